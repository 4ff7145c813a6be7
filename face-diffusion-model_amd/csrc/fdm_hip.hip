// libfdm_hip.so: C ABI (include/fdm_hip.h) over the gfx950 kernels.  No torch, no exceptions across
// the boundary, no allocation or synchronisation inside fdm_op_* (so a caller may capture them).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/fdm_hip.h"
#include "elementwise.hpp"
#include "kernels.hpp"

namespace fdm {
thread_local std::string g_err;
// records the message returned by fdm_last_error() on this thread and hands back `code` (shared with plan.hip)
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace fdm

namespace {
using fdm::fail;
using fdm::g_err;

int hip_fail(hipError_t e, const char* what) {
  return fail(FDM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

using Op = std::function<hipError_t(hipStream_t)>;

}  // namespace

struct fdm_prog {
  std::vector<Op> ops;
  hipGraph_t graph = nullptr;                // the captured sequence
  hipGraphExec_t exec = nullptr;
};

namespace {

thread_local fdm_prog* g_rec = nullptr;

// Either record the launch closure into the program being built or launch it now.
int submit(Op op, void* stream, const char* what) {
  if (g_rec) {
    g_rec->ops.push_back(std::move(op));
    return FDM_OK;
  }
  hipError_t e = op((hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, what);
  return FDM_OK;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int grid_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

const char* fdm_last_error(void) { return g_err.c_str(); }
int fdm_version(void) { return 105; }      // 1.05: round 5 (fdm_gemm_args.ksplit / batch2, fdm_ln_args.x_planes, plan keys ksplit.*; lanes, FDM_BF16X3 and three tile ids removed)

// sizeof() of a public struct as THIS build sees it: a binding compares it with its own mirror before the first call
int fdm_abi_struct_size(const char* name) {
  if (!name) return fdm::fail(FDM_ERR_ARG, "abi_struct_size: null name");
  const std::string n(name);
  if (n == "fdm_sched_args") return (int)sizeof(fdm_sched_args);
  if (n == "fdm_gemm_args") return (int)sizeof(fdm_gemm_args);
  if (n == "fdm_attn_args") return (int)sizeof(fdm_attn_args);
  if (n == "fdm_ln_args") return (int)sizeof(fdm_ln_args);
  if (n == "fdm_model_desc") return (int)sizeof(fdm_model_desc);
  if (n == "fdm_sample_args") return (int)sizeof(fdm_sample_args);
  if (n == "fdm_vq_desc") return (int)sizeof(fdm_vq_desc);
  return fdm::fail(FDM_ERR_ARG, "abi_struct_size: unknown struct '%s'", name);
}

int fdm_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

static bool gemm_act_heavy_host(int act) { return act == FDM_ACT_MISH || act == FDM_ACT_GELU_ERF || act == FDM_ACT_GELU_TANH; }

int fdm_op_gemm(const fdm_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W) return fail(FDM_ERR_ARG, "gemm: null operand");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return fail(FDM_ERR_SHAPE, "gemm: M,N,K must be positive (%d,%d,%d)", a->M, a->N, a->K);
  if (a->dtype < FDM_F32 || a->dtype > FDM_F16) return fail(FDM_ERR_ARG, "gemm: bad dtype %d", a->dtype);
  const bool split = a->dtype == FDM_F16X3;
  const int bk = a->dtype == FDM_F32 ? 32 : 64, epc = a->dtype == FDM_F32 ? 4 : 8;
  if (split && (a->a_lo_off <= 0 || a->w_lo_off <= 0 || a->a_lo_off % epc || a->w_lo_off % epc))
    return fail(FDM_ERR_ARG, "gemm: split operands need positive a_lo_off / w_lo_off (multiples of 8 elements)");
  if (split && a->out_t && a->out_t_lo_off <= 0) return fail(FDM_ERR_ARG, "gemm: split out_t needs out_t_lo_off");
  if (a->K % bk) return fail(FDM_ERR_SHAPE, "gemm: K=%d not a multiple of %d", a->K, bk);
  if (a->lda % epc || a->ldw % epc || !aligned16(a->A) || !aligned16(a->W)) return fail(FDM_ERR_ARG, "gemm: operands need 16-byte aligned rows");
  if (a->lda < 0 || a->ldw < 0 || a->lda > 0x7fffffffLL || a->ldw > 0x7fffffffLL) return fail(FDM_ERR_SHAPE, "gemm: lda / ldw are passed to the kernel as 32-bit element counts");
  if (a->a_batch_stride % epc || a->w_batch_stride % epc) return fail(FDM_ERR_ARG, "gemm: batch strides need 16-byte alignment");
  if (!a->out_f32 && !a->out_t && !a->out_vp && !a->out_kp) return fail(FDM_ERR_ARG, "gemm: no output");
  if (a->out_kp || a->out_vp) {
    const int hi = a->out_vp ? a->vp_col0 : a->N;
    if (a->kv_hd <= 0 || a->kv_hd % 16 || a->kv_L <= 0 || a->kv_Lpad < a->kv_L || a->kv_Lpad % 32 || a->M % a->kv_L)
      return fail(FDM_ERR_SHAPE, "gemm: bad packed K/V geometry (hd %d, L %d, Lpad %d)", a->kv_hd, a->kv_L, a->kv_Lpad);
    if (a->out_vp && (a->vp_col0 < 0 || a->vp_col0 % 4 || (a->N - a->vp_col0) % a->kv_hd)) return fail(FDM_ERR_SHAPE, "gemm: bad packed V column range");
    if (a->out_kp && (a->kp_col0 < 0 || a->kp_col0 % 4 || hi < a->kp_col0 || (hi - a->kp_col0) % a->kv_hd)) return fail(FDM_ERR_SHAPE, "gemm: bad packed K column range");
    if (!aligned16(a->out_kp) || !aligned16(a->out_vp)) return fail(FDM_ERR_ARG, "gemm: packed K/V buffers must be 16-byte aligned");
    if (a->dtype == FDM_F16X3 && (a->kv_lo_off <= 0 || a->kv_lo_off % 8)) return fail(FDM_ERR_ARG, "gemm: split packed K/V outputs need kv_lo_off");
  }
  if (a->bias && !aligned16(a->bias)) return fail(FDM_ERR_ARG, "gemm: bias must be 16-byte aligned");
  if (a->ln_stat_in && (a->ln_nparts <= 0 || a->ln_dim <= 0)) return fail(FDM_ERR_ARG, "gemm: ln_stat_in needs ln_nparts and ln_dim");
  if ((a->ln_colsum || a->rln_gamma) && !a->ln_stat_in) return fail(FDM_ERR_ARG, "gemm: LayerNorm folding needs ln_stat_in");
  if (a->rln_gamma && (!a->rln_beta || !a->resid)) return fail(FDM_ERR_ARG, "gemm: rln_gamma needs rln_beta and resid");
  if ((a->stat_out || a->ln_stat_in) && (a->batch > 1 || a->out_batch_stride)) return fail(FDM_ERR_ARG, "gemm: LayerNorm folding is not batched");
  if (a->tile < 0 || (a->tile & FDM_TILE_ID_MASK) > FDM_TILE_MAX) return fail(FDM_ERR_ARG, "gemm: unknown tile %d", a->tile);
  if (a->sched_fuse) {
    const fdm_sched_args& sc = a->sched;
    if (sc.mode != 0 && sc.mode != 1) return fail(FDM_ERR_ARG, "gemm: fused scheduler supports mode 0 (DDPM) and 1 (DDIM)");
    if (!a->resid || !a->out_f32 || a->N % 64 || a->ldo_f32 != a->N || a->ldr != a->N || a->resid_row_mod || a->batch > 1 ||
        !aligned16(a->resid) || !aligned16(a->out_f32) || (a->out_t && (a->ldo_t != a->N || !aligned16(a->out_t))))
      return fail(FDM_ERR_SHAPE, "gemm: fused scheduler needs resid = x_t and out_f32 = x_{t-1} as dense [M, N] arrays, N %% 64 == 0");
    if (gemm_act_heavy_host(a->act) || a->stat_out || a->out_kp || a->out_vp || a->rln_gamma)
      return fail(FDM_ERR_ARG, "gemm: fused scheduler cannot be combined with Mish/GELU, stat_out, packed K/V or rln outputs");
    if (sc.mode == 0 && (!sc.c1 || !sc.c2 || !sc.sigma)) return fail(FDM_ERR_ARG, "gemm: fused DDPM needs c1, c2, sigma");
    if (sc.mode == 1 && (!sc.sra || !sc.srm1 || !sc.sqrt_an || !sc.c_n)) return fail(FDM_ERR_ARG, "gemm: fused DDIM needs sra, srm1, sqrt_an, c_n");
    if (sc.mode == 0 && !sc.noise && sc.n_per_clip <= 0) return fail(FDM_ERR_ARG, "gemm: fused DDPM with Philox noise needs n_per_clip");
  }
  if (a->ksplit < 0 || a->ksplit > 4) return fail(FDM_ERR_ARG, "gemm: ksplit %d outside 0..4", a->ksplit);
  if (a->ksplit > 1) {
    const int tl = a->tile & FDM_TILE_ID_MASK;
    if (a->batch > 1 || a->out_batch_stride || a->act != FDM_ACT_NONE || !a->out_f32 || a->out_t || a->out_kp || a->out_vp || a->stat_out || a->ln_stat_in ||
        a->sched_fuse || a->resid_row_mod || (a->tile & FDM_TILE_GENERAL))
      return fail(FDM_ERR_ARG, "gemm: ksplit needs a plain launch (one batch, no activation, out_f32 as the only output, no folds)");
    if ((a->K / bk) % a->ksplit) return fail(FDM_ERR_SHAPE, "gemm: ksplit %d does not divide the %d k-tiles of K=%d", a->ksplit, a->K / bk, a->K);
    if (a->N % 64 || a->ldo_f32 % 4 || !aligned16(a->out_f32) || a->ksplit_stride % 4 || a->ksplit_stride < (long long)(a->M - 1) * a->ldo_f32 + a->N ||
        (a->resid && (a->ldr % 4 || !aligned16(a->resid))))
      return fail(FDM_ERR_SHAPE, "gemm: ksplit needs N %% 64 == 0, 16-byte aligned rows and ksplit_stride >= one output plane");
    if (tl != 0 && tl != FDM_TILE_64x64 && tl != FDM_TILE_64x64_S2 && tl != FDM_TILE_32x64_S3)
      return fail(FDM_ERR_ARG, "gemm: ksplit runs on the 64-column tiles (FDM_TILE_64x64, FDM_TILE_64x64_S2, FDM_TILE_32x64_S3), not tile %d", tl);
  }
  if (a->batch2 < 0) return fail(FDM_ERR_ARG, "gemm: negative batch2");
  if (a->batch2 >= 1) {
    const int tl = a->tile & FDM_TILE_ID_MASK;
    if (a->ksplit > 1 || a->out_kp || a->out_vp || a->stat_out || a->ln_stat_in || a->sched_fuse || a->resid_row_mod || a->incr_counter)
      return fail(FDM_ERR_ARG, "gemm: batch2 cannot be combined with ksplit, packed K/V, LayerNorm folds, the fused scheduler or resid_row_mod");
    if (a->a_batch_stride2 % epc) return fail(FDM_ERR_ARG, "gemm: a_batch_stride2 needs 16-byte alignment");
    if (tl != 0 && tl != FDM_TILE_64x64 && tl != FDM_TILE_64x64_S2 && tl != FDM_TILE_128x64)
      return fail(FDM_ERR_ARG, "gemm: batch2 runs on the 64-column tiles (FDM_TILE_64x64, FDM_TILE_64x64_S2, FDM_TILE_128x64), not tile %d", tl);
    if (a->batch < 1) return fail(FDM_ERR_ARG, "gemm: batch2 needs batch >= 1 (the group count of the second batch level)");
  }
  fdm_gemm_args c = *a;
  return submit([c](hipStream_t s) { return fdm::gemm_launch(c, s); }, stream, "gemm");
}

int fdm_gemm_heuristic_tile(const fdm_gemm_args* a) {
  if (!a) return fail(FDM_ERR_ARG, "gemm_heuristic_tile: null argument");
  return fdm::gemm_heuristic_tile_of(*a);
}

int fdm_op_attention(const fdm_attn_args* a, void* stream) {
  if (!a || !a->Q || !a->Kp || !a->Vp || !a->O) return fail(FDM_ERR_ARG, "attention: null operand");
  if (a->hd != 64 && a->hd != 128 && a->hd != 256) return fail(FDM_ERR_SHAPE, "attention: head_dim %d unsupported (64, 128, 256)", a->hd);
  if (a->B <= 0 || a->H <= 0 || a->L <= 0) return fail(FDM_ERR_SHAPE, "attention: B,H,L must be positive");
  if (a->Lpad < a->L || a->Lpad % 32) return fail(FDM_ERR_SHAPE, "attention: Lpad=%d must be a multiple of 32 >= L", a->Lpad);
  if (a->dtype < FDM_F32 || a->dtype > FDM_F16) return fail(FDM_ERR_ARG, "attention: bad dtype %d (FDM_F32, FDM_BF16, FDM_F16X3, FDM_F16)", a->dtype);
  if (a->o_split && (a->dtype != FDM_F32 || a->o_split != FDM_F16X3 || a->o_lo_off <= 0))
    return fail(FDM_ERR_ARG, "attention: o_split needs dtype FDM_F32, a split kind and o_lo_off");
  if (a->dtype == FDM_F16X3 && (a->q_lo_off <= 0 || a->kv_lo_off <= 0 || a->o_lo_off <= 0 || a->q_lo_off % 8 || a->kv_lo_off % 8 || a->o_lo_off % 4))
    return fail(FDM_ERR_ARG, "attention: split operands need q_lo_off, kv_lo_off and o_lo_off");
  if (a->q_lo_off > 0x7fffffffLL || a->kv_lo_off > 0x7fffffffLL || a->ldq > 0x7fffffffLL)
    return fail(FDM_ERR_SHAPE, "attention: q_lo_off, kv_lo_off and ldq are passed to the kernel as 32-bit element counts");
  const int epc = a->dtype == FDM_F32 ? 4 : 8;
  if (a->ldq % epc || a->ldo % 4 || !aligned16(a->Q) || !aligned16(a->Kp) || !aligned16(a->Vp) || !aligned16(a->O))
    return fail(FDM_ERR_ARG, "attention: operands need 16-byte aligned rows");
  if (a->slopes && a->period <= 0) return fail(FDM_ERR_ARG, "attention: period must be positive");
  fdm_attn_args c = *a;
  return submit([c](hipStream_t s) { return fdm::attn_launch(c, s); }, stream, "attention");
}

int fdm_op_pack_kv(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp,
                   int B, int H, int L, int Lpad, int hd, int dtype, void* stream) {
  if (!K || !V || !Kp || !Vp) return fail(FDM_ERR_ARG, "pack_kv: null operand");
  if (B <= 0 || H <= 0 || L <= 0 || Lpad < L || Lpad % 32 || hd <= 0 || hd % 16) return fail(FDM_ERR_SHAPE, "pack_kv: bad geometry");
  if (dtype != FDM_F32 && dtype != FDM_BF16) return fail(FDM_ERR_ARG, "pack_kv: bad dtype %d", dtype);
  return submit([=](hipStream_t s) {
    return dtype == FDM_BF16 ? fdm::pack_kv_launch_bf16(K, ldk, V, ldv, Kp, Vp, B, H, L, Lpad, hd, s)
                             : fdm::pack_kv_launch_f32(K, ldk, V, ldv, Kp, Vp, B, H, L, Lpad, hd, s);
  }, stream, "pack_kv");
}

int fdm_op_layernorm(const fdm_ln_args* a, void* stream) {
  if (!a || !a->x || !a->gamma || !a->beta) return fail(FDM_ERR_ARG, "layernorm: null operand");
  if (a->d != 256 && a->d != 512 && a->d != 768 && a->d != 1024) return fail(FDM_ERR_SHAPE, "layernorm: d=%d unsupported (256, 512, 768, 1024)", a->d);
  if (a->M <= 0) return fail(FDM_ERR_SHAPE, "layernorm: M must be positive");
  if (!a->y_f32 && !a->y_t) return fail(FDM_ERR_ARG, "layernorm: no output");
  if (a->dtype < FDM_F32 || a->dtype > FDM_F16) return fail(FDM_ERR_ARG, "layernorm: bad dtype %d", a->dtype);
  if (a->y_t && a->dtype == FDM_F16X3 && (a->y_t_lo_off <= 0 || a->y_t_lo_off % 4)) return fail(FDM_ERR_ARG, "layernorm: split y_t needs y_t_lo_off");
  if (a->add_mat_group < 0 || a->add_mat_wrap < 0 ||
      (a->add_mat_group > 0 && (a->add_mat_L <= 0 || a->add_mat_group % a->add_mat_L || (a->add_mat_wrap > 0 && a->add_mat_wrap % a->add_mat_group))))
    return fail(FDM_ERR_SHAPE, "layernorm: shared add_mat needs add_mat_L | add_mat_group | add_mat_wrap (got %d, %d, %d)", a->add_mat_L, a->add_mat_group, a->add_mat_wrap);
  if (a->x_planes < 0 || a->x_planes > 4 || (a->x_planes > 1 && (a->x_plane_stride < (long long)a->M * a->d || a->x_plane_stride % 4)))
    return fail(FDM_ERR_ARG, "layernorm: x_planes %d (0..4) needs x_plane_stride >= M * d, a multiple of 4", a->x_planes);
  fdm_ln_args c = *a;
  return submit([c](hipStream_t s) {
    switch (c.dtype) {
      case FDM_BF16: return fdm::ln_launch_t<fdm::bf16>(c, s);
      case FDM_F16X3: return fdm::ln_launch_t<fdm::f16x3_t>(c, s);
      case FDM_F16: return fdm::ln_launch_t<fdm::f16>(c, s);
      default: return fdm::ln_launch_t<float>(c, s);
    }
  }, stream, "layernorm");
}

int fdm_op_sched_step(const fdm_sched_args* a, void* stream) {
  if (!a || !a->x0 || !a->x_out) return fail(FDM_ERR_ARG, "sched: null operand");
  if (a->n <= 0 || a->n % 4) return fail(FDM_ERR_SHAPE, "sched: n=%lld must be a positive multiple of 4", a->n);
  if (a->mode == 0 && (!a->x || !a->c1 || !a->c2 || !a->sigma)) return fail(FDM_ERR_ARG, "sched: DDPM tables missing");
  if (a->mode == 1 && (!a->x || !a->sra || !a->srm1 || !a->sqrt_an || !a->c_n)) return fail(FDM_ERR_ARG, "sched: DDIM tables missing");
  if (a->mode < 0 || a->mode > 2) return fail(FDM_ERR_ARG, "sched: bad mode %d", a->mode);
  if (a->mode == 0 && !a->noise && (a->n_per_clip <= 0 || a->n_per_clip % 4)) return fail(FDM_ERR_SHAPE, "sched: n_per_clip must be a positive multiple of 4");
  if (a->x_out_t && (a->out_dtype < FDM_F32 || a->out_dtype > FDM_F16)) return fail(FDM_ERR_ARG, "sched: bad out_dtype %d", a->out_dtype);
  if (a->x_out_t && a->out_dtype == FDM_F16X3 && (a->x_out_t_lo_off <= 0 || a->x_out_t_lo_off % 4)) return fail(FDM_ERR_ARG, "sched: split x_out_t needs x_out_t_lo_off");
  fdm_sched_args c = *a;
  return submit([c](hipStream_t s) { return fdm::sched_launch(c, s); }, stream, "sched");
}

int fdm_op_cast(const float* src, void* dst, long long n, int dtype, void* stream) {
  if (!src || !dst || n <= 0) return fail(FDM_ERR_ARG, "cast: bad argument");
  if (dtype < FDM_F32 || dtype > FDM_F16) return fail(FDM_ERR_ARG, "cast: bad dtype %d", dtype);
  return submit([=](hipStream_t s) {
    const dim3 g(grid_for(n)), b(256);
    if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::cast_kernel<fdm::bf16>), g, b, 0, s, src, (fdm::bf16*)dst, n);
    else if (dtype == FDM_F16X3) hipLaunchKernelGGL((fdm::cast_kernel<fdm::f16x3_t>), g, b, 0, s, src, (fdm::f16*)dst, n);
    else if (dtype == FDM_F16) hipLaunchKernelGGL((fdm::cast_kernel<fdm::f16>), g, b, 0, s, src, (fdm::f16*)dst, n);
    else hipLaunchKernelGGL((fdm::cast_kernel<float>), g, b, 0, s, src, (float*)dst, n);
    return hipGetLastError();
  }, stream, "cast");
}

int fdm_op_bias_act(const float* in, const float* vec, float* out, long long rows, int d, int act, void* stream) {
  if (!in || !out || rows <= 0 || d <= 0) return fail(FDM_ERR_ARG, "bias_act: bad argument");
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::bias_act_kernel, dim3(grid_for(rows * d)), dim3(256), 0, s, in, vec, out, rows, d, act);
    return hipGetLastError();
  }, stream, "bias_act");
}

int fdm_op_add_rows(const float* a, int a_div, int a_mod, const float* b, int b_div, int b_mod,
                    const float* c, int c_div, int c_mod, float* out, long long M, int d, void* stream) {
  if (!a || !out || M <= 0 || d <= 0 || a_div <= 0 || a_mod <= 0) return fail(FDM_ERR_ARG, "add_rows: bad argument");
  if ((b && (b_div <= 0 || b_mod <= 0)) || (c && (c_div <= 0 || c_mod <= 0))) return fail(FDM_ERR_ARG, "add_rows: bad div/mod");
  fdm::AddRowsArgs p{a, a_div, a_mod, b, b ? b_div : 1, b ? b_mod : 1, c, c ? c_div : 1, c ? c_mod : 1, out, M, d};
  return submit([p](hipStream_t s) {
    hipLaunchKernelGGL(fdm::add_rows_kernel, dim3(grid_for(p.M * p.d)), dim3(256), 0, s, p);
    return hipGetLastError();
  }, stream, "add_rows");
}

int fdm_op_small_linear(const float* x, const float* W, const float* bias, float* out, int B, int K, int d, int act, void* stream) {
  if (!x || !W || !out || B <= 0 || K <= 0 || d <= 0) return fail(FDM_ERR_ARG, "small_linear: bad argument");
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::small_linear_kernel, dim3((B * d + 255) / 256), dim3(256), 0, s, x, W, bias, out, B, K, d, act);
    return hipGetLastError();
  }, stream, "small_linear");
}

int fdm_op_pad_rows(const void* in, void* out, int B, int L, int d, int pad, int dtype, int zero, void* stream) {
  if (!in || !out || B <= 0 || L <= 0 || d <= 0 || pad < 0) return fail(FDM_ERR_ARG, "pad_rows: bad argument");
  const long long n = (long long)B * (L + 2 * pad) * d;
  return submit([=](hipStream_t s) {
    if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::pad_rows_kernel<fdm::bf16>), dim3(grid_for(n)), dim3(256), 0, s, (const fdm::bf16*)in, (fdm::bf16*)out, B, L, d, pad, zero);
    else hipLaunchKernelGGL((fdm::pad_rows_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (const float*)in, (float*)out, B, L, d, pad, zero);
    return hipGetLastError();
  }, stream, "pad_rows");
}

int fdm_op_group_pad(const void* in, void* out, int B, int T, int d, int groups, int pad, int dtype, void* stream) {
  if (!in || !out || B <= 0 || T <= 0 || d <= 0 || groups <= 0 || d % groups || pad < 0) return fail(FDM_ERR_ARG, "group_pad: bad argument");
  const long long n = (long long)B * (T + 2 * pad) * d;
  return submit([=](hipStream_t s) {
    if (dtype == FDM_BF16 || dtype == FDM_F16X3)      // (one plane of a split operand moves like any 2-byte matrix: the caller passes each plane)
      hipLaunchKernelGGL((fdm::group_pad_kernel<fdm::bf16>), dim3(grid_for(n)), dim3(256), 0, s, (const fdm::bf16*)in, (fdm::bf16*)out, B, T, d, groups, pad);
    else hipLaunchKernelGGL((fdm::group_pad_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (const float*)in, (float*)out, B, T, d, groups, pad);
    return hipGetLastError();
  }, stream, "group_pad");
}

int fdm_op_conv0(const float* wav, const float* w, const float* bias, float* out, int B, int n, int T0, void* stream) {
  if (!wav || !w || !out || B <= 0 || n < 10 || T0 != (n - 10) / 5 + 1) return fail(FDM_ERR_SHAPE, "conv0: bad shape (n=%d, T0=%d)", n, T0);
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::conv0_kernel, dim3((T0 + 15) / 16, B), dim3(256), 0, s, wav, w, bias, out, n, T0);
    return hipGetLastError();
  }, stream, "conv0");
}

int fdm_op_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out,
                         long long out_lo_off, int B, int n, int T0, float eps, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || B <= 0 || n < 10 || T0 != (n - 10) / 5 + 1) return fail(FDM_ERR_SHAPE, "conv0_ln_gelu: bad shape (n=%d, T0=%d)", n, T0);
  if (dtype != FDM_F32 && dtype != FDM_BF16 && dtype != FDM_F16X3) return fail(FDM_ERR_ARG, "conv0_ln_gelu: dtype %d (fp32, bf16 or FDM_F16X3 output)", dtype);
  if (dtype == FDM_F16X3 && out_lo_off <= 0) return fail(FDM_ERR_ARG, "conv0_ln_gelu: split output needs out_lo_off");
  return submit([=](hipStream_t s) {
    const dim3 grid((T0 + 31) / 32, B);       // 4 waves x 8 frames per workgroup
    if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::conv0_ln_gelu_kernel<fdm::bf16>), grid, dim3(256), 0, s, wav, w, bias, gamma, beta, (fdm::bf16*)out, 0LL, n, T0, eps);
    else if (dtype == FDM_F16X3) hipLaunchKernelGGL((fdm::conv0_ln_gelu_kernel<fdm::f16x3_t>), grid, dim3(256), 0, s, wav, w, bias, gamma, beta, (fdm::f16*)out, out_lo_off, n, T0, eps);
    else hipLaunchKernelGGL((fdm::conv0_ln_gelu_kernel<float>), grid, dim3(256), 0, s, wav, w, bias, gamma, beta, (float*)out, 0LL, n, T0, eps);
    return hipGetLastError();
  }, stream, "conv0_ln_gelu");
}

int fdm_op_leaky_instnorm(const float* x, float* y_f32, void* y_t, int B, int L, int d, float eps, int dtype, void* stream) {
  if (!x || (!y_f32 && !y_t) || B <= 0 || L <= 0 || d <= 0) return fail(FDM_ERR_ARG, "leaky_instnorm: bad argument");
  return submit([=](hipStream_t s) {
    dim3 grid((d + 63) / 64, B);
    if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::leaky_instnorm_kernel<fdm::bf16>), grid, dim3(1024), 0, s, x, y_f32, (fdm::bf16*)y_t, L, d, eps);
    else hipLaunchKernelGGL((fdm::leaky_instnorm_kernel<float>), grid, dim3(1024), 0, s, x, y_f32, (float*)y_t, L, d, eps);
    return hipGetLastError();
  }, stream, "leaky_instnorm");
}

int fdm_op_time_groupnorm(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_t, long long y_t_lo_off, int B, int T, int C,
                          float eps, int act, int dtype, void* scratch, long long scratch_bytes, void* stream) {
  if (!x || (!y_f32 && !y_t) || B <= 0 || T <= 0 || C <= 0) return fail(FDM_ERR_ARG, "time_groupnorm: bad argument");
  if (y_t && dtype == FDM_F16X3 && y_t_lo_off <= 0) return fail(FDM_ERR_ARG, "time_groupnorm: split y_t needs y_t_lo_off");
  // long clips with a scratch buffer: statistics and normalisation over time chunks (two launches, hundreds of workgroups)
  const int nch = T >= 4096 ? std::min(64, (T + 1023) / 1024) : 1;
  const int chunk = ((T + nch - 1) / nch + 15) / 16 * 16;
  const long long need = (long long)B * nch * C * 2 * (long long)sizeof(double);
  const long long lo = y_t_lo_off;
  if (scratch && nch > 1 && scratch_bytes >= need && ((uintptr_t)scratch % 8) == 0) {
    double* part = (double*)scratch;
    return submit([=](hipStream_t s) {
      const dim3 grid((C + 63) / 64, nch, B);
      hipLaunchKernelGGL(fdm::time_stats_kernel, grid, dim3(1024), 0, s, x, part, T, C, chunk);
      if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::time_norm_apply_kernel<fdm::bf16>), grid, dim3(1024), 0, s, x, (const double*)part, gamma, beta, y_f32, (fdm::bf16*)y_t, 0LL, T, C, chunk, eps, act);
      else if (dtype == FDM_F16X3) hipLaunchKernelGGL((fdm::time_norm_apply_kernel<fdm::f16x3_t>), grid, dim3(1024), 0, s, x, (const double*)part, gamma, beta, y_f32, (fdm::f16*)y_t, lo, T, C, chunk, eps, act);
      else hipLaunchKernelGGL((fdm::time_norm_apply_kernel<float>), grid, dim3(1024), 0, s, x, (const double*)part, gamma, beta, y_f32, (float*)y_t, 0LL, T, C, chunk, eps, act);
      return hipGetLastError();
    }, stream, "time_groupnorm");
  }
  return submit([=](hipStream_t s) {
    dim3 grid((C + 63) / 64, B);
    if (dtype == FDM_BF16) hipLaunchKernelGGL((fdm::time_groupnorm_kernel<fdm::bf16>), grid, dim3(1024), 0, s, x, gamma, beta, y_f32, (fdm::bf16*)y_t, 0LL, T, C, eps, act);
    else if (dtype == FDM_F16X3) hipLaunchKernelGGL((fdm::time_groupnorm_kernel<fdm::f16x3_t>), grid, dim3(1024), 0, s, x, gamma, beta, y_f32, (fdm::f16*)y_t, lo, T, C, eps, act);
    else hipLaunchKernelGGL((fdm::time_groupnorm_kernel<float>), grid, dim3(1024), 0, s, x, gamma, beta, y_f32, (float*)y_t, 0LL, T, C, eps, act);
    return hipGetLastError();
  }, stream, "time_groupnorm");
}

int fdm_op_mean_diff(const float* a, const float* b, float* partial, float* out, long long n, int l1, void* stream) {
  if (!a || !b || !partial || !out || n <= 0) return fail(FDM_ERR_ARG, "mean_diff: bad argument");
  return submit([=](hipStream_t s) {
    const int nb = grid_for(n) > 1024 ? 1024 : grid_for(n);
    hipLaunchKernelGGL(fdm::diff_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, partial, n, l1);
    hipLaunchKernelGGL(fdm::diff_final_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, nb, 1.0f / (float)n, out);
    return hipGetLastError();
  }, stream, "mean_diff");
}

int fdm_op_vertex_err(const float* gt, const float* pred, const int* region, int R, int F, int V,
                      float* frame_max, double* frame_sum, double* out, void* stream) {
  if (!gt || !pred || !frame_max || !frame_sum || !out) return fail(FDM_ERR_ARG, "vertex_err: null operand");
  if (F <= 0 || V <= 0 || R <= 0 || (!region && R != V)) return fail(FDM_ERR_SHAPE, "vertex_err: bad shape (F %d, V %d, R %d)", F, V, R);
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::vertex_err_kernel, dim3(F), dim3(256), 0, s, gt, pred, region, R, V, frame_max, frame_sum);
    hipLaunchKernelGGL(fdm::vertex_err_final_kernel, dim3(1), dim3(256), 0, s, (const float*)frame_max, (const double*)frame_sum, F, R, out);
    return hipGetLastError();
  }, stream, "vertex_err");
}

int fdm_op_motion_std(const float* verts, const float* tmpl, const int* region, int R, int F, int V,
                      double* partial, double* out, void* stream) {
  if (!verts || !tmpl || !partial || !out) return fail(FDM_ERR_ARG, "motion_std: null operand");
  if (F <= 0 || V <= 0 || R <= 0 || (!region && R != V)) return fail(FDM_ERR_SHAPE, "motion_std: bad shape (F %d, V %d, R %d)", F, V, R);
  return submit([=](hipStream_t s) {
    const int FC = F < 64 ? F : 64;
    hipLaunchKernelGGL(fdm::motion_partial_kernel, dim3((R + 255) / 256, FC), dim3(256), 0, s, verts, tmpl, region, R, F, V, partial);
    hipLaunchKernelGGL(fdm::motion_final_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, FC, R, F, out);
    return hipGetLastError();
  }, stream, "motion_std");
}

int fdm_op_linear_interp(const float* x, float* y, int B, int Tin, int Tout, int C, void* stream) {
  if (!x || !y || B <= 0 || Tin <= 0 || Tout <= 0 || C <= 0) return fail(FDM_ERR_ARG, "linear_interp: bad argument");
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::linear_interp_kernel, dim3(grid_for((long long)B * Tout * C)), dim3(256), 0, s, x, y, B, Tin, Tout, C);
    return hipGetLastError();
  }, stream, "linear_interp");
}

int fdm_op_adain(const float* content, const float* style, float* out, int NC, int Lc, int Ls, float eps, void* stream) {
  if (!content || !style || !out || NC <= 0 || Lc < 2 || Ls < 2) return fail(FDM_ERR_ARG, "adain: bad argument");
  return submit([=](hipStream_t s) {
    hipLaunchKernelGGL(fdm::adain_kernel, dim3((NC + 3) / 4), dim3(256), 0, s, content, style, out, NC, Lc, Ls, eps);
    return hipGetLastError();
  }, stream, "adain");
}

int fdm_op_vq_quant(const float* z, const float* codebook, const int* book, int B, int R, int c, int K,
                    float* zq_bcl, long long* idx, void* stream) {
  if (!z || !codebook || !zq_bcl || !idx) return fail(FDM_ERR_ARG, "vq_quant: null operand");
  if (B <= 0 || R <= 0 || K <= 0 || c <= 0 || c > 128) return fail(FDM_ERR_SHAPE, "vq_quant: bad shape (c=%d must be <= 128)", c);
  return submit([=](hipStream_t s) {
    const long long rows = (long long)B * R;
    hipLaunchKernelGGL(fdm::vq_quant_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, z, codebook, book, B, R, c, K, zq_bcl, idx);
    return hipGetLastError();
  }, stream, "vq_quant");
}

int fdm_op_vq_stats(const float* z, const float* codebook, const int* book, const long long* idx, int B, int R, int c, int K, float beta,
                    float* min_encodings, double* partial, int* hist, float* out, void* stream) {
  if (!z || !codebook || !idx || !partial || !hist || !out) return fail(FDM_ERR_ARG, "vq_stats: null operand");
  if (B <= 0 || R <= 0 || K <= 0 || c <= 0) return fail(FDM_ERR_SHAPE, "vq_stats: bad shape");
  const long long rows = (long long)B * R;
  const int nblk = (int)std::min<long long>(1024, (rows + 3) / 4);
  return submit([=](hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)K * sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fdm::vq_stats_partial_kernel, dim3(nblk), dim3(256), 0, s, z, codebook, book, idx, B, R, c, K, min_encodings, partial, hist);
    hipLaunchKernelGGL(fdm::vq_stats_final_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, nblk, (const int*)hist, K, rows, c, beta, out);
    return hipGetLastError();
  }, stream, "vq_stats");
}

// ---------------------------------------------------------------------------------------------
// step programs
// ---------------------------------------------------------------------------------------------
int fdm_prog_create(fdm_prog** out) {
  if (!out) return fail(FDM_ERR_ARG, "prog_create: null out");
  *out = new (std::nothrow) fdm_prog();
  return *out ? FDM_OK : fail(FDM_ERR_STATE, "prog_create: out of memory");
}

int fdm_prog_destroy(fdm_prog* p) {
  if (!p) return FDM_OK;
  if (g_rec == p) g_rec = nullptr;
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  delete p;
  return FDM_OK;
}

int fdm_prog_begin(fdm_prog* p) {
  if (!p) return fail(FDM_ERR_ARG, "prog_begin: null program");
  if (g_rec) return fail(FDM_ERR_STATE, "prog_begin: another program is recording on this thread");
  if (p->exec) return fail(FDM_ERR_STATE, "prog_begin: program already instantiated");
  g_rec = p;
  return FDM_OK;
}

int fdm_prog_end(fdm_prog* p) {
  if (!p || g_rec != p) return fail(FDM_ERR_STATE, "prog_end: program is not recording");
  g_rec = nullptr;
  return FDM_OK;
}

int fdm_prog_num_ops(fdm_prog* p) { return p ? (int)p->ops.size() : 0; }

int fdm_prog_run(fdm_prog* p, void* stream) {
  if (!p) return fail(FDM_ERR_ARG, "prog_run: null program");
  if (g_rec) return fail(FDM_ERR_STATE, "prog_run: a program is still recording");
  for (auto& op : p->ops) {
    hipError_t e = op((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "prog_run");
  }
  return FDM_OK;
}

int fdm_prog_instantiate(fdm_prog* p, void* stream) {
  if (!p) return fail(FDM_ERR_ARG, "prog_instantiate: null program");
  if (g_rec) return fail(FDM_ERR_STATE, "prog_instantiate: a program is still recording");
  if (p->exec) return FDM_OK;
  if (p->ops.empty()) return fail(FDM_ERR_STATE, "prog_instantiate: empty program");
  (void)stream;
  // Capture never executes anything, so it runs on a stream of the library's own: the caller's stream may be the legacy
  // default stream (torch's current stream usually is), which cannot be captured.  (Independent chains as parallel branches
  // of one graph, or as graphs on separate streams, were measured slower than one chain over all rows -- rounds 1 and 3,
  // profiles/README.md -- and are no longer offered.)
  hipStream_t cap = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
  if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
  struct CapGuard { hipStream_t s; ~CapGuard() { if (s) (void)hipStreamDestroy(s); } } guard{cap};
  e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return hip_fail(e, "hipStreamBeginCapture");
  hipError_t opErr = hipSuccess;
  for (size_t i = 0; opErr == hipSuccess && i < p->ops.size(); ++i) opErr = p->ops[i](cap);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(cap, &g);
  // a failure part-way leaves no half-built state behind: a later instantiate starts over, replay keeps failing loudly
  if (opErr != hipSuccess) { if (g) (void)hipGraphDestroy(g); return hip_fail(opErr, "prog_instantiate (launch during capture)"); }
  if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(e, "hipGraphInstantiate"); }
  p->graph = g;
  p->exec = x;
  return FDM_OK;
}

int fdm_prog_replay(fdm_prog* p, int n, void* stream) {
  if (!p || !p->exec) return fail(FDM_ERR_STATE, "prog_replay: program not instantiated");
  for (int i = 0; i < n; ++i) {
    const hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch");
  }
  return FDM_OK;
}

}  // extern "C"
