// HBM-bound kernels of the path: LayerNorm (wave per row), the fused scheduler update with CFG mix
// and in-kernel Philox noise, VQ nearest-code search, InstanceNorm, conv0, and layout helpers.
// All of these are bandwidth kernels: 16-byte accesses per lane, no re-reads, grid-stride where large.
#pragma once
#include "common.hpp"
#include "sched.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

// ------------------------------------------------------------------------------------------------
// LayerNorm (+addends, +fused second LayerNorm, +activation): one workgroup of NV wavefronts per row (d = 256 * NV), one
// float4 per thread, so every load of the row and of its addends is in flight at once; the statistics go wave_sum -> LDS
// slot -> fixed-order sum (one barrier each, a slot per statistic: no reuse hazard).  One kernel for every row count, so
// a clip's result never depends on the batch it is computed in.  (Round 1's wave-per-row form took 7.3 us at the step's
// 800 rows against 5.4 us here; HEAVY = the activation may be a transcendental one -- GELU after HuBERT's conv LayerNorm --
// and is templated out of the step's instances: the inlined libm forms were 90 % of that kernel's code.)
// ------------------------------------------------------------------------------------------------
template <typename T, int NV, bool HEAVY>
__global__ __launch_bounds__(64 * NV) void ln_row_kernel(const float* p_x, const float* p_add_mat, const float* p_add_tab, const int* p_tab_step,
                                                         const int* p_tab_index, int p_add_mat_group, int p_add_mat_wrap, int p_add_mat_L,
                                                         const fdm_ln_args p) {
  // (the leading arguments repeat fields of p: kernel-argument preload, 13 SGPRs -- what the row, the matrix addend and the table row's
  //  dependent loads need goes out without waiting for the argument block; see gemm_glds_kernel)
  __shared__ float red[4][NV];
  constexpr int d = 256 * NV;
#ifdef FDM_GEMM_STAMPS
  // instrumented build (tools/coldstart_probe.cpp): with x_planes == 0 the stamp buffer rides in x_plane_stride (8 words per row)
  const unsigned long long t_first = wall_clock64();
  unsigned long long t_loaded = 0, t_reduced = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x, col = tid * 4;
  const bool two = p.gamma2 != nullptr;
  // every load of the kernel is requested before the first value is used: the row, the matrix addend and the affine vectors
  // go out at once, the table row one dependent scalar load (the device-side step word) later -- one memory latency in
  // all instead of one per operand (the kernel is launch-to-launch latency, not bandwidth)
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 v = *(const f32x4*)(p_x + (size_t)row * d + col);
  // split-K partial planes of the row (fdm_gemm_args.ksplit): requested with everything else, summed in plane order below
  const int npl = p.x_planes;
  f32x4 vpl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) vpl[i] = (i + 1 < npl) ? *(const f32x4*)(p_x + (size_t)(i + 1) * p.x_plane_stride + (size_t)row * d + col) : zero;
  const bool has_e = p_add_mat || p_add_tab;
  int arow = row;                      // (uniform: scalar arithmetic) conditions of a clip share the clip's addend rows
  if (p_add_mat_group > 0) {
    const int m = p_add_mat_wrap > 0 ? row % p_add_mat_wrap : row;
    arow = (m / p_add_mat_group) * p_add_mat_L + m % p_add_mat_L;
  }
  const f32x4 em = p_add_mat ? *(const f32x4*)(p_add_mat + (size_t)arow * d + col) : zero;
  const f32x4 g1 = *(const f32x4*)(p.gamma + col), b1 = *(const f32x4*)(p.beta + col);
  const float *gp2 = two ? p.gamma2 : p.gamma, *bp2 = two ? p.beta2 : p.beta;      // (a select of pointers, not of loaded data)
  const f32x4 g2 = *(const f32x4*)(gp2 + col), b2 = *(const f32x4*)(bp2 + col);
  f32x4 et = zero;
  if (p_add_tab) {
    const int k = p_tab_step ? *p_tab_step : 0;
    const int idx = p_tab_index ? p_tab_index[k] : k;
    et = *(const f32x4*)(p_add_tab + (size_t)idx * d + col);
  }
  const f32x4 e = p_add_tab ? em + et : em;
  auto block_sum = [&](float x, int slot) {
    x = wave_sum(x);
    if (lane == 0) red[slot][wave] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NV; ++w) t += red[slot][w];
    return t;
  };
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i + 1 < npl) v += vpl[i];
  if (!two && has_e) v += e;
#ifdef FDM_GEMM_STAMPS
  asm volatile("" ::"v"(v[0]), "v"(e[0]), "v"(g2[0]), "v"(b2[0]));
  __builtin_amdgcn_sched_barrier(0);
  t_loaded = wall_clock64();
  __builtin_amdgcn_sched_barrier(0);
#endif
  float mean = block_sum((v[0] + v[1]) + (v[2] + v[3]), 0) * (1.f / d);
  v -= mean;
  float var = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), 1) * (1.f / d);
  float rstd = 1.f / sqrtf(var + p.eps);
  if (two) {       // h = LN1(x); stage 2 input = h + add_mat + add_tab[idx]
    v = v * rstd * g1 + b1;
    if (has_e) v += e;
    mean = block_sum((v[0] + v[1]) + (v[2] + v[3]), 2) * (1.f / d);
    v -= mean;
    var = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), 3) * (1.f / d);
    rstd = 1.f / sqrtf(var + p.eps);
  }
  f32x4 y = v * rstd * g2 + b2;
#ifdef FDM_GEMM_STAMPS
  asm volatile("" ::"v"(y[0]));
  __builtin_amdgcn_sched_barrier(0);
  t_reduced = wall_clock64();
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (HEAVY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = act_apply_t<T>(y[j], p.act);
  } else if (p.act == ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
  }
  if (p.y_f32) st16(p.y_f32 + (size_t)row * d + col, y);
  if (p.y_t) store_opnd4<T>((typename Opnd<T>::E*)p.y_t + (size_t)row * d + col, p.y_t_lo_off, y);
#ifdef FDM_GEMM_STAMPS
  if (p.x_planes == 0 && p.x_plane_stride) {
    const unsigned long long t_issued = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned long long* st = (unsigned long long*)p.x_plane_stride + (size_t)row * 8;
      st[5] = t_first; st[0] = t_first; st[1] = t_loaded; st[2] = t_reduced; st[3] = t_issued; st[4] = wall_clock64();
    }
  }
#endif
}

#define LN_PRELOAD_ARGS a.x, a.add_mat, a.add_tab, a.tab_step, a.tab_index, a.add_mat_group, a.add_mat_wrap, a.add_mat_L
template <typename T, bool HEAVY>
static hipError_t ln_launch_h(const fdm_ln_args& a, hipStream_t s) {
  dim3 grid(a.M);
  switch (a.d) {
    case 256: hipLaunchKernelGGL((ln_row_kernel<T, 1, HEAVY>), grid, dim3(64), 0, s, LN_PRELOAD_ARGS, a); break;
    case 512: hipLaunchKernelGGL((ln_row_kernel<T, 2, HEAVY>), grid, dim3(128), 0, s, LN_PRELOAD_ARGS, a); break;
    case 768: hipLaunchKernelGGL((ln_row_kernel<T, 3, HEAVY>), grid, dim3(192), 0, s, LN_PRELOAD_ARGS, a); break;
    case 1024: hipLaunchKernelGGL((ln_row_kernel<T, 4, HEAVY>), grid, dim3(256), 0, s, LN_PRELOAD_ARGS, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
template <typename T>
static hipError_t ln_launch_t(const fdm_ln_args& a, hipStream_t s) {
  return (a.act != ACT_NONE && a.act != ACT_RELU) ? ln_launch_h<T, true>(a, s) : ln_launch_h<T, false>(a, s);
}

__global__ __launch_bounds__(256) void sched_kernel(const fdm_sched_args p) {
  const SchedCoef c = sched_coef_load(p);
  const int k = c.k;
  if (p.arrive && p.advance && threadIdx.x == 0) {
    // every block reads *step first, then takes a ticket; the block holding the last ticket knows all
    // reads are done and advances the counter (and re-arms the ticket word) -- no separate launch
    const unsigned tk = atomicAdd(p.arrive, 1u);
    if (tk == gridDim.x - 1) {
      *p.arrive = 0u;
      *p.step = k + 1;
    }
  }
  const long long nq = p.n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long long)gridDim.x * blockDim.x) {
    f32x4 x0 = *(const f32x4*)(p.x0 + 4 * i);
    if (p.x0u) {
      f32x4 u = *(const f32x4*)(p.x0u + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) x0[j] = __fadd_rn(u[j], __fmul_rn(p.cfg_scale, __fsub_rn(x0[j], u[j])));
    }
    f32x4 o;
    if (p.mode == 2) {
      o = x0;
    } else {
      const f32x4 x = *(const f32x4*)(p.x + 4 * i);
      o = sched_update4(p, c, x0, x, 4 * i);
    }
    *(f32x4*)(p.x_out + 4 * i) = o;
    if (p.x_out_t) {
      if (p.out_dtype == FDM_BF16) store_opnd4<bf16>((bf16*)p.x_out_t + 4 * i, 0, o);
      else if (p.out_dtype == FDM_F16X3) store_opnd4<f16x3_t>((f16*)p.x_out_t + 4 * i, p.x_out_t_lo_off, o);
      else if (p.out_dtype == FDM_F16) store_opnd4<f16>((f16*)p.x_out_t + 4 * i, 0, o);
      else *(f32x4*)((float*)p.x_out_t + 4 * i) = o;
    }
  }
}

// single-thread epilogue kernel: advances the device-side step counter (separate tiny launch keeps
// the main kernel free of any inter-workgroup ordering requirement)
__global__ void step_advance_kernel(int* step) { *step += 1; }

static hipError_t sched_launch(const fdm_sched_args& a, hipStream_t s) {
  const long long nq = a.n / 4;
  int blocks = (int)((nq + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sched_kernel, dim3(blocks), dim3(256), 0, s, a);
  if (a.advance && a.step && !a.arrive) hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, a.step);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// small layout / elementwise helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* src, typename Opnd<T>::E* dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    store_opnd1<T>(dst + i, n, src[i]);      // split kinds: lo plane n elements after the hi plane
}

__global__ void bias_act_kernel(const float* in, const float* vec, float* out, long long rows, int d, int act) {
  const long long n = rows * d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = act_apply(in[i] + (vec ? vec[i % d] : 0.f), act);
}

struct AddRowsArgs {
  const float* a; int a_div, a_mod;
  const float* b; int b_div, b_mod;
  const float* c; int c_div, c_mod;
  float* out; long long M; int d;
};
__global__ void add_rows_kernel(const AddRowsArgs p) {
  const long long n = p.M * p.d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / p.d;
    const int col = (int)(i - m * p.d);
    float v = p.a[((m / p.a_div) % p.a_mod) * p.d + col];
    if (p.b) v += p.b[((m / p.b_div) % p.b_mod) * p.d + col];
    if (p.c) v += p.c[((m / p.c_div) % p.c_mod) * p.d + col];
    p.out[i] = v;
  }
}

// out[b, k, :] = in[b, clamp(k - pad), :] (replicate) or 0 outside (zero)
template <typename T>
__global__ void pad_rows_kernel(const T* in, T* out, int B, int L, int d, int pad, int zero) {
  const long long n = (long long)B * (L + 2 * pad) * d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % d);
    const long long r = i / d;
    const int k = (int)(r % (L + 2 * pad));
    const int b = (int)(r / (L + 2 * pad));
    int src = k - pad;
    T v;
    if (zero && (src < 0 || src >= L)) {
      v = from_f32<T>(0.f);
    } else {
      src = src < 0 ? 0 : (src >= L ? L - 1 : src);
      v = in[((size_t)b * L + src) * d + col];
    }
    out[i] = v;
  }
}

// [B, T, d] -> [groups, B, T + 2*pad, d/groups], zero padded in time
template <typename T>
__global__ void group_pad_kernel(const T* in, T* out, int B, int Tn, int d, int groups, int pad) {
  const int dg = d / groups;
  const long long n = (long long)groups * B * (Tn + 2 * pad) * dg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % dg);
    long long r = i / dg;
    const int k = (int)(r % (Tn + 2 * pad));
    r /= (Tn + 2 * pad);
    const int b = (int)(r % B);
    const int gi = (int)(r / B);
    const int src = k - pad;
    T v = from_f32<T>(0.f);
    if (src >= 0 && src < Tn) v = in[((size_t)b * Tn + src) * d + gi * dg + col];
    out[i] = v;
  }
}

// small dense layer for one-hot / conditioning vectors: out[b, j] = act(bias[j] + sum_k W[j, k] x[b, k])
__global__ void small_linear_kernel(const float* x, const float* W, const float* bias, float* out, int B, int K, int d, int act) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * d) return;
  const int b = i / d, j = i - b * d;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(W[(size_t)j * K + k], x[(size_t)b * K + k], acc);
  out[i] = act_apply(acc + (bias ? bias[j] : 0.f), act);
}

// HuBERT feature-extractor layer 0: Conv1d(1, 512, k=10, stride=5) + bias, channels-last output
__global__ __launch_bounds__(256) void conv0_kernel(const float* wav, const float* w, const float* bias, float* out,
                                                    int n, int T0) {
  constexpr int TT = 16;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const float* wv = wav + (size_t)b * n;
  float wa[10], wb[10];
  const int oa = threadIdx.x, ob = threadIdx.x + 256;
#pragma unroll
  for (int k = 0; k < 10; ++k) { wa[k] = w[oa * 10 + k]; wb[k] = w[ob * 10 + k]; }
  const float ba = bias ? bias[oa] : 0.f, bb = bias ? bias[ob] : 0.f;
  for (int tt = 0; tt < TT; ++tt) {
    const int t = t0 + tt;
    if (t >= T0) break;
    float xa = 0.f, xb = 0.f;
    // (the frame's ten samples through ONE vector load + readlane, not through the scalar data cache: see conv0_ln_gelu_kernel)
    const int lane = threadIdx.x & 63;
    const float xs = lane < 10 ? wv[5 * t + lane] : 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), k));
      xa = fmaf(wa[k], x, xa);
      xb = fmaf(wb[k], x, xb);
    }
    float* o = out + ((size_t)b * T0 + t) * 512;
    o[oa] = xa + ba;
    o[ob] = xb + bb;
  }
}

// The same layer with its LayerNorm(512) + GELU(erf) applied before anything is stored (HuBERT-large, feat_extract_norm = 'layer':
// transformers HubertLayerNormConvLayer).  ONE WAVE PER FRAME: a lane owns 8 adjacent channels (its 80 filter taps stay in
// registers for all the frames the wave handles), so a frame's two-pass statistics are two wave reductions -- no LDS, no workgroup
// barrier -- the ten waveform samples of a frame are wave-uniform (scalar loads), and a lane stores its 8 outputs as whole 16-byte
// (bf16) / 2 x 16-byte (fp32, fp16 plane pair) accesses.  Only the operand copy leaves the chip: 128 MB instead of 250 MB written +
// 250 MB read + 128 MB written at 4 x 10 s.  (Round 4: conv0 + LayerNorm kernels 88 + 99 us; first fused form -- 2 channels x 16
// frames per thread, 2-byte stores -- 128-143 us.)
template <typename T>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* wav, const float* w, const float* bias, const float* gamma,
                                                            const float* beta, typename Opnd<T>::E* out, long long lo_off, int n, int T0, float eps) {
  constexpr int FPW = 8;                      // frames per wave; a workgroup of 4 waves covers 32 consecutive frames
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = lane * 8;
  const float* wv = wav + (size_t)b * n;
  float wt[8][10], bs[8], ga[8], be[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int k = 0; k < 10; ++k) wt[c][k] = w[(c0 + c) * 10 + k];
    bs[c] = bias ? bias[c0 + c] : 0.f;
    ga[c] = gamma[c0 + c]; be[c] = beta[c0 + c];
  }
  const int f0 = (blockIdx.x * 4 + wave) * FPW;
  for (int f = 0; f < FPW; ++f) {
    const int t = f0 + f;                     // (wave-uniform)
    if (t >= T0) break;
    // The frame's ten samples: ONE vector load by lanes 0..9, handed to every lane through readlane.  (Round 4 read them with
    // wave-uniform addresses, i.e. through the scalar data cache: bit-stable in a process that owns the device, but with a second
    // process on the same GPU a handful of the 128 k frames came back wrong on every call -- the only kernel of the library that
    // pulled bulk data through the scalar cache, and the only stage of the encoder that was not reproducible there: round 5.)
    const float xs = lane < 10 ? wv[5 * t + lane] : 0.f;
    float x[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) x[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), k));
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 10; ++k) a = fmaf(wt[c][k], x[k], a);
      v[c] = a + bs[c];
    }
    s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    const float mean = wave_sum(s) * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] -= mean; q += v[c] * v[c]; }
    const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / 512.f) + eps);
    f32x4 y0, y1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      y0[c] = act_apply_t<T>(v[c] * rstd * ga[c] + be[c], ACT_GELU_ERF);
      y1[c] = act_apply_t<T>(v[c + 4] * rstd * ga[c + 4] + be[c + 4], ACT_GELU_ERF);
    }
    typename Opnd<T>::E* o = out + ((size_t)b * T0 + t) * 512 + c0;
    store_opnd4<T>(o, lo_off, y0);
    store_opnd4<T>(o + 4, lo_off, y1);
  }
}

// LeakyReLU(0.2) then InstanceNorm1d (biased variance, no affine) over L per (clip, channel).  Workgroup = 16 time-lanes x 64
// channels (coalesced across channels): the time-lanes stride over l and combine through LDS in a fixed order; three passes
// (mean, centred variance, normalise) as the reference's two-pass statistics.  (Round 4: the first form ran one thread per
// channel over all of L -- 16 workgroups of 1500 dependent strided loads each, 261 us at 4 x 498 frames, 8 % of a VQ decode.)
template <typename T>
__global__ __launch_bounds__(1024) void leaky_instnorm_kernel(const float* x, float* y_f32, T* y_t, int L, int d, float eps) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cl, b = blockIdx.y;
  const bool ok = ch < d;
  const float* xp = x + (size_t)b * L * d + (ok ? ch : 0);
  float s = 0.f;
  if (ok) for (int l = tl; l < L; l += 16) s += act_apply(xp[(size_t)l * d], ACT_LEAKY02);
  red[tl][cl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mean += red[i][cl];
  mean /= L;
  __syncthreads();
  float q = 0.f;
  if (ok) for (int l = tl; l < L; l += 16) { const float c = act_apply(xp[(size_t)l * d], ACT_LEAKY02) - mean; q += c * c; }
  red[tl][cl] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) var += red[i][cl];
  const float rstd = 1.f / sqrtf(var / L + eps);
  if (!ok) return;
  for (int l = tl; l < L; l += 16) {
    const float v = (act_apply(xp[(size_t)l * d], ACT_LEAKY02) - mean) * rstd;
    const size_t o = ((size_t)b * L + l) * d + ch;
    if (y_f32) y_f32[o] = v;
    if (y_t) y_t[o] = from_f32<T>(v);
  }
}

// GroupNorm(num_groups = C) of wav2vec2's first conv layer: per (clip, channel) statistics over time
// (biased variance), affine, then activation.  Workgroup = 16 time-lanes x 64 channels; the time-lanes
// stride over t and combine through LDS; three passes (mean, centred variance, normalise).
template <typename T>
__global__ __launch_bounds__(1024) void time_groupnorm_kernel(const float* x, const float* gamma, const float* beta, float* y_f32,
                                                              typename Opnd<T>::E* y_t, long long lo_off, int Tn, int C, float eps, int act) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cl, b = blockIdx.y;
  const bool ok = ch < C;
  const float* xp = x + (size_t)b * Tn * C + (ok ? ch : 0);
  float s = 0.f;
  if (ok) for (int t = tl; t < Tn; t += 16) s += xp[(size_t)t * C];
  red[tl][cl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mean += red[i][cl];
  mean /= Tn;
  __syncthreads();
  float q = 0.f;
  if (ok) for (int t = tl; t < Tn; t += 16) { const float c = xp[(size_t)t * C] - mean; q += c * c; }
  red[tl][cl] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) var += red[i][cl];
  const float rstd = 1.f / sqrtf(var / Tn + eps);
  if (!ok) return;
  const float gm = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
  for (int t = tl; t < Tn; t += 16) {
    const float v = act_apply_t<T>((xp[(size_t)t * C] - mean) * rstd * gm + bt, act);
    const size_t o = ((size_t)b * Tn + t) * C + ch;
    if (y_f32) y_f32[o] = v;
    if (y_t) store_opnd1<T>(y_t + o, lo_off, v);
  }
}

// The same normalisation for long clips, in two launches over time CHUNKS (round 4): the single-launch form above has C / 64 = 8
// workgroups per clip whatever the clip's length -- 2.13 ms of a 3.1 ms wav2vec2-base forward at 10 s of audio (32 k frames).
//   time_stats_kernel        grid (C / 64, chunks, B): per (clip, chunk, channel) sum and sum of squares in fp64 (one pass is exact
//                            enough in fp64: the variance is formed as q / T - mean^2 in double), time-lanes combined in a fixed order
//   time_norm_apply_kernel   grid (C / 64, chunks, B): every workgroup folds the chunk partials of its 64 channels in chunk order
//                            (so the statistics do not depend on which workgroup reads them), then normalises its own chunk
__global__ __launch_bounds__(1024) void time_stats_kernel(const float* x, double* part, int Tn, int C, int chunk) {
  __shared__ double rs[16][64], rq[16][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cl, c = blockIdx.y, b = blockIdx.z, nch = gridDim.y;
  const bool ok = ch < C;
  const float* xp = x + (size_t)b * Tn * C + (ok ? ch : 0);
  const int t1 = min(Tn, (c + 1) * chunk);
  double s = 0.0, q = 0.0;
  if (ok) for (int t = c * chunk + tl; t < t1; t += 16) { const double v = (double)xp[(size_t)t * C]; s += v; q += v * v; }
  rs[tl][cl] = s; rq[tl][cl] = q;
  __syncthreads();
  if (tl == 0 && ok) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ss += rs[i][cl]; qq += rq[i][cl]; }
    double* o = part + (((size_t)b * nch + c) * C + ch) * 2;
    o[0] = ss; o[1] = qq;
  }
}
template <typename T>
__global__ __launch_bounds__(1024) void time_norm_apply_kernel(const float* x, const double* part, const float* gamma, const float* beta,
                                                               float* y_f32, typename Opnd<T>::E* y_t, long long lo_off, int Tn, int C, int chunk, float eps, int act) {
  __shared__ float st[64][2];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cl, c = blockIdx.y, b = blockIdx.z, nch = gridDim.y;
  const bool ok = ch < C;
  if (tl == 0) {
    double ss = 0.0, qq = 0.0;
    if (ok) for (int i = 0; i < nch; ++i) { const double* o = part + (((size_t)b * nch + i) * C + ch) * 2; ss += o[0]; qq += o[1]; }
    const double mean = ss / Tn, var = fmax(qq / Tn - mean * mean, 0.0);
    st[cl][0] = (float)mean; st[cl][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (!ok) return;
  const float mean = st[cl][0], rstd = st[cl][1];
  const float gm = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
  const float* xp = x + (size_t)b * Tn * C + ch;
  const int t1 = min(Tn, (c + 1) * chunk);
  for (int t = c * chunk + tl; t < t1; t += 16) {
    const float v = act_apply_t<T>((xp[(size_t)t * C] - mean) * rstd * gm + bt, act);
    const size_t o = ((size_t)b * Tn + t) * C + ch;
    if (y_f32) y_f32[o] = v;
    if (y_t) store_opnd1<T>(y_t + o, lo_off, v);
  }
}

// sum |a - b|^p (p = 1, 2) in two deterministic stages: per-block partials, then one block folds them in
// a fixed order (F.mse_loss / F.l1_loss of p_losses, diffusion_BIWI_encoder_decoder.py:744-749).
__global__ __launch_bounds__(256) void diff_partial_kernel(const float* a, const float* b, float* partial, long long n, int l1) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s += l1 ? fabsf(d) : d * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void diff_final_kernel(const float* partial, int nb, float scale, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// AdaIN (utiles/adaIN.py:4-22): one wavefront per (n, c) row; unbiased variance + eps.
__global__ __launch_bounds__(256) void adain_kernel(const float* content, const float* style, float* out, int NC, int Lc, int Ls, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= NC) return;
  const float* c = content + (size_t)row * Lc;
  const float* s = style + (size_t)row * Ls;
  float a = 0.f;
  for (int i = lane; i < Lc; i += 64) a += c[i];
  const float cm = wave_sum(a) / Lc;
  a = 0.f;
  for (int i = lane; i < Lc; i += 64) { const float e = c[i] - cm; a += e * e; }
  const float cs = sqrtf(wave_sum(a) / (Lc - 1) + eps);
  a = 0.f;
  for (int i = lane; i < Ls; i += 64) a += s[i];
  const float sm = wave_sum(a) / Ls;
  a = 0.f;
  for (int i = lane; i < Ls; i += 64) { const float e = s[i] - sm; a += e * e; }
  const float ss = sqrtf(wave_sum(a) / (Ls - 1) + eps);
  for (int i = lane; i < Lc; i += 64) out[(size_t)row * Lc + i] = (c[i] - cm) / cs * ss + sm;
}

// ------------------------------------------------------------------------------------------------
// VQ nearest code: one wavefront per latent vector; each lane scans codes lane, lane+64, ...
// d_k = (sum_i z_i^2 + sum_i e_ki^2) - 2 * dot, every sum a sequential fmaf chain over i (the
// documented order shared with oracle/fdm_oracle_c.c); first-min argmin; z_q = z + (e - z).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_quant_kernel(const float* z, const float* codebook, const int* book, int B, int R, int c,
                                                       int K, float* zq_bcl, long long* idx) {
  __shared__ float zs[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= (long long)B * R) return;
  const int b = (int)(row / R), r = (int)(row - (long long)b * R);
  const float* zr = z + (size_t)row * c;
  for (int i = lane; i < c; i += 64) zs[wave][i] = zr[i];
  __builtin_amdgcn_wave_barrier();
  const float* E = codebook + (size_t)(book ? book[b] : 0) * K * c;
  float z2 = 0.f;
  for (int i = 0; i < c; ++i) z2 = __fmaf_rn(zs[wave][i], zs[wave][i], z2);
  float best = INFINITY;
  int bk = 0x7fffffff;
  for (int k = lane; k < K; k += 64) {
    const float* e = E + (size_t)k * c;
    float e2 = 0.f, dot = 0.f;
    for (int i = 0; i < c; ++i) {
      const float ev = e[i];
      e2 = __fmaf_rn(ev, ev, e2);
      dot = __fmaf_rn(zs[wave][i], ev, dot);
    }
    const float dk = __fsub_rn(__fadd_rn(z2, e2), __fmul_rn(2.f, dot));
    if (dk < best) { best = dk; bk = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int ok = __shfl_xor(bk, o, 64);
    if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
  }
  if (bk >= K) bk = 0;   // all-NaN row: keep the access in bounds
  if (lane == 0) idx[row] = bk;
  const float* e = E + (size_t)bk * c;
  for (int i = lane; i < c; i += 64) {
    const float zv = zs[wave][i];
    zq_bcl[((size_t)b * c + i) * R + r] = __fadd_rn(zv, __fsub_rn(e[i], zv));
  }
}

// ------------------------------------------------------------------------------------------------
// The rest of VectorQuantizer.forward's return tuple (models/lib/quantizer.py:46-61, models/vq_vae_emotion.py:232-249):
//   min_encodings [rows, K] one-hot of the chosen code, loss = beta * mean((e - z)^2) + mean((e - z)^2) (the two means are the
//   same number in a forward pass), perplexity = exp(-sum_k p_k log(p_k + 1e-10)), p = mean over rows of min_encodings.
// Pass 1: one wavefront per row (grid-stride, fixed row -> wave assignment): squared error of the row in a fixed lane order,
//   accumulated per wave in double; code histogram through integer atomics (exact in any order).  Pass 2: one workgroup sums the
//   per-workgroup partials in index order and evaluates the two scalars.  Deterministic; not on the sampling path's clock.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_stats_partial_kernel(const float* z, const float* codebook, const int* book, const long long* idx,
                                                               int B, int R, int c, int K, float* min_enc, double* partial, int* hist) {
  __shared__ double wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long rows = (long long)B * R;
  double acc = 0.0;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / R);
    const int k = (int)idx[row];
    const float* e = codebook + ((size_t)(book ? book[b] : 0) * K + k) * c;
    const float* zr = z + (size_t)row * c;
    float d2 = 0.f;
    for (int i = lane; i < c; i += 64) { const float d = __fsub_rn(e[i], zr[i]); d2 = __fmaf_rn(d, d, d2); }
    acc += (double)wave_sum(d2);
    if (lane == 0) atomicAdd(hist + k, 1);
    if (min_enc)
      for (int j = lane; j < K; j += 64) min_enc[(size_t)row * K + j] = (j == k) ? 1.f : 0.f;
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(256) void vq_stats_final_kernel(const double* partial, int nparts, const int* hist, int K, long long rows, int c,
                                                             float beta, float* out) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < nparts; i += 256) s += partial[i];
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  const double sq = red[0];
  __syncthreads();
  double h = 0.0;
  for (int k = t; k < K; k += 256) {
    const float pk = (float)hist[k] / (float)rows;                      // torch.mean of a 0/1 column: exact count / rows in fp32
    h += (double)(pk * logf(pk + 1e-10f));
  }
  red[t] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) {
    const float m = (float)(sq / ((double)rows * c));
    out[0] = __fadd_rn(__fmul_rn(beta, m), m);                         // beta * mean(.) + mean(.)
    out[1] = expf(-(float)red[0]);
  }
}

// ---- evaluation metrics (computer_metrix.py:84-136, metric/metric.py:115-138): HBM-bound reductions -------------------
// One workgroup per frame: squared vertex error d2 = ((gx-px)^2 + (gy-py)^2) + (gz-pz)^2 (numpy's order over axis 2, so
// the per-frame maximum is bit-identical to the reference's), over a vertex region (or all vertices when region == NULL).
// frame_max[f] = max d2;  frame_sum[2f] = sum d2, frame_sum[2f+1] = sum sqrt(d2)  (double accumulators, fixed order).
__global__ __launch_bounds__(256) void vertex_err_kernel(const float* gt, const float* pred, const int* region, int R, int V,
                                                         float* frame_max, double* frame_sum) {
  __shared__ float rmax[4];
  __shared__ double rs[4], rn[4];
  const size_t base = (size_t)blockIdx.x * V * 3;
  float mx = 0.f;
  double s2 = 0.0, sn = 0.0;
  for (int r = threadIdx.x; r < R; r += 256) {
    const size_t o = base + (size_t)(region ? region[r] : r) * 3;
    const float dx = __fsub_rn(gt[o], pred[o]), dy = __fsub_rn(gt[o + 1], pred[o + 1]), dz = __fsub_rn(gt[o + 2], pred[o + 2]);
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    mx = fmaxf(mx, d2);
    s2 += (double)d2;
    sn += sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    s2 += __shfl_xor(s2, o, 64);
    sn += __shfl_xor(sn, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { rmax[threadIdx.x >> 6] = mx; rs[threadIdx.x >> 6] = s2; rn[threadIdx.x >> 6] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    frame_max[blockIdx.x] = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    frame_sum[2 * blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    frame_sum[2 * blockIdx.x + 1] = (rn[0] + rn[1]) + (rn[2] + rn[3]);
  }
}
// out[0] = mean_f frame_max (LVE / FVE), out[1] = sum d2 / (F R) (EME-style mean), out[2] = sum |d| / (F R) (mean vertex error)
__global__ __launch_bounds__(256) void vertex_err_final_kernel(const float* frame_max, const double* frame_sum, int F, int R, double* out) {
  __shared__ double red[3][4];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int f = threadIdx.x; f < F; f += 256) { a += (double)frame_max[f]; b += frame_sum[2 * f]; c += frame_sum[2 * f + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (double)F;
    out[1] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / ((double)F * R);
    out[2] = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / ((double)F * R);
  }
}
// Upper-face dynamics (FDD, computer_metrix.py:95-105): s[f, r] = |verts[f, region[r]] - tmpl[region[r]]|^2; population
// std of s over frames per region vertex, then the mean over the region.  Stage 1: thread = region vertex, block row =
// frame chunk, partial (sum s, sum s^2) in double; stage 2 folds the chunks in a fixed order.
__global__ __launch_bounds__(256) void motion_partial_kernel(const float* verts, const float* tmpl, const int* region, int R, int F, int V,
                                                             double* partial) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int FC = gridDim.y, fc = blockIdx.y;
  const int f0 = (int)((long long)F * fc / FC), f1 = (int)((long long)F * (fc + 1) / FC);
  const int v = region ? region[r] : r;
  const float tx = tmpl[3 * v], ty = tmpl[3 * v + 1], tz = tmpl[3 * v + 2];
  double s1 = 0.0, s2 = 0.0;
  for (int f = f0; f < f1; ++f) {
    const size_t o = ((size_t)f * V + v) * 3;
    const float dx = __fsub_rn(verts[o], tx), dy = __fsub_rn(verts[o + 1], ty), dz = __fsub_rn(verts[o + 2], tz);
    const double s = (double)__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    s1 += s;
    s2 += s * s;
  }
  partial[((size_t)fc * R + r) * 2] = s1;
  partial[((size_t)fc * R + r) * 2 + 1] = s2;
}
__global__ __launch_bounds__(256) void motion_final_kernel(const double* partial, int FC, int R, int F, double* out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int r = threadIdx.x; r < R; r += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int fc = 0; fc < FC; ++fc) { s1 += partial[((size_t)fc * R + r) * 2]; s2 += partial[((size_t)fc * R + r) * 2 + 1]; }
    const double mu = s1 / F;
    acc += sqrt(fmax(s2 / F - mu * mu, 0.0));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)R;
}

// F.interpolate(mode='linear', align_corners=True) over time, channels-last x [B, Tin, C] -> y [B, Tout, C]
// (linear_interpolation, models/hubert.py:62-69 / models/wav2vec.py:61-67: the 50 -> 30 fps resampling of the audio features)
__global__ void linear_interp_kernel(const float* x, float* y, int B, int Tin, int Tout, int C) {
  const size_t n = (size_t)B * Tout * C;
  const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int t = (int)((i / C) % Tout);
    const int b = (int)(i / ((size_t)C * Tout));
    const float src = __fmul_rn(scale, (float)t);
    const int i0 = min((int)src, Tin - 1);
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float l1 = __fsub_rn(src, (float)i0), l0 = __fsub_rn(1.f, l1);
    const float* xb = x + (size_t)b * Tin * C + c;
    y[i] = __fadd_rn(__fmul_rn(l0, xb[(size_t)i0 * C]), __fmul_rn(l1, xb[(size_t)i1 * C]));
  }
}

}  // namespace fdm
