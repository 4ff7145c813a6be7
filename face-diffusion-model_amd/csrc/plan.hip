// Plan layer of libfdm_hip.so (include/fdm_hip.h, "Plan layer"): the FDM denoiser + diffusion scheduler of one model
// behind plain C calls.  What the reference does in FDM.__init__ / FDM.forward (models/fdm_vocaset.py:9-91,
// models/fdm_vqvae_mead.py:9-104, models/fdm.py:10-99) and GaussianDiffusion.p_sample_loop / ddim_sample
// (video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:649-710) is split here into
//   commit   (once per model)   operand-kind weight copies, tau table, folded cross-attention time tables, LayerNorm folds
//   prepare  (once per batch)   AF = audio_extract(features), per-layer tables C1_l, conditioning addend E0
//   step program (per step)     recorded once through fdm_op_*, captured into a hipGraph, replayed with t from a device counter
// All arithmetic runs in the library's own kernels (no torch, no vendor BLAS); host code only sequences launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/fdm_hip.h"
#include "common.hpp"
#include "kernels.hpp"

namespace {
using fdm::fail;

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(FDM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define FCK(x) do { int r_ = (x); if (r_ != FDM_OK) return r_; } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// one-time weight preparation kernels (plan commit)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* in, float* out, int rows, int cols) {      // out[c][r] = in[r][c]
  const long long n = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / rows), r = (int)(i % rows);
    out[i] = in[(size_t)r * cols + c];
  }
}
__global__ void scale_cols_kernel(const float* W, const float* gamma, float* out, long long n, int K) {   // out[j][k] = W[j][k] * gamma[k]
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = W[i] * gamma[i % K];
}
// out[j] = sum_k (float) Wt[j][k]: one wavefront per row, fixed order (lane-strided partial sums, then the DPP tree)
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const fdm::bf16* W, float* out, int N, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += (float)W[(size_t)row * K + k];
  s = fdm::wave_sum(s);
  if (lane == 0) out[row] = s;
}
// the same for a split operand: the value a GEMM sees is hi + lo / scale
template <typename E>
__global__ __launch_bounds__(256) void rowsum_split_kernel(const E* W, long long lo_off, float inv_scale, float* out, int N, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += (float)W[(size_t)row * K + k] + (float)W[lo_off + (size_t)row * K + k] * inv_scale;
  s = fdm::wave_sum(s);
  if (lane == 0) out[row] = s;
}
int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

// ---------------------------------------------------------------------------------------------------------------------
// host tables
// ---------------------------------------------------------------------------------------------------------------------
void alibi_slopes(int n, std::vector<double>& out) {       // get_slopes, models/fdm_vocaset.py:96-106
  auto p2 = [](int m, std::vector<double>& o) {
    const double start = std::pow(2.0, -std::pow(2.0, -(std::log2((double)m) - 3.0)));
    for (int i = 0; i < m; ++i) o.push_back(start * std::pow(start, (double)i));
  };
  const double l2 = std::log2((double)n);
  if (l2 == std::floor(l2)) { p2(n, out); return; }
  const int c = 1 << (int)std::floor(l2);
  p2(c, out);
  std::vector<double> more;
  alibi_slopes(2 * c, more);
  for (int i = 0; i < n - c; ++i) out.push_back(more[2 * i]);
}

}  // namespace

extern "C" {

int fdm_schedule_host(int T, float* out) {
  if (T <= 0 || !out) return fail(FDM_ERR_ARG, "schedule_host: bad argument");
  // cosine_beta_schedule (:537-547) and the 12 buffers (:565-603), fp64 in the reference's expression order
  const double s = 0.008;
  std::vector<double> ac(T + 1), betas(T), alphas(T), acp(T), acprev(T);
  for (int i = 0; i <= T; ++i) {
    const double c = std::cos((((double)i / T) + s) / (1 + s) * M_PI * 0.5);
    ac[i] = c * c;
  }
  const double a0 = ac[0];
  for (int i = 0; i <= T; ++i) ac[i] = ac[i] / a0;
  for (int i = 0; i < T; ++i) {
    double b = 1 - (ac[i + 1] / ac[i]);
    b = b < 0 ? 0 : (b > 0.9999 ? 0.9999 : b);
    betas[i] = b;
    alphas[i] = 1.0 - b;
  }
  double run = 1.0;
  for (int i = 0; i < T; ++i) { acprev[i] = run; run = (i == 0) ? alphas[0] : run * alphas[i]; acp[i] = run; }
  for (int i = 0; i < T; ++i) {
    const double pv = betas[i] * (1.0 - acprev[i]) / (1.0 - acp[i]);
    const double v[12] = {betas[i], acp[i], acprev[i], std::sqrt(acp[i]), std::sqrt(1.0 - acp[i]), std::log(1.0 - acp[i]),
                          std::sqrt(1.0 / acp[i]), std::sqrt(1.0 / acp[i] - 1), pv, std::log(pv < 1e-20 ? 1e-20 : pv),
                          betas[i] * std::sqrt(acprev[i]) / (1.0 - acp[i]), (1.0 - acprev[i]) * std::sqrt(alphas[i]) / (1.0 - acp[i])};
    for (int k = 0; k < 12; ++k) out[(size_t)k * T + i] = (float)v[k];
  }
  return FDM_OK;
}

int fdm_ddim_schedule_host(int steps, int T, int* t, int* t_next, float* sqrt_an, float* c_n) {
  if (steps <= 0 || T <= 0) return fail(FDM_ERR_ARG, "ddim_schedule_host: bad argument");
  // times = linspace(-1, T-1, steps+1).astype(int32) reversed, zipped (:684-687); numpy: arange(num) * step + start, last = stop
  std::vector<int> times(steps + 1);
  const double start = -1.0, stop = (double)T - 1.0, step = (stop - start) / steps;
  for (int i = 0; i <= steps; ++i) times[i] = (int)(i == steps ? stop : (double)i * step + start);
  std::vector<float> buf;
  if (sqrt_an || c_n) { buf.resize((size_t)12 * T); fdm_schedule_host(T, buf.data()); }
  int n = 0;
  for (int i = steps; i >= 1; --i) {
    const int tc = times[i], tn = times[i - 1];
    if (tn < 0) continue;                    // the dead last pair (:695-696)
    if (t) t[n] = tc;
    if (t_next) t_next[n] = tn;
    if (sqrt_an || c_n) {
      const float an = buf[(size_t)1 * T + tn];          // alphas_cumprod[t_next]; eta = 0 -> sigma = 0 (:699-708)
      if (sqrt_an) sqrt_an[n] = std::sqrt(an);
      if (c_n) c_n[n] = std::sqrt((1.f - an) - 0.f);
    }
    ++n;
  }
  return n;
}

int fdm_alibi_slopes_host(int n_head, float* out) {
  if (n_head <= 0 || !out) return fail(FDM_ERR_ARG, "alibi_slopes_host: bad argument");
  std::vector<double> v;
  alibi_slopes(n_head, v);
  for (int i = 0; i < n_head; ++i) out[i] = (float)v[i];
  return FDM_OK;
}

int fdm_pe_table_host(int d, int periodic, int period, int rows, float* out) {
  if (d <= 0 || d % 2 || rows <= 0 || !out || (periodic && period <= 0)) return fail(FDM_ERR_ARG, "pe_table_host: bad argument");
  // pe[p, 2k] = sin(p w_k), pe[p, 2k+1] = cos(p w_k), w_k = exp(2k * (-ln 10000 / d)); periodic: p -> p mod period (:150-184).
  // The reference evaluates these in fp32 torch ops; here each fp32 step is the correctly rounded value of the same function
  // (<= 1 ulp from any fp32 libm); callers that need the reference buffer bit for bit pass "PE.pe" to fdm_plan_set_weights.
  const float coef = (float)(-std::log(10000.0) / d);
  for (int p = 0; p < rows; ++p) {
    const float pos = (float)(periodic ? p % period : p);
    for (int k = 0; k < d; k += 2) {
      const float div = (float)std::exp((double)((float)k * coef));
      const float arg = pos * div;
      out[(size_t)p * d + k] = (float)std::sin((double)arg);
      out[(size_t)p * d + k + 1] = (float)std::cos((double)arg);
    }
  }
  return FDM_OK;
}

int fdm_model_preset(const char* name, fdm_model_desc* o) {
  if (!name || !o) return fail(FDM_ERR_ARG, "model_preset: null argument");
  const fdm_model_desc vocaset = {1024, 8, 8, 2048, 16, 64, 8, 0, 1024, 1, 1, 30, 1, 0, 600};
  const fdm_model_desc mead = {512, 4, 8, 1024, 8, 64, 25, 7, 2048, 2, 0, 30, 1, 0, 600};
  const fdm_model_desc biwi = {1024, 4, 8, 2048, 8, 128, 6, 0, 1536, 2, 0, 25, 0, 1, 600};
  const std::string n(name);
  if (n == "vocaset") *o = vocaset;
  else if (n == "mead") *o = mead;
  else if (n == "biwi") *o = biwi;
  else if (n == "vocaset_tiny") { *o = vocaset; o->d = 256; o->n_head = 2; o->n_layers = 2; o->ffn = 512; o->c = 16; }
  else if (n == "mead_tiny") { *o = mead; o->d = 256; o->n_head = 2; o->n_layers = 2; o->ffn = 512; o->c = 32; }
  else return fail(FDM_ERR_ARG, "model_preset: unknown preset '%s'", name);
  return FDM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// the plan
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Mat { void* p = nullptr; long long lo = 0; };       // operand-kind matrix: pointer + hi->lo plane distance (elements)
struct Fold { Mat w; float* colsum = nullptr; float* bias = nullptr; const float* gamma = nullptr; const float* beta = nullptr; };
struct Wt { float* p = nullptr; long long n = 0; };
}  // namespace

struct fdm_plan {
  fdm_model_desc m{};
  int dtype = FDM_F32, hd = 0;
  std::vector<void*> allocs, ws_allocs, commit_allocs;     // plan lifetime | per capacity | per commit (freed when weights change)
  std::map<std::string, Wt> w;               // fp32 weights / buffers by reference state-dict name (plan-owned copies)
  bool committed = false, in_commit = false;
  // ---- per model
  std::map<std::string, Mat> wt;             // operand-kind copies of the step's matrices
  float* tau = nullptr;
  std::vector<float*> TT;
  std::vector<const float*> Wv, bv, Wo, bo;
  bool fuse_ln3 = false;
  std::map<int, Fold> fold;                  // layer l (1 .. n_layers-1) reads norm3 of layer l-1; -1 = latent decoder
  float *slopes = nullptr, *pe = nullptr;
  float *c1 = nullptr, *c2 = nullptr, *sigma = nullptr, *sra = nullptr, *srm1 = nullptr;
  // ---- per shape (capacity cap*, current B, L, ...)
  int capB = 0, capL = 0, capRep = 0;
  int B = 0, L = 0, M = 0, rep = 1, R = 0, Lpad = 0, cfg = 0;     // B = row blocks ("virtual clips") = audio clips x S
  int S = 1;                                 // conditions per audio clip sharing the clip's AF / C1_l tables (fdm_audio_prepare_conds)
  bool prepared = false;
  float *h = nullptr, *h2 = nullptr, *x1 = nullptr, *x0 = nullptr, *x = nullptr, *x2 = nullptr, *stats = nullptr;
  Mat xt, ht, h2t, x2t, ctx, u;
  void *q = nullptr, *kp = nullptr, *vp = nullptr;
  long long q_lo = 0, kv_lo = 0;             // FDM_F16X3: plane distances of q and of the packed K / V buffers
  size_t kv_bytes = 0;
  float *AF = nullptr, *t1 = nullptr, *sty = nullptr, *em = nullptr, *emu = nullptr, *zeros = nullptr, *E0 = nullptr;
  std::vector<float*> C1;
  int* step = nullptr;                       // [device step counter, t of the current step]
  unsigned long long* seedbuf = nullptr;     // {Philox seed, global index of clip 0}: read by the scheduler at run time
  int* tseq = nullptr; int tseq_cap = 0;
  std::map<int, std::pair<int, float*>> ddim;   // ddim_steps -> (live pairs, device [san | cn])
  std::map<int, std::vector<int>> ddim_t;
  // ---- programs and tiles
  std::map<std::string, fdm_prog*> progs;
  std::vector<std::string> prog_order;       // least recently used first
  std::vector<std::string> pinned;           // programs handed out during the current API call: never evicted by it
  std::map<std::string, int> tiles;
  std::map<std::string, std::map<std::string, int>> tile_cache;     // by shape key
  std::map<std::string, long long> steps_seen;
  std::map<std::string, std::vector<fdm_gemm_args>>* tune_rec = nullptr;
  int tune_enabled = 1;
  int tune_failed = 0;                       // opt-in request-path tuning runs that failed (heuristic tiles kept)
  std::set<std::string> tune_failed_shapes;  // ... and their shapes: the request path tries a shape once (fdm_plan_tune retries)
  int want_fuse_ln3 = 0;                     // fdm_plan_set "fuse_ln3": fold norm3 into the GEMMs around it at the next commit
  // K slices of the two GEMMs whose fp32 output row is read next by a LayerNorm launch (out-proj -> LN1+LN2, FFN2 -> LN3): S > 1 = S
  // partial planes of x1, summed by that launch (fdm_gemm_args.ksplit / fdm_ln_args.x_planes).  A property of the plan, NOT of the
  // shape: results depend on S, and a clip must compute the same bits in every batch composition.
  int ksplit_out = 1, ksplit_ffn2 = 1;
  int x1_planes = 0;                 // fp32 planes the workspace's x1 holds (one per K slice)
  int lockstep = 0;                          // fdm_plan_set "lockstep": the lockstep k loop in every GEMM of the step (A/B against the loader-wave form; same bits)
  int tune_lazy = 0;                         // 1: fdm_sample_graph may tune in-call once a shape has run 2000 steps (opt-in)
  long long last_graph_launches = 0, launches_per_step = 0;
};

namespace {

size_t esize(int dtype) { return dtype == FDM_F32 ? 4 : 2; }
bool is_split(int dtype) { return dtype == FDM_F16X3; }

int dalloc(fdm_plan* P, void** out, size_t bytes, bool ws, bool zero = true) {
  void* p = nullptr;
  HIPCK(hipMalloc(&p, bytes ? bytes : 16));
  if (zero) HIPCK(hipMemset(p, 0, bytes ? bytes : 16));
  (ws ? P->ws_allocs : (P->in_commit ? P->commit_allocs : P->allocs)).push_back(p);     // commit-time tables are freed when a weight changes
  *out = p;
  return FDM_OK;
}
template <typename T> int dalloc_t(fdm_plan* P, T** out, size_t n, bool ws) { return dalloc(P, (void**)out, n * sizeof(T), ws); }
// operand-kind matrix [rows, cols]: plain, or two consecutive planes for the split kinds
int dalloc_mat(fdm_plan* P, Mat* out, size_t rows, size_t cols, bool ws) {
  const size_t n = rows * cols;
  out->lo = is_split(P->dtype) ? (long long)n : 0;
  return dalloc(P, &out->p, n * esize(P->dtype) * (is_split(P->dtype) ? 2 : 1), ws);
}
Mat mat_rows(const fdm_plan* P, const Mat& m, size_t row0, size_t cols) {
  Mat r = m;
  r.p = (char*)m.p + row0 * cols * esize(P->dtype);
  return r;
}

std::string shape_key(const fdm_plan* P) {
  // (the K-slice factors belong to the key: a tile set tuned without slices holds tiles a sliced launch cannot run on, and the other
  //  way round -- in the in-process cache and in the FDM_TILE_CACHE file alike)
  char b[80];
  snprintf(b, sizeof(b), "%d,%d,%d,%d,%d,ks%d.%d", P->R, P->M, P->L, P->rep, P->S, P->ksplit_out, P->ksplit_ffn2);
  return b;
}

int drop_programs(fdm_plan* P, void* stream) {
  if (P->progs.empty()) return FDM_OK;
  // graph execs / kernarg storage may still be referenced by queued replays: drain the stream they were launched on first
  if (stream) HIPCK(hipStreamSynchronize((hipStream_t)stream));
  else HIPCK(hipDeviceSynchronize());
  for (auto& kv : P->progs) fdm_prog_destroy(kv.second);
  P->progs.clear();
  P->prog_order.clear();
  P->pinned.clear();
  return FDM_OK;
}

const Wt* weight(const fdm_plan* P, const std::string& name) {
  auto it = P->w.find(name);
  return it == P->w.end() ? nullptr : &it->second;
}
int need(const fdm_plan* P, const std::string& name, long long n, const float** out) {
  const Wt* w = weight(P, name);
  if (!w) return fail(FDM_ERR_STATE, "plan: missing weight %s", name.c_str());
  if (w->n != n) return fail(FDM_ERR_SHAPE, "plan: weight %s has %lld elements, expected %lld", name.c_str(), w->n, n);
  *out = w->p;
  return FDM_OK;
}

// fp32 GEMM args with the defaults the op layer's callers use (dense row-major operands)
fdm_gemm_args gemm_f32(const float* A, const float* W, int M, int N, int K) {
  fdm_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = FDM_F32;
  a.ldr = N; a.ldo_f32 = N; a.ldo_t = N; a.ln_eps = 1e-5f;
  return a;
}
fdm_gemm_args gemm_op(const fdm_plan* P, const Mat& A, const Mat& W, int M, int N, int K) {
  fdm_gemm_args a = gemm_f32((const float*)A.p, (const float*)W.p, M, N, K);
  a.dtype = P->dtype; a.a_lo_off = A.lo; a.w_lo_off = W.lo;
  return a;
}
void set_out_t(fdm_gemm_args& a, const Mat& o) { a.out_t = o.p; a.out_t_lo_off = o.lo; }

int to_operand(fdm_plan* P, const float* src, long long n, Mat* out, void* stream) {
  if (P->dtype == FDM_F32) { out->p = (void*)src; out->lo = 0; return FDM_OK; }
  out->lo = is_split(P->dtype) ? n : 0;
  FCK(dalloc(P, &out->p, (size_t)n * 2 * (is_split(P->dtype) ? 2 : 1), false, false));
  return fdm_op_cast(src, out->p, n, P->dtype, stream);
}

int plan_gemm(fdm_plan* P, const char* label, fdm_gemm_args a, void* stream) {
  if (P->tune_rec) (*P->tune_rec)[label].push_back(a);
  auto it = P->tiles.find(label);
  a.tile = (it == P->tiles.end() ? 0 : it->second) | (P->lockstep ? FDM_TILE_LOCKSTEP : 0);
  return fdm_op_gemm(&a, stream);
}

std::string lname(int l, const char* suffix) {
  char b[96];
  snprintf(b, sizeof(b), "transformer_decoder.layers.%d.%s", l, suffix);
  return b;
}

// ---------------------------------------------------------------------------------------------------------------------
// commit: per-model tables
// ---------------------------------------------------------------------------------------------------------------------
int commit_impl(fdm_plan* P, void* stream) {
  const fdm_model_desc& m = P->m;
  const int d = m.d;
  hipStream_t s = (hipStream_t)stream;
  FCK(drop_programs(P, stream));
  const float* p = nullptr;
  FCK(need(P, "audio_extract.0.weight", (long long)d * m.audio_in, &p));
  FCK(need(P, "audio_extract.0.bias", d, &p));
  FCK(need(P, "audio_extract.2.weight", (long long)d * d, &p));
  FCK(need(P, "audio_extract.2.bias", d, &p));
  FCK(need(P, "style_embedd.weight", (long long)d * m.n_style, &p));
  FCK(need(P, "style_embedd.bias", d, &p));
  if (m.n_emo) { FCK(need(P, "emotion_embedd.weight", (long long)d * m.n_emo, &p)); FCK(need(P, "emotion_embedd.bias", d, &p)); }
  FCK(need(P, "latent_encoder.0.bias", d, &p));
  FCK(need(P, "latent_decoder.bias", d, &p));
  // operand-kind copies of the per-step matrices
  std::vector<std::pair<std::string, long long>> mats = {{"latent_encoder.0.weight", (long long)d * d}, {"latent_decoder.weight", (long long)d * d}};
  for (int l = 0; l < m.n_layers; ++l) {
    mats.push_back({lname(l, "self_attn.in_proj_weight"), 3LL * d * d});
    mats.push_back({lname(l, "self_attn.out_proj.weight"), (long long)d * d});
    mats.push_back({lname(l, "linear1.weight"), (long long)m.ffn * d});
    mats.push_back({lname(l, "linear2.weight"), (long long)d * m.ffn});
    for (const char* b : {"self_attn.in_proj_bias", "self_attn.out_proj.bias", "linear1.bias", "linear2.bias", "norm1.weight", "norm1.bias",
                          "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias"}) {
      const long long nb = !strcmp(b, "self_attn.in_proj_bias") ? 3LL * d : (!strcmp(b, "linear1.bias") ? m.ffn : d);
      FCK(need(P, lname(l, b), nb, &p));
    }
  }
  for (auto& mt : mats) {
    FCK(need(P, mt.first, mt.second, &p));
    Mat o;
    FCK(to_operand(P, p, mt.second, &o, stream));
    P->wt[mt.first] = o;
  }
  // tau table [1000, d] = Mish(W_t^T + b_t): Linear(one_hot(t)) is a column gather (models/fdm_vocaset.py:71-72)
  const float *Wt_ = nullptr, *bt = nullptr;
  FCK(need(P, "time_embedd.0.weight", (long long)d * 1000, &Wt_));
  FCK(need(P, "time_embedd.0.bias", d, &bt));
  float* wtT = nullptr;
  FCK(dalloc_t(P, &wtT, (size_t)1000 * d, false));
  FCK(dalloc_t(P, &P->tau, (size_t)1000 * d, false));
  hipLaunchKernelGGL(transpose_kernel, dim3(grid_for(1000LL * d)), dim3(256), 0, s, Wt_, wtT, d, 1000);
  FCK(fdm_op_bias_act(wtT, bt, P->tau, 1000, d, FDM_ACT_MISH, stream));
  // folded cross-attention time tables TT_l = (tau Wv_l^T) Wo_l^T: fp32 MFMA, one-time
  float* tmp = nullptr;
  FCK(dalloc_t(P, &tmp, (size_t)1000 * d, false));
  P->TT.assign(m.n_layers, nullptr);
  P->Wv.assign(m.n_layers, nullptr); P->bv = P->Wo = P->bo = P->Wv;
  for (int l = 0; l < m.n_layers; ++l) {
    const float *ipw = nullptr, *ipb = nullptr;
    FCK(need(P, lname(l, "multihead_attn.in_proj_weight"), 3LL * d * d, &ipw));
    FCK(need(P, lname(l, "multihead_attn.in_proj_bias"), 3LL * d, &ipb));
    FCK(need(P, lname(l, "multihead_attn.out_proj.weight"), (long long)d * d, &P->Wo[l]));
    FCK(need(P, lname(l, "multihead_attn.out_proj.bias"), d, &P->bo[l]));
    P->Wv[l] = ipw + 2LL * d * d;
    P->bv[l] = ipb + 2LL * d;
    FCK(dalloc_t(P, &P->TT[l], (size_t)1000 * d, false));
    fdm_gemm_args g = gemm_f32(P->tau, P->Wv[l], 1000, d, d);
    g.out_f32 = tmp;
    FCK(fdm_op_gemm(&g, stream));
    g = gemm_f32(tmp, P->Wo[l], 1000, d, d);
    g.out_f32 = P->TT[l];
    FCK(fdm_op_gemm(&g, stream));
  }
  // Optional (fdm_plan_set "fuse_ln3"; bf16 and f16x3): norm3 of layer l-1 folded into the QKV / out-proj GEMMs of layer l and into the
  // latent decoder,   LN(x) W^T + b = rstd (x W'^T - mu colsum(W')) + (W beta + b),  W' = W o gamma   -- 8 launches fewer.
  // Off by default since the GEMM kernels were specialised: the plain GEMM + LayerNorm launch is now as fast or faster than
  // the three GEMMs that carry the fold (profiles/README.md, A/B of the final build).
  P->fuse_ln3 = (P->dtype == FDM_BF16 || P->dtype == FDM_F16X3) && P->want_fuse_ln3;     // fdm_plan_set(p, "fuse_ln3", 1) before the commit
  if (P->fuse_ln3) {
    auto make_fold = [&](const std::string& wname, const std::string& bname, int N, int l_prev, Fold* f) -> int {
      const float *W = nullptr, *b = nullptr, *gam = nullptr, *bet = nullptr;
      FCK(need(P, wname, (long long)N * d, &W));
      FCK(need(P, bname, N, &b));
      FCK(need(P, lname(l_prev, "norm3.weight"), d, &gam));
      FCK(need(P, lname(l_prev, "norm3.bias"), d, &bet));
      float* wg = nullptr;
      FCK(dalloc_t(P, &wg, (size_t)N * d, false));
      hipLaunchKernelGGL(scale_cols_kernel, dim3(grid_for((long long)N * d)), dim3(256), 0, s, W, gam, wg, (long long)N * d, d);
      FCK(to_operand(P, wg, (long long)N * d, &f->w, stream));
      FCK(dalloc_t(P, &f->colsum, (size_t)N, false));
      if (P->dtype == FDM_BF16)
        hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((N + 3) / 4), dim3(256), 0, s, (const fdm::bf16*)f->w.p, f->colsum, N, d);
      else
        hipLaunchKernelGGL(rowsum_split_kernel<fdm::f16>, dim3((N + 3) / 4), dim3(256), 0, s, (const fdm::f16*)f->w.p, f->w.lo, 1.f / 2048.f, f->colsum, N, d);
      FCK(dalloc_t(P, &f->bias, (size_t)N, false));
      FCK(fdm_op_small_linear(bet, W, b, f->bias, 1, d, N, FDM_ACT_NONE, stream));      // W beta + b
      f->gamma = gam; f->beta = bet;
      return FDM_OK;
    };
    for (int l = 1; l < m.n_layers; ++l)
      FCK(make_fold(lname(l, "self_attn.in_proj_weight"), lname(l, "self_attn.in_proj_bias"), 3 * d, l - 1, &P->fold[l]));
    FCK(make_fold("latent_decoder.weight", "latent_decoder.bias", d, m.n_layers - 1, &P->fold[-1]));
  }
  HIPCK(hipGetLastError());
  // ALiBi slopes, positional table, schedule tables
  std::vector<float> hs(m.n_head);
  fdm_alibi_slopes_host(m.n_head, hs.data());
  FCK(dalloc_t(P, &P->slopes, (size_t)m.n_head, false));
  HIPCK(hipMemcpyAsync(P->slopes, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, s));
  const int pe_rows = m.max_len + 30;
  FCK(dalloc_t(P, &P->pe, (size_t)pe_rows * d, false));
  if (const Wt* pw = weight(P, "PE.pe")) {           // the reference's registered buffer [1, rows, d]
    if (pw->n < (long long)m.max_len * d) return fail(FDM_ERR_SHAPE, "plan: PE.pe has %lld elements, need >= %lld", pw->n, (long long)m.max_len * d);
    const long long n = pw->n < (long long)pe_rows * d ? pw->n : (long long)pe_rows * d;
    HIPCK(hipMemcpyAsync(P->pe, pw->p, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  } else {
    std::vector<float> hp((size_t)pe_rows * d);
    fdm_pe_table_host(d, m.pe_periodic, m.period, pe_rows, hp.data());
    HIPCK(hipMemcpyAsync(P->pe, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));
  }
  std::vector<float> sb((size_t)12 * 1000);
  fdm_schedule_host(1000, sb.data());
  std::vector<float> sg(1000);
  for (int i = 0; i < 1000; ++i) sg[i] = (float)std::exp((double)(0.5f * sb[9000 + i]));      // exp(0.5 logvar), p_sample :655
  struct { const char* name; float** dst; const float* host; } tabs[] = {
      {"sched.c1", &P->c1, &sb[10000]}, {"sched.c2", &P->c2, &sb[11000]}, {"sched.sigma", &P->sigma, sg.data()},
      {"sched.sra", &P->sra, &sb[6000]}, {"sched.srm1", &P->srm1, &sb[7000]}};
  for (auto& t : tabs) {
    FCK(dalloc_t(P, t.dst, (size_t)1000, false));
    if (const Wt* ow = weight(P, t.name)) {
      if (ow->n != 1000) return fail(FDM_ERR_SHAPE, "plan: %s must have 1000 elements", t.name);
      HIPCK(hipMemcpyAsync(*t.dst, ow->p, 4000, hipMemcpyDeviceToDevice, s));
    } else {
      HIPCK(hipMemcpyAsync(*t.dst, t.host, 4000, hipMemcpyHostToDevice, s));
    }
  }
  HIPCK(hipStreamSynchronize(s));         // host staging vectors go out of scope
  P->committed = true;
  return FDM_OK;
}

// Everything commit allocates (operand-kind weight copies, tau / TT tables, folds, schedule and PE tables) is derived from the
// weights: released when a weight changes under it (fdm_plan_set_weights) and rebuilt by the next commit.
int release_commit(fdm_plan* P, void* stream) {
  FCK(drop_programs(P, stream));           // recorded programs point into the tables
  if (!P->commit_allocs.empty()) {
    if (stream) HIPCK(hipStreamSynchronize((hipStream_t)stream)); else HIPCK(hipDeviceSynchronize());
    for (void* p : P->commit_allocs) (void)hipFree(p);
    P->commit_allocs.clear();
  }
  P->committed = false; P->prepared = false;
  P->wt.clear(); P->fold.clear(); P->TT.clear();
  P->tau = nullptr; P->slopes = P->pe = nullptr;
  P->c1 = P->c2 = P->sigma = P->sra = P->srm1 = nullptr;
  return FDM_OK;
}

int commit(fdm_plan* P, void* stream) {
  if (P->committed) return FDM_OK;
  if (!P->commit_allocs.empty()) FCK(release_commit(P, stream));      // a commit that failed half way
  P->in_commit = true;
  const int rc = commit_impl(P, stream);
  P->in_commit = false;
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------------------------------
int reserve(fdm_plan* P, int B, int L, int cfg) {
  const int rep = cfg ? 2 : 1;
  if (B <= P->capB && L <= P->capL && rep <= P->capRep) return FDM_OK;
  B = B > P->capB ? B : P->capB; L = L > P->capL ? L : P->capL;
  const int repc = rep > P->capRep ? rep : P->capRep;
  FCK(drop_programs(P, nullptr));
  for (void* p : P->ws_allocs) (void)hipFree(p);
  P->ws_allocs.clear();
  P->prepared = false;
  const fdm_model_desc& m = P->m;
  const size_t d = m.d, M = (size_t)B * L, R = M * repc;
  const size_t Lpad = ((size_t)L + 31) / 32 * 32;
  FCK(dalloc_t(P, &P->h, R * d, true)); FCK(dalloc_t(P, &P->h2, R * d, true)); P->x1_planes = std::max(1, std::max(P->ksplit_out, P->ksplit_ffn2));      // (x1: one fp32 plane per K slice of the plan's setting; plan_set grows it)
  FCK(dalloc_t(P, &P->x1, R * d * P->x1_planes, true));
  FCK(dalloc_t(P, &P->x0, R * d, true)); FCK(dalloc_t(P, &P->x, M * d, true));
  if (P->dtype != FDM_F32) {
    FCK(dalloc_mat(P, &P->xt, M, d, true)); FCK(dalloc_mat(P, &P->ht, R, d, true)); FCK(dalloc_mat(P, &P->h2t, R, d, true));
  } else {        // fp32 operands alias the fp32 residual-stream buffers
    P->xt = Mat{P->x, 0}; P->ht = Mat{P->h, 0}; P->h2t = Mat{P->h2, 0};
  }
  if (P->dtype != FDM_F32) {     // folded-norm3 buffers (raw rows, their operand copy, per-row partial sums)
    FCK(dalloc_t(P, &P->x2, R * d, true)); FCK(dalloc_mat(P, &P->x2t, R, d, true)); FCK(dalloc_t(P, &P->stats, (d / 64) * R * 2, true));
  }
  // q: row-major queries; kp / vp: fragment-packed keys / values written by the QKV GEMM's epilogue (zeroed: pad keys must be
  // finite).  Split modes: attention runs in fp32, ctx returns as a plane pair.
  // (FDM_F16X3: fp16 plane pairs, the same bytes as fp32)
  const size_t ea = (P->dtype == FDM_BF16 || P->dtype == FDM_F16) ? 2 : 4;
  FCK(dalloc(P, &P->q, R * d * ea, true));
  FCK(dalloc(P, &P->kp, (size_t)B * repc * Lpad * d * ea, true));
  FCK(dalloc(P, &P->vp, (size_t)B * repc * Lpad * d * ea, true));
  P->q_lo = (long long)(R * d);
  P->kv_lo = (long long)((size_t)B * repc * Lpad * d);
  P->kv_bytes = (size_t)B * repc * Lpad * d * ea;
  FCK(dalloc_mat(P, &P->ctx, R, d, true));
  FCK(dalloc_mat(P, &P->u, R, m.ffn, true));
  FCK(dalloc_t(P, &P->AF, M * d, true)); FCK(dalloc_t(P, &P->t1, M * d, true));
  FCK(dalloc_t(P, &P->sty, (size_t)B * d, true)); FCK(dalloc_t(P, &P->em, (size_t)B * d, true)); FCK(dalloc_t(P, &P->emu, (size_t)B * d, true));
  FCK(dalloc_t(P, &P->zeros, (size_t)B * (m.n_emo > 0 ? m.n_emo : 1), true));
  FCK(dalloc_t(P, &P->E0, R * d, true));
  P->C1.assign(m.n_layers, nullptr);
  for (int l = 0; l < m.n_layers; ++l) FCK(dalloc_t(P, &P->C1[l], R * d, true));
  FCK(dalloc_t(P, &P->step, (size_t)4, true));
  FCK(dalloc_t(P, &P->seedbuf, (size_t)2, true));
  if (!P->tseq) { P->tseq_cap = 1024; FCK(dalloc_t(P, &P->tseq, (size_t)P->tseq_cap, false)); }
  P->capB = B; P->capL = L; P->capRep = repc;
  return FDM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// the step program: one denoiser pass, ws.x (+ its operand copy) -> ws.x0, or -> x_{t-1} when the scheduler update is fused
// ---------------------------------------------------------------------------------------------------------------------
int record_chain(fdm_plan* P, const fdm_sched_args* sched, void* stream) {
  const fdm_model_desc& m = P->m;
  const int d = m.d, M = P->M, R = P->R, L = P->L;
  const bool both = P->dtype != FDM_F32, split = is_split(P->dtype), fuse = P->fuse_ln3;
  int* step = P->step; int* tcur = P->step + 1;
  const float* none = nullptr; (void)none;
  const float *b = nullptr;
  {
    // latent encoder: h = act(x W^T + b) + E0.  A CFG plan's cond and uncond rows read the same x rows and differ only in
    // the addend, so both halves are one batched launch (batch index = half: A / W / bias strides 0, outputs and the
    // addend advance by M rows).  First kernel of the step: the step counter += 1 and tcur = tseq[counter].
    FCK(need(P, "latent_encoder.0.bias", d, &b));
    fdm_gemm_args g = gemm_op(P, P->xt, P->wt["latent_encoder.0.weight"], M, d, d);
    g.bias = b; g.act = m.latent_mish ? FDM_ACT_MISH : FDM_ACT_NONE;
    g.resid = P->E0; g.out_f32 = P->h;
    if (both) set_out_t(g, P->ht);
    g.batch = P->rep; g.out_batch_stride = (long long)M * d;
    g.incr_counter = step; g.incr_table = P->tseq;
    FCK(plan_gemm(P, "enc", g, stream));
  }
  const int BB = P->B * P->rep;
  const float eps = 1e-5f;
  const int np = d / 64;
  for (int l = 0; l < m.n_layers; ++l) {
    const Fold* f = (fuse && l > 0) ? &P->fold[l] : nullptr;
    fdm_gemm_args g;
    if (!f) {
      FCK(need(P, lname(l, "self_attn.in_proj_bias"), 3LL * d, &b));
      g = gemm_op(P, P->ht, P->wt[lname(l, "self_attn.in_proj_weight")], R, 3 * d, d);
      g.bias = b;
    } else {          // layer input = LN3(x2) of the previous layer, never materialised
      g = gemm_op(P, P->x2t, f->w, R, 3 * d, d);
      g.bias = f->bias; g.ln_stat_in = P->stats; g.ln_nparts = np; g.ln_dim = d; g.ln_eps = eps; g.ln_colsum = f->colsum;
    }
    // FDM_F16X3: split attention on plane pairs (head_dim 64 / 128 hold a key tile's fragments in registers, 256 -- BIWI --
    // streams them at one wave per SIMD).
    g.out_t = P->q; g.ldo_t = d; g.out_t_lo_off = split ? P->q_lo : 0;
    g.kv_lo_off = split ? P->kv_lo : 0;
    g.out_kp = P->kp; g.kp_col0 = d; g.out_vp = P->vp; g.vp_col0 = 2 * d; g.kv_L = L; g.kv_Lpad = P->Lpad; g.kv_hd = P->hd;
    FCK(plan_gemm(P, f ? "qkv_ln" : "qkv", g, stream));
    fdm_attn_args at;
    memset(&at, 0, sizeof(at));
    at.Q = P->q; at.ldq = d;
    at.Kp = P->kp; at.Vp = P->vp; at.Lpad = P->Lpad; at.O = P->ctx.p; at.ldo = d;
    at.B = BB; at.H = m.n_head; at.L = L; at.hd = P->hd; at.dtype = P->dtype;
    at.scale = 1.0f / std::sqrt((float)P->hd); at.causal = 1; at.slopes = P->slopes; at.period = m.period;
    if (split) { at.q_lo_off = P->q_lo; at.kv_lo_off = P->kv_lo; at.o_lo_off = P->ctx.lo; }
    FCK(fdm_op_attention(&at, stream));
    FCK(need(P, lname(l, "self_attn.out_proj.bias"), d, &b));
    g = gemm_op(P, P->ctx, P->wt[lname(l, "self_attn.out_proj.weight")], R, d, d);
    g.bias = b; g.out_f32 = P->x1;
    if (!f) {
      g.resid = P->h;
      if (P->ksplit_out > 1) { g.ksplit = P->ksplit_out; g.ksplit_stride = (long long)R * d; }
      FCK(plan_gemm(P, "out", g, stream));
    } else {
      g.resid = P->x2; g.ln_stat_in = P->stats; g.ln_nparts = np; g.ln_dim = d; g.ln_eps = eps; g.rln_gamma = f->gamma; g.rln_beta = f->beta;
      FCK(plan_gemm(P, "out_ln", g, stream));
    }
    // norm1 and norm2 back to back in one kernel: h2 = LN2(LN1(x1) + C1_l + TT_l[t])
    fdm_ln_args ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = P->x1; ln.M = R; ln.d = d; ln.add_mat = P->C1[l]; ln.add_tab = P->TT[l]; ln.tab_step = tcur; ln.eps = eps;
    if (P->S > 1) { ln.add_mat_L = L; ln.add_mat_group = P->S * L; ln.add_mat_wrap = M; }     // C1_l holds one block per audio clip
    if (!f && P->ksplit_out > 1) { ln.x_planes = P->ksplit_out; ln.x_plane_stride = (long long)R * d; }
    FCK(need(P, lname(l, "norm1.weight"), d, &ln.gamma)); FCK(need(P, lname(l, "norm1.bias"), d, &ln.beta));
    FCK(need(P, lname(l, "norm2.weight"), d, &ln.gamma2)); FCK(need(P, lname(l, "norm2.bias"), d, &ln.beta2));
    ln.y_f32 = P->h2; ln.dtype = P->dtype;
    if (both) { ln.y_t = P->h2t.p; ln.y_t_lo_off = P->h2t.lo; }
    FCK(fdm_op_layernorm(&ln, stream));
    FCK(need(P, lname(l, "linear1.bias"), m.ffn, &b));
    g = gemm_op(P, P->h2t, P->wt[lname(l, "linear1.weight")], R, m.ffn, d);
    g.bias = b; g.act = FDM_ACT_RELU; set_out_t(g, P->u);
    FCK(plan_gemm(P, "ffn1", g, stream));
    FCK(need(P, lname(l, "linear2.bias"), d, &b));
    g = gemm_op(P, P->u, P->wt[lname(l, "linear2.weight")], R, d, m.ffn);
    g.bias = b; g.resid = P->h2;
    if (fuse) {      // x2 = h2 + FFN(h2): fp32 + operand copy + per-row partial sums for the folded norm3
      g.out_f32 = P->x2; set_out_t(g, P->x2t); g.stat_out = P->stats;
      FCK(plan_gemm(P, "ffn2_stat", g, stream));
    } else {
      g.out_f32 = P->x1;
      if (P->ksplit_ffn2 > 1) { g.ksplit = P->ksplit_ffn2; g.ksplit_stride = (long long)R * d; }
      FCK(plan_gemm(P, "ffn2", g, stream));
      memset(&ln, 0, sizeof(ln));
      ln.x = P->x1; ln.M = R; ln.d = d; ln.eps = eps; ln.y_f32 = P->h; ln.dtype = P->dtype;
      if (P->ksplit_ffn2 > 1) { ln.x_planes = P->ksplit_ffn2; ln.x_plane_stride = (long long)R * d; }
      FCK(need(P, lname(l, "norm3.weight"), d, &ln.gamma)); FCK(need(P, lname(l, "norm3.bias"), d, &ln.beta));
      if (both) { ln.y_t = P->ht.p; ln.y_t_lo_off = P->ht.lo; }
      FCK(fdm_op_layernorm(&ln, stream));
    }
  }
  fdm_gemm_args g;
  if (fuse) {
    const Fold& f = P->fold[-1];
    g = gemm_op(P, P->x2t, f.w, R, d, d);
    g.bias = f.bias; g.ln_stat_in = P->stats; g.ln_nparts = np; g.ln_dim = d; g.ln_eps = eps; g.ln_colsum = f.colsum;
  } else {
    FCK(need(P, "latent_decoder.bias", d, &b));
    g = gemm_op(P, P->ht, P->wt["latent_decoder.weight"], R, d, d);
    g.bias = b;
  }
  if (sched) {       // x_{t-1} = update(x0_hat = this GEMM, x_t = ws.x) in the epilogue
    g.resid = P->x; g.out_f32 = P->x;
    if (both) set_out_t(g, P->xt);
    g.sched_fuse = 1; g.sched = *sched;
  } else {
    g.out_f32 = P->x0;
  }
  return plan_gemm(P, fuse ? "dec_ln" : "dec", g, stream);
}

struct ProgSpec {
  int kind;                  // 0 pass (denoiser only, + CFG mix), 1 DDPM, 2 DDIM
  const float* noise = nullptr; float cfg_scale = 0.f;
  const float* san = nullptr; const float* cn = nullptr;
  int reps = 1;              // diffusion steps recorded back to back (one graph launch runs them all)
};

// Recorded programs are kept per (shape, tile set, program kind): a serving loop that alternates between shapes (clips of
// different lengths, one clip vs a batch) finds its graphs again instead of re-recording and re-instantiating them (~3 ms per
// switch).  What a program captured stays valid until the workspaces are re-allocated or the weights / tables change: those
// events drop every program.
std::string tiles_sig(const fdm_plan* P) {
  std::string s;
  for (auto& kv : P->tiles)
    if (kv.second) s += kv.first + "=" + std::to_string(kv.second) + ",";
  s += "ks" + std::to_string(P->ksplit_out) + std::to_string(P->ksplit_ffn2) + (P->lockstep ? "L" : "");
  return s;
}

int get_program(fdm_plan* P, const ProgSpec& sp, void* stream, fdm_prog** out) {
  char key[512];
  snprintf(key, sizeof(key), "%s#%s|%d|%p|%a|%p|%d", shape_key(P).c_str(), tiles_sig(P).c_str(), sp.kind, (const void*)sp.noise, (double)sp.cfg_scale,
           (const void*)sp.san, sp.reps);
  auto it = P->progs.find(key);
  if (it != P->progs.end()) {            // hit: most recently used goes to the back, and the caller's handle stays valid for this call
    auto pos = std::find(P->prog_order.begin(), P->prog_order.end(), std::string(key));
    if (pos != P->prog_order.end()) { P->prog_order.erase(pos); P->prog_order.push_back(key); }
    P->pinned.push_back(key);
    if (sp.reps == 1) P->launches_per_step = fdm_prog_num_ops(it->second);
    *out = it->second;
    return FDM_OK;
  }
  if (P->progs.size() >= 16) {
    // programs are keyed by the pointers they captured (e.g. injected noise): cap the cache.  Victim = least recently used
    // program that was NOT handed out earlier in this API call (fdm_sample_graph holds two: the 1-step and the K-step one).
    auto victim = P->prog_order.end();
    for (auto v = P->prog_order.begin(); v != P->prog_order.end(); ++v)
      if (std::find(P->pinned.begin(), P->pinned.end(), *v) == P->pinned.end()) { victim = v; break; }
    if (victim != P->prog_order.end()) {
      HIPCK(hipStreamSynchronize((hipStream_t)stream));
      const std::string vk = *victim;
      P->prog_order.erase(victim);
      fdm_prog_destroy(P->progs[vk]);
      P->progs.erase(vk);
    }
  }
  const int d = P->m.d, M = P->M;
  const long long n = (long long)M * d;
  fdm_prog* prog = nullptr;
  FCK(fdm_prog_create(&prog));
  int rc = fdm_prog_begin(prog);
  const bool fuse_sched = !P->cfg && sp.kind != 0;
  for (int rep = 0; rc == FDM_OK && rep < sp.reps; ++rep) {
    fdm_sched_args sc;
    memset(&sc, 0, sizeof(sc));
    sc.x0 = P->x0; sc.x0u = P->cfg ? P->x0 + n : nullptr; sc.cfg_scale = sp.cfg_scale;
    sc.x = P->x; sc.x_out = P->x; sc.n = n; sc.tseq = P->tseq; sc.step = P->step; sc.advance = 0;
    if (P->dtype != FDM_F32) { sc.x_out_t = P->xt.p; sc.out_dtype = P->dtype; sc.x_out_t_lo_off = P->xt.lo; }
    if (sp.kind == 1) {
      sc.mode = 0; sc.n_per_clip = (long long)P->L * d; sc.c1 = P->c1; sc.c2 = P->c2; sc.sigma = P->sigma;
      sc.noise = sp.noise; sc.noise_stride = n; sc.seed_dev = P->seedbuf;
    } else if (sp.kind == 2) {
      sc.mode = 1; sc.sra = P->sra; sc.srm1 = P->srm1; sc.sqrt_an = sp.san; sc.c_n = sp.cn;
    }
    if (sp.kind != 0 && fuse_sched) {
      // non-CFG samplers: the update runs in the latent decoder GEMM's epilogue (bit-identical, one launch less)
      fdm_sched_args fs2 = sc;
      fs2.x0 = fs2.x0u = fs2.x = nullptr; fs2.x_out = nullptr; fs2.x_out_t = nullptr;
      rc = record_chain(P, &fs2, stream);
      continue;
    }
    rc = record_chain(P, nullptr, stream);
    if (rc != FDM_OK) break;
    if (sp.kind != 0) {
      rc = fdm_op_sched_step(&sc, stream);
    } else if (P->cfg) {
      sc.mode = 2; sc.x = nullptr; sc.x_out = P->x0; sc.x_out_t = nullptr;
      rc = fdm_op_sched_step(&sc, stream);
    }
  }
  const int rc2 = fdm_prog_end(prog);
  if (rc != FDM_OK || rc2 != FDM_OK) { fdm_prog_destroy(prog); return rc != FDM_OK ? rc : rc2; }
  if (sp.reps == 1) P->launches_per_step = fdm_prog_num_ops(prog);
  P->progs[key] = prog;
  P->prog_order.push_back(key);
  P->pinned.push_back(key);
  *out = prog;
  return FDM_OK;
}

int set_steps(fdm_plan* P, const int* ts, int n, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n > P->tseq_cap) {
    FCK(drop_programs(P, stream));          // (drains the stream: nothing reads the old list any more)
    auto old = std::find(P->allocs.begin(), P->allocs.end(), (void*)P->tseq);
    if (old != P->allocs.end()) { P->allocs.erase(old); (void)hipFree(P->tseq); }
    P->tseq_cap = n;
    FCK(dalloc_t(P, &P->tseq, (size_t)n, false));
  }
  HIPCK(hipMemcpyAsync(P->tseq, ts, (size_t)n * 4, hipMemcpyHostToDevice, s));
  const int init[2] = {-1, 0};       // the first GEMM of every step increments the counter before anything reads it
  HIPCK(hipMemcpyAsync(P->step, init, 8, hipMemcpyHostToDevice, s));
  HIPCK(hipStreamSynchronize(s));    // ts / init are caller / stack memory
  return FDM_OK;
}

int load_x(fdm_plan* P, const float* x, void* stream) {
  const long long n = (long long)P->M * P->m.d;
  HIPCK(hipMemcpyAsync(P->x, x, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (P->dtype != FDM_F32) {
    // (the plane distance of ws.xt is its capacity, not n: cast plane by plane through the op's contract lo = dst + n only
    //  when they coincide; otherwise use the scheduler's store path with identity coefficients -- simpler: a strided cast)
    if (!is_split(P->dtype) || P->xt.lo == n) return fdm_op_cast(P->x, P->xt.p, n, P->dtype, stream);
    // capacity > current shape: run the mix-only scheduler mode (x_out = x0), which writes the operand copy with any plane distance
    fdm_sched_args sc;
    memset(&sc, 0, sizeof(sc));
    sc.mode = 2; sc.x0 = P->x; sc.x_out = P->x; sc.n = n; sc.x_out_t = P->xt.p; sc.out_dtype = P->dtype; sc.x_out_t_lo_off = P->xt.lo;
    return fdm_op_sched_step(&sc, stream);
  }
  return FDM_OK;
}

int check_ready(const fdm_plan* P) {
  if (!P) return fail(FDM_ERR_ARG, "plan: null plan");
  if (!P->prepared) return fail(FDM_ERR_STATE, "plan: call fdm_audio_prepare first");
  return FDM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// plan-time tile tuning
// ---------------------------------------------------------------------------------------------------------------------
int time_prog(fdm_prog* prog, int warm, int reps, hipStream_t s, float* ms) {
  hipEvent_t e0, e1;
  HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  int rc = fdm_prog_instantiate(prog, s);
  if (rc == FDM_OK) rc = fdm_prog_replay(prog, warm, s);
  if (rc == FDM_OK) {
    (void)hipEventRecord(e0, s);
    rc = fdm_prog_replay(prog, reps, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(ms, e0, e1);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

void apply_tile_override(std::map<std::string, int>& tiles) {      // FDM_TILE_OVERRIDE="qkv_ln=5,ffn1=3": force call sites (experiments, pinned profiles)
  const char* ov = getenv("FDM_TILE_OVERRIDE");
  if (!ov) return;
  const std::string sov(ov);
  size_t pos = 0;
  while (pos < sov.size()) {
    const size_t comma = sov.find(',', pos), eq = sov.find('=', pos);
    const size_t end = comma == std::string::npos ? sov.size() : comma;
    if (eq != std::string::npos && eq < end) tiles[sov.substr(pos, eq - pos)] = atoi(sov.substr(eq + 1, end - eq - 1).c_str());
    pos = end + 1;
  }
}

// ---- tuned tiles kept across processes (opt-in: FDM_TILE_CACHE=<file>).  One text line per (library version, arithmetic mode,
// model geometry, shape): "<key>\t<site>=<tile>,...".  A plan whose shape is in the file takes the stored set at
// fdm_audio_prepare without a single timing launch; fdm_plan_tune / the serving-time tuning of fdm_audio_prepare add their
// result.  Written through a temporary file + rename: concurrent ranks may lose each other's additions, never corrupt the file.
std::string store_key(const fdm_plan* P) {
  const fdm_model_desc& m = P->m;
  char b[160];
  snprintf(b, sizeof(b), "v%d|dt%d|%d,%d,%d,%d,%d,%d,%d,%d|%s", fdm_version(), P->dtype, m.d, m.n_head, m.n_layers, m.ffn, m.G, m.c, m.audio_in, m.pair,
           shape_key(P).c_str());
  return b;
}

bool store_read(std::map<std::string, std::string>& lines) {
  const char* path = getenv("FDM_TILE_CACHE");
  if (!path || !*path) return false;
  std::ifstream f(path);
  std::string ln;
  while (f && std::getline(f, ln)) {
    const size_t tab = ln.find('\t');
    if (tab != std::string::npos) lines[ln.substr(0, tab)] = ln.substr(tab + 1);
  }
  return true;
}

bool store_lookup(const fdm_plan* P, std::map<std::string, int>& tiles) {
  std::map<std::string, std::string> lines;
  if (!store_read(lines)) return false;
  auto it = lines.find(store_key(P));
  if (it == lines.end()) return false;
  tiles.clear();
  const std::string& v = it->second;
  size_t pos = 0;
  while (pos < v.size()) {
    const size_t comma = v.find(',', pos), eq = v.find('=', pos);
    const size_t end = comma == std::string::npos ? v.size() : comma;
    if (eq != std::string::npos && eq < end) {
      const int t = atoi(v.substr(eq + 1, end - eq - 1).c_str());
      if (t < 0 || t > FDM_TILE_MAX) return false;      // a damaged line: tune again
      if (t) tiles[v.substr(pos, eq - pos)] = t;
    }
    pos = end + 1;
  }
  return true;
}

void store_save(const fdm_plan* P, const std::map<std::string, int>& tiles) {
  std::map<std::string, std::string> lines;
  if (!store_read(lines)) return;
  std::string v;
  for (auto& kv : tiles)
    if (kv.second) v += (v.empty() ? "" : ",") + kv.first + "=" + std::to_string(kv.second);
  lines[store_key(P)] = v;
  const std::string path = getenv("FDM_TILE_CACHE"), tmp = path + ".tmp" + std::to_string((long long)getpid());
  {
    std::ofstream f(tmp, std::ios::trunc);
    if (!f) return;                                     // an unwritable location disables the store, never the plan
    for (auto& kv : lines) f << kv.first << '\t' << kv.second << '\n';
  }
  if (rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

int tune_tiles_impl(fdm_plan* P, void* stream);

// Time the candidate output tiles of every GEMM call site of the step at this plan's shapes and keep the fastest.  Cached per
// shape.  Tuning is PLAN-TIME work (it records, instantiates and times dozens of graphs with stream drains in between):
//   force != 0          fdm_plan_tune
//   force == 0          fdm_audio_prepare, for a shape that earlier sampling calls have run >= 2000 steps at (serving)
// fdm_sample_graph itself only counts steps per shape, unless the caller opted in to in-call tuning
// (fdm_plan_set(p, "tune_lazy", 1)).  Every tile accumulates k in the same order, so the choice changes speed only, never results.
int tune_tiles(fdm_plan* P, int force, void* stream) {
  const std::string key = shape_key(P);
  const char* env = getenv("FDM_TUNE");
  if (P->tile_cache.count(key)) return FDM_OK;
  if (!P->tune_enabled || (env && !strcmp(env, "0"))) return FDM_OK;       // heuristic tiles, or the pinned set of FDM_TILE_OVERRIDE
  if (!force && (P->steps_seen[key] < 2000 || P->tune_failed_shapes.count(key))) return FDM_OK;     // (a failed request-path tune is not repeated per request)
  const std::map<std::string, int> before = P->tiles;
  const int rc = tune_tiles_impl(P, stream);
  if (rc != FDM_OK) { P->tiles = before; (void)drop_programs(P, stream); if (!force) P->tune_failed_shapes.insert(key); }   // never leave a trial set behind
  else { store_save(P, P->tiles); P->tune_failed_shapes.erase(key); }
  return rc;
}

// opt-in tuning on a request path: a failure is remembered (fdm_plan_get "tune_failed"), never returned
void tune_soft(fdm_plan* P, void* stream) {
  if (tune_tiles(P, 0, stream) != FDM_OK) ++P->tune_failed;
}

int tune_tiles_impl(fdm_plan* P, void* stream) {
  const std::string key = shape_key(P);
  hipStream_t s = (hipStream_t)stream;
  FCK(drop_programs(P, stream));
  P->tiles.clear();
  std::map<std::string, std::vector<fdm_gemm_args>> calls;
  {      // dry recording of one pass: captures each call site's arguments, runs nothing
    fdm_prog* dry = nullptr;
    FCK(fdm_prog_create(&dry));
    int rc = fdm_prog_begin(dry);
    P->tune_rec = &calls;
    if (rc == FDM_OK) rc = record_chain(P, nullptr, stream);
    P->tune_rec = nullptr;
    (void)fdm_prog_end(dry);
    fdm_prog_destroy(dry);
    FCK(rc);
  }
  auto timed = [&](const std::vector<fdm_gemm_args>& inst, int tile, float* best) -> int {
    *best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      fdm_prog* prog = nullptr;
      FCK(fdm_prog_create(&prog));
      int rc = fdm_prog_begin(prog);
      for (size_t i = 0; rc == FDM_OK && i < inst.size(); ++i) { fdm_gemm_args a = inst[i]; a.tile = tile | (P->lockstep ? FDM_TILE_LOCKSTEP : 0); a.incr_counter = nullptr; a.incr_table = nullptr; rc = fdm_op_gemm(&a, stream); }
      (void)fdm_prog_end(prog);
      float ms = 0.f;
      if (rc == FDM_OK) rc = time_prog(prog, 2, 5, s, &ms);
      HIPCK(hipStreamSynchronize(s));
      fdm_prog_destroy(prog);
      FCK(rc);
      if (ms < *best) *best = ms;
    }
    return FDM_OK;
  };
  std::vector<int> cands = {FDM_TILE_64x64, FDM_TILE_64x64_S2, FDM_TILE_32x64_S3, FDM_TILE_128x64, FDM_TILE_128x128, FDM_TILE_80x128, FDM_TILE_64x128};
  if (!is_split(P->dtype) && P->R >= 1024) cands.push_back(FDM_TILE_256x128_PP);
  // Which of them are worth a stopwatch is decided by a wave-quantisation model first.  What bounds these GEMMs is the
  // bytes a CU pulls from L2 into LDS: a BM x BN tile costs (BM + BN) * K * bytes-per-element (x planes)
  // and the busiest CU runs ceil(tiles / 256) of them (the whole grid is resident, or queued behind, at <= 160 KB / ring per CU),
  // on top of a fixed cost per kernel (~5 us: boundary, ring fill, epilogue; in-situ timings of profiles/README.md fit
  // 5 us + bytes / 70 GB/s within ~15 % for tiles up to 128x64; larger tiles run above the model).  Candidates modelled
  // more than 35 % above the best one are not timed.
  struct Geo { int bm, bn, nst; };
  auto geo = [&](int tile) -> Geo {
    const bool sp = is_split(P->dtype);
    switch (tile) {
      case FDM_TILE_64x64_S2: return {64, 64, 2};
      case FDM_TILE_32x64_S3: return {32, 64, 3};
      case FDM_TILE_128x64: return {128, 64, sp ? 3 : 4};
      case FDM_TILE_128x128: return {128, 128, sp ? 2 : 3};
      case FDM_TILE_80x128: return {80, 128, sp ? 3 : 4};
      case FDM_TILE_64x128: return {64, 128, sp ? 3 : 4};
      case FDM_TILE_256x128_PP: return {256, 128, 3};
      default: return {64, 64, 4};
    }
  };
  auto modelled_us = [&](const fdm_gemm_args& a, int tile) {
    const Geo g = geo(tile);
    const double planes = is_split(P->dtype) ? 2.0 : 1.0, eb = P->dtype == FDM_F32 ? 4.0 : 2.0;
    const long long tiles = (long long)((a.M + g.bm - 1) / g.bm) * ((a.N + g.bn - 1) / g.bn) * (a.batch > 0 ? a.batch : 1);
    const long long rounds = (tiles + 255) / 256;                                   // tiles on the busiest CU
    const double bytes = (double)rounds * (g.bm + g.bn) * a.K * eb * planes;
    return 5.0 + bytes / 70e3;          // us
  };
  std::map<std::string, int> tuned, runner_up;
  for (auto& kv : calls) {
    std::vector<fdm_gemm_args> inst = kv.second;
    while (inst.size() < 4) { auto c = inst; inst.insert(inst.end(), c.begin(), c.end()); }
    float base = 0.f;
    FCK(timed(inst, 0, &base));
    std::vector<std::pair<float, int>> cand = {{base * 0.97f, 0}};          // switch only for a > 3 % gain over the heuristic
    double best_model = 1e30;
    for (int tile : cands) best_model = std::min(best_model, modelled_us(kv.second[0], tile));
    // the heuristic's own tile is `base`: timing it again as a candidate only lets noise "pick" it (the split kinds alias
    // several ids to one kernel: compare what the ids launch)
    auto launched = [&](int tile) {
      if (!is_split(P->dtype)) return tile;
      return tile == FDM_TILE_256x128_PP ? FDM_TILE_128x128 : tile;
    };
    const int heur = launched(fdm_gemm_heuristic_tile(&kv.second[0]));
    for (int tile : cands) {
      if (launched(tile) == heur) continue;
      if (kv.second[0].ksplit > 1 && tile != FDM_TILE_64x64 && tile != FDM_TILE_64x64_S2 && tile != FDM_TILE_32x64_S3) continue;   // K-sliced sites
      if (kv.second[0].sched_fuse && tile != FDM_TILE_256x128_PP && tile != FDM_TILE_64x64) continue;      // the scheduler-fused decoder has two forms: 64x64 and the ping-pong tile
      if (modelled_us(kv.second[0], tile) > 1.35 * best_model) continue;
      float t = 0.f;
      FCK(timed(inst, tile, &t));
      cand.push_back({t, tile});
    }
    std::sort(cand.begin(), cand.end());
    tuned[kv.first] = cand[0].second;
    if (cand.size() > 1 && cand[1].first < cand[0].first * 1.05f) runner_up[kv.first] = cand[1].second;   // settled inside the chain below
  }
  // the isolated timings can mislead (cache state inside the step differs): keep the tuned set only if one whole denoiser
  // pass is faster with it than with the heuristic
  auto chain_time = [&](const std::map<std::string, int>& tiles, float* best) -> int {
    *best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      P->tiles = tiles;
      const int init[2] = {-1, 0};
      HIPCK(hipMemcpyAsync(P->step, init, 8, hipMemcpyHostToDevice, s));
      fdm_prog* prog = nullptr;
      FCK(fdm_prog_create(&prog));
      int rc = fdm_prog_begin(prog);
      if (rc == FDM_OK) rc = record_chain(P, nullptr, stream);
      (void)fdm_prog_end(prog);
      float ms = 0.f;
      if (rc == FDM_OK) rc = time_prog(prog, 2, 4, s, &ms);
      HIPCK(hipStreamSynchronize(s));
      fdm_prog_destroy(prog);
      FCK(rc);
      if (ms < *best) *best = ms;
    }
    return FDM_OK;
  };
  bool any = !runner_up.empty();
  for (auto& kv : tuned) any = any || kv.second != 0;
  std::map<std::string, int> keep;
  if (any) {
    float t_h = 0.f, t_t = 0.f;
    FCK(chain_time({}, &t_h));
    FCK(chain_time(tuned, &t_t));
    for (auto& kv : runner_up) {            // close calls: try the runner-up in place, keep what the chain prefers
      std::map<std::string, int> trial = tuned;
      trial[kv.first] = kv.second;
      float t_a = 0.f;
      FCK(chain_time(trial, &t_a));
      if (t_a < 0.997f * t_t) { tuned = trial; t_t = t_a; }
    }
    if (t_t < 0.99f * t_h) {                    // (below 1 % the chain timing's own spread decides: keep the heuristic set)
      float t_h2 = 0.f;                         // the heuristic chain once more, AFTER the trials: a slow first measurement
      FCK(chain_time({}, &t_h2));               // (clock ramp, cold caches) must not make a neutral set look faster
      t_h = std::min(t_h, t_h2);
    }
    if (t_t < 0.99f * t_h) keep = tuned;
    if (getenv("FDM_TUNE_VERBOSE")) {
      std::string desc;
      for (auto& kv : tuned) desc += kv.first + "=" + std::to_string(kv.second) + ",";
      fprintf(stderr, "[fdm tune] rows=%d candidates: %s chain %.3f -> %.3f ms: %s\n", P->R, desc.c_str(), t_h / 4, t_t / 4, keep.empty() ? "rejected" : "kept");
    }
  }
  P->tiles = keep;
  apply_tile_override(P->tiles);
  P->tile_cache[key] = P->tiles;
  return FDM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

int fdm_plan_create(const fdm_model_desc* desc, int B, int L, int cfg, int dtype, fdm_plan** out) {
  if (!desc || !out) return fail(FDM_ERR_ARG, "plan_create: null argument");
  if (dtype < FDM_F32 || dtype > FDM_F16) return fail(FDM_ERR_ARG, "plan_create: bad dtype %d", dtype);
  const fdm_model_desc& m = *desc;
  if (m.d <= 0 || m.n_head <= 0 || m.d % m.n_head || m.n_layers <= 0 || m.ffn <= 0 || m.G * m.c != m.d || m.pair <= 0 || m.max_len <= 0)
    return fail(FDM_ERR_SHAPE, "plan_create: inconsistent model geometry (d %d, heads %d, G*c %d)", m.d, m.n_head, m.G * m.c);
  const int hd = m.d / m.n_head;
  if (hd != 64 && hd != 128 && hd != 256) return fail(FDM_ERR_SHAPE, "plan_create: head_dim %d unsupported (64, 128, 256)", hd);
  if (m.d != 256 && m.d != 512 && m.d != 768 && m.d != 1024) return fail(FDM_ERR_SHAPE, "plan_create: feature_dim %d unsupported (256, 512, 768, 1024)", m.d);
  if (B < 1 || L < 1 || L > m.max_len) return fail(FDM_ERR_SHAPE, "plan_create: B=%d, L=%d outside [1, .] x [1, %d] (models/fdm_vocaset.py:44)", B, L, m.max_len);
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "plan_create: no gfx950 device visible (there is no CPU fallback)");
  fdm_plan* P = new (std::nothrow) fdm_plan();
  if (!P) return fail(FDM_ERR_STATE, "plan_create: out of memory");
  P->m = m; P->dtype = dtype; P->hd = hd;
  const int rc = reserve(P, B, L, cfg);
  if (rc != FDM_OK) { fdm_plan_destroy(P); return rc; }
  *out = P;
  return FDM_OK;
}

int fdm_plan_reserve(fdm_plan* P, int B, int L, int cfg) {
  if (!P) return fail(FDM_ERR_ARG, "plan_reserve: null plan");
  if (B < 1 || L < 1 || L > P->m.max_len) return fail(FDM_ERR_SHAPE, "plan_reserve: B=%d, L=%d outside [1, .] x [1, %d]", B, L, P->m.max_len);
  return reserve(P, B, L, cfg);
}

int fdm_plan_destroy(fdm_plan* P) {
  if (!P) return FDM_OK;
  (void)hipDeviceSynchronize();
  for (auto& kv : P->progs) fdm_prog_destroy(kv.second);
  for (void* p : P->ws_allocs) (void)hipFree(p);
  for (void* p : P->commit_allocs) (void)hipFree(p);
  for (void* p : P->allocs) (void)hipFree(p);
  delete P;
  return FDM_OK;
}

int fdm_plan_set_weights(fdm_plan* P, const char* name, const float* ptr, long long n, void* stream) {
  if (!P || !name || !ptr || n <= 0) return fail(FDM_ERR_ARG, "plan_set_weights: bad argument");
  // a weight changed under the derived tables (and under recorded programs that point at the fp32 masters): release them,
  // the next prepare / commit rebuilds
  if (P->committed || !P->commit_allocs.empty()) FCK(release_commit(P, stream));
  Wt& w = P->w[name];
  if (w.n != n) {
    if (w.p) {                               // size change: the old tensor goes (nothing references it after release_commit)
      FCK(drop_programs(P, stream));
      HIPCK(hipStreamSynchronize((hipStream_t)stream));
      auto it = std::find(P->allocs.begin(), P->allocs.end(), (void*)w.p);
      if (it != P->allocs.end()) P->allocs.erase(it);
      (void)hipFree(w.p);
      w.p = nullptr;
    }
    w.n = n;
    FCK(dalloc_t(P, &w.p, (size_t)n, false));
  }
  HIPCK(hipMemcpyAsync(w.p, ptr, (size_t)n * 4, hipMemcpyDefault, (hipStream_t)stream));
  return FDM_OK;
}

int fdm_plan_commit(fdm_plan* P, void* stream) {
  if (!P) return fail(FDM_ERR_ARG, "plan_commit: null plan");
  return commit(P, stream);
}

int fdm_audio_prepare(fdm_plan* P, const float* hub, int B, int N, int fw, const float* style, const float* emo, int L, int cfg, void* stream) {
  return fdm_audio_prepare_conds(P, hub, B, N, fw, 1, style, emo, L, cfg, stream);
}

int fdm_audio_prepare_conds(fdm_plan* P, const float* hub, int B0, int N, int fw, int S, const float* style, const float* emo, int L, int cfg, void* stream) {
  if (!P || !hub || !style) return fail(FDM_ERR_ARG, "audio_prepare: null argument");
  const fdm_model_desc& m = P->m;
  if (B0 < 1 || N < 1 || fw < 1) return fail(FDM_ERR_SHAPE, "audio_prepare: bad feature shape [%d, %d, %d]", B0, N, fw);
  if (S < 1) return fail(FDM_ERR_SHAPE, "audio_prepare: S=%d conditions per clip", S);
  if (m.pair * fw != m.audio_in) return fail(FDM_ERR_SHAPE, "audio_prepare: audio feature width %d x pair %d != audio_extract input %d", fw, m.pair, m.audio_in);
  if (L < 1 || L > N / m.pair || L > m.max_len) return fail(FDM_ERR_SHAPE, "audio_prepare: latent frames L=%d outside [1, min(%d, %d)] (models/fdm_vocaset.py:44,64-66)", L, N / m.pair, m.max_len);
  if (m.n_emo && !emo) return fail(FDM_ERR_ARG, "audio_prepare: this model needs an emotion one-hot");
  const int B = B0 * S;                      // row blocks of the step program
  FCK(commit(P, stream));
  FCK(reserve(P, B, L, cfg));
  // recorded programs hold the workspace pointers and the shape, not the clip tables' contents: a new batch of a shape seen
  // before (serving) finds them and their instantiated graphs again (get_program keys them by shape and tile set)
  hipStream_t s = (hipStream_t)stream;
  const int d = m.d, M0 = B0 * L, M = B * L, rep = cfg ? 2 : 1;
  P->B = B; P->S = S; P->L = L; P->M = M; P->rep = rep; P->R = M * rep; P->cfg = cfg ? 1 : 0; P->Lpad = (L + 31) / 32 * 32;
  // pad keys of the packed K / V buffers must be finite: the layout depends on (L, Lpad), so clear them per shape
  HIPCK(hipMemsetAsync(P->kp, 0, P->kv_bytes, s));
  HIPCK(hipMemsetAsync(P->vp, 0, P->kv_bytes, s));
  const float *w0 = nullptr, *b0 = nullptr, *w2 = nullptr, *b2 = nullptr;
  FCK(need(P, "audio_extract.0.weight", (long long)d * m.audio_in, &w0)); FCK(need(P, "audio_extract.0.bias", d, &b0));
  FCK(need(P, "audio_extract.2.weight", (long long)d * d, &w2)); FCK(need(P, "audio_extract.2.bias", d, &b2));
  // audio rows: `pair` consecutive encoder frames per latent frame (models/fdm_vqvae_mead.py:73), cropped to L (:64-66);
  // the rows of a clip are contiguous in hub, so each clip's GEMM reads them in place (fp32: once per clip, parity)
  for (int b = 0; b < B0; ++b) {
    fdm_gemm_args g = gemm_f32(hub + (size_t)b * N * fw, w0, L, d, m.audio_in);
    g.bias = b0; g.act = FDM_ACT_MISH; g.out_f32 = P->t1 + (size_t)b * L * d;
    FCK(fdm_op_gemm(&g, stream));
  }
  fdm_gemm_args g = gemm_f32(P->t1, w2, M0, d, d);
  g.bias = b2; g.out_f32 = P->AF;
  FCK(fdm_op_gemm(&g, stream));
  // folded cross-attention tables C1_l = Wo_l (Wv_l AF + bv_l) + bo_l.  S = 1: layout [rep][M, d] (the uncond half is a copy);
  // S > 1: ONE block of [B0 * L, d] per layer, every condition (and both CFG halves) of a clip reads its clip's rows through
  // the LayerNorm kernel's row map (fdm_ln_args.add_mat_group) -- no table work per condition
  for (int l = 0; l < m.n_layers; ++l) {
    g = gemm_f32(P->AF, P->Wv[l], M0, d, d);
    g.bias = P->bv[l]; g.out_f32 = P->t1;
    FCK(fdm_op_gemm(&g, stream));
    g = gemm_f32(P->t1, P->Wo[l], M0, d, d);
    g.bias = P->bo[l]; g.out_f32 = P->C1[l];
    FCK(fdm_op_gemm(&g, stream));
    if (rep == 2 && S == 1) HIPCK(hipMemcpyAsync(P->C1[l] + (size_t)M * d, P->C1[l], (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
  }
  // conditioning addend E0 = PE[l] + style[b] (+ emotion[b]) (:75-84), one row block per (clip, condition)
  const float *sw = nullptr, *sbias = nullptr;
  FCK(need(P, "style_embedd.weight", (long long)d * m.n_style, &sw)); FCK(need(P, "style_embedd.bias", d, &sbias));
  FCK(fdm_op_small_linear(style, sw, sbias, P->sty, B, m.n_style, d, m.style_mish ? FDM_ACT_MISH : FDM_ACT_NONE, stream));
  const float *ew = nullptr, *eb = nullptr;
  if (m.n_emo) {
    FCK(need(P, "emotion_embedd.weight", (long long)d * m.n_emo, &ew)); FCK(need(P, "emotion_embedd.bias", d, &eb));
    FCK(fdm_op_small_linear(emo, ew, eb, P->em, B, m.n_emo, d, FDM_ACT_NONE, stream));
    // null condition = zeros_like(emotion one-hot) (models/fdm_vqvae_mead.py:56-57) -> bias only
    if (cfg) FCK(fdm_op_small_linear(P->zeros, ew, eb, P->emu, B, m.n_emo, d, FDM_ACT_NONE, stream));
  }
  for (int r = 0; r < rep; ++r) {
    const float* e = m.n_emo ? (r == 1 ? P->emu : P->em) : nullptr;
    FCK(fdm_op_add_rows(P->pe, 1, L, P->sty, L, B, e, L, B, P->E0 + (size_t)r * M * d, M, d, stream));
  }
  P->prepared = true;
  auto it = P->tile_cache.find(shape_key(P));
  std::map<std::string, int> want;
  const char* tune_env = getenv("FDM_TUNE");
  if (it != P->tile_cache.end()) {
    want = it->second;
  } else if (P->tune_enabled && !(tune_env && !strcmp(tune_env, "0")) && store_lookup(P, want)) {
    apply_tile_override(want);             // a set tuned by an earlier process (FDM_TILE_CACHE): counts as tuned
    P->tile_cache[shape_key(P)] = want;
  } else {
    want.clear();
    apply_tile_override(want);             // FDM_TILE_OVERRIDE pins tiles with or without the tuner (heuristic tiles elsewhere)
  }
  P->tiles = want;                          // (programs are keyed by the tile set they were recorded with)
  // A request path never tunes by itself: fdm_plan_tune does, when the caller schedules it (fdm_plan_get "needs_tune" says when a
  // shape has served >= 2000 steps on heuristic tiles).  Opt-in (fdm_plan_set "tune_lazy"): tune here / inside fdm_sample_graph
  // once that is the case -- and even then a tuner failure keeps the heuristic tiles instead of failing the request.
  if (P->tune_lazy) tune_soft(P, stream);
  return FDM_OK;
}

int fdm_denoise_step(fdm_plan* P, const float* x_t, int t, float cfg_scale, float* x0_hat, float* x0_uncond, void* stream) {
  FCK(check_ready(P));
  P->pinned.clear();
  if (!x_t || !x0_hat || t < 0 || t >= 1000) return fail(FDM_ERR_ARG, "denoise_step: bad argument (t = %d)", t);
  FCK(load_x(P, x_t, stream));
  FCK(set_steps(P, &t, 1, stream));
  ProgSpec sp; sp.kind = 0; sp.cfg_scale = cfg_scale;
  fdm_prog* prog = nullptr;
  FCK(get_program(P, sp, stream, &prog));
  FCK(fdm_prog_run(prog, stream));
  const size_t nb = (size_t)P->M * P->m.d * 4;
  HIPCK(hipMemcpyAsync(x0_hat, P->x0, nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (x0_uncond && P->cfg) HIPCK(hipMemcpyAsync(x0_uncond, P->x0 + (size_t)P->M * P->m.d, nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return FDM_OK;
}

int fdm_sample_graph(fdm_plan* P, const fdm_sample_args* a, void* stream) {
  FCK(check_ready(P));
  P->pinned.clear();
  if (!a || !a->x_T || !a->out) return fail(FDM_ERR_ARG, "sample_graph: null argument");
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> ts;
  ProgSpec sp;
  sp.cfg_scale = a->cfg_scale;
  if (a->kind == 0) {
    if (!a->t_list || a->n_steps <= 0) return fail(FDM_ERR_ARG, "sample_graph: DDPM needs t_list / n_steps");
    for (int i = 0; i < a->n_steps; ++i) {
      if (a->t_list[i] < 0 || a->t_list[i] >= 1000) return fail(FDM_ERR_ARG, "sample_graph: timestep %d outside [0, 1000)", a->t_list[i]);
      ts.push_back(a->t_list[i]);
    }
    sp.kind = 1; sp.noise = a->noise;
    const unsigned long long sd[2] = {a->seed, (unsigned long long)(unsigned)a->clip0};     // set_steps() below drains the copy
    HIPCK(hipMemcpyAsync(P->seedbuf, sd, 16, hipMemcpyHostToDevice, s));
  } else if (a->kind == 1) {
    if (a->ddim_steps <= 0) return fail(FDM_ERR_ARG, "sample_graph: DDIM needs ddim_steps");
    if (!P->ddim.count(a->ddim_steps)) {
      std::vector<int> t(a->ddim_steps), tn(a->ddim_steps);
      std::vector<float> tab(2 * (size_t)a->ddim_steps);
      const int n = fdm_ddim_schedule_host(a->ddim_steps, 1000, t.data(), tn.data(), tab.data(), tab.data() + a->ddim_steps);
      if (n < 0) return n;
      float* dv = nullptr;
      FCK(dalloc_t(P, &dv, 2 * (size_t)a->ddim_steps, false));
      HIPCK(hipMemcpyAsync(dv, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
      HIPCK(hipStreamSynchronize(s));
      t.resize(n);
      P->ddim[a->ddim_steps] = {n, dv};
      P->ddim_t[a->ddim_steps] = t;
    }
    ts = P->ddim_t[a->ddim_steps];
    sp.kind = 2; sp.san = P->ddim[a->ddim_steps].second; sp.cn = sp.san + a->ddim_steps;
  } else {
    return fail(FDM_ERR_ARG, "sample_graph: kind %d (0 = DDPM, 1 = DDIM)", a->kind);
  }
  const int n_steps = (int)ts.size();
  const size_t nb = (size_t)P->M * P->m.d * 4;
  P->last_graph_launches = 0;
  if (n_steps == 0) {        // e.g. ddim_steps = 1: only the dead pair
    if (a->out != a->x_T) HIPCK(hipMemcpyAsync(a->out, a->x_T, nb, hipMemcpyDeviceToDevice, s));
    return FDM_OK;
  }
  if (P->tune_lazy) tune_soft(P, stream);
  P->steps_seen[shape_key(P)] += n_steps;
  FCK(load_x(P, a->x_T, stream));
  FCK(set_steps(P, ts.data(), n_steps, stream));
  fdm_prog* p1 = nullptr;
  FCK(get_program(P, sp, stream, &p1));
  if (a->record || a->eager) {
    for (int i = 0; i < n_steps; ++i) {
      if (a->eager) { FCK(fdm_prog_run(p1, stream)); }
      else { FCK(fdm_prog_instantiate(p1, stream)); FCK(fdm_prog_replay(p1, 1, stream)); ++P->last_graph_launches; }
      if (a->record) HIPCK(hipMemcpyAsync(a->record + (size_t)i * P->M * P->m.d, P->x, nb, hipMemcpyDeviceToDevice, s));
    }
  } else {
    int K = a->graph_steps > 0 ? a->graph_steps : 10;
    if (K > n_steps) K = n_steps;
    int left = n_steps;
    if (K > 1) {
      ProgSpec spk = sp; spk.reps = K;
      fdm_prog* pk = nullptr;
      FCK(get_program(P, spk, stream, &pk));
      FCK(fdm_prog_instantiate(pk, stream));
      FCK(fdm_prog_replay(pk, left / K, stream));
      P->last_graph_launches += left / K;
      left %= K;
    }
    if (left) {
      FCK(fdm_prog_instantiate(p1, stream));
      FCK(fdm_prog_replay(p1, left, stream));
      P->last_graph_launches += left;
    }
  }
  HIPCK(hipMemcpyAsync(a->out, P->x, nb, hipMemcpyDeviceToDevice, s));
  return FDM_OK;
}

int fdm_plan_tune(fdm_plan* P, void* stream) {
  FCK(check_ready(P));
  return tune_tiles(P, 1, stream);
}

int fdm_plan_get(fdm_plan* P, const char* key, long long* out) {
  if (!P || !key || !out) return fail(FDM_ERR_ARG, "plan_get: null argument");
  const std::string k(key);
  if (k == "launches_per_step") *out = P->launches_per_step;
  else if (k == "graph_launches") *out = P->last_graph_launches;
  else if (k == "fuse_ln3") *out = P->fuse_ln3;
  else if (k == "rows") *out = P->R;
  else if (k == "ksplit.out") *out = P->ksplit_out;
  else if (k == "ksplit.ffn2") *out = P->ksplit_ffn2;
  else if (k == "tuned") *out = P->tile_cache.count(shape_key(P)) ? 1 : 0;
  else if (k == "needs_tune") { const std::string sk = shape_key(P); *out = (!P->tile_cache.count(sk) && P->tune_enabled && P->steps_seen.count(sk) && P->steps_seen[sk] >= 2000 && !P->tune_failed_shapes.count(sk)) ? 1 : 0; }
  else if (k == "tune_failed") *out = P->tune_failed;
  else if (k.rfind("tile.", 0) == 0) { auto it = P->tiles.find(k.substr(5)); *out = it == P->tiles.end() ? 0 : it->second; }
  else return fail(FDM_ERR_ARG, "plan_get: unknown key '%s'", key);
  return FDM_OK;
}

int fdm_plan_set(fdm_plan* P, const char* key, long long value) {
  if (!P || !key) return fail(FDM_ERR_ARG, "plan_set: null argument");
  const std::string k(key);
  if (k == "tune") { P->tune_enabled = value != 0; return FDM_OK; }
  if (k == "tune_lazy") { P->tune_lazy = value != 0; return FDM_OK; }
  if (k == "fuse_ln3") {      // takes effect at the next commit (the folded weights are commit-time tables)
    if ((value != 0) != (P->want_fuse_ln3 != 0)) { P->want_fuse_ln3 = value != 0; P->committed = false; P->prepared = false; return drop_programs(P, nullptr); }
    return FDM_OK;
  }
  if (k == "ksplit.out" || k == "ksplit.ffn2") {      // K slices of the out-proj / FFN2 GEMMs (1 = none); tuned tiles of other split factors no longer apply
    const int kt = (k == "ksplit.out" ? P->m.d : P->m.ffn) / (P->dtype == FDM_F32 ? 32 : 64);
    if (value < 1 || value > 4 || kt % value) return fail(FDM_ERR_ARG, "plan_set: %s = %lld must be 1..4 and divide the %d k-tiles", key, value, kt);
    int& cur = k == "ksplit.out" ? P->ksplit_out : P->ksplit_ffn2;
    if (cur == (int)value) return FDM_OK;
    cur = (int)value;
    P->tile_cache.clear(); P->tiles.erase(k == "ksplit.out" ? "out" : "ffn2");
    FCK(drop_programs(P, nullptr));
    if (P->capB > 0 && (int)value > P->x1_planes) {     // more partial planes than the workspace holds: a larger x1 (its contents do not outlive a step)
      P->x1_planes = (int)value;
      FCK(dalloc_t(P, &P->x1, (size_t)P->capB * P->capL * P->capRep * P->m.d * P->x1_planes, true));
    }
    return FDM_OK;
  }
  if (k == "lockstep") { P->lockstep = value != 0; P->tile_cache.clear(); P->tiles.clear(); return drop_programs(P, nullptr); }
  if (k == "untune") {      // forget the tuned tiles of every shape (tests)
    P->tile_cache.clear(); P->steps_seen.clear(); P->tiles.clear(); P->tune_failed_shapes.clear();
    return drop_programs(P, nullptr);
  }
  if (k.rfind("tile.", 0) == 0) {
    if (value < 0 || value > FDM_TILE_MAX) return fail(FDM_ERR_ARG, "plan_set: unknown tile %lld", value);
    P->tiles[k.substr(5)] = (int)value;
    P->tile_cache[shape_key(P)] = P->tiles;      // an explicit choice counts as tuned: sampling calls keep it
    return drop_programs(P, nullptr);
  }
  return fail(FDM_ERR_ARG, "plan_set: unknown key '%s'", key);
}

}  // extern "C"
