// Once-per-clip stages of the path behind plain C calls (include/fdm_hip.h, "Audio encoder" and "VQ" sections):
//   fdm_hubert_*  HuBERT-large / wav2vec2-base encoder: models/hubert.py:75-146, models/wav2vec.py:69-143 over transformers'
//                 Hubert/Wav2Vec2 FeatureEncoder, FeatureProjection, PositionalConvEmbedding, Encoder(StableLayerNorm)
//   fdm_vq_*      (E)VQ-VAE quantiser, decoder and encoder: models/lib/quantizer.py:35-64, models/vq_vae_emotion.py:221-252,
//                 models/vq_vae_vocaset.py:35-43,134-258, models/vq_vae_emotion.py:279-352, models/vq_vae.py:275-347 with
//                 models/lib/base_models.py:37-87,138-174,286-301
// Weights arrive by reference state-dict name (fp32, host or device) and are repacked once; forward passes only sequence
// fdm_op_* launches on the caller's stream.  Workspaces grow on demand (the only allocations after create).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fdm_hip.h"
#include "common.hpp"
#include "kernels.hpp"

namespace {
using fdm::fail;

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(FDM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define FCK(x) do { int r_ = (x); if (r_ != FDM_OK) return r_; } while (0)

int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

// ---- one-time repack kernels -----------------------------------------------------------------------------------------
// Conv1d weight [out, in, k] -> [out, k, in]: a strided Conv1d over a channels-last signal is then a GEMM whose A rows overlap
__global__ void permute_oik_oki_kernel(const float* w, float* out, int O, int I, int K) {
  const long long n = (long long)O * I * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ii = (int)(i % I), k = (int)((i / I) % K), o = (int)(i / ((long long)I * K));
    out[i] = w[((size_t)o * I + ii) * K + k];
  }
}
// weight_norm(dim = 2) of the grouped positional conv: norm[k] = || v[:, :, k] ||_2 (one block per k, fixed order)
__global__ __launch_bounds__(256) void posconv_norm_kernel(const float* v, float* norm, int OI, int K) {
  __shared__ float red[4];
  const int k = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < OI; i += 256) { const float x = v[(size_t)i * K + k]; s += x * x; }
  s = fdm::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm[k] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}
// w[o, i, k] = g[k] * v[o, i, k] / norm[k], repacked per group to [G][dg (out)][K][dg (in)]
__global__ void posconv_pack_kernel(const float* g, const float* v, const float* norm, float* out, int G, int dg, int K) {
  const long long n = (long long)G * dg * K * dg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ii = (int)(i % dg), k = (int)((i / dg) % K), o = (int)((i / ((long long)dg * K)) % dg), gi = (int)(i / ((long long)dg * K * dg));
    const float x = v[((size_t)(gi * dg + o) * dg + ii) * K + k];
    out[i] = g[k] * x / norm[k];
  }
}
// out[r, 0:src_cols] = in[r, :], zero beyond (K of a GEMM padded to a whole k-tile)
__global__ void pad_cols_kernel(const float* in, float* out, long long rows, int src_cols, int dst_cols) {
  const long long n = rows * dst_cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dst_cols);
    out[i] = c < src_cols ? in[(i / dst_cols) * src_cols + c] : 0.f;
  }
}
// [B, C, R] -> [B, R, C]
__global__ void bcr_to_brc_kernel(const float* in, float* out, int B, int C, int R) {
  const long long n = (long long)B * C * R;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), r = (int)((i / C) % R), b = (int)(i / ((long long)C * R));
    out[i] = in[((size_t)b * C + c) * R + r];
  }
}
// book[b] = argmax(one_hot[b, :]) (first maximum, as torch.argmax)
__global__ void argmax_rows_kernel(const float* x, int* out, int B, int n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int best = 0;
  float bv = x[(size_t)b * n];
  for (int i = 1; i < n; ++i) { const float v = x[(size_t)b * n + i]; if (v > bv) { bv = v; best = i; } }
  out[b] = best;
}

// ---- shared bookkeeping -----------------------------------------------------------------------------------------------
struct Wt { float* p = nullptr; long long n = 0; };
struct Mat { void* p = nullptr; long long lo = 0; };      // operand-kind matrix; lo = elements between the hi and lo planes (split kind)

struct Arena {
  std::vector<void*> allocs;
  int alloc(void** out, size_t bytes, bool zero = false) {
    void* p = nullptr;
    HIPCK(hipMalloc(&p, bytes ? bytes : 16));
    if (zero) HIPCK(hipMemset(p, 0, bytes ? bytes : 16));
    allocs.push_back(p);
    *out = p;
    return FDM_OK;
  }
  template <typename T> int alloc_t(T** out, size_t n, bool zero = false) { return alloc((void**)out, n * sizeof(T), zero); }
  void release() { for (void* p : allocs) (void)hipFree(p); allocs.clear(); }
};

struct Store {       // fp32 weights by reference state-dict name (plan-owned copies)
  std::map<std::string, Wt> w;
  Arena mem;
  int set(const char* name, const float* ptr, long long n, void* stream) {
    Wt& t = w[name];
    if (t.n != n) { t.n = n; FCK(mem.alloc_t(&t.p, (size_t)n)); }
    HIPCK(hipMemcpyAsync(t.p, ptr, (size_t)n * 4, hipMemcpyDefault, (hipStream_t)stream));
    return FDM_OK;
  }
  const Wt* find(const std::string& name) const { auto it = w.find(name); return it == w.end() ? nullptr : &it->second; }
  int need(const std::string& name, long long n, const float** out) const {
    const Wt* t = find(name);
    if (!t) return fail(FDM_ERR_STATE, "missing weight %s", name.c_str());
    if (t->n != n) return fail(FDM_ERR_SHAPE, "weight %s has %lld elements, expected %lld", name.c_str(), t->n, n);
    *out = t->p;
    return FDM_OK;
  }
};

size_t esize(int dtype) { return dtype == FDM_BF16 ? 2 : 4; }      // (FDM_F16X3: two 2-byte planes = 4)

int to_operand(Arena& mem, int dtype, const float* src, long long n, Mat* out, void* stream) {
  if (dtype == FDM_F32) { out->p = (void*)src; return FDM_OK; }
  FCK(mem.alloc(&out->p, (size_t)n * esize(dtype)));
  out->lo = dtype == FDM_F16X3 ? n : 0;          // fdm_op_cast writes the lo plane n elements after the hi plane
  return fdm_op_cast(src, out->p, n, dtype, stream);
}

fdm_gemm_args gemm_args(int dtype, const void* A, const void* W, int M, int N, int K) {
  fdm_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = dtype;
  a.ldr = N; a.ldo_f32 = N; a.ldo_t = N; a.ln_eps = 1e-5f;
  return a;
}
// (split dtype: y_t is a plane pair of M * d elements each, lo plane right behind the hi plane)
int layernorm(const float* x, const float* gamma, const float* beta, int M, int d, int act, float* y32, void* yt, int dtype, void* stream) {
  fdm_ln_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.M = M; a.d = d; a.gamma = gamma; a.beta = beta; a.eps = 1e-5f; a.act = act; a.y_f32 = y32; a.y_t = yt; a.dtype = dtype;
  if (yt && dtype == FDM_F16X3) a.y_t_lo_off = (long long)M * d;
  return fdm_op_layernorm(&a, stream);
}
int kv_pad(int L) { return (L + 31) / 32 * 32; }

// one pre-LN transformer block pair shared by the audio encoders' and the VQ transformers' layer loops
struct Layer {
  const float *ln1g, *ln1b, *ln2g, *ln2b, *bqkv, *bo, *b1, *b2;
  Mat wqkv, wo, w1, w2;
};

const int CONV_K[7] = {10, 3, 3, 3, 3, 2, 2};
const int CONV_S[7] = {5, 2, 2, 2, 2, 2, 2};
const int CD = 512, POS_K = 128, POS_G = 16;

}  // namespace

// =====================================================================================================================
// audio encoder
// =====================================================================================================================
struct fdm_audio_encoder {
  int kind = 0, n_layers = 0, dtype = FDM_F32;
  // FDM_F16X3: every GEMM of the encoder runs on split-fp16 operands -- the transformer layers (84 % of the FLOPs) since round 3,
  // the front (conv stack, projection, positional conv: 26 % of the round-3 f16x3 forward on the fp32 matrix pipe) since round 4
  int front_dtype() const { return dtype; }
  int D = 1024, H = 16, FFN = 4096;
  bool conv_layer_norm = true, conv_bias = true, stable_ln = true;
  Store st;
  Arena mem, ws;
  bool committed = false;
  // repacked weights
  const float* conv0_w = nullptr;
  Mat conv_w[7];
  const float *conv_b[7] = {}, *conv_g[7] = {}, *conv_beta[7] = {};
  const float *fp_lng = nullptr, *fp_lnb = nullptr, *fp_b = nullptr, *pc_b = nullptr, *fin_g = nullptr, *fin_b = nullptr;
  Mat fp_w, pc_w;
  std::vector<Layer> layers;
  // workspace (capacity: capB clips x capN samples)
  int capB = 0, capN = 0;
  float *x32 = nullptr, *y32 = nullptr, *g6 = nullptr, *gi = nullptr, *h = nullptr, *h2 = nullptr, *hb = nullptr, *x1 = nullptr;
  void *xa = nullptr, *xb = nullptr, *ft = nullptr, *ht = nullptr, *xg = nullptr, *xt = nullptr, *q = nullptr, *kp = nullptr, *vp = nullptr, *ctx = nullptr, *u = nullptr;
};

namespace {

void conv_lengths(int n, int* T) {
  for (int i = 0; i < 7; ++i) { n = (n - CONV_K[i]) / CONV_S[i] + 1; T[i] = n; }
}

int enc_commit(fdm_audio_encoder* E, void* stream) {
  if (E->committed) return FDM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int D = E->D, dt = E->front_dtype(), dtl = E->dtype;
  Store& st = E->st;
  const float* p = nullptr;
  for (int i = 0; i < 7; ++i) {
    char nm[96];
    snprintf(nm, sizeof(nm), "feature_extractor.conv_layers.%d.conv.weight", i);
    FCK(st.need(nm, (long long)CD * (i == 0 ? 1 : CD) * CONV_K[i], &p));
    if (i == 0) {
      E->conv0_w = p;            // [512, 1, 10] == [512, 10]: direct kernel, fp32
    } else {
      float* r = nullptr;
      FCK(E->mem.alloc_t(&r, (size_t)CD * CD * CONV_K[i]));
      hipLaunchKernelGGL(permute_oik_oki_kernel, dim3(grid_for((long long)CD * CD * CONV_K[i])), dim3(256), 0, s, p, r, CD, CD, CONV_K[i]);
      FCK(to_operand(E->mem, dt, r, (long long)CD * CD * CONV_K[i], &E->conv_w[i], stream));
    }
    E->conv_b[i] = nullptr;
    if (E->conv_bias) {
      snprintf(nm, sizeof(nm), "feature_extractor.conv_layers.%d.conv.bias", i);
      FCK(st.need(nm, CD, &E->conv_b[i]));
    }
    E->conv_g[i] = E->conv_beta[i] = nullptr;
    if (E->conv_layer_norm || i == 0) {
      snprintf(nm, sizeof(nm), "feature_extractor.conv_layers.%d.layer_norm.weight", i);
      FCK(st.need(nm, CD, &E->conv_g[i]));
      snprintf(nm, sizeof(nm), "feature_extractor.conv_layers.%d.layer_norm.bias", i);
      FCK(st.need(nm, CD, &E->conv_beta[i]));
    }
  }
  FCK(st.need("feature_projection.layer_norm.weight", CD, &E->fp_lng));
  FCK(st.need("feature_projection.layer_norm.bias", CD, &E->fp_lnb));
  FCK(st.need("feature_projection.projection.weight", (long long)D * CD, &p));
  FCK(to_operand(E->mem, dt, p, (long long)D * CD, &E->fp_w, stream));
  FCK(st.need("feature_projection.projection.bias", D, &E->fp_b));
  // weight-normalised grouped positional conv (weight_norm dim = 2): torch >= 2.1 parametrization names or torch 2.0's weight_g / weight_v
  const int dg = D / POS_G;
  const float *wg = nullptr, *wv = nullptr;
  const char* n0 = st.find("encoder.pos_conv_embed.conv.parametrizations.weight.original0") ? "encoder.pos_conv_embed.conv.parametrizations.weight.original0" : "encoder.pos_conv_embed.conv.weight_g";
  const char* n1 = st.find("encoder.pos_conv_embed.conv.parametrizations.weight.original1") ? "encoder.pos_conv_embed.conv.parametrizations.weight.original1" : "encoder.pos_conv_embed.conv.weight_v";
  FCK(st.need(n0, POS_K, &wg));
  FCK(st.need(n1, (long long)D * dg * POS_K, &wv));
  float *nrm = nullptr, *wpc = nullptr;
  FCK(E->mem.alloc_t(&nrm, (size_t)POS_K));
  FCK(E->mem.alloc_t(&wpc, (size_t)D * dg * POS_K));
  hipLaunchKernelGGL(posconv_norm_kernel, dim3(POS_K), dim3(256), 0, s, wv, nrm, D * dg, POS_K);
  hipLaunchKernelGGL(posconv_pack_kernel, dim3(grid_for((long long)D * dg * POS_K)), dim3(256), 0, s, wg, wv, (const float*)nrm, wpc, POS_G, dg, POS_K);
  FCK(to_operand(E->mem, dt, wpc, (long long)D * dg * POS_K, &E->pc_w, stream));
  FCK(st.need("encoder.pos_conv_embed.conv.bias", D, &E->pc_b));
  E->layers.assign(E->n_layers, Layer());
  for (int l = 0; l < E->n_layers; ++l) {
    Layer& ly = E->layers[l];
    char pre[64];
    snprintf(pre, sizeof(pre), "encoder.layers.%d.", l);
    const std::string P(pre);
    float *wqkv = nullptr, *bqkv = nullptr;
    FCK(E->mem.alloc_t(&wqkv, (size_t)3 * D * D));
    FCK(E->mem.alloc_t(&bqkv, (size_t)3 * D));
    const char* proj[3] = {"attention.q_proj.", "attention.k_proj.", "attention.v_proj."};
    for (int j = 0; j < 3; ++j) {
      const float *w = nullptr, *b = nullptr;
      FCK(st.need(P + proj[j] + "weight", (long long)D * D, &w));
      FCK(st.need(P + proj[j] + "bias", D, &b));
      HIPCK(hipMemcpyAsync(wqkv + (size_t)j * D * D, w, (size_t)D * D * 4, hipMemcpyDeviceToDevice, s));
      HIPCK(hipMemcpyAsync(bqkv + (size_t)j * D, b, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    }
    FCK(to_operand(E->mem, dtl, wqkv, 3LL * D * D, &ly.wqkv, stream));
    ly.bqkv = bqkv;
    FCK(st.need(P + "attention.out_proj.weight", (long long)D * D, &p));
    FCK(to_operand(E->mem, dtl, p, (long long)D * D, &ly.wo, stream));
    FCK(st.need(P + "attention.out_proj.bias", D, &ly.bo));
    FCK(st.need(P + "layer_norm.weight", D, &ly.ln1g)); FCK(st.need(P + "layer_norm.bias", D, &ly.ln1b));
    FCK(st.need(P + "final_layer_norm.weight", D, &ly.ln2g)); FCK(st.need(P + "final_layer_norm.bias", D, &ly.ln2b));
    FCK(st.need(P + "feed_forward.intermediate_dense.weight", (long long)E->FFN * D, &p));
    FCK(to_operand(E->mem, dtl, p, (long long)E->FFN * D, &ly.w1, stream));
    FCK(st.need(P + "feed_forward.intermediate_dense.bias", E->FFN, &ly.b1));
    FCK(st.need(P + "feed_forward.output_dense.weight", (long long)D * E->FFN, &p));
    FCK(to_operand(E->mem, dtl, p, (long long)D * E->FFN, &ly.w2, stream));
    FCK(st.need(P + "feed_forward.output_dense.bias", D, &ly.b2));
  }
  FCK(st.need("encoder.layer_norm.weight", D, &E->fin_g));
  FCK(st.need("encoder.layer_norm.bias", D, &E->fin_b));
  HIPCK(hipGetLastError());
  E->committed = true;
  return FDM_OK;
}

int enc_reserve(fdm_audio_encoder* E, int B, int n) {
  if (B <= E->capB && n <= E->capN) return FDM_OK;
  B = B > E->capB ? B : E->capB; n = n > E->capN ? n : E->capN;
  (void)hipDeviceSynchronize();
  E->ws.release();
  int T[7];
  conv_lengths(n, T);
  const size_t es = esize(E->front_dtype()), esl = esize(E->dtype), D = E->D;
  const size_t r0 = (size_t)B * T[0], N = (size_t)T[6] + 2, M = (size_t)B * N;
  if (!E->conv_layer_norm) FCK(E->ws.alloc_t(&E->x32, r0 * CD));      // conv 0's fp32 output: only the GroupNorm front (wav2vec2-base) materialises it
  FCK(E->ws.alloc_t(&E->y32, (size_t)B * T[1] * CD));
  FCK(E->ws.alloc(&E->xa, r0 * CD * es)); FCK(E->ws.alloc(&E->xb, (size_t)B * T[1] * CD * es));
  FCK(E->ws.alloc_t(&E->g6, M * CD)); FCK(E->ws.alloc_t(&E->gi, M * CD)); FCK(E->ws.alloc(&E->ft, M * CD * es));
  FCK(E->ws.alloc_t(&E->h, M * D)); FCK(E->ws.alloc_t(&E->h2, M * D)); FCK(E->ws.alloc_t(&E->hb, M * D)); FCK(E->ws.alloc_t(&E->x1, M * D));
  FCK(E->ws.alloc(&E->ht, M * D * es)); FCK(E->ws.alloc(&E->xt, M * D * esl)); FCK(E->ws.alloc(&E->q, M * D * esl)); FCK(E->ws.alloc(&E->ctx, M * D * esl));
  FCK(E->ws.alloc(&E->xg, (size_t)POS_G * B * (N + POS_K) * (D / POS_G) * es));
  FCK(E->ws.alloc(&E->u, M * E->FFN * esl));
  FCK(E->ws.alloc(&E->kp, (size_t)B * kv_pad((int)N) * D * esl, true)); FCK(E->ws.alloc(&E->vp, (size_t)B * kv_pad((int)N) * D * esl, true));
  E->capB = B; E->capN = n;
  return FDM_OK;
}

}  // namespace

extern "C" {

int fdm_hubert_frames(int n_samples) {
  if (n_samples < 400) return 0;
  int T[7];
  conv_lengths(n_samples, T);
  return T[6] - (T[6] % 2);
}

int fdm_hubert_create(int kind, int n_layers, int dtype, fdm_audio_encoder** out) {
  if (!out) return fail(FDM_ERR_ARG, "hubert_create: null out");
  if (kind != 0 && kind != 1) return fail(FDM_ERR_ARG, "hubert_create: kind %d (0 = HuBERT-large, 1 = wav2vec2-base)", kind);
  if (dtype != FDM_F32 && dtype != FDM_BF16 && dtype != FDM_F16X3)
    return fail(FDM_ERR_ARG, "hubert_create: dtype %d (fp32, bf16, or FDM_F16X3 = split-fp16 operands in the conv front and the transformer layers)", dtype);
  fdm_audio_encoder* E = new (std::nothrow) fdm_audio_encoder();
  if (!E) return fail(FDM_ERR_STATE, "hubert_create: out of memory");
  E->kind = kind; E->dtype = dtype;
  if (kind == 0) { E->D = 1024; E->H = 16; E->FFN = 4096; E->n_layers = 24; E->conv_layer_norm = true; E->conv_bias = true; E->stable_ln = true; }
  else { E->D = 768; E->H = 12; E->FFN = 3072; E->n_layers = 12; E->conv_layer_norm = false; E->conv_bias = false; E->stable_ln = false; }
  if (n_layers > 0) E->n_layers = n_layers;
  *out = E;
  return FDM_OK;
}

int fdm_hubert_destroy(fdm_audio_encoder* E) {
  if (!E) return FDM_OK;
  (void)hipDeviceSynchronize();
  E->ws.release(); E->mem.release(); E->st.mem.release();
  delete E;
  return FDM_OK;
}

int fdm_hubert_set_weights(fdm_audio_encoder* E, const char* name, const float* ptr, long long n, void* stream) {
  if (!E || !name || !ptr || n <= 0) return fail(FDM_ERR_ARG, "hubert_set_weights: bad argument");
  if (E->committed) return fail(FDM_ERR_STATE, "hubert_set_weights: weights are frozen after the first forward (create a new encoder)");
  return E->st.set(name, ptr, n, stream);
}

int fdm_hubert_forward(fdm_audio_encoder* E, const float* wav, int B, int n, int frame_num, int interp_in_fps, int interp_out_fps,
                       float* out, int* n_frames, void* stream) {
  if (!E || !wav || !out) return fail(FDM_ERR_ARG, "hubert_forward: null argument");
  if (B < 1 || n < 400) return fail(FDM_ERR_SHAPE, "hubert_forward: audio too short (%d samples)", n);
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "hubert_forward: no gfx950 device visible (there is no CPU fallback)");
  int T[7];
  conv_lengths(n, T);
  if (T[6] < 2) return fail(FDM_ERR_SHAPE, "hubert_forward: audio too short (%d samples)", n);
  const bool interp = interp_in_fps > 0 && interp_out_fps > 0;
  int N = T[6] - (T[6] % 2);                                        // drop the last frame if odd (models/hubert.py:95-96)
  if (frame_num > 0 && !interp && N > frame_num * 2) N = frame_num * 2;     // :97-98
  int T6 = T[6];
  if (interp) {
    N = frame_num > 0 ? frame_num : (int)((double)T6 / (double)interp_in_fps * (double)interp_out_fps);
    if (N < 2) return fail(FDM_ERR_SHAPE, "hubert_forward: interpolated length %d too short", N);
    if (N > T[6] + 2) return fail(FDM_ERR_SHAPE, "hubert_forward: interpolated length %d exceeds the workspace (%d conv frames)", N, T[6]);
  }
  FCK(enc_commit(E, stream));
  FCK(enc_reserve(E, B, n));
  const int D = E->D, dt = E->front_dtype(), dtl = E->dtype, H = E->H, FFN = E->FFN, HD = 64;
  const size_t es = esize(dt), esl = esize(dtl);
  const bool split = dtl == FDM_F16X3;
  // --- conv feature extractor (channels-last) ---
  // split front (FDM_F16X3): every activation matrix [rows, C] is a plane pair, lo plane rows * C elements behind the hi plane
  const bool fsplit = dt == FDM_F16X3;
  auto lo_of = [&](long long rows, long long cols) { return fsplit ? rows * cols : 0LL; };
  void* xt = E->xa;
  if (E->conv_layer_norm) {        // conv 0 + LayerNorm + GELU in one kernel: only the operand copy is stored
    FCK(fdm_op_conv0_ln_gelu(wav, E->conv0_w, E->conv_b[0], E->conv_g[0], E->conv_beta[0], xt, lo_of((long long)B * T[0], CD), B, n, T[0], 1e-5f, dt, stream));
  } else {                         // wav2vec2-base: GroupNorm over time needs the whole clip's conv output first
    FCK(fdm_op_conv0(wav, E->conv0_w, E->conv_b[0], E->x32, B, n, T[0], stream));
    // (chunk statistics go through y32: the conv stack's fp32 scratch, free until layer 1's GEMM writes it)
    FCK(fdm_op_time_groupnorm(E->x32, E->conv_g[0], E->conv_beta[0], nullptr, xt, lo_of((long long)B * T[0], CD), B, T[0], CD, 1e-5f, FDM_ACT_GELU_ERF, dt,
                              E->y32, (long long)B * T[1] * CD * 4, stream));
  }
  int Tin = T[0];
  for (int i = 1; i < 7; ++i) {
    const int k = CONV_K[i], sd = CONV_S[i], To = T[i];
    fdm_gemm_args g = gemm_args(dt, xt, E->conv_w[i].p, To, CD, k * CD);
    g.lda = (long long)sd * CD; g.bias = E->conv_b[i]; g.batch = B; g.a_batch_stride = (long long)Tin * CD; g.out_batch_stride = (long long)To * CD;
    g.a_lo_off = lo_of((long long)B * Tin, CD); g.w_lo_off = E->conv_w[i].lo;
    void* nx = (xt == E->xa) ? E->xb : E->xa;
    if (E->conv_layer_norm) {
      g.out_f32 = E->y32;
      FCK(fdm_op_gemm(&g, stream));
      if (i < 6) FCK(layernorm(E->y32, E->conv_g[i], E->conv_beta[i], B * To, CD, FDM_ACT_GELU_ERF, nullptr, nx, dt, stream));
      else FCK(layernorm(E->y32, E->conv_g[i], E->conv_beta[i], B * To, CD, FDM_ACT_GELU_ERF, E->g6, nullptr, dt, stream));
    } else {        // conv (no norm) + GELU fused in the GEMM epilogue
      g.act = FDM_ACT_GELU_ERF;
      if (i < 6) { g.out_t = nx; g.out_t_lo_off = lo_of((long long)B * To, CD); } else g.out_f32 = E->g6;
      FCK(fdm_op_gemm(&g, stream));
    }
    xt = nx; Tin = To;
  }
  // --- optional 50 -> 30 fps resampling (linear_interpolation, models/hubert.py:62-69), else the even crop via batch strides ---
  const float* g6 = E->g6;
  if (interp) { FCK(fdm_op_linear_interp(E->g6, E->gi, B, T6, N, CD, stream)); g6 = E->gi; T6 = N; }
  FCK(layernorm(g6, E->fp_lng, E->fp_lnb, B * T6, CD, FDM_ACT_NONE, nullptr, E->ft, dt, stream));
  const int M = B * N;
  fdm_gemm_args g = gemm_args(dt, E->ft, E->fp_w.p, N, D, CD);
  g.bias = E->fp_b; g.out_f32 = E->h; g.batch = B; g.a_batch_stride = (long long)T6 * CD; g.out_batch_stride = (long long)N * D;
  g.a_lo_off = lo_of((long long)B * T6, CD); g.w_lo_off = E->fp_w.lo;
  void* ht = dt != FDM_F32 ? E->ht : (void*)E->h;
  if (dt != FDM_F32) { g.out_t = E->ht; g.out_t_lo_off = lo_of((long long)M, D); }
  FCK(fdm_op_gemm(&g, stream));
  // --- positional conv embedding: h += GELU(grouped conv(h)), k = 128, groups = 16, pad 64, last output dropped ---
  const int dg = D / POS_G;
  const long long xg_plane = (long long)POS_G * B * (N + POS_K) * dg;      // elements of one [groups, B, N + K, dg] plane
  const size_t ee = fsplit ? 2 : es;                                     // bytes per element of a plane
  FCK(fdm_op_group_pad(ht, E->xg, B, N, D, POS_G, POS_K / 2, dt, stream));
  if (fsplit) FCK(fdm_op_group_pad((const char*)ht + (size_t)M * D * 2, (char*)E->xg + (size_t)xg_plane * 2, B, N, D, POS_G, POS_K / 2, dt, stream));
  {
    // ONE launch for all clips (round 5; one launch per clip before): z = (clip, group), the weights of a group shared by the clips
    // and by the row tiles, workgroups dealt so that every XCD works on two groups only -- its L2 streams 1 / 8 of the 16.8 MB
    // weight once (fdm_gemm_args.batch2).  Per clip the launch fetched 133 MB for 18 MB of operands (profiles/r4_pmc_hubert_bf16_B4).
    (void)ee;
    fdm_gemm_args pg = gemm_args(dt, E->xg, E->pc_w.p, N, dg, POS_K * dg);
    pg.lda = dg; pg.batch = POS_G; pg.a_batch_stride = (long long)B * (N + POS_K) * dg; pg.w_batch_stride = (long long)dg * POS_K * dg;
    pg.batch2 = B; pg.a_batch_stride2 = (long long)(N + POS_K) * dg; pg.out_batch_stride2 = (long long)N * D;
    pg.a_lo_off = fsplit ? xg_plane : 0; pg.w_lo_off = E->pc_w.lo;
    pg.bias = E->pc_b; pg.bias_batch_stride = dg; pg.act = FDM_ACT_GELU_ERF;
    pg.resid = E->h; pg.ldr = D; pg.out_f32 = E->h2; pg.ldo_f32 = D; pg.out_batch_stride = dg;
    FCK(fdm_op_gemm(&pg, stream));
  }
  float* h = E->h2;
  float* hb = E->hb;
  float* hx = E->h;         // free fp32 buffer (the pre-posconv h)
  // --- encoder layers ---
  const int Lpad = kv_pad(N);
  HIPCK(hipMemsetAsync(E->kp, 0, (size_t)B * Lpad * D * esl, (hipStream_t)stream));       // pad keys must be finite; the layout depends on N
  HIPCK(hipMemsetAsync(E->vp, 0, (size_t)B * Lpad * D * esl, (hipStream_t)stream));
  // split kind: every operand of the layer loop is a plane pair; lo planes sit one whole matrix after the hi planes
  const long long lo_md = split ? (long long)M * D : 0, lo_mf = split ? (long long)M * FFN : 0, lo_kv = split ? (long long)B * Lpad * D : 0;
  auto lgemm = [&](const void* A, long long a_lo, const Mat& W, int n_out, int k_in) {
    fdm_gemm_args a = gemm_args(dtl, A, W.p, M, n_out, k_in);
    a.a_lo_off = a_lo; a.w_lo_off = W.lo;
    return a;
  };
  auto lnorm = [&](const float* x, const float* gam, const float* bet, float* y32, void* yt) {
    fdm_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.M = M; a.d = D; a.gamma = gam; a.beta = bet; a.eps = 1e-5f; a.act = FDM_ACT_NONE; a.y_f32 = y32; a.y_t = yt; a.dtype = dtl;
    a.y_t_lo_off = yt ? lo_md : 0;
    return fdm_op_layernorm(&a, stream);
  };
  auto qkv = [&](const void* a_in, const Layer& ly) {
    fdm_gemm_args a = lgemm(a_in, lo_md, ly.wqkv, 3 * D, D);
    a.bias = ly.bqkv; a.out_t = E->q; a.ldo_t = D; a.out_t_lo_off = lo_md; a.out_kp = E->kp; a.kp_col0 = D; a.out_vp = E->vp; a.vp_col0 = 2 * D;
    a.kv_L = N; a.kv_Lpad = Lpad; a.kv_hd = HD; a.kv_lo_off = lo_kv;
    return fdm_op_gemm(&a, stream);
  };
  auto attn = [&]() {
    fdm_attn_args a;
    memset(&a, 0, sizeof(a));
    a.Q = E->q; a.ldq = D; a.Kp = E->kp; a.Vp = E->vp; a.Lpad = Lpad; a.O = E->ctx; a.ldo = D; a.B = B; a.H = H; a.L = N; a.hd = HD;
    a.dtype = dtl; a.scale = 0.125f; a.causal = 0; a.period = 1;
    a.q_lo_off = lo_md; a.kv_lo_off = lo_kv; a.o_lo_off = lo_md;
    return fdm_op_attention(&a, stream);
  };
  if (E->stable_ln) {       // pre-LN layers, final LayerNorm (HubertEncoderStableLayerNorm)
    for (const Layer& ly : E->layers) {
      FCK(lnorm(h, ly.ln1g, ly.ln1b, nullptr, E->xt));
      FCK(qkv(E->xt, ly));
      FCK(attn());
      fdm_gemm_args a = lgemm(E->ctx, lo_md, ly.wo, D, D);
      a.bias = ly.bo; a.resid = h; a.out_f32 = hb;
      FCK(fdm_op_gemm(&a, stream));
      FCK(lnorm(hb, ly.ln2g, ly.ln2b, nullptr, E->xt));
      a = lgemm(E->xt, lo_md, ly.w1, FFN, D);
      a.bias = ly.b1; a.act = FDM_ACT_GELU_ERF; a.out_t = E->u; a.out_t_lo_off = lo_mf;
      FCK(fdm_op_gemm(&a, stream));
      a = lgemm(E->u, lo_mf, ly.w2, D, FFN);
      a.bias = ly.b2; a.resid = hb; a.out_f32 = h;
      FCK(fdm_op_gemm(&a, stream));
    }
    FCK(lnorm(h, E->fin_g, E->fin_b, out, nullptr));
  } else {                  // LayerNorm before the stack, post-LN layers (Wav2Vec2Encoder / Wav2Vec2EncoderLayer)
    const bool both = dtl != FDM_F32;
    void* htt = both ? E->xt : nullptr;
    FCK(lnorm(h, E->fin_g, E->fin_b, hb, htt));
    for (const Layer& ly : E->layers) {
      const void* a_in = both ? (const void*)E->xt : (const void*)hb;
      FCK(qkv(a_in, ly));
      FCK(attn());
      fdm_gemm_args a = lgemm(E->ctx, lo_md, ly.wo, D, D);
      a.bias = ly.bo; a.resid = hb; a.out_f32 = E->x1;
      FCK(fdm_op_gemm(&a, stream));
      FCK(lnorm(E->x1, ly.ln1g, ly.ln1b, hb, htt));
      a = lgemm(a_in, lo_md, ly.w1, FFN, D);
      a.bias = ly.b1; a.act = FDM_ACT_GELU_ERF; a.out_t = E->u; a.out_t_lo_off = lo_mf;
      FCK(fdm_op_gemm(&a, stream));
      a = lgemm(E->u, lo_mf, ly.w2, D, FFN);
      a.bias = ly.b2; a.resid = hb; a.out_f32 = E->x1;
      FCK(fdm_op_gemm(&a, stream));
      FCK(lnorm(E->x1, ly.ln2g, ly.ln2b, hb, htt));
    }
    HIPCK(hipMemcpyAsync(out, hb, (size_t)M * D * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  (void)hx;
  if (n_frames) *n_frames = N;
  return FDM_OK;
}

}  // extern "C"

// =====================================================================================================================
// (E)VQ-VAE
// =====================================================================================================================
struct fdm_vq {
  fdm_vq_desc d{};
  int dtype = FDM_F32;
  // FDM_F16X3: the two 6-layer transformers on split-fp16 operands; convs, embeddings, the vertex map and the quantiser stay fp32
  int front_dtype() const { return dtype == FDM_F16X3 ? FDM_F32 : dtype; }
  Store st;
  Arena mem, ws;
  bool committed = false, has_encoder = false;
  const float* codebook = nullptr;
  Mat pre_w, conv_w, emb_w, out_w;
  const float *pre_b = nullptr, *conv_b = nullptr, *emb_b = nullptr, *out_b = nullptr;
  float* pe0 = nullptr;
  std::vector<Layer> dec_layers, enc_layers;
  // encoder
  int Kp = 0;
  Mat e_map_w, e_conv_w, e_emb_w, e_post_w;
  const float *e_map_b = nullptr, *e_emo_w = nullptr, *e_emo_b = nullptr, *e_conv_b = nullptr, *e_emb_b = nullptr, *e_post_b = nullptr;
  // workspace
  size_t capM = 0; int capB = 0, capL = 0;
  float *x32 = nullptr, *c32 = nullptr, *h = nullptr, *hb = nullptr, *h2 = nullptr, *em = nullptr, *xpad32 = nullptr;
  void *xt = nullptr, *y = nullptr, *xp = nullptr, *nt = nullptr, *q = nullptr, *kp = nullptr, *vp = nullptr, *ctx = nullptr, *u = nullptr, *a = nullptr, *xpt = nullptr;
  int* book = nullptr;
  double* stat_partial = nullptr; int* stat_hist = nullptr;      // fdm_vq_quant_stats scratch
};

namespace {
const int VQ_HIDDEN = 1024, VQ_LAYERS = 6, VQ_HEADS = 8, VQ_FFN = 1536;      // models/utils/config.py defaults

int vq_blocks(fdm_vq* V, const std::string& prefix, std::vector<Layer>* out, void* stream) {
  const int d = VQ_HIDDEN, dt = V->dtype;
  out->assign(VQ_LAYERS, Layer());
  const float* p = nullptr;
  for (int l = 0; l < VQ_LAYERS; ++l) {
    Layer& ly = (*out)[l];
    char a[128], m[128];
    snprintf(a, sizeof(a), "%s.net.%d.fn.", prefix.c_str(), 2 * l);
    snprintf(m, sizeof(m), "%s.net.%d.fn.", prefix.c_str(), 2 * l + 1);
    const std::string A(a), Mm(m);
    FCK(V->st.need(A + "norm.weight", d, &ly.ln1g)); FCK(V->st.need(A + "norm.bias", d, &ly.ln1b));
    FCK(V->st.need(A + "fn.to_qkv.weight", 3LL * d * d, &p));
    FCK(to_operand(V->mem, dt, p, 3LL * d * d, &ly.wqkv, stream));
    ly.bqkv = nullptr;
    FCK(V->st.need(A + "fn.to_out.weight", (long long)d * d, &p));
    FCK(to_operand(V->mem, dt, p, (long long)d * d, &ly.wo, stream));
    FCK(V->st.need(A + "fn.to_out.bias", d, &ly.bo));
    FCK(V->st.need(Mm + "norm.weight", d, &ly.ln2g)); FCK(V->st.need(Mm + "norm.bias", d, &ly.ln2b));
    FCK(V->st.need(Mm + "fn.l1.weight", (long long)VQ_FFN * d, &p));
    FCK(to_operand(V->mem, dt, p, (long long)VQ_FFN * d, &ly.w1, stream));
    FCK(V->st.need(Mm + "fn.l1.bias", VQ_FFN, &ly.b1));
    FCK(V->st.need(Mm + "fn.l2.weight", (long long)d * VQ_FFN, &p));
    FCK(to_operand(V->mem, dt, p, (long long)d * VQ_FFN, &ly.w2, stream));
    FCK(V->st.need(Mm + "fn.l2.bias", d, &ly.b2));
  }
  return FDM_OK;
}

int vq_conv_pack(fdm_vq* V, const std::string& name, Mat* out, void* stream) {
  const int d = VQ_HIDDEN;
  const float* p = nullptr;
  FCK(V->st.need(name, (long long)d * d * 5, &p));
  float* r = nullptr;
  FCK(V->mem.alloc_t(&r, (size_t)d * d * 5));
  hipLaunchKernelGGL(permute_oik_oki_kernel, dim3(grid_for((long long)d * d * 5)), dim3(256), 0, (hipStream_t)stream, p, r, d, d, 5);
  return to_operand(V->mem, V->front_dtype(), r, (long long)d * d * 5, out, stream);
}

int vq_commit(fdm_vq* V, void* stream) {
  if (V->committed) return FDM_OK;
  const fdm_vq_desc& q = V->d;
  const int d = VQ_HIDDEN, dt = V->front_dtype();
  const float* p = nullptr;
  FCK(V->st.need("quantize.embedding.weight", (long long)q.K * q.n_books * q.c, &V->codebook));
  if (q.pre) {
    FCK(V->st.need("decoder.decoder_linear_embedding_pre.net.weight", (long long)d * q.G * q.c, &p));
    FCK(to_operand(V->mem, dt, p, (long long)d * q.G * q.c, &V->pre_w, stream));
    FCK(V->st.need("decoder.decoder_linear_embedding_pre.net.bias", d, &V->pre_b));
  }
  FCK(vq_conv_pack(V, "decoder.expander.0.0.weight", &V->conv_w, stream));
  FCK(V->st.need("decoder.expander.0.0.bias", d, &V->conv_b));
  FCK(V->st.need("decoder.decoder_linear_embedding.net.weight", (long long)d * d, &p));
  FCK(to_operand(V->mem, dt, p, (long long)d * d, &V->emb_w, stream));
  FCK(V->st.need("decoder.decoder_linear_embedding.net.bias", d, &V->emb_b));
  // pe[0] = (sin 0, cos 0, ...) = (0, 1, 0, 1, ...): the reference indexes its positional table by BATCH position
  // (models/lib/base_models.py:300), so bs = 1 usage adds pe[0] to every frame of every clip
  std::vector<float> pe0(d);
  for (int i = 0; i < d; ++i) pe0[i] = (i & 1) ? 1.f : 0.f;
  FCK(V->mem.alloc_t(&V->pe0, (size_t)d));
  HIPCK(hipMemcpyAsync(V->pe0, pe0.data(), (size_t)d * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCK(hipStreamSynchronize((hipStream_t)stream));
  FCK(vq_blocks(V, "decoder.decoder_transformer", &V->dec_layers, stream));
  FCK(V->st.need("decoder.vertice_map_reverse.weight", (long long)q.V3 * d, &p));
  FCK(to_operand(V->mem, dt, p, (long long)q.V3 * d, &V->out_w, stream));
  V->out_b = nullptr;
  if (V->st.find("decoder.vertice_map_reverse.bias")) FCK(V->st.need("decoder.vertice_map_reverse.bias", q.V3, &V->out_b));
  // encoder (models/vq_vae_vocaset.py:134-191): optional, only needed for fdm_vq_encode
  V->has_encoder = V->st.find("encoder.vertice_mapping.0.weight") != nullptr;
  if (V->has_encoder) {
    V->Kp = (q.V3 + 63) / 64 * 64;           // K of the GEMM must be a multiple of the k-tile: zero-pad once
    FCK(V->st.need("encoder.vertice_mapping.0.weight", (long long)d * q.V3, &p));
    float* wp = nullptr;
    FCK(V->mem.alloc_t(&wp, (size_t)d * V->Kp));
    hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for((long long)d * V->Kp)), dim3(256), 0, (hipStream_t)stream, p, wp, (long long)d, q.V3, V->Kp);
    FCK(to_operand(V->mem, dt, wp, (long long)d * V->Kp, &V->e_map_w, stream));
    FCK(V->st.need("encoder.vertice_mapping.0.bias", d, &V->e_map_b));
    if (q.n_books > 1) {
      FCK(V->st.need("encoder.emotion_mapping.0.weight", (long long)d * 7, &V->e_emo_w));
      FCK(V->st.need("encoder.emotion_mapping.0.bias", d, &V->e_emo_b));
    }
    FCK(vq_conv_pack(V, "encoder.squasher.0.0.weight", &V->e_conv_w, stream));
    FCK(V->st.need("encoder.squasher.0.0.bias", d, &V->e_conv_b));
    FCK(V->st.need("encoder.encoder_linear_embedding.net.weight", (long long)d * d, &p));
    FCK(to_operand(V->mem, dt, p, (long long)d * d, &V->e_emb_w, stream));
    FCK(V->st.need("encoder.encoder_linear_embedding.net.bias", d, &V->e_emb_b));
    if (q.pre) {
      FCK(V->st.need("encoder.encoder_linear_embedding_post.net.weight", (long long)q.G * q.c * d, &p));
      FCK(to_operand(V->mem, dt, p, (long long)q.G * q.c * d, &V->e_post_w, stream));
      FCK(V->st.need("encoder.encoder_linear_embedding_post.net.bias", q.G * q.c, &V->e_post_b));
    }
    FCK(vq_blocks(V, "encoder.encoder_transformer", &V->enc_layers, stream));
  }
  HIPCK(hipGetLastError());
  V->committed = true;
  return FDM_OK;
}

int vq_reserve(fdm_vq* V, int B, int L) {
  if (B <= V->capB && L <= V->capL) return FDM_OK;
  B = B > V->capB ? B : V->capB; L = L > V->capL ? L : V->capL;
  (void)hipDeviceSynchronize();
  V->ws.release();
  const size_t d = VQ_HIDDEN, M = (size_t)B * L, es = esize(V->front_dtype());
  const size_t wide = d > (size_t)V->d.G * V->d.c ? d : (size_t)V->d.G * V->d.c;
  FCK(V->ws.alloc_t(&V->x32, M * wide)); FCK(V->ws.alloc(&V->xt, M * wide * es)); FCK(V->ws.alloc(&V->y, M * d * es));
  FCK(V->ws.alloc(&V->xp, (size_t)B * (L + 4) * d * es)); FCK(V->ws.alloc_t(&V->c32, M * d)); FCK(V->ws.alloc(&V->nt, M * d * es));
  FCK(V->ws.alloc_t(&V->h, M * d)); FCK(V->ws.alloc_t(&V->hb, M * d)); FCK(V->ws.alloc_t(&V->h2, M * d)); FCK(V->ws.alloc_t(&V->em, (size_t)B * d));
  FCK(V->ws.alloc(&V->q, M * d * es)); FCK(V->ws.alloc(&V->ctx, M * d * es)); FCK(V->ws.alloc(&V->a, M * d * es)); FCK(V->ws.alloc(&V->u, M * VQ_FFN * es));
  FCK(V->ws.alloc(&V->kp, (size_t)B * kv_pad(L) * d * es, true)); FCK(V->ws.alloc(&V->vp, (size_t)B * kv_pad(L) * d * es, true));
  FCK(V->ws.alloc_t(&V->book, (size_t)B));
  FCK(V->ws.alloc_t(&V->stat_partial, (size_t)1024)); FCK(V->ws.alloc_t(&V->stat_hist, (size_t)V->d.K));
  if (V->has_encoder) { FCK(V->ws.alloc_t(&V->xpad32, M * V->Kp)); FCK(V->ws.alloc(&V->xpt, M * V->Kp * es)); }
  V->capB = B; V->capL = L;
  return FDM_OK;
}

// 6 pre-LN blocks on the fp32 residual stream h [B*L, 1024] (updated in place)
int vq_transformer(fdm_vq* V, float* h, const std::vector<Layer>& layers, int B, int L, void* stream) {
  const int d = VQ_HIDDEN, dt = V->dtype, M = B * L, hd = d / VQ_HEADS, Lpad = kv_pad(L);
  const size_t es = esize(dt);
  HIPCK(hipMemsetAsync(V->kp, 0, (size_t)B * Lpad * d * es, (hipStream_t)stream));
  HIPCK(hipMemsetAsync(V->vp, 0, (size_t)B * Lpad * d * es, (hipStream_t)stream));
  // split kind: every operand of the loop is a plane pair, the lo plane one whole matrix after the hi plane
  const bool split = dt == FDM_F16X3;
  const long long lo_md = split ? (long long)M * d : 0, lo_mf = split ? (long long)M * VQ_FFN : 0, lo_kv = split ? (long long)B * Lpad * d : 0;
  auto lnorm = [&](const float* x, const float* gam, const float* bet) {
    fdm_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.M = M; a.d = d; a.gamma = gam; a.beta = bet; a.eps = 1e-5f; a.act = FDM_ACT_NONE; a.y_t = V->a; a.y_t_lo_off = lo_md; a.dtype = dt;
    return fdm_op_layernorm(&a, stream);
  };
  for (const Layer& ly : layers) {
    FCK(lnorm(h, ly.ln1g, ly.ln1b));
    fdm_gemm_args g = gemm_args(dt, V->a, ly.wqkv.p, M, 3 * d, d);
    g.a_lo_off = lo_md; g.w_lo_off = ly.wqkv.lo; g.out_t_lo_off = lo_md; g.kv_lo_off = lo_kv;
    g.out_t = V->q; g.ldo_t = d; g.out_kp = V->kp; g.kp_col0 = d; g.out_vp = V->vp; g.vp_col0 = 2 * d; g.kv_L = L; g.kv_Lpad = Lpad; g.kv_hd = hd;
    FCK(fdm_op_gemm(&g, stream));
    fdm_attn_args at;
    memset(&at, 0, sizeof(at));
    at.Q = V->q; at.ldq = d; at.Kp = V->kp; at.Vp = V->vp; at.Lpad = Lpad; at.O = V->ctx; at.ldo = d; at.B = B; at.H = VQ_HEADS; at.L = L; at.hd = hd;
    at.dtype = dt; at.scale = 1.0f / std::sqrt((float)d); at.causal = 0; at.period = 1;      // scale = hidden^-0.5 (base_models.py:144)
    at.q_lo_off = lo_md; at.kv_lo_off = lo_kv; at.o_lo_off = lo_md;
    FCK(fdm_op_attention(&at, stream));
    g = gemm_args(dt, V->ctx, ly.wo.p, M, d, d);
    g.a_lo_off = lo_md; g.w_lo_off = ly.wo.lo;
    g.bias = ly.bo; g.resid = h; g.out_f32 = V->hb;
    FCK(fdm_op_gemm(&g, stream));
    FCK(lnorm(V->hb, ly.ln2g, ly.ln2b));
    g = gemm_args(dt, V->a, ly.w1.p, M, VQ_FFN, d);
    g.a_lo_off = lo_md; g.w_lo_off = ly.w1.lo; g.out_t_lo_off = lo_mf;
    g.bias = ly.b1; g.act = FDM_ACT_GELU_TANH; g.out_t = V->u;
    FCK(fdm_op_gemm(&g, stream));
    g = gemm_args(dt, V->u, ly.w2.p, M, d, VQ_FFN);
    g.a_lo_off = lo_mf; g.w_lo_off = ly.w2.lo;
    g.bias = ly.b2; g.resid = V->hb; g.out_f32 = h;
    FCK(fdm_op_gemm(&g, stream));
  }
  return FDM_OK;
}

// Conv1d(k = 5, replicate) -> LeakyReLU -> InstanceNorm1d -> Linear + pe[0]; xt [B*L, 1024] operand kind -> V->h fp32
int vq_conv_norm_embed(fdm_vq* V, const void* xt, const Mat& conv_w, const float* conv_b, const Mat& emb_w, const float* emb_b, int B, int L, void* stream) {
  const int d = VQ_HIDDEN, dt = V->front_dtype(), M = B * L;
  FCK(fdm_op_pad_rows(xt, V->xp, B, L, d, 2, dt, 0, stream));
  fdm_gemm_args g = gemm_args(dt, V->xp, conv_w.p, L, d, 5 * d);
  g.lda = d; g.bias = conv_b; g.out_f32 = V->c32; g.batch = B; g.a_batch_stride = (long long)(L + 4) * d; g.out_batch_stride = (long long)L * d;
  FCK(fdm_op_gemm(&g, stream));
  FCK(fdm_op_leaky_instnorm(V->c32, nullptr, V->nt, B, L, d, 1e-5f, dt, stream));
  g = gemm_args(dt, V->nt, emb_w.p, M, d, d);
  g.bias = emb_b; g.resid = V->pe0; g.ldr = d; g.resid_row_mod = 1; g.out_f32 = V->h;
  return fdm_op_gemm(&g, stream);
}

int vq_operand(fdm_vq* V, const float* src32, void* dst, long long n, const void** out, void* stream) {
  if (V->front_dtype() == FDM_F32) { *out = src32; return FDM_OK; }
  *out = dst;
  return fdm_op_cast(src32, dst, n, V->front_dtype(), stream);
}

}  // namespace

extern "C" {

int fdm_vq_create(const fdm_vq_desc* desc, int dtype, fdm_vq** out) {
  if (!desc || !out) return fail(FDM_ERR_ARG, "vq_create: null argument");
  if (dtype != FDM_F32 && dtype != FDM_BF16 && dtype != FDM_F16X3)
    return fail(FDM_ERR_ARG, "vq_create: dtype %d (fp32, bf16, or FDM_F16X3 = split-fp16 transformer layers, everything else fp32)", dtype);
  if (desc->G <= 0 || desc->c <= 0 || desc->c > 128 || desc->K <= 0 || desc->n_books <= 0 || desc->V3 <= 0)
    return fail(FDM_ERR_SHAPE, "vq_create: inconsistent geometry (G %d, c %d, K %d, books %d, V3 %d)", desc->G, desc->c, desc->K, desc->n_books, desc->V3);
  if (!desc->pre && desc->G * desc->c != VQ_HIDDEN) return fail(FDM_ERR_SHAPE, "vq_create: decoder input width G*c = %d must equal %d when there is no pre-embedding", desc->G * desc->c, VQ_HIDDEN);
  if ((desc->G * desc->c) % 64) return fail(FDM_ERR_SHAPE, "vq_create: G*c = %d must be a multiple of 64", desc->G * desc->c);
  fdm_vq* V = new (std::nothrow) fdm_vq();
  if (!V) return fail(FDM_ERR_STATE, "vq_create: out of memory");
  V->d = *desc; V->dtype = dtype;
  *out = V;
  return FDM_OK;
}

int fdm_vq_destroy(fdm_vq* V) {
  if (!V) return FDM_OK;
  (void)hipDeviceSynchronize();
  V->ws.release(); V->mem.release(); V->st.mem.release();
  delete V;
  return FDM_OK;
}

int fdm_vq_set_weights(fdm_vq* V, const char* name, const float* ptr, long long n, void* stream) {
  if (!V || !name || !ptr || n <= 0) return fail(FDM_ERR_ARG, "vq_set_weights: bad argument");
  if (V->committed) return fail(FDM_ERR_STATE, "vq_set_weights: weights are frozen after the first call (create a new object)");
  return V->st.set(name, ptr, n, stream);
}

int fdm_vq_quant(fdm_vq* V, const float* z, const float* emo_one_hot, int B, int R, float* zq_bcl, long long* idx, void* stream) {
  if (!V || !z || !zq_bcl || !idx) return fail(FDM_ERR_ARG, "vq_quant: null argument");
  if (B < 1 || R < 1) return fail(FDM_ERR_SHAPE, "vq_quant: bad shape");
  if (V->d.n_books > 1 && !emo_one_hot) return fail(FDM_ERR_ARG, "vq_quant: this model needs the emotion one-hot to pick the codebook slice");
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "vq_quant: no gfx950 device visible (there is no CPU fallback)");
  FCK(vq_commit(V, stream));
  FCK(vq_reserve(V, B, 2));
  const int* book = nullptr;
  if (V->d.n_books > 1) {        // pos = argmax(one_hot) (models/vq_vae_emotion.py:223)
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, emo_one_hot, V->book, B, V->d.n_books);
    book = V->book;
  }
  return fdm_op_vq_quant(z, V->codebook, book, B, R, V->d.c, V->d.K, zq_bcl, idx, stream);
}

int fdm_vq_quant_stats(fdm_vq* V, const float* z, const float* emo_one_hot, const long long* idx, int B, int R, float beta,
                       float* min_encodings, float* out2, void* stream) {
  if (!V || !z || !idx || !out2) return fail(FDM_ERR_ARG, "vq_quant_stats: null argument");
  if (B < 1 || R < 1) return fail(FDM_ERR_SHAPE, "vq_quant_stats: bad shape");
  if (V->d.n_books > 1 && !emo_one_hot) return fail(FDM_ERR_ARG, "vq_quant_stats: this model needs the emotion one-hot to pick the codebook slice");
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "vq_quant_stats: no gfx950 device visible (there is no CPU fallback)");
  FCK(vq_commit(V, stream));
  FCK(vq_reserve(V, B, 2));
  const int* book = nullptr;
  if (V->d.n_books > 1) {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, emo_one_hot, V->book, B, V->d.n_books);
    book = V->book;
  }
  return fdm_op_vq_stats(z, V->codebook, book, idx, B, R, V->d.c, V->d.K, beta, min_encodings, V->stat_partial, V->stat_hist, out2, stream);
}

int fdm_vq_decode(fdm_vq* V, const float* zq_bcl, int B, int R, float* out, void* stream) {
  if (!V || !zq_bcl || !out) return fail(FDM_ERR_ARG, "vq_decode: null argument");
  const fdm_vq_desc& q = V->d;
  if (B < 1 || R < 1 || R % q.G) return fail(FDM_ERR_SHAPE, "vq_decode: bad quantised latent shape [%d, %d, %d]", B, q.c, R);
  const int L = R / q.G;
  if (L < 2) return fail(FDM_ERR_SHAPE, "vq_decode: needs at least 2 frames (InstanceNorm1d over one element is undefined in the reference)");
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "vq_decode: no gfx950 device visible (there is no CPU fallback)");
  FCK(vq_commit(V, stream));
  FCK(vq_reserve(V, B, L));
  const int d = VQ_HIDDEN, dt = V->front_dtype(), M = B * L, W = q.G * q.c;
  // [B, c, L*G] -> [B, L*G, c] == [B*L, G*c] (layout only; models/vq_vae_vocaset.py:37-40)
  hipLaunchKernelGGL(bcr_to_brc_kernel, dim3(grid_for((long long)B * q.c * R)), dim3(256), 0, (hipStream_t)stream, zq_bcl, V->x32, B, q.c, R);
  const void* xt = nullptr;
  FCK(vq_operand(V, V->x32, V->xt, (long long)M * W, &xt, stream));
  if (q.pre) {
    fdm_gemm_args g = gemm_args(dt, xt, V->pre_w.p, M, d, W);
    g.bias = V->pre_b;
    if (dt == FDM_F32) g.out_f32 = (float*)V->y; else g.out_t = V->y;
    FCK(fdm_op_gemm(&g, stream));
    xt = V->y;
  }
  FCK(vq_conv_norm_embed(V, xt, V->conv_w, V->conv_b, V->emb_w, V->emb_b, B, L, stream));
  FCK(vq_transformer(V, V->h, V->dec_layers, B, L, stream));
  const void* ht = nullptr;
  FCK(vq_operand(V, V->h, V->a, (long long)M * d, &ht, stream));
  fdm_gemm_args g = gemm_args(dt, ht, V->out_w.p, M, q.V3, d);
  g.bias = V->out_b; g.out_f32 = out;
  return fdm_op_gemm(&g, stream);
}

int fdm_vq_encode(fdm_vq* V, const float* x, const float* emo_one_hot, int B, int L, float* latent, void* stream) {
  if (!V || !x || !latent) return fail(FDM_ERR_ARG, "vq_encode: null argument");
  if (B < 1 || L < 2) return fail(FDM_ERR_SHAPE, "vq_encode: bad vertex tensor shape [%d, %d, .]", B, L);
  if (!fdm_device_ok()) return fail(FDM_ERR_STATE, "vq_encode: no gfx950 device visible (there is no CPU fallback)");
  FCK(vq_commit(V, stream));
  if (!V->has_encoder) return fail(FDM_ERR_STATE, "vq_encode: this object was built without encoder.* weights");
  const fdm_vq_desc& q = V->d;
  if (q.n_books > 1 && !emo_one_hot) return fail(FDM_ERR_ARG, "vq_encode: this model's encoder needs the emotion one-hot");
  FCK(vq_reserve(V, B, L));
  const int d = VQ_HIDDEN, dt = V->front_dtype(), M = B * L;
  hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for((long long)M * V->Kp)), dim3(256), 0, (hipStream_t)stream, x, V->xpad32, (long long)M, q.V3, V->Kp);
  const void* xp = nullptr;
  FCK(vq_operand(V, V->xpad32, V->xpt, (long long)M * V->Kp, &xp, stream));
  fdm_gemm_args g = gemm_args(dt, xp, V->e_map_w.p, M, d, V->Kp);
  g.bias = V->e_map_b; g.act = FDM_ACT_LEAKY02; g.out_f32 = V->h2;
  FCK(fdm_op_gemm(&g, stream));
  const float* h = V->h2;
  if (q.n_books > 1) {
    FCK(fdm_op_small_linear(emo_one_hot, V->e_emo_w, V->e_emo_b, V->em, B, 7, d, FDM_ACT_LEAKY02, stream));
    FCK(fdm_op_add_rows(V->h2, 1, M, V->em, L, B, nullptr, 1, 1, V->c32, M, d, stream));
    h = V->c32;
  }
  const void* ht = nullptr;
  FCK(vq_operand(V, h, V->y, (long long)M * d, &ht, stream));
  FCK(vq_conv_norm_embed(V, ht, V->e_conv_w, V->e_conv_b, V->e_emb_w, V->e_emb_b, B, L, stream));
  FCK(vq_transformer(V, V->h, V->enc_layers, B, L, stream));
  if (q.pre) {
    FCK(vq_operand(V, V->h, V->a, (long long)M * d, &ht, stream));
    g = gemm_args(dt, ht, V->e_post_w.p, M, q.G * q.c, d);
    g.bias = V->e_post_b; g.out_f32 = latent;
    return fdm_op_gemm(&g, stream);
  }
  HIPCK(hipMemcpyAsync(latent, V->h, (size_t)M * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return FDM_OK;
}

}  // extern "C"
