// Fused attention for short sequences (L <= ~1500, head_dim 64 / 128) on gfx950 MFMA.
//
// One workgroup (4 wavefronts) owns 16 query rows of one (clip, head).  The key tiles are dealt
// round-robin to the 4 waves, each running its own online softmax straight from L2 (K and V^T tiles
// are a few KB per head), and the four partial (m, l, O) states are merged through LDS at the end:
// with L <= 600 a query tile has at most 19 key tiles, so the per-wave dependent chain is <= 5 tiles
// instead of 19 -- these kernels are latency-bound, not FLOP-bound.
//
// "Swapped" formulation so that nothing has to be transposed between the two products:
//   S^T[key][query] = K * Q^T      A-port rows = keys,   B-port cols = queries
//   O^T[e][query]   = V^T * P^T    A-port rows = e (V^T is stored [hd][Lpad]), B-port = P^T
// The accumulator of the first product has query on the lane (lane & 15) and keys in the 4
// registers x 4 lane groups, which is exactly the B-port fragment of the second product; the
// softmax statistics of a query are therefore lane-local plus two cross-group shuffles
// (xor 16, 32) and the O^T accumulator is rescaled by a per-lane factor.
// For bf16 the key rows fed to the first product are permuted (row i -> key 8*(i>>2) + 4*s + (i&3))
// so that lane group g ends up holding keys 8g..8g+7 of the 32-key tile = the natural k order
// of v_mfma_f32_16x16x32_bf16; for fp32 (16-key tile, v_mfma_f32_16x16x4_f32) the order is natural.
//
// The causal + periodic-ALiBi bias of models/fdm_vocaset.py:95-116 is generated from (h, i, j)
// in-kernel: no [H, 600, 600] mask tensor is ever read.
#pragma once
#include "common.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

// accurate expf on the fp32 (parity) path, hardware exp2 on the bf16 (throughput) path
template <typename T> __device__ __forceinline__ float fexp(float x) {
  if constexpr (sizeof(T) == 4) return expf(x);
  else return __expf(x);
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_kernel(const fdm_attn_args p) {
  constexpr int EPC = 16 / (int)sizeof(T);     // elements per 16 B fragment chunk
  constexpr int NKS = HD / (4 * EPC);          // MFMA k-steps over the head dim
  constexpr int KT = 4 * EPC;                  // keys per tile (bf16 32, fp32 16)
  constexpr int NSUB = KT / 16;                // 16-key sub-tiles per tile
  constexpr int NC = HD / 16;                  // 16-row chunks of O^T

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int qt = blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int L = p.L;
  const int q0 = qt * 16;
  __shared__ __attribute__((aligned(16))) float part_o[4][16][HD];
  __shared__ float part_m[4][16], part_l[4][16];

  const T* Q = (const T*)p.Q + (size_t)b * L * p.ldq + (size_t)h * HD;
  const T* K = (const T*)p.K + (size_t)b * L * p.ldk + (size_t)h * HD;
  const T* Vt = (const T*)p.Vt + (size_t)(b * p.H + h) * HD * p.Lpad;

  const int qi = q0 + r16;                 // this lane's query index
  const int qrow = min(qi, L - 1);
  u32x4 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    qf[ks] = *(const u32x4*)(Q + (size_t)qrow * p.ldq + (ks * 4 + g) * EPC);

  f32x4 o[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;

  const float slope = p.slopes ? p.slopes[h] : 0.f;
  const int kend = p.causal ? min(q0 + 16, L) : L;     // keys [0, kend) can be visible to this tile
  const int ntiles = (kend + KT - 1) / KT;

  // Measured: software prefetch of the next tile's fragments (register double buffers, copy or ping-pong) is
  // SLOWER here (9.1 / 10.4 us vs 8.4 us at cfg2): the extra 64 VGPRs cost residency, and residency is what hides
  // the L2 round trips of these short per-wave chains.  So each tile simply loads its fragments and uses them.
  const float inv_period = 1.f / (float)p.period;

  for (int kt = wave; kt < ntiles; kt += 4) {
    const int kbase = kt * KT;
    u32x4 kcur[NSUB][NKS];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      // key fed by A-port row r16 of sub-tile s
      const int krow = (NSUB == 2) ? (kbase + 8 * (r16 >> 2) + 4 * s + (r16 & 3)) : (kbase + r16);
      const T* kp = K + (size_t)min(krow, L - 1) * p.ldk;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) kcur[s][ks] = *(const u32x4*)(kp + (ks * 4 + g) * EPC);
    }
    u32x4 vf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) vf[c] = *(const u32x4*)(Vt + (size_t)(c * 16 + r16) * p.Lpad + kbase + g * EPC);
    f32x4 sc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) Mma<T>::run(a, kcur[s][ks], qf[ks]);
      sc[s] = a;
    }
    // scores -> scaled, biased, masked; this lane holds keys kbase + (NSUB==2 ? 8g+4s+r : 4g+r)
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = kbase + ((NSUB == 2) ? (8 * g + 4 * s + r) : (4 * g + r));
        float v = sc[s][r] * p.scale;
        // floor((qi - kj) / period) without an integer divide: (n + 0.5) / period never rounds across an integer
        if (p.slopes) v -= slope * floorf(((float)(qi - kj) + 0.5f) * inv_period);
        const bool masked = (kj >= L) || (p.causal && kj > qi);
        v = masked ? -INFINITY : v;
        sc[s][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // m_new is finite: key 0 is visible to every query in tile 0 (causal) / every key < L (non-causal)
    const float alpha = fexp<T>(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = fexp<T>(sc[s][r] - m_new);
        sc[s][r] = e;
        psum += e;
      }
    l_part = l_part * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] *= alpha;

    // P^T fragment for the B port
    u32x4 pf;
    if constexpr (sizeof(T) == 2) {
      bf16x8 pb = {(bf16)sc[0][0], (bf16)sc[0][1], (bf16)sc[0][2], (bf16)sc[0][3],
                   (bf16)sc[NSUB - 1][0], (bf16)sc[NSUB - 1][1], (bf16)sc[NSUB - 1][2], (bf16)sc[NSUB - 1][3]};
      pf = __builtin_bit_cast(u32x4, pb);
    } else {
      pf = __builtin_bit_cast(u32x4, sc[0]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) Mma<T>::run(o[c], vf[c], pf);
  }

  float l_tot = l_part;
  l_tot += __shfl_xor(l_tot, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  // ---- merge the four waves' partial states ----
  if (g == 0) { part_m[wave][r16] = m_run; part_l[wave][r16] = l_tot; }
#pragma unroll
  for (int c = 0; c < NC; ++c) *(f32x4*)&part_o[wave][r16][c * 16 + 4 * g] = o[c];
  __syncthreads();
  {
    const int q = threadIdx.x >> 4;                  // 16 threads per query row
    const int e0 = (threadIdx.x & 15) * (HD / 16);   // HD/16 consecutive head-dim elements per thread
    const int qq = q0 + q;
    float mw[4], ms = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = part_m[w][q]; ms = fmaxf(ms, mw[w]); }
    float sw[4], lsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sw[w] = fexp<T>(mw[w] - ms); lsum += sw[w] * part_l[w][q]; }   // exp(-inf) = 0 for idle waves
    const float inv = 1.f / lsum;
    if (qq < L) {
      T* op = (T*)p.O + ((size_t)b * L + qq) * p.ldo + (size_t)h * HD + e0;
#pragma unroll
      for (int j = 0; j < HD / 16; j += 4) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) v += *(const f32x4*)&part_o[w][q][e0 + j] * sw[w];
        v *= inv;
        if constexpr (sizeof(T) == 4) {
          *(f32x4*)(op + j) = v;
        } else {
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 ob = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(op + j) = ob;
        }
      }
    }
  }
}

static hipError_t attn_launch(const fdm_attn_args& a, hipStream_t s) {
  dim3 grid((a.L + 15) / 16, a.H, a.B);
  dim3 block(256);
  if (a.dtype == FDM_BF16) {
    if (a.hd == 256) hipLaunchKernelGGL((attn_kernel<bf16, 256>), grid, block, 0, s, a);
    else if (a.hd == 128) hipLaunchKernelGGL((attn_kernel<bf16, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<bf16, 64>), grid, block, 0, s, a);
  } else {
    if (a.hd == 256) hipLaunchKernelGGL((attn_kernel<float, 256>), grid, block, 0, s, a);
    else if (a.hd == 128) hipLaunchKernelGGL((attn_kernel<float, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<float, 64>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace fdm
