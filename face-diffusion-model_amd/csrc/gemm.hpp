// C[M,N] = epilogue(A[M,K] * W[N,K]^T) on MFMA, for gfx950.
//
// Layout / tiling
//  * A and W are both K-contiguous ("NT" GEMM: nn.Linear weights are [out, in]).
//  * Workgroup = WM x WN wavefronts, tile BM x BN, k-tile = 128 bytes of K per row (64 16-bit / 32 fp32
//    elements), so the LDS image and the staging code are byte-identical for every operand kind.  LDS rows
//    are 128 B with a 16-byte-chunk XOR swizzle (chunk ^ (row & 7)) so that the ds_read_b128 fragment reads
//    of 16 different rows at one k-chunk spread over 8 slots.
//  * Operand kinds (common.hpp, Opnd<T>): fp32 (exact, v_mfma_f32_16x16x4_f32), bf16 (v_mfma_f32_16x16x32_bf16),
//    and the split kind f16x3: hi and lo planes of both operands ride the same ring (a stage holds
//    A_hi, A_lo, W_hi, W_lo) and every fragment pair takes three 16-bit MFMAs (hi.hi into the main accumulator,
//    hi.lo + lo.hi into a second one that is scaled by 1 / SCALE once, after the k loop).
//  * Operands are swapped into the MFMA (W rows feed the A port, activation rows the B port), so a
//    lane ends up with 4 consecutive output columns n of one row m: the epilogue then reads
//    bias/residual and writes outputs with 16-byte (fp32) / 8-byte (16-bit) vector accesses.
//  * Epilogue is fused: bias, activation, residual add, dual-dtype stores, transposed "V^T" scatter.
//  * One kernel template, several instantiations per tile: what a launch is timed by on this path is its fixed cost (boundary,
//    cold kernel arguments, cold first tile, cold instruction fetch -- tools/gemm_timeline.py), so kernels are specialised
//    by what the launch can need (SPEC: lean / packed K,V / LayerNorm fold; HEAVY: transcendental activations; SCHED: the
//    fused scheduler update) and the general edge-handling kernel serves only the shapes that need it.
#pragma once
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "sched.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

// LDS scratch of the LayerNorm-folding epilogue behind the ring: rowstat[BM][2] (mu, rstd of this block's rows).  The
// producer side's per-fragment (sum, sum of squares) pairs [BN / 16][BM][2] live INSIDE the ring (free after the k
// loop), behind the area of the transposed V tile.
template <int BM, int BN> constexpr int gemm_ln_scratch_bytes() { return BM * 2 * 4; }
template <typename T, int BM, int BN> constexpr int gemm_epi_ring_bytes() {
  using KK = typename Opnd<T>::KV;
  return BM * BN * (int)sizeof(typename Opnd<KK>::E) * Opnd<KK>::NP + BM * 2 * (BN / 16) * 4;
}

// Consumer side: mu / rstd of the block's rows from the producer's per-64-column partial sums (fixed order).
template <int BM>
__device__ __forceinline__ void gemm_load_rowstats(const fdm_gemm_args& p, int m0, float* rowstat) {
  if (!p.ln_stat_in) return;
  const int t = threadIdx.x;
  if (t < BM) {
    const int m = min(m0 + t, p.M - 1);
    float s = 0.f, q = 0.f;
    // issue all partial loads before the first use (one L2 round trip instead of ln_nparts of them)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    for (int i0 = 0; i0 < p.ln_nparts; i0 += 16) {
      f32x2 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ii = min(i0 + i, p.ln_nparts - 1);
        v[i] = *(const f32x2*)(p.ln_stat_in + ((size_t)ii * p.M + m) * 2);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i0 + i < p.ln_nparts) { s += v[i][0]; q += v[i][1]; }
    }
    const float inv = 1.f / (float)p.ln_dim;
    const float mu = s * inv;
    const float var = fmaxf(q * inv - mu * mu, 0.f);
    rowstat[2 * t] = mu;
    rowstat[2 * t + 1] = 1.f / sqrtf(var + p.ln_eps);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes are ordered by the k loop's first raw s_barrier
}

// Everything the epilogue reads from memory (bias, folded-LayerNorm vectors, the residual tile) is fetched by
// gemm_epi_preload BEFORE the k loop, so the epilogue is compute + stores only.  Left inside the epilogue these loads
// sit behind the stores of earlier fragments (the output pointers may alias the inputs as far as the compiler knows),
// one exposed load latency per (mi, ni) fragment: measured 11 us of fixed cost on a 96x128 tile, ~1 us on 64x64.
// They are requested right after the first ring tiles (older than every later tile), so the k loop's counted vmcnt waits
// cover them too.
template <int MI, int NI> struct EpiPre { f32x4 csv[NI], bv[NI], gmv[NI], btv[NI], rv[MI][NI]; SchedCoef sc; };

template <typename T, int BM, int BN, int WM, int WN, bool SCHED = false>
__device__ __forceinline__ void gemm_epi_preload(const fdm_gemm_args& p, int m0, int n0, int z, int wm, int wn, int g, int r16,
                                                 bool ln_capable, EpiPre<BM / WM / 16, BN / WN / 16>& e) {
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  if constexpr (SCHED) e.sc = sched_coef_load(p.sched);     // k -> t -> table entries: three dependent loads, hidden by the k loop
  const int M = p.M, N = p.N;
  const bool use_ln = ln_capable && p.ln_stat_in;
  const float* bias = p.bias ? p.bias + (size_t)z * p.bias_batch_stride : nullptr;
  const size_t ocol = (size_t)z * p.out_batch_stride;
  const bool vec_r = p.resid && (p.ldr % 4 == 0) && (((uintptr_t)(p.resid + ocol)) % 16 == 0);
  const bool pre_cs = use_ln && p.ln_colsum, pre_rln = use_ln && p.rln_gamma && p.resid;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (BN / WN) + ni * 16 + 4 * g;
    const bool full = (n + 3 < N);
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    e.csv[ni] = (pre_cs && full) ? *(const f32x4*)(p.ln_colsum + n) : zero;
    e.bv[ni] = (bias && full) ? *(const f32x4*)(bias + n) : zero;
    e.gmv[ni] = (pre_rln && full) ? *(const f32x4*)(p.rln_gamma + n) : zero;
    e.btv[ni] = (pre_rln && full) ? *(const f32x4*)(p.rln_beta + n) : zero;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * (BM / WM) + mi * 16 + r16;
      const size_t rrow = p.resid_row_mod > 0 ? (size_t)(min(m, M - 1) % p.resid_row_mod) : (size_t)min(m, M - 1);
      e.rv[mi][ni] = (p.resid && vec_r && full && m < M) ? *(const f32x4*)(p.resid + ocol + rrow * p.ldr + n) : zero;
    }
  }
}

// HEAVY = the activation may be one of the transcendental ones (Mish, GELU): their libm expansions are hundreds of
// instructions per element, so kernels for the plain / ReLU / LeakyReLU GEMMs (33 of the step's 34) are built without.
template <bool HEAVY, typename T> __device__ __forceinline__ float gemm_act(float v, int act) {
  if constexpr (!HEAVY) {
    return act == ACT_RELU ? fmaxf(v, 0.f) : (act == ACT_LEAKY02 ? (v > 0.f ? v : 0.2f * v) : v);
  } else if constexpr (is_fast16<T>::value) {
    // bf16 / fp16 (throughput) modes: hardware exp / log forms, ~1e-6 relative -- far inside the bf16 operand rounding
    if (act == ACT_MISH) {
      const float sp = v > 20.f ? v : __logf(1.f + __expf(v));
      return v * (1.f - 2.f / (1.f + __expf(2.f * sp)));
    }
    if (act == ACT_GELU_TANH) {
      const float c = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return v * (0.5f * (2.f - 2.f / (1.f + __expf(2.f * c))));
    }
    return act_apply_t<T>(v, act);      // (GELU(erf): the fast form too)
  } else {
    return act_apply(v, act);     // fp32 and split (parity) modes: accurate libm forms
  }
}
__host__ __device__ inline bool gemm_act_is_heavy(int act) { return act == ACT_MISH || act == ACT_GELU_ERF || act == ACT_GELU_TANH; }

// LEAN = every tile of the launch takes the straight-line path (the host checked it: gemm_all_tiles_lean), so the general
// edge-handling path -- two thirds of the kernel's instructions, never executed by the step's GEMMs but in the way of the
// instruction fetch -- is not compiled in.
// SPEC further says what the launch can need: GEMM_KV = packed K / V outputs, GEMM_FOLD = the LayerNorm-folding producer /
// consumer forms; a lean kernel without either is bias + activation + residual + stores (a few hundred instructions).
constexpr int GEMM_LEAN = 1, GEMM_KV = 2, GEMM_FOLD = 4, GEMM_KSPLIT = 8, GEMM_B2 = 16;
// GEMM_KSPLIT (with GEMM_LEAN): blockIdx.z is a K slice, not a batch index -- slice s runs k-tiles [s nk / S, (s + 1) nk / S) and
// stores its fp32 partial tile to plane s of out_f32; slice 0 alone adds bias and residual (fdm_gemm_args.ksplit).  The
// epilogue sees the slice's view of the arguments:
// GEMM_B2: a second batch level (fdm_gemm_args.batch2): the epilogue's outputs and residual advance by out_batch_stride2 per c.
template <int MODE> struct KSliceArgs;      // MODE: 0 plain, 1 K slice, 2 second batch level
template <> struct KSliceArgs<0> {
  const fdm_gemm_args& a;
  __device__ __forceinline__ KSliceArgs(const fdm_gemm_args& p, int, int) : a(p) {}
};
template <> struct KSliceArgs<1> {
  fdm_gemm_args a;
  __device__ __forceinline__ KSliceArgs(const fdm_gemm_args& p, int slice, int) : a(p) {
    a.out_f32 = p.out_f32 + (size_t)slice * p.ksplit_stride;
    if (slice > 0) { a.bias = nullptr; a.resid = nullptr; }
  }
};
template <> struct KSliceArgs<2> {
  fdm_gemm_args a;
  __device__ __forceinline__ KSliceArgs(const fdm_gemm_args& p, int c, int out_t_elem_bytes) : a(p) {
    const size_t off = (size_t)c * p.out_batch_stride2;
    if (p.out_f32) a.out_f32 = p.out_f32 + off;
    if (p.resid) a.resid = p.resid + off;
    if (p.out_t) a.out_t = (char*)p.out_t + off * out_t_elem_bytes;
  }
};
template <typename T, int BM, int BN, int WM = 2, int WN = 2, bool HEAVY = true, bool SCHED = false, int SPEC = 0>
__device__ __forceinline__ void gemm_epilogue(const fdm_gemm_args& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16],
                                              const EpiPre<BM / WM / 16, BN / WN / 16>& e, int m0, int n0, int z,
                                              int wm, int wn, int g, int r16, float* rowstat = nullptr, char* tile_lds = nullptr) {
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr bool LEAN = (SPEC & GEMM_LEAN) != 0;
  constexpr bool KVC = !LEAN || (SPEC & GEMM_KV), FOLDC = !LEAN || (SPEC & GEMM_FOLD);     // capabilities compiled in
  using E = typename Opnd<T>::E;      // element type of out_t (split kinds: two planes of it)
  using KK = typename Opnd<T>::KV;    // operand kind of the packed K / V outputs (a plane pair for f16x3)
  using KV = typename Opnd<KK>::E;    // their element type
  constexpr int KNP = Opnd<KK>::NP;
  const int M = p.M, N = p.N;
  // Whole-tile packed-V fast path: stage the tile transposed in LDS ([n][m], the ring is free after the k loop) and
  // store 16-byte runs of consecutive keys (one packed chunk each).  Needs the tile to lie entirely in the V columns
  // and clip boundaries on 16-byte multiples (L % (16 / sizeof(KV)) == 0); otherwise the per-element scatter is used.
  constexpr int EPC_T = 16 / (int)sizeof(KV);
  const bool vt_tile = KVC && tile_lds && p.out_vp && n0 >= p.vp_col0 && n0 + BN <= N && (p.kv_L % EPC_T == 0);
  KV* tl = (KV*)tile_lds;
  float* comb = tile_lds ? (float*)(tile_lds + BM * BN * (int)sizeof(KV) * KNP) : nullptr;
  const bool use_ln = FOLDC && rowstat && p.ln_stat_in;
  const bool do_stat = FOLDC && rowstat && comb && p.stat_out;
  if (vt_tile || do_stat) __syncthreads();      // every wave is done reading the last ring stage
  // ---- fused epilogue: lane owns C[m = .. + r16][n = .. + 4g + (0..3)] ----
  const float* bias = p.bias ? p.bias + (size_t)z * p.bias_batch_stride : nullptr;
  const size_t ocol = (size_t)z * p.out_batch_stride;
  const bool vec_f32 = p.out_f32 && (p.ldo_f32 % 4 == 0) && (((uintptr_t)(p.out_f32 + ocol)) % 16 == 0);
  const bool vec_t = p.out_t && (p.ldo_t % 4 == 0) && (((uintptr_t)((E*)p.out_t + ocol)) % (4 * sizeof(E)) == 0) &&
                     (Opnd<T>::NP == 1 || p.out_t_lo_off % 4 == 0);
  const bool vec_r = p.resid && (p.ldr % 4 == 0) && (((uintptr_t)(p.resid + ocol)) % 16 == 0);
  const int kv_H = p.out_vp ? (N - p.vp_col0) / p.kv_hd : 0;
  const size_t kv_blk = (size_t)p.kv_Lpad * p.kv_hd;        // elements per (clip, head) block of the packed buffers
  const int kcol_lo = p.out_kp ? p.kp_col0 : N, kcol_hi = p.out_kp ? (p.out_vp ? p.vp_col0 : N) : N;
  const bool pre_cs = use_ln && p.ln_colsum, pre_rln = use_ln && p.rln_gamma && p.resid;
  const f32x4 (&csv)[NI] = e.csv; const f32x4 (&bv)[NI] = e.bv; const f32x4 (&gmv)[NI] = e.gmv; const f32x4 (&btv)[NI] = e.btv;
  const f32x4 (&rv)[MI][NI] = e.rv;
  // Interior tiles (no column edge, vectorisable pointers, the whole tile in one of the Q / K / V column ranges) take a
  // lean straight-line path.  The general path below handles every edge case but is ~1000 instructions of branches per
  // fragment: measured, the epilogue's code size IS the kernel's fixed cost (~0.3 us per 1000 lines of ISA on top of the
  // launch floor: 4.5 us for this 64x64 kernel, 16 us for a 128x128 tile on 4 waves), so the common case must be short.
  const int kv_mode = (!KVC || (!p.out_kp && !p.out_vp)) ? 0
                    : (n0 + BN <= (p.out_kp ? kcol_lo : p.vp_col0)) ? 0
                    : (p.out_kp && n0 >= kcol_lo && n0 + BN <= kcol_hi) ? 1
                    : vt_tile ? 2
                    : (p.out_vp && n0 >= p.vp_col0 && n0 + BN <= N) ? 3 : -1;      // 3: V columns, clip length not in whole packed chunks
  const bool lean = LEAN || (kv_mode >= 0 && n0 + BN <= N && (vec_f32 || !p.out_f32) && (vec_t || !p.out_t) && (vec_r || !p.resid) &&
                            (p.kv_hd % 16 == 0 || kv_mode != 1));
  if (lean) {
    const int ncol = n0 + wn * (BN / WN) + 4 * g;            // this lane's column in fragment ni = 0
    const int KH = (kv_mode == 1) ? (kcol_hi - kcol_lo) / p.kv_hd : 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int lrow = wm * (BM / WM) + mi * 16 + r16;
      const int m = m0 + lrow;
      if (m >= M) continue;
      const float mu = use_ln ? rowstat[2 * lrow] : 0.f, rs = use_ln ? rowstat[2 * lrow + 1] : 1.f;
      float* o32 = p.out_f32 ? p.out_f32 + ocol + (size_t)m * p.ldo_f32 + ncol : nullptr;
      E* ot = p.out_t ? (E*)p.out_t + ocol + (size_t)m * p.ldo_t + ncol : nullptr;
      KV* okp = nullptr;
      int kv_l = 0;
      if (kv_mode == 1) {
        const int kv_b = m / p.kv_L;
        kv_l = m - kv_b * p.kv_L;
        okp = (KV*)p.out_kp + (size_t)kv_b * KH * kv_blk;
      } else if (kv_mode == 3) {
        const int kv_b = m / p.kv_L;
        kv_l = m - kv_b * p.kv_L;
        okp = (KV*)p.out_vp + (size_t)kv_b * kv_H * kv_blk;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x4 v = acc[mi][ni];
        if (pre_cs) v = (v - mu * csv[ni]) * rs;
        if (bias) v += bv[ni];
        if (p.act != ACT_NONE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gemm_act<HEAVY, T>(v[j], p.act);
        }
        if constexpr (SCHED) {
          // fused scheduler: v = x0_hat, rv = x_t (same element), flat element index of the chain's x buffer
          v = sched_update4(p.sched, e.sc, v, rv[mi][ni], (long long)m * N + (ncol + ni * 16));
        } else {
          if (p.resid) v += pre_rln ? (rv[mi][ni] - mu) * rs * gmv[ni] + btv[ni] : rv[mi][ni];
        }
        if (do_stat) {      // canonical order (tile-independent): 4 columns per lane, 4 lane groups, then fragments in column order
          const float fs = rows_sum((v[0] + v[1]) + (v[2] + v[3]));
          const float fq = rows_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
          if (g == 0) { float* cb = comb + ((size_t)(wn * NI + ni) * BM + lrow) * 2; cb[0] = fs; cb[1] = fq; }
        }
        if (kv_mode == 2) {
          const int nl = wn * (BN / WN) + ni * 16 + 4 * g;
#pragma unroll
          for (int j = 0; j < 4; ++j) store_opnd1<KK>(tl + (nl + j) * BM + lrow, BM * BN, v[j]);
          continue;
        }
        if (kv_mode == 1) {
          const int cc = ncol + ni * 16 - kcol_lo;
          const int h = cc / p.kv_hd, e2 = cc - h * p.kv_hd;
          store_opnd4<KK>(okp + (size_t)h * kv_blk + kp_offset<KV>(kv_l, e2, p.kv_hd), p.kv_lo_off, v);
          continue;
        }
        if (kv_mode == 3) {       // per-element scatter into the packed V^T layout (the lane's 4 columns share a head: kv_hd % 16 == 0)
          const int cc = ncol + ni * 16 - p.vp_col0;
          const int h = cc / p.kv_hd, e2 = cc - h * p.kv_hd;
#pragma unroll
          for (int j = 0; j < 4; ++j) store_opnd1<KK>(okp + (size_t)h * kv_blk + vp_offset<KV>(kv_l, e2 + j, p.kv_hd), p.kv_lo_off, v[j]);
          continue;
        }
        if (o32) st16(o32 + ni * 16, v);
        if (ot) store_opnd4<T>(ot + ni * 16, p.out_t_lo_off, v);
      }
    }
  } else if constexpr (!LEAN) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int lrow = wm * (BM / WM) + mi * 16 + r16;
    const int m = m0 + lrow;
    if (m >= M) continue;
    const size_t rrow = p.resid_row_mod > 0 ? (size_t)(m % p.resid_row_mod) : (size_t)m;
    const float mu = use_ln ? rowstat[2 * lrow] : 0.f, rs = use_ln ? rowstat[2 * lrow + 1] : 1.f;
    int kv_b = 0, kv_l = 0;
    if (p.out_kp || p.out_vp) { kv_b = m / p.kv_L; kv_l = m - kv_b * p.kv_L; }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (BN / WN) + ni * 16 + 4 * g;
      if (n >= N) continue;
      f32x4 v = acc[mi][ni];
      const bool full = (n + 3 < N);
      if (pre_cs) {      // LN(x) W^T == rstd (x W'^T - mu colsum(W'))
        if (full) {
          v = (v - mu * csv[ni]) * rs;
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] = (v[j] - mu * p.ln_colsum[n + j]) * rs;
        }
      }
      if (bias) {
        if (full) {
          v += bv[ni];
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] += bias[n + j];
        }
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gemm_act<HEAVY, T>(v[j], p.act);
      }
      if (p.resid) {
        const float* rp = p.resid + ocol + rrow * p.ldr + n;
        if (pre_rln) {    // residual = LayerNorm(raw row) computed on the fly
          if (full && vec_r) {
            v += (rv[mi][ni] - mu) * rs * gmv[ni] + btv[ni];
          } else {
            for (int j = 0; j < 4; ++j)
              if (n + j < N) v[j] += (rp[j] - mu) * rs * p.rln_gamma[n + j] + p.rln_beta[n + j];
          }
        } else if (full && vec_r) {
          v += rv[mi][ni];
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] += rp[j];
        }
      }
      if (do_stat) {      // same canonical order as the lean path (columns past N contribute zeros)
        float w4[4];
        for (int j = 0; j < 4; ++j) w4[j] = (n + j < N) ? v[j] : 0.f;
        const float fs = rows_sum((w4[0] + w4[1]) + (w4[2] + w4[3]));
        const float fq = rows_sum((w4[0] * w4[0] + w4[1] * w4[1]) + (w4[2] * w4[2] + w4[3] * w4[3]));
        if (g == 0) { float* cb = comb + ((size_t)(wn * NI + ni) * BM + lrow) * 2; cb[0] = fs; cb[1] = fq; }
      }
      if (vt_tile) {
        const int nl = wn * (BN / WN) + ni * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) store_opnd1<KK>(tl + (nl + j) * BM + lrow, BM * BN, v[j]);
        continue;
      }
      if (n >= kcol_lo && n < kcol_hi) {
        // packed K: this lane's 4 columns sit inside one 16-byte chunk of key kv_l (kv_hd % 4 == 0, column ranges % 4 == 0)
        const int cc = n - kcol_lo;
        const int h = cc / p.kv_hd, e = cc - h * p.kv_hd;
        const int KH = (kcol_hi - kcol_lo) / p.kv_hd;
        store_opnd4<KK>((KV*)p.out_kp + (size_t)(kv_b * KH + h) * kv_blk + kp_offset<KV>(kv_l, e, p.kv_hd), p.kv_lo_off, v);
        continue;
      }
      if (p.out_vp && n >= p.vp_col0) {
        for (int j = 0; j < 4; ++j) {
          if (n + j >= N) break;
          const int cc = n + j - p.vp_col0;
          const int h = cc / p.kv_hd, e = cc - h * p.kv_hd;
          store_opnd1<KK>((KV*)p.out_vp + (size_t)(kv_b * kv_H + h) * kv_blk + vp_offset<KV>(kv_l, e, p.kv_hd), p.kv_lo_off, v[j]);
        }
        continue;
      }
      if (p.out_f32) {
        float* op = p.out_f32 + ocol + (size_t)m * p.ldo_f32 + n;
        if (full && vec_f32) {
          *(f32x4*)op = v;
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) op[j] = v[j];
        }
      }
      if (p.out_t) {
        E* op = (E*)p.out_t + ocol + (size_t)m * p.ldo_t + n;
        if (full && vec_t) {
          store_opnd4<T>(op, p.out_t_lo_off, v);
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) store_opnd1<T>(op + j, p.out_t_lo_off, v[j]);
        }
      }
    }
  }
  }
  if (vt_tile) {
    __syncthreads();
    constexpr int CPR = BM / EPC_T;                       // 16-byte chunks per tile column
    for (int c = threadIdx.x; c < BN * CPR; c += WM * WN * 64) {
      const int nl = c / CPR, ml = (c % CPR) * EPC_T;
      const int m = m0 + ml;
      if (m >= M) continue;                               // M is a multiple of L, L of EPC_T: chunks are all-in or all-out
      const int b = m / p.kv_L, l = m - b * p.kv_L;
      const int cc = n0 + nl - p.vp_col0;
      const int h = cc / p.kv_hd, e = cc - h * p.kv_hd;
#pragma unroll
      for (int pl = 0; pl < KNP; ++pl)
        *(u32x4*)((KV*)p.out_vp + (size_t)pl * p.kv_lo_off + (size_t)(b * kv_H + h) * kv_blk + vp_offset<KV>(l, e, p.kv_hd)) =
            *(const u32x4*)(tl + pl * BM * BN + nl * BM + ml);
    }
  }
  if (do_stat) {
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BM && m0 + t < M) {
      // one (sum, sum of squares) slot per 64-column group: its four 16-column fragments in column order, whatever the
      // tile / wave layout that produced them -> the folded LayerNorm statistics do not depend on the tile choice
#pragma unroll
      for (int gq = 0; gq < BN / 64; ++gq) {
        if (n0 + gq * 64 >= N) break;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) { s += comb[((size_t)(gq * 4 + f) * BM + t) * 2]; q += comb[((size_t)(gq * 4 + f) * BM + t) * 2 + 1]; }
        float* so = p.stat_out + ((size_t)(n0 / 64 + gq) * M + (m0 + t)) * 2;
        so[0] = s; so[1] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Main loop: HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging) into an NST-stage
// ring, tiles prefetched NST-1 deep, counted s_waitcnt vmcnt (never 0 in steady state), one raw
// s_barrier per k-tile.  The LDS image of a stage is lane-linear (a wave instruction writes 1 KiB =
// 8 rows x 128 B), so the bank-conflict swizzle is applied to the SOURCE chunk (slot ^ (row & 7)) and
// undone by the same XOR on the ds_read side.
// The step's GEMMs have M = B*L of a few hundred rows: a block owns one tile for the whole K loop, so
// what bounds it is the latency chain of its own k-tiles, not bandwidth -- hence the deep ring.
// ---------------------------------------------------------------------------------------------------
// FDM_LW_VARIANT (tools/lw_probe.cpp only; 0 in every build of the library): the k loop of gemm_glds_kernel with parts removed -- bit 0 no
// MFMAs, bit 1 no fragment reads, bit 2 no LDS-DMA after the prologue's NST - 1 tiles.  Results of the reduced variants are meaningless;
// only their durations are read (round 6: which phase bounds the loader-wave loop at 1992-3984 rows).
#ifndef FDM_LW_VARIANT
#define FDM_LW_VARIANT 0
#endif
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LOADER WAVES (round 5, LW > 0).  Split K showed that the step's k loop is bounded per CU by throughput, not by the latency of
// its tiles: per k-tile every wave issues its LDS-DMA pieces (each costs the ISSUING wave 60-185 cycles of back-pressure from the
// memory pipe, MI355X_MICROARCH.md), reads its fragments and runs its MFMAs, in lockstep behind one barrier.  With LW > 0 the
// WM x WN compute waves never issue a tile load: LW extra waves do nothing but wait for their pieces of tile kt (counted vmcnt),
// meet the compute waves at the k-tile's barrier and request tile kt + NST - 1 -- the back-pressure stalls loader waves only, and
// the loop runs at the L2 -> LDS rate of the CU instead of at the sum of its phases.  Same LDS image, same k order, same
// epilogue: bit-identical to LW = 0 (FDM_TILE_LOCKSTEP selects it for A/B and tests).
template <typename T, int BM, int BN, int WM, int WN, int NST, int KCH = 8, bool HEAVY = false, bool SCHED = false, int SPEC = 0, int LW = 0>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void gemm_glds_kernel(const void* pA, const void* pW, long long p_a_lo_off, long long p_w_lo_off,
                                                                        int pM, int pN, int pK, int p_lda, int p_ldw, const fdm_gemm_args p) {
  // The nine leading arguments repeat fields of p: they are the 13 SGPRs of kernel-argument PRELOAD (the build passes
  // -amdgpu-kernarg-preload-count=13; gfx950 fills them at wave launch), everything the first tile requests need -- so the first
  // LDS-DMA instructions go out without waiting for a scalar load of the argument block (cold at every launch: ~0.5 us).
  // KCH = 16-byte chunks of K per LDS row: 8 (128-B rows) or 16 (256-B rows: half the barriers per K)
  using E = typename Opnd<T>::E;
  constexpr int NP = Opnd<T>::NP;                // operand planes (2 for the split kinds: hi, lo)
  constexpr int NW = WM * WN;
  constexpr int ROWB = KCH * 16;                 // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;               // rows written by one wave-wide LDS-DMA instruction
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  // glds instructions (1 KB pieces) per ISSUING wave (IW of them: the LW loader waves, or every wave of the lockstep loop), k-tile
  // and plane.  A tile whose A pieces do not divide by IW (80 rows = 10 pieces on 4 loaders / 8 waves) deals them out strided --
  // piece i * IW + wave -- and the first NA % IW issuing waves carry one more: their counted waits use their own piece count (a
  // wave-uniform branch).
  constexpr int NA = BM / RPI, NWP = BN / RPI;
  constexpr int IW = LW > 0 ? LW : NW;          // waves that issue the tile loads: the loader waves, or every wave
  constexpr bool UNEVEN = NA % IW != 0;
  static_assert(NWP % IW == 0 && BM % RPI == 0 && MI >= 1 && NI >= 1, "tile does not fit the wave grid");
  constexpr int A_IPW = (NA + IW - 1) / IW, W_IPW = NWP / IW;
  constexpr int P = NP * (A_IPW + W_IPW), P_LO = NP * (A_IPW - 1 + W_IPW);
  constexpr int STAGE = NP * (BM + BN) * ROWB;   // [A planes][W planes]
  static_assert(NST * STAGE >= gemm_epi_ring_bytes<T, BM, BN>(), "epilogue staging does not fit in the ring");
  constexpr bool EARLY_READS = NP * (KCH / 4) * (MI + NI) <= 12;     // fragment registers for a whole k-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FDM_GEMM_STAMPS
  const unsigned long long t_first = wall_clock64();      // before any kernel argument is read
#endif
  float* rowstat = (float*)(smem + NST * STAGE);      // LayerNorm-folding scratch behind the ring

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef FDM_GEMM_STAMPS
  unsigned long long* stamps = (!p.incr_counter && p.incr_table) ? (unsigned long long*)p.incr_table + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (stamps && tid == 0) { stamps[0] = wall_clock64(); stamps[5] = t_first; }
#endif
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LW > 0 && wave_id >= NW;
  const int wave = LW > 0 ? (loader ? wave_id - NW : 0) : wave_id;      // index among the ISSUING waves (piece ownership below)
  const int wm = wave_id / WN, wn = wave_id % WN;                        // compute-wave coordinates (unused by loader waves)
  const int g = lane >> 4, r16 = lane & 15;
  constexpr bool KSP = (SPEC & GEMM_KSPLIT) != 0, B2 = (SPEC & GEMM_B2) != 0;
  int z = KSP ? 0 : blockIdx.z;                  // batch index (a K-sliced launch is not batched)
  int by = blockIdx.y, bx = blockIdx.x, c2 = 0;
  if constexpr (B2) {
    // second batch level: grid z = batch2 * batch.  Workgroups are dispatched in linear-id order round-robin over the 8 XCDs, so
    // id % 8 names the L2 a workgroup runs behind: deal the ids so that XCD x works on groups [x G / 8, (x + 1) G / 8) only,
    // group-major (all clips and row tiles of one group back to back): each L2 streams its share of W exactly once.
    const int nx = gridDim.x, ny = gridDim.y, G = p.batch > 0 ? p.batch : 1, C = p.batch2;
    const int id = (blockIdx.z * ny + blockIdx.y) * nx + blockIdx.x;
    int g_, r;
    if (G % 8 == 0) {
      const int per_g = ny * nx * C, slot = id >> 3;
      const int gl = slot / per_g;
      g_ = (id & 7) * (G / 8) + gl; r = slot - gl * per_g;
    } else {
      const int per_g = ny * nx * C;
      g_ = id / per_g; r = id - g_ * per_g;
    }
    z = g_; c2 = r / (ny * nx);
    const int r2 = r - c2 * ny * nx;
    by = r2 / nx; bx = r2 - by * nx;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int M = pM, N = pN;
  constexpr int EPC = 16 / (int)sizeof(E);
  int nk = pK / (KCH * EPC);
  const E* A = (const E*)pA;
  const E* W = (const E*)pW;
  if (z) {                                       // (unbatched launches -- the step's -- do not touch the argument block before their first requests)
    A += (size_t)z * p.a_batch_stride;
    W += (size_t)z * p.w_batch_stride;
  }
  if constexpr (KSP) {                           // this slice's k-tiles: [slice * nk, (slice + 1) * nk)
    nk /= p.ksplit;
    A += (size_t)blockIdx.z * nk * (KCH * EPC);
    W += (size_t)blockIdx.z * nk * (KCH * EPC);
  }
  if constexpr (B2) A += (size_t)c2 * p.a_batch_stride2;
  const KSliceArgs<KSP ? 1 : (B2 ? 2 : 0)> ksa(p, KSP ? (int)blockIdx.z : c2, (int)sizeof(E));
  const fdm_gemm_args& pe = ksa.a;               // what the epilogue reads (bias, residual, outputs)
  const size_t a_lo = (size_t)p_a_lo_off * sizeof(E), w_lo = (size_t)p_w_lo_off * sizeof(E);   // bytes hi plane -> lo plane

  // per-lane source pointers (k-tile 0, plane 0); LDS row groups are wave-uniform.  The LDS image is
  // lane-linear, so the swizzle sits on the source: lane (row, slot) fetches chunk slot ^ (row % KCH).
  const int lrow = lane / KCH, slot = lane % KCH;
  const char* a_src[A_IPW];
  const char* w_src[W_IPW];
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int piece = UNEVEN ? min(i * IW + wave, NA - 1) : wave * A_IPW + i;
    const int row = RPI * piece + lrow;
    a_src[i] = (const char*)(A + (size_t)min(m0 + row, M - 1) * p_lda) + ((slot ^ (row % KCH)) << 4);
  }
#pragma unroll
  for (int i = 0; i < W_IPW; ++i) {
    const int row = RPI * (wave * W_IPW + i) + lrow;
    w_src[i] = (const char*)(W + (size_t)min(n0 + row, N - 1) * p_ldw) + ((slot ^ (row % KCH)) << 4);
  }
  auto issue = [&](int kt) {
    char* sb = smem + (kt % NST) * STAGE;
    const size_t off = (size_t)kt * ROWB;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < A_IPW; ++i) {
        if constexpr (UNEVEN) {
          if (i * IW + wave < NA)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + off + pl * a_lo), (lptr_t)(sb + pl * BM * ROWB + (i * IW + wave) * 1024), 16, 0, 0);
        } else {
          __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + off + pl * a_lo), (lptr_t)(sb + pl * BM * ROWB + (wave * A_IPW + i) * 1024), 16, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < W_IPW; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + off + pl * w_lo), (lptr_t)(sb + (NP * BM + pl * BN) * ROWB + (wave * W_IPW + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
  f32x4 accl[NP == 2 ? MI : 1][NP == 2 ? NI : 1];     // split kinds: the two small products (hi.lo + lo.hi), scaled once at the end
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (NP == 2) accl[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // The first ring tiles are requested before anything else that touches memory (the first k-tile's latency is the longest
  // wait of the kernel); the epilogue operands follow, younger than those tiles and older than every later one: the counted
  // waits of the first k iterations are stricter by their number until they have returned, and the loop's final vmcnt(0)
  // covers them in any case.
  if (LW == 0 || loader) {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
      if (t < nk) issue(t);
  }
  // (the step counter's increment rides on one thread of the launch; after the first requests so that they do not wait for p)
  if (p.incr_counter && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    const int nv = *p.incr_counter + 1;
    *p.incr_counter = nv;
    if (p.incr_table) p.incr_counter[1] = p.incr_table[nv];    // e.g. t = tseq[step]: saves later kernels one dependent load
  }
  if constexpr (LW > 0) {
    if (loader) {
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + NST - 2 < nk) {
          if (!UNEVEN || wave < NA % IW) wait_vmcnt<(NST - 2) * P>();
          else wait_vmcnt<(NST - 2) * P_LO>();
        } else {
          wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();           // tile kt landed (every loader's share); the compute waves are done with stage (kt-1) % NST
        if constexpr (!(FDM_LW_VARIANT & 4)) { if (kt + NST - 1 < nk) issue(kt + NST - 1); }
      }
      return;                                   // (a finished wave leaves the workgroup's barrier count: the epilogue's barriers are the compute waves')
    }
  }
  EpiPre<MI, NI> epre;
  constexpr bool FOLDC = !(SPEC & GEMM_LEAN) || (SPEC & GEMM_FOLD);
  gemm_epi_preload<T, BM, BN, WM, WN, SCHED>(pe, m0, n0, z, wm, wn, g, r16, FOLDC, epre);
  if constexpr (FOLDC) gemm_load_rowstats<BM>(p, m0, rowstat);   // visible to every wave after the first barrier of the k loop

  // fragment (plane pl, k-step s) of tile row `row` in the stage at `base`: one ds_read_b128 through the XOR swizzle
  auto frag_a = [&](const char* base, int pl, int s, int mi) {
    const int row = wm * (BM / WM) + mi * 16 + r16;
    return *(const u32x4*)(base + (pl * BM + row) * ROWB + (((4 * s + g) ^ (row % KCH)) << 4));
  };
  auto frag_w = [&](const char* base, int pl, int s, int ni) {
    const int row = wn * (BN / WN) + ni * 16 + r16;
    return *(const u32x4*)(base + (NP * BM + pl * BN + row) * ROWB + (((4 * s + g) ^ (row % KCH)) << 4));
  };
  auto mma = [&](int mi, int ni, const u32x4 (&wf)[NP], const u32x4 (&af)[NP]) {
    if constexpr (NP == 1) {
      Mma<T>::run(acc[mi][ni], wf[0], af[0]);            // D[row = n][col = m]
    } else {
      mma16<E>(acc[mi][ni], wf[0], af[0]);
      mma16<E>(accl[mi][ni], wf[0], af[1]);
      mma16<E>(accl[mi][ni], wf[1], af[0]);
    }
  };

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the (NST-2) younger tiles of this wave are still in flight
    if constexpr (LW == 0) {
      if (kt + NST - 2 < nk) {
        if (!UNEVEN || wave < NA % IW) wait_vmcnt<(NST - 2) * P>();
        else wait_vmcnt<(NST - 2) * P_LO>();
      } else {
        wait_vmcnt<0>();
      }
    }
    __builtin_amdgcn_s_barrier();           // everyone's part of tile kt landed; stage (kt-1)%NST is free
#ifdef FDM_GEMM_STAMPS
    if (stamps && tid == 0 && kt == 0) stamps[1] = wall_clock64();
#endif
    const char* base = smem + (kt % NST) * STAGE;
    if constexpr (EARLY_READS) {
      // fragment reads first, THEN the LDS-DMA issue for tile kt+NST-1: a DMA piece costs the issuing wave 60-180
      // cycles (MI355X_MICROARCH.md) which now overlap the ds_read latency instead of preceding it (measured -3..-7 %).
      // (Going further -- reading tile kt+1's fragments before tile kt's MFMAs, two register sets -- measured slower.)
      u32x4 af[KCH / 4][MI][NP], wf[KCH / 4][NI][NP];
#pragma unroll
      for (int s = 0; s < KCH / 4; ++s)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[s][mi][pl] = (FDM_LW_VARIANT & 2) ? u32x4{(unsigned)kt, 1u, 2u, 3u} : frag_a(base, pl, s, mi);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[s][ni][pl] = (FDM_LW_VARIANT & 2) ? u32x4{(unsigned)kt, 5u, 6u, 7u} : frag_w(base, pl, s, ni);
        }
      if (LW == 0 && kt + NST - 1 < nk) issue(kt + NST - 1);
#pragma unroll
      for (int s = 0; s < KCH / 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (FDM_LW_VARIANT & 1) { asm volatile("" ::"v"(wf[s][ni][0]), "v"(af[s][mi][0])); }      // (keeps the fragment reads alive)
            else mma(mi, ni, wf[s][ni], af[s][mi]);
          }
    } else {
      if (LW == 0 && kt + NST - 1 < nk) issue(kt + NST - 1);
#pragma unroll
      for (int s = 0; s < KCH / 4; ++s) {
        u32x4 af[MI][NP], wf[NI][NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[mi][pl] = (FDM_LW_VARIANT & 2) ? u32x4{(unsigned)kt, 1u, 2u, 3u} : frag_a(base, pl, s, mi);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[ni][pl] = (FDM_LW_VARIANT & 2) ? u32x4{(unsigned)kt, 5u, 6u, 7u} : frag_w(base, pl, s, ni);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (FDM_LW_VARIANT & 1) { asm volatile("" ::"v"(wf[ni][0]), "v"(af[mi][0])); }
            else mma(mi, ni, wf[ni], af[mi]);
          }
      }
    }
  }
  if constexpr (NP == 2) {
    constexpr float inv = 1.f / Opnd<T>::SCALE;      // a power of two: exact
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += accl[mi][ni] * inv;
  }
#ifdef FDM_GEMM_STAMPS
  if (stamps && tid == 0) stamps[2] = wall_clock64();
#endif
  if constexpr (LW > 0) wait_vmcnt<0>();      // the epilogue operands requested before the loop (a compute wave has no other load in flight)
#ifndef FDM_GEMM_STAMPS
  gemm_epilogue<T, BM, BN, WM, WN, HEAVY, SCHED, SPEC>(pe, acc, epre, m0, n0, z, wm, wn, g, r16, rowstat, smem);
#else
  // (instrumented build: with FDM_EPI_PASSES = 2 in the environment of the probe -- carried in fdm_gemm_args.sched.advance -- the epilogue runs twice over
  //  the SAME instructions, the second pass with a warm instruction cache: stamps [6] / [7] = its stores issued / acknowledged)
  const int passes = (stamps && p.sched.advance == 2 && !SCHED) ? 2 : 1;
#pragma clang loop unroll(disable)
  for (int pass = 0; pass < passes; ++pass) {
    gemm_epilogue<T, BM, BN, WM, WN, HEAVY, SCHED, SPEC>(pe, acc, epre, m0, n0, z, wm, wn, g, r16, rowstat, smem);
    if (stamps && tid == 0) stamps[pass == 0 ? 3 : 6] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stamps && tid == 0) stamps[pass == 0 ? 4 : 7] = wall_clock64();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Large-M main loop ("ping-pong", round 3).  With thousands of rows a workgroup's k loop is no longer a latency chain but
// a throughput problem: per 64-deep k-tile a 256x128 bf16 tile needs 48 LDS-DMA pieces (each costs its issuing wave
// 60-185 cycles of issue, MI355X_MICROARCH.md) and 128 fragment reads beside 256 MFMAs.  In gemm_glds_kernel all eight waves
// do these three things in lockstep behind one barrier per k-tile, so the matrix pipes idle while every wave issues DMA and
// waits for its fragments (measured at M = 6400: 17-26 % of the MFMA peak in every tile).
// Here the eight waves form two groups of four (waves w and w + 4 share a SIMD) that run the SAME program half a period
// apart: while group 0 computes tile kt (32 MFMAs per wave, nothing else), group 1 reads its fragments of tile kt from LDS
// and issues its share of the DMA for tile kt + D; then they swap.  A SIMD's matrix pipe always has one wave feeding it
// and the other wave's LDS / DMA issue beside it (matrix beside memory: the complementary pairing of
// MI355X_MICROARCH.md "Two waves per SIMD", item 5).  Slots are separated by raw s_barriers; group 1 enters the loop one
// barrier late and group 0 pays one extra barrier at the end.
//   slot 2kt   : G0 LOAD(kt)     | G1 COMPUTE(kt-1)
//   slot 2kt+1 : G0 COMPUTE(kt)  | G1 LOAD(kt)
//   LOAD(kt)   = ds_read the wave's fragments of tile kt (one register set: the previous COMPUTE has consumed it),
//                issue its pieces of tile kt + D into stage (kt + D) % NST, D = NST - 1
//   landing    : a wave's counted vmcnt for its pieces of tile kt + 1 sits before the barrier that ends the slot in which
//                BOTH groups still have to read tile kt + 1 (G0: end of COMPUTE(kt), G1: end of LOAD(kt))
//   reuse      : stage (kt + D) % NST last held tile kt + D - NST = kt - 1, read in slots 2kt-2 (G0) and 2kt-1 (G1): both
//                lie before slot 2kt, the first slot that issues into it
// Same operands, same LDS image (lane-linear DMA, source-side XOR swizzle), same k order per accumulator and the same
// epilogue as gemm_glds_kernel: results are bit-identical to every other tile.
// ---------------------------------------------------------------------------------------------------
#ifndef FDM_PP_PRIO
#define FDM_PP_PRIO 1         // experiments: 0 = no s_setprio, 1 = raised around the MFMA cluster, 2 = raised around the LOAD phase
#endif
#ifndef FDM_PP_VARIANT
#define FDM_PP_VARIANT 0      // tools/pp_probe.cpp only: bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no DMA after the prologue
#endif
template <typename T, int BM, int BN, int WM, int WN, int NST, bool HEAVY = false, bool SCHED = false, int SPEC = 0, int KCH = 8>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const fdm_gemm_args p) {
  using E = typename Opnd<T>::E;
  static_assert(WM * WN == 8, "ping-pong loop: two groups of four waves");
  constexpr int NP = Opnd<T>::NP;
  // KCH = 16-byte chunks of K per LDS row: 8 (64-deep stages, two k-steps) or 4 (32-deep stages, one k-step: half the bytes
  // per stage, so a 256x256 tile gets a 4-stage ring in 128 KB)
  static_assert(KCH == 8 || KCH == 4, "LDS rows of 128 or 64 bytes");
  constexpr int NW = 8, ROWB = KCH * 16, RPI = 1024 / ROWB, KS = KCH / 4;
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr int A_IPW = BM / RPI / NW, W_IPW = BN / RPI / NW;
  static_assert(A_IPW >= 1 && W_IPW >= 1 && MI >= 1 && NI >= 1, "tile too small for the wave grid");
  constexpr int P = NP * (A_IPW + W_IPW);          // DMA pieces per wave and k-tile
  constexpr int D = NST - 1;                       // k-tiles in flight
  constexpr int STAGE = NP * (BM + BN) * ROWB;
  // (the ring doubles as the epilogue's staging area only in the kernels that can pack V or fold a LayerNorm)
  constexpr bool EPI_RING = !(SPEC & GEMM_LEAN) || (SPEC & (GEMM_KV | GEMM_FOLD));
  static_assert(NST >= 2 && (!EPI_RING || NST * STAGE >= gemm_epi_ring_bytes<T, BM, BN>()), "epilogue staging does not fit in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rowstat = (float*)(smem + NST * STAGE);
#ifdef FDM_PP_PHASES
  const unsigned long long t_entry = wall_clock64();
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  if (p.incr_counter && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    const int nv = *p.incr_counter + 1;
    *p.incr_counter = nv;
    if (p.incr_table) p.incr_counter[1] = p.incr_table[nv];
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD (cyclic placement)
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, r16 = lane & 15;
  const int z = blockIdx.z;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so consecutive ids land on different L2s.
  // Give each XCD a contiguous run of tiles, column-tile fastest: the tiles of one row block (same A rows) share an L2.
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
  int bid = blockIdx.y * gx + blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective for any nwg
  }
  const int m0 = (bid / gx) * BM, n0 = (bid % gx) * BN;
  const int M = p.M, N = p.N;
  const E* A = (const E*)p.A + (size_t)z * p.a_batch_stride;
  const E* W = (const E*)p.W + (size_t)z * p.w_batch_stride;
  const size_t a_lo = (size_t)p.a_lo_off * sizeof(E), w_lo = (size_t)p.w_lo_off * sizeof(E);

  // 16-byte-chunk swizzle of a row: 128-B rows chunk ^ (row % 8); 64-B rows (four rows per 256-B bank line) chunk ^ f(row / 4 % 4)
  // with f = {0, 2, 3, 1}: the 16 lanes of every ds_read_b128 group then hit 16 different 16-byte slots
  auto swz = [](int row) { return KCH == 8 ? (row & 7) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3); };
  const int lrow = lane / KCH, slot = lane % KCH;
  const char* a_src[A_IPW];
  const char* w_src[W_IPW];
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int row = RPI * (wave * A_IPW + i) + lrow;
    a_src[i] = (const char*)(A + (size_t)min(m0 + row, M - 1) * p.lda) + ((slot ^ swz(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < W_IPW; ++i) {
    const int row = RPI * (wave * W_IPW + i) + lrow;
    w_src[i] = (const char*)(W + (size_t)min(n0 + row, N - 1) * p.ldw) + ((slot ^ swz(row)) << 4);
  }
  auto issue = [&](int kt) {
    char* sb = smem + (kt % NST) * STAGE;
    const size_t off = (size_t)kt * ROWB;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < A_IPW; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + off + pl * a_lo), (lptr_t)(sb + pl * BM * ROWB + (wave * A_IPW + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < W_IPW; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + off + pl * w_lo), (lptr_t)(sb + (NP * BM + pl * BN) * ROWB + (wave * W_IPW + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
  f32x4 accl[NP == 2 ? MI : 1][NP == 2 ? NI : 1];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (NP == 2) accl[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  constexpr int EPC = 16 / (int)sizeof(E);
  const int nk = p.K / (KCH * EPC);
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) issue(t);
  constexpr bool FOLDC = !(SPEC & GEMM_LEAN) || (SPEC & GEMM_FOLD);
  if constexpr (FOLDC) gemm_load_rowstats<BM>(p, m0, rowstat);

  // per-lane LDS byte offsets of the fragment reads (k-step 0); k-step 1 flips chunk bit 2: (4 + g) ^ r = (g ^ r) ^ 4
  int a_off[MI], w_off[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * (BM / WM) + mi * 16 + r16;
    a_off[mi] = row * ROWB + ((g ^ swz(row)) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * (BN / WN) + ni * 16 + r16;
    w_off[ni] = (NP * BM + row) * ROWB + ((g ^ swz(row)) << 4);
  }
  u32x4 af[KS][MI][NP], wf[KS][NI][NP];
  auto load_frags = [&](int kt) {
    const char* base = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[s][mi][pl] = *(const u32x4*)(base + pl * BM * ROWB + (a_off[mi] ^ (s << 6)));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[s][ni][pl] = *(const u32x4*)(base + pl * BN * ROWB + (w_off[ni] ^ (s << 6)));
      }
  };
  auto compute = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (NP == 1) {
            Mma<T>::run(acc[mi][ni], wf[s][ni][0], af[s][mi][0]);
          } else {
            mma16<E>(acc[mi][ni], wf[s][ni][0], af[s][mi][0]);
            mma16<E>(accl[mi][ni], wf[s][ni][0], af[s][mi][1]);
            mma16<E>(accl[mi][ni], wf[s][ni][1], af[s][mi][0]);
          }
        }
  };

  // tile 0 has landed once at most the D - 1 younger tiles of this wave are still in flight
  if (D - 1 < nk) wait_vmcnt<(D - 1) * P>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();            // group 1 runs one slot behind group 0
#ifdef FDM_PP_PHASES
  // tools/pp_probe.cpp: 100 MHz wall-clock stamps of every workgroup -> p.rln_gamma as [workgroup][4] u64: entry, loop start, loop end, stores done
  unsigned long long* phases = (tid == 0) ? (unsigned long long*)p.rln_gamma + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 : nullptr;
  if (phases) { phases[0] = t_entry; phases[1] = wall_clock64(); }
#endif
#ifdef FDM_PP_STAMPS
  // tools/pp_probe.cpp: shader-clock stamps of one workgroup's waves around every slot -> p.rln_beta as [wave][kt < 16][4] u64
  unsigned long long* stamps = (blockIdx.x == 1 && blockIdx.y == 3 && lane == 0) ? (unsigned long long*)p.rln_beta + (size_t)wave * 64 : nullptr;
#define PP_STAMP(i) do { if (stamps && kt < 16) stamps[kt * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    // ---- LOAD(kt)
    PP_STAMP(0);
    if constexpr (FDM_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if constexpr (!(FDM_PP_VARIANT & 2)) load_frags(kt);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(FDM_PP_VARIANT & 4)) { if (kt + D < nk) issue(kt + D); }
    if (grp) {                                      // own pieces of tile kt + 1 landed (group 0 waits after its COMPUTE)
      if (kt + D < nk) wait_vmcnt<(D - 1) * P>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FDM_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    PP_STAMP(1);
    // this wave's fragment reads of tile kt have RETURNED before the barrier: the other group's next LOAD slot issues LDS-DMA
    // into the stage a wave of this group may still be reading otherwise (the stage-reuse argument above counts a read as done
    // once it is issued; the first MFMA of COMPUTE waits for the same counter anyway)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- COMPUTE(kt)
    PP_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FDM_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (!(FDM_PP_VARIANT & 1)) {
      compute();
    } else if constexpr (!(FDM_PP_VARIANT & 2)) {      // keep the fragment reads alive
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(af[s][mi][0]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[s][ni][0]));
      }
    }
    if constexpr (FDM_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (!grp) {
      if (kt + D < nk) wait_vmcnt<(D - 1) * P>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(3);
    __builtin_amdgcn_s_barrier();
  }
  if (!grp) __builtin_amdgcn_s_barrier();           // matches group 1's late start
#ifdef FDM_PP_PHASES
  if (phases) phases[2] = wall_clock64();
#endif
  if constexpr (NP == 2) {
    constexpr float inv = 1.f / Opnd<T>::SCALE;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += accl[mi][ni] * inv;
  }
  // the epilogue operands are fetched after the k loop here (one exposed round trip per workgroup, against 16+ k-tiles of
  // work): held across the loop they would cost MI * NI * 4 registers beside the accumulators and the fragment set
  if constexpr (MI > 4 && !EPI_RING) {
    // 128-row wave tiles: the epilogue's operand preload (MI x NI residual fragments) would not fit beside the accumulators, so it
    // runs over the wave's rows in two halves -- the lean epilogue only needs each row's global index, which the per-wave row
    // base carries (a tile of BM / 2 rows with this wave's rows shifted into place)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x4 (&acch)[MI / 2][NI] = *reinterpret_cast<f32x4 (*)[MI / 2][NI]>(&acc[hh * (MI / 2)]);
      const int m0h = m0 + wm * (BM / WM) - wm * (BM / 2 / WM) + hh * (BM / 2 / WM);
      EpiPre<MI / 2, NI> epre;
      gemm_epi_preload<T, BM / 2, BN, WM, WN, SCHED>(p, m0h, n0, z, wm, wn, g, r16, false, epre);
      gemm_epilogue<T, BM / 2, BN, WM, WN, HEAVY, SCHED, SPEC>(p, acch, epre, m0h, n0, z, wm, wn, g, r16, nullptr, nullptr);
    }
  } else {
    EpiPre<MI, NI> epre;
    gemm_epi_preload<T, BM, BN, WM, WN, SCHED>(p, m0, n0, z, wm, wn, g, r16, FOLDC, epre);
    gemm_epilogue<T, BM, BN, WM, WN, HEAVY, SCHED, SPEC>(p, acc, epre, m0, n0, z, wm, wn, g, r16, rowstat, smem);
  }
#ifdef FDM_PP_PHASES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (phases) phases[3] = wall_clock64();
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int NST, bool HEAVY, bool SCHED = false, int SPEC = 0, int KCH = 8>
static hipError_t gemm_pp_launch_h(const fdm_gemm_args& a, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.batch > 0 ? a.batch : 1);
  constexpr int lds = NST * Opnd<T>::NP * (BM + BN) * KCH * 16 + gemm_ln_scratch_bytes<BM, BN>();
  static_assert(lds <= 160 * 1024, "ring does not fit the CU's LDS");
  static bool once = [] {
    return hipFuncSetAttribute((const void*)gemm_pp_kernel<T, BM, BN, WM, WN, NST, HEAVY, SCHED, SPEC, KCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
  }();
  (void)once;
  hipLaunchKernelGGL((gemm_pp_kernel<T, BM, BN, WM, WN, NST, HEAVY, SCHED, SPEC, KCH>), grid, dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN, int NST, int KCH, bool HEAVY, bool SCHED = false, int SPEC = 0, int LW = 0>
static hipError_t gemm_glds_launch_h(const fdm_gemm_args& a, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, (SPEC & GEMM_KSPLIT) ? a.ksplit : (a.batch > 0 ? a.batch : 1) * ((SPEC & GEMM_B2) ? a.batch2 : 1));
  constexpr int lds = NST * Opnd<T>::NP * (BM + BN) * KCH * 16 + gemm_ln_scratch_bytes<BM, BN>();
  static_assert(lds <= 160 * 1024, "ring does not fit the CU's LDS");
  static bool once = [] {
    return hipFuncSetAttribute((const void*)gemm_glds_kernel<T, BM, BN, WM, WN, NST, KCH, HEAVY, SCHED, SPEC, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
  }();
  (void)once;
  hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, WM, WN, NST, KCH, HEAVY, SCHED, SPEC, LW>), grid, dim3(64 * (WM * WN + LW)), lds, s,
                     a.A, a.W, (long long)a.a_lo_off, (long long)a.w_lo_off, a.M, a.N, a.K, (int)a.lda, (int)a.ldw, a);
  return hipGetLastError();
}
// FDM_GEMM_LOCKSTEP=1 (env, read once): FDM_TILE_LOCKSTEP for every launch of the process -- A/B of the once-per-clip stages, whose
// GEMMs carry no plan switch
static bool gemm_lockstep_env() {
  static bool v = [] { const char* e = getenv("FDM_GEMM_LOCKSTEP"); return e && atoi(e) != 0; }();
  return v;
}
// Host mirror of the epilogue's `lean` predicate for EVERY tile of the launch: interior column tiles, vectorisable outputs /
// residual, and (QKV projections) tile-aligned Q | K | V column ranges with whole packed chunks.
template <typename T, int BM, int BN>
static bool gemm_all_tiles_lean(const fdm_gemm_args& a) {
  using E = typename Opnd<T>::E;
  auto al = [](const void* p, size_t n) { return ((uintptr_t)p % n) == 0; };
  if (a.N % BN || a.out_batch_stride % 4 || a.resid_row_mod) return false;
  if (a.out_f32 && (a.ldo_f32 % 4 || !al(a.out_f32, 16))) return false;
  if (a.out_t && (a.ldo_t % 4 || !al(a.out_t, 4 * sizeof(E)) || (Opnd<T>::NP == 2 && a.out_t_lo_off % 4))) return false;
  if (a.resid && (a.ldr % 4 || !al(a.resid, 16))) return false;
  if (a.out_kp || a.out_vp) {
    const int kcol_lo = a.out_kp ? a.kp_col0 : a.N, kcol_hi = a.out_kp ? (a.out_vp ? a.vp_col0 : a.N) : a.N;
    if (a.kv_hd % 16) return false;
    if (a.out_kp && (kcol_lo % BN || kcol_hi % BN)) return false;
    if (a.out_vp && a.vp_col0 % BN) return false;
    if (!a.out_kp && a.out_vp) return false;          // (V-only projections: the general kernel)
  }
  return true;
}
template <typename T, int BM, int BN, int WM, int WN, int NST, int KCH, int LW>
static hipError_t gemm_glds_launch_lw(const fdm_gemm_args& a, hipStream_t s) {
  const bool no_lean = (a.tile & FDM_TILE_GENERAL) != 0;      // tests: the edge-handling kernel on a shape that does not need it
  if (a.ksplit > 1) {       // K-sliced launch (validated by fdm_op_gemm: plain fp32 output, no activation): the 64-column tiles only
    if constexpr (BN == 64 && (BM == 64 || BM == 32)) {
      if (!gemm_all_tiles_lean<T, BM, BN>(a)) return hipErrorInvalidValue;
      return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, GEMM_LEAN | GEMM_KSPLIT, LW>(a, s);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.batch2 >= 1) {       // second batch level (validated by fdm_op_gemm): two kernels per tile, both with every activation compiled in
    if constexpr (BN == 64 && (BM == 64 || BM == 128)) {
      if (!no_lean && gemm_all_tiles_lean<T, BM, BN>(a) && a.out_batch_stride2 % 4 == 0)
        return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, true, false, GEMM_LEAN | GEMM_B2, LW>(a, s);
      return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, true, false, GEMM_B2, LW>(a, s);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (gemm_act_is_heavy(a.act)) {
    if (!no_lean && !a.out_kp && !a.out_vp && !a.stat_out && !a.ln_stat_in && gemm_all_tiles_lean<T, BM, BN>(a))
      return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, true, false, GEMM_LEAN, LW>(a, s);
    return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, true, false, 0, LW>(a, s);
  }
  if (!no_lean && gemm_all_tiles_lean<T, BM, BN>(a)) {
    const bool kv = a.out_kp || a.out_vp, fold = a.stat_out || a.ln_stat_in;
    if (kv && fold) return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, GEMM_LEAN | GEMM_KV | GEMM_FOLD, LW>(a, s);
    if (kv) return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, GEMM_LEAN | GEMM_KV, LW>(a, s);
    if (fold) return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, GEMM_LEAN | GEMM_FOLD, LW>(a, s);
    return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, GEMM_LEAN, LW>(a, s);
  }
  return gemm_glds_launch_h<T, BM, BN, WM, WN, NST, KCH, false, false, 0, LW>(a, s);
}

// Tile launcher: the loader-wave form (LW_ waves that only issue the tile loads) for the 16-bit operand kinds unless the caller asks
// for the lockstep loop (FDM_TILE_LOCKSTEP: A/B and tests -- the two are bit-identical); fp32 is MFMA-bound and keeps the lockstep loop.
template <typename T, int BM, int BN, int WM, int WN, int NST, int LW_ = 0, int KCH = 8>
static hipError_t gemm_glds_launch_t(const fdm_gemm_args& a, hipStream_t s) {
  if constexpr (LW_ > 0 && !std::is_same<T, float>::value) {
    if (!(a.tile & FDM_TILE_LOCKSTEP) && !gemm_lockstep_env()) return gemm_glds_launch_lw<T, BM, BN, WM, WN, NST, KCH, LW_>(a, s);
  }
  return gemm_glds_launch_lw<T, BM, BN, WM, WN, NST, KCH, 0>(a, s);
}

template <typename T, int BM, int BN, int WM, int WN, int NST>
static hipError_t gemm_pp_launch_t(const fdm_gemm_args& a, hipStream_t s) {
  const bool no_lean = (a.tile & FDM_TILE_GENERAL) != 0;      // tests: the edge-handling kernel on a shape that does not need it
  if (gemm_act_is_heavy(a.act)) {
    if (!no_lean && !a.out_kp && !a.out_vp && !a.stat_out && !a.ln_stat_in && gemm_all_tiles_lean<T, BM, BN>(a))
      return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, true, false, GEMM_LEAN>(a, s);
    return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, true>(a, s);
  }
  if (!no_lean && gemm_all_tiles_lean<T, BM, BN>(a)) {
    const bool kv = a.out_kp || a.out_vp, fold = a.stat_out || a.ln_stat_in;
    if (kv && fold) return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, false, false, GEMM_LEAN | GEMM_KV | GEMM_FOLD>(a, s);
    if (kv) return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, false, false, GEMM_LEAN | GEMM_KV>(a, s);
    if (fold) return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, false, false, GEMM_LEAN | GEMM_FOLD>(a, s);
    return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, false, false, GEMM_LEAN>(a, s);
  }
  return gemm_pp_launch_h<T, BM, BN, WM, WN, NST, false>(a, s);
}

// ---- tile choice -------------------------------------------------------------------------------------------------------
// fdm_gemm_args.tile (the caller's plan-time choice; FDM_TILE_GENERAL may be or-ed in), else the FDM_GEMM_TILE override (env, read
// once: the FDM_TILE_* value forced for every GEMM, for A/B measurements), else the heuristic below: gemm_heuristic_tile names the FDM_TILE_* a launch
// with tile = 0 resolves to (also exported as fdm_gemm_heuristic_tile, so that the plan-time tuner does not time a candidate
// against itself).  Every tile accumulates k in the same order: the choice changes speed, never results.
static int gemm_tile_override() {
  static int v = [] { const char* e = getenv("FDM_GEMM_TILE"); return e ? atoi(e) : 0; }();
  return v;
}
static bool gemm_one_round(long long tiles) { return tiles > 192 && tiles <= 256; }
// 80x128 tiles that fill the chip in exactly one round (225..256 workgroups, e.g. 800 rows x 3072 columns = 240): every CU
// streams one (80 + 128)-row operand pair instead of two or three 64x64 ones (12.7 vs 14.6 us bf16, 22.2 vs 28.4 us f16x3
// on that shape; profiles/README.md round 3)
static bool gemm_one_round_80(const fdm_gemm_args& a) {
  const long long t80 = (long long)((a.M + 79) / 80) * ((a.N + 127) / 128) * (a.batch > 0 ? a.batch : 1);
  return t80 > 224 && t80 <= 256 && (a.M % 80 == 0 || a.M % 80 > 40);   // (a mostly empty last row tile wastes the round)
}

// 64x128 tiles that fill the chip in exactly one round for a wide projection (FFN1 at the step's 800 rows: 13 x 16 = 208 workgroups): with
// loader waves (round 5) the k loop runs at the CU's L2 -> LDS rate, so the tile with fewer bytes per CU wins -- 8.45 us against 9.71 for two
// co-resident 64x64 tiles per CU in bf16, 14.9 against 18.7 in f16x3 (profiles/r5_ldw/isolated.txt)
static bool gemm_one_round_64x128(const fdm_gemm_args& a) {
  const long long t = (long long)((a.M + 63) / 64) * ((a.N + 127) / 128) * (a.batch > 0 ? a.batch : 1);
  return gemm_one_round(t) && a.N >= 2048 && a.N % 128 == 0;
}

// elem_bytes: 4 (fp32) or 2; split: the two-plane kinds (a ring stage is twice as large there, so their tile set is the part
// of the one-plane set whose ring fits 160 KB of LDS)
static int gemm_heuristic_tile(const fdm_gemm_args& a, int elem_bytes, bool split) {
  if (a.batch2 >= 1) {       // grouped conv over clips: 128-row tiles once they fill the chip (one round of 256: 4 row tiles x 16 groups x 4 clips), else 64
    const long long t128b = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64) * (a.batch > 0 ? a.batch : 1) * a.batch2;
    return t128b >= 192 ? FDM_TILE_128x64 : FDM_TILE_64x64;
  }
  const long long batch = a.batch > 0 ? a.batch : 1;
  const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * batch;
  const long long t128x64 = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64) * batch;
  const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64) * batch;
  const long long t64x128 = (long long)((a.M + 63) / 64) * ((a.N + 127) / 128) * batch;
  // Round 5 (the loop runs at the CU's L2 -> LDS rate: fewer bytes on the busiest CU wins; rules from the tuner's picks over 66 shapes on
  // the loader-wave build, profiles/r5_tile_sweep/):
  //  * narrow outputs (N <= 512: MEAD's out-proj / FFN2 / decoder at 2400-4000 rows) whose 64x64 grid needs two workgroups per CU or two
  //    rounds while the 64x128 grid is one round: 64x128 (-4...-8 % on those chains)
  const bool narrow_one_round = elem_bytes == 2 && a.N <= 512 && a.N % 128 == 0 && t64 > 256 && t64x128 <= 256;
  //  * a few hundred rows and a wide projection whose 128x64 grid is (nearly) one round: 128x64 instead of two co-resident 64x64 tiles
  //    per CU (QKV at 400-600 rows, FFN1 at 600: -2.5...-4 %)
  const bool wide_128x64 = elem_bytes == 2 && a.M <= 1024 && a.N >= 2048 && t128x64 >= 160 && t128x64 <= 256;
  if (!split) {
    // Measured on MI355X (profiles/README.md): the biggest tile wins only once it still yields >= 2 blocks per CU;
    // below that the 64x64 tile's extra blocks beat its higher L2->LDS traffic.
    constexpr long long thr128 = 512, thr128x64 = 700;
    const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128) * batch;
    // thousands of rows: the ping-pong loop on the 256x128 tile (the tuner's pick for the N <= 2048 sites from 6400 rows and
    // for FFN1 from 3200; `profiles/r3_rows_sweep`, `r3_tile_sweep`; bf16 only, whole column tiles)
    const bool pp_ok = elem_bytes == 2 && a.N % 128 == 0 && a.K >= 1024;
    if (pp_ok && a.M >= 6000 && a.N <= 2048 && t256 >= 150) return FDM_TILE_256x128_PP;
    // a wide projection whose 256x128 grid is exactly one round (HuBERT's FFN1 at 4 x 10 s: 1992 x 4096 = 8 x 32 tiles; 29.6 us
    // against 33.3 on the 512 tiles of 128x128, profiles/r4_pmc_hubert; the lockstep 256x128 kernel of rounds 2-4 measured 29.9 against
    // 29.4 for the ping-pong loop there and is retired: profiles/r5_xcd_band/gemm_tiles_hubert_ffn.txt)
    if (elem_bytes == 2 && a.M > 1024 && t128 >= thr128 && gemm_one_round(t256)) return FDM_TILE_256x128_PP;
    if (t128 >= thr128) {
      // 128x128 unless its last round is less than half full (600 tiles = 2.34 rounds: QKV at 3200 rows, MEAD's at 6400): 128x64 on the
      // 3-stage ring, two workgroups per CU, balances that tail (-4 % on those chains)
      const long long tail = t128 % 256;
      if (elem_bytes == 2 && t128 < 1024 && tail > 0 && tail < 128) return FDM_TILE_128x64;
      return FDM_TILE_128x128;
    }
    if (gemm_one_round_80(a)) return FDM_TILE_80x128;
    if (elem_bytes == 2 && gemm_one_round_64x128(a)) return FDM_TILE_64x128;
    if (wide_128x64) return FDM_TILE_128x64;
    if (narrow_one_round && a.M > 1024) return FDM_TILE_64x128;
    // one short clip (and MEAD's d = 512 sites up to ~800 rows): 64x64 tiles leave most CUs idle -> 32-row tiles, twice the workgroups
    // (the split kinds have had this rule since round 3; with loader waves it pays in bf16 too: -3.5...-5.5 % on MEAD's chains at 200-800 rows)
    if (elem_bytes == 2 && t64 <= 128 && a.N <= 1536) return FDM_TILE_32x64_S3;
    // (measured in bf16 only: the fp32 kind keeps its rules; short-K products whose 64x64 grid is resident in one round -- two
    //  64 KB rings per CU -- stay on it: MEAD's d = 512 sites at 1200-1600 rows lost 3-5 % on larger tiles)
    if (elem_bytes == 2 && a.M > 1024 && (t64 > 512 || a.K >= 1024)) {
      // 1100..4000 rows (batched clips, long clips, CFG): what the plan-time tuner picks there (profiles/r3_tile_sweep/), as rules.
      // A grid that fills the chip in exactly ONE round wins; else 128x64 -- on the 4-stage ring while its grid is one round, on the
      // 3-stage ring (72 KB: two workgroups per CU, all of <= 512 tiles resident) beyond.
      if (gemm_one_round(t128)) return FDM_TILE_128x128;
      if (gemm_one_round(t256)) return FDM_TILE_256x128_PP;
      if (t128x64 > 128) return FDM_TILE_128x64;          // (4-stage ring while its grid is one round, 3-stage beyond: the tile picks)
    }
    if (t128x64 >= thr128x64) return FDM_TILE_128x64;
    return FDM_TILE_64x64;
  }
  if (t128 >= 512) return FDM_TILE_128x128;
  if (gemm_one_round_80(a)) return FDM_TILE_80x128;
  if (gemm_one_round_64x128(a)) return FDM_TILE_64x128;
  if (wide_128x64) return FDM_TILE_128x64;
  if (a.M > 1024) {            // the tuner's picks at 1100..4000 rows as rules (see above)
    if (gemm_one_round(t128)) return FDM_TILE_128x128;
    if (narrow_one_round) return FDM_TILE_64x128;
    if (t128x64 > 128 && t128x64 <= 256) return FDM_TILE_128x64;     // (split kinds: 3-stage, 144 KB ring -- one per CU, so one round only)
    // Beyond the rules (1600+ rows in the split kinds, where every ring is one workgroup per CU): the tile with the fewest operand rows on
    // the busiest CU, rounds x (BM + BN) -- the model the tuner prunes with; it reproduces the tuner's picks at 1600-4800 rows (80x128 for
    // QKV at 1600 and FFN1 at 2400, 128x128 for QKV at 2400: -3.6...-8.6 % on those chains)
    struct { int id, bm, bn; } cand[] = {{FDM_TILE_64x64, 64, 64}, {FDM_TILE_128x64, 128, 64}, {FDM_TILE_64x128, 64, 128}, {FDM_TILE_80x128, 80, 128}, {FDM_TILE_128x128, 128, 128}};
    int best = FDM_TILE_64x64; long long best_rows = -1;
    for (auto& c : cand) {
      if (a.N % c.bn) continue;
      const long long tiles = (long long)((a.M + c.bm - 1) / c.bm) * (a.N / c.bn) * batch;
      const long long rows = ((tiles + 255) / 256) * (c.bm + c.bn);
      if (best_rows < 0 || rows < best_rows) { best = c.id; best_rows = rows; }
    }
    return best;
  } else {
    // a single short clip: 64x64 tiles leave half the CUs idle -> 32-row tiles (72 KB rings, two per CU); 257..512 tiles of a
    // wide projection: the 2-stage ring (64 KB) keeps all of them resident in one round instead of two
    if (t64 <= 128) return FDM_TILE_32x64_S3;
    // more than one round of 64x64 tiles (one 128 KB ring per CU) where the 64x128 grid is a single round: MEAD's QKV at 800 rows (-6...-8 %)
    if (t64 > 256 && t64x128 <= 256 && a.N % 128 == 0) return FDM_TILE_64x128;
    if (t64 > 256 && t64 <= 512 && a.N >= 2048) return FDM_TILE_64x64_S2;
  }
  if (t128x64 >= 700) return FDM_TILE_128x64;
  return FDM_TILE_64x64;
}

// the scheduler-fused latent decoder has two forms (64x64, ping-pong 256x128): the same rule as its unfused shape
static bool gemm_sched_fuse_heuristic_pp(const fdm_gemm_args& a, int elem_bytes) {
  fdm_gemm_args b = a;
  b.sched_fuse = 0;
  return gemm_heuristic_tile(b, elem_bytes, false) == FDM_TILE_256x128_PP;
}

// the scheduler-fused latent decoder on the 64x64 tile (lean kernels only), loader-wave form for the 16-bit kinds
template <typename T>
static hipError_t gemm_sched_fuse_launch(const fdm_gemm_args& a, hipStream_t s) {
  if constexpr (!std::is_same<T, float>::value) {
    if (!(a.tile & FDM_TILE_LOCKSTEP) && !gemm_lockstep_env())
      return a.ln_stat_in ? gemm_glds_launch_h<T, 64, 64, 2, 4, 4, 8, false, true, GEMM_LEAN | GEMM_FOLD, 4>(a, s)
                          : gemm_glds_launch_h<T, 64, 64, 2, 4, 4, 8, false, true, GEMM_LEAN, 4>(a, s);
  }
  return a.ln_stat_in ? gemm_glds_launch_h<T, 64, 64, 2, 4, 4, 8, false, true, GEMM_LEAN | GEMM_FOLD>(a, s)
                      : gemm_glds_launch_h<T, 64, 64, 2, 4, 4, 8, false, true, GEMM_LEAN>(a, s);
}

// K-sliced launches (fdm_gemm_args.ksplit): the 64-column tiles, ring depth by tile id.  With S slices per output tile the grid
// is S times as large and a slice's chain 1 / S as long, so shallower rings (more co-resident slices per CU) are the candidates.
static int gemm_ksplit_heuristic_tile(const fdm_gemm_args& a) { return a.M <= 256 ? FDM_TILE_32x64_S3 : FDM_TILE_64x64; }   // (profiles/r5_splitk/; the tuner's pick at 249 rows on the loader-wave build)
template <typename T>
static hipError_t gemm_dispatch_ksplit(const fdm_gemm_args& a, int tile_id, hipStream_t s) {
  switch (tile_id > 0 ? tile_id : gemm_ksplit_heuristic_tile(a)) {
    case FDM_TILE_64x64_S2: return gemm_glds_launch_t<T, 64, 64, 2, 4, 2, 4>(a, s);
    case FDM_TILE_32x64_S3: return gemm_glds_launch_t<T, 32, 64, 2, 2, 3, 2>(a, s);
    default: return gemm_glds_launch_t<T, 64, 64, 2, 4, 4, 4>(a, s);
  }
}

template <typename T>
static hipError_t gemm_dispatch(const fdm_gemm_args& a, hipStream_t s) {
  if (a.sched_fuse) {     // (validated: interior tiles only -> the lean epilogue)
    // thousands of rows: the scheduler-fused latent decoder on the ping-pong tile when the plan's tuner picked it
    const int tile_id = a.tile & FDM_TILE_ID_MASK;
    const bool pp = tile_id == FDM_TILE_256x128_PP || (tile_id == 0 && gemm_tile_override() == 0 && gemm_sched_fuse_heuristic_pp(a, (int)sizeof(typename Opnd<T>::E)));
    if (pp && !a.ln_stat_in && a.N % 128 == 0)
      return gemm_pp_launch_h<T, 256, 128, 4, 2, 3, false, true, GEMM_LEAN>(a, s);
    return gemm_sched_fuse_launch<T>(a, s);
  }
  const int tile_id = a.tile & FDM_TILE_ID_MASK;
  if (a.ksplit > 1) return gemm_dispatch_ksplit<T>(a, tile_id, s);
  // (FDM_GEMM_TILE forces one tile on the plain launches only: a second-batch-level launch runs on its own tile set -- like ksplit above)
  const int want = tile_id > 0 ? tile_id : (a.batch2 >= 1 ? 0 : gemm_tile_override());
  switch (want > 0 ? want : gemm_heuristic_tile(a, (int)sizeof(typename Opnd<T>::E), false)) {
    case FDM_TILE_128x64_S3:                                                       // (retired id: the tile picks its ring depth)
    case FDM_TILE_128x64: {                                                        // 8 waves, 32x32 per wave
      // 4-stage ring (96 KB, one per CU) while the grid is one round; beyond that the 3-stage ring (72 KB): two workgroups per CU
      const long long wgs = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64) * (a.batch > 0 ? a.batch : 1) * (a.batch2 > 0 ? a.batch2 : 1);
      // (Round 6, measured and not adopted: four compute waves of 64x32 instead of eight of 32x32 beyond one round -- fewer fragment bytes read
      //  from LDS per staged byte -- is 7-11 % faster on the isolated QKV / FFN1 shapes at 1992 rows with a plain epilogue and 1-1.7 % SLOWER
      //  inside cfg5's step program and HuBERT-large, where the epilogue's residual / packed-K,V operands share the registers:
      //  profiles/r6_loop_ablation/.)
      return wgs <= 256 ? gemm_glds_launch_t<T, 128, 64, 4, 2, 4, 4>(a, s) : gemm_glds_launch_t<T, 128, 64, 4, 2, 3, 4>(a, s);
    }
    case FDM_TILE_96x128:                                                          // (retired id: nearest member)
    case FDM_TILE_128x128: return gemm_glds_launch_t<T, 128, 128, 2, 4, 3, 4>(a, s);  // 8 waves, 64x32 per wave
    case FDM_TILE_64x64_S2: return gemm_glds_launch_t<T, 64, 64, 2, 4, 2, 4>(a, s);    // 32 KB -> 4 workgroups per CU
    case FDM_TILE_32x64_S3: return gemm_glds_launch_t<T, 32, 64, 2, 2, 3, 2>(a, s);    // 4 waves, 16x32 per wave, 36 KB
    case FDM_TILE_256x128:                                                           // (retired id: the lockstep loop on this tile)
    case FDM_TILE_256x128_PP: return gemm_pp_launch_t<T, 256, 128, 4, 2, 3>(a, s);   // ping-pong loop, 64x64 per wave, 146 KB
    case FDM_TILE_80x128: return gemm_glds_launch_t<T, 80, 128, 1, 8, 4, 4>(a, s);      // 8 waves, 80x16 per wave: 800 rows x 3072 = 240 workgroups
    case FDM_TILE_64x128: return gemm_glds_launch_t<T, 64, 128, 2, 4, 4, 4>(a, s);      // 8 waves, 32x32 per wave
    default: return gemm_glds_launch_t<T, 64, 64, 2, 4, 4, 4>(a, s);                    // FDM_TILE_64x64: 8 waves, 32x16 per wave
  }
}

// Split kinds: a ring stage is twice as large (hi and lo planes of both operands), so the tile set is the part of the
// one above whose ring fits 160 KB of LDS; other FDM_TILE_* values map to the nearest member.
template <typename T>
static hipError_t gemm_dispatch_split(const fdm_gemm_args& a, hipStream_t s) {
  if (a.sched_fuse)
    return gemm_sched_fuse_launch<T>(a, s);
  const int tile_id = a.tile & FDM_TILE_ID_MASK;
  if (a.ksplit > 1) return gemm_dispatch_ksplit<T>(a, tile_id, s);
  const int want = tile_id > 0 ? tile_id : (a.batch2 >= 1 ? 0 : gemm_tile_override());
  switch (want > 0 ? want : gemm_heuristic_tile(a, 2, true)) {
    case FDM_TILE_64x64_S2: return gemm_glds_launch_t<T, 64, 64, 2, 4, 2, 4>(a, s);  // 64 KB -> 2 workgroups per CU
    case FDM_TILE_32x64_S3: return gemm_glds_launch_t<T, 32, 64, 2, 2, 3, 2>(a, s);  // 72 KB -> 2 workgroups per CU
    case FDM_TILE_128x64:
    case FDM_TILE_128x64_S3: return gemm_glds_launch_t<T, 128, 64, 4, 2, 3, 4>(a, s);   // 144 KB
    case FDM_TILE_128x128:
    case FDM_TILE_96x128:
    case FDM_TILE_256x128:
    case FDM_TILE_256x128_PP: return gemm_glds_launch_t<T, 128, 128, 2, 4, 2, 4>(a, s);    // 128 KB, one tile in flight
    case FDM_TILE_80x128: return gemm_glds_launch_t<T, 80, 128, 1, 8, 3, 4>(a, s);      // 156 KB
    case FDM_TILE_64x128: return gemm_glds_launch_t<T, 64, 128, 2, 4, 3, 4>(a, s);      // 144 KB
    default: return gemm_glds_launch_t<T, 64, 64, 2, 4, 4, 4>(a, s);                    // FDM_TILE_64x64: 128 KB ring
  }
}

}  // namespace fdm
