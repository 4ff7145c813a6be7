// Fused attention for short sequences (L <= ~1500, head_dim 64 / 128) on gfx950 MFMA.
//
// One workgroup (4 wavefronts) owns 16 or 32 query rows of one (clip, head).  The key tiles are dealt
// round-robin to the 4 waves, each running its own online softmax straight from L2 (K and V^T tiles
// are a few KB per head), and the four partial (m, l, O) states are merged through LDS at the end:
// with L <= 600 a query tile has at most 19 key tiles, so the per-wave dependent chain is <= 5 tiles
// instead of 19 -- these kernels are latency-bound, not FLOP-bound.
//
// "Swapped" formulation so that nothing has to be transposed between the two products:
//   S^T[key][query] = K * Q^T      A-port rows = keys,   B-port cols = queries
//   O^T[e][query]   = V^T * P^T    A-port rows = e, B-port = P^T
// K and V arrive "fragment-packed" (include/fdm_hip.h, fdm_attn_args; written by the QKV GEMM's epilogue):
// each A-port fragment of a key tile is one contiguous 1 KB run, so every fragment load is a single fully
// coalesced request (measured: the 16-rows x 64 B gathers of a row-major K cost ~30 % of the kernel).
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

// accurate expf on the parity paths (fp32, split fp16), hardware exp2 on the bf16 (throughput) path
// (bf16 callers pre-multiply the exponent by log2(e))
template <typename T> __device__ __forceinline__ float fexp(float x) {
  if constexpr (is_fast16<T>::value) return __builtin_amdgcn_exp2f(x);
  else return expf(x);
}

// T = float | bf16 | f16x3_t.  The split kind keeps Q, K and V as fp16 plane pairs (x = hi + lo / 2^11) and evaluates both
// products in three 16-bit MFMA passes (hi.hi into the main accumulator, hi.lo + lo.hi into a second one, combined once per
// key tile for the scores and once at the end for O); the probabilities are split the same way in registers.  fp32-class
// results at a fifth of the fp32 kernel's MFMA time and half its key tiles.
// Split kind at head_dim 256 (BIWI: 4 heads x 256): the fragments of a whole key tile (K 128 + V 128 registers) cannot be held
// AT ONCE beside Q (64) and the two O^T accumulators (128), so they are held ONE AFTER THE OTHER at one wave per SIMD: the whole
// K tile for the scores, then -- requested once the scores are done, its latency under the softmax arithmetic -- the whole V
// tile into the registers K has freed (512 registers, no scratch).  Two dependent L2 round trips per key tile; streaming K and V
// in halves (round 3's first form: four round trips, 463 registers) was 3.7 us slower per launch (cfg4 f16x3 0.826 -> 0.795 ms).
template <typename T, int HD> constexpr bool attn_streamed() { return Opnd<T>::NP == 2 && HD == 256; }

template <typename T, int HD, int QS>
__global__ __launch_bounds__(256, (attn_streamed<T, HD>() ? 1 : 2)) void attn_kernel(const void* pQ, const void* pKp, const void* pVp, int p_q_lo_off, int p_kv_lo_off, int pL, int pB, int pH, int p_ldq,
                                                                                     int pLpad, const fdm_attn_args p) {
  // (the leading arguments repeat fields of p: kernel-argument preload, 13 SGPRs -- the Q / K fragment loads go out without waiting for
  //  the argument block; see gemm_glds_kernel.  The plane offsets travel as 32-bit element counts: attn_launch_t checks them.)
  // QS = 16-query sub-tiles per workgroup (1 or 2).  With QS = 2 every K / V^T fragment fetched from L2 feeds
  // two S^T and two O^T products: the kernel is bound by L2 -> register fragment traffic, which this halves.
  using E = typename Opnd<T>::E;
  constexpr int NP = Opnd<T>::NP;
  constexpr float SCL = Opnd<T>::SCALE;
  constexpr int EPC = 16 / (int)sizeof(E);     // elements per 16 B fragment chunk
  constexpr int NKS = HD / (4 * EPC);          // MFMA k-steps over the head dim
  constexpr int KT = 4 * EPC;                  // keys per tile (16-bit kinds 32, fp32 16)
  constexpr int NSUB = KT / 16;                // 16-key sub-tiles per tile
  constexpr int NC = HD / 16;                  // 16-row chunks of O^T
  constexpr int BQ = 16 * QS;
  constexpr bool STREAM = attn_streamed<T, HD>();
  static_assert(!STREAM || QS == 1, "the K-then-V form runs one query sub-tile");

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  // XCD-aware 1-D launch: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so the id is
  // decoded such that all query tiles of one (clip, head) land on ONE XCD -- its K / V block is then fetched into one
  // L2 instead of eight (FETCH_SIZE showed ~2.3x the algorithmic bytes with the plain (tile, head, clip) grid) -- and
  // the longest (last, causal) query tiles are dispatched first.
  const int L = pL;
  const int nqt = (L + BQ - 1) / BQ;
  const int grp = blockIdx.x / (8 * nqt), rem = blockIdx.x % (8 * nqt);
  const int bh = grp * 8 + (rem & 7);
  if (bh >= pB * pH) return;
  const int b = bh / pH, h = bh - b * pH;
  const int q0 = (nqt - 1 - (rem >> 3)) * BQ;
  // (+4 floats per row: the merge writes below put 8 lanes on 8 consecutive rows at one column; without the pad they share
  //  a bank group -- SQ_LDS_BANK_CONFLICT was 79 % of this kernel's LDS cycles)
  __shared__ __attribute__((aligned(16))) float part_o[4][BQ][HD + 4];
  __shared__ float part_m[4][BQ], part_l[4][BQ];

  const E* Q = (const E*)pQ + (size_t)b * L * p_ldq + (size_t)h * HD;
  const E* Kp = (const E*)pKp + (size_t)(b * pH + h) * HD * pLpad + (size_t)lane * EPC;
  const E* Vp = (const E*)pVp + (size_t)(b * pH + h) * HD * pLpad + (size_t)lane * EPC;
  const size_t q_lo = NP == 2 ? (size_t)p_q_lo_off : 0, kv_lo = NP == 2 ? (size_t)p_kv_lo_off : 0;

  int qi[QS];                              // this lane's query index in each sub-tile
  u32x4 qf[QS][NKS][NP];
  f32x4 o[QS][NC];
  f32x4 ol[NP == 2 ? QS : 1][NP == 2 ? NC : 1];       // split kind: the two small products of O^T
  float m_run[QS], l_part[QS];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    qi[u] = q0 + 16 * u + r16;
    const int qrow = min(qi[u], L - 1);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) qf[u][ks][pl] = *(const u32x4*)(Q + pl * q_lo + (size_t)qrow * p_ldq + (ks * 4 + g) * EPC);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      o[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (NP == 2) ol[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    m_run[u] = -INFINITY;
    l_part[u] = 0.f;
  }

  // bf16 path: softmax in the log2 domain (scale and slope carry log2(e); v_exp_f32 is a bare exp2)
  constexpr float LG = is_fast16<T>::value ? 1.4426950408889634f : 1.f;
  const float sc_mul = p.scale * LG;
  const float slope = p.slopes ? p.slopes[h] * LG : 0.f;
  const float inv_period = 1.f / (float)p.period;
  const bool fastbias = !p.slopes || p.period >= 8;
  const int goff = (NSUB == 2 ? 8 : 4) * g;
  const int kend = p.causal ? min(q0 + BQ, L) : L;     // keys [0, kend) can be visible to this workgroup
  const int ntiles = (kend + KT - 1) / KT;

  // Measured (twice: with the accumulators in AGPRs and again after they moved to VGPRs, 6.4 vs 6.1 us at L = 200):
  // software prefetch of the next tile's fragments (register double buffers, copy or ping-pong) is SLOWER here: the extra VGPRs cost residency, and residency is what hides the L2 round trips of these
  // short per-wave chains.  So each tile simply loads its fragments and uses them.
  for (int kt = wave; kt < ntiles; kt += 4) {
    const int kbase = kt * KT;
    // fragment-packed K / V: every operand fragment of this key tile is one contiguous 1 KB run
    u32x4 kcur[NSUB][NKS][NP];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) kcur[s][ks][pl] = *(const u32x4*)(Kp + pl * kv_lo + (size_t)((kt * NSUB + s) * NKS + ks) * (64 * EPC));
    u32x4 vf[NC][NP];
    auto load_v = [&](int c0, int c1) {
#pragma unroll
      for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) vf[c][pl] = *(const u32x4*)(Vp + pl * kv_lo + (size_t)(kt * NC + c) * (64 * EPC));
    };
    // split kind: the second half of V (all of it at head_dim 256) is requested after the scores, into the registers the K
    // fragments free, which keeps the kernel at two (one) waves per SIMD; its latency overlaps the softmax arithmetic
    constexpr int NC_EARLY = STREAM ? 0 : ((NP == 2) ? NC / 2 : NC);
    load_v(0, NC_EARLY);
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      // a causal sub-tile whose last query precedes this key tile sees none of it (wave-uniform skip);
      // it also guarantees every processed tile starts at a key visible to all 16 queries (finite row maxima)
      if (p.causal && kbase > q0 + 16 * u + 15) continue;
      f32x4 sc[NSUB];
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NP == 1) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) Mma<T>::run(a, kcur[s][ks][0], qf[u][ks][0]);
        } else {
          f32x4 al = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            mma16<E>(a, kcur[s][ks][0], qf[u][ks][0]);
            mma16<E>(al, kcur[s][ks][0], qf[u][ks][1]);
            mma16<E>(al, kcur[s][ks][1], qf[u][ks][0]);
          }
          a += al * (1.f / SCL);
        }
        sc[s] = a;
      }
      if constexpr (NP == 2 && QS == 1) load_v(NC_EARLY, NC);
      // scores -> scaled, biased, masked.  This lane holds keys kbase + goff + j with j = 4s + r (j < 8), so with
      // D = qi - (kbase + goff):  floor((qi - kj) / period) = floor(D / period) - (j > D mod period)   (period >= 8)
      // -> one compare/select/fma per score instead of an int->float convert, floor and two multiplies.
      const int D = qi[u] - kbase - goff;
      float b0 = 0.f, b1 = 0.f;
      int rem = 1 << 30;
      if (p.slopes) {
        const float f0 = floorf(((float)D + 0.5f) * inv_period);     // (n + 0.5) / period never rounds across an integer
        rem = D - (int)f0 * p.period;
        b0 = slope * f0;
        b1 = b0 - slope;
      }
      // only tiles that straddle the causal diagonal or the end of the sequence need the mask
      const bool edge = (kbase + KT > L) || (p.causal && kbase + KT - 1 > q0 + 16 * u);
      float mx = -INFINITY;
      if (fastbias) {
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 4 * s + r;
            sc[s][r] = sc[s][r] * sc_mul - ((j > rem) ? b1 : b0);
          }
      } else {
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 4 * s + r;
            float v = sc[s][r] * sc_mul;
            if (p.slopes) v -= slope * floorf(((float)(D - j) + 0.5f) * inv_period);
            sc[s][r] = v;
          }
      }
      if (edge) {
        const int jmax = min(p.causal ? D : (1 << 30), L - 1 - kbase - goff);     // largest visible j of this lane
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[s][r] = (4 * s + r > jmax) ? -INFINITY : sc[s][r];
      }
#pragma unroll
      for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[s][r]);
      mx = rows_max(mx);
      const float m_new = fmaxf(m_run[u], mx);
      const float alpha = fexp<T>(m_run[u] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = fexp<T>(sc[s][r] - m_new);
          sc[s][r] = e;
          psum += e;
        }
      l_part[u] = l_part[u] * alpha + psum;
      m_run[u] = m_new;
      // the running maximum usually stops moving after the first tiles: skip the (AGPR round-trip) rescale then
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          o[u][c] *= alpha;
          if constexpr (NP == 2) ol[u][c] *= alpha;
        }
      }
      // P^T fragment(s) for the B port
      if constexpr (NP == 1) {
        u32x4 pf;
        if constexpr (sizeof(E) == 2) {
          typedef __attribute__((ext_vector_type(8))) E e8p;
          e8p pb = {(E)sc[0][0], (E)sc[0][1], (E)sc[0][2], (E)sc[0][3],
                    (E)sc[NSUB - 1][0], (E)sc[NSUB - 1][1], (E)sc[NSUB - 1][2], (E)sc[NSUB - 1][3]};
          pf = __builtin_bit_cast(u32x4, pb);
        } else {
          pf = __builtin_bit_cast(u32x4, sc[0]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) Mma<T>::run(o[u][c], vf[c][0], pf);
      } else {
        typedef __attribute__((ext_vector_type(8))) E e8;
        e8 ph, plo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = sc[j >> 2][j & 3];         // probabilities in [0, 1]: no clamp needed
          const E hi = (E)x;
          ph[j] = hi;
          plo[j] = (E)((x - (float)hi) * SCL);
        }
        const u32x4 pfh = __builtin_bit_cast(u32x4, ph), pfl = __builtin_bit_cast(u32x4, plo);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          mma16<E>(o[u][c], vf[c][0], pfh);
          mma16<E>(ol[u][c], vf[c][0], pfl);
          mma16<E>(ol[u][c], vf[c][1], pfh);
        }
      }
    }
  }

  // ---- merge the four waves' partial states ----
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    float l_tot = l_part[u];
    l_tot = rows_sum(l_tot);
    if (g == 0) { part_m[wave][16 * u + r16] = m_run[u]; part_l[wave][16 * u + r16] = l_tot; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 ov = o[u][c];
      if constexpr (NP == 2) ov += ol[u][c] * (1.f / SCL);
      *(f32x4*)&part_o[wave][16 * u + r16][c * 16 + 4 * g] = ov;
    }
  }
  __syncthreads();
  {
    constexpr int TPQ = 256 / BQ;                    // threads per query row (16 or 8)
    constexpr int EPT = HD / TPQ;                    // head-dim elements per thread (multiple of 4)
    // A thread takes runs of RUN consecutive floats, EPT / RUN of them TPQ * RUN floats apart.  RUN = EPT (one contiguous run)
    // is conflict-free with the +4 row pad up to 8 floats per thread (measured: SQ_LDS_BANK_CONFLICT 0 at head dim 64 / 128);
    // 16 contiguous floats per thread (head dim 256) put lanes l and l + 4 of a read group on one 16-byte slot (21 % of the
    // kernel's LDS cycles), so that case reads two runs of 8, i.e. the head-dim-128 pattern twice.
    constexpr int RUN = (EPT == 16 && TPQ == 16) ? 8 : EPT;
    const int q = threadIdx.x / TPQ;
    const int e0 = (threadIdx.x % TPQ) * RUN;
    const int qq = q0 + q;
    float mw[4], ms = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = part_m[w][q]; ms = fmaxf(ms, mw[w]); }
    float sw[4], lsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sw[w] = fexp<T>(mw[w] - ms); lsum += sw[w] * part_l[w][q]; }   // exp(-inf) = 0 for idle waves
    const float inv = 1.f / lsum;
    if (qq < L) {
      const size_t oo = ((size_t)b * L + qq) * p.ldo + (size_t)h * HD + e0;
#pragma unroll
      for (int rr = 0; rr < EPT / RUN; ++rr)
#pragma unroll
      for (int jj = 0; jj < RUN; jj += 4) {
        const int j = rr * TPQ * RUN + jj;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) v += *(const f32x4*)&part_o[w][q][e0 + j] * sw[w];
        v *= inv;
        if constexpr (std::is_same<T, float>::value) {
          // fp32 attention feeding a split-operand GEMM: O is written as the plane pair
          if (p.o_split == FDM_F16X3) store_opnd4<f16x3_t>((f16*)p.O + oo + j, p.o_lo_off, v);
          else *(f32x4*)((float*)p.O + oo + j) = v;
        } else {
          store_opnd4<T>((E*)p.O + oo + j, p.o_lo_off, v);
        }
      }
    }
  }
}

#define ATTN_PRELOAD_ARGS a.Q, a.Kp, a.Vp, (int)a.q_lo_off, (int)a.kv_lo_off, a.L, a.B, a.H, (int)a.ldq, a.Lpad
template <typename T, int HD>
static void attn_launch_t(const fdm_attn_args& a, hipStream_t s) {
  // two query sub-tiles per workgroup once the sequence is long enough that halving the K / V traffic matters more
  // than the number of workgroups; head_dim 256 and the split kind keep one (register budget)
  constexpr int qs2 = 384;
  const int groups = (a.B * a.H + 7) / 8 * 8;        // (clip, head) pairs padded to whole XCD rounds
  if constexpr (HD <= 128 && Opnd<T>::NP == 1) {
    if (a.L >= qs2) {
      dim3 grid((a.L + 31) / 32 * groups);
      hipLaunchKernelGGL((attn_kernel<T, HD, 2>), grid, dim3(256), 0, s, ATTN_PRELOAD_ARGS, a);
      return;
    }
  }
  dim3 grid((a.L + 15) / 16 * groups);
  hipLaunchKernelGGL((attn_kernel<T, HD, 1>), grid, dim3(256), 0, s, ATTN_PRELOAD_ARGS, a);
}

template <typename T>
static hipError_t attn_launch_dtype(const fdm_attn_args& a, hipStream_t s) {
  if constexpr (Opnd<T>::NP == 2) {       // split kind: head_dim 64 / 128 hold a key tile's fragments, 256 streams them (one wave per SIMD)
    if (a.hd == 256) attn_launch_t<T, 256>(a, s);
    else if (a.hd == 128) attn_launch_t<T, 128>(a, s);
    else if (a.hd == 64) attn_launch_t<T, 64>(a, s);
    else return hipErrorInvalidValue;
  } else {
    if (a.hd == 256) attn_launch_t<T, 256>(a, s);
    else if (a.hd == 128) attn_launch_t<T, 128>(a, s);
    else attn_launch_t<T, 64>(a, s);
  }
  return hipGetLastError();
}

// row-major K, V -> fragment-packed (for C-ABI users whose projections do not come from fdm_op_gemm, and for tests)
template <typename T>
__global__ void pack_kv_kernel(const T* K, long long ldk, const T* V, long long ldv, T* Kp, T* Vp, int B, int H, int L, int Lpad, int hd) {
  const size_t n = (size_t)B * L * H * hd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % hd);
    const int h = (int)((i / hd) % H);
    const size_t row = i / ((size_t)hd * H);
    const int b = (int)(row / L), l = (int)(row % L);
    const size_t blk = (size_t)(b * H + h) * Lpad * hd;
    Kp[blk + kp_offset<T>(l, e, hd)] = K[row * ldk + (size_t)h * hd + e];
    Vp[blk + vp_offset<T>(l, e, hd)] = V[row * ldv + (size_t)h * hd + e];
  }
}
template <typename T>
static hipError_t pack_kv_launch(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp,
                                 int B, int H, int L, int Lpad, int hd, hipStream_t s) {
  const size_t n = (size_t)B * L * H * hd;
  const int blocks = (int)min((size_t)4096, (n + 255) / 256);
  hipLaunchKernelGGL(pack_kv_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)K, ldk, (const T*)V, ldv, (T*)Kp, (T*)Vp, B, H, L, Lpad, hd);
  return hipGetLastError();
}

}  // namespace fdm
