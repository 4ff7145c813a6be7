// C[M,N] = epilogue(A[M,K] * W[N,K]^T) on MFMA, for gfx950.
//
// Layout / tiling
//  * A and W are both K-contiguous ("NT" GEMM: nn.Linear weights are [out, in]).
//  * Workgroup = 256 threads = 4 wavefronts (2 x 2), tile BM x BN, k-tile = 128 bytes of K per
//    row (64 bf16 / 32 fp32), so the LDS image and the staging code are byte-identical for both
//    operand types.  LDS rows are 128 B with a 16-byte-chunk XOR swizzle (chunk ^ (row & 7)) so
//    that the ds_read_b128 fragment reads of 16 different rows at one k-chunk spread over 8 slots.
//  * HBM/L2 -> VGPR (global_load_dwordx4) -> LDS (ds_write_b128), double-buffered: the loads of
//    k-tile t+1 are issued before the MFMAs of k-tile t and written to the other LDS buffer after
//    them; one barrier per k-tile.
//  * Operands are swapped into the MFMA (W rows feed the A port, activation rows the B port), so a
//    lane ends up with 4 consecutive output columns n of one row m: the epilogue then reads
//    bias/residual and writes outputs with 16-byte (fp32) / 8-byte (bf16) vector accesses.
//  * Epilogue is fused: bias, activation, residual add, dual-dtype stores, transposed "V^T" scatter.
#pragma once
#include "common.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const fdm_gemm_args p) {
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                // elements of K per k-tile (128 B)
  constexpr int MI = BM / 32, NI = BN / 32;  // 16x16 MFMA tiles per wave along m, n
  constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
  constexpr int BUF = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, r16 = lane & 15;
  const int z = blockIdx.z;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int M = p.M, N = p.N;

  const T* A = (const T*)p.A + (size_t)z * p.a_batch_stride;
  const T* W = (const T*)p.W + (size_t)z * p.w_batch_stride;

  const char* a_src[A_CH];
  const char* w_src[W_CH];
  int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int c = tid + i * 256, row = c >> 3, kc = c & 7;
    int grow = min(m0 + row, M - 1);
    a_src[i] = (const char*)(A + (size_t)grow * p.lda) + kc * 16;
    a_dst[i] = row * 128 + ((kc ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < W_CH; ++i) {
    int c = tid + i * 256, row = c >> 3, kc = c & 7;
    int grow = min(n0 + row, N - 1);
    w_src[i] = (const char*)(W + (size_t)grow * p.ldw) + kc * 16;
    w_dst[i] = BM * 128 + row * 128 + ((kc ^ (row & 7)) << 4);
  }

  u32x4 ra[A_CH], rw[W_CH];
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
#pragma unroll
  for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i]);
#pragma unroll
  for (int i = 0; i < W_CH; ++i) rw[i] = *(const u32x4*)(w_src[i]);
#pragma unroll
  for (int i = 0; i < A_CH; ++i) *(u32x4*)(smem + a_dst[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < W_CH; ++i) *(u32x4*)(smem + w_dst[i]) = rw[i];
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      const size_t off = (size_t)(kt + 1) * 128;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i] + off);
#pragma unroll
      for (int i = 0; i < W_CH; ++i) rw[i] = *(const u32x4*)(w_src[i] + off);
    }
    const char* base = smem + (kt & 1) * BUF;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 af[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int row = wm * (BM / 2) + mi * 16 + r16;
        af[mi] = *(const u32x4*)(base + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int row = wn * (BN / 2) + ni * 16 + r16;
        wf[ni] = *(const u32x4*)(base + BM * 128 + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], wf[ni], af[mi]);  // D[row=n][col=m]
    }
    if (more) {
      char* nb = smem + ((kt + 1) & 1) * BUF;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) *(u32x4*)(nb + a_dst[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < W_CH; ++i) *(u32x4*)(nb + w_dst[i]) = rw[i];
    }
    __syncthreads();
  }

  // ---- fused epilogue: lane owns C[m = .. + r16][n = .. + 4g + (0..3)] ----
  const float* bias = p.bias ? p.bias + (size_t)z * p.bias_batch_stride : nullptr;
  const size_t ocol = (size_t)z * p.out_batch_stride;
  const bool vec_f32 = p.out_f32 && (p.ldo_f32 % 4 == 0) && (((uintptr_t)(p.out_f32 + ocol)) % 16 == 0);
  const bool vec_t = p.out_t && (p.ldo_t % 4 == 0) && (((uintptr_t)((T*)p.out_t + ocol)) % (4 * sizeof(T)) == 0);
  const bool vec_r = p.resid && (p.ldr % 4 == 0) && (((uintptr_t)(p.resid + ocol)) % 16 == 0);
  const int vt_H = p.out_vt ? (N - p.vt_col0) / p.vt_hd : 0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * (BM / 2) + mi * 16 + r16;
    if (m >= M) continue;
    const size_t rrow = p.resid_row_mod > 0 ? (size_t)(m % p.resid_row_mod) : (size_t)m;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (BN / 2) + ni * 16 + 4 * g;
      if (n >= N) continue;
      f32x4 v = acc[mi][ni];
      const bool full = (n + 3 < N);
      if (bias) {
        if (full) {
          f32x4 b = *(const f32x4*)(bias + n);
          v += b;
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] += bias[n + j];
        }
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], p.act);
      }
      if (p.resid) {
        const float* rp = p.resid + ocol + rrow * p.ldr + n;
        if (full && vec_r) {
          v += *(const f32x4*)rp;
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] += rp[j];
        }
      }
      if (p.out_vt && n >= p.vt_col0) {
        // scatter transposed: Vt[((b*H + h)*hd + e)*Lpad + l]
        const int b = m / p.vt_L, l = m - b * p.vt_L;
        for (int j = 0; j < 4; ++j) {
          if (n + j >= N) break;
          const int cc = n + j - p.vt_col0;
          const int h = cc / p.vt_hd, e = cc - h * p.vt_hd;
          ((T*)p.out_vt)[((size_t)(b * vt_H + h) * p.vt_hd + e) * p.vt_Lpad + l] = from_f32<T>(v[j]);
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
        T* op = (T*)p.out_t + ocol + (size_t)m * p.ldo_t + n;
        if (full && vec_t) {
          if constexpr (sizeof(T) == 4) {
            *(f32x4*)op = v;
          } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4*)op = o;
          }
        } else {
          for (int j = 0; j < 4; ++j)
            if (n + j < N) op[j] = from_f32<T>(v[j]);
        }
      }
    }
  }
}

template <typename T, int BM, int BN>
static hipError_t gemm_launch_t(const fdm_gemm_args& a, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.batch > 0 ? a.batch : 1);
  const int lds = 2 * (BM + BN) * 128;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

// Tile choice: the path's GEMMs have M = B*L of a few hundred to a few thousand rows, so a
// 128x128 tiling often leaves most of the 256 CUs idle; fall back to 64x64 until the 128x128
// grid covers the chip at least ~1.5 times.
static hipError_t gemm_launch(const fdm_gemm_args& a, hipStream_t s) {
  const long long batch = a.batch > 0 ? a.batch : 1;
  const long long big = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * batch;
  const bool use_big = big >= 384;
  if (a.dtype == FDM_BF16) {
    return use_big ? gemm_launch_t<bf16, 128, 128>(a, s) : gemm_launch_t<bf16, 64, 64>(a, s);
  }
  return use_big ? gemm_launch_t<float, 128, 128>(a, s) : gemm_launch_t<float, 64, 64>(a, s);
}

}  // namespace fdm
