// Device functions of the diffusion scheduler update (DDPM posterior sample / DDIM eta = 0) and its Philox noise,
// shared by sched_kernel (elementwise.hpp) and the GEMM epilogue's fused form (gemm.hpp) so both produce the same bits.
#pragma once
#include "common.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: counter = (element/4, step, global clip, 0), key = seed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, unsigned quad, unsigned step, unsigned clip) {
  unsigned r[4];
  philox4x32_10(quad, step, clip, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r[0] + 0.5f) * k, u1 = (float)r[1] * k;
  const float u2 = ((float)r[2] + 0.5f) * k, u3 = (float)r[3] * k;
  // hardware log2 / sin / cos (v_log_f32, v_sin_f32, v_cos_f32 take the angle in turns): the noise is a
  // sampling input, not a parity surface, and the accurate libm forms made this kernel VALU-bound
  const float ra = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(fminf(u0, 1.f)));   // -2 ln u = -2 ln2 log2 u
  const float rb = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(fminf(u2, 1.f)));
  const float s0 = __builtin_amdgcn_sinf(u1), c0 = __builtin_amdgcn_cosf(u1);
  const float s1 = __builtin_amdgcn_sinf(u3), c1 = __builtin_amdgcn_cosf(u3);
  return f32x4{ra * c0, ra * s0, rb * c1, rb * s1};
}

// ------------------------------------------------------------------------------------------------
// Scheduler step (DDPM posterior sample / DDIM eta=0 update / CFG mix), 16 B per lane.
// Explicit _rn operations (no FMA contraction) so the fp32 result is bit-identical to the
// reference's unfused torch expression order.
// ------------------------------------------------------------------------------------------------
// Per-step scalars of the update (k = step index, t = tseq[k]) and the update of 4 consecutive latent elements starting at
// flat element e of the x buffer.  Shared by sched_kernel and by the GEMM epilogue's fused form (fdm_gemm_args.sched_fuse),
// so both produce the same bits.
struct SchedCoef { int k, t; float c1, c2, sg, sra, srm1, san, cn; unsigned long long seed; int clip0; };
__device__ __forceinline__ SchedCoef sched_coef_load(const fdm_sched_args& p) {
  SchedCoef c;
  c.k = p.step ? *(volatile const int*)p.step : 0;
  c.t = p.tseq ? p.tseq[c.k] : c.k;
  c.c1 = c.c2 = c.sg = c.sra = c.san = c.cn = 0.f;
  c.srm1 = 1.f;
  c.seed = p.seed_dev ? p.seed_dev[0] : p.seed;
  c.clip0 = p.seed_dev ? (int)p.seed_dev[1] : p.clip0;
  if (p.mode == 0) { c.c1 = p.c1[c.t]; c.c2 = p.c2[c.t]; c.sg = p.sigma[c.t]; }
  if (p.mode == 1) { c.sra = p.sra[c.t]; c.srm1 = p.srm1[c.t]; c.san = p.sqrt_an[c.k]; c.cn = p.c_n[c.k]; }
  return c;
}
__device__ __forceinline__ f32x4 sched_update4(const fdm_sched_args& p, const SchedCoef& c, f32x4 x0, f32x4 x, long long e) {
  f32x4 o;
  if (p.mode == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(__fmul_rn(c.c1, x0[j]), __fmul_rn(c.c2, x[j]));
    if (c.t > 0) {
      f32x4 z;
      if (p.noise) {
        z = *(const f32x4*)(p.noise + (size_t)c.k * (p.noise_stride > 0 ? p.noise_stride : p.n) + e);
      } else {
        const int clip = (int)(e / p.n_per_clip);
        z = philox_normal4(c.seed, (unsigned)((e - (long long)clip * p.n_per_clip) >> 2), (unsigned)c.k,
                           (unsigned)(c.clip0 + clip));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(o[j], __fmul_rn(c.sg, z[j]));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(c.sra, x[j]), x0[j]), c.srm1);
      o[j] = __fadd_rn(__fmul_rn(x0[j], c.san), __fmul_rn(c.cn, eps));
    }
  }
  return o;
}

}  // namespace fdm
