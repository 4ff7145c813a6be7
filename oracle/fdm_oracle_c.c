/* Plain-C restatement with a FIXED operation order (oracle / test infrastructure only).
 *
 * vq_argmin_ref: VectorQuantizer.forward nearest-code search (models/lib/quantizer.py:38-45):
 *   d_k = (sum_i z_i^2 + sum_i e_ki^2) - 2 * sum_i z_i e_ki,  argmin with lowest-index tie-break.
 * torch leaves the summation order of sum/matmul to its BLAS, so indices of exact mathematical ties
 * are not reproducible across libraries; this file pins ONE order (sequential fmaf chains over i),
 * the same order the HIP kernel (csrc/elementwise.hpp: vq_quant_kernel) uses, so the two are
 * bit-identical on every input, near-ties included.
 * sched_ddpm_ref / sched_ddim_ref: the unfused fp32 expression order of p_sample / ddim_sample
 * (video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:632-656, 693-708).                */
#include <math.h>
#include <stdint.h>

void vq_argmin_ref(const float* z, const float* E, long R, int c, int K, int64_t* idx) {
  for (long r = 0; r < R; ++r) {
    const float* zr = z + r * c;
    float z2 = 0.f;
    for (int i = 0; i < c; ++i) z2 = fmaf(zr[i], zr[i], z2);
    float best = INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const float* e = E + (long)k * c;
      float e2 = 0.f, dot = 0.f;
      for (int i = 0; i < c; ++i) {
        e2 = fmaf(e[i], e[i], e2);
        dot = fmaf(zr[i], e[i], dot);
      }
      volatile float s = z2 + e2;
      volatile float t = 2.f * dot;
      float dk = s - t;
      if (dk < best) { best = dk; bk = k; }
    }
    idx[r] = bk;
  }
}

void sched_ddpm_ref(const float* x0, const float* x, const float* z, float c1, float c2, float sigma, int t, long n, float* out) {
  for (long i = 0; i < n; ++i) {
    volatile float a = c1 * x0[i];
    volatile float b = c2 * x[i];
    float m = a + b;
    if (t > 0) {
      volatile float s = sigma * z[i];
      m = m + s;
    }
    out[i] = m;
  }
}

void sched_ddim_ref(const float* x0, const float* x, float sra, float srm1, float san, float cn, long n, float* out) {
  for (long i = 0; i < n; ++i) {
    volatile float a = sra * x[i];
    volatile float b = a - x0[i];
    volatile float eps = b / srm1;
    volatile float p = x0[i] * san;
    volatile float q = cn * eps;
    out[i] = p + q;
  }
}
