// Where does a k-tile of the ping-pong GEMM loop (gemm_pp_kernel, csrc/gemm.hpp) spend its time?  The same kernel built with
// parts of the loop removed (-DFDM_PP_VARIANT: bit 0 no MFMAs, bit 1 no fragment reads, bit 2 no LDS-DMA after the prologue),
// timed on one shape.  Results of the reduced variants are meaningless; only their durations are read.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DFDM_PP_VARIANT=<v> -o pp_probe_<v> tools/pp_probe.cpp
//   ./pp_probe_<v> M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>

#include "../face-diffusion-model_amd/csrc/gemm.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef PP_BM
#define PP_BM 256
#define PP_BN 128
#define PP_WM 4
#define PP_WN 2
#define PP_NST 3
#endif
#ifndef PP_KCH
#define PP_KCH 8
#endif

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 2048;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::vector<unsigned short> h((size_t)M * K);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));      // bf16 around +-0.01..0.03, random sign / mantissa
  void *A, *W[8];
  float* out;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
  for (int i = 0; i < 8; ++i) {
    CK(hipMalloc(&W[i], (size_t)N * K * 2));
    CK(hipMemcpy(W[i], h.data() + i * 64, (size_t)N * K * 2 < (size_t)M * K * 2 - 1024 ? (size_t)N * K * 2 : (size_t)M * K * 2 - 1024, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&out, (size_t)M * N * 4));
  fdm_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = K; a.ldw = K; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = FDM_BF16;
  a.out_f32 = out; a.ldo_f32 = N; a.ldr = N; a.ldo_t = N; a.ln_eps = 1e-5f;
  auto run = [&](int i) {
    a.W = W[i % 8];
    hipError_t e = fdm::gemm_pp_launch_h<fdm::bf16, PP_BM, PP_BN, PP_WM, PP_WN, PP_NST, false, false, fdm::GEMM_LEAN, PP_KCH>(a, s);
    if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); exit(1); }
  };
#ifdef FDM_PP_PHASES
  const int nwg = ((M + PP_BM - 1) / PP_BM) * ((N + PP_BN - 1) / PP_BN);
  unsigned long long* ph;
  CK(hipMalloc(&ph, (size_t)nwg * 32));
  CK(hipMemset(ph, 0, (size_t)nwg * 32));
  a.rln_gamma = (const float*)ph;
#endif
#ifdef FDM_PP_STAMPS
  unsigned long long* st;
  CK(hipMalloc(&st, 8 * 64 * 8));
  CK(hipMemset(st, 0, 8 * 64 * 8));
  a.rln_beta = (const float*)st;
#endif
  for (int i = 0; i < 8; ++i) run(i);
  CK(hipStreamSynchronize(s));
#if FDM_PP_VARIANT == 0 && !defined(FDM_PP_MFMA32)
  {   // spot check of the last launch (weights W[7]) against a host dot product of the same bf16 operands
    auto bf = [](unsigned short v) { unsigned int u = (unsigned int)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
    std::vector<float> ho((size_t)M * N);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < 256; ++t) {
      const int m = (int)(((long long)t * 7919 + 13) % M), n = (int)(((long long)t * 104729 + 7) % N);
      double r = 0.0;
      for (int k = 0; k < K; ++k) r += bf(h[(size_t)m * K + k]) * bf(h[7 * 64 + (size_t)n * K + k]);
      const double e = fabs(r - (double)ho[(size_t)m * N + n]) / (fabs(r) + 1e-3);
      if (e > worst) worst = e;
    }
    printf("spot check (256 outputs vs host fp64): worst relative error %.2e %s\n", worst, worst < 2e-3 ? "ok" : "MISMATCH");
  }
#endif
#ifdef FDM_PP_STAMPS
  {
    std::vector<unsigned long long> hs(8 * 64);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    const unsigned long long t0 = hs[0];
    printf("stamps (shader cycles since wave 0's first LOAD): per wave and k-tile: LOAD start, LOAD end (at barrier), COMPUTE start, COMPUTE end (at barrier)\n");
    for (int w : {0, 1, 4, 5})
      for (int kt = 0; kt < 12; ++kt)
        printf("  wave %d kt %2d: %6lld %6lld %6lld %6lld   load %4lld  bar %4lld  compute %4lld\n", w, kt, (long long)(hs[w * 64 + kt * 4] - t0),
               (long long)(hs[w * 64 + kt * 4 + 1] - t0), (long long)(hs[w * 64 + kt * 4 + 2] - t0), (long long)(hs[w * 64 + kt * 4 + 3] - t0),
               (long long)(hs[w * 64 + kt * 4 + 1] - hs[w * 64 + kt * 4]), (long long)(hs[w * 64 + kt * 4 + 2] - hs[w * 64 + kt * 4 + 1]),
               (long long)(hs[w * 64 + kt * 4 + 3] - hs[w * 64 + kt * 4 + 2]));
  }
#endif
#ifdef FDM_PP_PHASES
  {
    std::vector<unsigned long long> hp((size_t)nwg * 4);
    CK(hipMemcpy(hp.data(), ph, hp.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nwg; ++i) { if (hp[i * 4] < t0) t0 = hp[i * 4]; if (hp[i * 4 + 3] > t1) t1 = hp[i * 4 + 3]; }
    auto med = [&](int a_, int b_) {
      std::vector<double> v;
      for (int i = 0; i < nwg; ++i) v.push_back((double)(hp[i * 4 + b_] - hp[i * 4 + a_]) * 0.01);
      std::sort(v.begin(), v.end());
      printf(" min %.2f med %.2f max %.2f |", v[0], v[v.size() / 2], v.back());
    };
    auto rel = [&](int a_) {
      std::vector<double> v;
      for (int i = 0; i < nwg; ++i) v.push_back((double)(hp[i * 4 + a_] - t0) * 0.01);
      std::sort(v.begin(), v.end());
      printf(" min %.2f med %.2f max %.2f |", v[0], v[v.size() / 2], v.back());
    };
    printf("phases (us, %d workgroups; last launch): first entry -> last store %.2f\n  entry since first:", nwg, (double)(t1 - t0) * 0.01); rel(0);
    printf("\n  entry -> loop start (prologue, first tile):"); med(0, 1);
    printf("\n  k loop:"); med(1, 2);
    printf("\n  epilogue until stores acknowledged:"); med(2, 3);
    printf("\n  loop end since first entry:"); rel(2);
    printf("\n");
  }
#endif
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 80;
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) run(i);
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, nk = K / (PP_KCH * 8.0);
  printf("variant %d tile %dx%d M=%d N=%d K=%d: %.2f us per launch, %.3f us per k-tile, %.1f TFLOP/s if it were the full kernel\n", FDM_PP_VARIANT,
         PP_BM, PP_BN, M, N, K, us, us / nk, 2.0 * M * N * K / us / 1e6);
  return 0;
}
