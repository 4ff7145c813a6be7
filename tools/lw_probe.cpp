// Round 6, item 3: which phase bounds the loader-wave k loop of gemm_glds_kernel at cfg5's / HuBERT's row counts (1992-3984 rows, where a
// launch is 2-4x its fixed cost)?  The same kernel built with parts of the loop removed (-DFDM_LW_VARIANT: bit 0 no MFMAs, bit 1 no
// fragment reads, bit 2 no LDS-DMA after the prologue), timed on one shape with one tile; variant 7 is the launch with an empty loop.
// Results of the reduced variants are meaningless; only their durations are read.  (tools/pp_probe.cpp is the same table for the ping-pong kernel.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=13 -DFDM_LW_VARIANT=<v> \
//         -DLWP_BM=128 -DLWP_BN=64 -DLWP_WM=4 -DLWP_WN=2 -DLWP_NST=4 -DLWP_LW=4 -o lw_probe_<v> tools/lw_probe.cpp;  ./lw_probe_<v> M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../face-diffusion-model_amd/csrc/gemm.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#ifndef LWP_KCH
#define LWP_KCH 8      // 16-byte chunks of K per LDS row: 8 (64-deep k-tiles) or 16 (128-deep: half the barriers per K)
#endif
#ifndef LWP_BM
#define LWP_BM 128
#define LWP_BN 64
#define LWP_WM 4
#define LWP_WN 2
#define LWP_NST 4
#define LWP_LW 4
#endif

int main(int argc, char** argv) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int a0 = 1; a0 + 2 < argc; a0 += 3) {
    const int M = atoi(argv[a0]), N = atoi(argv[a0 + 1]), K = atoi(argv[a0 + 2]);
    std::vector<unsigned short> h((size_t)(M > N ? M : N) * K + 1024);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    void *A, *W[8], *out;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&W[i], (size_t)N * K * 2)); CK(hipMemcpy(W[i], h.data() + i * 64, (size_t)N * K * 2, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&out, (size_t)M * N * 2));
    float* bias; CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
    fdm_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = K; a.ldw = K; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = FDM_BF16; a.bias = bias;
    a.out_t = out; a.ldo_t = N; a.ldr = N; a.ldo_f32 = N; a.ln_eps = 1e-5f;
    auto run = [&](int i) {
      a.W = W[i % 8];
      hipError_t e = fdm::gemm_glds_launch_h<fdm::bf16, LWP_BM, LWP_BN, LWP_WM, LWP_WN, LWP_NST, LWP_KCH, false, false, fdm::GEMM_LEAN, LWP_LW>(a, s);
      if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); exit(1); }
    };
    for (int i = 0; i < 8; ++i) run(i);
    CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t x;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 16; ++i) run(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(x, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(x, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / 16, nk = K / 64.0;
    const long long wgs = (long long)((M + LWP_BM - 1) / LWP_BM) * ((N + LWP_BN - 1) / LWP_BN);
    printf("variant %d tile %dx%d%s (%dx%d waves, %d stages, %d loaders) M=%d N=%d K=%d %lld WGs: %.2f us per launch, %.3f us per k-tile, %.0f TFLOP/s if it were the full kernel\n",
           FDM_LW_VARIANT, LWP_BM, LWP_BN, LWP_KCH == 16 ? "k128" : "", LWP_WM, LWP_WN, LWP_NST, LWP_LW, M, N, K, wgs, us, us / nk, 2.0 * M * N * K / us / 1e6);
    CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
    CK(hipFree(A)); CK(hipFree(out)); CK(hipFree(bias));
    for (int i = 0; i < 8; ++i) CK(hipFree(W[i]));
  }
  return 0;
}
