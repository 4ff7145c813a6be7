// The C ABI used without Python or torch: plain HIP runtime + libfdm_hip.so.
// GEMM (fp32 path: exact fp32 MFMA chain) against a host loop, then the recorded-program / hipGraph replay mechanism.
// Built and run by tests/test_abi_c_gpu.py:  hipcc abi_smoke.cpp -I include -L <dir> -lfdm_hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fdm_hip.h"

#define CK(x) do { if ((x) != 0) { printf("FAIL %s: %s\n", #x, fdm_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  if (!fdm_device_ok()) { printf("no gfx950 device\n"); return 2; }
  const int M = 70, N = 96, K = 64;
  std::vector<float> A(M * K), W(N * K), bias(N), C(M * N), ref(M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : W) v = rnd() * 0.125f;
  for (auto& v : bias) v = rnd();
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)A[m * K + k] * W[n * K + k];
      const double v = acc + bias[n];
      ref[m * N + n] = (float)(v > 0 ? v : 0);      // FDM_ACT_RELU
    }
  float *dA, *dW, *db, *dC;
  HK(hipMalloc(&dA, A.size() * 4)); HK(hipMalloc(&dW, W.size() * 4)); HK(hipMalloc(&db, bias.size() * 4)); HK(hipMalloc(&dC, C.size() * 4));
  HK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  HK(hipStreamCreate(&st));

  fdm_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.M = M; g.N = N; g.K = K; g.batch = 1; g.dtype = FDM_F32;
  g.bias = db; g.act = FDM_ACT_RELU; g.out_f32 = dC; g.ldo_f32 = N;
  CK(fdm_op_gemm(&g, st));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (size_t i = 0; i < C.size(); ++i) worst = fmax(worst, fabs((double)C[i] - ref[i]));
  printf("gemm max|hip - host| = %.3e\n", worst);
  if (worst > 1e-5) return 1;

  // invalid arguments are reported, not executed
  g.K = 48;
  if (fdm_op_gemm(&g, st) != FDM_ERR_SHAPE) { printf("FAIL: bad K accepted\n"); return 1; }
  g.K = K;

  // recorded program: two launches captured into a hipGraph and replayed
  HK(hipMemset(dC, 0, C.size() * 4));
  fdm_prog* prog = nullptr;
  CK(fdm_prog_create(&prog));
  CK(fdm_prog_begin(prog));
  CK(fdm_op_gemm(&g, nullptr));
  CK(fdm_op_gemm(&g, nullptr));
  CK(fdm_prog_end(prog));
  if (fdm_prog_num_ops(prog) != 2) { printf("FAIL: program holds %d ops\n", fdm_prog_num_ops(prog)); return 1; }
  CK(fdm_prog_instantiate(prog, st));
  CK(fdm_prog_replay(prog, 3, st));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  worst = 0;
  for (size_t i = 0; i < C.size(); ++i) worst = fmax(worst, fabs((double)C[i] - ref[i]));
  printf("graph replay max|hip - host| = %.3e\n", worst);
  CK(fdm_prog_destroy(prog));
  if (worst > 1e-5) return 1;
  printf("abi_smoke ok (libfdm_hip version %d)\n", fdm_version());
  return 0;
}
