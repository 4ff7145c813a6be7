// Launch-floor probe: time per dependent kernel boundary on one stream, eager and as a replayed hipGraph, for a
// trivial kernel with a small kernarg block and with a ~500-byte one (fdm_gemm_args is that size).
//   hipcc -O2 --offload-arch=gfx950 -o launch_floor launch_floor.cpp && ./launch_floor
// Run it under HIP_FORCE_DEV_KERNARG / DEBUG_CLR_GRAPH_PACKET_CAPTURE settings to see which knob moves the floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { int* p; char pad[480]; int last; };

__global__ void k_small(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = b.last; }
// touches memory like a small elementwise op (1 MB in, 1 MB out)
__global__ void k_copy(const float4* a, float4* o, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = a[i];
}

template <typename F> float timed(hipStream_t s, int reps, F f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  int* p; CK(hipMalloc(&p, 64));
  float4 *a, *o; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&o, 1 << 20));
  CK(hipMemset(a, 0, 1 << 20));
  const int NL = 50;
  Big b{}; b.p = p; b.last = 7;
  for (int wg : {1, 256, 1024}) {
    auto small = [&] { for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_small, dim3(wg), dim3(256), 0, s, p, i); };
    auto big = [&] { for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_big, dim3(wg), dim3(256), 0, s, b); };
    auto copy = [&] { for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, s, (const float4*)a, o, (1 << 20) / 16); };
    printf("grid %4d eager : small-arg %.3f us  big-arg %.3f us  copy1MB %.3f us per launch\n", wg,
           timed(s, 20, small) * 1e3 / NL, timed(s, 20, big) * 1e3 / NL, timed(s, 20, copy) * 1e3 / NL);
    auto graph_of = [&](auto f) {
      hipGraph_t g; hipGraphExec_t x;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      f();
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
      return x;
    };
    hipGraphExec_t xs = graph_of(small), xb = graph_of(big), xc = graph_of(copy);
    printf("grid %4d graph : small-arg %.3f us  big-arg %.3f us  copy1MB %.3f us per launch\n", wg,
           timed(s, 20, [&] { CK(hipGraphLaunch(xs, s)); }) * 1e3 / NL, timed(s, 20, [&] { CK(hipGraphLaunch(xb, s)); }) * 1e3 / NL,
           timed(s, 20, [&] { CK(hipGraphLaunch(xc, s)); }) * 1e3 / NL);
  }
  return 0;
}
