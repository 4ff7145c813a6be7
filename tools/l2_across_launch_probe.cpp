// Do lines survive in an XCD's L2 across a kernel boundary of a replayed graph?  Each launch: the first 8 workgroups (one per
// XCD under round-robin placement) time a dependent pointer chase (s_memtime around 16 global loads, one lane) three ways:
//   warm-prev : lines every one of those workgroups also touched in the PREVIOUS launch
//   warm-now  : the same lines again, inside this launch (L2 / L1 hit reference)
//   cold      : lines nothing has touched since they were written at start-up
//   hipcc -O2 --offload-arch=gfx950 -o l2_across_launch_probe l2_across_launch_probe.cpp && ./l2_across_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NLOADS = 16, STRIDE = 4096 / 8;      // chase over 16 lines, 4 KB apart (u64 elements)

__device__ unsigned long long chase(const unsigned long long* base, unsigned long long* sink) {
  unsigned long long idx = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < NLOADS; ++i) idx = __builtin_nontemporal_load(base + idx);      // (glc-style load: not served by L1)
  const unsigned long long t1 = __builtin_readcyclecounter();
  *sink = idx;
  return t1 - t0;
}

__global__ void probe(const unsigned long long* warm, const unsigned long long* cold, unsigned long long* out, unsigned long long* sink, int launch, int spin) {
  if (blockIdx.x < 8 && threadIdx.x == 0) {
    unsigned long long* o = out + ((size_t)launch * 8 + blockIdx.x) * 4;
    o[0] = chase(warm, sink + blockIdx.x);
    o[1] = chase(warm, sink + blockIdx.x);
    o[2] = chase(cold + (size_t)launch * NLOADS * STRIDE, sink + blockIdx.x);
  }
  // every workgroup streams some bytes so the launch is not trivial
  unsigned long long acc = 0;
  for (int i = 0; i < spin; ++i) acc += warm[(threadIdx.x * 8 + i) % (NLOADS * STRIDE)] ;
  if (acc == 0x1234567) sink[100] = acc;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int NL = 16;
  const size_t n = (size_t)(NL + 1) * NLOADS * STRIDE;
  std::vector<unsigned long long> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 0;
  for (int blk = 0; blk <= NL; ++blk)
    for (int i = 0; i < NLOADS; ++i) h[(size_t)blk * NLOADS * STRIDE + (size_t)i * STRIDE] = (unsigned long long)((i + 1) % NLOADS) * STRIDE;
  unsigned long long *d, *out, *sink;
  CK(hipMalloc(&d, n * 8)); CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)NL * 8 * 4 * 8)); CK(hipMalloc(&sink, 1024));
  hipGraph_t g; hipGraphExec_t x;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, s, (const unsigned long long*)d, (const unsigned long long*)(d + NLOADS * STRIDE), out, sink, l, 4);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(x, s));      // ONE replay: the "cold" lines of every launch have never been read
  CK(hipStreamSynchronize(s));
  std::vector<unsigned long long> r((size_t)NL * 8 * 4);
  CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
  double a[3] = {0, 0, 0};
  int cnt = 0;
  for (int l = 2; l < NL; ++l) for (int w = 0; w < 8; ++w) { for (int k = 0; k < 3; ++k) a[k] += (double)r[((size_t)l * 8 + w) * 4 + k] / NLOADS; ++cnt; }
  printf("cycles per dependent load (shader clock): touched in the previous launch %.0f | touched again in this launch %.0f | untouched %.0f\n", a[0] / cnt, a[1] / cnt, a[2] / cnt);
  return 0;
}
