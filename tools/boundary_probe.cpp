// What does a kernel boundary inside a replayed hipGraph cost, and what makes it grow beyond the 1.7 us of a trivial kernel?
// A chain of NL dependent launches of one kernel; every workgroup stamps the 100 MHz clock at its first instruction and at its
// end, so   boundary = (first stamp of launch i+1) - (last end stamp of launch i)   is separated from the time inside the kernel.
// Variants: workgroup size, dynamic LDS per workgroup, time spent inside (spin), bytes written (dirty lines in L2), bytes read
// (streamed through the XCD L2s: what pushes code, descriptors and kernel arguments out between launches).
//   hipcc -O2 --offload-arch=gfx950 -o boundary_probe boundary_probe.cpp && ./boundary_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { unsigned long long* stamps; float4* out; const float4* in; int spin_ticks; int write_f4_per_thread; int read_f4_per_thread; char pad[420]; };

__global__ void probe(const Args a) {
  extern __shared__ char smem[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1;
  for (int i = 0; i < a.write_f4_per_thread; ++i)
    a.out[((size_t)blockIdx.x * a.write_f4_per_thread + i) * blockDim.x + threadIdx.x] = float4{1.f, 2.f, 3.f, (float)i};
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < a.read_f4_per_thread; ++i) {       // streams through this XCD's L2 (every workgroup its own range)
    const float4 v = a.in[((size_t)blockIdx.x * a.read_f4_per_thread + i) * blockDim.x + threadIdx.x];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 12345.f) a.out[0] = acc;
  while ((long long)(wall_clock64() - t0) < a.spin_ticks) {}
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) { a.stamps[blockIdx.x * 2] = t0; a.stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int NL = 24, MAXWG = 1024;
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)NL * MAXWG * 16));
  float4* out; CK(hipMalloc(&out, (size_t)64 << 20));
  CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float4* in; CK(hipMalloc(&in, (size_t)256 << 20)); CK(hipMemset(in, 0, (size_t)256 << 20));
  struct V { int wg, threads, lds_kb, spin_us, write_kb_per_wg, read_kb_per_wg; };
  const V vs[] = {{208, 256, 0, 0, 0, 0},  {208, 512, 0, 0, 0, 0},  {208, 512, 64, 0, 0, 0}, {208, 512, 128, 0, 0, 0}, {208, 512, 64, 5, 0, 0}, {208, 512, 64, 10, 0, 0},
                  {208, 512, 64, 5, 16, 0}, {208, 512, 64, 5, 32, 0}, {416, 256, 32, 5, 16, 0}, {1024, 256, 0, 5, 4, 0}, {208, 512, 0, 5, 16, 0}, {208, 512, 64, 0, 16, 0},
                  {208, 512, 64, 5, 0, 64}, {208, 512, 64, 5, 0, 256}, {208, 512, 64, 5, 16, 256}, {208, 512, 64, 8, 16, 1024}};
  printf("(last column of the configuration: KB read per workgroup, streamed from a 256 MB buffer)\n");
  printf("workgroups threads LDS/WG spin  written/WG |  per launch   boundary (last end -> first start)   inside (first start -> last end)\n");
  for (const V& v : vs) {
    hipGraph_t g; hipGraphExec_t x;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NL; ++l) {
      Args a{};
      a.stamps = stamps + (size_t)l * MAXWG * 2; a.out = out; a.spin_ticks = v.spin_us * 100; a.write_f4_per_thread = v.write_kb_per_wg * 1024 / 16 / v.threads; a.in = in; a.read_f4_per_thread = v.read_kb_per_wg * 1024 / 16 / v.threads;
      hipLaunchKernelGGL(probe, dim3(v.wg), dim3(v.threads), v.lds_kb * 1024 + 16, s, a);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(x, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)NL * MAXWG * 2);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double gap = 0, inside = 0;
    for (int l = 1; l < NL; ++l) {
      unsigned long long first = ~0ull, last = 0, prev_last = 0;
      for (int w = 0; w < v.wg; ++w) {
        first = std::min(first, h[((size_t)l * MAXWG + w) * 2]); last = std::max(last, h[((size_t)l * MAXWG + w) * 2 + 1]);
        prev_last = std::max(prev_last, h[((size_t)(l - 1) * MAXWG + w) * 2 + 1]);
      }
      gap += (double)(first - prev_last) * 0.01; inside += (double)(last - first) * 0.01;
    }
    printf("%9d %7d %4d KB %3d us %6d KB %5d KB | %7.2f us %12.2f us %35.2f us\n", v.wg, v.threads, v.lds_kb, v.spin_us, v.write_kb_per_wg, v.read_kb_per_wg, ms * 1e3 / reps / NL,
           gap / (NL - 1), inside / (NL - 1));
    CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
  }
  return 0;
}
