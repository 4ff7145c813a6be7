// Probe: what does an in-launch exchange of per-row LayerNorm partials between the column tiles of one GEMM row block
// cost on gfx950?  Every workgroup emulates a GEMM tile (streams `kbytes` through its CU, then R exchange rounds, then
// its output stores).  An exchange round: 64 rows x 2 words per workgroup are published as 8-byte {tag, value} granules
// (relaxed agent-scope atomics: the data is the flag, cdna_hip_programming.md Guideline 16 R2), then every workgroup
// collects the granules of all NT column tiles of its row block (NT x 64 x 2 words).  Spins are bounded.
//   hipcc -O2 --offload-arch=gfx950 -o row_exchange_probe row_exchange_probe.cpp && ./row_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef unsigned long long u64;

struct Args {
  const float4* src; float4* dst; u64* slots; const unsigned* epoch; unsigned* err;
  int RB, NT, rounds, same_xcd, launch, kbytes;
};

__device__ __forceinline__ float expect(int G, int c, int row, int which, int r, unsigned tag) {
  return (float)(G * 131 + c * 17 + row * 3 + which + r * 7) + (float)(tag & 1023) * 0.25f;
}

__global__ __launch_bounds__(512) void bump(unsigned* epoch) { if (threadIdx.x == 0 && blockIdx.x == 0) epoch[0] += 1; }

__global__ __launch_bounds__(512) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, id = blockIdx.x;
  int G, c;
  if (a.same_xcd) {       // the NT column tiles of a row block have equal id % 8 (one XCD under round-robin placement)
    const int x = id & 7, j = id >> 3;
    G = (j / a.NT) * 8 + x; c = j % a.NT;
  } else {
    G = id / a.NT; c = id % a.NT;
  }
  if (G >= a.RB) return;
  // "k loop": stream kbytes through this CU
  float4 s = {0.f, 0.f, 0.f, 0.f};
  const int iters = a.kbytes / (512 * 16);
  const float4* sp = a.src + ((size_t)(G * 7 + c * 13) * 4096) % (1 << 20);
  for (int i = 0; i < iters; ++i) { const float4 v = sp[(size_t)i * 512 + t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const unsigned tag = a.epoch[0] * 32u + (unsigned)a.launch;
  float chk = s.x + s.y + s.z + s.w;
  const size_t per_round = (size_t)((a.RB + 7) / 8 * 8) * a.NT * 128;
  for (int r = 0; r < a.rounds; ++r) {
    u64* base = a.slots + r * per_round + (size_t)G * a.NT * 128;
    if (t < 128) {
      const float v = expect(G, c, t >> 1, t & 1, r, tag) + chk * 0.f;
      __hip_atomic_store((gu64*)(base + (size_t)c * 128 + t), ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int n = a.NT * 128;
    float got[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned spins = 0;
    bool done[4] = {false, false, false, false};
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = t + 512 * k;
        if (i < n && !done[k]) {
          const u64 x = __hip_atomic_load((gu64*)(base + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(x >> 32) == tag) { done[k] = true; got[k] = __uint_as_float((unsigned)x); } else ok = false;
        }
      }
      if (ok) break;
      if (++spins > (1u << 18)) { if (t == 0) atomicAdd(a.err, 1u << 16); break; }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = t + 512 * k;
      if (i < n) {
        const int cc = i / 128, row = (i % 128) >> 1, which = i & 1;
        if (done[k] && got[k] != expect(G, cc, row, which, r, tag)) atomicAdd(a.err, 1u);
        chk += got[k];
      }
    }
    ((float*)smem)[t] = chk;
    __syncthreads();
    chk = ((float*)smem)[(t * 7) & 511];
    __syncthreads();
  }
  // "epilogue": 16 B per thread x 2
  float4* dp = a.dst + ((size_t)G * a.NT + c) * 1024;
  dp[t] = float4{chk, s.x, s.y, s.z};
  dp[t + 512] = float4{chk, s.w, s.y, s.z};
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float4 *src, *dst; u64* slots; unsigned *epoch, *err;
  CK(hipMalloc(&src, (size_t)(1 << 20) * 16 + (64 << 20)));
  CK(hipMemset(src, 0, (size_t)(1 << 20) * 16 + (64 << 20)));
  CK(hipMalloc(&dst, (size_t)1024 * 1024 * 16));
  CK(hipMalloc(&slots, (size_t)2 * 64 * 16 * 128 * 8));
  CK(hipMemset(slots, 0, (size_t)2 * 64 * 16 * 128 * 8));
  CK(hipMalloc(&epoch, 64)); CK(hipMemset(epoch, 0, 64));
  CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
  const int NL = 32;
  struct Shape { int RB, NT; const char* name; };
  const Shape shapes[] = {{13, 16, "cfg2 800x1024"}, {32, 16, "cfg5 1992x1024"}, {38, 8, "cfg3 2400x512"}};
  for (const Shape& sh : shapes)
    for (int kb : {0, 256})
      for (int same : {1, 0})
        for (int R : {0, 1, 2}) {
          if (R == 0 && same == 0 && kb == 0) continue;
          Args a{src, dst, slots, epoch, err, sh.RB, sh.NT, R, same, 0, kb * 1024};
          const int grid = (sh.RB + 7) / 8 * 8 * sh.NT;
          hipGraph_t g; hipGraphExec_t x;
          CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
          hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, epoch);
          for (int i = 0; i < NL; ++i) { a.launch = i; hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 48 * 1024, s, a); }
          CK(hipStreamEndCapture(s, &g));
          CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipGraphLaunch(x, s)); CK(hipStreamSynchronize(s));
          const int reps = 30;
          CK(hipEventRecord(e0, s));
          for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(x, s));
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          unsigned he; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
          printf("%-16s grid %3d k-stream %3d KB  %s  rounds %d : %.2f us per launch   (mismatches %u, timeouts %u)\n", sh.name, grid, kb,
                 same ? "same-XCD" : "spread  ", R, ms * 1e3 / reps / NL, he & 0xffff, he >> 16);
          CK(hipMemset(err, 0, 64));
          CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
        }
  return 0;
}
