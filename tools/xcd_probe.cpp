// Probe (round 4): what does it cost to keep a producer -> consumer chain INSIDE one XCD on gfx950?
//   A. placement: which XCD (s_getreg HW_REG_XCC_ID) do workgroups 0..15 of consecutive launches of one replayed graph land
//      on, for the grid sizes of the step program (is "id % 8" the same physical XCD from launch to launch?)
//   B. an XCD-local barrier inside a persistent launch (one workgroup of 512 threads per CU): the workgroups of one XCD
//      arrive on a counter of their own (one line per XCD), lane 0 polls it with an L1-bypassing load; nothing is fenced.
//      Variants: agent-scope vs workgroup-scope atomic add.
//   C. the same barrier as a data hand-off: every workgroup stores 16 KB (plain stores), waits for its stores, passes the
//      barrier and reads the 16 KB of ANOTHER workgroup of its XCD -- with plain loads (L1 may be stale) and with sc1 loads
//      (served by the XCD's L2); every word is checked, the slots are rewritten every iteration (consumer L1-warm).
//   D. LDS-DMA streaming rate per CU, all CUs streaming: 256 KB per workgroup from lines its own XCD stored just before (L2),
//      from lines it read just before (L2) and from lines nobody touched (Infinity Cache / HBM).
//   hipcc -O2 --offload-arch=gfx950 -o xcd_probe xcd_probe.cpp && ./xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 15;
}

// ---- A ------------------------------------------------------------------------------------------------------------------
__global__ void where(int* out, int launch) {
  if (threadIdx.x == 0 && blockIdx.x < 16) out[launch * 16 + blockIdx.x] = xcc_id();
}

// ---- B / C / D ----------------------------------------------------------------------------------------------------------
struct PArgs {
  unsigned* ticket;      // [8 * 32] one line per XCD
  unsigned* cnt;         // [8 * 32] barrier counters, one line per XCD
  unsigned* err;         // [8]: 0 stale words (plain loads), 1 stale words (sc1 loads), 2 spin time-outs, 3 bad XCD population
  u32x4* slots;          // [256][1024] 16 KB per workgroup
  char* big;             // streaming buffer
  u64* times;            // [256][8]
  int iters, scope_wg, mode;
};

__device__ __forceinline__ u32x4 load_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ bool xcd_barrier(unsigned* c, unsigned target, int scope_wg, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached L2
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (scope_wg) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { atomicAdd(err + 2, 1u); ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(512) void persistent(const PArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_idx;
  const int t = threadIdx.x;
  const int x = xcc_id();
  if (t == 0) s_idx = (int)atomicAdd(a.ticket + x * 32, 1u);
  __syncthreads();
  const int idx = s_idx;                   // this workgroup's number inside its XCD
  if (idx >= 32) { if (t == 0) atomicAdd(a.err + 3, 1u); return; }
  const int me = x * 32 + idx;
  unsigned* c = a.cnt + x * 32;
  unsigned gen = 0;
  // all 32 workgroups of the XCD are present
  if (!xcd_barrier(c, 32 * ++gen, a.scope_wg, a.err)) return;
  const u64 t0 = wall_clock64();
  if (a.mode == 0) {                       // B: barriers only
    for (int i = 0; i < a.iters; ++i)
      if (!xcd_barrier(c, 32 * ++gen, a.scope_wg, a.err)) return;
  } else if (a.mode == 1 || a.mode == 2) { // C: 16 KB hand-off to the next workgroup of the XCD, every word checked
    const int src = x * 32 + (idx + 1 + (t >> 8)) % 32;      // threads 0..255 read neighbour +1, 256..511 neighbour +2 (half each)
    unsigned bad = 0;
    for (int i = 0; i < a.iters; ++i) {
      u32x4* mine = a.slots + (size_t)me * 1024;
      for (int k = t; k < 1024; k += 512) mine[k] = u32x4{(unsigned)i, (unsigned)me, (unsigned)k, (unsigned)(i * 2654435761u + k)};
      if (!xcd_barrier(c, 32 * ++gen, a.scope_wg, a.err)) return;
      const u32x4* theirs = a.slots + (size_t)src * 1024;
      for (int k = (t & 255); k < 1024; k += 256) {
        const u32x4 v = (a.mode == 2) ? load_sc1(theirs + k) : theirs[k];
        if (v[0] != (unsigned)i || v[1] != (unsigned)src || v[2] != (unsigned)k || v[3] != (unsigned)(i * 2654435761u + k)) ++bad;
      }
      // nobody may overwrite a slot before its readers are done
      if (!xcd_barrier(c, 32 * ++gen, a.scope_wg, a.err)) return;
    }
    if (bad) atomicAdd(a.err + (a.mode == 2 ? 1 : 0), bad);
  } else {                                 // D: LDS-DMA streaming of 256 KB per workgroup, three sources
    const int wave = t >> 6, lane = t & 63;
    char* own = a.big + (size_t)me * (256 << 10);                       // written by this workgroup below
    char* nb = a.big + (size_t)(x * 32 + (idx + 1) % 32) * (256 << 10);  // written by a neighbour of the same XCD
    char* cold = a.big + (size_t)(256 + me) * (256 << 10);               // never touched in this launch
    for (int k = t; k < (256 << 10) / 16; k += 512) ((u32x4*)own)[k] = u32x4{(unsigned)k, 1u, 2u, 3u};
    if (!xcd_barrier(c, 32 * ++gen, a.scope_wg, a.err)) return;
    auto stream = [&](const char* src_base, int aux_sc1) {
      const u64 s0 = wall_clock64();
      for (int blk = 0; blk < 16; ++blk) {         // 16 x 16 KB, two ring slots of 16 KB
        char* lds = smem + (blk & 1) * 16384;
        const char* src = src_base + (size_t)blk * 16384;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int piece = wave * 2 + i;
          if (aux_sc1) __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024 + lane * 16), (lptr_t)(lds + piece * 1024), 16, 0, 16);
          else __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024 + lane * 16), (lptr_t)(lds + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return wall_clock64() - s0;
    };
    const u64 d_nb = stream(nb, 0);            // stored by the same XCD just before
    const u64 d_nb2 = stream(nb, 0);           // read by this CU just before (L2; LDS-DMA does not stay in L1?)
    const u64 d_nb_sc1 = stream(nb, 1);        // the same with the sc1 policy on the DMA
    const u64 d_cold = stream(cold, 0);        // untouched lines
    const u64 d_cold2 = stream(cold, 0);       // read just before
    if (t == 0) { u64* o = a.times + (size_t)me * 8; o[2] = d_nb; o[3] = d_nb2; o[4] = d_nb_sc1; o[5] = d_cold; o[6] = d_cold2; }
  }
  const u64 t1 = wall_clock64();
  if (t == 0) { u64* o = a.times + (size_t)me * 8; o[0] = t1 - t0; o[1] = (u64)x; }
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  // ---- A
  {
    const int grids[] = {208, 240, 800, 416, 256, 208, 800, 208, 800, 200, 208, 256, 240, 416, 800, 208};
    const int NL = (int)(sizeof(grids) / sizeof(int));
    int* out; CK(hipMalloc(&out, NL * 16 * 4)); CK(hipMemset(out, 0xff, NL * 16 * 4));
    hipGraph_t g; hipGraphExec_t gx;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(where, dim3(grids[l]), dim3(256), 0, s, out, l);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
    std::vector<int> h(NL * 16);
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipGraphLaunch(gx, s)); CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
      printf("A. replay %d: XCC_ID of workgroups 0..15 per launch (grid size in front)\n", rep);
      for (int l = 0; l < NL; ++l) {
        printf("   grid %4d:", grids[l]);
        for (int w = 0; w < 16; ++w) printf(" %d", h[l * 16 + w]);
        printf("\n");
      }
    }
    CK(hipFree(out));
  }
  // ---- B / C / D
  PArgs a;
  CK(hipMalloc(&a.ticket, 8 * 32 * 4)); CK(hipMalloc(&a.cnt, 8 * 32 * 4)); CK(hipMalloc(&a.err, 8 * 4));
  CK(hipMalloc(&a.slots, (size_t)256 * 1024 * 16)); CK(hipMalloc(&a.big, (size_t)512 * (256 << 10))); CK(hipMalloc(&a.times, 256 * 8 * 8));
  CK(hipMemset(a.big, 1, (size_t)512 * (256 << 10)));
  const int lds = 100 << 10;     // one workgroup per CU
  CK(hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  auto run = [&](int mode, int scope_wg, int iters, const char* what) {
    a.mode = mode; a.scope_wg = scope_wg; a.iters = iters;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(a.ticket, 0, 8 * 32 * 4)); CK(hipMemset(a.cnt, 0, 8 * 32 * 4)); CK(hipMemset(a.err, 0, 8 * 4)); CK(hipMemset(a.times, 0, 256 * 8 * 8));
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(persistent, dim3(256), dim3(512), lds, s, a);
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned err[8]; unsigned tick[8 * 32]; u64 tm[256 * 8];
      CK(hipMemcpy(err, a.err, sizeof(err), hipMemcpyDeviceToHost)); CK(hipMemcpy(tick, a.ticket, sizeof(tick), hipMemcpyDeviceToHost));
      CK(hipMemcpy(tm, a.times, sizeof(tm), hipMemcpyDeviceToHost));
      double avg = 0, mx = 0; int n = 0;
      double dd[5] = {0, 0, 0, 0, 0};
      for (int w = 0; w < 256; ++w) if (tm[w * 8]) { const double us = tm[w * 8] / 100.0; avg += us; if (us > mx) mx = us; ++n; for (int k = 0; k < 5; ++k) dd[k] += tm[w * 8 + 2 + k] / 100.0; }
      printf("%s rep %d: kernel %.1f us by events; in-kernel avg %.2f max %.2f us over %d workgroups", what, rep, ms * 1e3, n ? avg / n : 0., mx, n);
      if (mode <= 2) printf(" => %.3f us per %s", (n ? avg / n : 0.) / iters / (mode ? 2 : 1), mode ? "barrier (two per hand-off, incl. 16 KB store + check)" : "barrier");
      if (mode == 3 && n) printf("\n      256 KB by LDS-DMA per workgroup, us (GB/s per CU): stored by the XCD %.2f (%.0f) | again %.2f (%.0f) | again, sc1 %.2f (%.0f) | untouched %.2f (%.0f) | again %.2f (%.0f)",
                                 dd[0] / n, 262.144 / (dd[0] / n), dd[1] / n, 262.144 / (dd[1] / n), dd[2] / n, 262.144 / (dd[2] / n), dd[3] / n, 262.144 / (dd[3] / n), dd[4] / n, 262.144 / (dd[4] / n));
      printf("\n      XCD population:");
      for (int x = 0; x < 8; ++x) printf(" %u", tick[x * 32]);
      printf(" | stale words plain %u, sc1 %u | spin time-outs %u | over-populated %u\n", err[0], err[1], err[2], err[3]);
    }
  };
  run(0, 0, 1000, "B. XCD-local barrier, agent-scope add");
  run(0, 1, 1000, "B. XCD-local barrier, workgroup-scope add");
  run(1, 0, 200, "C. 16 KB hand-off inside the XCD, PLAIN loads");
  run(2, 0, 200, "C. 16 KB hand-off inside the XCD, sc1 loads");
  run(2, 1, 200, "C. 16 KB hand-off inside the XCD, sc1 loads, workgroup-scope add");
  run(3, 0, 1, "D. LDS-DMA streaming");
  return 0;
}
