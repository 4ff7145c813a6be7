// Round 6, item 1(a): what is the ~1.3 us by which a step GEMM's launch-to-launch boundary (previous kernel's last store acknowledged ->
// first instruction of the first wave: 2.0-2.4 us at 208 workgroups) exceeds that of a trivial kernel of the same grid (0.9-1.4 us,
// tools/boundary_probe.cpp) -- the START of the GEMM (cold code / descriptor / argument preload / resources) or the END of the kernel in
// front of it (what the GEMM leaves for the end-of-kernel processing), and does it follow the workgroup count or the coldness of L2?
// Chains of launches replayed as one hipGraph; every workgroup of every launch stamps the 100 MHz clock at its first instruction and
// after its stores are acknowledged.  Launch kinds:
//   G  the production out-proj GEMM (gemm_glds_kernel<bf16,64,64,2,4,4,8,false,false,LEAN,LW=4>, stamped build), M x 1024 x 1024,
//      distinct weights per launch (g: the SAME weights in every launch -- its working set stays in the L2s)
//   T  a trivial kernel of the same grid, 768 threads, 66 KB of LDS, the same argument block (t: 256 threads, no LDS;
//      V: T with 128 VGPRs declared; P: T with the GEMM's 13 preloaded argument SGPRs)
//   L  the step's fused LN1+LN2 launch (ln_row_kernel<bf16,4>: fp32 row + matrix addend + table row in, fp32 row + bf16 copy out), one
//      workgroup per row, stamped: [1] operands arrived [2] four block reductions done [3] stores issued [4] acknowledged
//   X  a kernel that streams 128 MB through the L2s (evicts code, descriptors, arguments, tables)
// Every adjacent pair (a, b) of a chain gives one sample of boundary(a -> b) = first instruction of b - last acknowledgement of a.
// Part (b) -- the same GEMM chain with the exiting loader waves touching the next launch's first weight k-tiles / whole weight matrix /
// first instruction lines -- measured null (profiles/r6_coldstart/probe_touch.txt) and is kept as profiles/r6_coldstart/tail_touch_experiment.patch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=13 -DFDM_GEMM_STAMPS \
//         -o tools/_build/coldstart_probe tools/coldstart_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../face-diffusion-model_amd/csrc/gemm.hpp"
#include "../face-diffusion-model_amd/csrc/elementwise.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAXWG = 1600, SW = 8;      // stamp words per workgroup: [0] entry [1] first k-tile [2] loop end [3] stores issued [4] acknowledged [5] first instruction

struct TArgs { unsigned long long* stamps; char pad[sizeof(fdm_gemm_args) - 8]; };      // the size of fdm_gemm_args

template <int THREADS, int VG>
__global__ __launch_bounds__(THREADS) void triv_kernel(const TArgs a) {
  extern __shared__ char smem[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1;
  if constexpr (VG > 0) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SW;
    const unsigned long long t1 = wall_clock64();
    st[5] = t0; st[0] = t0; st[4] = t1;
  }
}
// the GEMM's signature: nine leading scalars (13 SGPRs, preloaded at wave launch) and the argument struct
__global__ __launch_bounds__(768) void triv_preload_kernel(const void* pA, const void* pW, long long o1, long long o2, int pM, int pN, int pK, int l1, int l2, const TArgs a) {
  extern __shared__ char smem[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1;
  if (pM == 12345 && pA == pW && o1 == o2 && pN == pK && l1 == l2) smem[1] = 2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SW;
    const unsigned long long t1 = wall_clock64();
    st[5] = t0; st[0] = t0; st[4] = t1;
  }
}
struct XArgs { unsigned long long* stamps; const float4* in; float4* out; int f4_per_thread; };
__global__ __launch_bounds__(256) void thrash_kernel(const XArgs a) {
  const unsigned long long t0 = wall_clock64();
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < a.f4_per_thread; ++i) {
    const float4 v = a.in[((size_t)blockIdx.x * a.f4_per_thread + i) * 256 + threadIdx.x];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 12345.f) a.out[0] = acc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stamps + (size_t)blockIdx.x * SW;
    const unsigned long long t1 = wall_clock64();
    st[5] = t0; st[0] = t0; st[4] = t1;
  }
}

struct Ctx {
  hipStream_t s;
  float *lx[2], *ladd, *ltab, *lgam; int* lstep;
  void* W[12]; void* buf[2]; float* bias; unsigned long long* stamps; float4* big; float4* sink; unsigned long long* code_slot;
  int NL;
};

static int nwg_of(char kind, int M) { return kind == 'X' ? 1024 : (kind == 'L' ? M : ((M + 63) / 64) * 16); }

static void launch(Ctx& c, const std::string& pattern, int l, int M, int touch) {
  const char kind = pattern[l];
  const int NLp = (int)pattern.size();
  int ln = (l + 1) % NLp;                       // the next GEMM of the chain (the graph is replayed back to back: it wraps)
  for (int i = 0; i < NLp && pattern[ln] != 'G' && pattern[ln] != 'g'; ++i) ln = (ln + 1) % NLp;
  unsigned long long* st = c.stamps + (size_t)l * MAXWG * SW;
  const dim3 grid(16, (M + 63) / 64);
  if (kind == 'G' || kind == 'g') {
    fdm_gemm_args a;
    memset(&a, 0, sizeof(a));
    const int N = 1024, K = 1024;
    a.A = c.buf[l % 2]; a.lda = K; a.W = c.W[kind == 'g' ? 0 : l % 12]; a.ldw = K; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = FDM_BF16;
    a.bias = c.bias; a.out_t = c.buf[(l + 1) % 2]; a.ldo_t = N; a.ldr = N; a.ldo_f32 = N; a.ln_eps = 1e-5f;
    a.incr_table = (const int*)st;
    a.sched.advance = getenv("FDM_EPI_PASSES") ? atoi(getenv("FDM_EPI_PASSES")) : 0;      // 2: the stamped kernel runs its epilogue twice (second pass: warm instruction cache)
    (void)ln; (void)touch;
    hipError_t e = fdm::gemm_glds_launch_h<fdm::bf16, 64, 64, 2, 4, 4, 8, false, false, fdm::GEMM_LEAN, 4>(a, c.s);
    if (e != hipSuccess) { printf("gemm launch: %s\n", hipGetErrorString(e)); exit(1); }
  } else if (kind == 'L') {
    fdm_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = c.lx[l % 2]; a.M = M; a.d = 1024; a.add_mat = c.ladd; a.add_tab = c.ltab; a.tab_step = c.lstep; a.gamma = c.lgam; a.beta = c.lgam + 1024;
    a.gamma2 = c.lgam + 2048; a.beta2 = c.lgam + 3072; a.eps = 1e-5f; a.y_f32 = c.lx[(l + 1) % 2]; a.y_t = c.buf[(l + 1) % 2]; a.dtype = FDM_BF16;
    a.x_plane_stride = (long long)st;        // (instrumented build: the stamp buffer rides here when x_planes == 0)
    hipError_t e = fdm::ln_launch_t<fdm::bf16>(a, c.s);
    if (e != hipSuccess) { printf("ln launch: %s\n", hipGetErrorString(e)); exit(1); }
  } else if (kind == 'X') {
    XArgs a{st, c.big, c.sink, (128 << 20) / 16 / (1024 * 256)};
    hipLaunchKernelGGL(thrash_kernel, dim3(1024), dim3(256), 0, c.s, a);
  } else {
    TArgs a{};
    a.stamps = st;
    const int lds = 66 * 1024;
    if (kind == 'T') hipLaunchKernelGGL((triv_kernel<768, 0>), grid, dim3(768), lds, c.s, a);
    else if (kind == 't') hipLaunchKernelGGL((triv_kernel<256, 0>), grid, dim3(256), 0, c.s, a);
    else if (kind == 'V') hipLaunchKernelGGL((triv_kernel<768, 128>), grid, dim3(768), lds, c.s, a);
    else if (kind == 'P') hipLaunchKernelGGL(triv_preload_kernel, grid, dim3(768), lds, c.s, (const void*)c.buf[0], (const void*)c.W[0], 0ll, 0ll, M, 1024, 1024, 1024, 1024, a);
    else { printf("unknown kind %c\n", kind); exit(1); }
  }
}

static void run_chain(Ctx& c, const char* name, const std::string& pattern, int M, int touch = 0) {
  const int NL = (int)pattern.size();
  hipGraph_t g; hipGraphExec_t x;
  fprintf(stderr, "-> %s %s M=%d touch %d\n", name, pattern.c_str(), M, touch);
  CK(hipMemsetAsync(c.stamps, 0, (size_t)NL * MAXWG * SW * 8, c.s));
  CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) launch(c, pattern, l, M, touch);
  CK(hipStreamEndCapture(c.s, &g));
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(x, c.s));
  CK(hipStreamSynchronize(c.s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0, c.s));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(x, c.s));
  CK(hipEventRecord(e1, c.s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)NL * MAXWG * SW);
  CK(hipMemcpy(h.data(), c.stamps, h.size() * 8, hipMemcpyDeviceToHost));
  auto S = [&](int l, int w, int i) { return (double)h[((size_t)l * MAXWG + w) * SW + i] * 0.01; };
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  struct Acc { double sum = 0; int n = 0; void add(double v) { sum += v; ++n; } double mean() const { return n ? sum / n : 0.0; } };
  std::map<std::string, Acc> bnd;
  std::map<char, Acc> span, kloop, first_tile, epi, ack, spread, entry, epi2, ack2;
  for (int l = 2; l < NL; ++l) {
    const char a = pattern[l - 1], b = pattern[l];
    const int na = nwg_of(a, M), nb = nwg_of(b, M);
    double prev_end = 0, first = 1e30, last = 0, efirst = 1e30, elast = 0;
    for (int w = 0; w < na; ++w) prev_end = std::max(prev_end, S(l - 1, w, 4));
    std::vector<double> t10, t21, t32, t43, ent;
    for (int w = 0; w < nb; ++w) {
      first = std::min(first, S(l, w, 5)); last = std::max(last, S(l, w, 4));
      efirst = std::min(efirst, S(l, w, 0)); elast = std::max(elast, S(l, w, 0));
      ent.push_back(S(l, w, 0) - S(l, w, 5));
      if (b == 'G' || b == 'g' || b == 'L') { t10.push_back(S(l, w, 1) - S(l, w, 0)); t21.push_back(S(l, w, 2) - S(l, w, 1)); t32.push_back(S(l, w, 3) - S(l, w, 2)); t43.push_back(S(l, w, 4) - S(l, w, 3)); }
    }
    bnd[std::string(1, a) + "->" + std::string(1, b)].add(first - prev_end);
    span[b].add(last - first); spread[b].add(elast - efirst); entry[b].add(med(ent));
    if (!t10.empty()) { first_tile[b].add(med(t10)); kloop[b].add(med(t21)); epi[b].add(med(t32)); ack[b].add(med(t43)); }
    if ((b == 'G' || b == 'g') && h[((size_t)l * MAXWG) * SW + 6]) {      // second epilogue pass stamped
      std::vector<double> e2, a2;
      for (int w = 0; w < nb; ++w) { e2.push_back(S(l, w, 6) - S(l, w, 4)); a2.push_back(S(l, w, 7) - S(l, w, 6)); }
      epi2[b].add(med(e2)); ack2[b].add(med(a2));
    }
  }
  printf("%-34s %-14s M=%4d (%3d WG) touch %d : %6.2f us per launch |", name, pattern.c_str(), M, nwg_of('G', M), touch, ms * 1e3 / reps / NL);
  for (auto& kv : bnd) printf("  %s %.2f", kv.first.c_str(), kv.second.mean());
  printf("  ||");
  for (auto& kv : span) {
    printf("  %c: span %.2f spread %.2f", kv.first, kv.second.mean(), spread[kv.first].mean());
    if (kloop.count(kv.first)) printf(" first->entry %.2f tile0 %.2f loop %.2f epilogue %.2f ack %.2f", entry[kv.first].mean(), first_tile[kv.first].mean(), kloop[kv.first].mean(), epi[kv.first].mean(), ack[kv.first].mean());
    if (epi2.count(kv.first)) printf(" | second epilogue pass (warm code): issued %.2f ack %.2f", epi2[kv.first].mean(), ack2[kv.first].mean());
  }
  printf("\n");
  fflush(stdout);
  CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
}

int main(int argc, char** argv) {
  Ctx c{};
  if (const char* pr = getenv("FDM_PROBE_PRIO")) {      // FDM_PROBE_PRIO=high|low: does the queue's priority move the dependent-launch boundary?
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // (numerically lower = higher priority)
    CK(hipStreamCreateWithPriority(&c.s, hipStreamNonBlocking, pr[0] == 'h' ? hi : lo));
    printf("stream priority %s (range %d..%d)\n", pr, lo, hi);
  } else {
    CK(hipStreamCreate(&c.s));
  }
  const int N = 1024, K = 1024, MMAX = 1600;
  std::vector<unsigned short> hw((size_t)N * K);
  srand(1);
  for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15) - 0x400);
  for (int i = 0; i < 12; ++i) { CK(hipMalloc(&c.W[i], (size_t)N * K * 2)); CK(hipMemcpy(c.W[i], hw.data(), (size_t)N * K * 2, hipMemcpyHostToDevice)); }
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&c.buf[i], (size_t)MMAX * K * 2)); CK(hipMemset(c.buf[i], 0, (size_t)MMAX * K * 2)); }
  CK(hipMalloc(&c.bias, N * 4)); CK(hipMemset(c.bias, 0, N * 4));
  CK(hipMalloc(&c.stamps, (size_t)24 * MAXWG * SW * 8));
  CK(hipMalloc(&c.big, (size_t)128 << 20)); CK(hipMemset(c.big, 0, (size_t)128 << 20));
  CK(hipMalloc(&c.sink, 4096));
  CK(hipMalloc(&c.code_slot, 64)); CK(hipMemset(c.code_slot, 0, 64));
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&c.lx[i], (size_t)MMAX * 1024 * 4)); CK(hipMemset(c.lx[i], 0, (size_t)MMAX * 1024 * 4)); }
  CK(hipMalloc(&c.ladd, (size_t)MMAX * 1024 * 4)); CK(hipMemset(c.ladd, 0, (size_t)MMAX * 1024 * 4));
  CK(hipMalloc(&c.ltab, (size_t)1000 * 1024 * 4)); CK(hipMemset(c.ltab, 0, (size_t)1000 * 1024 * 4));
  { std::vector<float> hg(4096, 1.f); CK(hipMalloc(&c.lgam, 4096 * 4)); CK(hipMemcpy(c.lgam, hg.data(), 4096 * 4, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&c.lstep, 64)); CK(hipMemset(c.lstep, 0, 64));
  CK(hipFuncSetAttribute((const void*)triv_kernel<768, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)triv_kernel<768, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)triv_preload_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const std::string G12(12, 'G'), g12(12, 'g');
  const std::string grp = argc > 1 ? argv[1] : "gemm";
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;
  for (int r = 0; r < rounds; ++r) {
    printf("== %s, round %d: boundary a->b = first instruction of b - last store acknowledgement of a (us)\n", grp.c_str(), r);
    if (grp == "gemm") {        // (1) the GEMM chain by grid size, distinct weights (the r2-r5 table on the final build) and with the weights kept in L2
      for (int M : {64, 256, 448, 800, 1600}) run_chain(c, "gemm chain", G12, M);
      for (int M : {64, 800}) run_chain(c, "gemm chain, same W", g12, M);
    } else if (grp == "triv") { // (2) trivial kernels of the same grids
      for (int M : {64, 800}) { run_chain(c, "trivial 256 thr", std::string(12, 't'), M); run_chain(c, "trivial 768 thr 66 KB", std::string(12, 'T'), M); }
      run_chain(c, "trivial + 128 VGPRs", std::string(12, 'V'), 800);
      run_chain(c, "trivial + argument preload", std::string(12, 'P'), 800);
    } else if (grp == "alt") {  // (3) who pays: the start of the GEMM or the end of the kernel in front of it
      for (int M : {64, 800}) { run_chain(c, "gemm / small trivial alternating", "GtGtGtGtGtGt", M); run_chain(c, "gemm / trivial alternating", "GTGTGTGTGTGT", M); }
      run_chain(c, "gemm / preload-trivial alternating", "GPGPGPGPGPGP", 800);
    } else if (grp == "thrash") {   // (4) coldness without the grid: an L2-thrashing launch in between
      for (int M : {64, 800}) {
        run_chain(c, "gemm / thrash alternating", "GXGXGXGXGXGX", M);
        run_chain(c, "gemm(same W) / thrash alternating", "gXgXgXgXgXgX", M);
        run_chain(c, "small trivial / thrash alternating", "tXtXtXtXtXtX", M);
      }
    } else if (grp == "ln") {       // the LayerNorm launch of the step: where its 5 us go
      for (int M : {64, 800}) { run_chain(c, "LayerNorm chain", std::string(12, 'L'), M); run_chain(c, "gemm / LayerNorm alternating", "GLGLGLGLGLGL", M); }
      run_chain(c, "LayerNorm / thrash alternating", "LXLXLXLXLXLX", 800);
    }
  }
  return 0;
}
