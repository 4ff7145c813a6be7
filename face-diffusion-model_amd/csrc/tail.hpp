// The row-local tail of a decoder layer as ONE launch (round 4):
//     out-proj GEMM (+bias, +residual)  ->  LayerNorm1 + LayerNorm2 (+folded cross-attention addends)  ->  FFN1 GEMM (+bias, ReLU)
//     ->  FFN2 GEMM (+bias, +residual)  ->  LayerNorm3                    (models/fdm_vocaset.py:45-46,87 after the self-attention)
// Nothing in this sequence mixes rows, so the rows are cut into 8 contiguous blocks, one per XCD, and every phase of a block runs
// on the 32 CUs of its XCD.  The phases are separated by XCD-LOCAL barriers -- one counter per XCD, plain stores + s_waitcnt
// vmcnt(0) before arriving, an L1-bypassing poll, no fence: every byte a phase reads from an earlier phase was stored by a CU of
// the same XCD and is served by that XCD's L2 (tools/xcd_probe.cpp: 0.79 us per barrier, no stale word in a 16 KB hand-off test;
// loads of such bytes carry sc1 / nt so that a line the CU's L1 kept from an earlier step is never used).  That replaces four of the
// layer's seven kernel boundaries (2.0-2.5 us each in situ, plus cold kernel arguments and a cold first tile) by barriers.
//
// One workgroup of 512 threads per CU, 256 workgroups, all resident (the launch follows its predecessor in stream order and the
// ring's LDS footprint admits one workgroup per CU).  Which XCD a workgroup runs on is read from the hardware (HW_REG_XCC_ID); its
// index among the XCD's workgroups is a ticket.  The ticket and barrier words only ever grow: every launch adds exactly 32 tickets
// and 32 arrivals per barrier and XCD, so a workgroup derives its targets from the value its first add returned -- nothing is
// re-zeroed between launches.  Spins are bounded: a workgroup that times out (an XCD that did not receive its 32 workgroups) raises
// the error word and the host falls back to the per-operator chain for good.
//
// Every phase body is the operator's own (gemm_glds_tile, ln_row_body): same k order, same reduction order -> the same bits as
// the five separate launches.
#pragma once
#include "gemm.hpp"
#include "layernorm.hpp"

namespace fdm {

__device__ __forceinline__ int tail_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & (FDM_XCD - 1);
}

// arrive on the XCD's counter, wait until `target` arrivals (wrap-safe), every wave's stores already in L2
__device__ __forceinline__ bool tail_barrier(unsigned* c, unsigned target, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int ok = 1;
    while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      if (++spins > (1u << 21)) { ok = 0; atomicAdd(err, 1u); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

template <typename T, int NV>
__device__ __forceinline__ void tail_ln_phase(const fdm_ln_args& a, int r0, int r1, int idx, bool coh, float (*red)[4][NV]) {
  // groups of NV waves take rows idx * G + grp, + 32 * G, ...; every group runs the same number of passes (workgroup barriers inside)
  constexpr int G = 8 / NV;
  const int grp = threadIdx.x / (64 * NV), tid = threadIdx.x % (64 * NV);
  const int per_pass = 32 * G;
  const int passes = (r1 - r0 + per_pass - 1) / per_pass;
  for (int ps = 0; ps < passes; ++ps) {
    const int row = r0 + ps * per_pass + idx * G + grp;
    const bool live = row < r1;
    const int rr = live ? row : r1 - 1;
    if (coh) ln_row_body<T, NV, false, true>(a, rr, tid, red[grp], live);
    else ln_row_body<T, NV, false, false>(a, rr, tid, red[grp], live);
  }
}

template <typename T, int BM, int BN, int WM, int WN, int NST, bool COH>
__device__ __forceinline__ void tail_gemm_phase(const fdm_gemm_args& a, int r0, int r1, int idx) {
  const int nct = a.N / BN, nrt = (r1 - r0 + BM - 1) / BM;
  for (int i = idx; i < nrt * nct; i += 32) {
    const int rt = i / nct, ct = i - rt * nct;
    gemm_glds_tile<T, BM, BN, WM, WN, NST, 8, false, false, GEMM_LEAN, COH>(a, r0 + rt * BM, ct * BN, 0, r1, false);
    __syncthreads();            // the ring is reused by this workgroup's next tile / the next phase's first DMA
  }
}

// LDS: the largest ring of the three GEMM phases (64x128, 3 stages in the split kind, 4 otherwise) + the LayerNorm slots
template <typename T> constexpr int tail_ffn1_stages() { return Opnd<T>::NP == 2 ? 3 : 4; }
template <typename T> constexpr int tail_lds_bytes() {
  return tail_ffn1_stages<T>() * Opnd<T>::NP * (64 + 128) * 128 + gemm_ln_scratch_bytes<64, 128>();
}

template <typename T, int NV>
__global__ __launch_bounds__(512) void layer_tail_kernel(const fdm_tail_args p) {
  __shared__ int s_idx;
  __shared__ unsigned s_base;
  __shared__ float red[8 / NV][4][NV];
  const unsigned long long t_entry = p.stamps ? wall_clock64() : 0ull;
  const int x = tail_xcc_id();
  unsigned* ticket = p.sync + x * 32;
  unsigned* bar = p.sync + (FDM_XCD + x) * 32;
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_idx = (int)(old & 31u);
    // this launch's barriers count arrivals [base, base + 4 * 32): every earlier launch added exactly 4 * 32 per XCD
    s_base = (old >> 5) * (4u * 32u);
  }
  __syncthreads();
  const int idx = s_idx;
  const unsigned base = s_base;
  unsigned long long* st = (p.stamps && threadIdx.x == 0) ? p.stamps + (size_t)(x * 32 + idx) * 12 : nullptr;
  int sn = 0;
  auto stamp = [&]() { if (st) st[sn++] = wall_clock64(); };
  if (st) st[sn++] = t_entry;
  stamp();
  // row block of this XCD: cuts on multiples of 8 rows
  const int R = p.rows;
  const int r0 = (int)(((long long)R * x / FDM_XCD) / 8 * 8), r1 = x == FDM_XCD - 1 ? R : (int)(((long long)R * (x + 1) / FDM_XCD) / 8 * 8);
  // 1. x1 = ctx Wo^T + bo + h          (ctx and h come from earlier launches: default cache policy)
  tail_gemm_phase<T, 64, 64, 2, 4, 4, false>(p.out_proj, r0, r1, idx);
  stamp();
  if (!tail_barrier(bar, base + 32u, p.err)) return;
  stamp();
  // 2. h2 = LN2(LN1(x1) + C1 + TT[t])
  tail_ln_phase<T, NV>(p.ln12, r0, r1, idx, true, red);
  stamp();
  if (!tail_barrier(bar, base + 64u, p.err)) return;
  stamp();
  // 3. u = relu(h2 W1^T + b1)
  tail_gemm_phase<T, 64, 128, 2, 4, tail_ffn1_stages<T>(), true>(p.ffn1, r0, r1, idx);
  stamp();
  if (!tail_barrier(bar, base + 96u, p.err)) return;
  stamp();
  // 4. x1 = u W2^T + b2 + h2
  tail_gemm_phase<T, 64, 64, 2, 4, 4, true>(p.ffn2, r0, r1, idx);
  stamp();
  if (!tail_barrier(bar, base + 128u, p.err)) return;
  stamp();
  // 5. h = LN3(x1)
  tail_ln_phase<T, NV>(p.ln3, r0, r1, idx, true, red);
  if (st) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[sn++] = wall_clock64(); }
}

template <typename T>
static hipError_t tail_launch_t(const fdm_tail_args& a, hipStream_t s) {
  constexpr int lds = tail_lds_bytes<T>();
  static_assert(lds <= 160 * 1024 && lds > 80 * 1024, "one workgroup per CU");
  const int d = a.ln12.d;
  if (d == 1024) {
    static bool once = [] { return hipFuncSetAttribute((const void*)layer_tail_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }();
    (void)once;
    hipLaunchKernelGGL((layer_tail_kernel<T, 4>), dim3(FDM_XCD * 32), dim3(512), lds, s, a);
  } else if (d == 512) {
    static bool once = [] { return hipFuncSetAttribute((const void*)layer_tail_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }();
    (void)once;
    hipLaunchKernelGGL((layer_tail_kernel<T, 2>), dim3(FDM_XCD * 32), dim3(512), lds, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace fdm
