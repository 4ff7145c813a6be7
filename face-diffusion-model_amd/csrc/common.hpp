// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the FDM sampling path.
// Wavefront = 64 lanes; MFMA fragments are expressed as 16-byte lane chunks so that the bf16
// (v_mfma_f32_16x16x32_bf16) and exact-fp32 (v_mfma_f32_16x16x4_f32) paths share one kernel body.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdm {

typedef __bf16 bf16;
typedef _Float16 f16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2, ACT_GELU_ERF = 3, ACT_GELU_TANH = 4, ACT_LEAKY02 = 5 };

// One MFMA "k-step" on a 16-byte-per-lane fragment pair.
//  bf16 : 8 elements per lane  -> one 16x16x32 MFMA (lane group g = lane>>4 holds k = 8g..8g+7)
//  fp32 : 4 elements per lane  -> four 16x16x4 MFMAs; MFMA j takes element j of both fragments, so
//         lane group g contributes k = 4g + j; both operands use the same permutation, the sum is exact.
template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<f16> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    // (bit-cast the whole vector: __builtin_bit_cast(float, a[j]) on a vector element folds to element 0)
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
  }
};

// One k-step of a 16-bit fragment pair whose element type is E (bf16 or fp16): same lane map for both.
template <typename E> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
  if constexpr (std::is_same<E, f16>::value)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// Operand kinds.  Besides plain fp32 and bf16 matrices the GEMMs take SPLIT operands: x = hi + lo / SCALE with hi, lo
// stored as two planes of a 16-bit type (plane 1 starts `lo_off` elements after plane 0), and the product evaluated as
// hi.hi + (hi.lo + lo.hi) / SCALE on the 16-bit MFMA (three passes; the lo.lo term is below the representation error).
//   f16x3_t : fp16 planes, 11 significant bits each -> 22 bits per operand, fp32-class results (SCALE = 2^11 keeps the
//             residual plane in fp16's normal range; the two small products are summed in their own accumulator)
// E = element type in memory, NP = planes, KV = operand kind of the packed K / V outputs of a QKV projection in that mode
// (f16x3 feeds the split attention kernel plane pairs).
// ---------------------------------------------------------------------------------------------------
struct f16x3_t {};
// one-plane 16-bit kinds (bf16, and fp16 = the split kind's hi plane alone): the throughput modes -- hardware exp / log forms in
// their activations and softmax, far inside the rounding of the 8- / 11-bit operand they feed
template <typename T> struct is_fast16 { static constexpr bool value = std::is_same<T, bf16>::value || std::is_same<T, f16>::value; };
template <typename T> struct Opnd { using E = T; using KV = T; static constexpr int NP = 1; static constexpr float SCALE = 1.f; };
template <> struct Opnd<f16x3_t> { using E = f16; using KV = f16x3_t; static constexpr int NP = 2; static constexpr float SCALE = 2048.f; };

// Result stores are agent-scope write-through (`sc1`): the line goes to the memory side as it is written instead of sitting dirty in
// the storing XCD's L2 until the end-of-kernel release writes it back.  Every launch of the step program is followed by a launch that
// reads its output from other XCDs, so that write-back is on the launch-to-launch path: cfg2 +2.0 % in both modes, cfg3 +4.5...5 %, cfg4
// +1.4...2.7 %, single clips +0.6...2.7 %, cfg5 (1992 rows) -0.2...-0.7 %, HuBERT 0...+1.7 % (same box, alternating;
// profiles/r5_store_policy/).  `sc0 sc1` measures the same; `nt` (with or without the scope bits) is 13 % slower.  Cache policy only:
// the bytes stored are the same.
// FDM_PLAIN_STORES (`make plainstores`): the same stores written in C++ -- the compiler's own instruction selection and hazard handling.
// tests/test_store_policy_gpu.py runs the whole path on both builds and compares every output bit for bit: the hand-kept wait states
// below are correct exactly as long as that test is green on the toolchain in use.
__device__ __forceinline__ void st16(void* p, const f32x4& v) {
#ifdef FDM_PLAIN_STORES
  *(f32x4*)p = v;
#else
  // (a VMEM store of more than 64 bits needs 2 wait states before a VALU write to its data registers on gfx940+; the compiler's
  //  hazard recogniser does not see into the asm, so the wait states travel with the store)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
template <typename V8> __device__ __forceinline__ void st8(void* p, const V8& v) {
  static_assert(sizeof(V8) == 8, "st8 stores 8 bytes");
#ifdef FDM_PLAIN_STORES
  *(V8*)p = v;
#else
  typedef __attribute__((ext_vector_type(2))) int i2;
  const i2 w = __builtin_bit_cast(i2, v);
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
#endif
}
// fp16 overflows at 65504: operand copies are clamped (one v_med3_f32; activations / weights on this path are O(1..100), the clamp keeps a stray
// value -- or a NaN, which med3 maps to the lower bound -- finite)
__device__ __forceinline__ float clamp_f16(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
// Store 4 consecutive fp32 values v as operand kind T at dst (split kinds: hi plane at dst, lo plane at dst + lo_off).
template <typename T> __device__ __forceinline__ void store_opnd4(void* dst, long long lo_off, const f32x4& v) {
  using E = typename Opnd<T>::E;
  if constexpr (std::is_same<T, float>::value) {
    st16(dst, v);
  } else if constexpr (Opnd<T>::NP == 1) {
    typedef __attribute__((ext_vector_type(4))) E e4;
    e4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (E)(std::is_same<E, f16>::value ? clamp_f16(v[j]) : v[j]);      // (fp16 overflows at 65504)
    st8(dst, o);
  } else {
    typedef __attribute__((ext_vector_type(4))) E e4;
    e4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // fp16 overflows at 65504: clamp (activations / weights on this path are O(1..100); a clamp keeps a stray value finite)
      const float x = std::is_same<E, f16>::value ? clamp_f16(v[j]) : v[j];
      h[j] = (E)x;
      l[j] = (E)((x - (float)h[j]) * Opnd<T>::SCALE);
    }
    st8(dst, h);
    st8((E*)dst + lo_off, l);
  }
}
template <typename T> __device__ __forceinline__ void store_opnd1(void* dst, long long lo_off, float v) {
  using E = typename Opnd<T>::E;
  if constexpr (Opnd<T>::NP == 1) {
    *(E*)dst = (E)(std::is_same<E, f16>::value ? clamp_f16(v) : v);
  } else {
    const float x = std::is_same<E, f16>::value ? clamp_f16(v) : v;
    const E h = (E)x;
    *(E*)dst = h;
    *((E*)dst + lo_off) = (E)((x - (float)h) * Opnd<T>::SCALE);
  }
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_MISH: {  // x * tanh(softplus(x)), softplus threshold 20 as torch
      float sp = v > 20.f ? v : log1pf(expf(v));
      return v * tanhf(sp);
    }
    case ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case ACT_GELU_TANH: {  // models/utils/base_model_util.py:81-94
      float c = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return v * (0.5f * (1.f + tanhf(c)));
    }
    case ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
    default: return v;
  }
}

// GELU(erf) for the bf16 (throughput) kind: erf by Abramowitz-Stegun 7.1.26 (ABSOLUTE error <= 1.5e-7 on erf: far inside the 2^-9
// rounding of the bf16 value it feeds wherever |GELU| > 1e-4; on the negative tail, v < -4, where |GELU(v)| itself is below 1e-4, the
// relative error is of the order of bf16's own rounding -- absolute 1e-7 there) on the hardware exp -- a dozen instructions instead of libm erff's ~50 per element (HuBERT's FFN1 epilogue,
// the conv LayerNorm + GELU kernels).  The fp32 and split (parity) kinds keep erff.
__device__ __forceinline__ float gelu_erf_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = 1.f / (1.f + 0.3275911f * x);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = 1.f - poly * __expf(-x * x);              // erf(|v| / sqrt 2)
  return 0.5f * v * (1.f + (v < 0.f ? -e : e));
}
// activation in the arithmetic of operand kind T: the bf16 kind takes the fast GELU above, every other kind act_apply
template <typename T> __device__ __forceinline__ float act_apply_t(float v, int act) {
  if constexpr (is_fast16<T>::value) { if (act == ACT_GELU_ERF) return gelu_erf_fast(v); }
  return act_apply(v, act);
}

// Reductions across the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48).  (gfx950's v_permlane16_swap /
// v_permlane32_swap would do this in VALU, but neither the builtin nor inline asm gave the documented pair of results
// in a probe on this toolchain, so these stay on the ds_bpermute path.)
__device__ __forceinline__ float rows_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float rows_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Sum over the 64 lanes, result in every lane.  Four DPP adds (quad swaps, half-row and row mirrors: VALU, a few cycles
// each) give every lane its 16-lane row sum; the four row sums are combined through v_readlane.  The butterfly of
// __shfl_xor it replaces is six dependent ds_bpermute round trips through the LDS crossbar (~0.25 us per reduction,
// and the fused LN1+LN2 kernel chains four of them).
__device__ __forceinline__ float wave_sum(float v) {
  auto dpp_add = [](float x, auto ctrl) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});     // row_mirror
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}


// Element offsets inside one (clip, head) block of the fragment-packed K / V buffers (include/fdm_hip.h, fdm_attn_args).
template <typename T> __device__ __forceinline__ size_t kp_offset(int l, int e, int hd) {
  constexpr int EPC = 16 / (int)sizeof(T), KT = 4 * EPC, NSUB = KT / 16;
  const int NKS = hd / (4 * EPC);
  const int kt = l / KT, w = l % KT;
  const int s = (NSUB == 2) ? ((w >> 2) & 1) : 0;
  const int r = (NSUB == 2) ? (((w >> 3) << 2) | (w & 3)) : w;
  const int ch = e / EPC;
  return ((size_t)((((kt * NSUB + s) * NKS + (ch >> 2)) * 4 + (ch & 3)) * 16 + r)) * EPC + (e % EPC);
}
template <typename T> __device__ __forceinline__ size_t vp_offset(int l, int e, int hd) {
  constexpr int EPC = 16 / (int)sizeof(T), KT = 4 * EPC;
  const int kt = l / KT, w = l % KT;
  return ((size_t)(((kt * (hd >> 4) + (e >> 4)) * 4 + w / EPC) * 16 + (e & 15))) * EPC + (w % EPC);
}

}  // namespace fdm
