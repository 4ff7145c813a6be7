// LayerNorm of one row (fdm_op_layernorm's arithmetic), shared by the operator's kernel (elementwise.hpp) and the fused layer-tail
// launch (tail.hpp): the same reduction order in both, so the fused launch reproduces the operator's bits.
#pragma once
#include "common.hpp"
#include "../../include/fdm_hip.h"

namespace fdm {

// One row by NV waves: `tid` = thread index inside the row's group (0 .. 64 * NV - 1), `red` = the group's [4][NV] reduction slots,
// `store` = false for a padding pass (a group with no row left still takes part in the workgroup barriers of block_sum).
// COH (fused launches, csrc/tail.hpp): x was written earlier in the same launch by other CUs of this XCD -> loaded past L1.
template <typename T, int NV, bool HEAVY, bool COH = false>
__device__ __forceinline__ void ln_row_body(const fdm_ln_args& p, const int row, const int tid, float (*red)[NV], const bool store = true) {
  constexpr int d = 256 * NV;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = tid * 4;
  const bool two = p.gamma2 != nullptr;
  // every load of the kernel is requested before the first value is used: the row, the matrix addend and the affine vectors
  // go out at once, the table row one dependent scalar load (the device-side step word) later -- one memory latency in
  // all instead of one per operand (the kernel is launch-to-launch latency, not bandwidth)
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 v;
  if constexpr (COH) v = __builtin_nontemporal_load((const f32x4*)(p.x + (size_t)row * d + col));
  else v = *(const f32x4*)(p.x + (size_t)row * d + col);
  const bool has_e = p.add_mat || p.add_tab;
  int arow = row;                      // (uniform: scalar arithmetic) conditions of a clip share the clip's addend rows
  if (p.add_mat_group > 0) {
    const int m = p.add_mat_wrap > 0 ? row % p.add_mat_wrap : row;
    arow = (m / p.add_mat_group) * p.add_mat_L + m % p.add_mat_L;
  }
  const f32x4 em = p.add_mat ? *(const f32x4*)(p.add_mat + (size_t)arow * d + col) : zero;
  const f32x4 g1 = *(const f32x4*)(p.gamma + col), b1 = *(const f32x4*)(p.beta + col);
  const float *gp2 = two ? p.gamma2 : p.gamma, *bp2 = two ? p.beta2 : p.beta;      // (a select of pointers, not of loaded data)
  const f32x4 g2 = *(const f32x4*)(gp2 + col), b2 = *(const f32x4*)(bp2 + col);
  f32x4 et = zero;
  if (p.add_tab) {
    const int k = p.tab_step ? *p.tab_step : 0;
    const int idx = p.tab_index ? p.tab_index[k] : k;
    et = *(const f32x4*)(p.add_tab + (size_t)idx * d + col);
  }
  const f32x4 e = p.add_tab ? em + et : em;
  auto block_sum = [&](float x, int slot) {
    x = wave_sum(x);
    if (lane == 0) red[slot][wave] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NV; ++w) t += red[slot][w];
    return t;
  };
  if (!two && has_e) v += e;
  float mean = block_sum((v[0] + v[1]) + (v[2] + v[3]), 0) * (1.f / d);
  v -= mean;
  float var = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), 1) * (1.f / d);
  float rstd = 1.f / sqrtf(var + p.eps);
  if (two) {       // h = LN1(x); stage 2 input = h + add_mat + add_tab[idx]
    v = v * rstd * g1 + b1;
    if (has_e) v += e;
    mean = block_sum((v[0] + v[1]) + (v[2] + v[3]), 2) * (1.f / d);
    v -= mean;
    var = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), 3) * (1.f / d);
    rstd = 1.f / sqrtf(var + p.eps);
  }
  f32x4 y = v * rstd * g2 + b2;
  if constexpr (HEAVY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = act_apply(y[j], p.act);
  } else if (p.act == ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
  }
  if (!store) return;
  if (p.y_f32) *(f32x4*)(p.y_f32 + (size_t)row * d + col) = y;
  if (p.y_t) store_opnd4<T>((typename Opnd<T>::E*)p.y_t + (size_t)row * d + col, p.y_t_lo_off, y);
}

}  // namespace fdm
