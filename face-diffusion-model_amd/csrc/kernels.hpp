// Launchers of the heavy kernel families, one translation unit per operand kind (gemm_*.hip, attn_*.hip) so that the
// library builds in parallel; fdm_hip.hip (the C ABI + the bandwidth kernels) calls them through these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fdm_hip.h"

namespace fdm {
int fail(int code, const char* fmt, ...);      // fdm_hip.hip: sets the thread's fdm_last_error() text, returns code
hipError_t gemm_launch_f32(const fdm_gemm_args& a, hipStream_t s);
hipError_t gemm_launch_bf16(const fdm_gemm_args& a, hipStream_t s);
hipError_t gemm_launch_f16x3(const fdm_gemm_args& a, hipStream_t s);
hipError_t gemm_launch_f16(const fdm_gemm_args& a, hipStream_t s);
int gemm_heuristic_tile_of(const fdm_gemm_args& a);      // gemm_bf16.hip: the tile a launch with tile = 0 resolves to
hipError_t attn_launch_f32(const fdm_attn_args& a, hipStream_t s);
hipError_t attn_launch_bf16(const fdm_attn_args& a, hipStream_t s);
hipError_t attn_launch_f16x3(const fdm_attn_args& a, hipStream_t s);
hipError_t attn_launch_f16(const fdm_attn_args& a, hipStream_t s);
hipError_t pack_kv_launch_f32(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp, int B, int H, int L, int Lpad, int hd, hipStream_t s);
hipError_t pack_kv_launch_bf16(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp, int B, int H, int L, int Lpad, int hd, hipStream_t s);

inline hipError_t gemm_launch(const fdm_gemm_args& a, hipStream_t s) {
  switch (a.dtype) {
    case FDM_BF16: return gemm_launch_bf16(a, s);
    case FDM_F16X3: return gemm_launch_f16x3(a, s);
    case FDM_F16: return gemm_launch_f16(a, s);
    default: return gemm_launch_f32(a, s);
  }
}
inline hipError_t attn_launch(const fdm_attn_args& a, hipStream_t s) {
  return a.dtype == FDM_BF16 ? attn_launch_bf16(a, s) : (a.dtype == FDM_F16 ? attn_launch_f16(a, s) : (a.dtype == FDM_F16X3 ? attn_launch_f16x3(a, s) : attn_launch_f32(a, s)));
}
}  // namespace fdm
