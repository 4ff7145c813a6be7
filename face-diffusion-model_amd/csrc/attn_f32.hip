// Fused attention kernels for f32 operands (see attention.hpp).
#include "attention.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t attn_launch_f32(const fdm_attn_args& a, hipStream_t s) { return attn_launch_dtype<float>(a, s); }
hipError_t pack_kv_launch_f32(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp, int B, int H, int L, int Lpad, int hd, hipStream_t s) {
  return pack_kv_launch<float>(K, ldk, V, ldv, Kp, Vp, B, H, L, Lpad, hd, s);
}
}
