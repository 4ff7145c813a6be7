// Fused attention kernels for bf16 operands (see attention.hpp).
#include "attention.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t attn_launch_bf16(const fdm_attn_args& a, hipStream_t s) { return attn_launch_dtype<bf16>(a, s); }
hipError_t pack_kv_launch_bf16(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp, int B, int H, int L, int Lpad, int hd, hipStream_t s) {
  return pack_kv_launch<bf16>(K, ldk, V, ldv, Kp, Vp, B, H, L, Lpad, hd, s);
}
}
