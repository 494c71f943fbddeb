// lane_xor.h — the value of lane ^ MASK for f64, without ds_bpermute wherever the hardware offers something cheaper (round 6).
// __shfl_xor compiles to ds_bpermute_b32 (two per double): an LDS-pipe instruction with an address register and ~100 cycles of latency, and a butterfly is a CHAIN of
// them.  Inside a row of 16 lanes DPP does the same as a vector-ALU move: xor 8 = row_ror:8, xor 4 = row_half_mirror then every quad reversed, xor 2 / xor 1 = quad_perm;
// xor 16 is a ds_swizzle (bit-mask mode: still the LDS crossbar, but no address and no bank access); xor 32 stays a ds_bpermute.  A sum formed with these is the same
// sum (same pairs, same order of the steps): same bits.  Every lane of the wave must be active where these are called (DPP reads nothing from a disabled lane).
#pragma once
#include <hip/hip_runtime.h>

namespace lanex {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

template <int MASK>
__device__ __forceinline__ double from_partner(double v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "lane ^ MASK inside a wave of 64");
  if constexpr (MASK == 1) return dpp_f64<0xB1>(v);                  // quad_perm:[1,0,3,2]
  else if constexpr (MASK == 2) return dpp_f64<0x4E>(v);             // quad_perm:[2,3,0,1]
  else if constexpr (MASK == 4) return dpp_f64<0x1B>(dpp_f64<0x141>(v));   // row_half_mirror (i -> 7 - i), then quad_perm:[3,2,1,0] (j -> j ^ 3): i -> i ^ 4
  else if constexpr (MASK == 8) return dpp_f64<0x128>(v);            // row_ror:8
  else if constexpr (MASK == 16) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);   // bit-mask mode: and 0x1f, or 0, xor 0x10
    return __hiloint2double(hi, lo);
  } else return __shfl_xor(v, 32, 64);
}

// the same for a step that is a constant after unrolling (off = 32, 16, ... 1)
__device__ __forceinline__ double from_partner_c(double v, int off) {
  switch (off) {
    case 1: return from_partner<1>(v);
    case 2: return from_partner<2>(v);
    case 4: return from_partner<4>(v);
    case 8: return from_partner<8>(v);
    case 16: return from_partner<16>(v);
    default: return from_partner<32>(v);
  }
}

// v summed over the wave by the xor butterfly 32, 16, 8, 4, 2, 1 (the order every wave sum of this library has always used)
__device__ __forceinline__ double wave_sum(double v) {
  v += from_partner<32>(v); v += from_partner<16>(v); v += from_partner<8>(v);
  v += from_partner<4>(v); v += from_partner<2>(v); v += from_partner<1>(v);
  return v;
}

}  // namespace lanex
