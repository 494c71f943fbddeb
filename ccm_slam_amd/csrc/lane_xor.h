// lane_xor.h — the value of lane ^ MASK for f64, without ds_bpermute wherever the hardware offers something cheaper (round 6).
// __shfl_xor compiles to ds_bpermute_b32 (two per double): an LDS-pipe instruction with an address register and ~100 cycles of latency, and a butterfly is a CHAIN of
// them.  Inside a row of 16 lanes DPP does the same as a vector-ALU move: xor 8 = row_ror:8, xor 4 = row_half_mirror then every quad reversed, xor 2 / xor 1 = quad_perm;
// xor 16 and xor 32 go through gfx950's v_permlane16_swap / v_permlane32_swap: applied to two copies of a register they leave [r0 r0 r2 r2] / [r1 r1 r3 r3] (rows of 16
// lanes; for 32: [lo lo] / [hi hi]) — own value and partner's value side by side in every lane, so `v + partner` is one addition of the two results (in the upper lanes
// with its operands exchanged: the same bits), and the partner's value alone costs a select.  Nothing goes through LDS.  A sum formed with these is the same sum (same
// pairs, same order of the steps): same bits.  Every lane of the wave must be active where these are called (DPP reads nothing from a disabled lane).
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

// MASK = 16 / 32: a = the value of the lane with bit MASK clear, b = that of the lane with it set, of every pair (lane, lane ^ MASK), in both lanes of the pair
template <int MASK>
__device__ __forceinline__ void swap_rows(double v, double& a, double& b) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (MASK == 32) {
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double(h[0], l[0]); b = __hiloint2double(h[1], l[1]);
  } else {
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double(h[0], l[0]); b = __hiloint2double(h[1], l[1]);
  }
}

template <int MASK>
__device__ __forceinline__ double from_partner(double v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "lane ^ MASK inside a wave of 64");
  if constexpr (MASK == 1) return dpp_f64<0xB1>(v);                  // quad_perm:[1,0,3,2]
  else if constexpr (MASK == 2) return dpp_f64<0x4E>(v);             // quad_perm:[2,3,0,1]
  else if constexpr (MASK == 4) return dpp_f64<0x1B>(dpp_f64<0x141>(v));   // row_half_mirror (i -> 7 - i), then quad_perm:[3,2,1,0] (j -> j ^ 3): i -> i ^ 4
  else if constexpr (MASK == 8) return dpp_f64<0x128>(v);            // row_ror:8
  else {   // 16, 32
    double a, b;
    swap_rows<MASK>(v, a, b);
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (lane & MASK) ? a : b;
  }
}

// v + (the value of lane ^ MASK): one step of an xor butterfly
template <int MASK>
__device__ __forceinline__ double add_partner(double v) {
  if constexpr (MASK >= 16) { double a, b; swap_rows<MASK>(v, a, b); return a + b; }   // (own + partner in the lower lane of a pair, partner + own in the upper one: the same bits)
  else return v + from_partner<MASK>(v);
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

// inclusive prefix sum of a 32-bit integer over the 64 lanes of the wave (exact: integer additions): four shifts inside the rows of 16 lanes (row_shr, lanes without a
// source get 0), then the rows' totals handed on by row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) — six vector-ALU instructions instead of six ds_bpermute
// round trips with a select each.  Every lane of the wave must be active.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
  return v;
}

// v summed over the wave by the xor butterfly 32, 16, 8, 4, 2, 1 (the order every wave sum of this library has always used)
__device__ __forceinline__ double wave_sum(double v) {
  v = add_partner<32>(v); v = add_partner<16>(v); v = add_partner<8>(v);
  v = add_partner<4>(v); v = add_partner<2>(v); v = add_partner<1>(v);
  return v;
}

}  // namespace lanex
