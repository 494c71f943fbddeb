// block_red.h — deterministic workgroup-wide f64 sums for the single-workgroup LM kernels (poseopt.hip, sim3opt.hip).
// Fixed association order => run-to-run deterministic, and every thread of every wave ends with the same bits, which
// keeps the LM control flow workgroup-uniform without broadcasting decisions.
#pragma once
#include <hip/hip_runtime.h>
#include "lane_xor.h"

namespace blockred {

constexpr int kWave = 64;
constexpr int kNW = 4;                      // waves per workgroup
constexpr int kThreads = kWave * kNW;
constexpr int kRedDoubles = 2 * kNW * 64;   // two ping-pong buffers of per-wave partials at the head of the LDS

// value select on the bit patterns: a plain `c ? acc[i] : acc[j]` on a register array is turned into a lane-dependent
// ADDRESS select by the compiler, which moves the whole array to scratch memory (measured: 7.5 us per reduction instead of <1)
__device__ __forceinline__ double sel_bits(bool c, double a, double b) {
  const long long m = -(long long)c;
  return __longlong_as_double((__double_as_longlong(a) & m) | (__double_as_longlong(b) & ~m));
}

__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Block-wide sums with a fixed association order (=> deterministic, and every thread of every wave ends with the
// same bits, which keeps the whole LM control flow uniform without broadcasting decisions).
template <int NW>
struct BlockRedT {
  double* buf;   // LDS [2][NW][64]
  int phase;
  int lane, wave;

  // one value: wave butterfly, then the NW partials through LDS
  __device__ __forceinline__ double sum1(double v) {
    v = lanex::wave_sum(v);   // (xor butterfly 32 ... 1; round 6: by v_permlane*_swap / DPP, lane_xor.h; same pairs, same order)
    double* b = buf + phase * (NW * 64);
    phase ^= 1;
    if (lane == 0) b[wave * 64] = v;
    __syncthreads();
    double s = b[0];
#pragma unroll
    for (int w = 1; w < NW; w++) s += b[w * 64];
    return s;
  }

  // 27 values at once (upper triangle of H + b): a halving butterfly moves 16+8+4+2+1+1 = 32 f64 values through
  // the cross-lane network instead of 27*6 = 162; lane l then owns the wave total of value l>>1, the waves meet
  // in LDS, lanes 0..26 add the NW partials and v_readlane hands every total to the whole wave as a scalar.
  __device__ __forceinline__ void sum27(double* acc /* [32], entries 28..31 zero */) {
#pragma unroll
    for (int c = 16, off = 32; c >= 1; c >>= 1, off >>= 1) {
      // (round 6) the upper lanes swap their halves — a real branch: the empty asm keeps the swaps from becoming selects again —, then every lane keeps [0, c) and sends
      // [c, 2c): 3 moves instead of 8 bit operations per exchanged value, and the partner's value comes by DPP / ds_swizzle where they reach (lane_xor.h); same sums
      if ((lane & off) != 0) {
#pragma unroll
        for (int k = 0; k < c; k++) { asm volatile("" : "+v"(acc[k]), "+v"(acc[k + c])); const double tmp = acc[k]; acc[k] = acc[k + c]; acc[k + c] = tmp; }
      }
#pragma unroll
      for (int k = 0; k < c; k++) acc[k] = acc[k] + lanex::from_partner_c(acc[k + c], off);
    }
    acc[0] += lanex::from_partner<1>(acc[0]);
    double* b = buf + phase * (NW * 64);
    phase ^= 1;
    if ((lane & 1) == 0) b[wave * 64 + (lane >> 1)] = acc[0];
    __syncthreads();
    double v = 0.0;
    if (lane < 32) {
      v = b[lane];
#pragma unroll
      for (int w = 1; w < NW; w++) v += b[w * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = bcast_lane(v, k);   // entry 27 is free for a 28th value (e.g. a chi2 partial)
  }
  // 28 values at once THROUGH LDS (round 4).  The halving butterfly above is six dependent levels of cross-lane moves (ds_bpermute: 1.35 us per reduction, measured with the
  // phase clocks of poseopt.hip: 26 reductions = 35 of the kernel's 115 us).  Here every thread stores its 28 values column-wise ([28][64 NW + 8]: the row pad keeps the rows
  // a half-wave reads on different banks), 224 threads add 64 NW / 8 entries each (value = t / 8, every 8th entry from t % 8, four running sums), three cross-lane steps join
  // the 8 parts, and every thread reads the 28 totals back (LDS broadcast): two barriers, 1.15 us.  tr: LDS [kTrDoubles], not shared with `buf`.
  static constexpr int kTrCols = NW * 64, kTrStride = kTrCols + 8, kTrDoubles = 28 * kTrStride + 32;
  __device__ __forceinline__ void sum28_lds(double* acc /* [32], entries 28..31 unused */, double* tr) {
    static_assert(NW * 64 >= 224, "sum28_lds: 224 threads add the columns");
    const int t = wave * 64 + lane;
#pragma unroll
    for (int k = 0; k < 28; k++) tr[k * kTrStride + t] = acc[k];
    __syncthreads();
    double s = 0.0;
    if (t < 224) {
      const double* row = tr + (t >> 3) * kTrStride + (t & 7);
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
      for (int i = 0; i < kTrCols / 8; i += 4) { s0 += row[8 * i]; s1 += row[8 * (i + 1)]; s2 += row[8 * (i + 2)]; s3 += row[8 * (i + 3)]; }
      s = (s0 + s1) + (s2 + s3);
    }
    s = lanex::add_partner<1>(s); s = lanex::add_partner<2>(s); s = lanex::add_partner<4>(s);
    double* out = tr + 28 * kTrStride;
    if (t < 224 && (t & 7) == 0) out[t >> 3] = s;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = out[k];
    // (no trailing barrier: the columns are rewritten only by threads that have passed the second barrier, i.e. after every column read; `out` lies outside the columns and is
    // rewritten only after the NEXT call's first barrier, i.e. after every thread's reads of it)
  }
  // up to 64 values (NV of them meaningful): halving butterfly 32+16+8+4+2+1 = 63 cross-lane moves, lane l then owns
  // the wave total of value l
  template <int NV>
  __device__ __forceinline__ void sum64(double* acc /* [64], entries NV..63 zero */) {
#pragma unroll
    for (int c = 32, off = 32; c >= 1; c >>= 1, off >>= 1) {
      // (round 6) the upper lanes swap their halves — a real branch: the empty asm keeps the swaps from becoming selects again —, then every lane keeps [0, c) and sends
      // [c, 2c): 3 moves instead of 8 bit operations per exchanged value, and the partner's value comes by DPP / ds_swizzle where they reach (lane_xor.h); same sums
      if ((lane & off) != 0) {
#pragma unroll
        for (int k = 0; k < c; k++) { asm volatile("" : "+v"(acc[k]), "+v"(acc[k + c])); const double tmp = acc[k]; acc[k] = acc[k + c]; acc[k + c] = tmp; }
      }
#pragma unroll
      for (int k = 0; k < c; k++) acc[k] = acc[k] + lanex::from_partner_c(acc[k + c], off);
    }
    double* b = buf + phase * (NW * 64);
    phase ^= 1;
    b[wave * 64 + lane] = acc[0];
    __syncthreads();
    double v = b[lane];
#pragma unroll
    for (int w = 1; w < NW; w++) v += b[w * 64 + lane];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = bcast_lane(v, k);
  }
};
using BlockRed = BlockRedT<kNW>;   // the 4-wave kernels (sim3opt.hip)


}  // namespace blockred
