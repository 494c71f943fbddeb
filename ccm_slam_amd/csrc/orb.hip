// orb.hip — ORB extraction on MI355X (gfx950): pyramid, FAST-9/16 score, per-cell NMS with threshold
// fallback, intensity-centroid orientation, 7x7 Gaussian blur, steered BRIEF.  Integer / exactly
// specified f32 arithmetic: bit-exact against the reference semantics restated in oracle/orb_ref.cpp.
//
// Replaces cslam::ORBextractor (cslam/src/ORBextractor.cpp):
//   ctor :579-639            -> ccm_orb_create (host tables)
//   ComputePyramid :1280     -> orb_resize_kernel, one launch per level (level l depends on l-1)
//   ComputeKeyPointsOctTree  -> orb_fast_score_kernel (all levels, one launch) +
//        :933-998               orb_cells_kernel (one workgroup per 30-px cell: threshold, 3x3 strict NMS with
//                               zero padding at the cell's interior edge, empty-cell retry with minThFAST,
//                               raster-order compaction) + orb_compact_kernel
//   DistributeOctTree :707   -> host (serial by nature, <= ~10 N candidates), own
//                               index-linked implementation below
//   IC_Angle :68 + computeOrbDescriptor :100 -> orb_orient_desc_kernel (one wave per keypoint; the 256
//                               comparisons are 4 x 64-lane ballots = 32 bytes)
//   GaussianBlur :1259       -> orb_blur_kernel (LDS tiled separable 8.8 fixed point, REFLECT_101)
//
// MI355X notes: one frame moves ~9 MB, far below what the Infinity Cache holds, so a single frame is
// launch-latency bound (12 launches + one host round trip for the octree); all levels live in ONE
// device allocation and the score / cell / blur kernels cover every level per launch to keep the launch
// count down.  The batch entry point pipelines frames over several streams so the host octree of frame
// i overlaps the device phases of frame i+1.
#include "common.h"
#include "lane_xor.h"
#include "test_internal.h"
#include "orb_math.h"
#include "orb_pattern.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <list>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr int kWave = 64;
constexpr int kMaxLevels = 16;
constexpr int kEdge = 19, kHalfPatch = 15, kPatch = 31;
constexpr int kCellMax = 60;       // largest cell edge supported by the LDS tile: a level with ONE column of cells has cells of up to 59 px (:949-955)
constexpr int kCellCap = 900;      // >= ceil(60/2)^2: strict 3x3 NMS keeps no two adjacent pixels
constexpr int kCandFirstCopy = 16384;

struct LevelInfo {
  int w, h, stride, off;           // off: pixel offset of the level in the pyramid buffer (and in the score / blur buffers that mirror it)
  long long poff; int pstride;     // where the level's PIXELS are read, relative to the pyramid buffer: (off, stride), except level 0 of a batch frame, which is read in
                                   // place from the caller's image (round 4: no device-to-device copy of every frame into the pyramid buffer)
  int nCols, nRows, wCell, hCell, cellBase;
  int rowBase;                     // first global row index (sum of h of previous levels)
  int tabOff;                      // offset into the resize tables
  float scale;
};

struct OrbDev {
  int nlevels, ncells, totalRows, maxW;
  LevelInfo lv[kMaxLevels];
  int umax[16];
};

// Several frames in ONE launch (round 4, batch API): blockIdx.y is the frame.  Every device buffer of a frame's buffer set is carved from one block with the same layout in
// every set (orb_alloc_bufs), so frame f's buffers are the kernel's pointer arguments (those of the group's first set) moved by ONE byte offset; the caller's output arrays and
// the frames' level-0 pixels have their own strides.  n = 1 with zero offsets is the single-frame form of the same kernels.
constexpr int kOrbGroup = 4;
struct OrbFrames {
  int n, pstride0;                 // frames of the launch; row stride of level 0 where the frames' pixels are read
  long long set[kOrbGroup];        // bytes from the first set's block to frame f's
  long long poff0[kOrbGroup];      // level 0 of frame f relative to ITS pyramid buffer (in place in the caller's frames for the batch API)
  long long kout[kOrbGroup], desc[kOrbGroup], cnt[kOrbGroup];   // bytes from the output pointers given to frame f's outputs
};
template <typename T> __device__ __forceinline__ T* orb_shift(T* p, long long bytes) { return p ? (T*)((uintptr_t)p + (uintptr_t)bytes) : p; }

__constant__ int8_t c_pattern[1024];

// ---- pyramid: cv::resize INTER_LINEAR 8UC1 fixed point (oracle/orb_ref.cpp resize_linear_u8) ----
__global__ __launch_bounds__(256) void orb_resize_kernel(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                                         uint8_t* __restrict__ dst, int dw, int dh, int dstride,
                                                         const int16_t* __restrict__ xofs, const int16_t* __restrict__ ialpha,
                                                         const int16_t* __restrict__ yofs, const int16_t* __restrict__ ibeta) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y;
  if (dx >= dw) return;
  const int sy = yofs[dy];
  const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
  const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
  const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
  const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
  const uint8_t* S0 = src + (size_t)sy0 * sstride;
  const uint8_t* S1 = src + (size_t)sy1 * sstride;
  const int r0 = S0[sx] * a0 + S0[sx1] * a1;
  const int r1 = S1[sx] * a0 + S1[sx1] * a1;
  dst[(size_t)dy * dstride + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// ---- the whole pyramid in ONE launch (round 4) ------------------------------------------------------------------------
// Seven dependent launches of orb_resize_kernel are ~42 us of a single frame's ~130 us chain (3.9 us + a 2 us gap each, 40 K pixels or fewer
// per launch after level 2).  Here a workgroup owns a tile of level 1 and, recursively, the pixels of every later level whose top-left source
// pixel it owns (the offset tables are monotone, so that is a rectangle per level and the rectangles of all workgroups tile the level).  It
// stages its level-0 source rectangle and the table entries of all its levels in LDS once, then walks down the levels LDS -> LDS, writing the
// owned rectangle of each level to the pyramid buffer.  A level's computed rectangle is its owned one plus the columns / rows to the right /
// below that the NEXT level's owned + halo pixels read (1, 2.2, 3.6 ... ~10 pixels at level 1 for 8 levels at 1.2): the halo is recomputed
// instead of exchanged.  The per-pixel arithmetic is orb_resize_kernel's.
// Tile tables (host, orb_prepare): per (tile column, level): x0 (first owned column; level 0: first source column), x1 (end of the owned
// columns), cx1 (end of the computed columns), start of the level's entries in the staged tables; the same per (tile row, level).
constexpr int kPyrTW = 32, kPyrTH = 32, kPyrTPB = 1024;

struct PyrArgs {
  const int4* tcol; const int4* trow;   // [tile column / row][level] {x0, x1, cx1, start of the level's entries among the tile column's / row's}
  const uint2* ent;                     // table entries: a block of ent_sx entries per tile column, then (from ent_y0) a block of ent_sy entries per tile row
  int ntx, buf_bytes, ent_sx, ent_sy, ent_y0;
  int ex0, ey0;                         // extent of the level-0 rectangle every workgroup stages (the largest any tile needs; clipped to the image)
  double scale_x, scale_y;              // level 0 / level 1 size ratios: a tile's first source column / row is computed, not loaded (see the kernel)
  unsigned long long* dbg;              // CCM_ORB_OCT_DBG: phase clocks of workgroup 0, else nullptr
};

// barrier that orders LDS traffic only: __syncthreads() also waits for the level's global stores (~0.8 us per level, measured)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kPyrTPB) void orb_pyramid_kernel(OrbDev d, uint8_t* __restrict__ pyr, PyrArgs a, OrbFrames fr) {
  extern __shared__ __attribute__((aligned(16))) uint8_t plds[];
  const int fy = blockIdx.y;
  pyr = orb_shift(pyr, fr.set[fy]);
  __shared__ int4 s_c[kMaxLevels], s_r[kMaxLevels];
  const int t = threadIdx.x;
  const bool timing = a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && t == 0;
  long long tk = timing ? wall_clock64() : 0;
  const int tc = blockIdx.x % a.ntx, tr = blockIdx.x / a.ntx;
  const int4* C = a.tcol + (size_t)tc * d.nlevels;
  const int4* Rw = a.trow + (size_t)tr * d.nlevels;
  uint8_t* buf[2] = {plds, plds + a.buf_bytes};
  uint2* tx = reinterpret_cast<uint2*>(plds + 2 * (size_t)a.buf_bytes);   // per computed column of every level {sx - x0_prev, sx1 - x0_prev, a0, a1} (4 x int16); from a.ent_sx: per computed row {sy0 - y0_prev, sy1 - y0_prev, b0, b1}
  // Staging: level descriptors, the tile column's / row's table entries (built by the host: they depend on the tile column / row only) and the level-0 source
  // rectangle, in ONE memory round trip: no address depends on a loaded value (the rectangle's origin is the offset-table formula of orb_prepare evaluated here,
  // its extent the largest any tile needs; entry blocks have a fixed stride), and every load is issued before the first LDS store.  (Measured on the way: a loop of
  // load -> store pairs 27 us per launch; descriptors loaded first, then entries and pixels: 5 us of staging.)
  const LevelInfo& L0 = d.lv[0];
  int x0, y0;
  {
    const float fx = (float)((tc * kPyrTW + 0.5) * a.scale_x - 0.5), fy = (float)((tr * kPyrTH + 0.5) * a.scale_y - 0.5);
    x0 = min(max((int)floorf(fx), 0), L0.w - 1); y0 = min(max((int)floorf(fy), 0), L0.h - 1);
  }
  const int cw0 = min(a.ex0, L0.w - x0), ch0 = min(a.ey0, L0.h - y0);
  {
    const uint2* gx = a.ent + (size_t)tc * a.ent_sx;
    const uint2* gy = a.ent + a.ent_y0 + (size_t)tr * a.ent_sy;
    const uint8_t* src = pyr + fr.poff0[fy] + (long long)y0 * fr.pstride0 + x0;
    const float rcw = 1.f / (float)cw0;
    const int ne = a.ent_sx + a.ent_sy;
    constexpr int UE = 1024 / kPyrTPB, UP = 3072 / kPyrTPB;
    uint2 e[UE]; uint8_t v[UP];
    int4 dsc = make_int4(0, 0, 0, 0);
    if (t < d.nlevels) dsc = C[t]; else if (t >= 64 && t < 64 + d.nlevels) dsc = Rw[t - 64];
#pragma unroll
    for (int k = 0; k < UE; k++) { const int i = k * kPyrTPB + t; e[k] = make_uint2(0, 0); if (i < a.ent_sx) e[k] = gx[i]; else if (i < ne) e[k] = gy[i - a.ent_sx]; }
#pragma unroll
    for (int k = 0; k < UP; k++) {
      const int i = k * kPyrTPB + t;
      v[k] = 0;
      if (i < cw0 * ch0) { const int yy = (int)(((float)i + 0.5f) * rcw), xx = i - yy * cw0; v[k] = src[(long long)yy * fr.pstride0 + xx]; }
    }
    if (t < d.nlevels) s_c[t] = dsc; else if (t >= 64 && t < 64 + d.nlevels) s_r[t - 64] = dsc;
#pragma unroll
    for (int k = 0; k < UE; k++) { const int i = k * kPyrTPB + t; if (i < ne) tx[i] = e[k]; }
#pragma unroll
    for (int k = 0; k < UP; k++) { const int i = k * kPyrTPB + t; if (i < cw0 * ch0) buf[0][i] = v[k]; }
    // (tiles beyond the unrolled batches: plain loops)
    for (int i = UE * kPyrTPB + t; i < ne; i += kPyrTPB) tx[i] = i < a.ent_sx ? gx[i] : gy[i - a.ent_sx];
    for (int i = UP * kPyrTPB + t; i < cw0 * ch0; i += kPyrTPB) { const int yy = (int)(((float)i + 0.5f) * rcw), xx = i - yy * cw0; buf[0][i] = src[(long long)yy * fr.pstride0 + xx]; }
  }
  __syncthreads();
  if (timing) { const long long tn = wall_clock64(); a.dbg[0] += tn - tk; tk = tn; a.dbg[23] += 1; }
  int pcw = cw0;     // width of the previous level's rectangle in LDS
  for (int l = 1; l < d.nlevels; l++) {
    const LevelInfo& L = d.lv[l];
    const int4 cl = s_c[l], rl = s_r[l];
    const int lx0 = cl.x, x1 = cl.y, cw = cl.z - lx0, ly0 = rl.x, y1 = rl.y, ch = rl.z - ly0;
    const uint8_t* S = buf[(l - 1) & 1];
    uint8_t* D = buf[l & 1];
    uint8_t* dst = pyr + L.off + (size_t)ly0 * L.stride + lx0;
    const uint2* ex = tx + cl.w;
    const uint2* ey = tx + a.ent_sx + rl.w;
    const float rcw = 1.f / (float)max(cw, 1);
    const int n = cw * ch, ow = x1 - lx0, oh = y1 - ly0;
    constexpr int UL = 1024 / kPyrTPB;   // pixels per thread and batch: their LDS reads (entries, then four source bytes) overlap
    for (int base = 0; base < n; base += UL * kPyrTPB) {
      uint8_t v[UL]; int xx[UL], yy[UL];
#pragma unroll
      for (int k = 0; k < UL; k++) {
        v[k] = 0; xx[k] = yy[k] = 0;
        if (base + k * kPyrTPB + (t & ~63) >= n) continue;   // the whole wave is beyond the rectangle (the small levels keep half of the 16 waves busy: the loop is issue-bound)
        const int i = min(base + k * kPyrTPB + t, n - 1);
        yy[k] = (int)(((float)i + 0.5f) * rcw); xx[k] = i - yy[k] * cw;
        const uint2 qx = ex[xx[k]], qy = ey[yy[k]];
        const int sx = (int16_t)(qx.x & 0xffff), sx1 = (int16_t)(qx.x >> 16), a0 = (int16_t)(qx.y & 0xffff), a1 = (int16_t)(qx.y >> 16);
        const int sy0 = (int16_t)(qy.x & 0xffff), sy1 = (int16_t)(qy.x >> 16), b0 = (int16_t)(qy.y & 0xffff), b1 = (int16_t)(qy.y >> 16);
        const uint8_t* S0 = S + sy0 * pcw;
        const uint8_t* S1 = S + sy1 * pcw;
        const int r0 = S0[sx] * a0 + S0[sx1] * a1;
        const int r1 = S1[sx] * a0 + S1[sx1] * a1;
        v[k] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
      }
#pragma unroll
      for (int k = 0; k < UL; k++) {
        const int i = base + k * kPyrTPB + t;
        if (i < n) {
          D[i] = v[k];
          if (xx[k] < ow && yy[k] < oh) dst[(size_t)yy[k] * L.stride + xx[k]] = v[k];
        }
      }
    }
    pcw = cw;
    lds_barrier();
    if (timing && l < 20) { const long long tn = wall_clock64(); a.dbg[l] += tn - tk; tk = tn; }
  }
}

__device__ __forceinline__ int find_level(const OrbDev& d, int grow) {
  int l = 0;
#pragma unroll 1
  for (int k = 1; k < d.nlevels; k++) if (grow >= d.lv[k].rowBase) l = k;
  return l;
}

// ---- FAST-9/16 corner score for every pixel of every level ----------------------------------------
// score = max over the 16 arcs of 9 contiguous ring pixels of min(v - p)  (dark ring)  or of min(p - v)
// (bright ring), minus 1  ==  cornerScore<16>() of OpenCV for any pixel that is a corner at threshold t
// (then score >= t); a pixel is a corner at threshold t iff score >= t.  Stored clamped to [0,255].
// score of the pixel at p (row stride s, any address space): see above
__device__ __forceinline__ int orb_fast_score_at(const uint8_t* p, int s) {
  const int v = p[0];
  int dd[16];
  dd[0] = v - p[3 * s];          dd[1] = v - p[3 * s + 1];      dd[2] = v - p[2 * s + 2];      dd[3] = v - p[s + 3];
  dd[4] = v - p[3];              dd[5] = v - p[-s + 3];         dd[6] = v - p[-2 * s + 2];     dd[7] = v - p[-3 * s + 1];
  dd[8] = v - p[-3 * s];         dd[9] = v - p[-3 * s - 1];     dd[10] = v - p[-2 * s - 2];    dd[11] = v - p[-s - 3];
  dd[12] = v - p[-3];            dd[13] = v - p[s - 3];         dd[14] = v - p[2 * s - 2];     dd[15] = v - p[3 * s - 1];
  // sliding min / max of width 9 over the circular ring by doubling: 2,4,8 then +1
  int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { mn2[k] = min(dd[k], dd[(k + 1) & 15]); mx2[k] = max(dd[k], dd[(k + 1) & 15]); }
#pragma unroll
  for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
#pragma unroll
  for (int k = 0; k < 16; k++) { mn8[k] = min(mn4[k], mn4[(k + 4) & 15]); mx8[k] = max(mx4[k], mx4[(k + 4) & 15]); }
  int A = -256, B = 256;
#pragma unroll
  for (int k = 0; k < 16; k++) { A = max(A, min(mn8[k], dd[(k + 8) & 15])); B = min(B, max(mx8[k], dd[(k + 8) & 15])); }
  const int sc = max(A, -B) - 1;
  return min(max(sc, 0), 255);
}
// the whole score map (every level in one launch): since round 4 only for ccm_orb_debug_level — the extraction computes the scores inside orb_cells_kernel
__global__ __launch_bounds__(256) void orb_fast_score_kernel(OrbDev d, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ score) {
  const int grow = blockIdx.y;
  const int l = find_level(d, grow);
  const LevelInfo L = d.lv[l];
  const int y = grow - L.rowBase;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < kEdge || x >= L.w - kEdge || y < kEdge || y >= L.h - kEdge) return;
  score[L.off + (size_t)y * L.stride + x] = (uint8_t)orb_fast_score_at(pyr + L.poff + (size_t)y * L.pstride + x, L.pstride);
}

// ---- per-cell threshold + NMS + fallback + ordered compaction ------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane) { (void)lane; return lanex::wave_incl_scan_i32(v); }   // (round 6: by DPP, lane_xor.h; every lane of the wave is active at both call sites)

// Round 4: the FAST scores of the cell's interior are computed HERE, from a (w + 6) x (h + 6) pixel tile staged in LDS (the score kernel over all levels and its
// 1.1 MB score map are gone from the extraction: one launch and one round trip through memory less per frame; 13 % of the pixels are scored twice, by the
// two cells whose tiles overlap).  `pyr` = the pyramid; the scores are the same integers, so everything downstream is unchanged.
__global__ __launch_bounds__(256) void orb_cells_kernel(OrbDev d, const uint8_t* __restrict__ pyr, int iniTh, int minTh,
                                                        uint32_t* __restrict__ cell_slots, int* __restrict__ cell_counts, OrbFrames fr) {
  const int fy = blockIdx.y;
  pyr = orb_shift(pyr, fr.set[fy]); cell_slots = orb_shift(cell_slots, fr.set[fy]); cell_counts = orb_shift(cell_counts, fr.set[fy]);
  __shared__ uint8_t tile[(kCellMax + 2) * (kCellMax + 2)];
  __shared__ uint8_t pix[(kCellMax + 6) * (kCellMax + 6 + 2)];
  __shared__ uint8_t sct[kCellMax * kCellMax];
  __shared__ int wsum[4];
  __shared__ int total_s;
  const int cell = blockIdx.x;
  int l = 0;
  for (int k = 1; k < d.nlevels; k++) if (cell >= d.lv[k].cellBase) l = k;
  const LevelInfo L = d.lv[l];
  const int ci = (cell - L.cellBase) / L.nCols, cj = (cell - L.cellBase) % L.nCols;
  const int minB = kEdge - 3, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
  const int iniX = minB + cj * L.wCell, iniY = minB + ci * L.hCell;
  int maxX = iniX + L.wCell + 6, maxY = iniY + L.hCell + 6;
  if (iniY >= maxBY - 3 || iniX >= maxBX - 6) { if (threadIdx.x == 0) cell_counts[cell] = 0; return; }
  if (maxX > maxBX) maxX = maxBX;
  if (maxY > maxBY) maxY = maxBY;
  // interior processed by cv::FAST on the ROI [iniX,maxX) x [iniY,maxY): 3-pixel margin
  const int x0 = iniX + 3, y0 = iniY + 3;
  const int iw = (maxX - iniX) - 6, ih = (maxY - iniY) - 6;
  if (iw <= 0 || ih <= 0) { if (threadIdx.x == 0) cell_counts[cell] = 0; return; }
  const int tw = iw + 2;   // zero-padded tile width
  const int npx = iw * ih;
  const int per = (npx + 255) / 256;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  {
    // pixels of [x0 - 3, x0 + iw + 3) x [y0 - 3, y0 + ih + 3): inside the level, every cell interior lies >= 19 px from the level's edges
    const int pw = iw + 6, ps = (pw + 3) & ~3;
    for (int t = threadIdx.x; t < (ih + 6) * pw; t += 256) {
      const int ty = t / pw, tx = t % pw;
      pix[ty * ps + tx] = pyr[(l == 0 ? fr.poff0[fy] : (long long)L.off) + (long long)(y0 - 3 + ty) * (l == 0 ? fr.pstride0 : L.stride) + (x0 - 3 + tx)];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < npx; t += 256) {
      const int ty = t / iw, tx = t % iw;
      sct[t] = (uint8_t)orb_fast_score_at(pix + (ty + 3) * ps + tx + 3, ps);
    }
    __syncthreads();
  }
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? iniTh : minTh;
    for (int t = threadIdx.x; t < (ih + 2) * tw; t += 256) {
      const int ty = t / tw - 1, tx = t % tw - 1;
      int m = 0;
      if (tx >= 0 && tx < iw && ty >= 0 && ty < ih) {
        const int sc = sct[ty * iw + tx];
        m = sc >= th ? sc : 0;
      }
      tile[t] = (uint8_t)m;
    }
    __syncthreads();
    uint32_t flags = 0;
    int cnt = 0;
    const int p0 = threadIdx.x * per;
    for (int k = 0; k < per; k++) {
      const int p = p0 + k;
      if (p < npx) {
        const int ty = p / iw, tx = p % iw;
        const uint8_t* c = tile + (ty + 1) * tw + tx + 1;
        const int m = c[0];
        if (m > 0 && m > c[-1] && m > c[1] && m > c[-tw - 1] && m > c[-tw] && m > c[-tw + 1] && m > c[tw - 1] && m > c[tw] && m > c[tw + 1]) {
          flags |= 1u << k; cnt++;
        }
      }
    }
    const int incl = wave_incl_scan(cnt, lane);
    if (lane == kWave - 1) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wsum[w];
    if (threadIdx.x == 0) total_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int total = total_s;
    if (total > 0 || pass == 1) {
      int pos = base + incl - cnt;
      for (int k = 0; k < per; k++) {
        if (flags & (1u << k)) {
          const int p = p0 + k;
          const int ty = p / iw, tx = p % iw;
          const int sc = tile[(ty + 1) * tw + tx + 1];
          // candidate coordinates relative to (minBorderX, minBorderY), as in vToDistributeKeys (:989-994)
          const uint32_t rx = (uint32_t)(x0 + tx - minB), ry = (uint32_t)(y0 + ty - minB);
          if (pos < kCellCap) cell_slots[(size_t)cell * kCellCap + pos] = rx | (ry << 12) | ((uint32_t)sc << 24);
          pos++;
        }
      }
      if (threadIdx.x == 0) cell_counts[cell] = min(total, kCellCap);
      return;
    }
    __syncthreads();
  }
}

// concatenate the per-cell lists in cell order: out = [offsets (ncells+1)] [records ...].
// One workgroup per cell: its offset is the sum of the counts of all earlier cells (<= ~1000 ints, cheaper
// than a separate scan launch), then it copies its own records.
__global__ __launch_bounds__(256) void orb_compact_kernel(int ncells, const uint32_t* __restrict__ cell_slots,
                                                           const int* __restrict__ cell_counts, int* __restrict__ offsets,
                                                           uint32_t* __restrict__ records) {
  __shared__ int wsum[4];
  const int c = blockIdx.x, t = threadIdx.x;
  int s = 0;
  for (int k = t; k < c; k += 256) s += cell_counts[k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
  if ((t & (kWave - 1)) == 0) wsum[t / kWave] = s;
  __syncthreads();
  const int o = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const int n = cell_counts[c];
  if (t == 0) { offsets[c] = o; if (c == ncells - 1) offsets[ncells] = o + n; }
  for (int k = t; k < n; k += 256) records[o + k] = cell_slots[(size_t)c * kCellCap + k];
}

// ---- GaussianBlur 7x7 sigma 2, 8.8 fixed point {18,34,48,56,48,34,18}, REFLECT_101 -------------------
__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

constexpr int kBlurTW = 64, kBlurTH = 16;
constexpr int kBlurInBytes = (kBlurTH + 6) * (kBlurTW + 6), kBlurHsBytes = (kBlurTH + 6) * kBlurTW * 2;
constexpr int kBlurLds = (kBlurInBytes + kBlurHsBytes + 15) & ~15;   // LDS of one 256-thread tile team
// one 64 x 16 tile by a team of 256 threads (tid 0..255); `live` = the team has a tile (a team without one only keeps the two barriers of its workgroup)
__device__ __forceinline__ void orb_blur_tile(const OrbDev& d, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, bool live, int l, int bx, int by, int tid,
                                              uint8_t* in, uint16_t* hs, long long poff0, int pstride0) {
  LevelInfo L = d.lv[live ? l : 0];
  if (!live || l == 0) { L.poff = poff0; L.pstride = pstride0; }   // level 0 of THIS frame (OrbFrames)
  const uint8_t* src = pyr + L.poff;
  if (live)
    for (int t = tid; t < (kBlurTH + 6) * (kBlurTW + 6); t += 256) {
      const int ty = t / (kBlurTW + 6), tx = t % (kBlurTW + 6);
      const int gx = reflect101(bx + tx - 3, L.w), gy = reflect101(by + ty - 3, L.h);
      in[t] = src[(size_t)gy * L.pstride + gx];
    }
  __syncthreads();
  if (live)
    for (int t = tid; t < (kBlurTH + 6) * kBlurTW; t += 256) {
      const int ty = t / kBlurTW, tx = t % kBlurTW;
      const uint8_t* r = in + ty * (kBlurTW + 6) + tx;
      hs[t] = (uint16_t)(18 * (r[0] + r[6]) + 34 * (r[1] + r[5]) + 48 * (r[2] + r[4]) + 56 * r[3]);
    }
  __syncthreads();
  if (live)
    for (int t = tid; t < kBlurTH * kBlurTW; t += 256) {
      const int ty = t / kBlurTW, tx = t % kBlurTW;
      const int gx = bx + tx, gy = by + ty;
      if (gx < L.w && gy < L.h) {
        const uint16_t* c = hs + ty * kBlurTW + tx;
        const uint32_t s = 18u * (c[0] + c[6 * kBlurTW]) + 34u * (c[kBlurTW] + c[5 * kBlurTW]) + 48u * (c[2 * kBlurTW] + c[4 * kBlurTW]) + 56u * c[3 * kBlurTW];
        blur[L.off + (size_t)gy * L.stride + gx] = (uint8_t)((s + (1u << 15)) >> 16);
      }
    }
}
__global__ __launch_bounds__(256) void orb_blur_kernel(OrbDev d, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                       const int* __restrict__ tile_level, const int* __restrict__ tile_xy) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kBlurLds];
  orb_blur_tile(d, pyr, blur, true, tile_level[blockIdx.x], tile_xy[2 * blockIdx.x], tile_xy[2 * blockIdx.x + 1], threadIdx.x, lds, reinterpret_cast<uint16_t*>(lds + ((kBlurInBytes + 1) & ~1)),
                d.lv[0].poff, d.lv[0].pstride);
}

// the blur tiles of a GROUP of frames (batch API; frame = blockIdx.y) with the tile's own 5 KB of LDS.  Round 4 let the blur tiles of a frame ride in the octree kernel's
// launch, whose workgroups all ask for a whole CU's LDS (one plan size per launch): right for ONE frame — 8 octree workgroups leave the chip idle —, but a group of
// four frames brought ~600 blur workgroups that could only run one per CU: 128 us per group against 36 for a single frame.  Groups launch this kernel and an
// octree kernel without blur workgroups instead (round 5).
__global__ __launch_bounds__(256) void orb_blur_frames_kernel(OrbDev d, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                              const int* __restrict__ tile_level, const int* __restrict__ tile_xy, OrbFrames fr) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kBlurLds];
  const int fy = blockIdx.y;
  const long long sh_ = fr.set[fy];
  orb_blur_tile(d, orb_shift(pyr, sh_), orb_shift(blur, sh_), true, tile_level[blockIdx.x], tile_xy[2 * blockIdx.x], tile_xy[2 * blockIdx.x + 1], threadIdx.x, lds,
                reinterpret_cast<uint16_t*>(lds + ((kBlurInBytes + 1) & ~1)), fr.poff0[fy], fr.pstride0);
}

// ---- orientation + descriptor: one wave per keypoint -----------------------------------------------
struct KpIn { int16_t x, y; int16_t level; int16_t response; };

// ---- DistributeOctTree on the device (ORBextractor.cpp:707-931): one workgroup per pyramid level, everything in LDS ----------------------------
// The reference walks a std::list and splits nodes one by one; the order of that list decides ties and the output order.  The walk is
// restated in ROUNDS that a workgroup can execute in parallel (validated against the oracle's sequential restatement before it was written
// as a kernel):
//  * keys never move: every stable partition keeps the keys of a node in ascending candidate index, so a key only needs the slot of its
//    current node, and "first maximum of the response" at the end is max(response, then smallest index);
//  * one split pass: the nodes to split carry a processing rank (main loop: list order of the expandable nodes; final phase: descending
//    (size, creation index) with the cut where the list reaches N — prefix sums of children-1 in rank order); children are created in rank
//    order x quadrant order, and because every child is pushed to the FRONT the new list is reverse(creation order) followed by the old
//    list without the split nodes;
//  * node ids of the reference (creation order) are only ever compared inside one pass, so the creation index of the pass is enough.
// A level needs 12 B per candidate and 52 B per list slot (<= 4 N + 16 slots) of LDS; levels that do not fit raise a flag and the frame is
// redone through the host octree.
constexpr int kOctTPB = 1024;
struct OctNode { int16_t ulx, uly, urx, bry; uint16_t size; uint16_t nomore; };
struct OctArgs {
  const int* cand;        // [ncells + 1 offsets][records]
  int ncells, nlevels;
  int nfeat[kMaxLevels];
  int lcap[kMaxLevels];   // list slots per level
  int kcap;               // candidates per level that fit
  KpIn* stage;            // [nlevels][stage_stride]
  int stage_stride;
  int* counts;            // [nlevels] list sizes, [nlevels] = arrival counter
  KpIn* kin; int* n_out; int kp_cap;   // n_out: [keypoint count, overflow flag]
  unsigned long long* dbg;   // nullable: phase clocks of level 0 (timing experiments)
  // round 4: the per-cell lists of orb_cells_kernel read directly (no orb_compact_kernel in front); nullptr = `cand` holds the compacted lists
  const uint32_t* cell_slots; const int* cell_counts;
  // round 4: workgroups behind the `nlevels` octree ones blur 64 x 16 tiles (TPB / 256 tiles each): the 7 x 7 Gaussian does not depend on the octree and the
  // octree's one-workgroup-per-level rounds leave the rest of the chip idle, so both travel in ONE launch.  n_blur_tiles = 0: no blur workgroups.
  const uint8_t* pyr; uint8_t* blur; const int* tile_level; const int* tile_xy; int n_blur_tiles;
};

// exclusive scan of two consecutive items per thread over the workgroup (items 2t, 2t+1); returns the total.  wsum: LDS [TPB / 64 + 1]
template <int TPB>
__device__ __forceinline__ int oct_scan2(int v0, int v1, int& e0, int& e1, int* wsum) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int a = v0 + v1;
  const int inc = lanex::wave_incl_scan_i32(a);   // (round 6: by DPP, lane_xor.h; the whole workgroup calls this — barriers below)
  __syncthreads();                       // wsum may still be read by the previous call
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TPB / 64; w++) { const int x = wsum[w]; if (w < wv) base += x; tot += x; }
  e0 = base + inc - a;
  e1 = e0 + v0;
  return tot;
}

template <int TPB>
__global__ __launch_bounds__(TPB) void orb_octree_kernel(OrbDev d, OctArgs a, OrbFrames fr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oct_lds[];
  // 1-D grid, the octree workgroups of ALL frames first (level-major inside a frame), then the frames' blur workgroups: every workgroup of this kernel asks for a whole CU's
  // LDS, and with the frames along blockIdx.y the octree workgroups of the later frames queued behind hundreds of blur workgroups (132 us for four frames, 36 for one)
  const int n_oct = a.nlevels * fr.n;
  const int bwg = (int)(gridDim.x - n_oct) / fr.n;                         // blur workgroups per frame
  const int fy = (int)blockIdx.x < n_oct ? (int)blockIdx.x / a.nlevels : ((int)blockIdx.x - n_oct) / bwg;
  const int bx = (int)blockIdx.x < n_oct ? (int)blockIdx.x % a.nlevels : a.nlevels + ((int)blockIdx.x - n_oct) % bwg;   // the frame's own index: [levels | blur workgroups]
  // frame fy's buffers (the arguments are those of the group's first set).  Local copies: writing into the by-value argument block would move all of it — the per-level
  // tables are indexed at run time — to scratch memory (272 bytes per lane, the kernel 36 -> 54 us).
  const long long sh_ = fr.set[fy];
  const int* const f_cand = orb_shift(a.cand, sh_); KpIn* const f_stage = orb_shift(a.stage, sh_); int* const f_counts = orb_shift(a.counts, sh_);
  KpIn* const f_kin = orb_shift(a.kin, sh_); int* const f_n_out = orb_shift(a.n_out, sh_);
  const uint32_t* const f_cell_slots = orb_shift(a.cell_slots, sh_); const int* const f_cell_counts = orb_shift(a.cell_counts, sh_);
  const uint8_t* const f_pyr = orb_shift(a.pyr, sh_); uint8_t* const f_blur = orb_shift(a.blur, sh_);
  unsigned long long* const f_dbg = fy ? nullptr : a.dbg;
  (void)f_kin;
  __shared__ int wsum[TPB / 64 + 1];
  __shared__ int sh[16];                 // [0] list size  [1] cur buffer  [2] created  [3] new candidates  [4] finish  [5] cut rank  [6] overflow
  __shared__ int rootcnt[16], rootslot[16];
  const int t = threadIdx.x;
  if (bx >= a.nlevels) {   // a blur workgroup: TPB / 256 teams, one tile each
    const int team = t >> 8, tile = (bx - a.nlevels) * (TPB / 256) + team;
    const bool live = tile < a.n_blur_tiles;
    uint8_t* in = oct_lds + (size_t)team * kBlurLds;
    orb_blur_tile(d, f_pyr, f_blur, live, live ? a.tile_level[tile] : 0, live ? a.tile_xy[2 * tile] : 0, live ? a.tile_xy[2 * tile + 1] : 0, t & 255, in,
                  reinterpret_cast<uint16_t*>(in + ((kBlurInBytes + 1) & ~1)), fr.poff0[fy], fr.pstride0);
    return;
  }
  const int l = bx;
  const LevelInfo Lv = d.lv[l];
  const int N = a.nfeat[l], Lcap = a.lcap[l];
  const int ncl = Lv.nCols * Lv.nRows;
  const bool from_cells = f_cell_counts != nullptr;      // (the host chooses this form only when every level has at most 2 TPB cells)
  int c0 = 0, n;
  if (from_cells) {
    // exclusive scan of the level's cell counts in LDS (the list slots are not in use yet): cell i's records go to [coff[i], coff[i + 1])
    int* coff = reinterpret_cast<int*>(oct_lds + (((size_t)a.kcap * 7 + 15) & ~(size_t)15));
    const int i0 = 2 * t, i1 = 2 * t + 1;
    const int v0 = i0 < ncl ? f_cell_counts[Lv.cellBase + i0] : 0, v1 = i1 < ncl ? f_cell_counts[Lv.cellBase + i1] : 0;
    int e0, e1;
    n = oct_scan2<TPB>(v0, v1, e0, e1, wsum);
    if (i0 <= ncl) coff[i0] = e0;
    if (i1 <= ncl) coff[i1] = e1;
    __syncthreads();
  } else {
    c0 = f_cand[Lv.cellBase];
    n = f_cand[Lv.cellBase + ncl] - c0;
  }
  const uint32_t* grec = from_cells ? nullptr : reinterpret_cast<const uint32_t*>(f_cand + a.ncells + 1) + c0;
  // LDS carve-up
  uint32_t* rec = reinterpret_cast<uint32_t*>(oct_lds);                       // [kcap]
  uint16_t* knode = reinterpret_cast<uint16_t*>(rec + a.kcap);                // [kcap]
  uint8_t* kq = reinterpret_cast<uint8_t*>(knode + a.kcap);                   // [kcap]
  size_t o = ((size_t)a.kcap * 7 + 15) & ~(size_t)15;
  OctNode* nodes0 = reinterpret_cast<OctNode*>(oct_lds + o); o += (size_t)Lcap * sizeof(OctNode);
  OctNode* nodes1 = reinterpret_cast<OctNode*>(oct_lds + o); o += (size_t)Lcap * sizeof(OctNode);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(oct_lds + o); o += (size_t)Lcap * 16;     // [Lcap][4] quadrant counts, then child slots / new slot
  int16_t* prank = reinterpret_cast<int16_t*>(oct_lds + o); o += (size_t)Lcap * 2;      // processing rank of a slot, -1 = not split
  int16_t* byrank = reinterpret_cast<int16_t*>(oct_lds + o); o += (size_t)Lcap * 2;     // rank -> children (then creation base)
  o = (o + 15) & ~(size_t)15;
  uint32_t* ckey = reinterpret_cast<uint32_t*>(oct_lds + o); o += (size_t)Lcap * 4;     // expandable children of the last pass: size << 16 | creation index
  uint16_t* cslot = reinterpret_cast<uint16_t*>(oct_lds + o); o += (size_t)Lcap * 2;    //   and their list slot
  uint8_t* mark = reinterpret_cast<uint8_t*>(oct_lds + o);                                  // [Lcap] node takes part in the counting
  int* g_over = f_n_out + 1;                                                                  // [count, overflow flag]
  unsigned long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = f_dbg ? __builtin_amdgcn_s_memtime() : 0;
#define OCT_T(slot) do { if (f_dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tq[slot] += t_ - tl; tl = t_; } } while (0)
  if (t < 16) { sh[t] = 0; rootcnt[t] = 0; }
  __syncthreads();
  const int W = (Lv.w - kEdge + 3) - (kEdge - 3), H = (Lv.h - kEdge + 3) - (kEdge - 3);
  const int nIni = (int)roundf(static_cast<float>(W) / static_cast<float>(H));
  bool overflow = n > 0 && (n > a.kcap || nIni > 15 || nIni < 1);   // (a level without cells has n = 0 and a meaningless box)
  if (!overflow && n > 0) {
    const float hX = static_cast<float>(W) / nIni;
    // roots: bin the candidates (order inside a node never matters, see above)
    if (from_cells) {
      // record k of the level = record k - coff[i] of cell i, i = the last cell whose offset is <= k (binary search over the offsets in LDS); the offsets
      // live where the list slots will: they are copied to registers before the first slot is written
      const int* coff = reinterpret_cast<const int*>(oct_lds + (((size_t)a.kcap * 7 + 15) & ~(size_t)15));
      for (int k0 = t; k0 < n; k0 += 4 * TPB) {   // four global loads in flight per thread
        uint32_t r4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k = k0 + q * TPB;
          r4[q] = 0u;
          if (k < n) {
            int lo = 0, hi = ncl - 1;       // invariant: coff[lo] <= k < coff[hi + 1]
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (coff[mid] <= k) lo = mid; else hi = mid - 1; }
            r4[q] = f_cell_slots[(size_t)(Lv.cellBase + lo) * kCellCap + (k - coff[lo])];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) if (k0 + q * TPB < n) rec[k0 + q * TPB] = r4[q];
      }
    } else
    for (int k0 = t; k0 < n; k0 += 4 * TPB) {   // four global loads in flight per thread
      uint32_t r4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) r4[q] = (k0 + q * TPB < n) ? grec[k0 + q * TPB] : 0u;
#pragma unroll
      for (int q = 0; q < 4; q++) if (k0 + q * TPB < n) rec[k0 + q * TPB] = r4[q];
    }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += TPB) {   // every wave adds its count per root once (thousands of atomics on one or two counters serialise)
      const int k = k0 + t;
      int root = -1;
      if (k < n) { root = (int)(static_cast<float>(rec[k] & 0xFFF) / hX); root = root < 15 ? root : 15; kq[k] = (uint8_t)root; }
      for (int i = 0; i <= nIni && i < 16; i++) {
        const int c = __popcll(__ballot(root == i));
        if ((t & 63) == 0 && c) atomicAdd(&rootcnt[i], c);
      }
    }
    __syncthreads();
    if (t == 0) {
      int ls = 0;
      for (int i = 0; i < nIni; i++) {
        rootslot[i] = -1;
        if (rootcnt[i] == 0) continue;   // empty root: erased (:754-757)
        OctNode nd;
        nd.ulx = (int16_t)(int)(hX * static_cast<float>(i)); nd.urx = (int16_t)(int)(hX * static_cast<float>(i + 1)); nd.uly = 0; nd.bry = (int16_t)H;
        nd.size = (uint16_t)rootcnt[i]; nd.nomore = rootcnt[i] == 1;
        nodes0[ls] = nd; rootslot[i] = ls++;
      }
      sh[0] = ls; sh[1] = 0;
    }
    __syncthreads();
    for (int k = t; k < n; k += TPB) knode[k] = (uint16_t)rootslot[kq[k]];
    __syncthreads();
    OCT_T(0);
    // ---- rounds ----
    bool final_phase = false;
    for (int guard = 0; guard < 64; guard++) {
      const int Ls = sh[0];
      OctNode* cur = sh[1] ? nodes1 : nodes0;
      OctNode* nxt = sh[1] ? nodes0 : nodes1;
      const int m = sh[3];                                   // expandable children of the previous pass (final phase input)
      const int i0 = 2 * t, i1 = 2 * t + 1;
      // 1. which nodes take part in the counting: every expandable node (main loop) / the candidates of the previous pass (final phase)
      if (!final_phase) {
        if (i0 < Ls) mark[i0] = cur[i0].nomore ? 0 : 1;
        if (i1 < Ls) mark[i1] = cur[i1].nomore ? 0 : 1;
      } else {
        if (i0 < Ls) mark[i0] = 0;
        if (i1 < Ls) mark[i1] = 0;
        __syncthreads();
        for (int c = t; c < m; c += TPB) mark[cslot[c]] = 1;
      }
      for (int e = t; e < 4 * Ls; e += TPB) cnt[e] = 0;
      if (t == 0) sh[3] = 0;
      __syncthreads();
      // 2. quadrant of every key of a participating node (DivideNode :650-705), counts per (node, quadrant)
      for (int k = t; k < n; k += TPB) {
        const int nd = knode[k];
        if (!mark[nd]) continue;
        const OctNode Pn = cur[nd];
        const int midx = Pn.ulx + ((Pn.urx - Pn.ulx + 1) >> 1), midy = Pn.uly + ((Pn.bry - Pn.uly + 1) >> 1);   // (int)ceil((float)extent / 2)
        const uint32_t r = rec[k];
        const int x = r & 0xFFF, y = (r >> 12) & 0xFFF;
        const int q = (x < midx) ? ((y < midy) ? 0 : 2) : ((y < midy) ? 1 : 3);
        kq[k] = (uint8_t)q;
        atomicAdd(&cnt[4 * nd + q], 1u);
      }
      __syncthreads();
      OCT_T(1);
      auto nchild = [&](int i) { return (cnt[4 * i] ? 1 : 0) + (cnt[4 * i + 1] ? 1 : 0) + (cnt[4 * i + 2] ? 1 : 0) + (cnt[4 * i + 3] ? 1 : 0); };
      int P, C;                                              // nodes split in this pass, children created
      int kp0 = 0, kp1 = 0;                                  // kept nodes before slot i0 / i1
      if (!final_phase) {
        // 3a. main loop: every expandable node is split, in list order.  One scan of (split flag | children << 12) gives the processing
        // rank, the creation base of the node's children and — slot minus rank — the position of a kept node behind them
        const int f0 = i0 < Ls ? mark[i0] : 0, f1 = i1 < Ls ? mark[i1] : 0;
        const int v0 = f0 ? (1 | (nchild(i0) << 12)) : 0, v1 = f1 ? (1 | (nchild(i1) << 12)) : 0;
        int e0, e1;
        const int tot = oct_scan2<TPB>(v0, v1, e0, e1, wsum);
        P = tot & 0xFFF; C = tot >> 12;
        if (i0 < Ls) { prank[i0] = f0 ? (int16_t)(e0 & 0xFFF) : (int16_t)-1; if (f0) byrank[e0 & 0xFFF] = (int16_t)(e0 >> 12); }
        if (i1 < Ls) { prank[i1] = f1 ? (int16_t)(e1 & 0xFFF) : (int16_t)-1; if (f1) byrank[e1 & 0xFFF] = (int16_t)(e1 >> 12); }
        kp0 = i0 - (e0 & 0xFFF); kp1 = i1 - (e1 & 0xFFF);
        __syncthreads();
      } else {
        // 3b. final phase: processing order = descending (size, creation index); the pass stops with the node that takes the list to N.
        // ASSUMPTION behind the tie-break: the reference sorts pair<int, ExtractorNode*> (ORBextractor.cpp:852), i.e. nodes with equally many keypoints
        // by HEAP ADDRESS, which equals creation order only while list nodes are allocated at growing addresses (true under the monotonic
        // operator new of oracle/_ref/orb_ref_cli, not guaranteed by glibc malloc: DESIGN 2, "The reference is not deterministic").  The oracle and
        // this kernel DEFINE the tie as creation order.
        if (i0 < Ls) prank[i0] = -1;
        if (i1 < Ls) prank[i1] = -1;
        __syncthreads();
        for (int e = m + t; e < ((m + 3) & ~3); e += TPB) ckey[e] = 0;   // pad to whole 16-byte reads (0 is below every key)
        __syncthreads();
        for (int c = t; c < m; c += TPB) {
          const uint32_t key = ckey[c];
          int r = 0;
          for (int o2 = 0; o2 < m; o2 += 4) {   // (size, creation index) compare as one word; four candidates per LDS read
            const uint4 k4 = *reinterpret_cast<const uint4*>(ckey + o2);
            r += (k4.x > key ? 1 : 0) + (k4.y > key ? 1 : 0) + (k4.z > key ? 1 : 0) + (k4.w > key ? 1 : 0);
          }
          const int nd = cslot[c];
          prank[nd] = (int16_t)r;
          byrank[r] = (int16_t)nchild(nd);
        }
        if (t == 0) sh[5] = m;                                // no cut: every candidate is split
        __syncthreads();
        const int n0 = i0 < m ? byrank[i0] : 0, n1 = i1 < m ? byrank[i1] : 0;   // children by rank
        int e0, e1;
        oct_scan2<TPB>(n0, n1, e0, e1, wsum);                        // children created before this rank
        // list size after splitting ranks 0..r = Ls + children - (r + 1); first rank that reaches N
        const int a0 = Ls + e0 + n0 - (i0 + 1), b0 = Ls + e0 - i0;
        const int a1 = Ls + e1 + n1 - (i1 + 1), b1 = Ls + e1 - i1;
        if (i0 < m && a0 >= N && b0 < N) sh[5] = i0 + 1;
        if (i1 < m && a1 >= N && b1 < N) sh[5] = i1 + 1;
        __syncthreads();
        P = sh[5];
        if (i0 < P) byrank[i0] = (int16_t)e0;                 // creation base of the rank's children
        if (i1 < P) byrank[i1] = (int16_t)e1;
        for (int c = t; c < m; c += TPB) { const int nd = cslot[c]; if (prank[nd] >= P) prank[nd] = -1; }
        if (t == 0) sh[2] = 0;
        __syncthreads();
        // children created by the ranks below the cut, and the kept nodes in list order
        int k0, k1;
        const int K = oct_scan2<TPB>((i0 < Ls && prank[i0] < 0) ? 1 : 0, (i1 < Ls && prank[i1] < 0) ? 1 : 0, k0, k1, wsum);
        kp0 = k0; kp1 = k1;
        // C = children of ranks 0..P-1 = base of rank P-1 + its children
        if (P > 0 && i0 == P - 1) sh[2] = e0 + n0;
        if (P > 0 && i1 == P - 1) sh[2] = e1 + n1;
        __syncthreads();
        C = sh[2];
        (void)K;
      }
      OCT_T(final_phase ? 3 : 2);
      const int K = Ls - P;
      if (C + K > Lcap) overflow = true;                      // (cannot happen: a pass at most quadruples a list shorter than N)
      // 5. the new list
      if (!overflow) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
          const int i = w ? i1 : i0;
          if (i >= Ls) continue;
          const OctNode Pn = cur[i];
          if (prank[i] < 0) {
            const int ns = C + (w ? kp1 : kp0);
            nxt[ns] = Pn;
            cnt[4 * i] = (uint32_t)ns;
            continue;
          }
          const int midx = Pn.ulx + ((Pn.urx - Pn.ulx + 1) >> 1), midy = Pn.uly + ((Pn.bry - Pn.uly + 1) >> 1);
          int ci = byrank[prank[i]];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int sz = (int)cnt[4 * i + q];
            if (sz == 0) continue;
            OctNode ch;
            ch.ulx = (q & 1) ? (int16_t)midx : Pn.ulx; ch.urx = (q & 1) ? Pn.urx : (int16_t)midx;
            ch.uly = (q & 2) ? (int16_t)midy : Pn.uly; ch.bry = (q & 2) ? Pn.bry : (int16_t)midy;
            ch.size = (uint16_t)sz; ch.nomore = sz == 1;
            const int ns = C - 1 - ci;                        // pushed to the front: reverse creation order
            nxt[ns] = ch;
            cnt[4 * i + q] = (uint32_t)ns;
            if (sz > 1) { const int e = atomicAdd(&sh[3], 1); cslot[e] = (uint16_t)ns; ckey[e] = ((uint32_t)sz << 16) | (uint32_t)ci; }
            ci++;
          }
        }
      }
      __syncthreads();
      OCT_T(4);
      // 6. keys follow their nodes
      if (!overflow)
        for (int k = t; k < n; k += TPB) {
          const int nd = knode[k];
          knode[k] = (uint16_t)cnt[4 * nd + (prank[nd] >= 0 ? kq[k] : 0)];
        }
      // 7. the reference's loop conditions (:835-845, :901-905)
      const int newLs = C + K, nExp = sh[3];
      const bool done = newLs >= N || newLs == Ls;
      if (!final_phase && !done && newLs + 3 * nExp > N) final_phase = true;
      __syncthreads();
      if (t == 0) { sh[0] = newLs; sh[1] ^= 1; }
      __syncthreads();
      OCT_T(5); tq[7]++;
      if (done || overflow) break;
    }
  }
  // ---- best response per node, in list order ----
  const int Ls = (overflow || n == 0) ? 0 : sh[0];
  if (Ls > 0) {
    OctNode* cur = sh[1] ? nodes1 : nodes0;
    (void)cur;
    for (int e = t; e < Ls; e += TPB) cnt[e] = 0;
    __syncthreads();
    for (int k = t; k < n; k += TPB) atomicMax(&cnt[knode[k]], ((rec[k] >> 24) << 16) | (uint32_t)(0xFFFF - k));   // first maximum = smallest index
    __syncthreads();
    const int minB = kEdge - 3;
    for (int e = t; e < Ls && e < a.stage_stride; e += TPB) {
      const uint32_t r = rec[0xFFFF - (cnt[e] & 0xFFFF)];
      f_stage[(size_t)l * a.stage_stride + e] = KpIn{(int16_t)((int)(r & 0xFFF) + minB), (int16_t)((int)((r >> 12) & 0xFFF) + minB), (int16_t)l, (int16_t)(r >> 24)};
    }
    if (Ls > a.stage_stride) overflow = true;
  }
  // ---- the last level to finish concatenates the levels ----
  OCT_T(6);
  if (f_dbg && l == 0 && t == 0) for (int q = 0; q < 8; q++) f_dbg[q] += tq[q];
#undef OCT_T
  // (round 4) no concatenation here: the level leaves its count and its staged keypoints, the orientation + descriptor kernel behind it maps its waves to
  // (level, entry) through the eight counts — the device-scope fences, the arrival counter and the last workgroup's copy are gone from the frame's longest kernel
  if (t == 0) {
    f_counts[l] = Ls;
    if (overflow) atomicExch(g_over, 1);
  }
}


// Keypoints come either as one list (`kin`, n entries: host octree) or STAGED per level by the device octree (`stage` [level][stage_stride] with `lcounts`: output
// position i belongs to the level whose running count passes it — the concatenation in level order that ComputeKeyPointsOctTree's loop over levels produces); in the
// staged form thread 0 of workgroup 0 also publishes the frame's count (`n_dev`[0], and `count_out` of the batch API), clamped to the caller's capacity.
__global__ __launch_bounds__(256) void orb_orient_desc_kernel(OrbDev d, const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                              const KpIn* __restrict__ kin, int n, int* __restrict__ n_dev /* staged form: receives the count */,
                                                              ccm_keypoint* __restrict__ kout, uint8_t* __restrict__ desc, int* __restrict__ count_out /* nullable */,
                                                              const KpIn* __restrict__ stage, int stage_stride, const int* __restrict__ lcounts, int kp_cap, OrbFrames fr) {
  const int fy = blockIdx.y;
  {
    const long long sh_ = fr.set[fy];
    pyr = orb_shift(pyr, sh_); blur = orb_shift(blur, sh_); kin = orb_shift(kin, sh_); n_dev = orb_shift(n_dev, sh_); stage = orb_shift(stage, sh_); lcounts = orb_shift(lcounts, sh_);
    kout = orb_shift(kout, fr.kout[fy]); desc = orb_shift(desc, fr.desc[fy]); count_out = orb_shift(count_out, fr.cnt[fy]);
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  KpIn kp;
  if (stage) {
    int off = 0, lv = -1, e = 0;
    for (int q = 0; q < d.nlevels; q++) {      // uniform: scalar loads of the counts
      const int c = lcounts[q];
      if (lv < 0 && i < off + c) { lv = q; e = i - off; }
      off += c;
    }
    n = off < kp_cap ? off : kp_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *n_dev = n; if (count_out) *count_out = n; }
    if (i >= n) return;
    kp = stage[(size_t)lv * stage_stride + e];
  } else {
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = n;
    if (i >= n) return;
    kp = kin[i];
  }
  LevelInfo L = d.lv[kp.level];
  if (kp.level == 0) { L.poff = fr.poff0[fy]; L.pstride = fr.pstride0; }   // level 0 of THIS frame
  // IC_Angle (:68-95): integer moments over the radius-15 disc of the UNBLURRED level
  const uint8_t* center = pyr + L.poff + (long long)kp.y * L.pstride + kp.x;
  int m10 = 0, m01 = 0;
  {
    const int u = lane - kHalfPatch;   // lanes 0..30 <-> u = -15..15
    for (int v = -kHalfPatch; v <= kHalfPatch; v++) {
      const int av = v < 0 ? -v : v;
      if (lane <= 2 * kHalfPatch && (u <= d.umax[av] && u >= -d.umax[av])) {
        const int val = center[v * L.pstride + u];
        m10 += u * val; m01 += v * val;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { m10 += __shfl_xor(m10, off, kWave); m01 += __shfl_xor(m01, off, kWave); }
  }
  const float angle = orbm::fast_atan2((float)m01, (float)m10);
  // computeOrbDescriptor (:100-316) on the blurred level
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float ang = angle * factorPI;
  const float a = orbm::cosf_glibc(ang), b = orbm::sinf_glibc(ang);
  const uint8_t* bc = blur + L.off + (size_t)kp.y * L.stride + kp.x;
  unsigned long long words[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pair = j * 64 + lane;
    const float x0 = (float)c_pattern[pair * 4], y0 = (float)c_pattern[pair * 4 + 1];
    const float x1 = (float)c_pattern[pair * 4 + 2], y1 = (float)c_pattern[pair * 4 + 3];
    const int t0 = bc[orbm::cv_round(x0 * b + y0 * a) * L.stride + orbm::cv_round(x0 * a - y0 * b)];
    const int t1 = bc[orbm::cv_round(x1 * b + y1 * a) * L.stride + orbm::cv_round(x1 * a - y1 * b)];
    words[j] = __ballot(t0 < t1);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(desc + (size_t)i * 32)[lane] = words[lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 2 : 3];
  if (lane == 0) {
    ccm_keypoint o;
    // operator() epilogue (:1268-1274): pt *= mvScaleFactor[level] for level != 0
    o.x = kp.level != 0 ? (float)kp.x * L.scale : (float)kp.x;
    o.y = kp.level != 0 ? (float)kp.y * L.scale : (float)kp.y;
    o.size = (float)(int)(kPatch * L.scale);   // scaledPatchSize = PATCH_SIZE*mvScaleFactor[level] truncated to int (:1006)
    o.angle = angle;
    o.response = (float)kp.response;
    o.octave = kp.level;
    kout[i] = o;
  }
}

// =================================================================================================
// host: DistributeOctTree (ORBextractor.cpp:707-931), index-linked list instead of std::list
// =================================================================================================
struct Cand { float x, y, response; };

// Nodes own a contiguous range [kb, ke) of a shared key array (candidate indices); splitting a node is a stable
// 4-way partition of its range, so no per-node allocation happens (the first version kept a std::vector per
// node and spent ~220 us per frame in malloc/free — more than all device work of the frame).
struct ONode {
  int ulx, uly, urx, bry;          // UL.x, UL.y, UR.x, BR.y  (BL.x = UL.x, BL.y = BR.y, UR.y = UL.y, BR.x = UR.x)
  int kb, ke;                      // key range, insertion order preserved
  int prev, next;
  bool noMore;
};

struct Octree {
  std::vector<ONode> nodes;
  std::vector<int> keys, tmp;
  std::vector<std::pair<int, int>> sizeAndNode, prevList;
  int head = -1, tail = -1, count = 0;
  void push_back(int id) { ONode& n = nodes[id]; n.prev = tail; n.next = -1; if (tail >= 0) nodes[tail].next = id; else head = id; tail = id; count++; }
  void push_front(int id) { ONode& n = nodes[id]; n.next = head; n.prev = -1; if (head >= 0) nodes[head].prev = id; else tail = id; head = id; count++; }
  int erase(int id) {
    ONode& n = nodes[id];
    const int nx = n.next;
    if (n.prev >= 0) nodes[n.prev].next = n.next; else head = n.next;
    if (n.next >= 0) nodes[n.next].prev = n.prev; else tail = n.prev;
    count--;
    return nx;
  }
};

// splits node `id` into its four children (n1 UL, n2 UR, n3 BL, n4 BR); returns child ids (or -1 when empty)
static void divide_node(Octree& T, int id, const Cand* c, int child[4]) {
  const ONode P = T.nodes[id];
  const int halfX = (int)std::ceil(static_cast<float>(P.urx - P.ulx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(P.bry - P.uly) / 2);
  const int midx = P.ulx + halfX, midy = P.uly + halfY;
  int cnt[4] = {0, 0, 0, 0};
  int* K = T.keys.data();
  int* Tm = T.tmp.data();
  for (int s = P.kb; s < P.ke; s++) {
    const Cand& kp = c[K[s]];
    const int q = (kp.x < midx) ? ((kp.y < midy) ? 0 : 2) : ((kp.y < midy) ? 1 : 3);
    Tm[s] = q; cnt[q]++;
  }
  int start[4] = {P.kb, P.kb + cnt[0], P.kb + cnt[0] + cnt[1], P.kb + cnt[0] + cnt[1] + cnt[2]};
  // stable scatter through a small stack buffer when the node is small, else a heap scratch (rare: only near the roots)
  {
    const int n = P.ke - P.kb;
    int stackbuf[512];
    std::vector<int> heap;
    int* buf = stackbuf;
    if (n > 512) { heap.resize(n); buf = heap.data(); }
    int pos[4] = {start[0] - P.kb, start[1] - P.kb, start[2] - P.kb, start[3] - P.kb};
    for (int s = P.kb; s < P.ke; s++) buf[pos[Tm[s]]++] = K[s];
    std::memcpy(K + P.kb, buf, sizeof(int) * n);
  }
  const int bx[4][4] = {{P.ulx, P.uly, midx, midy}, {midx, P.uly, P.urx, midy}, {P.ulx, midy, midx, P.bry}, {midx, midy, P.urx, P.bry}};
  for (int q = 0; q < 4; q++) {
    if (cnt[q] == 0) { child[q] = -1; continue; }
    ONode ch;
    ch.ulx = bx[q][0]; ch.uly = bx[q][1]; ch.urx = bx[q][2]; ch.bry = bx[q][3];
    ch.kb = start[q]; ch.ke = start[q] + cnt[q]; ch.prev = ch.next = -1; ch.noMore = cnt[q] == 1;
    child[q] = (int)T.nodes.size();
    T.nodes.push_back(ch);
  }
}

// returns the selected candidate indices in the reference's output order (front-to-back list order)
// `T` is a caller-owned workspace whose vectors keep their capacity from frame to frame
static void distribute_octree(Octree& T, const Cand* c, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& result) {
  T.nodes.clear(); T.head = T.tail = -1; T.count = 0;
  T.nodes.reserve((size_t)std::min(n, 4 * N + 64) * 2 + 16);
  T.keys.resize(n); T.tmp.resize(n);
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  {
    // bin the candidates by root (kp.pt.x / hX), stable
    std::vector<int>& rootOf = T.tmp;
    std::vector<int> cnt(nIni + 1, 0);
    for (int k = 0; k < n; k++) { rootOf[k] = (int)(c[k].x / hX); cnt[rootOf[k] + 1]++; }
    for (int i = 0; i < nIni; i++) cnt[i + 1] += cnt[i];
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (int k = 0; k < n; k++) T.keys[pos[rootOf[k]]++] = k;
    for (int i = 0; i < nIni; i++) {
      ONode r;
      r.ulx = (int)(hX * static_cast<float>(i)); r.urx = (int)(hX * static_cast<float>(i + 1)); r.uly = 0; r.bry = maxY - minY;
      r.kb = cnt[i]; r.ke = cnt[i + 1]; r.prev = r.next = -1; r.noMore = false;
      T.nodes.push_back(r);
      T.push_back(i);
    }
  }
  for (int id = T.head; id >= 0;) {
    ONode& nd = T.nodes[id];
    const int sz = nd.ke - nd.kb;
    if (sz == 1) { nd.noMore = true; id = nd.next; }
    else if (sz == 0) id = T.erase(id);
    else id = nd.next;
  }
  bool finish = false;
  std::vector<std::pair<int, int>>& sizeAndNode = T.sizeAndNode; std::vector<std::pair<int, int>>& prevList = T.prevList;   // (size, node id)
  while (!finish) {
    const int prevSize = T.count;
    int nToExpand = 0;
    sizeAndNode.clear();
    for (int id = T.head; id >= 0;) {
      if (T.nodes[id].noMore) { id = T.nodes[id].next; continue; }
      int child[4];
      divide_node(T, id, c, child);
      for (int q = 0; q < 4; q++) {
        if (child[q] < 0) continue;
        T.push_front(child[q]);
        const int sz = T.nodes[child[q]].ke - T.nodes[child[q]].kb;
        if (sz > 1) { nToExpand++; sizeAndNode.emplace_back(sz, child[q]); }
      }
      id = T.erase(id);
    }
    if (T.count >= N || T.count == prevSize) finish = true;
    else if (T.count + nToExpand * 3 > N) {
      while (!finish) {
        const int prev2 = T.count;
        prevList = sizeAndNode;
        sizeAndNode.clear();
        // reference sorts pair<int,ExtractorNode*>; equal sizes tie on the heap address there — here on
        // creation order (the node id), the tie-break the oracle defines
        std::sort(prevList.begin(), prevList.end());
        for (int j = (int)prevList.size() - 1; j >= 0; j--) {
          const int id = prevList[j].second;
          int child[4];
          divide_node(T, id, c, child);
          for (int q = 0; q < 4; q++) {
            if (child[q] < 0) continue;
            T.push_front(child[q]);
            const int sz = T.nodes[child[q]].ke - T.nodes[child[q]].kb;
            if (sz > 1) sizeAndNode.emplace_back(sz, child[q]);
          }
          T.erase(id);
          if (T.count >= N) break;
        }
        if (T.count >= N || T.count == prev2) finish = true;
      }
    }
  }
  result.clear();
  result.reserve(T.count);
  for (int id = T.head; id >= 0; id = T.nodes[id].next) {
    const ONode& nd = T.nodes[id];
    int best = T.keys[nd.kb];
    float maxResp = c[best].response;
    for (int s = nd.kb + 1; s < nd.ke; s++) { const int k = T.keys[s]; if (c[k].response > maxResp) { best = k; maxResp = c[k].response; } }
    result.push_back(best);
  }
}

}  // namespace

// =================================================================================================
constexpr int kOrbSets = 2 * kOrbGroup;   // buffer sets: two groups of kOrbGroup frames in flight in the batch API with the device octree (sets 0 / 1 also serve the single-frame and host-octree paths)
struct ccm_orb {
  ccm_ctx* ctx = nullptr;
  int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
  float scaleFactor = 1.2f;
  std::vector<float> sf, isf, s2, is2;
  std::vector<int> nfeat;
  int umax[16];
  // geometry-dependent state
  int w = 0, h = 0;
  OrbDev dev{};
  size_t pyr_bytes = 0;
  // everything one frame in flight owns.  Two sets: the single-frame entry points use set 0; ccm_orb_extract_batch_dev alternates, so
  // that frame t+1's device phase 1 runs while the host selects keypoints (DistributeOctTree) for frame t.
  struct Bufs {
    uint8_t* d_block = nullptr;   // every device buffer below is carved from this one block, with the same layout in every set (OrbFrames)
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    uint32_t* d_cell_slots = nullptr; int* d_cell_counts = nullptr; int* d_cand = nullptr;   // d_cand: [ncells+1 offsets][records]
    KpIn* d_kin = nullptr; ccm_keypoint* d_kout = nullptr; uint8_t* d_desc = nullptr;
    int* h_cand = nullptr;   // pinned
    KpIn* h_kin = nullptr;   // pinned
    int* h_count = nullptr;  // pinned: keypoint count of the frame (batch API); device octree: [0] count, [1] overflow flag
    KpIn* d_oct_stage = nullptr; int* d_oct_counts = nullptr; int* d_n = nullptr;   // device octree: per-level results, [levels | arrival | overflow], keypoint count
    hipEvent_t ev_cand = nullptr;
  } B[kOrbSets];
  int cur = 0;
  int16_t* d_tabs = nullptr; std::vector<int> tab_xofs, tab_ialpha, tab_yofs, tab_ibeta;   // per level offsets into d_tabs
  // one-launch pyramid (orb_pyramid_kernel): tile tables [tile column / row][level][4], grid, LDS plan; pyr_fused = false -> one orb_resize_kernel per level
  int4* d_pyr_tcol = nullptr; int4* d_pyr_trow = nullptr; int pyr_ntx = 0, pyr_nty = 0, pyr_buf_bytes = 0; size_t pyr_lds = 0; bool pyr_fused = false;
  uint2* d_pyr_ent = nullptr;   // staged-table entries of every tile column, then of every tile row
  PyrArgs pyr_args{};
  int cand_cap = 0;
  int* d_tile_level = nullptr; int* d_tile_xy = nullptr; int n_blur_tiles = 0;
  int kp_cap = 0;
  uint8_t* h_io = nullptr;  // pinned: the input image on the way in, [keypoints | descriptors] on the way out (one DMA each)
  size_t h_io_bytes = 0;
  double t_phase[6] = {0, 0, 0, 0, 0, 0}; double t_wait_cand = 0;   // host wall clock of the last frame, ms: upload+queue, wait cand, octree, queue2, wait+D2H, total
  Octree tree_ws; std::vector<int> sel_ws;   // reusable host workspaces
  // device octree (orb_octree_kernel): LDS plan of this geometry; oct_ok = false -> host octree
  bool oct_ok = false; int oct_tpb = 1024; int oct_kcap = 0, oct_stride = 0; size_t oct_lds = 0; int oct_lcap[kMaxLevels] = {0};
  bool oct_cells = false;   // the octree kernel reads the per-cell lists itself (no orb_compact_kernel on the device-octree path)
  hipStream_t st = nullptr;        // stream the phase functions queue on (the context's, or stream2 for every other frame of a batch)
  hipStream_t stream2 = nullptr; hipEvent_t ev_a = nullptr, ev_b = nullptr;   // host-octree batch path
  hipStream_t bstream[kOrbSets] = {}; hipEvent_t bev[kOrbSets] = {};   // device-octree batch path: stream / "done" event of sets 1..3
  unsigned long long* d_oct_dbg = nullptr;   // CCM_DBG=orb: phase clocks of the octree kernel, printed by ccm_orb_destroy
  uint8_t* d_score_dbg = nullptr;            // score map of the test hook (orb_debug_level), allocated on its first call: the extraction itself has none since round 4
  bool last_call_read_level0_in_place = false;   // the batch API reads level 0 from the caller's frames: the pyramid buffer then holds no level 0 to score
  // last-frame debug
  std::vector<std::vector<Cand>> last_cand; bool last_cand_valid = false;
};

static void orb_free_bufs(ccm_orb::Bufs& b) {
  hipFree(b.d_block);   // every device buffer of the set
  if (b.h_cand) hipHostFree(b.h_cand);
  if (b.h_kin) hipHostFree(b.h_kin);
  if (b.h_count) hipHostFree(b.h_count);
  hipEvent_t ev = b.ev_cand;
  b = ccm_orb::Bufs();
  b.ev_cand = ev;   // events do not depend on the geometry
}
static void orb_free_geometry(ccm_orb* o) {
  for (int k = 0; k < kOrbSets; k++) orb_free_bufs(o->B[k]);
  hipFree(o->d_tabs); hipFree(o->d_tile_level); hipFree(o->d_tile_xy); hipFree(o->d_pyr_tcol); hipFree(o->d_pyr_trow); hipFree(o->d_pyr_ent);
  o->d_pyr_tcol = o->d_pyr_trow = nullptr; o->d_pyr_ent = nullptr; o->pyr_fused = false;
  if (o->h_io) hipHostFree(o->h_io);
  if (o->d_score_dbg) { hipFree(o->d_score_dbg); o->d_score_dbg = nullptr; }   // (sized for the geometry)
  o->d_tabs = nullptr; o->d_tile_level = o->d_tile_xy = nullptr; o->h_io = nullptr; o->h_io_bytes = 0;
}

extern "C" int ccm_orb_create(ccm_ctx* ctx, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                              int min_th_fast, ccm_orb** out) {
  if (!ctx || !out || nfeatures <= 0 || nlevels <= 0 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || min_th_fast < 1 || ini_th_fast < 1)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_orb_create: bad args");
  ccm_orb* o = new ccm_orb();
  o->ctx = ctx; o->nfeatures = nfeatures; o->nlevels = nlevels; o->iniTh = ini_th_fast; o->minTh = min_th_fast; o->scaleFactor = scale_factor;
  // ORBextractor ctor (:584-638), f32 arithmetic as written there
  o->sf.resize(nlevels); o->s2.resize(nlevels); o->isf.resize(nlevels); o->is2.resize(nlevels); o->nfeat.resize(nlevels);
  o->sf[0] = 1.0f; o->s2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) { o->sf[i] = o->sf[i - 1] * scale_factor; o->s2[i] = o->sf[i] * o->sf[i]; }
  for (int i = 0; i < nlevels; i++) { o->isf[i] = 1.0f / o->sf[i]; o->is2[i] = 1.0f / o->s2[i]; }
  const float factor = 1.0f / scale_factor;
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) { o->nfeat[level] = (int)lrintf(nDesired); sum += o->nfeat[level]; nDesired *= factor; }
  o->nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);
  {
    int v, v0;
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) o->umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) { while (o->umax[v0] == o->umax[v0 + 1]) ++v0; o->umax[v] = v0; ++v0; }
  }
  hipSetDevice(ctx->device);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), kOrbPattern31, 1024);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&o->B[0].ev_cand, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&o->B[1].ev_cand, hipEventDisableTiming);
  if (e != hipSuccess) { delete o; return ccm_set_error(ctx, CCM_E_HIP, std::string("orb: pattern upload: ") + hipGetErrorString(e)); }
  *out = o;
  return CCM_OK;
}

extern "C" void ccm_orb_destroy(ccm_orb* o) {
  if (!o) return;
  if (o->ctx) { hipSetDevice(o->ctx->device); hipStreamSynchronize(o->ctx->stream); }
  if (o->d_oct_dbg) {
    unsigned long long h[8];
    hipMemcpy(h, o->d_oct_dbg, sizeof(h), hipMemcpyDeviceToHost);
    const double nl = (double)std::max<unsigned long long>(h[7], 1);
    fprintf(stderr, "[ccm_orb] octree kernel, level 0, clock ticks: load+roots %llu | per pass (%llu passes): count %.0f scan(main) %.0f rank+scan(final) %.0f new list %.0f remap+loop %.0f | final select %llu\n",
            h[0], h[7], h[1] / nl, h[2] / nl, h[3] / nl, h[4] / nl, h[5] / nl, h[6]);
    unsigned long long hp[24];
    hipMemcpy(hp, o->d_oct_dbg + 8, sizeof(hp), hipMemcpyDeviceToHost);
    if (hp[23]) {
      fprintf(stderr, "[ccm_orb] pyramid kernel, workgroup 0, clock ticks per launch (%llu launches): staging %.1f | levels", hp[23], (double)hp[0] / hp[23]);
      for (int l = 1; l < o->nlevels && l < 20; l++) fprintf(stderr, " %.1f", (double)hp[l] / hp[23]);
      fprintf(stderr, "\n");
    }
    hipFree(o->d_oct_dbg);
  }
  orb_free_geometry(o);
  for (int k = 0; k < 2; k++) if (o->B[k].ev_cand) hipEventDestroy(o->B[k].ev_cand);
  if (o->stream2) { hipStreamSynchronize(o->stream2); hipStreamDestroy(o->stream2); }
  for (int k = 1; k < kOrbSets; k++) { if (o->bstream[k]) { hipStreamSynchronize(o->bstream[k]); hipStreamDestroy(o->bstream[k]); } if (o->bev[k]) hipEventDestroy(o->bev[k]); }
  if (o->ev_a) hipEventDestroy(o->ev_a);
  if (o->ev_b) hipEventDestroy(o->ev_b);
  delete o;
}

extern "C" int ccm_orb_get_table(const ccm_orb* o, int which, float* out, int cap) {
  if (!o || !out || which < 0 || which > 3) return CCM_E_ARG;
  const std::vector<float>& t = which == 0 ? o->sf : which == 1 ? o->isf : which == 2 ? o->s2 : o->is2;
  for (int i = 0; i < o->nlevels && i < cap; i++) out[i] = t[i];
  return CCM_OK;
}
extern "C" int ccm_orb_features_per_level(const ccm_orb* o, int32_t* out, int cap) {
  if (!o || !out) return CCM_E_ARG;
  for (int i = 0; i < o->nlevels && i < cap; i++) out[i] = o->nfeat[i];
  return CCM_OK;
}
extern "C" int ccm_orb_level_size(const ccm_orb* o, int w, int h, int level, int* lw, int* lh) {
  if (!o || level < 0 || level >= o->nlevels || !lw || !lh) return CCM_E_ARG;
  *lw = (int)lrintf((float)w * o->isf[level]);   // cvRound((float)image.cols*scale) (:1285)
  *lh = (int)lrintf((float)h * o->isf[level]);
  return CCM_OK;
}
// Upper bound of a frame's keypoint count.  Per level DistributeOctTree returns at most max(N_l + 3, 4 nIni) nodes: the first pass splits every root
// (nIni = round(W / H) of them) before any count is checked (:759-843), later passes stop within 3 of N_l (:901-905).  nIni <= 16 is assumed here
// (levels up to 16.5 times as wide as high); wider levels are truncated at this capacity.
extern "C" int ccm_orb_max_keypoints(const ccm_orb* o) { return o ? o->nfeatures + 67 * o->nlevels : 0; }

// the buffers one frame in flight owns (set k); geometry (pyr_bytes, cand_cap, kp_cap, ncells) must be known
static int orb_alloc_bufs(ccm_orb* o, int k) {
  ccm_ctx* ctx = o->ctx;
  ccm_orb::Bufs& b = o->B[k];
  if (b.d_pyr) return CCM_OK;
  const OrbDev& d = o->dev;
  // one block per set, the same layout in every set: a launch over several frames reaches frame f's buffers by adding ONE byte offset to the pointers of the group's first set
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += ccm_align256(std::max<size_t>(bytes, 1)); return at; };
  const size_t o_pyr = take(o->pyr_bytes), o_blur = take(o->pyr_bytes);
  const size_t o_slots = take(std::max<size_t>(o->cand_cap, 1) * sizeof(uint32_t)), o_counts = take(std::max<size_t>(d.ncells, 1) * sizeof(int));
  const size_t o_cand = take(((size_t)d.ncells + 1 + o->cand_cap) * sizeof(int)), o_kin = take((size_t)o->kp_cap * sizeof(KpIn));
  const size_t o_stage = take(o->oct_ok ? (size_t)o->nlevels * o->oct_stride * sizeof(KpIn) : 1), o_octc = take((size_t)(o->nlevels + 2) * sizeof(int)), o_n = take(2 * sizeof(int));
  const size_t o_d = ccm_align256((size_t)o->kp_cap * sizeof(ccm_keypoint));
  const size_t o_kout = take(o_d + (size_t)o->kp_cap * 32 + 256);   // keypoints and descriptors in one piece [kout | desc] so that the results leave with one copy
  CCM_HIP_CHECK(ctx, hipMalloc(&b.d_block, off));
  CCM_HIP_CHECK(ctx, hipMemsetAsync(b.d_block, 0, off, ctx->stream));
  uint8_t* base = b.d_block;
  b.d_pyr = base + o_pyr; b.d_blur = base + o_blur;
  b.d_cell_slots = reinterpret_cast<uint32_t*>(base + o_slots); b.d_cell_counts = reinterpret_cast<int*>(base + o_counts); b.d_cand = reinterpret_cast<int*>(base + o_cand);
  b.d_kin = reinterpret_cast<KpIn*>(base + o_kin);
  if (o->oct_ok) { b.d_oct_stage = reinterpret_cast<KpIn*>(base + o_stage); b.d_oct_counts = reinterpret_cast<int*>(base + o_octc); b.d_n = reinterpret_cast<int*>(base + o_n); }
  b.d_kout = reinterpret_cast<ccm_keypoint*>(base + o_kout);
  b.d_desc = base + o_kout + o_d;
  CCM_HIP_CHECK(ctx, hipHostMalloc((void**)&b.h_cand, ((size_t)d.ncells + 1 + o->cand_cap) * sizeof(int), hipHostMallocDefault));
  CCM_HIP_CHECK(ctx, hipHostMalloc((void**)&b.h_kin, (size_t)o->kp_cap * sizeof(KpIn), hipHostMallocDefault));
  CCM_HIP_CHECK(ctx, hipHostMalloc((void**)&b.h_count, 64, hipHostMallocDefault));
  return CCM_OK;
}

// (re)build everything that depends on the image size
static int orb_prepare(ccm_orb* o, int w, int h) {
  ccm_ctx* ctx = o->ctx;
  if (o->w == w && o->h == h && o->B[0].d_pyr) return CCM_OK;
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  orb_free_geometry(o);
  OrbDev& d = o->dev;
  d.nlevels = o->nlevels;
  for (int i = 0; i < 16; i++) d.umax[i] = o->umax[i];
  int off = 0, cellBase = 0, rowBase = 0, maxW = 0;
  std::vector<int16_t> tabs;
  o->tab_xofs.assign(o->nlevels, 0); o->tab_ialpha.assign(o->nlevels, 0); o->tab_yofs.assign(o->nlevels, 0); o->tab_ibeta.assign(o->nlevels, 0);
  std::vector<int> tile_level, tile_xy;
  for (int l = 0; l < o->nlevels; l++) {
    LevelInfo& L = d.lv[l];
    ccm_orb_level_size(o, w, h, l, &L.w, &L.h);
    if (L.w < 1 || L.h < 1) return ccm_set_error(ctx, CCM_E_ARG, "orb: a pyramid level is empty (cv::resize rejects an empty size, ORBextractor.cpp:1293)");
    if (L.w > 4000 || L.h > 4000) return ccm_set_error(ctx, CCM_E_ARG, "orb: image too large (packed candidate coordinates are 12 bit)");
    L.stride = (L.w + 63) & ~63;
    L.off = off; off += L.stride * L.h;
    L.poff = L.off; L.pstride = L.stride;
    L.rowBase = rowBase; rowBase += L.h;
    L.scale = o->sf[l];
    maxW = std::max(maxW, L.w);
    // cell grid (:949-955), f32 arithmetic as in the reference
    const float width = (float)((L.w - kEdge + 3) - (kEdge - 3)), height = (float)((L.h - kEdge + 3) - (kEdge - 3));
    L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
    // A level smaller than one cell: the reference's cell loops (`for i < nRows`, `for j < nCols`, :957-998) do not run, so the level contributes no
    // keypoint; its image is still part of mvImagePyramid.  Level sizes only shrink, so such levels are a suffix of the pyramid.
    const bool live = L.nCols > 0 && L.nRows > 0;
    if (!live) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 0; }
    else {
      L.wCell = (int)std::ceil(width / L.nCols); L.hCell = (int)std::ceil(height / L.nRows);
      if (L.wCell > kCellMax || L.hCell > kCellMax) return ccm_set_error(ctx, CCM_E_STATE, "orb: cell larger than the LDS tile");
      // DistributeOctTree (:711-716): nIni = round(W / H) roots; 0 roots (H > 2 W) is a division by zero in the reference
      if ((int)std::round(width / height) < 1) return ccm_set_error(ctx, CCM_E_ARG, "orb: a level is more than twice as high as wide (the reference's DistributeOctTree divides by nIni = 0)");
    }
    L.cellBase = cellBase; cellBase += L.nCols * L.nRows;
    // resize tables for level l (from level l-1)
    if (l > 0) {
      const LevelInfo& P = d.lv[l - 1];
      const double scale_x = 1. / ((double)L.w / P.w), scale_y = 1. / ((double)L.h / P.h);
      o->tab_xofs[l] = (int)tabs.size();
      for (int dx = 0; dx < L.w; dx++) { float fx = (float)((dx + 0.5) * scale_x - 0.5); int sx = (int)std::floor(fx); if (sx < 0) sx = 0; if (sx >= P.w - 1) sx = P.w - 1; tabs.push_back((int16_t)sx); }
      o->tab_ialpha[l] = (int)tabs.size();
      for (int dx = 0; dx < L.w; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5); int sx = (int)std::floor(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= P.w - 1) { fx = 0; sx = P.w - 1; }
        tabs.push_back((int16_t)std::min(std::max((int)lrintf((1.f - fx) * 2048.f), -32768), 32767));
        tabs.push_back((int16_t)std::min(std::max((int)lrintf(fx * 2048.f), -32768), 32767));
      }
      o->tab_yofs[l] = (int)tabs.size();
      for (int dy = 0; dy < L.h; dy++) { float fy = (float)((dy + 0.5) * scale_y - 0.5); tabs.push_back((int16_t)(int)std::floor(fy)); }
      o->tab_ibeta[l] = (int)tabs.size();
      for (int dy = 0; dy < L.h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5); int sy = (int)std::floor(fy); fy -= sy;
        tabs.push_back((int16_t)std::min(std::max((int)lrintf((1.f - fy) * 2048.f), -32768), 32767));
        tabs.push_back((int16_t)std::min(std::max((int)lrintf(fy * 2048.f), -32768), 32767));
      }
      if (tabs.size() & 1) tabs.push_back(0);
    }
    if (live)   // the blurred level is only read around keypoints
      for (int by = 0; by < L.h; by += kBlurTH)
        for (int bx = 0; bx < L.w; bx += kBlurTW) { tile_level.push_back(l); tile_xy.push_back(bx); tile_xy.push_back(by); }
  }
  d.ncells = cellBase; d.totalRows = rowBase; d.maxW = maxW;
  o->pyr_bytes = (size_t)off + 4096;
  o->n_blur_tiles = (int)tile_level.size();
  o->cand_cap = d.ncells * kCellCap;
  o->kp_cap = ccm_orb_max_keypoints(o);
  o->cur = 0;
  {   // LDS plan of the device octree: 52 B per list slot (4 N + 16 slots of the level with most features), the rest for candidates at 7 B each
    static const bool host_oct = getenv("CCM_ORB_HOST_OCTREE") && atoi(getenv("CCM_ORB_HOST_OCTREE")) != 0;
    int nmax = 0;
    for (int l = 0; l < o->nlevels; l++) { o->oct_lcap[l] = std::max(4 * o->nfeat[l] + 16, 64); nmax = std::max(nmax, o->nfeat[l]); }   // 64: the first pass turns up to 15 roots into 60 nodes whatever N is
    const size_t slots = (size_t)std::max(4 * nmax + 16, 64) * (2 * sizeof(OctNode) + 16 + 5 * 2 + 1) + 64;
    const size_t budget = 150 * 1024;
    o->oct_stride = nmax + 64;
    o->oct_kcap = slots < budget ? (int)std::min<size_t>((budget - slots) / 7, 65000) & ~15 : 0;
    o->oct_ok = !host_oct && o->oct_kcap >= 2048 && 4 * nmax + 16 <= 2 * kOctTPB;
    o->oct_tpb = (4 * nmax + 16 <= 1024) ? 512 : 1024;
    if (getenv("CCM_ORB_OCT_KCAP")) o->oct_kcap = std::min(o->oct_kcap, std::max(256, atoi(getenv("CCM_ORB_OCT_KCAP"))) & ~15);   // tests: force the host fallback
    o->oct_lds = ((size_t)o->oct_kcap * 7 + 15) / 16 * 16 + slots;
    {
      int max_cells = 0;
      for (int l = 0; l < o->nlevels; l++) max_cells = std::max(max_cells, d.lv[l].nCols * d.lv[l].nRows);
      o->oct_cells = o->oct_ok && max_cells <= 2 * o->oct_tpb && ((size_t)max_cells + 1) * sizeof(int) <= slots;
    }
  }
  o->pyr_fused = false;
  if (o->nlevels > 1 && !(getenv("CCM_ORB_PYR_FUSED") && atoi(getenv("CCM_ORB_PYR_FUSED")) == 0)) {
    // tile tables of orb_pyramid_kernel.  One axis at a time: size[l], offset table of level l (clamped to the source as the kernels clamp it), tile size.
    int max_ent = 0;
    std::vector<int16_t> ent;   // 4 x int16 per computed column / row of every level, tile column by tile column, then tile row by tile row
    std::vector<size_t> blk_start; int max_src = 0;
    bool plan_ok = true;        // the plan is re-checked below against what the kernel assumes; a plan that fails falls back to one launch per level
    auto axis = [&](bool is_x, int tile, std::vector<int>& out) -> int {   // returns the largest extent of a computed range over tiles and levels, fills out[tile][level][4]
      const int nl = o->nlevels;
      std::vector<int> size(nl);
      for (int l = 0; l < nl; l++) size[l] = is_x ? d.lv[l].w : d.lv[l].h;
      auto ofs = [&](int l, int i) { const int v = tabs[(is_x ? o->tab_xofs[l] : o->tab_yofs[l]) + i]; return std::min(std::max(v, 0), size[l - 1] - 1); };
      const int nt = ccm_div_up(size[1], tile);
      out.assign((size_t)4 * nt * nl, 0);
      int max_ext = 1;
      std::vector<int> a0(nl), a1(nl), c1(nl), own_end(nl, 0);
      for (int tI = 0; tI < nt; tI++) {
        a0[1] = tI * tile; a1[1] = std::min((tI + 1) * tile, size[1]);
        for (int l = 2; l < nl; l++) {   // owned range: the pixels whose first source pixel is owned one level up (the offset tables are non-decreasing)
          int lo = 0; while (lo < size[l] && ofs(l, lo) < a0[l - 1]) lo++;
          int hi = lo; while (hi < size[l] && ofs(l, hi) < a1[l - 1]) hi++;
          a0[l] = lo; a1[l] = hi;
        }
        // computed range, from the last level up: owned + what the next level's computed range reads (its last pixel's first source + 1)
        c1[nl - 1] = a1[nl - 1];
        for (int l = nl - 1; l >= 1; l--) {
          const int src_end = c1[l] > a0[l] ? std::min(ofs(l, c1[l] - 1) + 2, size[l - 1]) : 0;
          if (l - 1 >= 1) c1[l - 1] = std::max(a1[l - 1], src_end);
          else { a0[0] = c1[1] > a0[1] ? ofs(1, a0[1]) : 0; a1[0] = a0[0]; c1[0] = std::max(src_end, a0[0]); }
        }
        int n_ent = 0;
        for (int l = 0; l < nl; l++) {
          int* e = out.data() + 4 * ((size_t)tI * nl + l);
          e[0] = a0[l]; e[1] = a1[l]; e[2] = c1[l];
          if (l >= 1) { e[3] = n_ent; n_ent += c1[l] - a0[l]; }
          max_ext = std::max(max_ext, c1[l] - a0[l]);
        }
        max_ent = std::max(max_ent, n_ent);
        blk_start.push_back(ent.size() / 4);
        max_src = std::max(max_src, c1[0] - a0[0]);
        // what the kernel relies on: (1) the owned ranges of consecutive tiles tile every level without gap or overlap (checked against `own_end`), (2) every computed
        // pixel's two sources lie inside the previous level's computed range, (3) the level-0 origin is the offset-table formula the kernel evaluates, (4) int16 entries
        for (int l = 1; l < nl; l++) {
          if (a0[l] != own_end[l] || a1[l] < a0[l] || c1[l] < a1[l] || c1[l] > size[l]) plan_ok = false;
          own_end[l] = a1[l];
          for (int i = a0[l]; i < c1[l]; i++) {
            const int v = tabs[(is_x ? o->tab_xofs[l] : o->tab_yofs[l]) + i];
            const int s0 = std::min(std::max(v, 0), size[l - 1] - 1), s1 = std::min(std::max(v + 1, 0), size[l - 1] - 1);
            if (s0 < a0[l - 1] || s1 >= c1[l - 1] || s1 - a0[l - 1] > 32767) plan_ok = false;
          }
        }
        {
          const double sc = 1. / ((double)size[1] / size[0]);
          const float f0 = (float)((tI * tile + 0.5) * sc - 0.5);
          if (std::min(std::max((int)std::floor(f0), 0), size[0] - 1) != a0[0]) plan_ok = false;
        }
        for (int l = 1; l < nl; l++)
          for (int i = a0[l]; i < c1[l]; i++) {
            const int v = tabs[(is_x ? o->tab_xofs[l] : o->tab_yofs[l]) + i], wo = (is_x ? o->tab_ialpha[l] : o->tab_ibeta[l]) + 2 * i;
            const int s0 = std::min(std::max(v, 0), size[l - 1] - 1), s1 = std::min(std::max(v + 1, 0), size[l - 1] - 1);
            ent.push_back((int16_t)(s0 - a0[l - 1])); ent.push_back((int16_t)(s1 - a0[l - 1])); ent.push_back(tabs[wo]); ent.push_back(tabs[wo + 1]);
          }
      }
      for (int l = 1; l < nl; l++) if (own_end[l] != size[l]) plan_ok = false;
      return max_ext;
    };
    std::vector<int> tcol, trow;
    const int ex = axis(true, kPyrTW, tcol);
    const int ent_x = max_ent, src_x = max_src; max_ent = 0; max_src = 0;
    const size_t n_xblk = blk_start.size();
    const int ey = axis(false, kPyrTH, trow);
    const int ent_y = max_ent, src_y = max_src;
    o->pyr_ntx = ccm_div_up(d.lv[1].w, kPyrTW); o->pyr_nty = ccm_div_up(d.lv[1].h, kPyrTH);
    o->pyr_buf_bytes = (ex * ey + 15) & ~15;
    // staged tables: 8 bytes per computed column / row of every level
    o->pyr_lds = 2 * (size_t)o->pyr_buf_bytes + 8 * (size_t)(ent_x + ent_y) + 64;
    if (o->pyr_lds <= 60 * 1024 && plan_ok) {
      // entry blocks at a fixed stride per tile column / row (the kernel's staging loads must not depend on a loaded offset)
      std::vector<int16_t> entp((size_t)4 * (n_xblk * ent_x + (blk_start.size() - n_xblk) * ent_y), 0);
      blk_start.push_back(ent.size() / 4);
      for (size_t bI = 0; bI + 1 < blk_start.size(); bI++) {
        const size_t dst = bI < n_xblk ? bI * ent_x : n_xblk * ent_x + (bI - n_xblk) * ent_y;
        std::copy(ent.begin() + 4 * blk_start[bI], ent.begin() + 4 * blk_start[bI + 1], entp.begin() + 4 * dst);
      }
      PyrArgs& pa = o->pyr_args;
      pa.ntx = o->pyr_ntx; pa.buf_bytes = o->pyr_buf_bytes; pa.ent_sx = ent_x; pa.ent_sy = ent_y; pa.ent_y0 = (int)(n_xblk * ent_x);
      pa.ex0 = src_x; pa.ey0 = src_y;
      pa.scale_x = 1. / ((double)d.lv[1].w / d.lv[0].w); pa.scale_y = 1. / ((double)d.lv[1].h / d.lv[0].h);
      pa.dbg = nullptr;
      CCM_HIP_CHECK(ctx, hipMalloc(&o->d_pyr_ent, std::max<size_t>(entp.size(), 4) * sizeof(int16_t)));
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_pyr_ent, entp.data(), entp.size() * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
      CCM_HIP_CHECK(ctx, hipMalloc(&o->d_pyr_tcol, tcol.size() * sizeof(int)));
      CCM_HIP_CHECK(ctx, hipMalloc(&o->d_pyr_trow, trow.size() * sizeof(int)));
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_pyr_tcol, tcol.data(), tcol.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_pyr_trow, trow.data(), trow.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // tcol / trow / entp are locals
      pa.tcol = o->d_pyr_tcol; pa.trow = o->d_pyr_trow; pa.ent = o->d_pyr_ent;
      o->pyr_fused = true;
    }
  }
  if (ccm_dbg("orb")) fprintf(stderr, "[ccm_orb] %d x %d, %d levels: pyramid %s\n", w, h, o->nlevels, o->pyr_fused ? "in one launch" : "one launch per level");
  if (ccm_dbg("orb") && !o->d_oct_dbg) { CCM_HIP_CHECK(ctx, hipMalloc(&o->d_oct_dbg, 256)); CCM_HIP_CHECK(ctx, hipMemset(o->d_oct_dbg, 0, 256)); }
  if (int rc = orb_alloc_bufs(o, 0)) return rc;
  CCM_HIP_CHECK(ctx, hipMalloc(&o->d_tabs, std::max<size_t>(tabs.size(), 2) * sizeof(int16_t)));
  if (!tabs.empty()) CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_tabs, tabs.data(), tabs.size() * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMalloc(&o->d_tile_level, std::max<size_t>(tile_level.size(), 1) * sizeof(int)));
  CCM_HIP_CHECK(ctx, hipMalloc(&o->d_tile_xy, std::max<size_t>(tile_xy.size(), 2) * sizeof(int)));
  if (!tile_level.empty()) {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_tile_level, tile_level.data(), tile_level.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->d_tile_xy, tile_xy.data(), tile_xy.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  }
  {
    const size_t o_d = ccm_align256((size_t)o->kp_cap * sizeof(ccm_keypoint));
    o->h_io_bytes = std::max((size_t)w * h, o_d + (size_t)o->kp_cap * 32) + 256;
    CCM_HIP_CHECK(ctx, hipHostMalloc((void**)&o->h_io, o->h_io_bytes, hipHostMallocDefault));
  }
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  o->w = w; o->h = h;
  return CCM_OK;
}

// the single-frame form of OrbFrames: the buffers are those of the pointer arguments, level 0 lies in the pyramid buffer
static OrbFrames orb_one_frame(const ccm_orb* o) {
  OrbFrames fr;
  memset(&fr, 0, sizeof(fr));
  fr.n = 1; fr.pstride0 = o->dev.lv[0].pstride; fr.poff0[0] = o->dev.lv[0].poff;
  return fr;
}

// device phase 1: pyramid, scores, cells, compaction; blur is queued too (it does not depend on the octree)
// fr != nullptr: a group of frames in one launch per kernel (batch API, device octree): o->cur is the group's FIRST buffer set
static int orb_phase1(ccm_orb* o, bool copy_cand = true, bool dev_octree = false, const OrbFrames* frp = nullptr) {
  ccm_ctx* ctx = o->ctx;
  const OrbDev& d = o->dev;
  const OrbFrames fr = frp ? *frp : orb_one_frame(o);
  if (o->pyr_fused) {
    ccm_prof_scope ps(ctx, CCM_K_PYR_RESIZE, o->st);
    PyrArgs pa = o->pyr_args;
    pa.dbg = o->d_oct_dbg ? o->d_oct_dbg + 8 : nullptr;
    hipLaunchKernelGGL(orb_pyramid_kernel, dim3(o->pyr_ntx * o->pyr_nty, fr.n), dim3(kPyrTPB), o->pyr_lds, o->st, d, o->B[o->cur].d_pyr, pa, fr);
  } else for (int fI = 0; fI < fr.n; fI++) for (int l = 1; l < o->nlevels; l++) {   // (one launch per frame and level)
    const LevelInfo &P = d.lv[l - 1], &L = d.lv[l];
    uint8_t* pyr_f = o->B[o->cur].d_pyr + fr.set[fI];
    ccm_prof_scope ps(ctx, CCM_K_PYR_RESIZE, o->st);
    hipLaunchKernelGGL(orb_resize_kernel, dim3(ccm_div_up(L.w, 256), L.h), dim3(256), 0, o->st, pyr_f + (l == 1 ? fr.poff0[fI] : (long long)P.off), P.w, P.h, l == 1 ? fr.pstride0 : P.stride,
                       pyr_f + L.off, L.w, L.h, L.stride, o->d_tabs + o->tab_xofs[l], o->d_tabs + o->tab_ialpha[l],
                       o->d_tabs + o->tab_yofs[l], o->d_tabs + o->tab_ibeta[l]);
  }
  if (d.ncells == 0) {   // no level holds a cell (image below 62 px): the candidate list is empty, the pyramid above is the whole result
    for (int fI = 0; fI < fr.n; fI++) CCM_HIP_CHECK(ctx, hipMemsetAsync(reinterpret_cast<uint8_t*>(o->B[o->cur].d_cand) + fr.set[fI], 0, sizeof(int), o->st));
  } else {
    ccm_prof_scope ps(ctx, CCM_K_FAST_NMS, o->st);
    hipLaunchKernelGGL(orb_cells_kernel, dim3(d.ncells, fr.n), dim3(256), 0, o->st, d, o->B[o->cur].d_pyr, o->iniTh, o->minTh, o->B[o->cur].d_cell_slots, o->B[o->cur].d_cell_counts, fr);
    if (!(dev_octree && o->oct_cells))   // (the octree kernel gathers from the cell lists itself; the compacted form is only made when the host or a test asks for it)
      for (int fI = 0; fI < fr.n; fI++) {
        ccm_orb::Bufs& bf = o->B[o->cur + fI];   // (the sets of a group are consecutive)
        hipLaunchKernelGGL(orb_compact_kernel, dim3(d.ncells), dim3(256), 0, o->st, d.ncells, bf.d_cell_slots, bf.d_cell_counts, bf.d_cand, (uint32_t*)(bf.d_cand + d.ncells + 1));
      }
  }
  if (copy_cand) {
    const size_t first = ((size_t)d.ncells + 1 + std::min(o->cand_cap, kCandFirstCopy)) * sizeof(int);
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->B[o->cur].h_cand, o->B[o->cur].d_cand, first, hipMemcpyDeviceToHost, o->st));
    CCM_HIP_CHECK(ctx, hipEventRecord(o->B[o->cur].ev_cand, o->st));
  }
  if (o->n_blur_tiles > 0 && !dev_octree) {   // (device octree: the blur tiles ride in the octree kernel's launch, orb_phase2_dev)
    ccm_prof_scope ps(ctx, CCM_K_BLUR, o->st);
    hipLaunchKernelGGL(orb_blur_kernel, dim3(o->n_blur_tiles), dim3(256), 0, o->st, d, o->B[o->cur].d_pyr, o->B[o->cur].d_blur, o->d_tile_level, o->d_tile_xy);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// host phase: read candidates, run the octree per level, fill h_kin; returns the keypoint count.
// The levels are independent (DistributeOctTree is called once per level, ORBextractor.cpp:1027-1040), so a batch call spreads them over a few
// helper threads that live for the duration of the call (LevelPool); the single-frame path runs them in order on the calling thread.
static void orb_select_level(ccm_orb* o, const int* offs, const uint32_t* rec, int l, Octree& ws, std::vector<int>& sel) {
  const LevelInfo& L = o->dev.lv[l];
  const int c0 = offs[L.cellBase], c1 = offs[L.cellBase + L.nCols * L.nRows];
  std::vector<Cand>& cand = o->last_cand[l];
  cand.resize(c1 - c0);
  for (int k = c0; k < c1; k++) {
    const uint32_t r = rec[k];
    cand[k - c0] = Cand{(float)(r & 0xFFF), (float)((r >> 12) & 0xFFF), (float)(r >> 24)};
  }
  sel.clear();
  if (cand.empty()) return;
  const int minB = kEdge - 3;
  distribute_octree(ws, cand.data(), (int)cand.size(), minB, L.w - kEdge + 3, minB, L.h - kEdge + 3, o->nfeat[l], sel);
}

// helper threads of one batch call: they spin on a generation counter (a frame's levels arrive every ~0.1 ms; a condition variable's
// wake-up would cost as much as the work) and take levels from a shared counter, largest level first
struct LevelPool {
  ccm_orb* o = nullptr;
  const int* offs = nullptr; const uint32_t* rec = nullptr;
  std::vector<std::thread> th;
  std::vector<Octree> ws; std::vector<std::vector<int>> sel;   // per level
  std::atomic<int> gen{0}, next{0}, done{0};
  std::atomic<bool> stop{false};
  void work() {
    const int nl = o->nlevels;
    for (int l; (l = next.fetch_add(1, std::memory_order_acq_rel)) < nl;) {   // acquire: a helper that is late leaving the previous frame may pick up this one's first level
      orb_select_level(o, offs, rec, l, ws[l], sel[l]);
      done.fetch_add(1, std::memory_order_release);
    }
  }
  void start(ccm_orb* o_, int helpers) {
    o = o_; ws.resize(o->nlevels); sel.resize(o->nlevels);
    for (int i = 0; i < helpers; i++)
      th.emplace_back([this] {
        int seen = 0;
        while (!stop.load(std::memory_order_acquire)) {
          const int g = gen.load(std::memory_order_acquire);
          if (g == seen) { __builtin_ia32_pause(); continue; }
          seen = g;
          work();
        }
      });
  }
  void run(const int* offs_, const uint32_t* rec_) {   // the caller takes levels too and returns when all are done
    offs = offs_; rec = rec_;
    done.store(0, std::memory_order_relaxed); next.store(0, std::memory_order_release);
    gen.fetch_add(1, std::memory_order_release);
    work();
    while (done.load(std::memory_order_acquire) < o->nlevels) __builtin_ia32_pause();
  }
  ~LevelPool() { stop.store(true, std::memory_order_release); for (auto& t : th) t.join(); }
};

static int orb_host_select(ccm_orb* o, int* n_out, LevelPool* pool = nullptr) {
  ccm_ctx* ctx = o->ctx;
  const OrbDev& d = o->dev;
  // wait for the candidate copy only (the blur kernel queued behind it keeps running)
  {
    const double w0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    CCM_HIP_CHECK(ctx, hipEventSynchronize(o->B[o->cur].ev_cand));
    o->t_wait_cand = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - w0;
  }
  const int* offs = o->B[o->cur].h_cand;
  const int total = offs[d.ncells];
  if (total > kCandFirstCopy) {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->B[o->cur].h_cand + d.ncells + 1 + kCandFirstCopy, o->B[o->cur].d_cand + d.ncells + 1 + kCandFirstCopy,
                                      (size_t)(total - kCandFirstCopy) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  const uint32_t* rec = (const uint32_t*)(o->B[o->cur].h_cand + d.ncells + 1);
  o->last_cand.resize(o->nlevels); o->last_cand_valid = true;
  if (pool) pool->run(offs, rec);
  const int minB = kEdge - 3;
  int n = 0;
  for (int l = 0; l < o->nlevels; l++) {
    if (!pool) orb_select_level(o, offs, rec, l, o->tree_ws, o->sel_ws);
    const std::vector<int>& sel = pool ? pool->sel[l] : o->sel_ws;
    const std::vector<Cand>& cand = o->last_cand[l];
    for (int id : sel) {
      if (n >= o->kp_cap) break;
      o->B[o->cur].h_kin[n++] = KpIn{(int16_t)((int)cand[id].x + minB), (int16_t)((int)cand[id].y + minB), (int16_t)l, (int16_t)cand[id].response};
    }
  }
  *n_out = n;
  return CCM_OK;
}

static int orb_phase2(ccm_orb* o, int n) {
  ccm_ctx* ctx = o->ctx;
  if (n == 0) return CCM_OK;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->B[o->cur].d_kin, o->B[o->cur].h_kin, (size_t)n * sizeof(KpIn), hipMemcpyHostToDevice, ctx->stream));
  {
    ccm_prof_scope ps(ctx, CCM_K_BRIEF);
    hipLaunchKernelGGL(orb_orient_desc_kernel, dim3(ccm_div_up(n, 4)), dim3(256), 0, ctx->stream, o->dev, o->B[o->cur].d_pyr, o->B[o->cur].d_blur, o->B[o->cur].d_kin, n, (int*)nullptr, o->B[o->cur].d_kout, o->B[o->cur].d_desc, (int*)nullptr,
                       (const KpIn*)nullptr, 0, (const int*)nullptr, o->kp_cap, orb_one_frame(o));
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// keypoint selection and phase 2 without the host: octree kernel (one workgroup per level), then orientation + descriptors for the count the
// device wrote (the grid covers the capacity, surplus waves leave at once)
static int orb_phase2_dev(ccm_orb* o, int out_cap, ccm_keypoint* kout = nullptr, uint8_t* dout = nullptr, int* count_out = nullptr, const OrbFrames* frp = nullptr) {
  ccm_ctx* ctx = o->ctx;
  ccm_orb::Bufs& b = o->B[o->cur];
  const OrbFrames fr = frp ? *frp : orb_one_frame(o);
  OctArgs a;
  a.cand = b.d_cand; a.ncells = o->dev.ncells; a.nlevels = o->nlevels;
  for (int l = 0; l < o->nlevels; l++) { a.nfeat[l] = o->nfeat[l]; a.lcap[l] = o->oct_lcap[l]; }
  a.kcap = o->oct_kcap; a.stage = b.d_oct_stage; a.stage_stride = o->oct_stride; a.counts = b.d_oct_counts;
  a.kin = b.d_kin; a.n_out = b.d_n; a.kp_cap = std::min(o->kp_cap, out_cap);
  a.dbg = o->d_oct_dbg;
  a.cell_slots = o->oct_cells ? b.d_cell_slots : nullptr; a.cell_counts = o->oct_cells ? b.d_cell_counts : nullptr;
  const bool split_blur = fr.n > 1 && o->n_blur_tiles > 0;   // a group of frames: the blur tiles get their own launch (see orb_blur_frames_kernel)
  a.pyr = b.d_pyr; a.blur = b.d_blur; a.tile_level = o->d_tile_level; a.tile_xy = o->d_tile_xy; a.n_blur_tiles = split_blur ? 0 : o->n_blur_tiles;
  if (split_blur) {
    ccm_prof_scope ps(ctx, CCM_K_BLUR, o->st);
    hipLaunchKernelGGL(orb_blur_frames_kernel, dim3(o->n_blur_tiles, fr.n), dim3(256), 0, o->st, o->dev, b.d_pyr, b.d_blur, o->d_tile_level, o->d_tile_xy, fr);
  }
  {
    ccm_prof_scope ps(ctx, CCM_K_FAST_NMS, o->st);
    const int blur_wgs = split_blur ? 0 : ccm_div_up(o->n_blur_tiles, o->oct_tpb / 256);
    if (o->oct_tpb == 512) {   // two list slots per thread: 512 threads hold up to 4 N + 16 = 1024 slots, and a barrier of 8 waves is cheaper than one of 16
      CCM_LDS_ATTR(ctx, CCM_LDS_ORB_OCT, orb_octree_kernel<512>, 152 * 1024);
      hipLaunchKernelGGL(orb_octree_kernel<512>, dim3((o->nlevels + blur_wgs) * fr.n), dim3(512), o->oct_lds, o->st, o->dev, a, fr);
    } else {
      CCM_LDS_ATTR(ctx, CCM_LDS_ORB_OCT2, orb_octree_kernel<1024>, 152 * 1024);
      hipLaunchKernelGGL(orb_octree_kernel<1024>, dim3((o->nlevels + blur_wgs) * fr.n), dim3(1024), o->oct_lds, o->st, o->dev, a, fr);
    }
  }
  {
    ccm_prof_scope ps(ctx, CCM_K_BRIEF, o->st);
    hipLaunchKernelGGL(orb_orient_desc_kernel, dim3(ccm_div_up(o->kp_cap, 4), fr.n), dim3(256), 0, o->st, o->dev, b.d_pyr, b.d_blur, b.d_kin, 0, b.d_n, kout ? kout : b.d_kout, dout ? dout : b.d_desc, count_out,
                       (const KpIn*)b.d_oct_stage, o->oct_stride, (const int*)b.d_oct_counts, a.kp_cap, fr);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

extern "C" int ccm_orb_extract(ccm_orb* o, const uint8_t* img, int w, int h, int stride, ccm_keypoint* kps, uint8_t* desc,
                               int cap, int* n_out, uint8_t* const* pyramid_out) {
  if (!o || !img || w <= 0 || h <= 0 || stride < w || !kps || !desc || !n_out) return ccm_set_error(o ? o->ctx : nullptr, CCM_E_ARG, "ccm_orb_extract: bad args");
  ccm_ctx* ctx = o->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  o->st = ctx->stream;
  o->last_call_read_level0_in_place = false;
  int rc = orb_prepare(o, w, h);
  if (rc) return rc;
  const LevelInfo& L0 = o->dev.lv[0];
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // the image goes through the pinned block (packed rows): a pageable source makes the runtime stage and wait
  for (int y = 0; y < h; y++) memcpy(o->h_io + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(o->B[o->cur].d_pyr + L0.off, L0.stride, o->h_io, w, w, h, hipMemcpyHostToDevice, ctx->stream));
  const size_t o_d = ccm_align256((size_t)o->kp_cap * sizeof(ccm_keypoint));
  o->last_cand_valid = false;
  if (o->oct_ok) {
    // no host in the middle: pyramid, FAST, octree, orientation and descriptors are queued back to back; ONE wait at the end
    ccm_orb::Bufs& b = o->B[o->cur];
    if ((rc = orb_phase1(o, false, true))) return rc;
    if ((rc = orb_phase2_dev(o, cap))) return rc;
    const double t1 = now();
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->h_io, b.d_kout, o_d + (size_t)std::min(cap, o->kp_cap) * 32, hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(b.h_count, b.d_n, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (pyramid_out)
      for (int l = 0; l < o->nlevels; l++)
        if (pyramid_out[l]) {
          const LevelInfo& L = o->dev.lv[l];
          CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(pyramid_out[l], L.w, b.d_pyr + L.off, L.stride, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
        }
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (b.h_count[1] == 0) {
      const int nc = std::min(b.h_count[0], cap);
      if (nc) { memcpy(kps, o->h_io, (size_t)nc * sizeof(ccm_keypoint)); memcpy(desc, o->h_io + o_d, (size_t)nc * 32); }
      const double t5 = now();
      o->t_phase[0] = t1 - t0; o->t_phase[1] = 0; o->t_phase[2] = 0; o->t_phase[3] = 0; o->t_phase[4] = t5 - t1; o->t_phase[5] = t5 - t0;
      *n_out = nc;
      return CCM_OK;
    }
    // a level did not fit the kernel's LDS plan: clear the flag and select on the host from the candidates that are still on the device
    CCM_HIP_CHECK(ctx, hipMemsetAsync(b.d_n + 1, 0, sizeof(int), ctx->stream));
    if (o->oct_cells && o->dev.ncells > 0)   // the compacted lists were not made on the way
      hipLaunchKernelGGL(orb_compact_kernel, dim3(o->dev.ncells), dim3(256), 0, ctx->stream, o->dev.ncells, b.d_cell_slots, b.d_cell_counts, b.d_cand, (uint32_t*)(b.d_cand + o->dev.ncells + 1));
    const size_t first = ((size_t)o->dev.ncells + 1 + std::min(o->cand_cap, kCandFirstCopy)) * sizeof(int);
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(b.h_cand, b.d_cand, first, hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipEventRecord(b.ev_cand, ctx->stream));
  } else if ((rc = orb_phase1(o))) return rc;
  const double t1 = now();
  int n = 0;
  if ((rc = orb_host_select(o, &n))) return rc;
  const double t3 = now();
  if ((rc = orb_phase2(o, n))) return rc;
  const double t4 = now();
  const int nc = std::min(n, cap);
  // (the image upload from h_io completed before the host octree ran: the candidate read-back waited behind it)
  if (nc) CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->h_io, o->B[o->cur].d_kout, o_d + (size_t)nc * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (pyramid_out)
    for (int l = 0; l < o->nlevels; l++)
      if (pyramid_out[l]) {
        const LevelInfo& L = o->dev.lv[l];
        CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(pyramid_out[l], L.w, o->B[o->cur].d_pyr + L.off, L.stride, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
      }
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (nc) { memcpy(kps, o->h_io, (size_t)nc * sizeof(ccm_keypoint)); memcpy(desc, o->h_io + o_d, (size_t)nc * 32); }
  const double t5 = now();
  o->t_phase[0] = t1 - t0; o->t_phase[1] = o->t_wait_cand; o->t_phase[2] = (t3 - t1) - o->t_wait_cand; o->t_phase[3] = t4 - t3; o->t_phase[4] = t5 - t4; o->t_phase[5] = t5 - t0;
  *n_out = nc;
  return CCM_OK;
}

int ccm_internal::orb_debug_timing(const ccm_orb* o, double out_ms[6]) {
  if (!o || !out_ms) return CCM_E_ARG;
  for (int i = 0; i < 6; i++) out_ms[i] = o->t_phase[i];
  return CCM_OK;
}

extern "C" int ccm_orb_extract_batch_dev(ccm_orb* o, const uint8_t* d_imgs, int n_frames, int w, int h, ccm_keypoint* d_kps,
                                         uint8_t* d_desc, int cap, int32_t* d_counts) {
  if (!o || !d_imgs || n_frames < 0 || !d_kps || !d_desc || !d_counts) return ccm_set_error(o ? o->ctx : nullptr, CCM_E_ARG, "ccm_orb_extract_batch_dev: bad args");
  ccm_ctx* ctx = o->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  o->st = ctx->stream;
  o->last_call_read_level0_in_place = true;   // (group path below; the host-octree path copies level 0 and clears the flag)
  int rc = orb_prepare(o, w, h);
  if (rc) return rc;
  if ((rc = orb_alloc_bufs(o, 1))) return rc;
  const LevelInfo& L0 = o->dev.lv[0];
  if (o->oct_ok && cap > 0) {
    // device octree: the whole batch is queued without a single host wait, in groups of kOrbGroup frames per launch (below), two groups in flight on two streams with
    // their own buffer sets; the orientation + descriptor kernel writes keypoints, descriptors and the counts straight into the caller's arrays
    // (round 2: two frames in flight and three device-to-device copies per frame: 0.069 ms per frame; round 4 until its last step: one frame per launch on four streams, 0.0345).
    for (int k = 1; k < kOrbSets; k++) {
      if ((rc = orb_alloc_bufs(o, k))) return rc;
      if (!o->bstream[k]) {
        CCM_HIP_CHECK(ctx, hipStreamCreateWithFlags(&o->bstream[k], hipStreamNonBlocking));
        CCM_HIP_CHECK(ctx, hipEventCreateWithFlags(&o->bev[k], hipEventDisableTiming));
      }
    }
    if (!o->ev_a) CCM_HIP_CHECK(ctx, hipEventCreateWithFlags(&o->ev_a, hipEventDisableTiming));
    CCM_HIP_CHECK(ctx, hipEventRecord(o->ev_a, ctx->stream));            // whatever produced the images on the context's stream comes first
    for (int k = 1; k < kOrbSets; k++) CCM_HIP_CHECK(ctx, hipStreamWaitEvent(o->bstream[k], o->ev_a, 0));
    const bool inplace = true;
    // groups of kOrbGroup frames, ONE launch per kernel and group (blockIdx.y = frame, OrbFrames), two groups in flight on two streams with their own buffer sets:
    // the kernels of a frame are small (8 octree workgroups, ~260 pyramid tiles), four frames per launch fill the chip four times better and cost a quarter of the launches
    // (round 4: one frame per launch on four streams 0.0345 ms per frame).  Level 0 is read IN PLACE from the caller's frames.
    const int group_env = kOrbGroup;
    int gi = 0;
    for (int f0 = 0; f0 < n_frames; f0 += group_env, gi++) {
      const int nf = std::min(group_env, n_frames - f0);
      const int s0 = (gi & 1) * kOrbGroup;
      o->cur = s0;
      o->st = (gi & 1) ? o->bstream[1] : ctx->stream;
      OrbFrames fr;
      memset(&fr, 0, sizeof(fr));
      fr.n = nf; fr.pstride0 = inplace ? w : L0.stride;
      for (int j = 0; j < nf; j++) {
        ccm_orb::Bufs& bj = o->B[s0 + j];
        fr.set[j] = (long long)(bj.d_block - o->B[s0].d_block);
        if (inplace) fr.poff0[j] = (long long)((d_imgs + (size_t)(f0 + j) * w * h) - bj.d_pyr);
        else {
          fr.poff0[j] = L0.off;
          CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(bj.d_pyr + L0.off, L0.stride, d_imgs + (size_t)(f0 + j) * w * h, w, w, h, hipMemcpyDeviceToDevice, o->st));
        }
        fr.kout[j] = (long long)j * cap * (long long)sizeof(ccm_keypoint); fr.desc[j] = (long long)j * cap * 32; fr.cnt[j] = (long long)j * (long long)sizeof(int32_t);
      }
      rc = orb_phase1(o, false, true, &fr);
      if (!rc) rc = orb_phase2_dev(o, cap, d_kps + (size_t)f0 * cap, d_desc + (size_t)f0 * cap * 32, d_counts + f0, &fr);
      if (rc) { o->cur = 0; o->st = ctx->stream; return rc; }
    }
    o->cur = 0; o->st = ctx->stream;
    // the overflow flags are sticky over the batch
    CCM_HIP_CHECK(ctx, hipEventRecord(o->bev[1], o->bstream[1]));
    CCM_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, o->bev[1], 0));
    for (int k = 0; k < kOrbSets; k++) CCM_HIP_CHECK(ctx, hipMemcpyAsync(o->B[k].h_count, o->B[k].d_n, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    o->last_cand_valid = false;
    bool overflow = false;
    for (int k = 0; k < kOrbSets; k++) overflow = overflow || o->B[k].h_count[1] != 0;
    if (!overflow) return CCM_OK;
    // some level of some frame did not fit the kernel's LDS plan: redo the batch with the host octree
    for (int k = 0; k < kOrbSets; k++) CCM_HIP_CHECK(ctx, hipMemsetAsync(o->B[k].d_n + 1, 0, sizeof(int), ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  o->last_call_read_level0_in_place = false;   // (the host-octree path copies every frame's level 0 into the pyramid buffer)
  // the octrees of a frame's levels run on this thread + three helpers for the duration of the call
  const int n_helpers = 3;
  std::unique_ptr<LevelPool> pool;
  if (n_helpers > 0 && n_frames > 1) { pool.reset(new LevelPool()); pool->start(o, n_helpers); }
  // Two frames in flight on ONE in-order stream: iteration f queues the device phase 1 of frame f (into buffer set f & 1), then finishes
  // frame f - 1: the host waits only for ITS candidate list, selects keypoints (DistributeOctTree, ~0.1 ms) while the GPU is busy with
  // frame f, and queues phase 2 + the device-to-device copies of the results.  Stream order alone keeps the two sets apart: phase 2 of
  // frame f - 2 (the last reader of set f & 1) was queued before phase 1 of frame f.
  for (int f = 0; f <= n_frames; f++) {
    if (f < n_frames) {
      o->cur = f & 1;
      CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(o->B[o->cur].d_pyr + L0.off, L0.stride, d_imgs + (size_t)f * w * h, w, w, h, hipMemcpyDeviceToDevice, ctx->stream));
      if ((rc = orb_phase1(o))) { o->cur = 0; return rc; }
    }
    if (f >= 1) {
      const int g = f - 1;
      o->cur = g & 1;
      int n = 0;
      if ((rc = orb_host_select(o, &n, pool.get()))) { o->cur = 0; return rc; }
      if ((rc = orb_phase2(o, n))) { o->cur = 0; return rc; }
      const int nc = std::min(n, cap);
      if (nc) {
        CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_kps + (size_t)g * cap, o->B[o->cur].d_kout, (size_t)nc * sizeof(ccm_keypoint), hipMemcpyDeviceToDevice, ctx->stream));
        CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_desc + (size_t)g * cap * 32, o->B[o->cur].d_desc, (size_t)nc * 32, hipMemcpyDeviceToDevice, ctx->stream));
      }
      *o->B[o->cur].h_count = nc;   // pinned; rewritten two frames later, after this copy has long run (the host waits for frame g + 2's candidates first)
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_counts + g, o->B[o->cur].h_count, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  o->cur = 0;
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

int ccm_internal::orb_debug_level(ccm_orb* o, int level, uint8_t* score_out, uint8_t* blur_out) {
  if (!o || !o->B[o->cur].d_pyr || level < 0 || level >= o->nlevels) return CCM_E_ARG;
  ccm_ctx* ctx = o->ctx;
  const LevelInfo& L = o->dev.lv[level];
  if (score_out) {   // (the extraction no longer makes the score map: computed here for the caller)
    // (advisor, round 4) after a batch call level 0 was read in place from the caller's frames and never copied into the pyramid buffer: there is nothing to score
    if (level == 0 && o->last_call_read_level0_in_place) return ccm_set_error(ctx, CCM_E_STATE, "ccm_orb_debug_level: level 0 of the last (batch) call was read in place from the caller's frames");
    if (!o->d_score_dbg) { CCM_HIP_CHECK(ctx, hipMalloc(&o->d_score_dbg, o->pyr_bytes)); CCM_HIP_CHECK(ctx, hipMemsetAsync(o->d_score_dbg, 0, o->pyr_bytes, ctx->stream)); }
    ccm_prof_scope ps(ctx, CCM_K_FAST_SCORE);
    hipLaunchKernelGGL(orb_fast_score_kernel, dim3(ccm_div_up(o->dev.maxW, 256), o->dev.totalRows), dim3(256), 0, ctx->stream, o->dev, o->B[o->cur].d_pyr, o->d_score_dbg);
    CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(score_out, L.w, o->d_score_dbg + L.off, L.stride, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (blur_out) CCM_HIP_CHECK(ctx, hipMemcpy2DAsync(blur_out, L.w, o->B[o->cur].d_blur + L.off, L.stride, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

int ccm_internal::orb_debug_candidates(ccm_orb* o, int level, ccm_keypoint* out, int cap, int* n_out) {
  if (!o || level < 0 || level >= o->nlevels || !n_out) return CCM_E_ARG;
  if (!o->last_cand_valid) {   // the device octree never brings the candidates to the host: fetch them for this call
    ccm_ctx* ctx = o->ctx;
    ccm_orb::Bufs& b = o->B[o->cur];
    if (!b.d_cand) return CCM_E_ARG;
    if (o->oct_cells && o->dev.ncells > 0)   // the device-octree path skips the compaction: make the lists for this call
      hipLaunchKernelGGL(orb_compact_kernel, dim3(o->dev.ncells), dim3(256), 0, ctx->stream, o->dev.ncells, b.d_cell_slots, b.d_cell_counts, b.d_cand, (uint32_t*)(b.d_cand + o->dev.ncells + 1));
    else if (o->dev.ncells == 0) CCM_HIP_CHECK(ctx, hipMemsetAsync(b.d_cand, 0, sizeof(int), ctx->stream));
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(b.h_cand, b.d_cand, ((size_t)o->dev.ncells + 1 + o->cand_cap) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const int* offs = b.h_cand;
    const uint32_t* rec = (const uint32_t*)(b.h_cand + o->dev.ncells + 1);
    o->last_cand.resize(o->nlevels);
    for (int l = 0; l < o->nlevels; l++) {
      const LevelInfo& L = o->dev.lv[l];
      const int c0 = offs[L.cellBase], c1 = offs[L.cellBase + L.nCols * L.nRows];
      o->last_cand[l].resize(c1 - c0);
      for (int k = c0; k < c1; k++) { const uint32_t r = rec[k]; o->last_cand[l][k - c0] = Cand{(float)(r & 0xFFF), (float)((r >> 12) & 0xFFF), (float)(r >> 24)}; }
    }
    o->last_cand_valid = true;
  }
  if ((int)o->last_cand.size() != o->nlevels) return CCM_E_ARG;
  const std::vector<Cand>& c = o->last_cand[level];
  *n_out = (int)c.size();
  for (int i = 0; i < (int)c.size() && i < cap && out; i++) out[i] = ccm_keypoint{c[i].x, c[i].y, 7.f, -1.f, c[i].response, 0};
  return CCM_OK;
}

// Test hook: the device octree kernel alone on a caller-supplied candidate set of ONE level (integer positions relative to the level's border
// box of W x H, responses 1..255); sel_out receives the indices of the kept candidates in output order.  *overflow = 1 when the set does not
// fit the kernel's LDS plan for N features.
int ccm_internal::orb_debug_octree_dev(ccm_ctx* ctx, const int32_t* x, const int32_t* y, const int32_t* response, int n, int W, int H, int N,
                                        int32_t* sel_out, int cap, int* n_out, int* overflow) {
  if (!ctx || n < 0 || !n_out || !overflow || (n && (!x || !y || !response)) || W <= 0 || H <= 0 || N <= 0 || W > 4000 || H > 4000)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_orb_debug_octree_dev: bad args");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  OrbDev d{};
  d.nlevels = 1; d.ncells = 1;
  d.lv[0].w = W + 2 * kEdge - 6; d.lv[0].h = H + 2 * kEdge - 6; d.lv[0].nCols = 1; d.lv[0].nRows = 1; d.lv[0].cellBase = 0;
  std::vector<int> cand(2 + (size_t)n);
  cand[0] = 0; cand[1] = n;
  for (int k = 0; k < n; k++) {
    if (x[k] < 0 || x[k] > 0xFFF || y[k] < 0 || y[k] > 0xFFF || response[k] < 1 || response[k] > 255) return ccm_set_error(ctx, CCM_E_ARG, "ccm_orb_debug_octree_dev: candidate out of range");
    cand[2 + k] = (int)((uint32_t)x[k] | ((uint32_t)y[k] << 12) | ((uint32_t)response[k] << 24));
  }
  const int lcap = std::max(4 * N + 16, 64), stride = lcap;
  const size_t slots = (size_t)lcap * (2 * sizeof(OctNode) + 16 + 5 * 2 + 1) + 64;
  const size_t budget = 150 * 1024;
  const int kcap = slots < budget ? (int)std::min<size_t>((budget - slots) / 7, 65000) & ~15 : 0;
  if (kcap < 16 || lcap > 2 * kOctTPB) { *overflow = 1; *n_out = 0; return CCM_OK; }
  int *d_cand = nullptr, *d_counts = nullptr, *d_nout = nullptr; KpIn *d_stage = nullptr, *d_kin = nullptr;
  CCM_HIP_CHECK(ctx, hipMalloc(&d_cand, cand.size() * sizeof(int)));
  CCM_HIP_CHECK(ctx, hipMalloc(&d_counts, 4 * sizeof(int)));
  CCM_HIP_CHECK(ctx, hipMalloc(&d_nout, 2 * sizeof(int)));
  CCM_HIP_CHECK(ctx, hipMalloc(&d_stage, (size_t)stride * sizeof(KpIn)));
  CCM_HIP_CHECK(ctx, hipMalloc(&d_kin, (size_t)stride * sizeof(KpIn)));
  hipMemcpyAsync(d_cand, cand.data(), cand.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
  hipMemsetAsync(d_counts, 0, 4 * sizeof(int), ctx->stream);
  hipMemsetAsync(d_nout, 0, 2 * sizeof(int), ctx->stream);
  OctArgs a{};   // (cell lists / blur payload: none)
  a.cand = d_cand; a.ncells = 1; a.nlevels = 1; a.nfeat[0] = N; a.lcap[0] = lcap; a.kcap = kcap; a.stage = d_stage; a.stage_stride = stride;
  a.counts = d_counts; a.kin = d_kin; a.n_out = d_nout; a.kp_cap = stride;
  const size_t lds = ((size_t)kcap * 7 + 15) / 16 * 16 + slots;
  int rc = CCM_OK;
  if (hipFuncSetAttribute((const void*)orb_octree_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess || hipFuncSetAttribute((const void*)orb_octree_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, "octree: LDS attribute");
  std::vector<KpIn> out(stride);
  int hn[2] = {0, 0};
  if (rc == CCM_OK) {
    OrbFrames fr1;
    memset(&fr1, 0, sizeof(fr1));
    fr1.n = 1;
    if (lcap <= 1024) hipLaunchKernelGGL(orb_octree_kernel<512>, dim3(1), dim3(512), lds, ctx->stream, d, a, fr1);
    else hipLaunchKernelGGL(orb_octree_kernel<1024>, dim3(1), dim3(1024), lds, ctx->stream, d, a, fr1);
    hipMemcpyAsync(out.data(), d_stage, out.size() * sizeof(KpIn), hipMemcpyDeviceToHost, ctx->stream);   // the one level's staged list IS the output
    hipMemcpyAsync(hn, d_nout, sizeof(hn), hipMemcpyDeviceToHost, ctx->stream);                              // [1] = overflow flag
    hipMemcpyAsync(hn, d_counts, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);                           // [0] = the level's count
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, "ccm_orb_debug_octree_dev: kernel");
  }
  hipFree(d_cand); hipFree(d_counts); hipFree(d_nout); hipFree(d_stage); hipFree(d_kin);
  if (rc) return rc;
  *overflow = hn[1]; *n_out = hn[1] ? 0 : hn[0];
  if (!hn[1] && sel_out) {
    // the kernel returns positions; positions are unique in a candidate set, so they identify the candidate
    const int minB = kEdge - 3;
    for (int e = 0; e < hn[0] && e < cap; e++) {
      const int px = out[e].x - minB, py = out[e].y - minB;
      int found = -1;
      for (int k = 0; k < n; k++) if (x[k] == px && y[k] == py) { found = k; break; }
      sel_out[e] = found;
    }
  }
  return CCM_OK;
}

// host-only entry point for the CPU tests of the octree (no GPU involved): selects from candidates
// given relative to (minX,minY); writes the indices of the chosen candidates in output order.
extern "C" int ccm_orb_distribute_octree(const float* x, const float* y, const float* response, int n, int minX, int maxX,
                                         int minY, int maxY, int N, int32_t* sel_out, int cap, int* n_out) {
  if (n < 0 || !n_out || (n && (!x || !y || !response)) || maxX <= minX || maxY <= minY || N <= 0) return CCM_E_ARG;
  std::vector<Cand> c(n);
  for (int i = 0; i < n; i++) c[i] = Cand{x[i], y[i], response[i]};
  std::vector<int> sel;
  Octree ws;
  if (n) distribute_octree(ws, c.data(), n, minX, maxX, minY, maxY, N, sel);
  *n_out = (int)sel.size();
  for (int i = 0; i < (int)sel.size() && i < cap && sel_out; i++) sel_out[i] = sel[i];
  return CCM_OK;
}
