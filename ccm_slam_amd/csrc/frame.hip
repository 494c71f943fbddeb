// frame.hip — the per-frame glue between extraction and matching on the device:
// Frame::UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid (cslam/src/Frame.cpp:103-118, 284-330) and the
// candidate enumeration of Frame::GetFeaturesInArea (:200-253) + KeyFrame::GetFeaturesInArea (KeyFrame.cpp:1162-1201) for a
// whole batch of window queries, followed by the Hamming distances of every (query, candidate) slot.
// The host receives the candidate CSR in the reference's order (ix-major, iy, insertion order) and replays the
// sequential claim rules on it; it no longer builds the grid or walks cells itself.
#include "common.h"
#include "frame_math.h"
#include <algorithm>
#include <cstring>
#include <vector>

struct ccm_frame {
  ccm_ctx* ctx = nullptr;
  FrameCam cam{};
  FrameBounds b{};
  int distorted = 0, N = 0, cap = 0;
  // device: [xy_un 2N f32 | octave N i32 | cell_idx N i32 | cell_off 3601 i32] + descriptors N x 32
  float* d_xy = nullptr; int* d_oct = nullptr; int* d_cell_idx = nullptr; int* d_cell_off = nullptr; uint8_t* d_desc = nullptr;
  ccm_keypoint* d_kps = nullptr;
  uint8_t* h_stage = nullptr;   // pinned mirror of the [desc | kps] block (own buffer: the upload is not synchronised at return)
};

namespace {

constexpr int kCells = kGridCols * kGridRows;
constexpr int kGridTPB = 1024;

// one workgroup: undistort, bin, exclusive scan, place, restore insertion order inside every cell
__global__ __launch_bounds__(kGridTPB) void frame_grid_kernel(const ccm_keypoint* kps, int N, FrameCam cam, FrameBounds b, int distorted,
                                                              float* xy, int* oct, int* cell_off, int* cell_idx) {
  __shared__ int cnt[kCells + 1];
  __shared__ int part[kGridTPB];
  const int t = threadIdx.x;
  for (int i = t; i <= kCells; i += kGridTPB) cnt[i] = 0;
  __syncthreads();
  for (int i = t; i < N; i += kGridTPB) {
    float x = kps[i].x, y = kps[i].y;
    if (distorted) frame_undistort_point(cam, x, y, x, y);
    xy[2 * i] = x; xy[2 * i + 1] = y;
    oct[i] = kps[i].octave;
    const int c = frame_cell_of(b, x, y);
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  // exclusive scan of 3600 counters: 4 cells per thread
  constexpr int kPer = (kCells + kGridTPB - 1) / kGridTPB;
  int loc[kPer], s = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) { const int c = t * kPer + k; loc[k] = (c < kCells) ? cnt[c] : 0; s += loc[k]; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < kGridTPB; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = part[t] - s;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int c = t * kPer + k;
    if (c < kCells) { cell_off[c] = base; cnt[c] = base; base += loc[k]; }
  }
  if (t == kGridTPB - 1) cell_off[kCells] = part[t];
  __syncthreads();
  for (int i = t; i < N; i += kGridTPB) {
    const int c = frame_cell_of(b, xy[2 * i], xy[2 * i + 1]);
    if (c >= 0) cell_idx[atomicAdd(&cnt[c], 1)] = i;
  }
  __syncthreads();
  // push_back order = ascending feature index: insertion sort of every (tiny) cell
  for (int c = t; c < kCells; c += kGridTPB) {
    const int e0 = cell_off[c], e1 = cnt[c];
    for (int a = e0 + 1; a < e1; a++) {
      const int v = cell_idx[a];
      int q = a - 1;
      while (q >= e0 && cell_idx[q] > v) { cell_idx[q + 1] = cell_idx[q]; q--; }
      cell_idx[q + 1] = v;
    }
  }
}

struct WinQ { const float* u; const float* v; const float* r; const int* minl; const int* maxl; };

// pass 0: count, pass 1: fill.  16 lanes per query: the cells of the window are enumerated in the reference's order
// (ix-major, then iy) and dealt to the lanes 16 at a time; a lane walks its (tiny) cell, a 16-wide prefix sum gives every
// cell its slot range, so the list comes out in exactly the order GetFeaturesInArea produces.  (One thread per query
// walked ~36 cells through three dependent loads each: 50 us per pass.)
constexpr int kQL = 16;   // lanes per query
template <int FILL>
__global__ void frame_window_kernel(int Q, WinQ q, FrameBounds b, const float* xy, const int* oct, const int* cell_off, const int* cell_idx,
                                    int* cnt, const int* off, int* out_idx, int cap = 0x7fffffff, const uint32_t* qdesc = nullptr, const uint32_t* tdesc = nullptr,
                                    uint16_t* out_dist = nullptr) {
  // FILL with descriptors: the Hamming distance of every candidate is written with its index (round 4: one launch less than running ccm_hamming_csr_dev over the lists
  // afterwards; the same XOR + popcount of eight words)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid / kQL, sub = gid % kQL;
  if (i >= Q) return;                                       // whole 16-lane groups leave together
  const float x = q.u[i], y = q.v[i], r = q.r[i];
  const int minLevel = q.minl[i], maxLevel = q.maxl[i];
  if (FILL && off[i + 1] > cap) return;   // would cross the caller's capacity: the host reports the shortfall
  int x0, x1, y0, y1;
  int total = 0;
  if (frame_cell_range(b, x, y, r, x0, x1, y0, y1)) {
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const int ny = y1 - y0 + 1, ncell = (x1 - x0 + 1) * ny;
    int carry = FILL ? off[i] : 0;
    for (int base = 0; base < ncell; base += kQL) {
      const int j = base + sub;
      int n = 0, e0 = 0, e1 = 0;
      if (j < ncell) {
        const int c = (x0 + j / ny) * kGridRows + (y0 + j % ny);
        e0 = cell_off[c]; e1 = cell_off[c + 1];
        for (int s = e0; s < e1; s++) {
          const int k = cell_idx[s];
          if (bCheckLevels) {
            if (oct[k] < minLevel) continue;
            if (maxLevel >= 0 && oct[k] > maxLevel) continue;
          }
          if (fabsf(xy[2 * k] - x) < r && fabsf(xy[2 * k + 1] - y) < r) n++;
        }
      }
      int incl = n;                                          // inclusive prefix over the 16 lanes of the group
#pragma unroll
      for (int d = 1; d < kQL; d <<= 1) { const int v = __shfl_up(incl, d, kQL); if (sub >= d) incl += v; }
      const int chunk_total = __shfl(incl, kQL - 1, kQL);
      if (FILL && n) {
        int w = carry + incl - n;
        for (int s = e0; s < e1; s++) {
          const int k = cell_idx[s];
          if (bCheckLevels) {
            if (oct[k] < minLevel) continue;
            if (maxLevel >= 0 && oct[k] > maxLevel) continue;
          }
          if (fabsf(xy[2 * k] - x) < r && fabsf(xy[2 * k + 1] - y) < r) {
            if (out_dist) {
              const uint4* qp = reinterpret_cast<const uint4*>(qdesc + 8 * (size_t)i);
              const uint4* tp = reinterpret_cast<const uint4*>(tdesc + 8 * (size_t)k);
              const uint4 ql = qp[0], qh = qp[1], tl = tp[0], th = tp[1];
              out_dist[w] = (uint16_t)(__builtin_popcount(ql.x ^ tl.x) + __builtin_popcount(ql.y ^ tl.y) + __builtin_popcount(ql.z ^ tl.z) + __builtin_popcount(ql.w ^ tl.w) +
                                       __builtin_popcount(qh.x ^ th.x) + __builtin_popcount(qh.y ^ th.y) + __builtin_popcount(qh.z ^ th.z) + __builtin_popcount(qh.w ^ th.w));
            }
            out_idx[w++] = k;
          }
        }
      }
      carry += chunk_total;
      total += chunk_total;
    }
  }
  if (!FILL && sub == 0) cnt[i] = total;
}

// exclusive scan of the per-query counts (one workgroup; Q is at most a few 10^4)
__global__ __launch_bounds__(1024) void frame_scan_kernel(const int* cnt, int Q, int* off) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (Q + 1023) / 1024;
  const int i0 = t * per, i1 = min(Q, i0 + per);
  int s = 0;
  for (int i = i0; i < i1; i++) s += cnt[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = part[t] - s;
  for (int i = i0; i < i1; i++) { off[i] = base; base += cnt[i]; }
  if (t == 1023) off[Q] = part[t];
}

// Tracking::SearchLocalPoints' isInFrustum loop (Tracking.cpp:~770-800 / Frame.cpp:139-198): one thread per map point
__global__ void frame_frustum_kernel(FrustumFrame fr, int n, const float* P, const float* Pn, const float* dmin, const float* dmax,
                                     float cos_limit, uint8_t* in_view, float* pu, float* pv, int* lvl, float* pcos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {P[3 * i], P[3 * i + 1], P[3 * i + 2]}, nn[3] = {Pn[3 * i], Pn[3 * i + 1], Pn[3 * i + 2]};
  float u = 0, v = 0, c = 0;
  int l = 0;
  const bool ok = frame_in_frustum(fr, p, nn, dmin[i], dmax[i], cos_limit, u, v, l, c);
  in_view[i] = ok ? 1 : 0;
  pu[i] = u; pv[i] = v; lvl[i] = l; pcos[i] = c;
}

// MapPoint::UpdateNormalAndDepth (MapPoint.cpp:779-823) for a batch of map points, one thread per point.  cv::Mat CV_32F arithmetic
// restated ([EXT]): `a - b` elementwise in f32; cv::norm accumulates the squares in double and returns sqrt in double; `m / s` is
// convertTo with alpha = (float)(1.0 / s); `normal + v` is an f32 add.  The observers are summed in the order of the caller's list
// (the reference iterates a std::map keyed by shared_ptr, i.e. in heap-address order: any order is "the reference's").
__global__ void frame_normal_depth_kernel(int n, const float* pos, const int* obs_off, const int* obs_kf, const float* kf_center, const int* ref_kf,
                                          const int* ref_level, const float* scale_factors, int n_levels, float* normal_out, float* dmin, float* dmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o0 = obs_off[i], o1 = obs_off[i + 1];
  if (o1 == o0) return;                              // observations.empty(): the point keeps what it has
  const float p0 = pos[3 * i], p1 = pos[3 * i + 1], p2 = pos[3 * i + 2];
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  for (int o = o0; o < o1; o++) {
    const float* O = kf_center + 3 * (size_t)obs_kf[o];
    const float d0 = p0 - O[0], d1 = p1 - O[1], d2 = p2 - O[2];
    const double nrm = sqrt((double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2);
    const float a = (float)(1.0 / nrm);
    n0 = n0 + d0 * a; n1 = n1 + d1 * a; n2 = n2 + d2 * a;
  }
  const float* Or = kf_center + 3 * (size_t)ref_kf[i];
  const float c0 = p0 - Or[0], c1 = p1 - Or[1], c2 = p2 - Or[2];
  const float dist = (float)sqrt((double)c0 * (double)c0 + (double)c1 * (double)c1 + (double)c2 * (double)c2);
  const float mx = dist * scale_factors[ref_level[i]];
  dmax[i] = mx;
  dmin[i] = mx / scale_factors[n_levels - 1];
  const float an = (float)(1.0 / (double)(o1 - o0));
  normal_out[3 * i] = n0 * an; normal_out[3 * i + 1] = n1 * an; normal_out[3 * i + 2] = n2 * an;
}

int frame_reserve(ccm_frame* f, int n) {
  if (n <= f->cap) return CCM_OK;
  ccm_ctx* ctx = f->ctx;
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // allocate the new set first and swap only when all of it exists: a failed allocation leaves the frame as it was
  const int cap = std::max(2048, n + n / 2);
  const size_t blk_bytes = (32 + sizeof(ccm_keypoint)) * (size_t)cap + 256;
  float* n_xy = nullptr; int *n_oct = nullptr, *n_cell = nullptr; uint8_t* n_blk = nullptr; void* n_stage = nullptr;
  // descriptors and keypoints in ONE block [desc 32 cap | kps]: ccm_frame_set_keypoints uploads both with one copy
  const bool ok = hipMalloc(&n_xy, 2 * sizeof(float) * (size_t)cap) == hipSuccess && hipMalloc(&n_oct, sizeof(int) * (size_t)cap) == hipSuccess &&
                  hipMalloc(&n_cell, sizeof(int) * (size_t)cap) == hipSuccess && hipMalloc(&n_blk, blk_bytes) == hipSuccess &&
                  hipHostMalloc(&n_stage, blk_bytes, hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (n_xy) hipFree(n_xy); if (n_oct) hipFree(n_oct); if (n_cell) hipFree(n_cell); if (n_blk) hipFree(n_blk); if (n_stage) hipHostFree(n_stage);
    return ccm_set_error(ctx, CCM_E_HIP, "ccm_frame: out of memory growing the keypoint buffers");
  }
  for (void* p : {(void*)f->d_xy, (void*)f->d_oct, (void*)f->d_cell_idx, (void*)f->d_desc}) if (p) hipFree(p);   // d_kps lives inside d_desc's block
  if (f->h_stage) hipHostFree(f->h_stage);
  f->d_xy = n_xy; f->d_oct = n_oct; f->d_cell_idx = n_cell; f->d_desc = n_blk;
  f->d_kps = reinterpret_cast<ccm_keypoint*>(n_blk + ccm_align256(32 * (size_t)cap));
  f->h_stage = static_cast<uint8_t*>(n_stage);
  f->cap = cap;
  return CCM_OK;
}

}  // namespace

extern "C" int ccm_frame_create(ccm_ctx* ctx, const float K[4], const float* dist, int n_dist, int img_w, int img_h, ccm_frame** out) {
  if (!ctx || !K || !out || n_dist < 0 || n_dist > 5 || (n_dist && !dist) || img_w <= 0 || img_h <= 0)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_frame_create: bad args");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ccm_frame* f = new ccm_frame();
  f->ctx = ctx;
  f->cam.fx = K[0]; f->cam.fy = K[1]; f->cam.cx = K[2]; f->cam.cy = K[3];
  double d5[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n_dist; i++) d5[i] = dist[i];
  f->cam.k1 = d5[0]; f->cam.k2 = d5[1]; f->cam.p1 = d5[2]; f->cam.p2 = d5[3]; f->cam.k3 = d5[4];
  f->distorted = (n_dist > 0 && dist[0] != 0.0f) ? 1 : 0;          // mDistCoef.at<float>(0) != 0.0 (Frame.cpp:286, 316)
  // ComputeImageBounds (Frame.cpp:314-347)
  if (f->distorted) {
    const float cx[4] = {0.f, (float)img_w, 0.f, (float)img_w}, cy[4] = {0.f, 0.f, (float)img_h, (float)img_h};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) frame_undistort_point(f->cam, cx[i], cy[i], ux[i], uy[i]);
    f->b.minX = std::min(ux[0], ux[2]); f->b.maxX = std::max(ux[1], ux[3]);
    f->b.minY = std::min(uy[0], uy[1]); f->b.maxY = std::max(uy[2], uy[3]);
  } else {
    f->b.minX = 0.f; f->b.maxX = (float)img_w; f->b.minY = 0.f; f->b.maxY = (float)img_h;
  }
  f->b.wInv = static_cast<float>(kGridCols) / static_cast<float>(f->b.maxX - f->b.minX);   // Frame.cpp:87-88
  f->b.hInv = static_cast<float>(kGridRows) / static_cast<float>(f->b.maxY - f->b.minY);
  if (hipMalloc(&f->d_cell_off, sizeof(int) * (kCells + 1)) != hipSuccess) { delete f; return ccm_set_error(ctx, CCM_E_HIP, "ccm_frame_create: hipMalloc"); }
  *out = f;
  return CCM_OK;
}

extern "C" void ccm_frame_destroy(ccm_frame* f) {
  if (!f) return;
  hipSetDevice(f->ctx->device);
  hipStreamSynchronize(f->ctx->stream);
  for (void* p : {(void*)f->d_xy, (void*)f->d_oct, (void*)f->d_cell_idx, (void*)f->d_cell_off, (void*)f->d_desc}) if (p) hipFree(p);   // d_kps is part of d_desc's block
  if (f->h_stage) hipHostFree(f->h_stage);
  delete f;
}

extern "C" int ccm_frame_bounds(const ccm_frame* f, float bounds[4]) {
  if (!f || !bounds) return CCM_E_ARG;
  bounds[0] = f->b.minX; bounds[1] = f->b.minY; bounds[2] = f->b.maxX; bounds[3] = f->b.maxY;
  return CCM_OK;
}

extern "C" int ccm_frame_set_keypoints(ccm_frame* f, const ccm_keypoint* kps, const uint8_t* desc, int n) {
  if (!f || n < 0 || (n && (!kps || !desc))) return f ? ccm_set_error(f->ctx, CCM_E_ARG, "ccm_frame_set_keypoints: bad args") : CCM_E_ARG;
  ccm_ctx* ctx = f->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rc = frame_reserve(f, n)) return rc;
  f->N = n;
  if (n) {   // both arrays through the pinned staging block, laid out like the device block: one DMA instead of two pageable copies
    const size_t o_k = ccm_align256(32 * (size_t)f->cap), bytes = o_k + sizeof(ccm_keypoint) * (size_t)n;
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // the staging block may still feed the previous frame's copy
    memcpy(f->h_stage, desc, 32 * (size_t)n);
    memcpy(f->h_stage + o_k, kps, sizeof(ccm_keypoint) * (size_t)n);
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(f->d_desc, f->h_stage, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(frame_grid_kernel, dim3(1), dim3(kGridTPB), 0, ctx->stream, f->d_kps, n, f->cam, f->b, f->distorted, f->d_xy, f->d_oct,
                     f->d_cell_off, f->d_cell_idx);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

extern "C" int ccm_frame_get(ccm_frame* f, float* xy_un, int32_t* cell_off, int32_t* cell_idx) {
  if (!f) return CCM_E_ARG;
  ccm_ctx* ctx = f->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (xy_un && f->N) CCM_HIP_CHECK(ctx, hipMemcpyAsync(xy_un, f->d_xy, 2 * sizeof(float) * (size_t)f->N, hipMemcpyDeviceToHost, ctx->stream));
  if (cell_off) CCM_HIP_CHECK(ctx, hipMemcpyAsync(cell_off, f->d_cell_off, sizeof(int) * (kCells + 1), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (cell_idx) {
    int total = 0;
    CCM_HIP_CHECK(ctx, hipMemcpy(&total, f->d_cell_off + kCells, sizeof(int), hipMemcpyDeviceToHost));
    if (total) CCM_HIP_CHECK(ctx, hipMemcpy(cell_idx, f->d_cell_idx, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost));
  }
  return CCM_OK;
}

extern "C" int ccm_frame_window_search(ccm_frame* f, int Q, const float* u, const float* v, const float* r, const int32_t* min_level,
                                       const int32_t* max_level, const uint8_t* qdesc, int32_t* cand_off, int32_t* cand_idx, uint16_t* cand_dist,
                                       int64_t cap, int64_t* n_cand) {
  if (!f || Q < 0 || !cand_off || !n_cand || (Q && (!u || !v || !r || !min_level || !max_level || !qdesc)) || cap < 0 || (cap && (!cand_idx || !cand_dist)))
    return f ? ccm_set_error(f->ctx, CCM_E_ARG, "ccm_frame_window_search: bad args") : CCM_E_ARG;
  ccm_ctx* ctx = f->ctx;
  *n_cand = 0;
  cand_off[0] = 0;
  if (Q == 0) return CCM_OK;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // device staging, mirrored byte for byte in one pinned block so that the inputs travel in ONE H2D copy and the results in
  // ONE D2H copy:  in = [u | v | r | minl | maxl] (5Q words) | qdesc (32Q);  cnt (Q);  out = off (Q+1) | idx (cap) | dist (cap u16)
  const size_t wQ = (size_t)Q;
  const size_t o_desc = ccm_align256(20 * wQ), in_bytes = o_desc + ccm_align256(32 * wQ);
  const size_t o_idx = ccm_align256(4 * (wQ + 1)), o_dist = o_idx + ccm_align256(4 * (size_t)cap), out_bytes = o_dist + ccm_align256(2 * (size_t)cap);
  const size_t cnt_bytes = ccm_align256(4 * wQ);
  void* scratch = nullptr;
  if (int rc = ccm_io_scratch(ctx, in_bytes + cnt_bytes + out_bytes + 256, &scratch)) return rc;
  uint8_t* d_in = (uint8_t*)scratch;
  float* d_q = (float*)d_in;
  uint8_t* d_qdesc = d_in + o_desc;
  int* d_cnt = (int*)(d_in + in_bytes);
  uint8_t* d_out = d_in + in_bytes + cnt_bytes;
  int* d_off = (int*)d_out;
  int* d_idx = (int*)(d_out + o_idx);
  uint16_t* d_dist = (uint16_t*)(d_out + o_dist);
  void* pin = nullptr;
  if (int rc = ccm_pin_scratch(ctx, in_bytes + out_bytes + 64, &pin)) return rc;
  uint8_t* h = (uint8_t*)pin;
  memcpy(h, u, 4 * wQ); memcpy(h + 4 * wQ, v, 4 * wQ); memcpy(h + 8 * wQ, r, 4 * wQ);
  memcpy(h + 12 * wQ, min_level, 4 * wQ); memcpy(h + 16 * wQ, max_level, 4 * wQ);
  memcpy(h + o_desc, qdesc, 32 * wQ);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h, o_desc + 32 * wQ, hipMemcpyHostToDevice, ctx->stream));
  WinQ wq{d_q, d_q + wQ, d_q + 2 * wQ, (const int*)(d_q + 3 * wQ), (const int*)(d_q + 4 * wQ)};
  const int nb = ccm_div_up((int64_t)Q * kQL, 128);
  hipLaunchKernelGGL(frame_window_kernel<0>, dim3(nb), dim3(128), 0, ctx->stream, Q, wq, f->b, f->d_xy, f->d_oct, f->d_cell_off, f->d_cell_idx, d_cnt,
                     (const int*)nullptr, (int*)nullptr, 0);
  hipLaunchKernelGGL(frame_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_cnt, Q, d_off);
  uint8_t* h_out = h + in_bytes;
  int* h_off = (int*)h_out;
  uint8_t* h_idx = h_out + o_idx;
  uint8_t* h_dist = h_out + o_dist;
  if (cap > 0) {
    // optimistic single-sync path: fill and distances are queued behind the scan without waiting for the total; the
    // fill kernel never writes past cap (queries whose list would cross it are skipped) and the total is checked after
    hipLaunchKernelGGL(frame_window_kernel<1>, dim3(nb), dim3(128), 0, ctx->stream, Q, wq, f->b, f->d_xy, f->d_oct, f->d_cell_off, f->d_cell_idx,
                       (int*)nullptr, (const int*)d_off, d_idx, (int)std::min<int64_t>(cap, 0x7fffffff), (const uint32_t*)d_qdesc, (const uint32_t*)f->d_desc, d_dist);
    CCM_HIP_CHECK(ctx, hipGetLastError());
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, o_dist + 2 * (size_t)cap, hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t total = h_off[Q];
    memcpy(cand_off, h_off, 4 * (wQ + 1));
    *n_cand = total;
    if (total > cap) return ccm_set_error(ctx, CCM_E_ARG, "ccm_frame_window_search: candidate capacity too small");
    memcpy(cand_idx, h_idx, 4 * (size_t)total);
    memcpy(cand_dist, h_dist, 2 * (size_t)total);
    return CCM_OK;
  }
  // sizing call (cap == 0): counts only
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_off, d_off, 4 * (wQ + 1), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(cand_off, h_off, 4 * (wQ + 1));
  *n_cand = h_off[Q];
  return CCM_OK;
}

extern "C" int ccm_frame_frustum(ccm_ctx* ctx, const ccm_frustum_frame* fr, int n, const float* P, const float* normal, const float* min_dist,
                                 const float* max_dist, float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y, int32_t* level,
                                 float* view_cos) {
  if (!ctx || !fr || n < 0 || (n && (!P || !normal || !min_dist || !max_dist || !in_view || !proj_x || !proj_y || !level || !view_cos)))
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_frame_frustum: bad args");
  if (n == 0) return CCM_OK;
  static_assert(sizeof(ccm_frustum_frame) == sizeof(FrustumFrame), "ccm_frustum_frame must mirror FrustumFrame");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  FrustumFrame f;
  memcpy(&f, fr, sizeof(f));
  const size_t wn = (size_t)n;
  // device: P 3n | normal 3n | dmin n | dmax n (8n floats in) ; u n | v n | cos n | level n | in_view n bytes (out)
  void* scratch = nullptr;
  if (int rc = ccm_io_scratch(ctx, 4 * 8 * wn + 4 * 4 * wn + wn + 256, &scratch)) return rc;
  void* pin = nullptr;
  if (int rc = ccm_pin_scratch(ctx, 4 * 8 * wn + 4 * 4 * wn + wn + 64, &pin)) return rc;
  float* d_in = (float*)scratch;
  float* d_out = d_in + 8 * wn;
  uint8_t* d_flag = (uint8_t*)(d_out + 4 * wn);
  float* h = (float*)pin;
  memcpy(h, P, 12 * wn); memcpy(h + 3 * wn, normal, 12 * wn); memcpy(h + 6 * wn, min_dist, 4 * wn); memcpy(h + 7 * wn, max_dist, 4 * wn);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h, 32 * wn, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(frame_frustum_kernel, dim3(ccm_div_up(n, 256)), dim3(256), 0, ctx->stream, f, n, d_in, d_in + 3 * wn, d_in + 6 * wn, d_in + 7 * wn,
                     viewing_cos_limit, d_flag, d_out, d_out + wn, (int*)(d_out + 3 * wn), d_out + 2 * wn);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  float* h_out = h + 8 * wn;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, 16 * wn + wn, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(proj_x, h_out, 4 * wn); memcpy(proj_y, h_out + wn, 4 * wn); memcpy(view_cos, h_out + 2 * wn, 4 * wn);
  memcpy(level, h_out + 3 * wn, 4 * wn); memcpy(in_view, h_out + 4 * wn, wn);
  return CCM_OK;
}

extern "C" int ccm_update_normal_and_depth(ccm_ctx* ctx, int n_pt, const float* pos, const int32_t* obs_off, const int32_t* obs_kf, int n_kf,
                                           const float* kf_center, const int32_t* ref_kf, const int32_t* ref_level, const float* scale_factors,
                                           int n_levels, float* normal, float* min_dist, float* max_dist) {
  if (!ctx || n_pt < 0 || n_kf < 0 || n_levels <= 0 || (n_pt && (!pos || !obs_off || !kf_center || !ref_kf || !ref_level || !scale_factors || !normal || !min_dist || !max_dist)))
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_update_normal_and_depth: bad args");
  if (n_pt == 0) return CCM_OK;
  const int n_obs = obs_off[n_pt];
  if (n_obs < 0 || (n_obs && !obs_kf)) return ccm_set_error(ctx, CCM_E_ARG, "ccm_update_normal_and_depth: bad observation lists");
  for (int i = 0; i < n_pt; i++) {
    if (obs_off[i + 1] < obs_off[i]) return ccm_set_error(ctx, CCM_E_ARG, "ccm_update_normal_and_depth: obs_off must be non-decreasing");
    if (obs_off[i + 1] > obs_off[i] && (ref_kf[i] < 0 || ref_kf[i] >= n_kf || ref_level[i] < 0 || ref_level[i] >= n_levels))
      return ccm_set_error(ctx, CCM_E_ARG, "ccm_update_normal_and_depth: reference keyframe / level out of range");
  }
  for (int o = 0; o < n_obs; o++) if (obs_kf[o] < 0 || obs_kf[o] >= n_kf) return ccm_set_error(ctx, CCM_E_ARG, "ccm_update_normal_and_depth: keyframe index out of range");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t np = (size_t)n_pt, no = (size_t)n_obs, nk = (size_t)n_kf;
  // one staging block in, one out: [pos 3np | kf_center 3nk | scale n_levels | obs_off np+1 | obs_kf no | ref_kf np | ref_level np] -> [normal 3np | dmin np | dmax np]
  const size_t in_words = 3 * np + 3 * nk + (size_t)n_levels + (np + 1) + no + 2 * np, out_words = 5 * np;
  void* scratch = nullptr; void* pin = nullptr;
  if (int rc = ccm_io_scratch(ctx, 4 * (in_words + out_words) + 256, &scratch)) return rc;
  if (int rc = ccm_pin_scratch(ctx, 4 * (in_words + out_words) + 64, &pin)) return rc;
  float* h = (float*)pin; float* dv = (float*)scratch;
  size_t w = 0;
  const size_t o_pos = w; memcpy(h + w, pos, 12 * np); w += 3 * np;
  const size_t o_kc = w; memcpy(h + w, kf_center, 12 * nk); w += 3 * nk;
  const size_t o_sf = w; memcpy(h + w, scale_factors, 4 * (size_t)n_levels); w += (size_t)n_levels;
  const size_t o_off = w; memcpy(h + w, obs_off, 4 * (np + 1)); w += np + 1;
  const size_t o_okf = w; if (no) memcpy(h + w, obs_kf, 4 * no); w += no;
  const size_t o_rk = w; memcpy(h + w, ref_kf, 4 * np); w += np;
  const size_t o_rl = w; memcpy(h + w, ref_level, 4 * np); w += np;
  float* d_out = dv + in_words; float* h_out = h + in_words;
  // points without observers keep the caller's values: seed the output block with them
  memcpy(h_out, normal, 12 * np); memcpy(h_out + 3 * np, min_dist, 4 * np); memcpy(h_out + 4 * np, max_dist, 4 * np);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(dv, h, 4 * (in_words + out_words), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(frame_normal_depth_kernel, dim3(ccm_div_up(n_pt, 256)), dim3(256), 0, ctx->stream, n_pt, dv + o_pos, (const int*)(dv + o_off), (const int*)(dv + o_okf),
                     dv + o_kc, (const int*)(dv + o_rk), (const int*)(dv + o_rl), dv + o_sf, n_levels, d_out, d_out + 3 * np, d_out + 4 * np);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, 4 * out_words, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(normal, h_out, 12 * np); memcpy(min_dist, h_out + 3 * np, 4 * np); memcpy(max_dist, h_out + 4 * np, 4 * np);
  return CCM_OK;
}
