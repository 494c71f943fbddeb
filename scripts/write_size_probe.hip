// Calibration of rocprofv3's WRITE_SIZE on gfx950 for the store patterns of the BA kernels (the microarch guide calls WRITE_SIZE "uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel writes EXACTLY 18.9 MB (65 694 blocks x 288 B, the S blocks of gba_c4) in a different shape:
//   k_stream16   16-byte stores, fully coalesced, 256-byte aligned                      (the reference point)
//   k_stream8    8-byte stores, fully coalesced
//   k_blocks36   what ba_schur_row3's final sums do: thread (g, el) = (t / 36, t % 36) of a 1008-thread group writes element el of block b0 + g — consecutive
//                threads, consecutive 8-byte addresses, 288-byte (32-byte aligned) blocks, waves straddling block and line boundaries
//   k_blocks36_rows   the same, but every workgroup starts at the block offset of a "row" (varying 288-byte multiples): the alignment the row kernel really has
// build: hipcc --offload-arch=gfx950 -O2 -o write_size_probe write_size_probe.hip ; run under rocprofv3 --pmc WRITE_SIZE (scripts/write_size_probe.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr size_t kBlocks = 65694, kDoubles = kBlocks * 36;
__global__ void k_stream16(double2* out, size_t n2) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n2) out[i] = double2{1.0, 2.0}; }
__global__ void k_stream8(double* out, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = 3.0; }
__global__ __launch_bounds__(1024) void k_blocks36(double* out, size_t nblk) {
  const int g = threadIdx.x / 36, el = threadIdx.x % 36;
  if (g >= 28) return;
  for (size_t b = (size_t)blockIdx.x * 28 + g; b < nblk; b += (size_t)gridDim.x * 28) out[36 * b + el] = -4.0;
}
__global__ __launch_bounds__(1024) void k_blocks36_rows(double* out, size_t nblk, int rows) {   // row r owns blocks [r * nblk / rows, (r + 1) * nblk / rows): ~33 blocks, written 28 at a time
  const int g = threadIdx.x / 36, el = threadIdx.x % 36;
  if (g >= 28) return;
  const size_t b0 = (size_t)blockIdx.x * nblk / rows, b1 = (size_t)(blockIdx.x + 1) * nblk / rows;
  for (size_t b = b0 + g; b < b1; b += 28) out[36 * b + el] = -5.0;
}
int main() {
  double* d; hipMalloc(&d, kDoubles * 8 + 4096);
  hipMemset(d, 0, kDoubles * 8 + 4096); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream16, dim3((kDoubles / 2 + 255) / 256), dim3(256), 0, 0, (double2*)d, kDoubles / 2);
    hipLaunchKernelGGL(k_stream8, dim3((kDoubles + 255) / 256), dim3(256), 0, 0, d, kDoubles);
    hipLaunchKernelGGL(k_blocks36, dim3(1999), dim3(1024), 0, 0, d, kBlocks);
    hipLaunchKernelGGL(k_blocks36_rows, dim3(1999), dim3(1024), 0, 0, d, kBlocks, 1999);
    hipDeviceSynchronize();
  }
  printf("each kernel wrote %zu bytes\n", kDoubles * 8);
  return 0;
}
