// Probe used for DESIGN.md §4.1: prints the XCC_ID of the first workgroups of a normal and a cooperative launch (hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID, bits [3:0]
    out[blockIdx.x] = (int)x;
  }
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  int h[4096];
  for (int coop = 0; coop < 2; coop++) {
    int grid = 128;
    if (coop) { void* args[1] = {&d}; hipLaunchCooperativeKernel((const void*)k, dim3(grid), dim3(1024), args, 150 * 1024, 0); }
    else hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
    printf("coop=%d err=%d:", coop, (int)e);
    for (int i = 0; i < 32; i++) printf(" %d", h[i]);
    int ok = 0; for (int i = 0; i < grid; i++) ok += (h[i] == i % 8);
    printf("  | match b%%8: %d/%d\n", ok, grid);
  }
  return 0;
}
