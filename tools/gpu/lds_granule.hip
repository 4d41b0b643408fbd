// How many 64-thread workgroups fit a CU as a function of their dynamic LDS size (MI355X: 160 KB per CU)?  Prints the sizes at
// which the count changes -- the allocation granule and the thresholds the tree kernel's LDS block has to stay under.
//   hipcc --offload-arch=gfx950 tools/gpu/lds_granule.hip -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) probe(float* out) {
  extern __shared__ float buf[];
  buf[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = buf[63 - threadIdx.x];
}
int main() {
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int prev = -1;
  for (int bytes = 8192; bytes <= 32768; bytes += 64) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 64, (size_t)bytes) != hipSuccess) { printf("query failed at %d\n", bytes); return 1; }
    if (nb != prev) { printf("%6d B -> %2d workgroups per CU\n", bytes, nb); prev = nb; }
  }
  return 0;
}
