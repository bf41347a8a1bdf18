// occupancy probe: how many 64-thread workgroups with N bytes of static LDS fit on one CU (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kBytes>
__global__ __launch_bounds__(64) void k(int* out) {
  __shared__ unsigned char lds[kBytes];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  if (out) out[threadIdx.x] = lds[(threadIdx.x * 7) % kBytes];
}
template <int kBytes>
void probe() {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<kBytes>, 64, 0);
  printf("static LDS %6d B -> %d workgroups (wavefronts) per CU\n", kBytes, n);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName, p.multiProcessorCount,
         p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  probe<16384>(); probe<16384 + 512>(); probe<22528>(); probe<32768>(); probe<32768 - 1280>(); probe<32768 + 64>(); probe<40960>(); probe<65536>();
  return 0;
}
