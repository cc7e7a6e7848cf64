// Does a kernel with more than 64 KiB of DYNAMIC LDS launch on gfx950, and what does it take?
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void k_dyn(unsigned* out, int words) {
  extern __shared__ __attribute__((aligned(16))) unsigned sm[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) sm[i] = i;
  __syncthreads();
  unsigned s = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) s += sm[words - 1 - i];
  atomicAdd(out, s);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
  unsigned* d;
  hipMalloc(&d, 4);
  for (int pass = 0; pass < 3; pass++) {
    for (size_t kb : {48, 64, 66, 96, 128, 160}) {
      size_t bytes = kb * 1024;
      hipError_t ea = hipSuccess;
      if (pass == 1) ea = hipFuncSetAttribute((const void*)k_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (pass == 2) ea = hipFuncSetAttribute((const void*)k_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipMemset(d, 0, 4);
      hipLaunchKernelGGL(k_dyn, dim3(1), dim3(256), bytes, 0, d, (int)(bytes / 4));
      hipError_t el = hipGetLastError();
      hipError_t es = hipDeviceSynchronize();
      printf("pass %d  %3zu KiB: attr %s, launch %s, sync %s\n", pass, kb, hipGetErrorName(ea), hipGetErrorName(el), hipGetErrorName(es));
    }
  }
  return 0;
}
