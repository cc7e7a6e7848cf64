// Allocation-churn check, independent of the library: buffers of assorted sizes are allocated, filled by a kernel on a
// non-blocking stream, verified by another kernel, and freed/reallocated in a shuffled order -- the allocation pattern
// of short-lived handles.  Any mismatch means recycled device memory was read through a stale mapping or cache.
// hipcc --offload-arch=gfx950 -O2 -o alloc_churn alloc_churn.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

__global__ void k_fill(uint32_t* p, size_t n, uint32_t tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = tag ^ (uint32_t)(i * 2654435761u);
}
__global__ void k_check(const uint32_t* p, size_t n, uint32_t tag, unsigned long long* bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != (tag ^ (uint32_t)(i * 2654435761u))) atomicAdd(bad, 1ull);
}

struct Buf {
  uint32_t* p = nullptr;
  size_t n = 0;
  uint32_t tag = 0;
};

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned long long* bad;
  CK(hipMalloc(&bad, 8));
  CK(hipMemset(bad, 0, 8));
  const size_t sizes[] = {8280, 20496, 32768, 65536, 114688, 131072, 262144, 393216, 524288, 1048576, 2097152, 4194304, 8388608, 12582912};
  std::vector<Buf> live;
  uint32_t rng = 12345, tag = 1;
  auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  unsigned long long total_bad = 0;
  for (int it = 0; it < iters; it++) {
    // allocate a handful, fill on s1 (some through hipMemcpy from the host on the null stream, as tables are)
    const int na = 3 + rnd() % 6;
    for (int a = 0; a < na; a++) {
      Buf b;
      b.n = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))] / 4;
      b.tag = tag++;
      CK(hipMalloc(&b.p, b.n * 4));
      if (rnd() % 4 == 0) {
        std::vector<uint32_t> h(b.n);
        for (size_t i = 0; i < b.n; i++) h[i] = b.tag ^ (uint32_t)(i * 2654435761u);
        CK(hipMemcpy(b.p, h.data(), b.n * 4, hipMemcpyHostToDevice));
      } else {
        hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s1, b.p, b.n, b.tag);
        CK(hipStreamSynchronize(s1));
      }
      live.push_back(b);
    }
    // verify everything that is live on the other stream
    for (auto& b : live) hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, s2, b.p, b.n, b.tag, bad);
    CK(hipStreamSynchronize(s2));
    unsigned long long h = 0;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    if (h != total_bad) {
      printf("iteration %d: %llu new mismatching words (%zu live buffers)\n", it, h - total_bad, live.size());
      total_bad = h;
    }
    // free a shuffled subset
    for (size_t i = 0; i < live.size();) {
      if (rnd() % 2 == 0 || live.size() > 40) {
        CK(hipFree(live[i].p));
        live[i] = live.back();
        live.pop_back();
      } else {
        i++;
      }
    }
  }
  printf("alloc_churn: %d iterations, %llu mismatching words\n", iters, total_bad);
  return total_bad ? 1 : 0;
}
