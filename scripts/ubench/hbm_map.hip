// Bandwidth map of the GPU's memory: allocate (almost) all of it in CHUNK_GB pieces and time, per piece, a read with the
// sweep kernel's access pattern (4096 concurrent sequential 448-KiB streams, 28 B per lane per step) and a plain
// grid-stride 16-byte read.  Are some physical regions slower than others, and at what granularity?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(256) void k_units(const unsigned* base, long units, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * 4;
  unsigned acc = 0;
  for (long u = (long)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += nw) {
    const unsigned* p = base + u * (448 * 256);  // 448 KiB per unit = 256 row pairs x 448 words
#pragma unroll 4
    for (int jp = 0; jp < 256; jp++) {
      const unsigned* q = p + jp * 448;
      u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q + lane * 4));
      u32x3 b = __builtin_nontemporal_load(reinterpret_cast<const u32x3*>(q + 256 + lane * 3));
      acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_linear(const u32x4* base, long n16, unsigned* sink) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    u32x4 a = __builtin_nontemporal_load(base + i);
    acc += a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
static void time_quarters(unsigned char* big, size_t quarter) {
  hipMemset(big, 1, 4 * quarter);
  unsigned* sink;
  hipMalloc(&sink, 4);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  double tot = 0;
  for (int qd = 0; qd < 4; qd++) {
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
      hipEventRecord(a);
      k_units<<<1024, 256>>>((const unsigned*)(big + qd * quarter), (long)(quarter / (448 * 1024)), sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float t; hipEventElapsedTime(&t, a, b);
      best = t < best ? t : best;
    }
    tot += best;
    printf("  q%d %.3f ms (%.0f GB/s)", qd, best, quarter / best / 1e6);
  }
  printf("  | sum %.3f ms\n", tot);
}
// "big" mode: one allocation of 56 GiB (after an optional pad allocation), the units-pattern read timed per 14-GiB quarter
static int big_mode(double pad_gb, int contiguous) {
  void* pad = nullptr;
  if (pad_gb > 0) hipMalloc(&pad, (size_t)(pad_gb * (1ull << 30)));
  const size_t quarter = (size_t)(14ull << 30) / (448 * 1024) * (448 * 1024);
  unsigned char* big = nullptr;
  hipError_t e = contiguous ? hipExtMallocWithFlags((void**)&big, 4 * quarter, hipDeviceMallocContiguous) : hipMalloc((void**)&big, 4 * quarter);
  if (e != hipSuccess) { printf("alloc failed: %s\n", hipGetErrorName(e)); return 1; }
  if (contiguous) printf("(hipDeviceMallocContiguous) ");
  printf("pad %.0f GiB, db at %p:", pad_gb, (void*)big);
  time_quarters(big, quarter);
  return 0;
}
// "vmm" mode: the same 56 GiB as ONE virtual range backed by separately created physical chunks
// (hipMemAddressReserve + hipMemCreate / hipMemMap): placement decided chunk by chunk, nothing is relocated.
static int vmm_mode(double chunk_gb, double pad_gb) {
  void* pad = nullptr;
  if (pad_gb > 0) hipMalloc(&pad, (size_t)(pad_gb * (1ull << 30)));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess || gran == 0) { printf("hipMemGetAllocationGranularity: %s\n", hipGetErrorName(e)); return 1; }
  const size_t quarter = (size_t)(14ull << 30) / (448 * 1024) * (448 * 1024);
  size_t chunk = (size_t)(chunk_gb * (1ull << 30));
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t total = (4 * quarter + chunk - 1) / chunk * chunk;
  void* va = nullptr;
  e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
  if (e != hipSuccess) { printf("hipMemAddressReserve: %s\n", hipGetErrorName(e)); return 1; }
  int n = 0;
  for (size_t off = 0; off < total; off += chunk, n++) {
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, chunk, &prop, 0);
    if (e != hipSuccess) { printf("hipMemCreate chunk %d: %s\n", n, hipGetErrorName(e)); return 1; }
    e = hipMemMap((char*)va + off, chunk, 0, h, 0);
    if (e != hipSuccess) { printf("hipMemMap chunk %d: %s\n", n, hipGetErrorName(e)); return 1; }
    hipMemRelease(h);  // the mapping keeps the physical memory alive
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(va, total, &acc, 1);
  if (e != hipSuccess) { printf("hipMemSetAccess: %s\n", hipGetErrorName(e)); return 1; }
  printf("(VMM: %d chunks of %.2f GiB, granularity %zu KiB) pad %.0f GiB, db at %p:", n, chunk / 1073741824.0, gran >> 10, pad_gb, va);
  time_quarters((unsigned char*)va, quarter);
  return 0;
}
int main(int argc, char** argv) {
  if (argc > 2 && argv[1][0] == 'b') return big_mode(atof(argv[2]), 0);
  if (argc > 2 && argv[1][0] == 'c') return big_mode(atof(argv[2]), 1);
  if (argc > 2 && argv[1][0] == 'v') return vmm_mode(atof(argv[2]), argc > 3 ? atof(argv[3]) : 0.0);
  const double chunk_gb = argc > 1 ? atof(argv[1]) : 2.0;
  const size_t chunk = (size_t)(chunk_gb * (1ull << 30)) / (448 * 1024) * (448 * 1024);
  size_t fr, tot;
  hipMemGetInfo(&fr, &tot);
  const int n = (int)((fr - (6ull << 30)) / chunk);
  printf("free %.1f GiB of %.1f; %d chunks of %.2f GiB\n", fr / 1073741824.0, tot / 1073741824.0, n, chunk / 1073741824.0);
  std::vector<void*> bufs;
  for (int i = 0; i < n; i++) {
    void* p = nullptr;
    if (hipMalloc(&p, chunk) != hipSuccess) break;
    hipMemsetAsync(p, i + 1, chunk, 0);
    bufs.push_back(p);
  }
  unsigned* sink;
  hipMalloc(&sink, 4);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<double> gu, gl;
  for (size_t i = 0; i < bufs.size(); i++) {
    float ms[2];
    for (int mode = 0; mode < 2; mode++) {
      float best = 1e9;
      for (int r = 0; r < 3; r++) {
        hipEventRecord(a);
        if (mode == 0) k_units<<<1024, 256>>>((const unsigned*)bufs[i], (long)(chunk / (448 * 1024)), sink);
        else k_linear<<<4096, 256>>>((const u32x4*)bufs[i], (long)(chunk / 16), sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        best = t < best ? t : best;
      }
      ms[mode] = best;
    }
    gu.push_back(chunk / ms[0] / 1e6); gl.push_back(chunk / ms[1] / 1e6);
  }
  printf("chunk: units-pattern GB/s | linear GB/s   (address)\n");
  for (size_t i = 0; i < bufs.size(); i++) printf("%3zu: %7.0f | %7.0f   %p\n", i, gu[i], gl[i], bufs[i]);
  std::vector<double> s = gu; std::sort(s.begin(), s.end());
  printf("units pattern: min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f GB/s\n", s[0], s[s.size() / 4], s[s.size() / 2], s[3 * s.size() / 4], s.back());
  s = gl; std::sort(s.begin(), s.end());
  printf("linear:        min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f GB/s\n", s[0], s[s.size() / 4], s[s.size() / 2], s[3 * s.size() / 4], s.back());
  return 0;
}
