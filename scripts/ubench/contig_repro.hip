// Stand-alone check of the claim in profiles/r02_stale_reads.md: does a large PHYSICALLY CONTIGUOUS allocation
// (hipExtMallocWithFlags(..., hipDeviceMallocContiguous)) change the contents of other live device buffers of the
// process?  No library code involved: pattern-fill NBUF small buffers, allocate + fill the big one, re-read the small
// ones (device-to-host copy AND a kernel-side checksum), report.  Exit code 1 if anything changed.
// Mode 2 ("fragment"): first fill most of the device with 1-GiB buffers, free every other one (free memory = many 1-GiB
// holes), THEN ask for the contiguous range: it can only be satisfied by relocating live buffers -- the situation the
// round-2 record blames.  The surviving 1-GiB buffers are checked as well.
// Mode 3 ("history"): what the library did when the corruption was seen (scripts/archive/r03/diag_c2.py): a few SMALL contiguous
// allocations are made, filled and freed first (the databases of small configurations), then the plain small buffers
// (public parameters), then the big contiguous one.
// hipcc --offload-arch=gfx950 -O2 contig_repro.hip -o contig_repro ;  ./contig_repro [big_gib=56] [nbuf=96] [rounds=3] [contiguous=1|2|3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorName(e_)); exit(2); } } while (0)
__host__ __device__ inline unsigned pat(unsigned buf, size_t i) { return (unsigned)(i * 2654435761u) ^ (buf * 0x9E3779B9u) ^ 0xA5A5A5A5u; }
__global__ void k_fill(unsigned* p, size_t n, unsigned buf) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = pat(buf, i);
}
__global__ void k_check(const unsigned* p, size_t n, unsigned buf, unsigned long long* bad) {
  unsigned long long b = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b += p[i] != pat(buf, i);
  if (b) atomicAdd(bad, b);
}
// Mode 4 ("reuse"): the sequence that actually fails in the library (scripts/archive/r03/diag_c2b.py: the public parameters are wrong
// right after sp_pp_deserialize, before any large allocation): small contiguous buffers are written by a KERNEL and freed;
// a plain hipMalloc then reuses the memory, is zero-filled (hipMemset), receives host data (hipMemcpy H2D) and is read by
// a kernel.  Does the kernel see the host data?
__global__ void k_scramble(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned)(i * 747796405u) ^ seed ^ 0xDEAD0000u;
}
__global__ void k_copy(unsigned* dst, const unsigned* src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static int reuse_mode(int rounds, int use_contig, int use_memset) {
  int any = 0;
  for (int round = 0; round < rounds; round++) {
    for (int i = 0; i < 5; i++) {   // the databases of the small configurations
      void* p = nullptr;
      const size_t b = (size_t)(i == 1 || i == 2 ? 32 : 16) << 20;
      hipError_t e = use_contig ? hipExtMallocWithFlags(&p, b, hipDeviceMallocContiguous) : hipMalloc(&p, b);
      if (e != hipSuccess) { printf("small allocation failed: %s\n", hipGetErrorName(e)); return 2; }
      k_scramble<<<1024, 256>>>((unsigned*)p, b / 4, (unsigned)(round * 16 + i));
      k_check<<<64, 256>>>((const unsigned*)p, b / 4, 1u, nullptr == p ? nullptr : (unsigned long long*)p);  // a second kernel reads it (result unused)
      CK(hipDeviceSynchronize());
      CK(hipFree(p));
    }
    const size_t words = 17956864 / 4;
    std::vector<unsigned> host(words), back(words);
    for (size_t i = 0; i < words; i++) host[i] = pat(4242u + round, i);
    unsigned *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, words * 4));
    if (use_memset) { CK(hipMemset(a, 0, words * 4)); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(a, host.data(), words * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b, words * 4));
    if (use_memset) { CK(hipMemset(b, 0, words * 4)); CK(hipDeviceSynchronize()); }
    k_copy<<<2048, 256>>>(b, a, words);          // the kernel that consumes the upload (the NTT of the public parameters)
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(back.data(), b, words * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0, zeros = 0;
    for (size_t i = 0; i < words; i++)
      if (back[i] != host[i]) { if (!bad) first = i; bad++; zeros += back[i] == 0; }
    printf("round %d (%s small buffers, %s hipMemset): %zu of %zu words read by the kernel differ from the upload (first at byte 0x%zx, %zu of them zero)\n", round,
           use_contig ? "CONTIGUOUS" : "plain", use_memset ? "with" : "without", bad, words, first * 4, zeros);
    any |= bad != 0;
    CK(hipFree(a));
    CK(hipFree(b));
  }
  printf(any ? "RESULT: a kernel read STALE data from a freshly uploaded buffer\n" : "RESULT: uploads were read correctly\n");
  return any;
}
int main(int argc, char** argv) {
  if (argc > 4 && atoi(argv[4]) >= 4) return reuse_mode(argc > 3 ? atoi(argv[3]) : 3, atoi(argv[4]) == 4 || atoi(argv[4]) == 6, atoi(argv[4]) != 6);   // 4: contiguous + memset, 5: plain + memset, 6: contiguous, no memset
  const double big_gib = argc > 1 ? atof(argv[1]) : 56.0;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 96, rounds = argc > 3 ? atoi(argv[3]) : 3, contiguous = argc > 4 ? atoi(argv[4]) : 1;
  unsigned long long* d_bad;
  CK(hipMalloc(&d_bad, 8));
  int any = 0;
  for (int round = 0; round < rounds; round++) {
    // some churn first so that the small buffers do not all sit at the bottom of an empty heap
    std::vector<void*> churn;
    for (int i = 0; i < 24; i++) { void* p; CK(hipMalloc(&p, (size_t)(64 + 37 * i) << 20)); churn.push_back(p); }
    std::vector<unsigned*> gib;  // mode 2: the surviving 1-GiB buffers
    const size_t gib_words = (size_t)1 << 28;
    if (contiguous == 2) {
      size_t fr = 0, tot = 0;
      CK(hipMemGetInfo(&fr, &tot));
      const int n_gib = (int)((fr >> 30) - 24);
      std::vector<unsigned*> all;
      for (int i = 0; i < n_gib; i++) { unsigned* p; if (hipMalloc(&p, gib_words * 4) != hipSuccess) break; all.push_back(p); }
      for (size_t i = 0; i < all.size(); i++) {
        if (i & 1) { CK(hipFree(all[i])); } else { k_fill<<<2048, 256>>>(all[i], gib_words, 7000u + (unsigned)gib.size()); gib.push_back(all[i]); }
      }
      CK(hipDeviceSynchronize());
      CK(hipMemGetInfo(&fr, &tot));
      printf("round %d: fragmented: %zu live 1-GiB buffers, %.1f GiB free in 1-GiB holes\n", round, gib.size(), fr / 1073741824.0);
    }
    if (contiguous == 3) {
      for (int i = 0; i < 5; i++) {
        void* p = nullptr;
        const size_t b = (size_t)(16 + 8 * i) << 20;
        hipError_t e3 = hipExtMallocWithFlags(&p, b, hipDeviceMallocContiguous);
        if (e3 == hipSuccess) {
          CK(hipMemset(p, 0x11 * (i + 1), b));
          CK(hipDeviceSynchronize());
          CK(hipFree(p));
        } else {
          (void)hipGetLastError();
        }
      }
      printf("round %d: five small contiguous allocations made, filled and freed\n", round);
    }
    std::vector<unsigned*> bufs(nbuf);
    std::vector<size_t> words(nbuf);
    for (int i = 0; i < nbuf; i++) {
      words[i] = ((size_t)(1 + (i * 7) % 33) << 20) / 4 + 1024 * (i % 5);   // 1 .. 33 MiB, like tables / public parameters
      CK(hipMalloc(&bufs[i], words[i] * 4));
      k_fill<<<512, 256>>>(bufs[i], words[i], (unsigned)(round * 1000 + i));
      if (i % 3 == 0 && !churn.empty()) { CK(hipFree(churn.back())); churn.pop_back(); }
    }
    CK(hipDeviceSynchronize());
    void* big = nullptr;
    const size_t big_bytes = (size_t)(big_gib * (1ull << 30));
    hipError_t e = contiguous != 0 ? hipExtMallocWithFlags(&big, big_bytes, hipDeviceMallocContiguous) : hipMalloc(&big, big_bytes);
    printf("round %d: %s allocation of %.1f GiB: %s (%p)\n", round, contiguous != 0 ? "CONTIGUOUS" : "plain", big_gib, hipGetErrorName(e), big);
    if (e == hipSuccess) CK(hipMemset(big, 0x5A, big_bytes));
    CK(hipDeviceSynchronize());
    size_t bad_kernel = 0, bad_copy = 0;
    int bufs_bad = 0;
    for (int i = 0; i < nbuf; i++) {
      CK(hipMemset(d_bad, 0, 8));
      k_check<<<512, 256>>>(bufs[i], words[i], (unsigned)(round * 1000 + i), d_bad);
      unsigned long long b = 0;
      CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
      std::vector<unsigned> h(words[i]);
      CK(hipMemcpy(h.data(), bufs[i], words[i] * 4, hipMemcpyDeviceToHost));
      size_t bc = 0;
      for (size_t k = 0; k < words[i]; k++) bc += h[k] != pat((unsigned)(round * 1000 + i), k);
      if (b || bc) { bufs_bad++; if (bufs_bad <= 6) printf("  buffer %d (%zu words at %p): %llu words differ in the kernel's view, %zu in the copy's view\n", i, words[i], (void*)bufs[i], b, bc); }
      bad_kernel += b; bad_copy += bc;
    }
    printf("round %d: %d of %d small buffers changed (%zu words kernel view, %zu words copy view)\n", round, bufs_bad, nbuf, bad_kernel, bad_copy);
    any |= bufs_bad != 0;
    int gib_bad = 0;
    for (size_t i = 0; i < gib.size(); i++) {
      CK(hipMemset(d_bad, 0, 8));
      k_check<<<2048, 256>>>(gib[i], gib_words, 7000u + (unsigned)i, d_bad);
      unsigned long long b = 0;
      CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
      if (b) { gib_bad++; if (gib_bad <= 6) printf("  1-GiB buffer %zu at %p: %llu words differ\n", i, (void*)gib[i], b); }
    }
    if (!gib.empty()) printf("round %d: %d of %zu live 1-GiB buffers changed\n", round, gib_bad, gib.size());
    any |= gib_bad != 0;
    for (auto* p : gib) CK(hipFree(p));
    if (big) CK(hipFree(big));
    for (auto* p : bufs) CK(hipFree(p));
    for (auto* p : churn) CK(hipFree(p));
  }
  printf(any ? "RESULT: live buffers CHANGED across the allocation\n" : "RESULT: no live buffer changed\n");
  return any;
}
