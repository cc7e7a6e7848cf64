// Replays an allocation trace of libspiral_hip.so (SPIRAL_ALLOC_DEBUG=1: "[spiral] hipMalloc N bytes -> [p, q)",
// "[spiral] contiguous allocation of N bytes: hipSuccess -> [p, q)", "[spiral] hipFree p") with plain HIP calls and
// nothing else -- no library kernels, no streams -- and checks two things after every allocation:
//   * does the new VIRTUAL range overlap a live one?            (the runtime handed the same addresses out twice)
//   * do all live buffers still hold the pattern they were filled with?
// profiles/r03_contiguous_alloc.md: with physically contiguous allocations (hipDeviceMallocContiguous) in the history the
// runtime returns a range that overlaps a LIVE allocation; zero-filling the new buffer then wipes the old one.  That is
// what round 2 saw as "public parameters change when the database is allocated".
// hipcc --offload-arch=gfx950 -O2 alloc_replay.hip -o alloc_replay ;  ./alloc_replay trace.txt [contiguous=1] [max_gib=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorName(e_)); exit(2); } } while (0)
struct Live { char* p; size_t bytes; unsigned id; bool checked; };
__global__ void k_fill(unsigned* p, size_t n, unsigned id) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned)(i * 2654435761u) ^ id;
}
__global__ void k_check(const unsigned* p, size_t n, unsigned id, unsigned long long* bad) {
  unsigned long long b = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b += p[i] != ((unsigned)(i * 2654435761u) ^ id);
  if (b) atomicAdd(bad, b);
}
int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: alloc_replay trace.txt [contiguous=1] [max_gib=2]\n"); return 2; }
  const int use_contig = argc > 2 ? atoi(argv[2]) : 1;
  const double max_gib = argc > 3 ? atof(argv[3]) : 2.0;   // larger requests are scaled down to this (the fill takes time)
  FILE* f = fopen(argv[1], "r");
  if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
  std::map<unsigned long long, Live> live;         // key = the TRACE's address (names the buffer for the later free)
  unsigned long long* d_bad;
  CK(hipMalloc(&d_bad, 8));
  char line[512];
  unsigned next_id = 1;
  long n_alloc = 0, n_free = 0, overlaps = 0, corrupt = 0;
  auto check_all = [&](const char* when) {
    for (auto& kv : live) {
      Live& L = kv.second;
      CK(hipMemset(d_bad, 0, 8));
      k_check<<<256, 256>>>((const unsigned*)L.p, L.bytes / 4, L.id, d_bad);
      unsigned long long b = 0;
      CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
      if (b && !L.checked) {
        printf("  CONTENT CHANGED %s: buffer #%u [%p, %p) (%zu bytes): %llu words differ\n", when, L.id, (void*)L.p, (void*)(L.p + L.bytes), L.bytes, b);
        L.checked = true;
        corrupt++;
      }
    }
  };
  while (fgets(line, sizeof line, f)) {
    unsigned long long bytes = 0, a = 0, b = 0;
    const char* s;
    bool contig = false;
    if ((s = strstr(line, "[spiral] hipMalloc ")) && sscanf(s, "[spiral] hipMalloc %llu bytes -> [%llx, %llx)", &bytes, &a, &b) == 3) {
    } else if ((s = strstr(line, "[spiral] contiguous allocation of ")) && sscanf(s, "[spiral] contiguous allocation of %llu bytes: hipSuccess -> [%llx, %llx)", &bytes, &a, &b) == 3) {
      contig = true;
    } else if ((s = strstr(line, "[spiral] hipFree ")) && sscanf(s, "[spiral] hipFree %llx", &a) == 1) {
      auto it = live.find(a);
      if (it != live.end()) { CK(hipFree(it->second.p)); live.erase(it); n_free++; }
      continue;
    } else {
      continue;
    }
    size_t want = (size_t)bytes;
    if (want > (size_t)(max_gib * (1ull << 30))) want = (size_t)(max_gib * (1ull << 30));
    want = (want + 3) / 4 * 4;
    void* p = nullptr;
    hipError_t e = (contig && use_contig) ? hipExtMallocWithFlags(&p, want, hipDeviceMallocContiguous) : hipMalloc(&p, want);
    if (e != hipSuccess) { printf("allocation of %zu bytes failed: %s\n", want, hipGetErrorName(e)); return 2; }
    n_alloc++;
    for (auto& kv : live) {
      const Live& L = kv.second;
      if ((char*)p < L.p + L.bytes && L.p < (char*)p + want) {
        printf("  VA OVERLAP: new %s[%p, %p) (%zu bytes) overlaps LIVE buffer #%u [%p, %p) (%zu bytes)\n", contig && use_contig ? "contiguous " : "", p,
               (void*)((char*)p + want), want, L.id, (void*)L.p, (void*)(L.p + L.bytes), L.bytes);
        overlaps++;
      }
    }
    const unsigned id = next_id++;
    k_fill<<<1024, 256>>>((unsigned*)p, want / 4, id);
    CK(hipDeviceSynchronize());
    live[a] = Live{(char*)p, want, id, false};
    if (want >= (1u << 20)) check_all("after an allocation");   // cheap enough: only after the larger ones
  }
  check_all("at the end");
  printf("replayed %ld allocations, %ld frees (%s contiguous allocations): %ld VA overlaps with live buffers, %ld live buffers changed\n", n_alloc, n_free,
         use_contig ? "WITH" : "without", overlaps, corrupt);
  return overlaps || corrupt ? 1 : 0;
}
