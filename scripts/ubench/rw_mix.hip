// What do HBM WRITES cost when they are mixed into a streaming READ?  (profiles/r03_mfma_batch_sweep.md: the batched
// sweep's 512 MB of outputs per plane cost as much as 5 GB of reads.)  One wave per 448-KiB read stream (the sweep's
// access pattern); after every `every` row pairs the wave stores `wbytes` contiguous bytes.  Variants: plain stores,
// non-temporal stores, stores that stay in L2 (all waves write the same 64 KiB), no stores.
// hipcc --offload-arch=gfx950 -O3 rw_mix.hip -o rw_mix ; ./rw_mix [gib = 14]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorName(e_)); exit(2); } } while (0)
// MODE 0 none, 1 plain, 2 non-temporal, 3 L2-resident target
template <int MODE>
__global__ __launch_bounds__(256) void k_rw(const unsigned* base, long units, unsigned* out, int every, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const long u = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= units) return;
  const unsigned* p = base + u * (448 * 256);
  unsigned acc = 0;
  long wpos = u * (256 / every);
  for (int jp = 0; jp < 256; jp += 4) {
    u32x4 a[4];
    u32x3 b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned* q = p + (jp + k) * 448;
      a[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q + lane * 4));
      b[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x3*>(q + 256 + lane * 3));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc += a[k].x ^ a[k].y ^ a[k].z ^ a[k].w ^ b[k].x ^ b[k].y ^ b[k].z;
    if (MODE != 0 && ((jp + 4) % every) == 0) {
      // 64 lanes x 8 B = 512 contiguous bytes per store instruction
      unsigned* o = out + (MODE == 3 ? (long)(u & 127) * 128 : wpos * 128) + lane * 2;
      if (MODE == 2) {
        __builtin_nontemporal_store(acc, o);
        __builtin_nontemporal_store(acc + 1, o + 1);
      } else {
        *reinterpret_cast<uint2*>(o) = make_uint2(acc, acc + 1);
      }
      wpos++;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE>
static void run(const char* name, const unsigned* db, long units, unsigned* out, int every, unsigned* sink, double rbytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_rw<MODE><<<(unsigned)((units + 3) / 4), 256>>>(db, units, out, every, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (r) best = t < best ? t : best;
  }
  const double wbytes = MODE == 0 ? 0 : (double)units * (256 / every) * 512;
  printf("%-34s every %3d row pairs: %.3f ms  read %.0f GB/s, written %.0f MB (%.2f %% of the reads)\n", name, every, best,
         rbytes / best / 1e6, wbytes / 1e6, 100 * wbytes / rbytes);
}
int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 14.0;
  const long units = (long)(gib * (1ull << 30) / (448 * 1024));
  const size_t bytes = (size_t)units * 448 * 1024;
  unsigned *db, *out, *sink;
  CK(hipMalloc(&db, bytes));
  CK(hipMemset(db, 1, bytes));
  CK(hipMalloc(&out, (size_t)units * 64 * 512 + (1 << 20)));
  CK(hipMalloc(&sink, 64));
  CK(hipDeviceSynchronize());
  run<0>("no stores", db, units, out, 256, sink, (double)bytes);
  for (int every : {256, 64, 32, 16, 8, 4}) {
    run<1>("plain stores", db, units, out, every, sink, (double)bytes);
    run<2>("non-temporal stores", db, units, out, every, sink, (double)bytes);
    run<3>("stores to an L2-resident region", db, units, out, every, sink, (double)bytes);
  }
  return 0;
}
