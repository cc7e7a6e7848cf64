// How fast is the wave-per-transform NTT (sdk_amd/csrc/wave_ntt.hpp) by itself, and what do more resident waves buy?
// Registers only: every wave chains `reps` forward transforms (output -> input), tables in LDS as in the product kernels, no
// global traffic inside the loop.  Occupancy is set with dynamic LDS: 1 .. 4 workgroups (= waves per SIMD) per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sdk_amd/csrc -I include scripts/ubench/wave_ntt_rate.hip -o scripts/ubench/wave_ntt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wave_ntt.hpp"
using namespace spiral;

namespace spiral {  // the product headers declare these; nothing here calls them
long tunable(const char*, long d) { return d; }
void launched(u64, const char*) {}
void note_path(u64) {}
}

template <bool CANON>
__global__ __launch_bounds__(256) void k_rate(u32* out, const u32* tw, int reps, u32 q) {
  extern __shared__ __attribute__((aligned(16))) u32 smem[];
  u32* ltw = smem + 4 * WBUF_WORDS;
  const int tau = threadIdx.x, lane = tau & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tau >> 6);
  wtw_stage(ltw, tw + 2 * N, tau);   // (any 2N words do: the image layout only matters for the values)
  __syncthreads();
  WaveScalarTw stw;
  wntt_scalar_tw(stw, tw);
  u32 v[32];
#pragma unroll
  for (int k = 0; k < 32; k++) v[k] = (u32)(((unsigned long long)(tau * 2654435761u + k * 40503u + blockIdx.x)) % q);
  WaveNoHooks nh;
#pragma unroll 1
  for (int r = 0; r < reps; r++) {
    int ln = lane;
    const u32* twi = tw;
    asm volatile("" : "+v"(ln));
    asm volatile("" : "+s"(twi));
    wntt_fwd<CANON>(v, ln, smem + wv * WBUF_WORDS, twi, stw, ltw, q, 2 * q, nh);
    if (!CANON) {
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] &= 0x0fffffffu;   // back below 2q for the next round (one instruction per value)
    }
  }
  u32 acc = 0;
#pragma unroll
  for (int k = 0; k < 32; k++) acc ^= v[k];
  out[blockIdx.x * 256 + tau] = acc;
}

int main() {
  const u32 q = 268369921u;
  std::vector<u32> tw(4 * N);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < 2 * N; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const u32 w = (u32)(s % q);
    tw[i < N ? i : i] = 0;
    if (i < N) { tw[i] = w; tw[N + i] = (u32)(((unsigned __int128)w << 32) / q); }
  }
  u32 *dtw, *dout;
  hipMalloc(&dtw, 4 * N * 4);
  hipMemcpy(dtw, tw.data(), 4 * N * 4, hipMemcpyHostToDevice);
  const int blocks = 256 * 4 * 4, reps = 32;
  hipMalloc(&dout, (size_t)blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t base = (4 * WBUF_WORDS + 2 * N) * 4;
  printf("# wave NTT alone, %d workgroups x 4 waves x %d chained transforms; LDS per workgroup sets the waves per SIMD\n", blocks, reps);
  for (int canon = 0; canon < 2; canon++)
    for (int per_cu = 1; per_cu <= 4; per_cu++) {
      size_t lds = per_cu == 4 ? base : per_cu == 3 ? 52 * 1024 : per_cu == 2 ? 78 * 1024 : 120 * 1024;
      auto kern = canon ? k_rate<true> : k_rate<false>;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      float best = 1e30f;
      for (int it = 0; it < 4; it++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, dout, dtw, reps, q);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
      }
      const double n = (double)blocks * 4 * reps;
      printf("%s  %d waves per SIMD (LDS %6zu B): %8.3f ms  %6.3f ns per transform chip-wide  %6.2f us per transform per SIMD\n",
             canon ? "canonical" : "lazy     ", per_cu, lds, best, best * 1e6 / n, best * 1e3 / n * 1024);
    }
  hipError_t e = hipGetLastError();
  printf("# last error: %s\n", hipGetErrorString(e));
  return 0;
}
