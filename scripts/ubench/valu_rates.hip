// Issue cost of the VALU ops used by the NTT / sweep kernels on gfx950, relative to v_fma_f32 (2 cycles per wave64,
// guides/MI355X_MICROARCH.md).  16 independent chains per lane, 8 waves per SIMD, ~1e9 wave-instructions per run.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define CH 16
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[CH], b = seed | 1u;
  unsigned long long w[CH];
  double d[CH], e = 1.0000001 + seed * 1e-9;
#pragma unroll
  for (int i = 0; i < CH; i++) { a[i] = threadIdx.x * 2654435761u + i; w[i] = a[i]; d[i] = a[i] * 1e-3; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e));
      if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 6) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
      if (OP == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
      if (OP == 11) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
      if (OP == 12) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      if (OP == 13) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      if (OP == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
      if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 16) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 17) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 18) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 19) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < CH; i++) s += a[i] + (uint32_t)w[i] + (uint32_t)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static double base_ms = 0;
template <int OP>
void run(const char* name, uint32_t* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8;
  k<OP><<<blocks, 256>>>(out, 7);
  hipEventRecord(a);
  for (int r = 0; r < 3; r++) k<OP><<<blocks, 256>>>(out, 7);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  if (OP == 8) base_ms = ms;
  double wave_instr = 3.0 * blocks * 4.0 * ITERS * CH;
  double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
  printf("%-18s %8.2f ms  %.3f G wave-instr/s/SIMD  = %.2f x v_fma_f32 time (-> %.1f cycles if fma_f32 = 2)\n", name, ms,
         per_simd_per_s / 1e9, base_ms > 0 ? ms / base_ms : 0.0, base_ms > 0 ? 2.0 * ms / base_ms : 0.0);
}
int main() {
  uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<8>("v_fma_f32", out); run<3>("v_add_u32", out); run<17>("v_sub_u32", out); run<18>("v_min_u32", out);
  run<16>("v_cmp_ge_u32", out); run<15>("v_cndmask_b32", out); run<19>("v_lshl_add_u32", out); run<14>("v_alignbit_b32", out);
  run<0>("v_mul_lo_u32", out); run<1>("v_mul_hi_u32", out); run<2>("v_mad_u64_u32", out);
  run<5>("v_mul_u32_u24", out); run<6>("v_mul_hi_u32_u24", out); run<7>("v_mad_u32_u24", out);
  run<4>("v_fma_f64", out); run<9>("v_add_f64", out); run<10>("v_mul_f64", out); run<11>("v_rndne_f64", out);
  run<12>("v_cvt_f64_u32", out); run<13>("v_cvt_u32_f64", out);
  return 0;
}
