// micro-benchmarks of fp64 / conversion / LDS-atomic issue rates on gfx950 (diagnostic; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void kb(double* out, int iters, double seed) {
  __shared__ uint32_t lds[4096];
  double a = threadIdx.x * 1e-3 + seed, b = a * 1.0001 + 1.0, c = a + 0.5, d = b + 0.25;
  double e = a * 3.0, f = b * 0.5, g = c * 1.5, h = d * 0.75;
  uint32_t acc = 0;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // 8 independent fp64 adds x 8
#pragma unroll
      for (int r = 0; r < 8; ++r) { a = a + 1.25; b = b + 1.5; c = c + 0.75; d = d + 2.0; e = e + 1.125; f = f + 0.5; g = g + 3.0; h = h + 0.25; }
    } else if (MODE == 1) { // 8 independent fp64 muls x 8
#pragma unroll
      for (int r = 0; r < 8; ++r) { a = a * 1.0000001; b = b * 0.9999999; c = c * 1.0000002; d = d * 0.9999998; e = e * 1.0000003; f = f * 0.9999997; g = g * 1.0000004; h = h * 0.9999996; }
    } else if (MODE == 2) { // cvt f64 -> u32 (64 per iter)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t0) : "v"(a)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t1) : "v"(b));
        asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t2) : "v"(c)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t3) : "v"(d));
        asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t4) : "v"(e)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t5) : "v"(f));
        asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t6) : "v"(g)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t7) : "v"(h));
        acc += t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
      }
    } else if (MODE == 3) { // dependent fp64 add chain (latency): 64 adds
#pragma unroll
      for (int r = 0; r < 64; ++r) a = a + 1.25;
    } else if (MODE == 4) { // LDS atomics, conflict-free (address = lane): 16 per iter
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&lds[threadIdx.x & 63], 1u);
    } else if (MODE == 5) { // LDS atomics, pseudo-random addresses in 2048 words: 16 per iter
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc = acc * 1664525u + 1013904223u + threadIdx.x; atomicAdd(&lds[(acc >> 12) & 2047], 1u); }
    } else if (MODE == 6) { // the LCG alone (to subtract)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc = acc * 1664525u + 1013904223u + threadIdx.x; }
    } else if (MODE == 7) { // ds_read_b128 random 16-B aligned addresses + use: 16 per iter
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc = acc * 1664525u + 1013904223u + threadIdx.x; const uint4 v = *(const uint4*)&lds[((acc >> 12) & 511) * 4]; acc += v.x ^ v.w; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (double)acc + (double)lds[threadIdx.x];
}
template <int MODE>
void run(const char* name, int wavesPerSimd, double ops_per_iter) {
  double* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, threads = 256, blocks = 256 * wavesPerSimd;
  hipLaunchKernelGGL(kb<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 50, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kb<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * ops_per_iter * wavesPerSimd);
  printf("%-34s waves/SIMD %d: %8.3f ms -> %6.2f ns per wave-op per SIMD (= %6.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, per, per * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("fp64 add (64 indep)", w, 64);
    run<1>("fp64 mul (64 indep)", w, 64);
    run<2>("v_cvt_u32_f64 (64)", w, 64);
    run<3>("fp64 add dependent chain (64)", w, 64);
    run<4>("ds_add_u32 conflict-free (16)", w, 16);
    run<5>("ds_add_u32 random + lcg (16)", w, 16);
    run<6>("lcg alone (16)", w, 16);
    run<7>("ds_read_b128 random + lcg (16)", w, 16);
  }
  return 0;
}
