// micro-benchmarks of VALU issue rates on gfx950 (diagnostic; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../bayesnmf_amd/csrc/dmath.h"
using namespace bnmf;
__device__ __forceinline__ void cmp_acc4(uint32_t T, uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3, uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  unsigned long long s0, s1, s2, s3;
  asm volatile("v_cmp_le_u32_e64 %4, %8, %9\n\tv_cmp_le_u32_e64 %5, %8, %10\n\tv_cmp_le_u32_e64 %6, %8, %11\n\tv_cmp_le_u32_e64 %7, %8, %12\n\t"
               "v_addc_co_u32_e64 %0, %4, 0, %0, %4\n\tv_addc_co_u32_e64 %1, %5, 0, %1, %5\n\tv_addc_co_u32_e64 %2, %6, 0, %2, %6\n\tv_addc_co_u32_e64 %3, %7, 0, %3, %7"
               : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(T), "v"(u0), "v"(u1), "v"(u2), "v"(u3));
}
template <int MODE>
__global__ void kb(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, dd = a + 7;
  uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  uint32_t T[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) T[i] = a * (i + 1) * 2654435761u;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 160 cmp/addc
#pragma unroll
      for (int n = 0; n < 20; ++n) cmp_acc4(T[n], a, b, c, dd, b0, b1, b2, b3);
      a += b0; b ^= b1;
    } else if (MODE == 1) {  // 160 independent v_add / v_xor
#pragma unroll
      for (int n = 0; n < 20; ++n) { b0 += T[n] ^ a; b1 += T[n] ^ b; b2 += T[n] ^ c; b3 += T[n] ^ dd; }
      a += b0; b ^= b1;
    } else if (MODE == 2) {  // philox
      u32x4 w = philox4x32_10(a, b, c, dd, seed, 17);
      a = w.x; b = w.y; c = w.z; dd = w.w;
    } else if (MODE == 3) {  // C++ compare form
#pragma unroll
      for (int n = 0; n < 20; ++n) { b0 += (T[n] <= a) ? 1u : 0u; b1 += (T[n] <= b) ? 1u : 0u; b2 += (T[n] <= c) ? 1u : 0u; b3 += (T[n] <= dd) ? 1u : 0u; }
      a += b0 * 2654435761u; b ^= b1 * 40503u;
    } else if (MODE == 4) {  // sub-sat form
#pragma unroll
      for (int n = 0; n < 20; ++n) { b0 += min(__builtin_elementwise_sub_sat(T[n], a), 1u); b1 += min(__builtin_elementwise_sub_sat(T[n], b), 1u); b2 += min(__builtin_elementwise_sub_sat(T[n], c), 1u); b3 += min(__builtin_elementwise_sub_sat(T[n], dd), 1u); }
      a += b0 * 2654435761u; b ^= b1 * 40503u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + dd + b0 + b1 + b2 + b3;
}
template <int MODE>
void run(const char* name, int wavesPerSimd, double instr_per_iter) {
  uint32_t* d; hipMalloc(&d, 256 * 4 * 64 * 16 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, threads = 256, blocks = 256 * wavesPerSimd;   // 256-thread blocks: 4 waves = 1 per SIMD per block
  hipLaunchKernelGGL(kb<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kb<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_simd = (double)iters * instr_per_iter * wavesPerSimd;
  printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("cmp/addc asm (160)", w, 160);
    run<1>("xor/add (160)", w, 160);
    run<2>("philox (~60+)", w, 60);
    run<3>("C++ compare (160+)", w, 160);
    run<4>("sub_sat/min/add (240)", w, 240);
  }
  return 0;
}
