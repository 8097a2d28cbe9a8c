// micro-benchmark (diagnostic; not part of the product): the inner loop of a lane-per-cell allocation kernel.
//   MODE 0: Philox block + linear register scan of NT thresholds for the 4 words (compare-accumulate per threshold)
//   MODE 1: Philox only
//   MODE 2: Philox + pivots in registers + ONE conflict-free 128-bit LDS read per count + 3 compares + lane-private LDS histogram atomic
//   MODE 3: scan only (words from an LCG)
//   MODE 5 (round 5): the product's loop as it is now — Philox4x32-7, pivots = every 5th threshold in registers (3 for N = 20), ONE 128-bit
//           LDS read, 4 compares, one LDS add into the lane's histogram
//   MODE 6 (round 5): the loop an ALIAS table would allow (another stream spec: the table is not the inverse CDF) — Philox4x32-7, column
//           j = hi32(u * N), one 32-bit LDS read of the lane's (cut, alias) word, one compare, one select, one LDS add.  The per-cell
//           table build (N serial steps of Vose's pairing per lane) is NOT in the loop: this is the upper bound of what the change could give.
// Prints cycles per wave-quad per SIMD at several waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../bayesnmf_amd/csrc/dmath.h"
using namespace bnmf;

__device__ __forceinline__ uint32_t gt1(uint32_t a, uint32_t u) {   // 1 if a > u else 0, without the condition-code path
  uint32_t r;
  asm("v_sub_u32_e64 %0, %1, %2 clamp\n\tv_min_u32_e32 %0, 1, %0" : "=v"(r) : "v"(a), "v"(u));
  return r;
}
template <int MODE, int NT>
__global__ __launch_bounds__(256) void kb(uint32_t* out, int quads, uint32_t seed) {
  __shared__ uint4 blk[8 * 256];        // MODE 2: [block][lane] threshold blocks (conflict-free 128-bit reads)
  __shared__ uint32_t hist[32 * 256];   // MODE 2: [bucket][lane]
  const uint32_t tid = threadIdx.x, gid = blockIdx.x * blockDim.x + tid;
  uint32_t thr[NT], c[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { thr[n] = (uint32_t)(((uint64_t)(n + 1) << 32) / (NT + 1)) + gid * 7u; c[n] = 0; }
  if (MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6) {
    if (MODE == 6) { uint32_t* al = reinterpret_cast<uint32_t*>(blk); for (int j = 0; j <= NT; ++j) al[j * 256 + tid] = ((thr[j % NT] * 2654435761u) & ~31u) | (uint32_t)((j * 7 + tid) % (NT + 1)); }
    else if (MODE == 5) for (int j = 0; j < 4; ++j) blk[j * 256 + tid] = uint4{thr[(5 * j) % NT], thr[(5 * j + 1) % NT], thr[(5 * j + 2) % NT], j == 3 ? 0xFFFFFFFFu : thr[(5 * j + 3) % NT]};
    else for (int j = 0; j < 8; ++j) blk[j * 256 + tid] = uint4{thr[(4 * j) % NT], thr[(4 * j + 1) % NT], thr[(4 * j + 2) % NT], thr[(4 * j + 3) % NT]};
    for (int j = 0; j < 32; ++j) hist[j * 256 + tid] = 0;
    __syncthreads();
  }
  uint32_t lcg = gid * 2654435761u + seed;
  for (int q = 0; q < quads; ++q) {
    u32x4 w;
    if (MODE == 3) { lcg = lcg * 1664525u + 1013904223u; w = u32x4{lcg, lcg ^ 0x9E3779B9u, lcg * 3u, ~lcg}; }
    else if (MODE == 5 || MODE == 6) w = philox4x32_7((uint32_t)q, gid, seed, 5u, 17u, 29u);
    else w = philox4x32_10((uint32_t)q, gid, seed, 5u, 17u, 29u);
    const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        c[n] += (thr[n] <= u0) ? 1u : 0u; c[n] += (thr[n] <= u1) ? 1u : 0u;
        c[n] += (thr[n] <= u2) ? 1u : 0u; c[n] += (thr[n] <= u3) ? 1u : 0u;
      }
    } else if (MODE == 1) {
      c[0] += u0 ^ u1 ^ u2 ^ u3;
    } else if (MODE == 5) {
      constexpr int NPV = (NT + 1 + 4) / 5 - 1;       // NT thresholds + the closing 0xFFFFFFFF in blocks of five: four in LDS, the fifth a pivot
      uint32_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
#pragma unroll
      for (int p = 0; p < NPV; ++p) {
        const uint32_t pv = thr[5 * p + 4];
        j0 += (pv <= u0) ? 1u : 0u; j1 += (pv <= u1) ? 1u : 0u; j2 += (pv <= u2) ? 1u : 0u; j3 += (pv <= u3) ? 1u : 0u;
      }
      const uint4 k0 = blk[j0 * 256 + tid], k1 = blk[j1 * 256 + tid], k2 = blk[j2 * 256 + tid], k3 = blk[j3 * 256 + tid];
      const uint32_t b0 = 5 * j0 + (k0.x <= u0) + (k0.y <= u0) + (k0.z <= u0) + (k0.w <= u0);
      const uint32_t b1 = 5 * j1 + (k1.x <= u1) + (k1.y <= u1) + (k1.z <= u1) + (k1.w <= u1);
      const uint32_t b2 = 5 * j2 + (k2.x <= u2) + (k2.y <= u2) + (k2.z <= u2) + (k2.w <= u2);
      const uint32_t b3 = 5 * j3 + (k3.x <= u3) + (k3.y <= u3) + (k3.z <= u3) + (k3.w <= u3);
      atomicAdd(&hist[b0 * 256 + tid], 1u); atomicAdd(&hist[b1 * 256 + tid], 1u);
      atomicAdd(&hist[b2 * 256 + tid], 1u); atomicAdd(&hist[b3 * 256 + tid], 1u);
    } else if (MODE == 6) {
      const uint32_t* al = reinterpret_cast<const uint32_t*>(blk);     // [column][lane]: cut in the upper 27 bits, alias in the lower 5
      auto pick = [&](uint32_t u) {
        const uint64_t x = (uint64_t)u * (uint32_t)(NT + 1);
        const uint32_t j = (uint32_t)(x >> 32), lo = (uint32_t)x;
        const uint32_t e = al[j * 256 + tid];
        return lo < (e | 31u) ? j : (e & 31u);
      };
      const uint32_t b0 = pick(u0), b1 = pick(u1), b2 = pick(u2), b3 = pick(u3);
      atomicAdd(&hist[b0 * 256 + tid], 1u); atomicAdd(&hist[b1 * 256 + tid], 1u);
      atomicAdd(&hist[b2 * 256 + tid], 1u); atomicAdd(&hist[b3 * 256 + tid], 1u);
    } else if (MODE == 4) {   // as MODE 2 with the saturating-subtract compare form
      constexpr int NPV = (NT + 3) / 4 - 1;
      uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
#pragma unroll
      for (int p = 0; p < NPV; ++p) {
        const uint32_t pv = thr[4 * p + 3];
        g0 += gt1(pv, u0); g1 += gt1(pv, u1); g2 += gt1(pv, u2); g3 += gt1(pv, u3);
      }
      const uint32_t j0 = NPV - g0, j1 = NPV - g1, j2 = NPV - g2, j3 = NPV - g3;
      const uint4 k0 = blk[j0 * 256 + tid], k1 = blk[j1 * 256 + tid], k2 = blk[j2 * 256 + tid], k3 = blk[j3 * 256 + tid];
      const uint32_t b0 = 4 * j0 + 3 - (gt1(k0.x, u0) + gt1(k0.y, u0) + gt1(k0.z, u0));
      const uint32_t b1 = 4 * j1 + 3 - (gt1(k1.x, u1) + gt1(k1.y, u1) + gt1(k1.z, u1));
      const uint32_t b2 = 4 * j2 + 3 - (gt1(k2.x, u2) + gt1(k2.y, u2) + gt1(k2.z, u2));
      const uint32_t b3 = 4 * j3 + 3 - (gt1(k3.x, u3) + gt1(k3.y, u3) + gt1(k3.z, u3));
      atomicAdd(&hist[b0 * 256 + tid], 1u); atomicAdd(&hist[b1 * 256 + tid], 1u);
      atomicAdd(&hist[b2 * 256 + tid], 1u); atomicAdd(&hist[b3 * 256 + tid], 1u);
    } else {
      constexpr int NPV = (NT + 3) / 4 - 1;
      uint32_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
#pragma unroll
      for (int p = 0; p < NPV; ++p) {
        const uint32_t pv = thr[4 * p + 3];
        j0 += (pv <= u0) ? 1u : 0u; j1 += (pv <= u1) ? 1u : 0u; j2 += (pv <= u2) ? 1u : 0u; j3 += (pv <= u3) ? 1u : 0u;
      }
      const uint4 k0 = blk[j0 * 256 + tid], k1 = blk[j1 * 256 + tid], k2 = blk[j2 * 256 + tid], k3 = blk[j3 * 256 + tid];
      const uint32_t b0 = 4 * j0 + (k0.x <= u0) + (k0.y <= u0) + (k0.z <= u0);
      const uint32_t b1 = 4 * j1 + (k1.x <= u1) + (k1.y <= u1) + (k1.z <= u1);
      const uint32_t b2 = 4 * j2 + (k2.x <= u2) + (k2.y <= u2) + (k2.z <= u2);
      const uint32_t b3 = 4 * j3 + (k3.x <= u3) + (k3.y <= u3) + (k3.z <= u3);
      atomicAdd(&hist[b0 * 256 + tid], 1u); atomicAdd(&hist[b1 * 256 + tid], 1u);
      atomicAdd(&hist[b2 * 256 + tid], 1u); atomicAdd(&hist[b3 * 256 + tid], 1u);
    }
  }
  uint32_t s = lcg;
#pragma unroll
  for (int n = 0; n < NT; ++n) s += c[n] * (n + 1);
  if (MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6) for (int j = 0; j < 32; ++j) s += hist[j * 256 + tid];
  out[gid] = s;
}
template <int MODE, int NT>
void run(const char* name, int wavesPerSimd) {
  uint32_t* d; hipMalloc(&d, (size_t)256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int quads = 4000, threads = 256, blocks = 256 * wavesPerSimd;   // 256-thread blocks: one wave per SIMD per block
  hipLaunchKernelGGL((kb<MODE, NT>), dim3(blocks), dim3(threads), 0, 0, d, 50, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((kb<MODE, NT>), dim3(blocks), dim3(threads), 0, 0, d, quads, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)quads * wavesPerSimd);
  printf("%-44s NT %3d waves/SIMD %d: %8.3f ms -> %7.1f ns = %7.1f cycles@2.4GHz per wave-quad per SIMD;  all 1024 SIMDs: %.3g counts/s\n",
         name, NT, wavesPerSimd, ms, ns, ns * 2.4, 1024.0 * 256.0 / (ns * 1e-9));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 3, 4}) {
    run<1, 19>("philox only", w);
    run<2, 19>("philox + pivots + LDS block + LDS hist", w);
    run<4, 19>("same, saturating-subtract compares", w);
    run<5, 19>("product loop (Philox-7, blocks of five)", w);
    run<6, 19>("alias loop (Philox-7, no table build)", w);
  }
  return 0;
}
