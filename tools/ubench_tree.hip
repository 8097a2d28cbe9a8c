#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
template <int WHICH>
__device__ inline double tree64(double v) {
  {
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double o = __longlong_as_double(((long long)b[WHICH] << 32) | (unsigned)a[WHICH]);
    v = v + o;
  }
  {
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double o = __longlong_as_double(((long long)b[WHICH] << 32) | (unsigned)a[WHICH]);
    v = v + o;
  }
#define STEP(CTRL) { int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32); \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true); \
    v = v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo); }
  STEP(0x108) STEP(0x104) STEP(0x102) STEP(0x101)
  return v;
}
__device__ inline double tree_ref(double v) { for (int h = 32; h >= 1; h >>= 1) v = v + __shfl_down(v, h, 64); return v; }
__global__ void k(double* o, const double* in) {
  double v = in[blockIdx.x * 64 + threadIdx.x];
  double a0 = tree64<0>(v), a1 = tree64<1>(v), b = tree_ref(v);
  if (threadIdx.x == 0) { o[3 * blockIdx.x] = a0; o[3 * blockIdx.x + 1] = a1; o[3 * blockIdx.x + 2] = b; }
}
int main() {
  const int NBK = 1000;
  double* h = (double*)malloc(NBK * 64 * 8); srand(1);
  for (int i = 0; i < NBK * 64; ++i) h[i] = (rand() / (double)RAND_MAX - 0.3) * exp((rand() % 40) - 20.0);
  double *din, *dout; hipMalloc(&din, NBK * 64 * 8); hipMalloc(&dout, NBK * 3 * 8);
  hipMemcpy(din, h, NBK * 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NBK), dim3(64), 0, 0, dout, din);
  double* r = (double*)malloc(NBK * 3 * 8); hipMemcpy(r, dout, NBK * 3 * 8, hipMemcpyDeviceToHost);
  int ok0 = 0, ok1 = 0;
  for (int i = 0; i < NBK; ++i) { ok0 += memcmp(&r[3 * i], &r[3 * i + 2], 8) == 0; ok1 += memcmp(&r[3 * i + 1], &r[3 * i + 2], 8) == 0; }
  printf("variant0 bit-equal %d / %d, variant1 bit-equal %d / %d\n", ok0, NBK, ok1, NBK);
  return 0;
}
