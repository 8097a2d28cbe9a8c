// Cross-stream release latency and dependent-dispatch gap: completion events (hipExtLaunchKernelGGL stop event +
// hipStreamWaitEvent) against a flag in signal memory the producer kernel raises itself + hipStreamWaitValue32 on the
// consumer stream.  Diagnostic; not part of the product.   hipcc --offload-arch=gfx950 -O2 -o ubench_wv tools/ubench_wv.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long start, end; };
// busy kernel: every workgroup spins `ticks` of the 100 MHz clock; first start / last end are recorded; the last workgroup
// to finish raises *flag = epoch (system scope) when flag != nullptr
__global__ void k_busy(Stamp* st, unsigned long long ticks, unsigned* counter, unsigned* flag, unsigned epoch) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) atomicMin(&st->start, t0);
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&st->end, wall_clock64());
    if (flag) {
      const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

int main(int argc, char** argv) {
  const int iters = 60, nB = 256, nC = 782;
  const int nA = argc > 1 ? atoi(argv[1]) : 782, tA = argc > 2 ? atoi(argv[2]) : 256;   // producer grid: workgroups x lanes
  printf("producer kernel: %d workgroups x %d lanes\n", nA, tA);
  int can = 0;
  CHK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t mainS, side;
  CHK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  Stamp* st; CHK(hipMalloc(&st, 3 * iters * sizeof(Stamp)));
  unsigned* counter; CHK(hipMalloc(&counter, 64)); CHK(hipMemset(counter, 0, 64));
  unsigned* sig = nullptr;
  if (can) { CHK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CHK(hipMemset(sig, 0, 8)); }
  hipEvent_t ev; CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int mode = 0; mode < (can ? 2 : 1); ++mode) {
    std::vector<Stamp> init(3 * iters, Stamp{~0ull, 0ull});
    CHK(hipMemcpy(st, init.data(), init.size() * sizeof(Stamp), hipMemcpyHostToDevice));
    CHK(hipDeviceSynchronize());
    for (int i = 0; i < iters; ++i) {
      const unsigned epoch = (unsigned)(mode * 1000 + i + 1);
      Stamp *sa = st + 3 * i, *sb = sa + 1, *sc = sa + 2;
      if (mode == 0) {        // events
        hipExtLaunchKernelGGL(k_busy, dim3(nA), dim3(tA), 0, mainS, nullptr, ev, 0, sa, 1000ull, counter, (unsigned*)nullptr, epoch);
        CHK(hipStreamWaitEvent(side, ev, 0));
      } else {                // flag in signal memory (1), or in plain device memory (2)
        unsigned* f = mode == 1 ? sig : counter + 8;
        hipLaunchKernelGGL(k_busy, dim3(nA), dim3(tA), 0, mainS, sa, 1000ull, counter, f, epoch);
        CHK(hipStreamWaitValue32(side, f, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
      }
      hipLaunchKernelGGL(k_busy, dim3(nC), dim3(256), 0, side, sc, 3000ull, counter + 4, (unsigned*)nullptr, epoch);
      hipLaunchKernelGGL(k_busy, dim3(nB), dim3(256), 0, mainS, sb, 8000ull, counter + 4, (unsigned*)nullptr, epoch);
      CHK(hipGetLastError());
    }
    CHK(hipStreamSynchronize(mainS)); CHK(hipStreamSynchronize(side));
    std::vector<Stamp> out(3 * iters);
    CHK(hipMemcpy(out.data(), st, out.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> gapB, latC, gapA;
    for (int i = 5; i < iters; ++i) {
      gapB.push_back(((double)out[3 * i + 1].start - (double)out[3 * i].end) * 0.01);
      latC.push_back(((double)out[3 * i + 2].start - (double)out[3 * i].end) * 0.01);
      gapA.push_back(((double)out[3 * i].start - (double)out[3 * i - 2].end) * 0.01);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
    printf("%-28s main A_end->B_start median %6.2f us (max %6.2f)   side A_end->C_start median %6.2f us (max %6.2f)   B_end->next A_start %6.2f us\n",
           mode == 0 ? "stop event + StreamWaitEvent" : mode == 1 ? "flag (signal mem) + WaitValue" : "flag (device mem) + WaitValue", med(gapB), mx(gapB), med(latC), mx(latC), med(gapA));
  }
  return 0;
}
