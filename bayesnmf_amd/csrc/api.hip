// bayesnmf_amd/csrc/api.hip — C ABI of libbnmf.so (include/bnmf.h) over the gfx950 kernels.
// Host side: device memory, one HIP stream per handle, launch sequencing of the sweep
// (R/bayesNMF_sampler.R:273-285) and of the constructor draws (:232-257).  There is no CPU path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <algorithm>
#include <map>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include "kernels.h"
#include "zalloc_reg.h"
#include "zalloc_sort.h"
#include "zalloc_tile.h"
#include "zalloc_step.h"
#include "rank.h"
#include "mh.h"

using namespace bnmf;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(BNMF_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

enum { KN_PDRAW = 0, KN_EDRAW = 1, KN_ZALLOC = 2, KN_REDUCE = 3, KN_SIDE = 4, KN_RANK = 5, KN_MH = 6, KN_OTHER = 7 };
static const char* k_names[BNMF_NKERNEL] = {"k_pdraw", "k_edraw", "k_zalloc", "k_reduce", "k_side", "k_rank", "k_mh", "other"};

// ---- device memory of the handles: a small per-device pool ----
// bnmf_destroy followed by bnmf_create (bayesNMF() after bayesNMF(), a BIC sweep, the fuzzers) freed ~60 device allocations and made them
// again.  The re-made allocations cost the next handle a wait of 8-13 ms (rounds 3-4; 27 ms with round 5's Mhat buffers: about a
// millisecond per MB) at its first synchronisation — the driver remaps freed memory lazily, at the first submission that touches it; a
// first handle of a process never saw it (tools/recreate.py: the wait sat in the first bnmf_set_array's stream synchronisation, lives
// 1, 2, 3 of a process, not life 0).  Freed blocks now go to a per-device free list keyed by their size and the next handle of the same
// shape takes them from there: no free, no map, no wait.  Blocks above 256 MB (the record_sample rings: their own cache below) and
// whatever exceeds BNMF_POOL_GB (default 1) are really freed; bnmf_trim() empties the list; an allocation that fails empties it and
// is tried again.
struct DevPool {
  std::mutex m;
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live;
  size_t cached = 0;
};
static DevPool g_pool[64];
static size_t pool_cap() { static const size_t cap = [] { const char* e = getenv("BNMF_POOL_GB"); return (size_t)((e ? atof(e) : 1.0) * 1e9); }(); return cap; }
static size_t pool_drop(int device) {
  if (device < 0 || device >= 64) return 0;
  std::vector<void*> drop;
  size_t tot = 0;
  { std::lock_guard<std::mutex> lk(g_pool[device].m); for (auto& kv : g_pool[device].free_blocks) drop.push_back(kv.second); tot = g_pool[device].cached; g_pool[device].free_blocks.clear(); g_pool[device].cached = 0; }
  for (void* q : drop) (void)hipFree(q);
  return tot;
}
static hipError_t dmalloc_raw(void** out, size_t bytes) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || bytes == 0 || bytes > ((size_t)256 << 20)) return hipMalloc(out, bytes);
  DevPool& P = g_pool[dev];
  {
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.free_blocks.find(bytes);
    if (it != P.free_blocks.end()) { *out = it->second; P.free_blocks.erase(it); P.cached -= bytes; P.live[*out] = bytes; return hipSuccess; }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); pool_drop(dev); e = hipMalloc(out, bytes); }
  if (e == hipSuccess) { std::lock_guard<std::mutex> lk(P.m); P.live[*out] = bytes; }
  return e;
}
template <class T> static hipError_t dmalloc(T** out, size_t bytes) { return dmalloc_raw((void**)out, bytes); }
static hipError_t dfree(void* p) {
  if (!p) return hipSuccess;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipFree(p);
  DevPool& P = g_pool[dev];
  {
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      const size_t bytes = it->second;
      P.live.erase(it);
      if (P.cached + bytes <= pool_cap()) { P.free_blocks.insert({bytes, p}); P.cached += bytes; return hipSuccess; }
    }
  }
  return hipFree(p);
}


struct Arr { double* d = nullptr; size_t n = 0; int stride = 1; bool set = false; std::vector<int> redraw; double* ring = nullptr;
             bool slab = false; };   // slab: d points into the handle's block of scalars (a broadcast hyper-prior value), not an allocation of its own

struct bnmf_handle {
  bnmf_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;        // main stream: draws, k_zalloc, reductions
  hipStream_t side = nullptr;          // side stream: k_side (E part) of the next iteration (overlaps k_zalloc), k_reduce
  hipStream_t side2 = nullptr;         // second side stream: k_side P part (starts right after k_pdraw) and Esum
  hipEvent_t ev_draw = nullptr, ev_side = nullptr, ev_sideP = nullptr, ev_p = nullptr, ev_z = nullptr, ev_red = nullptr, ev_rank = nullptr;
  bool side_valid = false;             // k_side of iteration iter+1 has been issued
  bool side_main = false;              // ... on the main stream (MH / Normal sweeps: launch_side_main)
  bool mh_side_tail = true;            // BNMF_MHSIDETAIL=0 (diagnostics): the main-stream hyper sweep as a launch of its own in front of k_mh_tail
  bool mh_side_main = true;            // BNMF_MHSIDE=0 (diagnostics / tests): the MH / Normal sweeps' hyper sweep on the side stream, as in round 3
  int gate_forced = -1;                // BNMF_GATE at bnmf_create: 0 / 1 forces the merged draw kernel off / on, else by size
  int draw_bw = 0;                     // lanes per workgroup of the merged draw kernel (chosen at the first launch)
  double* dScal = nullptr;              // [BNMF_ID_MAX] broadcast scalars of bnmf_set_array (hyper-prior values given as one number)
  unsigned* dDrawOwn = nullptr; unsigned draw_seq = 0;   // k_draw: owner word per column of P, launch sequence number (kernels.h)
  int dbg_draw_no_p = 0;               // BNMF_DEBUG_DRAW_NO_P (tests): the P workgroups of k_draw leave without claiming their columns
  int dbg_main_delay_us = 0;           // BNMF_DEBUG_MAIN_DELAY_US (tests): a delay kernel in front of the main stream's kernels of every sweep
  int dbg_allside_delay_us = 0;        // BNMF_DEBUG_ALLSIDE_DELAY_US (tests): a delay kernel in front of EVERY kernel launched on the two side streams
  int dbg_side_delay_us = 0;           // BNMF_DEBUG_SIDE_DELAY_US (tests): a delay kernel in front of the P-side hyper sweep of launch_side_merged
  int gate_f0 = 1;                     // flag the gate waits for beside [3]: [1] E-side sweep (k_side), [9] P-side sweep on its own stream (merged draw path)
  uint32_t z_gate_next = 0;            // != 0: the allocation kernel being launched waits at its end for the hyper sweep of this iteration
  uint32_t z_gated_for = 0;            // the last allocation kernel gated for this iteration's hyper sweep (merged draw kernel, BNMF_GATE)
  bool side_ev_stale = false;          // ev_sideP / ev_side not recorded since the last side launches (fixed-rank sweep: recorded on demand)
  bool mhe_k128 = false;               // BNMF_MHE_K128=1 (diagnostics / tests): k_mh_ecol16's 128-row form also where K <= 96
  bool mhe16 = false;                  // the MH / Normal column sweep by k_mh_ecol16 (K <= 128 and its LDS fits)
  bool red_on_side2 = false;           // the last k_reduce was issued on side2 (then side2 needs no event to be ordered behind it)
  double* E_alt = nullptr;             // Gibbs sweep: the other E buffer (k_edraw of t+1 does not overwrite what k_lpe of t still reads)
  bool mh_prep_valid = false;          // MH / Normal models: Et, nzE are current and nzP is zero (k_mh_tail of the previous iteration)
  bool mh_pipe = false;                // Poisson MH models at fixed rank through k_mh_ecol16: what followed the two sweeps is hosted BY them (mh.h; BNMF_MHPIPE=0: k_mh_tail)
  bool mh_pipe_valid = false;          // ... and its Et / parity flag buffers are current
  uint32_t mh_etail_pending = 0;       // ... the iteration whose E side (hyper sweep of the next, log-prior, record) has not been issued yet
  const void* z_attr_kernel = nullptr;   // allocation kernel whose dynamic-LDS limit has been raised for this handle
  bool red_pending = false, red_issued = false; uint32_t red_t = 0; int red_row = 0;   // k_reduce of the previous iteration, issued late
  int iter = 0;
  bool inited = false;
  Dev dev{};
  Arr arr[BNMF_ID_MAX];
  int32_t *dM = nullptr, *dZsumK = nullptr, *dZsumG = nullptr, *dZ = nullptr;
  int* dR = nullptr; int* dRedraw = nullptr;
  double *dEsum = nullptr, *dPsum = nullptr, *dlpPn = nullptr, *dlpE = nullptr, *dcol = nullptr;
  double* hMetrics = nullptr;          // the metric rows live in mapped host memory (dMetrics is its device address): k_compose writes them
                                       // where the host reads them, no device-to-host copy at the end of a call
  double *dLut = nullptr, *dTemp = nullptr, *dMetrics = nullptr, *dRaw = nullptr, *dRankCol = nullptr, *dRankMhat = nullptr;
  uint32_t* dRankSync = nullptr; int rank_grid = 0; bool rank_reg = false, rank_half = false; void* dRankDbg = nullptr;
  int32_t* dMt = nullptr; double* dEt = nullptr;
  int32_t* zring = nullptr;            // save_Z with a window: samples$Z, [wcap][K*N*G] int32 (only if it fits BNMF_ZRING_GB, default 32)
  double *dMhat = nullptr, *dAccPn = nullptr, *dAccEpart = nullptr; int* dNzE = nullptr; int mh_S = 1; size_t mhe_lds = 0; int mhe_gw = 0;
  size_t metrics_rows = 0;
  int maxM = 0, nblkE = 0;
  int wcap = 0;                        // ring capacity = window + 1: the hyper sweep of iteration t+1 is issued (and, in the
                                       // Gibbs sweep, recorded) during iteration t, one slot ahead of the oldest kept sample
  int z_grid = 0, z_zw = 8, z_ablate = 0; bool z_reg = false; size_t z_lds = 0; ZGeom zg{};
  bool z_tile = false, z_lean = false; ZTGeom ztg{}; double* dMhatZ = nullptr;   // k_zalloc_tile (zalloc_tile.h): N > 24 / large K
  // k_zalloc_sort (zalloc_sort.h): stats mode, N <= 24 — the static schedule built from M at bnmf_create
  bool z_sort = false, zs_pk = false; ZSGeom zsg{}; int zs_nblk = 0, zs_w = 0; size_t zs_lds = 0; int32_t* dZsM = nullptr;
  int zs_it16 = 0, zs_qmax = ZS_QMAX;   // 2-byte items; quads per item
  uint32_t* dZsRec = nullptr; int zx_cols = 0; size_t zx_lds = 0;   // save_Z on the sorted schedule: the items' records, k_zexpand's columns per pass and LDS bytes
  // Round 5: Z of the sorted schedule is kept AS RECORDS (two 16-bit counts per word and item: 44 MB per iteration at the metric configuration
  // against 77 MB of Z) and expanded when somebody reads it (bnmf_get_array, bnmf_window): k_zexpand left the loop.  With a window the
  // records of iteration t live in slot (t - 1) % wcap of dZsRecRing (samples$Z); zs_eager (BNMF_ZEAGER=1, measurements): expand every iteration.
  uint32_t* dZsRecRing = nullptr; size_t zs_recwords = 0; int z_expanded_iter = 0; bool zs_eager = false;
  uint32_t* dZsItems = nullptr; ZSBlock* dZsBlocks = nullptr; int* dZsCols = nullptr; unsigned long long* dZsProf = nullptr;
  double* dZsMh = nullptr;              // [G][K] Mhat left by k_zalloc_sort for the per-column metric terms (colterms.h)
  uint32_t ct_pending = 0;              // iteration whose column terms have not been summed yet (0: none)
  long colmax = 0;                      // largest column total of M
  bool zs_shared = false;               // the sorted schedule spreads large cells over the blocks: ZsumK is accumulated (atomics), the draw kernels zero it
  int n_cu = 256;
  // k_zalloc_step (zalloc_step.h): stats mode, 25 <= N <= 100, any K — the static schedule built from M at bnmf_create
  bool z_step = false; ZPGeom zpg{}; int zp_ns = 0 /* waves per workgroup */, zp_gbp = 0; size_t zp_lds = 0;
  bool zp_it16 = false;                // k_zalloc_step's items as uint16
  uint32_t* dZpItems = nullptr; ZPWg* dZpWgs = nullptr; ZPBatch* dZpBatches = nullptr; ZPStep* dZpSteps = nullptr; int* dZpCols = nullptr;
  hipEvent_t ev[2 * BNMF_NKERNEL]{};
  bool have_ev = false;
  double* dMap = nullptr; size_t map_words = 0;   // scratch of bnmf_map (grown on demand)
  int devlock_fd = -1;                 // the device's lock file (<BNMF_LOCKDIR or /tmp>/bnmf_dev_<PCI bus id>.lock): the device gate's rule across the PROCESSES that share the device
  int devgate_fd = -1;                 // ... and its turnstile (.gate): a process that wants the device exclusively holds it while it waits, new sharers queue behind it
  bool devlock_off = false;            // BNMF_DEVLOCK=0: the caller vouches that no other process uses the device
  unsigned char* dAsg = nullptr; size_t asg_bytes = 0;   // scratch of bnmf_assign (grown on demand): catalogue, norms, cosines, slot / signature lists
  unsigned* dFlags = nullptr;                     // [0] counter, [1] flag of the E-side hyper sweep; [2], [3] of P part + Esum; [5], [6] P inside k_draw;
                                                  // [8], [9] P-side sweep on its own stream
  int* hErr = nullptr; int* dErr = nullptr;       // time-out words of the bounded in-kernel waits, in mapped host memory (read without a copy):
                                                  // [0] a draw kernel waiting for the hyper sweep, [1] the rank sweep's exchange
  bool flags_valid = false;                       // the side work of the next iteration publishes its flags
  bool serial = false;                            // no kernel may wait for a kernel of another stream (serialised dispatch: counter
                                                  // collection, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING; or BNMF_SERIAL=1): stream waits only
  bool poisoned = false;                          // a bounded in-kernel wait timed out: the state is no longer the chain's, every call fails
  std::vector<double> hist;                       // [wcap][4]: loglikelihood, logposterior, P / E mean acceptance of the last iterations
  std::vector<double> temp_host;                  // temperature schedule (host copy, for the convergence rule)
};

static size_t id_len(const bnmf_handle* h, int id) {
  const size_t K = h->cfg.K, G = h->cfg.G, N = h->cfg.N;
  switch (id) {
    case BNMF_P: case BNMF_ZSUMG: case BNMF_ALPHA_P: case BNMF_BETA_P: case BNMF_MU_P: case BNMF_SIGMASQ_P:
    case BNMF_LAMBDA_P: case BNMF_HA_P: case BNMF_HB_P: case BNMF_HC_P: case BNMF_HD_P: case BNMF_HM_P:
    case BNMF_HS_P: case BNMF_ACC_P: return K * N;
    case BNMF_E: case BNMF_ZSUMK: case BNMF_ALPHA_E: case BNMF_BETA_E: case BNMF_MU_E: case BNMF_SIGMASQ_E:
    case BNMF_LAMBDA_E: case BNMF_HA_E: case BNMF_HB_E: case BNMF_HC_E: case BNMF_HD_E: case BNMF_HM_E:
    case BNMF_HS_E: case BNMF_ACC_E: return N * G;
    case BNMF_A: return N;
    case BNMF_R: return 1;
    case BNMF_Z: return K * N * G;
    case BNMF_SIGMASQ: case BNMF_ALPHA: case BNMF_BETA: return G;
    case BNMF_MHAT: return K * G;
    default: return 0;
  }
}
static bool is_hyper(int id) { return (id >= 30 && id < 50) || id == BNMF_ALPHA || id == BNMF_BETA; }
static bool is_prior_param(int id) { return id >= BNMF_ALPHA_P && id <= BNMF_LAMBDA_E; }   // 2 slots, slot(t) = t & 1
static int cur_slot(const bnmf_handle* h) { return (h->iter > 0 ? h->iter : 1) & 1; }
static bool is_pside(int id) { size_t dummy = 0; (void)dummy; return id == BNMF_P || id == BNMF_ALPHA_P || id == BNMF_BETA_P || id == BNMF_MU_P || id == BNMF_SIGMASQ_P || id == BNMF_LAMBDA_P; }

static int ensure_Z(bnmf_handle* h);   // save_Z on the sorted schedule: Z of the current iteration expanded from its records, if it is not
static int ensure(bnmf_handle* h, int id) {
  Arr& a = h->arr[id];
  if (a.d) return 0;
  const size_t n = id_len(h, id), tot = n * (is_prior_param(id) ? 2 : 1);
  HIPCHK(dmalloc(&a.d, tot * sizeof(double)));
  std::vector<double> nan(tot, std::nan(""));
  HIPCHK(hipMemcpy(a.d, nan.data(), tot * sizeof(double), hipMemcpyHostToDevice));
  a.n = n; a.stride = 1;
  return 0;
}

static void refresh_dev(bnmf_handle* h) {
  Dev& d = h->dev;
  const bnmf_config& c = h->cfg;
  d.K = c.K; d.G = c.G; d.N = c.N;
  d.prior = c.prior; d.likelihood = c.likelihood; d.MH = c.MH; d.learning_rank = c.learning_rank;
  d.rank_method = c.rank_method; d.save_Z = c.save_Z;
  d.zsumk_accum = (h->z_tile || (h->z_sort && h->zs_shared)) ? 1 : 0;
  d.k0 = (uint32_t)c.seed; d.k1 = (uint32_t)(c.seed >> 32) ^ c.chain_id;
  d.maxM = h->maxM;
  d.M = h->dM; d.Mt = h->dMt; d.Et = h->dEt; d.R = h->dR;
  d.P = h->arr[BNMF_P].d; d.E = h->arr[BNMF_E].d; d.A = h->arr[BNMF_A].d;
  d.ZsumK = h->dZsumK; d.ZsumG = h->dZsumG; d.Z = h->dZ;
  d.Alpha_p = h->arr[BNMF_ALPHA_P].d; d.Beta_p = h->arr[BNMF_BETA_P].d;
  d.Alpha_e = h->arr[BNMF_ALPHA_E].d; d.Beta_e = h->arr[BNMF_BETA_E].d;
  d.Mu_p = h->arr[BNMF_MU_P].d; d.Sig_p = h->arr[BNMF_SIGMASQ_P].d;
  d.Mu_e = h->arr[BNMF_MU_E].d; d.Sig_e = h->arr[BNMF_SIGMASQ_E].d;
  d.Lam_p = h->arr[BNMF_LAMBDA_P].d; d.Lam_e = h->arr[BNMF_LAMBDA_E].d;
  auto hr = [&](int id) { return HRef{h->arr[id].d, h->arr[id].stride}; };
  d.hA_p = hr(BNMF_HA_P); d.hB_p = hr(BNMF_HB_P); d.hC_p = hr(BNMF_HC_P); d.hD_p = hr(BNMF_HD_P);
  d.hM_p = hr(BNMF_HM_P); d.hS_p = hr(BNMF_HS_P);
  d.hA_e = hr(BNMF_HA_E); d.hB_e = hr(BNMF_HB_E); d.hC_e = hr(BNMF_HC_E); d.hD_e = hr(BNMF_HD_E);
  d.hM_e = hr(BNMF_HM_E); d.hS_e = hr(BNMF_HS_E);
  d.sigmasq = h->arr[BNMF_SIGMASQ].d; d.hAlphaS = hr(BNMF_ALPHA); d.hBetaS = hr(BNMF_BETA);
  d.Esum = h->dEsum; d.Psum = h->dPsum; d.lpPn = h->dlpPn; d.lpE_part = h->dlpE;
  d.colsse = h->dcol; d.colll = h->dcol + c.G; d.colkl = h->dcol + 2 * (size_t)c.G;
  d.lgfact = h->dLut; d.logm = h->dLut + (h->maxM + 1);
  d.temperature = h->dTemp; d.n_temperature = c.n_temperature;
  d.metrics = h->dMetrics; d.raw = h->dRaw;
  d.lenP = (size_t)c.K * c.N; d.lenE = (size_t)c.N * c.G;
}

// BNMF_TIMING=1 (diagnostics): wall-clock marks of bnmf_create on stderr
struct CreateClock {
  bool on = getenv("BNMF_TIMING") != nullptr; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what) { if (!on) return; const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[bnmf_create] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; }
};
__global__ void k_set_scalar(double* p, double v) { *p = v; }

extern "C" {

int bnmf_destroy(bnmf_handle* h);
int bnmf_version(void) { return BNMF_VERSION; }
const char* bnmf_last_error(void) { return g_err; }
const char* bnmf_kernel_name(int i) { return (i >= 0 && i < BNMF_NKERNEL) ? k_names[i] : ""; }

int bnmf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int bnmf_device_info(int device, char* buf, size_t buflen) {
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, device));
  snprintf(buf, buflen, "%s arch=%s CUs=%d clock=%dMHz mem=%.1fGiB lds/block=%zu",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000,
           (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), (size_t)p.sharedMemPerBlock);
  return 0;
}

static int create_impl(const bnmf_config* cfg, const int32_t* M, bnmf_handle* h);
// lgamma / digamma table of the Alpha sampler's tangent points (dsamplers.h g_alut): filled once per device
static int ensure_alut(int device) {
  static bool done[64] = {};
  static std::mutex mtx;                                  // bnmf_create may be called from several host threads (one chain each)
  if (device < 0 || device >= 64) return fail(BNMF_EINVAL, "device %d out of range", device);
  std::lock_guard<std::mutex> lock(mtx);
  if (done[device]) return 0;
  hipLaunchKernelGGL(k_alut_fill, dim3((ALUT_N + 255) / 256), dim3(256), 0, 0, 0);
  hipLaunchKernelGGL(k_alut_fill, dim3((ALUT_N + 255) / 256), dim3(256), 0, 0, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  done[device] = true;
  return 0;
}

int bnmf_create(const bnmf_config* cfg, const int32_t* M, bnmf_handle** out) {
  if (!cfg || !M || !out) return fail(BNMF_EINVAL, "bnmf_create: null argument");
  if (cfg->K < 1 || cfg->G < 1 || cfg->N < 1) return fail(BNMF_EINVAL, "bnmf_create: dims must be positive");
  if (cfg->N > 1024) return fail(BNMF_EINVAL, "bnmf_create: N > 1024 unsupported");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(BNMF_ENODEVICE, "bnmf_create: no HIP device visible (libbnmf has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(BNMF_ENODEVICE, "bnmf_create: device %d out of range (%d visible)", cfg->device, ndev);
  // model check: check_model R/bayesNMF_sampler.R:623-645
  if (cfg->likelihood == BNMF_NORMAL) {
    if (cfg->prior == BNMF_GAMMA) return fail(BNMF_EMODEL, "prior must be one of c('truncnormal','exponential') with `likelihood = 'normal'`");
  } else if (cfg->likelihood == BNMF_POISSON) {
    if (cfg->prior == BNMF_GAMMA && cfg->MH) return fail(BNMF_EMODEL, "gamma prior cannot be used in a MH-within-gibbs sampler");
    if (cfg->prior == BNMF_TRUNCNORMAL && !cfg->MH) return fail(BNMF_EMODEL, "truncnormal prior can only be used in a MH-within-gibbs sampler");
  } else return fail(BNMF_EMODEL, "likelihood must be one of normal, poisson");
  HIPCHK(hipSetDevice(cfg->device));
  if (int rc = ensure_alut(cfg->device)) return rc;
  bnmf_handle* h = new bnmf_handle();
  h->cfg = *cfg;
  h->device = cfg->device;
  if (int rc = create_impl(cfg, M, h)) {             // every failure path releases the handle and its device memory
    char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep);
    bnmf_destroy(h);
    memcpy(g_err, keep, sizeof keep);
    return rc;
  }
  *out = h;
  return 0;
}

// Static schedule of k_zalloc_sort (zalloc_sort.h): columns dealt into blocks of equal total count (largest column first,
// to the lightest block that still has room), the non-empty cells of a block as items sorted by their number of quads,
// 64 items per task.  M is fixed for the life of the handle, so this runs once.
static int build_zsort(bnmf_handle* h, const int32_t* M, int n_cu) {
  const bnmf_config& c = h->cfg;
  const size_t K = c.K, G = c.G, N = c.N;
  h->z_sort = false;
  if (!h->z_reg || N > (size_t)ZS_NMAX - 1 || K > 1024) return 0;
  // save_Z: a cell's counts per factor meet as 16-bit halves in k_zexpand's slab
  if (c.save_Z && h->maxM > 65535) return 0;
  if (const char* e = getenv("BNMF_ZSORT")) if (atoi(e) == 0) return 0;          // diagnostics / tests: the register kernel
  // an item word holds 16 bits of fragment index (k | gl << 10 | f << 16), and f = 65535 with k = 1023, gl = 63 is the empty-lane
  // sentinel: a cell above 65,534 fragments of 4 ZS_QMAX counts stays with the register kernel
  if ((long long)h->maxM > 65534LL * 4 * ZS_QMAX16) return 0;
  // Round 5: LARGE CELLS ARE SPREAD OVER THE BLOCKS.  A block's work is the counts of its columns, and the columns are dealt whole: a cell of
  // 10^6 counts (six times an average block at the metric configuration) made its block, and with it the launch, six times as long.  The
  // fragments of a cell above ZS_BIG counts beyond its first ZS_HOME are now "exported" in units of ZS_UNIT fragments to the lightest blocks,
  // which host the cell's column as a GUEST column (up to GX extra column slots per block: its A E products, a row of zK); Mhat is still left
  // by the lane of fragment 0, which stays at home.  ZsumK of a column then has several writers: every block adds its share with integer
  // atomics (exact, order-independent) and the draw kernels zero what they have consumed (Dev::zsumk_accum, as for the tile kernel).  Not with
  // save_Z (k_zexpand writes whole columns of Z per block).  The per-count work stays O(sum M) — the reference's rmultinom is O(N) per cell
  // (R/sample_params.R:263) — but a 10^7-count cell is 25 % more counts for the whole chip, not a 60-fold longer block.
  constexpr int ZS_BIG = 8192, ZS_HOME = 16, ZS_UNIT = 32;
  const bool spread = !c.save_Z && (long long)h->maxM > ZS_BIG && !(getenv("BNMF_ZSSPREAD") && atoi(getenv("BNMF_ZSSPREAD")) == 0);   // (tests: 0 = every cell at home)
  const int nblk = (int)((N + 4) / 5);                                             // threshold blocks per cell
  const int KP = (K % 32 == 0) ? (int)K + 1 : (int)(K | 1);
  size_t budget = 156 * 1024;                                                     // of 160: the side streams' workgroups (2 KB each) keep room on the CU
  if (const char* e = getenv("BNMF_ZSLDS")) budget = (size_t)atol(e) * 1024;
  long nb = std::min<long>((long)G, n_cu);
  int GBc = 0, W = 0;
  for (int tries = 0; tries < 12; ++tries, nb = std::min<long>((long)G, nb * 2)) {
    GBc = (int)((G + nb - 1) / nb);
    if (spread) GBc = std::min(64, GBc + 8);                                      // guest column slots
    if (GBc <= 64 && (long)nb * GBc >= (long)G) {
      const size_t sh = zsort_shared_bytes((int)K, (int)N, KP, GBc, false), wv = zsort_wave_bytes(nblk, (int)N);
      W = 0;
      for (int w : {16, 14, 12, 8, 6, 4}) if (sh + (size_t)w * wv <= budget) { W = w; break; }
      if (W) break;
    }
    if (nb >= (long)G) break;
  }
  if (!W || GBc > 64) return 0;
  if (const char* e = getenv("BNMF_ZSW")) { const int w = atoi(e); if (w == 4 || w == 6 || w == 8 || w == 12 || w == 14 || w == 16) W = w; }
  // columns -> blocks
  const bool it16_pre = K <= 127 && GBc <= 64 && (long long)h->maxM <= 8LL * 4 * ZS_QMAX16;
  const int qmax_pre = (it16_pre || (long long)h->maxM > 65534LL * 4 * ZS_QMAX) ? ZS_QMAX16 : ZS_QMAX;   // quads per fragment (the item format is fixed below: the same rule)
  struct Unit { int g, k, f0, nf; long counts; };
  std::vector<Unit> units;
  std::vector<long> ctot(G, 0), cfull(G, 0);
  for (size_t g = 0; g < G; ++g) {
    long sacc = 0, exported = 0;
    for (size_t k = 0; k < K; ++k) {
      const long m = M[k + K * g];
      sacc += m;
      if (spread && m > ZS_BIG) {
        const long qt = (m + 3) >> 2, F = (qt + qmax_pre - 1) / qmax_pre;
        for (long f0 = ZS_HOME; f0 < F; f0 += ZS_UNIT) {
          const long nf = std::min<long>(ZS_UNIT, F - f0);
          const long cnt = std::min<long>(m, (f0 + nf) * 4L * qmax_pre) - f0 * 4L * qmax_pre;
          units.push_back({(int)g, (int)k, (int)f0, (int)nf, cnt});
          exported += cnt;
        }
      }
    }
    cfull[g] = sacc; ctot[g] = sacc - exported;
  }
  std::vector<int> order(G);
  for (size_t g = 0; g < G; ++g) order[g] = (int)g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ctot[a] > ctot[b]; });
  std::vector<std::vector<int>> bcols(nb);
  std::vector<long> bload(nb, 0);
  const int own_cap = spread ? std::max(1, (int)((G + nb - 1) / nb)) : GBc;        // own columns per block (the rest of GBc: guest slots)
  {
    // min-heap of (load, block) over the blocks that still have room
    std::vector<std::pair<long, int>> heap;
    for (int b = 0; b < nb; ++b) heap.push_back({0L, b});
    auto cmp = [](const std::pair<long, int>& a, const std::pair<long, int>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (int g : order) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      auto top = heap.back(); heap.pop_back();
      bcols[top.second].push_back(g);
      top.first += ctot[g];
      if ((int)bcols[top.second].size() < own_cap) { heap.push_back(top); std::push_heap(heap.begin(), heap.end(), cmp); }
    }
  }
  for (int b = 0; b < nb; ++b) { std::sort(bcols[b].begin(), bcols[b].end()); long l = 0; for (int g : bcols[b]) l += ctot[g]; bload[b] = l; }
  // the exported units -> the lightest blocks (largest unit first); a block takes a unit if it owns the column, hosts it already, or has a
  // guest slot left; a unit nobody can take stays with its column's owner
  std::vector<std::vector<int>> gcols(nb);                                         // guest columns per block, in slot order
  struct BUnit { int k, gl, f0, nf; };
  std::vector<std::vector<BUnit>> bunits(nb);
  if (!units.empty()) {
    std::vector<int> owner(G, -1);
    for (int b = 0; b < nb; ++b) for (int g : bcols[b]) owner[g] = b;
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.counts > b.counts; });
    std::vector<std::pair<long, int>> heap;
    for (int b = 0; b < nb; ++b) heap.push_back({bload[b], b});
    auto cmp = [](const std::pair<long, int>& a, const std::pair<long, int>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    auto slot_of = [&](int b, int g, bool take) -> int {
      if (owner[g] == b) return (int)(std::lower_bound(bcols[b].begin(), bcols[b].end(), g) - bcols[b].begin());
      for (size_t i = 0; i < gcols[b].size(); ++i) if (gcols[b][i] == g) return (int)(bcols[b].size() + i);
      if (take && (int)(bcols[b].size() + gcols[b].size()) < GBc) { gcols[b].push_back(g); return (int)(bcols[b].size() + gcols[b].size() - 1); }
      return -1;
    };
    for (const Unit& u : units) {
      std::vector<std::pair<long, int>> skipped;
      int dst = -1, gl = -1;
      for (int tries = 0; tries < 16 && !heap.empty(); ++tries) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        auto top = heap.back(); heap.pop_back();
        gl = slot_of(top.second, u.g, true);
        if (gl >= 0) { dst = top.second; top.first += u.counts; heap.push_back(top); std::push_heap(heap.begin(), heap.end(), cmp); break; }
        skipped.push_back(top);
      }
      for (auto& x : skipped) { heap.push_back(x); std::push_heap(heap.begin(), heap.end(), cmp); }
      if (dst < 0) {                                                               // home: its load grows (the heap entry is found and raised)
        dst = owner[u.g]; gl = slot_of(dst, u.g, false);
        for (auto& x : heap) if (x.second == dst) x.first += u.counts;
        std::make_heap(heap.begin(), heap.end(), cmp);
      }
      bunits[dst].push_back({u.k, gl, u.f0, u.nf});
    }
  }
  // Quads per item.  Round 5 (end): chosen per data set where no cell is large enough to be spread.  A wave alone with its task runs a quad in
  // ~0.33 us (one dependent chain; tools/zstamps.py, tools/zsmall.py) and the waves of a SIMD share its issue at about twice that per wave
  // and quad: with few cells per block (K = 96, G = 2,000: 13 tasks for 14 waves) the kernel WAS its largest task — 64 quads with 2-byte
  // items, 20 of its 34 us.  Estimated per block in quad units, T = 6 for a task's thresholds: max(T + largest item, 2 (quads / 64 + T tasks)
  // / W); the candidate with the smallest worst block wins, the larger one on a near-tie (fewer items, fewer thresholds).  Measured, device
  // time of the kernel in us at K = 96, N = 20 with 64 / 32 / 16 / 8 / 4 quads per item: G = 250: 35.1 / 24.8 / 19.6 / 16.9 / 16.3; 1,000: 35.8 /
  // 25.3 / 20.2 / 19.6 / 21.2; 2,000: 36.2 / 26.5 / 23.3 / 24.0 / 29.1; 4,000: 37.2 / 31.6 / 30.1 / 34.0 / 44.1; 10,000: 52.7 / 53.4 / 56.7 / 69.0 /
  // 94.1.  The draws do not depend on it (Philox counter = cell, count index).  BNMF_ZSQMAX: tests.
  int qsel = 0;
  if (!spread && (long long)h->maxM <= ZS_BIG) {          // (above: the fragment index of a 4-byte item is 16 bits)
    const int cand[5] = {64, 32, 16, 8, 4};
    double worst[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
      long Q = 0, I[5] = {0, 0, 0, 0, 0}; int maxqt = 0;
      for (int g : bcols[b]) for (size_t k = 0; k < K; ++k) {
        const int m = M[k + K * (size_t)g], qt = m > 0 ? (m + 3) >> 2 : 0;
        Q += qt; maxqt = std::max(maxqt, qt);
        for (int c = 0; c < 5; ++c) I[c] += qt ? (qt + cand[c] - 1) / cand[c] : 1;
      }
      for (int c = 0; c < 5; ++c) {
        const double est = std::max(6.0 + std::min(cand[c], maxqt), 2.0 * ((double)Q / 64.0 + 6.0 * (double)((I[c] + 63) / 64)) / (double)W);
        worst[c] = std::max(worst[c], est);
      }
    }
    double best = 1e300;
    for (int c = 0; c < 5; ++c) if (worst[c] < 0.95 * best) { best = worst[c]; qsel = cand[c]; }
    if (const char* e = getenv("BNMF_ZSQMAX")) { const int v = atoi(e); if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) qsel = v; }
  }
  // 2-byte items where row, column-in-block and fragment index fit 7 + 6 + 3 bits (and 0xFFFF stays free for the empty lane)
  bool it16 = K <= 127 && GBc <= 64 && (long long)h->maxM <= 8LL * 4 * (qsel ? qsel : ZS_QMAX16);
  if (const char* e = getenv("BNMF_ZSIT16")) it16 = it16 && atoi(e) != 0;           // diagnostics / tests: 0 = 4-byte items
  // (large cells spread over the blocks: 4-byte items of 128 counts per fragment, 256 where a cell would need more than 65,534 of them)
  const int qmax = qsel ? qsel : (it16 || (long long)h->maxM > 65534LL * 4 * ZS_QMAX) ? ZS_QMAX16 : ZS_QMAX;
  std::vector<ZSBlock> blocks(nb);
  std::vector<int> cols;
  std::vector<uint32_t> items;
  // two factors per word in the block's zG / zK tables (16-bit halves): only if no half can overflow, i.e. every column total
  // (bound of a ZsumK entry) and every row total over a block's columns (bound of the block's share of a ZsumG entry) < 2^16
  bool pk = *std::max_element(cfull.begin(), cfull.end()) < 65536;                  // (the WHOLE column: units of a large cell may be dealt back to its owner)
  // the blocks are independent: their item lists are built by a few host threads (the schedule was 20 of the 50 ms of bnmf_create
  // at the metric configuration)
  std::vector<std::vector<uint32_t>> bitems(nb);
  std::vector<char> bpk(nb, 1);
  auto build_blocks = [&](long b0, long b1) {
    std::vector<std::pair<int, uint32_t>> tmp;
    for (long b = b0; b < b1; ++b) {
      for (size_t k = 0; k < K && bpk[b]; ++k) {
        long r = 0;
        for (int g : bcols[b]) r += M[k + K * (size_t)g];
        for (int g : gcols[b]) r += M[k + K * (size_t)g];                          // (a guest column's share: bounded by the whole cell)
        if (r >= 65536) bpk[b] = 0;
      }
      tmp.clear();
      for (size_t gl = 0; gl < bcols[b].size(); ++gl) {
        const size_t g = (size_t)bcols[b][gl];
        for (size_t k = 0; k < K; ++k) {
          const int m = M[k + K * g];
          if (m <= 0) { tmp.push_back({0, (uint32_t)k | ((uint32_t)gl << 10)}); continue; }   // an item without counts: its lane leaves Mhat of the cell (s.mh)
          const int qt = (m + 3) >> 2;
          const int fend = (spread && m > ZS_BIG) ? ZS_HOME : INT_MAX;                // a large cell: the fragments beyond the first ZS_HOME are units (below, or in other blocks)
          for (int f = 0; f * qmax < qt && f < fend; ++f)
            tmp.push_back({std::min(qmax, qt - f * qmax), (uint32_t)k | ((uint32_t)gl << 10) | ((uint32_t)f << 16)});
        }
      }
      for (const BUnit& u : bunits[b]) {                                             // units of large cells this block works on (its own columns' or guests')
        const size_t g = u.gl < (int)bcols[b].size() ? (size_t)bcols[b][u.gl] : (size_t)gcols[b][u.gl - (int)bcols[b].size()];
        const int m = M[u.k + K * g], qt = (m + 3) >> 2;
        for (int f = u.f0; f < u.f0 + u.nf && f * qmax < qt; ++f)
          tmp.push_back({std::min(qmax, qt - f * qmax), (uint32_t)u.k | ((uint32_t)u.gl << 10) | ((uint32_t)f << 16)});
      }
      std::stable_sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
      // inside a task (64 consecutive items) the order is free: ascending row, so that neighbouring lanes read neighbouring
      // rows of the LDS copy of P and add to neighbouring words of the block's ZsumG table (few bank conflicts)
      for (size_t i0 = 0; i0 < tmp.size(); i0 += 64)
        std::sort(tmp.begin() + i0, tmp.begin() + std::min(tmp.size(), i0 + 64), [](const auto& a, const auto& b) { return (a.second & 1023u) < (b.second & 1023u); });
      std::vector<uint32_t>& it = bitems[b];
      it.reserve(tmp.size() + 64);
      for (const auto& x : tmp) it.push_back(x.second);
      while (it.size() % 64) it.push_back(0xFFFFFFFFu);
    }
  };
  {
    const long nthr = std::max<long>(1, std::min<long>({(long)std::thread::hardware_concurrency(), 16L, nb}));
    std::vector<std::thread> pool;
    for (long i = 1; i < nthr; ++i) pool.emplace_back(build_blocks, nb * i / nthr, nb * (i + 1) / nthr);
    build_blocks(0, nb / nthr);
    for (auto& th : pool) th.join();
  }
  for (int b = 0; b < nb; ++b) pk = pk && bpk[b];
  if (const char* e = getenv("BNMF_ZSPK")) pk = pk && atoi(e) != 0;                 // diagnostics / tests: 0 = one factor per word
  // the waves per workgroup were sized for one factor per word (the block tables' larger form); with two per word the tables are half as
  // large and, at the metric configuration, 14 waves fit where 12 did.  tools/ablong.py, sixteen processes alternating on one box: 12 waves
  // 80.3 us per iteration in three of eight processes and 81.4-83.0 in the others (the stop-event mode of DESIGN.md 5b), 14 waves 81.3-81.6
  // in seven of eight (80.3 in one): 82.1 against 81.3 us on average
  if (pk && !getenv("BNMF_ZSW")) {
    const size_t sh = zsort_shared_bytes((int)K, (int)N, KP, GBc, true), wv = zsort_wave_bytes(nblk, (int)N);
    for (int w : {16, 14, 12, 8, 6, 4}) if (sh + (size_t)w * wv <= budget) { W = std::max(W, w); break; }
  }
  std::vector<int32_t> Mblk(K * G);
  for (int b = 0; b < nb; ++b) {
    ZSBlock& bk = blocks[b];
    bk.item0 = (int)items.size(); bk.col0 = (int)cols.size(); bk.ncols = (int)(bcols[b].size() + gcols[b].size());
    items.insert(items.end(), bitems[b].begin(), bitems[b].end());
    bk.ntask = (int)(bitems[b].size() / 64);
    for (int pass = 0; pass < 2; ++pass)
      for (int g : (pass ? gcols[b] : bcols[b])) {
        if (Mblk.size() < K * (cols.size() + 1)) Mblk.resize(K * (cols.size() + 1));
        memcpy(Mblk.data() + K * cols.size(), M + K * (size_t)g, K * sizeof(int32_t)); cols.push_back(g);
      }
  }
  if (items.empty()) items.push_back(0xFFFFFFFFu);
  if (it16) {
    std::vector<uint16_t> i16(items.size());
    for (size_t i = 0; i < items.size(); ++i) {
      const uint32_t v = items[i];
      i16[i] = v == 0xFFFFFFFFu ? (uint16_t)0xFFFFu : (uint16_t)((v & 127u) | (((v >> 10) & 63u) << 7) | ((v >> 16) << 13));
    }
    HIPCHK(dmalloc(&h->dZsItems, ((i16.size() * sizeof(uint16_t) + 3) & ~(size_t)3)));
    HIPCHK(hipMemcpy(h->dZsItems, i16.data(), i16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  } else {
    HIPCHK(dmalloc(&h->dZsItems, items.size() * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(h->dZsItems, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  h->zs_it16 = it16 ? 1 : 0; h->zs_qmax = qmax;
  HIPCHK(dmalloc(&h->dZsBlocks, blocks.size() * sizeof(ZSBlock)));
  HIPCHK(hipMemcpy(h->dZsBlocks, blocks.data(), blocks.size() * sizeof(ZSBlock), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dZsCols, cols.size() * sizeof(int)));
  HIPCHK(hipMemcpy(h->dZsCols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dZsM, Mblk.size() * sizeof(int32_t)));
  HIPCHK(hipMemcpy(h->dZsM, Mblk.data(), Mblk.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  // three copies, iteration t in copy t % 3 (as the per-column partial sums): the terms of t are summed beside the allocation kernel of t + 1
  // (k_side_lp of t + 2, stream side2), which is known to be over before the draw kernel of t + 3 starts — the allocation kernel of t + 2
  // has waited for the flag of the k_side_lp behind it on that stream
  HIPCHK(dmalloc(&h->dZsMh, 3 * K * G * sizeof(double)));
  HIPCHK(hipMemset(h->dZsMh, 0, 3 * K * G * sizeof(double)));
  if (c.save_Z) {
    const size_t hw = (N + 1) / 2;
    h->zs_recwords = items.size() * hw;
    HIPCHK(dmalloc(&h->dZsRec, h->zs_recwords * sizeof(uint32_t)));
    h->zs_eager = getenv("BNMF_ZEAGER") && atoi(getenv("BNMF_ZEAGER")) != 0;
    const size_t lds_max = 160 * 1024;
    h->zx_cols = std::max(1, std::min(GBc, zexpand_cols((int)K, (int)N, lds_max)));
    h->zx_lds = ((size_t)h->zx_cols * N * ((K + 1) / 2) * 4 + 15) & ~(size_t)15;
    if (h->zx_lds > lds_max) return fail(BNMF_EINVAL, "bnmf_create: K = %zu, N = %zu: a column of Z does not fit the LDS of k_zexpand", K, N);
    HIPCHK(hipFuncSetAttribute((const void*)k_zexpand, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  h->zsg = ZSGeom{KP, GBc, (int)nb};
  h->zs_shared = !units.empty();                           // columns with several writers of ZsumK: atomics + zeroing by the consumer (refresh_dev)
  h->zs_nblk = nblk; h->zs_w = W; h->zs_pk = pk;
  h->zs_lds = (zsort_shared_bytes((int)K, (int)N, KP, GBc, pk) + (size_t)W * zsort_wave_bytes(nblk, (int)N) + 15) & ~(size_t)15;
  h->z_sort = true;
#ifdef ZSPROF
  HIPCHK(dmalloc(&h->dZsProf, (8 + 3 * 16 * 8) * sizeof(unsigned long long)));   // section ticks, then the stamps of three blocks' waves (-DZSPROF)
  HIPCHK(hipMemset(h->dZsProf, 0, (8 + 3 * 16 * 8) * sizeof(unsigned long long)));
#endif
  return 0;
}

// Static schedule of k_zalloc_step (zalloc_step.h): columns dealt to the workgroups by total count (largest first, to the
// lightest workgroup that still has room), a workgroup's columns cut into batches of <= GBP, a batch's rows into chunks of
// 32; the cells of a step (chunk x batch) as items — zero-count cells too: their Mhat feeds the metric terms — sorted by
// their number of quads (counting sort) and dealt to the workgroup's waves in snake order, 64 per task: the waves of a step
// get the same number of items of the same sizes.  M is fixed for the life of the handle, so this runs once.
static int build_zstep(bnmf_handle* h, const int32_t* M, int n_cu) {
  const bnmf_config& c = h->cfg;
  const size_t K = c.K, G = c.G, N = c.N;
  h->z_step = false;
  if (c.save_Z || N <= (size_t)ZNMAX || N > (size_t)ZP_NMAX) return 0;
  if (const char* e = getenv("BNMF_ZSTEP")) if (atoi(e) == 0) return 0;            // diagnostics / tests: the tile kernel
  const int L = 4;                                                                 // lanes per cell (with <= 20 included factors the search then skips a level)
  size_t budget = 156 * 1024;                                                      // of 160: the side streams' workgroups keep room on the CU
  int GBP = 0, W = 0;
  // 8 waves (two per SIMD).  12 waves fit the LDS up to N = 60 and were measured at config 4: 114.5 against 121 us per launch, but
  // at the 168 registers three waves per SIMD leave, the kernel spills 16-36 bytes per lane — not kept
  for (int gbp : {40, 32}) if (zstep_shared_bytes((int)N, gbp) + 8 * zstep_wave_bytes(L) <= budget) { GBP = gbp; W = 8; break; }
  if (const char* e = getenv("BNMF_ZPGB")) { const int v = atoi(e); if (v == 32 || v == 40) GBP = v; }   // diagnostics / tests
  if (!GBP) return 0;
  const int nch = (int)((K + ZP_KC - 1) / ZP_KC);
  const long nwg = std::min<long>((long)G, n_cu);
  // columns -> workgroups
  std::vector<long> ctot(G, 0);
  for (size_t g = 0; g < G; ++g) { long sacc = 0; for (size_t k = 0; k < K; ++k) sacc += M[k + K * g]; ctot[g] = sacc; }
  std::vector<int> order(G);
  for (size_t g = 0; g < G; ++g) order[g] = (int)g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ctot[a] > ctot[b]; });
  const size_t cap = (G + nwg - 1) / nwg + 2;                                      // the threshold work of a column does not depend on its counts
  std::vector<std::vector<int>> wcols(nwg);
  {
    std::vector<std::pair<long, int>> heap;
    for (int b = 0; b < nwg; ++b) heap.push_back({0L, b});
    auto cmp = [](const std::pair<long, int>& a, const std::pair<long, int>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (int g : order) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      auto top = heap.back(); heap.pop_back();
      wcols[top.second].push_back(g);
      top.first += ctot[g] + 64 * (long)K;                                         // + the cells' fixed cost (Mhat, thresholds), in counts
      if (wcols[top.second].size() < cap) { heap.push_back(top); std::push_heap(heap.begin(), heap.end(), cmp); }
    }
  }
  std::vector<ZPWg> wgs(nwg);
  std::vector<ZPBatch> batches;
  std::vector<int> cols;
  for (int b = 0; b < nwg; ++b) {
    std::sort(wcols[b].begin(), wcols[b].end());
    const int nc = (int)wcols[b].size(), nb = (nc + GBP - 1) / GBP;
    wgs[b] = ZPWg{(int)batches.size(), nb};
    for (int i = 0; i < nb; ++i) {
      const int c0 = (int)((long)nc * i / nb), c1 = (int)((long)nc * (i + 1) / nb);
      batches.push_back(ZPBatch{(int)cols.size(), c1 - c0});
      for (int x = c0; x < c1; ++x) cols.push_back(wcols[b][x]);
    }
  }
  std::vector<ZPStep> steps(batches.size() * (size_t)nch);
  std::vector<uint32_t> items;
  items.reserve((size_t)((double)K * (double)G * 1.05) + 64 * steps.size());
  std::vector<uint32_t> bucket[ZP_QMAX + 1], wlist[ZP_WMAX], sorted;
  int maxfrag = 0;
  for (size_t bi = 0; bi < batches.size(); ++bi) {
    const ZPBatch& bt = batches[bi];
    for (int ch = 0; ch < nch; ++ch) {
      const size_t k0 = (size_t)ch * ZP_KC, kc = std::min<size_t>(ZP_KC, K - k0);
      for (auto& v : bucket) v.clear();
      for (int gl = 0; gl < bt.ncols; ++gl) {
        const size_t g = (size_t)cols[bt.col0 + gl];
        for (size_t kl = 0; kl < kc; ++kl) {
          const int m = M[k0 + kl + K * g];
          const int qt = m > 0 ? (m + 3) >> 2 : 0;
          const uint32_t base = (uint32_t)kl | ((uint32_t)gl << 5);
          if (qt == 0) { bucket[0].push_back(base); continue; }
          for (int f = 0; f * ZP_QMAX < qt; ++f) {
            if (f >= (1 << 21)) return fail(BNMF_EINVAL, "bnmf_create: a cell of M holds %d counts: unsupported", m);
            bucket[std::min(ZP_QMAX, qt - f * ZP_QMAX)].push_back(base | ((uint32_t)f << 11));
            maxfrag = std::max(maxfrag, f);
          }
        }
      }
      // sorted by size, then dealt to the waves in snake order: every wave gets the same number of items (+-1) of the same sizes
      sorted.clear();
      for (int qn = ZP_QMAX; qn >= 0; --qn) sorted.insert(sorted.end(), bucket[qn].begin(), bucket[qn].end());
      for (int w = 0; w < W; ++w) wlist[w].clear();
      for (size_t i = 0; i < sorted.size(); ++i) { const int r = (int)(i % (2 * (size_t)W)); wlist[r < W ? r : 2 * W - 1 - r].push_back(sorted[i]); }
      size_t mx = 0;
      for (int w = 0; w < W; ++w) mx = std::max(mx, wlist[w].size());
      ZPStep& st = steps[bi * (size_t)nch + ch];
      st.item0 = (long long)items.size(); st.pad = 0;
      st.ntw = (int)((mx + 63) / 64);
      for (int w = 0; w < W; ++w) { const auto& v = wlist[w]; items.insert(items.end(), v.begin(), v.end()); items.insert(items.end(), (size_t)st.ntw * 64 - v.size(), 0xFFFFFFFFu); }
    }
  }
  if (items.empty()) items.push_back(0xFFFFFFFFu);
  // 2-byte items where the fragment index fits 5 bits beside row (5) and column (6), 0xFFFF staying the empty lane (column 63 does not
  // occur): BNMF_ZPIT16=0 keeps the 4-byte form (diagnostics / tests)
  h->zp_it16 = maxfrag <= 30 && !(getenv("BNMF_ZPIT16") && atoi(getenv("BNMF_ZPIT16")) == 0);
  if (h->zp_it16) {
    std::vector<uint16_t> i16(items.size());
    for (size_t i = 0; i < items.size(); ++i) i16[i] = items[i] == 0xFFFFFFFFu ? (uint16_t)0xFFFFu : (uint16_t)items[i];
    HIPCHK(dmalloc(&h->dZpItems, (i16.size() * sizeof(uint16_t) + 3) & ~(size_t)3));
    HIPCHK(hipMemcpy(h->dZpItems, i16.data(), i16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  } else {
    HIPCHK(dmalloc(&h->dZpItems, items.size() * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(h->dZpItems, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(dmalloc(&h->dZpWgs, wgs.size() * sizeof(ZPWg)));
  HIPCHK(hipMemcpy(h->dZpWgs, wgs.data(), wgs.size() * sizeof(ZPWg), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dZpBatches, batches.size() * sizeof(ZPBatch)));
  HIPCHK(hipMemcpy(h->dZpBatches, batches.data(), batches.size() * sizeof(ZPBatch), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dZpSteps, steps.size() * sizeof(ZPStep)));
  HIPCHK(hipMemcpy(h->dZpSteps, steps.data(), steps.size() * sizeof(ZPStep), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dZpCols, cols.size() * sizeof(int)));
  HIPCHK(hipMemcpy(h->dZpCols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice));
  h->zpg = ZPGeom{nch, (int)nwg, nullptr};
#ifdef ZPPROF
  HIPCHK(dmalloc(&h->zpg.prof, 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(h->zpg.prof, 0, 8 * sizeof(unsigned long long)));
#endif
  h->zp_ns = W; h->zp_gbp = GBP;
  h->zp_lds = (zstep_shared_bytes((int)N, GBP) + W * zstep_wave_bytes(L) + 15) & ~(size_t)15;
  h->z_step = true;
  return 0;
}

// A handle's three streams come from a per-device pool and go back to it at bnmf_destroy: creating a stream and bringing its
// hardware queue up at the first submission cost 15-30 ms of a 50 ms bnmf_create (a BIC sweep creates one handle per rank).
static std::mutex g_stream_mtx;
static std::vector<hipStream_t> g_stream_pool[64];
// (Round 5, measured and NOT adopted: stream priorities.  The steady iteration is 80.6 us in most processes and 83.3 us in about a quarter
// of them — in those the side-stream kernels run faster and the allocation kernel slower: how the queues' dispatches interleave differs
// from process to process (tools/ablong.py, tools/bimodal.sh).  hipStreamCreateWithPriority with the main stream high and the side streams
// low — or the reverse — makes the slow order the common one, and all three streams in one non-default class puts them on one hardware
// queue: 106-111 us.  Plain streams it is; the allocation kernel raises its waves' issue priority instead: zalloc_sort.h.)
static int take_stream(int device, hipStream_t* out, int kind = 0) {
  (void)kind;
  if (device < 0 || device >= 64) return fail(BNMF_EINVAL, "device ordinal %d out of range", device);
  {
    std::lock_guard<std::mutex> lock(g_stream_mtx);
    auto& v = g_stream_pool[device];
    if (!v.empty()) { *out = v.back(); v.pop_back(); return 0; }
  }
  HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return 0;
}
static void give_stream(int device, hipStream_t st, int kind = 0) {
  (void)kind;
  if (!st) return;
  if (device < 0 || device >= 64) { hipStreamDestroy(st); return; }
  std::lock_guard<std::mutex> lock(g_stream_mtx);
  g_stream_pool[device].push_back(st);
}
// The record_sample rings (gigabytes per handle) go back to a per-device cache at bnmf_destroy and are handed to the next handle
// that asks for the same size: releasing them and mapping new ones stalled the first kernel of the next bnmf_create by 20-30 ms (a
// BIC sweep, or bayesNMF() called in a loop, creates handle after handle of the same shape).  At most BNMF_RING_CACHE_GB (default
// 32, 0 = off) stay cached per device, oldest out first.
struct CachedBuf { void* p; size_t bytes; };
static std::mutex g_ring_mtx;
static std::vector<CachedBuf> g_ring_cache[64];
// What a destroyed handle leaves cached per device: its record_sample rings, for the next handle of the same shape (bayesNMF() after
// bayesNMF(), a BIC sweep).  Round 5: 8 GB by default (it was 32: room for one ring set of the metric configuration, 6.5 GB,
// not for several), dropped whenever a device allocation of the library fails (the allocation is then tried again), and bnmf_trim()
// releases it — and the pooled streams — on request (the library usually shares its process with PyTorch / RCCL).
static size_t ring_cache_cap() { const char* e = getenv("BNMF_RING_CACHE_GB"); return (size_t)((e ? atof(e) : 8.0) * 1e9); }
static size_t ring_cache_drop(int device) {
  std::vector<CachedBuf> drop;
  { std::lock_guard<std::mutex> lock(g_ring_mtx); if (device >= 0 && device < 64) drop.swap(g_ring_cache[device]); }
  size_t tot = 0;
  for (const auto& c : drop) { hipFree(c.p); tot += c.bytes; }
  return tot;
}
static int ring_alloc(int device, size_t bytes, double** out) {
  if (device < 0 || device >= 64) return fail(BNMF_EINVAL, "device ordinal %d out of range", device);
  {
    std::lock_guard<std::mutex> lock(g_ring_mtx);
    auto& v = g_ring_cache[device];
    for (size_t i = v.size(); i-- > 0;) if (v[i].bytes == bytes) { *out = (double*)v[i].p; v.erase(v.begin() + (long)i); return 0; }
  }
  if (hipMalloc(out, bytes) != hipSuccess) {               // out of memory with rings of other shapes cached: give them back, try once more
    (void)hipGetLastError();
    ring_cache_drop(device);
    HIPCHK(hipMalloc(out, bytes));
  }
  return 0;
}
static void ring_release(int device, void* p, size_t bytes) {
  if (!p) return;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lock(g_ring_mtx);
    auto& v = g_ring_cache[device];
    const size_t cap = ring_cache_cap();
    if (bytes > cap || bytes < ((size_t)1 << 20)) drop.push_back(p);        // small rings are not worth keeping
    else {
      v.push_back({p, bytes});
      size_t tot = 0;
      for (const auto& c : v) tot += c.bytes;
      while (tot > cap && !v.empty()) { tot -= v.front().bytes; drop.push_back(v.front().p); v.erase(v.begin()); }
    }
  }
  for (void* q : drop) hipFree(q);
}
// ---- who may use a device at the same time ----
// A rank-learning call takes its device exclusively, every other bnmf_run shares it (why: run_impl).  Round 5: the
// in-process gate is WRITER-PREFERRING — a std::shared_mutex of glibc admits new readers while a writer waits, so a rank-learning
// chain beside chains that keep issuing overlapping calls could wait for ever.  Here a waiting exclusive caller closes the gate for
// new sharers; those already inside finish their call (at most one block of iterations).
struct DeviceGate {
  std::mutex m; std::condition_variable cv;
  int sharers = 0, waiting_excl = 0; bool excl = false;
  unsigned long admitted_while_excl_waited = 0;           // must stay 0 (bnmf_test_gate)
  void lock_shared() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !excl && waiting_excl == 0; });
    if (waiting_excl) ++admitted_while_excl_waited;
    ++sharers;
  }
  void unlock_shared() { std::lock_guard<std::mutex> lk(m); if (--sharers == 0) cv.notify_all(); }
  void lock() {
    std::unique_lock<std::mutex> lk(m);
    ++waiting_excl;
    cv.wait(lk, [&] { return !excl && sharers == 0; });
    --waiting_excl; excl = true;
  }
  void unlock() { std::lock_guard<std::mutex> lk(m); excl = false; cv.notify_all(); }
};
static DeviceGate g_dev_gate[64];
// The same rule between PROCESSES: two lock files per device.  <dir>/bnmf_dev_<bus>.lock is held shared / exclusive for the length of
// a call; <dir>/bnmf_dev_<bus>.gate is the turnstile that makes it fair: an exclusive caller holds the gate exclusively from before
// it asks for the lock until it has it, a sharer passes through the gate (shared, released at once) before it asks — so sharers
// that arrive while an exclusive caller waits queue behind it (flock alone lets them overtake).
// The files: an existing one is opened read-only (flock needs no write access; O_CREAT on a file of another user in a sticky /tmp is
// refused under fs.protected_regular); a missing one is created and made world-accessible.  BNMF_LOCKDIR replaces /tmp (containers
// that share a GPU but not /tmp).  Failure is said once on stderr and remembered in the handle (run_impl refuses rank learning then).
static int open_lock_file(const char* path) {
  for (int attempt = 0; attempt < 3; ++attempt) {
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd >= 0) return fd;
    if (errno != ENOENT) return -1;
    fd = open(path, O_RDWR | O_CREAT | O_EXCL | O_CLOEXEC, 0666);
    if (fd >= 0) { (void)fchmod(fd, 0666); return fd; }
    if (errno != EEXIST) return -1;                        // (EEXIST: another process has just created it — open it)
  }
  return -1;
}
static void devlock_open(const char* bus, int* lock_fd, int* gate_fd) {
  const char* dir = getenv("BNMF_LOCKDIR");
  if (!dir || !*dir) dir = "/tmp";
  char path[512];
  snprintf(path, sizeof path, "%s/bnmf_dev_%s.lock", dir, bus);
  *lock_fd = open_lock_file(path);
  const int e1 = errno;
  snprintf(path, sizeof path, "%s/bnmf_dev_%s.gate", dir, bus);
  *gate_fd = *lock_fd >= 0 ? open_lock_file(path) : -1;
  if (*lock_fd < 0 || *gate_fd < 0) {
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
      fprintf(stderr, "[bnmf] warning: cannot open the device lock files %s/bnmf_dev_%s.{lock,gate} (%s): chains of OTHER processes on this device are not "
                      "kept apart from rank-learning chains; set BNMF_LOCKDIR to a directory all of them can write, or BNMF_DEVLOCK=0 if there are none\n",
              dir, bus, strerror(*lock_fd < 0 ? e1 : errno));
    if (*lock_fd >= 0) { close(*lock_fd); *lock_fd = -1; }
  }
}
// host-only checks of the two mechanisms above (tests/test_abi_host.py; no GPU involved)
extern "C" int bnmf_test_gate(int n_sharers, int calls_per_sharer, int hold_us, long* excl_wait_us, long* admitted_while_waiting) {
  // sharers keep the gate busy with overlapping calls (a reader-preferring lock would never let the exclusive caller in before they
  // stop); the exclusive caller asks once they are all running.  Reports how long it waited and how many sharers were admitted
  // while it did (must be 0).
  DeviceGate g;
  std::atomic<int> running{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> th;
  for (int i = 0; i < n_sharers; ++i)
    th.emplace_back([&, i] {
      for (int c = 0; c < calls_per_sharer && !stop.load(); ++c) {
        g.lock_shared();
        running.fetch_add(1);
        std::this_thread::sleep_for(std::chrono::microseconds(hold_us + 37 * i));
        g.unlock_shared();
      }
    });
  while (running.load() < n_sharers) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  g.lock();
  const long waited = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  stop.store(true);
  g.unlock();
  for (auto& t : th) t.join();
  if (excl_wait_us) *excl_wait_us = waited;
  if (admitted_while_waiting) *admitted_while_waiting = (long)g.admitted_while_excl_waited;
  return 0;
}
extern "C" int bnmf_test_devlock(const char* bus_tag, int* lock_ok, int* gate_ok) {
  int a = -1, b = -1;
  devlock_open(bus_tag ? bus_tag : "test", &a, &b);
  if (lock_ok) *lock_ok = a >= 0;
  if (gate_ok) *gate_ok = b >= 0;
  int rc = 0;
  if (a >= 0 && b >= 0) {                                  // the turnstile order of run_impl, both kinds of caller
    if (flock(b, LOCK_EX) || flock(a, LOCK_EX) || flock(b, LOCK_UN) || flock(a, LOCK_UN)) rc = -1;
    if (flock(b, LOCK_SH) || flock(b, LOCK_UN) || flock(a, LOCK_SH) || flock(a, LOCK_UN)) rc = -1;
  }
  if (a >= 0) close(a);
  if (b >= 0) close(b);
  return rc;
}

// ---- can two kernels on two streams of this device run at the same time?  (once per device and process) ----
// The steady-state sweeps hand results from the side streams to the main stream through words in memory that a lane of the
// main-stream kernel polls (kernels.h side_wait, the allocation kernels' gate): that needs the polled-for kernel to be able to
// START while the polling one is resident.  A runtime or tool that serialises dispatches (counter collection, AMD_SERIALIZE_KERNEL,
// HIP_LAUNCH_BLOCKING, a debugger, ...) breaks it — whatever its name.  The probe: kernel A on one stream spins (bounded: 2 ms)
// on a word that kernel B on another stream sets; A reports whether it saw the word.  Overlap -> flag polling; no overlap -> every
// hand-off is a stream wait on an event (serial-safe mode).  BNMF_DEBUG_PROBE=serial|overlap (tests) replaces the measurement.
__global__ void k_probe_wait(unsigned* word, unsigned* out, long long ticks_100mhz) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  unsigned seen = 0;
  while (!(seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > ticks_100mhz) break;
    __builtin_amdgcn_s_sleep(8);
  }
  *out = seen ? 1u : 2u;
}
__global__ void k_probe_set(unsigned* word) { __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static std::mutex g_probe_mtx;
static int g_probe[64];   // 0 unknown, 1 overlap, 2 serial
static int probe_overlap(int device, int* overlap) {
  std::lock_guard<std::mutex> lk(g_probe_mtx);
  if (const char* e = getenv("BNMF_DEBUG_PROBE")) {            // tests: the probe's outcome, not the mode (BNMF_SERIAL is the caller's switch)
    if (!strcmp(e, "serial")) { *overlap = 0; return 0; }
    if (!strcmp(e, "overlap")) { *overlap = 1; return 0; }
  }
  if (device < 0 || device >= 64) { *overlap = 0; return 0; }
  if (!g_probe[device]) {
    hipStream_t a = nullptr, b = nullptr;
    if (int rc = take_stream(device, &a)) return rc;
    if (int rc = take_stream(device, &b)) { give_stream(device, a); return rc; }
    unsigned* w = nullptr;
    HIPCHK(dmalloc(&w, 2 * sizeof(unsigned)));
    HIPCHK(hipMemset(w, 0, 2 * sizeof(unsigned)));
    HIPCHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, w, w + 1, 200000LL);   // 2 ms of the 100 MHz counter
    hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(a));
    HIPCHK(hipStreamSynchronize(b));
    unsigned res[2] = {0, 0};
    HIPCHK(hipMemcpy(res, w, sizeof res, hipMemcpyDeviceToHost));
    dfree(w);
    give_stream(device, a); give_stream(device, b);
    g_probe[device] = res[1] == 1u ? 1 : 2;
    if (getenv("BNMF_TIMING")) fprintf(stderr, "[bnmf] device %d: kernels on two streams %s\n", device, g_probe[device] == 1 ? "overlap" : "do NOT overlap: serial-safe mode");
  }
  *overlap = g_probe[device] == 1;
  return 0;
}
extern "C" int bnmf_trim(int device, size_t* bytes_released) {
  if (device < 0 || device >= 64) return fail(BNMF_EINVAL, "bnmf_trim: device ordinal %d out of range", device);
  HIPCHK(hipSetDevice(device));
  const size_t b = ring_cache_drop(device) + pool_drop(device);
  std::vector<hipStream_t> st;
  { std::lock_guard<std::mutex> lock(g_stream_mtx); st.swap(g_stream_pool[device]); }
  for (hipStream_t q : st) hipStreamDestroy(q);
  if (bytes_released) *bytes_released = b;
  return 0;
}
extern "C" int bnmf_probe_overlap(int device, int* overlap) {
  if (!overlap) return fail(BNMF_EINVAL, "bnmf_probe_overlap: null argument");
  HIPCHK(hipSetDevice(device));
  return probe_overlap(device, overlap);
}

static int create_impl(const bnmf_config* cfg, const int32_t* M, bnmf_handle* h) {
  const size_t K = cfg->K, G = cfg->G, N = cfg->N;
  CreateClock clk;
  if (int rc = take_stream(h->device, &h->stream)) return rc;
  if (int rc = take_stream(h->device, &h->side, 1)) return rc;
  if (int rc = take_stream(h->device, &h->side2, 1)) return rc;
  clk.mark("streams");
  // one open file description per handle: flock() then also separates the handles of ONE process (BNMF_DEVLOCK=0: no file lock)
  h->devlock_off = getenv("BNMF_DEVLOCK") && atoi(getenv("BNMF_DEVLOCK")) == 0;
  if (!h->devlock_off) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, h->device) == hipSuccess) {
      for (char* c = bus; *c; ++c) if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z'))) *c = '_';
      devlock_open(bus, &h->devlock_fd, &h->devgate_fd);
    } else (void)hipGetLastError();
  }
  HIPCHK(hipEventCreateWithFlags(&h->ev_draw, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_sideP, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_p, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_rank, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_z, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&h->ev_red, hipEventDisableTiming | hipEventDisableSystemFence));
  clk.mark("events");
  HIPCHK(dmalloc(&h->dM, K * G * sizeof(int32_t)));
  HIPCHK(hipMemcpy(h->dM, M, K * G * sizeof(int32_t), hipMemcpyHostToDevice));
  clk.mark("M to the device");
  int mx = 0;
  for (size_t i = 0; i < K * G; ++i) { if (M[i] < 0) return fail(BNMF_EINVAL, "bnmf_create: negative count in M"); if (M[i] > mx) mx = M[i]; }
  h->maxM = mx;
  HIPCHK(dmalloc(&h->dZsumK, N * G * sizeof(int32_t)));
  HIPCHK(dmalloc(&h->dZsumG, K * N * sizeof(int32_t)));
  HIPCHK(hipMemset(h->dZsumK, 0, N * G * sizeof(int32_t)));
  HIPCHK(hipMemset(h->dZsumG, 0, K * N * sizeof(int32_t)));
  if (cfg->save_Z) HIPCHK(dmalloc(&h->dZ, K * N * G * sizeof(int32_t)));
  if (const char* e = getenv("BNMF_GATE")) h->gate_forced = atoi(e) != 0 ? 1 : 0;   // diagnostics / tests
  if (const char* e = getenv("BNMF_MHSIDE")) h->mh_side_main = atoi(e) != 0;
  if (const char* e = getenv("BNMF_MHSIDETAIL")) h->mh_side_tail = atoi(e) != 0;
  if (const char* e = getenv("BNMF_DEBUG_DRAW_NO_P")) h->dbg_draw_no_p = atoi(e) != 0 ? 1 : 0;   // tests only
  if (const char* e = getenv("BNMF_DEBUG_MAIN_DELAY_US")) h->dbg_main_delay_us = std::max(0, std::min(20000, atoi(e)));   // tests only
  if (const char* e = getenv("BNMF_DEBUG_ALLSIDE_DELAY_US")) h->dbg_allside_delay_us = std::max(0, std::min(20000, atoi(e)));   // tests only
  if (const char* e = getenv("BNMF_DEBUG_SIDE_DELAY_US")) h->dbg_side_delay_us = std::max(0, std::min(20000, atoi(e)));   // tests only
  {
    // A lane that polls inside a main-stream kernel for a side-stream kernel deadlocks (until its bound) when dispatches cannot
    // overlap: the waited-for kernel only starts once the waiting one has ended.  Where the process is known to serialise its
    // dispatches, every hand-off is a stream wait on an event instead (the structure profile mode has always used).
    // Round 5: MEASURED, not guessed from the environment (rounds 3-4 looked for AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING and the
    // profiler's variables; any other serialising tool ended in a time-out): probe_overlap runs a two-stream hand-off once per device.
    int ov = 0;
    if (int rc = probe_overlap(h->device, &ov)) return rc;
    h->serial = ov == 0;
    if (const char* e = getenv("BNMF_SERIAL")) h->serial = atoi(e) != 0;      // the caller's explicit choice
  }
  HIPCHK(dmalloc(&h->dFlags, 64));
  HIPCHK(hipMemset(h->dFlags, 0, 64));
  HIPCHK(dmalloc(&h->dScal, BNMF_ID_MAX * sizeof(double)));
  HIPCHK(dmalloc(&h->dDrawOwn, N * sizeof(unsigned)));
  HIPCHK(hipMemset(h->dDrawOwn, 0, N * sizeof(unsigned)));
  HIPCHK(hipHostMalloc((void**)&h->hErr, 64, hipHostMallocMapped));
  memset(h->hErr, 0, 64);
  HIPCHK(hipHostGetDevicePointer((void**)&h->dErr, h->hErr, 0));
  HIPCHK(dmalloc(&h->dR, sizeof(int)));
  int Rinit = (int)N;
  HIPCHK(hipMemcpy(h->dR, &Rinit, sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(dmalloc(&h->dRedraw, N * sizeof(int)));
  HIPCHK(dmalloc(&h->dEsum, N * sizeof(double)));
  HIPCHK(dmalloc(&h->dPsum, N * sizeof(double)));
  HIPCHK(dmalloc(&h->dlpPn, 3 * N * sizeof(double)));
  h->nblkE = (int)((N * G + ES_T - 1) / ES_T);
  HIPCHK(dmalloc(&h->dlpE, 3 * (size_t)h->nblkE * sizeof(double)));
  HIPCHK(dmalloc(&h->dcol, 3 * 3 * G * sizeof(double)));   // per-column partials, 3 slots (t % 3)
  if (cfg->learning_rank) {
    const size_t gran_words = (size_t)RK_REP * 4 * 2 * ((G + RK_MAXC - 1) / RK_MAXC);   // [RK_REP copies][4 buffers][2 granules per block sum]
    HIPCHK(dmalloc(&h->dRankCol, gran_words * sizeof(double)));
    HIPCHK(hipMemset(h->dRankCol, 0, gran_words * sizeof(double)));           // tag 0 is never used
    HIPCHK(dmalloc(&h->dRankSync, 32));
    HIPCHK(hipMemset(h->dRankSync, 0, 32));
    // grid of the persistent rank sweep: co-resident by construction (one 512-lane workgroup per CU)
    hipDeviceProp_t prop0;
    HIPCHK(hipGetDeviceProperties(&prop0, cfg->device));
    const long NB = ((long)G + RK_MAXC - 1) / RK_MAXC;                 // blocks of 8 columns, one compute wave each
    const long wg_needed = (NB + RK_CW - 1) / RK_CW;                   // RK_CW compute waves + the decision wave per workgroup
    // every workgroup of the sweep waits for all others: the grid must be co-resident.  Ask the runtime how many
    // workgroups of each variant fit a CU (registers, LDS); where the answer is SGPR-limited (>= 6 per CU) the query can
    // be one high (MI355X_MICROARCH.md), so one is kept in reserve there; never plan more than two per CU.  Should the
    // grid still not be co-resident, the bounded spins time out and bnmf_run reports it (no hang).
    const size_t rlds = (3 * (size_t)N + 1) * sizeof(double);
    auto fit = [&](const void* fn) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, RK_T, rlds) != hipSuccess || nb < 1) nb = 1;
      return std::min(1, nb >= 6 ? nb - 1 : nb);          // one per CU: even compute times, and co-resident with margin
    };
    const bool nrm0 = cfg->likelihood == BNMF_NORMAL;
    const long cap_reg = (long)fit(nrm0 ? (const void*)k_rank_sweep<true, true> : (const void*)k_rank_sweep<true, false>) * prop0.multiProcessorCount;
    const long cap_gen = (long)fit(nrm0 ? (const void*)k_rank_sweep<false, true> : (const void*)k_rank_sweep<false, false>) * prop0.multiProcessorCount;
    h->rank_reg = K <= 96 && wg_needed <= cap_reg;       // register variant: rows 64..95 of two columns share a register (rank.h)
    // ... with HALF a block per compute wave where the grid of ten-half-block workgroups is co-resident too (704 lanes, one per CU)
    const long wg_half = (NB + RK_CWH / 2 - 1) / (RK_CWH / 2);
    h->rank_half = h->rank_reg && wg_half <= (long)prop0.multiProcessorCount && NB <= 1536;   // (1,536: one round of its decision wave's gather, rank.h)
    if (const char* e = getenv("BNMF_RANKHALF")) h->rank_half = h->rank_half && atoi(e) != 0;   // diagnostics / tests: 0 = whole blocks
    // (a wider grid with the blocks dealt wave-major over all CUs was measured: no gain, the sweep is bound by the
    // per-factor exchange, not by VALU contention)
    h->rank_grid = h->rank_half ? (int)wg_half : (int)std::min<long>(wg_needed, h->rank_reg ? cap_reg : cap_gen);
    if (const char* e = getenv("BNMF_RANKGRID")) { const long v = atol(e); if (!h->rank_half && v >= wg_needed && v <= (h->rank_reg ? cap_reg : cap_gen)) h->rank_grid = (int)v; }   // diagnostics only
    if (!h->rank_reg) HIPCHK(dmalloc(&h->dRankMhat, K * G * sizeof(double)));
    if (getenv("BNMF_RANKDBG")) { HIPCHK(dmalloc(&h->dRankDbg, (size_t)h->rank_grid * 16 * 8 * 8)); HIPCHK(hipMemset(h->dRankDbg, 0, (size_t)h->rank_grid * 16 * 8 * 8)); }   // diagnostics only
  }
  if (cfg->MH || cfg->likelihood == BNMF_NORMAL) {
    h->mh_S = (int)((G + MH_SEG - 1) / MH_SEG);
    HIPCHK(dmalloc(&h->dMhat, 3 * K * G * sizeof(double)));      // rows of Mhat maintained by the P sweep; log(Mhat) and its candidates (MH step)
    if (const char* e = getenv("BNMF_MHE_K128")) h->mhe_k128 = atoi(e) != 0;
    if (const char* e = getenv("BNMF_MHE_GW")) h->mhe_gw = atoi(e) == 32 ? 32 : atoi(e) == 16 ? 16 : 0;   // diagnostics / tests: lanes per column of k_mh_ecol16 (0 = by mode)
    h->mhe_lds = 4 * (2 * N + 3 * K) * sizeof(double);             // k_mh_ecol: per wave E column, A, Mhat column, log(Mhat) and candidates
    if (h->mhe_lds > 64 * 1024) {
      if (h->mhe_lds > 160 * 1024) return fail(BNMF_EINVAL, "bnmf_create: K = %zu too large for the column kernel of the MH / Normal models (LDS)", K);
      HIPCHK(hipFuncSetAttribute((const void*)k_mh_ecol<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)k_mh_ecol<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    // k_mh_ecol16 (K <= 128: several columns per wave): its LDS grows with N — above 64 KiB it needs the attribute, above the CU's 160 KiB
    // the sweep takes k_mh_ecol
    {
      const size_t lds16_max = (4 * (size_t)4 * N * (1 + PRE_W) + 2 * N) * sizeof(double);   // 16 lanes per column: 4 columns per wave
      h->mhe16 = K <= (size_t)MHE16_KMAX && lds16_max <= 160 * 1024;
      if (h->mhe16 && lds16_max > 64 * 1024) {
        const void* ks[] = {(const void*)k_mh_ecol16<false, false, 16>, (const void*)k_mh_ecol16<false, true, 16>, (const void*)k_mh_ecol16<false, false, 32>,
                            (const void*)k_mh_ecol16<false, true, 32>, (const void*)k_mh_ecol16<true, false, 16>, (const void*)k_mh_ecol16<true, false, 32>,
                            (const void*)k_mh_ecol16<false, false, 16, 96>, (const void*)k_mh_ecol16<false, true, 16, 96>, (const void*)k_mh_ecol16<false, false, 32, 96>,
                            (const void*)k_mh_ecol16<false, true, 32, 96>, (const void*)k_mh_ecol16<true, false, 16, 96>, (const void*)k_mh_ecol16<true, false, 32, 96>};
        for (const void* kf : ks) HIPCHK(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      }
    }
    // Round 5: k_mh_tail's work hosted by the two sweep kernels (mh.h) — the Poisson MH models at fixed rank where the column sweep is k_mh_ecol16
    h->mh_pipe = h->mhe16 && cfg->MH && cfg->likelihood == BNMF_POISSON && !cfg->learning_rank && h->mh_side_main && h->mh_side_tail && N <= RT;
    if (const char* e = getenv("BNMF_MHPIPE")) h->mh_pipe = h->mh_pipe && atoi(e) != 0;      // diagnostics / tests: 0 = k_mh_tail between the sweeps
    HIPCHK(dmalloc(&h->dAccPn, 3 * N * sizeof(double)));
    HIPCHK(dmalloc(&h->dAccEpart, 3 * (size_t)h->nblkE * sizeof(double)));
    HIPCHK(dmalloc(&h->dNzE, 6 * N * sizeof(int)));         // nzE[N], nzP[N] (k_mh_tail's), then the hosted form's nzE[2][N], nzP[2][N] by iteration parity
    HIPCHK(dmalloc(&h->dEt, N * G * sizeof(double)));
    HIPCHK(dmalloc(&h->dMt, K * G * sizeof(int32_t)));
    {
      std::vector<int32_t> mt(K * G);
      for (size_t g = 0; g < G; ++g) for (size_t k = 0; k < K; ++k) mt[g + G * k] = M[k + K * g];
      HIPCHK(hipMemcpy(h->dMt, mt.data(), K * G * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(dmalloc(&h->dLut, 2 * (size_t)(mx + 1) * sizeof(double)));
  if (cfg->n_temperature > 0 && cfg->temperature) {
    HIPCHK(dmalloc(&h->dTemp, cfg->n_temperature * sizeof(double)));
    HIPCHK(hipMemcpy(h->dTemp, cfg->temperature, cfg->n_temperature * sizeof(double), hipMemcpyHostToDevice));
    h->temp_host.assign(cfg->temperature, cfg->temperature + cfg->n_temperature);
  } else h->cfg.n_temperature = 0;
  h->cfg.temperature = nullptr;
  h->metrics_rows = 1024;
  HIPCHK(hipHostMalloc((void**)&h->hMetrics, h->metrics_rows * BNMF_NMETRIC * sizeof(double), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&h->dMetrics, h->hMetrics, 0));
  HIPCHK(dmalloc(&h->dRaw, h->metrics_rows * 8 * sizeof(double)));
  hipLaunchKernelGGL(k_luts, dim3((mx + 256) / 256), dim3(256), 0, h->stream, h->dLut, h->dLut + (mx + 1), mx);
  HIPCHK(hipGetLastError());
  clk.mark("small buffers, tables");
  // wait for the tables now: left to the end of this function, the wait cost 20-25 ms on every bnmf_create after the first of a
  // process (the launch sat unsubmitted behind the allocations and copies of the schedule)
  HIPCHK(hipStreamSynchronize(h->stream));
  clk.mark("tables kernel done");
  // Allocation-kernel geometry: independent waves, one LDS slab per wave, zacc (and P) shared per workgroup.
  // N <= 24 takes k_zalloc_reg (zalloc_reg.h), larger N the general LDS-search kernel k_zalloc (kernels.h).
  {
    ZGeom& zg = h->zg;
    zg.KP = (K % 32 == 0) ? (int)K + 1 : (int)(K | 1);
    zg.HW = (int)((N + 3) / 4);
    zg.TR = N <= 8 ? 8 : N <= 16 ? 16 : N <= 20 ? 20 : 24;   // threshold registers of the k_zalloc_reg instantiation
    h->z_reg = N <= (size_t)ZNMAX;
    if (const char* e = getenv("BNMF_ZREG")) h->z_reg = h->z_reg && atoi(e) != 0;   // diagnostics only
    long colmax = 0;
    for (size_t g = 0; g < G; ++g) { long cs = 0; for (size_t k = 0; k < K; ++k) cs += M[k + K * g]; if (cs > colmax) colmax = cs; }
    h->colmax = colmax;                                      // (checked behind build_zsort: the sorted schedule spreads large cells over the blocks and takes more)
    // LDS need of a geometry; the general kernel (k_zalloc) takes the whole column in one row chunk when that
    // leaves room for at least two waves per workgroup, else row chunks of 64 with ZsumG kept in global memory
    bool force_chunk = false;
    if (const char* e = getenv("BNMF_ZCHUNK")) force_chunk = atoi(e) != 0;   // diagnostics / tests
    size_t slab = 0, shared_words = 0;
    auto geometry = [&](bool chunked) {
      if (h->z_reg) {
        zg.KC = (int)K;
        slab = (size_t)zg.HW * ZH + (K + 1) * (size_t)zreg_row_words(zg.TR) + 2 * N + N + (cfg->save_Z ? N * (size_t)zg.KP : 0);
        zg.zacc_words = (int)((N * (size_t)zg.KP + 3) & ~(size_t)3);
        zg.p_words = (int)((2 * K * N + 3) & ~(size_t)3);                  // workgroup copy of P (fp64)
      } else {
        zg.KC = chunked ? 64 : (int)((K + 63) & ~(size_t)63);
        zg.KP = chunked ? 65 : ((K % 32 == 0) ? (int)K + 1 : (int)(K | 1));
        const bool loc = chunked || cfg->save_Z;
        slab = (size_t)zg.HW * ZH + 2 * N + (N - 1) * (size_t)zg.KP + (zg.KC + 1) + zg.KC + N + (loc ? N * (size_t)zg.KP : 0);
        zg.zacc_words = chunked ? 0 : (int)((N * (size_t)zg.KP + 3) & ~(size_t)3);
        zg.p_words = 0;
      }
      slab = (slab + 3) & ~(size_t)3;
      zg.slab_words = (int)slab;
      shared_words = (size_t)zg.zacc_words + zg.p_words;
    };
    // workgroup width.  k_zalloc holds 128 VGPRs per lane: at 16 waves/CU it owns the whole register file and
    // starves k_side (side stream) until its tail.  Measured end to end at the metric config (tools/e2e.py):
    // 16 waves/CU 178 us/iter, 12: 169, 10: 176, 8: 161, 2x4: 165, 6: 179.  So: at most 8 waves per CU, in one
    // workgroup when LDS allows.
    constexpr int Z_MAX_WAVES_PER_CU = 8;
    int best_w = 0, best_per_cu = 0, best_total = 0;
    auto pick = [&]() {
      best_w = best_per_cu = best_total = 0;
      for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int w : {16, 8, 6, 4, 2, 1}) {
          const size_t lds = (shared_words + (size_t)w * slab) * 4;
          if (lds * per_cu <= 160 * 1024 && w * per_cu <= Z_MAX_WAVES_PER_CU && w * per_cu > best_total) { best_total = w * per_cu; best_w = w; best_per_cu = per_cu; }
        }
    };
    geometry(force_chunk && !h->z_reg);
    pick();
    // the register path keeps (K+1) threshold rows per wave and a workgroup copy of P: for large K (e.g. K = 1,536
    // with N <= 24) that exceeds LDS, so fall back to the general kernel, which can walk the rows in chunks
    if (h->z_reg && best_total < 2) { h->z_reg = false; geometry(force_chunk); pick(); }
    if (!h->z_reg && best_total < 2 && !force_chunk) { geometry(true); pick(); }
    if (const char* e = getenv("BNMF_ZW")) { best_w = atoi(e); best_per_cu = (shared_words + (size_t)best_w * slab) * 4 * 2 <= 160 * 1024 ? 2 : 1; }
    if (best_w == 0) return fail(BNMF_EINVAL, "bnmf_create: K = %zu, N = %zu needs %zu B of LDS per wavefront for the allocation kernel (limit 160 KiB): unsupported", K, N, slab * 4);
    h->z_zw = best_w;
    h->z_lds = ((shared_words + (size_t)best_w * slab) * 4 + 15) & ~(size_t)15;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    const long resident = (long)prop.multiProcessorCount * best_per_cu;
    const long want = ((long)G + best_w - 1) / best_w;
    h->z_grid = (int)(want < resident ? want : resident);
    if (const char* e = getenv("BNMF_ZGRID")) h->z_grid = atoi(e);          // diagnostics only
    // N > 24, stats mode: the statically scheduled lane-per-item kernel (zalloc_step.h), where its LDS layout fits
    if (!h->z_reg) if (int rc = build_zstep(h, M, prop.multiProcessorCount)) return rc;
    // N > 24 (or K too large for the register kernel): the tile kernel, when at least two waves per CU fit
    if (!h->z_reg && !h->z_step) {
      bool want_tile = true;
      if (const char* e = getenv("BNMF_ZTILE")) want_tile = atoi(e) != 0;   // diagnostics / tests: 0 = k_zalloc
      ZTGeom& tg = h->ztg;
      tg.HW = zg.HW;
      tg.nch = (int)((K + ZTR - 1) / ZTR);
      tg.p_words = 2 * ztile_np8((int)N) * ZTR;
      tg.zacc_words = (int)((N * (size_t)ZTR + 3) & ~(size_t)3);
      tg.slab_words = (int)ztile_slab_words((int)N, tg.HW, cfg->save_Z != 0);
      const size_t shw = (size_t)tg.p_words + tg.zacc_words;
      int tw = 0, tper = 0, ttot = 0;
      for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int w : {8, 6, 4, 2}) {
          const size_t lds = (shw + (size_t)w * tg.slab_words) * 4;
          // 8 waves per CU: the kernel holds ~190 VGPRs (double-buffered LDS reads), i.e. two waves per SIMD
          if (lds * per_cu <= 160 * 1024 && w * per_cu <= 8 && w * per_cu > ttot) { ttot = w * per_cu; tw = w; tper = per_cu; }
        }
      // ... or 16 (one workgroup of 1024 lanes) with the register-lean variant, where LDS allows it
      h->z_lean = false;
      if ((shw + 16 * (size_t)tg.slab_words) * 4 <= 160 * 1024 && !getenv("BNMF_ZNOLEAN")) { ttot = 16; tw = 16; tper = 1; h->z_lean = true; }
      if (want_tile && ttot >= 2) {
        h->z_tile = true;
        h->z_zw = tw;
        h->z_lds = ((shw + (size_t)tw * tg.slab_words) * 4 + 15) & ~(size_t)15;
        // column slices: the fewest rounds (1..4) of resident workgroups that fill >= 97 % of the CUs (the chunk's P rows
        // are staged and its ZsumG counts flushed once per workgroup); every wave with at least two columns
        const long res = (long)prop.multiProcessorCount * tper;
        long ns = 1; double best_util = 0.0;
        for (long r = 1; r <= 4; ++r) {
          const long c = (r * res) / tg.nch;
          if (c < 1) continue;
          const double util = (double)(c * tg.nch) / (double)(r * res);
          if (util > best_util + 1e-9) { best_util = util; ns = c; }
          if (util >= 0.97) break;
        }
        const long nsmax = ((long)G + 2 * tw - 1) / (2 * tw);
        if (ns > nsmax) ns = nsmax;
        if (ns < 1) ns = 1;
        tg.nslice = (int)ns;
        h->z_grid = tg.nch * tg.nslice;
        HIPCHK(dmalloc(&h->dMhatZ, K * G * sizeof(double)));
        tg.dbg = nullptr;
        if (getenv("BNMF_ZTDBG")) { HIPCHK(dmalloc(&tg.dbg, 8 * sizeof(unsigned long long))); HIPCHK(hipMemset(tg.dbg, 0, 8 * sizeof(unsigned long long))); }   // diagnostics only
      }
    }
#ifdef BNMF_DIAG   /* the builder's diagnostic builds only (tools/bin/, never libbnmf.so): phases of the allocation kernels switched off */
    if (const char* e = getenv("BNMF_ABLATE")) h->z_ablate = atoi(e);
#endif
    h->n_cu = prop.multiProcessorCount;
    if (!h->z_tile && !h->z_step && !h->z_ablate) if (int rc = build_zsort(h, M, prop.multiProcessorCount)) return rc;
    // the other allocation kernels take a column whole (k_zalloc_reg: one wave per column; tile / step: a workgroup's batch): 4,000,000 counts
    // per column is where that stops being a sensible launch.  The sorted schedule (N <= 24, K <= 1,024, no save_Z) spreads the counts of a
    // large cell over all blocks and is bounded by its item format only (a cell <= 65,534 fragments of 256 counts = 16.7 M counts).
    if (h->colmax > 4000000 && !(h->z_sort && h->zs_shared))
      return fail(BNMF_EINVAL, "bnmf_create: a column of M sums to %ld (> 4,000,000 counts): unsupported for this model / shape (N <= 24 and K <= 1,024 without save_Z take up to 16.7 M counts per cell)", h->colmax);
  }
  clk.mark("allocation-kernel schedule");
  HIPCHK(hipStreamSynchronize(h->stream));
  refresh_dev(h);
  clk.mark("drain");
  return 0;
}

int bnmf_destroy(bnmf_handle* h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->side) hipStreamSynchronize(h->side);
  if (h->side2) hipStreamSynchronize(h->side2);
  for (int id = 0; id < BNMF_ID_MAX; ++id) {
    Arr& a = h->arr[id];
    if (a.d && !a.slab) dfree(a.d);
    if (a.ring) ring_release(h->device, a.ring, (size_t)h->wcap * id_len(h, id) * sizeof(double));
  }
  dfree(h->dM); dfree(h->dZsumK); dfree(h->dZsumG); if (h->dZ) dfree(h->dZ);
  dfree(h->dR); dfree(h->dRedraw); dfree(h->dEsum); dfree(h->dPsum); dfree(h->dlpPn);
  dfree(h->dlpE); dfree(h->dcol); dfree(h->dLut); if (h->dTemp) dfree(h->dTemp); if (h->hMetrics) hipHostFree(h->hMetrics); dfree(h->dRaw); if (h->dRankCol) dfree(h->dRankCol); if (h->dRankMhat) dfree(h->dRankMhat); if (h->dRankSync) dfree(h->dRankSync);
  if (h->E_alt) dfree(h->E_alt);
  if (h->dMhatZ) dfree(h->dMhatZ);
  if (h->zpg.prof) dfree(h->zpg.prof);
  if (h->dZpItems) dfree(h->dZpItems); if (h->dZpWgs) dfree(h->dZpWgs); if (h->dZpBatches) dfree(h->dZpBatches); if (h->dZpSteps) dfree(h->dZpSteps); if (h->dZpCols) dfree(h->dZpCols);
  if (h->dZsItems) dfree(h->dZsItems); if (h->dZsBlocks) dfree(h->dZsBlocks); if (h->dZsCols) dfree(h->dZsCols); if (h->dZsProf) dfree(h->dZsProf); if (h->dZsM) dfree(h->dZsM); if (h->dZsRec) dfree(h->dZsRec); if (h->dZsRecRing) dfree(h->dZsRecRing); if (h->dZsMh) dfree(h->dZsMh);
  if (h->dMhat) dfree(h->dMhat); if (h->dAccPn) dfree(h->dAccPn); if (h->dAccEpart) dfree(h->dAccEpart); if (h->dNzE) dfree(h->dNzE);
  if (h->dEt) dfree(h->dEt); if (h->dMt) dfree(h->dMt); if (h->zring) dfree(h->zring);
  if (h->ev_draw) hipEventDestroy(h->ev_draw); if (h->ev_side) hipEventDestroy(h->ev_side); if (h->ev_sideP) hipEventDestroy(h->ev_sideP); if (h->ev_p) hipEventDestroy(h->ev_p); if (h->ev_rank) hipEventDestroy(h->ev_rank); if (h->ev_z) hipEventDestroy(h->ev_z); if (h->ev_red) hipEventDestroy(h->ev_red); give_stream(h->device, h->side, 1); give_stream(h->device, h->side2, 1);
  if (h->have_ev) for (auto& e : h->ev) hipEventDestroy(e);
  if (h->dMap) dfree(h->dMap);
  if (h->dAsg) dfree(h->dAsg);
  if (h->devlock_fd >= 0) close(h->devlock_fd);
  if (h->devgate_fd >= 0) close(h->devgate_fd);
  if (h->dFlags) dfree(h->dFlags); if (h->dDrawOwn) dfree(h->dDrawOwn); if (h->dScal) dfree(h->dScal); if (h->hErr) hipHostFree(h->hErr);
  give_stream(h->device, h->stream);                    // synchronised at the top of this function
  delete h;
  return 0;
}

int bnmf_set_array(bnmf_handle* h, int id, const double* x, size_t n) {
  if (!h || !x) return fail(BNMF_EINVAL, "bnmf_set_array: null argument");
  const size_t len = id_len(h, id);
  if (len == 0) return fail(BNMF_EINVAL, "bnmf_set_array: unknown id %d", id);
  HIPCHK(hipSetDevice(h->device));
  CreateClock clk;
  HIPCHK(hipStreamSynchronize(h->stream));
  clk.mark("set_array: sync main");
  HIPCHK(hipStreamSynchronize(h->side));
  clk.mark("set_array: sync side");
  HIPCHK(hipStreamSynchronize(h->side2));
  clk.mark("set_array: sync side2");
  h->side_valid = false; h->side_main = false;   // state changed: the pre-issued k_side must be redone
  h->mh_prep_valid = false; h->mh_pipe_valid = false;
  if (id == BNMF_R) { int r = (int)x[0]; HIPCHK(hipMemcpy(h->dR, &r, sizeof(int), hipMemcpyHostToDevice)); h->arr[BNMF_R].set = true; return 0; }
  if (id == BNMF_ZSUMK || id == BNMF_ZSUMG || id == BNMF_Z) {
    int32_t* dst = id == BNMF_ZSUMK ? h->dZsumK : id == BNMF_ZSUMG ? h->dZsumG : h->dZ;
    if (!dst) return fail(BNMF_EUNSET, "bnmf_set_array: Z is not materialised (save_Z = 0)");
    if (n != len) return fail(BNMF_ESIZE, "bnmf_set_array: id %d expects %zu values, got %zu", id, len, n);
    std::vector<int32_t> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = (int32_t)x[i];
    HIPCHK(hipMemcpy(dst, tmp.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    if (id == BNMF_Z) h->z_expanded_iter = h->iter;        // (what the caller stored is what a read returns until the next sweep)
    return 0;
  }
  Arr& a = h->arr[id];
  if (is_hyper(id) && n == 1) {
    // a broadcast scalar lives in the handle's block of scalars: the eight device allocations of the default hyper-prior values
    // were 20 ms of a bayesNMF() call
    if (a.d && !a.slab) HIPCHK(dfree(a.d));
    a.d = h->dScal + id; a.slab = true;
    // the value travels as a kernel argument: the first small host-to-device copy of a handle created after another one had been
    // destroyed took 13-24 ms (BNMF_TIMING marks), every time
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, h->stream, a.d, x[0]);
    HIPCHK(hipGetLastError());
    clk.mark("set_array: a scalar");
    a.n = 1; a.stride = 0; a.set = true;
    refresh_dev(h);
    return 0;
  }
  if (n != len) return fail(BNMF_ESIZE, "bnmf_set_array: id %d expects %zu values, got %zu", id, len, n);
  const size_t nslot = is_prior_param(id) ? 2 : 1;
  if (a.d && (a.n != len || a.slab)) { if (!a.slab) HIPCHK(dfree(a.d)); a.d = nullptr; a.slab = false; }
  if (!a.d) HIPCHK(dmalloc(&a.d, nslot * len * sizeof(double)));
  HIPCHK(hipMemcpy(a.d + (nslot == 2 ? (size_t)cur_slot(h) * len : 0), x, len * sizeof(double), hipMemcpyHostToDevice));
  a.n = len; a.stride = 1; a.set = true;
  // which columns n (P side) / rows n (E side) carry a missing (NaN) entry
  if (!is_hyper(id) && len >= (size_t)h->cfg.N && id != BNMF_A && id < 30) {
    const size_t K = h->cfg.K, N = h->cfg.N;
    a.redraw.assign(N, 0);
    const bool ps = is_pside(id);
    for (size_t i = 0; i < len; ++i) if (x[i] != x[i]) a.redraw[ps ? i / K : i % N] = 1;
  }
  refresh_dev(h);
  return 0;
}

int bnmf_get_array(bnmf_handle* h, int id, double* out, size_t n) {
  if (!h || !out) return fail(BNMF_EINVAL, "bnmf_get_array: null argument");
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_get_array: the handle timed out inside a kernel; its state is invalid");
  const size_t len = id_len(h, id);
  if (len == 0 || n != len) return fail(BNMF_ESIZE, "bnmf_get_array: id %d expects %zu values, got %zu", id, len, n);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->side));
  HIPCHK(hipStreamSynchronize(h->side2));
  if (id == BNMF_R) { int r; HIPCHK(hipMemcpy(&r, h->dR, sizeof(int), hipMemcpyDeviceToHost)); out[0] = r; return 0; }
  if (id == BNMF_ZSUMK || id == BNMF_ZSUMG || id == BNMF_Z) {
    const int32_t* src = id == BNMF_ZSUMK ? h->dZsumK : id == BNMF_ZSUMG ? h->dZsumG : h->dZ;
    if (!src) return fail(BNMF_EUNSET, "bnmf_get_array: Z is not materialised (save_Z = 0)");
    if (id == BNMF_Z) if (int rc = ensure_Z(h)) return rc;
    std::vector<int32_t> tmp(n);
    HIPCHK(hipMemcpy(tmp.data(), src, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) out[i] = tmp[i];
    return 0;
  }
  const Arr& a = h->arr[id];
  if (!a.d) return fail(BNMF_EUNSET, "bnmf_get_array: id %d has no value", id);
  if (a.stride == 0) { double v; HIPCHK(hipMemcpy(&v, a.d, sizeof(double), hipMemcpyDeviceToHost)); for (size_t i = 0; i < n; ++i) out[i] = v; return 0; }
  HIPCHK(hipMemcpy(out, a.d + (is_prior_param(id) ? (size_t)cur_slot(h) * len : 0), n * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int bnmf_get_array_i32(bnmf_handle* h, int id, int32_t* out, size_t n) {
  if (!h || !out) return fail(BNMF_EINVAL, "bnmf_get_array_i32: null argument");
  const size_t len = id_len(h, id);
  const int32_t* src = id == BNMF_ZSUMK ? h->dZsumK : id == BNMF_ZSUMG ? h->dZsumG : id == BNMF_Z ? h->dZ : nullptr;
  if (!(id == BNMF_ZSUMK || id == BNMF_ZSUMG || id == BNMF_Z)) return fail(BNMF_EINVAL, "bnmf_get_array_i32: id %d is not an integer array", id);
  if (!src) return fail(BNMF_EUNSET, "bnmf_get_array_i32: Z is not materialised (save_Z = 0)");
  if (n != len) return fail(BNMF_ESIZE, "bnmf_get_array_i32: id %d expects %zu values, got %zu", id, len, n);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (id == BNMF_Z) if (int rc = ensure_Z(h)) return rc;
  HIPCHK(hipMemcpy(out, src, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}

int bnmf_debug_rank(bnmf_handle* h, unsigned long long* out, size_t n) {   // diagnostics: phase time stamps of k_rank_sweep
  if (!h || !h->dRankDbg) return fail(BNMF_ESTATE, "BNMF_RANKDBG not set");
  HIPCHK(hipMemcpy(out, h->dRankDbg, n * 8, hipMemcpyDeviceToHost));
  return h->rank_grid;
}
#ifdef ZSPROF
int bnmf_debug_zstamps(bnmf_handle* h, unsigned long long* out) {   // diagnostics (-DZSPROF): [3 blocks][16 waves][8] s_memrealtime stamps of the last k_zalloc_sort
  if (!h || !h->dZsProf) return fail(BNMF_ESTATE, "not a -DZSPROF build, or the kernel is not in use");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(out, h->dZsProf + 8, 3 * 16 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
#endif
int bnmf_debug_zsort(bnmf_handle* h, unsigned long long* out) {   // diagnostics (-DZSPROF / -DZPPROF builds): section ticks of k_zalloc_sort / k_zalloc_step, then reset
  unsigned long long* src = h ? (h->dZsProf ? h->dZsProf : h->zpg.prof) : nullptr;
  if (!src) return fail(BNMF_ESTATE, "not a -DZSPROF / -DZPPROF build, or the kernel is not in use");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(out, src, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(src, 0, 8 * sizeof(unsigned long long)));
  return 0;
}
#ifdef ZSPROF
int bnmf_debug_draw(bnmf_handle* h, unsigned long long* out, int reset) {   // diagnostics (-DZSPROF builds): [wave][8] section ticks of k_draw's E waves
  if (!h) return fail(BNMF_EINVAL, "null");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_drprof), 8 * DRPROF_W * sizeof(unsigned long long)));
  if (reset) {
    std::vector<unsigned long long> z(8 * DRPROF_W, 0ull);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_drprof), z.data(), z.size() * sizeof(unsigned long long)));
  }
  return DRPROF_W;
}
#endif
int bnmf_debug_set_timeout(bnmf_handle* h, int word) {   // tests: what a bounded in-kernel wait does when it gives up
  if (!h || word < 0 || word > 1) return fail(BNMF_EINVAL, "bnmf_debug_set_timeout: bad argument");
  ((volatile int*)h->hErr)[word] = 2;
  return 0;
}
int bnmf_get_stat(bnmf_handle* h, int what, double* out) {   // sizes of the schedule's buffers, for bench.py's byte counts
  if (!h || !out) return fail(BNMF_EINVAL, "bnmf_get_stat: null argument");
  switch (what) {
    case 0: *out = (double)h->zs_recwords * 4.0; return 0;                                  // bytes of item records per iteration (save_Z, sorted schedule; else 0)
    case 1: *out = h->dZsMh ? (double)h->cfg.K * h->cfg.G * 8.0 : 0.0; return 0;            // bytes of Mhat left per iteration for the column terms
    case 2: *out = h->dZsRecRing ? 1.0 : 0.0; return 0;                                     // samples$Z kept as a ring of records
    case 3: *out = h->zs_eager ? 1.0 : 0.0; return 0;
    case 5: *out = h->z_sort ? (double)h->zs_qmax : 0.0; return 0;                          // quads per item of the sorted schedule (chosen per data set)
    case 4: *out = h->mh_pipe ? 1.0 : 0.0; return 0;                                        // MH sweep: k_mh_tail's work hosted by the two sweep kernels
    default: return fail(BNMF_EINVAL, "bnmf_get_stat: unknown statistic %d", what);
  }
}
int bnmf_get_iter(bnmf_handle* h, int* iter) { if (!h || !iter) return fail(BNMF_EINVAL, "null"); *iter = h->iter; return 0; }

}  // extern "C"

// ------------------------------------------------------------------ launch helpers
static int need_hyper(bnmf_handle* h, std::initializer_list<int> ids) {
  for (int id : ids) if (!h->arr[id].d) return fail(BNMF_EUNSET, "hyper-prior array id %d was not set (fill_hyperprior_params, R/setup.R:15-88)", id);
  return 0;
}
static int ensure_metrics(bnmf_handle* h, size_t rows) {
  if (rows <= h->metrics_rows) return 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipHostFree(h->hMetrics));
  h->hMetrics = nullptr; h->dMetrics = nullptr;
  h->metrics_rows = rows;
  HIPCHK(hipHostMalloc((void**)&h->hMetrics, rows * BNMF_NMETRIC * sizeof(double), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&h->dMetrics, h->hMetrics, 0));
  HIPCHK(dfree(h->dRaw));
  HIPCHK(dmalloc(&h->dRaw, rows * 8 * sizeof(double)));
  refresh_dev(h);
  return 0;
}
// Per-iteration partial sums (per-column metric terms, log-prior partials, MH acceptance partials) live in
// three slots (t % 3): k_reduce of iteration t is issued during iteration t+1 (see launch_side / launch_side_E), and the
// next writers of its slot are the kernels of iteration t+3.  Fixed-rank sweep: k_reduce(t) sits on side2 in front of
// Esum(t+2), whose flag releases k_pdraw(t+2) and with it everything of iteration t+2 and later on the main stream; the
// log-prior workgroups of side2 follow it in stream order.  Other sweeps: through ev_side (main stream) and ev_red (side2).
static void set_slot(const bnmf_handle* h, Dev& d, uint32_t t) {
  const size_t sl = t % 3u, G = h->cfg.G, N = h->cfg.N;
  d.colsse = h->dcol + sl * 3 * G; d.colll = d.colsse + G; d.colkl = d.colsse + 2 * G;
  d.lpE_part = h->dlpE + sl * (size_t)h->nblkE;
  d.lpPn = h->dlpPn + sl * N;
}
static void use_slot(bnmf_handle* h, uint32_t t) { set_slot(h, h->dev, t); }
static double* accPn_slot(const bnmf_handle* h, uint32_t t) { return h->dAccPn ? h->dAccPn + (size_t)(t % 3u) * h->cfg.N : nullptr; }
static double* accEp_slot(const bnmf_handle* h, uint32_t t) { return h->dAccEpart ? h->dAccEpart + (size_t)(t % 3u) * h->nblkE : nullptr; }
struct Timer {   // optional per-kernel HIP-event bracketing (serialises the two streams: profile mode only)
  bnmf_handle* h; bool on; double acc[BNMF_NKERNEL]{}; int cnt[BNMF_NKERNEL]{};
  void begin(int k, hipStream_t st) { if (on) { hipStreamSynchronize(h->stream); hipStreamSynchronize(h->side); hipStreamSynchronize(h->side2); hipEventRecord(h->ev[2 * k], st); } }
  void end(int k, hipStream_t st) { if (on) { hipEventRecord(h->ev[2 * k + 1], st); hipEventSynchronize(h->ev[2 * k + 1]); float ms = 0; hipEventElapsedTime(&ms, h->ev[2 * k], h->ev[2 * k + 1]); acc[k] += ms; cnt[k]++; } }
};
static RecDst rec_at(const bnmf_handle* h, uint32_t t, bool on);
static bool fused_rec(const bnmf_handle* h);
static double* ring_at(const bnmf_handle* h, int id, uint32_t t);
static RecDst rec_pdraw(const bnmf_handle* h, uint32_t t, bool on) {   // what k_pdraw records
  RecDst r = rec_at(h, t, on);
  if (h->cfg.learning_rank) { r.A = nullptr; r.R = nullptr; }          // then k_sumA records them, after the rank update
  return r;
}
static void launch_pdraw(bnmf_handle* h, uint32_t t, int from_prior, bool rec) {
  const size_t lds = 2 * (size_t)h->cfg.K * sizeof(double);
  hipLaunchKernelGGL(k_pdraw, dim3(h->cfg.N), dim3(PD_T), lds, h->stream, h->dev, t, from_prior, 1, rec_pdraw(h, t, rec), SideWait{});
}
static void launch_edraw(bnmf_handle* h, uint32_t t, int from_prior, bool rec) {
  hipLaunchKernelGGL(k_edraw, dim3(h->nblkE), dim3(ES_T), 0, h->stream, h->dev, t, from_prior, 1, rec_at(h, t, rec).E);
}
// k_side for iteration t (reads P_{t-1}, E_{t-1}): issued on the side stream right after the draws
// of iteration t-1, so that it overlaps k_zalloc of iteration t-1
// tests (BNMF_DEBUG_ALLSIDE_DELAY_US): hold a side stream back in front of its next kernel — whatever then reads too early or writes too early shows
// as a bit that differs from the oracle's
static void dbg_delay(bnmf_handle* h, hipStream_t st) {
  if (h->dbg_allside_delay_us && st != h->stream) hipLaunchKernelGGL(k_debug_delay, dim3(1), dim3(64), 0, st, h->dbg_allside_delay_us);
}
// ... and the other way round (BNMF_DEBUG_MAIN_DELAY_US): the main stream held back, so that a side-stream kernel that runs on the main stream's
// results without waiting for them reads the values of the iteration before
static void dbg_delay_main(bnmf_handle* h) {
  if (h->dbg_main_delay_us) hipLaunchKernelGGL(k_debug_delay, dim3(1), dim3(64), 0, h->stream, h->dbg_main_delay_us);
}
static void issue_reduce(bnmf_handle* h, uint32_t t, int row, Timer& tm, hipStream_t st = nullptr) {
  if (!st) st = h->side;
  dbg_delay(h, st);
  Dev dr = h->dev;
  set_slot(h, dr, t);
  tm.begin(KN_REDUCE, st);
  hipLaunchKernelGGL(k_reduce, dim3(h->cfg.MH ? 5 : 4), dim3(RT), 0, st, dr, row, h->nblkE, (const double*)accPn_slot(h, t), (const double*)accEp_slot(h, t));
  tm.end(KN_REDUCE, st);
  // k_lpp's workgroups (side2) rewrite an lpPn slot this kernel read three iterations earlier: side2 waits for ev_red, unless the
  // reduce was issued on side2 itself (the fixed-rank sweep), where stream order does it without two runtime calls
  h->red_on_side2 = st == h->side2;
  if (!h->red_on_side2) { hipEventRecord(h->ev_red, st); h->red_issued = true; }
}
// The per-column metric terms of iteration t from the Mhat k_zalloc_sort left (colterms.h), into the iteration's slot of the partial sums
static CtArgs ct_args(const bnmf_handle* h, uint32_t t) {
  const size_t G = h->cfg.G;
  double* sse = h->dcol + (size_t)(t % 3u) * 3 * G;
  return CtArgs{h->dZsMh + (size_t)(t % 3u) * h->cfg.K * G, h->dev.M, h->dev.lgfact, h->dev.logm, sse, sse + G, sse + 2 * G, h->cfg.K, h->cfg.G, h->dev.maxM};
}
// ... as a launch of its own on the main stream (behind the allocation kernel in stream order): whenever the next kernel on that stream
// is not a merged draw kernel that could take the work along (first sweeps of a chain, two-kernel sweep, profile mode, end of a call)
static void flush_colterms(bnmf_handle* h) {
  if (!h->ct_pending) return;
  const CtArgs a = ct_args(h, h->ct_pending);
  hipLaunchKernelGGL(k_colterms, dim3((unsigned)((h->cfg.G + 2 * (CT_T / 64) - 1) / (2 * (CT_T / 64)))), dim3(CT_T), 0, h->stream, a);
  h->ct_pending = 0;
}
// ev_sideP (side2 done) and ev_side (side done, behind ev_sideP) are what a main-stream wait or flush_reduce needs; in the
// steady state of the fixed-rank sweep nobody waits for them (k_pdraw polls flags), so they are recorded on demand: a later
// record covers everything enqueued before it
static void refresh_side_events(bnmf_handle* h) {
  if (!h->side_ev_stale) return;
  hipEventRecord(h->ev_sideP, h->side2);
  hipStreamWaitEvent(h->side, h->ev_sideP, 0);
  hipEventRecord(h->ev_side, h->side);
  h->side_ev_stale = false;
}
static void launch_side(bnmf_handle* h, uint32_t t, Timer& tm, bool publish = false) {
  const int nbP = (int)(((size_t)h->cfg.K * h->cfg.N + RT - 1) / RT);
  const int nbE = (int)(((size_t)h->cfg.N * h->cfg.G + RT - 1) / RT);
  hipEventRecord(h->ev_draw, h->stream);
  hipStreamWaitEvent(h->side, h->ev_draw, 0);
  tm.begin(KN_SIDE, h->side);
  // publish (MH / Normal sweeps): the last workgroup raises flag [1] = t, which the next P-row kernel polls (no barrier packet)
  dbg_delay(h, h->side);
  hipLaunchKernelGGL(k_side, dim3(h->cfg.N + nbP + nbE), dim3(RT), 0, h->side, h->dev, t, nbP, 0, rec_at(h, t, fused_rec(h)),
                     publish ? SideDone{h->dFlags, h->dFlags + 1, (unsigned)(h->cfg.N + nbP + nbE), t} : SideDone{});
  h->flags_valid = publish;
  tm.end(KN_SIDE, h->side);
  hipEventRecord(h->ev_side, h->side);
  hipEventRecord(h->ev_sideP, h->side);
  h->side_ev_stale = false;
  h->side_valid = true;
  h->side_main = false;
  // k_reduce of the PREVIOUS iteration: its inputs are complete once the draws of this iteration have run
  // (main-stream order), which ev_draw above implies, so the main stream needs no marker after k_zalloc
  if (h->red_pending) { issue_reduce(h, h->red_t, h->red_row, tm); h->red_pending = false; }
}
// MH / Normal sweeps, steady state (round 4): the hyper sweep of iteration t on the MAIN stream, between the column kernel of t-1 and
// its tail kernel.  On the side stream it was released by the column kernel through an event (12 us late), ran 20 us beside a 12 us tail
// kernel, and the next P-row kernel waited 13.6 us of its 101 for the flag (profiles/r04_cfg3_kernel_stats.csv, r04_mh_prow_stamps.txt);
// alone on the device it is shorter than that wait, and the row kernel behind it needs neither flag nor event.
static void launch_side_main(bnmf_handle* h, uint32_t t, Timer& tm) {
  const int nbP = (int)(((size_t)h->cfg.K * h->cfg.N + RT - 1) / RT);
  const int nbE = (int)(((size_t)h->cfg.N * h->cfg.G + RT - 1) / RT);
  tm.begin(KN_SIDE, h->stream);
  hipLaunchKernelGGL(k_side, dim3(h->cfg.N + nbP + nbE), dim3(RT), 0, h->stream, h->dev, t, nbP, 0, rec_at(h, t, fused_rec(h)), SideDone{});
  tm.end(KN_SIDE, h->stream);
  h->flags_valid = false;
  h->side_valid = true;
  h->side_main = true;
  // (k_reduce of the PREVIOUS iteration: inside this iteration's k_mh_tail, see launch_mh_metrics)
}
// The same work in three launches, for the Gibbs sweep.  The P-side hyper sweep depends on P_{t-1} only and has
// the longest per-lane latency (rejection sampling of Alpha): it starts right behind k_pdraw on its own stream.
// Esum follows it once k_edraw is done; both are over long before k_zalloc, so that the event the next k_pdraw
// waits for is already satisfied when the main stream reaches it (a late cross-stream event costs ~12 us).
// The E-side sweep (needed only by the next k_edraw) shares the CUs with k_zalloc and ends with it.
// what k_lpe reads as E_t: the ring slot of iteration t when the sweep records (safe for a whole window), else the live E
// (then sweep() double-buffers E)
static bool lpe_from_ring(const bnmf_handle* h) { return fused_rec(h) && h->arr[BNMF_E].ring != nullptr; }
static const double* lpe_src(const bnmf_handle* h, uint32_t t) { return lpe_from_ring(h) ? ring_at(h, BNMF_E, t) : h->dev.E; }
static void launch_side_P(bnmf_handle* h, uint32_t t, hipEvent_t after = nullptr) {   // ev_p = completion of k_pdraw(t-1)
  const int nbP = (int)(((size_t)h->cfg.K * h->cfg.N + RT - 1) / RT);
  hipStreamWaitEvent(h->side2, after ? after : h->ev_p, 0);
  // k_lpp below rewrites lpPn slot (t-1) % 3, last read by k_reduce of iteration t-4 (side stream): order behind it
  if (h->red_issued && !h->red_on_side2) hipStreamWaitEvent(h->side2, h->ev_red, 0);
  // ... and the log-prior of the P just drawn (k_lpp's work, iteration t-1) in the same launch
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_side_lp, dim3(nbP + h->cfg.N), dim3(RT), 0, h->side2, h->dev, t, nbP, h->cfg.N, rec_at(h, t, fused_rec(h)), SideDone{},
                     SideExtra{nbP, h->cfg.N, 0, t - 1, nullptr, 0}, CtArgs{});
}
static void launch_side_E(bnmf_handle* h, uint32_t t, Timer& tm, bool e_done = false) {   // ev_draw = completion of k_edraw(t-1); e_done: k_draw ran the E-side sweep
  const int nbP = (int)(((size_t)h->cfg.K * h->cfg.N + RT - 1) / RT);
  const int nbE = (int)(((size_t)h->cfg.N * h->cfg.G + RT - 1) / RT);
  hipStreamWaitEvent(h->side2, h->ev_draw, 0);
  // Esum closes the side2 work the next k_pdraw needs (the P part ran before it on the same stream): it publishes flag [3]
  // ... and, in the same launch, the log-prior of the E just drawn (k_lpe's work; iteration t-1, whose slot pointers h->dev
  // still holds): off the critical path
  dbg_delay(h, h->side2);
  // (the per-column metric terms of the iteration before ride along as in launch_side_merged: workgroups behind the log-prior ones)
  CtArgs ct{};
  int n_ct = 0;
  if (h->ct_pending) { ct = ct_args(h, h->ct_pending); n_ct = (h->cfg.G + 2 * (RT / 64) - 1) / (2 * (RT / 64)); h->ct_pending = 0; }
  hipLaunchKernelGGL(k_side_lp, dim3(h->cfg.N + h->nblkE + n_ct), dim3(RT), 0, h->side2, h->dev, t, nbP, 0, RecDst{}, SideDone{h->dFlags + 2, h->dFlags + 3, (unsigned)(h->cfg.N + h->nblkE), t},
                     SideExtra{h->cfg.N, 0, h->nblkE, t - 1, lpe_src(h, t - 1), 1}, ct);
  // k_reduce of the PREVIOUS iteration here, behind the kernels that produce its inputs on this stream (k_lpp, k_lpe) and
  // behind ev_draw (k_zalloc of that iteration): on the E part's stream it sat in front of the next E-side sweep, and the
  // P part waited for its event
  if (h->red_pending) { issue_reduce(h, h->red_t, h->red_row, tm, h->side2); h->red_pending = false; }
  if (!e_done) {
    hipStreamWaitEvent(h->side, h->ev_draw, 0);
    dbg_delay(h, h->side);
    hipLaunchKernelGGL(k_side, dim3(nbE), dim3(RT), 0, h->side, h->dev, t, nbP, h->cfg.N + nbP, rec_at(h, t, fused_rec(h)), SideDone{h->dFlags, h->dFlags + 1, (unsigned)nbE, t});
  }
  h->flags_valid = true;
  // ev_side (the E part AND the P part / Esum / log-priors done) for a main-stream wait: on demand, see refresh_side_events
  h->side_ev_stale = true;
  h->side_valid = true;
}
// Behind the merged draw kernel (which runs the E-side sweep itself): the P-side hyper sweep of iteration t on `side`, with a
// flag of its own ([9]); Esum(t) and the log-priors of iteration t-1 in ONE launch on side2 (flag [3] counts all its
// workgroups), k_reduce behind it.  The two no longer share a stream: the P-side sweep is a few long per-lane chains and
// held Esum's flag back.
static void launch_side_merged(bnmf_handle* h, uint32_t t, Timer& tm) {
  const int N = h->cfg.N;
  const int nbP = (int)(((size_t)h->cfg.K * N + RT - 1) / RT);
  hipStreamWaitEvent(h->side, h->ev_draw, 0);
  if (h->dbg_side_delay_us) hipLaunchKernelGGL(k_debug_delay, dim3(1), dim3(64), 0, h->side, h->dbg_side_delay_us);   // tests: a late P-side sweep
  dbg_delay(h, h->side);
  hipLaunchKernelGGL(k_side, dim3(nbP), dim3(RT), 0, h->side, h->dev, t, nbP, N, rec_at(h, t, fused_rec(h)), SideDone{h->dFlags + 8, h->dFlags + 9, (unsigned)nbP, t});
  // (Round 5, measured and NOT adopted: releasing the side streams by the draw kernel's flag — one polling wavefront at the head of each side
  // stream, P / E stored write-through, no stop event on the draw kernel.  The stop event costs ~6 us between the draw kernel's end and the
  // allocation kernel's start in the traces and its signal reaches the side queues 12-20 us later, differently from process to process
  // (tools/bimodal.sh: steady iteration 80.6 us in most processes, 83.3 us in about a quarter) — but with the flag the side kernels start
  // WITH the allocation kernel and take its issue slots from its first task on: 80.3 -> 96.4 us per iteration, bit-exact.)
  hipStreamWaitEvent(h->side2, h->ev_draw, 0);
  if (h->red_issued && !h->red_on_side2) hipStreamWaitEvent(h->side2, h->ev_red, 0);   // lpPn slot reuse, see launch_side_P
  dbg_delay(h, h->side2);
  // the per-column metric terms of iteration t - 2 ride along (its allocation kernel ran before the draw kernel whose stop event this stream
  // has just waited for): extra workgroups behind the log-prior ones, beside the allocation kernel of t - 1; k_reduce(t - 2) follows below
  CtArgs ct{};
  int n_ct = 0;
  if (h->ct_pending) { ct = ct_args(h, h->ct_pending); n_ct = (h->cfg.G + 2 * (RT / 64) - 1) / (2 * (RT / 64)); h->ct_pending = 0; }
  hipLaunchKernelGGL(k_side_lp, dim3(2 * N + h->nblkE + n_ct), dim3(RT), 0, h->side2, h->dev, t, nbP, 0, RecDst{}, SideDone{h->dFlags + 2, h->dFlags + 3, (unsigned)(2 * N + h->nblkE), t},
                     SideExtra{N, N, h->nblkE, t - 1, lpe_src(h, t - 1), 1}, ct);
  if (h->red_pending) { issue_reduce(h, h->red_t, h->red_row, tm, h->side2); h->red_pending = false; }
  h->flags_valid = true;
  h->side_ev_stale = true;
  h->side_valid = true;
  h->gate_f0 = 9;
}
// Rank learning: the hyper sweep of t+1 in two parts.  Early (released by k_edraw): the k_side kernels.  They hold 64+ VGPRs
// and cannot be scheduled on a CU whose SIMDs carry two waves of the rank sweep (230 VGPRs each): they run on the ~100 CUs
// the rank sweep leaves free and are done before k_zalloc starts.  Late (released by the rank sweep): the small log-prior
// kernels (16-28 VGPRs), which DO fit beside rank-sweep waves and delayed the whole co-resident grid at every factor, and
// Esum, whose flag releases the next iteration's draws and therefore has to come after them.
static void launch_side_early(bnmf_handle* h, uint32_t t) {
  const int nbP = (int)(((size_t)h->cfg.K * h->cfg.N + RT - 1) / RT);
  const int nbE = (int)(((size_t)h->cfg.N * h->cfg.G + RT - 1) / RT);
  hipStreamWaitEvent(h->side2, h->ev_draw, 0);
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_side, dim3(nbP), dim3(RT), 0, h->side2, h->dev, t, nbP, h->cfg.N, rec_at(h, t, fused_rec(h)), SideDone{});
  // Esum of t needs E only: summed here, beside the rank sweep (end of round 5).  Behind the rank sweep — where its flag has to be raised, see
  // launch_side_late — its 50 workgroups sat beside the allocation kernel for 121 us of a 10 us reduction, and the next k_pdraw polled for them
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_side, dim3(h->cfg.N), dim3(RT), 0, h->side2, h->dev, t, nbP, 0, RecDst{}, SideDone{});
  hipStreamWaitEvent(h->side, h->ev_draw, 0);
  dbg_delay(h, h->side);
  hipLaunchKernelGGL(k_side, dim3(nbE), dim3(RT), 0, h->side, h->dev, t, nbP, h->cfg.N + nbP, rec_at(h, t, fused_rec(h)), SideDone{h->dFlags, h->dFlags + 1, (unsigned)nbE, t});
  h->flags_valid = true;
}
static void launch_side_late(bnmf_handle* h, uint32_t t, Timer& tm) {
  hipStreamWaitEvent(h->side2, h->ev_rank, 0);
  // k_lpp rewrites lpPn slot (t-1) % 3, last read by k_reduce of iteration t-4 (side stream): order behind it
  if (h->red_issued) hipStreamWaitEvent(h->side2, h->ev_red, 0);
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_lpp, dim3(h->cfg.N), dim3(64), 0, h->side2, h->dev, t - 1);   // log-prior of the P just drawn
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_lpe, dim3(h->nblkE), dim3(ES_T), 0, h->side2, h->dev, t - 1, lpe_src(h, t - 1)); // ... and of the E just drawn
  // Esum's flag [3] last: it releases the next iteration's draws, which overwrite the P and E the two kernels above read (the sums
  // themselves were made by launch_side_early on this stream)
  dbg_delay(h, h->side2);
  hipLaunchKernelGGL(k_raise_flag, dim3(1), dim3(64), 0, h->side2, h->dFlags + 3, t);
  hipEventRecord(h->ev_sideP, h->side2);
  hipStreamWaitEvent(h->side, h->ev_sideP, 0);
  hipEventRecord(h->ev_side, h->side);
  h->side_ev_stale = false;
  h->side_valid = true;
  if (h->red_pending) { issue_reduce(h, h->red_t, h->red_row, tm); h->red_pending = false; }
}
template <typename KernelT, typename ArgT>
static int launch_z(bnmf_handle* h, uint32_t t, KernelT kern, const ArgT& arg, int zt) {
  if (h->z_attr_kernel != (const void*)kern) {           // once per handle (per device): allow > 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->z_attr_kernel = (const void*)kern;
  }
  hipLaunchKernelGGL(kern, dim3(h->z_grid), dim3(zt), h->z_lds, h->stream, arg, t, h->zg, h->z_ablate);
  return 0;
}
static ZArgs zargs(const bnmf_handle* h) {
  const Dev& d = h->dev;
  ZArgs za{d.K, d.G, d.N, d.maxM, d.k0, d.k1, d.M, d.P, d.E, d.A, d.ZsumK, d.ZsumG, d.Z, d.colsse, d.colll, d.colkl, d.lgfact, d.logm, nullptr, nullptr, 0u, nullptr};
  if (h->z_gate_next) { za.gate0 = h->dFlags + h->gate_f0; za.gate1 = h->dFlags + 3; za.gate_epoch = h->z_gate_next; za.gate_err = h->dErr; }
  return za;
}
template <bool SZ, int ZT_, bool DIAG>
static int launch_zreg_t(bnmf_handle* h, uint32_t t) {
  const ZArgs za = zargs(h);
  switch (h->zg.TR) {
    case 8: return launch_z(h, t, k_zalloc_reg<SZ, ZT_, 8, DIAG>, za, ZT_);
    case 16: return launch_z(h, t, k_zalloc_reg<SZ, ZT_, 16, DIAG>, za, ZT_);
    case 20: return launch_z(h, t, k_zalloc_reg<SZ, ZT_, 20, DIAG>, za, ZT_);
    default: return launch_z(h, t, k_zalloc_reg<SZ, ZT_, 24, DIAG>, za, ZT_);
  }
}
template <bool SZ, int ZT_, bool LEAN_>
static int launch_ztile(bnmf_handle* h, uint32_t t) {
  auto kern = k_zalloc_tile<SZ, ZT_, LEAN_>;
  if (h->z_attr_kernel != (const void*)kern) {
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->z_attr_kernel = (const void*)kern;
  }
  const ZArgs za = zargs(h);
  hipLaunchKernelGGL(kern, dim3(h->z_grid), dim3(ZT_), h->z_lds, h->stream, za, h->dMhatZ, t, h->ztg);
  hipLaunchKernelGGL(k_colmetrics<256>, dim3((h->cfg.G + 3) / 4), dim3(256), 0, h->stream, za, (const double*)h->dMhatZ);
  if (h->ztg.dbg) {                                        // BNMF_ZTDBG: section cycles (100 MHz s_memtime ticks) per launch
    unsigned long long v[8];
    hipStreamSynchronize(h->stream);
    hipMemcpy(v, h->ztg.dbg, sizeof v, hipMemcpyDeviceToHost);
    hipMemset(h->ztg.dbg, 0, sizeof v);
    if (v[0]) fprintf(stderr, "[ztile t=%u] waves %llu grid %d w %d lds %zu  per wave: phase1 %.1f  phase2 %.1f  flush %.1f  columns %.1f  kernel %.1f (s_memtime ticks)\n",
                      t, v[0], h->z_grid, h->z_zw, h->z_lds, (double)v[1] / v[0], (double)v[2] / v[0], (double)v[3] / v[0], (double)v[4] / v[0], (double)v[5] / v[0]);
  }
  return 0;
}
template <bool SZ, int ZT_>
static int launch_zalloc_t(bnmf_handle* h, uint32_t t) {
  if (h->z_tile) return (h->z_lean && ZT_ == 1024) ? launch_ztile<SZ, ZT_, (ZT_ == 1024)>(h, t) : launch_ztile<SZ, ZT_, false>(h, t);
  if (!h->z_reg) return launch_z(h, t, k_zalloc<SZ, ZT_>, h->dev, ZT_);
#ifdef BNMF_DIAG
  if (h->z_ablate) return launch_zreg_t<SZ, ZT_, true>(h, t);   // the DIAG instantiation honours BNMF_ABLATE
#endif
  return launch_zreg_t<SZ, ZT_, false>(h, t);
}
static int zs_prio() { static const int v = getenv("BNMF_ZSPRIO") ? atoi(getenv("BNMF_ZSPRIO")) : 1; return v; }   // A/B: 0 = the allocation kernel at default issue priority
// where the item records of iteration t go (save_Z on the sorted schedule): the sample's slot of the record ring, or the one buffer
static uint32_t* zs_rec_at(const bnmf_handle* h, uint32_t t) {
  if (!h->dZsRec) return nullptr;
  return h->dZsRecRing ? h->dZsRecRing + (size_t)((t - 1) % (uint32_t)h->wcap) * h->zs_recwords : h->dZsRec;
}
// Z[k, n, g] of iteration t from its records into h->dZ (main stream)
static void launch_zexpand(bnmf_handle* h, uint32_t t) {
  const ZSArgs sa{zargs(h), h->dZsItems, h->dZsBlocks, h->dZsCols, h->dZsM, h->zs_it16, h->zs_qmax, zs_rec_at(h, t), nullptr, 0, 0, h->dZsProf};
  hipLaunchKernelGGL(k_zexpand, dim3(h->zsg.nblocks), dim3(ZX_T), h->zx_lds, h->stream, sa, h->zx_cols);
  h->z_expanded_iter = (int)t;
}
static int ensure_Z(bnmf_handle* h) {
  if (!h->z_sort || !h->dZsRec || !h->cfg.save_Z || h->iter < 1 || h->z_expanded_iter == h->iter) return 0;
  HIPCHK(hipSetDevice(h->device));
  launch_zexpand(h, (uint32_t)h->iter);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
template <int ZT_>
static int launch_zsort_t(bnmf_handle* h, uint32_t t) {
  const ZSArgs sa{zargs(h), h->dZsItems, h->dZsBlocks, h->dZsCols, h->dZsM, h->zs_it16, h->zs_qmax, zs_rec_at(h, t), h->dZsMh + (size_t)(t % 3u) * h->cfg.K * h->cfg.G, zs_prio(), h->zs_shared ? 1 : 0, h->dZsProf};
  auto go = [&](auto kern) -> int {
    if (h->z_attr_kernel != (const void*)kern) {
      HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      h->z_attr_kernel = (const void*)kern;
    }
    hipLaunchKernelGGL(kern, dim3(h->zsg.nblocks), dim3(ZT_), h->zs_lds, h->stream, sa, t, h->zsg);
    return 0;
  };
#ifdef BNMF_FASTBUILD   /* builder's experiment builds only (one allocation kernel: the metric configuration's): never the product */
  if (h->zs_nblk != 4) return fail(BNMF_EMODEL, "BNMF_FASTBUILD: only N = 16..20");
  return h->zs_pk ? go(k_zalloc_sort<ZT_, 4, true>) : go(k_zalloc_sort<ZT_, 4, false>);
#else
  if (h->zs_pk) switch (h->zs_nblk) {
    case 1: return go(k_zalloc_sort<ZT_, 1, true>);
    case 2: return go(k_zalloc_sort<ZT_, 2, true>);
    case 3: return go(k_zalloc_sort<ZT_, 3, true>);
    case 4: return go(k_zalloc_sort<ZT_, 4, true>);
    default: return go(k_zalloc_sort<ZT_, 5, true>);
  }
  switch (h->zs_nblk) {
    case 1: return go(k_zalloc_sort<ZT_, 1, false>);
    case 2: return go(k_zalloc_sort<ZT_, 2, false>);
    case 3: return go(k_zalloc_sort<ZT_, 3, false>);
    case 4: return go(k_zalloc_sort<ZT_, 4, false>);
    default: return go(k_zalloc_sort<ZT_, 5, false>);
  }
#endif
}
static int launch_zsort(bnmf_handle* h, uint32_t t) {
#ifdef BNMF_FASTBUILD
  if (h->zs_w == 14) return launch_zsort_t<896>(h, t);
  if (h->zs_w != 12) return fail(BNMF_EMODEL, "BNMF_FASTBUILD: only 12 or 14 waves");
  return launch_zsort_t<768>(h, t);
#else
  switch (h->zs_w) {
    case 16: return launch_zsort_t<1024>(h, t);
    case 14: return launch_zsort_t<896>(h, t);
    case 12: return launch_zsort_t<768>(h, t);
    case 8: return launch_zsort_t<512>(h, t);
    case 6: return launch_zsort_t<384>(h, t);
    default: return launch_zsort_t<256>(h, t);
  }
#endif
}
static int launch_zstep(bnmf_handle* h, uint32_t t) {
  const ZPArgs pa{zargs(h), h->dZpItems, h->zp_it16 ? 1 : 0, h->dZpWgs, h->dZpBatches, h->dZpSteps, h->dZpCols};
  auto go = [&](auto kern) -> int {
    if (h->z_attr_kernel != (const void*)kern) {
      HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      h->z_attr_kernel = (const void*)kern;
    }
    hipLaunchKernelGGL(kern, dim3(h->zpg.nwg), dim3(h->zp_ns * 64), h->zp_lds, h->stream, pa, t, h->zpg);
    return 0;
  };
  return h->zp_gbp == 40 ? go(k_zalloc_step<4, 40, 8>) : go(k_zalloc_step<4, 32, 8>);
}
static int launch_zalloc(bnmf_handle* h, uint32_t t) {
  if (h->z_sort) {
    if (int rc = launch_zsort(h, t)) return rc;
    // save_Z: the items' records ARE the sample (zs_rec_at); Z is expanded from them when it is read (ensure_Z, bnmf_window)
    if (h->cfg.save_Z && h->zs_eager) launch_zexpand(h, t);
    return 0;
  }
  if (h->z_step) return launch_zstep(h, t);
  const bool sz = h->cfg.save_Z != 0;
#ifdef BNMF_FASTBUILD
  if (h->z_zw != 16) return fail(BNMF_EMODEL, "BNMF_FASTBUILD: only 16 waves");
  return sz ? launch_zalloc_t<true, 1024>(h, t) : launch_zalloc_t<false, 1024>(h, t);
#else
  switch (h->z_zw) {
    case 16: return sz ? launch_zalloc_t<true, 1024>(h, t) : launch_zalloc_t<false, 1024>(h, t);
    case 8: return sz ? launch_zalloc_t<true, 512>(h, t) : launch_zalloc_t<false, 512>(h, t);
    case 6: return sz ? launch_zalloc_t<true, 384>(h, t) : launch_zalloc_t<false, 384>(h, t);
    case 4: return sz ? launch_zalloc_t<true, 256>(h, t) : launch_zalloc_t<false, 256>(h, t);
    case 2: return sz ? launch_zalloc_t<true, 128>(h, t) : launch_zalloc_t<false, 128>(h, t);
    default: return sz ? launch_zalloc_t<true, 64>(h, t) : launch_zalloc_t<false, 64>(h, t);
  }
#endif
}
// sample_R then sample_An for n = 1..N (R/sample_params.R:67-74): one persistent launch for the N sequential updates
// row >= 0 (Gibbs sweep): the kernel also records A, R and sum(A) of the iteration (k_sumA's work)
static void launch_rank(bnmf_handle* h, uint32_t t, hipEvent_t stop = nullptr, int row = -1) {
  const int N = h->cfg.N;
  const int NB = (h->cfg.G + RK_MAXC - 1) / RK_MAXC;
  const size_t lds = (3 * (size_t)N + 1) * sizeof(double);            // A, sample_R weights, sample_An uniforms
  const RecDst rr = row >= 0 ? rec_at(h, t, fused_rec(h)) : RecDst{};
  auto go = [&](auto kern) {
    hipExtLaunchKernelGGL(kern, dim3(h->rank_grid), dim3(h->rank_half ? RK_TH : RK_T), (uint32_t)lds, h->stream, nullptr, stop, 0, h->dev, t, (unsigned long long*)h->dRankCol, NB, h->dErr + 1, h->dRankMhat, (unsigned long long*)h->dRankDbg, row, rr.A, rr.R);
  };
  const bool nrm = h->cfg.likelihood == BNMF_NORMAL;
  if (h->rank_half) { if (nrm) go(k_rank_sweep<true, true, true>); else go(k_rank_sweep<true, false, true>); }
  else if (h->rank_reg) { if (nrm) go(k_rank_sweep<true, true>); else go(k_rank_sweep<true, false>); }
  else { if (nrm) go(k_rank_sweep<false, true>); else go(k_rank_sweep<false, false>); }
}
// ids recorded per iteration (names(self$params) + names(self$prior_params), R/bayesNMF_sampler.R:245-252)
static std::vector<int> recorded_ids(const bnmf_handle* h) {
  std::vector<int> ids = {BNMF_P, BNMF_E, BNMF_A, BNMF_R};
  if (h->cfg.prior == BNMF_GAMMA) ids.insert(ids.end(), {BNMF_ALPHA_P, BNMF_BETA_P, BNMF_ALPHA_E, BNMF_BETA_E});
  else if (h->cfg.prior == BNMF_EXPONENTIAL) ids.insert(ids.end(), {BNMF_LAMBDA_P, BNMF_LAMBDA_E});
  else ids.insert(ids.end(), {BNMF_MU_P, BNMF_SIGMASQ_P, BNMF_MU_E, BNMF_SIGMASQ_E});
  if (h->cfg.MH) ids.insert(ids.end(), {BNMF_ACC_P, BNMF_ACC_E});
  if (h->cfg.likelihood == BNMF_NORMAL) ids.push_back(BNMF_SIGMASQ);
  return ids;
}
static int ensure_rings(bnmf_handle* h) {
  if (h->cfg.window <= 0) return 0;
  h->wcap = h->cfg.window + 1;
  for (int id : recorded_ids(h)) {
    Arr& a = h->arr[id];
    if (!a.ring) if (int rc = ring_alloc(h->device, (size_t)h->wcap * id_len(h, id) * sizeof(double), &a.ring)) return rc;
  }
  if (h->dZ && h->z_sort && h->dZsRec) {                   // samples$Z on the sorted schedule: a ring of item records (zs_rec_at)
    if (!h->dZsRecRing) {
      const double gb = (double)h->wcap * (double)h->zs_recwords * 4.0 / 1e9;
      const char* e = getenv("BNMF_ZRING_GB");
      if (gb <= (e ? atof(e) : 32.0)) HIPCHK(dmalloc(&h->dZsRecRing, (size_t)h->wcap * h->zs_recwords * sizeof(uint32_t)));
    }
  } else if (h->dZ && !h->zring) {                         // samples$Z (R/bayesNMF_sampler.R:245-252): K*N*G ints per kept sample
    const double gb = (double)h->wcap * (double)id_len(h, BNMF_Z) * 4.0 / 1e9;
    const char* e = getenv("BNMF_ZRING_GB");
    if (gb <= (e ? atof(e) : 32.0)) HIPCHK(dmalloc(&h->zring, (size_t)h->wcap * id_len(h, BNMF_Z) * sizeof(int32_t)));
  }
  return 0;
}
static void record_Z(bnmf_handle* h, uint32_t t) {
  if (!h->zring) return;                                   // (sorted schedule: the allocation kernel wrote the sample's records into its ring slot)
  const size_t len = id_len(h, BNMF_Z);
  hipMemcpyAsync(h->zring + (size_t)((t - 1) % (uint32_t)h->wcap) * len, h->dZ, len * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream);
}
// the Gibbs sweep (Poisson, no MH) records inside its producers: k_pdraw (P, and A, R when the rank is fixed), k_edraw (E),
// k_side (prior parameters), k_sumA (A, R when the rank is learned).  The MH / Normal sweeps copy with k_record.
static bool fused_rec(const bnmf_handle* h) { return h->cfg.window > 0 && !h->cfg.MH && h->cfg.likelihood == BNMF_POISSON; }
static double* ring_at(const bnmf_handle* h, int id, uint32_t t) {
  const Arr& a = h->arr[id];
  return a.ring ? a.ring + (size_t)((t - 1) % (uint32_t)h->wcap) * id_len(h, id) : nullptr;
}
static RecDst rec_at(const bnmf_handle* h, uint32_t t, bool on) {
  RecDst r{};
  if (!on || h->wcap <= 0) return r;
  r.P = ring_at(h, BNMF_P, t); r.E = ring_at(h, BNMF_E, t); r.A = ring_at(h, BNMF_A, t); r.R = ring_at(h, BNMF_R, t);
  static const int PP[3][4] = {{BNMF_MU_P, BNMF_SIGMASQ_P, BNMF_MU_E, BNMF_SIGMASQ_E},      // indexed by the prior enum
                               {BNMF_LAMBDA_P, -1, BNMF_LAMBDA_E, -1},
                               {BNMF_ALPHA_P, BNMF_BETA_P, BNMF_ALPHA_E, BNMF_BETA_E}};
  const int* pp = PP[h->cfg.prior];
  for (int i = 0; i < 4; ++i) r.pp[i] = pp[i] >= 0 ? ring_at(h, pp[i], t) : nullptr;
  return r;
}
static int record_args(bnmf_handle* h, uint32_t t, RecArgs& ra) {
  ra = RecArgs{}; ra.n = 0; ra.R = nullptr; ra.Rdst = nullptr;
  const int W = h->cfg.window;
  if (W <= 0) return 0;
  const size_t slot = (size_t)((t - 1) % (uint32_t)h->wcap);
  for (int id : recorded_ids(h)) {
    Arr& a = h->arr[id];
    const size_t len = id_len(h, id);
    if (!a.ring) return fail(BNMF_ESTATE, "record: ring of id %d missing", id);
    if (id == BNMF_R) { ra.R = h->dR; ra.Rdst = a.ring + slot; continue; }
    if (!a.d) continue;
    ra.src[ra.n] = a.d + (is_prior_param(id) ? (size_t)(t & 1u) * len : 0);
    ra.dst[ra.n] = a.ring + slot * len;
    ra.len[ra.n] = len;
    ra.n++;
  }
  return 0;
}
static int launch_record(bnmf_handle* h, uint32_t t) {
  RecArgs ra;
  if (int rc = record_args(h, t, ra)) return rc;
  if (ra.n > 0 || ra.Rdst) hipLaunchKernelGGL(k_record, dim3(512), dim3(256), 0, h->stream, ra);
  return 0;
}
// k_reduce of iteration t: on the side stream, after the main stream has finished k_zalloc / metrics of t
// metrics of iteration t: sum(A) now (main stream, right after the rank update); the canonical reductions later
static void launch_reduce(bnmf_handle* h, uint32_t t, int row, Timer& tm, bool rank_wrote = false) {   // rank_wrote: k_rank_sweep recorded A, R, sum(A)
  // sum(A) and the A-masked acceptance sum (MH / Normal sweeps, init); the Gibbs sweep's rank kernel writes sum(A), A, R itself
  if (h->cfg.learning_rank && !rank_wrote) hipLaunchKernelGGL(k_sumA, dim3(1), dim3(64), 0, h->stream, h->dev, row, (const double*)accPn_slot(h, t), rec_at(h, t, fused_rec(h)));
  h->red_pending = true; h->red_t = t; h->red_row = row;
}
// ... or at once (init, end of a run), on the MAIN stream: behind the allocation / metrics kernel of the last iteration in
// stream order, and behind the log-prior workgroups of that iteration on the side streams through ONE event wait (they ran
// beside the allocation kernel and are long done).  On the side stream it took two cross-stream hops in a row (side waits
// for main, main waits for side: ~15 us each) at the end of every bnmf_run.
static void flush_reduce(bnmf_handle* h, Timer& tm) {
  if (!h->red_pending) return;
  if (h->side_ev_stale) hipEventRecord(h->ev_sideP, h->side2);   // fixed-rank sweep: k_lpp / k_lpe workgroups live on side2
  hipStreamWaitEvent(h->stream, h->ev_sideP, 0);
  issue_reduce(h, h->red_t, h->red_row, tm, h->stream);
  h->red_pending = false;
}
// P and E updates of the MH models (R/sample_params.R:56-64 with sample_Pn/_En -> *_normal -> MH_*_poisson)
struct MhPipe { MhETail et; MhPTail pt; };
static void launch_mh_PE(bnmf_handle* h, uint32_t t, int converged, bool poll = false, const MhPipe* pp = nullptr) {
  const int K = h->cfg.K, N = h->cfg.N, G = h->cfg.G, S = h->mh_S;
  const bool normal = h->cfg.likelihood == BNMF_NORMAL;
  const int mhstep = (h->cfg.MH && converged && !normal) ? 1 : 0;
  if (pp) {
    if (!h->mh_pipe_valid) {                                 // first hosted sweep after init / set_array / a sweep of the other form
      hipMemsetAsync(h->dNzE + 2 * N, 0, 4 * N * sizeof(int), h->stream);
      hipLaunchKernelGGL(k_mh_nz, dim3(N), dim3(256), 0, h->stream, h->dev, h->dNzE + 2 * N + ((t - 1) & 1u) * N);
      h->mh_pipe_valid = true; h->mh_prep_valid = false;
    }
  } else if (!h->mh_prep_valid) {                            // first sweep after init / set_array; afterwards k_mh_tail prepares them
    hipMemsetAsync(h->dNzE, 0, 2 * N * sizeof(int), h->stream);         // nzE[N], nzP[N]
    hipLaunchKernelGGL(k_mh_nz, dim3(N), dim3(256), 0, h->stream, h->dev, h->dNzE);
    h->mh_prep_valid = true; h->mh_pipe_valid = false;
  }
  double* accP = h->arr[BNMF_ACC_P].d; double* accE = h->arr[BNMF_ACC_E].d;
  const bool regP = S <= MHP_W;                              // one 320-column segment per wave: the row's cells stay in registers
  const size_t ldsP = (4 * (size_t)S + 2 * N + 2 + (size_t)(PRE_W + 2) * N + ((regP && mhstep) ? (size_t)MH_CPL * MHP_T : 0)) * sizeof(double);
  const bool pipe = pp != nullptr;                         // hosted form (sweep_mh): parity flag buffers, hosted workgroups behind the rows / the column blocks
  int* const nzb = h->dNzE + 2 * N;                        // nzE[2][N], nzP[2][N]
  const int* nzE_in = pipe ? nzb + ((t - 1) & 1u) * N : h->dNzE;
  int* nzP_io = pipe ? nzb + 2 * N + (t & 1u) * N : h->dNzE + N;
  const MhETail et = pipe ? pp->et : MhETail{};
  const int nhostP = pipe ? mh_etail_groups(et, N, MHP_T / ES_T) : 0;
  const size_t ldsPx = pipe ? std::max<size_t>(ldsP, MHP_T * sizeof(double)) : ldsP;
  auto goP = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(K + nhostP), dim3(MHP_T), ldsPx, h->stream, h->dev, t, S, nzE_in, nzP_io, accP, h->dMhat, h->dMhat + (size_t)K * h->cfg.G,
                                                poll ? SideWait{h->dFlags + 1, h->dFlags + 1, t, h->dErr} : SideWait{}, et); };
  if (normal) { if (regP) goP(k_mh_prow<true, true, false>); else goP(k_mh_prow<true, false, false>); }
  else if (mhstep) { if (regP) goP(k_mh_prow<false, true, true>); else goP(k_mh_prow<false, false, true>); }
  else { if (regP) goP(k_mh_prow<false, true, false>); else goP(k_mh_prow<false, false, false>); }
  int grid = (G + 3) / 4; if (grid > 2048) grid = 2048;
  if (h->mhe16) {                                          // several columns per wave
    // lanes per column: 16 for the Gibbs-only sweep, 32 with the MH step (measured at config 3: 117 / 126 us and 276 / 205 us)
    const int gw = h->mhe_gw ? h->mhe_gw : (mhstep ? 32 : 16), cpw = 64 / gw;
    int g16 = ((G + cpw - 1) / cpw + 3) / 4; if (g16 > 2048) g16 = 2048;
    const size_t lds16 = (4 * (size_t)cpw * N * (1 + PRE_W) + 2 * (size_t)N) * sizeof(double);
    const MhPTail pt = pipe ? pp->pt : MhPTail{};
    const int nhostE = pipe ? mh_ptail_blocks(pt, N, h->cfg.MH) : 0;
    const size_t lds16x = pipe ? std::max<size_t>(lds16, RT * sizeof(double)) : lds16;
    int* nzE_set = pipe ? nzb + (t & 1u) * N : nullptr;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(g16 + nhostE), dim3(MHE_T), lds16x, h->stream, h->dev, t, (const int*)nzP_io, accE, 0, nzE_set, g16, pt); };
    if (K <= 96 && !h->mhe_k128) {                         // register arrays for 96 rows (BNMF_MHE_K128=1: the 128-row form)
      if (gw == 16) { if (mhstep) go(k_mh_ecol16<false, true, 16, 96>); else go(k_mh_ecol16<false, false, 16, 96>); }
      else { if (mhstep) go(k_mh_ecol16<false, true, 32, 96>); else go(k_mh_ecol16<false, false, 32, 96>); }
    } else if (gw == 16) { if (mhstep) go(k_mh_ecol16<false, true, 16>); else go(k_mh_ecol16<false, false, 16>); }
    else { if (mhstep) go(k_mh_ecol16<false, true, 32>); else go(k_mh_ecol16<false, false, 32>); }
  } else
  hipLaunchKernelGGL(k_mh_ecol<false>, dim3(grid), dim3(MHE_T), h->mhe_lds, h->stream, h->dev, t, mhstep, (const int*)(h->dNzE + N), accE, 0);
}
static int launch_mh_metrics(bnmf_handle* h, uint32_t t, bool cells, bool with_record = false, bool with_side = false) {   // with_record: record_sample inside k_mh_tail; with_side: and the hyper sweep of t + 1
  const int draw_sig = h->cfg.likelihood == BNMF_NORMAL ? 1 : 0;
  if (draw_sig) cells = true;                 // sigmasq is drawn after R, A (R/sample_params.R:86-88) in the metrics pass
  const int N = h->cfg.N, G = h->cfg.G;
  if (cells) {
    if (h->mhe16) {
      const int gw = h->mhe_gw ? h->mhe_gw : 16, cpw = 64 / gw;
      int g16 = ((G + cpw - 1) / cpw + 3) / 4; if (g16 > 2048) g16 = 2048;
      const size_t lds16 = (4 * (size_t)cpw * N * (1 + PRE_W) + 2 * (size_t)N) * sizeof(double);
      const bool k96 = h->cfg.K <= 96 && !h->mhe_k128;
      if (gw == 16 && k96) hipLaunchKernelGGL((k_mh_ecol16<true, false, 16, 96>), dim3(g16), dim3(MHE_T), lds16, h->stream, h->dev, t, (const int*)nullptr, h->arr[BNMF_ACC_E].d, draw_sig, (int*)nullptr, g16, MhPTail{});
      else if (gw == 16) hipLaunchKernelGGL((k_mh_ecol16<true, false, 16>), dim3(g16), dim3(MHE_T), lds16, h->stream, h->dev, t, (const int*)nullptr, h->arr[BNMF_ACC_E].d, draw_sig, (int*)nullptr, g16, MhPTail{});
      else if (k96) hipLaunchKernelGGL((k_mh_ecol16<true, false, 32, 96>), dim3(g16), dim3(MHE_T), lds16, h->stream, h->dev, t, (const int*)nullptr, h->arr[BNMF_ACC_E].d, draw_sig, (int*)nullptr, g16, MhPTail{});
      else hipLaunchKernelGGL((k_mh_ecol16<true, false, 32>), dim3(g16), dim3(MHE_T), lds16, h->stream, h->dev, t, (const int*)nullptr, h->arr[BNMF_ACC_E].d, draw_sig, (int*)nullptr, g16, MhPTail{});
    } else {
      int grid = (G + 3) / 4; if (grid > 2048) grid = 2048;
      hipLaunchKernelGGL(k_mh_ecol<true>, dim3(grid), dim3(MHE_T), h->mhe_lds, h->stream, h->dev, t, 0, (const int*)nullptr, h->arr[BNMF_ACC_E].d, draw_sig);
    }
  }
  // log-priors and acceptance sums, (for the next iteration's P sweep) Et, nzE, nzP = 0, and record_sample: one launch
  RecArgs ra{};
  if (with_record) { if (int rc = record_args(h, t, ra)) return rc; }
  const int nrec = (ra.n > 0 || ra.Rdst) ? 256 : 0;
  // the canonical reductions of the iteration BEFORE ride in this launch when the hyper sweep runs on the main stream (launch_side_main):
  // as a kernel of their own on the side stream nothing ordered the writers of their slot, three iterations on, behind them
  RedSlots rs{};
  if (h->red_pending && h->mh_side_main) {
    Dev dr = h->dev;
    set_slot(h, dr, h->red_t);
    rs = RedSlots{dr.colsse, dr.colll, dr.colkl, dr.lpE_part, dr.lpPn, accPn_slot(h, h->red_t), accEp_slot(h, h->red_t), h->red_row, 1};
    h->red_pending = false;
  }
  SideInTail sx{};
  if (with_side) {                                           // launch_side_main's kernel as the first blocks of this one
    const int nbP = (int)(((size_t)h->cfg.K * N + RT - 1) / RT), nbE = (int)(((size_t)N * G + RT - 1) / RT);
    sx = SideInTail{N + nbP + nbE, nbP, t + 1, rec_at(h, t + 1, fused_rec(h))};
    h->flags_valid = false; h->side_valid = true; h->side_main = true;
  }
  hipLaunchKernelGGL(k_mh_tail, dim3(sx.n + 2 * N + h->nblkE + nrec + (rs.on ? (h->cfg.MH ? 5 : 4) : 0)), dim3(ES_T), 0, h->stream, h->dev, t, (const double*)h->arr[BNMF_ACC_P].d, accPn_slot(h, t),
                     (const double*)h->arr[BNMF_ACC_E].d, accEp_slot(h, t), h->dNzE, h->dNzE + N, h->nblkE, ra, nrec, rs, sx);
  h->mh_prep_valid = true;
  return 0;
}
// record_sample's arrays of iteration t in two groups: what the row sweep of t + 1 rewrites (P, its prior parameters of t, its acceptance
// rates: copied beside the column sweep of t) and the rest (copied beside the row sweep of t + 1)
static int record_args_split(bnmf_handle* h, uint32_t t, RecArgs& raP, RecArgs& raE) {
  RecArgs ra;
  if (int rc = record_args(h, t, ra)) return rc;
  raP = RecArgs{}; raE = RecArgs{};
  raE.R = ra.R; raE.Rdst = ra.Rdst;
  const int pids[] = {BNMF_P, BNMF_ACC_P, BNMF_MU_P, BNMF_SIGMASQ_P, BNMF_LAMBDA_P, BNMF_ALPHA_P, BNMF_BETA_P};
  for (int j = 0; j < ra.n; ++j) {
    bool isP = false;
    for (int id : pids) { const Arr& a = h->arr[id]; if (a.ring && ra.dst[j] >= a.ring && ra.dst[j] < a.ring + (size_t)h->wcap * id_len(h, id)) isP = true; }
    RecArgs& o = isP ? raP : raE;
    o.src[o.n] = ra.src[j]; o.dst[o.n] = ra.dst[j]; o.len[o.n] = ra.len[j]; o.n++;
  }
  return 0;
}
static MhETail mh_etail_args(bnmf_handle* h, uint32_t te, uint32_t t_next, int& rc) {   // the E side of iteration te (0: none pending); t_next: the iteration of the next column sweep
  const int N = h->cfg.N;
  MhETail et{};
  rc = 0;
  et.nz_zero = h->dNzE + 2 * N + (t_next & 1u) * N;
  if (!te) return et;
  RecArgs raP;
  if ((rc = record_args_split(h, te, raP, et.ra))) return et;
  et.on = 1; et.t = te;
  et.nbE = (int)(((size_t)N * h->cfg.G + RT - 1) / RT); et.nblkE = h->nblkE;
  et.nrec = (et.ra.n > 0 || et.ra.Rdst) ? 128 : 0;
  et.accE = h->arr[BNMF_ACC_E].d; et.accE_part = accEp_slot(h, te); et.lpE_part = h->dlpE + (size_t)(te % 3u) * h->nblkE;
  return et;
}
// the E side of the last iteration of a call: no row sweep behind it
static int flush_mh_etail(bnmf_handle* h) {
  if (!h->mh_etail_pending) return 0;
  int rc = 0;
  const MhETail et = mh_etail_args(h, h->mh_etail_pending, h->mh_etail_pending + 1, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mh_etail, dim3(mh_etail_groups(et, h->cfg.N, 4)), dim3(1024), 0, h->stream, h->dev, et);
  h->mh_etail_pending = 0;
  return 0;
}
// The hosted form of the sweep (Poisson MH models at fixed rank, k_mh_ecol16): two launches per iteration, k_mh_tail's work inside them (mh.h)
static int sweep_mh_pipe(bnmf_handle* h, int row, int converged, Timer& tm) {
  h->iter += 1;
  const uint32_t t = (uint32_t)h->iter;
  const int N = h->cfg.N;
  use_slot(h, t);
  if (!h->side_valid) launch_side(h, t, tm);              // first sweep after init / set_array: the prior parameters of t on the side streams
  if (!h->side_main) { refresh_side_events(h); hipStreamWaitEvent(h->stream, h->ev_side, 0); hipStreamWaitEvent(h->stream, h->ev_sideP, 0); }
  MhPipe pp{};
  int rc = 0;
  pp.et = mh_etail_args(h, h->mh_etail_pending, t, rc);
  if (rc) return rc;
  MhPTail& pt = pp.pt;
  pt.on = 1; pt.t = t;
  pt.nbP = (int)(((size_t)h->cfg.K * N + RT - 1) / RT);
  RecArgs raE;
  if ((rc = record_args_split(h, t, pt.ra, raE))) return rc;
  pt.nrec = pt.ra.n > 0 ? 8 : 0;
  pt.accP = h->arr[BNMF_ACC_P].d; pt.accPn = accPn_slot(h, t);
  pt.nblkE = h->nblkE;
  pt.nz_zero = h->dNzE + 2 * N + 2 * N + ((t + 1) & 1u) * N;
  if (h->red_pending) {                                    // k_reduce's work for the iteration before: its E-side sums are issued with pp.et above
    Dev dr = h->dev;
    set_slot(h, dr, h->red_t);
    pt.rs = RedSlots{dr.colsse, dr.colll, dr.colkl, dr.lpE_part, dr.lpPn, accPn_slot(h, h->red_t), accEp_slot(h, h->red_t), h->red_row, 1};
    h->red_pending = false;
  }
  dbg_delay_main(h);
  launch_mh_PE(h, t, converged, false, &pp);
  dbg_delay_main(h);
  h->mh_etail_pending = t;
  h->flags_valid = false; h->side_valid = true; h->side_main = true;
  launch_reduce(h, t, row, tm);
  return 0;
}
static int sweep_mh(bnmf_handle* h, int row, int converged, Timer& tm) {
  if (h->mh_pipe && !tm.on) return sweep_mh_pipe(h, row, converged, tm);
  if (int rc = flush_mh_etail(h)) return rc;
  h->iter += 1;
  const uint32_t t = (uint32_t)h->iter;
  use_slot(h, t);
  if (!h->side_valid) launch_side(h, t, tm);
  // prior parameters of iteration t: in the steady state the P-row kernel polls the flag k_side publishes (a stream wait is a
  // barrier packet: ~16 us of bubble per iteration here); after init / set_array / in profile mode a stream wait
  const bool on_main = h->side_main;                       // the hyper sweep of t ran on this stream (launch_side_main): nothing to wait for
  const bool poll = !on_main && h->flags_valid && !tm.on && !h->serial;
  if (!poll && !on_main) { refresh_side_events(h); hipStreamWaitEvent(h->stream, h->ev_side, 0); hipStreamWaitEvent(h->stream, h->ev_sideP, 0); }
  dbg_delay_main(h);
  tm.begin(KN_MH, h->stream); launch_mh_PE(h, t, converged, poll); tm.end(KN_MH, h->stream);
  dbg_delay_main(h);
  // the hyper sweep of t + 1: on the main stream — inside k_mh_tail below (its own launch in profile mode, which times it) — or on the side stream
  const bool side_in_tail = h->mh_side_main && !tm.on && h->mh_side_tail;
  if (side_in_tail) {} else if (h->mh_side_main) launch_side_main(h, t + 1, tm); else launch_side(h, t + 1, tm, !tm.on);
  if (h->cfg.learning_rank) { tm.begin(KN_RANK, h->stream); launch_rank(h, t); tm.end(KN_RANK, h->stream); }
  // record_sample rides in k_mh_tail: after sample_sigmasq, like record_sample (:279) after sample_params (:276)
  tm.begin(KN_OTHER, h->stream); if (int rc = launch_mh_metrics(h, t, h->cfg.learning_rank != 0, true, side_in_tail)) return rc; tm.end(KN_OTHER, h->stream);
  launch_reduce(h, t, row, tm);
  return 0;
}
// Merged draw kernel + gate at the end of the allocation kernel (k_draw, zalloc_reg.h): pays when the allocation kernel is long
// enough to cover the side streams' kernels that its last lane waits for: 107 -> 92.5 us per iteration at K = 96, G = 10,000, 59.4 -> 52.7 at
// G = 3,000, but 45.2 -> 51.1 us at G = 2,000, where they outlast the kernel and the gate puts them on the main stream's path.
// BNMF_GATE=0 / 1 forces it off / on (diagnostics).
static bool gate_enabled(const bnmf_handle* h) {
  if (h->gate_forced >= 0) return h->gate_forced != 0;
  // tools/gatesize.py (us per iteration without / with, K = 96, N = 20, recording on).  Round 3: 45.2 / 51.1 at G = 2,000; 59.4 / 52.7 at G = 3,000;
  // 107 / 92.5 at G = 10,000 -> from 250,000 cells.  Round 5 (the sorted schedule without metric tasks, the two draw kernels' and the side
  // kernels' chains shortened): 48.3 / 55.4 at G = 3,000; 56.4 / 58.9 at 4,000; 63.6 / 59.3 at 5,000; 72.4 / 67.0 at 7,000; 85.9 / 84.1 at
  // 10,000 -> the crossover has moved up; with the quads per item chosen per data set (build_zsort): 45.5 / 52.9 at G = 3,000; 49.0 / 54.3 at 4,000;
  // 59.8 / 60.6 at 5,000; 72.7 / 68.9 at 7,000; 85.6 / 83.5 at 10,000
  return (size_t)h->cfg.K * h->cfg.G >= 550000;
}
static int sweep(bnmf_handle* h, int row, Timer& tm) {
  h->iter += 1;
  const uint32_t t = (uint32_t)h->iter;
  use_slot(h, t);
  const bool rec = fused_rec(h);
  if (!h->side_valid) launch_side(h, t, tm);               // first sweep after init / set_array
  // The log-prior kernel of iteration t-1 (k_lpe, side stream) is ordered only behind its own inputs, not before this
  // iteration's k_edraw.  With recording on it reads E_{t-1} from the ring; without a ring E is double-buffered: this k_edraw
  // writes the buffer that held E_{t-2}.  Nothing below reads E_{t-1}: the draws use ZsumK / Psum / Esum, everything after
  // k_edraw works on E_t.  (The P side needs nothing: k_lpp precedes Esum, whose flag releases k_pdraw.)
  if (!lpe_from_ring(h)) {
    if (!h->E_alt) {
      HIPCHK(dmalloc(&h->E_alt, (size_t)h->cfg.N * h->cfg.G * sizeof(double)));
      HIPCHK(hipMemsetAsync(h->E_alt, 0, (size_t)h->cfg.N * h->cfg.G * sizeof(double), h->stream));
    }
    std::swap(h->arr[BNMF_E].d, h->E_alt);
    h->dev.E = h->arr[BNMF_E].d;
  }
  // prior parameters + Esum of iteration t: in the steady state k_pdraw polls the flags their kernels publish (no barrier
  // packet on the main stream); after init / set_array / in profile mode a stream wait
  const bool poll = h->flags_valid && !tm.on && !h->serial;
  if (!poll) { refresh_side_events(h); hipStreamWaitEvent(h->stream, h->ev_side, 0); }
  dbg_delay_main(h);
  if (tm.on) {                                             // profile mode: one kernel at a time
    flush_colterms(h);
    tm.begin(KN_PDRAW, h->stream); launch_pdraw(h, t, 0, rec); tm.end(KN_PDRAW, h->stream);
    tm.begin(KN_EDRAW, h->stream); launch_edraw(h, t, 0, rec); tm.end(KN_EDRAW, h->stream);
    launch_side(h, t + 1, tm);
  } else if (gate_enabled(h) && !h->cfg.learning_rank && h->z_reg && !h->z_tile && (!poll || h->z_gated_for == t)) {
    // merged draw kernel: the allocation kernel of t-1 has waited for this iteration's hyper sweep (its gate), or the main
    // stream has (the event wait above: first sweep after init / set_array, serial mode)
    // workgroup width: the E elements spread over (almost) all CUs in ONE round of workgroups — 1,024-lane workgroups left
    // 60 of 256 CUs idle at N G = 200,000 — while the workgroup count stays small (the gap to the next kernel grows with it)
    if (!h->draw_bw) {
      hipDeviceProp_t pr;
      HIPCHK(hipGetDeviceProperties(&pr, h->device));
      const size_t per_cu = ((size_t)h->cfg.N * h->cfg.G + pr.multiProcessorCount - 1) / pr.multiProcessorCount;
      size_t bw = ((per_cu + 63) / 64) * 64 + 64;        // one wave of slack: a few CUs take two small workgroups rather than one a second round
      h->draw_bw = (int)std::min<size_t>(DW, std::max<size_t>(256, bw));
      if (const char* e = getenv("BNMF_DRAWBW")) { const int v = atoi(e); if (v >= 64 && v <= DW && v % 64 == 0) h->draw_bw = v; }   // diagnostics
    }
    const unsigned bw = (unsigned)h->draw_bw;
    const unsigned nE = (unsigned)(((size_t)h->cfg.N * h->cfg.G + bw - 1) / bw);
    hipExtLaunchKernelGGL(k_draw, dim3(h->cfg.N + nE), dim3(bw), 0, h->stream, nullptr, h->ev_draw, 0, h->dev, t, rec_at(h, t, rec),
                          SideDone{h->dFlags + 5, h->dFlags + 6, (unsigned)h->cfg.N, t}, SideWait{h->dFlags + 6, h->dFlags + 6, t, h->dErr},
                          rec_at(h, t + 1, rec), SideDone{h->dFlags, h->dFlags + 1, nE, t + 1}, h->dDrawOwn, ++h->draw_seq, h->dbg_draw_no_p);

    launch_side_merged(h, t + 1, tm);
  } else {
    // The side work of this iteration may have been issued by launch_side_merged (the sweep before took the merged path without
    // arming the allocation kernel's gate: first sweep after init / set_array).  Its P-side sweep then runs on `side` under flag
    // [9], which k_pdraw does not poll ([1] was raised by k_draw, [3] covers side2 only): the main stream waits for it here, and
    // with it launch_side_P below (released by k_pdraw's stop event) cannot overwrite the slot that sweep still reads.
    if (poll && h->gate_f0 == 9) { hipEventRecord(h->ev_side, h->side); hipStreamWaitEvent(h->stream, h->ev_side, 0); }
    if (h->cfg.learning_rank) flush_colterms(h);           // (fixed rank: launch_side_E below takes the column terms of t - 1 along)
    // completion events ride on the dispatches themselves (stop events): no marker packets on the main stream
    hipExtLaunchKernelGGL(k_pdraw, dim3(h->cfg.N), dim3(PD_T), (uint32_t)(2 * (size_t)h->cfg.K * sizeof(double)), h->stream,
                          nullptr, h->ev_p, 0, h->dev, t, 0, 0, rec_pdraw(h, t, rec),
                          poll ? SideWait{h->dFlags + 1, h->dFlags + 3, t, h->dErr} : SideWait{});
    h->gate_f0 = 1;
    if (!h->cfg.learning_rank) {
      launch_side_P(h, t + 1);
      hipExtLaunchKernelGGL(k_edraw, dim3(h->nblkE), dim3(ES_T), 0, h->stream, nullptr, h->ev_draw, 0, h->dev, t, 0, 0, rec_at(h, t, rec).E);
      launch_side_E(h, t + 1, tm);                         // overlaps k_zalloc below
    } else {
      // rank learning: every workgroup of the rank sweep waits for all others at every factor, so a kernel sharing a CU
      // with one of them delays the whole grid: see launch_side_early / launch_side_late
      hipExtLaunchKernelGGL(k_edraw, dim3(h->nblkE), dim3(ES_T), 0, h->stream, nullptr, h->ev_draw, 0, h->dev, t, 0, 0, rec_at(h, t, rec).E);
      launch_side_early(h, t + 1);
      launch_rank(h, t, h->ev_rank, row);
      launch_side_late(h, t + 1, tm);
    }
  }
  if (h->cfg.learning_rank && tm.on) { tm.begin(KN_RANK, h->stream); launch_rank(h, t, nullptr, row); tm.end(KN_RANK, h->stream); }
  const bool gate = gate_enabled(h) && !h->cfg.learning_rank && poll && h->z_reg && !h->z_tile;
  h->z_gate_next = gate ? t + 1 : 0u;
  dbg_delay_main(h);
  tm.begin(KN_ZALLOC, h->stream);
  if (int rc = launch_zalloc(h, t)) return rc;
  tm.end(KN_ZALLOC, h->stream);
  if (h->z_sort) {
    h->ct_pending = t;
    if (tm.on) { tm.begin(KN_OTHER, h->stream); flush_colterms(h); tm.end(KN_OTHER, h->stream); }   // profile mode: the column terms as a launch of their own ("other")
  }
  h->z_gated_for = h->z_gate_next; h->z_gate_next = 0;
  record_Z(h, t);
  launch_reduce(h, t, row, tm, h->cfg.learning_rank != 0);
  return 0;
}

extern "C" {

int bnmf_init(bnmf_handle* h, double* metrics_row1) {
  if (!h) return fail(BNMF_EINVAL, "bnmf_init: null handle");
  HIPCHK(hipSetDevice(h->device));
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_init: the handle timed out inside a kernel; destroy it");
  if (h->inited) {
    // Re-initialisation restarts the iteration counter at 1, and with it every epoch the kernels compare their sync words
    // against (flags < epoch, the rank sweep's granule tags): stale words from the first life of the handle would satisfy
    // those waits at once.  Drain the streams and clear the words and the host-side pipeline state.
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->side));
    HIPCHK(hipStreamSynchronize(h->side2));
    HIPCHK(hipMemset(h->dFlags, 0, 64));
    memset(h->hErr, 0, 64);
    if (h->dRankSync) HIPCHK(hipMemset(h->dRankSync, 0, 32));
    if (h->dRankCol) HIPCHK(hipMemset(h->dRankCol, 0, (size_t)RK_REP * 4 * 2 * (((size_t)h->cfg.G + RK_MAXC - 1) / RK_MAXC) * sizeof(double)));
    HIPCHK(hipMemset(h->dZsumK, 0, (size_t)h->cfg.N * h->cfg.G * sizeof(int32_t)));
    HIPCHK(hipMemset(h->dZsumG, 0, (size_t)h->cfg.K * h->cfg.N * sizeof(int32_t)));
    h->side_valid = false; h->side_main = false; h->flags_valid = false; h->z_gate_next = 0; h->z_gated_for = 0; h->gate_f0 = 1;
    h->side_ev_stale = false; h->red_on_side2 = false; h->red_pending = false; h->red_issued = false; h->mh_prep_valid = false; h->mh_pipe_valid = false; h->mh_etail_pending = 0; h->ct_pending = 0; h->z_expanded_iter = 0;
    h->inited = false;
  }
  const bnmf_config& c = h->cfg;
  const int N = c.N;
  const long KN = (long)c.K * N, NG = (long)N * c.G;
  struct Spec { int id; uint32_t var; int side; int hs, hr; };
  std::vector<Spec> specs;
  if (c.prior == BNMF_GAMMA) {
    if (int rc = need_hyper(h, {BNMF_HA_P, BNMF_HB_P, BNMF_HC_P, BNMF_HD_P, BNMF_HA_E, BNMF_HB_E, BNMF_HC_E, BNMF_HD_E})) return rc;
    specs = {{BNMF_BETA_P, BNMF_V_BETA_P, 0, BNMF_HA_P, BNMF_HB_P}, {BNMF_ALPHA_P, BNMF_V_ALPHA_P, 0, BNMF_HC_P, BNMF_HD_P},
             {BNMF_BETA_E, BNMF_V_BETA_E, 1, BNMF_HA_E, BNMF_HB_E}, {BNMF_ALPHA_E, BNMF_V_ALPHA_E, 1, BNMF_HC_E, BNMF_HD_E}};
  } else if (c.prior == BNMF_EXPONENTIAL) {
    if (int rc = need_hyper(h, {BNMF_HA_P, BNMF_HB_P, BNMF_HA_E, BNMF_HB_E})) return rc;
    specs = {{BNMF_LAMBDA_P, BNMF_V_LAMBDA_P, 0, BNMF_HA_P, BNMF_HB_P}, {BNMF_LAMBDA_E, BNMF_V_LAMBDA_E, 1, BNMF_HA_E, BNMF_HB_E}};
  }
  for (const Spec& sp : specs) {
    Arr& a = h->arr[sp.id];
    std::vector<int> redraw(N, 1);
    if (a.set) redraw = a.redraw.empty() ? std::vector<int>(N, 0) : a.redraw;
    if (int rc = ensure(h, sp.id)) return rc;
    refresh_dev(h);
    HIPCHK(hipMemcpyAsync(h->dRedraw, redraw.data(), N * sizeof(int), hipMemcpyHostToDevice, h->stream));
    const long len = sp.side ? NG : KN;
    const HRef hs{h->arr[sp.hs].d, h->arr[sp.hs].stride}, hr{h->arr[sp.hr].d, h->arr[sp.hr].stride};
    double* slot1 = a.d + (size_t)len;                      // iteration 1 lives in slot 1
    if (sp.side) hipLaunchKernelGGL(k_init_gamma<1>, dim3((len + 255) / 256), dim3(256), 0, h->stream, h->dev, slot1, hs, hr, sp.var, h->dRedraw);
    else hipLaunchKernelGGL(k_init_gamma<0>, dim3((len + 255) / 256), dim3(256), 0, h->stream, h->dev, slot1, hs, hr, sp.var, h->dRedraw);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));   // redraw is reused by the next spec
  }
  if (c.prior == BNMF_TRUNCNORMAL) {
    if (int rc = need_hyper(h, {BNMF_HM_P, BNMF_HS_P, BNMF_HA_P, BNMF_HB_P, BNMF_HM_E, BNMF_HS_E, BNMF_HA_E, BNMF_HB_E})) return rc;
    struct TS { int id; uint32_t var; int side, is_mu; };
    const TS ts[4] = {{BNMF_MU_P, BNMF_V_MU_P, 0, 1}, {BNMF_SIGMASQ_P, BNMF_V_SIGSQ_P, 0, 0}, {BNMF_MU_E, BNMF_V_MU_E, 1, 1}, {BNMF_SIGMASQ_E, BNMF_V_SIGSQ_E, 1, 0}};
    for (const TS& sp : ts) {
      Arr& a = h->arr[sp.id];
      std::vector<int> redraw(N, 1);
      if (a.set) redraw = a.redraw.empty() ? std::vector<int>(N, 0) : a.redraw;
      if (int rc = ensure(h, sp.id)) return rc;
      refresh_dev(h);
      HIPCHK(hipMemcpyAsync(h->dRedraw, redraw.data(), N * sizeof(int), hipMemcpyHostToDevice, h->stream));
      const long len = sp.side ? NG : KN;
      double* slot1 = a.d + (size_t)len;
      if (sp.side) hipLaunchKernelGGL(k_init_tn<1>, dim3((len + 255) / 256), dim3(256), 0, h->stream, h->dev, slot1, sp.is_mu, sp.var, h->dRedraw);
      else hipLaunchKernelGGL(k_init_tn<0>, dim3((len + 255) / 256), dim3(256), 0, h->stream, h->dev, slot1, sp.is_mu, sp.var, h->dRedraw);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  }
  const bool haveP = h->arr[BNMF_P].set, haveE = h->arr[BNMF_E].set, haveA = h->arr[BNMF_A].set;
  if (int rc = ensure(h, BNMF_P)) return rc;
  if (int rc = ensure(h, BNMF_E)) return rc;
  if (int rc = ensure(h, BNMF_A)) return rc;
  if (c.MH) { if (int rc = ensure(h, BNMF_ACC_P)) return rc; if (int rc = ensure(h, BNMF_ACC_E)) return rc; }
  if (c.likelihood == BNMF_NORMAL) {
    const double three = 3.0;                                   // alpha = beta = 3 (R/bayesNMF_sampler.R:222-230)
    if (!h->arr[BNMF_ALPHA].d) if (int rc = bnmf_set_array(h, BNMF_ALPHA, &three, 1)) return rc;
    if (!h->arr[BNMF_BETA].d) if (int rc = bnmf_set_array(h, BNMF_BETA, &three, 1)) return rc;
    if (int rc = ensure(h, BNMF_SIGMASQ)) return rc;
  }
  if (!haveA) { std::vector<double> ones(N, 1.0); HIPCHK(hipMemcpy(h->arr[BNMF_A].d, ones.data(), N * sizeof(double), hipMemcpyHostToDevice)); }
  refresh_dev(h);
  h->iter = 1;
  Timer tm{h, false};
  use_slot(h, 1u);
  if (int rc = ensure_rings(h)) return rc;
  launch_pdraw(h, 1u, haveP ? 2 : 1, false);                // skip = names(init_params): supplied P / E kept verbatim
  launch_edraw(h, 1u, haveE ? 2 : 1, false);                // (iteration 1 is recorded by k_record below, every model)
  launch_side(h, 2u, tm);
  if (!haveA && c.learning_rank) {                            // R ~ Uniform{0..N}, A[n] ~ Bernoulli(pi(R))
    hipLaunchKernelGGL(k_rank_R, dim3(1), dim3(64), (size_t)(N + 1) * sizeof(double), h->stream, h->dev, 1u, 1);
    hipLaunchKernelGGL(k_rank_Aprior, dim3((N + 63) / 64), dim3(64), 0, h->stream, h->dev, 1u);
  }
  if (c.MH || c.likelihood == BNMF_NORMAL) launch_mh_metrics(h, 1u, true);
  else { if (int rc = launch_zalloc(h, 1u)) return rc; if (h->z_sort) { h->ct_pending = 1u; flush_colterms(h); } record_Z(h, 1u); }
  if (int rc = launch_record(h, 1u)) return rc;
  launch_reduce(h, 1u, 0, tm);
  flush_reduce(h, tm);
  hipEventRecord(h->ev_z, h->side); hipStreamWaitEvent(h->stream, h->ev_z, 0);
  hipLaunchKernelGGL(k_compose, dim3(1), dim3(64), 0, h->stream, h->dev, 1, 1u);
  HIPCHK(hipGetLastError());
  double row1[BNMF_NMETRIC];
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(row1, h->hMetrics, BNMF_NMETRIC * sizeof(double));
  HIPCHK(hipStreamSynchronize(h->side));
  HIPCHK(hipStreamSynchronize(h->side2));
  if (metrics_row1) memcpy(metrics_row1, row1, sizeof row1);
  if (h->wcap > 0) { h->hist.assign((size_t)h->wcap * 4, std::nan("")); h->hist[0] = row1[3]; h->hist[1] = row1[4]; h->hist[2] = row1[9]; h->hist[3] = row1[10]; }
  h->inited = true;
  return 0;
}

// Two handles that learn the rank must not run on one device at the same time: each persistent rank sweep sizes its grid
// as if it owned the device (one workgroup per CU, every workgroup waits for all others), and two half-resident grids
// would wait for each other until their bounded spins give up.  Their calls take turns (a call is at most one block of
// iterations between MAP checks).  Nor may a rank-learning call run beside ANY other chain's call on the device (round 4, found
// by tools/concurrent_check.py with full-size chains): a workgroup that waits inside a kernel for its chain's side streams (the
// allocation kernel's gate, the draw kernels' polls) holds a CU the rank sweep's grid needs, while the rank sweep's resident
// workgroups hold the registers the side-stream kernel needs — a cycle only the time-outs broke.  Rank-learning calls take the
// device's lock exclusively, all other calls shared.
static int flock_retry(int fd, int op) { int rc; while ((rc = flock(fd, op)) != 0 && errno == EINTR) {} return rc; }
static int run_impl(bnmf_handle* h, int n_iter, int converged, double* metrics, Timer& tm) {
  if (!h) return fail(BNMF_EINVAL, "bnmf_run: null handle");
  const bool excl = h->cfg.learning_rank != 0;
  // The rule between the PROCESSES that share the device first (each learns nothing of the others' launches), then the one between the
  // chains of this process — a call blocked on another process must not hold this process's gate.  Without the lock files a
  // rank-learning call is refused (two such chains of two processes end in each other's time-outs) unless the caller has said
  // BNMF_DEVLOCK=0: no other process uses the device.
  struct FileTurn { int fd; ~FileTurn() { if (fd >= 0) flock(fd, LOCK_UN); } } fturn{-1};
  if (!h->devlock_off) {
    if (h->devlock_fd < 0 || h->devgate_fd < 0) {
      if (excl) return fail(BNMF_ESTATE, "bnmf_run: a rank-learning chain needs its device to itself, and the device's lock files could not be opened "
                                         "(see the warning at bnmf_create): set BNMF_LOCKDIR, or BNMF_DEVLOCK=0 if no other process uses this GPU");
    } else {
      int rc;
      if (excl) { rc = flock_retry(h->devgate_fd, LOCK_EX); if (!rc) { rc = flock_retry(h->devlock_fd, LOCK_EX); flock(h->devgate_fd, LOCK_UN); } }
      else { rc = flock_retry(h->devgate_fd, LOCK_SH); if (!rc) { flock(h->devgate_fd, LOCK_UN); rc = flock_retry(h->devlock_fd, LOCK_SH); } }
      if (rc) return fail(BNMF_ESTATE, "bnmf_run: the device's lock file could not be taken (%s)", strerror(errno));
      fturn.fd = h->devlock_fd;
    }
  }
  struct GateTurn { DeviceGate* g; bool ex; ~GateTurn() { if (g) { if (ex) g->unlock(); else g->unlock_shared(); } } } gturn{nullptr, excl};
  if (h->device >= 0 && h->device < 64) {
    gturn.g = &g_dev_gate[h->device];
    if (excl) gturn.g->lock(); else gturn.g->lock_shared();
  }
  if (!h->inited) return fail(BNMF_ESTATE, "bnmf_run: call bnmf_init first");
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_run: an earlier call timed out inside a kernel; the handle's state is invalid, destroy it");
  if (n_iter < 0) return fail(BNMF_EINVAL, "bnmf_run: n_iter < 0");
  if (n_iter == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  if (int rc = ensure_metrics(h, (size_t)n_iter)) return rc;
  const uint32_t t0 = (uint32_t)h->iter + 1;
  // BNMF_RUNCLOCK=1 (diagnostics): host time of the call's phases on stderr
  static const bool runclock = getenv("BNMF_RUNCLOCK") != nullptr;
  const auto rc0 = std::chrono::steady_clock::now();
  auto rc_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - rc0).count(); };
  double rc_first = 0.0, rc_issued = 0.0, rc_tail = 0.0, rc_main = 0.0;
  for (int i = 0; i < n_iter; ++i) {
    if (runclock && i == 1) rc_first = rc_us();
    if (int rc = ((h->cfg.MH || h->cfg.likelihood == BNMF_NORMAL) ? sweep_mh(h, i, h->cfg.MH ? converged : 0, tm) : sweep(h, i, tm))) return rc;
    HIPCHK(hipGetLastError());                               // a refused launch of this iteration (bad geometry, LDS size)
    // a bounded in-kernel wait that timed out has set its word (mapped host memory): issue nothing more, so that one stuck
    // hand-off costs one spin bound and not one per remaining launch
    if (((volatile int*)h->hErr)[0] | ((volatile int*)h->hErr)[1]) break;
  }
  if (runclock) rc_issued = rc_us();
  const bool reduces_on_side2 = h->red_on_side2;          // fixed-rank sweep: every earlier k_reduce sits on side2, which flush_reduce's wait covers
  flush_colterms(h);                                       // the last iteration's column terms: no draw kernel behind it in this call
  if (int rc = flush_mh_etail(h)) return rc;              // (hosted MH sweep) the last iteration's E side: no row sweep behind it in this call
  // Round 5: the main stream waits for EVERYTHING issued on the two side streams (a fresh event each) in front of the last reduction:
  // when it is idle so are they, and the two host-side synchronisations of idle streams that stood below (6 us each, at the end of
  // every call) are gone.  (flush_reduce's own wait for side2 is then a wait for an event that has fired.)
  hipEventRecord(h->ev_z, h->side); hipStreamWaitEvent(h->stream, h->ev_z, 0);
  hipEventRecord(h->ev_sideP, h->side2); hipStreamWaitEvent(h->stream, h->ev_sideP, 0);
  if (h->side_ev_stale) { hipStreamWaitEvent(h->side, h->ev_sideP, 0); hipEventRecord(h->ev_side, h->side); h->side_ev_stale = false; }   // (what refresh_side_events would record)
  flush_reduce(h, tm);
  (void)reduces_on_side2;
  hipLaunchKernelGGL(k_compose, dim3((n_iter + 63) / 64), dim3(64), 0, h->stream, h->dev, n_iter, t0);
  HIPCHK(hipGetLastError());
  std::vector<double> own;
  if (!metrics && h->wcap > 0) { own.resize((size_t)n_iter * BNMF_NMETRIC); metrics = own.data(); }
  if (runclock) rc_tail = rc_us();
  HIPCHK(hipStreamSynchronize(h->stream));
  if (runclock) rc_main = rc_us();
  if (metrics) memcpy(metrics, h->hMetrics, (size_t)n_iter * BNMF_NMETRIC * sizeof(double));

  if (runclock) fprintf(stderr, "[bnmf_run %d] first iteration issued %.1f us, all issued %.1f, tail issued %.1f, main stream idle %.1f, side streams idle %.1f\n",
                        n_iter, rc_first, rc_issued, rc_tail, rc_main, rc_us());
  if (h->wcap > 0 && metrics) {                             // loglik / logpost of the recorded iterations (MAP metrics are window means)
    if (h->hist.size() != (size_t)h->wcap * 4) h->hist.assign((size_t)h->wcap * 4, std::nan(""));
    for (int i = 0; i < n_iter; ++i) {
      const double* r = metrics + (size_t)i * BNMF_NMETRIC;
      double* d = h->hist.data() + (size_t)((t0 + i - 1) % (uint32_t)h->wcap) * 4;
      d[0] = r[3]; d[1] = r[4]; d[2] = r[9]; d[3] = r[10];
    }
  }
  // all three streams are idle: the time-out words (mapped host memory) are final
  if (((volatile int*)h->hErr)[0] | ((volatile int*)h->hErr)[1]) {
    // the kernels behind the time-out ran on inputs that were never published: P, E, the rings and the metric rows of this call
    // are not the chain's.  The handle stays poisoned (every later call fails with BNMF_ESTATE) until it is destroyed.
    h->poisoned = true;
    unsigned fl[16] = {};
    hipMemcpy(fl, h->dFlags, sizeof fl, hipMemcpyDeviceToHost);
    if (((volatile int*)h->hErr)[0])
      return fail(BNMF_EHIP, "bnmf_run: a kernel timed out waiting for the hyper-parameter sweep of its iteration (iteration %d; flags E-side %u, Esum %u, P-side %u, draw %u; "
                  "serialised dispatch? set BNMF_SERIAL=1); the handle is now invalid", h->iter, fl[1], fl[3], fl[9], fl[6]);
    return fail(BNMF_EHIP, "bnmf_run: the grid barrier of the rank sweep timed out at iteration %d (workgroups not co-resident?); the handle is now invalid", h->iter);
  }
  return 0;
}
int bnmf_run(bnmf_handle* h, int n_iter, int converged, double* metrics) {
  Timer tm{h, false};
  return run_impl(h, n_iter, converged, metrics, tm);
}
int bnmf_profile(bnmf_handle* h, int n_iter, int converged, double* out_ms) {
  if (!h || !out_ms) return fail(BNMF_EINVAL, "bnmf_profile: null argument");
  HIPCHK(hipSetDevice(h->device));
  if (!h->have_ev) { for (auto& e : h->ev) HIPCHK(hipEventCreate(&e)); h->have_ev = true; }
  Timer tm{h, true};
  if (int rc = run_impl(h, n_iter, converged, nullptr, tm)) return rc;
  for (int k = 0; k < BNMF_NKERNEL; ++k) out_ms[k] = tm.cnt[k] ? tm.acc[k] / tm.cnt[k] : 0.0;
  return 0;
}
int bnmf_window(bnmf_handle* h, int id, int last_n, double* out) {
  if (!h || !out) return fail(BNMF_EINVAL, "bnmf_window: null argument");
  const int W = h->cfg.window;
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_window: the handle timed out inside a kernel; its state is invalid");
  if (W <= 0) return fail(BNMF_ESTATE, "bnmf_window: the handle was created with window = 0");
  if (last_n < 1 || last_n > W || last_n > h->iter) return fail(BNMF_ESIZE, "bnmf_window: last_n = %d but only min(window = %d, iter = %d) samples are kept", last_n, W, h->iter);
  if (id < 0 || id >= BNMF_ID_MAX) return fail(BNMF_EINVAL, "bnmf_window: unknown id %d", id);
  if (id == BNMF_Z) {
    if (!h->zring && !h->dZsRecRing) return fail(BNMF_EUNSET, "bnmf_window: Z is not kept per sample (needs save_Z, a window, and the window's samples within BNMF_ZRING_GB)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t lenz = id_len(h, BNMF_Z);
    std::vector<int32_t> tmp(lenz);
    for (int i = 0; i < last_n; ++i) {
      const size_t slot = (size_t)(h->iter - last_n + i) % (size_t)h->wcap;
      if (h->dZsRecRing) {                                   // sorted schedule: the sample is its item records; expand them (k_zexpand) and copy
        launch_zexpand(h, (uint32_t)(h->iter - last_n + i + 1));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(tmp.data(), h->dZ, lenz * sizeof(int32_t), hipMemcpyDeviceToHost));
      } else
      HIPCHK(hipMemcpy(tmp.data(), h->zring + slot * lenz, lenz * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (size_t j = 0; j < lenz; ++j) out[(size_t)i * lenz + j] = tmp[j];
    }
    return 0;
  }
  const Arr& a = h->arr[id];
  const size_t len = id_len(h, id);
  if (!a.ring || len == 0) return fail(BNMF_EUNSET, "bnmf_window: id %d is not recorded for this model", id);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  // sample `it` lives in slot (it-1) % (W+1): the last_n samples, oldest first, are at most two contiguous runs of the ring
  const size_t C = (size_t)h->wcap, s0 = (size_t)(h->iter - last_n) % C;
  const size_t n1 = (s0 + (size_t)last_n <= C) ? (size_t)last_n : C - s0;
  HIPCHK(hipMemcpy(out, a.ring + s0 * len, n1 * len * sizeof(double), hipMemcpyDeviceToHost));
  if (n1 < (size_t)last_n) HIPCHK(hipMemcpy(out + n1 * len, a.ring, ((size_t)last_n - n1) * len * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}


// copy `n` consecutive samples (oldest first) of a ring to the host: at most two contiguous runs
static int ring_read(const bnmf_handle* h, int id, int last_n, double* out) {
  const Arr& a = h->arr[id];
  const size_t len = id_len(h, id), C = (size_t)h->wcap, s0 = (size_t)(h->iter - last_n) % C;
  const size_t n1 = (s0 + (size_t)last_n <= C) ? (size_t)last_n : C - s0;
  HIPCHK(hipMemcpy(out, a.ring + s0 * len, n1 * len * sizeof(double), hipMemcpyDeviceToHost));
  if (n1 < (size_t)last_n) HIPCHK(hipMemcpy(out + n1 * len, a.ring, ((size_t)last_n - n1) * len * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int bnmf_map(bnmf_handle* h, int last_n, double ci, double* P_mean, double* E_mean, double* A_mode, double* top_A,
             double* P_lower, double* P_upper, double* E_lower, double* E_upper, int32_t* used, bnmf_map_info* info) {
  if (!h || !A_mode || !info) return fail(BNMF_EINVAL, "bnmf_map: null argument");
  const int W = h->cfg.window;
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_map: the handle timed out inside a kernel; its state is invalid");
  if (W <= 0) return fail(BNMF_ESTATE, "bnmf_map: the handle was created with window = 0");
  if (last_n < 1 || last_n > W || last_n > h->iter) return fail(BNMF_ESIZE, "bnmf_map: last_n = %d but only min(window = %d, iter = %d) samples are kept", last_n, W, h->iter);
  if (ci >= 1.0) return fail(BNMF_EINVAL, "bnmf_map: credible_interval must be below 1");
  const int K = h->cfg.K, N = h->cfg.N, G = h->cfg.G;
  const size_t lenP = (size_t)K * N, lenE = (size_t)N * G;
  if (!h->arr[BNMF_P].ring || !h->arr[BNMF_E].ring || !h->arr[BNMF_A].ring) return fail(BNMF_ESTATE, "bnmf_map: nothing recorded yet");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->side));
  HIPCHK(hipStreamSynchronize(h->side2));
  // i. mode of A (get_mode): patterns as strings, most frequent first, ties in alphabetical order
  std::vector<double> Aw((size_t)last_n * N);
  if (int rc = ring_read(h, BNMF_A, last_n, Aw.data())) return rc;
  std::vector<std::string> keys(last_n, std::string(N, '0'));
  std::map<std::string, int> tab;
  for (int s = 0; s < last_n; ++s) { for (int n = 0; n < N; ++n) if (Aw[(size_t)s * N + n] != 0.0) keys[s][n] = '1'; tab[keys[s]]++; }
  std::vector<std::pair<std::string, int>> ord(tab.begin(), tab.end());        // std::map iterates alphabetically
  std::stable_sort(ord.begin(), ord.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
  const std::string& mode = ord[0].first;
  info->n_patterns = (int)ord.size();
  for (int i = 0; i < 5; ++i) {
    info->top_counts[i] = i < (int)ord.size() ? ord[i].second : 0;
    if (top_A) for (int n = 0; n < N; ++n) top_A[(size_t)i * N + n] = i < (int)ord.size() ? (ord[i].first[n] == '1' ? 1.0 : 0.0) : std::nan("");
  }
  std::vector<int> slots;
  for (int s = 0; s < last_n; ++s) {
    const bool u = keys[s] == mode;
    if (used) used[s] = u ? 1 : 0;
    if (u) slots.push_back((int)((size_t)(h->iter - last_n + s) % (size_t)h->wcap));
  }
  const int nu = (int)slots.size();
  info->n_used = nu; info->_pad = 0;
  std::vector<double> Am(N);
  for (int n = 0; n < N; ++n) Am[n] = A_mode[n] = mode[n] == '1' ? 1.0 : 0.0;
  // ii-iii. renormalised means (and quantiles) on the device
  const bool want_ci = ci > 0.0 && (P_lower || P_upper || E_lower || E_upper);
  int kt = 0, jlo = 0, jhi = 0; double glo = 0.0, ghi = 0.0;
  if (want_ci) {                                    // quantile type 7: h = (n-1) p, j = floor(h), g = h - j
    const double plo = 0.5 - ci / 2.0, phi = 0.5 + ci / 2.0;
    const double hl = (nu - 1) * plo, hh = (nu - 1) * phi;
    jlo = (int)std::floor(hl); glo = hl - jlo; jhi = (int)std::floor(hh); ghi = hh - jhi;
    kt = std::min(nu, std::max(jlo + 2, nu - jhi));
  }
  // the bounds by sorting (k_map_quant) when the samples of 8 elements fit the LDS; else by the kt smallest / largest per lane
  int qS = 0;
  if (want_ci) { qS = ((nu + 63) / 64) * 64; if (qS > 2048) qS = 0; }
  if (want_ci && !qS && (size_t)kt * 2 * 64 * sizeof(double) > 160 * 1024)
    return fail(BNMF_EINVAL, "bnmf_map: credible_interval %.3g over %d samples needs %d order statistics per element (device limit 160): take the window with bnmf_window", ci, nu, kt);
  const size_t words = (size_t)nu * N + 3 * (lenP + lenE) + 2 * (size_t)G + N + ((size_t)nu + 1) / 2 + 8;
  if (words > h->map_words) { if (h->dMap) HIPCHK(dfree(h->dMap)); h->dMap = nullptr; HIPCHK(dmalloc(&h->dMap, words * sizeof(double))); h->map_words = words; }
  double* cs = h->dMap; double* mP = cs + (size_t)nu * N; double* loP = mP + lenP; double* hiP = loP + lenP;
  double* mE = hiP + lenP; double* loE = mE + lenE; double* hiE = loE + lenE;
  double* colsse = hiE + lenE; double* colkl = colsse + G; double* dA = colkl + G; int* dslots = (int*)(dA + N);
  HIPCHK(hipMemcpyAsync(dslots, slots.data(), nu * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dA, Am.data(), N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_map_colsum, dim3(nu, N), dim3(64), 0, h->stream, (const double*)h->arr[BNMF_P].ring, lenP, K, N, (const int*)dslots, cs);
  if (!qS) {                                              // means (and, beyond 2,048 samples, the bounds) by a lane per element
    const size_t lds = (size_t)kt * 2 * 64 * sizeof(double);
    if (lds > 64 * 1024) {
      HIPCHK(hipFuncSetAttribute((const void*)k_map_stats<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)k_map_stats<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(k_map_stats<0>, dim3((unsigned)((lenP + 63) / 64)), dim3(64), lds, h->stream, (const double*)h->arr[BNMF_P].ring, lenP, K, N,
                       (const int*)dslots, nu, (const double*)cs, kt, jlo, glo, jhi, ghi, mP, loP, hiP);
    hipLaunchKernelGGL(k_map_stats<1>, dim3((unsigned)((lenE + 63) / 64)), dim3(64), lds, h->stream, (const double*)h->arr[BNMF_E].ring, lenE, K, N,
                       (const int*)dslots, nu, (const double*)cs, kt, jlo, glo, jhi, ghi, mE, loE, hiE);
  }
  if (qS) {
    const bool r16 = qS <= 1024;                           // 16 or 32 samples per lane column
    const size_t qS2 = r16 ? 1024 : 2048, qlds = qS2 * MQ_E * sizeof(double) + qS2 * sizeof(int);
    auto go = [&](auto kern, const double* ring, size_t len, double* mn, double* lo, double* hi) {
      if (qlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds));
      hipLaunchKernelGGL(kern, dim3((unsigned)((len + MQ_E - 1) / MQ_E)), dim3(MQ_T), qlds, h->stream, ring, len, K, N, (const int*)dslots, nu,
                         (const double*)cs, jlo, glo, jhi, ghi, mn, lo, hi);
      return 0;
    };
    if (r16) { if (int rc = go(k_map_quant<0, 16>, h->arr[BNMF_P].ring, lenP, mP, loP, hiP)) return rc; if (int rc = go(k_map_quant<1, 16>, h->arr[BNMF_E].ring, lenE, mE, loE, hiE)) return rc; }
    else { if (int rc = go(k_map_quant<0, 32>, h->arr[BNMF_P].ring, lenP, mP, loP, hiP)) return rc; if (int rc = go(k_map_quant<1, 32>, h->arr[BNMF_E].ring, lenE, mE, loE, hiE)) return rc; }
  }
  hipLaunchKernelGGL(k_map_fit, dim3((G + 3) / 4), dim3(256), 0, h->stream, (const int32_t*)h->dM, (const double*)mP, (const double*)dA, (const double*)mE, K, N, G, colsse, colkl);
  HIPCHK(hipGetLastError());
  if (P_mean) HIPCHK(hipMemcpyAsync(P_mean, mP, lenP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (E_mean) HIPCHK(hipMemcpyAsync(E_mean, mE, lenE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (want_ci) {
    if (P_lower) HIPCHK(hipMemcpyAsync(P_lower, loP, lenP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (P_upper) HIPCHK(hipMemcpyAsync(P_upper, hiP, lenP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (E_lower) HIPCHK(hipMemcpyAsync(E_lower, loE, lenE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (E_upper) HIPCHK(hipMemcpyAsync(E_upper, hiE, lenE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  std::vector<double> col(2 * (size_t)G);
  HIPCHK(hipMemcpyAsync(col.data(), colsse, 2 * (size_t)G * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double sse = 0.0, kl = 0.0;
  for (int g = 0; g < G; ++g) { sse += col[g]; kl += col[(size_t)G + g]; }
  info->rmse = std::sqrt(sse / ((double)K * (double)G));
  info->kl = kl;
  return 0;
}

// One MAP check (R/bayesNMF_sampler.R:297-321): get_MAP_ over the last min(MAP_over, iter) samples on the device, the
// state$MAP_metrics row (update_MAP_metrics_, R/utils.R:356-397) and check_convergence_ (R/convergence.R:60-154) into mr.
static int map_check(bnmf_handle* h, const bnmf_convergence_control* cc, bnmf_convergence_state* st, const double* lastrow, double* mr,
                     std::vector<double>& Am, std::vector<int32_t>& used) {
  const int K = h->cfg.K, G = h->cfg.G, N = h->cfg.N;
  const int it = h->iter;
  const int win = it < cc->MAP_over ? it : cc->MAP_over;
  bnmf_map_info info;
  if (int rc = bnmf_map(h, win, 0.0, nullptr, nullptr, Am.data(), nullptr, nullptr, nullptr, nullptr, nullptr, used.data(), &info)) return rc;
  // update_MAP_metrics_: loglikelihood / logposterior are means of the per-sample values over the whole window
  double ll = 0.0, lp = 0.0, mt = 0.0, sumA = 0.0;
  for (int i = it - win + 1; i <= it; ++i) {
    const double* hrow = h->hist.data() + (size_t)((i - 1) % h->wcap) * 4;
    ll += hrow[0]; lp += hrow[1];
    mt += h->temp_host.empty() ? 1.0 : h->temp_host[std::min<size_t>((size_t)i - 1, h->temp_host.size() - 1)];
  }
  ll /= win; lp /= win; mt /= win;
  for (int j = 0; j < N; ++j) sumA += Am[j];
  const double n_params = sumA * (double)(G + K);
  mr[0] = it; mr[1] = info.rmse; mr[2] = info.kl; mr[3] = ll; mr[4] = lp; mr[5] = n_params;
  mr[6] = -2.0 * ll + n_params * std::log((double)G); mr[7] = sumA; mr[8] = info.top_counts[0]; mr[9] = mt;
  mr[10] = lastrow[9]; mr[11] = lastrow[10];            // compute_metrics_ uses the CURRENT acceptance matrices
  // check_convergence_
  static const int col_of[5] = {3, 4, 1, 2, 6};
  double m = mr[col_of[cc->metric]];
  if (cc->metric <= 1) m = -m;
  if (!st->have_prev) { st->prev_MAP_metric = m + 1.0; st->best_MAP_metric = m + 1.0; st->inarow_na = st->inarow_no_change = st->inarow_no_best = 0; st->have_prev = 1; }
  const double pc = (m - st->prev_MAP_metric) / st->prev_MAP_metric;
  st->prev_percent_change = pc; st->prev_MAP_metric = m;
  if (pc != pc) { st->inarow_no_change = 0; st->inarow_no_best = 0; st->inarow_na += 1; }
  else if (std::fabs(pc) < cc->tol) { st->inarow_no_change += 1; st->inarow_na = 0; }
  else { st->inarow_no_change = 0; st->inarow_na = 0; }
  bool temps_one = true;                                  // temperature_schedule[(iter - MAP_over):iter] == 1 (R drops index 0)
  for (int i = std::max(it - cc->MAP_over, 1); i <= it && temps_one; ++i)
    temps_one = h->temp_host.empty() || h->temp_host[std::min<size_t>((size_t)i - 1, h->temp_host.size() - 1)] == 1.0;
  if (temps_one && it >= cc->miniters) {
    if (m < st->best_MAP_metric) { st->best_MAP_metric = m; st->best_iter = it; st->inarow_no_best = 0; }
    else st->inarow_no_best += 1;
    if (st->inarow_no_change >= cc->Ninarow_nochange) { st->converged = 1; st->why = 1; }
    else if (st->inarow_no_best >= cc->Ninarow_nobest) { st->converged = 1; st->why = 2; }
    else if (it >= cc->maxiters) { st->converged = 1; st->why = 3; }
  }
  mr[12] = pc; mr[13] = st->inarow_no_change; mr[14] = st->inarow_no_best; mr[15] = st->inarow_na; mr[16] = st->converged;
  return 0;
}

// The sampling loop up to convergence as ONE call: blocks of iterations up to the next MAP check, get_MAP_ on the device,
// update_MAP_metrics_ (R/utils.R:356-397) and check_convergence_ (R/convergence.R:60-154) here on the host side of the ABI.
int bnmf_run_until(bnmf_handle* h, const bnmf_convergence_control* cc, bnmf_convergence_state* st, double* metrics, int cap_rows,
                   int* n_rows, double* map_rows, int cap_checks, int* n_checks) {
  if (!h || !cc || !st || !metrics || !n_rows || !map_rows || !n_checks) return fail(BNMF_EINVAL, "bnmf_run_until: null argument");
  if (!h->inited) return fail(BNMF_ESTATE, "bnmf_run_until: call bnmf_init first");
  if (h->cfg.window < cc->MAP_over) return fail(BNMF_ESTATE, "bnmf_run_until: the handle keeps %d samples, MAP_over = %d", h->cfg.window, cc->MAP_over);
  if (cc->MAP_every < 1 || cc->MAP_over < 1 || cc->metric < 0 || cc->metric > 4) return fail(BNMF_EINVAL, "bnmf_run_until: bad convergence control");
  const int N = h->cfg.N;
  *n_rows = 0; *n_checks = 0;
  std::vector<double> Am(N);
  std::vector<int32_t> used(cc->MAP_over);
  Timer tm{h, false};
  while (!st->converged && h->iter < cc->maxiters) {      // R/bayesNMF_sampler.R:268
    const int it0 = h->iter;
    int nxt = (it0 / cc->MAP_every + 1) * cc->MAP_every;
    if (nxt > cc->maxiters) nxt = cc->maxiters;
    const int n = nxt - it0;
    if (*n_rows + n > cap_rows) return fail(BNMF_ESIZE, "bnmf_run_until: metrics buffer too small (%d rows)", cap_rows);
    if (int rc = run_impl(h, n, 0, metrics + (size_t)*n_rows * BNMF_NMETRIC, tm)) return rc;
    const double* lastrow = metrics + (size_t)(*n_rows + n - 1) * BNMF_NMETRIC;
    *n_rows += n;
    const int it = h->iter;
    const int over = cc->MAP_over > cc->MAP_every ? cc->MAP_over : cc->MAP_every;
    if (!((it % cc->MAP_every == 0 && it >= over) || it >= cc->maxiters)) continue;      // :288-296
    if (*n_checks >= cap_checks) return fail(BNMF_ESIZE, "bnmf_run_until: MAP-metrics buffer too small (%d rows)", cap_checks);
    if (int rc = map_check(h, cc, st, lastrow, map_rows + (size_t)*n_checks * BNMF_NMAPROW, Am, used)) return rc;
    *n_checks += 1; st->n_checks += 1;
  }
  return 0;
}

// The post-warm-up tail of the MH models as ONE call (R/bayesNMF_sampler.R:332-384): post_warmup more iterations with
// state$converged = TRUE (true accept / reject), a MAP check whenever iter is a multiple of MAP_every and after the last one.
int bnmf_run_post_warmup(bnmf_handle* h, const bnmf_convergence_control* cc, bnmf_convergence_state* st, int post_warmup,
                         double* metrics, int cap_rows, int* n_rows, double* map_rows, int cap_checks, int* n_checks) {
  if (!h || !cc || !st || !metrics || !n_rows || !map_rows || !n_checks) return fail(BNMF_EINVAL, "bnmf_run_post_warmup: null argument");
  if (!h->inited) return fail(BNMF_ESTATE, "bnmf_run_post_warmup: call bnmf_init first");
  if (post_warmup < 0) return fail(BNMF_EINVAL, "bnmf_run_post_warmup: post_warmup < 0");
  if (h->cfg.window < cc->MAP_over) return fail(BNMF_ESTATE, "bnmf_run_post_warmup: the handle keeps %d samples, MAP_over = %d", h->cfg.window, cc->MAP_over);
  if (cc->MAP_every < 1 || cc->MAP_over < 1 || cc->metric < 0 || cc->metric > 4) return fail(BNMF_EINVAL, "bnmf_run_post_warmup: bad convergence control");
  *n_rows = 0; *n_checks = 0;
  std::vector<double> Am(h->cfg.N);
  std::vector<int32_t> used(cc->MAP_over);
  Timer tm{h, false};
  int done = 0;
  while (done < post_warmup) {
    const int it0 = h->iter;
    const int nxt = (it0 / cc->MAP_every + 1) * cc->MAP_every;
    const int n = std::min(nxt - it0, post_warmup - done);
    if (*n_rows + n > cap_rows) return fail(BNMF_ESIZE, "bnmf_run_post_warmup: metrics buffer too small (%d rows)", cap_rows);
    if (int rc = run_impl(h, n, 1, metrics + (size_t)*n_rows * BNMF_NMETRIC, tm)) return rc;
    const double* lastrow = metrics + (size_t)(*n_rows + n - 1) * BNMF_NMETRIC;
    *n_rows += n; done += n;
    if (!(h->iter % cc->MAP_every == 0 || done == post_warmup)) continue;
    if (*n_checks >= cap_checks) return fail(BNMF_ESIZE, "bnmf_run_post_warmup: MAP-metrics buffer too small (%d rows)", cap_checks);
    if (int rc = map_check(h, cc, st, lastrow, map_rows + (size_t)*n_checks * BNMF_NMAPROW, Am, used)) return rc;
    *n_checks += 1; st->n_checks += 1;
  }
  return 0;
}

static double quantile7(std::vector<double> x, double prob) {
  std::sort(x.begin(), x.end());
  const double hq = (x.size() - 1) * prob;
  const size_t j = (size_t)std::floor(hq);
  const double g = hq - (double)j;
  return (1.0 - g) * x[j] + g * x[std::min(j + 1, x.size() - 1)];
}

int bnmf_assign(bnmf_handle* h, int last_n, const int32_t* used, const double* ref, int R, const int32_t* keep, const double* MAP_P,
                double ci, double* votes, int32_t* assigned, double* MAP_cosine, double* lower, double* upper) {
  if (!h || !ref || !votes || !assigned) return fail(BNMF_EINVAL, "bnmf_assign: null argument");
  const int W = h->cfg.window;
  if (h->poisoned) return fail(BNMF_ESTATE, "bnmf_assign: the handle timed out inside a kernel; its state is invalid");
  if (W <= 0 || !h->arr[BNMF_P].ring) return fail(BNMF_ESTATE, "bnmf_assign: no recorded samples (window = 0)");
  if (last_n < 1 || last_n > W || last_n > h->iter) return fail(BNMF_ESIZE, "bnmf_assign: last_n = %d but only min(window = %d, iter = %d) samples are kept", last_n, W, h->iter);
  if (R < 1) return fail(BNMF_EINVAL, "bnmf_assign: empty reference");
  const int K = h->cfg.K, N = h->cfg.N;
  std::vector<int> slots, sig;
  for (int s = 0; s < last_n; ++s) if (!used || used[s]) slots.push_back((int)((size_t)(h->iter - last_n + s) % (size_t)h->wcap));
  for (int n = 0; n < N; ++n) if (!keep || keep[n]) sig.push_back(n);
  const int nu = (int)slots.size(), nk = (int)sig.size();
  for (int i = 0; i < N * R; ++i) votes[i] = 0.0;
  for (int n = 0; n < N; ++n) { assigned[n] = -1; if (MAP_cosine) MAP_cosine[n] = std::nan(""); if (lower) lower[n] = std::nan(""); if (upper) upper[n] = std::nan(""); }
  if (nu == 0 || nk == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<double> refT((size_t)K * R), rn2(R, 0.0);
  for (int j = 0; j < R; ++j) for (int k = 0; k < K; ++k) { const double v = ref[k + (size_t)K * j]; refT[(size_t)k * R + j] = v; rn2[j] += v * v; }
  const size_t nout = (size_t)nu * nk * R;
  // one scratch allocation per handle, grown on demand (the ensemble assignment is called once per result, but BIC sweeps call it per rank)
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t oRef = 0, oN2 = oRef + up(refT.size() * 8), oOut = oN2 + up((size_t)R * 8), oSl = oOut + up(nout * 8), oSig = oSl + up((size_t)nu * sizeof(int)),
               oCol = oSig + up((size_t)nk * sizeof(int)), need = oCol + up((size_t)nu * std::min(nk, R) * sizeof(int32_t));
  if (need > h->asg_bytes) {
    if (h->dAsg) { HIPCHK(dfree(h->dAsg)); h->dAsg = nullptr; h->asg_bytes = 0; }
    HIPCHK(dmalloc(&h->dAsg, need));
    h->asg_bytes = need;
  }
  double *dRef = (double*)(h->dAsg + oRef), *dN2 = (double*)(h->dAsg + oN2), *dOut = (double*)(h->dAsg + oOut);
  int *dSl = (int*)(h->dAsg + oSl), *dSig = (int*)(h->dAsg + oSig);
  int32_t* dCol = (int32_t*)(h->dAsg + oCol);
  HIPCHK(hipMemcpy(dRef, refT.data(), refT.size() * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dN2, rn2.data(), R * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dSl, slots.data(), nu * sizeof(int), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dSig, sig.data(), nk * sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ref_cosine, dim3(nu, nk), dim3(128), 0, h->stream, (const double*)h->arr[BNMF_P].ring, (size_t)K * N, K, (const int*)dSl,
                     (const int*)dSig, nk, (const double*)dRef, (const double*)dN2, R, dOut);
  HIPCHK(hipGetLastError());
  // one Hungarian assignment per sample (maximise the total cosine), one wave each; with more signatures than references the
  // references are the rows
  const bool tr = nk > R;
  const int nrow = tr ? R : nk, ncol = tr ? nk : R;
  const size_t hung_lds = (size_t)(ncol + 1) * (16 + 12) + (size_t)(nrow + 1) * 8;
  if (hung_lds > 160 * 1024) return fail(BNMF_ESIZE, "bnmf_assign: %d x %d assignment problem exceeds the LDS of one workgroup", nrow, ncol);
  HIPCHK(hipFuncSetAttribute((const void*)k_hungarian, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hung_lds));
  HIPCHK(hipMemsetAsync(dCol, 0xff, (size_t)nu * nrow * sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(k_hungarian, dim3(nu), dim3(64), hung_lds, h->stream, (const double*)dOut, nk, R, tr ? 1 : 0, dCol);
  HIPCHK(hipGetLastError());
  std::vector<double> cosv(nout);
  std::vector<int32_t> col((size_t)nu * nrow);
  HIPCHK(hipMemcpyAsync(cosv.data(), dOut, nout * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(col.data(), dCol, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  // the cosine of a chosen pair is its vote; the votes are summed in sample order
  for (int s = 0; s < nu; ++s) {
    const double* c = cosv.data() + (size_t)s * nk * R;
    const int32_t* a = col.data() + (size_t)s * nrow;
    for (int r = 0; r < nrow; ++r) if (a[r] < 0 || a[r] >= ncol) return fail(BNMF_ESTATE, "bnmf_assign: sample %d has no assignment (a cosine is not finite)", s);
    if (!tr) for (int i = 0; i < nk; ++i) votes[sig[i] + (size_t)N * a[i]] += c[(size_t)i * R + a[i]];
    else for (int j = 0; j < R; ++j) votes[sig[a[j]] + (size_t)N * j] += c[(size_t)a[j] * R + j];
  }
  for (int i = 0; i < nk; ++i) {                           // which.max(prop_votes): first maximum
    const int n = sig[i];
    int best = -1; double bv = 0.0;
    for (int j = 0; j < R; ++j) if (votes[n + (size_t)N * j] > bv) { bv = votes[n + (size_t)N * j]; best = j; }
    assigned[n] = best;
    if (best < 0) continue;
    if (MAP_P && MAP_cosine) {
      double dot = 0.0, nn = 0.0;
      for (int k = 0; k < K; ++k) { const double p = MAP_P[k + (size_t)K * n]; dot += p * ref[k + (size_t)K * best]; nn += p * p; }
      MAP_cosine[n] = dot / std::sqrt(nn * rn2[best]);
    }
    if (ci > 0.0 && ci < 1.0 && (lower || upper)) {
      std::vector<double> x(nu);
      for (int s = 0; s < nu; ++s) x[s] = cosv[((size_t)s * nk + i) * R + best];
      if (lower) lower[n] = quantile7(x, (1.0 - ci) / 2.0);
      if (upper) upper[n] = quantile7(x, 1.0 - (1.0 - ci) / 2.0);
    }
  }
  return 0;
}

// ---- measured ceilings for bench.py's roofline: Philox4x32-7 words per second (the count-allocation stream's generator) with
// nothing else in the loop (the floor of any allocation kernel: one word per count), and the device-to-device copy bandwidth ----
__global__ __launch_bounds__(256) void k_ubench_philox(uint32_t* out, int blocks_per_lane, uint32_t seed) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int q = 0; q < blocks_per_lane; ++q) {
    const u32x4 w = philox4x32_7((uint32_t)q, gid, seed, 5u, 17u, 29u);
    acc += w.x ^ w.y ^ w.z ^ w.w;
  }
  out[gid] = acc;
}
int bnmf_ubench(int device, double* philox_words_per_s, double* copy_gbs) {
  if (!philox_words_per_s || !copy_gbs) return fail(BNMF_EINVAL, "bnmf_ubench: null argument");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  {
    const int wgs = prop.multiProcessorCount * 8, nq = 4000;                 // 8 waves per SIMD
    uint32_t* d = nullptr;
    HIPCHK(dmalloc(&d, (size_t)wgs * 256 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_ubench_philox, dim3(wgs), dim3(256), 0, 0, d, 200, 1u);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_ubench_philox, dim3(wgs), dim3(256), 0, 0, d, nq, 1u);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *philox_words_per_s = 4.0 * nq * (double)wgs * 256.0 / (ms * 1e-3);
    dfree(d);
  }
  {
    const size_t bytes = (size_t)1 << 30;
    char *a = nullptr, *b = nullptr;
    HIPCHK(dmalloc(&a, bytes)); HIPCHK(dmalloc(&b, bytes));
    HIPCHK(hipMemset(a, 1, bytes));
    HIPCHK(hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 4; ++r) HIPCHK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *copy_gbs = 4.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;               // read + write
    dfree(a); dfree(b);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

// ---- device-side probes for the parity tests ----
int bnmf_test_math(int device, int fn, const double* in, double* out, size_t n) {
  HIPCHK(hipSetDevice(device));
  double *di, *dou;
  HIPCHK(dmalloc(&di, n * sizeof(double)));
  HIPCHK(dmalloc(&dou, n * sizeof(double)));
  HIPCHK(hipMemcpy(di, in, n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_math, dim3((n + 255) / 256), dim3(256), 0, 0, fn, di, dou, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dou, n * sizeof(double), hipMemcpyDeviceToHost));
  dfree(di); dfree(dou);
  return 0;
}
int bnmf_test_sampler(int device, int which, uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter,
                      const double* a, const double* b, const double* c, double* out, size_t n) {
  HIPCHK(hipSetDevice(device));
  if (int rc = ensure_alut(device)) return rc;
  double *da, *db, *dc, *dou;
  HIPCHK(dmalloc(&da, n * sizeof(double))); HIPCHK(dmalloc(&db, n * sizeof(double)));
  HIPCHK(dmalloc(&dc, n * sizeof(double))); HIPCHK(dmalloc(&dou, n * sizeof(double)));
  std::vector<double> zeros(n, 0.0);
  HIPCHK(hipMemcpy(da, a ? a : zeros.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db, b ? b : zeros.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dc, c ? c : zeros.data(), n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_sampler, dim3((n + 255) / 256), dim3(256), 0, 0, which, (uint32_t)seed, (uint32_t)(seed >> 32) ^ chain,
                     var, elem0, iter, da, db, dc, dou, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dou, n * sizeof(double), hipMemcpyDeviceToHost));
  dfree(da); dfree(db); dfree(dc); dfree(dou);
  return 0;
}
static int test_philox_r(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
  HIPCHK(hipSetDevice(device));
  uint32_t* d;
  HIPCHK(dmalloc(&d, 16));
  hipLaunchKernelGGL(k_test_philox, dim3(1), dim3(1), 0, 0, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], d, rounds);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
  dfree(d);
  return 0;
}
int bnmf_test_philox(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { return test_philox_r(device, ctr, key, out, 10); }
int bnmf_test_philox7(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { return test_philox_r(device, ctr, key, out, 7); }

}  // extern "C"
