/* include/bnmf.h — C ABI of libbnmf.so, the MI355X (gfx950) Gibbs engine for Bayesian NMF.
 *
 * This is the drop-in boundary.  The reference (jennalandy/bayesNMF, pure R) has no FFI;
 * the seam this library replaces is the body of the sampling loop
 *   R/bayesNMF_sampler.R:273-285  (sample_prior_params -> sample_params -> record_sample ->
 *                                  update_sample_metrics)
 * and the constructor's prior draws (R/bayesNMF_sampler.R:232-257).  An R `.Call` shim
 * (r/bnmf_shim.c) or Python ctypes (bayesnmf_amd/engine.py) binds exactly these symbols.
 *
 * Conventions
 *  - plain C types only; every matrix is column-major double exactly as R stores it:
 *    M[k+K*g], P[k+K*n], E[n+N*g], Z[k+K*(n+N*g)]; A is length N; R one value.
 *  - every function returns 0 on success, a negative BNMF_E* code otherwise; the message is
 *    available from bnmf_last_error() (thread-local).  No exceptions cross the boundary.
 *  - the caller owns every buffer it passes; the handle owns all device memory and one HIP
 *    stream.  One handle = one chain = one device.  Handles are independent.
 *  - there is NO CPU fallback: bnmf_create fails with BNMF_ENODEVICE when no gfx950 GPU is
 *    visible.
 */
#ifndef BNMF_H
#define BNMF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BNMF_VERSION 100
#define BNMF_NMETRIC 11   /* iter,RMSE,KL,loglikelihood,logposterior,n_params,BIC,rank,temp,
                             P_mean_acceptance_rate,E_mean_acceptance_rate
                             (state$sample_metrics columns, R/bayesNMF_sampler.R:190-207) */

enum { BNMF_OK = 0, BNMF_EINVAL = -1, BNMF_ESIZE = -2, BNMF_EUNSET = -3, BNMF_ENODEVICE = -4,
       BNMF_EHIP = -5, BNMF_EMODEL = -6, BNMF_ESTATE = -7 };

enum { BNMF_POISSON = 0, BNMF_NORMAL = 1 };                       /* likelihood */
enum { BNMF_TRUNCNORMAL = 0, BNMF_EXPONENTIAL = 1, BNMF_GAMMA = 2 }; /* prior */
enum { BNMF_SBFI = 0, BNMF_BFI = 1 };                             /* rank_method */

/* array ids for bnmf_set_array / bnmf_get_array / bnmf_window */
enum {
  BNMF_P = 0, BNMF_E = 1, BNMF_A = 2, BNMF_R = 3, BNMF_Z = 4, BNMF_ZSUMK = 5, BNMF_ZSUMG = 6,
  BNMF_SIGMASQ = 7,
  BNMF_ALPHA_P = 10, BNMF_BETA_P = 11, BNMF_ALPHA_E = 12, BNMF_BETA_E = 13,
  BNMF_MU_P = 14, BNMF_SIGMASQ_P = 15, BNMF_MU_E = 16, BNMF_SIGMASQ_E = 17,
  BNMF_LAMBDA_P = 18, BNMF_LAMBDA_E = 19, BNMF_ALPHA = 20, BNMF_BETA = 21,
  /* hyper-prior matrices (R/setup.R:15-88); n==1 means a scalar broadcast */
  BNMF_HA_P = 30, BNMF_HB_P = 31, BNMF_HC_P = 32, BNMF_HD_P = 33, BNMF_HM_P = 34, BNMF_HS_P = 35,
  BNMF_HA_E = 40, BNMF_HB_E = 41, BNMF_HC_E = 42, BNMF_HD_E = 43, BNMF_HM_E = 44, BNMF_HS_E = 45,
  BNMF_ACC_P = 50, BNMF_ACC_E = 51, BNMF_MHAT = 60, BNMF_ID_MAX = 64
};

/* Philox variable ids (counter word 3 of the stream spec) */
enum { BNMF_V_Z = 1, BNMF_V_P = 2, BNMF_V_E = 3, BNMF_V_BETA_P = 4, BNMF_V_ALPHA_P = 5,
       BNMF_V_BETA_E = 6, BNMF_V_ALPHA_E = 7, BNMF_V_MU_P = 8, BNMF_V_SIGSQ_P = 9,
       BNMF_V_MU_E = 10, BNMF_V_SIGSQ_E = 11, BNMF_V_LAMBDA_P = 12, BNMF_V_LAMBDA_E = 13,
       BNMF_V_A = 14, BNMF_V_R = 15, BNMF_V_MHU_P = 16, BNMF_V_MHU_E = 17, BNMF_V_SIGMASQ = 18 };

typedef struct bnmf_handle bnmf_handle;

typedef struct {
  int32_t K, G, N;          /* dims (R/bayesNMF_sampler.R:141-145) */
  int32_t likelihood;       /* BNMF_POISSON | BNMF_NORMAL */
  int32_t prior;            /* BNMF_TRUNCNORMAL | BNMF_EXPONENTIAL | BNMF_GAMMA */
  int32_t MH;               /* Metropolis-Hastings within Gibbs (Poisson only) */
  int32_t learning_rank;    /* rank given as a range -> A, R sampled */
  int32_t rank_method;      /* BNMF_SBFI | BNMF_BFI */
  int32_t save_Z;           /* materialise Z (K x N x G int32) every iteration ("full mode") */
  int32_t window;           /* samples kept on device for bnmf_window (MAP_over, or all) */
  uint64_t seed;            /* Philox key = (seed_lo, seed_hi ^ chain_id) */
  uint32_t chain_id;
  int32_t device;           /* HIP device ordinal */
  const double* temperature;/* temperature_schedule, length n_temperature (may be NULL = all 1) */
  int64_t n_temperature;
} bnmf_config;

/* create: copies M (int32, column-major K x G) to the device.  Replaces the data/dims part of
 * bayesNMF_sampler$new (R/bayesNMF_sampler.R:140-145). */
int bnmf_create(const bnmf_config* cfg, const int32_t* M_colmajor, bnmf_handle** out);
int bnmf_destroy(bnmf_handle* h);

/* set/get any array by id (column-major doubles; integer arrays are converted).  Hyper-prior
 * ids accept n==1 (scalar) — fill_matrix_ R/setup.R:102-113.  State arrays given before
 * bnmf_init are kept verbatim (init_prior_params / init_params, R/sample_priors.R:15-141,
 * R/sample_params.R:16-41); NaN entries mark "missing" and make that column n be re-drawn. */
int bnmf_set_array(bnmf_handle* h, int id, const double* colmajor, size_t n);
int bnmf_get_array(bnmf_handle* h, int id, double* colmajor_out, size_t n);
int bnmf_get_array_i32(bnmf_handle* h, int id, int32_t* out, size_t n);   /* Z, ZsumK, ZsumG */

/* constructor draws: prior params from hyper-priors unless supplied, then
 * sample_params(from_prior=TRUE), record iteration 1 and its metrics row
 * (R/bayesNMF_sampler.R:232-257).  metrics_row1 may be NULL. */
int bnmf_init(bnmf_handle* h, double* metrics_row1 /* BNMF_NMETRIC */);

/* n_iter iterations of the loop body (R/bayesNMF_sampler.R:273-285).  `converged` is
 * state$converged: it only matters for MH (accept-all before, true accept/reject after,
 * R/sample_Pn.R:199-204).  metrics_rowmajor: n_iter x BNMF_NMETRIC doubles (may be NULL). */
int bnmf_run(bnmf_handle* h, int n_iter, int converged, double* metrics_rowmajor);

/* last `last_n` recorded samples (<= window) of array `id`, oldest first:
 * out[last_n][len(id)] (record_sample, R/bayesNMF_sampler.R:651-672). */
int bnmf_window(bnmf_handle* h, int id, int last_n, double* out);

/* MAP estimate over the last `last_n` recorded samples, computed on the device (get_MAP_, R/utils.R:194-288;
 * get_mode / renormalize, R/helpers.R:35-79): mode of A over the window (ties: alphabetically first pattern, as
 * sort(table(.), decreasing = TRUE) gives); over the samples whose A equals the mode, P / colSums(P) and
 * E * colSums(P) are averaged element-wise (P_mean K x N, E_mean N x G, every factor; `final = TRUE` of the
 * reference is the caller dropping the factors with A_mode == 0); credible_interval in (0,1) also returns the
 * quantile(., type 7) bounds at (1 -+ credible_interval) / 2 (any of the four pointers may be NULL; <= 0: none);
 * used[last_n] flags the samples that entered (oldest first); top_A (5 x N, row-major) the most frequent patterns.
 * info.rmse / info.kl are compute_metrics_(P = MAP$P, A = MAP$A, E = MAP$E, MAP = TRUE) (R/utils.R:412-455). */
typedef struct {
  int32_t n_used;          /* samples whose A equals the mode */
  int32_t n_patterns;      /* distinct A patterns in the window */
  int32_t top_counts[5];   /* counts of the (up to) five most frequent patterns, descending */
  int32_t _pad;
  double rmse, kl;
} bnmf_map_info;
int bnmf_map(bnmf_handle* h, int last_n, double credible_interval, double* P_mean, double* E_mean,
             double* A_mode, double* top_A, double* P_lower, double* P_upper, double* E_lower,
             double* E_upper, int32_t* used, bnmf_map_info* info);

/* The sampling loop up to convergence as ONE call (run_gibbs_sampler, R/bayesNMF_sampler.R:268-330, warm-up part):
 * blocks of iterations up to the next MAP check; at a check get_MAP_ over the last MAP_over samples on the device,
 * update_MAP_metrics_ (R/utils.R:356-397) and check_convergence_ (R/convergence.R:60-154).
 *   metrics  [cap_rows][BNMF_NMETRIC]   one row per iteration run (n_rows returned)
 *   map_rows [cap_checks][BNMF_NMAPROW] one row per check: iter, RMSE, KL, loglikelihood, logposterior, n_params, BIC,
 *            rank, MAP_A_counts, mean_temp, P/E_mean_acceptance_rate (state$MAP_metrics columns), then percent change,
 *            inarow_no_change, inarow_no_best, inarow_na, converged after that check.
 * `st` carries state$prev_MAP_metric ... between calls (zero-initialise it for a fresh chain). */
#define BNMF_NMAPROW 17
typedef struct {
  int32_t MAP_over, MAP_every, Ninarow_nochange, Ninarow_nobest, miniters, maxiters;
  int32_t metric;                /* 0 loglikelihood, 1 logposterior, 2 RMSE, 3 KL, 4 BIC */
  int32_t _pad;
  double tol;
} bnmf_convergence_control;
typedef struct {
  int32_t converged, why /* 0 -, 1 "no change", 2 "no best", 3 "max iters" */, best_iter, inarow_na, inarow_no_change,
          inarow_no_best, have_prev, n_checks;
  double prev_MAP_metric, best_MAP_metric, prev_percent_change;
} bnmf_convergence_state;
int bnmf_run_until(bnmf_handle* h, const bnmf_convergence_control* cc, bnmf_convergence_state* st,
                   double* metrics_rowmajor, int cap_rows, int* n_rows, double* map_rows, int cap_checks,
                   int* n_checks);

/* The post-warm-up tail of the MH models as ONE call (R/bayesNMF_sampler.R:332-384): post_warmup more iterations with
 * state$converged = TRUE, i.e. true accept / reject in MH_Pn_poisson / MH_En_poisson (R/sample_Pn.R:199-248), a MAP check
 * (rows as in bnmf_run_until) whenever iter is a multiple of MAP_every and after the last iteration. */
int bnmf_run_post_warmup(bnmf_handle* h, const bnmf_convergence_control* cc, bnmf_convergence_state* st,
                         int post_warmup, double* metrics_rowmajor, int cap_rows, int* n_rows, double* map_rows,
                         int cap_checks, int* n_checks);

/* Posterior reference assignment (assign_signatures_ensemble_, R/postprocessing.R:175-341; hungarian_assignment,
 * pairwise_sim, R/helpers.R:218-398) over the recorded samples flagged in used[last_n] (MAP$idx; NULL = all):
 * cosine similarities of every sample's included signatures (keep[N] flags; NULL = all) with the reference catalogue
 * reference_P (K x R, column-major) and one Hungarian assignment per sample maximising the total cosine, both on the device,
 * votes[n + N*j] = sum over samples of the cosine of the pairs (n, j) chosen; assigned_ref[n] = which.max of the votes
 * (-1 for a signature that is not kept); MAP_P (K x N, may be NULL) -> MAP_cosine[n]; lower/upper_cosine[n] =
 * quantile(type 7) of the per-sample cosine between signature n and its assigned reference. */
int bnmf_assign(bnmf_handle* h, int last_n, const int32_t* used, const double* reference_P, int R,
                const int32_t* keep, const double* MAP_P, double credible_interval, double* votes,
                int32_t* assigned_ref, double* MAP_cosine, double* lower_cosine, double* upper_cosine);

int bnmf_get_iter(bnmf_handle* h, int* iter);
/* sizes of the handle's per-iteration buffers (bench.py's byte counts): what = 0 bytes of item records written per iteration (save_Z on the
 * sorted schedule: samples$Z is kept as these records and expanded when read), 1 bytes of Mhat left for the per-column metric terms,
 * 2 whether samples$Z is a ring of records, 3 whether Z is expanded every iteration (BNMF_ZEAGER=1), 4 whether the MH sweep hosts what followed
 * its two kernels inside them (Poisson MH models at fixed rank; BNMF_MHPIPE=0: no), 5 the quads (4 counts) per item of the sorted schedule */
int bnmf_get_stat(bnmf_handle* h, int what, double* out);

/* average device time (ms) of each kernel class over n_iter iterations, measured with HIP
 * events on the handle's own stream (advances the chain by n_iter iterations).
 * out_ms[BNMF_NKERNEL]; names from bnmf_kernel_name(). */
#define BNMF_NKERNEL 8
int bnmf_profile(bnmf_handle* h, int n_iter, int converged, double* out_ms);
const char* bnmf_kernel_name(int i);

/* measured ceilings of the device for bench.py's roofline: Philox4x32-7 words per second (the count-allocation generator) alone in a loop
 * (one word per allocated count: the floor of sample_Zkg, R/sample_params.R:253-265) and the device-to-device copy
 * bandwidth in GB/s (read + write), next to the nominal 8 TB/s */
int bnmf_ubench(int device, double* philox_words_per_s, double* copy_gbs);

/* device-side unit probes used by the parity tests (tests/test_gpu_*.py) */
int bnmf_test_math(int device, int fn, const double* in, double* out, size_t n);
int bnmf_test_sampler(int device, int which, uint64_t seed, uint32_t chain, uint32_t var,
                      uint32_t elem0, uint32_t iter, const double* a, const double* b,
                      const double* c, double* out, size_t n);
int bnmf_test_philox(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);    /* Philox4x32-10 */
int bnmf_test_philox7(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);   /* Philox4x32-7: the count-allocation words */

/* diagnostics of the builder's own tools and tests (tools/rankdbg.py, tools/zsprof.py, tools/zpprof.py, tests/test_gpu_parity.py):
 * phase time stamps of the rank sweep (BNMF_RANKDBG=1 at bnmf_create; returns the grid size), section ticks of the allocation
 * kernel (-DZSPROF / -DZPPROF builds only, else BNMF_ESTATE), and what a bounded in-kernel wait does when it gives up
 * (word 0: a draw kernel waiting for the hyper sweep, word 1: the rank sweep's exchange).  Not part of the drop-in boundary. */
/* host-only checks (no GPU) of what keeps the chains that share a device apart (api.hip DeviceGate, devlock_open; tests/test_abi_host.py):
 * the writer-preferring gate under overlapping sharers (how long the exclusive caller waited, sharers admitted while it did: 0), and the
 * opening of the device's two lock files under BNMF_LOCKDIR (or /tmp) with the lock order of bnmf_run.  Not part of the drop-in boundary. */
int bnmf_test_gate(int n_sharers, int calls_per_sharer, int hold_us, long* excl_wait_us, long* admitted_while_waiting);
int bnmf_test_devlock(const char* bus_tag, int* lock_ok, int* gate_ok);
int bnmf_debug_rank(bnmf_handle* h, unsigned long long* out, size_t n);
int bnmf_debug_zsort(bnmf_handle* h, unsigned long long* out);
int bnmf_debug_set_timeout(bnmf_handle* h, int word);

/* Can two kernels on two streams of this device run at the same time?  Measured once per device and process by a two-stream
 * hand-off (kernel A spins, bounded, on a word kernel B sets).  bnmf_create uses it to choose between flag polling inside the
 * kernels (overlap) and stream waits on events ("serial-safe mode": counter collection, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING
 * and any other tool that serialises dispatches).  *overlap = 1 / 0. */
int bnmf_probe_overlap(int device, int* overlap);

/* What destroyed handles leave cached on a device for the next handle — their record_sample rings (up to BNMF_RING_CACHE_GB, default 8)
 * and three HIP streams each — is released; bytes_released (may be NULL) = device memory given back. */
int bnmf_trim(int device, size_t* bytes_released);

int bnmf_device_info(int device, char* buf, size_t buflen);
int bnmf_device_count(void);
const char* bnmf_last_error(void);
int bnmf_version(void);

#ifdef __cplusplus
}
#endif
#endif
