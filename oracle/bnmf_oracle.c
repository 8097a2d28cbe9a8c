/* oracle/bnmf_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, optional OpenMP over columns) of the Gibbs sweep
 * of the reference R package jennalandy/bayesNMF, function by function, driven by
 * the Philox stream spec of DESIGN.md §4.  It exists to CHECK the HIP engine
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The product
 * (bayesnmf_amd/, libbnmf.so) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference is pure R with no tests, no golden vectors and
 * no seeds; R is not installed in this image; R's
 * Mersenne-Twister + nmath samplers cannot be reproduced by a counter-based
 * generator.  What this file pins instead: each conditional is the one the cited R
 * line specifies (checked distributionally against scipy in tests/), and the HIP
 * engine must agree with this file bit-for-bit on integers and to stated ulps on fp.
 *
 * Reference files followed (all under /root/reference/R/):
 *   sample_params.R:51-89 (sweep order), :101-206 (A), :217-241 (R), :253-265 (Z)
 *   sample_Pn.R:11-42,98-120 ; sample_En.R:11-42,97-119 (conjugate Gamma draws)
 *   sample_priors.R:15-141 (init from hyper-priors), :150-200, :284-397 (hyper sweep)
 *   utils.R:29-183, :412-471 (Mhat, log-lik, log-posterior, metrics)
 *   bayesNMF_sampler.R:232-257 (constructor order), :273-285 (loop body)
 * All matrices are column-major exactly like R: M[k+K*g], P[k+K*n], E[n+N*g],
 * Z[k+K*(n+N*g)].
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc_math.h"
#include "orc_samplers.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NMETRIC 11
enum { LIK_POISSON = 0, LIK_NORMAL = 1 };
enum { PRIOR_TRUNCNORMAL = 0, PRIOR_EXPONENTIAL = 1, PRIOR_GAMMA = 2 };
enum { RANK_SBFI = 0, RANK_BFI = 1 };
/* Philox variable ids (counter word 3) — shared with include/bnmf.h */
enum { V_Z = 1, V_P = 2, V_E = 3, V_BETA_P = 4, V_ALPHA_P = 5, V_BETA_E = 6, V_ALPHA_E = 7,
       V_MU_P = 8, V_SIGSQ_P = 9, V_MU_E = 10, V_SIGSQ_E = 11, V_LAMBDA_P = 12, V_LAMBDA_E = 13,
       V_A = 14, V_R = 15, V_MHU_P = 16, V_MHU_E = 17, V_SIGMASQ = 18 };
/* array ids — shared with include/bnmf.h */
enum { ID_P = 0, ID_E = 1, ID_A = 2, ID_R = 3, ID_Z = 4, ID_ZSUMK = 5, ID_ZSUMG = 6, ID_SIGMASQ = 7,
       ID_ALPHA_P = 10, ID_BETA_P = 11, ID_ALPHA_E = 12, ID_BETA_E = 13,
       ID_MU_P = 14, ID_SIGSQ_P = 15, ID_MU_E = 16, ID_SIGSQ_E = 17,
       ID_LAMBDA_P = 18, ID_LAMBDA_E = 19, ID_ALPHA_S = 20, ID_BETA_S = 21,
       ID_HA_P = 30, ID_HB_P = 31, ID_HC_P = 32, ID_HD_P = 33, ID_HM_P = 34, ID_HS_P = 35,
       ID_HA_E = 40, ID_HB_E = 41, ID_HC_E = 42, ID_HD_E = 43, ID_HM_E = 44, ID_HS_E = 45,
       ID_ACC_P = 50, ID_ACC_E = 51, ID_MHAT = 60, ID_MAX = 64 };

typedef struct {
  int32_t K, G, N;
  int32_t likelihood, prior, MH, learning_rank, rank_method;
  int32_t save_Z, nthreads;
  uint64_t seed;
  uint32_t chain_id;
  uint32_t _pad;
  const double* temperature;
  int64_t n_temperature;
} orc_config;

typedef struct { double* p; long n; long stride; int set; } arr_t;

typedef struct orc_handle {
  orc_config cfg;
  int iter, converged, R;
  int fresh_mhat;   /* checker of the checker (orc_set_fresh_mhat): recompute P diag(A) E from scratch wherever the stream spec maintains it */
  int32_t* M;
  int32_t *Z, *ZsumK, *ZsumG;
  arr_t a[ID_MAX];
  double* temperature;
  double *colsse, *colll, *colkl, *Mhat, *lpE, *lpP;
  char err[256];
} orc_handle;

static long id_len(const orc_handle* o, int id) {
  long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  switch (id) {
    case ID_P: case ID_ZSUMG: case ID_ALPHA_P: case ID_BETA_P: case ID_MU_P: case ID_SIGSQ_P:
    case ID_LAMBDA_P: case ID_HA_P: case ID_HB_P: case ID_HC_P: case ID_HD_P: case ID_HM_P:
    case ID_HS_P: case ID_ACC_P: return K * N;
    case ID_E: case ID_ZSUMK: case ID_ALPHA_E: case ID_BETA_E: case ID_MU_E: case ID_SIGSQ_E:
    case ID_LAMBDA_E: case ID_HA_E: case ID_HB_E: case ID_HC_E: case ID_HD_E: case ID_HM_E:
    case ID_HS_E: case ID_ACC_E: return N * G;
    case ID_A: return N;
    case ID_R: return 1;
    case ID_Z: return K * N * G;
    case ID_SIGMASQ: case ID_ALPHA_S: case ID_BETA_S: return G;
    case ID_MHAT: return K * G;
    default: return -1;
  }
}
static int is_hyper(int id) { return (id >= 30 && id < 50) || id == ID_ALPHA_S || id == ID_BETA_S; }
#define AR(id) (o->a[id].p)
#define HY(id, e) (o->a[id].p[(e) * o->a[id].stride])

orc_handle* orc_create(const orc_config* cfg, const int32_t* M) {
  orc_handle* o = (orc_handle*)calloc(1, sizeof(orc_handle));
  o->cfg = *cfg;
  long K = cfg->K, G = cfg->G, N = cfg->N;
  o->M = (int32_t*)malloc(sizeof(int32_t) * K * G);
  memcpy(o->M, M, sizeof(int32_t) * K * G);
  o->ZsumK = (int32_t*)calloc(N * G, sizeof(int32_t));
  o->ZsumG = (int32_t*)calloc(K * N, sizeof(int32_t));
  o->Z = cfg->save_Z ? (int32_t*)calloc(K * N * G, sizeof(int32_t)) : NULL;
  o->colsse = (double*)calloc(G, 8); o->colll = (double*)calloc(G, 8); o->colkl = (double*)calloc(G, 8);
  o->lpE = (double*)calloc((N * G + 255) / 256 * 256, 8);
  o->lpP = (double*)calloc(K * N, 8);
  o->temperature = (double*)malloc(8 * (cfg->n_temperature > 0 ? cfg->n_temperature : 1));
  if (cfg->n_temperature > 0) memcpy(o->temperature, cfg->temperature, 8 * cfg->n_temperature);
  o->cfg.temperature = o->temperature;
  o->iter = 0; o->converged = 0; o->R = (int)N;
  return o;
}
void orc_destroy(orc_handle* o) {
  if (!o) return;
  for (int i = 0; i < ID_MAX; ++i) free(o->a[i].p);
  free(o->M); free(o->Z); free(o->ZsumK); free(o->ZsumG); free(o->temperature);
  free(o->colsse); free(o->colll); free(o->colkl); free(o->lpE); free(o->lpP); free(o->Mhat);
  free(o);
}
static double* ensure(orc_handle* o, int id) {
  if (!o->a[id].p) {
    long n = id_len(o, id);
    o->a[id].p = (double*)malloc(8 * n);
    for (long i = 0; i < n; ++i) o->a[id].p[i] = NAN;
    o->a[id].n = n; o->a[id].stride = 1;
  }
  return o->a[id].p;
}
/* set: hyper arrays may be scalars (n==1, broadcast).  State arrays may carry NaN = "not supplied". */
int orc_set_array(orc_handle* o, int id, const double* x, long n) {
  long len = id_len(o, id);
  if (len < 0) return -1;
  if (id == ID_R) { o->R = (int)x[0]; o->a[ID_R].set = 1; return 0; }
  if (id == ID_ZSUMK || id == ID_ZSUMG || id == ID_Z) {
    int32_t* dst = id == ID_ZSUMK ? o->ZsumK : id == ID_ZSUMG ? o->ZsumG : o->Z;
    if (!dst || n != len) return -2;
    for (long i = 0; i < n; ++i) dst[i] = (int32_t)x[i];
    return 0;
  }
  if (is_hyper(id) && n == 1) {
    free(o->a[id].p);
    o->a[id].p = (double*)malloc(8); o->a[id].p[0] = x[0];
    o->a[id].n = 1; o->a[id].stride = 0; o->a[id].set = 1;
    return 0;
  }
  if (n != len) return -2;
  free(o->a[id].p);
  o->a[id].p = (double*)malloc(8 * n);
  memcpy(o->a[id].p, x, 8 * n);
  o->a[id].n = n; o->a[id].stride = 1; o->a[id].set = 1;
  return 0;
}
int orc_get_array(orc_handle* o, int id, double* out, long n) {
  long len = id_len(o, id);
  if (len < 0 || n != len) return -1;
  if (id == ID_R) { out[0] = o->R; return 0; }
  if (id == ID_ZSUMK || id == ID_ZSUMG || id == ID_Z) {
    const int32_t* src = id == ID_ZSUMK ? o->ZsumK : id == ID_ZSUMG ? o->ZsumG : o->Z;
    if (!src) return -3;
    for (long i = 0; i < n; ++i) out[i] = src[i];
    return 0;
  }
  if (!o->a[id].p) return -3;
  for (long i = 0; i < n; ++i) out[i] = o->a[id].p[i * o->a[id].stride];
  return 0;
}
int orc_get_iter(orc_handle* o) { return o->iter; }

static inline orc_stream ST(const orc_handle* o, uint32_t var, uint32_t elem, uint32_t iter) {
  return orc_stream_make(o->cfg.seed, o->cfg.chain_id, var, elem, iter);
}
static inline double clamp_tiny(double v) { return (v < 1e-300) ? 1e-300 : v; }

/* ---- hyper sweep: R/sample_priors.R:150-200 (element-wise; every draw depends only on
 * element (k,n)/(n,g) of the previous iteration's P/E and its just-updated sibling) ---- */
static void hyper_elem(orc_handle* o, int side /*0=p,1=e*/, long e, uint32_t t) {
  const int pr = o->cfg.prior;
  double v = side ? AR(ID_E)[e] : AR(ID_P)[e];
  if (pr == PRIOR_GAMMA) {
    int idA = side ? ID_HA_E : ID_HA_P, idB = side ? ID_HB_E : ID_HB_P;
    int idC = side ? ID_HC_E : ID_HC_P, idD = side ? ID_HD_E : ID_HD_P;
    double* Al = side ? AR(ID_ALPHA_E) : AR(ID_ALPHA_P);
    double* Be = side ? AR(ID_BETA_E) : AR(ID_BETA_P);
    /* sample_Beta_Pn / _En  R/sample_priors.R:323-345 */
    orc_stream s = ST(o, side ? V_BETA_E : V_BETA_P, (uint32_t)e, t);
    double b = orc_rgamma(&s, HY(idA, e) + Al[e], HY(idB, e) + v);
    Be[e] = b;
    /* sample_Alpha_Pkn / _Eng  R/sample_priors.R:356-397 */
    double tau = (HY(idD, e) - orc_log(clamp_tiny(b))) - orc_log(clamp_tiny(v));
    orc_stream s2 = ST(o, side ? V_ALPHA_E : V_ALPHA_P, (uint32_t)e, t);
    Al[e] = orc_ralpha_fast(&s2, HY(idC, e), tau, Al[e], NULL);
  } else if (pr == PRIOR_EXPONENTIAL) {
    /* sample_Lambda_Pn / _En  R/sample_priors.R:284-308 */
    int idA = side ? ID_HA_E : ID_HA_P, idB = side ? ID_HB_E : ID_HB_P;
    double* La = side ? AR(ID_LAMBDA_E) : AR(ID_LAMBDA_P);
    orc_stream s = ST(o, side ? V_LAMBDA_E : V_LAMBDA_P, (uint32_t)e, t);
    La[e] = orc_rgamma(&s, HY(idA, e) + 1.0, HY(idB, e) + v);
  } else {
    /* truncnormal: sample_Mu_* R/sample_priors.R:214-236 (sd = 1/denom, quirk kept),
     * sample_Sigmasq_* :246-270 (E side uses A_e where B_e is meant, quirk kept) */
    int idM = side ? ID_HM_E : ID_HM_P, idS = side ? ID_HS_E : ID_HS_P;
    int idA = side ? ID_HA_E : ID_HA_P, idB = side ? ID_HB_E : ID_HB_P;
    double* Mu = side ? AR(ID_MU_E) : AR(ID_MU_P);
    double* Sg = side ? AR(ID_SIGSQ_E) : AR(ID_SIGSQ_P);
    double num = HY(idM, e) / HY(idS, e) + v / Sg[e];
    double den = 1.0 / HY(idS, e) + 1.0 / Sg[e];
    orc_stream s = ST(o, side ? V_MU_E : V_MU_P, (uint32_t)e, t);
    double mu = num / den + (1.0 / den) * orc_rnorm_std(&s);
    Mu[e] = mu;
    double dlt = v - mu;
    double rate = (side ? HY(idA, e) : HY(idB, e)) + (dlt * dlt) / 2.0;
    orc_stream s2 = ST(o, side ? V_SIGSQ_E : V_SIGSQ_P, (uint32_t)e, t);
    Sg[e] = orc_rinvgamma(&s2, HY(idA, e) + 0.5, rate);
  }
}

/* draw element e of P (side 0) / E (side 1) from its prior: R/sample_Pn.R:12-30, R/sample_En.R:12-30 */
static double prior_draw(orc_handle* o, int side, long e, uint32_t t) {
  orc_stream s = ST(o, side ? V_E : V_P, (uint32_t)e, t);
  switch (o->cfg.prior) {
    case PRIOR_GAMMA:
      return orc_rgamma(&s, (side ? AR(ID_ALPHA_E) : AR(ID_ALPHA_P))[e], (side ? AR(ID_BETA_E) : AR(ID_BETA_P))[e]);
    case PRIOR_EXPONENTIAL:
      return orc_rexp(&s, (side ? AR(ID_LAMBDA_E) : AR(ID_LAMBDA_P))[e]);
    default:
      return orc_rtnorm0(&s, (side ? AR(ID_MU_E) : AR(ID_MU_P))[e], sqrt((side ? AR(ID_SIGSQ_E) : AR(ID_SIGSQ_P))[e]));
  }
}
/* log prior density of element e at value x: R/utils.R:132-175 */
static double prior_logdens(orc_handle* o, int side, long e, double x) {
  switch (o->cfg.prior) {
    case PRIOR_GAMMA: {
      double al = (side ? AR(ID_ALPHA_E) : AR(ID_ALPHA_P))[e], be = (side ? AR(ID_BETA_E) : AR(ID_BETA_P))[e];
      return ((al * orc_log(be) - orc_lgamma(al)) + (al - 1.0) * orc_log(x)) - be * x;
    }
    case PRIOR_EXPONENTIAL: {
      double la = (side ? AR(ID_LAMBDA_E) : AR(ID_LAMBDA_P))[e];
      return orc_log(la) - la * x;
    }
    default: {
      double mu = (side ? AR(ID_MU_E) : AR(ID_MU_P))[e], sg = sqrt((side ? AR(ID_SIGSQ_E) : AR(ID_SIGSQ_P))[e]);
      double zz = (x - mu) / sg;
      return ((-0.91893853320467274178 - orc_log(sg)) - 0.5 * (zz * zz)) - orc_log_pnorm(mu / sg);
    }
  }
}

/* ---- conjugate Gamma updates, Poisson without MH ---- */
/* sample_Pn_poisson  R/sample_Pn.R:98-120 ; sample_Pn dispatch :11-42 */
static void sample_P_poisson(orc_handle* o, uint32_t t, int from_prior) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  for (long n = 0; n < N; ++n) {
    double a_n = AR(ID_A)[n];
    double Esum = from_prior ? 0.0 : orc_canon_sum(AR(ID_E) + n, G, N, 1024);
    for (long k = 0; k < K; ++k) {
      long e = k + K * n;
      if (from_prior || a_n == 0.0) { AR(ID_P)[e] = prior_draw(o, 0, e, t); continue; }
      double shape, rate;
      if (o->cfg.prior == PRIOR_GAMMA) { shape = AR(ID_ALPHA_P)[e] + (double)o->ZsumG[e]; rate = AR(ID_BETA_P)[e] + a_n * Esum; }
      else { shape = 1.0 + (double)o->ZsumG[e]; rate = AR(ID_LAMBDA_P)[e] + a_n * Esum; }
      orc_stream s = ST(o, V_P, (uint32_t)e, t);
      AR(ID_P)[e] = orc_rgamma(&s, shape, rate);
    }
  }
}
/* sample_En_poisson  R/sample_En.R:97-119 */
static void sample_E_poisson(orc_handle* o, uint32_t t, int from_prior) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  double Psum[N];
  for (long n = 0; n < N; ++n) Psum[n] = orc_canon_sum(AR(ID_P) + K * n, K, 1, 64);
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g)
    for (long n = 0; n < N; ++n) {
      long e = n + N * g;
      double a_n = AR(ID_A)[n];
      if (from_prior || a_n == 0.0) { AR(ID_E)[e] = prior_draw(o, 1, e, t); continue; }
      double shape, rate;
      if (o->cfg.prior == PRIOR_GAMMA) { shape = AR(ID_ALPHA_E)[e] + (double)o->ZsumK[e]; rate = AR(ID_BETA_E)[e] + a_n * Psum[n]; }
      else { shape = 1.0 + (double)o->ZsumK[e]; rate = AR(ID_LAMBDA_E)[e] + a_n * Psum[n]; }
      orc_stream s = ST(o, V_E, (uint32_t)e, t);
      AR(ID_E)[e] = orc_rgamma(&s, shape, rate);
    }
}

/* ---- Z allocation + fused per-cell metric terms ----
 * sample_Zkg R/sample_params.R:253-265: probs[n] = P[k,n]*A[n]*E[n,g]; sum==0 -> zeros;
 * else Multinomial(M[k,g], probs/sum).  Stream spec: the M[k,g] counts of a cell are
 * allocated one by one; count j uses 32-bit word (j&3) of block (j>>2) of stream
 * (V_Z, cell=k+K*g, iter) — Philox4x32 with SEVEN rounds for this stream —; it lands in the first n whose cumulative threshold
 * thr[n] = floor(cum[n] * 2^32 / sum) exceeds the word; the word is clamped to 2^32-2 and
 * thresholds at or beyond the last n with a positive probability (or that saturate) are
 * 2^32-1 = "never", so no count can land on a factor of zero probability.  The sum of M independent categorical draws is
 * exactly the multinomial R's rmultinom samples by conditional binomials.
 * The same pass yields Mhat[k,g] = sum (get_Mhat_, R/utils.R:29-49) and the per-cell
 * terms of RMSE / KL / Poisson log-lik (R/utils.R:62-112, :412-471). */
static void z_cell(const orc_handle* o, long k, long g, uint32_t t, int32_t* zrow, double* mhat) {
  const long K = o->cfg.K, N = o->cfg.N;
  const double *P = o->a[ID_P].p, *E = o->a[ID_E].p, *A = o->a[ID_A].p;
  double cum[N]; uint32_t thr[N];
  double c = 0.0; long nlast = -1;
  for (long n = 0; n < N; ++n) {
    double p = (P[k + K * n] * A[n]) * E[n + N * g];
    c = c + p; cum[n] = c;
    if (p > 0.0) nlast = n;
    zrow[n] = 0;
  }
  *mhat = c;
  int32_t m = o->M[k + K * g];
  if (!(c > 0.0) || m <= 0 || nlast < 0) return;
  double scale = 4294967296.0 / c;
  for (long n = 0; n < N; ++n) {
    double tt = cum[n] * scale;
    thr[n] = (n >= nlast || tt >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)tt;   /* 2^32-1 = "never" */
  }
  orc_stream s = ST(o, V_Z, (uint32_t)(k + K * g), t);
  uint32_t w[4];
  for (int32_t j = 0; j < m; ++j) {
    if ((j & 3) == 0) { orc_philox4x32_r(s.blk, s.elem, s.iter, s.var, s.k0, s.k1, w, 7); s.blk++; }
    uint32_t u = w[j & 3];
    if (u > 0xFFFFFFFEu) u = 0xFFFFFFFEu;
    long b = 0;
    while (b < N - 1 && thr[b] <= u) ++b;
    zrow[b]++;
  }
}
static void cell_terms(const orc_handle* o, int32_t m, double mhat, double* sse, double* ll, double* kl) {
  (void)o;
  double d = mhat - (double)m;
  *sse = d * d;
  double mh = mhat < 1e-6 ? 1e-6 : mhat;     /* pmax(Mhat, 1e-6)  R/utils.R:100, :468 */
  double lmh = orc_log(mh);
  double mt = m < 1 ? 1e-6 : (double)m;        /* pmax(M, 1e-6)     R/utils.R:469 */
  *ll = ((double)m * lmh - mh) - orc_lgamma((double)m + 1.0);   /* dpois(M, Mhat, log=TRUE) */
  *kl = mt * (orc_log(mt) - lmh);
}
static void sample_Z_and_metrics(orc_handle* o, uint32_t t) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  memset(o->ZsumG, 0, sizeof(int32_t) * K * N);
  int nth = o->cfg.nthreads > 0 ? o->cfg.nthreads : 1;
  int32_t* zg_part = (int32_t*)calloc((size_t)nth * K * N, sizeof(int32_t));
#pragma omp parallel num_threads(nth)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    int32_t* zg = zg_part + (size_t)tid * K * N;
    int32_t zrow[N];
    double* csse = (double*)malloc(8 * K * 3);
    double *cll = csse + K, *ckl = csse + 2 * K;
#pragma omp for schedule(static)
    for (long g = 0; g < G; ++g) {
      for (long n = 0; n < N; ++n) o->ZsumK[n + N * g] = 0;
      for (long k = 0; k < K; ++k) {
        double mhat;
        z_cell(o, k, g, t, zrow, &mhat);
        for (long n = 0; n < N; ++n) {
          o->ZsumK[n + N * g] += zrow[n];
          zg[k + K * n] += zrow[n];
          if (o->Z) o->Z[k + K * (n + N * g)] = zrow[n];
        }
        cell_terms(o, o->M[k + K * g], mhat, &csse[k], &cll[k], &ckl[k]);
      }
      o->colsse[g] = orc_canon_sum(csse, K, 1, 64);
      o->colll[g] = orc_canon_sum(cll, K, 1, 64);
      o->colkl[g] = orc_canon_sum(ckl, K, 1, 64);
    }
    free(csse);
  }
  for (int th = 0; th < nth; ++th)
    for (long i = 0; i < K * N; ++i) o->ZsumG[i] += zg_part[(size_t)th * K * N + i];
  free(zg_part);
}

/* ---- rank learning: sample_R R/sample_params.R:217-241, sample_An :101-166 ---- */
static double prior_prob_1(double R, double N) {   /* compute_prior_prob_1 :178-187 */
  double p = R / N;
  if (p < 0.4 / N) p = 0.4 / N;
  if (p > 1.0 - 0.4 / N) p = 1.0 - 0.4 / N;
  return p;
}
static double temp_at(const orc_handle* o, int iter) {
  if (o->cfg.n_temperature <= 0) return 1.0;
  long i = iter - 1;
  if (i < 0) i = 0;
  if (i >= o->cfg.n_temperature) i = o->cfg.n_temperature - 1;
  return o->temperature[i];
}
static void sample_R(orc_handle* o, uint32_t t, int from_prior) {
  const int N = o->cfg.N;
  orc_stream s = ST(o, V_R, 0, t);
  double u = orc_runif(&s);
  if (from_prior) { int r = (int)(u * (double)(N + 1)); if (r > N) r = N; o->R = r; return; }
  double T = temp_at(o, (int)t);
  double sumA = 0.0;
  for (int n = 0; n < N; ++n) sumA = sumA + AR(ID_A)[n];
  double w[4097], tot = 0.0;
  for (int r = 0; r <= N; ++r) {
    double p1 = prior_prob_1((double)r, (double)N);
    double lw = T * (sumA * orc_log(p1) + ((double)N - sumA) * orc_log(1.0 - p1));
    w[r] = orc_exp(lw);
    tot = tot + w[r];
  }
  double target = u * tot, cum = 0.0;
  int pick = N;
  for (int r = 0; r <= N; ++r) { cum = cum + w[r]; if (target < cum) { pick = r; break; } }
  o->R = pick;
}
static double dnorm_log_fwd(double x, double mean, double var) {   /* dnorm(x, mean, sqrt(var), log = TRUE) */
  double sd = sqrt(var);
  double z = (x - mean) / sd;
  return (-0.91893853320467274178 - orc_log(sd)) - 0.5 * (z * z);
}
static double pois_ll_cell(int32_t m, double mhat) {
  double mh = mhat < 1e-6 ? 1e-6 : mhat;
  return ((double)m * orc_log(mh) - mh) - orc_lgamma((double)m + 1.0);
}
/* sample_An R/sample_params.R:101-166, for n = 1..N in order.  R evaluates two full log-likelihoods per factor,
 * get_loglik(A = A0) and get_loglik(A = A1), each from a BLAS product P diag(A^j) E.  Stream spec (same conditional,
 * O(N K G) instead of O(N^2 K G)): Mhat = P diag(A) E is computed fresh (factor order, as get_Mhat_ R/utils.R:29-49)
 * once at the start of the rank sweep and then maintained per cell; for factor n only the ALTERNATIVE state is
 * evaluated, alt = Mhat -/+ P[k,n] E[n,g] (A[n] = 1 / 0 now), the log-likelihood of the current state being the one
 * carried from the previous decision.  Sums: per column 64-strided over k + tree; the column sums are added in blocks
 * of 8 consecutive columns (sequentially, ascending); W = 1024 over the blocks. */
static double ll_cell_rank(const orc_handle* o, int32_t m, double mhat, long g) {
  if (o->cfg.likelihood == LIK_NORMAL) return dnorm_log_fwd((double)m, mhat, o->a[ID_SIGMASQ].p[g]);
  return pois_ll_cell(m, mhat);
}
static double hsum_cols(const double* col, long G) {
  long nb = (G + 7) / 8;
  double* blk = (double*)malloc(8 * nb);
  for (long b = 0; b < nb; ++b) { double a = 0.0; for (long g = 8 * b; g < 8 * b + 8 && g < G; ++g) a = a + col[g]; blk[b] = a; }
  double r = orc_canon_sum(blk, nb, 1, 1024);
  free(blk);
  return r;
}
static void sample_A(orc_handle* o, uint32_t t, int from_prior) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  double pi1 = prior_prob_1((double)o->R, (double)N);
  if (from_prior) {
    for (long n = 0; n < N; ++n) { orc_stream s = ST(o, V_A, (uint32_t)n, t); AR(ID_A)[n] = (orc_runif(&s) < pi1) ? 1.0 : 0.0; }
    return;
  }
  double T = temp_at(o, (int)t);
  double* mh = (double*)malloc(8 * K * G);     /* [k + K g] */
  double* col = (double*)malloc(8 * G);
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g) {
    double l[K];
    for (long k = 0; k < K; ++k) {
      double c = 0.0;
      for (long j = 0; j < N; ++j) c = c + (AR(ID_P)[k + K * j] * AR(ID_A)[j]) * AR(ID_E)[j + N * g];
      mh[k + K * g] = c;
      l[k] = ll_cell_rank(o, o->M[k + K * g], c, g);
    }
    col[g] = orc_canon_sum(l, K, 1, 64);
  }
  double ll_cur = hsum_cols(col, G);
  for (long n = 0; n < N; ++n) {
    double a_old = AR(ID_A)[n];
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
    for (long g = 0; g < G; ++g) {
      double l[K];
      double en = AR(ID_E)[n + N * g];
      for (long k = 0; k < K; ++k) {
        double tt = AR(ID_P)[k + K * n] * en;
        double alt = (a_old == 1.0) ? mh[k + K * g] - tt : mh[k + K * g] + tt;
        l[k] = ll_cell_rank(o, o->M[k + K * g], alt, g);
      }
      col[g] = orc_canon_sum(l, K, 1, 64);
    }
    double ll_alt = hsum_cols(col, G);
    double ll0 = (a_old == 1.0) ? ll_alt : ll_cur, ll1 = (a_old == 1.0) ? ll_cur : ll_alt;
    if (o->fresh_mhat) {                                          /* as R: both log-likelihoods from P diag(A^j) E, from scratch */
      for (int state = 0; state < 2; ++state) {
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
        for (long g = 0; g < G; ++g) {
          double l[K];
          for (long k = 0; k < K; ++k) {
            double c = 0.0;
            for (long j = 0; j < N; ++j) c = c + (AR(ID_P)[k + K * j] * (j == n ? (double)state : AR(ID_A)[j])) * AR(ID_E)[j + N * g];
            l[k] = ll_cell_rank(o, o->M[k + K * g], c, g);
          }
          col[g] = orc_canon_sum(l, K, 1, 64);
        }
        if (state == 0) ll0 = hsum_cols(col, G); else ll1 = hsum_cols(col, G);
      }
    }
    double sumA = 0.0;
    for (long j = 0; j < N; ++j) sumA = sumA + AR(ID_A)[j];
    double sumA0 = sumA - a_old, sumA1 = sumA0 + 1.0;
    double s0 = ll0, s1 = ll1;
    if (o->cfg.rank_method == RANK_SBFI) {
      double lg = orc_log((double)G);
      s0 = ll0 - (sumA0 * (double)(G + K)) * lg / 2.0;
      s1 = ll1 - (sumA1 * (double)(G + K)) * lg / 2.0;
    }
    double lp0 = orc_log(1.0 - pi1) + T * s0;
    double lp1 = orc_log(pi1) + T * s1;
    double hi = lp0 > lp1 ? lp0 : lp1, lo = lp0 > lp1 ? lp1 : lp0;
    double lse = hi + orc_log(1.0 + orc_exp(lo - hi));          /* sumLog :199-206 */
    double p = orc_exp(lp1 - lse);
    if (p != p) {                                                 /* overflow clamp :136-162 */
      if (lp1 != lp1 && lp0 != lp0) p = 0.5; else if (lp1 != lp1) p = 0.0; else if (lp0 != lp0) p = 1.0;
      else if (lp1 > lp0) p = 1.0; else if (lp1 < lp0) p = 0.0; else p = 0.5;
    }
    orc_stream s = ST(o, V_A, (uint32_t)n, t);
    double a_new = (orc_runif(&s) < p) ? 1.0 : 0.0;
    if (a_new != a_old) {
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
      for (long g = 0; g < G; ++g) {
        double en = AR(ID_E)[n + N * g];
        for (long k = 0; k < K; ++k) {
          double tt = AR(ID_P)[k + K * n] * en;
          mh[k + K * g] = (a_old == 1.0) ? mh[k + K * g] - tt : mh[k + K * g] + tt;
        }
      }
      ll_cur = ll_alt;
    }
    AR(ID_A)[n] = a_new;
  }
  free(mh); free(col);
}

/* ================= Poisson likelihood with Metropolis-Hastings (truncnormal / exponential prior), and the
 * Normal likelihood (plain Gibbs) =====
 * sample_Pn -> sample_Pn_normal(as_proposal) -> MH_Pn_poisson (R/sample_Pn.R:11-42, :54-87, :132-187, :199-248) and
 * the E mirror (R/sample_En.R); sample_sigmasq R/sample_params.R:275-286.  Columns/rows are updated for n = 1..N in
 * order.  R recomputes Mhat = P diag(A) E (and Mhat with A[n] = 0) by BLAS products at every step.  Stream spec (same
 * conditionals, O(N K G) instead of O(N^2 K G)): Mhat is computed fresh (factor order) at the start of the P sweep
 * and again at the start of the E sweep, then maintained per cell: Mhat_no_n = Mhat - (P[k,n] A[n]) E[n,g], and after
 * the update of factor n, Mhat = Mhat_no_n + (P_new[k,n] A[n]) E[n,g].  Rows of P are mutually independent within the
 * P sweep, columns of E within the E sweep.
 * Canonical sums: over k (column sums) 64-strided + tree; over g (row sums) segments of 320 columns, each
 * 64-strided + tree, segments added in ascending order. */
#define ORC_SEG 320
static double canon_rowsum(const double* x /* G values */, long G) {
  double acc = 0.0;
  for (long s0 = 0; s0 < G; s0 += ORC_SEG) {
    long len = G - s0 < ORC_SEG ? G - s0 : ORC_SEG;
    acc = acc + orc_canon_sum(x + s0, len, 1, 64);
  }
  return acc;
}
static inline double mhat_cell(const orc_handle* o, long k, long g, long skip /* -1: none */, const double* Prow_override, long n_over) {
  const long K = o->cfg.K, N = o->cfg.N;
  double c = 0.0;
  for (long j = 0; j < N; ++j) {
    double a = (j == skip) ? 0.0 : o->a[ID_A].p[j];
    double pkj = (Prow_override && j == n_over) ? *Prow_override : o->a[ID_P].p[k + K * j];
    c = c + (pkj * a) * o->a[ID_E].p[j + N * g];
  }
  return c;
}
static inline double mhat_cell_e(const orc_handle* o, long k, long g, long skip, double e_over, long n_over) {   /* E[n_over, g] replaced */
  const long K = o->cfg.K, N = o->cfg.N;
  double c = 0.0;
  for (long j = 0; j < N; ++j) {
    double a = (j == skip) ? 0.0 : o->a[ID_A].p[j];
    double ejg = (j == n_over) ? e_over : o->a[ID_E].p[j + N * g];
    c = c + (o->a[ID_P].p[k + K * j] * a) * ejg;
  }
  return c;
}
static inline double dpois_log(int32_t m, double lam) {   /* get_loglik_ poisson branch R/utils.R:98-106 */
  double mh = lam < 1e-6 ? 1e-6 : lam;
  return ((double)m * orc_log(mh) - mh) - orc_lgamma((double)m + 1.0);
}
/* dnorm(x, mean, sqrt(var), log = TRUE) in variance form, -(log(2 pi) + log(var)) / 2 - (x - mean)^2 / (2 var): with
 * var = max(Mhat, 1) the logarithm is the one the Poisson term of the same cell already needs (stream spec) */
static inline double dnorm_log(double x, double mean, double var) {
  double dlt = x - mean;
  return (-0.91893853320467274178 - 0.5 * orc_log(var)) - 0.5 * ((dlt * dlt) / var);
}
static void mh_prior_or_cond(orc_handle* o, int side, long e, uint32_t t, int use_prior, double num1, double den, double* out) {
  if (use_prior) { *out = prior_draw(o, side, e, t); return; }
  double mu, var;
  if (o->cfg.prior == PRIOR_EXPONENTIAL) {
    double la = (side ? AR(ID_LAMBDA_E) : AR(ID_LAMBDA_P))[e];
    mu = (num1 - la) / den; var = 1.0 / den;
  } else {
    double mp = (side ? AR(ID_MU_E) : AR(ID_MU_P))[e], sg = (side ? AR(ID_SIGSQ_E) : AR(ID_SIGSQ_P))[e];
    double den2 = den + 1.0 / sg;
    mu = (num1 + mp / sg) / den2; var = 1.0 / den2;
  }
  orc_stream s = ST(o, side ? V_E : V_P, (uint32_t)e, t);
  *out = orc_rtnorm0(&s, mu, sqrt(var));
}
/* P sweep: rows are independent, so each row k runs its N sequential updates on its own maintained Mhat[k, .] */
static void sample_P_seq(orc_handle* o, uint32_t t) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  const int normal = o->cfg.likelihood == LIK_NORMAL;
  const int mhstep = o->cfg.MH && o->converged && !normal;       /* true accept/reject (MH_Pn_poisson :206-247) */
  double* P = AR(ID_P);
  double* acc = o->cfg.MH ? ensure(o, ID_ACC_P) : NULL;
  const int nth = o->cfg.nthreads > 0 ? o->cfg.nthreads : 1;
  double* tmp = (double*)malloc(8 * (size_t)G * 7 * (size_t)nth);
  int* allzero = (int*)malloc(sizeof(int) * N);
  for (long n = 0; n < N; ++n) {                                    /* all(E[n,] == 0): E is fixed during the P sweep */
    allzero[n] = 1;
    for (long g = 0; g < G; ++g) if (AR(ID_E)[n + N * g] != 0.0) { allzero[n] = 0; break; }
  }
#pragma omp parallel for schedule(static) num_threads(nth)
  for (long k = 0; k < K; ++k) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double* row = tmp + (size_t)tid * 7 * G; double* x1 = row + G; double* x2 = x1 + G; double* y = x2 + G;
    for (long g = 0; g < G; ++g) row[g] = mhat_cell(o, k, g, -1, NULL, 0);         /* fresh at the start of the sweep */
    for (long n = 0; n < N; ++n) {
      const long e = k + K * n;
      const double a_n = AR(ID_A)[n];
      if (a_n == 0.0) { P[e] = prior_draw(o, 0, e, t); continue; }                /* sample_Pn :12; term of Mhat is 0 */
      const double pa = P[e] * a_n;
      const int fresh = o->fresh_mhat;
      if (fresh) for (long g = 0; g < G; ++g) row[g] = mhat_cell(o, k, g, -1, NULL, 0);   /* as R: from scratch at every factor */
      double num1 = 0.0, den = 0.0;
      if (!allzero[n]) {
        for (long g = 0; g < G; ++g) {
          double en = AR(ID_E)[n + N * g];
          double mno = fresh ? mhat_cell(o, k, g, n, NULL, 0) : row[g] - pa * en;  /* Mhat_no_n */
          double V = normal ? AR(ID_SIGMASQ)[g] : row[g];                          /* sigmasq_kg :137-147 */
          double rV = 1.0 / V;                                                     /* one reciprocal serves both sums (stream spec) */
          x1[g] = en * (((double)o->M[k + K * g] - mno) * rV);                     /* :155-161 */
          x2[g] = (a_n * (en * en)) * rV;                                          /* :163-169 */
        }
        num1 = canon_rowsum(x1, G); den = canon_rowsum(x2, G);
      }
      double pr;
      mh_prior_or_cond(o, 0, e, t, allzero[n], num1, den, &pr);
      const double pra = pr * a_n;
      int take = 1;
      if (!mhstep) { if (acc) acc[e] = 1.0; }                                      /* MH_Pn_poisson :201-204 / plain Gibbs */
      else {
        for (long g = 0; g < G; ++g) {
          double en = AR(ID_E)[n + N * g];
          double m0 = row[g], m1 = fresh ? mhat_cell(o, k, g, -1, &pr, n) : (m0 - pa * en) + pra * en;
          int32_t m = o->M[k + K * g];
          y[g] = dpois_log(m, m1);                                       /* loglik_poisson_new :216-218 */
          y[G + g] = dnorm_log((double)m, m0, m1 < 1.0 ? 1.0 : m1);     /* loglik_normal_old :219-224 */
          y[2 * G + g] = dpois_log(m, m0);                               /* loglik_poisson_old :213-215 */
          y[3 * G + g] = dnorm_log((double)m, m1, m0 < 1.0 ? 1.0 : m0); /* loglik_normal_new :225-231 */
        }
        double A_ = canon_rowsum(y, G), B_ = canon_rowsum(y + G, G), C_ = canon_rowsum(y + 2 * G, G), D_ = canon_rowsum(y + 3 * G, G);
        double ratio = orc_exp((A_ + B_) - (C_ + D_));
        if (ratio > 1.0) ratio = 1.0;                                    /* pmin(accept_ratio, 1) :239 */
        acc[e] = ratio;
        orc_stream s = ST(o, V_MHU_P, (uint32_t)e, t);
        take = orc_runif(&s) < ratio;
      }
      if (take) {
        P[e] = pr;
        for (long g = 0; g < G; ++g) { double en = AR(ID_E)[n + N * g]; row[g] = (row[g] - pa * en) + pra * en; }
      }
    }
  }
  free(tmp); free(allzero);
}
/* E sweep: columns are independent; column g keeps Mhat[., g] */
static void sample_E_seq(orc_handle* o, uint32_t t) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  const int normal = o->cfg.likelihood == LIK_NORMAL;
  const int mhstep = o->cfg.MH && o->converged && !normal;
  double* E = AR(ID_E);
  double* acc = o->cfg.MH ? ensure(o, ID_ACC_E) : NULL;
  int* allzero = (int*)malloc(sizeof(int) * N);
  for (long n = 0; n < N; ++n) {                                    /* all(P[,n] == 0): P is fixed during the E sweep */
    allzero[n] = 1;
    for (long k = 0; k < K; ++k) if (AR(ID_P)[k + K * n] != 0.0) { allzero[n] = 0; break; }
  }
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g) {
    double mhc[K], x1[K], x2[K], y[4 * K];
    for (long k = 0; k < K; ++k) mhc[k] = mhat_cell(o, k, g, -1, NULL, 0);
    const double sg = normal ? AR(ID_SIGMASQ)[g] : 1.0;
    for (long n = 0; n < N; ++n) {
      const long e = n + N * g;
      const double a_n = AR(ID_A)[n];
      if (a_n == 0.0) { E[e] = prior_draw(o, 1, e, t); continue; }     /* sample_En :12 */
      const double eold = E[e];
      const int fresh = o->fresh_mhat;
      if (fresh) for (long k = 0; k < K; ++k) mhc[k] = mhat_cell(o, k, g, -1, NULL, 0);
      double num1 = 0.0, den = 0.0;
      if (!allzero[n]) {
        for (long k = 0; k < K; ++k) {
          double pn = AR(ID_P)[k + K * n];
          double mno = fresh ? mhat_cell(o, k, g, n, NULL, 0) : mhc[k] - (pn * a_n) * eold;
          double V = normal ? sg : mhc[k];
          double rV = 1.0 / V;
          x1[k] = pn * (((double)o->M[k + K * g] - mno) * rV);
          x2[k] = (a_n * (pn * pn)) * rV;
        }
        num1 = orc_canon_sum(x1, K, 1, 64); den = orc_canon_sum(x2, K, 1, 64);
      }
      double pr;
      mh_prior_or_cond(o, 1, e, t, allzero[n], num1, den, &pr);
      int take = 1;
      if (!mhstep) { if (acc) acc[e] = 1.0; }
      else {
        for (long k = 0; k < K; ++k) {
          double pna = AR(ID_P)[k + K * n] * a_n;
          double m0 = mhc[k], m1 = fresh ? mhat_cell_e(o, k, g, -1, pr, n) : (m0 - pna * eold) + pna * pr;
          int32_t m = o->M[k + K * g];
          y[k] = dpois_log(m, m1);
          y[K + k] = dnorm_log((double)m, m0, m1 < 1.0 ? 1.0 : m1);
          y[2 * K + k] = dpois_log(m, m0);
          y[3 * K + k] = dnorm_log((double)m, m1, m0 < 1.0 ? 1.0 : m0);
        }
        double A_ = orc_canon_sum(y, K, 1, 64), B_ = orc_canon_sum(y + K, K, 1, 64), C_ = orc_canon_sum(y + 2 * K, K, 1, 64), D_ = orc_canon_sum(y + 3 * K, K, 1, 64);
        double ratio = orc_exp((A_ + B_) - (C_ + D_));
        if (ratio > 1.0) ratio = 1.0;
        acc[e] = ratio;
        orc_stream s = ST(o, V_MHU_E, (uint32_t)e, t);
        take = orc_runif(&s) < ratio;
      }
      if (take) {
        E[e] = pr;
        for (long k = 0; k < K; ++k) { double pna = AR(ID_P)[k + K * n] * a_n; mhc[k] = (mhc[k] - pna * eold) + pna * pr; }
      }
    }
  }
  free(allzero);
}
static void sample_P_mh(orc_handle* o, uint32_t t) { sample_P_seq(o, t); }
static void sample_E_mh(orc_handle* o, uint32_t t) { sample_E_seq(o, t); }
static void sample_P_normal(orc_handle* o, uint32_t t) { sample_P_seq(o, t); }
static void sample_E_normal(orc_handle* o, uint32_t t) { sample_E_seq(o, t); }
/* per-cell metric terms for the models without Z (fresh Mhat) */
static void metrics_cells_mh(orc_handle* o) {
  const long K = o->cfg.K, G = o->cfg.G;
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g) {
    double a[K], b[K], c[K];
    for (long k = 0; k < K; ++k) cell_terms(o, o->M[k + K * g], mhat_cell(o, k, g, -1, NULL, 0), &a[k], &b[k], &c[k]);
    o->colsse[g] = orc_canon_sum(a, K, 1, 64);
    o->colll[g] = orc_canon_sum(b, K, 1, 64);
    o->colkl[g] = orc_canon_sum(c, K, 1, 64);
  }
}

/* ================= Normal likelihood: sample_sigmasq R/sample_params.R:275-286 and its metrics ================= */
static void sample_sigmasq(orc_handle* o, uint32_t t) {
  const long K = o->cfg.K, G = o->cfg.G;
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g) {
    double r2[K];
    for (long k = 0; k < K; ++k) { double r = (double)o->M[k + K * g] - mhat_cell(o, k, g, -1, NULL, 0); r2[k] = r * r; }
    double ss = orc_canon_sum(r2, K, 1, 64);
    orc_stream s = ST(o, V_SIGMASQ, (uint32_t)g, t);
    AR(ID_SIGMASQ)[g] = orc_rinvgamma(&s, HY(ID_ALPHA_S, g) + (double)K / 2.0, HY(ID_BETA_S, g) + 0.5 * ss);
  }
}
/* per-cell metric terms, Normal log-likelihood (get_loglik_ normal branch R/utils.R:72-97) */
static void metrics_cells_normal(orc_handle* o) {
  const long K = o->cfg.K, G = o->cfg.G;
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long g = 0; g < G; ++g) {
    double a[K], b[K], c[K];
    double sg = AR(ID_SIGMASQ)[g];
    for (long k = 0; k < K; ++k) {
      double mh = mhat_cell(o, k, g, -1, NULL, 0), dummy;
      cell_terms(o, o->M[k + K * g], mh, &a[k], &dummy, &c[k]);
      b[k] = dnorm_log_fwd((double)o->M[k + K * g], mh, sg);
    }
    o->colsse[g] = orc_canon_sum(a, K, 1, 64);
    o->colll[g] = orc_canon_sum(b, K, 1, 64);
    o->colkl[g] = orc_canon_sum(c, K, 1, 64);
  }
}

/* ---- metrics row: compute_metrics_ R/utils.R:412-455, update_sample_metrics_ :339-348 ---- */
static void metrics_row(orc_handle* o, double* row) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  double sse = orc_canon_sum(o->colsse, G, 1, 1024);
  double ll = orc_canon_sum(o->colll, G, 1, 1024);
  double kl = orc_canon_sum(o->colkl, G, 1, 1024);
  /* log prior of P and E under the current prior parameters (R/utils.R:132-175) */
  double lpP = 0.0;
  for (long n = 0; n < N; ++n) {
    for (long k = 0; k < K; ++k) o->lpP[k + K * n] = prior_logdens(o, 0, k + K * n, AR(ID_P)[k + K * n]);
    lpP = lpP + orc_canon_sum(o->lpP + K * n, K, 1, 64);
  }
  long NE = N * G, nblk = (NE + 255) / 256;
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long e = 0; e < NE; ++e) o->lpE[e] = prior_logdens(o, 1, e, AR(ID_E)[e]);
  double* part = (double*)malloc(8 * nblk);
  for (long b = 0; b < nblk; ++b) {
    long len = NE - b * 256 < 256 ? NE - b * 256 : 256;
    part[b] = orc_canon_sum(o->lpE + b * 256, len, 1, 256);
  }
  double lpE = orc_canon_sum(part, nblk, 1, 1024);
  free(part);
  double sumA = 0.0;
  for (long n = 0; n < N; ++n) sumA = sumA + AR(ID_A)[n];
  double n_params = sumA * (double)(G + K);
  row[0] = (double)o->iter;
  row[1] = sqrt(sse / ((double)K * (double)G));
  row[2] = kl;
  row[3] = ll;
  row[4] = ll + (lpP + lpE);
  row[5] = n_params;
  row[6] = -2.0 * ll + n_params * orc_log((double)G);
  row[7] = sumA;
  row[8] = temp_at(o, o->iter);
  row[9] = NAN; row[10] = NAN;
  if (o->cfg.MH) {
    /* mean of the acceptance matrices over active factors (R/utils.R:444-452) */
    double sp = 0.0;
    for (long n = 0; n < N; ++n) if (AR(ID_A)[n] == 1.0) sp = sp + orc_canon_sum(AR(ID_ACC_P) + K * n, K, 1, 64);
    double* part2 = (double*)malloc(8 * nblk);
    for (long b = 0; b < nblk; ++b) {
      double tmpv[256];
      long len = NE - b * 256 < 256 ? NE - b * 256 : 256;
      for (long i = 0; i < len; ++i) { long e = b * 256 + i; tmpv[i] = (AR(ID_A)[e % N] == 1.0) ? AR(ID_ACC_E)[e] : 0.0; }
      part2[b] = orc_canon_sum(tmpv, len, 1, 256);
    }
    double se = orc_canon_sum(part2, nblk, 1, 1024);
    free(part2);
    row[9] = sp / ((double)K * sumA);
    row[10] = se / ((double)G * sumA);
  }
}

/* ---- constructor part: init_prior_params_ R/sample_priors.R:15-141 then
 * sample_params(from_prior=TRUE) R/bayesNMF_sampler.R:241 ; iteration index 0 is used for the
 * hyper-prior draws, 1 for the prior draws of P,E,(R,A),Z (state$iter starts at 1). ---- */
static int col_has_nan(const double* x, long n, long stride) {
  for (long i = 0; i < n; ++i) if (x[i * stride] != x[i * stride]) return 1;
  return 0;
}
int orc_init(orc_handle* o, double* metrics_row1) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  const int pr = o->cfg.prior;
  /* prior params: redraw column n (P side) / row n (E side) when any entry is missing (NaN) */
  struct ispec { int id, var, side; int hs, hr; } spec[4]; int nspec = 0;
  if (pr == PRIOR_GAMMA) {
    spec[nspec++] = (struct ispec){ID_BETA_P, V_BETA_P, 0, ID_HA_P, ID_HB_P};
    spec[nspec++] = (struct ispec){ID_ALPHA_P, V_ALPHA_P, 0, ID_HC_P, ID_HD_P};
    spec[nspec++] = (struct ispec){ID_BETA_E, V_BETA_E, 1, ID_HA_E, ID_HB_E};
    spec[nspec++] = (struct ispec){ID_ALPHA_E, V_ALPHA_E, 1, ID_HC_E, ID_HD_E};
  } else if (pr == PRIOR_EXPONENTIAL) {
    spec[nspec++] = (struct ispec){ID_LAMBDA_P, V_LAMBDA_P, 0, ID_HA_P, ID_HB_P};
    spec[nspec++] = (struct ispec){ID_LAMBDA_E, V_LAMBDA_E, 1, ID_HA_E, ID_HB_E};
  }
  if (pr == PRIOR_TRUNCNORMAL) {
    /* Mu ~ N(M, sqrt(S)), Sigmasq ~ InvGamma(A, B)  (R/sample_priors.R:32-61) */
    int ids[4] = {ID_MU_P, ID_SIGSQ_P, ID_MU_E, ID_SIGSQ_E};
    for (int i = 0; i < 4; ++i) {
      int side = i >= 2, is_mu = (i % 2) == 0;
      double* x = ensure(o, ids[i]);
      for (long n = 0; n < N; ++n) {
        int redraw = side ? col_has_nan(x + n, G, N) : col_has_nan(x + K * n, K, 1);
        if (!redraw) continue;
        long cnt = side ? G : K;
        for (long j = 0; j < cnt; ++j) {
          long e = side ? n + N * j : j + K * n;
          if (is_mu) { orc_stream s = ST(o, side ? V_MU_E : V_MU_P, (uint32_t)e, 0);
            x[e] = HY(side ? ID_HM_E : ID_HM_P, e) + sqrt(HY(side ? ID_HS_E : ID_HS_P, e)) * orc_rnorm_std(&s); }
          else { orc_stream s = ST(o, side ? V_SIGSQ_E : V_SIGSQ_P, (uint32_t)e, 0);
            x[e] = orc_rinvgamma(&s, HY(side ? ID_HA_E : ID_HA_P, e), HY(side ? ID_HB_E : ID_HB_P, e)); }
        }
      }
    }
  }
  for (int i = 0; i < nspec; ++i) {
    double* x = ensure(o, spec[i].id);
    for (long n = 0; n < N; ++n) {
      int redraw = spec[i].side ? col_has_nan(x + n, G, N) : col_has_nan(x + K * n, K, 1);
      if (!redraw) continue;
      long cnt = spec[i].side ? G : K;
      for (long j = 0; j < cnt; ++j) {
        long e = spec[i].side ? n + N * j : j + K * n;
        orc_stream s = ST(o, (uint32_t)spec[i].var, (uint32_t)e, 0);
        x[e] = orc_rgamma(&s, HY(spec[i].hs, e), HY(spec[i].hr, e));   /* rgamma(shape, rate) :72-127 */
      }
    }
  }
  o->iter = 1;
  /* params: skip = names(init_params) — whole parameter kept when supplied */
  int haveP = o->a[ID_P].set, haveE = o->a[ID_E].set, haveA = o->a[ID_A].set;
  ensure(o, ID_P); ensure(o, ID_E); ensure(o, ID_A);
  if (!haveA) for (long n = 0; n < N; ++n) AR(ID_A)[n] = 1.0;
  if (!haveP) sample_P_poisson(o, 1, 1);
  if (!haveE) sample_E_poisson(o, 1, 1);
  if (o->cfg.MH) { double* ap = ensure(o, ID_ACC_P); double* ae_ = ensure(o, ID_ACC_E); (void)ap; (void)ae_; }
  if (!haveA && o->cfg.learning_rank) { sample_R(o, 1, 1); sample_A(o, 1, 1); }
  else if (!o->a[ID_R].set) o->R = (int)N;
  if (o->cfg.likelihood == LIK_NORMAL) {
    if (!o->a[ID_ALPHA_S].p) { double three = 3.0; orc_set_array(o, ID_ALPHA_S, &three, 1); }
    if (!o->a[ID_BETA_S].p) { double three = 3.0; orc_set_array(o, ID_BETA_S, &three, 1); }
    if (!o->a[ID_SIGMASQ].set) { ensure(o, ID_SIGMASQ); sample_sigmasq(o, 1); }
    metrics_cells_normal(o);
  } else if (o->cfg.MH) metrics_cells_mh(o); else sample_Z_and_metrics(o, 1);
  if (metrics_row1) metrics_row(o, metrics_row1);
  return 0;
}

/* one iteration of the loop body R/bayesNMF_sampler.R:273-285 */
static void sweep(orc_handle* o) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  o->iter += 1;
  uint32_t t = (uint32_t)o->iter;
  for (long e = 0; e < K * N; ++e) hyper_elem(o, 0, e, t);
#pragma omp parallel for schedule(static) num_threads(o->cfg.nthreads)
  for (long e = 0; e < N * G; ++e) hyper_elem(o, 1, e, t);
  if (o->cfg.likelihood == LIK_NORMAL) { sample_P_normal(o, t); sample_E_normal(o, t); }
  else if (o->cfg.MH) { sample_P_mh(o, t); sample_E_mh(o, t); }
  else { sample_P_poisson(o, t, 0); sample_E_poisson(o, t, 0); }
  if (o->cfg.learning_rank) { sample_R(o, t, 0); sample_A(o, t, 0); }
  if (o->cfg.likelihood == LIK_NORMAL) { sample_sigmasq(o, t); metrics_cells_normal(o); }
  else if (o->cfg.MH) metrics_cells_mh(o); else sample_Z_and_metrics(o, t);
}
int orc_run(orc_handle* o, int n_iter, int converged, double* metrics /* n_iter x ORC_NMETRIC row-major */) {
  o->converged = converged;
  for (int i = 0; i < n_iter; ++i) {
    sweep(o);
    if (metrics) metrics_row(o, metrics + (size_t)i * ORC_NMETRIC);
  }
  return 0;
}
const char* orc_last_error(orc_handle* o) { return o->err; }

/* ---- single-step hooks for the law tests (tests/test_oracle_laws.py): one conditional of the sweep on the
 * current state with the streams of iteration t; orc_set_M swaps the data (Geweke joint-distribution test) ---- */
/* The stream spec maintains Mhat incrementally in the rank sweep and the MH / Normal sweeps (O(N K G)); the R code recomputes
 * P diag(A) E by BLAS products at every factor (R/sample_params.R:101-166, R/sample_Pn.R:132-187, R/sample_En.R).  With this switch on the
 * oracle does what R does — every Mhat, Mhat without factor n and Mhat with the proposed value from scratch, in factor order —
 * so that tests/test_oracle.py can tie the maintained form to the R semantics without the GPU (agreement to rounding). */
int orc_set_fresh_mhat(orc_handle* o, int on) { o->fresh_mhat = on; return 0; }
int orc_set_M(orc_handle* o, const int32_t* M) {
  memcpy(o->M, M, sizeof(int32_t) * (size_t)o->cfg.K * o->cfg.G);
  return 0;
}
enum { STEP_HYPER = 0, STEP_P = 1, STEP_E = 2, STEP_R = 3, STEP_A = 4, STEP_Z = 5, STEP_SIGMASQ = 6 };
int orc_t_step(orc_handle* o, int what, uint32_t t, int converged) {
  const long K = o->cfg.K, G = o->cfg.G, N = o->cfg.N;
  const int normal = o->cfg.likelihood == LIK_NORMAL, mh = o->cfg.MH;
  o->converged = converged;
  switch (what) {
    case STEP_HYPER:
      for (long e = 0; e < K * N; ++e) hyper_elem(o, 0, e, t);
      for (long e = 0; e < N * G; ++e) hyper_elem(o, 1, e, t);
      return 0;
    case STEP_P: if (normal) sample_P_normal(o, t); else if (mh) sample_P_mh(o, t); else sample_P_poisson(o, t, 0); return 0;
    case STEP_E: if (normal) sample_E_normal(o, t); else if (mh) sample_E_mh(o, t); else sample_E_poisson(o, t, 0); return 0;
    case STEP_R: sample_R(o, t, 0); return 0;
    case STEP_A: sample_A(o, t, 0); return 0;
    case STEP_Z: if (normal || mh) return -1; sample_Z_and_metrics(o, t); return 0;
    case STEP_SIGMASQ: if (!normal) return -1; sample_sigmasq(o, t); return 0;
    default: return -1;
  }
}

/* ---- unit-test exports ---- */
void orc_t_philox7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc_philox4x32_r(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out, 7);
}
void orc_t_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
double orc_t_log(double x) { return orc_log(x); }
double orc_t_exp(double x) { return orc_exp(x); }
double orc_t_lgamma(double x) { return orc_lgamma(x); }
double orc_t_digamma(double x) { return orc_digamma(x); }
double orc_t_qnorm(double p) { return orc_qnorm(p); }
double orc_t_log_pnorm(double z) { return orc_log_pnorm(z); }
double orc_t_u52(uint32_t a, uint32_t b) { return orc_u52(a, b); }
double orc_t_canon_sum(const double* x, long L, long stride, int W) { return orc_canon_sum(x, L, stride, W); }
/* vectorised sampler probes: element e of out uses stream (var, elem0+e, iter) */
void orc_t_rgamma(uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter,
                  const double* shape, const double* rate, double* out, long n) {
  for (long e = 0; e < n; ++e) { orc_stream s = orc_stream_make(seed, chain, var, elem0 + (uint32_t)e, iter); out[e] = orc_rgamma(&s, shape[e], rate[e]); }
}
void orc_t_rtnorm0(uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter,
                   const double* mu, const double* sd, double* out, long n) {
  for (long e = 0; e < n; ++e) { orc_stream s = orc_stream_make(seed, chain, var, elem0 + (uint32_t)e, iter); out[e] = orc_rtnorm0(&s, mu[e], sd[e]); }
}
void orc_t_rnorm(uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter, double* out, long n) {
  for (long e = 0; e < n; ++e) { orc_stream s = orc_stream_make(seed, chain, var, elem0 + (uint32_t)e, iter); out[e] = orc_rnorm_std(&s); }
}
void orc_t_ralpha(uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter,
                  const double* c, const double* tau, const double* xprev, double* out, int32_t* attempts, long n) {
  for (long e = 0; e < n; ++e) { orc_stream s = orc_stream_make(seed, chain, var, elem0 + (uint32_t)e, iter); int na; out[e] = orc_ralpha(&s, c[e], tau[e], xprev[e], &na); if (attempts) attempts[e] = na; }
}
void orc_t_ralpha_fast(uint64_t seed, uint32_t chain, uint32_t var, uint32_t elem0, uint32_t iter,
                       const double* c, const double* tau, const double* xprev, double* out, int32_t* attempts, long n) {
  for (long e = 0; e < n; ++e) { orc_stream s = orc_stream_make(seed, chain, var, elem0 + (uint32_t)e, iter); int na; out[e] = orc_ralpha_fast(&s, c[e], tau[e], xprev[e], &na); if (attempts) attempts[e] = na; }
}
void orc_t_alpha_h(double x, double c, double tau, double* h, double* hp) { orc_alpha_h(x, c, tau, h, hp); }
