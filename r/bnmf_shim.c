/* r/bnmf_shim.c — thin `.Call` shim between R and the C ABI of libbnmf.so (include/bnmf.h).
 *
 * R API only (no Rcpp, no CUDA-compat headers); logic-free marshalling: every entry point is one C-ABI
 * call.  Must be called from the R main thread; never calls back into R; inputs are read-only SEXPs
 * (copied to the device by the library); outputs are freshly allocated and PROTECTed here; errors become
 * Rf_error() with bnmf_last_error().  Build inside an R package: src/bnmf_shim.c + PKG_LIBS = -lbnmf.
 * (Not compiled in this repository's container: there is no R installation / Rinternals.h.)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include "bnmf.h"

static void chk(int rc) { if (rc != 0) Rf_error("%s", bnmf_last_error()); }

static void handle_finalizer(SEXP ptr) {
  bnmf_handle* h = (bnmf_handle*)R_ExternalPtrAddr(ptr);
  if (h) { bnmf_destroy(h); R_ClearExternalPtr(ptr); }
}
static bnmf_handle* get_handle(SEXP ptr) {
  bnmf_handle* h = (bnmf_handle*)R_ExternalPtrAddr(ptr);
  if (!h) Rf_error("bnmf: handle was destroyed");
  return h;
}

/* C_bnmf_create(data (integer K x G), dims c(K,G,N), spec c(likelihood, prior, MH, learning_rank,
 *               rank_method, save_Z, window), temperature (double), seed (double), chain_id, device) */
SEXP C_bnmf_create(SEXP data, SEXP dims, SEXP spec, SEXP temperature, SEXP seed, SEXP chain_id, SEXP device) {
  bnmf_config cfg;
  const int* d = INTEGER(dims); const int* s = INTEGER(spec);
  cfg.K = d[0]; cfg.G = d[1]; cfg.N = d[2];
  cfg.likelihood = s[0]; cfg.prior = s[1]; cfg.MH = s[2]; cfg.learning_rank = s[3];
  cfg.rank_method = s[4]; cfg.save_Z = s[5]; cfg.window = s[6];
  cfg.seed = (uint64_t)REAL(seed)[0]; cfg.chain_id = (uint32_t)INTEGER(chain_id)[0]; cfg.device = INTEGER(device)[0];
  cfg.temperature = REAL(temperature); cfg.n_temperature = (int64_t)XLENGTH(temperature);
  bnmf_handle* h = NULL;
  chk(bnmf_create(&cfg, INTEGER(data), &h));
  SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ptr, handle_finalizer, TRUE);
  UNPROTECT(1);
  return ptr;
}
SEXP C_bnmf_set_array(SEXP ptr, SEXP id, SEXP value) {
  chk(bnmf_set_array(get_handle(ptr), INTEGER(id)[0], REAL(value), (size_t)XLENGTH(value)));
  return R_NilValue;
}
SEXP C_bnmf_get_array(SEXP ptr, SEXP id, SEXP n) {
  const R_xlen_t len = (R_xlen_t)REAL(n)[0];
  SEXP out = PROTECT(Rf_allocVector(REALSXP, len));
  chk(bnmf_get_array(get_handle(ptr), INTEGER(id)[0], REAL(out), (size_t)len));
  UNPROTECT(1);
  return out;
}
SEXP C_bnmf_init(SEXP ptr) {
  SEXP row = PROTECT(Rf_allocVector(REALSXP, BNMF_NMETRIC));
  chk(bnmf_init(get_handle(ptr), REAL(row)));
  UNPROTECT(1);
  return row;
}
/* returns a (BNMF_NMETRIC x n_iter) double matrix (column-major: one column per iteration) */
SEXP C_bnmf_run(SEXP ptr, SEXP n_iter, SEXP converged) {
  const int n = INTEGER(n_iter)[0];
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, BNMF_NMETRIC, n));
  chk(bnmf_run(get_handle(ptr), n, LOGICAL(converged)[0], REAL(out)));
  UNPROTECT(1);
  return out;
}
/* returns a (len x last_n) double matrix: one recorded sample per column, oldest first */
SEXP C_bnmf_window(SEXP ptr, SEXP id, SEXP last_n, SEXP len) {
  const int n = INTEGER(last_n)[0]; const R_xlen_t l = (R_xlen_t)REAL(len)[0];
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)l, n));
  chk(bnmf_window(get_handle(ptr), INTEGER(id)[0], n, REAL(out)));
  UNPROTECT(1);
  return out;
}
SEXP C_bnmf_get_iter(SEXP ptr) {
  int it = 0;
  chk(bnmf_get_iter(get_handle(ptr), &it));
  return Rf_ScalarInteger(it);
}
static SEXP named_list(int n, const char** names) {
  SEXP out = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
  for (int i = 0; i < n; ++i) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
  Rf_setAttrib(out, R_NamesSymbol, nm);
  UNPROTECT(2);
  return out;
}
/* get_MAP_ on the device (R/utils.R:194-288): C_bnmf_map(ptr, last_n, credible_interval (<= 0: no bounds), dims c(K,G,N)) ->
 * list(P K x N, E N x G, A, top_A 5 x N (row i = i-th most frequent pattern), P_lower, P_upper, E_lower, E_upper, used (logical),
 *      n_used, n_patterns, top_counts, rmse, kl) */
SEXP C_bnmf_map(SEXP ptr, SEXP last_n, SEXP ci, SEXP dims) {
  const int n = INTEGER(last_n)[0]; const int* d = INTEGER(dims); const int K = d[0], G = d[1], N = d[2];
  const double c = REAL(ci)[0]; const int want = c > 0.0;
  static const char* nms[] = {"P", "E", "A", "top_A", "P_lower", "P_upper", "E_lower", "E_upper", "used", "n_used", "n_patterns", "top_counts", "rmse", "kl"};
  SEXP out = PROTECT(named_list(14, nms));
  SEXP P = PROTECT(Rf_allocMatrix(REALSXP, K, N)), E = PROTECT(Rf_allocMatrix(REALSXP, N, G)), A = PROTECT(Rf_allocMatrix(REALSXP, 1, N));
  SEXP top = PROTECT(Rf_allocVector(REALSXP, 5 * (R_xlen_t)N)), used = PROTECT(Rf_allocVector(INTSXP, n));
  SEXP Pl = PROTECT(want ? Rf_allocMatrix(REALSXP, K, N) : R_NilValue), Pu = PROTECT(want ? Rf_allocMatrix(REALSXP, K, N) : R_NilValue);
  SEXP El = PROTECT(want ? Rf_allocMatrix(REALSXP, N, G) : R_NilValue), Eu = PROTECT(want ? Rf_allocMatrix(REALSXP, N, G) : R_NilValue);
  bnmf_map_info info;
  chk(bnmf_map(get_handle(ptr), n, c, REAL(P), REAL(E), REAL(A), REAL(top), want ? REAL(Pl) : NULL, want ? REAL(Pu) : NULL,
               want ? REAL(El) : NULL, want ? REAL(Eu) : NULL, INTEGER(used), &info));
  SEXP topm = PROTECT(Rf_allocMatrix(REALSXP, 5, N));                     /* row-major 5 x N -> R matrix */
  for (int i = 0; i < 5; ++i) for (int j = 0; j < N; ++j) REAL(topm)[i + 5 * j] = REAL(top)[(size_t)i * N + j];
  SEXP usedl = PROTECT(Rf_allocVector(LGLSXP, n));
  for (int i = 0; i < n; ++i) LOGICAL(usedl)[i] = INTEGER(used)[i] != 0;
  SEXP tc = PROTECT(Rf_allocVector(INTSXP, 5));
  for (int i = 0; i < 5; ++i) INTEGER(tc)[i] = info.top_counts[i];
  SET_VECTOR_ELT(out, 0, P); SET_VECTOR_ELT(out, 1, E); SET_VECTOR_ELT(out, 2, A); SET_VECTOR_ELT(out, 3, topm);
  SET_VECTOR_ELT(out, 4, Pl); SET_VECTOR_ELT(out, 5, Pu); SET_VECTOR_ELT(out, 6, El); SET_VECTOR_ELT(out, 7, Eu);
  SET_VECTOR_ELT(out, 8, usedl); SET_VECTOR_ELT(out, 9, Rf_ScalarInteger(info.n_used)); SET_VECTOR_ELT(out, 10, Rf_ScalarInteger(info.n_patterns));
  SET_VECTOR_ELT(out, 11, tc); SET_VECTOR_ELT(out, 12, Rf_ScalarReal(info.rmse)); SET_VECTOR_ELT(out, 13, Rf_ScalarReal(info.kl));
  UNPROTECT(13);
  return out;
}
/* convergence control / state marshalling: cc_int = c(MAP_over, MAP_every, Ninarow_nochange, Ninarow_nobest, miniters, maxiters,
 * metric (0 loglikelihood, 1 logposterior, 2 RMSE, 3 KL, 4 BIC)); state = c(converged, why, best_iter, inarow_na, inarow_no_change,
 * inarow_no_best, have_prev, n_checks, prev_MAP_metric, best_MAP_metric, prev_percent_change) */
static void cc_in(SEXP cc_int, SEXP tol, bnmf_convergence_control* cc) {
  const int* c = INTEGER(cc_int);
  cc->MAP_over = c[0]; cc->MAP_every = c[1]; cc->Ninarow_nochange = c[2]; cc->Ninarow_nobest = c[3]; cc->miniters = c[4];
  cc->maxiters = c[5]; cc->metric = c[6]; cc->_pad = 0; cc->tol = REAL(tol)[0];
}
static void st_in(SEXP state, bnmf_convergence_state* st) {
  const double* v = REAL(state);
  st->converged = (int)v[0]; st->why = (int)v[1]; st->best_iter = (int)v[2]; st->inarow_na = (int)v[3]; st->inarow_no_change = (int)v[4];
  st->inarow_no_best = (int)v[5]; st->have_prev = (int)v[6]; st->n_checks = (int)v[7];
  st->prev_MAP_metric = v[8]; st->best_MAP_metric = v[9]; st->prev_percent_change = v[10];
}
static SEXP loop_out(const double* met, int n_rows, const double* maps, int n_checks, const bnmf_convergence_state* st) {
  static const char* nms[] = {"metrics", "map_rows", "state"};
  SEXP out = PROTECT(named_list(3, nms));
  SEXP m = PROTECT(Rf_allocMatrix(REALSXP, BNMF_NMETRIC, n_rows)), r = PROTECT(Rf_allocMatrix(REALSXP, BNMF_NMAPROW, n_checks));
  SEXP s = PROTECT(Rf_allocVector(REALSXP, 11));
  for (R_xlen_t i = 0; i < (R_xlen_t)BNMF_NMETRIC * n_rows; ++i) REAL(m)[i] = met[i];      /* one column per iteration */
  for (R_xlen_t i = 0; i < (R_xlen_t)BNMF_NMAPROW * n_checks; ++i) REAL(r)[i] = maps[i];   /* one column per MAP check */
  double* v = REAL(s);
  v[0] = st->converged; v[1] = st->why; v[2] = st->best_iter; v[3] = st->inarow_na; v[4] = st->inarow_no_change; v[5] = st->inarow_no_best;
  v[6] = st->have_prev; v[7] = st->n_checks; v[8] = st->prev_MAP_metric; v[9] = st->best_MAP_metric; v[10] = st->prev_percent_change;
  SET_VECTOR_ELT(out, 0, m); SET_VECTOR_ELT(out, 1, r); SET_VECTOR_ELT(out, 2, s);
  UNPROTECT(4);
  return out;
}
/* the warm-up loop to convergence in one call (R/bayesNMF_sampler.R:268-330) */
SEXP C_bnmf_run_until(SEXP ptr, SEXP cc_int, SEXP tol, SEXP state) {
  bnmf_convergence_control cc; bnmf_convergence_state st;
  cc_in(cc_int, tol, &cc); st_in(state, &st);
  int it = 0;
  chk(bnmf_get_iter(get_handle(ptr), &it));
  const int cap_rows = (cc.maxiters > it ? cc.maxiters - it : 0) + 1, cap_checks = cap_rows / cc.MAP_every + 2;
  double* met = (double*)R_alloc((size_t)cap_rows * BNMF_NMETRIC, sizeof(double));
  double* maps = (double*)R_alloc((size_t)cap_checks * BNMF_NMAPROW, sizeof(double));
  int n_rows = 0, n_checks = 0;
  chk(bnmf_run_until(get_handle(ptr), &cc, &st, met, cap_rows, &n_rows, maps, cap_checks, &n_checks));
  return loop_out(met, n_rows, maps, n_checks, &st);
}
/* the MH models' post-warm-up iterations in one call (R/bayesNMF_sampler.R:332-384) */
SEXP C_bnmf_run_post_warmup(SEXP ptr, SEXP cc_int, SEXP tol, SEXP state, SEXP post_warmup) {
  bnmf_convergence_control cc; bnmf_convergence_state st;
  cc_in(cc_int, tol, &cc); st_in(state, &st);
  const int pw = INTEGER(post_warmup)[0];
  const int cap_rows = pw + 1, cap_checks = cap_rows / cc.MAP_every + 3;
  double* met = (double*)R_alloc((size_t)cap_rows * BNMF_NMETRIC, sizeof(double));
  double* maps = (double*)R_alloc((size_t)cap_checks * BNMF_NMAPROW, sizeof(double));
  int n_rows = 0, n_checks = 0;
  chk(bnmf_run_post_warmup(get_handle(ptr), &cc, &st, pw, met, cap_rows, &n_rows, maps, cap_checks, &n_checks));
  return loop_out(met, n_rows, maps, n_checks, &st);
}
/* assign_signatures_ensemble_ (R/postprocessing.R:175-341): C_bnmf_assign(ptr, last_n, used (logical, or NULL = all), reference_P
 * (K x R), keep (logical length N, or NULL), MAP_P (K x N, or NULL), credible_interval, dims c(K,G,N)) ->
 * list(votes N x R, assigned (1-based column of reference_P, NA = not kept), MAP_cosine, lower, upper) */
SEXP C_bnmf_assign(SEXP ptr, SEXP last_n, SEXP used, SEXP reference_P, SEXP keep, SEXP MAP_P, SEXP ci, SEXP dims) {
  const int n = INTEGER(last_n)[0], N = INTEGER(dims)[2], R = Rf_ncols(reference_P);
  int32_t* u = NULL; int32_t* kp = NULL;
  if (used != R_NilValue) { u = (int32_t*)R_alloc(n, sizeof(int32_t)); for (int i = 0; i < n; ++i) u[i] = LOGICAL(used)[i] == TRUE; }
  if (keep != R_NilValue) { kp = (int32_t*)R_alloc(N, sizeof(int32_t)); for (int i = 0; i < N; ++i) kp[i] = LOGICAL(keep)[i] == TRUE; }
  static const char* nms[] = {"votes", "assigned", "MAP_cosine", "lower", "upper"};
  SEXP out = PROTECT(named_list(5, nms));
  SEXP votes = PROTECT(Rf_allocMatrix(REALSXP, N, R)), asg = PROTECT(Rf_allocVector(INTSXP, N));
  SEXP mc = PROTECT(Rf_allocVector(REALSXP, N)), lo = PROTECT(Rf_allocVector(REALSXP, N)), hi = PROTECT(Rf_allocVector(REALSXP, N));
  chk(bnmf_assign(get_handle(ptr), n, u, REAL(reference_P), R, kp, MAP_P == R_NilValue ? NULL : REAL(MAP_P), REAL(ci)[0], REAL(votes),
                  INTEGER(asg), REAL(mc), REAL(lo), REAL(hi)));
  for (int i = 0; i < N; ++i) INTEGER(asg)[i] = INTEGER(asg)[i] < 0 ? NA_INTEGER : INTEGER(asg)[i] + 1;
  SET_VECTOR_ELT(out, 0, votes); SET_VECTOR_ELT(out, 1, asg); SET_VECTOR_ELT(out, 2, mc); SET_VECTOR_ELT(out, 3, lo); SET_VECTOR_ELT(out, 4, hi);
  UNPROTECT(6);
  return out;
}
SEXP C_bnmf_destroy(SEXP ptr) { handle_finalizer(ptr); return R_NilValue; }
SEXP C_bnmf_device_info(SEXP device) {
  char buf[512];
  chk(bnmf_device_info(INTEGER(device)[0], buf, sizeof buf));
  return Rf_mkString(buf);
}

static const R_CallMethodDef call_methods[] = {
  {"C_bnmf_create", (DL_FUNC)&C_bnmf_create, 7}, {"C_bnmf_set_array", (DL_FUNC)&C_bnmf_set_array, 3},
  {"C_bnmf_get_array", (DL_FUNC)&C_bnmf_get_array, 3}, {"C_bnmf_init", (DL_FUNC)&C_bnmf_init, 1},
  {"C_bnmf_run", (DL_FUNC)&C_bnmf_run, 3}, {"C_bnmf_window", (DL_FUNC)&C_bnmf_window, 4},
  {"C_bnmf_get_iter", (DL_FUNC)&C_bnmf_get_iter, 1}, {"C_bnmf_map", (DL_FUNC)&C_bnmf_map, 4},
  {"C_bnmf_run_until", (DL_FUNC)&C_bnmf_run_until, 4}, {"C_bnmf_run_post_warmup", (DL_FUNC)&C_bnmf_run_post_warmup, 5},
  {"C_bnmf_assign", (DL_FUNC)&C_bnmf_assign, 8},
  {"C_bnmf_destroy", (DL_FUNC)&C_bnmf_destroy, 1}, {"C_bnmf_device_info", (DL_FUNC)&C_bnmf_device_info, 1},
  {NULL, NULL, 0}};
void R_init_bayesNMFhip(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
