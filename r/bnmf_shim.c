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
  {"C_bnmf_destroy", (DL_FUNC)&C_bnmf_destroy, 1}, {"C_bnmf_device_info", (DL_FUNC)&C_bnmf_device_info, 1},
  {NULL, NULL, 0}};
void R_init_bayesNMFhip(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
