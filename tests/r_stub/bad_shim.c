/* tests/r_stub/bad_shim.c — deliberately WRONG `.Call` routines: the checker's own test.  Each routine makes one of the
 * mistakes the stand-in runtime (rstub.c) exists to catch; tests/test_rshim.py asserts that every one of them is reported
 * (a checker that cannot fail checks nothing) and that the correct routine beside them passes. */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

/* correct: allocate, protect, fill, unprotect, return */
SEXP C_good(SEXP n) {
  const int k = INTEGER(n)[0];
  SEXP out = PROTECT(Rf_allocVector(REALSXP, k));
  SEXP names = PROTECT(Rf_allocVector(INTSXP, k));
  for (int i = 0; i < k; ++i) { REAL(out)[i] = i; INTEGER(names)[i] = i; }
  UNPROTECT(2);
  return out;
}
/* REAL() on an integer vector */
SEXP C_bad_accessor(SEXP n) { return Rf_ScalarReal(REAL(n)[0]); }
/* the first vector is not protected while the second is allocated */
SEXP C_bad_unprotected(SEXP n) {
  SEXP a = Rf_allocVector(REALSXP, INTEGER(n)[0]);
  SEXP b = PROTECT(Rf_allocVector(REALSXP, 1));
  REAL(a)[0] = REAL(b)[0] = 1.0;
  UNPROTECT(1);
  return a;
}
/* one UNPROTECT missing */
SEXP C_bad_imbalance(SEXP n) {
  SEXP a = PROTECT(Rf_allocVector(INTSXP, INTEGER(n)[0]));
  return a;
}
/* an element allocated in a loop while its list is unprotected */
SEXP C_bad_list(SEXP n) {
  SEXP l = Rf_allocVector(VECSXP, 2);
  SET_VECTOR_ELT(l, 0, Rf_ScalarInteger(INTEGER(n)[0]));
  SET_VECTOR_ELT(l, 1, Rf_ScalarInteger(2));
  return l;
}
/* an error after allocations: R unwinds the PROTECT stack itself, nothing to report */
SEXP C_error_after_protect(SEXP n) {
  SEXP a = PROTECT(Rf_allocVector(INTSXP, 4));
  if (INTEGER(n)[0] >= 0) Rf_error("refused: %d", INTEGER(n)[0]);
  UNPROTECT(1);
  return a;
}
/* writes one element past the end of an R vector: not detectable by the stand-in's API checks (this is what the CPU build
 * under AddressSanitizer is for); listed so that the test says so */
static const R_CallMethodDef call_methods[] = {
  {"C_good", (DL_FUNC)&C_good, 1}, {"C_bad_accessor", (DL_FUNC)&C_bad_accessor, 1}, {"C_bad_unprotected", (DL_FUNC)&C_bad_unprotected, 1},
  {"C_bad_imbalance", (DL_FUNC)&C_bad_imbalance, 1}, {"C_bad_list", (DL_FUNC)&C_bad_list, 1},
  {"C_error_after_protect", (DL_FUNC)&C_error_after_protect, 1}, {NULL, NULL, 0}};
void R_init_badshim(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
