/* tests/r_stub/Rinternals.h — STAND-IN for R's C API (test infrastructure, see README.md in this directory).
 * Only what r/bnmf_shim.c uses, declared from the documented behaviour of the API ("Writing R Extensions", 5.9);
 * not R, not derived from R's sources.  The runtime behind it is rstub.c. */
#ifndef RSTUB_RINTERNALS_H
#define RSTUB_RINTERNALS_H
#include <stddef.h>
#include <stdint.h>
#include <limits.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rstub_sexprec* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE = 1 } Rboolean;

/* the type codes the shim names (values as documented) */
#define NILSXP 0
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22

#define NA_INTEGER INT_MIN
#define NA_LOGICAL INT_MIN

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;

SEXP Rf_allocVector(unsigned type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)

int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
double* REAL(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int LENGTH(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);

SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP Rf_mkChar(const char* s);
SEXP Rf_mkString(const char* s);
const char* R_CHAR(SEXP x);
#define CHAR(x) R_CHAR(x)
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_ScalarReal(double v);
SEXP Rf_ScalarLogical(int v);

typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot);
void* R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);

/* transient storage, reclaimed when the .Call returns (or unwinds) */
char* R_alloc(size_t n, int size);

#if defined(__GNUC__)
__attribute__((noreturn, format(printf, 1, 2)))
#endif
void Rf_error(const char* fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
