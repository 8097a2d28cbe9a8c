/* tests/r_stub/rstub.c — STAND-IN for the R runtime behind Rinternals.h / Rdynload.h in this directory (test infrastructure
 * only; see README.md).  Not R, not derived from R's sources: the documented behaviour of the handful of API functions
 * r/bnmf_shim.c uses, made STRICT so that marshalling mistakes fail deterministically:
 *   - typed accessors: REAL / INTEGER / LOGICAL on a vector of another type fail the call;
 *   - a checked PROTECT stack: underflow fails the call, a non-empty stack when a .Call returns is reported;
 *   - gctorture: EVERY allocation collects the objects of the running call that are neither protected nor reachable from a
 *     protected object, an argument or a preserved object; their payload is poisoned and any later use fails the call;
 *   - Rf_error unwinds by longjmp to the .Call boundary, the PROTECT stack is reset and R_alloc memory reclaimed, as in R;
 *   - external pointers run their C finalizer when they are released and collected (rstub_release + rstub_gc).
 * The rstub_* functions at the end are the ctypes-facing driver: build argument vectors, call a registered routine BY NAME
 * with the registered argument count, inspect the result. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"
#include "R_ext/Rdynload.h"

#define MAGIC_LIVE 0x52535842u
#define MAGIC_DEAD 0xDEADDEADu

struct rstub_sexprec {
  uint32_t magic;
  int type;
  R_xlen_t length;
  int nrow, ncol;              /* -1: no dim attribute */
  void* data;                  /* int / double payload, SEXP[] for VECSXP / STRSXP, char[] for CHARSXP */
  SEXP names;                  /* the one attribute the shim sets */
  void* ext; R_CFinalizer_t fin;
  int preserved;               /* owned by the driver (arguments, returned results): a root, never collected inside a call */
  int mark;
  struct rstub_sexprec* next;
};
struct rstub_dllinfo { int dynamic_symbols; };

static struct rstub_sexprec nil_rec = {MAGIC_LIVE, NILSXP, 0, -1, -1, NULL, NULL, NULL, NULL, 1, 0, NULL};
static struct rstub_sexprec names_sym = {MAGIC_LIVE, 1 /* SYMSXP */, 0, -1, -1, NULL, NULL, NULL, NULL, 1, 0, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol = &names_sym;

static SEXP heap = NULL;
#define PSTACK_MAX 10000
static SEXP pstack[PSTACK_MAX];
static int pdepth = 0;
static int in_call = 0;
static jmp_buf call_jmp;
static char last_error[1024] = "";
static char last_violation[1024] = "";
static int n_violations = 0, n_finalized = 0, n_collected_in_call = 0;
static SEXP* cur_args = NULL; static int cur_nargs = 0;
typedef struct ralloc_blk { struct ralloc_blk* next; } ralloc_blk;
static ralloc_blk* ralloc_head = NULL;
static const R_CallMethodDef* call_table = NULL;
static struct rstub_dllinfo the_dll = {1};

/* a mistake of the code under test (not an R-level error): recorded, and the running call is abandoned */
#if defined(__GNUC__)
__attribute__((format(printf, 1, 2)))
#endif
static void violation(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(last_violation, sizeof last_violation, fmt, ap); va_end(ap);
  ++n_violations;
  if (in_call) longjmp(call_jmp, 2);
  fprintf(stderr, "rstub: %s (outside a .Call)\n", last_violation);
  abort();
}
static const char* tname(int t) {
  switch (t) { case NILSXP: return "NULL"; case CHARSXP: return "CHARSXP"; case LGLSXP: return "logical"; case INTSXP: return "integer";
    case REALSXP: return "double"; case STRSXP: return "character"; case VECSXP: return "list"; case EXTPTRSXP: return "externalptr"; default: return "?"; }
}
static SEXP chk_live(SEXP x, const char* who) {
  if (!x) violation("%s: NULL C pointer used as a SEXP", who);
  if (x->magic == MAGIC_DEAD) violation("%s: use of an object that was garbage-collected (it was neither PROTECTed nor reachable at an allocation)", who);
  if (x->magic != MAGIC_LIVE) violation("%s: not a SEXP", who);
  return x;
}

/* ---------------- collection */
static void mark(SEXP x) {
  if (!x || x->magic != MAGIC_LIVE || x->mark) return;
  x->mark = 1;
  if (x->names) mark(x->names);
  if (x->type == VECSXP || x->type == STRSXP) for (R_xlen_t i = 0; i < x->length; ++i) mark(((SEXP*)x->data)[i]);
}
static void mark_roots(void) {
  for (SEXP p = heap; p; p = p->next) p->mark = 0;
  for (SEXP p = heap; p; p = p->next) if (p->preserved) mark(p);
  for (int i = 0; i < pdepth; ++i) mark(pstack[i]);
  for (int i = 0; i < cur_nargs; ++i) mark(cur_args[i]);
}
static size_t payload_bytes(SEXP x) {
  switch (x->type) { case LGLSXP: case INTSXP: return (size_t)x->length * sizeof(int); case REALSXP: return (size_t)x->length * sizeof(double);
    case VECSXP: case STRSXP: return (size_t)x->length * sizeof(SEXP); case CHARSXP: return (size_t)x->length + 1; default: return 0; }
}
/* inside a call: poison, keep the record so that a later use is recognised; finalizers of collected external pointers run */
static void torture_collect(void) {
  mark_roots();
  for (SEXP p = heap; p; p = p->next) {
    if (p->mark || p->magic != MAGIC_LIVE) continue;
    if (p->type == EXTPTRSXP && p->fin) { R_CFinalizer_t f = p->fin; p->fin = NULL; f(p); ++n_finalized; }
    if (p->data) memset(p->data, 0xA5, payload_bytes(p));
    p->magic = MAGIC_DEAD;
    ++n_collected_in_call;
  }
}
/* outside a call: really free everything that is not reachable from a preserved object */
static int full_gc(void) {
  int fin0 = n_finalized;
  cur_args = NULL; cur_nargs = 0;
  mark_roots();
  SEXP* link = &heap;
  while (*link) {
    SEXP p = *link;
    if (p->mark && p->magic == MAGIC_LIVE) { link = &p->next; continue; }
    if (p->magic == MAGIC_LIVE && p->type == EXTPTRSXP && p->fin) { R_CFinalizer_t f = p->fin; p->fin = NULL; in_call = 0; f(p); ++n_finalized; }
    *link = p->next;
    free(p->data);
    p->magic = 0;
    free(p);
  }
  return n_finalized - fin0;
}

static SEXP new_obj(int type, R_xlen_t n) {
  if (n < 0) violation("allocVector: negative length");
  if (in_call) torture_collect();
  SEXP x = (SEXP)calloc(1, sizeof *x);
  x->magic = MAGIC_LIVE; x->type = type; x->length = n; x->nrow = x->ncol = -1;
  size_t b = 0;
  switch (type) { case LGLSXP: case INTSXP: b = (size_t)n * sizeof(int); break; case REALSXP: b = (size_t)n * sizeof(double); break;
    case VECSXP: case STRSXP: b = (size_t)n * sizeof(SEXP); break; case CHARSXP: b = (size_t)n + 1; break; case EXTPTRSXP: b = 0; break;
    default: free(x); violation("allocVector: type %d is not supported by the stand-in", type); }
  if (b) {
    x->data = malloc(b);
    /* R does not zero numeric vectors: fill them with a recognisable pattern so that reading an element the shim never wrote
     * shows up in the comparison with the engine */
    memset(x->data, (type == VECSXP || type == STRSXP || type == CHARSXP) ? 0 : 0x7B, b);
    if (type == VECSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)x->data)[i] = R_NilValue;
    if (type == STRSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)x->data)[i] = NULL;
  }
  x->preserved = in_call ? 0 : 1;
  x->next = heap; heap = x;
  return x;
}

/* ---------------- the API */
SEXP Rf_allocVector(unsigned type, R_xlen_t n) { return new_obj((int)type, n); }
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol) {
  if (nrow < 0 || ncol < 0) violation("allocMatrix: negative extent (%d x %d)", nrow, ncol);
  SEXP x = new_obj((int)type, (R_xlen_t)nrow * ncol);
  x->nrow = nrow; x->ncol = ncol;
  return x;
}
SEXP Rf_protect(SEXP x) {
  chk_live(x, "PROTECT");
  if (pdepth >= PSTACK_MAX) violation("PROTECT: stack overflow");
  pstack[pdepth++] = x;
  return x;
}
void Rf_unprotect(int n) {
  if (n < 0 || n > pdepth) violation("UNPROTECT(%d): only %d object(s) on the stack", n, pdepth);
  pdepth -= n;
}
static void* typed(SEXP x, int type, const char* who) {
  chk_live(x, who);
  if (x->type != type) violation("%s() applied to a %s vector", who, tname(x->type));
  return x->data;
}
int* INTEGER(SEXP x) { return (int*)typed(x, INTSXP, "INTEGER"); }
int* LOGICAL(SEXP x) { return (int*)typed(x, LGLSXP, "LOGICAL"); }
double* REAL(SEXP x) { return (double*)typed(x, REALSXP, "REAL"); }
R_xlen_t XLENGTH(SEXP x) { return chk_live(x, "XLENGTH")->length; }
int LENGTH(SEXP x) { return (int)chk_live(x, "LENGTH")->length; }
int Rf_nrows(SEXP x) { chk_live(x, "nrows"); if (x->nrow < 0) { if (x->type == NILSXP) violation("nrows: object is not a matrix"); return (int)x->length; } return x->nrow; }
int Rf_ncols(SEXP x) { chk_live(x, "ncols"); if (x->ncol < 0) { if (x->type == NILSXP) violation("ncols: object is not a matrix"); return 1; } return x->ncol; }
static void chk_index(SEXP x, R_xlen_t i, const char* who) { if (i < 0 || i >= x->length) violation("%s: index %ld out of bounds (length %ld)", who, (long)i, (long)x->length); }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { chk_live(x, "VECTOR_ELT"); if (x->type != VECSXP) violation("VECTOR_ELT on a %s", tname(x->type)); chk_index(x, i, "VECTOR_ELT"); return ((SEXP*)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
  chk_live(x, "SET_VECTOR_ELT"); chk_live(v, "SET_VECTOR_ELT (value)");
  if (x->type != VECSXP) violation("SET_VECTOR_ELT on a %s", tname(x->type));
  chk_index(x, i, "SET_VECTOR_ELT");
  ((SEXP*)x->data)[i] = v;
  return v;
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) {
  chk_live(x, "SET_STRING_ELT"); chk_live(v, "SET_STRING_ELT (value)");
  if (x->type != STRSXP) violation("SET_STRING_ELT on a %s", tname(x->type));
  if (v->type != CHARSXP) violation("SET_STRING_ELT: value is a %s, not a CHARSXP", tname(v->type));
  chk_index(x, i, "SET_STRING_ELT");
  ((SEXP*)x->data)[i] = v;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i) { chk_live(x, "STRING_ELT"); if (x->type != STRSXP) violation("STRING_ELT on a %s", tname(x->type)); chk_index(x, i, "STRING_ELT"); return ((SEXP*)x->data)[i]; }
SEXP Rf_mkChar(const char* s) {
  if (!s) violation("mkChar(NULL)");
  SEXP x = new_obj(CHARSXP, (R_xlen_t)strlen(s));
  memcpy(x->data, s, strlen(s) + 1);
  return x;
}
const char* R_CHAR(SEXP x) { chk_live(x, "CHAR"); if (x->type != CHARSXP) violation("CHAR on a %s", tname(x->type)); return (const char*)x->data; }
SEXP Rf_mkString(const char* s) {
  SEXP x = Rf_protect(new_obj(STRSXP, 1));
  SET_STRING_ELT(x, 0, Rf_mkChar(s));
  Rf_unprotect(1);
  return x;
}
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
  chk_live(x, "setAttrib"); chk_live(val, "setAttrib (value)");
  if (name != R_NamesSymbol) violation("setAttrib: only the names attribute exists in the stand-in");
  if (val->type != STRSXP || val->length != x->length) violation("setAttrib(names): value must be a character vector of the object's length");
  x->names = val;
  return val;
}
SEXP Rf_getAttrib(SEXP x, SEXP name) { chk_live(x, "getAttrib"); return (name == R_NamesSymbol && x->names) ? x->names : R_NilValue; }
SEXP Rf_ScalarInteger(int v) { SEXP x = new_obj(INTSXP, 1); ((int*)x->data)[0] = v; return x; }
SEXP Rf_ScalarLogical(int v) { SEXP x = new_obj(LGLSXP, 1); ((int*)x->data)[0] = v; return x; }
SEXP Rf_ScalarReal(double v) { SEXP x = new_obj(REALSXP, 1); ((double*)x->data)[0] = v; return x; }

SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot) {
  (void)tag; (void)prot;
  SEXP x = new_obj(EXTPTRSXP, 0);
  x->ext = p;
  return x;
}
static SEXP chk_ext(SEXP s, const char* who) { chk_live(s, who); if (s->type != EXTPTRSXP) violation("%s on a %s", who, tname(s->type)); return s; }
void* R_ExternalPtrAddr(SEXP s) { return chk_ext(s, "R_ExternalPtrAddr")->ext; }
void R_ClearExternalPtr(SEXP s) { chk_ext(s, "R_ClearExternalPtr")->ext = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; chk_ext(s, "R_RegisterCFinalizerEx")->fin = fun; }

char* R_alloc(size_t n, int size) {
  if (size < 0) violation("R_alloc: negative element size");
  ralloc_blk* b = (ralloc_blk*)malloc(sizeof(ralloc_blk) + 16 + n * (size_t)size);
  b->next = ralloc_head; ralloc_head = b;
  char* p = (char*)(b + 1);
  p += (16 - ((uintptr_t)p & 15)) & 15;
  memset(p, 0x5C, n * (size_t)size);
  return p;
}
static void ralloc_reset(void) { while (ralloc_head) { ralloc_blk* b = ralloc_head; ralloc_head = b->next; free(b); } }

void Rf_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(last_error, sizeof last_error, fmt, ap); va_end(ap);
  if (!in_call) { fprintf(stderr, "rstub: Rf_error outside a .Call: %s\n", last_error); abort(); }
  longjmp(call_jmp, 1);
}

int R_registerRoutines(DllInfo* info, const R_CMethodDef* const c, const R_CallMethodDef* const call, const R_FortranMethodDef* const f,
                       const R_ExternalMethodDef* const ext) {
  (void)info; (void)c; (void)f; (void)ext;
  call_table = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) { Rboolean old = info->dynamic_symbols ? TRUE : FALSE; info->dynamic_symbols = value; return old; }

/* ================= the driver (called from Python through ctypes) ================= */
/* what R does after dlopen of a package's shared object: call R_init_<pkg>(dll).  The caller looks the symbol up (ctypes) */
int rstub_load(void (*init)(DllInfo*)) {
  if (!init) return -1;
  call_table = NULL;
  the_dll.dynamic_symbols = 1;
  init(&the_dll);
  return (call_table && the_dll.dynamic_symbols == 0) ? 0 : -2;
}
int rstub_n_routines(void) { int n = 0; if (call_table) while (call_table[n].name) ++n; return n; }
const char* rstub_routine_name(int i) { return call_table[i].name; }
int rstub_routine_nargs(int i) { return call_table[i].numArgs; }

typedef SEXP (*fn0)(void); typedef SEXP (*fn1)(SEXP); typedef SEXP (*fn2)(SEXP, SEXP); typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP); typedef SEXP (*fn5)(SEXP, SEXP, SEXP, SEXP, SEXP); typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP); typedef SEXP (*fn8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

static void preserve_tree(SEXP x) {
  if (!x || x->magic != MAGIC_LIVE || x->preserved) return;
  x->preserved = 1;
  if (x->names) preserve_tree(x->names);
  if (x->type == VECSXP || x->type == STRSXP) for (R_xlen_t i = 0; i < x->length; ++i) preserve_tree(((SEXP*)x->data)[i]);
}
static void sweep_dead(void) {                /* the records kept poisoned during a call */
  SEXP* link = &heap;
  while (*link) { SEXP p = *link; if (p->magic == MAGIC_DEAD) { *link = p->next; free(p->data); p->magic = 0; free(p); } else link = &p->next; }
}
/* the part of a .Call that runs under setjmp: everything it touches after the jump is static or volatile */
static const R_CallMethodDef* cur_method = NULL;
static SEXP cur_result = NULL;
static int invoke_current(void) {
  const int j = setjmp(call_jmp);
  if (j != 0) return j;                       /* 1: Rf_error, 2: violation */
  DL_FUNC f = cur_method->fun;
  SEXP* a = cur_args;
  switch (cur_nargs) {
    case 0: cur_result = ((fn0)f)(); break;
    case 1: cur_result = ((fn1)f)(a[0]); break;
    case 2: cur_result = ((fn2)f)(a[0], a[1]); break;
    case 3: cur_result = ((fn3)f)(a[0], a[1], a[2]); break;
    case 4: cur_result = ((fn4)f)(a[0], a[1], a[2], a[3]); break;
    case 5: cur_result = ((fn5)f)(a[0], a[1], a[2], a[3], a[4]); break;
    case 6: cur_result = ((fn6)f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 7: cur_result = ((fn7)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 8: cur_result = ((fn8)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    default: cur_result = ((fn9)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
  }
  return 0;
}
/* .Call(name, args...): 0 = returned normally (*out preserved for the caller until rstub_release), 1 = the routine raised an R
 * error (rstub_last_error), 2 = the code under test broke a rule of the API (rstub_last_violation), 3 = no such routine / wrong
 * argument count (as R reports for a registered routine) */
int rstub_call(const char* name, int nargs, SEXP* args, SEXP* out) {
  *out = NULL;
  if (!call_table) { snprintf(last_violation, sizeof last_violation, "no routines registered: call rstub_load first"); return 3; }
  cur_method = NULL;
  for (const R_CallMethodDef* p = call_table; p->name; ++p) if (!strcmp(p->name, name)) cur_method = p;
  if (!cur_method) { snprintf(last_violation, sizeof last_violation, "\"%s\" not available for .Call()", name); return 3; }
  if (cur_method->numArgs != nargs) { snprintf(last_violation, sizeof last_violation, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, cur_method->numArgs, name); return 3; }
  if (nargs > 9) { snprintf(last_violation, sizeof last_violation, "more than 9 arguments: not supported by the stand-in"); return 3; }
  for (int i = 0; i < nargs; ++i) if (!args[i] || args[i]->magic != MAGIC_LIVE) { snprintf(last_violation, sizeof last_violation, "argument %d is not a live SEXP", i); return 3; }
  last_error[0] = 0; last_violation[0] = 0;
  pdepth = 0; n_collected_in_call = 0;
  cur_args = args; cur_nargs = nargs; cur_result = NULL;
  in_call = 1;
  int rc = invoke_current();
  in_call = 0;
  if (rc == 0) {
    if (pdepth != 0) {
      snprintf(last_violation, sizeof last_violation, "stack imbalance in .Call(\"%s\"): %d object(s) left PROTECTed", name, pdepth);
      ++n_violations; rc = 2;
    } else if (!cur_result || cur_result->magic != MAGIC_LIVE) {
      snprintf(last_violation, sizeof last_violation, ".Call(\"%s\") returned %s", name, !cur_result ? "a NULL C pointer" : "a collected object");
      ++n_violations; rc = 2;
    }
  }
  pdepth = 0;                                 /* R resets the PROTECT stack when it unwinds */
  ralloc_reset();
  if (rc == 0) { preserve_tree(cur_result); *out = cur_result; }
  sweep_dead();
  cur_args = NULL; cur_nargs = 0; cur_result = NULL;
  return rc;
}

SEXP rstub_nil(void) { return R_NilValue; }
SEXP rstub_mk_int(long n, const int* v) { SEXP x = new_obj(INTSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(int)); return x; }
SEXP rstub_mk_lgl(long n, const int* v) { SEXP x = new_obj(LGLSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(int)); return x; }
SEXP rstub_mk_real(long n, const double* v) { SEXP x = new_obj(REALSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(double)); return x; }
SEXP rstub_mk_int_matrix(int nr, int nc, const int* v) { SEXP x = rstub_mk_int((long)nr * nc, v); x->nrow = nr; x->ncol = nc; return x; }
SEXP rstub_mk_real_matrix(int nr, int nc, const double* v) { SEXP x = rstub_mk_real((long)nr * nc, v); x->nrow = nr; x->ncol = nc; return x; }
int rstub_typeof(SEXP x) { return x->type; }
long rstub_xlength(SEXP x) { return (long)x->length; }
int rstub_nrow(SEXP x) { return x->nrow; }
int rstub_ncol(SEXP x) { return x->ncol; }
void* rstub_dataptr(SEXP x) { return x->data; }
SEXP rstub_elt(SEXP x, long i) { return (x->type == VECSXP && i >= 0 && i < x->length) ? ((SEXP*)x->data)[i] : NULL; }
const char* rstub_string(SEXP x, long i) { return (x->type == STRSXP && i >= 0 && i < x->length && ((SEXP*)x->data)[i]) ? (const char*)((SEXP*)x->data)[i]->data : NULL; }
const char* rstub_name(SEXP x, long i) { return x->names ? rstub_string(x->names, i) : NULL; }
void* rstub_extptr_addr(SEXP x) { return x->type == EXTPTRSXP ? x->ext : NULL; }
/* the caller drops its reference; the object (tree) is collected by the next rstub_gc unless something preserved still holds it */
static void unpreserve_tree(SEXP x) {
  if (!x || x->magic != MAGIC_LIVE || !x->preserved || x == R_NilValue || x == R_NamesSymbol) return;
  x->preserved = 0;
  if (x->names) unpreserve_tree(x->names);
  if (x->type == VECSXP || x->type == STRSXP) for (R_xlen_t i = 0; i < x->length; ++i) unpreserve_tree(((SEXP*)x->data)[i]);
}
void rstub_release(SEXP x) { unpreserve_tree(x); }
int rstub_gc(void) { return full_gc(); }      /* returns the number of finalizers that ran */
int rstub_live_objects(void) { int n = 0; for (SEXP p = heap; p; p = p->next) ++n; return n; }
int rstub_protect_depth(void) { return pdepth; }
int rstub_violations(void) { return n_violations; }
int rstub_finalizers_run(void) { return n_finalized; }
int rstub_collected_in_last_call(void) { return n_collected_in_call; }
const char* rstub_last_error(void) { return last_error; }
const char* rstub_last_violation(void) { return last_violation; }
