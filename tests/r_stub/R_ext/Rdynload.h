/* tests/r_stub/R_ext/Rdynload.h — STAND-IN (test infrastructure, see ../README.md): routine registration as documented in
 * "Writing R Extensions" 5.4.  The runtime (rstub.c) keeps the table and calls routines by their registered name with the
 * registered argument count — what `.Call("C_bnmf_run", ...)` does in R. */
#ifndef RSTUB_RDYNLOAD_H
#define RSTUB_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct { const char* name; DL_FUNC fun; int numArgs; void* types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef struct rstub_dllinfo DllInfo;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* const c, const R_CallMethodDef* const call,
                       const R_FortranMethodDef* const f, const R_ExternalMethodDef* const ext);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
