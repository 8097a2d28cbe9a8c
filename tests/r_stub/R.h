/* tests/r_stub/R.h — STAND-IN (test infrastructure, see README.md in this directory): R.h of a real installation pulls in
 * the standard C headers and R's utility declarations; the shim needs nothing from it beyond what Rinternals.h declares. */
#ifndef RSTUB_R_H
#define RSTUB_R_H
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "Rinternals.h"
#endif
