// What this binary is: mcpc_build_info() (include/mcpc.h) reports the hash of the sources it was compiled from, the commit, the
// compiler flags, and whether ANY timing-experiment switch was on.
//
// The kernel sources carry `#ifdef MCPC_EXP_*` / `MCPC_HEB_EXP` blocks marked "timing experiment only (wrong results)": builds that
// skip loads, MFMAs or stores to measure what each costs (profiles/r04_k1_bounds.txt).  They exist only as `make variant` libraries
// under scripts/bin/ -- and such a library must never be benchmarked or tested as the product.  This header is the ONE place that
// knows every switch (tests/test_build_info.py fails when a source mentions a switch this list does not): a build with any of them
// reports exp=1, the Python binding refuses to load it unless MCPC_ALLOW_EXP=1, and bench.py's self_check asserts exp=0.
#pragma once

#if defined(MCPC_EXP_NOLOAD) || defined(MCPC_EXP_NOSPLIT) || defined(MCPC_EXP_NOTAILMASK) || defined(MCPC_EXP_HALFMFMA) || \
    defined(MCPC_EXP_FULLFENCE) || defined(MCPC_EXP_NOGEMM) || defined(MCPC_EXP_NOSPILL) || defined(MCPC_EXP_NOX) || \
    defined(MCPC_EXP_NOELOAD) || defined(MCPC_EXP_SPILL_LINEAR) || defined(MCPC_EXP_NOY) || defined(MCPC_EXP_NOLEAN) || \
    defined(MCPC_EXP_NOEPI) || defined(MCPC_EXP_NOROWEXP) || defined(MCPC_HEB_EXP)
#define MCPC_TIMING_BUILD 1
#else
#define MCPC_TIMING_BUILD 0
#endif

#ifdef MCPC_STAMPS          // in-kernel phase stamps: a diagnostic build with the product's results (never the one bench.py times)
#define MCPC_STAMPS_BUILD 1
#else
#define MCPC_STAMPS_BUILD 0
#endif

// set by the Makefile; a build by hand reports "unknown"
#ifndef MCPC_BUILD_CSRC_SHA
#define MCPC_BUILD_CSRC_SHA "unknown"
#endif
#ifndef MCPC_BUILD_COMMIT
#define MCPC_BUILD_COMMIT "unknown"
#endif
#ifndef MCPC_BUILD_FLAGS
#define MCPC_BUILD_FLAGS "unknown"
#endif
