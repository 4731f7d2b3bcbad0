/*
 * ref_digamma_wrap.c -- C-ABI wrapper around the reference's OWN fp32 digamma (TEST INFRASTRUCTURE, see tmvb_oracle.h).
 *
 * "digamma_cl.inc" is NOT in the repository: oracle/_ref/Makefile cuts the text of `const DIGAMMA_c` out of
 * /root/reference/src/utils.jl:21-53 into a temporary directory and compiles this file against it, unmodified.  The reference
 * compiles that text as OpenCL C, where floor / log are overloaded on float; <tgmath.h> gives C the same overloads (floorf, logf
 * for a float argument), and `inline` gets internal linkage so the one translation unit links.  -ffp-contract=off: OpenCL's
 * default does not fuse a * b + c either.
 */
#include <stdint.h>
#include <tgmath.h>

#define inline static inline
#include "digamma_cl.inc"
#undef inline

float ref_digamma_f32(float x) { return digamma(x); }

void ref_digamma_f32_vec(const float* x, float* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = digamma(x[i]);
}
