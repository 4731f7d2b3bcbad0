/*
 * oracle_special.c -- special functions the reference gets from SpecialFunctions.jl
 * (Project.toml:14-19 pins only the range 0.8-0.10; the package source is NOT under the
 * reference tree).  Restated from the published algorithm of that package; the reference's
 * own fp32 OpenCL helper (src/utils.jl:21-53) spells out the same recurrence-to-x>=7 +
 * 8-coefficient asymptotic series, which anchors the digamma coefficients.
 *
 * TEST INFRASTRUCTURE ONLY (see tmvb_oracle.h).  PARITY UNPINNED (no reference vectors) except orc_digamma, which
 * tests/test_oracle_special.py checks at fp32 accuracy against that OpenCL helper compiled from the reference tree
 * (oracle/_ref/Makefile); all three are checked against mpmath to 1e-14 there.
 */
#include "tmvb_oracle.h"
#include <float.h>
#include <math.h>

/* digamma(x::Float64): reflection for x<=0, upward recurrence until x>=7, then
 * psi(x) ~ log x - 1/(2x) - sum_k B_2k / (2k x^2k), k=1..8.
 * Coefficients = bernoulli[2:9] ./ (2*(1:8)); cf. src/utils.jl:42-49. */
double orc_digamma(double x)
{
    double psi = 0.0;
    if (x <= 0.0) {
        psi -= M_PI / tan(M_PI * x);
        x = 1.0 - x;
    }
    if (x < 7.0) {
        int n = 7 - (int)floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    double t = 1.0 / x;
    psi += log(x) - 0.5 * t;
    t *= t;
    /* Horner, highest coefficient first */
    double p = -0.4432598039215686;
    p = p * t + 0.08333333333333333;
    p = p * t + -0.021092796092796094;
    p = p * t + 0.007575757575757576;
    p = p * t + -0.004166666666666667;
    p = p * t + 0.003968253968253968;
    p = p * t + -0.008333333333333333;
    p = p * t + 0.08333333333333333;
    psi -= t * p;
    return psi;
}

/* trigamma(x::Float64): recurrence until x>=8 then
 * psi'(x) ~ 1/x + 1/(2x^2) + sum_k B_2k / x^(2k+1), k=1..8. */
double orc_trigamma(double x)
{
    double psi = 0.0;
    if (x <= 0.0) {
        double s = M_PI / sin(M_PI * x);
        return s * s - orc_trigamma(1.0 - x);
    }
    if (x < 8.0) {
        int n = 8 - (int)floor(x);
        psi += 1.0 / (x * x);
        for (int v = 1; v < n; ++v) {
            double y = x + (double)v;
            psi += 1.0 / (y * y);
        }
        x += (double)n;
    }
    double t = 1.0 / x;
    double w = t * t;
    psi += t + 0.5 * w;
    double p = -7.092156862745098;
    p = p * w + 1.1666666666666667;
    p = p * w + -0.2531135531135531;
    p = p * w + 0.07575757575757576;
    p = p * w + -0.03333333333333333;
    p = p * w + 0.023809523809523808;
    p = p * w + -0.03333333333333333;
    p = p * w + 0.16666666666666666;
    psi += t * w * p;
    return psi;
}

/* loggamma(x::Float64) is libm lgamma for real positive x. */
double orc_lgamma(double x) { return lgamma(x); }

/* finite(x) = sign(x) * min(|x|, floatmax)   src/utils.jl:107 */
double orc_finite(double x)
{
    if (isnan(x)) return x;
    double a = fabs(x);
    if (a > DBL_MAX) a = DBL_MAX;
    return (x > 0) ? a : ((x < 0) ? -a : 0.0);
}

void orc_digamma_vec(const double* x, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_digamma(x[i]);
}

void orc_trigamma_vec(const double* x, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_trigamma(x[i]);
}
