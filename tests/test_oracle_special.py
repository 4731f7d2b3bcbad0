"""Special functions of the C oracle vs mpmath."""
import mpmath
import numpy as np


def test_digamma_trigamma_vs_mpmath(oracle):
    xs = np.concatenate([np.logspace(-6, 6, 300), np.linspace(0.01, 20.0, 500), [1.0, 7.0, 8.0, 6.999999, 7.000001]])
    dg = oracle.digamma(xs)
    tg = oracle.trigamma(xs)
    for x, a, b in zip(xs, dg, tg):
        ra = float(mpmath.digamma(x)); rb = float(mpmath.polygamma(1, x))
        assert abs(a - ra) <= 1e-14 * max(1.0, abs(ra)), x
        assert abs(b - rb) <= 1e-14 * max(1.0, abs(rb)), x


def test_digamma_coefficients_match_reference_opencl_helper():
    # src/utils.jl:42-49 lists the same 8 asymptotic coefficients in fp32 spelling.
    coef = [1 / 12, -1 / 120, 1 / 252, -1 / 240, 1 / 132, -691 / 32760, 1 / 12, -3617 / 8160]
    ours = [0.08333333333333333, -0.008333333333333333, 0.003968253968253968, -0.004166666666666667,
            0.007575757575757576, -0.021092796092796094, 0.08333333333333333, -0.4432598039215686]
    assert np.allclose(coef, ours, rtol=1e-15)


def test_epsilon_is_eps_1e14(oracle):
    # src/utils.jl:3
    assert oracle.EPSILON == float(np.spacing(1e-14)) == 2.0 ** -99
