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


def test_oracle_digamma_against_the_reference_fp32_helper_compiled_here(oracle):
    """oracle/_ref: `const DIGAMMA_c` of src/utils.jl:21-53 cut out of the reference tree and compiled unmodified (oracle/_ref/Makefile).
    It is fp32 code, so it pins orc_digamma (fp64) at fp32 accuracy only: its own error against mpmath / SciPy is 4.6 ulp of
    max(|psi|, 1) on [1e-3, 1e6] (the recurrence sum 1/x + 1/(x+1) + ... rounds in fp32); the bound here is 6 ulp."""
    import pytest
    if oracle.build_ref() is None:
        pytest.skip("no reference tree and no prebuilt oracle/_ref/libref_digamma.so")
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 100000)), np.linspace(1.40, 1.52, 2000),
                        [1e-3, 1.0, 6.0, 6.9999995, 7.0, 1e6]]).astype(np.float32)
    ref = oracle.ref_digamma_f32(x).astype(np.float64)
    ours = oracle.digamma(x.astype(np.float64))
    err = np.abs(ours - ref) / np.maximum(np.abs(ours), 1.0)
    assert err.max() <= 6 * 2.0 ** -23, (err.max() / 2.0 ** -23, x[err.argmax()])
    # same branch structure: both shift x < 7 up by n = 7 - floor(x) and sum the same n reciprocals, so at exactly representable
    # small arguments, where fp32 rounding is the only difference, the two agree to fp32 rounding of each term
    for xv in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 100.0):
        a = float(oracle.ref_digamma_f32(np.float32(xv))); b = float(oracle.digamma(xv))
        assert abs(a - b) <= 3 * 2.0 ** -23 * max(1.0, abs(b)), (xv, a, b)
