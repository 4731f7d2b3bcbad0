"""
Accuracy pins of the fp32 device special functions (tmvb_special_f32) against fp64 (SciPy):
  digamma_f  (6-step recurrence + 4-term series, rcp-based reciprocals): abs error <= 6 ulp of max(|psi(x)|, 1) over
             x in [1e-3, 1e6] (measured 4.96 ulp, at x = 3.16 where psi(x + 6) and the recurrence sum cancel; relative
             error is meaningless at the root x0 = 1.4616);
  fast_exp   (2^n * 2^f with a two-float product): rel error <= 4 ulp for x in [-87, 0], 0 below, monotone clamp at -150;
  fast_rcp   (v_rcp_f32 + one Newton step): rel error <= 1.5 ulp over 1e-30..1e30.
"""
import ctypes as C

import numpy as np
import pytest
from scipy import special

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -23
DIGAMMA_REF_ULP = 8.0          # device fp32 digamma vs the reference's fp32 helper (oracle/_ref); measured on MI355X: see the test


def run(tmvb, which, x):
    ctx = tmvb.DeviceContext(0)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rc = tmvb.lib().tmvb_special_f32(ctx.handle, C.c_int32(which), fp(x), fp(y), C.c_int64(x.size))
    assert rc == 0, tmvb.lib().tmvb_last_error()
    return x.astype(np.float64), y.astype(np.float64)


def test_digamma(tmvb):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 200000)), np.linspace(1.40, 1.52, 5000), [1e-3, 1.0, 6.0, 1e6]])
    x, y = run(tmvb, 0, x)
    ref = special.digamma(x)
    err = np.abs(y - ref) / np.maximum(np.abs(ref), 1.0)
    assert err.max() <= 6 * ULP, (err.max() / ULP, x[err.argmax()])


def test_exp(tmvb):
    x = np.concatenate([np.linspace(-87.0, 0.0, 200001), [-100.0, -150.0, -1e30, -np.inf]])
    x, y = run(tmvb, 1, x)
    body = x >= -87.0
    ref = np.exp(x[body])
    assert (np.abs(y[body] - ref) / ref).max() <= 4 * ULP
    assert np.all(y[~body] >= 0.0) and np.all(y[~body] < 1e-37) and np.all(np.isfinite(y))


def test_rcp(tmvb):
    rng = np.random.default_rng(1)
    x = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 200000)) * rng.choice([-1.0, 1.0], 200000)
    x, y = run(tmvb, 2, x)
    assert (np.abs(y * x - 1.0)).max() <= 1.5 * ULP


def test_device_digamma_against_the_reference_fp32_helper_compiled_here(tmvb, oracle):
    """tmvb_special_f32(which = 0) against the reference's own fp32 digamma (`const DIGAMMA_c`, src/utils.jl:21-53) compiled from the
    reference tree by oracle/_ref/Makefile (the library travels to the GPU box prebuilt).  Both are fp32 evaluations of the same
    published algorithm: the reference's is 4.6 ulp from fp64 (of max(|psi|, 1)), the device's 4.96 ulp (test_digamma above), so
    two correct implementations may differ by the sum; DIGAMMA_REF_ULP is the frozen bound, <= 10x the measured difference."""
    if oracle.build_ref() is None:
        pytest.skip("oracle/_ref/libref_digamma.so was not shipped")
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 200000)), np.linspace(1.40, 1.52, 5000), [1e-3, 1.0, 6.0, 7.0, 1e6]])
    x, y = run(tmvb, 0, x)
    ref = oracle.ref_digamma_f32(x.astype(np.float32)).astype(np.float64)
    err = np.abs(y - ref) / np.maximum(np.abs(ref), 1.0)
    print(f"device digamma vs reference fp32 helper: max {err.max() / ULP:.2f} ulp at x = {x[err.argmax()]:.6g}, mean {err.mean() / ULP:.3f} ulp")
    assert err.max() <= DIGAMMA_REF_ULP * ULP, (err.max() / ULP, x[err.argmax()])
