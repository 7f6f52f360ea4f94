"""The kernels divide by level constants (dr^2, dz^2, 2 dr, a_C, R) through precomputed reciprocals; parity with NumPy
requires the correctly rounded quotient every time.  Random operands almost never separate a faithful quotient from a
correctly rounded one, so this test constructs operands whose quotient lies within ~2^-53 ulp of a rounding boundary."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def hard_cases(rng, n_div=300):
    """x, d with x/d a hair (|s| / D half-ulps, |s| small) away from the midpoint of two adjacent doubles: with d = D 2^k
    (D a 53-bit odd integer) and a 54-bit odd M = s D^-1 mod 2^54, M D = K 2^54 + s, so x = K is exact and x/D = (M - s/D) 2^-54."""
    xs, ds = [], []
    divisors = [0.0078125 * 0.0078125 * 1.37, 3.0, 0.1, 6.2, 1.0 / 3.0, 0.0515625, 7.999999999999999, 1.0000000000000002]
    divisors += list(rng.uniform(0.5, 8.0, size=n_div))
    for d in divisors:
        fr, _ = math.frexp(float(d))
        D = int(fr * 2**53)
        while D % 2 == 0:
            D //= 2
        inv = pow(D, -1, 2**54)
        for s in range(-63, 64, 2):
            M = (s * inv) % 2**54
            if M < 2**53:
                continue
            K = (M * D - s) // 2**54
            if K.bit_length() > 53:
                continue
            xs.append(float(K))
            ds.append(float(D))
    return np.array(xs), np.array(ds)


def test_division_sequences_are_correctly_rounded():
    from scpn_fusion_core_amd import _lib
    from scpn_fusion_core_amd._lib import check

    L = _lib.require_gpu()
    rng = np.random.default_rng(11)
    hx, hd = hard_cases(rng)
    assert hx.size > 5000
    # random operands over many binades, with divisors like the level constants (grid spacings, radii)
    rx = rng.normal(size=200000) * 10.0 ** rng.integers(-12, 12, size=200000)
    rd = rng.uniform(1e-6, 1e3, size=200000)
    x = np.ascontiguousarray(np.concatenate([hx, -hx, rx]))
    d = np.ascontiguousarray(np.concatenate([hd, hd, rd]))
    q5, q4 = np.empty_like(x), np.empty_like(x)
    check(L.scpn_verify_division(x, d, int(x.size), q5, q4))
    ref = x / d
    assert np.array_equal(q5, ref)
    assert np.array_equal(q4, ref)
    # the cases are hard: the uncorrected product x * RN(1/d) misses some of them
    assert np.count_nonzero(x * (1.0 / d) != ref) > 0

