"""No GPU: the error bound of the wide variant's float32 pre-test (csrc/frame_kernel.hip match_roots_wide /
match_pairs_wide), checked in exact rational arithmetic.

The kernel drops a blob for a (root, camera) pair without the exact double decision of helpers.py:373,375 when
    |t32| > thr,   t32 = fmaf(a, x, fmaf(b, y, c)) in float32,   thr = RU(gate * den * (1 + 1e-12) + E * (1 + 1e-6)),
    E = 2.5 * 2^-24 * (2 * omax + |c|),   omax >= |x|, |y|,
with a, b, c the float32 line of cv.computeCorrespondEpilines (helpers.py:363-364: with the float32 roundings on, the line
the reference measures with IS these float32 values), x, y float32 blob coordinates and den = sqrt(a^2 + b^2) in double.
The drop is exact iff it never drops a blob whose true distance |a x + b y + c| / den is < gate, i.e. iff
    |t32 - (a x + b y + c)| <= E          (then |t32| > gate den + E  =>  |a x + b y + c| > gate den).
Here the two fused operations are emulated with Fractions (one correctly rounded float32 result each), the exact value is
a Fraction, and the inequality is checked on random and adversarial inputs -- including the 16 k-pixel coordinates and the
un-normalised lines the adversarial GPU tests use.  (The GPU side of the same claim: tests/test_gpu_wide_adversarial.py,
self-check build -DMOCAP_DEBUG_PRETEST.)"""
from fractions import Fraction

import numpy as np


def round_f32(q):
    """Correctly rounded (nearest, ties to even) float32 of the exact rational q, as a Fraction."""
    if q == 0:
        return Fraction(0)
    s = -1 if q < 0 else 1
    q = abs(q)
    e = q.numerator.bit_length() - q.denominator.bit_length()      # 2^(e-1) <= q < 2^(e+1)
    if Fraction(2) ** e > q:
        e -= 1                                                     # 2^e <= q < 2^(e+1)
    e = max(e, -126)                                               # subnormals share the smallest exponent
    ulp = Fraction(2) ** (e - 23)
    n = q / ulp
    lo = n.numerator // n.denominator
    rem = n - lo
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and lo % 2 == 1):
        lo += 1
    return s * lo * ulp


def fma32(a, b, c):
    return round_f32(Fraction(a) * Fraction(b) + Fraction(c))


def check(a, b, c, x, y, omax):
    a, b, c, x, y = (np.float32(v) for v in (a, b, c, x, y))
    exact = Fraction(float(a)) * Fraction(float(x)) + Fraction(float(b)) * Fraction(float(y)) + Fraction(float(c))
    t32 = fma32(float(a), float(x), fma32(float(b), float(y), float(c)))
    # sanity of the emulation: the float32 result is what NumPy's float64 evaluation rounds to whenever that is unambiguous
    E = Fraction(5, 2) * Fraction(1, 2 ** 24) * (2 * Fraction(float(omax)) + abs(Fraction(float(c))))
    assert abs(t32 - exact) <= E, (float(a), float(b), float(c), float(x), float(y), float(abs(t32 - exact)), float(E))
    return float(abs(t32 - exact) / E) if E else 0.0


def test_round_f32_is_numpy_float32_rounding():
    rng = np.random.default_rng(0)
    for v in np.concatenate([rng.standard_normal(200) * 10.0 ** rng.integers(-20, 20, 200), [0.1, 1 / 3, 16777217.0, 2.5e-45, 1e-40]]):
        assert float(round_f32(Fraction(float(v)))) == float(np.float32(v)), v
    # ties to even: 2^24 + 1 lies exactly between two float32 values
    assert float(round_f32(Fraction(2 ** 24 + 1))) == 2.0 ** 24 and float(round_f32(Fraction(2 ** 24 + 3))) == 2.0 ** 24 + 4


def test_pretest_bound_holds_for_normalised_float32_lines():
    rng = np.random.default_rng(1)
    worst = 0.0
    for trial in range(6000):
        ang = rng.uniform(0, 2 * np.pi)
        a, b = np.float32(np.cos(ang)), np.float32(np.sin(ang))          # cv.computeCorrespondEpilines normalises (a, b)
        scale = [320.0, 640.0, 16000.0][trial % 3]
        x, y = rng.uniform(0, scale, 2)
        if trial % 5 == 0:                                               # blobs right on the line: the cancelling case
            c = -(float(a) * float(np.float32(x)) + float(b) * float(np.float32(y))) + rng.uniform(-1, 1)
        else:
            c = rng.uniform(-2 * scale, 2 * scale)
        omax = np.float32(max(abs(np.float32(x)), abs(np.float32(y)), rng.uniform(0, scale)))
        worst = max(worst, check(a, b, c, x, y, omax))
    assert 0.05 < worst <= 1.0        # the bound is used (not vacuous) and never exceeded


def test_pretest_bound_holds_for_adversarial_inputs():
    """Coordinates at omax, |c| up to 2 omax (the line through the far corner), float32 neighbours, sign patterns that make the
    inner sum cancel, lines that are NOT normalised (nu = 1 branch of computeCorrespondEpilines: a = b = 0 keeps c)."""
    f32 = np.float32
    vals = [f32(16000.0), np.nextafter(f32(16000.0), f32(0)), f32(8191.999), f32(4096.0), np.nextafter(f32(4096.0), f32(1e9)), f32(0.5), f32(1e-3)]
    lines = [(f32(1.0), f32(0.0)), (f32(0.0), f32(-1.0)), (f32(0.70710677), f32(0.70710677)), (f32(-0.6), f32(0.8)),
             (np.nextafter(f32(1.0), f32(0)), f32(3.4526698e-4)), (f32(0.0), f32(0.0))]
    worst = 0.0
    for a, b in lines:
        for x in vals:
            for y in vals:
                om = max(float(x), float(y))
                for c in (-(float(a) * float(x) + float(b) * float(y)), -2 * om, 2 * om, 0.0, 1e-3, -(float(b) * float(y))):
                    for dc in (0.0, 1e-4, -3e-4):
                        worst = max(worst, check(a, b, f32(c + dc), x, y, f32(om)))
    assert worst <= 1.0
