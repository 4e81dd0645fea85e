"""The arithmetic of the x6 kernels, restated in numpy (no GPU): an fp32 value is cut into three bf16 pieces by truncation
(proba-v_amd/csrc/x6_device.h::pieces) and a product is the sum of the six largest piece products (mac6).  These tests pin
the two claims DESIGN.md §4.1 (docs/notebook_r1-r5.md §4.1) rests on: the cut is exact, and the six-product sum is as accurate as an fp32 product."""
import numpy as np

HI = np.uint32(0xFFFF0000)


def pieces(x):
    x = np.asarray(x, dtype=np.float32)
    p0 = (x.view(np.uint32) & HI).view(np.float32)
    r = x - p0                                   # exact: the low 16 significand bits
    p1 = (r.view(np.uint32) & HI).view(np.float32)
    p2 = r - p1                                  # <= 8 significant bits left
    return p0, p1, p2


def is_bf16(v):
    return np.all((np.asarray(v, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0)


def test_three_truncation_pieces_are_bf16_and_sum_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=20000) * 10.0 ** rng.integers(-20, 20, size=20000),
                        np.float32([0.0, -0.0, 1.0, -1.0, 3.4e38, 1.2e-30, 65504.0, np.pi])]).astype(np.float32)
    p0, p1, p2 = pieces(x)
    assert is_bf16(p0) and is_bf16(p1) and is_bf16(p2)
    np.testing.assert_array_equal(p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64), x.astype(np.float64))
    # same sign, shrinking by >= 2^-8 per piece
    assert np.all(np.abs(p1) <= np.abs(p0) * 2.0 ** -7) and np.all(np.abs(p2) <= np.abs(p0) * 2.0 ** -15)


def test_piece_products_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    a, b = (rng.normal(size=5000).astype(np.float32) for _ in range(2))
    for pa in pieces(a):
        for pb in pieces(b):
            np.testing.assert_array_equal((pa * pb).astype(np.float64), pa.astype(np.float64) * pb.astype(np.float64))


def test_six_product_sum_matches_fp32_accuracy():
    rng = np.random.default_rng(2)
    K = 675                                      # taps x channels of normConv
    a = rng.normal(size=(400, K)).astype(np.float32)
    b = (rng.normal(size=(400, K)) / np.sqrt(K)).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    a0, a1, a2 = pieces(a)
    b0, b1, b2 = pieces(b)
    acc = np.zeros(400, np.float32)
    for k in range(K):                           # fp32 accumulation, smallest terms first inside a k-step (mac6 order)
        for pa, pb in ((a2, b0), (a1, b1), (a0, b2), (a1, b0), (a0, b1), (a0, b0)):
            acc = (acc + pa[:, k] * pb[:, k]).astype(np.float32)
    plain = np.zeros(400, np.float32)
    for k in range(K):
        plain = (plain + a[:, k] * b[:, k]).astype(np.float32)
    scale = np.abs(exact).max()
    e6, e32 = np.abs(acc - exact).max() / scale, np.abs(plain - exact).max() / scale
    assert e6 < 2e-6 and e6 < 3 * e32 + 1e-7, (e6, e32)
    # the dropped products alone: bounded by 3 * 2^-24 per product
    dropped = (a1.astype(np.float64) * b2 + a2.astype(np.float64) * b1 + a2.astype(np.float64) * b2)
    assert np.all(np.abs(dropped) <= 3 * 2.0 ** -22 * np.abs(a.astype(np.float64) * b) + 1e-300)


# ---- H3: two fp16 round-to-nearest pieces of the SCALED value, three products (x6_device.h::cut_pair<H3>, mac<H3>) -----------------
def h3_exp(amax):
    """exponent e with amax * 2^e in [2^14, 2^15) -- x6_device.h::h3_exp on the bit pattern of a non-negative float."""
    E = (np.float32(amax).view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)
    return 0 if E == 0 else min(int(141 - int(E)), 126)


def h3_pieces(x, e):
    v = (np.asarray(x, np.float32) * np.float32(2.0 ** e)).astype(np.float32)     # exact: a power of two
    h0 = v.astype(np.float16)
    h1 = (v - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0, h1


def test_h3_scale_puts_the_maximum_below_fp16_overflow():
    rng = np.random.default_rng(3)
    for amax in np.concatenate([10.0 ** rng.uniform(-30, 30, 200), [1.0, 2.0, 0.99999994, 65504.0, 3.0e38, 1e-38]]).astype(np.float32):
        e = h3_exp(amax)
        s = float(amax) * 2.0 ** e
        assert s < 2.0 ** 15, (amax, e)
        if e < 126 and amax >= np.float32(1.1754944e-38):   # (tensors below 2^-113 keep the largest representable scale; subnormal maxima none)
            assert s >= 2.0 ** 14, (amax, e)
    assert h3_exp(0.0) == 0


def test_h3_two_pieces_carry_24_bits():
    rng = np.random.default_rng(4)
    x = (rng.normal(size=50000) * 10.0 ** rng.uniform(-3, 0, size=50000)).astype(np.float32)    # down to 2^-17 of the maximum and below
    e = h3_exp(np.abs(x).max())
    h0, h1 = h3_pieces(x, e)
    assert np.all(np.isfinite(h0.astype(np.float32))) and np.all(np.isfinite(h1.astype(np.float32)))
    v = x.astype(np.float64) * 2.0 ** e
    err = np.abs(h0.astype(np.float64) + h1.astype(np.float64) - v)
    big = np.abs(v) >= 2.0 ** -2                            # second piece still a normal fp16 number: full relative precision
    assert np.all(err[big] <= 2.0 ** -23 * np.abs(v[big]))
    assert np.all(err <= 2.0 ** -25 + 2.0 ** -23 * np.abs(v))   # below: absolute error of half an fp16 subnormal step (2^-40 of the maximum)
    # piece products are exact in fp32 (11 x 11 significant bits)
    a0, a1 = h0.astype(np.float32)[:5000], h1.astype(np.float32)[:5000]
    b0 = h0.astype(np.float32)[5000:10000]
    for pa in (a0, a1):
        np.testing.assert_array_equal((pa * b0).astype(np.float64), pa.astype(np.float64) * b0.astype(np.float64))


def test_h3_three_product_sum_matches_fp32_accuracy():
    rng = np.random.default_rng(5)
    K = 675
    for sa_, sb_ in ((1.0, 1.0), (3e-6, 4e4), (7e5, 2e-9)):           # the tensors' magnitudes must not matter
        a = (rng.normal(size=(400, K)) * sa_).astype(np.float32)
        b = (rng.normal(size=(400, K)) / np.sqrt(K) * sb_).astype(np.float32)
        exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
        ea, eb = h3_exp(np.abs(a).max()), h3_exp(np.abs(b).max())
        a0, a1 = (p.astype(np.float32) for p in h3_pieces(a, ea))
        b0, b1 = (p.astype(np.float32) for p in h3_pieces(b, eb))
        acc = np.zeros(400, np.float32)
        for k in range(K):                                   # fp32 accumulation, smallest terms first (mac<H3> order)
            for pa, pb in ((a1, b0), (a0, b1), (a0, b0)):
                acc = (acc + pa[:, k] * pb[:, k]).astype(np.float32)
        got = np.ldexp(acc.astype(np.float64), -(ea + eb))
        plain = np.zeros(400, np.float32)
        for k in range(K):
            plain = (plain + a[:, k] * b[:, k]).astype(np.float32)
        scale = np.abs(exact).max()
        e3, e32 = np.abs(got - exact).max() / scale, np.abs(plain - exact).max() / scale
        assert e3 < 2e-6 and e3 < 3 * e32 + 1e-7, (e3, e32)
