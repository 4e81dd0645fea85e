"""The arithmetic of the x6 kernels, restated in numpy (no GPU): an fp32 value is cut into three bf16 pieces by truncation
(proba-v_amd/csrc/x6_device.h::pieces) and a product is the sum of the six largest piece products (mac6).  These tests pin
the two claims DESIGN.md §4.1 rests on: the cut is exact, and the six-product sum is as accurate as an fp32 product."""
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
