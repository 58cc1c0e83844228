"""Host-side arithmetic the HIP kernels rely on, restated in numpy and checked exhaustively (no GPU needed)."""
import numpy as np


def _div24_make(d):
    """radix_sort.h: rs_div24_make — floor(x / d) for x < 2^24 as ((x << 8) * mul >> 32) >> sh"""
    L = 0
    while (1 << L) < d:
        L += 1
    m = ((1 << (24 + L)) + d - 1) // d
    return m << 7, L + 7


def test_div24_is_exact_for_every_24_bit_numerator():
    # the MSD-first sort's generated pass divides 24-bit part numbers (three symbols as a number) by the alphabet size,
    # its square and the bucket span with one multiply-high (radix_sort.h: rs_div24); a wrong quotient puts a suffix into
    # the wrong bucket.  Every numerator below 2^24 for alphabets from 2 to 128 symbols, their squares and typical spans.
    x = np.arange(1 << 24, dtype=np.uint64)
    divisors = sorted(set([2, 3, 5, 7, 16, 50, 64, 65, 76, 95, 96, 97, 127, 128, 204, 255, 256, 1000, 4096, 5776, 9025, 9216, 9409,
                           16129, 16384, 65535, 1 << 20, (1 << 24) - 1]))
    for d in divisors:
        mul, sh = _div24_make(d)
        assert mul < (1 << 32), d
        q = (((x << np.uint64(8)) * np.uint64(mul)) >> np.uint64(32)) >> np.uint64(sh)
        assert np.array_equal(q, x // np.uint64(d)), d


def test_pair_form_split_of_the_key_space():
    # top digit = (first two symbols as a number A) / span with span = floor(2^32 / B^4): M = span * B^4 <= 2^32, at most 256
    # buckets for the alphabets the pair form accepts in practice, and key - top * M < M for every 6-symbol key
    rng = np.random.default_rng(1)
    for B in (5, 27, 65, 76, 96, 128):
        P4 = B ** 4
        span = min((1 << 32) // P4, B * B)
        M = span * P4
        assert span >= 1 and M <= (1 << 32)
        buckets = -(-B * B // span)
        if buckets > 256:
            continue  # (the build then takes key >> 32 as the top digit)
        codes = rng.integers(0, B, size=(100000, 6)).astype(object)
        key = sum(codes[:, i] * (B ** (5 - i)) for i in range(6))
        A = codes[:, 0] * B + codes[:, 1]
        top = A // span
        rest = key - top * M
        assert all(0 <= int(r) < M for r in rest[:2000]) and int(max(top)) < 256
        # order: (top, rest) sorts like the key
        order = np.lexsort((np.array([int(r) for r in rest]), np.array([int(t) for t in top])))
        ks = np.array([int(k) for k in key])[order]
        assert np.all(ks[:-1] <= ks[1:])
