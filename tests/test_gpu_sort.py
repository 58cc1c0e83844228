"""The radix-sort primitive (radix_sort.h) against torch's stable sort, every kernel configuration."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 21, 26, 1, 31, 36, 32, 4, 41])
@pytest.mark.parametrize("val_bytes", [4, 8, 0])
def test_radix_sort_matches_stable_sort(variant, val_bytes):
    import torch
    from coffeedb_amd import capi
    if val_bytes == 0 and variant != 0:
        pytest.skip("key-only sort has one configuration")
    g = torch.Generator(device="cuda").manual_seed(1234 + variant)
    # (bits 3 / 1: eight values / one value — every lane of a wave lands on the same few counters, the
    #  worst case for the one-atomic ranking, whose stability rests on lane-ordered LDS atomics)
    for n, bits in ((1, 8), (255, 13), (4097, 40), (100_003, 61), (1_000_000, 24), (300_000, 64), (500_000, 3),
                    (200_000, 1)):
        hi = (1 << min(bits, 62)) - 1
        keys = torch.randint(0, hi, (n,), dtype=torch.int64, device="cuda", generator=g)
        if bits == 24:
            keys = keys & 0xFF00FF  # constant middle digit -> that pass is skipped
        if bits == 64:
            keys = keys * 4 + 1    # use the top bits (values stay non-negative as uint64 order == int64 here)
            keys = keys & 0x7FFFFFFFFFFFFFFF
        vals = torch.arange(n, dtype=torch.int64 if val_bytes == 8 else torch.int32, device="cuda")
        ref_keys, ref_idx = torch.sort(keys, stable=True)
        k = keys.clone()
        v = vals.clone()
        torch.cuda.synchronize()  # the library runs on its own non-blocking stream (header contract)
        ms, passes = capi.debug_radix_sort(k.data_ptr(), v.data_ptr() if val_bytes else 0, n, val_bytes, bits, variant)
        torch.cuda.synchronize()
        assert torch.equal(k, ref_keys), (n, bits)
        if val_bytes:
            assert torch.equal(v.to(torch.int64), ref_idx), (n, bits)  # stability: ties keep input order
