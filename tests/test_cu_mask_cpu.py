"""The CU shares of the two instances (the reference sets CUDA_MPS_ACTIVE_THREAD_PERCENTAGE at
entrypoints/engine.py:591-593, 632-634): host arithmetic only -- which logical CUs a percentage means, that the default
pair is a hard partition, that every share takes the same number of CUs from each of the 8 XCDs, and the HSA_CU_MASK text."""
import pytest

from semi_pd_amd.semi_pd.utils import cu_mask_env, cu_mask_words


def bits_of(words, n):
    return [i for i in range(n) if words[i >> 5] >> (i & 31) & 1]


@pytest.mark.parametrize("num_cus", [256, 304, 64])
@pytest.mark.parametrize("p,d", [(62, 38), (50, 50), (75, 25), (88, 12)])
def test_prefill_from_the_bottom_and_decode_from_the_top_partition_the_device(num_cus, p, d):
    pre = bits_of(cu_mask_words(num_cus, p, False), num_cus)
    dec = bits_of(cu_mask_words(num_cus, d, True), num_cus)
    assert pre == list(range(len(pre))) and dec == list(range(num_cus - len(dec), num_cus))   # contiguous ends
    assert not set(pre) & set(dec), "the two shares overlap"
    # granule: 32 logical CUs on a 256-CU device (one per shader engine of every XCD: the dispatcher deals workgroups to the
    # shader engines round-robin, so a share with unequal engines runs at the pace of its smallest one), 8 otherwise
    g = 32 if (num_cus % 32 == 0 and num_cus >= 256) else 8
    for share in (pre, dec):
        assert len(share) % g == 0 and len(share) >= g
        per_xcd = [sum(1 for i in share if i % 8 == x) for x in range(8)]   # logical CU i lives on XCD i % 8
        assert len(set(per_xcd)) == 1, per_xcd
    # whole granules: the shares may leave at most one granule per side unclaimed by rounding, never claim more than asked + one
    assert abs(len(pre) - num_cus * p / 100) < g and abs(len(dec) - num_cus * d / 100) < g
    assert len(pre) + len(dec) <= num_cus


def test_default_shares_of_an_mi355x():
    assert len(bits_of(cu_mask_words(256, 62, False), 256)) == 160
    assert len(bits_of(cu_mask_words(256, 38, True), 256)) == 96
    assert cu_mask_env(0, 256, 62, False) == {"HSA_CU_MASK": "0:0-159"}
    assert cu_mask_env(0, 256, 38, True) == {"HSA_CU_MASK": "0:160-255"}
    assert cu_mask_env(5, 256, 50, True) == {"HSA_CU_MASK": "5:128-255"}
    assert cu_mask_env(3, 256, 100, True) == {}                       # a whole device needs no mask
    assert len(bits_of(cu_mask_words(256, 1, False), 256)) == 32      # never less than one CU per shader engine of every XCD
    # the reference's own percentages (prefill 80 %, nested in decode 100 %): 192 CUs, not 208 -- a 6.5-CU-per-engine share
    # runs like a 6-CU one (profiles/r05_hbm_probe_cu_ranges.txt), and 88 % is 224
    assert len(bits_of(cu_mask_words(256, 80, False), 256)) == 192
    assert len(bits_of(cu_mask_words(256, 88, False), 256)) == 224
    assert len(bits_of(cu_mask_words(256, 75, False), 256)) == 192 and len(bits_of(cu_mask_words(256, 25, True), 256)) == 64
    env = cu_mask_env(1, 304, 50, True, library_grid=True)            # the library's stream-K grids learn the share too
    assert env == {"HSA_CU_MASK": "1:152-303", "TENSILE_STREAMK_MAX_CUS": "152"}


def test_defaults_of_the_reference_environment_variables():
    """SEMI_PD_PREFILL_SM_PERCENTILE / SEMI_PD_DECODE_SM_PERCENTILE (semi_pd/utils.py:10-11) exist; the defaults are nested
    shares like the reference's (prefill 80 %, decode 100 %) at the share the MI355X measurements chose: 88 % = 224 CUs with
    the decode-step deadline (semi_pd/utils.py)."""
    from semi_pd_amd.semi_pd import utils
    assert (utils.PREFILL_ENGINE_SM_PERCENTILE, utils.DECODE_ENGINE_SM_PERCENTILE) == (88, 100) or \
        "SEMI_PD_PREFILL_SM_PERCENTILE" in __import__("os").environ
