"""choose_kv_splits (layers/attention_backend.py): the split-KV factor of decode attention as a function of the batch and
the CUs the process OWNS (the reference fixes it: --triton-attention-num-kv-splits, server_args.py:321-323).  Host
arithmetic; the measurements behind the rule are cited in its comments."""
from semi_pd_amd.layers.attention_backend import MLA_SHARED_MIN_WORKGROUPS, choose_kv_splits


def test_gqa_fills_one_round_of_waves_on_the_share():
    # one work item = one wave, 8 per CU: B x Hkv x splits ~ 8 x CUs
    assert choose_kv_splits(32, 8, 8192, 256, 32) == 8
    assert choose_kv_splits(30, 8, 8192, 96, 32) == 3          # the decode share of the default policy at its usual batch
    assert choose_kv_splits(30, 8, 8192, 128, 32) == 4
    assert choose_kv_splits(256, 8, 8192, 256, 32) == 1         # a full batch already is a round
    assert choose_kv_splits(1, 8, 8192, 256, 32) == 32          # the cap
    assert choose_kv_splits(1, 8, 8192, 256, 16) == 16
    # never a split shorter than 64 tokens
    assert choose_kv_splits(1, 8, 100, 256, 32) == 1
    assert choose_kv_splits(1, 8, 640, 256, 32) == 10
    # monotone: more CUs never mean fewer splits, a larger batch never more
    for b in (1, 4, 30, 64):
        got = [choose_kv_splits(b, 8, 4096, c, 32) for c in (64, 96, 128, 160, 256)]
        assert got == sorted(got)
    for c in (96, 256):
        got = [choose_kv_splits(b, 8, 4096, c, 32) for b in (1, 2, 8, 30, 64, 256)]
        assert got == sorted(got, reverse=True)


def test_mla_sixteen_heads_and_the_shared_tile_kernel():
    # 16 heads per rank: one work item = one 4-wave workgroup, 2 per CU, splits of at least 128 tokens
    assert choose_kv_splits(32, 1, 8192, 256, 32, mla=True, mla_heads=16) == 16
    assert choose_kv_splits(32, 1, 1100, 256, 32, mla=True, mla_heads=16) == 8
    assert choose_kv_splits(45, 1, 8192, 128, 32, mla=True, mla_heads=16) == 5
    # 128 heads: the shared-tile kernel (one workgroup per CU, >= 256 tokens per split) from 160 workgroups up ...
    s = choose_kv_splits(128, 1, 8192, 256, 32, mla=True, mla_heads=128)
    assert s == 2 and 128 * s >= MLA_SHARED_MIN_WORKGROUPS
    assert choose_kv_splits(32, 1, 8192, 256, 32, mla=True, mla_heads=128) == 8
    # ... below that the rule of the 16-head kernels
    assert choose_kv_splits(32, 1, 1100, 256, 32, mla=True, mla_heads=128) == choose_kv_splits(32, 1, 1100, 256, 32, mla=True, mla_heads=16)
