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


def test_fused_decode_launch_rule(monkeypatch):
    """HipAttnBackend.fused_decode_waves / fused_decode_zsplits: the RoPE + KV store + attention + split merge launch
    (csrc/decode_attention_fused.hip): one workgroup per (request, kv head) from about half a workgroup per CU up, several
    per pair (and stage 2 behind them) below that; four waves per workgroup from two workgroups per CU; never for MLA, one q
    head per kv head, a fixed --triton-attention-num-kv-splits, or when switched off."""
    import types
    import torch
    from semi_pd_amd.layers.attention_backend import HipAttnBackend

    def backend(heads, kv_heads, cus, kind="mha", fixed=None, kv_dtype=None):
        mr = types.SimpleNamespace(device="cpu", num_attention_heads_local=heads, num_kv_heads_local=kv_heads, v_head_dim=128,
                                   req_to_token_pool=types.SimpleNamespace(req_to_token=None), max_context_len=8192,
                                   kv_geometry={"kind": kind}, num_kv_splits=fixed, num_cus_owned=cus, dtype=torch.bfloat16,
                                   kv_cache_dtype=kv_dtype or torch.bfloat16)
        return HipAttnBackend(mr)

    monkeypatch.delenv("SEMIPD_FUSED_DECODE_ATTN", raising=False)
    b = backend(32, 8, 256)                                   # Llama-3-8B on the whole chip
    # 8 waves from the smallest batch (several workgroups per pair below 16 requests), 4 waves from two workgroups per CU
    assert [b.fused_decode_waves(n, 128, 32) for n in (1, 15, 16, 32, 63, 64, 256)] == [8, 8, 8, 8, 8, 4, 4]
    assert [b.fused_decode_zsplits(n, s) for n, s in ((1, 32), (8, 32), (8, 17), (15, 17), (16, 16), (32, 8))] == [4, 4, 3, 3, 1, 1]
    assert backend(32, 8, 96).fused_decode_zsplits(6, 16) == 1           # a 96-CU share: one workgroup per pair from 6 requests
    r70 = backend(8, 1, 256)                                             # the 70B TP = 8 rank: one kv head
    assert r70.fused_decode_waves(32, 128, 32) == 8 and r70.fused_decode_zsplits(32, 32) == 4
    assert [r70.fused_decode_waves(n, 128, 32) for n in (1, 4, 7, 8)] == [0, 0, 0, 8]   # under cus / 8 workgroups: the separate launches
    assert backend(8, 1, 256).fused_decode_zsplits(128, 16) == 1
    assert backend(32, 8, 256, kv_dtype=torch.float8_e4m3fn).fused_decode_waves(32, 128) == 8   # fp8 pool rows
    assert backend(32, 8, 256).fused_decode_waves(32, 96) == 0          # head size without an instantiation
    assert backend(12, 12, 256).fused_decode_waves(64, 64) == 0         # MHA: the shuffle kernel's shape
    assert backend(128, 1, 256, kind="mla").fused_decode_waves(64, 128) == 0
    assert backend(32, 8, 256, fixed=16).fused_decode_waves(32, 128) == 0
    monkeypatch.setenv("SEMIPD_FUSED_DECODE_ATTN", "0")
    assert b.fused_decode_waves(32, 128) == 0
    monkeypatch.setenv("SEMIPD_FUSED_DECODE_ATTN", "2")                  # one workgroup per pair whatever the batch
    assert b.fused_decode_waves(1, 128) == 8 and b.fused_decode_zsplits(1, 32) == 1 and b.fused_decode_waves(256, 128) == 4
    monkeypatch.setenv("SEMIPD_FUSED_DECODE_ATTN", "4")                  # only where one workgroup per pair fills the chip
    assert [b.fused_decode_waves(n, 128) for n in (1, 15, 16, 64)] == [0, 0, 8, 4]
