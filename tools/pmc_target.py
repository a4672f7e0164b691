"""One fixed launch shape of the dominant kernels, for rocprofv3 --pmc passes (HBM traffic check).
Usage: python tools/pmc_target.py [decode|decode32|decode32_fused|mla|fp8mm|...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "decode"
torch.manual_seed(0)
if which == "decode":
    B, ctx, Hq, Hkv, D, splits = 128, 4096, 32, 8, 128, 1
    N = B * ctx + 1
    kb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
    vb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    for _ in range(5):
        ops.decode_attention_fwd(q, kb, vb, o, indptr, idx, None, splits, D ** -0.5)
    print("algorithmic_bytes_per_launch", B * ctx * Hkv * 2 * D * 2 + 2 * B * Hq * D * 2)
elif which == "decode32":
    # the decode attention as the serving line runs it: B = 32, ctx 1100, Llama-3-8B heads, the split count of the engine
    from semi_pd_amd.layers.attention_backend import choose_kv_splits
    B, ctx, Hq, Hkv, D = 32, 1100, 32, 8, 128
    splits = choose_kv_splits(B, Hkv, 2048, 256, 32)
    N = B * ctx + 1
    sets = []
    for _ in range(3):   # three KV pools in rotation: 3 x 144 MB is beyond the Infinity Cache
        sets.append((torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16), torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)))
    q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    lg = torch.empty(B, Hq, splits, D + 1, device=dev, dtype=torch.float32)
    for i in range(12):
        kb, vb = sets[i % 3]
        ops.decode_attention_fwd(q, kb, vb, o, indptr, idx, lg, splits, D ** -0.5)
    print("splits", splits, "algorithmic_bytes_per_launch", B * ctx * Hkv * 2 * D * 2 + 2 * B * Hq * D * 2)
elif which == "decode32_fused":
    # the same call through the fused decode launch (csrc/decode_attention_fused.hip): planes of a qkv GEMM in, RoPE + KV store +
    # attention + split merge; 8 waves = 8 kv splits per (request, kv head)
    B, ctx, Hq, Hkv, D, K = 32, 1100, 32, 8, 128, 4096
    N = B * ctx + 1
    sets = []
    for _ in range(3):
        sets.append((torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16), torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)))
    x = torch.randn(B, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn((Hq + 2 * Hkv) * D, K, device=dev, dtype=torch.bfloat16) * 0.02
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    loc = idx[(indptr[1:] - 1).long()].to(torch.int64)
    pos = torch.full((B,), ctx - 1, device=dev, dtype=torch.int64)
    inv = 1.0 / (500000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j -> ij", torch.arange(2048, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dev)
    for i in range(12):
        kb, vb = sets[i % 3]
        planes = ops.stream_linear_planes(x, w)
        ops.decode_rope_attention_planes(pos, planes, Hq, Hkv, D, cache, kb, vb, loc, indptr, idx, 8, D ** -0.5)
    print("planes", planes.ksplit, "algorithmic_bytes_per_launch",
          B * ctx * Hkv * 2 * D * 2 + B * Hq * D * 2 + planes.ksplit * B * (Hq + 2 * Hkv) * D * 4 + 2 * B * Hkv * D * 2)
elif which == "mla":
    B, ctx, H, splits = 128, 8192, 16, 4
    N = B * ctx + 1
    kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
    for _ in range(5):
        ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
    print("algorithmic_bytes_per_launch", B * ctx * 576 * 2 + B * H * (576 + 512) * 2)
elif which.startswith("extend"):
    # extend attention, Llama-3-8B heads: "extend1k" = ONE 1024-token request (the serving regime), "extend8k" = 8192
    ext = 1024 if which == "extend1k" else 8192
    Hq, Hkv, D, B = 32, 8, 128, 1
    T = B * ext
    q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    kb = torch.randn(8, Hkv, D, device=dev, dtype=torch.bfloat16)
    qo = torch.arange(B + 1, device=dev, dtype=torch.int32) * ext
    kvp = torch.zeros(B + 1, device=dev, dtype=torch.int32)
    idx = torch.ones(1, device=dev, dtype=torch.int32)
    for _ in range(5):
        ops.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, idx, None, None, ext)
    print("flop_per_launch", 4.0 * Hq * D * B * ext * (ext + 1) / 2)
elif which == "mla128_b32":
    # the same kernel at B = 32, ctx = 8192, 8 splits: 256 workgroups of 32 tiles (fixed cost per workgroup)
    B, ctx, H = 32, 8192, 128
    N = B * ctx + 1
    kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    for splits in (8, 4, 16):
        lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
        for _ in range(4):
            ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
        torch.cuda.synchronize()
elif which == "mla128":
    # MLA decode, 128 heads on one latent tile (mla_decode_shared.hip): B = 128, ctx = 8192, two splits
    B, ctx, H, splits = 128, 8192, 128, 2
    N = B * ctx + 1
    kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
    for _ in range(4):
        ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
    print("rows_bytes_per_call", B * ctx * 1152, "flop", 2.0 * B * ctx * H * (576 + 512))
elif which == "moe8k":
    # the tiled MoE GEMM1 (with the SiLU epilogue) and GEMM2 of DeepSeek-V2-Lite experts at T = 8192
    from semi_pd_amd.layers.moe import fused_experts
    E, k, K, N, T = 64, 6, 2048, 1408, 8192
    w1 = torch.randn(E, 2 * N, K, device=dev, dtype=torch.bfloat16) * 0.02
    w2 = torch.randn(E, K, N, device=dev, dtype=torch.bfloat16) * 0.02
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
    for _ in range(4):
        fused_experts(x, w1, w2, tw, ti)
    print("flop_per_call_gemm1", 2.0 * T * k * 2 * N * K, "gemm2", 2.0 * T * k * N * K)
elif which == "moe1k":
    # ONE prefill request of DeepSeek-V2-Lite through the fused MoE (~96 rows per expert: the 128-row x 512-column geometry
    # of the grouped ping-pong GEMM); the expert weights (738 + 369 MB) exceed the Infinity Cache by themselves
    from semi_pd_amd.layers.moe import fused_experts
    E, k, K, N, T = 64, 6, 2048, 1408, 1024
    w1 = torch.randn(E, 2 * N, K, device=dev, dtype=torch.bfloat16) * 0.02
    w2 = torch.randn(E, K, N, device=dev, dtype=torch.bfloat16) * 0.02
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
    for _ in range(6):
        fused_experts(x, w1, w2, tw, ti)
    print("algorithmic_bytes gemm1", E * 2 * N * K * 2 + T * k * K * 2 + T * k * N * 2, "gemm2",
          E * K * N * 2 + T * k * N * 2 + T * k * K * 2, "flop gemm1", 2.0 * T * k * 2 * N * K, "gemm2", 2.0 * T * k * N * K)
elif which == "stream":
    # the streaming GEMM of a decode batch: Llama-3-8B gate_up + SiLU*mul (235 MB of weights, read once), 6 weight
    # copies in rotation so that the 256 MB Infinity Cache cannot serve a re-read
    M, N, K = 32, 28672, 4096
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(6)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for i in range(12):
        ops.stream_linear(x, ws[i % 6], fuse_silu_mul=True)
    print("algorithmic_bytes_per_launch", N * K * 2 + M * K * 2 + M * (N // 2) * 2)
elif which in ("gemm_tall256", "gemm_tall4k"):
    # the tiled ping-pong GEMM (csrc/gemm8p.hip): gate_up + SiLU*mul of Llama-3-8B at 256 rows (weight stream, 6 weight
    # copies in rotation) or at 4096 rows (matrix bound)
    M = 256 if which == "gemm_tall256" else 4096
    N, K = 28672, 4096
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(6)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for i in range(12):
        ops.gemm_tall(x, ws[i % 6], fuse_silu_mul=True)
    print("algorithmic_bytes_per_launch", N * K * 2 + M * K * 2 + M * (N // 2) * 2, "flop_per_launch", 2.0 * M * N * K)
elif which == "fp8mm":
    # decode-sized block-fp8 linear on a DeepSeek-V3 shape: the fp8 weights (176 MB) are read once
    M, N, K = 32, 24576, 7168
    wq = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
    ws = torch.rand(N // 128, K // 128, device=dev) * 1e-2
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    xq, xs = ops.per_token_group_quant_fp8(x, 128)
    junk = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
    for _ in range(5):
        junk.fill_(1)  # 256 MB: evicts the weights from the Infinity Cache between launches
        ops.w8a8_block_fp8_matmul(xq, wq, xs, ws, [128, 128], torch.bfloat16)
    print("algorithmic_bytes_per_launch", N * K + ws.numel() * 4 + M * K + xs.numel() * 4 + M * N * 2)
torch.cuda.synchronize()
