"""Generate tests/golden/*.npz from the REFERENCE itself.

Run only in the build container, where /root/reference exists:
    python tests/golden/make_golden.py                 # every fixture, one subprocess per generator
    python tests/golden/make_golden.py decode extend   # just these, in this process
    SEMIPD_GOLDEN_OUT=/tmp/g python tests/golden/make_golden.py   # write somewhere else (tests/test_golden_regen_cpu.py)
(one process per generator because some reference modules can be loaded only once per interpreter: `gen_fp8` loads
fp8_kernel.py stand-alone and `gen_silu` imports it again through the package, which registers the same torch op twice)
It imports the reference's leaf modules (stubbing the packages that are not installed, SURVEY.md
Appendix A), runs them on CPU (Triton kernels under TRITON_INTERPRET=1) on seeded inputs and
stores inputs + outputs.  Only data is written to the repo; no reference source travels.
bf16 tensors are stored as their uint16 bit patterns (numpy has no bf16).
"""
import os
import sys
import types
import importlib.abc
import importlib.machinery

os.environ["TRITON_INTERPRET"] = "1"
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/python")

MISSING = {"IPython", "zmq", "setproctitle", "orjson", "uvloop", "decord", "interegular", "llguidance",
           "xgrammar", "outlines", "torchao", "vllm", "sgl_kernel", "flashinfer", "semi_pd_ipc",
           "modelscope", "multipart", "torchvision", "aiter"}


class _AnyMeta(type):
    """Fabricated classes answer any class attribute with another fabricated class (e.g. vllm's
    scalar_types.uint4b8, read at import time by quantization/gptq.py)."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _AnyMeta(name, (), {"__init__": lambda s, *a, **k: None})


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _AnyMeta(name, (), {"__init__": lambda s, *a, **k: None})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Finder())
import triton.runtime.cache as _c  # noqa: E402

for _n in ("default_cache_dir", "default_dump_dir", "default_override_dir"):
    if not hasattr(_c, _n):
        setattr(_c, _n, lambda: "/tmp/triton_cache")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import sglang.srt.utils as U  # noqa: E402

U.is_cuda_available = lambda: True

from sglang.srt.layers.layernorm import RMSNorm  # noqa: E402
from sglang.srt.layers import rotary_embedding as RE  # noqa: E402
from sglang.srt.layers.attention.triton_ops.decode_attention import decode_attention_fwd  # noqa: E402
from sglang.srt.layers.attention.triton_ops.extend_attention import extend_attention_fwd  # noqa: E402
from sglang.srt.layers.attention.utils import create_flashinfer_kv_indices_triton  # noqa: E402
from sglang.srt.layers.moe import topk as TK  # noqa: E402
from sglang.srt.mem_cache.memory_pool import ReqToTokenPool, TokenToKVPoolAllocator  # noqa: E402

OUT = os.environ.get("SEMIPD_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))


def bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach()
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


# ---------------------------------------------------------------- RMSNorm (layernorm.py:59-76)
def gen_rmsnorm():
    g = torch.Generator().manual_seed(0)
    arrs = {}
    for ci, (T, H, dt) in enumerate([(3, 64, torch.bfloat16), (5, 111, torch.float16), (2, 256, torch.float32),
                                     (4, 128, torch.bfloat16)]):
        x = torch.randn(T, H, generator=g).to(dt)
        r = torch.randn(T, H, generator=g).to(dt)
        w = (torch.randn(H, generator=g) * 0.5 + 1).to(dt)
        m = RMSNorm(H, eps=1e-5)
        m.weight.data = w.clone()
        y = m.forward_native(x.clone())
        y2, r2 = m.forward_native(x.clone(), r.clone())
        tag = f"c{ci}_"
        arrs.update({tag + "x": bits(x), tag + "r": bits(r), tag + "w": bits(w), tag + "y": bits(y),
                     tag + "y_fused": bits(y2), tag + "r_fused": bits(r2),
                     tag + "dtype": np.array(str(dt))})
    arrs["eps"] = np.array(1e-5)
    arrs["n"] = np.array(4)
    save("rmsnorm", **arrs)


# ---------------------------------------------------------------- RoPE (rotary_embedding.py)
def gen_rope():
    g = torch.Generator().manual_seed(1)
    arrs = {}
    cases = []
    cases.append(("base_neox", RE.RotaryEmbedding(64, 64, 128, 10000, True, torch.float32), 64))
    cases.append(("base_gptj", RE.RotaryEmbedding(64, 64, 128, 10000, False, torch.float32), 64))
    cases.append(("partial_neox", RE.RotaryEmbedding(64, 32, 128, 10000, True, torch.float32), 64))
    cases.append(("llama3", RE.Llama3RotaryEmbedding(128, 128, 256, 500000, True, torch.float32, 8.0, 1.0,
                                                     4.0, 64), 128))
    cases.append(("deepseek_yarn", RE.DeepseekScalingRotaryEmbedding(
        64, 64, 64, 10000, False, 4.0, torch.float32, extrapolation_factor=1, attn_factor=1,
        beta_fast=32, beta_slow=1, mscale=0.707, mscale_all_dim=0.707, device="cpu"), 64))
    for name, mod, head in cases:
        cache = mod.cos_sin_cache.float()
        T, Hq, Hk = 7, 4, 2
        pos = torch.randint(0, cache.shape[0], (T,), generator=g)
        q = torch.randn(T, Hq * head, generator=g)
        k = torch.randn(T, Hk * head, generator=g)
        if name == "deepseek_yarn":
            qo, ko = mod.forward(pos, q.view(T, Hq, head).clone(), k.view(T, Hk, head).clone())
            qo, ko = qo.reshape(T, -1), ko.reshape(T, -1)
        else:
            qo, ko = mod.forward_native(pos, q.clone(), k.clone())
        arrs.update({f"{name}_cache": cache.numpy(), f"{name}_pos": pos.numpy(), f"{name}_q": q.numpy(),
                     f"{name}_k": k.numpy(), f"{name}_qo": qo.numpy(), f"{name}_ko": ko.numpy(),
                     f"{name}_head": np.array(head), f"{name}_neox": np.array(bool(mod.is_neox_style))})
    arrs["names"] = np.array([c[0] for c in cases])
    save("rope", **arrs)


# ---------------------------------------------------------------- kv pool helpers
def make_paged(g, B, lens, Hkv, Dk, Dv, dtype, extra=7):
    total = int(sum(lens))
    N = total + extra
    perm = torch.randperm(N - 1, generator=g)[:total] + 1  # slot 0 reserved
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    kv_indices = perm.to(torch.int32)
    k_buf = torch.randn(N, Hkv, Dk, generator=g).to(dtype)
    v_buf = torch.randn(N, Hkv, Dv, generator=g).to(dtype)
    return k_buf, v_buf, kv_indptr, kv_indices


# ---------------------------------------------------------------- decode attention (Triton, interpreter)
def gen_decode():
    g = torch.Generator().manual_seed(2)
    arrs = {}
    cases = [
        # name, B, lens, Hq, Hkv, Dk, Dv, splits, logit_cap
        ("gqa4_d64", 3, [5, 33, 70], 8, 2, 64, 64, 4, 0.0),
        ("mha_d32", 2, [1, 19], 2, 2, 32, 32, 2, 0.0),
        ("gqa8_d128_cap", 2, [17, 40], 8, 1, 128, 128, 8, 30.0),
        ("mla_like", 2, [9, 21], 4, 1, 96, 64, 3, 0.0),
    ]
    for name, B, lens, Hq, Hkv, Dk, Dv, splits, cap in cases:
        dt = torch.float32
        k_buf, v_buf, kv_indptr, kv_indices = make_paged(g, B, lens, Hkv, Dk, Dv, dt)
        q = torch.randn(B, Hq, Dk, generator=g).to(dt)
        o = torch.zeros(B, Hq, Dv, dtype=dt)
        logits = torch.zeros(B, Hq, splits, Dv + 1, dtype=torch.float32)
        sm_scale = 1.0 / (Dk ** 0.5)
        decode_attention_fwd(q, k_buf, v_buf, o, kv_indptr, kv_indices, logits, splits, sm_scale, cap)
        arrs.update({f"{name}_q": q.numpy(), f"{name}_k": k_buf.numpy(), f"{name}_v": v_buf.numpy(),
                     f"{name}_indptr": kv_indptr.numpy(), f"{name}_indices": kv_indices.numpy(),
                     f"{name}_o": o.numpy(), f"{name}_logits": logits.numpy(),
                     f"{name}_meta": np.array([splits, sm_scale, cap], dtype=np.float64)})
    arrs["names"] = np.array([c[0] for c in cases])
    save("decode_attention", **arrs)


def gen_decode_8c():
    """The shapes SURVEY 8(c) lists for the reference's Triton decode kernels: MLA 576 / 512 through the grouped
    kernel (_fwd_grouped_kernel_stage1, decode_attention.py:234-390), head sizes 80 and 13 (BLOCK_DMODEL padding),
    GQA group 16, 16 kv splits, a logit cap on the grouped path."""
    g = torch.Generator().manual_seed(12)
    arrs = {}
    cases = [
        # name, B, lens, Hq, Hkv, Dk, Dv, splits, logit_cap
        ("mla_576_512_h16", 2, [37, 90], 16, 1, 576, 512, 4, 0.0),
        ("mla_576_512_h128", 1, [70], 128, 1, 576, 512, 2, 0.0),
        ("gqa4_d80", 2, [23, 51], 8, 2, 80, 80, 4, 0.0),
        ("mha_d13", 2, [9, 30], 3, 3, 13, 13, 2, 0.0),
        ("gqa16_d64_splits16", 2, [100, 257], 16, 1, 64, 64, 16, 0.0),
        ("gqa8_d128_splits16_cap", 1, [300], 16, 2, 128, 128, 16, 50.0),
        ("mha_d64_splits16", 2, [64, 129], 4, 4, 64, 64, 16, 0.0),
    ]
    for name, B, lens, Hq, Hkv, Dk, Dv, splits, cap in cases:
        dt = torch.float32
        k_buf, v_buf, kv_indptr, kv_indices = make_paged(g, B, lens, Hkv, Dk, Dv, dt)
        if name.startswith("mla"):
            v_buf = k_buf[..., :Dv]   # the latent row is K; V is its first 512 columns (memory_pool.py:439-452)
        q = torch.randn(B, Hq, Dk, generator=g).to(dt)
        o = torch.zeros(B, Hq, Dv, dtype=dt)
        logits = torch.zeros(B, Hq, splits, Dv + 1, dtype=torch.float32)
        sm_scale = 1.0 / (Dk ** 0.5)
        decode_attention_fwd(q, k_buf, v_buf, o, kv_indptr, kv_indices, logits, splits, sm_scale, cap)
        arrs.update({f"{name}_q": q.numpy(), f"{name}_k": k_buf.numpy(),
                     f"{name}_indptr": kv_indptr.numpy(), f"{name}_indices": kv_indices.numpy(),
                     f"{name}_o": o.numpy(),
                     f"{name}_meta": np.array([splits, sm_scale, cap, Dv], dtype=np.float64)})
        if not name.startswith("mla"):
            arrs[f"{name}_v"] = v_buf.numpy()
    arrs["names"] = np.array([c[0] for c in cases])
    save("decode_attention_8c", **arrs)


def gen_extend_8c():
    """extend_attention_fwd at the MLA prefill shape (Dk 192 / Dv 128, extend_attention.py:291-410 with
    BLOCK_DPE), head sizes 80 and 13, GQA group 16."""
    g = torch.Generator().manual_seed(13)
    arrs = {}
    cases = [
        ("mla_prefill_192_128", [0, 21], [40, 17], 4, 4, 192, 128, 0.0),
        ("mla_absorbed_576_512", [11], [19], 4, 1, 576, 512, 0.0),
        ("gqa2_d80", [5, 0], [33, 12], 4, 2, 80, 80, 0.0),
        ("mha_d13", [3], [20], 2, 2, 13, 13, 0.0),
        ("gqa16_d64", [30], [70], 16, 1, 64, 64, 0.0),
    ]
    for name, pre, ext, Hq, Hkv, Dk, Dv, cap in cases:
        dt = torch.float32
        B = len(pre)
        k_buf, v_buf, kv_indptr, kv_indices = make_paged(g, B, pre, Hkv, Dk, Dv, dt)
        T = sum(ext)
        qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
        qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
        q = torch.randn(T, Hq, Dk, generator=g).to(dt)
        k = torch.randn(T, Hkv, Dk, generator=g).to(dt)
        v = torch.randn(T, Hkv, Dv, generator=g).to(dt)
        o = torch.zeros(T, Hq, Dv, dtype=dt)
        sm_scale = 1.0 / (Dk ** 0.5)
        extend_attention_fwd(q, k, v, o, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, None, None,
                             max(ext), sm_scale, cap)
        arrs.update({f"{name}_q": q.numpy(), f"{name}_k": k.numpy(), f"{name}_v": v.numpy(),
                     f"{name}_kbuf": k_buf.numpy(), f"{name}_vbuf": v_buf.numpy(),
                     f"{name}_qo_indptr": qo_indptr.numpy(), f"{name}_kv_indptr": kv_indptr.numpy(),
                     f"{name}_kv_indices": kv_indices.numpy(), f"{name}_o": o.numpy(),
                     f"{name}_meta": np.array([sm_scale, cap], dtype=np.float64)})
    arrs["names"] = np.array([c[0] for c in cases])
    save("extend_attention_8c", **arrs)


# ---------------------------------------------------------------- SiLU * mul (layers/activation.py:41-44)
def gen_silu():
    from sglang.srt.layers.activation import SiluAndMul
    g = torch.Generator().manual_seed(14)
    arrs = {}
    m = SiluAndMul()
    for i, (T, d, dt) in enumerate([(3, 64, torch.bfloat16), (5, 88, torch.float16), (2, 256, torch.float32),
                                    (7, 1408, torch.bfloat16)]):
        x = (torch.randn(T, 2 * d, generator=g) * 3).to(dt)
        y = m.forward_native(x)
        arrs[f"c{i}_x"], arrs[f"c{i}_y"] = bits(x) if dt == torch.bfloat16 else x.numpy(), \
            bits(y) if dt == torch.bfloat16 else y.numpy()
        arrs[f"c{i}_dtype"] = np.array(str(dt))
    arrs["n"] = np.array(4)
    save("silu_and_mul", **arrs)


# ---------------------------------------------------------------- moe_align_block_size (Triton reference of the test)
def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_moe_align():
    """sgl-kernel/tests/test_moe_align.py compares the CUDA op with a four-stage Triton implementation; that
    Triton implementation is run here (interpreter) on the test's own kind of input (randperm ids per token)."""
    TM = _load_by_path("ref_test_moe_align", "/root/reference/sgl-kernel/tests/test_moe_align.py")
    g = torch.Generator().manual_seed(15)
    arrs = {}
    cases = [(32, 7, 2, 8), (64, 33, 6, 64), (16, 1, 8, 64), (128, 50, 8, 160), (64, 130, 4, 16)]
    for i, (bs, T, k, E) in enumerate(cases):
        topk_ids = torch.stack([torch.randperm(E, generator=g, dtype=torch.int32)[:k] for _ in range(T)])
        max_padded = topk_ids.numel() + E * (bs - 1)
        sorted_ids = torch.full((max_padded,), topk_ids.numel(), dtype=torch.int32)
        expert_ids = torch.zeros((max_padded // bs,), dtype=torch.int32)
        n_post = torch.empty((1,), dtype=torch.int32)
        TM.moe_align_block_size_triton(topk_ids, E, bs, sorted_ids, expert_ids, n_post)
        arrs.update({f"c{i}_topk_ids": topk_ids.numpy(), f"c{i}_sorted": sorted_ids.numpy(),
                     f"c{i}_expert_ids": expert_ids.numpy(), f"c{i}_n_post": n_post.numpy(),
                     f"c{i}_meta": np.array([bs, T, k, E])})
    arrs["n"] = np.array(len(cases))
    save("moe_align", **arrs)


# ---------------------------------------------------------------- fused MoE (fused_moe_native.py, test_fused_moe.py)
def _reference_torch_naive_moe():
    """test/srt/test_fused_moe.py::TestFusedMOE.torch_naive_moe (the formula the reference's own test pins
    fused_moe to): its module cannot be imported (vLLM's fused_moe), so exactly that method is compiled from the
    file at generation time (nothing of it is stored)."""
    import ast
    from sglang.srt.layers.activation import SiluAndMul
    path = "/root/reference/test/srt/test_fused_moe.py"
    tree = ast.parse(open(path).read())
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) for n in c.body
              if isinstance(n, ast.FunctionDef) and n.name == "torch_naive_moe")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "SiluAndMul": _native_silu_and_mul()}
    exec(compile(mod, path, "exec"), ns)
    return lambda *a: ns["torch_naive_moe"](None, *a)


def _native_silu_and_mul():
    """CustomOp.__call__ dispatches to forward_cuda here (is_cuda_available is patched for the import), whose
    sgl_kernel op is a stub: pick the reference's own forward_native instead — dispatch only, same class."""
    from sglang.srt.layers.activation import SiluAndMul

    class NativeSiluAndMul(SiluAndMul):
        def __call__(self, x):
            return self.forward_native(x)
    return NativeSiluAndMul


def gen_fused_moe():
    from types import SimpleNamespace as NS
    import sglang.srt.layers.moe.fused_moe_native as FN
    from sglang.srt.layers.moe.fused_moe_native import fused_moe_forward_native, moe_forward_native
    FN.SiluAndMul = _native_silu_and_mul()
    naive = _reference_torch_naive_moe()
    g = torch.Generator().manual_seed(16)
    arrs = {}
    cases = [(5, 32, 48, 8, 2), (33, 48, 64, 64, 6), (1, 64, 32, 8, 2)]   # m, n (intermediate), k (hidden), e, topk
    for i, (m, n, k, e, topk) in enumerate(cases):
        a = torch.randn(m, k, generator=g) * 0.5
        w1 = torch.randn(e, 2 * n, k, generator=g) * 0.2
        w2 = torch.randn(e, k, n, generator=g) * 0.2
        score = torch.randn(m, e, generator=g)
        layer = NS(w13_weight=w1, w2_weight=w2, num_experts=e)
        out_naive = naive(a.clone(), w1, w2, score.clone(), topk)
        # select_experts(torch_native=True, renormalize=False) = softmax + topk, the routing of torch_naive_moe
        out_native = fused_moe_forward_native(layer, a.clone(), False, topk, score.clone(), False)
        out_grouped = moe_forward_native(layer, a.clone(), False, topk, score.clone(), True)
        arrs.update({f"c{i}_a": a.numpy(), f"c{i}_w1": w1.numpy(), f"c{i}_w2": w2.numpy(), f"c{i}_score": score.numpy(),
                     f"c{i}_out_naive": out_naive.numpy(), f"c{i}_out_native": out_native.numpy(),
                     f"c{i}_out_renorm": out_grouped.numpy(), f"c{i}_meta": np.array([m, n, k, e, topk])})
    arrs["n"] = np.array(len(cases))
    save("fused_moe", **arrs)


# ---------------------------------------------------------------- extend attention (Triton, interpreter)
def gen_extend():
    g = torch.Generator().manual_seed(3)
    arrs = {}
    cases = [
        # name, prefix lens, extend lens, Hq, Hkv, Dk, Dv, logit_cap
        ("gqa_d64", [0, 13, 40], [20, 7, 70], 4, 2, 64, 64, 0.0),
        ("mha_d32_noprefix", [0, 0], [5, 130], 2, 2, 32, 32, 0.0),
        ("gqa_d128_cap", [9], [33], 4, 1, 128, 128, 25.0),
        ("dk96_dv64", [6, 0], [10, 3], 2, 1, 96, 64, 0.0),
    ]
    for name, pre, ext, Hq, Hkv, Dk, Dv, cap in cases:
        dt = torch.float32
        B = len(pre)
        k_buf, v_buf, kv_indptr, kv_indices = make_paged(g, B, pre, Hkv, Dk, Dv, dt)
        T = sum(ext)
        qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
        qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
        q = torch.randn(T, Hq, Dk, generator=g).to(dt)
        k = torch.randn(T, Hkv, Dk, generator=g).to(dt)
        v = torch.randn(T, Hkv, Dv, generator=g).to(dt)
        o = torch.zeros(T, Hq, Dv, dtype=dt)
        sm_scale = 1.0 / (Dk ** 0.5)
        extend_attention_fwd(q, k, v, o, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, None, None,
                             max(ext), sm_scale, cap)
        arrs.update({f"{name}_q": q.numpy(), f"{name}_k": k.numpy(), f"{name}_v": v.numpy(),
                     f"{name}_kbuf": k_buf.numpy(), f"{name}_vbuf": v_buf.numpy(),
                     f"{name}_qo_indptr": qo_indptr.numpy(), f"{name}_kv_indptr": kv_indptr.numpy(),
                     f"{name}_kv_indices": kv_indices.numpy(), f"{name}_o": o.numpy(),
                     f"{name}_meta": np.array([sm_scale, cap], dtype=np.float64)})
    arrs["names"] = np.array([c[0] for c in cases])
    save("extend_attention", **arrs)


# ---------------------------------------------------------------- kv indices (Triton) + pools
def gen_kv_indices():
    g = torch.Generator().manual_seed(4)
    R, C = 9, 40
    req_to_token = torch.randint(1, 1000, (R, C), generator=g, dtype=torch.int32)
    req_pool_indices = torch.tensor([3, 0, 7, 5], dtype=torch.int64)
    lens = torch.tensor([5, 40, 1, 17], dtype=torch.int64)
    start = torch.tensor([0, 3, 0, 2], dtype=torch.int32)  # start <= len for every request
    outs = {}
    for tag, st, ln in (("nostart", None, lens), ("start", start, lens - start.long())):
        kv_indptr = torch.zeros(5, dtype=torch.int32)
        kv_indptr[1:] = torch.cumsum(ln, 0)
        kv_indices = torch.zeros(int(kv_indptr[-1]), dtype=torch.int32)
        create_flashinfer_kv_indices_triton[(4,)](req_to_token, req_pool_indices, ln, kv_indptr, st,
                                                  kv_indices, req_to_token.stride(0))
        outs[tag + "_lens"] = ln.numpy()
        outs[tag + "_indptr"] = kv_indptr.numpy()
        outs[tag + "_indices"] = kv_indices.numpy()
    # pool behaviour (memory_pool.py:46-184): alloc / free ordering
    pool = ReqToTokenPool(6, 16, "cpu", False)
    a = pool.alloc(2)
    b = pool.alloc(3)
    pool.free(a)
    c = pool.alloc(3)
    over = pool.alloc(5)
    alloc = TokenToKVPoolAllocator(20, torch.bfloat16, "cpu", None)
    x = alloc.alloc(5)
    y = alloc.alloc(7)
    alloc.free(x)
    z = alloc.alloc(10)
    w = alloc.alloc(100)
    save("kv_indices", req_to_token=req_to_token.numpy(), req_pool_indices=req_pool_indices.numpy(),
         start=start.numpy(), pool_a=np.array(a), pool_b=np.array(b), pool_c=np.array(c),
         pool_over=np.array(-1 if over is None else 0), alloc_x=x.numpy(), alloc_y=y.numpy(),
         alloc_z=z.numpy(), alloc_w=np.array(-1 if w is None else 0),
         alloc_avail=np.array(alloc.available_size()), **outs)


# ---------------------------------------------------------------- MoE routing (topk.py)
def gen_topk():
    g = torch.Generator().manual_seed(5)
    arrs = {}
    T = 9
    # fused_topk_native: E=8 top2 ; E=64 top6
    for name, E, k, ren in (("native_e8", 8, 2, True), ("native_e64", 64, 6, False)):
        gate = torch.randn(T, E, generator=g)
        w, ids = TK.fused_topk_native(torch.zeros(T, 4), gate, k, ren)
        arrs.update({name + "_gate": gate.numpy(), name + "_w": w.numpy(), name + "_ids": ids.numpy().astype(np.int32),
                     name + "_meta": np.array([k, int(ren)])})
    # grouped_topk: E=64, 8 groups top 3, k=6, softmax & sigmoid, fp32 and bf16 gating
    for name, dt, scoring in (("grouped_f32", torch.float32, "softmax"), ("grouped_bf16", torch.bfloat16, "softmax"),
                              ("grouped_sigmoid", torch.float32, "sigmoid")):
        gate = torch.randn(T, 64, generator=g).to(dt)
        w, ids = TK.grouped_topk(torch.zeros(T, 4), gate, 6, True, 8, 3, scoring)
        arrs.update({name + "_gate": bits(gate), name + "_w": w.numpy(), name + "_ids": ids.numpy(),
                     name + "_meta": np.array([6, 1, 8, 3])})
    # biased_grouped_topk: E=64, 8 groups top 4, k=8 (DeepSeek-V3 style, scaled down)
    for name, dt in (("biased_f32", torch.float32), ("biased_bf16", torch.bfloat16)):
        gate = torch.randn(T, 64, generator=g).to(dt)
        bias = torch.randn(64, generator=g) * 0.1
        w, ids = TK.biased_grouped_topk(torch.zeros(T, 4), gate, bias, 8, True, 8, 4)
        arrs.update({name + "_gate": bits(gate), name + "_bias": bias.numpy(), name + "_w": w.numpy(),
                     name + "_ids": ids.numpy(), name + "_meta": np.array([8, 1, 8, 4])})
    save("moe_topk", **arrs)


def _reference_sampler_functions():
    """layers/sampler.py cannot be imported here (its module imports pull the scheduler), but the two
    torch-native helpers it defines are self-contained: compile just those two function definitions
    from the reference file at generation time (nothing of it is stored)."""
    import ast
    path = "/root/reference/python/sglang/srt/layers/sampler.py"
    tree = ast.parse(open(path).read())
    want = {"top_k_top_p_min_p_sampling_from_probs_torch", "top_p_normalize_probs_torch"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want],
                     type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["top_k_top_p_min_p_sampling_from_probs_torch"], ns["top_p_normalize_probs_torch"]


def gen_sampling():
    """Empirical token counts of the reference's torch-native sampler (sampler.py:207-243) and the
    output of top_p_normalize_probs_torch, on small seeded distributions."""
    sample_fn, top_p_norm = _reference_sampler_functions()
    g = torch.Generator().manual_seed(11)
    B, V, draws = 6, 48, 4000
    logits = torch.randn(B, V, generator=g) * 2.0
    temps = torch.tensor([1.0, 0.7, 1.3, 1.0, 0.5, 2.0]).view(-1, 1)
    probs = torch.softmax(logits / temps, dim=-1)
    top_ks = torch.tensor([1 << 30, 5, 12, 1, 1 << 30, 20], dtype=torch.int32)
    top_ps = torch.tensor([1.0, 1.0, 0.8, 1.0, 0.6, 0.95])
    min_ps = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    min_ps2 = torch.tensor([0.05, 0.2, 0.0, 0.0, 0.1, 0.02])
    arrs = {"logits": logits.numpy(), "temperatures": temps.numpy(), "probs": probs.numpy(),
            "top_ks": top_ks.numpy(), "top_ps": top_ps.numpy(), "min_ps": min_ps2.numpy(),
            "draws": np.array([draws])}
    torch.manual_seed(5)
    for name, mp, need in (("counts", min_ps, False), ("counts_min_p", min_ps2, True)):
        counts = torch.zeros(B, V, dtype=torch.int64)
        for _ in range(draws):
            ids = sample_fn(probs.clone(), top_ks, top_ps, mp, need)
            counts[torch.arange(B), ids.long()] += 1
        arrs[name] = counts.numpy()
    arrs["top_p_normalized"] = top_p_norm(probs.clone(), top_ps).numpy()
    save("sampling", **arrs)


def _reference_fp8_kernels():
    """layers/quantization/__init__.py imports every quantisation method (vLLM's scalar types included), so
    fp8_kernel.py is loaded as a stand-alone module.  Its HIP branch is written for MI300 (e4m3fnuz, 224):
    the golden vectors are made on the other branch, OCP e4m3fn / 448, the fp8 of gfx950."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_fp8_kernel", "/root/reference/python/sglang/srt/layers/quantization/fp8_kernel.py")
    FK = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(FK)
    FK._is_hip = False
    FK._is_cuda = False
    FK.get_device_name = lambda *a, **k: "cpu"
    return FK


def _reference_native_fp8_helpers():
    """The torch helpers the reference's own test compares its kernels with (test/test_block_fp8.py:19-44):
    compiled from the file at generation time, nothing of it is stored."""
    import ast
    path = "/root/reference/python/sglang/test/test_block_fp8.py"
    tree = ast.parse(open(path).read())
    want = {"native_per_token_group_quant_fp8"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["native_per_token_group_quant_fp8"]


def f8bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.uint8).numpy()


def gen_fp8():
    """per_token_group_quant_fp8 and w8a8_block_fp8_matmul: the reference's Triton kernels run by the Triton
    interpreter (fp8_kernel.py:75-115, 409-491)."""
    FK = _reference_fp8_kernels()
    native_quant = _reference_native_fp8_helpers()
    g = torch.Generator().manual_seed(21)
    arrs = {}
    cases = []
    for i, (rows, hidden, group, dtype) in enumerate([(3, 256, 128, torch.float32), (7, 512, 64, torch.bfloat16),
                                                       (5, 1024, 256, torch.float16), (2, 1024, 512, torch.bfloat16),
                                                       (9, 384, 128, torch.bfloat16)]):
        x = (torch.randn(rows, hidden, generator=g) * (10.0 ** torch.randint(-3, 3, (rows, 1), generator=g))).to(dtype)
        x[0, :group] = 0  # an all-zero group: the eps path
        if rows > 1:
            x[1, 3] = 30000.0 if dtype != torch.float16 else 6e4  # one outlier sets the scale of its group
        q, s = FK.per_token_group_quant_fp8(x, group, dtype=torch.float8_e4m3fn)
        arrs[f"quant{i}_x"] = bits(x) if dtype == torch.bfloat16 else x.numpy()
        arrs[f"quant{i}_q"] = f8bits(q)
        arrs[f"quant{i}_s"] = s.numpy()
        # the Triton INTERPRETER casts f32 -> fp8 with its own numpy code (ties away from zero, and a value
        # that rounds up into the next binade comes out halved); the torch helper casts with torch (RNE)
        qn, sn = native_quant(x, group)
        arrs[f"quant{i}_q_native"] = f8bits(qn)
        arrs[f"quant{i}_s_native"] = sn.numpy()
        cases.append([rows, hidden, group, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]])
    arrs["quant_cases"] = np.array(cases)
    mm = []
    for i, (M, N, K, out_dtype) in enumerate([(1, 144, 256, torch.float32), (7, 512, 384, torch.bfloat16),
                                               (83, 200, 1024, torch.float16), (5, 128, 128, torch.bfloat16),
                                               (33, 272, 400, torch.float32)]):
        A = (torch.rand(M, K, generator=g) - 0.5) * 2 * 448
        B = (torch.rand(N, K, generator=g) - 0.5) * 2 * 448
        Aq = A.clamp(-448, 448).to(torch.float8_e4m3fn)
        Bq = B.clamp(-448, 448).to(torch.float8_e4m3fn)
        kt, nt = (K + 127) // 128, (N + 127) // 128
        As = torch.rand(M, kt, generator=g) * 1e-2
        Bs = torch.rand(nt, kt, generator=g) * 1e-2
        C = FK.w8a8_block_fp8_matmul(Aq, Bq, As, Bs, [128, 128], out_dtype)
        arrs[f"mm{i}_a"], arrs[f"mm{i}_b"] = f8bits(Aq), f8bits(Bq)
        arrs[f"mm{i}_as"], arrs[f"mm{i}_bs"] = As.numpy(), Bs.numpy()
        arrs[f"mm{i}_c"] = bits(C) if out_dtype == torch.bfloat16 else C.numpy()
        mm.append([M, N, K, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[out_dtype]])
    arrs["mm_cases"] = np.array(mm)
    save("block_fp8", **arrs)


def _reference_fp8_utils_functions():
    """layers/quantization/fp8_utils.py imports the CUDA / HIP kernel packages at module level; its two tensor-wise
    helpers are plain torch: compile exactly those two function definitions from the file at generation time, on the
    OCP branch (_is_hip = False: finfo.max = 448, what gfx950 implements; the HIP branch there is MI300's fnuz / 224)."""
    import ast
    from typing import List, Tuple
    path = "/root/reference/python/sglang/srt/layers/quantization/fp8_utils.py"
    tree = ast.parse(open(path).read())
    want = {"input_to_float8", "block_quant_to_tensor_quant"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"torch": torch, "List": List, "Tuple": Tuple, "_is_hip": False}
    exec(compile(mod, path, "exec"), ns)
    return ns["input_to_float8"], ns["block_quant_to_tensor_quant"]


def gen_bmm_fp8():
    """input_to_float8 and block_quant_to_tensor_quant of the reference (fp8_utils.py:137-188), run here; and the
    inputs of a bmm_fp8 call shaped like the MLA absorption (models/deepseek_v2.py:659-665, 690-700) with the
    product the reference's own test compares against (torch.bmm of the unquantised operands,
    sgl-kernel/tests/test_bmm_fp8.py:33-44)."""
    to_f8, block_to_tensor = _reference_fp8_utils_functions()
    g = torch.Generator().manual_seed(31)
    arrs = {}
    # 1. input_to_float8 on a transposed bf16 view (q_nope.transpose(0, 1)) and on an fp32 tensor
    q_nope = (torch.randn(7, 4, 128, generator=g) * 3).to(torch.bfloat16)
    x = q_nope.transpose(0, 1)
    qv, sc = to_f8(x, torch.float8_e4m3fn)
    arrs.update(q_nope=bits(q_nope), q_nope_f8=f8bits(qv), q_nope_scale_inv=sc.numpy())
    y = torch.randn(3, 5, 64, generator=g) * 100
    y[1, 2, 3] = 7e4
    yv, ys = to_f8(y, torch.float8_e5m2)
    arrs.update(y=y.numpy(), y_f8=f8bits(yv), y_scale_inv=ys.numpy())
    # 2. block_quant_to_tensor_quant on a block-quantised kv_b_proj-like weight
    wq = (torch.rand(384, 256, generator=g) * 2 - 1).mul(448).to(torch.float8_e4m3fn)
    ws = torch.rand(3, 2, generator=g) * 1e-2 + 1e-3
    tq, ts = block_to_tensor(wq, ws, [128, 128])
    arrs.update(w_block_q=f8bits(wq), w_block_s=ws.numpy(), w_tensor_q=f8bits(tq), w_tensor_scale_inv=ts.numpy())
    # 3. bmm inputs: A [h, T, 128] x W_kc [h, 128, 512] (column-major), unquantised product for the cos-sim bar
    a = torch.randn(4, 9, 128, generator=g).to(torch.bfloat16)
    b = torch.randn(4, 512, 128, generator=g).to(torch.bfloat16)          # memory [h, n, k]
    a8, a_s = to_f8(a, torch.float8_e4m3fn)
    b8, b_s = to_f8(b, torch.float8_e4m3fn)
    ref = torch.bmm(a.float(), b.float().transpose(1, 2))
    arrs.update(bmm_a=bits(a), bmm_b=bits(b), bmm_a8=f8bits(a8), bmm_b8=f8bits(b8), bmm_a_s=a_s.numpy(), bmm_b_s=b_s.numpy(),
                bmm_ref_unquantised=ref.numpy())
    save("bmm_fp8", **arrs)


def gen_penalties():
    """The reference's batched penalizers (sampling/penaltylib/*.py) driven the way ScheduleBatch drives them:
    an orchestrator per new prefill batch, apply -> sample -> cumulate every step, merge into the running batch
    (orchestrator.py merge, before reqs is extended), filter when a request finishes.  Stored: raw logits, the
    penalised logits and the sampled ids of every step plus which requests formed the batch."""
    from types import SimpleNamespace as NS
    import sglang.srt.sampling.penaltylib as PL
    g = torch.Generator().manual_seed(8)
    V, EOS = 64, 7
    # frequency, presence, min_new_tokens, stop ids (-1 padded)
    params = [(0.7, -0.3, 0, []), (0.0, 0.0, 0, []), (0.0, 0.0, 5, [11, 12]), (0.0, 1.5, 2, []), (2.0, 0.0, 0, [3])]

    def req(i):
        f, p, m, stops = params[i]
        return NS(sampling_params=NS(frequency_penalty=f, presence_penalty=p, min_new_tokens=m,
                                     stop_token_ids=set(stops) or None),
                  tokenizer=NS(additional_stop_token_ids=None, eos_token_id=EOS), idx=i)

    def orch(reqs):
        # the reference passes a SET of classes (schedule_batch.py), whose iteration order -- the order in which the
        # penalties are subtracted, i.e. the last bit of the result -- changes from process to process; a tuple in a
        # fixed order makes the fixture reproducible (presence, min_new_tokens, frequency: the order the committed fixture was made with)
        order = [int(c) for c in os.environ.get("SEMIPD_PENALIZER_ORDER", "210")]
        classes = (PL.BatchedFrequencyPenalizer, PL.BatchedMinNewTokensPenalizer, PL.BatchedPresencePenalizer)
        return PL.BatchedPenalizerOrchestrator(V, NS(reqs=reqs, device="cpu"), tuple(classes[i] for i in order))

    arrs = {"params": np.array([[f, p, m] for f, p, m, _ in params], dtype=np.float64),
            "stops": np.array([(s + [-1, -1])[:2] for *_, s in params], dtype=np.int64), "eos": np.array(EOS)}
    step = [0]

    def run(o):
        rows = [r.idx for r in o.batch.reqs]
        logits = torch.randn(len(rows), V, generator=g)
        want = logits.clone()
        o.apply(want)
        ids = torch.randint(0, 16, (len(rows),), generator=g)   # a small range, so that tokens repeat
        k = step[0]
        arrs[f"rows_{k}"], arrs[f"logits_{k}"], arrs[f"want_{k}"], arrs[f"ids_{k}"] = (
            np.array(rows), logits.numpy(), want.numpy(), ids.numpy())
        step[0] += 1
        o.cumulate_output_tokens(ids)

    a = orch([req(0), req(1), req(2)])
    for _ in range(4):                   # the prefill sample and three decode steps
        run(a)
    b = orch([req(3), req(4)])
    run(b)                               # the new batch's prefill sample
    a.merge(b)
    a.batch.reqs.extend(b.batch.reqs)
    for _ in range(3):
        run(a)
    keep = [1, 2, 3, 4]                  # request 0 finishes
    a.batch.reqs = [a.batch.reqs[i] for i in keep]
    a.filter(torch.tensor(keep))
    for _ in range(3):
        run(a)
    keep = [0, 2]                        # the requests with min_new_tokens / frequency penalty finish
    a.batch.reqs = [a.batch.reqs[i] for i in keep]
    a.filter(torch.tensor(keep))
    for _ in range(2):
        run(a)
    arrs["n_steps"] = np.array(step[0])
    save("penalties", **arrs)


# ---------------------------------------------------------------- extend attention under a custom mask
def tree_mask(g, pre, ext, prefix_bits):
    """[ext][pre + ext] bool: a sub-causal triangle with the diagonal kept (every token sees itself: the shape of a
    speculative tree mask, triton_backend.py:136-149), prefix columns all ones or random (first column kept)."""
    tri = torch.tril(torch.rand(ext, ext, generator=g) < 0.55)
    tri |= torch.eye(ext, dtype=torch.bool)
    if prefix_bits:
        pm = torch.rand(ext, pre, generator=g) < 0.6
        if pre:
            pm[:, 0] = True
    else:
        pm = torch.ones(ext, pre, dtype=torch.bool)
    return torch.cat([pm, tri], 1)


def gen_extend_mask():
    """extend_attention_fwd with custom_mask / mask_indptr / skip_prefix_custom_mask (extend_attention.py:291-307,
    :164-177, :233-252; the reference's own test builds the mask the same way, test_triton_attention_kernels.py:121-139)."""
    g = torch.Generator().manual_seed(11)
    arrs = {}
    cases = [
        # name, prefix lens, extend lens, Hq, Hkv, Dk, Dv, logit_cap, skip_prefix_custom_mask
        ("tree_gqa_d64_skip", [0, 13, 40], [20, 7, 70], 4, 2, 64, 64, 0.0, True),
        ("tree_gqa_d64_prefix_bits", [5, 13, 40], [20, 7, 70], 4, 2, 64, 64, 0.0, False),
        ("tree_d128_skip", [9, 0], [33, 140], 4, 1, 128, 128, 0.0, True),
        ("tree_d128_cap_prefix_bits", [70], [66], 2, 1, 128, 128, 25.0, False),
        ("tree_dk96_dv64", [6, 0], [10, 3], 2, 1, 96, 64, 0.0, False),
        ("causal_as_mask_mha_d32", [3, 0], [5, 130], 2, 2, 32, 32, 0.0, True),
        ("tree_mla_576_512", [11, 2], [6, 14], 4, 1, 576, 512, 0.0, False),   # the absorbed-MLA row: BLOCK_DMODEL 512 + DPE 64
    ]
    for name, pre, ext, Hq, Hkv, Dk, Dv, cap, skip in cases:
        dt = torch.float32
        B = len(pre)
        k_buf, v_buf, kv_indptr, kv_indices = make_paged(g, B, pre, Hkv, Dk, Dv, dt)
        T = sum(ext)
        qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
        qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
        q = torch.randn(T, Hq, Dk, generator=g).to(dt)
        k = torch.randn(T, Hkv, Dk, generator=g).to(dt)
        v = torch.randn(T, Hkv, Dv, generator=g).to(dt)
        o = torch.zeros(T, Hq, Dv, dtype=dt)
        sm_scale = 1.0 / (Dk ** 0.5)
        blocks = []
        for b in range(B):
            if name.startswith("causal_as_mask"):
                blocks.append(torch.cat([torch.ones(ext[b], pre[b], dtype=torch.bool),
                                         torch.tril(torch.ones(ext[b], ext[b], dtype=torch.bool))], 1).flatten())
            else:
                blocks.append(tree_mask(g, pre[b], ext[b], not skip).flatten())
        custom_mask = torch.cat(blocks)
        mask_indptr = torch.zeros(B + 1, dtype=torch.int64)
        mask_indptr[1:] = torch.cumsum(torch.tensor([m.numel() for m in blocks]), 0)
        extend_attention_fwd(q, k, v, o, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, custom_mask, mask_indptr,
                             max(ext), sm_scale, cap, skip)
        assert torch.isfinite(o).all(), name
        arrs.update({f"{name}_q": q.numpy(), f"{name}_k": k.numpy(), f"{name}_v": v.numpy(),
                     f"{name}_kbuf": k_buf.numpy(), f"{name}_vbuf": v_buf.numpy(),
                     f"{name}_qo_indptr": qo_indptr.numpy(), f"{name}_kv_indptr": kv_indptr.numpy(),
                     f"{name}_kv_indices": kv_indices.numpy(), f"{name}_o": o.numpy(),
                     f"{name}_mask": custom_mask.numpy().astype(np.uint8), f"{name}_mask_indptr": mask_indptr.numpy(),
                     f"{name}_meta": np.array([sm_scale, cap, 1.0 if skip else 0.0], dtype=np.float64)})
    arrs["names"] = np.array([c[0] for c in cases])
    save("extend_attention_mask", **arrs)


GENERATORS = {"rmsnorm": gen_rmsnorm, "rope": gen_rope, "kv_indices": gen_kv_indices, "topk": gen_topk,
              "decode": gen_decode, "extend": gen_extend, "sampling": gen_sampling, "fp8": gen_fp8,
              "penalties": gen_penalties, "decode_8c": gen_decode_8c, "extend_8c": gen_extend_8c, "extend_mask": gen_extend_mask, "silu": gen_silu,
              "moe_align": gen_moe_align, "fused_moe": gen_fused_moe, "bmm_fp8": gen_bmm_fp8}


if __name__ == "__main__":
    names = sys.argv[1:]
    unknown = [n for n in names if n not in GENERATORS]
    if unknown:
        raise SystemExit(f"unknown generator(s) {unknown}; known: {sorted(GENERATORS)}")
    if names:
        for n in names:
            GENERATORS[n]()
    else:
        # everything: a fresh interpreter per generator
        import subprocess
        os.makedirs(OUT, exist_ok=True)
        failed = []
        for n in GENERATORS:
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), n])
            if rc != 0:
                failed.append(n)
        if failed:
            raise SystemExit(f"generators failed: {failed}")
        print("all", len(GENERATORS), "generators ran")
