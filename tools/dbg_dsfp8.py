import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd"), os.path.join(ROOT, "tests")]
import torch
from test_gpu_deepseek import tiny_deepseek, server_args, make_prompts
from oracle.model import OracleDeepseekV2
from semi_pd_amd.entrypoints.engine import Engine
from semi_pd_amd.managers.io_struct import SamplingParams
qc = {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic"}
cfg = tiny_deepseek(quantization_config=qc)
prompts = make_prompts(cfg.vocab_size, [5, 37, 130, 1, 64, 17])
sp = SamplingParams(max_new_tokens=4, ignore_eos=True)
for graph in (False, True):
    eng = Engine(server_args(cfg, disable_cuda_graph=not graph))
    sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
    outs = eng.generate(prompts, sp)
    a = eng.model_runner.model.model.layers[0].self_attn
    print("graph", graph, "w_scale", a.w_scale, a.w_kc.dtype, tuple(a.w_kc.shape), tuple(a.w_vc.shape))
    eng.shutdown()
    for absorb in (True, False):
        oracle = OracleDeepseekV2(cfg, sd, act_dtype=torch.bfloat16, absorb_fp8=absorb)
        _, logits = oracle.generate(prompts, 4, forced=outs)
        worst = []
        for b, toks in enumerate(outs):
            for s, t in enumerate(toks):
                row = logits[b, s]
                worst.append((float(row.max() - row[t]), b, s))
        worst.sort(reverse=True)
        print("  oracle absorb_fp8 =", absorb, "worst gaps", [(round(g, 3), b, s) for g, b, s in worst[:5]])
