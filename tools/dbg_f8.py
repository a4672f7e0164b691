import sys; sys.path[:0]=['/root/repo','/root/repo/semi-pd_amd']
import torch
from oracle import ops as O
from semi_pd_amd import ops
dev=torch.device("cuda:0")
torch.manual_seed(1)
for dt, f8 in ((torch.bfloat16, torch.float8_e5m2), (torch.float16, torch.float8_e4m3fn), (torch.bfloat16, torch.float8_e4m3fn)):
    x = (torch.randn(5, 33, 64) * 7).to(dt); x[2,3,4]=300.0
    qo, so = O.input_to_float8(x, f8)
    qg, sg = ops.input_to_float8(x.to(dev), f8)
    a=qg.view(torch.uint8).cpu(); b=qo.view(torch.uint8)
    bad=(a!=b).nonzero()
    print(dt, f8, "scale", float(sg), float(so), "mismatch", bad.shape[0], "of", a.numel())
    for i in bad[:5]:
        i=tuple(i.tolist()); print("  at", i, "x", float(x[i]), "gpu", int(a[i]), "ref", int(b[i]), "x*scale", float(x[i].float()*(1/so)))
