"""Run under HSA_CU_MASK / ROC_GLOBAL_CU_MASK to see how many (XCD, CU) slots a process really gets."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import _lib
lib = _lib.load()
nwg = 8192
out = torch.full((nwg, 2), -1, dtype=torch.int32, device="cuda:0")
_lib.check(lib.semipd_probe_cu_placement(out.data_ptr(), nwg, 20000, None), "probe")
torch.cuda.synchronize()
slots = {(int(x), int(c)) for x, c in out.cpu().tolist()}
xcds = sorted({x for x, _ in slots})
# bandwidth-ish check: time a big copy
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); b = torch.empty_like(a)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): b.copy_(a)
torch.cuda.synchronize(); dt = (time.time() - t) / 10
x = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): y = x @ x
torch.cuda.synchronize(); dg = (time.time() - t) / 10
print(f"HSA_CU_MASK={os.environ.get('HSA_CU_MASK')} ROC_GLOBAL_CU_MASK={os.environ.get('ROC_GLOBAL_CU_MASK')} "
      f"cu_slots={len(slots)} xcds={xcds} copy={2 * (1 << 30) / dt / 1e9:.0f} GB/s gemm={2 * 8192**3 / dg / 1e12:.0f} TF/s")
