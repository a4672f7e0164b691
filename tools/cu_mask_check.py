"""Which (XCD, CU) slots does a process / a stream really get?

    python tools/cu_mask_check.py                 # run under HSA_CU_MASK / ROC_GLOBAL_CU_MASK: the process-wide mask
    python tools/cu_mask_check.py --streams       # hipExtStreamCreateWithCUMask streams (the dynamic shares) against the
                                                  # same shares as HSA_CU_MASK of a child process, per-XCD CU counts, and
                                                  # a hipGraph captured elsewhere replayed on the masked stream
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import _lib
lib = _lib.load()


def placement(stream_ptr=None, nwg=8192):
    out = torch.full((nwg, 2), -1, dtype=torch.int32, device="cuda:0")
    _lib.check(lib.semipd_probe_cu_placement(out.data_ptr(), nwg, 20000, stream_ptr), "probe")
    torch.cuda.synchronize()
    return {(int(x), int(c)) for x, c in out.cpu().tolist()}


def per_xcd(slots):
    d = {}
    for x, _ in slots:
        d[x] = d.get(x, 0) + 1
    return [d.get(x, 0) for x in range(8)]


if "--slots" in sys.argv:     # child: print the slot set of this (masked) process
    print(sorted(placement()))
    sys.exit(0)

if "--streams" in sys.argv:
    from semi_pd_amd.semi_pd.utils import cu_mask_env, cu_masked_stream, get_device_sm_count
    n = get_device_sm_count(0)
    full = placement()
    print(f"unmasked: {len(full)} slots, per XCD {per_xcd(full)}")
    for pct, top in ((81, False), (75, False), (62, False), (38, True), (50, False), (50, True)):
        st = cu_masked_stream(0, pct, top)
        s = placement(st.cuda_stream)
        env = cu_mask_env(0, n, pct, top)
        child = subprocess.run([sys.executable, __file__, "--slots"], env={**os.environ, **env}, capture_output=True, text=True)
        e = set(map(tuple, eval(child.stdout.strip().splitlines()[-1]))) if child.returncode == 0 else set()
        print(f"{pct:3d} % from the {'top' if top else 'bottom'}: stream mask {len(s)} slots per XCD {per_xcd(s)} | "
              f"HSA_CU_MASK={env.get('HSA_CU_MASK')} {len(e)} slots per XCD {per_xcd(e)} | same set: {s == e}, "
              f"stream-only {len(s - e)}, env-only {len(e - s)}")
    sys.exit(0)

nwg = 8192
slots = placement(None, nwg)
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
      f"cu_slots={len(slots)} xcds={xcds} per_xcd={per_xcd(slots)} copy={2 * (1 << 30) / dt / 1e9:.0f} GB/s gemm={2 * 8192**3 / dg / 1e12:.0f} TF/s")
