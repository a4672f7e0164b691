"""What brings a process back after a failed stream capture on this ROCm? (dev probe)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
x = torch.ones(1024, device=dev)
raw = ctypes.c_void_p()
print("hipStreamCreateWithFlags rc", hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1))
stream = torch.cuda.ExternalStream(raw.value, device=dev)
stream.wait_stream(torch.cuda.current_stream())

def ok(tag):
    try:
        v = float((x + 1).cpu().sum())
        print(tag, "-> works", v, flush=True)
        return True
    except Exception as e:
        print(tag, "-> still broken:", str(e).splitlines()[0], flush=True)
        return False

g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=stream):
        y = x * 2
        y.sum().item()
except Exception as e:
    print("capture failed as intended:", str(e).splitlines()[0], flush=True)
ok("right after")
graph = ctypes.c_void_p()
print("hipStreamEndCapture rc", hip.hipStreamEndCapture(raw, ctypes.byref(graph)))
print("hipStreamDestroy rc", hip.hipStreamDestroy(raw))
print("hipGetLastError", hip.hipGetLastError())
ok("after destroying the capture stream")
raw2 = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(raw2), 1)
s2 = torch.cuda.ExternalStream(raw2.value, device=dev)
s2.wait_stream(torch.cuda.current_stream())
try:
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s2):
        y2 = x * 3
    g2.replay(); torch.cuda.synchronize()
    print("fresh capture on a new stream ok", float(y2.sum()))
except Exception as e:
    print("fresh capture failed:", str(e).splitlines()[0])
ok("at the end")
os._exit(0)
