"""How many processes can spin on each other on one GPU?  Launches `world` ranks of tests/ar_worker.py in
probe mode (a few synchronised 8 KB all-reduces each) under different environments and prints the call
latencies.  The launcher itself never touches the GPU.  Usage: python tools/ar_concurrency_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ar_worker.py")


def run(world, label, env_extra, masks=True, timeout=90, probe=True):
    procs = []
    port = 25000 + (os.getpid() + hash(label)) % 3000
    share = 256 // world // 8 * 8
    for r in range(world):
        env = dict(os.environ)
        if probe:
            env["AR_PROBE"] = "40"
        if masks:
            env["HSA_CU_MASK"] = f"0:{r * share}-{(r + 1) * share - 1}"
        env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    print(f"== world {world} {label}", flush=True)
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n(timed out)"
        lines = [l[:160] for l in out.splitlines() if l.startswith("[rank ") or "timed out" in l or l.startswith("AR_REPORT")]
        print(f"  rank {r}: " + (" | ".join(lines) if lines else out[-300:]), flush=True)


if __name__ == "__main__":
    if "--copies" in sys.argv:
        run(4, "no copies", {})
        run(4, "copies between calls", {"AR_PROBE_COPY": "1"})
        run(4, "copies between calls, HSA_ENABLE_SDMA=0", {"AR_PROBE_COPY": "1", "HSA_ENABLE_SDMA": "0"})
        run(4, "copies between calls, GPU_MAX_HW_QUEUES=1", {"AR_PROBE_COPY": "1", "GPU_MAX_HW_QUEUES": "1"})
        run(8, "copies between calls", {"AR_PROBE_COPY": "1"})
        sys.exit(0)
    if "--full" in sys.argv:
        run(4, "full worker, masks", {}, probe=False, timeout=150)
        run(8, "full worker, masks", {}, probe=False, timeout=150)
        sys.exit(0)
    run(2, "masks", {})
    run(4, "masks", {})
    run(4, "no masks", {}, masks=False)
    run(4, "masks, GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"})
    run(4, "masks, 1 block", {"SEMIPD_AR_MAX_BLOCKS": "1"})
    run(8, "masks, GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"})
