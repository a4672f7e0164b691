"""Alias of semi_pd_amd.launch_server (see sglang/__init__.py)."""
import multiprocessing as mp
import sys

from semi_pd_amd.launch_server import main

if __name__ == "__main__":
    mp.set_start_method("spawn", force=True)
    main(sys.argv[1:])
