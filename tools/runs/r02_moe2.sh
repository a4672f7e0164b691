for a in 0 1 3; do echo abl=$a; SEMIPD_MTG_ABL=$a timeout 300 python tools/kbench_moe_stages.py 2>&1 | grep "T=8192"; done
