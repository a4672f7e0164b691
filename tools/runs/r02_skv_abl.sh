for a in 0 8 16 32 48 56; do SEMIPD_SKV_ABL=$a timeout 120 python tools/kbench_ext_quick.py 2>&1 | grep "ext=8192"; done
