for a in 0 8 10 26 24 14 30; do SEMIPD_SKV_ABL=$a timeout 120 python tools/kbench_ext_quick.py 2>&1 | grep ABL; done
