for f in 1 2; do echo "form $f"; SEMIPD_EXTEND_KV_FORM=$f python tools/kbench_ext_scan.py 2>&1 | grep "ext=" | head -3; done
echo "old kernel"; SEMIPD_EXTEND_SHARED_KV=0 python tools/kbench_ext_scan.py 2>&1 | grep "ext=" | head -3
