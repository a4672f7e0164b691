timeout 300 python tools/kbench.py moe 2>&1 | grep "^moe" | head -3
timeout 300 python tools/kbench.py moe 2>&1 | grep "^moe" | head -3
