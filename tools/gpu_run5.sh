timeout 600 python tools/x3_check.py 512 1.4e-8 2>&1 | tail -20
timeout 300 python tools/x3_check.py 512 1e-3 2>&1 | tail -8
