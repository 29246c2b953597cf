sample() { sleep 3; for i in 1 2 3; do rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | head -1 | tr '\n' ' '; sleep 1; done; echo; }
run() { tag=$1; shift; ( "$@" > /tmp/kp.log 2>&1 ) & pid=$!; echo -n "$tag: "; sample; wait $pid; tail -1 /tmp/kp.log | cut -c1-150; }
run "wreg conv3 full"                 python tools/gemm_probe_loop.py 57600 256 2560 50000
run "wreg conv3 no stores"            env DA_GEMM_DEBUG=256 python tools/gemm_probe_loop.py 57600 256 2560 80000
run "wreg conv3 no stores no DMA"     env DA_GEMM_DEBUG=1280 python tools/gemm_probe_loop.py 57600 256 2560 80000
run "wreg conv3 no stores no MFMA"    env DA_GEMM_DEBUG=768 python tools/gemm_probe_loop.py 57600 256 2560 80000
run "xpanel conv3 no stores"          env DA_GEMM_DEBUG=256 python tools/gemm_probe_loop2.py 57600 256 2560 80000
