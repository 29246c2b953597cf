sample() { sleep 3; for i in 1 2 3; do rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | head -1 | tr '\n' ' '; sleep 1; done; echo; }
run() { tag=$1; shift; ( "$@" > /tmp/kp.log 2>&1 ) & pid=$!; echo -n "$tag: "; sample; wait $pid; tail -1 /tmp/kp.log | cut -c1-150; }
export STORE_PROBE_ITERS=200000
run "stores only (mode 1, 295 MB)"   tools/bin/store_probe 57600 2560 1 512 1
run "MFMA only (mode 13)"            tools/bin/store_probe 57600 2560 1 512 13
run "MFMA + stores (mode 12)"        tools/bin/store_probe 57600 2560 1 512 12
run "loads only (mode 6)"            tools/bin/store_probe 57600 2560 1 512 6
export STORE_PROBE_ITERS=20
run "gemm xpanel conv3"              env PACKED=1 python tools/gemm_probe_loop2.py 57600 256 2560 60000
