#!/bin/bash
# GPU box: after gating the prologue-projection variant into the experiments build -- resident-kernel suite on both builds, benched-mode suite, one bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_attn_resident.py tests/test_gpu_benched_mode.py tests/test_gpu_parity.py -m gpu -x -q > $O/r06_x_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_x_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['ms_per_step'],4), d['roofline']['frac'])"
