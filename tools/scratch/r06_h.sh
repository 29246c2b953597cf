#!/bin/bash
# GPU box, round 6 call 8: adjacency-masked resident kernel (hybrid graphs' hidden layers): parity suites, in-process A/B on configuration 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_softmax_fallbacks.py tests/test_gpu_scripted.py tests/test_gpu_attn_resident.py -m gpu -x -q > $O/r06_h_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/r06_h_tests.log
L=$O/r06_masked_resident_ab.log; : > $L
echo "== configuration 3 (32 x 900 pieces, d = 539): A = masked ring kernel (attn_level 1), B = masked resident kernel (attn_level 2)" >> $L
timeout 900 python tools/ab_config.py --config 3 --a attn_level=1 --b attn_level=2 --pairs 10 >> $L 2>&1
echo "== configuration 3, d = 90" >> $L
timeout 900 python tools/ab_config.py --config 3 --degree 90 --a attn_level=1 --b attn_level=2 --pairs 10 >> $L 2>&1
cat $L
timeout 900 python bench.py --config 3 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('config3', d['ms_per_step'], d['value'], d.get('two_batches_in_flight'))"
