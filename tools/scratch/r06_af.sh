#!/bin/bash
# GPU box: whole GPU suite on HEAD (after the counter-name fix), the driver's exact command once more
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > $O/r06_final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/r06_final_gpu_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/r06_bench_driver_command.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_driver_command.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
