#!/bin/bash
# GPU box: last verification of the round on HEAD -- whole GPU suite, smoke, the driver's exact command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > $O/r06_final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/r06_final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r06_final_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/r06_bench_driver_command.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_driver_command.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
