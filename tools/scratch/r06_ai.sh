#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --config 5 > gpurun_out/r06_bench_config_5.json 2> gpurun_out/r06_bench_config_5.err; echo rc=$?
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_config_5.json') if l.startswith('{')][-1])
print(d['ms_per_step'], json.dumps(d['gradient_exchange_prediction'], indent=0))"
timeout 900 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 > gpurun_out/r06_bench_config_5_exophormer_d539.json 2>/dev/null; echo rc=$?
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_config_5_exophormer_d539.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['gradient_exchange_prediction']['by_gpus'])"
