#!/bin/bash
# GPU box: virtual rows in the masked attention's launch (small Batches) -- hybrid / exophormer / sampler / train suites on the product build, scripted + config 3 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_benched_mode.py tests/test_gpu_parity.py tests/test_gpu_samplers.py tests/test_gpu_train.py -m gpu -x -q > $O/r06_ac_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_ac_tests.log
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "bench $tag rc=$?"; }
b scripted --config scripted
b config_3 --config 3
python - <<'PY'
import json
for t in ("scripted","config_3"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    f=d.get("two_batches_in_flight") or d.get("batches_in_flight") or {}
    print(t, round(d["ms_per_step"],4), round(d["value"],1), {k:round(v["ms_per_batch_step"],4) for k,v in f.items() if k in ("2","4")} or (round(f.get("ms_per_batch_step",0),4) if f else None))
    c=d["roofline"].get("classes",{}); print({k:round(v["us_per_step"],1) for k,v in c.items()})
PY
