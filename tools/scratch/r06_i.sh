#!/bin/bash
# GPU box, round 6 call 10: final code -- whole GPU suite, smoke, and the bench lines touched by the last kernel changes (virtual rows' kernel)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > $O/r06_final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/r06_final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r06_final_smoke.log
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "bench $tag rc=$?"; }
b scripted --config scripted
b config_3 --config 3
b config_3_d90 --config 3 --degree 90
python - <<'PY'
import json
for t in ("scripted","config_3","config_3_d90"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    print(t, d["ms_per_step"], d["value"], d.get("two_batches_in_flight"))
PY
