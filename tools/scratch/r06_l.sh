#!/bin/bash
# GPU box: N Batches in flight (scripted, csr), tests of the generalised sample_loop_batches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_scripted.py -m gpu -x -q 2>&1 | tail -2
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "bench $tag rc=$?"; }
b scripted --config scripted
b csr --config csr
python - <<'PY'
import json
for t in ("scripted","csr"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    f=d.get("batches_in_flight") or {}
    print(t, round(d["ms_per_step"],4), d["value"], {k:(round(v["ms_per_batch_step"],4), round(v["vs_one_batch_in_flight"],2)) for k,v in f.items() if k in ("2","4")})
    tr=d.get("training_step_same_batch"); print("   train", json.dumps(tr)[:300] if tr else None)
PY
