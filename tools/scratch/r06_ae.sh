#!/bin/bash
# GPU box: verification of HEAD after the virtual-rows change -- whole GPU suite, smoke, the driver's exact bench command, rocprof + PMC of the scripted Batch
# (its kernels changed), refreshed csr line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > $O/r06_final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/r06_final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r06_final_smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/r06_bench_driver_command.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_driver_command.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","scaling","vs_baseline")})
print("roofline", {k:d["roofline"][k] for k in ("bound","kernel","achieved","peak","unit","frac","traffic")})
print("cpu_baseline", {k:d["cpu_baseline"][k] for k in ("value","unit","cores","kind")})
PY
ROUND=r06 bash tools/collect_profiles.sh scripted --config scripted --no-train-side
timeout 900 python bench.py --config scripted > $O/r06_bench_scripted.json 2> $O/r06_bench_scripted.err; echo "bench scripted rc=$?"
timeout 900 python bench.py --config csr > $O/r06_bench_csr.json 2> $O/r06_bench_csr.err; echo "bench csr rc=$?"
python - <<'PY'
import json
for t in ("scripted","csr"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    f=d.get("batches_in_flight") or {}
    r=d["roofline"]
    print(t, round(d["ms_per_step"],4), round(d["value"],1), {k:round(v["ms_per_batch_step"],4) for k,v in f.items() if k in ("2","4")}, r.get("kernel"), round(r["frac"],4), r.get("traffic"))
PY
