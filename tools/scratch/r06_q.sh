#!/bin/bash
# GPU box: final verification of HEAD -- whole GPU suite, smoke, the driver's exact bench command, refreshed lines of the configurations whose kernels changed last
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
print("cpu_baseline", {k:d["cpu_baseline"][k] for k in ("value","unit","cores","kind")}, d["cpu_baseline"]["sample"][:80])
PY
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "bench $tag rc=$?"; }
b config_3 --config 3
b config_3_d90 --config 3 --degree 90
b scripted --config scripted
b config_5 --config 5
b config_5_exophormer_d539 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16
python - <<'PY'
import json
for t in ("config_3","config_3_d90","scripted","config_5","config_5_exophormer_d539"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    f=d.get("two_batches_in_flight") or d.get("batches_in_flight") or {}
    print(t, round(d["ms_per_step"],4), round(d["value"],1), {k:round(v["ms_per_batch_step"],4) for k,v in f.items() if k in ("2","4")} or (round(f.get("ms_per_batch_step",0),4) if f else None))
PY
