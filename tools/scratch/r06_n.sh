#!/bin/bash
# GPU box: k_attn_csr with 12-byte loads for the 144-wide bf16 rows: parity + the csr line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py --config csr > $O/r06_bench_csr.json 2> $O/r06_bench_csr.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_csr.json") if l.startswith("{")][-1])
r=d["roofline"]; print("csr", round(d["ms_per_step"],4), d["value"], r["kernel"], round(r["avg_launch_us"],1), "us", round(r["achieved"],1), "GB/s frac", round(r["frac"],4), {k:(round(v["ms_per_batch_step"],4), round(v["vs_one_batch_in_flight"],2)) for k,v in (d.get("batches_in_flight") or {}).items() if k in ("2","4")})
for k,v in r["classes"].items():
    if "attn" in k: print("  ", k, round(v["avg_launch_us"],1), round(v["compulsory_GBps"],1))
PY
