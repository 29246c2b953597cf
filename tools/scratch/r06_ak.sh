#!/bin/bash
# GPU box: refresh, on HEAD, the artefacts of the configurations whose kernels changed late in the round (exophormer inference, edge-list kernel,
# training glue): rocprof + PMC of configuration 3 (both degrees) and csr, their bench lines, the training lines and their kernel statistics
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export ROUND=r06
O=gpurun_out
bash tools/collect_profiles.sh config3_d539 --config 3
bash tools/collect_profiles.sh config3_d90 --config 3 --degree 90
bash tools/collect_profiles.sh csr --config csr
mkdir -p profiles/$ROUND
cp $O/${ROUND}_pmc_*.txt $O/${ROUND}_rocprof_*.txt profiles/$ROUND/ 2>/dev/null
python tools/make_pmc_json.py $ROUND > $O/${ROUND}_make_pmc_json.log 2>&1
cp profiles/$ROUND/pmc_traffic.json profiles/$ROUND/pmc_attention_sq.json $O/ 2>/dev/null
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/${ROUND}_bench_$tag.json 2> $O/${ROUND}_bench_$tag.err; echo "bench $tag rc=$?"; }
b config_1 --config 1
b config_3 --config 3
b config_3_d90 --config 3 --degree 90
b csr --config csr
b scripted --config scripted
b config_5 --config 5
b config_5_fp32 --config 5 --precision fp32
b config_5_exophormer_d539 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16
b config_5_exophormer_d539_fp32 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision fp32
( cd /tmp && rm -rf /tmp/prof_t5 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $O/${ROUND}_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r06_bench_*.json")):
    try: d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    fl=d.get("two_batches_in_flight") or d.get("batches_in_flight") or {}
    print(f"{os.path.basename(f):48s} {d.get('ms_per_step',0):9.4f} {d.get('value',0):12.1f} frac={(d.get('roofline') or {}).get('frac')}", {k:round(v['ms_per_batch_step'],4) for k,v in fl.items() if k in ('2','4')} if 'batches_in_flight' in d else (round(fl.get('ms_per_batch_step',0),4) if fl else ''))
PY
