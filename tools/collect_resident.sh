#!/bin/bash
# GPU box: re-collection of the headline's profiles after the K / V-resident hidden-layer kernel became the default (the other
# configurations' kernels did not change): kernel statistics + PMC traffic of the three launch shapes, SQ counters of the attention
# kernels, the counter json, the headline bench line.  Outputs: gpurun_out/r05_*; copy to profiles/r05/.
set -u
export ROUND=r05
O=gpurun_out
bash tools/collect_profiles.sh headline --config 3p
bash tools/collect_profiles.sh headline_half --config 3p --puzzles 32
DA_TWO_BRANCH=1 bash tools/collect_profiles.sh headline_two_branch --config 3p
bash tools/collect_attn_pmc.sh > /dev/null 2>&1
cp $O/${ROUND}_pmc_*.txt $O/${ROUND}_rocprof_*.txt profiles/$ROUND/ 2>/dev/null
python tools/make_pmc_json.py $ROUND > $O/${ROUND}_make_pmc_json.log 2>&1
cp profiles/$ROUND/pmc_traffic.json profiles/$ROUND/pmc_attention_sq.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 100 --warmup 10 > $O/${ROUND}_bench_config_3p.json 2> $O/${ROUND}_bench_config_3p.err
tail -c 600 $O/${ROUND}_bench_config_3p.json
