#!/bin/bash
# GPU box, round 6 call 4: the refactored library (da_config, product / experiments builds): whole GPU suite on both, then the A/B protocol
# (tools/ab_config.py: one process, interleaved pairs, sign test) re-judging the step rule, the pair split and the resident kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > $O/r06_gpu_suite_product.log 2>&1; echo "suite rc=$?"; tail -5 $O/r06_gpu_suite_product.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r06_smoke.log
L=$O/r06_ab_protocol.log; : > $L
echo "== step rule: A = both off, B = default (-1 / -1)" >> $L
timeout 900 python tools/ab_config.py --a xpanel=0,tail_next=0 --b xpanel=-1,tail_next=-1 --pairs 12 >> $L 2>&1
echo "== xpanel alone: A = off, B = on (tail_next off in both)" >> $L
timeout 900 python tools/ab_config.py --a xpanel=0,tail_next=0 --b xpanel=1,tail_next=0 --pairs 12 >> $L 2>&1
echo "== tail_next alone: A = off, B = on (xpanel off in both)" >> $L
timeout 900 python tools/ab_config.py --a xpanel=0,tail_next=0 --b xpanel=0,tail_next=1 --pairs 12 >> $L 2>&1
echo "== pair split: A = one graph with two branches, B = two graphs on two streams (default)" >> $L
timeout 900 python tools/ab_config.py --a pair_split=0 --b pair_split=1 --pairs 12 >> $L 2>&1
echo "== resident hidden-layer kernel: A = ring kernel (attn_level 1), B = resident (2, default)" >> $L
timeout 900 python tools/ab_config.py --a attn_level=1 --b attn_level=2 --pairs 12 >> $L 2>&1
echo "== configuration 2 (512 x 144 pieces), step rule forced on vs default (off below 512 pieces)" >> $L
timeout 900 python tools/ab_config.py --config 2 --a xpanel=-1,tail_next=-1 --b xpanel=1,tail_next=1 --pairs 12 >> $L 2>&1
cat $L
# parity mode and the fragment encoder after the spill fixes (occupancy 1 for the fp32 C = 144 general kernel / fp32 QKV GEMM; knn64 prefetch)
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r06_bench_3p_quick.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/r06_bench_3p_quick.json') if l.startswith('{')][-1]); print('headline', d['ms_per_step'], d['value'], 'parity_mode', d.get('parity_mode'))"
timeout 600 python bench.py --mode encode --config 4 --no-cpu-baseline > $O/r06_bench_pcd_quick.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/r06_bench_pcd_quick.json') if l.startswith('{')][-1]); print('pcd encode', d['ms_per_step'], d['value'])"
