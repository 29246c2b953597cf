# timing experiment: what the adjacency masking itself costs in k_attn_optt<MASKED> (rocprof kernel averages, config 3, d = 539)
export TMPDIR=/tmp
for dbg in none full partial; do
W=/tmp/prof_mp_$dbg; rm -rf $W; mkdir -p $W
( cd /tmp && DA_EXPANDER_CLS_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats -d $W -o s -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --replays 0 > $W/log 2>&1 )
echo "== classes: $dbg"; python profiles/rocpd_stats.py $(find $W -name "*results.db" | head -1) 2>&1 | grep -E "k_attn_optt|k_attn_csr_cont_heavy|total kernel" | cut -c1-150
done
