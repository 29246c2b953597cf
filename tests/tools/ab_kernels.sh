# usage: bash tests/tools/ab_kernels.sh ENVVAR -- per-class kernel time (us per launch, HIP events) with ENVVAR=1 / 0 at G = 32
for off in 1 0 1 0; do
  env $1=$off python bench.py --puzzles 32 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "
import json; d=json.load(open('/tmp/o.json')); k=d['kernels']
print('$1=$off', round(d['ms_per_step'],4), {n: (round(v['ms_per_launch']*1e3,1) if isinstance(v, dict) and 'ms_per_launch' in v else v) for n, v in k.items()})"
done
