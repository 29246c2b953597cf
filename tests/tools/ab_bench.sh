# usage: bash tests/tools/ab_bench.sh ENVVAR  -- headline bench at G = 32 / 8 / 1 with ENVVAR=1 (feature off) and =0 (on)
for g in 32 8 1; do for off in 1 0; do
  env $1=$off python bench.py --puzzles $g --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('G', d['config']['puzzles_per_gpu'], '$1=$off', round(d['value']), round(d['ms_per_step'],4))"
done; done
