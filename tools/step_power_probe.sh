( python bench.py --steps 100 --warmup 10 --no-parity-mode --no-cpu-baseline --replays 3000 > /tmp/loop.log 2>&1 ) &
sleep 45
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -4 | tr '\n' ' '; echo; sleep 1; done
wait
python -c "import json; d=json.loads(open('/tmp/loop.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['replay'])"
echo idle:; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -6
