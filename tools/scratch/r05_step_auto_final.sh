#!/bin/bash
# GPU box: final line of the round with DA_STEP_AUTO's defaults (roofline object, no CPU legs), smoke, and the hybrid-graph / sampler parity tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_config_3p_step_auto.json 2> gpurun_out/r05_bench_step_auto.err; echo "bench rc=$?"
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
(time timeout 110 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_samplers.py -x -q -k "exo900 or config3 or banded_expander_plans or samplers or ddim or ddpm or cfg") > gpurun_out/r05_step_auto_tests2.log 2>&1
tail -5 gpurun_out/r05_step_auto_tests2.log
