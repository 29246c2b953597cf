#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/small_batch_graph_probe.py 8 2>&1 | grep -v Warning | tee gpurun_out/r06_small_batch_graph_vs_stream.log
timeout 600 python tools/small_batch_graph_probe.py 2 2>&1 | grep -v Warning | tee -a gpurun_out/r06_small_batch_graph_vs_stream.log
