#!/bin/bash
# The standalone harnesses of tools/ (attn_bench, corun_probe) against the EXPERIMENTS build of the library: they call internal launchers and
# select kernel variants through experiment switches.  Run after `DA_EXPERIMENTS=1 python __graft_entry__.py`.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
for t in attn_bench corun_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffassemble_amd/csrc -Iinclude tools/$t.hip -Ldiffassemble_amd/lib_exp -ldiffassemble_hip \
      -Wl,-rpath,'$ORIGIN/../../diffassemble_amd/lib_exp' -o tools/bin/$t
  echo built tools/bin/$t
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_probe.hip -o tools/bin/gather_probe && echo built tools/bin/gather_probe
