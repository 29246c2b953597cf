#!/bin/bash
# tools/bin/corun_probe against the EXPERIMENTS build of the library (run after `DA_EXPERIMENTS=1 python __graft_entry__.py`)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffassemble_amd/csrc -Iinclude tools/corun_probe.hip -Ldiffassemble_amd/lib_exp -ldiffassemble_hip \
    -Wl,-rpath,'$ORIGIN/../../diffassemble_amd/lib_exp' -o tools/bin/corun_probe
echo built tools/bin/corun_probe
