#!/bin/bash
# tools/bin/corun_probe against the in-tree library (run after `python __graft_entry__.py`)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffassemble_amd/csrc -Iinclude tools/corun_probe.hip -Ldiffassemble_amd/lib -ldiffassemble_hip \
    -Wl,-rpath,'$ORIGIN/../../diffassemble_amd/lib' -o tools/bin/corun_probe
echo built tools/bin/corun_probe
