#!/bin/bash
# Build an ALTERNATE copy of the library for compile-flag A/Bs: tools/build_ab.sh <tag> <file.hip> "<extra flags>" [<file2.hip> "<flags2>" ...]
# -> diffassemble_amd/lib_<tag>/libdiffassemble_hip.so (objects of the other files are reused from diffassemble_amd/lib) and
# tools/bin/attn_bench_<tag> linked against it.  Python picks it with DA_LIB_PATH=diffassemble_amd/lib_<tag>/libdiffassemble_hip.so.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
out=diffassemble_amd/lib_$tag
mkdir -p $out tools/bin
cp diffassemble_amd/lib/*.o $out/
while [ $# -gt 0 ]; do
  f=$1; fl=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $fl -c diffassemble_amd/csrc/$f -o $out/${f%.hip}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdiffassemble_hip.so $out/*.o
rm -f $out/*.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idiffassemble_amd/csrc -Iinclude tools/attn_bench.hip -L$out -ldiffassemble_hip \
    -Wl,-rpath,"\$ORIGIN/../../$out" -o tools/bin/attn_bench_$tag
echo built $out tools/bin/attn_bench_$tag
