#!/usr/bin/env python3
"""Resource table of every kernel in the built library: VGPRs, AGPRs, SGPRs, spills, LDS, scratch.

Reads the AMDGPU metadata notes of the gfx950 code objects embedded in each `diffassemble_amd/lib/*.o`
(`llvm-objcopy --dump-section .hip_fatbin`, `clang-offload-bundler --unbundle`, then `llvm-readelf --notes`).  Used for two things:
  * the spill audit of the default build (0 kernels with vgpr_spill_count > 0),
  * sizing co-resident kernels (what a CU has left beside one workgroup of kernel X).

    python tools/kernel_resources.py [--spills | --vgpr-spills] [--match REGEX] [--json OUT] [--libdir DIR]
"""
import argparse, glob, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def kernels_of(obj, tmp):
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
    if r.returncode or not os.path.exists(fat):
        return []
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
    if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.split("\n"):
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr": int(m.group(1))}
            out.append(cur)
            continue
        if cur is None:
            continue
        m = re.match(r"\s+\.(\w+):\s+(.*)$", line)
        if m and m.group(1) in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count",
                                "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size", "name"):
            v = m.group(2).strip()
            cur[m.group(1)] = int(v) if v.isdigit() else v.strip("'")
    names = demangle([k.get("name", "?") for k in out])
    for k, n in zip(out, names):
        k["demangled"] = n
        k["file"] = os.path.basename(obj)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spills", action="store_true", help="only kernels that spill; exit 1 if any")
    ap.add_argument("--vgpr-spills", action="store_true", help="only kernels that spill VECTOR registers (to scratch memory); exit 1 if any")
    ap.add_argument("--match", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--libdir", default=os.path.join(ROOT, "diffassemble_amd", "lib"))
    a = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(glob.glob(os.path.join(a.libdir, "*.o"))):
            rows += kernels_of(o, tmp)
    if a.match:
        rows = [r for r in rows if re.search(a.match, r["demangled"])]
    if a.spills:
        rows = [r for r in rows if r.get("vgpr_spill_count", 0) or r.get("sgpr_spill_count", 0)]
    if a.vgpr_spills:
        rows = [r for r in rows if r.get("vgpr_spill_count", 0)]
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'lds':>7} {'scratch':>7} {'wg':>5}  kernel")
    for r in rows:
        print(f"{r.get('vgpr_count',0):5d} {r['agpr']:5d} {r.get('sgpr_count',0):5d} {r.get('vgpr_spill_count',0):6d} "
              f"{r.get('sgpr_spill_count',0):6d} {r.get('group_segment_fixed_size',0):7d} {r.get('private_segment_fixed_size',0):7d} "
              f"{r.get('max_flat_workgroup_size',0):5d}  {r['file']}: {r['demangled'][:150]}")
    print(f"{len(rows)} kernels")
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
    if (a.spills or a.vgpr_spills) and rows:
        sys.exit(1)


if __name__ == "__main__":
    main()
