#!/usr/bin/env python
"""profiles/<round>/<round>_pmc_traffic_<tag>.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, summarised per kernel by
profiles/rocpd_pmc.py) -> profiles/<round>/pmc_traffic.json  (round = argv[1], default r03), the per-kernel-CLASS bytes-per-launch table bench.py attaches to
its roofline entries.  Values stay in the counters' KiB; bench.py applies the gfx950 correction (FETCH_SIZE x 2,
MI355X_MICROARCH.md) when it converts."""
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
CLASS = [  # (regex on the kernel name, class)
    (r"k_attn_res<", "attn_hidden"), (r"k_attn_optt<32", "attn_hidden"), (r"k_attn_optt<144", "attn_last"), (r"k_attn_opt\(", "attn_hidden"),
    (r"k_attn_dual<144", "attn_last"), (r"k_attn_dual<32", "attn_hidden"),
    (r"k_attn_dense<unsigned short, 32,", "attn_hidden"), (r"k_attn_dense<unsigned short, 144,", "attn_last"),
    (r"k_attn_csr<unsigned short, 4>", "attn_hidden"), (r"k_attn_csr<unsigned short, 18>", "attn_last"),
    (r"k_attn_csr_cont", "attn_hidden"),
    (r"k_gemm_astat_rs<unsigned short, true", "linear_qkvs"), (r"k_gemm_wreg<256, true", "linear_qkvs"),
    (r"k_gemm_wreg2<", "linear_qkvs"),
    (r"k_gemm_wreg<256, false", "linear_qkvs"), (r"k_embed_pos_time", "embed"), (r"k_head_fold", "head"), (r"k_tail_fused", "head"),
]
# tag -> (bench config key, puzzles per GPU, launches of the class per denoising step)
RUNS = {"headline": ("3p", 64, None), "headline_half": ("3p", 32, None), "config3_d539": ("3_d539", 32, False), "config3_d90": ("3_d90", 32, False),
        "config3_d539_csr_only": ("3_d539_csr", 32, True), "config3_d90_csr_only": ("3_d90_csr", 32, True),
        "scripted": ("scripted", 8, None), "csr": ("csr", 64, None)}


def parse(path, want_csr):
    """want_csr: keep ONLY (True) / drop (False) the pure edge-list kernels k_attn_csr<T, EPL> -- in a hybrid run they
    belong to bench.py's `sparse_path` comparison leg, not to the timed loop."""
    per = {}
    for line in open(path):
        m = re.match(r"(.{56}) (FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
        if not m:
            continue
        name, ctr, n, avg = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
        is_csr = bool(re.search(r"k_attn_csr<unsigned short, \d+>", name))
        if want_csr is not None and is_csr != want_csr and ("k_attn" in name):
            continue
        for rx, cls in CLASS:
            if re.search(rx, name):
                d = per.setdefault(cls, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                d[ctr][0] += avg * n
                d[ctr][1] += n
                break
    return {c: {"fetch_kib": v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1), "write_kib": v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1),
                "launches": v["FETCH_SIZE"][1]} for c, v in per.items()}


out = {"_comment": "KiB per launch, averaged over the launches of a kernel class in `python bench.py --steps 20 --warmup 2` "
                   "(tools/collect_profiles.sh); FETCH_SIZE is NOT yet doubled here"}
for tag, (key, G, want_csr) in RUNS.items():
    p = os.path.join(ROOT, "profiles", ROUND, f"{ROUND}_pmc_traffic_{tag}.txt")
    if os.path.exists(p):
        out.setdefault(key, {}).setdefault("bf16", {})[str(G)] = parse(p, want_csr)
json.dump(out, open(os.path.join(ROOT, "profiles", ROUND, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])

# ---- SQ counters of the attention kernels (tools/collect_attn_pmc.sh) -> pmc_attention_sq.json:
# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration of the SAME counter pass x 2.4 GHz nominal clock)
sq_path = os.path.join(ROOT, "profiles", ROUND, f"{ROUND}_pmc_attention_sq.txt")
if os.path.exists(sq_path):
    sq, hdr = {}, None
    for line in open(sq_path):
        m = re.match(r"== G=(\d+) C=(\d+) fold=(\d+) kernel=(\d+)", line)
        if m:
            hdr = m.groups()
            continue
        m = re.match(r"(.{56}) (SQ_\w+)\s+(\d+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+([\d.]+)", line)
        if m and hdr and hdr[3] == "2":                      # production dispatch only
            G, Cw = hdr[0], hdr[1]
            cls = "attn_hidden" if Cw == "32" else "attn_last"
            d = sq.setdefault("bf16", {}).setdefault(G, {}).setdefault(cls, {"kernel": m.group(1).strip(), "counters": {}})
            d["counters"][m.group(2)] = {"per_launch": float(m.group(4)), "avg_us": float(m.group(5))}
    for G, per in sq.get("bf16", {}).items():
        for cls, d in per.items():
            c = d["counters"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                b = c["SQ_VALU_MFMA_BUSY_CYCLES"]
                d["mfma_busy"] = b["per_launch"] / (1024 * b["avg_us"] * 1e-6 * 2.4e9)
                d["source"] = (f"profiles/{ROUND}/{ROUND}_pmc_attention_sq.txt: SQ_VALU_MFMA_BUSY_CYCLES {b['per_launch']:.4g} per launch / "
                               f"(1024 SIMDs x {b['avg_us']:.1f} us x 2.4 GHz), tools/bin/attn_bench at {G} puzzles per launch")
            if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
                d["valu_per_mfma"] = c["SQ_INSTS_VALU"]["per_launch"] / c["SQ_INSTS_MFMA"]["per_launch"]
    json.dump(sq, open(os.path.join(ROOT, "profiles", ROUND, "pmc_attention_sq.json"), "w"), indent=1)
    print(json.dumps({G: {k: {"mfma_busy": v.get("mfma_busy"), "valu_per_mfma": v.get("valu_per_mfma")} for k, v in per.items()} for G, per in sq.get("bf16", {}).items()}, indent=1))

