#!/usr/bin/env python
"""The "Configuration lines of the round" table of DESIGN.md out of a round's bench lines.
usage: python tools/config_table.py profiles/r05 [r05]"""
import json
import os
import sys

d = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
rows = [("1", "config_1"), ("2", "config_2"), ("3", "config_3"), ("3 `--degree 90`", "config_3_d90"), ("3p (headline)", "config_3p"),
        ("4", "config_4"), ("scripted", "scripted"), ("csr", "csr"), ("5 (default: bf16 operands + bf16 projection buffers)", "config_5"),
        ("5 `--precision fp32` (the reference's arithmetic)", "config_5_fp32"),
        ("5 `--arch exophormer --train-side 30 --degree 539 --train-puzzles 16`", "config_5_exophormer_d539"),
        ("… `--precision fp32`", "config_5_exophormer_d539_fp32"), ("5 `--pixels`", "config_5_pixels"),
        ("5 `--pixels --precision fp32`", "config_5_pixels_fp32"), ("`--mode e2e`", "e2e"), ("`--mode encode`", "encode"),
        ("`--mode encode --config 4`", "pcd_encode")]
print("| `--config` | workload | value | ms per step | `roofline` | CPU oracle, same unit |")
print("|---|---|---|---|---|---|")
for label, tag in rows:
    p = os.path.join(d, f"{rnd}_bench_{tag}.json")
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print(f"| {label} | (no line: {e.__class__.__name__}) | | | | |")
        continue
    r, c = j.get("roofline") or {}, j.get("cpu_baseline") or {}
    cpu = f"{c['value']:.3g} ({c.get('cores')})" if c.get("value") else "—"
    print(f"| {label} | {j['config']['workload'][:60]} | {j['value']:,.0f} {j['unit']} | {j['ms_per_step']:.3f} | "
          f"{r.get('bound', '—')} {r.get('frac', 0):.3f} | {cpu} |")
