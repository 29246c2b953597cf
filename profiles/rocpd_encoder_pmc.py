#!/usr/bin/env python
"""Per-layer HBM traffic of the piece encoder from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; rocpd
databases).  Values are KB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64, MI355X_MICROARCH.md).
usage: python profiles/rocpd_encoder_pmc.py <fetch.db> <write.db> <chunk>"""
import sqlite3
import sys

B = int(sys.argv[3])


def per_layer(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, value, dispatch_id from counters_collection where counter_name = ? "
                       "order by dispatch_id", (counter,)).fetchall()
    rows = [r for r in rows if "k_conv" in r[0] or "k_enc_stem" in r[0]]
    start = next(i for i, r in enumerate(rows) if "k_enc_stem" in r[0])
    return [r[1] for r in rows[start:start + 20]]


NAMES = ["stem", "l1.0.c1", "l1.0.c2", "l1.1.c1", "l1.1.c2", "l2.0.c1", "l2.0.sc", "l2.0.c2", "l2.1.c1", "l2.1.c2",
         "l3.0.c1", "l3.0.sc", "l3.0.c2", "l3.1.c1", "l3.1.c2", "l4.0.c1", "l4.0.sc", "l4.0.c2", "l4.1.c1", "l4.1.c2"]
# algorithmic bytes per piece (bf16 maps, interior only): input map read once (+ residual) + weights, output written once
IN = [3 * 32 * 32 * 4] + [32 * 32 * 256] * 6 + [16 * 16 * 512] * 5 + [8 * 8 * 512] * 5 + [4 * 4 * 1024] * 3
RES = [0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1]
OUT = [32 * 32 * 256] * 5 + [16 * 16 * 512] * 5 + [8 * 8 * 512] * 5 + [4 * 4 * 1024] * 5
f, w = per_layer(sys.argv[1], "FETCH_SIZE"), per_layer(sys.argv[2], "WRITE_SIZE")
print(f"{'layer':10s} {'fetch MB':>9s} {'alg. MB':>8s} {'write MB':>9s} {'alg. MB':>8s}   (chunk of {B} pieces)")
for i, n in enumerate(NAMES):
    rd = B * (IN[i] + RES[i] * OUT[i]) / 1e6
    print(f"{n:10s} {2 * f[i] * 1024 / 1e6:9.1f} {rd:8.1f} {w[i] * 1024 / 1e6:9.1f} {B * OUT[i] / 1e6:8.1f}")
