#!/usr/bin/env python
"""Per-layer timing of the piece encoder from a rocprofv3 rocpd database: the stem + 19 convolution
launches repeat per chunk.  usage: python profiles/rocpd_encoder_sequence.py <results.db> <chunk>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
B = int(sys.argv[2])
rows = con.execute("select name, start, end, grid_x from kernels order by start").fetchall()
rows = [r for r in rows if "k_conv" in r[0] or "k_enc_stem" in r[0]]
per = 20
LAYERS = [("stem 3->128 @32", 32 * 32 * 128 * 27 * 2)]
cin, h = 128, 32
for li, (cout, stride) in enumerate(((128, 1), (256, 2), (256, 2), (512, 2)), start=1):
    ho = h // stride
    LAYERS.append((f"l{li}.0.conv1 {cin}->{cout} @{ho}", ho * ho * cout * cin * 18))
    if stride != 1:
        LAYERS.append((f"l{li}.0.shortcut {cin}->{cout} @{ho}", ho * ho * cout * cin * 2))
    LAYERS.append((f"l{li}.0.conv2 {cout}->{cout} @{ho}", ho * ho * cout * cout * 18))
    LAYERS.append((f"l{li}.1.conv1 {cout}->{cout} @{ho}", ho * ho * cout * cout * 18))
    LAYERS.append((f"l{li}.1.conv2 {cout}->{cout} @{ho}", ho * ho * cout * cout * 18))
    cin, h = cout, ho
# full chunks only: a chunk starts at a stem launch with grid = B * 256 threads
starts = [i for i, r in enumerate(rows) if "k_enc_stem" in r[0] and r[3] == B * 256]
starts = [i for i in starts if i + per <= len(rows)][-10:]
print(f"{'#':>3s} {'layer':34s} {'avg_us':>8s} {'TFLOP/s':>8s}")
tot = 0.0
for k in range(per):
    avg = sum(rows[i + k][2] - rows[i + k][1] for i in starts) / len(starts) / 1e3
    tot += avg
    print(f"{k:3d} {LAYERS[k][0]:34s} {avg:8.1f} {B * LAYERS[k][1] / avg / 1e6:8.0f}")
print(f"sum per chunk of {B}: {tot:.1f} us = {B * sum(f for _, f in LAYERS) / tot / 1e6:.0f} TFLOP/s")
