#!/usr/bin/env python3
"""Concurrent timeline of the two-stream pair loop from a rocprofv3 --kernel-trace database (rocpd `*_results.db`).

    python tools/pair_timeline.py <results.db> [--steps N] [--dump K]

Takes the LAST graph replay in the trace (the dispatches of the two loop graphs sit on two queues), and reports
  * per kernel class: launches, mean duration in the concurrent loop, mean number of OTHER kernels active during it,
  * the share of wall time with 0 / 1 / 2+ kernels in flight, wall time per step, kernel-time sum per step, overlap gain,
  * --dump K: the first K dispatches of a steady-state step with queue, start, end (the artefact VERDICT r05 item 1 asks for)."""
import argparse, collections, re, sqlite3

CLASSES = [("attn_hidden", r"k_attn_res|k_attn_optt<32"), ("attn_last", r"k_attn_optt<144"), ("proj", r"k_gemm_wreg|k_gemm_xpanel|k_gemm_thin|k_gemm_mfma"),
           ("mlp", r"k_gemm_astat"), ("tail", r"k_tail_fused|k_head"), ("embed", r"k_embed_pos_time"), ("ddim", r"k_ddim|k_ddpm")]


def cls(name):
    for c, pat in CLASSES:
        if re.search(pat, name):
            return c
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=20, help="denoising steps per replay (bench.py --steps)")
    ap.add_argument("--dump", type=int, default=0)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, queue_id, start, end, grid_x from kernels order by start").fetchall()
    # the loop's dispatches: the last contiguous run of library kernels with gaps < 200 us
    lib = [r for r in rows if re.search(r"\bk_[a-z]", r[0]) and "da::" in r[0]]
    end = len(lib)
    i = end - 1
    while i > 0 and lib[i][2] - lib[i - 1][3] < 200_000:
        i -= 1
    run = lib[i:end]
    t0, t1 = run[0][2], max(r[3] for r in run)
    wall = (t1 - t0) / 1e3
    ksum = sum(r[3] - r[2] for r in run) / 1e3
    print(f"last replay: {len(run)} dispatches on queues {sorted(set(r[1] for r in run))}, wall {wall:.1f} us, kernel-time sum {ksum:.1f} us, "
          f"overlap gain {ksum / wall:.3f}, per step ({a.steps} steps): wall {wall / a.steps:.1f} us, kernel sum {ksum / a.steps:.1f} us")
    # concurrency histogram
    ev = sorted([(r[2], 1) for r in run] + [(r[3], -1) for r in run])
    hist, act, prev = collections.Counter(), 0, t0
    for t, d in ev:
        hist[min(act, 2)] += t - prev
        prev, act = t, act + d
    tot = sum(hist.values())
    print("time with 0 / 1 / 2+ kernels in flight: " + " / ".join(f"{100 * hist[k] / tot:.1f} %" for k in (0, 1, 2)))
    # per class: duration and what runs beside it
    agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for r in run:
        c = cls(r[0])
        agg[c][0] += 1
        agg[c][1] += (r[3] - r[2]) / 1e3
        for o in run:
            if o is not r and o[2] < r[3] and o[3] > r[2]:
                agg[c][2][cls(o[0])] += (min(o[3], r[3]) - max(o[2], r[2])) / 1e3
    print(f"{'class':12s} {'n':>5s} {'mean_us':>8s} {'us/step':>8s}   time shared with (us per launch)")
    for c, (n, t, beside) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:12s} {n:5d} {t / n:8.1f} {t / a.steps:8.1f}   " + ", ".join(f"{k} {v / n:.1f}" for k, v in beside.most_common(5)))
    if a.dump:
        mid = run[len(run) // 2][2]
        k0 = next(i for i, r in enumerate(run) if r[2] >= mid and cls(r[0]) == "embed")
        print("steady-state window (us from its first dispatch):")
        b = run[k0][2]
        for r in run[k0:k0 + a.dump]:
            nm = r[0].replace("void ", "").replace("da::", "")[:44]
            print(f"  q{r[1]:<3d} {(r[2] - b) / 1e3:8.1f} -> {(r[3] - b) / 1e3:8.1f}  ({(r[3] - r[2]) / 1e3:6.1f})  {cls(r[0]):11s} {nm}")


if __name__ == "__main__":
    main()
