#!/usr/bin/env python3
"""The A/B protocol for every default: ONE process on ONE box, the two settings of the library's run-time
switches (`da_config`, include/diffassemble_hip.h) captured as two loop graphs and replayed INTERLEAVED, A B B A A B ..., at the
driver's loop length (--steps 20) and at 100; reports per-setting medians, the median of the paired differences and a two-sided
sign test.  A default changes only on p < 0.05 at the driver's command line.

    python tools/ab_config.py --a xpanel=0,tail_next=0 --b xpanel=-1,tail_next=-1 [--config 3p] [--puzzles 64] [--pairs 12] [--steps 20 100]

Fields a denoiser fixes at creation (disable_folds, disable_mfma) need two engines: --recreate builds one per setting.
(Synthetic weights / inputs exactly as bench.py builds them; the oracle is not involved.)"""
import argparse
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diffassemble_amd import _lib  # noqa: E402


def parse(s):
    return {k: int(v) for k, v in (kv.split("=") for kv in s.split(",") if kv)}


def sign_test_p(diffs):
    """two-sided binomial sign test on the non-zero paired differences"""
    pos, neg = sum(d > 0 for d in diffs), sum(d < 0 for d in diffs)
    n, k = pos + neg, min(pos, neg)
    if n == 0:
        return 1.0
    return min(1.0, 2 * sum(math.comb(n, i) for i in range(k + 1)) / 2 ** n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", required=True)
    ap.add_argument("--b", required=True)
    ap.add_argument("--config", default="3p")
    ap.add_argument("--puzzles", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=12)
    ap.add_argument("--steps", type=int, nargs="+", default=[20, 100])
    ap.add_argument("--degree", type=int, default=539)
    ap.add_argument("--recreate", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = bench.CONFIGS[a.config]
    G, n = a.puzzles or cfg["G"], cfg["n"]
    A, B = parse(a.a), parse(a.b)
    base = _lib.config()

    def apply(fields):
        _lib.set_config(**{k: getattr(base, k) for k, _ in _lib.DaConfig._fields_ if k != "struct_bytes"})
        _lib.set_config(**fields)

    engines = {}

    def engine_for(tag, fields):
        key = tag if a.recreate else "shared"
        if key not in engines:
            apply(fields if a.recreate else {})
            model = bench.build_module(cfg, dev, cfg["prec"])
            engines[key] = (model, model.model.engine(dev))
        return engines[key]

    gen = torch.Generator(device=dev).manual_seed(1234)
    N = G * n
    threed = cfg["variant"] == "3d"
    feats = torch.randn((N, 768 if threed else 1088), generator=gen, device=dev)
    c = 7 if threed else (4 if cfg["rotation"] else 2)
    x_T = torch.randn((N, c), generator=gen, device=dev)
    mt = _lib.MEAN_START_X if cfg["mean"] == "START_X" else _lib.MEAN_EPSILON
    plans = {}

    def plan_for(eng):
        if id(eng) not in plans:
            if cfg["graph"] == "regular":
                perms = bench.expander_perms(cfg, G, 3).to(dev)
                plans[id(eng)] = eng.plan_expander(perms, a.degree)
            else:
                ei, batch = bench.dense_batch(G, n, dev, loops=cfg["graph"] == "dense")
                plans[id(eng)] = eng.plan(ei, batch)
            eng.set_features(plans[id(eng)], feats)
        return plans[id(eng)]

    def loop(tag, fields, k):
        model, eng = engine_for(tag, fields)
        apply(fields)
        return eng.sample_loop(plan_for(eng), model._schedule(), x_T, feats, ratio=cfg["ratio"], mean_type=mt, max_iters=k,
                               keep_trajectory=False, use_graph=True, restage=False)

    def timed(tag, fields, k, reps):
        loop(tag, fields, k)                                   # (the graph of this setting: recorded on first use, replayed afterwards)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            loop(tag, fields, k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * k) * 1e3

    print(f"config {a.config}, {G} puzzles of {n} pieces, A = {A}, B = {B}, {a.pairs} interleaved pairs per loop length", flush=True)
    for k in a.steps:
        reps = max(1, 100 // k)
        for tag, f in (("A", A), ("B", B)):                    # capture + warm both
            timed(tag, f, k, 2)
        ta, tb = [], []
        for i in range(a.pairs):
            order = (("A", A), ("B", B)) if i % 2 == 0 else (("B", B), ("A", A))
            for tag, f in order:
                (ta if tag == "A" else tb).append(timed(tag, f, k, reps))
        d = [y - x for x, y in zip(ta, tb)]
        print(f"steps {k:4d}: A median {statistics.median(ta):.4f} ms/step  B median {statistics.median(tb):.4f}  "
              f"median(B - A) {statistics.median(d) * 1e3:+.1f} us ({100 * statistics.median(d) / statistics.median(ta):+.2f} %)  "
              f"B faster in {sum(x < 0 for x in d)} of {len(d)} pairs, sign test p = {sign_test_p(d):.4f}", flush=True)
    apply({})


if __name__ == "__main__":
    main()
