"""Time graph_plan.build_plan (edge_index -> device plan) for the BASELINE configurations."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from diffassemble_amd.graph_plan import build_plan
from oracle import weights as W

dev = torch.device("cuda")


def timed(name, ei, batch, virt):
    ei, batch = ei.to(dev), batch.to(dev)
    for _ in range(2):
        build_plan(ei, batch, virt)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        p = build_plan(ei, batch, virt)
    torch.cuda.synchronize()
    print(f"{name:40s} E={ei.shape[1]:9d} plan {1e3 * (time.perf_counter() - t) / 5:8.2f} ms  dense={p.dense} hybrid={p.hybrid}")


for G in (1, 32):
    n = 900
    e1 = W.dense_edge_index(n, True)
    ei = torch.cat([e1 + g * n for g in range(G)], 1)
    batch = torch.arange(G).repeat_interleave(n)
    timed(f"dense 900 G={G}", ei, batch, 0)
    rng = np.random.default_rng(0)
    e1 = W.random_regular_edge_index(n, 539 if (n * 539) % 2 == 0 else 540, rng)
    ei = torch.cat([e1 + g * n for g in range(G)], 1)
    timed(f"expander d=540 exophormer V=8 G={G}", ei, batch, 8)
# Exphander graphs straight from their permutations (no edge list): graph_plan.expander_plan
from diffassemble_amd import expander  # noqa: E402
from diffassemble_amd.graph_plan import expander_plan  # noqa: E402
for G in (1, 32):
    for d in (90, 539 + 1):
        perms = expander.draw_permutations(900, G, np.random.default_rng(1)).to(dev)
        for _ in range(2):
            expander_plan(perms, d, dev, 8)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            p = expander_plan(perms, d, dev, 8)
        torch.cuda.synchronize()
        print(f"{'expander_plan d=%d V=8 G=%d' % (d, G):40s} E={p.n_edges:9d} plan {1e3 * (time.perf_counter() - t) / 5:8.2f} ms  hybrid={p.hybrid}")
n = 144
e1 = W.dense_edge_index(n, True)
ei = torch.cat([e1 + g * n for g in range(512)], 1)
timed("dense 144 G=512", ei, torch.arange(512).repeat_interleave(n), 0)
