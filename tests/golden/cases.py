"""Case specifications shared by make_golden.py (build container, reference imported
under stubs) and the tests (CPU oracle / GPU HIP path).  A case is a plain dict; inputs
and weights are regenerated from seeds by ``build_case`` (oracle/weights.py), so the
fixture file only stores the reference's OUTPUTS."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import weights as W  # noqa: E402

GOLDEN_FILE = os.path.join(os.path.dirname(__file__), "golden_v1.npz")

# fmt: off
FWD2D = [
    # name,            grid, G, c, loops, arch,          V, steps, seed, qk_gain, graph
    dict(name="k36_noloop_eps", sizes=[36],       c=2, graph="dense_noloop", arch="transformer", V=0, steps=50,  seed=0, qk_gain=1.0),
    dict(name="k36_loop_sharp", sizes=[36],       c=2, graph="dense",        arch="transformer", V=0, steps=50,  seed=1, qk_gain=4.0),
    dict(name="rot144_g1",      sizes=[144],      c=4, graph="dense",        arch="transformer", V=0, steps=100, seed=0, qk_gain=1.0),
    dict(name="rot144_g2_sharp", sizes=[144, 144], c=4, graph="dense",       arch="transformer", V=0, steps=100, seed=1, qk_gain=4.0),
    dict(name="ragged_dense",   sizes=[36, 64, 100], c=4, graph="dense",     arch="transformer", V=0, steps=100, seed=2, qk_gain=3.0),
    dict(name="exo144_v4_g1",   sizes=[144],      c=4, graph="dense",        arch="exophormer",  V=4, steps=300, seed=3, qk_gain=2.0),
    dict(name="exo144_v8_g2",   sizes=[144, 144], c=4, graph="dense",        arch="exophormer",  V=8, steps=300, seed=4, qk_gain=2.0),
    dict(name="exo_expander_d6", sizes=[64, 36],  c=4, graph="regular6",     arch="exophormer",  V=4, steps=300, seed=5, qk_gain=3.0),
    dict(name="tr_expander_d7", sizes=[64, 100],  c=4, graph="regular7",     arch="transformer", V=0, steps=100, seed=6, qk_gain=3.0),
]
LOOPS2D = [
    dict(name="ddim_t50_eps",   base="k36_noloop_eps", T=50,  ratio=1,  mean="EPSILON", noise_weight=1.0, sampling="DDIM"),
    dict(name="ddim_t300_x0",   base="exo144_v4_g1",   T=300, ratio=10, mean="START_X", noise_weight=1.0, sampling="DDIM"),
    dict(name="ddim_t100_x0_nw0", base="rot144_g1",    T=100, ratio=1,  mean="START_X", noise_weight=0.0, sampling="DDIM", max_iters=12),
]
FWD3D = [
    dict(name="bb_p2",  sizes=[2],        arch="transformer", V=0, steps=300, seed=7, qk_gain=2.0),
    dict(name="bb_p20", sizes=[20, 7, 13], arch="transformer", V=0, steps=300, seed=8, qk_gain=2.0),
]
LOOPS3D = [
    dict(name="ddim3d_t300", base="bb_p20", T=300, ratio=10, mean="START_X", noise_weight=1.0, max_iters=10),
]
TRAIN2D = [
    dict(name="train_rot144_g2", base="rot144_g2_sharp", mean="EPSILON", seed=11),
]
# fmt: on
SCHEDULE_T = [50, 100, 300]


def _graph(kind, n, rng):
    if kind == "dense":
        return W.dense_edge_index(n, True)
    if kind == "dense_noloop":
        return W.dense_edge_index(n, False)
    if kind.startswith("regular"):
        return W.random_regular_edge_index(n, int(kind[len("regular"):]), rng)
    raise ValueError(kind)


def build_case(spec, variant="2d"):
    """-> dict(sd, x, t, feats, edge_index, batch) regenerated from the spec's seeds."""
    sizes = spec["sizes"]
    N = sum(sizes)
    rng = np.random.default_rng(spec["seed"] + 77)
    if variant == "2d":
        sd = W.make_denoiser_state(spec["steps"], spec["c"], spec["c"], D=1152, hidden=128,
                                   variant="2d", arch=spec["arch"], virt_nodes=spec["V"],
                                   seed=spec["seed"], qk_gain=spec["qk_gain"])
        x, feats = W.make_inputs(N, spec["c"], 1088, spec["seed"])
        eis = [_graph(spec["graph"], n, rng) for n in sizes]
    else:
        sd = W.make_denoiser_state(spec["steps"], 7, None, D=832, hidden=256, variant="3d",
                                   arch=spec["arch"], virt_nodes=spec["V"], seed=spec["seed"],
                                   qk_gain=spec["qk_gain"])
        x, feats = W.make_inputs(N, 7, 768, spec["seed"])
        x[:, :4] = torch.nn.functional.normalize(x[:, :4], dim=-1)      # unit quaternions
        eis = [W.dense_edge_index(n, True) for n in sizes]
    edge_index, batch = W.collate(eis, sizes)
    # per-graph timestep broadcast to nodes, as training_step does (spatial_diffusion.py:710-712)
    tg = torch.from_numpy(rng.integers(0, spec["steps"], size=len(sizes)))
    t = tg[batch]
    return dict(sd=sd, x=x, t=t, feats=feats, edge_index=edge_index, batch=batch)


def by_name(name):
    for lst in (FWD2D, FWD3D):
        for s in lst:
            if s["name"] == name:
                return s
    raise KeyError(name)


def load_golden():
    return np.load(GOLDEN_FILE)
