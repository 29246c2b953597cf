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
GOLDEN3_FILE = os.path.join(os.path.dirname(__file__), "golden_v3.npz")

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
# golden_v4.npz (make_golden_v4.py): the scripted TRAINING configuration -- exophormer arch on Exphander graphs with virtual
# nodes (train_celeba_rot.sh:4-15) -- loss + every live gradient of the reference's p_losses
TRAIN2D_V4 = [
    dict(name="train_exo_expander_d6", base="exo_expander_d6", mean="START_X", seed=12),
    dict(name="train_exo144_v8_g2", base="exo144_v8_g2", mean="EPSILON", seed=13),
]
# golden_v2.npz (make_golden_v2.py): the BENCHED sizes -- 900-piece dense (headline) and the scripted
# Exphander degree d = 539 ("60 %", train_celeba_rot.sh:12) -- and the reference's own greedy_cost_assignment
FWD2D_BIG = [
    dict(name="rot900_g1",       sizes=[900], c=4, graph="dense",      arch="transformer", V=0, steps=100, seed=21, qk_gain=3.0),
    dict(name="exo900_d539_v8",  sizes=[900], c=4, graph="regular539", arch="exophormer",  V=8, steps=300, seed=22, qk_gain=3.0),
]
LOOPS2D_BIG = [
    dict(name="ddim_rot900_x0", base="rot900_g1", T=100, ratio=1, mean="START_X", noise_weight=1.0, sampling="DDIM", max_iters=3),
]
# greedy_cost_assignment (spatial_diffusion.py:179-216): pos1 = predicted positions, pos2 = grid cells
GREEDY = [
    dict(name="greedy_6x6_noisy",   rows=6,  cols=6,  noise=0.30, seed=31, kind="noisy"),
    dict(name="greedy_12x12_noisy", rows=12, cols=12, noise=0.10, seed=32, kind="noisy"),
    dict(name="greedy_30x30_noisy", rows=30, cols=30, noise=0.05, seed=33, kind="noisy"),
    dict(name="greedy_12x12_exact", rows=12, cols=12, noise=0.0,  seed=34, kind="exact"),     # all-zero-distance ties
    dict(name="greedy_rect_20v30",  rows=5,  cols=6,  noise=0.20, seed=35, kind="fewer_rows", n1=20),
    dict(name="greedy_dups",        rows=4,  cols=4,  noise=0.0,  seed=36, kind="dups"),      # ties at non-zero distances
]
# 3D evaluation metrics (utils_3d.py trans_metrics / rot_metrics / calc_part_acc): P parts of N points, prediction = ground
# truth perturbed by `noise` (the first part exactly right, the last one far off)
METRICS3D = [
    dict(name="metrics3d_p7", P=7, N=200, noise=0.05, seed=51),
    dict(name="metrics3d_p20", P=20, N=1000, noise=0.02, seed=52),
    dict(name="metrics3d_p2_far", P=2, N=64, noise=1.0, seed=53),
]
# Exphander graphs: generate_random_regular_graph(n, d, default_rng(seed)) (puzzle_dataset.py:115-152)
EXPANDER = [
    dict(name="expander_n64_d6", n=64, d=6, seed=61, full=True),
    dict(name="expander_n36_d7", n=36, d=7, seed=62, full=True),        # odd degree: + the perfect matching
    dict(name="expander_n900_d90", n=900, d=90, seed=63, full=False),
    dict(name="expander_n900_d539", n=900, d=539, seed=64, full=False),  # the scripted "60 %"
]
# fmt: on
# 3D piece encoder (vnn/vn_dgcnn.py): fragments x points, eval mode; inputs/weights from oracle/weights.py seeds
PCD_ENC = [
    dict(name="vn_p3_n256", P=3, N=256, seed=5, wseed=3, inv=False),
    dict(name="vn_p3_n256_inv", P=3, N=256, seed=6, wseed=4, inv=True),
    dict(name="vn_p2_n1000", P=2, N=1000, seed=7, wseed=3, inv=False),      # the Breaking Bad point count
    dict(name="vn_p5_n37", P=5, N=37, seed=8, wseed=9, inv=False),          # ragged: N not a multiple of anything
]
SCHEDULE_T = [50, 100, 300]
GOLDEN2_FILE = os.path.join(os.path.dirname(__file__), "golden_v2.npz")


def greedy_inputs(spec):
    """(pos1 [n, 2], pos2 [m, 2]) fp32 of a GREEDY case, regenerated from its seed.  pos2 is the grid the
    eval step builds (spatial_diffusion.py:925-930: linspace(-1, 1) meshgrid, 'xy' indexing)."""
    rng = np.random.default_rng(spec["seed"])
    y = torch.linspace(-1, 1, spec["rows"])
    x = torch.linspace(-1, 1, spec["cols"])
    grid = torch.stack(torch.meshgrid(x, y, indexing="xy"), -1).reshape(-1, 2)
    m = grid.shape[0]
    perm = torch.from_numpy(rng.permutation(m))
    noise = torch.from_numpy(rng.standard_normal((m, 2)).astype(np.float32)) * spec["noise"]
    if spec["kind"] in ("noisy", "exact"):
        pos1 = grid[perm] + noise
    elif spec["kind"] == "fewer_rows":
        pos1 = (grid[perm] + noise)[: spec["n1"]]
    elif spec["kind"] == "dups":
        pos1 = grid[perm].clone()
        pos1[: m // 2] = torch.tensor([0.1234, -0.4321])          # half of the pieces on one point
    else:
        raise ValueError(spec["kind"])
    return pos1.contiguous(), grid.contiguous()


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


def edge_checksum(s, r):
    """Order-dependent and order-independent digests of an edge list (int64 arrays / tensors)."""
    s, r = np.asarray(s, dtype=np.int64), np.asarray(r, dtype=np.int64)
    k = np.arange(1, s.size + 1, dtype=np.int64)
    return np.array([s.size, int(s.sum()), int(r.sum()), int(((s * 7 + r * 13) % 1000003).sum()),
                     int(((s * 31 + r * 17 + 5) * (k % 8191 + 1) % 1000003).sum())], dtype=np.int64)


def metrics3d_inputs(spec):
    """(pcds [P, N, 3], pred [P, 7], gt [P, 7]) fp32: poses are (unit quaternion wxyz | translation) rows."""
    rng = np.random.default_rng(spec["seed"])
    P, N = spec["P"], spec["N"]
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))  # noqa: E731
    pcds = f(P, N, 3) * 0.3
    gt = torch.cat([torch.nn.functional.normalize(f(P, 4), dim=-1), f(P, 3) * 0.5], 1)
    pred = gt + spec["noise"] * f(P, 7)
    pred[0] = gt[0]
    pred[-1, 4:] += 2.0
    pred[:, :4] = torch.nn.functional.normalize(pred[:, :4], dim=-1)
    return pcds.contiguous(), pred.contiguous(), gt.contiguous()


def by_name(name):
    for lst in (FWD2D, FWD3D, FWD2D_BIG):
        for s in lst:
            if s["name"] == name:
                return s
    raise KeyError(name)


def load_golden():
    return np.load(GOLDEN_FILE)


def load_golden2():
    return np.load(GOLDEN2_FILE)


def load_golden4():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v4.npz"))


def load_golden3():
    return np.load(GOLDEN3_FILE)


def pcd_encoder_case(spec):
    """(state dict, clouds [P, N, 3]) of a PCD_ENC entry."""
    return W.make_vn_dgcnn_state(128, spec["wseed"]), W.make_point_clouds(spec["P"], spec["N"], spec["seed"])
