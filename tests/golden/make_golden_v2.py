"""Generate tests/golden/golden_v2.npz from the REFERENCE's own code: the benched sizes.

BUILD-CONTAINER ONLY (imports /root/reference/puzzle_diff/model/*.py under the stubs of ref_import.py, exactly
like make_golden.py; golden_v1.npz stays as it is).  Stores the reference's OUTPUTS for

* ``rot900_g1``: one 900-piece puzzle on the complete graph with self loops (the headline workload):
  ``forward_with_feats`` out [900, 4], last-layer attention head / tail / checksums, per-layer activation
  statistics, and a 3-step START_X DDIM trajectory through the reference's ``p_sample``;
* ``exo900_d539_v8``: 900 pieces on a random 539-regular Exphander graph, exophormer arch with 8 virtual
  nodes (the scripted configuration, singularity/gianscarpe/train_celeba_rot.sh:4-15): forward out [900, 4];
* the reference's TorchScript ``greedy_cost_assignment`` (spatial_diffusion.py:179-216) on six position sets
  (6x6 / 12x12 / 30x30 noisy, exact grids = all-zero-distance ties, fewer pieces than cells, duplicated points);
* the reference's 3D evaluation metrics (model/utils_3d.py ``trans_metrics``, ``rot_metrics`` rmse / geodesic,
  ``calc_part_acc``) on three seeded pose sets;
* the reference's ``generate_random_regular_graph`` (dataset/puzzle_dataset.py:115-152) for four (n, d, seed):
  full edge lists of the small graphs, digests + head / tail of the 900-node ones.

Inputs and weights are regenerated from seeds by cases.py.   Run:  python tests/golden/make_golden_v2.py
(about two minutes on 8 cores).
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases as C  # noqa: E402
from ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)
sd2, _ = import_reference()
OUT = {}


def put(case, field, t):
    OUT[f"{case}/{field}"] = (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).float()


def load_weights(module, sd):
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    dead = ("linear1.", "linear2.", "visual_backbone.", "pcd_backbone.", "mean", "std")
    assert not [k for k in missing if not k.startswith(dead)]


def ref_model(spec, ratio=1, mean="START_X", noise_weight=1.0, steps=None):
    m = sd2.GNN_Diffusion(steps=steps or spec["steps"], sampling="DDIM", inference_ratio=ratio,
                          noise_weight=noise_weight, rotation=(spec["c"] == 4),
                          model_mean_type=getattr(sd2.ModelMeanType, mean), visual_pretrained=False,
                          architecture=spec["arch"], virt_nodes=spec["V"])
    m.eval()
    return m


for spec in C.FWD2D_BIG:
    case = C.build_case(spec)
    m = ref_model(spec)
    load_weights(m.model, case["sd"])
    acts = []
    hs = [m.model.mlp.register_forward_hook(lambda mod, i, o: acts.append(o))]
    for conv in m.model.gnn_backbone.module_list:
        hs.append(conv.register_forward_hook(lambda mod, i, o: acts.append(o[0] if isinstance(o, tuple) else o)))
    with torch.no_grad():
        out, att = m.forward_with_feats(case["x"], case["t"], None, case["edge_index"], case["feats"],
                                        case["batch"], return_attentions=True)
    for h in hs:
        h.remove()
    put(spec["name"], "out", out)
    ei_last, alpha_last = att[-1]
    put(spec["name"], "n_att", len(att))
    put(spec["name"], "alpha_last_stats", stats(alpha_last))
    put(spec["name"], "alpha_last_head", alpha_last[:256])
    put(spec["name"], "alpha_last_tail", alpha_last[-256:])
    put(spec["name"], "ei_last_shape", np.array(ei_last.shape))
    for i, a in enumerate(acts):
        put(spec["name"], f"act{i}_stats", stats(a))
        put(spec["name"], f"act{i}_rows", a[:: max(1, a.shape[0] // 8), :64])
    print("fwd", spec["name"], tuple(out.shape), "E'", ei_last.shape[1], flush=True)

for lp in C.LOOPS2D_BIG:
    spec = C.by_name(lp["base"])
    case = C.build_case(spec)
    m = ref_model(spec, ratio=lp["ratio"], mean=lp["mean"], noise_weight=lp["noise_weight"], steps=lp["T"])
    load_weights(m.model, case["sd"])
    torch.manual_seed(123)
    img = torch.randn(case["x"].shape) * lp["noise_weight"]
    put(lp["name"], "x_init", img)
    imgs = []
    b = img.shape[0]
    for i in list(reversed(range(0, lp["T"], lp["ratio"])))[: lp["max_iters"]]:
        img, _ = m.p_sample(img, torch.full((b,), i, dtype=torch.long), i, cond=None, edge_index=case["edge_index"],
                            patch_feats=case["feats"], batch=case["batch"])
        imgs.append(img)
    put(lp["name"], "imgs", torch.stack(imgs))
    print("loop", lp["name"], len(imgs), flush=True)

for g in C.GREEDY:
    pos1, pos2 = C.greedy_inputs(g)
    ass = sd2.greedy_cost_assignment(pos1, pos2)              # the reference's TorchScript function itself
    put(g["name"], "assignment", ass)
    print("greedy", g["name"], tuple(ass.shape), flush=True)

import importlib  # noqa: E402
ut3d = importlib.import_module("model.utils_3d")            # the reference's own metric functions
for ms in C.METRICS3D:
    pcds, pred, gt = C.metrics3d_inputs(ms)
    put(ms["name"], "rmse_t", ut3d.trans_metrics(pred[:, 4:], gt[:, 4:]))
    put(ms["name"], "rmse_r", ut3d.rot_metrics(pred[:, :4], gt[:, :4], metric="rmse"))
    put(ms["name"], "gd_r", ut3d.rot_metrics(pred[:, :4], gt[:, :4], metric="geodesic"))
    put(ms["name"], "part_acc", ut3d.calc_part_acc(pcds, pred[:, 4:], gt[:, 4:], pred[:, :4], gt[:, :4], None))
    print("metrics3d", ms["name"], [float(OUT[f"{ms['name']}/{k}"]) for k in ("rmse_t", "rmse_r", "gd_r", "part_acc")], flush=True)

from ref_import import import_reference_dataset  # noqa: E402
ds = import_reference_dataset()
for ex in C.EXPANDER:
    snd, rcv = ds.generate_random_regular_graph(ex["n"], ex["d"], np.random.default_rng(ex["seed"]))
    put(ex["name"], "checksum", C.edge_checksum(snd, rcv))
    put(ex["name"], "head", np.stack([snd[:512], rcv[:512]]))
    put(ex["name"], "tail", np.stack([snd[-512:], rcv[-512:]]))
    if ex["full"]:
        put(ex["name"], "edges", np.stack([snd, rcv]))
    print("expander", ex["name"], snd.size, flush=True)

np.savez_compressed(C.GOLDEN2_FILE, **OUT)
print("wrote", C.GOLDEN2_FILE, os.path.getsize(C.GOLDEN2_FILE), "bytes,", len(OUT), "arrays")
