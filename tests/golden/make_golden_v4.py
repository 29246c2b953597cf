"""Generate tests/golden/golden_v4.npz from the REFERENCE's own code: training-step fixtures of the SCRIPTED training
configuration (singularity/gianscarpe/train_celeba_rot.sh:4-15: ``--architecture exophormer --virt_nodes`` on Exphander
graphs), i.e. the loss of the reference's ``GNN_Diffusion.p_losses`` (spatial_diffusion.py:432-483) and the gradient of
every live parameter after ``loss.backward()``, for the exophormer cases of cases.TRAIN2D_V4 -- virtual-node quirk edges,
duplicated and cross-graph pairs included (exophormer_gnn.py:183-200).

BUILD-CONTAINER ONLY (imports /root/reference/puzzle_diff/model/*.py under the stubs of ref_import.py, like
make_golden.py).  Inputs and weights are regenerated from seeds by cases.py; only outputs are stored: the loss, and per
gradient its first 64 entries and (sum, abs-sum, sum of squares).   Run:  python tests/golden/make_golden_v4.py
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


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).float()


for tr in C.TRAIN2D_V4:
    spec = C.by_name(tr["base"])
    case = C.build_case(spec)
    m = sd2.GNN_Diffusion(steps=spec["steps"], sampling="DDIM", inference_ratio=1, noise_weight=1.0, rotation=(spec["c"] == 4),
                          model_mean_type=getattr(sd2.ModelMeanType, tr["mean"]), visual_pretrained=False,
                          architecture=spec["arch"], virt_nodes=spec["V"])
    missing, unexpected = m.model.load_state_dict(case["sd"], strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("linear1.", "linear2.", "visual_backbone.", "mean", "std")) for k in missing), missing
    m.train()
    m.visual_features = lambda cond, _f=case["feats"]: _f
    rng = np.random.default_rng(tr["seed"])
    noise = torch.from_numpy(rng.standard_normal(tuple(case["x"].shape)).astype(np.float32))
    loss = m.p_losses(case["x"], case["t"], noise=noise, loss_type="huber", cond=None, edge_index=case["edge_index"],
                      batch=case["batch"])
    loss.backward()
    OUT[f"{tr['name']}/loss"] = loss.detach().numpy()
    n = 0
    for k, p in m.model.named_parameters():
        if p.grad is not None and k in case["sd"]:
            OUT[f"{tr['name']}/grad_stats/{k}"] = stats(p.grad).numpy()
            OUT[f"{tr['name']}/grad_head/{k}"] = p.grad.flatten()[:64].detach().numpy()
            n += 1
    print("train", tr["name"], float(loss), n, "gradients", flush=True)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v4.npz")
np.savez_compressed(path, **OUT)
print("wrote", path, os.path.getsize(path), "bytes,", len(OUT), "arrays")
