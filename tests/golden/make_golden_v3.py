"""Generate tests/golden/golden_v3.npz from the REFERENCE's own 3D piece encoder.

BUILD-CONTAINER ONLY (imports /root/reference/puzzle_diff/model/backbones/vnn/vn_dgcnn.py from where it lies, under the
stubs of ref_import.py).  The reference hard-codes ``torch.device('cuda')`` in ``get_graph_feature`` (vn_dgcnn.py:94);
the module-level name ``torch`` of the imported module is replaced by a proxy whose ``device()`` answers "cpu", nothing
else changes.  Stores, per case of cases.PCD_ENC (eval mode, randomised BatchNorm running statistics):

* ``out``      VN_DGCNN(128, inv).forward(clouds)  [P, 768] or [P, 256]
* ``x1_stats`` sum / abs-sum / square-sum of the first pooled vector-neuron map (output of ``pool1``)
* ``idx1``     the reference's k-nearest-neighbour lists of the first stage (small cases only)

Inputs and weights are regenerated from seeds by cases.py.   Run:  python tests/golden/make_golden_v3.py
"""
import importlib
import os
import sys

sys.dont_write_bytecode = True
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases as C  # noqa: E402
from ref_import import REF, install_stubs  # noqa: E402

install_stubs()
sys.path.insert(0, REF)
vn = importlib.import_module("model.backbones.vnn.vn_dgcnn")


class _TorchOnCPU:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")


vn.torch = _TorchOnCPU()
OUT = {}
for spec in C.PCD_ENC:
    sd, pts = C.pcd_encoder_case(spec)
    net = vn.VN_DGCNN(128, inv=spec["inv"]).eval()
    net.load_state_dict(sd, strict=True)
    grabbed = {}
    real_knn = vn.knn

    def spy(x, k, _g=grabbed):
        idx = real_knn(x, k)
        _g.setdefault("idx", []).append(idx)
        return idx

    vn.knn = spy
    x1 = []
    orig_pool = net.pool1
    net.pool1 = lambda x, _o=orig_pool: (x1.append(_o(x)) or x1[-1])
    with torch.no_grad():
        out = net(pts)
    vn.knn = real_knn
    OUT[f"pcd_enc/{spec['name']}/out"] = out.numpy()
    t = x1[0].double()
    OUT[f"pcd_enc/{spec['name']}/x1_stats"] = np.array([t.sum(), t.abs().sum(), (t * t).sum()], dtype=np.float64)
    if spec["N"] <= 256:
        OUT[f"pcd_enc/{spec['name']}/idx1"] = grabbed["idx"][0].numpy().astype(np.int32)
    print(spec["name"], out.shape, float(out.abs().max()))

np.savez_compressed(os.path.join(os.path.dirname(__file__), "golden_v3.npz"), **OUT)
print("wrote golden_v3.npz", len(OUT), "arrays")
