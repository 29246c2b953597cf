"""Import the reference's own hot-path Python under inert ``sys.modules`` stubs.

BUILD-CONTAINER ONLY (needs /root/reference; never runs on the GPU box).  Used by
``make_golden.py`` to produce the fixtures in this directory.  No reference source is
copied: the reference files are imported from where they lie.

Missing third-party packages (torch_geometric, pytorch_lightning, timm, kornia, wandb,
torchmetrics, torchvision, pytorch3d, trimesh, torch_scatter) are replaced by inert
stubs, except the ones that carry arithmetic on the path, which are bound to the
oracle's restatement (oracle/pyg_restatement.py): ``torch_geometric.nn.TransformerConv``,
``pytorch3d.transforms.{matrix_to_quaternion, quaternion_to_matrix, quaternion_apply}`` and
``pytorch3d.ops.knn.knn_points`` (K = 1; only the 3D evaluation metrics call the last two).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference/puzzle_diff"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))


class _Anything:
    """Inert callable/attribute sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []          # behave as a package so submodule imports resolve
    sys.modules[name] = m
    return m


class _LightningModule(nn.Module):
    """pytorch_lightning.LightningModule minus the Trainer: nn.Module + the few hooks
    the reference's __init__ / sampling code touches."""

    def __init__(self, *a, **k):
        super().__init__()

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    @property
    def device(self):
        return torch.device("cpu")

    @property
    def local_rank(self):
        return 0


def install_stubs():
    import transformers.optimization  # noqa: F401  (real Adafactor; import BEFORE stubbing torchvision)
    from oracle import pyg_restatement as R

    _mod("pytorch_lightning", LightningModule=_LightningModule, Trainer=_Anything)
    _mod("timm", create_model=lambda *a, **k: nn.Identity())
    _mod("wandb", Image=_Anything, log=lambda *a, **k: None)
    _mod("trimesh")
    _mod("torch_scatter", scatter=_Anything())
    _mod("torchmetrics", MeanMetric=_Anything, SumMetric=_Anything, Metric=object)
    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms")
    tvf = _mod("torchvision.transforms.functional", rotate=_Anything())
    tv.transforms = tvt
    tvt.functional = tvf
    _mod("kornia")
    _mod("kornia.geometry")
    _mod("kornia.geometry.transform", Rotate=_Anything)
    tg = _mod("torch_geometric")
    tgnn = _mod("torch_geometric.nn", TransformerConv=R.TransformerConv, GraphNorm=_Anything,
                GCNConv=_Anything)
    _mod("torch_geometric.nn.models")
    _mod("torch_geometric.nn.conv")
    _mod("torch_geometric.nn.conv.transformer_conv", TransformerConv=R.TransformerConv)
    _mod("torch_geometric.graphgym")
    _mod("torch_geometric.graphgym.register", register_layer=lambda *a, **k: (lambda c: c))
    tg.nn = tgnn
    _mod("pytorch3d")
    _mod("pytorch3d.transforms", matrix_to_quaternion=R.matrix_to_quaternion,
         quaternion_to_matrix=R.quaternion_to_matrix, matrix_to_euler_angles=_Anything(),
         rotation_6d_to_matrix=_Anything(), matrix_to_rotation_6d=_Anything(),
         quaternion_apply=R.quaternion_apply, quaternion_multiply=_Anything(),
         quaternion_invert=_Anything(), euler_angles_to_matrix=_Anything(),
         axis_angle_to_matrix=_Anything(), matrix_to_axis_angle=_Anything(),
         random_quaternions=_Anything(), random_rotations=_Anything(),
         so3_exp_map=_Anything(), so3_log_map=_Anything(), Transform3d=_Anything,
         axis_angle_to_quaternion=_Anything(), quaternion_to_axis_angle=_Anything())
    _mod("pytorch3d.ops")
    _mod("pytorch3d.ops.knn", knn_gather=_Anything(), knn_points=R.knn_points)
    _mod("pytorch3d.structures")
    _mod("pytorch3d.structures.pointclouds", Pointclouds=_Anything)
    # backbones/__init__.py:1 imports a file that is not in the tree
    _mod("model.backbones.backbone_vist", Eff_GAT_Vist=_Anything)


def import_reference():
    """Returns (spatial_diffusion module, 3D double-diffusion module)."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sd2 = importlib.import_module("model.spatial_diffusion")
    sd3 = importlib.import_module("model.spatial_diffusion_3d_test_double_diffusion")
    return sd2, sd3


def import_reference_dataset():
    """The reference's dataset/puzzle_dataset.py (graph generators: generate_random_regular_graph :115-152) loaded from
    where it lies, with inert stand-ins for the PyG data / torchvision transform classes it subclasses."""
    install_stubs()

    class _Cls:
        def __init__(self, *a, **k):
            pass

    _mod("torch_geometric.data", Data=_Cls, Dataset=_Cls, Batch=_Cls)
    _mod("torch_geometric.data.datapipes", functional_transform=lambda *a, **k: (lambda c: c))
    _mod("torch_geometric.loader", DataLoader=_Cls)
    _mod("torch_geometric.transforms", BaseTransform=_Cls)
    _mod("torch_geometric.utils", get_laplacian=_Anything(), to_scipy_sparse_matrix=_Anything(), dense_to_sparse=_Anything())
    tvt = sys.modules["torchvision.transforms"]
    tvt.InterpolationMode = _Anything()
    tvt.RandomResizedCrop = _Cls
    tvt.RandomHorizontalFlip = _Cls
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_puzzle_dataset", os.path.join(REF, "dataset", "puzzle_dataset.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == "__main__":
    a, b = import_reference()
    print("imported", a.__file__, b.__file__)
