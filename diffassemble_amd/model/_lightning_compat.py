"""``pytorch_lightning`` / ``torchmetrics`` are the reference's trainer-side dependencies
 and are absent from the build image.  When they are importable the real
classes are used, so the reference drivers (train_*.py / viz_script.py) get a genuine
LightningModule; otherwise these minimal stand-ins keep the module surface importable and
usable from plain Python (tests, bench, smoke)."""
import torch
import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()
            self._logged = {}

        def save_hyperparameters(self, *args, **kwargs):
            pass

        def log(self, name, value, *args, **kwargs):
            self._logged[name] = value

        def log_dict(self, d, *args, **kwargs):
            self._logged.update(dict(d))

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            for b in self.buffers():
                return b.device
            return torch.device("cpu")

        @property
        def local_rank(self):
            return 0

try:  # pragma: no cover
    import torchmetrics
    MeanMetric, SumMetric = torchmetrics.MeanMetric, torchmetrics.SumMetric
except Exception:  # noqa: BLE001
    class _Acc(nn.Module):
        def __init__(self):
            super().__init__()
            self.total, self.count = 0.0, 0

        def update(self, value):
            v = torch.as_tensor(value, dtype=torch.float64).flatten()
            self.total += float(v.sum())
            self.count += v.numel()

        def reset(self):
            self.total, self.count = 0.0, 0

    class MeanMetric(_Acc):
        def compute(self):
            return torch.tensor(self.total / max(self.count, 1))

    class SumMetric(_Acc):
        def compute(self):
            return torch.tensor(self.total)
