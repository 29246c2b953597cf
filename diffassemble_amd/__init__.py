"""diffassemble_amd -- MI355X-native (gfx950) denoiser + sampling loop of DiffAssemble.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every
arithmetic op on the hot path runs in hand-written HIP kernels behind the C ABI of
include/diffassemble_hip.h.  ``diffassemble_amd.model`` mirrors the reference's
``puzzle_diff/model`` module surface.
"""
from . import _lib  # noqa: F401
from .engine import DenoiserEngine, Schedule  # noqa: F401
from .graph_plan import GraphPlan, build_plan, exophormer_edge_index  # noqa: F401

__all__ = ["DenoiserEngine", "Schedule", "GraphPlan", "build_plan", "exophormer_edge_index"]
