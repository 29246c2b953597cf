"""Training side of the denoiser: flat fp32 parameter / gradient buffers bound to the
module's own ``nn.Parameter``s, and the ``torch.autograd.Function`` that routes
``Eff_GAT.forward_with_feats`` + its backward through ``da_train_forward`` / ``da_train_backward``.

Layout.  All live denoiser parameters sit in ONE flat fp32 buffer (``flat``) and their gradients in a
second one of the same layout (``flat_grad``); every ``nn.Parameter`` of the module becomes a view of
its slot, so (a) the HIP kernels read the live weights with no packing or copy, (b) the four PyG
``lin_query | lin_key | lin_value | lin_skip`` matrices of a conv are adjacent and run as one fused
[4*H*C, Din] projection, (c) data-parallel training needs exactly one all-reduce -- of ``flat_grad`` --
per optimizer step (``sharding.allreduce_gradients``: RCCL over xGMI on the GPU box), and (d) the
reference's optimizer (``Adafactor(self.parameters())``, spatial_diffusion.py:701-705) keeps working on
the parameters unchanged, in place.

The kernels ADD into ``flat_grad``.  When a parameter's ``.grad`` is not (or no longer, after
``zero_grad(set_to_none=True)``) the engine's view, the next backward starts from zero and re-attaches the
views; otherwise it accumulates, like autograd does.

Precision.  Storage is fp32 throughout (the reference trains in fp32; its gradient fixtures are fp32).  ``TrainEngine.precision``
chooses how the matrix-core GEMMs of a step take their operands: ``"fp32"`` (default: exact products, the mode of the
fixtures) or ``"bf16"`` (operands rounded to bf16 on their way into the matrix cores, fp32 accumulation --
``DA_TRAIN_MMA_BF16`` of include/diffassemble_hip.h; what ``torch.autocast(bfloat16)`` does to the reference's Linear /
matmul calls).  Default from the environment (``DIFFASSEMBLE_TRAIN_PRECISION``, the switch the piece encoder's training
path reads too), settable per engine / through ``Eff_GAT.train_precision``.
"""
import ctypes as C
import os

import torch

from . import _lib
from .graph_plan import GraphPlan


def _param_order(module):
    """(name, fused-group id or None) in flat-buffer order; names relative to the Eff_GAT module."""
    names = ["time_emb.weight", "pos_mlp.0.weight", "pos_mlp.0.bias", "pos_mlp.2.weight", "pos_mlp.2.bias",
             "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"]
    n_layers = len(module.gnn_backbone.module_list)
    # (the virtual-node embedding sits in FRONT of the convs: its gradient is produced after conv 0's, so it belongs to the
    #  late bucket of the data-parallel exchange -- everything before conv 1 -- see TrainEngine.backward)
    if hasattr(module.gnn_backbone, "virt_node_embedding"):
        names.append("gnn_backbone.virt_node_embedding.weight")
    for l in range(n_layers):
        p = f"gnn_backbone.module_list.{l}."
        names += [p + f"lin_{k}.weight" for k in ("query", "key", "value", "skip")]
        names += [p + f"lin_{k}.bias" for k in ("query", "key", "value", "skip")]
    names += ["final_mlp.0.weight", "final_mlp.0.bias", "final_mlp.2.weight", "final_mlp.2.bias"]
    return names, n_layers


class TrainEngine:
    """Owns the flat buffers + training workspace of one ``Eff_GAT`` module on one GPU."""

    def __init__(self, module, device=None):
        self.lib = _lib.lib()
        params = dict(module.named_parameters())
        names, self.n_layers = _param_order(module)
        dev = torch.device(device) if device is not None else params[names[0]].device
        if dev.type != "cuda":
            raise _lib.DaError("TrainEngine needs a ROCm device (no CPU path in diffassemble_amd)")
        self.device = dev
        self.names = names
        self.params = [params[n] for n in names]
        offs, off = [], 0
        for n, p in zip(names, self.params):
            fused_tail = any(n.endswith(f"lin_{k}.{w}") for k in ("key", "value", "skip") for w in ("weight", "bias"))
            if not fused_tail:
                off = (off + 63) // 64 * 64            # 256-byte aligned slots; fused groups stay gap-free
            offs.append(off)
            off += p.numel()
        self.total = (off + 63) // 64 * 64
        # data-parallel buckets, in the order backward completes them: EARLY = [early_off, total) = convs 1 .. L-1 and final_mlp
        # (final once da_train_backward_stage(EARLY) has run), LATE = [0, early_off) = embeddings, mlp, virtual nodes, conv 0
        # (a one-layer backbone has no conv 1: its early bucket is final_mlp alone, which is what the library's stage cut leaves final)
        early_name = "gnn_backbone.module_list.1.lin_query.weight" if self.n_layers > 1 else "final_mlp.0.weight"
        self.early_off = offs[names.index(early_name)]
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.views, self.grad_views = [], []
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                p.data = v
                self.views.append(v)
                self.grad_views.append(self.flat_grad[o:o + p.numel()].view(p.shape))
        gnn = module.gnn_backbone
        self.arch = gnn.arch
        self.virt_nodes = int(getattr(gnn, "virt_nodes", 0)) if self.arch == "exophormer" else 0
        by = dict(zip(names, self.views))
        self.D = by["mlp.0.weight"].shape[1]
        self.F = self.D - 64
        self.c_in = by["pos_mlp.0.weight"].shape[1]
        self.c_out = by["final_mlp.2.weight"].shape[0]
        self.steps = by["time_emb.weight"].shape[0]
        self.hidden = by["mlp.0.weight"].shape[0]
        self.w = self._weights_struct(dict(zip(names, self.views)))
        self.gw = self._weights_struct(dict(zip(names, self.grad_views)))
        self._ws = None
        self._ws_key = None
        # the one parameter autograd sees (DenoiserTrainFn): the smallest trainable tensor
        self.anchor = min((p for p in self.params if p.requires_grad), key=lambda p: p.numel(), default=self.params[0])
        self.precision = os.environ.get("DIFFASSEMBLE_TRAIN_PRECISION", "fp32") or "fp32"      # "fp32" | "bf16" (module docstring)
        self.version = 0              # bumped by every raw-pointer update of ``flat`` (FusedAdafactor.step)
        self.grads_synced = False     # True between sync_gradients() and the next backward
        # bucketed exchange (Lightning DDP's reducer overlaps its buckets with backward, train_script.py:215-218): on by default
        # whenever a gradient exchange is active; ``overlap_exchange = False`` = one serial all-reduce after backward
        self.overlap_exchange = True
        self._side = None             # side stream of the early bucket's all-reduce
        self._early_pending = False   # the early bucket is being / has been averaged on the side stream since the last sync

    def sync_gradients(self, average=True):
        """Data-parallel exchange (the reference gets it from Lightning's ``strategy="ddp"``,
        train_script.py:215-218): ONE all-reduce of ``flat_grad`` over the default process group, at most once
        per set of accumulated backward passes.  No-op without an initialised multi-rank group."""
        from .sharding import allreduce_gradients, exchange_active
        if self.grads_synced:
            return
        if self._early_pending:
            # the early bucket was averaged on the side stream while the rest of backward ran; what is left is the late bucket
            allreduce_gradients(self.flat_grad[:self.early_off], average=average)
            self._join_side()
            if not average and exchange_active():
                import torch.distributed as dist
                self.flat_grad[self.early_off:].mul_(dist.get_world_size())
        else:
            allreduce_gradients(self.flat_grad, average=average)
        self.grads_synced = True

    def _join_side(self):
        """The caller's stream waits for the early bucket's exchange (before anything rewrites ``flat_grad``)."""
        if self._early_pending:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._early_pending = False

    def _exchange_early(self):
        """Average the early bucket over the ranks on the side stream, behind everything the caller's stream holds so far
        (= da_train_backward_stage(EARLY)).  AVERAGED here, not summed: with gradient accumulation every backward exchanges
        its early bucket, and avg(avg(g1) + g2_r) = avg(g1) + avg(g2) where a sum would count g1 world-size times."""
        from .sharding import allreduce_gradients
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            allreduce_gradients(self.flat_grad[self.early_off:], average=True)
        self._early_pending = True

    def _weights_struct(self, by):
        w = _lib.DaWeights()
        w.variant = _lib.VARIANT_2D
        w.arch = _lib.ARCH_EXOPHORMER if self.arch == "exophormer" else _lib.ARCH_TRANSFORMER
        w.steps, w.c_in, w.c_out, w.feat_dim, w.hidden = self.steps, self.c_in, self.c_out, self.F, self.hidden
        w.heads, w.n_layers, w.virt_nodes = 8, self.n_layers, self.virt_nodes
        P = lambda n: by[n].data_ptr()  # noqa: E731
        w.time_emb = P("time_emb.weight")
        w.pos_w0, w.pos_b0, w.pos_w1, w.pos_b1 = (P("pos_mlp.0.weight"), P("pos_mlp.0.bias"), P("pos_mlp.2.weight"),
                                                  P("pos_mlp.2.bias"))
        w.mlp_w0, w.mlp_b0, w.mlp_w1, w.mlp_b1 = (P("mlp.0.weight"), P("mlp.0.bias"), P("mlp.2.weight"),
                                                  P("mlp.2.bias"))
        for l in range(self.n_layers):
            p = f"gnn_backbone.module_list.{l}."
            w.conv_wq[l], w.conv_bq[l] = P(p + "lin_query.weight"), P(p + "lin_query.bias")
            w.conv_wk[l], w.conv_bk[l] = P(p + "lin_key.weight"), P(p + "lin_key.bias")
            w.conv_wv[l], w.conv_bv[l] = P(p + "lin_value.weight"), P(p + "lin_value.bias")
            w.conv_ws[l], w.conv_bs[l] = P(p + "lin_skip.weight"), P(p + "lin_skip.bias")
        if self.virt_nodes > 0:
            w.virt_emb = P("gnn_backbone.virt_node_embedding.weight")
        w.head_w0, w.head_b0 = P("final_mlp.0.weight"), P("final_mlp.0.bias")
        w.head_w1, w.head_b1 = P("final_mlp.2.weight"), P("final_mlp.2.bias")
        return w

    # ------------------------------------------------------------------ plumbing
    def still_bound(self):
        """False once somebody replaced a parameter's storage (``.to()``, ``load_state_dict`` with
        assign, ...): the caller rebuilds the engine."""
        return all(p.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    pair_cap_mb = 8192          # fp32 hybrid training: largest pair-matrix allocation before the edge-list kernels are preferred

    def _edge_list_route(self, plan: GraphPlan):
        """True: this plan trains on the edge-list (CSR) kernels although it carries a hybrid split.  The hybrid thresholds of
        ``graph_plan._hybrid_worth_it`` (32 nodes, 1 % density) were measured on inference steps; the fp32 training route of a hybrid
        plan allocates [n, n] pair matrices per head and layer (+ dP), which was only ever measured to pay from 256-node graphs at
        3 % density up (round 4) -- below that, or beyond ``pair_cap_mb`` (attribute, default 8192 MB) of pair matrices, fp32 training keeps
        the edge-list kernels.  The bf16-operand mode runs hybrid graphs flash-style (no pair matrix): it always takes them."""
        if _lib.config().train_attn == 0:          # da_config.train_attn (DA_TRAIN_ATTN=0): edge-list kernels only
            return True
        from .graph_plan import _hybrid_mode
        if not plan.hybrid or plan.dense or self.precision == "bf16" or _hybrid_mode() == "force":      # (force: the caller asked for the hybrid kernels)
            return False
        n = int(plan.max_graph_nodes)
        density = plan.n_edges / max(1.0, float(plan.n_graphs) * n * n)
        pair_mb = (self.n_layers + 1) * 8 * float(plan.n_graphs) * n * ((n + 63) // 64 * 64) * 4 / 2**20
        return n < 256 or density < 0.03 or pair_mb > float(self.pair_cap_mb)

    def _cg(self, plan: GraphPlan):
        """da_graph of a training plan: complete and hybrid graphs run on the grouped-GEMM attention (da_train_dense.hip) and
        never walk the full edge list, so its CSR is not built for them (unless ``_edge_list_route`` says so); hybrid plans carry
        the by-source orientation of their REMAINDER edges in out_ptr / out_dst.  A hybrid plan in the banded slot layout (the
        inference default of ``expander_plan``) cannot be trained on -- the hybrid training kernels index the adjacency by node --
        so it is refused HERE with advice that can be followed."""
        edge_list = self._edge_list_route(plan)
        if plan.hybrid and plan.slot_node is not None and not edge_list:
            raise _lib.DaError("training on a hybrid plan in the banded slot layout: build the training plan with "
                               "expander_plan(..., banded=False) (or DA_EXPANDER_LAYOUT=natural), or from the edge list with build_plan")
        g = plan.c_struct(edge_list or not (plan.dense or plan.hybrid), inference_hints=False)
        if edge_list and plan.hybrid:
            g.hybrid = 0               # the library then takes the CSR route (da_train.hip dims_of)
            g.slot_node = None
        return g

    def _workspace(self, plan: GraphPlan, mma=None):
        g = self._cg(plan)
        need = int(self.lib.da_train_workspace_bytes_ex(C.byref(self.w), C.byref(g), self._mma() if mma is None else mma))     # (bf16 mode on hybrid graphs: no pair matrices)
        if need == 0:
            _lib.check(1)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, plan: GraphPlan, x, t, feats):
        """da_train_forward: out [n_real, c_out] fp32; activations stay in the workspace for backward."""
        x = x.detach().to(self.device, torch.float32).contiguous()
        t = t.detach().to(self.device, torch.int64).contiguous()
        feats = feats.detach().to(self.device, torch.float32).contiguous()
        assert x.shape == (plan.n_real, self.c_in) and feats.shape == (plan.n_real, self.F), (x.shape, feats.shape)
        out = torch.empty((plan.n_real, self.c_out), dtype=torch.float32, device=self.device)
        ws = self._workspace(plan)
        g = self._cg(plan)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.da_train_forward_ex(C.byref(self.w), C.byref(g), _lib.ptr(x), _lib.ptr(t), _lib.ptr(feats),
                                                    _lib.ptr(out), _lib.ptr(ws), ws.numel(), self._mma(), _lib.stream_ptr(self.device)))
        self._fwd_mma = self._mma()
        return out

    def _mma(self):
        if self.precision not in ("fp32", "bf16"):
            raise _lib.DaError(f"TrainEngine.precision must be 'fp32' or 'bf16', not {self.precision!r}")
        return _lib.TRAIN_MMA_BF16 if self.precision == "bf16" else _lib.TRAIN_MMA_FP32

    def backward(self, plan: GraphPlan, x, t, d_out, want_dfeats=False):
        """da_train_backward: adds every parameter gradient into ``flat_grad`` (and attaches the views
        as ``.grad``); returns d_feats [n_real, F] or None.

        COLLECTIVE under the bucketed exchange: with an initialised multi-rank process group and ``overlap_exchange`` (the
        default), this call issues the early bucket's all-reduce on the side stream -- every rank must then run the SAME number
        of backward passes between two ``sync_gradients`` calls (what DDP's reducer requires as well).  A caller whose ranks
        run different numbers of passes sets ``engine.overlap_exchange = False`` : backward
        is then local and only ``sync_gradients`` communicates.  (A one-rank ``nccl`` group still executes the call -- it cannot deadlock.)"""
        from .sharding import exchange_active
        plan.with_source_csr(self._edge_list_route(plan))
        self._join_side()             # (accumulation: the previous backward's early bucket must have landed before this one adds to it)
        attached = all(p.grad is not None and p.grad.data_ptr() == gv.data_ptr()
                       for p, gv in zip(self.params, self.grad_views))
        if not attached:
            self.flat_grad.zero_()
        self.grads_synced = False
        x = x.detach().to(self.device, torch.float32).contiguous()
        t = t.detach().to(self.device, torch.int64).contiguous()
        d_out = d_out.detach().to(self.device, torch.float32).contiguous()
        d_feats = torch.empty((plan.n_real, self.F), dtype=torch.float32, device=self.device) if want_dfeats else None
        mma = getattr(self, "_fwd_mma", self._mma())       # the mode of the forward it follows
        ws = self._workspace(plan, mma)
        g = self._cg(plan)
        overlap = self.overlap_exchange and exchange_active()
        staged = overlap or getattr(self, "force_staged", False)       # (force_staged: tests -- the two halves without a process group)
        with torch.cuda.device(self.device):
            for stage in ((_lib.TRAIN_BWD_EARLY, _lib.TRAIN_BWD_LATE) if staged else (_lib.TRAIN_BWD_ALL,)):
                _lib.check(self.lib.da_train_backward_stage(C.byref(self.w), C.byref(self.gw), C.byref(g), _lib.ptr(x), _lib.ptr(t),
                                                            _lib.ptr(d_out), _lib.ptr(d_feats), _lib.ptr(ws), ws.numel(), mma, stage,
                                                            _lib.stream_ptr(self.device)))
                if stage == _lib.TRAIN_BWD_EARLY and overlap:
                    self._exchange_early()
        if not attached:
            for p, gv in zip(self.params, self.grad_views):
                p.grad = gv
        return d_feats


class DenoiserTrainFn(torch.autograd.Function):
    """out = Eff_GAT.forward_with_feats(x, t, feats) with the backward in the HIP library.  Parameter gradients are
    written straight into ``TrainEngine.flat_grad`` (aliased by ``param.grad``), not returned to autograd.

    ONE parameter -- the ``anchor`` (the engine's smallest tensor) -- is passed as an autograd input and receives a ZERO
    gradient tensor (its real gradient went into its ``flat_grad`` slot like everybody else's, so ``grad += 0`` changes
    nothing).  Reason: a ``torch.nn.parallel.DistributedDataParallel`` wrapper (what ``pl.Trainer(strategy="ddp")``
    builds, train_script.py:215-218) only finishes an iteration when at least one of its autograd hooks fires.  With the
    anchor the reducer sees one used parameter (whose bucket it averages -- harmless, the fused exchange averages the same
    slot again) and, under ``find_unused_parameters=True``, classifies every other parameter as unused on all ranks and
    leaves its ``.grad`` alone; ``GNN_Diffusion.on_before_optimizer_step`` then does the one fused all-reduce.  With
    ``find_unused_parameters=False`` torch raises "Expected to have finished reduction in the prior iteration" on the
    second iteration (tests/test_gpu_train.py::test_real_ddp_wrapper_*).
    One forward must be followed by its backward before the next forward (shared workspace)."""

    @staticmethod
    def forward(ctx, eng, plan, x, t, feats, anchor):
        out = eng.forward(plan, x, t, feats)
        ctx.eng, ctx.plan = eng, plan
        ctx.want_dfeats = bool(feats.requires_grad)
        ctx.save_for_backward(x, t)
        ctx.anchor_shape = anchor.shape
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, t = ctx.saved_tensors
        d_feats = ctx.eng.backward(ctx.plan, x, t, d_out, ctx.want_dfeats)
        return None, None, None, None, d_feats, torch.zeros(ctx.anchor_shape, dtype=torch.float32, device=d_out.device)


class FusedAdafactor(torch.optim.Optimizer):
    """The reference's optimizer (``Adafactor(self.parameters())`` with transformers' defaults,
    spatial_diffusion.py:701-705) as ONE library call over the training engine's flat buffers:
    ``da_adafactor_step`` -- four launches, deterministic reductions, no host sync -- instead of ~700 tiny
    torch launches.  Covers exactly the parameters the engine owns (the live denoiser parameters); any
    other parameter handed in must have no gradient (the reference's dead ``linear1/linear2`` never do).
    Hyper-parameters are the transformers defaults the reference relies on; they are arguments only so
    the parity test can vary them."""

    def __init__(self, params, engine: TrainEngine, eps=(1e-30, 1e-3), clip_threshold=1.0, decay_rate=-0.8):
        super().__init__(list(params), dict(eps=eps, clip_threshold=clip_threshold, decay_rate=decay_rate))
        import numpy as np
        self.engine = engine
        self.lib = _lib.lib()
        dev = engine.device
        base = engine.flat.data_ptr()
        ptab = np.zeros((len(engine.views), 7), dtype=np.int64)      # 56-byte records, see include/diffassemble_hip.h
        blocks = []
        s_off = c_off = 0
        for pid, v in enumerate(engine.views):
            off = (v.data_ptr() - base) // 4
            fact = v.dim() >= 2
            rows, cols = (v.numel() // v.shape[-1], v.shape[-1]) if fact else (1, v.numel())
            if fact and cols > 1280:
                raise _lib.DaError(f"FusedAdafactor: {cols} columns > 1280")
            blk0 = len(blocks)
            if fact:
                rb = max(1, 8192 // cols)
                for r0 in range(0, rows, rb):
                    blocks.append((pid, r0, min(rb, rows - r0)))
                row_off, col_off = s_off, s_off + rows
                s_off += rows + cols
            else:
                for e0 in range(0, cols, 4096):
                    blocks.append((pid, e0, min(4096, cols - e0)))
                row_off, col_off = s_off, 0
                s_off += cols
            nblk = len(blocks) - blk0
            rec = np.zeros(14, dtype=np.int32)
            rec[0:2] = np.array([off], dtype=np.int64).view(np.int32)
            rec[2], rec[3], rec[4] = rows, cols, int(fact)
            rec[6:8] = np.array([row_off], dtype=np.int64).view(np.int32)
            rec[8:10] = np.array([col_off], dtype=np.int64).view(np.int32)
            rec[10], rec[11] = blk0, nblk
            rec[12:14] = np.array([c_off], dtype=np.int64).view(np.int32)
            ptab[pid] = rec.view(np.int64)
            if fact:
                c_off += nblk * cols
        self.n_params, self.n_blocks = len(engine.views), len(blocks)
        self.ptab = torch.from_numpy(ptab).to(dev)
        self.btab = torch.from_numpy(np.asarray(blocks, dtype=np.int32)).to(dev)
        self.state_buf = torch.zeros(max(s_off, 1), dtype=torch.float32, device=dev)
        self.scratch = torch.empty(2 * self.n_blocks + 4 * self.n_params + c_off + 64, dtype=torch.float32, device=dev)
        self.step_count = 0

    # ------------------------------------------------------------------ checkpoint / resume
    def _slots(self):
        """(index in the optimizer's parameter list, parameter, view, (rows, cols, factored, row_off, col_off))"""
        order = {id(p): i for i, p in enumerate(q for g in self.param_groups for q in g["params"])}
        tab = self.ptab.cpu().numpy().view("int32").reshape(self.n_params, 14)
        for pid, (p, v) in enumerate(zip(self.engine.params, self.engine.views)):
            r = tab[pid]
            row_off = int(r[6:8].view("int64")[0])
            col_off = int(r[8:10].view("int64")[0])
            yield order[id(p)], p, v, (int(r[2]), int(r[3]), bool(r[4]), row_off, col_off)

    def state_dict(self):
        """The second-moment statistics in transformers' per-parameter layout (``step``, ``exp_avg_sq_row`` [shape[:-1]],
        ``exp_avg_sq_col`` [shape[:-2] + shape[-1:]] for matrices, ``exp_avg_sq`` for vectors), keyed by the parameter's index
        in this optimizer: what ``Adafactor(...).state_dict()`` holds, so Lightning's checkpoint / resume keeps them (the
        base class would save an empty state: the statistics live in one flat device buffer)."""
        sd = super().state_dict()
        state = {}
        for idx, p, v, (rows, cols, fact, ro, co) in self._slots():
            st = {"step": self.step_count}
            if fact:
                st["exp_avg_sq_row"] = self.state_buf[ro:ro + rows].clone().view(v.shape[:-1])
                st["exp_avg_sq_col"] = self.state_buf[co:co + cols].clone().view(v.shape[:-2] + v.shape[-1:])
            else:
                st["exp_avg_sq"] = self.state_buf[ro:ro + cols].clone().view(v.shape)
            state[idx] = st
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        state = sd.get("state", {})
        steps = set()
        with torch.no_grad():
            for idx, p, v, (rows, cols, fact, ro, co) in self._slots():
                st = state.get(idx, state.get(str(idx)))
                if st is None:
                    continue
                steps.add(int(st["step"]))
                if fact:
                    self.state_buf[ro:ro + rows].copy_(st["exp_avg_sq_row"].reshape(-1))
                    self.state_buf[co:co + cols].copy_(st["exp_avg_sq_col"].reshape(-1))
                else:
                    self.state_buf[ro:ro + cols].copy_(st["exp_avg_sq"].reshape(-1))
        if len(steps) > 1:
            raise _lib.DaError(f"FusedAdafactor.load_state_dict: parameters at different steps {sorted(steps)}")
        if steps:
            self.step_count = steps.pop()

    def zero_grad(self, set_to_none: bool = True):
        """One fill of the flat gradient buffer (the views stay attached as ``.grad``) instead of one
        launch per parameter; parameters outside the engine are handled the usual way."""
        mine = {id(p) for p in self.engine.params}
        self.engine._join_side()
        self.engine.flat_grad.zero_()
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) not in mine and p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self.engine
        mine = {id(p) for p in eng.params}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None and id(p) not in mine:
                    raise _lib.DaError("FusedAdafactor only updates the TrainEngine's parameters; use the reference's "
                                       "Adafactor for the others")
        if any(p.grad is None for p in eng.params):
            return loss                                   # nothing back-propagated yet
        g = self.defaults
        self.step_count += 1
        eng.sync_gradients()          # no-op on one rank / when GNN_Diffusion.on_before_optimizer_step already did it
        eng.version += 1              # invalidates packed inference weights (DenoiserBase._param_version)
        with torch.cuda.device(eng.device):
            _lib.check(self.lib.da_adafactor_step(
                self.n_params, _lib.ptr(self.ptab), self.n_blocks, _lib.ptr(self.btab), _lib.ptr(eng.flat),
                _lib.ptr(eng.flat_grad), _lib.ptr(self.state_buf), _lib.ptr(self.scratch), self.scratch.numel(),
                self.step_count, g["eps"][0], g["eps"][1], g["clip_threshold"], g["decay_rate"],
                _lib.stream_ptr(eng.device)))
        return loss


class HybridAdafactor(torch.optim.Optimizer):
    """The reference's single ``Adafactor(self.parameters())`` (spatial_diffusion.py:701-705) when a trainable piece
    encoder is attached: the denoiser's parameters go through ``FusedAdafactor`` (one library call over the flat buffers),
    every other parameter with a gradient (the encoder's 5-D group-convolution weights, BatchNorm affines, the two wide
    linears -- shapes the fused kernel's row / column tables do not cover) through transformers' own Adafactor with the
    same defaults, so the update rule is the reference's for all of them.  Presents itself as ONE optimizer (Lightning's
    automatic optimisation wants exactly one)."""

    def __init__(self, params, engine: TrainEngine):
        from transformers.optimization import Adafactor
        params = list(params)
        mine = {id(p) for p in engine.params}
        rest = [p for p in params if id(p) not in mine and p.requires_grad]
        super().__init__(params, {})
        self.fused = FusedAdafactor([p for p in params if id(p) in mine], engine)
        self.rest = Adafactor(rest) if rest else None

    def zero_grad(self, set_to_none: bool = True):
        self.fused.zero_grad(set_to_none)
        if self.rest is not None:
            self.rest.zero_grad(set_to_none)

    def state_dict(self):
        return {"fused": self.fused.state_dict(), "rest": self.rest.state_dict() if self.rest is not None else None,
                "param_groups": super().state_dict()["param_groups"], "state": {}}

    def load_state_dict(self, sd):
        """Accepts its own layout ({"fused", "rest"}) and a PLAIN transformers-Adafactor state dict over the same parameter
        list (a run with ``fused_optimizer = False``, or the reference's own checkpoint): the entries are routed
        to the fused / remaining halves by the parameter's index in ``Adafactor(self.parameters())`` order."""
        if "fused" in sd:
            self.fused.load_state_dict(sd["fused"])
            if self.rest is not None and sd.get("rest") is not None:
                self.rest.load_state_dict(sd["rest"])
            return
        state = sd.get("state", {})
        allp = [q for g in self.param_groups for q in g["params"]]
        pos = {id(p): i for i, p in enumerate(allp)}
        get = lambda i: state.get(i, state.get(str(i)))  # noqa: E731
        f_idx = {id(p): i for i, p in enumerate(q for g in self.fused.param_groups for q in g["params"])}
        self.fused.load_state_dict({"state": {f_idx[id(p)]: get(pos[id(p)]) for p in allp if id(p) in f_idx and get(pos[id(p)]) is not None}})
        if self.rest is not None:
            rest_params = [q for g in self.rest.param_groups for q in g["params"]]
            rsd = self.rest.state_dict()
            rsd["state"] = {i: get(pos[id(p)]) for i, p in enumerate(rest_params) if get(pos[id(p)]) is not None}
            self.rest.load_state_dict(rsd)

    @torch.no_grad()
    def step(self, closure=None):
        loss = self.fused.step(closure)
        if self.rest is not None and any(p.grad is not None for g in self.rest.param_groups for p in g["params"]):
            self.rest.step()
        return loss
