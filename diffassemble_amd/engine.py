"""Python owner of one ``da_denoiser`` (packed weights) + its workspaces.

This is plumbing around the C ABI (include/diffassemble_hip.h): PyTorch provides device
memory and the stream, every arithmetic op of the denoiser / sampling loop runs in
libdiffassemble_hip.so.  No fallback: tensors must live on a ROCm device.
"""
import ctypes as C
import os

import torch

from . import _lib
from .graph_plan import GraphPlan, build_plan

_PREC = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16}


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Schedule:
    """Device copies of the registered diffusion buffers (spatial_diffusion.py:282-321)."""

    KEYS = ("betas", "alphas_cumprod", "sqrt_recip_alphas", "sqrt_recip_alphas_cumprod",
            "sqrt_recipm1_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_variance")

    def __init__(self, buffers, device):
        self.t = {k: _f32(buffers[k], device) for k in self.KEYS}
        self.steps = int(self.t["betas"].numel())
        self.c = _lib.DaSchedule()
        self.c.steps = self.steps
        for k in self.KEYS:
            setattr(self.c, k, self.t[k].data_ptr())


class DenoiserEngine:
    """Packs an ``Eff_GAT`` / ``Eff_GAT_3d`` state dict (reference key layout, see
    the key list in DESIGN.md section 1) into the HIP library and runs forward / sampling on it."""

    def __init__(self, sd, *, variant="2d", arch="transformer", virt_nodes=0, precision="bf16",
                 device=None):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.DaError("DenoiserEngine needs a ROCm device (no CPU path in diffassemble_amd)")
        self.lib = _lib.lib()
        self.precision = precision
        self.prec = _PREC[precision]
        self.variant, self.arch = variant, arch
        self.virt_nodes = int(virt_nodes) if arch == "exophormer" else 0
        w = _lib.DaWeights()
        keep = []

        def P(name):
            t = _f32(sd[name], self.device)
            keep.append(t)
            return t.data_ptr()

        layers = sorted({int(k.split(".")[2]) for k in sd if k.startswith("gnn_backbone.module_list.")})
        self.n_layers = len(layers)
        D = sd["mlp.0.weight"].shape[1]
        self.D, self.F = D, D - 64
        self.c_in = sd["pos_mlp.0.weight"].shape[1]
        self.steps = sd["time_emb.weight"].shape[0]
        w.variant = _lib.VARIANT_3D if variant == "3d" else _lib.VARIANT_2D
        w.arch = _lib.ARCH_EXOPHORMER if arch == "exophormer" else _lib.ARCH_TRANSFORMER
        w.steps, w.c_in, w.feat_dim = self.steps, self.c_in, self.F
        w.hidden = sd["mlp.0.weight"].shape[0]
        w.heads, w.n_layers, w.virt_nodes = 8, self.n_layers, self.virt_nodes
        w.time_emb = P("time_emb.weight")
        w.pos_w0, w.pos_b0 = P("pos_mlp.0.weight"), P("pos_mlp.0.bias")
        w.pos_w1, w.pos_b1 = P("pos_mlp.2.weight"), P("pos_mlp.2.bias")
        w.mlp_w0, w.mlp_b0 = P("mlp.0.weight"), P("mlp.0.bias")
        w.mlp_w1, w.mlp_b1 = P("mlp.2.weight"), P("mlp.2.bias")
        for l in layers:
            p = f"gnn_backbone.module_list.{l}."
            w.conv_wq[l], w.conv_bq[l] = P(p + "lin_query.weight"), P(p + "lin_query.bias")
            w.conv_wk[l], w.conv_bk[l] = P(p + "lin_key.weight"), P(p + "lin_key.bias")
            w.conv_wv[l], w.conv_bv[l] = P(p + "lin_value.weight"), P(p + "lin_value.bias")
            w.conv_ws[l], w.conv_bs[l] = P(p + "lin_skip.weight"), P(p + "lin_skip.bias")
        if self.virt_nodes > 0:
            w.virt_emb = P("gnn_backbone.virt_node_embedding.weight")
        if variant == "3d":
            self.c_out, self.c_pose = 6, 7
            w.c_out = 6
            w.head_w0, w.head_b0 = P("mlp_t.0.weight"), P("mlp_t.0.bias")
            w.head_w1, w.head_b1 = P("mlp_t.2.weight"), P("mlp_t.2.bias")
            w.head_r_w0, w.head_r_b0 = P("mlp_r.0.weight"), P("mlp_r.0.bias")
            w.head_r_w1, w.head_r_b1 = P("mlp_r.2.weight"), P("mlp_r.2.bias")
        else:
            self.c_out = sd["final_mlp.2.weight"].shape[0]
            self.c_pose = self.c_out
            w.c_out = self.c_out
            w.head_w0, w.head_b0 = P("final_mlp.0.weight"), P("final_mlp.0.bias")
            w.head_w1, w.head_b1 = P("final_mlp.2.weight"), P("final_mlp.2.bias")
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.da_denoiser_create(C.byref(w), self.prec, _lib.stream_ptr(self.device),
                                                   C.byref(handle)))
        self.handle = handle
        self.flags = int(self.lib.da_denoiser_flags(handle))
        self.dense_only = bool(self.flags & 4)     # complete graphs never need the CSR arrays (unless alpha is wanted)
        del keep                      # create() synchronised: the fp32 staging copies may go
        self._ws = {}
        self._loop_bufs = {}
        self._pair_state = None            # two-branch loop: half plans, workspaces, pose buffers
        self._profiling = False

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                self.lib.da_denoiser_destroy(h)
            except Exception:  # noqa: BLE001
                pass
            self.handle = None

    # ------------------------------------------------------------------ graph + workspace
    def plan(self, edge_index, batch):
        return build_plan(edge_index.to(self.device), batch.to(self.device), self.virt_nodes)

    def plan_expander(self, perms, degree):
        """Plan of a Batch of Exphander graphs straight from their permutations (no edge list, no sort)."""
        from .graph_plan import expander_plan
        return expander_plan(perms, degree, self.device, self.virt_nodes)

    def _workspace(self, plan: GraphPlan, need_csr=None):
        if need_csr is None:       # complete / hybrid graphs on an all-MFMA denoiser never walk the edge list
            need_csr = not ((plan.dense or plan.hybrid) and self.dense_only)
        g = plan.c_struct(need_csr)
        need = int(self.lib.da_denoiser_workspace_bytes(self.handle, C.byref(g)))
        key = (plan.n_nodes, plan.n_real)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws = {key: ws}           # one live workspace per engine
        return g, ws

    def set_features(self, plan, feats):
        g, ws = self._workspace(plan)
        feats = _f32(feats, self.device)
        assert feats.shape == (plan.n_real, self.F), (feats.shape, plan.n_real, self.F)
        _lib.check(self.lib.da_denoiser_set_features(self.handle, C.byref(g), _lib.ptr(feats), _lib.ptr(ws),
                                                     ws.numel(), _lib.stream_ptr(self.device)))
        return g, ws

    # ------------------------------------------------------------------ operators
    def forward(self, plan, x, t, feats=None, return_alpha=False, return_pre_head=False, alpha_all_layers=False):
        """Eff_GAT(.._3d).forward_with_feats.  ``feats=None`` reuses the features already staged
        for this plan (sampling loop).  t: int64 [N] tensor or python int."""
        if return_alpha:
            plan.ensure_csr()              # alpha[E, H] is produced by the edge-list kernels
        if feats is not None:
            g, ws = self.set_features(plan, feats)
        else:
            g, ws = self._workspace(plan)
        x = _f32(x, self.device)
        out = torch.empty((plan.n_real, self.c_pose), dtype=torch.float32, device=self.device)
        ashape = (self.n_layers, plan.n_edges, 8) if alpha_all_layers else (plan.n_edges, 8)
        alpha = torch.empty(ashape, dtype=torch.float32, device=self.device) if return_alpha else None
        pre = (torch.empty((plan.n_real, 6), dtype=torch.float32, device=self.device)
               if return_pre_head and self.variant == "3d" else None)
        if torch.is_tensor(t):
            tt, ts = t.to(device=self.device, dtype=torch.int64).contiguous(), 0
        else:
            tt, ts = None, int(t)
        _lib.check(self.lib.da_denoiser_forward(
            self.handle, C.byref(g), _lib.ptr(x), _lib.ptr(tt), ts, _lib.ptr(out), _lib.ptr(alpha),
            int(bool(alpha_all_layers)), _lib.ptr(pre), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self.device)))
        res = [out]
        if return_alpha:
            res.append(alpha)
        if return_pre_head:
            res.append(pre)
        return res[0] if len(res) == 1 else tuple(res)

    def ddim_step(self, sched: Schedule, x, model_out, t, ratio, mean_type, eta=0.0, noise=None):
        x, model_out = _f32(x, self.device), _f32(model_out, self.device)
        out = torch.empty_like(x)
        if torch.is_tensor(t):
            tt, ts = t.to(device=self.device, dtype=torch.int64).contiguous(), 0
            nonneg = int(bool((tt - ratio >= 0).all()))        # the reference's host branch (:560)
        else:
            tt, ts = None, int(t)
            nonneg = int(ts - ratio >= 0)
        nz = None if noise is None else _f32(noise, self.device)
        variant = _lib.VARIANT_3D if self.variant == "3d" else _lib.VARIANT_2D
        _lib.check(self.lib.da_ddim_step(C.byref(sched.c), variant, mean_type, x.shape[0], x.shape[1],
                                         _lib.ptr(x), _lib.ptr(model_out), _lib.ptr(tt), ts, int(ratio), nonneg,
                                         float(eta), _lib.ptr(nz), _lib.ptr(out), _lib.stream_ptr(self.device)))
        return out

    def ddpm_step(self, sched: Schedule, x, model_out, t, noise=None):
        x, model_out = _f32(x, self.device), _f32(model_out, self.device)
        out = torch.empty_like(x)
        if torch.is_tensor(t):
            tt, ts = t.to(device=self.device, dtype=torch.int64).contiguous(), 0
        else:
            tt, ts = None, int(t)
        nz = None if noise is None else _f32(noise, self.device)
        _lib.check(self.lib.da_ddpm_step(C.byref(sched.c), x.shape[0], x.shape[1], _lib.ptr(x),
                                         _lib.ptr(model_out), _lib.ptr(tt), ts, _lib.ptr(nz), _lib.ptr(out),
                                         _lib.stream_ptr(self.device)))
        return out

    def sample_loop(self, plan, sched: Schedule, x_init, feats, *, ratio=1, mean_type=_lib.MEAN_START_X,
                    max_iters=0, keep_trajectory=True, use_graph=True, restage=True, sampler="DDIM", eta=0.0,
                    cfg_w=None, noise=None, generator=None):
        """p_sample_loop: all iterations in one C call (one hipGraph launch when use_graph).
        Returns (traj [n_iters, N, c] or None, x_final [N, c]) -- engine-owned buffers that the next
        call with the same loop shape overwrites (clone to keep).  Note the cached graph also
        borrows ``plan``'s arrays and the workspace.

        ``sampler="DDPM"``, ``eta > 0`` and ``cfg_w`` (classifier-free guidance weight; ``None`` = off) select the
        reference's other samplers (spatial_diffusion.py:485-510, 568-589, 620-627) INSIDE the same captured loop
        (da_sample_loop_ex).  The stochastic ones read one standard-normal draw per iteration from an engine-owned
        [n_iters, N, c] buffer that is refilled before every launch: from ``noise`` when given (tests inject the
        reference's saved draws), else by one ``torch.randn`` call (``generator`` optional)."""
        total = (sched.steps + ratio - 1) // ratio
        n_iters = min(max_iters, total) if max_iters and max_iters > 0 else total
        ex = sampler == "DDPM" or eta > 0 or cfg_w is not None
        if self._two_branch(plan, keep_trajectory, use_graph):
            opts = None
            if ex:                                  # the other samplers of the reference on both branches (da_sample_loop_pair_ex)
                opts = dict(sampler=1 if sampler == "DDPM" else 0, eta=float(eta), cfg_w=cfg_w, noise=noise, generator=generator,
                            stochastic=sampler == "DDPM" or eta > 0)
            return self._sample_loop_pair(plan, sched, x_init, feats, ratio, mean_type, n_iters, restage, keep_trajectory, opts)
        if restage:
            g, ws = self.set_features(plan, feats)
        else:
            g, ws = self._workspace(plan)
        c = x_init.shape[1]
        # persistent I/O buffers: the cached hipGraph is keyed on these pointers, so replays of the
        # same loop shape reuse them (results are overwritten by the next call of that shape)
        key = (plan.n_real, c, n_iters, bool(keep_trajectory))
        bufs = self._loop_bufs.get(key)
        if bufs is None:
            xi = torch.empty((plan.n_real, c), dtype=torch.float32, device=self.device)
            traj = (torch.empty((n_iters, plan.n_real, c), dtype=torch.float32, device=self.device)
                    if keep_trajectory else None)
            bufs = self._loop_bufs[key] = (xi, traj, torch.empty_like(xi))
        xi, traj, x_final = bufs
        xi.copy_(x_init)
        self._loop_keep = plan                                 # the cached hipGraph borrows its arrays
        if ex:
            o = _lib.DaLoopOpts()
            o.sampler = 1 if sampler == "DDPM" else 0
            o.eta = float(eta)
            o.cfg, o.cfg_w = (0, 0.0) if cfg_w is None else (1, float(cfg_w))
            if sampler == "DDPM" or eta > 0:
                nb = self._loop_bufs.get(("noise",) + key)
                if nb is None:
                    nb = self._loop_bufs[("noise",) + key] = torch.empty((n_iters, plan.n_real, c), dtype=torch.float32, device=self.device)
                if noise is not None:
                    nb.copy_(noise)
                else:
                    nb.normal_(generator=generator)
                o.noise = nb.data_ptr()
            _lib.check(self.lib.da_sample_loop_ex(
                self.handle, C.byref(g), C.byref(sched.c), int(mean_type), int(ratio), int(n_iters),
                _lib.ptr(xi), _lib.ptr(traj), _lib.ptr(x_final), _lib.ptr(ws), ws.numel(),
                int(bool(use_graph)), C.byref(o), _lib.stream_ptr(self.device)))
            return traj, x_final
        _lib.check(self.lib.da_sample_loop(
            self.handle, C.byref(g), C.byref(sched.c), int(mean_type), int(ratio), int(n_iters),
            _lib.ptr(xi), _lib.ptr(traj), _lib.ptr(x_final), _lib.ptr(ws), ws.numel(),
            int(bool(use_graph)), _lib.stream_ptr(self.device)))
        return traj, x_final


    # ------------------------------------------------------------------ two-branch loop
    two_branch_min_nodes = 40000       # DA_TWO_BRANCH=auto: the pair loop from this many pieces per Batch
    two_branch_min_graphs = 64         # DA_TWO_BRANCH=1: ... from this many puzzles
    pair_split_at = 0                  # puzzles in the first half Batch (0 = half)

    def _two_branch(self, plan, keep_trajectory, use_graph):
        """Large Batches of complete graphs run as TWO half Batches on two parallel branches of one hipGraph
        (da_sample_loop_pair), so that one half's projections and tail kernels overlap the other half's attention.
        Bit-identical poses.  Default ("auto") from 40 000 nodes up (``two_branch_min_nodes``): measured at the end of round 3,
        A/B on one box, 900-piece puzzles: 48 per GPU 81.0 k -> 83.8 k puzzle-steps/s, 64 per GPU 85.7 k -> 88.6 k (+3.4 %, three
        rounds), 128: 88.8 k; but 32: 77.7 k -> 75.8 k and 16: 64.0 k -> 60.9 k (half Batches that no longer fill the chip);
        BASELINE config 2 (512 puzzles of 144 pieces) 738 k -> 790 k.  (In round 2 the same switch measured +0.4 %: the step sits
        at the package power cap, and what two interleaved kernel streams buy depends on the kernels.)  DA_TWO_BRANCH=0 turns it
        off, =1 forces it from ``two_branch_min_graphs`` (64) graphs up; ``two_branch_min_nodes`` / ``two_branch_min_graphs`` /
        ``pair_split_at`` are attributes of the engine (class defaults below).  Loops that keep their trajectory (the module's p_sample_loop) take it
        too: each half writes its row range of every iteration (da_sample_loop_pair_traj)."""
        if not use_graph or self._profiling or not self.dense_only:
            return False
        if not plan.dense or plan.hybrid or plan.n_nodes != plan.n_real or plan.n_graphs < 2:
            return False
        mode = os.environ.get("DA_TWO_BRANCH", "auto")
        if mode == "0":
            return False
        if mode == "1":
            return plan.n_graphs >= int(self.two_branch_min_graphs)
        return plan.n_real >= int(self.two_branch_min_nodes)

    def _sample_loop_pair(self, plan, sched, x_init, feats, ratio, mean_type, n_iters, restage, keep_trajectory=False, opts=None):
        from .graph_plan import split_complete
        # pair_split_at (attribute, 0 = half): puzzles in the first branch -- unequal branches drift out of lockstep (measured: loses)
        g_split = int(self.pair_split_at) or plan.n_graphs // 2
        pa, pb, n0 = split_complete(plan, min(max(g_split, 1), plan.n_graphs - 1))
        c = x_init.shape[1]
        st = self._pair_state
        # keyed on the Batch's SHAPE (graph sizes), not on the plan object: the module re-plans every Batch (it releases the
        # edge list after each loop), and a same-shaped Batch must find its half plans, workspaces, pose buffers and --
        # through their addresses -- the cached two-branch hipGraph again
        key = (plan._shape_sig, c)
        if st is None or st["key"] != key:
            need = [int(self.lib.da_denoiser_workspace_bytes(self.handle, C.byref(q.c_struct(False)))) for q in (pa, pb)]
            st = self._pair_state = {
                "key": key, "halves": (pa, pb), "g": (pa.c_struct(False), pb.c_struct(False)),
                "ws": tuple(torch.empty(nb, dtype=torch.uint8, device=self.device) for nb in need),
                "xi": torch.empty((plan.n_real, c), dtype=torch.float32, device=self.device),
                "xf": torch.empty((plan.n_real, c), dtype=torch.float32, device=self.device), "staged": None}
        pa, pb = st["halves"]                      # (the cached graph reads THESE plans' device arrays)
        fkey = None if feats is None else (feats.data_ptr(), feats._version, tuple(feats.shape))
        if restage or st["staged"] is None or (fkey is not None and st["staged"] != fkey):
            f = _f32(feats, self.device)
            assert f.shape == (plan.n_real, self.F), (f.shape, plan.n_real, self.F)
            for q, g, ws, sl in ((pa, st["g"][0], st["ws"][0], slice(0, n0)), (pb, st["g"][1], st["ws"][1], slice(n0, None))):
                fh = f[sl]
                _lib.check(self.lib.da_denoiser_set_features(self.handle, C.byref(g), _lib.ptr(fh), _lib.ptr(ws), ws.numel(),
                                                             _lib.stream_ptr(self.device)))
            st["staged"] = fkey
        xi, xf = st["xi"], st["xf"]
        xi.copy_(x_init)
        (ga, gb), (wa, wb) = st["g"], st["ws"]
        traj = None
        if keep_trajectory:
            # the trajectory p_sample_loop returns: one [n_iters, N, c] buffer, each half writes its own row range of every iteration
            traj = st.get("traj")
            if traj is None or traj.shape[0] != n_iters:
                traj = st["traj"] = torch.empty((n_iters, plan.n_real, c), dtype=torch.float32, device=self.device)
        o, nz_b = None, None
        if opts is not None:
            o = _lib.DaLoopOpts()
            o.sampler, o.eta = opts["sampler"], opts["eta"]
            o.cfg, o.cfg_w = (0, 0.0) if opts["cfg_w"] is None else (1, float(opts["cfg_w"]))
            if opts["stochastic"]:
                # ONE [n_iters, N, c] draw for the whole Batch (what the one-branch loop reads), the halves take their row ranges
                nb = st.get("noise")
                if nb is None or nb.shape[0] != n_iters:
                    nb = st["noise"] = torch.empty((n_iters, plan.n_real, c), dtype=torch.float32, device=self.device)
                if opts["noise"] is not None:
                    nb.copy_(opts["noise"])
                else:
                    nb.normal_(generator=opts["generator"])
                o.noise = nb.data_ptr()
                nz_b = _lib.ptr(nb[0, n0:])
        _lib.check(self.lib.da_sample_loop_pair_ex(
            self.handle, C.byref(sched.c), int(mean_type), int(ratio), int(n_iters),
            C.byref(ga), _lib.ptr(xi), _lib.ptr(xf), _lib.ptr(wa), wa.numel(),
            C.byref(gb), _lib.ptr(xi[n0:]), _lib.ptr(xf[n0:]), _lib.ptr(wb), wb.numel(),
            _lib.ptr(traj), None if traj is None else _lib.ptr(traj[0, n0:]), plan.n_real * c,
            None if o is None else C.byref(o), nz_b, plan.n_real * c, _lib.stream_ptr(self.device)))
        return traj, xf

    def sample_loop_batches(self, plans, sched: Schedule, x_inits, feats, *, ratio=1, mean_type=_lib.MEAN_START_X, max_iters=0, restage=True):
        """SEVERAL independent Batches in flight: their DDIM loops as hipGraphs on separate streams, each Batch exactly the computation
        ``sample_loop`` runs for it alone -- bit-identical poses.  This is how Batches that cannot be SPLIT get the two-stream overlap: an
        exophormer Batch's virtual-node edges couple its puzzles (the quirk of exophormer_gnn.py:183-200 sends every real node to the first
        graphs' virtual nodes), so half of it is a different computation, but whole Batches are independent -- and how SMALL Batches (the
        reference's scripted 8-puzzle ones leave most of the chip idle in every kernel) fill the chip.  ``plans``, ``x_inits``, ``feats``: lists
        of equal length N >= 2.  N = 2 is one library call (da_sample_loop_pair: two graphs between a fork and a join event); N > 2 launches
        each Batch's own loop graph (da_sample_loop) on a stream of the engine's between fork / join waits on the caller's stream.  Returns the
        list of x_final tensors, engine-owned (the next call of the same shapes overwrites them)."""
        n_b = len(plans)
        assert n_b >= 2 and len(x_inits) == n_b and len(feats) == n_b
        if not _lib.config().pair_split or self._profiling:
            return [self.sample_loop(p, sched, x, f, ratio=ratio, mean_type=mean_type, max_iters=max_iters, keep_trajectory=False,
                                     restage=restage)[1].clone() for p, x, f in zip(plans, x_inits, feats)]
        total = (sched.steps + ratio - 1) // ratio
        n_iters = min(max_iters, total) if max_iters and max_iters > 0 else total
        c = x_inits[0].shape[1]
        st = getattr(self, "_batches_state", None)
        key = tuple(id(q) for q in plans) + (c,)
        if st is None or st["key"] != key:
            need_csr = [not ((q.dense or q.hybrid) and self.dense_only) for q in plans]
            gs = tuple(q.c_struct(nc) for q, nc in zip(plans, need_csr))
            need = [int(self.lib.da_denoiser_workspace_bytes(self.handle, C.byref(g))) for g in gs]
            st = self._batches_state = {
                "key": key, "plans": tuple(plans), "g": gs,
                "ws": tuple(torch.empty(nb, dtype=torch.uint8, device=self.device) for nb in need),
                "xi": tuple(torch.empty((q.n_real, c), dtype=torch.float32, device=self.device) for q in plans),
                "xf": tuple(torch.empty((q.n_real, c), dtype=torch.float32, device=self.device) for q in plans), "staged": None,
                "streams": [torch.cuda.Stream(device=self.device) for _ in range(n_b)] if n_b > 2 else None}
        fkey = tuple((f.data_ptr(), f._version, tuple(f.shape)) for f in feats)
        if restage or st["staged"] != fkey:
            for q, g, ws, f in zip(plans, st["g"], st["ws"], feats):
                f = _f32(f, self.device)
                assert f.shape == (q.n_real, self.F), (f.shape, q.n_real, self.F)
                _lib.check(self.lib.da_denoiser_set_features(self.handle, C.byref(g), _lib.ptr(f), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self.device)))
            st["staged"] = fkey
        for xi, x in zip(st["xi"], x_inits):
            xi.copy_(x)
        if n_b == 2:
            (ga, gb), (wa, wb), (xa, xb), (fa, fb) = st["g"], st["ws"], st["xi"], st["xf"]
            _lib.check(self.lib.da_sample_loop_pair(
                self.handle, C.byref(sched.c), int(mean_type), int(ratio), int(n_iters),
                C.byref(ga), _lib.ptr(xa), _lib.ptr(fa), _lib.ptr(wa), wa.numel(),
                C.byref(gb), _lib.ptr(xb), _lib.ptr(fb), _lib.ptr(wb), wb.numel(), _lib.stream_ptr(self.device)))
            return [fa, fb]
        cur = torch.cuda.current_stream(self.device)
        for g, ws, xi, xf, sx in zip(st["g"], st["ws"], st["xi"], st["xf"], st["streams"]):
            sx.wait_stream(cur)
            _lib.check(self.lib.da_sample_loop(self.handle, C.byref(g), C.byref(sched.c), int(mean_type), int(ratio), int(n_iters),
                                               _lib.ptr(xi), None, _lib.ptr(xf), _lib.ptr(ws), ws.numel(), 1, C.c_void_p(sx.cuda_stream)))
        for sx in st["streams"]:
            cur.wait_stream(sx)
        return list(st["xf"])

    # ------------------------------------------------------------------ measurement
    def profile(self, on=True):
        _lib.check(self.lib.da_profile_enable(self.handle, int(bool(on))))
        self._profiling = bool(on)

    def profile_read(self):
        """-> {class: (total_ms, launches)} measured with HIP events on the launch stream."""
        n = len(_lib.PROF_CLASSES)
        ms, cnt = (C.c_float * n)(), (C.c_int32 * n)()
        _lib.check(self.lib.da_profile_read(self.handle, ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(_lib.PROF_CLASSES)}


# ---------------------------------------------------------------------- kernel-level helpers
def linear(x, weight, bias=None, act=_lib.ACT_NONE, residual=None, precision="fp32"):
    """out = act(x @ weight^T + bias) + residual through da_linear (x, weight: [M,K], [N,K])."""
    prec = _PREC[precision]
    dt = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32
    x = x.to(dt).contiguous()
    weight = weight.to(dt).contiguous()
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=dt, device=x.device)
    b = None if bias is None else bias.float().contiguous()
    r = None if residual is None else residual.to(dt).contiguous()
    _lib.check(_lib.lib().da_linear(prec, M, K, N, _lib.ptr(x), K, _lib.ptr(weight), _lib.ptr(b), int(act),
                                    _lib.ptr(r), _lib.ptr(out), N, _lib.stream_ptr(x.device)))
    return out


class PackedLinear:
    """A constant bf16 weight [N, K] packed once for the row-panel kernel (da_linear_pack / da_linear_packed,
    da_gemm_xpanel.hip); shapes without a packed form, and inputs too short for it, run da_linear's kernels."""

    def __init__(self, weight, bias=None):
        self.weight = weight.to(torch.bfloat16).contiguous()
        self.bias = None if bias is None else bias.float().contiguous()
        self.N, self.K = self.weight.shape
        nb = _lib.lib().da_linear_packed_bytes(_lib.PREC_BF16, self.K, self.N)
        self.packed = None
        if nb:
            self.packed = torch.empty(nb, dtype=torch.uint8, device=self.weight.device)
            _lib.check(_lib.lib().da_linear_pack(_lib.PREC_BF16, self.K, self.N, _lib.ptr(self.weight), self.K, _lib.ptr(self.packed),
                                                 _lib.stream_ptr(self.weight.device)))

    def __call__(self, x, act=_lib.ACT_NONE):
        x = x.to(torch.bfloat16).contiguous()
        out = torch.empty((x.shape[0], self.N), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().da_linear_packed(_lib.PREC_BF16, x.shape[0], self.K, self.N, _lib.ptr(x), self.K, _lib.ptr(self.weight),
                                               _lib.ptr(self.packed), _lib.ptr(self.bias), int(act), None, _lib.ptr(out), self.N,
                                               _lib.stream_ptr(x.device)))
        return out


def attn_csr(plan: GraphPlan, qkvs, heads, C_head, residual=None, act=_lib.ACT_NONE, return_alpha=False,
             precision="fp32"):
    prec = _PREC[precision]
    dt = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32
    qkvs = qkvs.to(dt).contiguous()
    out = torch.empty((plan.n_nodes, heads * C_head), dtype=dt, device=qkvs.device)
    alpha = torch.empty((plan.n_edges, heads), dtype=torch.float32, device=qkvs.device) if return_alpha else None
    r = None if residual is None else residual.to(dt).contiguous()
    g = plan.c_struct()
    _lib.check(_lib.lib().da_attn_csr(prec, C.byref(g), heads, C_head, _lib.ptr(qkvs), _lib.ptr(r), int(act),
                                      _lib.ptr(out), _lib.ptr(alpha), _lib.stream_ptr(qkvs.device)))
    return (out, alpha) if return_alpha else out


def conv_dense(plan: GraphPlan, x, weight, bias, heads, C_head, residual=None, act=_lib.ACT_NONE, precision="fp32"):
    """One TransformerConv layer on COMPLETE graphs through the dense MFMA path (da_conv_dense):
    x [N, Din], weight [4*H*C, Din] (rows Q|K|V|skip), bias [4*H*C]."""
    prec = _PREC[precision]
    dt = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32
    x = x.to(dt).contiguous()
    weight = weight.to(dt).contiguous()
    bias = bias.float().contiguous()
    g = plan.c_struct()
    lib = _lib.lib()
    nbytes = int(lib.da_attn_dense_scratch_bytes(prec, C.byref(g), heads, C_head))
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty((plan.n_nodes, heads * C_head), dtype=dt, device=x.device)
    r = None if residual is None else residual.to(dt).contiguous()
    _lib.check(lib.da_conv_dense(prec, C.byref(g), heads, C_head, x.shape[1], _lib.ptr(x), _lib.ptr(weight),
                                 _lib.ptr(bias), _lib.ptr(r), int(act), _lib.ptr(out), _lib.ptr(scratch),
                                 _lib.stream_ptr(x.device)))
    return out


def conv_dense_ex(plan: GraphPlan, x, weight, bias, heads, C_head, residual=None, act=_lib.ACT_NONE, precision="fp32",
                  prescale_q=True, folded=False):
    """da_conv_dense_ex: the layer in the forms the packed denoiser runs it.  ``weight`` / ``bias`` arrive UNSCALED in fp32
    (rows Q | K | V | skip, or Q | K | V' [H * 32] when ``folded``); with ``prescale_q=True`` the Q rows are multiplied
    by log2(e) / sqrt(C) here, in fp32, before the rounding to the activation dtype -- what da_denoiser_create does
    (da_api.hip ConvW::wd) -- and the kernels take their shift-free softmax paths (``"done"``: the caller's Q rows
    already carry the factor; ``False``: plain da_conv_dense semantics).  Returns [N, H*C] or, folded,
    [H, n_real, 32] (per-head normalised outputs)."""
    import math
    prec = _PREC[precision]
    dt = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32
    HC = heads * C_head
    weight, bias = weight.float().clone(), bias.float().clone()
    if prescale_q is True:                          # "done": the caller already scaled the Q rows
        sc = math.log2(math.e) / math.sqrt(C_head)
        weight[:HC] *= sc
        bias[:HC] *= sc
    x = x.to(dt).contiguous()
    weight = weight.to(dt).contiguous()
    bias = bias.contiguous()
    g = plan.c_struct(need_csr=False)
    lib = _lib.lib()
    nbytes = int(lib.da_attn_dense_scratch_bytes(prec, C.byref(g), heads, C_head))
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
    out = (torch.empty((heads, plan.n_real, 32), dtype=dt, device=x.device) if folded
           else torch.empty((plan.n_nodes, HC), dtype=dt, device=x.device))
    r = None if residual is None else residual.to(dt).contiguous()
    flags = (_lib.CONV_Q_PRESCALED if prescale_q else 0) | (_lib.CONV_FOLDED_V32 if folded else 0)
    with torch.cuda.device(x.device):
        _lib.check(lib.da_conv_dense_ex(prec, C.byref(g), heads, C_head, x.shape[1], _lib.ptr(x), _lib.ptr(weight),
                                        _lib.ptr(bias), _lib.ptr(r), int(act), _lib.ptr(out), _lib.ptr(scratch), flags,
                                        _lib.stream_ptr(x.device)))
    return out


def debug_counters(reset=True):
    """da_debug_counters as a dict (fallback events of the shift-free softmax kernels since the last reset)."""
    buf = (C.c_int64 * 8)()
    _lib.check(_lib.lib().da_debug_counters(buf, 8, 1 if reset else 0))
    return {name: int(buf[k]) for k, name in enumerate(_lib.DBG_COUNTERS)}


def launch_counters(reset=True):
    """Which kernels took the layers since the last reset (da_debug_counters [DA_DBG_RES_LAUNCHES], [DA_DBG_VIRT_IN_LAUNCH]): launches of the
    K / V-resident hidden-layer kernel, masked hidden-layer launches that carried the exophormer's virtual rows.  NOTE: resets every counter."""
    buf = (C.c_int64 * 8)()
    _lib.check(_lib.lib().da_debug_counters(buf, 8, 1 if reset else 0))
    return {name: int(buf[k]) for name, k in _lib.DBG_LAUNCH_COUNTERS.items()}


def resident_attention_launches(reset=True):
    """Launches of the K / V-resident hidden-layer attention kernel (k_attn_res, da_attn_opt.hip) since the last reset
    (da_debug_counters [DA_DBG_RES_LAUNCHES]): lets a test assert which kernel took a layer.  NOTE: resets every counter."""
    buf = (C.c_int64 * 8)()
    _lib.check(_lib.lib().da_debug_counters(buf, 8, 1 if reset else 0))
    return int(buf[4])


def greedy_assign(pos1, pos2, ptr1=None, ptr2=None):
    """greedy_cost_assignment (spatial_diffusion.py:179-216) for one puzzle or a whole Batch in ONE launch
    (da_greedy_assign): pos1 [N, >=2], pos2 [M, >=2] fp32 on a ROCm device; ptr1 / ptr2 int32 [G + 1] row
    offsets of the puzzles (default: one puzzle).  Returns int64 [N', 3] rows (row, column, int(distance)) in
    assignment order, indices local to each puzzle, puzzle g at rows ptr1[g] .. ptr1[g] + min(n_g, m_g)."""
    lib = _lib.lib()
    dev = pos1.device
    if dev.type != "cuda":
        raise _lib.DaError("greedy_assign needs ROCm tensors (no CPU path in diffassemble_amd)")
    pos1 = pos1.detach().to(torch.float32).contiguous()
    pos2 = pos2.detach().to(device=dev, dtype=torch.float32).contiguous()
    if ptr1 is None:
        ptr1 = torch.tensor([0, pos1.shape[0]], dtype=torch.int32, device=dev)
        ptr2 = torch.tensor([0, pos2.shape[0]], dtype=torch.int32, device=dev)
    ptr1, ptr2 = ptr1.to(device=dev, dtype=torch.int32).contiguous(), ptr2.to(device=dev, dtype=torch.int32).contiguous()
    G = ptr1.numel() - 1
    n = (ptr1[1:] - ptr1[:-1])
    m = (ptr2[1:] - ptr2[:-1])
    out = torch.zeros((pos1.shape[0], 3), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.da_greedy_assign(G, _lib.ptr(pos1), pos1.shape[1], _lib.ptr(pos2), pos2.shape[1], _lib.ptr(ptr1),
                                        _lib.ptr(ptr2), int(n.max()), int(m.max()), _lib.ptr(out), _lib.stream_ptr(dev)))
    return out
