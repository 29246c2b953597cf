"""Host-side plan of a PyG ``Batch``'s piece graph for the HIP kernels.

Turns the reference's operator inputs -- ``edge_index[2, E]`` int64 (row 0 = source j,
row 1 = target i, per-graph offsets already applied by PyG collation) and the sorted
``batch[N]`` vector (efficient_gat.py:121-129) -- into the ``da_graph`` of
include/diffassemble_hip.h: int32 CSR-by-destination with multi-edges kept, the slot->edge
permutation for the ``alpha[E, H]`` output, per-graph node offsets, and a ``dense`` flag
when every graph is complete (so the block-diagonal MFMA attention kernel applies).

For ``architecture='exophormer'`` the V*G virtual rows and the extra edges are appended
exactly as backbones/exophormer_gnn.py:164-200 builds them, including its pairing quirk
(``src = cat[arange(N), virt_edges]``, ``dst = cat[virt_edges, arange(N)]`` paired
position by position), but vectorised: no per-graph Python loop, no ``batch.unique()``.

Index plumbing only (torch sort/bincount on whatever device the inputs live on); the plan
is built once per Batch and reused for all T sampling steps.
"""
from dataclasses import dataclass

import torch

from . import _lib


def exophormer_edge_index(edge_index, batch, virt_nodes, n_graphs=None):
    """Extended edge_index of exophormer_gnn.py:183-200 (see module docstring)."""
    N = batch.numel()
    V = int(virt_nodes)
    dev = batch.device
    G = int(batch.max()) + 1 if n_graphs is None else int(n_graphs)
    counts = torch.bincount(batch, minlength=G) + V                 # nodes of graph i in the EXTENDED batch
    seg = counts * V                                                # len(virt_edge_i) = V * (n_i + V)
    gid = torch.repeat_interleave(torch.arange(G, device=dev), seg)
    start = torch.cumsum(seg, 0) - seg
    k = torch.arange(int(seg.sum()), device=dev) - start[gid]
    virt_edges = N + gid * V + k % V
    ar = torch.arange(N, device=dev)
    src = torch.cat([ar, virt_edges])
    dst = torch.cat([virt_edges, ar])
    return torch.hstack((edge_index, torch.stack((src, dst))))


@dataclass
class GraphPlan:
    n_nodes: int
    n_real: int
    n_graphs: int
    dense: int
    n_edges: int
    max_graph_nodes: int
    row_ptr: torch.Tensor      # int32 [n_nodes + 1]   (None until ensure_csr() for complete graphs: the dense
    col_src: torch.Tensor      # int32 [E]              kernels never walk the edge list, so the sort is skipped
    edge_id: torch.Tensor      # int32 [E]              unless somebody needs it)
    graph_ptr: torch.Tensor    # int32 [G + 1]
    n_pad: int                 # dense mode: rows of the head-major Q / K / V buffers
    pad_ptr: torch.Tensor      # int32 [G + 1] (64-aligned slot of each graph)
    row_map: torch.Tensor      # int32 [n_nodes] node -> padded row
    _edge_index: torch.Tensor   # int64 [2, E] (extended for exophormer) -- returned with alpha; None until asked for
                                # when the plan was built from expander permutations (see ``edge_index``)
    out_ptr: torch.Tensor = None   # int32 [n_nodes + 1] CSR by SOURCE (training backward), lazily built
    out_dst: torch.Tensor = None   # int32 [E]
    # hybrid mode (sparse-but-heavy graphs, e.g. Exphander + exophormer virtual nodes): the unique
    # real->real in-graph edges as one adjacency bitmask per graph for the masked MFMA attention, every
    # other edge (virtual nodes, duplicates, cross-graph pairs) as a small CSR that is merged afterwards
    hybrid: int = 0
    mask: torch.Tensor = None      # uint8, rows of graph g start at mask_ptr[g], row stride n_pad_g / 8 bytes
    mask_ptr: torch.Tensor = None  # int64 [G + 1] byte offsets
    irr_row_ptr: torch.Tensor = None   # int32 [n_nodes + 1]
    irr_col_src: torch.Tensor = None   # int32 [E_irregular]
    edge_index_fn: object = None       # () -> the edge list, for plans built without one (expander_plan)
    # banded layout of expander plans (optional): slots of a graph ordered by the nodes' positions in the generator's permutation
    slot_node: torch.Tensor = None     # int32 [n_pad]: padded slot -> node (-1 = padding); None = slot order is node order
    blk_class: torch.Tensor = None     # uint8: per (32-slot query slab, 32-slot key block) 0 = no regular edge, 1 = some, 2 = all
    blk_class_ptr: torch.Tensor = None  # int64 [G + 1] byte offsets of the graphs' class tables (graphs of one shape share one)
    blk_class_stride: int = 0
    rm_meta: torch.Tensor = None       # int32 [n_pad, 4]: per slot (remainder begin, end, slot of the first remainder source, node)
    agg: tuple = None                  # (row_ptr int32 [n_nodes + 1], col_src int32, mult float32): virtual rows' remainder edges, duplicates merged

    @property
    def edge_index(self):
        if self._edge_index is None and self.edge_index_fn is not None:
            self._edge_index = self.edge_index_fn()
        return self._edge_index

    def ensure_csr(self):
        """CSR by destination of ``edge_index`` (stable: keeps the caller's order inside a segment)."""
        if self.row_ptr is None:
            self.row_ptr, self.col_src, self.edge_id = _csr_by_destination(self.edge_index, self.n_nodes)
        return self

    def with_source_csr(self, edge_list_only=None):
        """Add the by-source orientation of the same edge list (da_graph.out_ptr / out_dst), which
        the attention backward walks to form dK / dV without atomics.  ``edge_list_only``: the library walks the FULL edge list
        (da_config.train_attn = 0, or TrainEngine's fp32 route for small / sparse hybrid plans)."""
        if edge_list_only is None:
            from . import _lib
            edge_list_only = _lib.config().train_attn == 0          # da_config.train_attn (DA_TRAIN_ATTN=0)
        kind = "full" if (edge_list_only or not self.hybrid) else "remainder"
        if self.out_ptr is not None and getattr(self, "_out_kind", kind) != kind:
            self.out_ptr = self.out_dst = None          # built for the other route
        self._out_kind = kind
        if self.dense and not edge_list_only:
            return self                        # complete graphs train on the grouped GEMMs: no edge list is walked
        if self.out_ptr is None and self.hybrid and not edge_list_only:
            # hybrid graphs: only the REMAINDER edges are walked (the regular ones run as adjacency-masked grouped GEMMs,
            # da_train_dense.hip), so the by-source orientation is built from irr_row_ptr / irr_col_src alone -- the
            # 15 M regular edges of a Batch of 60 % Exphander graphs are never sorted
            dev = self.irr_col_src.device
            cnt = (self.irr_row_ptr[1:] - self.irr_row_ptr[:-1]).to(torch.int64)
            dst = torch.repeat_interleave(torch.arange(self.n_nodes, device=dev), cnt)
            src = self.irr_col_src.to(torch.int64)
            perm = torch.argsort(src, stable=True)
            ptr = torch.zeros(self.n_nodes + 1, dtype=torch.int64, device=dev)
            ptr[1:] = torch.cumsum(torch.bincount(src, minlength=self.n_nodes), 0)
            self.out_ptr = ptr.to(torch.int32)
            self.out_dst = dst[perm].to(torch.int32).contiguous()
        if self.out_ptr is None:
            src, dst = self.edge_index[0], self.edge_index[1]
            perm = torch.argsort(src, stable=True)
            ptr = torch.zeros(self.n_nodes + 1, dtype=torch.int64, device=src.device)
            ptr[1:] = torch.cumsum(torch.bincount(src, minlength=self.n_nodes), 0)
            self.out_ptr = ptr.to(torch.int32)
            self.out_dst = dst[perm].to(torch.int32).contiguous()
        return self

    def c_struct(self, need_csr=True, inference_hints=True):
        """``need_csr=False``: leave the CSR arrays out if they have not been built (only valid for complete
        graphs on a denoiser that reports da_denoiser_flags bit 2, without alpha).  ``inference_hints=False`` (the training
        path, which never reads them): do not build rm_meta / agg_* -- a torch.unique and host syncs per new Batch."""
        g = _lib.DaGraph()
        g.n_nodes, g.n_real, g.n_graphs, g.dense = self.n_nodes, self.n_real, self.n_graphs, self.dense
        g.n_edges = self.n_edges
        if need_csr:
            self.ensure_csr()
        if self.row_ptr is not None:
            g.row_ptr = self.row_ptr.data_ptr()
            g.col_src = self.col_src.data_ptr()
            g.edge_id = self.edge_id.data_ptr()
        g.graph_ptr = self.graph_ptr.data_ptr()
        g.max_graph_nodes = self.max_graph_nodes
        g.n_pad = self.n_pad
        g.pad_ptr = self.pad_ptr.data_ptr()
        g.row_map = self.row_map.data_ptr()
        g.out_ptr = self.out_ptr.data_ptr() if self.out_ptr is not None else None
        g.out_dst = self.out_dst.data_ptr() if self.out_dst is not None else None
        g.hybrid = self.hybrid
        if self.hybrid:
            g.mask, g.mask_ptr = self.mask.data_ptr(), self.mask_ptr.data_ptr()
            g.irr_row_ptr, g.irr_col_src = self.irr_row_ptr.data_ptr(), self.irr_col_src.data_ptr()
            if self.slot_node is not None:
                g.slot_node = self.slot_node.data_ptr()
            if self.rm_meta is None and inference_hints:
                self.rm_meta = _remainder_meta(self)
            if self.rm_meta is not None:
                g.rm_meta = self.rm_meta.data_ptr()
            if self.agg is None and self.n_nodes > self.n_real and inference_hints:
                self.agg = _aggregate_virtual_rows(self.irr_row_ptr, self.irr_col_src, self.n_real, self.n_nodes)
            if self.agg is not None:
                g.agg_row_ptr, g.agg_col_src, g.agg_mult = (t.data_ptr() for t in self.agg)
            if self.blk_class is not None:
                g.blk_class, g.blk_class_ptr, g.blk_class_stride = self.blk_class.data_ptr(), self.blk_class_ptr.data_ptr(), self.blk_class_stride
        return g


def _aggregate_virtual_rows(irr_row_ptr, irr_col_src, n_real, n_nodes):
    """da_graph.agg_*: the remainder edges of the rows >= n_real (exophormer virtual nodes) with duplicated (source, target)
    pairs merged into one entry + multiplicity (one sort of the virtual rows' edges only; shape-only for expander plans)."""
    dev = irr_col_src.device
    rp = irr_row_ptr.to(torch.int64)
    lo, hi = int(rp[n_real]), int(rp[n_nodes])
    cnt = rp[n_real + 1:] - rp[n_real:-1]
    dst = torch.repeat_interleave(torch.arange(n_real, n_nodes, device=dev), cnt)
    key = dst * n_nodes + irr_col_src[lo:hi].to(torch.int64)
    uk, mult = torch.unique(key, return_counts=True)                     # sorted: grouped by destination, sources ascending
    ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(uk // n_nodes, minlength=n_nodes), 0)
    return ptr.to(torch.int32).contiguous(), (uk % n_nodes).to(torch.int32).contiguous(), mult.to(torch.float32).contiguous()


def _remainder_meta(plan: GraphPlan):
    """da_graph.rm_meta of a hybrid plan: int32 [n_pad, 4], one record per padded slot -- (begin, end) of the node's remainder
    edges in irr_col_src, the slot of the first remainder source (any valid slot when there is none) and the node itself
    (-1 for padding and virtual slots, whose rows the masked attention does not own).  Index arithmetic only."""
    dev = plan.row_map.device
    n_pad = int(plan.n_pad)
    if plan.slot_node is not None:
        sn = plan.slot_node.to(torch.int64)
    else:
        sn = torch.full((n_pad,), -1, dtype=torch.int64, device=dev)
        sn[plan.row_map.to(torch.int64)] = torch.arange(plan.n_nodes, device=dev)
    real = (sn >= 0) & (sn < plan.n_real)
    nd = torch.where(real, sn, torch.zeros_like(sn))
    rp = plan.irr_row_ptr.to(torch.int64)
    beg, end = rp[nd], rp[nd + 1]
    has = real & (end > beg)
    n_irr = int(plan.irr_col_src.numel())
    if n_irr > 0:
        src = plan.irr_col_src.to(torch.int64)[torch.where(has, beg, torch.zeros_like(beg)).clamp(max=n_irr - 1)]
        first = plan.row_map.to(torch.int64)[src]
    else:
        first = torch.zeros_like(sn)
    slot = torch.arange(n_pad, device=dev)
    first = torch.where(has, first, torch.where(real, slot, torch.zeros_like(slot)))      # own slot: always a valid row
    z = torch.zeros_like(beg)
    meta = torch.stack([torch.where(real, beg, z), torch.where(real, end, z), first, torch.where(real, sn, torch.full_like(sn, -1))], 1)
    return meta.to(torch.int32).contiguous()


def split_complete(plan: GraphPlan, g_split: int):
    """Two plans over the graphs [0, g_split) and [g_split, G) of a complete-graph plan without virtual nodes (the
    puzzles of a Batch never interact), plus the node index where the second one starts -- for the two-branch
    sampling loop (da_sample_loop_pair).  Cached on the plan."""
    cached = getattr(plan, "_halves", None)
    if cached is not None and cached[0] == g_split:
        return cached[1]
    assert plan.dense and not plan.hybrid and plan.n_nodes == plan.n_real and 0 < g_split < plan.n_graphs
    gph, pph = plan.graph_ptr.cpu().tolist(), plan.pad_ptr.cpu().tolist()
    halves = []
    for g0, g1 in ((0, g_split), (g_split, plan.n_graphs)):
        n0, n1, p0, p1 = gph[g0], gph[g1], pph[g0], pph[g1]
        sizes = [gph[i + 1] - gph[i] for i in range(g0, g1)]
        halves.append(GraphPlan(
            n_nodes=n1 - n0, n_real=n1 - n0, n_graphs=g1 - g0, dense=plan.dense,
            n_edges=sum(k * k if plan.dense == 1 else k * (k - 1) for k in sizes), max_graph_nodes=max(sizes),
            row_ptr=None, col_src=None, edge_id=None, graph_ptr=(plan.graph_ptr[g0:g1 + 1] - n0).contiguous(),
            n_pad=p1 - p0, pad_ptr=(plan.pad_ptr[g0:g1 + 1] - p0).contiguous(),
            row_map=(plan.row_map[n0:n1] - p0).contiguous(), _edge_index=None))
    out = (halves[0], halves[1], gph[g_split])
    plan._halves = (g_split, out)
    plan._shape_sig = (plan.dense, tuple(gph))          # host-side signature of the Batch's shape (two-branch loop cache)
    return out


def _hybrid_mode():
    import os
    return os.environ.get("DA_HYBRID", "auto")


def _csr_by_destination(edge_index, n_nodes):
    src, dst = edge_index[0], edge_index[1]
    perm = torch.argsort(dst, stable=True)          # keeps the caller's order inside a segment
    row_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dst.device)
    row_ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_nodes), 0)
    return row_ptr.to(torch.int32), src[perm].to(torch.int32).contiguous(), perm.to(torch.int32).contiguous()


def build_plan(edge_index, batch, virt_nodes=0, detect_dense=True, hybrid=None):
    """edge_index [2,E] int64, batch [N] int64 (sorted graph ids) -> GraphPlan.
    ``hybrid``: "auto" (default; env DA_HYBRID) enables the masked-dense + CSR-remainder split for
    non-complete graphs that are large and dense enough for the matrix cores to win, "force" always
    (tests), "off" never."""
    assert edge_index.dim() == 2 and edge_index.shape[0] == 2, "edge_index must be [2, E]"
    N = batch.numel()
    dev = batch.device
    G = int(batch.max()) + 1 if N > 0 else 0
    counts = torch.bincount(batch, minlength=G)
    graph_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    graph_ptr[1:] = torch.cumsum(counts, 0)
    n_nodes = N
    dense = 0
    if detect_dense and virt_nodes == 0 and edge_index.shape[1] > 0:
        dense = _detect_dense(edge_index, batch, counts)
    real_ei, virt_ei = edge_index, None
    if virt_nodes > 0:
        edge_index = exophormer_edge_index(edge_index, batch, virt_nodes, G)
        virt_ei = edge_index[:, real_ei.shape[1]:]
        n_nodes = N + virt_nodes * G
    E = edge_index.shape[1]
    assert n_nodes < 2 ** 31 and E < 2 ** 31
    src, dst = edge_index[0], edge_index[1]
    # padded slots: the rows of graph g (its real nodes, then its virtual nodes) own a 64-aligned block
    padded = (counts + virt_nodes + 63) // 64 * 64
    pad_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    pad_ptr[1:] = torch.cumsum(padded, 0)
    row_map = torch.arange(N, device=dev) - graph_ptr[batch] + pad_ptr[batch]
    if n_nodes > N:                                   # virtual row N + g*V + v -> slot pad_ptr[g] + n_g + v
        vg = torch.arange(n_nodes - N, device=dev) // virt_nodes
        vv = torch.arange(n_nodes - N, device=dev) % virt_nodes
        row_map = torch.cat([row_map, pad_ptr[vg] + counts[vg] + vv])
    hyb = dict(hybrid=0)
    mode = hybrid if hybrid is not None else _hybrid_mode()
    if dense == 0 and mode != "off" and E > 0 and N > 0:
        hyb = _hybrid_split(real_ei, virt_ei, batch, counts, graph_ptr, padded, n_nodes, N, mode == "force")
    # complete and hybrid graphs: the CSR is built on demand (GraphPlan.ensure_csr) -- sorting the 15-26 M edges of 32
    # 900-piece puzzles was most of the plan time and the matrix-core kernels never read it
    row_ptr, col_src, edge_id = (None, None, None) if (dense or hyb["hybrid"]) else _csr_by_destination(edge_index, n_nodes)
    return GraphPlan(
        n_pad=int(pad_ptr[-1]), pad_ptr=pad_ptr.to(torch.int32), row_map=row_map.to(torch.int32).contiguous(),
        n_nodes=n_nodes, n_real=N, n_graphs=G, dense=dense, n_edges=E,
        max_graph_nodes=int(counts.max()) if G else 0,
        row_ptr=row_ptr, col_src=col_src, edge_id=edge_id, graph_ptr=graph_ptr.to(torch.int32),
        _edge_index=edge_index, **hyb)


def _pack_mask(counts, padded, graph_ptr, regular_cells=None, uniform_bool=None):
    """Adjacency bit matrix in the layout the masked attention reads: rows of graph g start at mask_ptr[g], row stride
    padded_g / 8 bytes, bit j of row i = edge j -> i.  Either from a [G, n, n] bool (all graphs the same size: pure
    elementwise packing) or from (graph, target, source) triples."""
    dev = counts.device
    G = counts.numel()
    stride = padded // 8
    mask_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    mask_ptr[1:] = torch.cumsum(counts * stride, 0)
    total = int(mask_ptr[-1]) + 64
    if uniform_bool is not None:
        n, npad = uniform_bool.shape[1], int(padded[0])
        m = torch.zeros((G, n, npad), dtype=torch.uint8, device=dev)
        m[:, :, :n] = uniform_bool
        w = (1 << torch.arange(8, device=dev, dtype=torch.int32)).to(torch.uint8)
        packed = (m.view(G, n, npad // 8, 8) * w).sum(-1, dtype=torch.uint8)
        out = torch.zeros(total, dtype=torch.uint8, device=dev)
        out[: G * n * (npad // 8)] = packed.reshape(-1)
        return out, mask_ptr
    g, i, j = regular_cells
    byte = mask_ptr[g] + i * stride[g] + (j >> 3)
    acc = torch.zeros(total, dtype=torch.int32, device=dev)
    acc.index_add_(0, byte, (1 << (j & 7)).to(torch.int32))           # bits of a byte are distinct pairs: sum == or
    return acc.to(torch.uint8), mask_ptr


def _irregular_csr(isrc, idst, n_nodes):
    order = torch.argsort(idst, stable=True)
    irr_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=idst.device)
    irr_ptr[1:] = torch.cumsum(torch.bincount(idst, minlength=n_nodes), 0)
    return irr_ptr.to(torch.int32), isrc[order].to(torch.int32).contiguous()


def _hybrid_worth_it(n_max, n_reg, n_edges, pairs):
    """ONE predicate for both entry points (build_plan on an edge list, expander_plan in closed form): the masked matrix-core
    attention pays when the regular edges (unique real -> real pairs inside a graph) are at least a quarter of all edges and at
    least 1 % of the (target, source) pairs, on graphs of at least 32 nodes.  Round 5 re-measured both thresholds (they were 256
    nodes and 3 %): on 256 exophormer puzzles of 6x6 / 8x8 / 12x12 pieces at Exphander degrees of 10 / 30 / 60 % the masked
    kernels beat the edge-list kernels in EVERY cell -- 0.32 vs 0.41 - 0.53 ms per step at 36 pieces, 0.41 vs 0.62 - 1.04 at 64,
    0.55 vs 1.5 - 3.6 at 144 -- and on 64 puzzles of 900 pieces at degree 2 % (d = 18) by 0.94 vs 4.49 ms: a (query, key) pair
    costs the matrix cores ~18 ps at 900 pieces (~100 ps at 144), a gathered edge costs the edge-list kernel 2 - 3 ns, so the
    crossover sits near 0.6 - 1 % density (profiles/r05/NOTES.md).  Every Batch the reference scripts (6x6 .. 20x20 and
    30x30 at degree 60 %) is far above it.  The edge-list kernels keep what is sparser than that, multi-edged, tiny, or asked for
    its attention weights -- and every remainder edge (virtual nodes, duplicates)."""
    return n_max >= 32 and n_reg >= 0.25 * n_edges and n_reg >= 0.01 * pairs


def _hybrid_split(real_ei, virt_ei, batch, counts, graph_ptr, padded, n_nodes, N, force):
    """Split the edge list into (a) "regular" edges -- both ends real, same graph, the pair occurs once --
    stored as one adjacency bit per (target, source) pair, and (b) everything else as CSR by destination
    (PyG's multi-edge semantics live there).  Worth it when the regular part is most of the edges and the
    graphs are big and dense enough that streaming whole K/V tiles beats gathering rows (measured on
    MI355X: 900-node Exphander graphs are 7-19x faster through the matrix cores).

    No sort over the edge list: multiplicities come from ONE scatter-add into the per-graph n_g x n_g cell tables
    (26 M cells for 32 puzzles of 900 pieces; the torch.unique this replaces sorted 15 M keys), the exophormer's virtual
    edges -- all irregular, and a few hundred thousand at most -- are the only thing that is sorted."""
    dev = batch.device
    src, dst = real_ei[0], real_ei[1]
    E0 = src.numel()
    G = counts.numel()
    gs = batch[src]
    same = gs == batch[dst]
    pbase = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    pbase[1:] = torch.cumsum(counts * counts, 0)
    li, lj = dst - graph_ptr[gs], src - graph_ptr[gs]
    cell = pbase[gs] + li * counts[gs] + lj                      # (only meaningful where `same`)
    cell = torch.where(same, cell, torch.zeros_like(cell))
    cnt = torch.zeros(int(pbase[-1]) + 1, dtype=torch.int32, device=dev)
    cnt.index_add_(0, cell, same.to(torch.int32))
    once = cnt == 1
    regular = same & once[cell]
    n_reg = int(regular.sum())
    E = E0 + (virt_ei.shape[1] if virt_ei is not None else 0)
    pairs = int(pbase[-1])
    if not force and not _hybrid_worth_it(int(counts.max()), n_reg, E, pairs):
        return dict(hybrid=0)
    n0 = int(counts[0])
    if bool((counts == n0).all()):
        mask, mask_ptr = _pack_mask(counts, padded, graph_ptr, uniform_bool=once[:-1].view(G, n0, n0))
    else:
        cells = once[:-1].nonzero().flatten()
        g = torch.searchsorted(pbase, cells, right=True) - 1
        loc = cells - pbase[g]
        mask, mask_ptr = _pack_mask(counts, padded, graph_ptr, regular_cells=(g, loc // counts[g], loc % counts[g]))
    irr = (~regular).nonzero().flatten()                           # duplicates / cross-graph pairs: usually none
    isrc, idst = src[irr], dst[irr]
    if virt_ei is not None:
        isrc, idst = torch.cat([isrc, virt_ei[0]]), torch.cat([idst, virt_ei[1]])
    irr_ptr, irr_src = _irregular_csr(isrc, idst, n_nodes)
    return dict(hybrid=1, mask=mask, mask_ptr=mask_ptr, irr_row_ptr=irr_ptr, irr_col_src=irr_src)


def expander_plan(perms, degree, batch_device=None, virt_nodes=0, banded=None):
    """``banded``: None = the process default (DA_EXPANDER_LAYOUT, "banded"); False = the natural slot order the TRAINING path needs
    (its hybrid kernels index the adjacency by node); True = banded.

    GraphPlan of a Batch of Exphander graphs (puzzle_dataset.py:115-152) straight from their permutations
    (``diffassemble_amd.expander``): in position space the graph is a circulant band, so the adjacency bit of a pair is
    a closed form of the two positions -- no edge list, no sort, no multiplicity table.  The exophormer's virtual-node
    edges (exophormer_gnn.py:183-200) depend only on the Batch shape and are planned as in ``build_plan``.  The edge
    list itself (``plan.edge_index``, needed when attention weights are returned) is materialised on demand.
    perms: int64 [G, n]; all graphs have n nodes."""
    from . import expander
    perms = torch.as_tensor(perms)
    dev = torch.device(batch_device) if batch_device is not None else perms.device
    perms = perms.to(dev)
    G, n = perms.shape
    d = int(degree)
    reps = d // 2
    if (n * d) % 2 != 0:
        raise TypeError("nodes * degree must be even")
    N = G * n
    V = int(virt_nodes)
    n_nodes = N + V * G

    def edge_list():
        ei, b = expander.regular_edge_index(perms, d)
        return exophormer_edge_index(ei, b, V, G) if V > 0 else ei

    dup_free = 2 * reps < n and not (d % 2 == 1 and reps >= n // 2)       # rolls k and n - k never coincide
    mode = _hybrid_mode()
    n_virt_edges = (N + G * V * (n + V)) if V > 0 else 0
    E = N * d + n_virt_edges
    if not dup_free or mode == "off" or not (_hybrid_worth_it(n, N * d, E, G * n * n) or mode == "force"):
        batch = torch.arange(G, device=dev).repeat_interleave(n)
        # generic route (multi-edges, tiny graphs, ...): build_plan appends the virtual-node edges itself
        return build_plan(expander.regular_edge_index(perms, d)[0], batch, V)
    # everything but the adjacency bits depends on the Batch SHAPE only: built once per (G, n, V, device)
    key = (G, n, V, str(dev))
    sh = _EXPANDER_SHAPES.get(key)
    if sh is None:
        batch = torch.arange(G, device=dev).repeat_interleave(n)
        counts = torch.full((G,), n, dtype=torch.int64, device=dev)
        graph_ptr = torch.arange(G + 1, device=dev, dtype=torch.int64) * n
        padded = (counts + V + 63) // 64 * 64
        pad_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        pad_ptr[1:] = torch.cumsum(padded, 0)
        row_map = torch.arange(N, device=dev) - graph_ptr[batch] + pad_ptr[batch]
        if V > 0:
            vg = torch.arange(V * G, device=dev) // V
            row_map = torch.cat([row_map, pad_ptr[vg] + n + torch.arange(V * G, device=dev) % V])
            ve = exophormer_edge_index(torch.zeros((2, 0), dtype=torch.int64, device=dev), batch, V, G)
            irr_ptr, irr_src = _irregular_csr(ve[0], ve[1], n_nodes)
        else:
            e = torch.zeros(0, dtype=torch.int64, device=dev)
            irr_ptr, irr_src = _irregular_csr(e, e, n_nodes)
        stride = int(padded[0]) // 8
        mask_ptr = torch.arange(G + 1, device=dev, dtype=torch.int64) * (n * stride)
        sh = dict(counts=counts, padded=padded, graph_ptr=graph_ptr, stride=stride, mask_ptr=mask_ptr,
                  n_pad=int(pad_ptr[-1]), pad_ptr=pad_ptr.to(torch.int32), row_map=row_map.to(torch.int32).contiguous(),
                  graph_ptr32=graph_ptr.to(torch.int32), irr_ptr=irr_ptr, irr_src=irr_src)
        if len(_EXPANDER_SHAPES) > 16:
            _EXPANDER_SHAPES.clear()
        _EXPANDER_SHAPES[key] = sh
    want_banded = (_expander_layout() == "banded") if banded is None else bool(banded)
    if want_banded and int(sh["padded"][0]) <= 3968:
        return _expander_plan_banded(perms, d, V, sh, edge_list, E)
    if dev.type == "cuda":
        # two launches: inverse permutations, then the bit rows (csrc/da_graph.hip)
        from . import _lib
        perms64 = perms.to(torch.int64).contiguous()
        mask = torch.zeros(G * n * sh["stride"] + 64, dtype=torch.uint8, device=dev)
        pos32 = torch.empty(G * n, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().da_expander_mask(G, n, d, _lib.ptr(perms64), _lib.ptr(pos32), sh["stride"], _lib.ptr(mask),
                                               _lib.stream_ptr(dev)))
    else:                                                                 # host tensors (CPU tests of the host logic)
        pos = torch.empty_like(perms)
        pos.scatter_(1, perms, torch.arange(n, device=dev).expand(G, n))  # pos[g, node] = its position in the permutation
        dist = (pos[:, :, None] - pos[:, None, :]) % n                    # [G, target, source]
        cd = torch.minimum(dist, n - dist)
        adj = (cd >= 1) & (cd <= reps)
        if d % 2 == 1:
            adj |= cd * 2 == n
        mask, _ = _pack_mask(sh["counts"], sh["padded"], sh["graph_ptr"], uniform_bool=adj)
    return GraphPlan(
        n_pad=sh["n_pad"], pad_ptr=sh["pad_ptr"], row_map=sh["row_map"],
        n_nodes=n_nodes, n_real=N, n_graphs=G, dense=0, n_edges=E, max_graph_nodes=n,
        row_ptr=None, col_src=None, edge_id=None, graph_ptr=sh["graph_ptr32"], _edge_index=None,
        edge_index_fn=edge_list, hybrid=1, mask=mask, mask_ptr=sh["mask_ptr"], irr_row_ptr=sh["irr_ptr"], irr_col_src=sh["irr_src"])


_EXPANDER_SHAPES = {}
_EXPANDER_BANDS = {}


def _expander_layout():
    """"banded" (default): a graph's padded slots hold its nodes in POSITION order (position in the generator's permutation),
    which turns the adjacency into a circulant band -- whole 32 x 32 blocks of pairs are empty or full, and the masked
    attention skips / un-masks them (da_attn_opt.hip).  "natural" (DA_EXPANDER_LAYOUT=natural): slot order = node order, the
    plan ``build_plan`` derives from the edge list (every block partial)."""
    import os
    return os.environ.get("DA_EXPANDER_LAYOUT", "banded")


def band_adjacency(n, d, device):
    """[n, n] bool, entry (a, b) = the nodes at positions a and b of an Exphander permutation are neighbours
    (puzzle_dataset.py:115-152: positions p and (p - k) mod n, k = 1 .. d // 2, plus p and p + n / 2 for odd d)."""
    a = torch.arange(n, device=device)
    dist = (a[:, None] - a[None, :]) % n
    cd = torch.minimum(dist, n - dist)
    adj = (cd >= 1) & (cd <= d // 2)
    if d % 2 == 1:
        adj |= cd * 2 == n
    return adj


def block_classes(adj, padded):
    """uint8 [padded / 32, stride]: class of every (32-row slab, 32-column block) of a [rows <= padded, cols <= padded] bool
    adjacency (rows / columns beyond it count as "no edge"): 0 = empty, 1 = partial, 2 = full; stride = blocks per row
    rounded up to 16 bytes."""
    dev = adj.device
    nb = padded // 32
    stride = (nb + 15) // 16 * 16
    full = torch.zeros((padded, padded), dtype=torch.bool, device=dev)
    full[: adj.shape[0], : adj.shape[1]] = adj
    blk = full.view(nb, 32, nb, 32)
    cls = blk.any(3).any(1).to(torch.uint8) + blk.all(3).all(1).to(torch.uint8)
    out = torch.zeros((nb, stride), dtype=torch.uint8, device=dev)
    out[:, :nb] = cls
    # behind the rows: one 64-bit word per 128-slot query tile (four slabs), bit kt = some slab of the tile has an edge into
    # 64-key tile kt -- the key tiles a workgroup of the masked attention walks (read with one scalar load)
    nqt, nkt = (nb + 3) // 4, (nb + 1) // 2
    anyb = torch.zeros((nqt * 4, nkt * 2), dtype=torch.bool, device=dev)
    anyb[:nb, :nb] = cls > 0
    need = anyb.view(nqt, 4, nkt, 2).any(3).any(1)                                   # [query tile, key tile]
    words = (need.to(torch.int64) << torch.arange(nkt, device=dev, dtype=torch.int64)[None, :]).sum(1) if nkt <= 62 else None
    if words is None:
        raise ValueError("block classes cover graphs of up to 62 key tiles")
    tail = words.view(torch.uint8).reshape(-1) if words.numel() else torch.zeros(0, dtype=torch.uint8, device=dev)
    return torch.cat([out.reshape(-1), tail]).contiguous(), stride


def _expander_plan_banded(perms, d, V, sh, edge_list, E):
    """The banded layout of ``expander_plan``: per Batch only the node <-> slot maps are computed (two small index ops);
    the adjacency bit rows and the block-class table live in SLOT space, where they depend on (n, d) alone -- ONE copy,
    shared by every graph of every Batch of that shape (115 KB at n = 900: it stays in the L2)."""
    dev = perms.device
    G, n = perms.shape
    N = G * n
    padded = int(sh["padded"][0])
    bkey = (n, d, padded, str(dev))
    band = _EXPANDER_BANDS.get(bkey)
    if band is None:
        adj = band_adjacency(n, d, dev)
        counts1 = torch.full((1,), n, dtype=torch.int64, device=dev)
        mask, _ = _pack_mask(counts1, torch.full((1,), padded, dtype=torch.int64, device=dev), None, uniform_bool=adj[None])
        cls, stride = block_classes(adj, padded)
        band = dict(mask=mask, cls=cls, stride=stride)
        if len(_EXPANDER_BANDS) > 16:
            _EXPANDER_BANDS.clear()
        _EXPANDER_BANDS[bkey] = band
    perms = perms.to(torch.int64)
    pos = torch.empty_like(perms)
    pos.scatter_(1, perms, torch.arange(n, device=dev).expand(G, n))          # pos[g, node] = its position in the permutation
    pad0 = sh["pad_ptr"][:-1].to(torch.int64)
    row_map = (pad0[:, None] + pos).reshape(-1).to(torch.int32)
    slot_node = torch.full((G, padded), -1, dtype=torch.int32, device=dev)
    slot_node[:, :n] = (perms + (torch.arange(G, device=dev) * n)[:, None]).to(torch.int32)
    if V > 0:
        row_map = torch.cat([row_map, sh["row_map"][N:]])
        slot_node[:, n:n + V] = (N + torch.arange(G, device=dev)[:, None] * V + torch.arange(V, device=dev)[None, :]).to(torch.int32)
    zero_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    return GraphPlan(
        n_pad=sh["n_pad"], pad_ptr=sh["pad_ptr"], row_map=row_map.contiguous(),
        n_nodes=N + V * G, n_real=N, n_graphs=G, dense=0, n_edges=E, max_graph_nodes=n,
        row_ptr=None, col_src=None, edge_id=None, graph_ptr=sh["graph_ptr32"], _edge_index=None,
        edge_index_fn=edge_list, hybrid=1, mask=band["mask"], mask_ptr=zero_ptr, irr_row_ptr=sh["irr_ptr"], irr_col_src=sh["irr_src"],
        slot_node=slot_node.reshape(-1).contiguous(), blk_class=band["cls"], blk_class_ptr=zero_ptr, blk_class_stride=band["stride"])


def _detect_dense(edge_index, batch, counts):
    """1 = every graph is complete WITH self loops (rotation dataset, puzzle_dataset.py:609-614),
    2 = complete WITHOUT self loops (non-rotation dataset default), 0 = anything else.
    Exact: edges unique, inside their graph, and the count matches."""
    src, dst = edge_index[0], edge_index[1]
    E = edge_index.shape[1]
    N = batch.numel()
    full = int((counts * counts).sum())
    if E != full and E != full - N:
        return 0
    gs = batch[src]
    if not bool((gs == batch[dst]).all()):
        return 0
    # all E pairs distinct <=> they hit E different cells of the per-graph n_g x n_g tables (one scatter, no sort)
    gptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=src.device)
    gptr[1:] = torch.cumsum(counts, 0)
    pbase = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=src.device)
    pbase[1:] = torch.cumsum(counts * counts, 0)
    cell = pbase[gs] + (dst - gptr[gs]) * counts[gs] + (src - gptr[gs])
    seen = torch.zeros(full, dtype=torch.bool, device=src.device)
    seen[cell] = True
    if int(seen.sum()) != E:
        return 0
    loops = int((src == dst).sum())
    if E == full and loops == N:
        return 1
    if E == full - N and loops == 0:
        return 2
    return 0
