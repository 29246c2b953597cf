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
import ctypes as C
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
    edge_index: torch.Tensor   # int64 [2, E] (extended for exophormer) -- returned with alpha
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

    def ensure_csr(self):
        """CSR by destination of ``edge_index`` (stable: keeps the caller's order inside a segment)."""
        if self.row_ptr is None:
            self.row_ptr, self.col_src, self.edge_id = _csr_by_destination(self.edge_index, self.n_nodes)
        return self

    def with_source_csr(self):
        """Add the by-source orientation of the same edge list (da_graph.out_ptr / out_dst), which
        the attention backward walks to form dK / dV without atomics."""
        if self.out_ptr is None:
            src, dst = self.edge_index[0], self.edge_index[1]
            perm = torch.argsort(src, stable=True)
            ptr = torch.zeros(self.n_nodes + 1, dtype=torch.int64, device=src.device)
            ptr[1:] = torch.cumsum(torch.bincount(src, minlength=self.n_nodes), 0)
            self.out_ptr = ptr.to(torch.int32)
            self.out_dst = dst[perm].to(torch.int32).contiguous()
        return self

    def c_struct(self, need_csr=True):
        """``need_csr=False``: leave the CSR arrays out if they have not been built (only valid for complete
        graphs on a denoiser that reports da_denoiser_flags bit 2, without alpha)."""
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
        return g


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
    if virt_nodes > 0:
        edge_index = exophormer_edge_index(edge_index, batch, virt_nodes, G)
        n_nodes = N + virt_nodes * G
    E = edge_index.shape[1]
    assert n_nodes < 2 ** 31 and E < 2 ** 31
    src, dst = edge_index[0], edge_index[1]
    # complete graphs: the CSR is built on demand (GraphPlan.ensure_csr) -- sorting 26 M edges of 32 900-piece
    # puzzles is most of the plan time and the dense kernels never read it
    row_ptr, col_src, edge_id = (None, None, None) if dense else _csr_by_destination(edge_index, n_nodes)
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
        hyb = _hybrid_split(src, dst, batch, counts, graph_ptr, padded, n_nodes, N, mode == "force")
    return GraphPlan(
        n_pad=int(pad_ptr[-1]), pad_ptr=pad_ptr.to(torch.int32), row_map=row_map.to(torch.int32).contiguous(),
        n_nodes=n_nodes, n_real=N, n_graphs=G, dense=dense, n_edges=E,
        max_graph_nodes=int(counts.max()) if G else 0,
        row_ptr=row_ptr, col_src=col_src, edge_id=edge_id, graph_ptr=graph_ptr.to(torch.int32),
        edge_index=edge_index, **hyb)


def _hybrid_split(src, dst, batch, counts, graph_ptr, padded, n_nodes, N, force):
    """Split the edge list into (a) "regular" edges -- both ends real, same graph, the pair occurs once --
    stored as one adjacency bit per (target, source) pair, and (b) everything else as CSR by destination
    (PyG's multi-edge semantics live there).  Worth it when the regular part is most of the edges and the
    graphs are big and dense enough that streaming whole K/V tiles beats gathering rows (measured on
    MI355X: 900-node Exphander graphs are 7-19x faster through the matrix cores)."""
    dev = src.device
    E = src.numel()
    real = (src < N) & (dst < N)
    same = torch.zeros(E, dtype=torch.bool, device=dev)
    same[real] = batch[src[real]] == batch[dst[real]]
    key = dst * n_nodes + src
    uniq, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    regular = same & (cnt[inv] == 1)
    n_reg = int(regular.sum())
    pairs = int((counts * counts).sum())
    if not force and not (int(counts.max()) >= 256 and n_reg >= 0.5 * E and n_reg >= 0.03 * pairs):
        return dict(hybrid=0)
    G = counts.numel()
    stride = padded // 8                                              # bytes per mask row of graph g
    mask_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    mask_ptr[1:] = torch.cumsum(counts * stride, 0)
    rs, rd = src[regular], dst[regular]
    g = batch[rd]
    i, j = rd - graph_ptr[g], rs - graph_ptr[g]
    byte = mask_ptr[g] + i * stride[g] + (j >> 3)
    acc = torch.zeros(int(mask_ptr[-1]) + 64, dtype=torch.int32, device=dev)
    acc.index_add_(0, byte, (1 << (j & 7)).to(torch.int32))           # bits of a byte are distinct pairs: sum == or
    irr = ~regular
    isrc, idst = src[irr], dst[irr]
    order = torch.argsort(idst, stable=True)
    irr_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
    irr_ptr[1:] = torch.cumsum(torch.bincount(idst, minlength=n_nodes), 0)
    return dict(hybrid=1, mask=acc.to(torch.uint8), mask_ptr=mask_ptr, irr_row_ptr=irr_ptr.to(torch.int32),
                irr_col_src=isrc[order].to(torch.int32).contiguous())


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
