"""Multi-GPU layout of the path: puzzles (graphs of a Batch) are independent, so they are
partitioned across ranks -- one process per GPU, NO data-path collective.  The only collectives are an
optional gather of the final poses to rank 0 (metrics) and the MAX-reduce of a timing.  Works on any
``torch.distributed`` backend ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_graphs, rank, world):
    """Contiguous, balanced slice [lo, hi) of the graph ids owned by ``rank``."""
    base, rem = divmod(n_graphs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, feats, edge_index, batch, rank, world):
    """Slice a collated PyG-style Batch (``batch`` sorted) down to this rank's graphs, re-basing node
    and graph indices.  Returns (x, feats, edge_index, batch, node_lo, node_hi)."""
    G = int(batch.max()) + 1
    lo, hi = shard_range(G, rank, world)
    node_mask = (batch >= lo) & (batch < hi)
    idx = node_mask.nonzero().flatten()
    n_lo = int(idx[0]) if idx.numel() else 0
    n_hi = n_lo + idx.numel()
    emask = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    return (x[n_lo:n_hi], feats[n_lo:n_hi], edge_index[:, emask] - n_lo, batch[n_lo:n_hi] - lo, n_lo, n_hi)


def gather_rows(local, total_rows, node_lo, dst=0):
    """Assemble per-rank row blocks [node_lo : node_lo + len(local)) into one [total_rows, c] tensor on
    every rank (all_gather of padded blocks; poses are tiny)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    meta = torch.tensor([node_lo, local.shape[0]], dtype=torch.int64, device=local.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    mx = max(int(m[1]) for m in metas)
    pad = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    blocks = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad)
    out = torch.zeros((total_rows, local.shape[1]), dtype=local.dtype, device=local.device)
    for m, b in zip(metas, blocks):
        out[int(m[0]): int(m[0]) + int(m[1])] = b[: int(m[1])]
    return out


def max_over_ranks(seconds, device=None):
    """Job time = slowest rank (bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    if dist.get_backend() != "nccl":
        device = "cpu"                 # gloo (tests, several ranks on one GPU): reduce on the host
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def exchange_active():
    """True when a gradient exchange has to run: an initialised process group with more than one rank -- or a ONE-rank
    RCCL ("nccl") group, where the collective is executed anyway (it costs one small launch) so that a single-GPU box
    exercises the very call the 8-GPU node makes."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or dist.get_backend() == "nccl")


def allreduce_gradients(flat_grad, average=True):
    """Data-parallel training exchange: ONE all-reduce of the flat fp32 gradient buffer
    (``TrainEngine.flat_grad``, 12.9 MB for the 2D denoiser) per optimizer step -- a single fused bucket,
    sized for xGMI's point-to-point links, instead of Lightning-DDP's per-bucket hooks.  Every rank holds
    an equal number of equally sized puzzles, so the mean over ranks of the per-rank mean losses'
    gradients is the global-batch gradient (spatial_diffusion.py:707-721 under DDP)."""
    if not exchange_active():
        return flat_grad
    if flat_grad.is_cuda and dist.get_backend() == "gloo":
        # test configuration only (several ranks sharing one GPU cannot use RCCL): stage through the host
        host = flat_grad.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat_grad.copy_(host)
    else:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    if average:
        flat_grad.div_(dist.get_world_size())
    return flat_grad
