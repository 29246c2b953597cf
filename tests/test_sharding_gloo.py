"""world_size-2 gloo test (CPU) of the multi-GPU layout: puzzles are partitioned across ranks without
any data-path collective, and the gathered result equals the single-process result."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import weights as W

def _free_port():
    """an unused TCP port for the rendezvous of one spawned world (fixed numbers collide with sockets in TIME_WAIT)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]



def _fake_denoise(x, feats, edge_index, batch):
    """A stand-in with the path's dependency structure: each node's output depends on its own graph
    only (sum of source features over incoming edges), so sharding by graph must not change it."""
    agg = torch.zeros_like(x).index_add_(0, edge_index[1], x[edge_index[0]])
    return agg * 0.5 + feats[:, : x.shape[1]] + batch[:, None].float() * 0  # graph ids are re-based per rank


def _worker(rank, world, port, sizes, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffassemble_amd import sharding as S
    eis = [W.dense_edge_index(n, True) for n in sizes]
    ei, batch = W.collate(eis, sizes)
    N = sum(sizes)
    x, feats = W.make_inputs(N, 4, 16, 3)
    xs, fs, eis_, bs, lo, hi = S.shard_batch(x, feats, ei, batch, rank, world)
    assert int(bs.min()) == 0 and (eis_ >= 0).all() and (eis_ < hi - lo).all()
    local = _fake_denoise(xs, fs, eis_, bs)
    full = S.gather_rows(local, N, lo)
    t = S.max_over_ranks(0.1 * (rank + 1))
    if rank == 0:
        ret["full"] = full
        ret["t"] = t
        ret["ref"] = _fake_denoise(x, feats, ei, batch)
    dist.destroy_process_group()


def test_puzzle_sharding_world2_gloo():
    sizes = [5, 9, 4, 7, 6]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), sizes, ret), nprocs=2, join=True)
    assert torch.allclose(ret["full"], ret["ref"])
    assert abs(ret["t"] - 0.2) < 1e-9


def test_shard_range_is_a_balanced_partition():
    from diffassemble_amd.sharding import shard_range
    for G in (1, 7, 8, 512, 513):
        for world in (1, 2, 4, 8):
            spans = [shard_range(G, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, ret):
    """Data-parallel training layout on the CPU oracle: each rank back-propagates p_losses on its shard
    of the puzzles into a flat gradient buffer, ONE all-reduce averages it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from diffassemble_amd import sharding as S
    from oracle import diffusion as DF
    sizes = [9, 9, 9, 9]
    ei, batch = W.collate([W.dense_edge_index(n, True) for n in sizes], sizes)
    N = sum(sizes)
    sd0 = W.make_denoiser_state(20, 4, 4, D=128, hidden=32, variant="2d", arch="transformer", virt_nodes=0,
                                n_layers=4, heads=8, seed=5, qk_gain=2.0)
    sch = DF.make_schedule(20)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32))
    feats = torch.from_numpy(rng.standard_normal((N, 64)).astype(np.float32))
    tg = torch.from_numpy(rng.integers(0, 20, len(sizes)))
    t = tg[batch]

    def flat_grads(xs, ts, ns, eis, fs, bs):
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        DF.p_losses(sd, sch, xs, ts, ns, eis, fs, bs, "EPSILON").backward()
        return torch.cat([sd[k].grad.flatten() for k in sorted(sd) if sd[k].grad is not None])

    xs, fs, eis, bs, lo, hi = S.shard_batch(x, feats, ei, batch, rank, world)
    flat = flat_grads(xs, t[lo:hi], noise[lo:hi], eis, fs, bs)
    S.allreduce_gradients(flat)
    if rank == 0:
        ret["dp"] = flat
        ret["full"] = flat_grads(x, t, noise, ei, feats, batch)
    dist.destroy_process_group()


def test_data_parallel_gradient_allreduce_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    err = float((ret["dp"] - ret["full"]).abs().max() / ret["full"].abs().max())
    assert err < 1e-5, err
