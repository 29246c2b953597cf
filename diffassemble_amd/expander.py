"""Exphander graphs on the device: the random d-regular graphs of
puzzle_diff/dataset/puzzle_dataset.py:115-152 (``generate_random_regular_graph``), generated where the
denoiser consumes them.

Structure of the reference's construction, which everything here rests on: it draws ONE permutation
``nodes = rng.permutation(n)`` and connects ``nodes[p]`` with ``nodes[(p - k) mod n]`` for k = 1 .. d // 2 (np.roll by
k), plus -- odd d -- the perfect matching ``nodes[p] <-> nodes[p + n // 2]`` (p < n // 2), and symmetrises.  In
POSITION space (p = position of a node in ``nodes``) the graph is a circulant band: p ~ q  iff  the cyclic distance of
p and q is in [1, d // 2] (or exactly n // 2 for the matching of an odd d).  So a graph is fully described by its
permutation and d, and both the edge list in the reference's order and the adjacency of any pair are closed-form --
no sort, no 26 M-edge intermediate.

``regular_edge_index`` emits the reference's int64 ``edge_index`` (same edges, same order as
``np.concatenate([ei[0], ei[1]]), np.concatenate([ei[1], ei[0]])``) for a whole Batch with torch index arithmetic on
the device; ``diffassemble_amd.graph_plan.expander_plan`` goes straight from the permutations to the int32 CSR /
adjacency mask the kernels walk (da_expander_plan, csrc/da_expander.hip).

The host draws the permutations (numpy PCG64, as the reference's DataLoader workers do: ``rng.permutation`` is the only
random call of the generator) -- 900 integers per puzzle; the spectral-gap retry loop of ``generate_random_expander``
(:33-103, scipy ``eigsh``) is dataset-side policy and stays with the caller.
"""
import numpy as np
import torch


def draw_permutations(n, n_graphs, rng):
    """[G, n] int64: ``rng.permutation(np.arange(n))`` per graph, in graph order (puzzle_dataset.py:136)."""
    return torch.from_numpy(np.stack([rng.permutation(np.arange(n)) for _ in range(n_graphs)]).astype(np.int64))


def regular_edge_index(perms, degree, device=None):
    """perms [G, n] (a permutation of 0..n-1 per graph), degree d -> (edge_index [2, G*n*d] int64 with PyG-collated
    node offsets, batch [G*n] int64).  Per graph the columns are exactly ``generate_random_regular_graph``'s
    (senders, receivers)."""
    perms = torch.as_tensor(perms)
    if perms.dim() == 1:
        perms = perms[None]
    dev = torch.device(device) if device is not None else perms.device
    perms = perms.to(dev)
    G, n = perms.shape
    d = int(degree)
    if (n * d) % 2 != 0:
        raise TypeError("nodes * degree must be even")
    reps = d // 2
    k = torch.arange(1, reps + 1, device=dev)
    p = torch.arange(n, device=dev)
    # ns = hstack([roll(nodes, i + 1)]): roll(nodes, k)[p] = nodes[(p - k) mod n]
    rolled = perms[:, (p[None, :] - k[:, None]) % n].reshape(G, reps * n)
    tiled = perms.repeat(1, reps)
    if d % 2 == 1:
        tiled = torch.cat([tiled, perms[:, : n // 2]], 1)
        rolled = torch.cat([rolled, perms[:, n // 2:]], 1)
    off = (torch.arange(G, device=dev) * n)[:, None]
    s = torch.cat([tiled, rolled], 1) + off
    r = torch.cat([rolled, tiled], 1) + off
    batch = torch.arange(G, device=dev).repeat_interleave(n)
    return torch.stack([s.reshape(-1), r.reshape(-1)]), batch


def percent_degree(n, percent):
    """``Puzzle_Dataset``'s "--degree P%" for an n-piece puzzle (puzzle_dataset.py:46-47): round(P (n - 1) / 100), made even-sum
    (n * d must be even for a d-regular graph: one less when it is not)."""
    d = int(round(percent * (n - 1) / 100.0))
    d = max(min(d, n - 1), 2)
    return d - (d * n) % 2


def ragged_regular_batch(sides, percent, rng, device=None):
    """The PyG ``Batch`` graph of a RAGGED set of puzzles, as the reference's DataLoader collates it for the scripted run
    (train_celeba_rot.sh:4-15: ``-puzzle_sizes 6 8 .. 20 -batch_size 8 --degree 60%``): puzzle g has sides[g]^2 pieces and its own
    Exphander graph of degree ``percent_degree(n_g, percent)``.  -> (edge_index [2, E] int64 with node offsets, batch [N] int64,
    degrees list)."""
    eis, batches, degs, off = [], [], [], 0
    for g, side in enumerate(sides):
        n = int(side) * int(side)
        d = percent_degree(n, percent)
        ei, _ = regular_edge_index(draw_permutations(n, 1, rng), d, device)
        eis.append(ei + off)
        batches.append(torch.full((n,), g, dtype=torch.int64, device=ei.device))
        degs.append(d)
        off += n
    return torch.cat(eis, 1), torch.cat(batches), degs
