"""
dist.py — multi-GPU sharding of the hot path: one process per GPU under torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in CPU rehearsals).

The FRI commit shards by WHOLE commits: a batch of independent codewords is split into contiguous per-rank
slices, every rank runs its slice through its own GPU context (lock-step batched commit), and only results
(roots, last codewords, proof-stream bytes — a few KB per commit) are gathered.  There is no data-path
collective.  `engine` is the local compute: by default the GPU (`Fri.commit_batch`); tests inject another
callable to rehearse the sharding/gather logic on CPU ranks — the product never does.
"""
import numpy as np


def shard_range(n_items, world, rank):
    """contiguous, balanced slice [start, stop) of `n_items` for `rank` of `world` (first n%world ranks get one more)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gpu_engine(N, expansion_factor=4, num_colinearity_tests=17, offset=None, ctx=None):
    """local engine: lock-step batched commit on this rank's GPU; returns (roots (B,R,64) u8, last (B,L,2) u64, [stream bytes])."""
    from .field import GENERATOR
    from .fri import Fri
    from .proof_stream import ProofStream
    fri = Fri(GENERATOR if offset is None else offset, None, N, expansion_factor, num_colinearity_tests, ctx=ctx)

    def run(codewords):
        B = len(codewords)
        if B == 0:
            R = fri.num_rounds()
            return np.zeros((0, R, 64), np.uint8), np.zeros((0, N >> max(R - 1, 0), 2), np.uint64), []
        streams = [ProofStream() for _ in range(B)]
        roots, last = fri.commit_batch(codewords, streams)
        r = np.array([[np.frombuffer(h.as_bytes(), dtype=np.uint8) for h in row] for row in roots], dtype=np.uint8).reshape(B, -1, 64)
        return r, last, [s.serialize() for s in streams]
    return run


def commit_batch_sharded(codewords, engine, group=None, dst=0):
    """`codewords`: the FULL batch (array-like of shape (B_total, N, 2)) or a callable `(start, stop) -> slice` that
    materialises only this rank's share.  Every rank commits its slice with `engine`; rank `dst` returns
    (roots, last_codewords, streams) for the whole batch in the original order, other ranks return None."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if callable(codewords):
        total = codewords(None, None)
        mine = lambda a, b: codewords(a, b)
    else:
        total = len(codewords)
        mine = lambda a, b: codewords[a:b]
    a, b = shard_range(total, world, rank)
    roots, last, streams = engine(mine(a, b))
    part = (a, b, np.asarray(roots), np.asarray(last), list(streams))
    if world == 1:
        return part[2], part[3], part[4]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(part, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    gathered.sort(key=lambda p: p[0])
    assert gathered[0][0] == 0 and gathered[-1][1] == total and all(x[1] == y[0] for x, y in zip(gathered, gathered[1:]))
    return (np.concatenate([p[2] for p in gathered]), np.concatenate([p[3] for p in gathered]),
            [s for p in gathered for s in p[4]])


def max_over_ranks(value, group=None, device=None):
    """max of a Python float over ranks (the timing reduction bench.py uses)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend(group) == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
