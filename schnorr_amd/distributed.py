"""Multi-GPU batch verification: one process per GPU, contiguous shards, one all-gather of the
verdict bytes (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).

Every signature is independent (the reference's verify is a pure function,
/root/reference/src/keys/public.rs:121-130), so there is no exchange step inside the data path:
rank g verifies items [g*n/G, (g+1)*n/G) and the only collective is the gather of n/G bytes per
rank (1 MiB per rank for the 2^23 batch of BASELINE.json configs[4]).
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous, balanced partition of range(n): sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def shard_sizes(n, world):
    return [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]


def split_mixed(kinds):
    """Index sets of a mixed batch by kind (0 = single, 1 = double), each to be sharded
    separately so every rank gets the same single:double ratio (a double costs ~2x)."""
    kinds = np.asarray(kinds)
    return np.nonzero(kinds == 0)[0], np.nonzero(kinds == 1)[0]


def gather_verdicts(local_ok, n, group=None):
    """all_gather of per-rank verdict vectors (torch uint8 tensors, possibly ragged) into the
    full [n] vector in original order.  Returns a tensor on local_ok's device."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(n, world)
    if local_ok.numel() != sizes[dist.get_rank(group)]:
        raise ValueError("local shard has %d verdicts, expected %d"
                         % (local_ok.numel(), sizes[dist.get_rank(group)]))
    maxlen = max(sizes) if sizes else 0
    padded = torch.zeros(maxlen, dtype=torch.uint8, device=local_ok.device)
    padded[: local_ok.numel()] = local_ok
    out = torch.empty(world * maxlen, dtype=torch.uint8, device=local_ok.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    if all(s == maxlen for s in sizes):
        return out
    return torch.cat([out[r * maxlen: r * maxlen + sizes[r]] for r in range(world)])


def verify_single_sharded(u, R, PK, m, verify_fn, to_tensor, group=None):
    """Verify the FULL batch (host numpy arrays, identical on every rank) cooperatively:
    each rank runs `verify_fn` on its shard only, then verdicts are all-gathered."""
    import torch.distributed as dist

    n = u.shape[0]
    lo, hi = shard_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    local = verify_fn(u[lo:hi], R[lo:hi], PK[lo:hi], m[lo:hi])
    return gather_verdicts(to_tensor(local), n, group)


def verify_mixed_sharded(kinds, single, double, verify_single_fn, verify_double_fn, to_tensor,
                         group=None):
    """BASELINE.json configs[4] shape: one batch holding single (kind 0) and double (kind 1)
    signatures in arbitrary interleaving, identical on every rank.

    `single` = (u, R, PK, m), `double` = (u, R, Rp, PK, PKp, m): host arrays holding only the items
    of that kind, in batch order.  Each kind is sharded evenly on its own (a double costs about
    twice a single, so every rank gets the same single:double ratio), each rank verifies its two
    shards, both verdict vectors are all-gathered and scattered back into the original order.
    Returns the full verdict vector (torch uint8, on the device `to_tensor` puts things on)."""
    import torch
    import torch.distributed as dist

    kinds = np.asarray(kinds)
    idx_s, idx_d = split_mixed(kinds)
    if single[0].shape[0] != idx_s.size or double[0].shape[0] != idx_d.size:
        raise ValueError("kind vector and per-kind arrays disagree")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = []
    for cols, fn, count in ((single, verify_single_fn, idx_s.size), (double, verify_double_fn, idx_d.size)):
        lo, hi = shard_bounds(count, rank, world)
        local = fn(*[c[lo:hi] for c in cols]) if hi > lo else np.zeros(0, np.uint8)
        parts.append(gather_verdicts(to_tensor(np.ascontiguousarray(local, dtype=np.uint8)), count, group))
    out = torch.empty(kinds.size, dtype=torch.uint8, device=parts[0].device)
    out[torch.from_numpy(idx_s).to(out.device)] = parts[0]
    out[torch.from_numpy(idx_d).to(out.device)] = parts[1]
    return out
