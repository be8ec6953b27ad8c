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


class MixedShardedVerifier:
    """Cooperative verification of one mixed batch (BASELINE.json configs[4]) with the batch
    resident in HBM: rank g holds the contiguous slice [g*n_local, (g+1)*n_local) of a global batch
    of world*n_local items as one structure of arrays + a kind vector.

    Per call (everything on the GPU, nothing synchronises with the host):
      1. split the local kind vector on the device (stable index compaction, libdsv k_kind_*),
      2. gather the rows of each kind into compact arrays (k_gather_rows),
      3. run each kind through its own entry point (dsv_verify_single_dev / _double_dev), one
         after the other on the current stream (running the two kinds on two side streams at
         once was measured 5 % SLOWER: 42.3 against 44.6 M items/s — four sub-batch streams
         compete for the same SIMDs),
      4. all_gather the two per-kind verdict vectors (RCCL when the backend is "nccl"),
      5. scatter them back into GLOBAL batch order with the index vectors obtained by splitting
         the global kind vector (every rank ends up with all world*n_local verdicts).
    The per-kind shard sizes must be equal on all ranks (n_single_local, n_double_local are given
    by the caller, who knows its batch): that is what "shard each kind evenly" means for a batch the
    ranks already hold."""

    def __init__(self, n_local, n_double_local, world, rank, device, group=None, collective=None):
        import torch

        from . import engine as E

        self.E, self.torch = E, torch
        self.n, self.nd, self.ns = n_local, n_double_local, n_local - n_double_local
        self.world, self.rank, self.group, self.dev = world, rank, group, device
        # collective=True runs the all_gathers even for one rank (RCCL rehearsal on a one-GPU box)
        self.collective = world > 1 if collective is None else collective
        u8 = lambda *shape: torch.empty(shape, dtype=torch.uint8, device=device)
        # zero-initialised: an entry the split never writes is a valid (if meaningless) index
        i32 = lambda k: torch.zeros(max(k, 1), dtype=torch.int32, device=device)
        self.idx_s, self.idx_d = i32(self.ns), i32(self.nd)
        self.scratch = u8(E.split_scratch_bytes(n_local))
        self.cs = {k: u8(max(self.ns, 1), w) for k, w in (("u", 32), ("R", 64), ("PK", 64), ("m", 32))}
        self.cd = {k: u8(max(self.nd, 1), w) for k, w in (("u", 32), ("R", 64), ("Rp", 64), ("PK", 64),
                                                          ("PKp", 64), ("m", 32))}
        self.ok_s, self.ok_d = u8(max(self.ns, 1)), u8(max(self.nd, 1))
        self.ws = u8(E.workspace_bytes(max(self.ns, self.nd, 1)))
        N = world * n_local
        self.all_s, self.all_d = u8(max(world * self.ns, 1)), u8(max(world * self.nd, 1))
        self.gidx_s, self.gidx_d = i32(world * self.ns), i32(world * self.nd)
        self.gscratch = u8(E.split_scratch_bytes(N))
        self.out = u8(N)

    def profile(self, batch, global_kinds, reps=3):
        """Where a step's time goes on THIS rank: mean ms of `reps` calls per stage, from HIP events on
        the current stream around each stage (split + row gathers, the single kind's kernels, the
        double kind's kernels, the two all_gathers incl. the wait for the slowest rank, the scatter
        back).  Synchronises; not for use inside a timed region."""
        torch = self.torch
        marks = []
        self._mark = lambda: marks.append(self._record())
        try:
            for _ in range(reps):
                self(batch, global_kinds)
        finally:
            self._mark = None
        torch.cuda.synchronize(self.dev)
        names = ("split_gather_ms", "single_kernels_ms", "double_kernels_ms", "all_gather_ms", "scatter_ms")
        per = len(names) + 1
        out = dict.fromkeys(names, 0.0)
        for r in range(reps):
            ev = marks[r * per:(r + 1) * per]
            for k, name in enumerate(names):
                out[name] += ev[k].elapsed_time(ev[k + 1]) / reps
        return out

    _mark = None

    def _record(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def __call__(self, batch, global_kinds):
        """batch: dict kinds,u,R,Rp,PK,PKp,m (local slice); global_kinds: uint8 [world*n_local].
        Returns the global verdict vector (uint8 [world*n_local], owned by this object)."""
        import torch.distributed as dist

        E, ns, nd = self.E, self.ns, self.nd
        mark = self._mark or (lambda: None)
        mark()
        E.split_kinds_dev(batch["kinds"], self.idx_s, self.idx_d, self.scratch)
        # every gather / scatter below is bounded ON THE DEVICE by the split's own counts: whatever
        # (ns, nd) the caller declared, an index entry the split did not write is never read
        cnt = E.split_counts(self.scratch)
        for k, dst in self.cs.items():
            E.gather_rows_dev(batch[k], self.idx_s, ns, dst, limit=cnt[0:1])
        for k, dst in self.cd.items():
            E.gather_rows_dev(batch[k], self.idx_d, nd, dst, limit=cnt[1:2])
        mark()
        if ns:
            E.verify_single_dev(self.cs["u"][:ns], self.cs["R"][:ns], self.cs["PK"][:ns],
                                self.cs["m"][:ns], self.ok_s, self.ws)
        mark()
        if nd:
            E.verify_double_dev(self.cd["u"][:nd], self.cd["R"][:nd], self.cd["Rp"][:nd],
                                self.cd["PK"][:nd], self.cd["PKp"][:nd], self.cd["m"][:nd],
                                self.ok_d, self.ws)
        mark()
        if self.collective:
            if ns:
                dist.all_gather_into_tensor(self.all_s[:self.world * ns], self.ok_s[:ns], group=self.group)
            if nd:
                dist.all_gather_into_tensor(self.all_d[:self.world * nd], self.ok_d[:nd], group=self.group)
            all_s, all_d = self.all_s, self.all_d
        else:
            all_s, all_d = self.ok_s, self.ok_d
        mark()
        # kind-k item number j of the global batch: rank-major, i.e. j-th in global order
        E.split_kinds_dev(global_kinds, self.gidx_s, self.gidx_d, self.gscratch)
        gcnt = E.split_counts(self.gscratch)
        self.out.zero_()
        if ns:
            E.scatter_verdicts_dev(all_s, self.gidx_s, self.world * ns, self.out, limit=gcnt[0:1])
        if nd:
            E.scatter_verdicts_dev(all_d, self.gidx_d, self.world * nd, self.out, limit=gcnt[1:2])
        # the declared per-kind counts must be what the kind vectors hold, locally and globally;
        # otherwise the verdict mapping is meaningless: all zeros (decided on the device, no sync)
        good = ((cnt[0] == ns) & (cnt[1] == nd) & (gcnt[0] == self.world * ns)
                & (gcnt[1] == self.world * nd))
        self.out.mul_(good.to(self.out.dtype))
        mark()
        return self.out

    def local_counts(self):
        """(n_single, n_double) the device found in the LAST local split (synchronises)."""
        t = self.E.split_counts(self.scratch).cpu()
        return int(t[0]), int(t[1])
