"""Row sharding of the observed class across GPUs (one process per GPU).

The reference is single-threaded; this layer is new.  Within a sweep every rank
holds replicas of the latent tables and sweeps only its block of observed rows.
The exchange step between sweeps is:
  * all-reduce(sum) of the int64 delta-reference-count vectors of each block's
    root table (the CRP sufficient statistics) — RCCL over xGMI when the backend
    is "nccl", gloo in the CPU tests;
  * all-gather of the (rare) new-row records, merged in global row order so every
    rank applies the identical commit and the replicas stay bit-identical for any
    number of ranks.
"""
import numpy as np


def shard_bounds(n_rows, rank, world):
    """Contiguous block partition of the observed rows."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Comm:
    """Thin wrapper over torch.distributed (or a no-op for a single process)."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        try:
            import torch.distributed as dist
            self.dist = dist if dist.is_available() and dist.is_initialized() else None
        except Exception:
            self.dist = None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = device if device is not None else "cpu"

    def allreduce_sum_i64(self, arr):
        """Sum of an int64 numpy vector over ranks (integer => order independent)."""
        if not self.dist:
            return arr
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_varlen_i32(self, arr):
        """Concatenation over ranks (rank order) of int32 vectors of differing length."""
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        if not self.dist:
            return arr
        n = self.torch.tensor([arr.size], dtype=self.torch.int64, device=self.device)
        sizes = [self.torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        m = max(max(sizes), 1)
        buf = self.torch.zeros(m, dtype=self.torch.int32, device=self.device)
        buf[:arr.size] = self.torch.from_numpy(arr).to(self.device)
        outs = [self.torch.zeros_like(buf) for _ in range(self.world)]
        self.dist.all_gather(outs, buf)
        return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)])

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_float(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def exchange_and_commit(trace, lowered, comm, row_lo, choice_local, stats_local, new_rows_local, global_cur=False,
                        moved_local=None):
    """Apply one sweep's result to the replicated trace.

    choice_local [n_blocks][n_local]: chosen referents of this rank's rows;
    stats_local  {block: int64 delta counts over the root table} from the kernel;
    new_rows_local {block: (local rows, vals)}.
    global_cur=False: trace.cur holds only this rank's rows (observed-class sweeps only, bench.py);
    global_cur=True : trace.cur holds every observed row on every rank (needed by the latent-class
    sweeps, whose evidence sets span all rows): the (row, new referent) pairs of the rows that moved
    are all-gathered as well, so the whole trace stays replicated.
    moved_local {block: (local rows ascending, new referent)} (pclean_get_moved): when given, only those rows
    are touched and choice_local is not scanned.
    Returns the global number of rows whose referent changed."""
    changed = 0
    n_local = np.asarray(choice_local).shape[1]
    for bi, blk in enumerate(lowered.blocks):
        if blk.get("score"):
            continue
        cname = blk["root_class"]
        t = trace.tables[cname]
        nn = len(blk["nodes"])
        n_before = t.n
        delta = comm.allreduce_sum_i64(stats_local[bi][:n_before])
        rows, vals = new_rows_local.get(bi, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
        g_rows = comm.allgather_varlen_i32(np.asarray(rows, np.int32) + row_lo)
        g_vals = comm.allgather_varlen_i32(np.asarray(vals, np.int32)).reshape(-1, nn)
        order = np.argsort(g_rows, kind="stable")  # identical order on every rank -> identical row ids
        g_rows = g_rows[order]
        new_ids = trace.materialise_bulk(bi, g_vals[order])
        t = trace.tables[cname]
        t.counts[new_ids] += 1  # each new row is referred to by its creator
        t.counts[:n_before] += delta
        # this rank's own rows
        if moved_local is not None:
            moved, ch_m = moved_local[bi]
            ch_m = np.array(ch_m, dtype=np.int32)
            fresh = np.flatnonzero(ch_m < 0)
            if len(fresh):
                ch_m[fresh] = new_ids[np.searchsorted(g_rows, moved[fresh] + row_lo)]
            changed += len(moved)
            if global_cur:
                g_moved = comm.allgather_varlen_i32(np.asarray(moved, np.int32) + row_lo)
                g_new = comm.allgather_varlen_i32(ch_m)
                trace.cur[bi, g_moved] = g_new
            else:
                trace.cur[bi][moved] = ch_m
            trace.delete_rows_bulk(cname, np.nonzero((t.counts[:t.n] == 0) & t.live[:t.n])[0])
            continue
        ch = np.asarray(choice_local[bi])
        fresh = np.flatnonzero(ch < 0)
        if len(fresh):
            ch = ch.copy()
            ch[fresh] = new_ids[np.searchsorted(g_rows, fresh + row_lo)]
        if global_cur:
            cur = trace.cur[bi, row_lo:row_lo + n_local]
            moved = np.nonzero(ch != cur)[0]
            changed += len(moved)
            g_moved = comm.allgather_varlen_i32(moved.astype(np.int32) + row_lo)
            g_new = comm.allgather_varlen_i32(ch[moved].astype(np.int32))
            trace.cur[bi, g_moved] = g_new
        else:
            cur = trace.cur[bi]
            moved = np.flatnonzero(ch != cur)
            changed += len(moved)
            cur[moved] = ch[moved]
        # garbage-collect rows nobody refers to any more (ascending id: deterministic)
        trace.delete_rows_bulk(cname, np.nonzero((t.counts[:t.n] == 0) & t.live[:t.n])[0])
    return int(comm.allreduce_sum_i64(np.array([changed], dtype=np.int64))[0])
