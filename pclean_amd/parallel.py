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

from .trace import unique_rows


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

    def allgather_list_i32(self, arr):
        """[rank 0's vector, rank 1's vector, ...] of int32 vectors of differing length (2 collectives)."""
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        if not self.dist:
            return [arr]
        n = self.torch.tensor([arr.size], dtype=self.torch.int64, device=self.device)
        sizes = [self.torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        m = max(max(sizes), 1)
        buf = self.torch.zeros(m, dtype=self.torch.int32, device=self.device)
        buf[:arr.size] = self.torch.from_numpy(arr).to(self.device)
        outs = [self.torch.zeros_like(buf) for _ in range(self.world)]
        self.dist.all_gather(outs, buf)
        return [o[:k].cpu().numpy() for o, k in zip(outs, sizes)]

    def allgather_known_i32(self, arr, sizes):
        """[rank 0's vector, rank 1's vector, ...] when every rank already knows all sizes: ONE collective
        (all_gather_into_tensor of chunks padded to the largest size) and one device-to-host copy."""
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        if not self.dist:
            return [arr]
        m = max(int(max(sizes)), 1)
        buf = self.torch.zeros(m, dtype=self.torch.int32, device=self.device)
        if arr.size:
            buf[:arr.size] = self.torch.from_numpy(arr).to(self.device)
        out = self.torch.empty(m * self.world, dtype=self.torch.int32, device=self.device)
        self.dist.all_gather_into_tensor(out, buf)
        host = out.cpu().numpy().reshape(self.world, m)
        return [host[r, :int(k)] for r, k in enumerate(sizes)]

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
                        moved_local=None, n_local=None, stats_reduced=False, sweep_idx=0):
    """Apply one sweep's result to the replicated trace.

    choice_local [n_blocks][n_local]: chosen referents of this rank's rows;
    stats_local  {block: int64 delta counts over the root table} from the kernel;
    new_rows_local {block: (local rows, vals)}.
    global_cur=False: trace.cur holds only this rank's rows (observed-class sweeps only, bench.py);
    global_cur=True : trace.cur holds every observed row on every rank (needed by the latent-class
    sweeps, whose evidence sets span all rows): the (row, new referent) pairs of the rows that moved
    are exchanged as well, so the whole trace stays replicated.
    moved_local {block: (local rows ascending, new referent)} (pclean_get_moved): when given, only those rows
    are touched and choice_local is not scanned (it may be None if n_local is given).

    stats_reduced: stats_local is already summed over the ranks on the device (Engine.sweep_stats_reduced).
    Two collectives per sweep, whatever the number of blocks: ONE all-reduce(sum) of the concatenated int64
    delta-count vectors + the moved-row counter + the per-rank message sizes, and ONE all-gather of the new-row
    records and moved rows of all blocks.
    Returns the global number of rows whose referent changed."""
    blocks = [bi for bi, blk in enumerate(lowered.blocks) if not blk.get("score")]
    if n_local is None:
        n_local = np.asarray(choice_local).shape[1]
    # ---- local payloads (tables are still in their pre-sweep state) --------------------------------
    n_before, moved_of, red, msg = {}, {}, [], []
    n_changed = 0
    for bi in blocks:
        blk = lowered.blocks[bi]
        t = trace.tables[blk["root_class"]]
        nn = len(blk["nodes"])
        n_before[bi] = t.n
        red.append(np.asarray(stats_local[bi][:t.n], dtype=np.int64))
        rows, vals = new_rows_local.get(bi, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
        if moved_local is not None:
            moved, ch_m = moved_local[bi]
            moved, ch_m = np.asarray(moved, np.int32), np.asarray(ch_m, np.int32)
        else:
            ch = np.asarray(choice_local[bi])
            cur = trace.cur[bi, row_lo:row_lo + n_local] if global_cur else trace.cur[bi]
            moved = np.flatnonzero(ch != cur).astype(np.int32)
            ch_m = ch[moved].astype(np.int32)
        moved_of[bi] = (moved, ch_m)
        n_changed += len(moved)
        msg += [np.array([len(rows)], np.int32), np.asarray(rows, np.int32) + row_lo, np.asarray(vals, np.int32).reshape(-1)]
        if global_cur:
            msg += [np.array([len(moved)], np.int32), moved + row_lo, ch_m]
    red.append(np.array([n_changed], dtype=np.int64))
    # ---- the exchange ------------------------------------------------------------------------------
    # The message sizes ride along in the all-reduce (every rank adds its size at its own index), so the
    # variable-length all-gather needs no size exchange of its own: TWO collectives per sweep (plus the fused
    # device-side all-reduce of the statistics when stats_reduced).
    msg = np.concatenate(msg)
    sizes = np.zeros(comm.world, dtype=np.int64)
    sizes[comm.rank] = msg.size
    if stats_reduced:  # stats_local already holds the sums over all ranks (Engine.sweep_stats_reduced: one RCCL
        total = np.concatenate(red)  # all-reduce of the device-resident buffers); only counter and sizes are local
        small = comm.allreduce_sum_i64(np.concatenate([[n_changed], sizes]))
        total[-1] = small[0]
        sizes = small[1:]
    else:
        both = comm.allreduce_sum_i64(np.concatenate(red + [sizes]))
        total, sizes = both[:-comm.world], both[-comm.world:]
    parts = comm.allgather_known_i32(msg, sizes)
    # ---- identical commit on every rank ------------------------------------------------------------
    cursor = [0] * len(parts)
    off = 0
    for bi in blocks:
        blk = lowered.blocks[bi]
        cname = blk["root_class"]
        nn = len(blk["nodes"])
        delta = total[off:off + n_before[bi]]
        off += n_before[bi]
        g_rows, g_vals, g_moved, g_new = [], [], [], []
        for r, p in enumerate(parts):  # rank order == global row order (contiguous shards)
            c = cursor[r]
            k = int(p[c])
            g_rows.append(p[c + 1:c + 1 + k])
            g_vals.append(p[c + 1 + k:c + 1 + k + k * nn].reshape(k, nn))
            c += 1 + k + k * nn
            if global_cur:
                m = int(p[c])
                g_moved.append(p[c + 1:c + 1 + m])
                g_new.append(p[c + 1 + m:c + 1 + 2 * m])
                c += 1 + 2 * m
            cursor[r] = c
        g_rows = np.concatenate(g_rows)
        g_vals = np.concatenate(g_vals)
        if len(g_rows) > 1 and np.any(g_rows[1:] < g_rows[:-1]):  # (contiguous shards arrive in global row order)
            order = np.argsort(g_rows, kind="stable")  # identical order on every rank -> identical row ids
            g_rows = g_rows[order]
            g_vals = g_vals[order]
        t = trace.tables[cname]
        t.counts[:n_before[bi]] += delta
        if len(g_rows):
            g_vals = np.array(g_vals)
            g_part = -1 - g_vals[:, 0]  # entry 0 of a record: -1 - chosen particle (pclean_get_new_rows)
            g_vals[:, 0] = -1
            # identical new-row proposals of one sweep become ONE latent row (as commit_batch does for the
            # initialisation): in the sequential reference the second row would have joined the first row's
            # new referent instead of creating a duplicate entity.  Rows are created in order of first occurrence.
            first, grp = unique_rows(g_vals)
            # a proposing row whose OLD referent has just lost its last reference and holds exactly the proposed
            # values keeps it (Trace.materialise_bulk): the row re-proposed its own private referent
            reuse = None
            if global_cur:
                old = trace.cur[bi, g_rows[first]].astype(np.int64)
                ok = old >= 0
                ok[ok] = (t.counts[old[ok]] == 0) & t.live[old[ok]]
                reuse = np.where(ok, old, -1)
            new_ids = trace.materialise_bulk(bi, g_vals[first], reuse,
                                             origin=(g_rows[first], g_part[first], sweep_idx))[grp]
        else:
            new_ids = np.empty(0, dtype=np.int64)
        np.add.at(t.counts, new_ids, 1)  # each new row is referred to by its creator(s)

        def resolve(rows_global, ch):
            ch = np.array(ch, dtype=np.int32)
            fresh = np.flatnonzero(ch < 0)
            if len(fresh):
                ch[fresh] = new_ids[np.searchsorted(g_rows, rows_global[fresh])]
            return ch

        if global_cur:
            g_moved = np.concatenate(g_moved)
            trace.cur[bi, g_moved] = resolve(g_moved, np.concatenate(g_new))
        else:
            moved, ch_m = moved_of[bi]
            trace.cur[bi][moved] = resolve(moved + row_lo, ch_m)
        # garbage-collect rows nobody refers to any more
        trace.delete_rows_bulk(cname, np.nonzero((t.counts[:t.n] == 0) & t.live[:t.n])[0])
    return int(total[-1])
