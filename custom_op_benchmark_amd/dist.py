"""Node-range partitioned attention step over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Not in the reference (single process, single GPU) -- BASELINE.json's north_star
asks for it.  Design:

* Nodes are cut into `world` contiguous ranges.  Rank p owns the Q/K/V/dO rows of its range and
  every edge whose SOURCE row is in its range (so all per-edge scalars s, a, ds, da of a row live
  with the row's owner and the row softmax is local).
* Columns are renumbered locally: own nodes first ([0, n_own)), then the distinct remote
  neighbours ("halo", sorted by global id, i.e. grouped by owner).  The local graph is an
  n_own x (n_own + n_halo) chunked CSR and runs through the unchanged single-GPU operators.
* Forward exchange: a variable-size all_to_all delivers the halo rows of K (and one more those of
  V) straight into the tail of preallocated extended tensors -- no packing, no concatenation.
  Both are started together (async).  ALL FOUR exchanges of a step now run under compute (round 5):
  - K under the OWN-COLUMN half of the SDDMM forward.  Inside every row of the local CSR the own
    columns come first, then the halo columns; the constructor cuts the row-major slots into an
    own-column and a halo-column sub-graph -- two (row, indptr, eid, indices) sets whose `eid` name
    positions of the ONE shared score array s.  Every edge score is written exactly once
    (graphop_kernel.cu:45-52), so the two halves need no zero fill and no add
    (graphop_maskedmm_csr_forward_partial, include/graphop_hip.h): the own half (~90 % of the
    edges at cut = 0.1) is launched right after the K exchange is STARTED, wait_k.wait(), then the
    halo half.  (Round 4 left this exchange exposed with the argument that halves cost "more zero
    fills and adds than they hide" -- true of SpMM-type outputs, not of the SDDMM.)
  - V under the SDDMM and the softmax, the returning dV rows under the softmax / SDDMM backward, the
    returning dK rows under the row-major half of the SDDMM backward (the op is called once per
    orientation).
  `pack_kv`: K and V halo rows can travel as ONE grouped exchange (torch.distributed.
  batch_isend_irecv: one RCCL group launch whose receives land directly in the two extended
  tensors; 3 collectives per step instead of 4, one wait).  A group completes as a whole, so the
  halo half of the SDDMM then waits for the V rows as well: worth it while the exchange is
  latency-bound (small halos), not when it is link-bound (DESIGN.md section 6 prices both); the
  default "auto" packs below PACK_KV_MAX_BYTES per exchange.  dK | dV cannot be packed: dV is ready one
  column-major pass earlier than dK and travels under it.
  Backward exchange: the partial dK / dV rows computed for halo columns (a contiguous slice of the
  operators' outputs) travel back with the transposed split sizes and are added into the owners' rows.
  xGMI is point-to-point: all_to_all drives all 7 links of a GPU at once; no ring collective.
* Index maps are integer and exact: re-assembling the shards reproduces the single-GPU result up
  to fp32 summation order (tests/test_dist.py).

The local operator set is injectable (`ops`): the product default is the HIP path
(custom_op_benchmark_amd.graphop); the CPU tests pass the oracle, since the HIP path has no CPU
implementation.
"""
import torch
import torch.distributed as dist

from . import graphs


def balanced_ranges(out_degree, world):
    """Contiguous node ranges with ~equal edge counts: boundaries[p] .. boundaries[p+1]."""
    n = out_degree.numel()
    cum = torch.cumsum(out_degree.to(torch.int64), 0)
    total = int(cum[-1]) if n else 0
    bounds = [0]
    for p in range(1, world):
        target = total * p // world
        b = int(torch.searchsorted(cum, torch.tensor([target], dtype=torch.int64, device=cum.device))[0])
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return bounds


class _Done:
    def wait(self):
        return True


_DONE = _Done()
# pack_kv = "auto": K | V halo rows travel as one grouped exchange while both directions of one exchange stay below
# this many bytes (latency-bound); above it K goes first on its own, because the halo half of the SDDMM waits for it
PACK_KV_MAX_BYTES = 16 << 20


class _Works:
    """wait() on every work object of a grouped exchange (batch_isend_irecv returns one per op, or one per group)."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        # idempotent: K | V share one group handle, awaited once in front of each consumer (a second wait() on a
        # finished gloo send / recv work never returns)
        works, self.works = self.works, []
        for w in works:
            w.wait()
        return True


class LocalGroup:
    """In-process stand-in for a process group: `world` shards live in one process (one Python
    thread each, all on the same device) and exchange rows by device-to-device copies.  This is the
    one-GPU form of SURVEY.md 8e's validation mode ("shards executed sequentially, exchange via
    device copies"): every byte a rank would receive over xGMI is copied from the peer shard's
    tensors, so the re-assembled result IS the distributed result."""

    def __init__(self, world, timeout=300.0):
        import threading
        self.world = world
        self._barrier = threading.Barrier(world, timeout=timeout)
        self._box = [None] * world

    def handle(self, rank):
        return _LocalHandle(self, rank)

    def abort(self):
        self._barrier.abort()

    def exchange_counts(self, rank, counts):
        self._box[rank] = list(counts)
        self._barrier.wait()
        got = [self._box[p][rank] for p in range(self.world)]
        self._barrier.wait()
        return got

    def all_to_all(self, rank, out, inp, out_splits, in_splits):
        """out rows [sum(out_splits[:p]), +out_splits[p]) = the rows peer p addressed to `rank`."""
        self._box[rank] = (inp, list(in_splits))
        self._barrier.wait()
        o = 0
        for p in range(self.world):
            src, splits = self._box[p]
            b = sum(splits[:rank])
            n = splits[rank]
            assert n == out_splits[p], "split sizes of ranks %d and %d disagree" % (rank, p)
            if n:
                out[o:o + n].copy_(src[b:b + n])
            o += n
        if out.is_cuda:
            torch.cuda.current_stream(out.device).synchronize()   # peers may reuse their send buffers after the barrier
        self._barrier.wait()


class _LocalHandle:
    def __init__(self, group, rank):
        self.group, self.rank = group, rank


def run_local_shards(world, fn, timeout=300.0):
    """Run fn(rank, group_handle) for every rank of a LocalGroup on `world` threads; returns the list
    of results.  An exception in one shard aborts the others (no hang at the barrier)."""
    import threading
    grp = LocalGroup(world, timeout)
    out, err = [None] * world, []

    def body(r):
        try:
            out[r] = fn(r, grp.handle(r))
        except BaseException as exc:   # noqa: BLE001 -- re-raised below
            err.append((r, exc))
            grp.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        err.sort(key=lambda e: isinstance(e[1], threading.BrokenBarrierError))
        raise err[0][1]
    return out


class ShardedAttention:
    def __init__(self, rank, world, bounds, src_global, dst_global, device, chunk_size=32, ops=None,
                 group=None, timing_only=False, force_collectives=False, halo_mask=None, split_forward=True,
                 pack_kv="auto"):
        """src_global/dst_global: the edges whose source lies in this rank's range (any order).
        group: a torch.distributed process group (None = default), or a LocalGroup handle (all shards
        in this process, exchanges are device copies: exact results on one GPU).
        timing_only=True builds ONE shard with no peers at all: exchanges become local copies of the
        right sizes, so the shard's kernels see the right shapes and the step can be timed when the
        other shards do not fit next to it (bench.py --emulate-world on papers100M-size shards).
        Values involving halo rows are then meaningless and must not be checked.
        force_collectives=True (or GRAPHOP_DIST_FORCE_COLLECTIVES=1) keeps the real collective calls at
        world == 1 too -- all_to_all_single on a one-rank process group, async handles, split-size views --
        so that the RCCL code path runs on a one-GPU box (tests/test_dist_gpu.py, bench.py --rccl-self);
        halo_mask (bool per edge) then marks edges whose destination is to be FETCHED THROUGH THE EXCHANGE
        although this rank owns it (the rank is its own peer): real bytes move through every exchange and
        the re-assembled result is still exact."""
        self.rank, self.world, self.bounds, self.group = rank, world, list(bounds), group
        self.emulate = timing_only
        import os
        self.force = bool(force_collectives) or os.environ.get("GRAPHOP_DIST_FORCE_COLLECTIVES", "0") == "1"
        self.local = group if isinstance(group, _LocalHandle) else None
        self._buffers = {}
        self.timers = None       # set to {} to collect per-exchange wall times (bench.py)
        # SDDMM backward as two calls (column-major half, then row-major half) so that the dK halo
        # exchange runs under the dQ pass; costs one extra zero fill of a K_ext-sized tensor
        self.split_backward = True
        # SDDMM forward as an own-column half (runs under the K exchange) and a halo-column half (after it)
        self.split_forward = bool(split_forward) and os.environ.get("GRAPHOP_DIST_SPLIT_FORWARD", "1") != "0"
        self.pack_kv = {"0": False, "1": True}.get(os.environ.get("GRAPHOP_DIST_PACK_KV", ""), pack_kv)   # True / False / "auto"
        # the two column-major passes of the backward (dV = SpMM(a, dO), dK = SpMM(ds, Q)) as ONE launch over the column CSR
        # with the weights packed as (E, 2) pairs (graphop_spmm_pair): one 8-byte random read per slot instead of two
        # 4-byte ones in two passes, ids / metadata streamed once.  Both gradient exchanges then start together behind it
        # and share the dQ pass as cover (split form: dV travels under softmax-backward + the dK pass, dK under dQ), so
        # which form is faster depends on the links: autotune() measures both.  HIP path, fp32, one head, d in {64,128,256}.
        self.fuse_columns = os.environ.get("GRAPHOP_DIST_FUSE_COLUMNS", "0") == "1"
        # whether step() USES the forward halves it cut (autotune: the halo half re-reads the Q row of every row that has
        # a halo slot -- ~2 ms at the papers100M shape -- which only pays where there is an exchange to hide)
        self.use_forward_split = True
        self.noop_exchange = False        # bench.py: exchanges do nothing at all (timing only: what a step costs without them)
        self.collectives_last_step = 0
        self.device = torch.device(device)
        self.ops = ops
        lo, hi = bounds[rank], bounds[rank + 1]
        self.lo, self.n_own = lo, hi - lo
        src_global = src_global.to(self.device, torch.int64)
        dst_global = dst_global.to(self.device, torch.int64)
        assert src_global.numel() == 0 or (int(src_global.min()) >= lo and int(src_global.max()) < hi)

        own = (dst_global >= lo) & (dst_global < hi)
        if halo_mask is not None:
            own &= ~halo_mask.to(self.device)
        remote = dst_global[~own]
        halo_ids = torch.unique(remote)                       # sorted global ids of halo nodes
        self.n_halo = int(halo_ids.numel())
        dst_local = torch.empty_like(dst_global)
        dst_local[own] = dst_global[own] - lo
        dst_local[~own] = self.n_own + torch.searchsorted(halo_ids, remote)
        self.halo_ids = halo_ids
        self.graph = graphs.graph_from_coo(src_global - lo, dst_local, self.n_own, self.n_own + self.n_halo,
                                           chunk_size)

        # who owns each halo node -> how many rows we receive from every peer (halo is owner-major)
        b = torch.tensor(bounds, dtype=torch.int64, device=self.device)
        owner = torch.searchsorted(b, halo_ids, right=True) - 1
        self.recv_counts = torch.bincount(owner, minlength=world).tolist()
        # tell every peer which of its rows we need (setup-time exchange of id lists)
        send_counts = self._exchange_counts(self.recv_counts)
        need_local = (halo_ids - b[owner]).contiguous()       # row index inside the owner's range
        serve = torch.empty(sum(send_counts), dtype=torch.int64, device=self.device)
        if timing_only:   # pretend the peers ask for as many of our rows as we ask of theirs
            serve = torch.arange(sum(send_counts), device=self.device) % max(1, self.n_own)
        else:
            self._all_to_all(serve, need_local, send_counts, self.recv_counts)
        self.send_counts = send_counts
        self.serve_rows = serve                                # our rows, grouped by destination peer
        self._serve_groups = None                              # (ptr, rows, pos) of serve_rows, for the HIP add-home kernel
        # every rank must take the SAME exchange form in a step (a grouped send / recv on one rank does not match an
        # all_to_all_single on another): what pack_kv = "auto" decides on is therefore the largest exchange of ANY rank
        self._max_exchange_rows = self._all_reduce_max(max(self.n_halo, int(serve.numel())))
        self.fwd_halves = self._cut_forward_halves(chunk_size) if (self.split_forward and self.n_halo > 0) else None

    def _cut_forward_halves(self, chunk_size):
        """Own-column and halo-column sub-graphs of the row-major local CSR.  A row's slots are sorted by local column
        id, own columns (< n_own) first; each half is (row, indptr, slots, indices, eid_local): a chunked CSR over the
        half's slots in their original order, `slots` = their positions in the full slot array (= the entries of the
        shared score array they write: the `eid` of graphop_maskedmm_csr_forward_partial), `indices` = their extended
        local column ids, `eid_local` = arange (for operator sets that only offer the reference surface)."""
        from .part_csr import partition_csr
        g = self.graph
        own = g.indices_r < self.n_own
        cum = torch.zeros(g.n_edges + 1, dtype=torch.int64, device=own.device)
        torch.cumsum(own, 0, out=cum[1:])
        ip_own = cum[g.indptr_r]
        halves = []
        for mask, ip in ((own, ip_own), (~own, g.indptr_r - ip_own)):
            slots = torch.nonzero(mask).flatten()
            row, ptr_ = partition_csr(ip.contiguous(), chunk_size)
            halves.append(dict(row=row, ptr=ptr_, slots=slots, indices=g.indices_r[slots].contiguous(),
                               eid_local=torch.arange(slots.numel(), dtype=torch.int64, device=slots.device)))
        return halves

    # ---- collectives ---------------------------------------------------------------------------
    def _exchange_counts(self, counts):
        t_in = torch.tensor(counts, dtype=torch.int64, device=self.device)
        t_out = torch.empty_like(t_in)
        if (self.world == 1 and not self.force) or self.emulate:
            return counts
        if self.local is not None:
            return self.local.group.exchange_counts(self.rank, counts)
        if t_in.is_cuda and dist.get_backend(self.group) == "gloo":
            t_in, t_out = t_in.cpu(), t_out.cpu()
        dist.all_to_all_single(t_out, t_in, group=self.group)
        return t_out.tolist()

    def _all_reduce_max(self, v):
        """max of an int over the ranks (setup-time collective; the local value without peers)."""
        if (self.world == 1 and not self.force) or self.emulate:
            return int(v)
        if self.local is not None:
            return max(self.local.group.exchange_counts(self.rank, [int(v)] * self.world))
        t = torch.tensor([int(v)], dtype=torch.int64, device=self.device)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _timed(self, name, fn):
        """Wall time of one exchange incl. device completion (only when self.timers is a dict: the
        syncs it needs defeat the overlap, so bench.py measures exchanges in a separate pass)."""
        if self.timers is None:
            return fn()
        import time
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        r = fn()
        r.wait()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.timers.setdefault(name, []).append(time.perf_counter() - t0)
        return _DONE

    def _all_to_all(self, out, inp, out_splits, in_splits, async_op=False):
        """Variable-size all-to-all (splits count rows).  Returns a handle whose wait() orders the
        current stream (RCCL) / the caller (gloo) after the exchange; already complete unless
        async_op was requested on a real process group."""
        if self.noop_exchange:
            return _DONE
        self.collectives_last_step += 1
        if self.world == 1 and not self.force:
            out.copy_(inp)
            return _DONE
        if self.emulate:
            n = min(out.shape[0], inp.shape[0])
            out[:n].copy_(inp[:n])
            if out.shape[0] > n:
                out[n:].zero_()
            return _DONE
        if self.local is not None:
            self.local.group.all_to_all(self.rank, out, inp, out_splits, in_splits)
            return _DONE
        if out.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through host memory
            o, i = out.cpu(), inp.cpu()
            dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            out.copy_(o)
            return _DONE
        work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)
        return work if async_op else _DONE

    def _all_to_all_group(self, pairs, out_splits, in_splits, async_op=False):
        """Several variable-size all-to-alls with the SAME split sizes as ONE grouped exchange: pairs = [(out, inp), ...].
        On a real process group this is torch.distributed.batch_isend_irecv -- under RCCL one ncclGroupStart/End launch
        whose receives land directly in the `out` tensors (which is also what all_to_all_single is made of); everywhere
        else (no peers, emulation, LocalGroup, gloo staging of device tensors) one all-to-all per pair."""
        if self.noop_exchange:
            return _DONE
        real = not ((self.world == 1 and not self.force) or self.emulate or self.local is not None)
        if real and not (pairs[0][0].is_cuda and dist.get_backend(self.group) == "gloo"):
            nccl = dist.get_backend(self.group) == "nccl"
            ops, oo, io = [], 0, 0
            for p in range(self.world):
                for out, inp in pairs:               # per peer: K send, K recv, V send, V recv -- the same order on every rank
                    src_seg, dst_seg = inp[io:io + in_splits[p]], out[oo:oo + out_splits[p]]
                    if p == self.rank and not nccl:  # (gloo has no send-to-self; RCCL matches it inside the group)
                        dst_seg.copy_(src_seg)
                        continue
                    if in_splits[p]:
                        ops.append(dist.P2POp(dist.isend, src_seg, group=self.group, group_peer=p))
                    if out_splits[p]:
                        ops.append(dist.P2POp(dist.irecv, dst_seg, group=self.group, group_peer=p))
                oo += out_splits[p]
                io += in_splits[p]
            self.collectives_last_step += 1
            if not ops:
                return _DONE
            w = _Works(dist.batch_isend_irecv(ops))
            if not async_op:
                w.wait()
                return _DONE
            return w
        return _Works([self._all_to_all(o, i, out_splits, in_splits, async_op) for o, i in pairs])

    def _pack_kv_now(self, row_bytes):
        if self.pack_kv == "auto":      # (decided from the largest exchange of any rank: the same answer on every rank)
            return self._max_exchange_rows * row_bytes <= PACK_KV_MAX_BYTES
        return bool(self.pack_kv)

    def gather_halos_grouped(self, pairs, async_op=False, role="KV"):
        """gather_halo_into for several (X_own, X_ext) pairs as one grouped exchange (K | V)."""
        n_own = self.n_own
        bufs = []
        for j, (X_own, X_ext) in enumerate(pairs):
            if not (X_own.data_ptr() == X_ext.data_ptr() and X_own.is_contiguous()):
                X_ext[:n_own].copy_(X_own)
            bufs.append((X_ext[n_own:], self._pack(X_own, role + str(j))))
        return self._timed("halo_" + role, lambda: self._all_to_all_group(bufs, self.recv_counts, self.send_counts, async_op))

    def gather_halo_into(self, X_own, X_ext, async_op=False, role="x"):
        """X_ext[:n_own] = X_own; X_ext[n_own:] = rows of X for the halo nodes, fetched from their
        owners straight into the tail of the (preallocated) extended tensor -- no concatenation.
        With async_op the exchange is only started: call wait() on the returned handle before the
        halo rows are read."""
        n_own = self.n_own
        if not (X_own.data_ptr() == X_ext.data_ptr() and X_own.is_contiguous()):
            X_ext[:n_own].copy_(X_own)      # (no copy when the caller keeps its rows in the extended buffer: own_rows_view)
        send = self._pack(X_own, role)
        return self._timed("halo_" + role, lambda: self._all_to_all(X_ext[n_own:], send, self.recv_counts,
                                                                   self.send_counts, async_op))

    def _pack(self, X_own, role):
        """Rows peers gather from -> persistent contiguous send buffer (HIP pack kernel on the GPU)."""
        key = ("send", role, tuple(X_own.shape[1:]), X_own.dtype)
        buf = self._buffers.get(key)
        if buf is None:
            buf = X_own.new_empty((int(self.serve_rows.numel()),) + tuple(X_own.shape[1:]))
            self._buffers[key] = buf
        if X_own.is_cuda and self.ops is None:
            from . import _lib
            return _lib.gather_rows(X_own.contiguous(), self.serve_rows, out=buf)
        torch.index_select(X_own, 0, self.serve_rows, out=buf)
        return buf

    def _add_home(self, dX_own, recv):
        """Partial gradient rows that came home += into the owners' rows (serve_rows may repeat)."""
        if dX_own.is_cuda and self.ops is None and dX_own.is_contiguous():
            from . import _lib
            # received rows grouped by the own row they belong to (built once): one launch for all peers, every own
            # row read and written once, no atomics, fixed summation order (round 3: one plain-add launch per peer)
            if self._serve_groups is None:
                self._serve_groups = _lib.group_rows(self.serve_rows)
            _lib.add_rows_grouped(dX_own, self._serve_groups, recv)
        else:
            dX_own.index_add_(0, self.serve_rows, recv)
        return dX_own

    def scatter_halo_grad_start(self, dX_own, dX_halo, async_op=False, role="x"):
        """Start sending the partial gradient rows of halo nodes back to their owners."""
        key = ("recv_grad", role, tuple(dX_own.shape[1:]), dX_own.dtype)
        recv = self._buffers.get(key)
        if recv is None:
            recv = dX_own.new_empty((int(self.serve_rows.numel()),) + tuple(dX_own.shape[1:]))
            self._buffers[key] = recv
        send = dX_halo.contiguous()
        handle = self._timed("grad_" + role, lambda: self._all_to_all(recv, send, self.send_counts,
                                                                      self.recv_counts, async_op))
        return handle, recv

    def scatter_halo_grad(self, dX_own, dX_halo, role="x"):
        """Send partial gradient rows of halo nodes back to their owners and add them there."""
        handle, recv = self.scatter_halo_grad_start(dX_own, dX_halo, role=role)
        handle.wait()
        return self._add_home(dX_own, recv)

    def own_rows_view(self, name, shape_tail, dtype=torch.float32):
        """The own-row part of the persistent extended buffer `name` ("K" / "V"): a caller that keeps its K / V rows
        there (writes them into this view instead of a tensor of its own) saves the n_own-row copy in front of every
        halo exchange -- step() recognises the view and only fetches the halo rows behind it."""
        like = torch.empty((0,) + tuple(shape_tail), dtype=dtype, device=self.device)
        return self._ext_buffer(name, like)[:self.n_own]

    def _ext_buffer(self, name, like):
        """Reusable (n_own + n_halo, ...) buffer keyed by role, shape tail and dtype."""
        key = (name, tuple(like.shape[1:]), like.dtype)
        buf = self._buffers.get(key)
        if buf is None:
            buf = like.new_empty((self.n_own + self.n_halo,) + tuple(like.shape[1:]))
            self._buffers[key] = buf
        return buf

    # ---- the step --------------------------------------------------------------------------------
    def _ops(self):
        if self.ops is not None:
            return self.ops
        from . import graphop
        return graphop

    def step(self, Q, K, V, dO):
        """fwd+bwd of SDDMM -> row softmax -> SpMM for this rank's rows.
        Q, K, V, dO: (n_own, [h,] d).  Returns dict(o, dQ, dK, dV, s, a) for the own rows / edges.
        If Q/K/V require grad their .grad is set (detached), mirroring functions.attention_step."""
        ops, g = self._ops(), self.graph
        Qd, Kd, Vd = Q.detach().contiguous(), K.detach(), V.detach()
        n_own = self.n_own
        # forward exchange: halo rows of K and of V land directly behind the own rows.  Both are
        # started at once; the V rows are only awaited in front of the SpMM, so that exchange
        # runs under the SDDMM and the softmax.
        K_ext, V_ext = self._ext_buffer("K", Kd), self._ext_buffer("V", Vd)
        self.collectives_last_step = 0
        row_bytes = Kd[0].numel() * Kd.element_size() if Kd.size(0) else 0
        if self._pack_kv_now(row_bytes):
            wait_k = wait_v = self.gather_halos_grouped([(Kd, K_ext), (Vd, V_ext)], async_op=True)
        else:
            wait_k = self.gather_halo_into(Kd, K_ext, async_op=True, role="K")
            wait_v = self.gather_halo_into(Vd, V_ext, async_op=True, role="V")
        a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
        a8 = g.csr_args()
        if self.fwd_halves is not None and self.use_forward_split:
            # own-column half while the K rows travel (it gathers K_ext[:n_own] only), halo-column half behind the wait
            s = self._sddmm_forward_halves(Qd, K_ext, wait_k)
        else:
            wait_k.wait()
            s = ops.maskedmm_csr_forward(*a4, Qd, K_ext)
        a = ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s)
        wait_v.wait()
        if self.ops is None and V_ext.is_cuda:
            o = self._spmm_forward_own_rows(a, V_ext)           # C ABI: n_y = n_own rows, no zero fill of the halo part
        else:
            o = ops.vector_spmm_forward(*a4, a, V_ext)[:n_own]  # reference surface: y = zeros_like(x), rows >= n_own are 0
        if self.fuse_columns and self._columns_fusable(Qd, a):
            dQ, dK, dV = self._backward_fused_columns(Qd, K_ext, V_ext, a, dO.detach().contiguous())
            for t, gr in ((Q, dQ), (K, dK), (V, dV)):
                if t.requires_grad:
                    t.grad = gr
            from . import _lib
            _lib.check_errors(sync=False)
            return dict(o=o, dQ=dQ, dK=dK, dV=dV, s=s, a=a)
        # dy is only indexed by row ids (< n_own): no need to pad it to the extended row count
        da, dV_ext = ops.vector_spmm_backward(*a8, a, dO.detach().contiguous(), V_ext)
        # backward exchange: partial rows computed for halo columns go home and are added there;
        # the dV rows travel while the softmax and SDDMM backward run
        exchange = self.n_halo or self.world > 1 or self.force
        dV = dV_ext[:n_own]                                     # views: updated in place
        if exchange:
            wait_dv, recv_dv = self.scatter_halo_grad_start(dV, dV_ext[n_own:], async_op=True, role="dV")
        ds = ops.sparse_softmax_backward(g.row, g.ptr_r, g.eid_r, a, da)
        if exchange and self.split_backward:
            # column-major half first (dK incl. the halo columns' partial rows), start sending those
            # home, then the row-major half (dQ) runs under that exchange.  Each half is the same
            # entry point with the other orientation's chunk list empty.
            er, ep = self._empty_chunks()
            hip = self.ops is None and K_ext.is_cuda
            if hip:
                dK_ext = self._sddmm_backward_half(Qd, K_ext, ds, col_half=True)
            else:
                _, dK_ext = ops.maskedmm_csr_backward(er, ep, g.eid_r, g.indices_r, g.col, g.ptr_c, g.eid_c,
                                                      g.indices_c, Qd, K_ext, ds)
            dK = dK_ext[:n_own]
            wait_dk, recv_dk = self.scatter_halo_grad_start(dK, dK_ext[n_own:], async_op=True, role="dK")
            if hip:
                dQ = self._sddmm_backward_half(Qd, K_ext, ds, col_half=False)
            else:
                dQ, _ = ops.maskedmm_csr_backward(g.row, g.ptr_r, g.eid_r, g.indices_r, er, ep, g.eid_c,
                                                  g.indices_c, Qd, K_ext, ds)
            wait_dk.wait()
            self._add_home(dK, recv_dk)
            wait_dv.wait()
            self._add_home(dV, recv_dv)
        else:
            dQ, dK_ext = ops.maskedmm_csr_backward(*a8, Qd, K_ext, ds)
            dK = dK_ext[:n_own]
            if exchange:
                self.scatter_halo_grad(dK, dK_ext[n_own:], role="dK")
                wait_dv.wait()
                self._add_home(dV, recv_dv)
        for t, gr in ((Q, dQ), (K, dK), (V, dV)):
            if t.requires_grad:
                t.grad = gr
        if self.ops is None and dQ.is_cuda:
            from . import _lib
            _lib.check_errors(sync=False)     # a device-side abort of a finished launch is raised before the gradients leave
        return dict(o=o, dQ=dQ, dK=dK, dV=dV, s=s, a=a)

    def _columns_fusable(self, Q, a):
        if self.ops is not None or not Q.is_cuda or Q.dim() != 2 or a.dim() != 1 or Q.dtype != torch.float32:
            return False
        from . import _lib
        g = self.graph
        plan_c = _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c, Q.size(0))
        return bool(_lib.lib().graphop_spmm_pair_supported(_lib.F32, g.col.size(0), g.n_edges, Q.size(0), 1, Q.size(-1),
                                                           plan_c.handle))

    def _backward_fused_columns(self, Q, K_ext, V_ext, a, dO):
        """Backward with ONE column-major launch: da (row-major SDDMM) -> ds (softmax backward) -> (dV | dK) over the
        column CSR with packed (a, ds) weights -> both gradient exchanges -> dQ (row-major) under them."""
        from . import _lib
        g, n_own, L = self.graph, self.n_own, _lib.lib()
        er, ep = self._empty_chunks()
        d = Q.size(-1)
        st = _lib.stream_of(Q)
        with _lib.device_guard(Q.device):
            plan_r = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, V_ext.size(0))
            plan_c = _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c, Q.size(0))
            da = torch.empty_like(a)
            # row-major half of vector_spmm_backward only: da = SDDMM(dO, V_ext); no dx, no column chunks
            _lib.check(L.graphop_vector_spmm_backward(
                _lib.F32, _lib.ptr(g.row), _lib.ptr(g.ptr_r), _lib.ptr(g.eid_r), _lib.ptr(g.indices_r), _lib.ptr(er), _lib.ptr(ep),
                _lib.ptr(g.eid_c), _lib.ptr(g.indices_c), _lib.ptr(a), _lib.ptr(dO), _lib.ptr(V_ext), _lib.ptr(da), None,
                g.row.size(0), 0, g.n_edges, V_ext.size(0), dO.size(0), 1, d, plan_r.handle, None, st))
        ds = self._ops().sparse_softmax_backward(g.row, g.ptr_r, g.eid_r, a, da)
        del da
        # (a, ds) as (E, 2) pairs: a slot's two weights are ONE 8-byte random read in the column pass (scattering them into
        # the column CSR's slot order first so that they stream was built and measured: the 200 M scattered 8-byte stores
        # cost 8.3 ms for a 5 ms faster pass -- profiles/r5_pair_columns_experiment.txt)
        w2 = torch.empty((g.n_edges, 2), dtype=a.dtype, device=a.device)
        dV_ext, dK_ext = torch.empty_like(V_ext), torch.empty_like(K_ext)
        with _lib.device_guard(Q.device):
            _lib.check(L.graphop_interleave_pairs(_lib.F32, _lib.ptr(a), _lib.ptr(ds), _lib.ptr(w2), g.n_edges, st))
            _lib.check(L.graphop_spmm_pair(
                _lib.F32, _lib.ptr(g.col), _lib.ptr(g.ptr_c), _lib.ptr(g.eid_c), _lib.ptr(g.indices_c), _lib.ptr(w2), _lib.ptr(dO),
                _lib.ptr(Q), _lib.ptr(dV_ext), _lib.ptr(dK_ext), g.col.size(0), g.n_edges, Q.size(0), K_ext.size(0), 1, d,
                plan_c.handle, st))
        del w2
        dV, dK = dV_ext[:n_own], dK_ext[:n_own]
        exchange = self.n_halo or self.world > 1 or self.force
        if exchange:
            wait_dv, recv_dv = self.scatter_halo_grad_start(dV, dV_ext[n_own:], async_op=True, role="dV")
            wait_dk, recv_dk = self.scatter_halo_grad_start(dK, dK_ext[n_own:], async_op=True, role="dK")
        dQ = self._sddmm_backward_half(Q, K_ext, ds, col_half=False)
        if exchange:
            wait_dv.wait()
            self._add_home(dV, recv_dv)
            wait_dk.wait()
            self._add_home(dK, recv_dk)
        return dQ, dK, dV

    def autotune(self, Q, K, V, dO, steps=2, candidates=None):
        """Measure, don't guess: the step under every schedule the shard supports -- K | V halo rows as one grouped
        exchange or two, the column-major backward passes as one launch or two, the SDDMM forward as own / halo halves
        (K exchange hidden, Q rows of the halo half re-read) or whole -- `steps` steps each after one warm-up,
        wall time between barriers, MAX over ranks; the fastest is adopted by EVERY rank (the choice is made from the
        reduced times, so all ranks agree).  Which one wins depends on the links: packing delays the halo half of the
        SDDMM until V has arrived too, fusing the columns leaves both gradient exchanges only the dQ pass as cover.
        -> {schedule: ms}."""
        import time
        cuda = self.device.type == "cuda"
        real = self.world > 1 and self.local is None and not self.emulate
        # the candidate list must be the SAME on every rank (each candidate is a fixed number of collective steps): what a
        # rank supports is reduced over the ranks first
        fusable = self._columns_fusable(Q.detach().contiguous(), torch.empty(0, device=self.device))
        fusable = bool(-self._all_reduce_max(0 if fusable else 1) + 1)        # AND over the ranks
        exchanging = bool(self._all_reduce_max(1 if (self.n_halo > 0 or self.force) else 0))
        splittable = bool(-self._all_reduce_max(0 if self.fwd_halves is not None else 1) + 1)
        fus = [False] + ([True] if fusable else [])
        packs = [False, True] if exchanging else [False]
        splits = [True, False] if splittable else [False]
        cands = candidates or [(p, f, sp) for sp in splits for f in fus for p in packs]

        def sync():
            if real:
                dist.barrier(group=self.group)
            if cuda:
                torch.cuda.synchronize(self.device)

        times = {}
        for cand in cands:
            pack, fuse = cand[0], cand[1]
            split = cand[2] if len(cand) > 2 else self.use_forward_split
            self.pack_kv, self.fuse_columns, self.use_forward_split = pack, fuse, split
            self.step(Q, K, V, dO)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(Q, K, V, dO)
            sync()
            times[(pack, fuse, split)] = (time.perf_counter() - t0) / steps
        keys = sorted(times)
        t = torch.tensor([times[k] for k in keys], dtype=torch.float64,
                         device=self.device if (real and dist.get_backend(self.group) == "nccl") else "cpu")
        if real:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        best = keys[int(torch.argmin(t))]
        self.pack_kv, self.fuse_columns, self.use_forward_split = best
        return {"kv_%s+columns_%s+forward_%s" % ("packed" if k[0] else "separate", "fused" if k[1] else "split",
                                                  "split" if k[2] else "whole"): round(1e3 * float(v), 3)
                for k, v in zip(keys, t.tolist())}

    def _sddmm_forward_halves(self, Q, K_ext, wait_k):
        """s = SDDMM(Q, K_ext) as two launches over disjoint slot sets that share s: own columns first (no halo row is
        read: runs under the K exchange), then wait, then the halo columns.  Every score is written once: no fill, no add."""
        g, (own, halo) = self.graph, self.fwd_halves
        h = Q.size(1) if Q.dim() == 3 else 1
        s = Q.new_empty((g.n_edges,) if h == 1 else (g.n_edges, h))
        if self.ops is None and Q.is_cuda:
            from . import _lib
            L = _lib.lib()
            with _lib.device_guard(Q.device):
                for half in (own, halo):
                    if half is halo:
                        wait_k.wait()
                    _lib.check(L.graphop_maskedmm_csr_forward_partial(
                        _lib.dtype_code(Q), _lib.ptr(half["row"]), _lib.ptr(half["ptr"]), _lib.ptr(half["slots"]),
                        _lib.ptr(half["indices"]), _lib.ptr(Q), _lib.ptr(K_ext), _lib.ptr(s), half["row"].size(0),
                        half["slots"].size(0), g.n_edges, Q.size(0), K_ext.size(0), h, Q.size(-1), _lib.stream_of(Q)))
            return s
        # reference surface (fresh output per call): each half computes its own score array, copied to its slots
        ops = self._ops()
        for half in (own, halo):
            if half is halo:
                wait_k.wait()
            if half["slots"].numel():
                s[half["slots"]] = ops.maskedmm_csr_forward(half["row"], half["ptr"], half["eid_local"], half["indices"], Q, K_ext)
        return s

    # ---- C-ABI calls the reference's Python surface cannot express (HIP path only) -----------------
    def _spmm_forward_own_rows(self, a, V_ext):
        """vector_spmm_forward with n_y = n_own output rows (the reference returns zeros_like(x):
        n_own + n_halo rows, of which the halo part is only ever zero-filled)."""
        from . import _lib
        g = self.graph
        h = a.size(1) if a.dim() == 2 else 1
        o = V_ext.new_empty((self.n_own,) + tuple(V_ext.shape[1:]))
        with _lib.device_guard(V_ext.device):
            plan = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, V_ext.size(0))
            _lib.check(_lib.lib().graphop_vector_spmm_forward(
                _lib.dtype_code(V_ext), _lib.ptr(g.row), _lib.ptr(g.ptr_r), _lib.ptr(g.eid_r), _lib.ptr(g.indices_r),
                _lib.ptr(a), _lib.ptr(V_ext), _lib.ptr(o), g.row.size(0), g.eid_r.size(0), V_ext.size(0), self.n_own,
                h, V_ext.size(-1), plan.handle, _lib.stream_of(V_ext)))
        return o

    def _sddmm_backward_half(self, Q, K_ext, ds, col_half):
        """One orientation of maskedmm_csr_backward: the other output is passed as NULL with an empty
        chunk list, so it is neither allocated nor zero-filled."""
        from . import _lib
        g = self.graph
        er, ep = self._empty_chunks()
        ds = ds.contiguous()
        h = ds.size(1) if ds.dim() == 2 else 1
        L = _lib.lib()
        with _lib.device_guard(Q.device):
            # the skipped orientation gets NO plan (NULL): a plan over the empty chunk list would still
            # validate all E slots and allocate 32-bit mirrors of the slot arrays that no kernel reads
            if col_half:
                out = torch.empty_like(K_ext)
                plan_r = None
                plan_c = _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c, Q.size(0)).handle
                row, ptr_r, col, ptr_c, dA, dB = er, ep, g.col, g.ptr_c, None, out
            else:
                out = torch.empty_like(Q)
                plan_r = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, K_ext.size(0)).handle
                plan_c = None
                row, ptr_r, col, ptr_c, dA, dB = g.row, g.ptr_r, er, ep, out, None
            _lib.check(L.graphop_maskedmm_csr_backward(
                _lib.dtype_code(Q), _lib.ptr(row), _lib.ptr(ptr_r), _lib.ptr(g.eid_r), _lib.ptr(g.indices_r),
                _lib.ptr(col), _lib.ptr(ptr_c), _lib.ptr(g.eid_c), _lib.ptr(g.indices_c), _lib.ptr(Q), _lib.ptr(K_ext),
                _lib.ptr(ds), _lib.ptr(dA), _lib.ptr(dB), row.size(0), col.size(0), g.eid_r.size(0),
                Q.size(0), K_ext.size(0), h, Q.size(-1), plan_r, plan_c, _lib.stream_of(Q)))
        return out

    def _empty_chunks(self):
        """(row[0], indptr[1] = [0]): a chunk list that covers nothing (int64, on this shard's device)."""
        e = self._buffers.get("empty_chunks")
        if e is None:
            e = (torch.zeros(0, dtype=torch.int64, device=self.device),
                 torch.zeros(1, dtype=torch.int64, device=self.device))
            self._buffers["empty_chunks"] = e
        return e

    def halo_stats(self, row_bytes):
        """What this rank moves per step: rows / bytes received (K and V forward) and sent back
        (dK, dV), for bench.py's config line."""
        return {"n_own": self.n_own, "n_halo": self.n_halo, "recv_rows_per_peer": list(self.recv_counts),
                "send_rows_per_peer": list(self.send_counts),
                "bytes_per_exchange_in": self.n_halo * row_bytes,
                "bytes_per_exchange_out": int(self.serve_rows.numel()) * row_bytes,
                "exchanges_per_step": 4, "collectives_last_step": self.collectives_last_step,
                "kv_packed": self._pack_kv_now(row_bytes), "forward_split": self.fwd_halves is not None and self.use_forward_split,
                "columns_fused": bool(self.fuse_columns)}

    # ---- builders --------------------------------------------------------------------------------
    @classmethod
    def from_global_coo(cls, src, dst, n_nodes, rank, world, device, chunk_size=32, ops=None, group=None,
                        force_collectives=False, halo_mask=None, **kw):
        """Every rank holds the full edge list (small graphs / tests); ranges balanced by edges."""
        deg = torch.bincount(src.to(torch.int64), minlength=n_nodes)
        bounds = balanced_ranges(deg, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        m = (src >= lo) & (src < hi)
        return cls(rank, world, bounds, src[m], dst[m], device, chunk_size, ops, group,
                   force_collectives=force_collectives, halo_mask=None if halo_mask is None else halo_mask.to(m.device)[m],
                   **kw)

    @classmethod
    def synthetic(cls, n_per_rank, e_per_rank, world, rank, device, alpha=0.5, seed=0, chunk_size=32,
                  ops=None, group=None, timing_only=False, cut=1.0, force_collectives=False, self_halo=False, **kw):
        """Weak-scaling bench graph: `world` equal node ranges of a Chung-Lu graph with
        world*n_per_rank nodes; each rank draws the e_per_rank edges of its own rows on its own
        device.  Sources come from its range; a destination comes from the GLOBAL weight vector with
        probability `cut` and from the rank's own range otherwise.  cut = 1 is a graph without any
        locality (every rank's halo is nearly every remote node: the worst case for a node-range
        partition); cut ~ 0.1 is what a locality-aware partition of a citation graph leaves
        (stated in bench.py's config line).  self_halo=True (with force_collectives, world == 1 rehearsals of the
        RCCL path): the destinations drawn from the global distribution are fetched through the exchange even
        where the rank owns them."""
        n_total = n_per_rank * world
        w = graphs.powerlaw_weights(n_total, alpha, seed, device)
        bounds = [p * n_per_rank for p in range(world + 1)]
        lo, hi = bounds[rank], bounds[rank + 1]
        cdf_all = torch.cumsum(w, 0); cdf_all[-1] = 1.0
        w_own = w[lo:hi] / w[lo:hi].sum()
        cdf_own = torch.cumsum(w_own, 0); cdf_own[-1] = 1.0
        gen = torch.Generator(device=device).manual_seed(seed + 1000 * (rank + 1))
        srcs, dsts, glob = [], [], []
        batch = 1 << 26
        for s0 in range(0, e_per_rank, batch):
            m = min(batch, e_per_rank - s0)
            u = torch.rand(m, generator=gen, device=device, dtype=torch.float64)
            srcs.append(lo + torch.searchsorted(cdf_own, u).clamp_(max=n_per_rank - 1))
            u = torch.rand(m, generator=gen, device=device, dtype=torch.float64)
            d_glob = torch.searchsorted(cdf_all, u).clamp_(max=n_total - 1)
            if cut < 1.0:
                u = torch.rand(m, generator=gen, device=device, dtype=torch.float64)
                d_own = lo + torch.searchsorted(cdf_own, u).clamp_(max=n_per_rank - 1)
                keep = torch.rand(m, generator=gen, device=device) < cut
                d_glob = torch.where(keep, d_glob, d_own)
                if self_halo:
                    glob.append(keep)
            elif self_halo:
                glob.append(torch.ones(m, dtype=torch.bool, device=device))
            dsts.append(d_glob)
        return cls(rank, world, bounds, torch.cat(srcs), torch.cat(dsts), device, chunk_size, ops, group,
                   timing_only, force_collectives=force_collectives,
                   halo_mask=torch.cat(glob) if self_halo else None, **kw)

    @classmethod
    def synthetic_rmat(cls, scale, e_per_rank, world, rank, device, seed=0, chunk_size=32, ops=None,
                       group=None, timing_only=False, **kw):
        """High-degree-row stress graph (BASELINE.json config 5): equal node ranges of an R-MAT graph
        on 2**scale nodes; each rank draws e_per_rank edges whose sources lie in its range
        (graphs.rmat_edges with the range's bit prefix).  NB an R-MAT graph cut into equal node
        ranges is not edge-balanced (range 0 holds 0.76**log2(world) of the edges); weak scaling
        needs equal work per rank, so every rank draws the same number of edges."""
        bits = (world - 1).bit_length()
        assert world == 1 << bits and bits <= scale, "synthetic_rmat needs a power-of-two world"
        n_per_rank = (1 << scale) >> bits
        bounds = [p * n_per_rank for p in range(world + 1)]
        src, dst = graphs.rmat_edges(scale, e_per_rank, seed + 7919 * (rank + 1), device, src_prefix_bits=bits,
                                     src_prefix=rank)
        return cls(rank, world, bounds, src, dst, device, chunk_size, ops, group, timing_only, **kw)
