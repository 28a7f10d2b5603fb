"""Data-parallel sharding of the fit problem over GPUs (one process per GPU).

LD blocks are independent and everything else is per-SNP, so the SNP set is split into
shards that are closed under "shares an LD block in any cohort" (connected components of the
block structure across cohorts -- cohorts may have different block partitions and different
`perm`s, reference vi_options.py:161-183).  The only cross-shard quantities are the small sums
of include/vilma_hip.h (`totals`, `delta_sums`, convergence statistics); `Comm` all-reduces
them with torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests).
"""
import numpy as np


def init_distributed_from_env():
    """Under a multi-process launcher (torchrun: WORLD_SIZE > 1) join the process group -- one
    rank per GPU, backend nccl (= RCCL) -- and bind this process to its GPU.  Does nothing for a
    plain single-process run or when the caller has already initialised torch.distributed.
    VILMA_DIST_BACKEND=gloo and VILMA_SAME_DEVICE=1 exist for rehearsals on a one-GPU box."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return True
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('VILMA_SAME_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('VILMA_DIST_BACKEND', 'nccl')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend)
    return True


def agree_on_rccl(comm, make_id, init_rccl):
    """Set up one RCCL communicator per context on every rank, or on none: all ranks leave with
    the same answer, and no rank is ever left waiting inside ncclCommInitRank for a rank that
    cannot take part.

    make_id() -> the 128 bytes of ncclGetUniqueId, or None when this rank cannot even load RCCL;
    init_rccl(id_bytes) -> None, raising on failure (ncclCommInitRank on this rank's context).
    Returns (ok, reason, n_failed): ok on every rank or on none."""
    ident = make_id()
    have_id = ident is not None and len(ident) == 128
    # every rank makes an id (only rank 0's is used): a rank that cannot load RCCL must be known
    # BEFORE the others enter ncclCommInitRank, which would wait for it forever
    cannot = int(round(float(comm.allreduce_np(np.array([0.0 if have_id else 1.0]))[0])))
    # rank 0 always broadcasts (an empty id = "not everybody can"), so nobody waits forever
    raw = comm.broadcast_bytes(bytes(ident) if (have_id and cannot == 0) else b'')
    err = None
    if len(raw) != 128:
        err = 'RCCL is not loadable on %d rank(s) (librccl.so?)' % max(cannot, 1)
    else:
        try:
            init_rccl(raw)
        except Exception as exc:        # noqa: BLE001 - reported to every rank below
            err = str(exc)
    failed = int(round(float(comm.allreduce_np(np.array([0.0 if err is None else 1.0]))[0])))
    if failed == 0:
        return True, None, 0
    return False, err or 'another rank failed', failed


def _is_identity(index, n):
    index = np.asarray(index)
    return index.shape == (n,) and (n == 0 or (index[0] == 0 and index[-1] == n - 1
                                               and bool(np.all(index[1:] - index[:-1] == 1))))


class Comm:
    """Thin wrapper over torch.distributed; a no-op for a single process."""

    def __init__(self, group=None, force=False):
        self.rank, self.world = 0, 1
        self._dist = None
        self._force = force
        self.backend = None
        # VILMA_COLLECTIVE=torch: the sweep's all-reduces go through torch.distributed (a callback
        # from the library) instead of the communicator the context owns -- A/B and fallback
        import os
        self.force_callback = os.environ.get('VILMA_COLLECTIVE') == 'torch'
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self._dist = dist
                self.group = group
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
                self.backend = dist.get_backend(group)
        except ImportError:
            pass

    @property
    def active(self):
        """True when collectives actually run: more than one rank, or `force` (a one-rank group
        used to rehearse the RCCL stream ordering on a single GPU)."""
        return self._dist is not None and (self.world > 1 or self._force)

    def broadcast_bytes(self, raw, src=0):
        """Rank `src`'s byte string on every rank (the RCCL unique id of a context-owned
        communicator travels this way)."""
        if self._dist is None or self.world == 1:
            return raw
        box = [raw if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def allreduce(self, tensor, op='sum'):
        """All-reduce a small float64 torch tensor (any device) and return it as numpy."""
        if self.active:
            dist = self._dist
            t = tensor
            if self.backend == 'gloo' and t.is_cuda:
                t = t.cpu()
            elif self.backend == 'nccl' and not t.is_cuda:
                t = t.cuda()
            rop = dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX
            dist.all_reduce(t, op=rop, group=self.group)
            return t.cpu().numpy().copy()
        return tensor.cpu().numpy().copy()

    def allreduce_inplace(self, tensor, op='sum'):
        """All-reduce a (view of a) torch tensor in place, without a host copy when the backend
        can work on its device (nccl/RCCL on GPU tensors, gloo on CPU tensors)."""
        if not self.active:
            return tensor
        dist = self._dist
        rop = dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX
        if (self.backend == 'gloo' and tensor.is_cuda) or (self.backend == 'nccl' and not tensor.is_cuda):
            t = tensor.cpu() if tensor.is_cuda else tensor.cuda()
            dist.all_reduce(t, op=rop, group=self.group)
            tensor.copy_(t)
        else:
            dist.all_reduce(tensor, op=rop, group=self.group)
        return tensor

    def allreduce_np(self, array, op='sum'):
        if not self.active:
            return np.array(array, dtype=np.float64)
        import torch
        return self.allreduce(torch.as_tensor(np.ascontiguousarray(array, dtype=np.float64)), op)

    # temporaries of gather_snps per collective: a [M,P,N] array goes through in slices of its
    # first axis so that no rank ever holds world x (the whole array) beside the result
    GATHER_CHUNK_BYTES = 256 << 20

    def gather_snps(self, local, snp_index, n_global):
        """Assemble an array whose LAST axis is the shard's SNPs into the global SNP order on
        every rank (the class API returns whole arrays on every rank).  `snp_index` = global
        indices of the local SNPs.  Large arrays are gathered slice by slice along their first
        axis: the temporaries stay below GATHER_CHUNK_BYTES x world."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.world == 1:
            if local.shape[-1] == n_global and _is_identity(snp_index, n_global):
                return local            # one rank holding every SNP in order: nothing to move
            out = np.empty(local.shape[:-1] + (n_global,))
            out[..., snp_index] = local
            return out
        import torch
        dist = self._dist
        counts = [None] * self.world
        dist.all_gather_object(counts, int(local.shape[-1]), group=self.group)
        width = max(counts)

        def padded(a, dtype):
            buf = np.zeros(a.shape[:-1] + (width,), dtype=dtype)
            buf[..., :a.shape[-1]] = a
            t = torch.as_tensor(buf)
            return t.cuda() if self.backend == 'nccl' else t

        idx = padded(np.asarray(snp_index, dtype=np.int64), np.int64)
        idxs = [torch.empty_like(idx) for _ in range(self.world)]
        dist.all_gather(idxs, idx, group=self.group)
        where = [idxs[r].cpu().numpy()[:counts[r]] for r in range(self.world)]
        out = np.empty(local.shape[:-1] + (n_global,))

        def gather_into(dst, src):
            mine = padded(src, np.float64)
            bufs = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(bufs, mine, group=self.group)
            for r in range(self.world):
                dst[..., where[r]] = bufs[r].cpu().numpy()[..., :counts[r]]

        lead = local.shape[0] if local.ndim > 1 else 1
        row_bytes = 8 * width * max(1, int(np.prod(local.shape[1:-1])))
        step = lead if local.ndim == 1 else max(1, int(self.GATHER_CHUNK_BYTES // max(1, row_bytes)))
        if local.ndim == 1 or step >= lead:
            gather_into(out, local)
        else:       # the same number of collectives on every rank: lead and width are global
            for lo in range(0, lead, step):
                gather_into(out[lo:lo + step], local[lo:lo + step])
        return out


def block_costs(ld):
    """Bytes one product with each block streams in the form the device will hold it."""
    from .matrix_structures import dense_is_cheaper, dense_bytes_moved, eigen_cost
    out = []
    for m in ld.matrices:
        n = m.shape[0]
        # a block that is not decomposed yet is costed as dense (its rank is unknown)
        r = n if m.is_deferred() else m.u.shape[1]
        out.append(8.0 * (dense_bytes_moved(n) if dense_is_cheaper(n, r) else eigen_cost(n, r)))
    return np.asarray(out)


def plan_shards(ld_mats, num_loci, world, per_snp_cost=0.0):
    """Split SNPs into `world` shards closed under LD-block membership in every cohort.

    Returns a list (one per rank) of dicts: 'snps' (sorted global SNP indices of the shard) and
    'blocks' (per cohort, the indices into ld_mats[p].matrices of the blocks it owns).
    Deterministic, so every rank computes the same plan."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    N = int(num_loci)
    rows, cols = [], []
    for ld in ld_mats:
        n_ld = int(ld.starts[-1])
        if n_ld == 0:
            continue
        sizes = np.diff(ld.starts)
        first = np.repeat(ld.perm[ld.starts[:-1]], sizes)
        rows.append(first)
        cols.append(ld.perm[:n_ld])
    if rows:
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        graph = coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(N, N))
        n_comp, label = connected_components(graph, directed=False)
    else:
        n_comp, label = N, np.arange(N)
    cost = np.bincount(label, minlength=n_comp).astype(np.float64) * per_snp_cost
    block_comp = []
    for ld in ld_mats:
        comp = label[ld.perm[ld.starts[:-1]]] if len(ld.matrices) else np.zeros(0, dtype=int)
        block_comp.append(comp)
        if len(comp):
            np.add.at(cost, comp, block_costs(ld))
    # longest-processing-time greedy: heaviest component to the lightest rank; ties by index
    order = np.lexsort((np.arange(n_comp), -cost))
    load = np.zeros(world)
    owner = np.empty(n_comp, dtype=np.int64)
    for c in order:
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += cost[c]
    snp_owner = owner[label]
    plan = []
    for r in range(world):
        plan.append({
            'snps': np.flatnonzero(snp_owner == r),
            'blocks': [np.flatnonzero(owner[bc] == r) for bc in block_comp],
            'cost': float(load[r]),
        })
    return plan


def local_ld(ld, shard_snps, block_ids, num_loci):
    """(blocks, perm, n_ld) of one cohort restricted to a shard, in local SNP numbering."""
    g2l = np.full(int(num_loci), -1, dtype=np.int64)
    g2l[shard_snps] = np.arange(len(shard_snps))
    mats = [ld.matrices[b] for b in block_ids]
    members = [ld.perm[ld.starts[b]:ld.starts[b + 1]] for b in block_ids]
    covered = (np.concatenate(members) if members else np.zeros(0, dtype=np.int64))
    local_cov = g2l[covered]
    assert np.all(local_cov >= 0), 'shard is not closed under LD-block membership'
    is_cov = np.zeros(len(shard_snps), dtype=bool)
    is_cov[local_cov] = True
    perm = np.concatenate([local_cov, np.flatnonzero(~is_cov)]).astype(np.int64)
    return mats, perm, int(len(local_cov))
