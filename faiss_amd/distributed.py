"""faiss_amd/distributed.py -- one-process-per-GPU database sharding (IndexShards semantics).

The reference shards a database over the GPUs of a node with host threads inside ONE process
(faiss/IndexShards.cpp:135-265: every shard searches all queries, the per-shard top-k are
merged on the host, faiss/utils/Heap.cpp:166-240).  The MI355X-native launch model is one
process per GPU under torch.distributed, so the same algorithm becomes:

    rank r holds rows [r*nb/N, (r+1)*nb/N) of the database (successive id ranges)
    step:  local search (all queries) -> gather of the per-rank (D, I) onto rank 0
           -> k-way merge under the (distance, id) order -> result on rank 0

The only exchange is that gather: nq*k*(4+8) bytes per rank (12 MB at nq=10k, k=100), sent
point-to-point to rank 0 over xGMI by RCCL (``torch.distributed.gather`` on the "nccl" backend
is a set of send/recv pairs, so each peer uses its own direct link).  No all-reduce / ring is
involved.  On CPU the identical code path runs over gloo with a host merge (tests/).

When the database fits one GPU the reference replicates instead (IndexReplicas: every GPU holds everything, the
QUERIES are split) -- ReplicatedSearcher below; its only exchange is the gather of the result blocks.
"""
import numpy as np
import torch
import torch.distributed as dist


def _fence(t):
    """The collective that read or wrote `t` has completed.  A blocking NCCL (RCCL) collective only orders itself against
    torch's CURRENT stream; the library searches on its own stream, so without this the next local search could rewrite
    the per-rank result buffers while the previous gather is still sending them.  (gloo / CPU tensors: nothing to do.)"""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


class ShardedSearcher:
    """IndexShards(successive_ids=True) across the ranks of a torch.distributed group.

    local_search(xq, k) -> (D, I) torch tensors [nq, k] (float32 / int64) on `device`, labels
    local to the shard; merge(all_D, all_I, base) -> (D, I): k-way merge of the gathered
    [world, nq, k] tensors (device merge kernel on GPU, host merge in the gloo tests).
    """

    def __init__(self, local_search, merge, shard_sizes, device, group=None, force_collectives=False):
        self.local_search = local_search
        self.merge = merge
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: run the gather even in a group of ONE rank (tests/test_gpu_rccl_world1.py: the RCCL path --
        # communicator set-up next to the library's own HIP streams, gather, fence, device merge -- on a one-GPU box)
        self.collect = dist.is_initialized() and (self.world > 1 or force_collectives)
        assert len(shard_sizes) == self.world
        # successive_ids: shard s's labels are shifted by the number of rows before it
        # (faiss/IndexShards.cpp:214-219)
        self.base = np.concatenate([[0], np.cumsum(shard_sizes)[:-1]]).astype(np.int64)
        self.device = device
        self._gD = self._gI = None

    def search(self, xq, k):
        D, I = self.local_search(xq, k)
        if not self.collect:
            return self.merge(D.unsqueeze(0), I.unsqueeze(0), self.base)
        nq = D.shape[0]
        if self.rank == 0:
            if self._gD is None or self._gD.shape != (self.world, nq, k):
                self._gD = torch.empty((self.world, nq, k), dtype=torch.float32, device=self.device)
                self._gI = torch.empty((self.world, nq, k), dtype=torch.int64, device=self.device)
            dist.gather(D, list(self._gD.unbind(0)), dst=0, group=self.group)
            dist.gather(I, list(self._gI.unbind(0)), dst=0, group=self.group)
            _fence(self._gI)
            return self.merge(self._gD, self._gI, self.base)
        dist.gather(D, None, dst=0, group=self.group)
        dist.gather(I, None, dst=0, group=self.group)
        _fence(I)
        return None


class ReplicatedSearcher:
    """IndexReplicas across the ranks of a torch.distributed group: every rank holds the WHOLE database and
    searches its block of the queries (faiss/IndexReplicas.cpp:141-164; the default multi-GPU layout of the
    reference, GpuMultipleClonerOptions::shard = false).  The only exchange is the gather of the result blocks
    onto rank 0 -- nq/N * k * 12 bytes per rank, no merge.

    local_search(lo, hi, k) -> (D, I) torch tensors [per, k] on `device` whose first hi-lo rows are the results of
    queries [lo, hi); `per` = replica_bounds(nq, world)[1] (equal block size, so one gather moves everything).
    """

    def __init__(self, local_search, nq, device, group=None, force_collectives=False):
        self.local_search = local_search
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collect = dist.is_initialized() and (self.world > 1 or force_collectives)
        self.nq = nq
        self.bounds, self.per = replica_bounds(nq, self.world)
        self.device = device
        self._gD = self._gI = None

    def search(self, k):
        lo, hi = self.bounds[self.rank]
        D, I = self.local_search(lo, hi, k)
        if not self.collect:
            return D[: self.nq], I[: self.nq]
        assert D.shape == (self.per, k) and I.shape == (self.per, k)
        if self.rank == 0:
            if self._gD is None or self._gD.shape != (self.world, self.per, k):
                self._gD = torch.empty((self.world, self.per, k), dtype=torch.float32, device=self.device)
                self._gI = torch.empty((self.world, self.per, k), dtype=torch.int64, device=self.device)
            dist.gather(D, list(self._gD.unbind(0)), dst=0, group=self.group)
            dist.gather(I, list(self._gI.unbind(0)), dst=0, group=self.group)
            _fence(self._gI)
            # blocks are consecutive query ranges of equal size: the gathered buffer IS the result
            return self._gD.view(-1, k)[: self.nq], self._gI.view(-1, k)[: self.nq]
        dist.gather(D, None, dst=0, group=self.group)
        dist.gather(I, None, dst=0, group=self.group)
        _fence(I)
        return None


def replica_bounds(nq, world, granule=128):
    """Query block of every rank: equal blocks of ceil(nq / world) queries like IndexReplicas::search, rounded up
    to the 128 queries one wavefront of the flat kernel holds.  Returns ([(lo, hi)], block size)."""
    per = -(-nq // world)
    per = -(-per // granule) * granule
    return [(min(nq, r * per), min(nq, (r + 1) * per)) for r in range(world)], per


def shard_bounds(nb, world):
    """Row range of every rank, as IndexShards::add splits them (faiss/IndexShards.cpp:172-175)."""
    return [(r * nb // world, (r + 1) * nb // world) for r in range(world)]


def shard_chunks(rows_per_rank, rank, chunk_rows=1000000):
    """Row chunks of rank `rank` of a database of world x rows_per_rank rows split IndexShards-style (rank r owns the
    global rows [r * rows_per_rank, (r + 1) * rows_per_rank), faiss/IndexShards.cpp:172-175): a list of
    (global chunk number, first global id, rows).  The global chunk number seeds the generator of the chunk, so the union
    of the ranks' rows is ONE well-defined database of world x rows_per_rank rows; the global ids are what add_with_ids
    gets, so the merge needs no label translation (IndexShards with successive_ids = false)."""
    per = -(-rows_per_rank // chunk_rows)
    out, done, c = [], 0, 0
    while done < rows_per_rank:
        n = min(chunk_rows, rows_per_rank - done)
        out.append((rank * per + c, rank * rows_per_rank + done, n))
        done += n
        c += 1
    return out


def broadcast_arrays(arrays, device, src=0, group=None, force=False):
    """float32 numpy arrays of rank `src` -> every rank (shapes known on every rank; the values of the other ranks are
    ignored): the trained coarse quantizer and product-quantizer codebook of a sharded IVF index -- the only collective of
    the layout, once per build (the reference clones one trained CPU index to every device, gpu/GpuCloner.cpp:368-391)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    out = []
    for a in arrays:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        dist.broadcast(t, src, group=group)
        out.append(t.cpu().numpy())
    return out
