"""Read sharding and result gather for N GPUs of one node (one process per GPU, torch.distributed).

The path partitions by reads (they are independent, src/main/taxor_search.cpp:214): the index is replicated,
rank r classifies its own shard, and the only exchange step is the gather of the per-read results on rank 0.
With backend "nccl" (= RCCL on ROCm) the gather is point-to-point send/recv, one xGMI link per peer, after an
all_gather of the sizes; the same code runs on CPU tensors with "gloo" (tests)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_reads, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: concatenating rank outputs restores input order."""
    per = (n_reads + world - 1) // world
    lo = min(rank * per, n_reads)
    return lo, min(lo + per, n_reads)


# what the last gather_csr on this rank moved (dst only): per-rank (reads, tuples) as announced AND received, payload bytes that
# arrived from peers -- bench.py's `comm` object is written from this, not from WORLD_SIZE
last_gather = {"sizes": [], "bytes_received": 0}


def gather_csr(read_off, user_bin, count, n_hashes, dst=0):
    """Gather every rank's CSR results on `dst`.  Inputs are 1-D torch tensors on the rank's device:
    read_off int64[n+1] (rank-local offsets), user_bin int64[t], count int32[t], n_hashes int32[n].
    Returns on dst the concatenated (read_off, user_bin, count, n_hashes) in rank order with offsets
    rebased; on other ranks None."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = read_off.device
    n, t = n_hashes.numel(), user_bin.numel()
    sizes = torch.tensor([n, t], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    if world == 1:
        last_gather["sizes"], last_gather["bytes_received"] = [(n, t)], 0
        return read_off, user_bin, count, n_hashes
    mine = [read_off.contiguous(), user_bin.contiguous(), count.contiguous(), n_hashes.contiguous()]
    if rank != dst:
        ops = [dist.P2POp(dist.isend, b, dst) for b in mine if b.numel()]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    parts = {dst: mine}
    ops = []
    for p in range(world):
        if p == dst:
            continue
        pn, pt = int(all_sizes[p][0]), int(all_sizes[p][1])
        bufs = [torch.empty(pn + 1, dtype=torch.int64, device=dev), torch.empty(pt, dtype=torch.int64, device=dev),
                torch.empty(pt, dtype=torch.int32, device=dev), torch.empty(pn, dtype=torch.int32, device=dev)]
        parts[p] = bufs
        ops += [dist.P2POp(dist.irecv, b, p) for b in bufs if b.numel()]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    last_gather["sizes"] = [(int(parts[p][3].numel()), int(parts[p][1].numel())) for p in range(world)]
    last_gather["bytes_received"] = sum(b.numel() * b.element_size() for p in range(world) if p != dst for b in parts[p])
    offs, base = [torch.zeros(1, dtype=torch.int64, device=dev)], 0
    for p in range(world):
        ro = parts[p][0]
        offs.append(ro[1:] + base)
        base += int(ro[-1])
    return (torch.cat(offs), torch.cat([parts[p][1] for p in range(world)]),
            torch.cat([parts[p][2] for p in range(world)]), torch.cat([parts[p][3] for p in range(world)]))


def results_to_torch(res, device="cpu"):
    """SearchResults (numpy) -> the four tensors gather_csr takes."""
    return (torch.from_numpy(res.read_off.astype(np.int64)).to(device), torch.from_numpy(res.user_bin.copy()).to(device),
            torch.from_numpy(res.count.astype(np.int32)).to(device), torch.from_numpy(res.n_hashes.astype(np.int32)).to(device))
