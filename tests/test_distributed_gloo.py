"""N>1 path on CPU: world-size-2 (and 3) gloo processes shard a read batch and gather per-read CSR results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from taxor_amd import distributed as td


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_results(n_reads, seed):
    """deterministic per-read tuples: read i has (i*7+seed) % 4 tuples"""
    rng = np.random.default_rng(seed)
    per = np.array([(i * 7 + 3) % 4 for i in range(n_reads)], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
    ub = rng.integers(0, 1000, size=int(off[-1])).astype(np.int64)
    cnt = rng.integers(0, 500, size=int(off[-1])).astype(np.int32)
    nh = rng.integers(0, 900, size=n_reads).astype(np.int32)
    return off, ub, cnt, nh


def _worker(rank, world, port, n_reads, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, ub, cnt, nh = _fake_results(n_reads, 1)           # the whole batch, identical on every rank
    lo, hi = td.shard_range(n_reads, rank, world)
    s_off = torch.from_numpy(off[lo:hi + 1] - off[lo])
    s_ub = torch.from_numpy(ub[off[lo]:off[hi]])
    s_cnt = torch.from_numpy(cnt[off[lo]:off[hi]])
    s_nh = torch.from_numpy(nh[lo:hi])
    out = td.gather_csr(s_off, s_ub, s_cnt, s_nh, dst=0)
    if rank == 0:
        g_off, g_ub, g_cnt, g_nh = out
        ok = (np.array_equal(g_off.numpy(), off) and np.array_equal(g_ub.numpy(), ub)
              and np.array_equal(g_cnt.numpy(), cnt) and np.array_equal(g_nh.numpy(), nh))
        # what the exchange moved, as bench.py's `comm` object reports it: every rank's (reads, tuples) as received, and the payload
        # bytes that came from the peers (the four arrays of every rank but this one)
        spans = [td.shard_range(n_reads, r, world) for r in range(world)]
        want_sizes = [(b - a, int(off[b] - off[a])) for a, b in spans]
        want_bytes = sum((n + 1) * 8 + t * 8 + t * 4 + n * 4 for n, t in want_sizes[1:])
        ok = ok and td.last_gather["sizes"] == want_sizes and td.last_gather["bytes_received"] == want_bytes
        q.put(ok)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_reads", [(2, 101), (2, 1), (3, 50)])
def test_shard_and_gather_gloo(world, n_reads):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            spans = [td.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
