"""The C-ABI communicator (taxor_gpu_comm_*, taxor_amd/csrc/comm.hip): index replication and the per-round gather of the
per-read results.  One GPU is what the usual test box has, so there RCCL runs as a communicator of ONE rank -- but through the
code of a larger run: creation sends known bytes through ncclBroadcast and a grouped ncclSend/ncclRecv and verifies them, the
index takes the create-empty / upload-with-watermark / broadcast-behind-it route (an in-place broadcast on one rank moves
nothing, every line still executes), and with the self-exchange hook rank 0's results travel through ncclSend/ncclRecv to
itself instead of a device-to-device copy.  The several-replica logic is exercised through the host transport, which accepts
a device twice.  On a box with two or more GPUs the tests at the end run the real thing: Comm([0, 1], "rccl").  Every
transport must hand out byte-identical CSRs, equal to the concatenation of the searchers' own results."""
import threading

import numpy as np
import pytest

from taxor_amd import Comm, GpuIndex, Searcher, synth
from taxor_amd._lib import TaxorError

pytestmark = pytest.mark.gpu


def _index_and_batches(n_batches, seed=7):
    g, go = synth.random_genomes(6, 20000, seed=seed)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(6)]
    lay = synth.make_layout(planted, root_bins=64, child_bins=32, n_children=3, seed=seed)
    host = synth.materialize_host(lay)
    batches = [synth.synth_reads(g, go, 150 + 37 * b, 1500, error_rate=0.02, frac_random=0.15, seed=seed + 10 + b)[:2] for b in range(n_batches)]
    return host, lay["n_user_bins"], batches


def _concat(results):
    off, base = [np.zeros(1, np.uint64)], 0
    for r in results:
        off.append(r.read_off[1:] + np.uint64(base))
        base += int(r.read_off[-1])
    return (np.concatenate(off), np.concatenate([r.user_bin for r in results]), np.concatenate([r.count for r in results]),
            np.concatenate([r.n_hashes for r in results]))


def _same(res, want):
    return (np.array_equal(res.read_off, want[0]) and np.array_equal(res.user_bin, want[1]) and np.array_equal(res.count, want[2])
            and np.array_equal(res.n_hashes, want[3]))


@pytest.mark.parametrize("transport", ["rccl", "host"])
def test_single_rank_communicator(transport):
    host, nub, batches = _index_and_batches(2)
    ref_idx = GpuIndex(host, nub)
    ref = Searcher(ref_idx)
    want = [ref.search_batch(*b) for b in batches]
    comm = Comm([0], transport)
    (idx,) = comm.replicate_index(host, nub)
    assert idx.data_bytes == ref_idx.data_bytes
    for i in range(len(host)):       # the replica holds the same fingerprint bytes
        assert np.array_equal(idx.download_ixf(i), ref_idx.download_ixf(i))
    sr = Searcher(idx)
    for b, w in zip(batches, want):
        sr.search_batch_begin(*b)
        got = comm.gather([sr])
        assert _same(got, (w.read_off, w.user_bin, w.count, w.n_hashes))
        assert got.user_bin.size > 0
    info = comm.info()
    assert info["n_devices"] == 1 and info["gathers"] == 2 and info["index_bytes"] == ref_idx.data_bytes
    if transport == "rccl":
        # the RCCL code of a larger run, executed on one rank: verified self-test bytes at creation, the broadcast loop
        # behind the upload, and (hook) the grouped send/recv of the gather
        assert info["rccl_version"] >= 2000 and info["selftest_bytes"] == 2 * 4 * 65536
        assert info["index_broadcast_calls"] >= 1 and info["index_upload_bytes"] == ref_idx.data_bytes and info["index_broadcast_bytes"] == 0
        comm.set_self_exchange(True)
        for b, w in zip(batches, want):
            sr.search_batch_begin(*b)
            assert _same(comm.gather([sr]), (w.read_off, w.user_bin, w.count, w.n_hashes))
        sr.search_batch_begin(np.zeros(0, np.uint8), np.zeros(1, np.uint64))        # an empty batch: nothing to send, the group is empty
        assert comm.gather([sr]).n_hashes.size == 0
        assert comm.info()["self_exchange_bytes"] == sum(12 * (w.n_hashes.size + w.user_bin.size) for w in want)
    else:
        assert info["rccl_version"] == 0 and info["selftest_bytes"] == 0
        with pytest.raises(TaxorError, match="only the RCCL transport"):
            comm.set_self_exchange(True)
    sr.close(); idx.close(); comm.close(); ref.close(); ref_idx.close()


@pytest.mark.parametrize("transport", ["rccl", "host"])
def test_replicas_of_an_index_in_a_foreign_fingerprint_layout(transport):
    """the one PCIe upload of a replicated index goes through the same re-layout as a plain index creation (bin-major source,
    unpadded columns, rows position-major): every replica holds the search layout's bytes and answers like the plain index"""
    from taxor_amd.search import to_source_layout
    host, nub, batches = _index_and_batches(1, seed=11)
    layout = 0x301
    foreign = []
    for f in host:
        raw, pitch = to_source_layout(f, layout)
        foreign.append(dict(f, data=raw, src_stride=pitch))
    ref_idx = GpuIndex(host, nub)
    want = Searcher(ref_idx).search_batch(*batches[0])
    comm = Comm([0] if transport == "rccl" else [0, 0], transport)
    idxs = comm.replicate_index(foreign, nub, layout=layout)
    for idx in idxs:
        for i, f in enumerate(host):          # (columns beyond `bins` are padding: the re-layout writes zeros there, the plain upload what the host array held)
            got, ref = idx.download_ixf(i).reshape(-1, f["stride"]), ref_idx.download_ixf(i).reshape(-1, f["stride"])
            assert np.array_equal(got[:, :f["bins"]], ref[:, :f["bins"]]) and not got[:, f["bins"]:].any()
    srs = [Searcher(i) for i in idxs]
    for s in srs:
        s.search_batch_begin(*batches[0])
    got = comm.gather(srs)
    assert _same(got, _concat([want] * len(srs))) and got.user_bin.size > 0
    for s in srs:
        s.close()
    for i in idxs:
        i.close()
    comm.close(); ref_idx.close()


def test_host_transport_three_replicas_one_round():
    """three replicas (the same device three times: only the host transport allows that), one batch each, an empty batch on the
    third searcher in the second round"""
    host, nub, batches = _index_and_batches(3)
    comm = Comm([0, 0, 0], "host")
    idxs = comm.replicate_index(host, nub)
    srs = [Searcher(i) for i in idxs]
    singles = [Searcher(idxs[0]).search_batch(*b) for b in batches]
    for s, b in zip(srs, batches):
        s.search_batch_begin(*b)
    got = comm.gather(srs)
    assert _same(got, _concat(singles))
    # second round: devices 0 and 1 swap batches, device 2 has none
    srs[0].search_batch_begin(*batches[1])
    srs[1].search_batch_begin(*batches[0])
    srs[2].search_batch_begin(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    got = comm.gather(srs)
    assert _same(got, _concat([singles[1], singles[0]]))
    assert comm.info()["index_upload_bytes"] == 3 * idxs[0].data_bytes
    for s in srs:
        s.close()
    for i in idxs:
        i.close()
    comm.close()


def test_rccl_refuses_a_repeated_device_and_unknown_devices():
    with pytest.raises(TaxorError, match="listed twice"):
        Comm([0, 0], "rccl")
    with pytest.raises(TaxorError, match="does not exist"):
        Comm([0, 99], "host")


def test_gather_checks_the_searchers_device():
    host, nub, batches = _index_and_batches(1)
    comm = Comm([0], "host")
    (idx,) = comm.replicate_index(host, nub)
    sr = Searcher(idx)
    with pytest.raises(TaxorError, match="no run in flight"):
        comm.gather([sr])
    sr.close(); idx.close(); comm.close()


def test_segments_are_one_batch():
    """taxor_gpu_search_segments_begin: reads in several host buffers (one with a non-zero first offset, one empty) give the
    CSR of the same reads concatenated into one buffer -- through the streamed sub-batch path, whose H2D copies then come
    from several segments per sub-batch"""
    host, nub, batches = _index_and_batches(4, seed=11)
    idx = GpuIndex(host, nub)
    sr = Searcher(idx, sub_batch_reads=64)           # several sub-batches, cut across segment borders
    singles = [sr.search_batch(*b) for b in batches]
    b2, o2 = batches[2]
    shifted = (np.concatenate([np.frombuffer(b"ACGTACGTAC", np.uint8), b2]), o2 + np.uint64(10))     # offsets[0] != 0
    empty = (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    got = sr.search_segments([batches[0], empty, batches[1], shifted, batches[3]])
    assert _same(got, _concat(singles))
    allb = np.concatenate([b for b, _ in batches])
    offs, base = [np.zeros(1, np.uint64)], 0
    for b, o in batches:
        offs.append(o[1:] + np.uint64(base))
        base += b.size
    one = sr.search_batch(allb, np.concatenate(offs))
    assert _same(got, (one.read_off, one.user_bin, one.count, one.n_hashes))
    assert _same(sr.search_segments([]), (np.zeros(1, np.uint64), np.zeros(0, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.uint32)))
    sr.close(); idx.close()


def test_gather_of_one_set_while_another_set_runs():
    """the communicator contract (taxor_gpu.h): a gather waits for the searchers it is given; OTHER searchers on the same
    devices may run batches meanwhile -- the CLI's two searcher sets, one gathering round i while the other classifies round
    i+1.  Two sets on two replicas, one thread gathering set A again and again while the other keeps set B busy."""
    host, nub, batches = _index_and_batches(4, seed=23)
    comm = Comm([0, 0], "host")
    idxs = comm.replicate_index(host, nub)
    set_a = [Searcher(i) for i in idxs]
    set_b = [Searcher(i) for i in idxs]
    singles = [Searcher(idxs[0]).search_batch(*b) for b in batches]
    want_a = _concat([singles[0], singles[1]])
    stop = threading.Event()
    errors = []

    def keep_b_busy():
        try:
            while not stop.is_set():
                for s, b, w in zip(set_b, batches[2:], singles[2:]):
                    got = s.search_batch(*b)
                    if not _same(got, (w.read_off, w.user_bin, w.count, w.n_hashes)):
                        errors.append("set B result changed")
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=keep_b_busy)
    th.start()
    try:
        for _ in range(25):
            for s, b in zip(set_a, batches[:2]):
                s.search_batch_begin(*b)
            assert _same(comm.gather(set_a), want_a)
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    for s in set_a + set_b:
        s.close()
    for i in idxs:
        i.close()
    comm.close()


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: RCCL between devices (the usual test box has one)")
def test_rccl_between_two_devices():
    """Comm([0, 1], "rccl"): the index crosses PCIe once and reaches device 1 by ncclBroadcast; the round's results reach device 0
    by grouped ncclSend/ncclRecv -- against the host transport and the concatenation of single-searcher results, including an
    empty batch on one rank"""
    host, nub, batches = _index_and_batches(3, seed=31)
    ref_idx = GpuIndex(host, nub)
    ref = Searcher(ref_idx)
    singles = [ref.search_batch(*b) for b in batches]
    for transport in ("rccl", "host"):
        comm = Comm([0, 1], transport)
        idxs = comm.replicate_index(host, nub)
        for idx in idxs:
            for i in range(len(host)):
                assert np.array_equal(idx.download_ixf(i), ref_idx.download_ixf(i)), (transport, idx.device, i)
        srs = [Searcher(i) for i in idxs]
        for s, b in zip(srs, batches):
            s.search_batch_begin(*b)
        assert _same(comm.gather(srs), _concat(singles[:2])), transport
        srs[0].search_batch_begin(*batches[2])
        srs[1].search_batch_begin(np.zeros(0, np.uint8), np.zeros(1, np.uint64))          # rank 1 has nothing this round
        assert _same(comm.gather(srs), _concat([singles[2]])), transport
        srs[0].search_batch_begin(np.zeros(0, np.uint8), np.zeros(1, np.uint64))          # rank 0 has nothing, rank 1 does
        srs[1].search_batch_begin(*batches[1])
        assert _same(comm.gather(srs), _concat([singles[1]])), transport
        info = comm.info()
        if transport == "rccl":
            assert info["index_upload_bytes"] == ref_idx.data_bytes and info["index_broadcast_bytes"] > 0 and info["gather_bytes"] > 0
        for s in srs:
            s.close()
        for i in idxs:
            i.close()
        comm.close()
    ref.close(); ref_idx.close()
