"""Host-side mirror (Python) of the reference's search interface on top of the C ABI.

Names follow the reference: GpuIndex stands where `taxor_index<hixf_t>` does (src/main/index.hpp:25-287),
Searcher where the worker's `membership_agent` + `threshold` do (src/main/taxor_search.cpp:196-313,
src/hixf/build/hierarchical_interleaved_xor_filter.hpp:290-413).  All computation happens in
libtaxor_gpu.so; this module only marshals numpy arrays."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def threshold_ratio(kmer_size=22, error_rate=0.04, percentage=-1.0):
    """hixf::threshold::threshold + get_min_syncmer_match_ratio (threshold.hpp:22-81, syncmer_model.hpp:38-50)"""
    r = _lib.lib().taxor_threshold_ratio(int(kmer_size), float(error_rate), float(percentage))
    if r < 0:
        raise ValueError(f"no syncmer threshold model for k={kmer_size}, error_rate={error_rate}")
    return r


def threshold(hash_count, ratio):
    return int(_lib.lib().taxor_threshold(int(hash_count), float(ratio)))


def threshold_kind(use_syncmer, kmer_size, window_size, percentage=-1.0):
    """threshold::threshold's choice of model (threshold.hpp:22-47) -> _lib.THR_*"""
    return int(_lib.lib().taxor_threshold_kind(1 if use_syncmer else 0, int(kmer_size), int(window_size), float(percentage)))


def threshold_model(kind, count, kmer_size, error_rate=0.04, percentage=-1.0, scaling_factor=1.0):
    """threshold::get for every kind (threshold.hpp:51-81), host arithmetic identical to the reference's"""
    return int(_lib.lib().taxor_threshold_model(int(kind), int(count), int(kmer_size), float(error_rate), float(percentage),
                                                float(scaling_factor)))


def arith_code(key_hash=0, seed_mode=0, rot=21, reduce=0, fp_mode=0):
    """code of a reading of the un-vendored IXF arithmetic (taxor_ixf_variant's five arithmetic fields); 0 = the library's"""
    v = _lib.IxfVariant(0, 1, 64, key_hash, seed_mode, rot, reduce, fp_mode, 0)
    return int(_lib.lib().taxor_ixf_arith_code(C.byref(v)))


def classify_filter(counts):
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    keep = np.zeros(c.size, dtype=np.uint8)
    _lib.lib().taxor_classify_filter(_p(c), c.size, _p(keep))
    return keep.astype(bool)


@dataclass
class SearchResults:
    """CSR per-read tuples in the reference's DFS emission order, before the 0.8*max filter."""
    read_off: np.ndarray   # uint64[n_reads+1]
    user_bin: np.ndarray   # int64[n_tuples]
    count: np.ndarray      # uint32[n_tuples]
    n_hashes: np.ndarray   # uint32[n_reads]   (QHASH_COUNT)

    def tuples(self, r):
        lo, hi = int(self.read_off[r]), int(self.read_off[r + 1])
        return [(int(a), int(b)) for a, b in zip(self.user_bin[lo:hi], self.count[lo:hi])]


def _results(res: _lib.Results, copy=True) -> SearchResults:
    """copy=False: views of the library's own result arrays, valid until the next call on that searcher (the C ABI's
    convention; what a C++ host sees) -- for callers that time the library and not numpy's memcpy of 100 MB of tuples"""
    n, t = int(res.n_reads), int(res.n_tuples)
    c = (lambda a: a.copy()) if copy else (lambda a: a)
    ro = c(np.ctypeslib.as_array(res.read_off, shape=(n + 1,)))
    ub = c(np.ctypeslib.as_array(res.user_bin, shape=(t,))) if t else np.zeros(0, np.int64)
    ct = c(np.ctypeslib.as_array(res.count, shape=(t,))) if t else np.zeros(0, np.uint32)
    nh = c(np.ctypeslib.as_array(res.n_hashes, shape=(n,))) if n else np.zeros(0, np.uint32)
    return SearchResults(ro, ub, ct, nh)


def source_bytes(layout, f):
    """bytes a source holds for IXF dict f under layout code `layout` (taxor_amd/csrc/ixf_layout.h ixf_src_bytes)"""
    rows, kind = 3 * f["seg_len"], int(layout) & 0xFF
    # (ixf_layout.h ixf_src_pitch: the explicit pitch, else what the code's pitch rule names -- `bins` when unpadded, the index's stride otherwise)
    pitch = int(f.get("src_stride", 0)) or (f["bins"] if (int(layout) & 0x600) == _lib.LAYOUT_PITCH_BINS else f["stride"])
    if kind == _lib.LAYOUT_BIT_SLICED:
        return rows * ((f["bins"] + 63) // 64) * 64
    return rows * pitch


def to_source_layout(f, layout):
    """numpy restatement of ixf_layout.h, independent of the library: the bytes ANOTHER writer would store for IXF dict f (data in
    the search layout data[row * stride + bin]) under `layout` -> (bytes, src_stride).  Test infrastructure for the re-layout."""
    bins, stride, seg = f["bins"], f["stride"], f["seg_len"]
    rows, kind, rule = 3 * seg, int(layout) & 0xFF, int(layout) & 0x600
    D = np.asarray(f["data"], dtype=np.uint8).reshape(rows, stride)[:, :bins]
    if layout & _lib.LAYOUT_POSITION_MAJOR:          # source row pos*3 + segment holds search row segment*seg_len + pos
        D = D.reshape(3, seg, bins).transpose(1, 0, 2).reshape(rows, bins)
    S = (bins + 63) // 64 * 64
    pitch = S if kind == _lib.LAYOUT_BIT_SLICED else bins if rule == _lib.LAYOUT_PITCH_BINS else stride if rule == _lib.LAYOUT_PITCH_STORED else S
    if kind == _lib.LAYOUT_ROWS:
        out = np.zeros((rows, pitch), np.uint8)
        out[:, :bins] = D
    elif kind == _lib.LAYOUT_BIN_MAJOR:
        out = np.zeros((pitch, rows), np.uint8)
        out[:bins, :] = D.T
    else:
        P = np.zeros((rows, S), np.uint8)
        P[:, :bins] = D
        planes = (P[:, :, None] >> np.arange(8, dtype=np.uint8)[None, None, :]) & 1            # [row][bin][plane]
        planes = planes.reshape(rows, S // 64, 64, 8).transpose(0, 1, 3, 2)                      # [row][group][plane][bin in group]
        out = np.packbits(planes, axis=-1, bitorder="little")                                    # 8 bytes per plane word, little endian
    return np.ascontiguousarray(out).reshape(-1), pitch


class GpuIndex:
    """A HIXF resident in one GPU's HBM."""

    def __init__(self, ixfs, n_user_bins, k=22, s=12, t=5, device=0, use_syncmer=True, scaling=1, window_size=None, arith=0, layout=0):
        """ixfs: list of dicts {bins, stride, seg_len, seed, next_ixf, fname_idx, data (np.uint8 or None)[, src_stride]};
        layout != 0 (_lib.LAYOUT_*): `data` holds the bytes as ANOTHER writer's file would (ixf_layout.h, pitch / column count in
        src_stride); the library transposes them into the search layout on the device while uploading"""
        L = _lib.lib()
        view, keep = self._view(ixfs, n_user_bins, k, s, t, use_syncmer, scaling, window_size, layout)
        view.ixf_arith = int(arith)      # 0 = the library's reading of the IXF arithmetic; else arith_code(...)
        view.ixf_layout = int(layout)
        h = C.c_void_p()
        check(L.taxor_gpu_index_create(C.byref(view), device, C.byref(h)))
        del keep                 # the library copied everything into HBM
        self._adopt(h, ixfs, n_user_bins, k, s, t, device, use_syncmer, window_size)

    @staticmethod
    def _view(ixfs, n_user_bins, k, s, t, use_syncmer, scaling, window_size, layout=0):
        """taxor_hixf_view over numpy arrays (+ the arrays that must stay alive while the library reads it)"""
        keep = []
        arr = (_lib.IxfView * len(ixfs))()
        for i, f in enumerate(ixfs):
            nx = np.ascontiguousarray(f["next_ixf"], dtype=np.int64)
            fn = np.ascontiguousarray(f["fname_idx"], dtype=np.int64)
            assert nx.size == f["bins"] and fn.size == f["bins"]
            d = f.get("data")
            if d is not None:
                d = np.ascontiguousarray(d, dtype=np.uint8)
                assert d.size == source_bytes(layout, f), "IXF data size mismatch"
            keep += [nx, fn, d]
            arr[i] = _lib.IxfView(f["bins"], f["stride"], f["seg_len"], f["seed"],
                                  d.ctypes.data if d is not None else None, nx.ctypes.data, fn.ctypes.data, int(f.get("src_stride", 0)))
        ws = int(window_size) if window_size is not None else k
        keep.append(arr)
        return _lib.HixfView(len(ixfs), arr, n_user_bins, k, s, t, 1 if use_syncmer else 0, scaling, ws), keep

    def _adopt(self, h, ixfs, n_user_bins, k, s, t, device, use_syncmer, window_size):
        self._h = h
        self.use_syncmer = bool(use_syncmer)
        self.window_size = int(window_size) if window_size is not None else k
        self.device = device
        self.k, self.s, self.t = k, s, t
        self.n_ixf = len(ixfs)
        self.n_user_bins = n_user_bins
        self.shapes = [(f["bins"], f["stride"], f["seg_len"]) for f in ixfs]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().taxor_gpu_index_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def data_bytes(self):
        return int(_lib.lib().taxor_gpu_index_data_bytes(self._h))

    @property
    def leaf_runs(self):
        return int(_lib.lib().taxor_gpu_index_leaf_runs(self._h))

    @property
    def depth(self):
        return int(_lib.lib().taxor_gpu_index_depth(self._h))

    def gather_ceiling(self, ixf=0, want_bytes=32 << 30, reps=3, span=1):
        """random whole-row reads of IXF `ixf` (or of up to `span` equally shaped IXFs from it on), nothing else
        -> (GB/s requested, bytes read per row[, IXFs covered])"""
        g, rb, used = C.c_double(), C.c_uint64(), C.c_uint64()
        check(_lib.lib().taxor_gpu_gather_ceiling_span(self._h, ixf, int(span), int(want_bytes), int(reps), C.byref(g), C.byref(rb),
                                                       C.byref(used)))
        return (g.value, int(rb.value)) if span == 1 else (g.value, int(rb.value), int(used.value))

    def gather_pattern(self, ixf, pattern, nt, want_bytes=8 << 30, reps=3):
        """known-size launches in the query kernel's access shapes (0 = whole rows, 1 = one 16-B load per row)
        -> (GB/s requested, requested bytes per launch, requests per launch)"""
        g, b, r = C.c_double(), C.c_uint64(), C.c_uint64()
        check(_lib.lib().taxor_gpu_gather_pattern(self._h, ixf, int(pattern), 1 if nt else 0, int(want_bytes), int(reps), C.byref(g),
                                                  C.byref(b), C.byref(r)))
        return g.value, int(b.value), int(r.value)

    def fill_random(self, ixf, seed):
        check(_lib.lib().taxor_gpu_index_fill_random(self._h, ixf, seed))

    def upload_bin(self, ixf, bin_, column):
        col = np.ascontiguousarray(column, dtype=np.uint8)
        check(_lib.lib().taxor_gpu_index_upload_bin(self._h, ixf, bin_, _p(col), col.size))

    def build_ixf(self, ixf, bin_keys, seed0=1):
        """GPU construction of IXF `ixf` in place from {bin: uint64 keys}; returns (seed, peeling rounds)."""
        bins = self.shapes[ixf][0]
        off = np.zeros(bins + 1, dtype=np.uint64)
        parts = []
        for b in range(bins):
            k = np.ascontiguousarray(bin_keys.get(b, np.zeros(0, np.uint64)), dtype=np.uint64)
            parts.append(k)
            off[b + 1] = off[b] + np.uint64(k.size)
        keys = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        seed, rounds = C.c_uint64(), C.c_uint32()
        check(_lib.lib().taxor_gpu_index_build_ixf(self._h, ixf, _p(keys) if keys.size else None, _p(off), int(seed0),
                                                   C.byref(seed), C.byref(rounds)))
        return int(seed.value), int(rounds.value)

    def build_hixf(self, leaf_keys, seed0=1):
        """GPU construction of the whole hierarchy: leaf_keys = {(ixf, bin): uint64 keys} for leaf bins only; merged
        bins receive the union of their child IXF on the device.  Returns the number of peeling rounds of the slowest IXF."""
        parts, sizes = [], []
        for i, (bins, _, _) in enumerate(self.shapes):
            for b in range(bins):
                k = np.ascontiguousarray(leaf_keys.get((i, b), np.zeros(0, np.uint64)), dtype=np.uint64)
                parts.append(k)
                sizes.append(k.size)
        off = np.zeros(len(sizes) + 1, dtype=np.uint64)
        np.cumsum(np.array(sizes, dtype=np.uint64), out=off[1:])
        keys = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        rounds = C.c_uint32()
        check(_lib.lib().taxor_gpu_index_build_hixf(self._h, _p(keys) if keys.size else None, _p(off), int(seed0), C.byref(rounds)))
        return int(rounds.value)

    def build_hixf_synth(self, bin_counts, salt=1, seed0=1):
        """GPU construction of the whole hierarchy from SYNTHETIC keys generated on the device: technical bin g (index bin
        order: all bins of IXF 0, then IXF 1, ...) receives the keys synth_key(i, salt) for i in [off[g], off[g+1]), off =
        cumulative bin_counts (merged bins: 0).  Nothing crosses PCIe; a checker regenerates the keys with
        synth.synth_keys_host.  Returns the run's figures (taxor_build_stats) as a dict."""
        L = _lib.lib()
        cnt = np.ascontiguousarray(bin_counts, dtype=np.uint64)
        assert cnt.size == sum(b for b, _, _ in self.shapes)
        off = np.zeros(cnt.size + 1, dtype=np.uint64)
        np.cumsum(cnt, out=off[1:])
        total = int(off[-1])
        d_keys = C.c_void_p()
        check(L.taxor_gpu_malloc(self.device, total * 8, C.byref(d_keys)))
        try:
            check(L.taxor_gpu_synth_keys(self.device, d_keys, 0, total, int(salt)))
            st = _lib.BuildStats()
            check(L.taxor_gpu_index_build_hixf_ex(self._h, d_keys, 1, _p(off), int(seed0), C.byref(st)))
        finally:
            L.taxor_gpu_free(d_keys)
        return {k: getattr(st, k) for k, _ in _lib.BuildStats._fields_ if k != "reserved"}, off

    def build_hixf_host_keys(self, keys, off, seed0=1):
        """GPU construction of the whole hierarchy from keys in HOST memory (what a binding has): keys = uint64 array, off[total_bins + 1]
        per technical bin in index bin order (merged bins: empty).  Returns the run's figures (upload inside seconds_total)."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        o = np.ascontiguousarray(off, dtype=np.uint64)
        st = _lib.BuildStats()
        check(_lib.lib().taxor_gpu_index_build_hixf_ex(self._h, _p(k) if k.size else None, 0, _p(o), int(seed0), C.byref(st)))
        return {kk: getattr(st, kk) for kk, _ in _lib.BuildStats._fields_ if kk != "reserved"}

    def ixf_seed(self, ixf):
        return int(_lib.lib().taxor_gpu_index_ixf_seed(self._h, ixf))

    def download_ixf(self, ixf, out=None):
        bins, stride, seg = self.shapes[ixf]
        if out is None:
            out = np.empty(3 * seg * stride, dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == 3 * seg * stride and out.flags.c_contiguous
        check(_lib.lib().taxor_gpu_index_download_ixf(self._h, ixf, _p(out), out.size))
        return out


class Searcher:
    """One GPU-side agent: syncmers -> dedup -> threshold -> HIXF query -> per-read tuples."""

    def __init__(self, index: GpuIndex, error_rate=0.04, percentage=-1.0, ratio=None, sub_batch_reads=0,
                 sub_batch_bases=0, time_kernels=False, prune=True, group_always=False, small_path=True, split_always=False,
                 force_tree_stall=False):
        self.index = index
        flags = ((0 if prune else _lib.SEARCH_NO_PRUNE) | (_lib.SEARCH_GROUP_ALWAYS if group_always else 0)
                 | (0 if small_path else _lib.SEARCH_NO_SMALL_PATH) | (_lib.SEARCH_SPLIT_ALWAYS if split_always else 0)
                 | (_lib.SEARCH_FORCE_TREE_STALL if force_tree_stall else 0))
        prm = _lib.SearchParams(0.0, sub_batch_reads, sub_batch_bases, 1 if time_kernels else 0, _lib.THR_PERCENTAGE, error_rate, flags)
        if ratio is not None:          # explicit (size_t)(n * ratio), whatever the index
            prm.ratio = float(ratio)
        else:                          # the reference's choice: percentage / syncmer / k-mer / FracMinHash model
            view = _lib.HixfView(0, None, index.n_user_bins, index.k, index.s, index.t, 1 if index.use_syncmer else 0, 1,
                                 index.window_size)
            if _lib.lib().taxor_threshold_select(C.byref(view), float(error_rate), float(percentage), C.byref(prm)) != 0:
                raise ValueError(f"no threshold model for k={index.k}, error_rate={error_rate}")
        self.ratio, self.model = prm.ratio, int(prm.model)
        h = C.c_void_p()
        check(_lib.lib().taxor_gpu_searcher_create(index._h, C.byref(prm), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().taxor_gpu_searcher_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _batch(bases, offsets):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        return b, o

    # --- the drop-in batch call ---------------------------------------------------------------------------
    def search_batch(self, bases, offsets, copy=True) -> SearchResults:
        b, o = self._batch(bases, offsets)
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_search_batch(self._h, _p(b), _p(o), o.size - 1, C.byref(res)))
        return _results(res, copy)

    def search_batch_begin(self, bases, offsets):
        """enqueue the drop-in call; keep the arrays alive until search_batch_end()"""
        b, o = self._batch(bases, offsets)
        self._inflight = (b, o)
        check(_lib.lib().taxor_gpu_search_batch_begin(self._h, _p(b), _p(o), o.size - 1))

    def search_segments(self, segments) -> SearchResults:
        """one batch over reads that live in several (bases, offsets) buffers, in the order given"""
        keep = [self._batch(b, o) for b, o in segments]
        arr = (_lib.ReadSegment * len(keep))()
        for i, (b, o) in enumerate(keep):
            arr[i] = _lib.ReadSegment(b.ctypes.data, o.ctypes.data, o.size - 1)
        check(_lib.lib().taxor_gpu_search_segments_begin(self._h, arr, len(keep)))
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_search_batch_end(self._h, C.byref(res)))
        return _results(res)

    def search_batch_end(self) -> SearchResults:
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_search_batch_end(self._h, C.byref(res)))
        self._inflight = None
        return _results(res)

    # --- phases -------------------------------------------------------------------------------------------
    def upload(self, bases, offsets):
        b, o = self._batch(bases, offsets)
        check(_lib.lib().taxor_gpu_batch_upload(self._h, _p(b), _p(o), o.size - 1))

    def run(self):
        check(_lib.lib().taxor_gpu_batch_run(self._h))

    def sync(self):
        check(_lib.lib().taxor_gpu_batch_sync(self._h))

    def fetch(self) -> SearchResults:
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_batch_fetch(self._h, C.byref(res)))
        return _results(res)

    def stats(self):
        st = _lib.RunStats()
        check(_lib.lib().taxor_gpu_batch_stats(self._h, C.byref(st)))
        out = {f: getattr(st, f) for f, _ in _lib.RunStats._fields_}
        for f in ("level_ms", "level_requested_bytes", "level_row_reads", "level_sparse_loads"):
            out[f] = list(out[f])
        return out

    def phase_profile(self):
        """per-phase cycle sums of k_syncmers [0..7] and k_query_level [8..15] since the last call (needs a searcher
        created under TAXOR_PROFILE_PHASES=1)"""
        out = np.zeros(16, dtype=np.uint64)
        check(_lib.lib().taxor_gpu_phase_profile(self._h, _p(out)))
        return out

    def result_sizes(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(_lib.lib().taxor_gpu_batch_result_sizes(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def export_device(self, read_off_ptr=None, user_bin_ptr=None, count_ptr=None, n_hashes_ptr=None):
        """D2D copy of the last run's results into caller-owned device buffers (raw device pointers)."""
        check(_lib.lib().taxor_gpu_batch_export_device(self._h, read_off_ptr, user_bin_ptr, count_ptr, n_hashes_ptr))

    # --- stage entry points ---------------------------------------------------------------------------------
    def seq_to_syncmers(self, bases, offsets):
        """hashing::seq_to_syncmers for a batch -> (hash_off uint64[n+1], hashes uint64[])"""
        b, o = self._batch(bases, offsets)
        ho, hs = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        n = o.size - 1
        check(_lib.lib().taxor_gpu_syncmers(self._h, _p(b), _p(o), n, C.byref(ho), C.byref(hs)))
        hoff = np.ctypeslib.as_array(ho, shape=(n + 1,)).copy()
        tot = int(hoff[n])
        hashes = np.ctypeslib.as_array(hs, shape=(tot,)).copy() if tot else np.zeros(0, np.uint64)
        return hoff, hashes

    def ixf_bulk_count(self, ixf, hashes):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.zeros(self.index.shapes[ixf][0], dtype=np.uint32)
        check(_lib.lib().taxor_gpu_ixf_bulk_count(self._h, ixf, _p(h), h.size, _p(out)))
        return out

    def bulk_contains(self, hashes, thr):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_bulk_contains(self._h, _p(h), h.size, int(thr), C.byref(res)))
        r = _results(res)
        return r.user_bin, r.count


class Comm:
    """Several GPUs of one node driven by this one process (the C++ host's `taxor search --gpus N` shape): the index is
    replicated (one upload + ncclBroadcast), every device classifies its own batch, results are gathered on devices[0].
    transport: "rccl" (RCCL over xGMI, one rank per device) or "host" (same calls staged through host memory)."""

    def __init__(self, devices, transport="rccl"):
        self.devices = [int(d) for d in devices]
        self.transport = transport
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        check(_lib.lib().taxor_gpu_comm_create(arr, len(self.devices), _lib.COMM_RCCL if transport == "rccl" else _lib.COMM_HOST, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().taxor_gpu_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def replicate_index(self, ixfs, n_user_bins, k=22, s=12, t=5, use_syncmer=True, scaling=1, window_size=None, layout=0):
        """-> [GpuIndex on devices[0], GpuIndex on devices[1], ...]; layout as in GpuIndex"""
        view, keep = GpuIndex._view(ixfs, n_user_bins, k, s, t, use_syncmer, scaling, window_size, layout)
        view.ixf_layout = int(layout)
        out = (C.c_void_p * len(self.devices))()
        check(_lib.lib().taxor_gpu_index_create_replicated(self._h, C.byref(view), out))
        del keep
        idx = []
        for d, h in zip(self.devices, out):
            g = GpuIndex.__new__(GpuIndex)
            g._adopt(C.c_void_p(h), ixfs, n_user_bins, k, s, t, d, use_syncmer, window_size)
            idx.append(g)
        return idx

    def set_self_exchange(self, on=True):
        """test hook: rank 0's own results go through ncclSend/ncclRecv to itself (one-GPU boxes execute the exchange code)"""
        check(_lib.lib().taxor_gpu_comm_set_self_exchange(self._h, 1 if on else 0))

    def gather(self, searchers) -> SearchResults:
        """per-read results of one round (searcher i ran its batch on devices[i]) as one CSR in device order"""
        assert len(searchers) == len(self.devices)
        arr = (C.c_void_p * len(searchers))(*[s._h for s in searchers])
        res = _lib.Results()
        check(_lib.lib().taxor_gpu_gather_results(self._h, arr, C.byref(res)))
        return _results(res)

    def info(self):
        st = _lib.CommStats()
        check(_lib.lib().taxor_gpu_comm_info(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in _lib.CommStats._fields_}
