"""ctypes loader for the CPU oracle (oracle/libtaxor_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (taxor_amd) never does.  See oracle/taxor_oracle.h for scope and pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtaxor_oracle.so")


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    """The oracle is compiled -march=native, so a library built on another machine (the build container) is
    rebuilt on the machine that runs it (the GPU box has the same image and gcc)."""
    src = [os.path.join(_HERE, f) for f in ("taxor_oracle.c", "taxor_oracle.h", "Makefile")]
    stamp = os.path.join(_HERE, ".build_host")
    host = _cpu_model()
    built_for = open(stamp).read().strip() if os.path.exists(stamp) else None
    if (force or not os.path.exists(_SO) or built_for != host
            or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libtaxor_oracle.so"])
        with open(stamp, "w") as f:
            f.write(host + "\n")
    return _SO


_REF_SO = os.path.join(_HERE, "_ref", "libtaxor_ref.so")
_ref = None


def ref_lib(reference_root="/root/reference"):
    """oracle/_ref/libtaxor_ref.so: the standalone-compilable pieces of the REAL reference (see ref_driver.cpp), built
    here when /root/reference is present; on the GPU box the prebuilt file travels with the repository.  None if
    neither is available."""
    global _ref
    if _ref is not None:
        return _ref
    src = os.path.join(_HERE, "ref_driver.cpp")
    if os.path.isdir(os.path.join(reference_root, "src")) and (
            not os.path.exists(_REF_SO) or os.path.getmtime(src) > os.path.getmtime(_REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "ref", f"REF={reference_root}"])
    if not os.path.exists(_REF_SO):
        return None
    L = C.CDLL(_REF_SO)
    L.ref_syncmer_match_ratio.restype = C.c_double
    L.ref_syncmer_match_ratio.argtypes = [C.c_size_t, C.c_double]
    L.ref_nmut_kmer_ci_high.restype = C.c_size_t
    L.ref_nmut_kmer_ci_high.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_double]
    L.ref_containment_index_ci_low.restype = C.c_double
    L.ref_containment_index_ci_low.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_double, C.c_double]
    L.ref_normal_cdf_inverse.restype = C.c_double
    L.ref_normal_cdf_inverse.argtypes = [C.c_double]
    L.ref_adjust_seed.restype = C.c_uint64
    L.ref_adjust_seed.argtypes = [C.c_uint8]
    L.ref_xor_build.restype = C.c_void_p
    L.ref_xor_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_xor_contain.restype = C.c_int
    L.ref_xor_contain.argtypes = [C.c_void_p, C.c_uint64]
    L.ref_xor_fingerprints.restype = C.POINTER(C.c_uint8)
    L.ref_xor_fingerprints.argtypes = [C.c_void_p]
    L.ref_xor_probe.restype = None
    L.ref_xor_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.ref_xor_free.restype = None
    L.ref_xor_free.argtypes = [C.c_void_p]
    L.ref_do_parallel_chunks.restype = C.c_uint64
    L.ref_do_parallel_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_double)]
    L.ref_do_parallel_slices.restype = None
    L.ref_do_parallel_slices.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p]
    L.ref_sync_out_lines.restype = None
    L.ref_sync_out_lines.argtypes = [C.c_char_p, C.c_int, C.c_int]
    _ref = L
    return _ref


class _Ixf(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("stride", C.c_uint64), ("seg_len", C.c_uint64),
                ("seed", C.c_uint64), ("data", C.c_void_p), ("arith", C.c_uint32)]


class _Hixf(C.Structure):
    _fields_ = [("n_ixf", C.c_size_t), ("ixf", C.POINTER(_Ixf)),
                ("next_ixf", C.POINTER(C.c_void_p)), ("fname_idx", C.POINTER(C.c_void_p))]


class _Params(C.Structure):
    _fields_ = [("k", C.c_int), ("s", C.c_int), ("t", C.c_int),
                ("error_rate", C.c_double), ("percentage", C.c_double), ("scaling", C.c_int),
                ("window", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_wyhash_u64.restype = C.c_uint64
        L.orc_wyhash_u64.argtypes = [C.c_uint64]
        L.orc_dna4_normalise.restype = C.c_int
        L.orc_dna4_normalise.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_seq_to_syncmers.restype = C.c_size_t
        L.orc_seq_to_syncmers.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_syncmer_match_ratio.restype = C.c_double
        L.orc_syncmer_match_ratio.argtypes = [C.c_size_t, C.c_double]
        L.orc_threshold.restype = C.c_size_t
        L.orc_threshold.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_double]
        L.orc_threshold_kind.restype = C.c_int
        L.orc_threshold_kind.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_double]
        L.orc_threshold_model.restype = C.c_size_t
        L.orc_threshold_model.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double]
        L.orc_nmut_kmer_ci_high.restype = C.c_size_t
        L.orc_nmut_kmer_ci_high.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_double]
        L.orc_containment_index_ci_low.restype = C.c_double
        L.orc_containment_index_ci_low.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_double, C.c_double]
        L.orc_normal_cdf_inverse.restype = C.c_double
        L.orc_normal_cdf_inverse.argtypes = [C.c_double]
        L.orc_adjust_seed.restype = C.c_uint64
        L.orc_adjust_seed.argtypes = [C.c_int]
        L.orc_minimiser_hash.restype = C.c_size_t
        L.orc_minimiser_hash.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_ixf_seg_len.restype = C.c_uint64
        L.orc_ixf_seg_len.argtypes = [C.c_uint64]
        L.orc_ixf_probe.restype = None
        L.orc_ixf_probe.argtypes = [C.POINTER(_Ixf), C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_ixf_bulk_count.restype = None
        L.orc_ixf_bulk_count.argtypes = [C.POINTER(_Ixf), C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_bulk_contains.restype = C.c_size_t
        L.orc_bulk_contains.argtypes = [C.POINTER(_Hixf), C.c_void_p, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_search_batch.restype = C.c_int
        L.orc_search_batch.argtypes = [C.POINTER(_Hixf), C.POINTER(_Params), C.c_void_p, C.c_void_p,
                                       C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_classify_filter.restype = None
        L.orc_classify_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_batch_begin.restype = C.c_void_p
        L.orc_batch_begin.argtypes = [C.POINTER(_Hixf), C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_batch_worker.restype = None
        L.orc_batch_worker.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_batch_finish.restype = C.c_int
        L.orc_batch_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def wyhash(x):
    return int(lib().orc_wyhash_u64(C.c_uint64(int(x) & (2**64 - 1))))


def dna4_normalise(seq: bytes) -> bytes:
    buf = np.frombuffer(seq, dtype=np.uint8).copy()
    if lib().orc_dna4_normalise(_p(buf), buf.size) != 0:
        raise ValueError("character outside the dna15 alphabet")
    return buf.tobytes()


def seq_to_syncmers(seq: bytes, k=22, s=12, t=5) -> np.ndarray:
    cap = max(len(seq), 1)
    out = np.empty(cap, dtype=np.uint64)
    n = lib().orc_seq_to_syncmers(seq, len(seq), k, s, t, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def syncmer_match_ratio(k, err):
    return float(lib().orc_syncmer_match_ratio(k, err))


def threshold(hash_count, k=22, err=0.04, percentage=-1.0):
    return int(lib().orc_threshold(hash_count, k, err, percentage))


THR_PERCENTAGE, THR_SYNCMER, THR_KMER, THR_FRACMINHASH = 0, 1, 2, 3


def threshold_kind(use_syncmer, k, window, percentage=-1.0):
    return int(lib().orc_threshold_kind(1 if use_syncmer else 0, k, window, percentage))


def threshold_model(kind, count, k=22, err=0.04, percentage=-1.0, scaling_factor=1.0):
    return int(lib().orc_threshold_model(kind, count, k, err, percentage, scaling_factor))


def adjust_seed(k):
    return int(lib().orc_adjust_seed(k))


def minimiser_hash(seq: bytes, k=20, w=20) -> np.ndarray:
    cap = max(len(seq), 1)
    out = np.empty(cap, dtype=np.uint64)
    n = lib().orc_minimiser_hash(seq, len(seq), k, w, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def ixf_seg_len(max_bin_elements):
    return int(lib().orc_ixf_seg_len(max_bin_elements))


class Hixf:
    """Host view of a HIXF for the oracle.

    ixfs: list of dicts {bins, stride, seg_len, seed, data(np.uint8 1-D, rows*stride)}
    next_ixf / fname_idx: list of np.int64 arrays (one per IXF, length bins)
    """

    def __init__(self, ixfs, next_ixf, fname_idx, arith=0):
        """arith: 0 = the restated reading of the un-vendored IXF arithmetic; else the code of another reading (taxor_oracle.h)"""
        self.n = len(ixfs)
        self._keep = []
        arr = (_Ixf * self.n)()
        for i, f in enumerate(ixfs):
            d = np.ascontiguousarray(f["data"], dtype=np.uint8)
            assert d.size == 3 * f["seg_len"] * f["stride"], "IXF data size mismatch"
            self._keep.append(d)
            arr[i] = _Ixf(f["bins"], f["stride"], f["seg_len"], f["seed"], d.ctypes.data, int(arith))
        self._ixf = arr
        self._nx = [np.ascontiguousarray(a, dtype=np.int64) for a in next_ixf]
        self._fn = [np.ascontiguousarray(a, dtype=np.int64) for a in fname_idx]
        self._nxp = (C.c_void_p * self.n)(*[a.ctypes.data for a in self._nx])
        self._fnp = (C.c_void_p * self.n)(*[a.ctypes.data for a in self._fn])
        self.c = _Hixf(self.n, arr, self._nxp, self._fnp)
        self.total_leaves = int(sum((a >= 0).sum() for a in self._fn))

    def ixf_probe(self, i, key):
        rows = np.zeros(3, dtype=np.uint64)
        fp = np.zeros(1, dtype=np.uint8)
        lib().orc_ixf_probe(C.byref(self._ixf[i]), C.c_uint64(int(key)), _p(rows), _p(fp))
        return rows, int(fp[0])

    def ixf_bulk_count(self, i, hashes):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.zeros(int(self._ixf[i].bins), dtype=np.uint32)
        lib().orc_ixf_bulk_count(C.byref(self._ixf[i]), _p(h), h.size, _p(out))
        return out

    def synth_keys_found(self, i, bin_, first, n, salt, sample_step=0, counts=None, threads=0):
        """the synthetic keys first .. first+n-1 against column `bin_` of IXF i (bulk_count's rule on one column) -> number found;
        every sample_step-th key also through orc_ixf_bulk_count over all bins: counts (uint64[bins]) accumulates.
        Returns (found, sampled)."""
        sampled = C.c_uint64(0)
        L = lib()
        L.orc_ixf_synth_keys_found.restype = C.c_uint64
        L.orc_ixf_synth_keys_found.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        threads = int(threads) if threads else min(32, len(os.sched_getaffinity(0)) or 8)
        found = L.orc_ixf_synth_keys_found(C.byref(self._ixf[i]), int(bin_), int(first), int(n), int(salt) & (2**64 - 1), int(sample_step),
                                           _p(counts) if counts is not None else None, C.byref(sampled), threads)
        return int(found), int(sampled.value)

    def bulk_contains(self, hashes, thr):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        cap = self.total_leaves + 1
        ub = np.empty(cap, dtype=np.int64)
        cnt = np.empty(cap, dtype=np.uint32)
        vb = np.zeros(1, dtype=np.uint64)
        n = lib().orc_bulk_contains(C.byref(self.c), _p(h), h.size, int(thr), _p(ub), _p(cnt), cap, _p(vb))
        assert n <= cap
        return ub[:n].copy(), cnt[:n].copy(), int(vb[0])

    def search_batch(self, bases: np.ndarray, offsets: np.ndarray, k=22, s=12, t=5, err=0.04,
                     percentage=-1.0, threads=1, scaling=1, window=0, scheduler="openmp", chunk=1024, timing=None):
        """bases: np.uint8 ASCII (already dna4-normalised), offsets: uint64[n+1].
        Returns (n_hashes u32[n], out_off u64[n+1], user_bin i64[], count u32[], visited_bytes).
        scheduler="reference": the worker runs under the REFERENCE's own hixf::do_parallel (oracle/_ref, do_parallel.hpp:17-36),
        one call per `chunk` records like taxor_search.cpp:315-326; timing (a dict) then receives its compute_time."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        prm = _Params(k, s, t, err, percentage, scaling, window)   # window > 0: index built without --use-syncmer
        nh = np.zeros(n, dtype=np.uint32)
        off = np.zeros(n + 1, dtype=np.uint64)
        cap = max(4 * n, 1024)
        vb = np.zeros(1, dtype=np.uint64)
        if scheduler == "reference":
            R = ref_lib()
            if R is None:
                raise RuntimeError("scheduler='reference' needs oracle/_ref/libtaxor_ref.so (make -C oracle ref)")
            L = lib()
            ctx = L.orc_batch_begin(C.byref(self.c), C.byref(prm), _p(bases), _p(offsets), n, _p(nh))
            ct = C.c_double(0.0)
            R.ref_do_parallel_chunks(C.cast(L.orc_batch_worker, C.c_void_p), ctx, n, int(chunk), max(1, int(threads)), C.byref(ct))
            if timing is not None:
                timing["compute_time"] = ct.value
            off[0] = 0
            # sizes are known only after the run: assemble with a generous buffer, grow once if needed
            ub = np.empty(max(cap, 64 * n + 1024), dtype=np.int64)
            cnt = np.empty(ub.size, dtype=np.uint32)
            rc = L.orc_batch_finish(ctx, _p(off), _p(ub), _p(cnt), ub.size, _p(vb))
            if rc == -1:
                raise RuntimeError(f"reference-scheduled batch produced {int(off[n])} tuples, more than the {ub.size} provided for")
            if rc != 0:
                raise RuntimeError("the scheduler's slices did not tile the batch")
            tot = int(off[n])
            return nh, off, ub[:tot].copy(), cnt[:tot].copy(), int(vb[0])
        while True:
            ub = np.empty(cap, dtype=np.int64)
            cnt = np.empty(cap, dtype=np.uint32)
            rc = lib().orc_search_batch(C.byref(self.c), C.byref(prm), _p(bases), _p(offsets), n, threads,
                                        _p(nh), _p(off), _p(ub), _p(cnt), cap, _p(vb))
            if rc == 0:
                tot = int(off[n])
                return nh, off, ub[:tot].copy(), cnt[:tot].copy(), int(vb[0])
            cap = int(off[n]) + 16


def classify_filter(counts):
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    keep = np.zeros(c.size, dtype=np.uint8)
    lib().orc_classify_filter(_p(c), c.size, _p(keep))
    return keep.astype(bool)
