"""`.hixf` files through the library's reader / writer (taxor_amd/csrc/hixf_io.cpp): the drop-in on-disk
format of `taxor search` (cereal binary envelope of src/main/index.hpp:208-244)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def store_hixf(path, ixfs, n_user_bins, species, k=22, s=12, t=5, filenames=None, window_size=None, scaling=1,
               schema=None, use_syncmer=True, data_of=None):
    """ixfs: list of dicts {bins, stride, seg_len, seed, data, next_ixf, fname_idx};
    species: list of dicts {organism_name, accession_id, taxid, taxnames_string, taxid_string, user_bin, seq_len}.
    data_of: optional callable ixf -> uint8 array of that IXF's fingerprint bytes, called once per IXF in order while the file
    is written (the ixfs' own "data" are then ignored): an index that is resident on a GPU is written without a host copy of
    all of it at once."""
    keep = []
    arr = (_lib.IxfView * len(ixfs))()
    for i, f in enumerate(ixfs):
        d = None if data_of is not None else np.ascontiguousarray(f["data"], dtype=np.uint8)
        nx = np.ascontiguousarray(f["next_ixf"], dtype=np.int64)
        fn = np.ascontiguousarray(f["fname_idx"], dtype=np.int64)
        keep += [d, nx, fn]
        arr[i] = _lib.IxfView(f["bins"], f["stride"], f["seg_len"], f["seed"], d.ctypes.data if d is not None else None, nx.ctypes.data, fn.ctypes.data)
    ws = window_size if window_size is not None else k
    view = _lib.HixfView(len(ixfs), arr, n_user_bins, k, s, t, 1 if use_syncmer else 0, scaling, ws)
    if data_of is not None:
        cache = {}

        def _read(_ctx, ixf, off, n, dst):
            try:
                if cache.get("ixf") != ixf:
                    cache.clear()
                    cache["ixf"], cache["data"] = ixf, np.ascontiguousarray(data_of(int(ixf)), dtype=np.uint8)
                C.memmove(dst, cache["data"].ctypes.data + off, n)
                return 0
            except Exception:        # a Python exception must not unwind through the C caller
                return -1

        cb = _lib.IXF_READ_FN(_read)
        src = _lib.IxfSource(cb, None)
        keep += [cb, src]
        view.source = C.cast(C.pointer(src), C.c_void_p)
    sp = (_lib.Species * len(species))()
    for i, x in enumerate(species):
        sp[i] = _lib.Species(x["organism_name"].encode(), x["accession_id"].encode(), x["taxid"].encode(),
                             x["taxnames_string"].encode(), x["taxid_string"].encode(), x["user_bin"], x["seq_len"])
    if filenames is None:
        filenames = [f"user_bin_{i}.fna" for i in range(n_user_bins)]
    fns = (C.c_char_p * len(filenames))(*[f.encode() for f in filenames])
    meta = _lib.HixfMeta(ws, 1, 0, len(species), sp, len(filenames), fns)
    if schema is None:
        check(_lib.lib().taxor_hixf_store(str(path).encode(), C.byref(view), C.byref(meta)))
    else:
        check(_lib.lib().taxor_hixf_store_schema(str(path).encode(), C.byref(view), C.byref(meta), C.byref(schema)))


def default_schema():
    sc = _lib.IxfSchema()
    _lib.lib().taxor_ixf_schema_default(C.byref(sc))
    return sc


def make_schema(n_before, n_after, idx_bins, idx_stride, idx_seg_len, idx_seed, seg_len_is_rows=0, layout=0, len_unit=1, skip_before_len=0,
                skip_after_len=0):
    """len_unit: what the vector's length word counts (1 bytes, 8 64-bit words, 64 bits in whole words); skip_*: filler bytes around it"""
    return _lib.IxfSchema(n_before, n_after, idx_bins, idx_stride, idx_seg_len, idx_seed, seg_len_is_rows,
                          13572355802537770549, layout, len_unit, skip_before_len, skip_after_len)


def parse_layout(spec):
    """"bin-major,unpadded,position-major" -> layout code (taxor_ixf_layout_parse)"""
    c = C.c_uint32()
    check(_lib.lib().taxor_ixf_layout_parse(spec.encode(), C.byref(c)))
    return int(c.value)


def describe_layout(code):
    buf = C.create_string_buffer(128)
    _lib.lib().taxor_ixf_layout_describe(int(code), buf, len(buf))
    return buf.value.decode()


def probe_hixf(path):
    """hixf-probe: infer the IXF record layout of a file -> (schema, report text)"""
    sc = _lib.IxfSchema()
    buf = C.create_string_buffer(8192)
    check(_lib.lib().taxor_hixf_probe(str(path).encode(), C.byref(sc), buf, len(buf)))
    return sc, buf.value.decode()


class HixfFile:
    """A parsed .hixf (mmap).  .ixfs are zero-copy numpy views valid while the object lives."""

    def __init__(self, path, schema=None):
        h = C.c_void_p()
        if schema is None:
            check(_lib.lib().taxor_hixf_load(str(path).encode(), C.byref(h)))
        else:
            check(_lib.lib().taxor_hixf_load_schema(str(path).encode(), C.byref(schema), C.byref(h)))
        self._h = h
        self._read_view()

    def set_layout(self, layout):
        """the fingerprint layout `taxor verify --variants` / `taxor pin` found the file to follow (taxor_hixf_set_layout)"""
        check(_lib.lib().taxor_hixf_set_layout(self._h, int(layout)))
        self._read_view()

    def _read_view(self):
        h = self._h
        v = _lib.lib().taxor_hixf_get_view(h).contents
        m = _lib.lib().taxor_hixf_get_meta(h).contents
        self.k, self.s, self.t = v.kmer_size, v.syncmer_size, v.t_syncmer
        self.use_syncmer, self.scaling = bool(v.use_syncmer), v.scaling
        self.n_user_bins = int(v.n_user_bins)
        self.window_size = int(m.window_size)
        self.foreign_schema = bool(m.foreign_schema)
        self.layout = int(v.ixf_layout)      # how the FILE lays each IXF's bytes out (ixf_layout.h); "data" below are those raw bytes
        self.ixfs = []
        for i in range(v.n_ixf):
            f = v.ixf[i]
            n = int(_lib.lib().taxor_hixf_ixf_raw_bytes(h, i))
            data = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_uint8)), shape=(n,))
            nx = np.ctypeslib.as_array(C.cast(f.next_ixf, C.POINTER(C.c_int64)), shape=(f.bins,))
            fn = np.ctypeslib.as_array(C.cast(f.fname_idx, C.POINTER(C.c_int64)), shape=(f.bins,))
            self.ixfs.append(dict(bins=int(f.bins), stride=int(f.stride), seg_len=int(f.seg_len), seed=int(f.seed),
                                  data=data, next_ixf=nx, fname_idx=fn, src_stride=int(f.src_stride)))
        self.species = [dict(organism_name=m.species[i].organism_name.decode("utf-8", "replace"), accession_id=m.species[i].accession_id.decode("utf-8", "replace"),
                             taxid=m.species[i].taxid.decode("utf-8", "replace"), taxnames_string=m.species[i].taxnames_string.decode("utf-8", "replace"),
                             taxid_string=m.species[i].taxid_string.decode("utf-8", "replace"), user_bin=int(m.species[i].user_bin),
                             seq_len=int(m.species[i].seq_len)) for i in range(m.n_species)]
        self.filenames = [m.user_bin_filenames[i].decode("utf-8", "replace") for i in range(m.n_user_bin_filenames)]

    def format_read(self, read_id: str, read_len, n_hashes, user_bin, count):
        ub = np.ascontiguousarray(user_bin, dtype=np.int64)
        ct = np.ascontiguousarray(count, dtype=np.uint32)
        rid = read_id.encode()
        buf = C.create_string_buffer(4096)
        n = _lib.lib().taxor_format_read(self._h, rid, len(rid), read_len, n_hashes, ub.ctypes.data, ct.ctypes.data,
                                         ub.size, buf, len(buf))
        if n > len(buf):
            buf = C.create_string_buffer(int(n))
            n = _lib.lib().taxor_format_read(self._h, rid, len(rid), read_len, n_hashes, ub.ctypes.data, ct.ctypes.data,
                                             ub.size, buf, len(buf))
        return buf.raw[:n].decode("utf-8", "replace")

    def format_reads(self, ids, read_lens, n_hashes, read_off, user_bin, count):
        """the text of a whole chunk of reads (taxor_format_reads)"""
        n = len(ids)
        enc = [i.encode() for i in ids]
        idp = (C.c_char_p * n)(*enc)
        idl = np.array([len(e) for e in enc], dtype=np.uint64)
        rl = np.ascontiguousarray(read_lens, dtype=np.uint64)
        nh = np.ascontiguousarray(n_hashes, dtype=np.uint32)
        ro = np.ascontiguousarray(read_off, dtype=np.uint64)
        ub = np.ascontiguousarray(user_bin, dtype=np.int64)
        ct = np.ascontiguousarray(count, dtype=np.uint32)
        args = (self._h, n, C.cast(idp, C.c_void_p), idl.ctypes.data, rl.ctypes.data, nh.ctypes.data, ro.ctypes.data, ub.ctypes.data, ct.ctypes.data)
        need = _lib.lib().taxor_format_reads(*args, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        got = _lib.lib().taxor_format_reads(*args, buf, int(need))
        assert got == need
        return buf.raw[:need].decode("utf-8", "replace")

    def close(self):
        if getattr(self, "_h", None):
            self.ixfs = []
            _lib.lib().taxor_hixf_free(self._h)
            self._h = None

    __del__ = close
