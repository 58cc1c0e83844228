"""ctypes wrapper over liboracle.so (the CPU restatement in cpu_ref.cpp).  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_oracle_lib(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cpu_ref.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build_oracle_lib())
        vp, u64, i64 = C.c_void_p, C.c_uint64, C.c_int64
        lib.orc_create.restype = vp
        lib.orc_destroy.argtypes = [vp]
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_last_error.argtypes = [vp]
        lib.orc_add.argtypes = [vp, i64, C.c_char_p, u64]
        lib.orc_add_bulk.argtypes = [vp, vp, vp, vp, u64]
        lib.orc_build.argtypes = [vp, C.c_uint]
        lib.orc_build.restype = C.c_int
        for f in ("orc_size", "orc_bits", "orc_mask", "orc_canonicalize", "orc_inversions"):
            getattr(lib, f).restype = u64
            getattr(lib, f).argtypes = [vp]
        lib.orc_canonicalize_mt.restype = u64
        lib.orc_canonicalize_mt.argtypes = [vp, C.c_uint]
        lib.orc_sa_width.restype = C.c_int
        lib.orc_sa_width.argtypes = [vp]
        lib.orc_sa_data.restype = vp
        lib.orc_sa_data.argtypes = [vp]
        lib.orc_query.restype = C.c_int
        lib.orc_query.argtypes = [vp, C.c_char_p, u64, vp, vp, u64, C.POINTER(u64)]
        lib.orc_query_batch.restype = C.c_int
        lib.orc_query_batch.argtypes = [vp, vp, vp, u64, C.c_uint, vp, vp, vp, u64, C.POINTER(u64)]
        lib.orc_brute_count.argtypes = [vp, vp, u64, C.c_char_p, u64, vp]
        lib.orc_highlight_spans.restype = u64
        lib.orc_highlight_spans.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64]
        lib.orc_filter_or.restype = u64
        lib.orc_filter_or.argtypes = [vp, vp, vp, u64, vp, vp, u64]
        _LIB = lib
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleIndex:
    """Mirrors string_index's add/build/query (reference src/index.h:54-86)."""

    def __init__(self):
        self._h = _lib().orc_create()

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_destroy(self._h)
            self._h = None

    def add(self, id_, value: bytes):
        _lib().orc_add(self._h, int(id_), value, len(value))

    def add_bulk(self, ids, blob, doc_start):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
        assert len(doc_start) == len(ids) + 1
        if len(doc_start) and int(doc_start[-1]) > len(blob):
            raise ValueError(f"blob holds {len(blob)} bytes, doc_start[-1] = {int(doc_start[-1])}")
        _lib().orc_add_bulk(self._h, _ptr(ids), _ptr(blob), _ptr(doc_start), len(ids))

    def build(self, nthreads=0):
        if _lib().orc_build(self._h, nthreads) != 0:
            raise RuntimeError(_lib().orc_last_error(self._h).decode())

    size = property(lambda s: _lib().orc_size(s._h))
    bits = property(lambda s: _lib().orc_bits(s._h))
    mask = property(lambda s: _lib().orc_mask(s._h))
    sa_width = property(lambda s: _lib().orc_sa_width(s._h))

    def sa(self):
        n, w = self.size, self.sa_width
        dt = np.uint32 if w == 4 else np.uint64
        if n == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n * w)).from_address(_lib().orc_sa_data(self._h))
        return np.frombuffer(buf, dtype=dt).copy()

    def canonicalize(self, nthreads=1):
        return _lib().orc_canonicalize_mt(self._h, nthreads) if nthreads > 1 else _lib().orc_canonicalize(self._h)

    def sa_view(self):
        """The oracle's own array without a copy (valid while the index lives)."""
        n, w = self.size, self.sa_width
        dt = np.uint32 if w == 4 else np.uint64
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.frombuffer((C.c_char * (n * w)).from_address(_lib().orc_sa_data(self._h)), dtype=dt)

    def inversions(self):
        return _lib().orc_inversions(self._h)

    def query(self, kw: bytes):
        cap = 1024
        while True:
            ids = np.empty(cap, dtype=np.int64)
            cnt = np.empty(cap, dtype=np.int64)
            n = C.c_uint64(0)
            if _lib().orc_query(self._h, kw, len(kw), _ptr(ids), _ptr(cnt), cap, C.byref(n)) != 0:
                raise RuntimeError(_lib().orc_last_error(self._h).decode())
            if n.value <= cap:
                return list(zip(ids[: n.value].tolist(), cnt[: n.value].tolist()))
            cap = n.value

    def filter_or(self, keywords):
        """interface.cpp:78-113: union over keywords by id with summed counts, ascending id."""
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        cap = 1 << 16
        while True:
            ids = np.empty(cap, dtype=np.int64)
            cnt = np.empty(cap, dtype=np.int64)
            n = _lib().orc_filter_or(self._h, _ptr(blob), _ptr(offs), len(keywords), _ptr(ids), _ptr(cnt), cap)
            if n <= cap:
                return list(zip(ids[:n].tolist(), cnt[:n].tolist()))
            cap = n

    def highlight_spans(self, keywords, ids):
        """[(object id, [(begin, end_inclusive), ...])] per document with a match, insertion order
        (ac_automaton::render's spans, database.cpp:58-76)."""
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        cap = 1 << 16
        while True:
            dd = np.empty(cap, dtype=np.uint64); bb = np.empty(cap, dtype=np.uint64); ee = np.empty(cap, dtype=np.uint64)
            n = _lib().orc_highlight_spans(self._h, _ptr(blob), _ptr(offs), len(keywords), _ptr(dd), _ptr(bb), _ptr(ee), cap)
            if n <= cap:
                break
            cap = n
        out = []
        for k in range(n):
            d = int(dd[k])
            if not out or out[-1][0] != d:
                out.append((d, []))
            out[-1][1].append((int(bb[k]), int(ee[k])))
        return [(int(ids[d]), sp) for d, sp in out]

    def query_batch(self, blob, offsets, nthreads=1, want_rows=True):
        """Returns (row_ptr, ids, counts, total_hits); ids/counts are None when want_rows is False."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npat = len(offsets) - 1
        row_ptr = np.zeros(npat + 1, dtype=np.uint64)
        hits = C.c_uint64(0)
        lib = _lib()
        rc = lib.orc_query_batch(self._h, _ptr(blob), _ptr(offsets), npat, nthreads, _ptr(row_ptr), None, None, 0,
                                 C.byref(hits))
        if rc != 0:
            raise RuntimeError(lib.orc_last_error(self._h).decode())
        if not want_rows:
            return row_ptr, None, None, hits.value
        rows = int(row_ptr[-1])
        ids = np.empty(max(rows, 1), dtype=np.int64)
        cnt = np.empty(max(rows, 1), dtype=np.int64)
        lib.orc_query_batch(self._h, _ptr(blob), _ptr(offsets), npat, nthreads, _ptr(row_ptr), _ptr(ids), _ptr(cnt),
                            rows, C.byref(hits))
        return row_ptr, ids[:rows], cnt[:rows], hits.value


def brute_count(blob, doc_start, kw: bytes):
    """Overlapping occurrence count per document (reference test/test-string.py:14-19)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
    nd = len(doc_start) - 1
    out = np.zeros(nd, dtype=np.int64)
    _lib().orc_brute_count(_ptr(blob), _ptr(doc_start), nd, kw, len(kw), _ptr(out))
    return out
