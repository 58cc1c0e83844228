"""ctypes binding of libcoffeedb_gpu.so (include/coffeedb_gpu.h).

`GpuStringIndex` mirrors the reference's string_index surface (add / build / query,
/root/reference/src/index.h:54-86) plus the batched query.  There is no CPU fallback: loading fails
loudly when the HIP library is missing, and cdb_create fails when no gfx950 device is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libcoffeedb_gpu.so")
if os.environ.get("CDB_LIB_PATH"):   # (timing experiments with ablated builds of the library: tools/experiments/abl/)
    LIB_PATH = os.environ["CDB_LIB_PATH"]
_LIB = None

EXPORTS = [
    "cdb_create", "cdb_destroy", "cdb_last_error", "cdb_add", "cdb_add_bulk", "cdb_build", "cdb_build_view", "cdb_build_views", "cdb_build_device", "cdb_build_resident", "cdb_raw_record_find_string", "cdb_add_raw_record", "cdb_add_raw_dir", "cdb_save", "cdb_load",
    "cdb_query", "cdb_query_or", "cdb_query_ranked", "cdb_query_and", "cdb_query_spans", "cdb_spans_free", "cdb_free", "cdb_query_batch", "cdb_query_batch_offsets", "cdb_hits_free", "cdb_result_free", "cdb_query_batch_device", "cdb_query_batch_offsets_device", "cdb_size", "cdb_bits",
    "cdb_mask", "cdb_sa_width", "cdb_sa_copy", "cdb_set_option", "cdb_get_stat", "cdb_profile_get",
    "cdb_profile_dump", "cdb_profile_reset", "cdb_release_cached_memory", "cdb_cached_memory_bytes", "cdb_set_cache_limit", "cdb_memory_stats", "cdb_memory_reset_peak",
    "cdb_debug_radix_sort", "cdb_debug_verify", "cdb_debug_verify_reference", "cdb_debug_self_check", "cdb_proof_wait", "cdb_layout_rule", "cdb_debug_query_latency",
    "cdb_shards_create", "cdb_shards_destroy", "cdb_shards_last_error", "cdb_shards_add", "cdb_shards_add_bulk", "cdb_shards_set_option",
    "cdb_shards_build", "cdb_shards_query", "cdb_shards_query_batch", "cdb_shards_query_or", "cdb_shards_query_ranked", "cdb_shards_query_spans", "cdb_shards_count", "cdb_shards_get", "cdb_shards_first_doc",
    "cdb_shards_transport", "cdb_shards_build_views", "cdb_shards_query_batch_offsets", "cdb_shards_query_and", "cdb_shards_add_raw_dir", "cdb_shards_save", "cdb_shards_load",
    "cdb_comm_unique_id", "cdb_comm_create", "cdb_comm_create_group", "cdb_comm_destroy", "cdb_comm_last_error", "cdb_comm_merge", "cdb_comm_merge_counts",
    "cdb_comm_world", "cdb_comm_transport", "cdb_reserve", "cdb_reserve_wait",
]


class CdbResult(C.Structure):
    _fields_ = [("npat", C.c_uint64), ("nrows", C.c_uint64), ("nhits", C.c_uint64),
                ("row_ptr", C.POINTER(C.c_uint64)), ("ids", C.POINTER(C.c_int64)), ("counts", C.POINTER(C.c_int64))]


class CdbSpans(C.Structure):
    _fields_ = [("ndocs", C.c_uint64), ("nspans", C.c_uint64), ("ids", C.POINTER(C.c_int64)),
                ("span_ptr", C.POINTER(C.c_uint64)), ("begin", C.POINTER(C.c_uint64)), ("end", C.POINTER(C.c_uint64))]


class CdbHits(C.Structure):
    _fields_ = [("hit_ptr", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_uint64))]


class CdbDeviceResult(C.Structure):
    _fields_ = [("npat", C.c_uint64), ("nrows", C.c_uint64), ("nhits", C.c_uint64),
                ("d_row_ptr", C.c_void_p), ("d_ids", C.c_void_p), ("d_counts", C.c_void_p)]


class CdbShardSlice(C.Structure):
    _fields_ = [("npat", C.c_uint64), ("nrows_total", C.c_uint64), ("nrows_local", C.c_uint64),
                ("d_row_ptr", C.c_void_p), ("d_row_base", C.c_void_p)]


class CdbKeyQuery(C.Structure):
    _fields_ = [("index", C.c_void_p), ("blob", C.c_void_p), ("offsets", C.c_void_p), ("nkw", C.c_uint64),
                ("ids", C.c_void_p), ("counts", C.c_void_p), ("nrows", C.c_size_t)]


class CdbDeviceHits(C.Structure):
    _fields_ = [("d_hit_ptr", C.c_void_p), ("d_offsets", C.c_void_p)]


def build_library(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process do not see each
        # other's devices.  Loading torch first makes this library bind to the runtime torch uses
        # (tests and bench.py share device pointers with torch).  C/C++ hosts simply use /opt/rocm's.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing — build it with `make -C {CSRC}` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, u64, i64, cp = C.c_void_p, C.c_uint64, C.c_int64, C.c_char_p
    lib.cdb_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.cdb_destroy.argtypes = [vp]
    lib.cdb_destroy.restype = None
    lib.cdb_last_error.argtypes = [vp]
    lib.cdb_last_error.restype = cp
    lib.cdb_add.argtypes = [vp, i64, cp, C.c_size_t]
    lib.cdb_add_bulk.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_raw_record_find_string.argtypes = [vp, C.c_size_t, cp, C.POINTER(i64), C.POINTER(cp), C.POINTER(C.c_size_t)]
    lib.cdb_add_raw_record.argtypes = [vp, cp, vp, C.c_size_t]
    lib.cdb_add_raw_dir.argtypes = [vp, cp, cp, C.POINTER(u64), C.POINTER(u64)]
    lib.cdb_save.argtypes = [vp, cp]
    lib.cdb_load.argtypes = [vp, cp]
    lib.cdb_build.argtypes = [vp]
    lib.cdb_build_device.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_build_resident.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_query.argtypes = [vp, cp, C.c_size_t, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)),
                              C.POINTER(C.c_size_t)]
    lib.cdb_query_ranked.argtypes = [vp, vp, vp, u64, i64, i64, u64, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)),
                                     C.POINTER(C.c_size_t)]
    lib.cdb_query_or.argtypes = [vp, vp, vp, u64, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)),
                                 C.POINTER(C.c_size_t)]
    lib.cdb_query_and.argtypes = [C.POINTER(CdbKeyQuery), C.c_int, C.c_int, i64, i64, u64, C.POINTER(C.POINTER(i64)),
                                  C.POINTER(C.POINTER(i64)), C.POINTER(C.c_size_t)]
    lib.cdb_query_spans.argtypes = [vp, vp, vp, u64, C.POINTER(CdbSpans)]
    lib.cdb_spans_free.argtypes = [C.POINTER(CdbSpans)]
    lib.cdb_spans_free.restype = None
    lib.cdb_free.argtypes = [vp]
    lib.cdb_free.restype = None
    lib.cdb_query_batch.argtypes = [vp, vp, vp, u64, C.POINTER(CdbResult)]
    lib.cdb_query_batch_offsets.argtypes = [vp, vp, vp, u64, C.POINTER(CdbResult), C.POINTER(CdbHits)]
    lib.cdb_hits_free.argtypes = [C.POINTER(CdbHits)]
    lib.cdb_hits_free.restype = None
    lib.cdb_result_free.argtypes = [C.POINTER(CdbResult)]
    lib.cdb_result_free.restype = None
    lib.cdb_query_batch_device.argtypes = [vp, vp, vp, u64, u64, C.POINTER(CdbDeviceResult)]
    lib.cdb_query_batch_offsets_device.argtypes = [vp, vp, vp, u64, u64, C.POINTER(CdbDeviceResult), C.POINTER(CdbDeviceHits)]
    for f in ("cdb_size", "cdb_bits", "cdb_mask"):
        getattr(lib, f).argtypes = [vp]
        getattr(lib, f).restype = u64
    lib.cdb_sa_width.argtypes = [vp]
    lib.cdb_sa_copy.argtypes = [vp, vp, u64]
    lib.cdb_set_option.argtypes = [vp, cp, i64]
    lib.cdb_get_stat.argtypes = [vp, cp, C.POINTER(C.c_double)]
    lib.cdb_profile_get.argtypes = [vp, cp, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]
    lib.cdb_profile_dump.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.cdb_profile_reset.argtypes = [vp]
    lib.cdb_profile_reset.restype = None
    lib.cdb_release_cached_memory.argtypes = []
    lib.cdb_release_cached_memory.restype = None
    lib.cdb_set_cache_limit.argtypes = [u64]
    lib.cdb_set_cache_limit.restype = None
    lib.cdb_memory_stats.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.cdb_memory_stats.restype = None
    lib.cdb_memory_reset_peak.argtypes = []
    lib.cdb_memory_reset_peak.restype = None
    lib.cdb_cached_memory_bytes.argtypes = []
    lib.cdb_cached_memory_bytes.restype = u64
    lib.cdb_debug_verify.argtypes = [vp, C.POINTER(u64)]
    lib.cdb_debug_verify_reference.argtypes = [vp, C.POINTER(u64)]
    lib.cdb_debug_self_check.argtypes = [vp, C.c_int, C.POINTER(u64)]
    lib.cdb_proof_wait.argtypes = [vp, C.c_double]
    lib.cdb_proof_wait.restype = C.c_int
    lib.cdb_debug_radix_sort.argtypes = [C.c_int, vp, vp, u64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_int)]
    lib.cdb_shards_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    lib.cdb_shards_destroy.argtypes = [vp]
    lib.cdb_shards_destroy.restype = None
    lib.cdb_shards_last_error.argtypes = [vp]
    lib.cdb_shards_last_error.restype = cp
    lib.cdb_shards_add.argtypes = [vp, i64, cp, C.c_size_t]
    lib.cdb_shards_add_bulk.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_shards_set_option.argtypes = [vp, cp, i64]
    lib.cdb_shards_build.argtypes = [vp]
    lib.cdb_shards_query.argtypes = [vp, cp, C.c_size_t, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)), C.POINTER(C.c_size_t)]
    lib.cdb_shards_query_batch.argtypes = [vp, vp, vp, u64, C.POINTER(CdbResult)]
    lib.cdb_shards_query_or.argtypes = [vp, vp, vp, u64, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)), C.POINTER(C.c_size_t)]
    lib.cdb_shards_query_ranked.argtypes = [vp, vp, vp, u64, i64, i64, u64, C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(i64)),
                                            C.POINTER(C.c_size_t)]
    lib.cdb_shards_query_spans.argtypes = [vp, vp, vp, u64, C.POINTER(CdbSpans)]
    lib.cdb_shards_count.argtypes = [vp]
    lib.cdb_shards_get.argtypes = [vp, C.c_int]
    lib.cdb_shards_get.restype = vp
    lib.cdb_shards_first_doc.argtypes = [vp, C.c_int]
    lib.cdb_shards_first_doc.restype = u64
    lib.cdb_shards_transport.argtypes = [vp]
    lib.cdb_shards_transport.restype = cp
    lib.cdb_comm_unique_id.argtypes = [vp]
    lib.cdb_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int]
    lib.cdb_reserve.argtypes = [C.c_int, u64, u64, C.c_char_p, C.c_size_t]
    lib.cdb_reserve_wait.restype = None
    lib.cdb_comm_create_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    lib.cdb_comm_destroy.argtypes = [vp]
    lib.cdb_comm_destroy.restype = None
    lib.cdb_comm_last_error.argtypes = [vp]
    lib.cdb_comm_last_error.restype = cp
    lib.cdb_comm_merge.argtypes = [vp, C.POINTER(CdbDeviceResult), C.POINTER(CdbDeviceResult)]
    lib.cdb_comm_merge_counts.argtypes = [vp, C.POINTER(CdbDeviceResult), C.POINTER(CdbShardSlice)]
    lib.cdb_comm_world.argtypes = [vp]
    lib.cdb_comm_transport.argtypes = [vp]
    lib.cdb_comm_transport.restype = cp
    lib.cdb_build_view.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_build_views.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_shards_build_views.argtypes = [vp, vp, vp, vp, u64]
    lib.cdb_shards_query_batch_offsets.argtypes = [vp, vp, vp, u64, C.POINTER(CdbResult), C.POINTER(CdbHits)]
    lib.cdb_shards_query_and.argtypes = [C.POINTER(CdbKeyQuery), C.c_int, C.c_int, i64, i64, u64, C.POINTER(C.POINTER(i64)),
                                         C.POINTER(C.POINTER(i64)), C.POINTER(C.c_size_t)]
    lib.cdb_shards_add_raw_dir.argtypes = [vp, cp, cp, C.POINTER(u64), C.POINTER(u64)]
    lib.cdb_shards_save.argtypes = [vp, cp]
    lib.cdb_shards_load.argtypes = [vp, cp]
    _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class GpuStringIndex:
    """string_index on the GPU: add() / build() / query() as in the reference, plus query_batch()."""

    def __init__(self, device=-1):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.cdb_create(C.byref(h), device)
        if rc != 0:
            raise RuntimeError(f"cdb_create failed (code {rc}): no usable gfx950 device — there is no CPU fallback")
        self._h = h
        # measurement plumbing of THIS binding (tools/, bench.py A/B runs): CDB_OPTIONS="name=value,..." is applied through
        # cdb_set_option; the library itself reads no option from the environment
        for kv in filter(None, os.environ.get("CDB_OPTIONS", "").split(",")):
            name, _, val = kv.partition("=")
            self._lib.cdb_set_option(self._h, name.strip().encode(), int(val))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cdb_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.cdb_last_error(self._h).decode(errors="replace"))

    # ---- reference surface
    def add(self, id_, value: bytes):
        self._check(self._lib.cdb_add(self._h, int(id_), value, len(value)))

    def add_bulk(self, ids, blob, doc_start):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
        assert len(doc_start) == len(ids) + 1
        if len(doc_start) and int(doc_start[-1]) > len(blob):   # (the C ABI takes plain pointers: a short blob would be read past its end)
            raise ValueError(f"blob holds {len(blob)} bytes, doc_start[-1] = {int(doc_start[-1])}")
        self._check(self._lib.cdb_add_bulk(self._h, _ptr(ids), _ptr(blob), _ptr(doc_start), len(ids)))

    def build(self):
        self._check(self._lib.cdb_build(self._h))

    def build_view(self, ids, blob, doc_start):
        """cdb_build_view: build straight from the caller's host column (no staging copy inside the handle)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
        assert len(doc_start) == len(ids) + 1
        if len(doc_start) and int(doc_start[-1]) > len(blob):   # (the C ABI takes plain pointers: a short blob would be read past its end)
            raise ValueError(f"blob holds {len(blob)} bytes, doc_start[-1] = {int(doc_start[-1])}")
        self._check(self._lib.cdb_build_view(self._h, _ptr(ids), _ptr(blob) if len(blob) else None, _ptr(doc_start), len(ids)))

    def build_views(self, ids, docs):
        """cdb_build_views: documents as separate bytes objects (string_index's views), gathered by the library."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        ptrs = (C.c_char_p * len(docs))(*docs)
        lens = np.array([len(d) for d in docs], dtype=np.uint64)
        self._check(self._lib.cdb_build_views(self._h, _ptr(ids), C.cast(ptrs, C.c_void_p), _ptr(lens), len(docs)))

    def add_raw_record(self, key: bytes, record: bytes):
        self._check(self._lib.cdb_add_raw_record(self._h, key, record, len(record)))

    def add_raw_dir(self, directory, key: bytes):
        """Every record file of `directory` (ascending name order); returns (records parsed, documents added)."""
        nrec, nadd = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.cdb_add_raw_dir(self._h, os.fsencode(directory), key, C.byref(nrec), C.byref(nadd)))
        return nrec.value, nadd.value

    def save(self, path):
        self._check(self._lib.cdb_save(self._h, os.fsencode(path)))

    def load(self, path):
        self._check(self._lib.cdb_load(self._h, os.fsencode(path)))

    def build_device(self, d_text_ptr, doc_start, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
        self._check(self._lib.cdb_build_device(self._h, C.c_void_p(d_text_ptr), _ptr(doc_start), _ptr(ids), len(ids)))

    def build_resident(self, d_text_ptr, d_doc_start_ptr, d_ids_ptr, ndocs):
        """Build over text AND document tables that already live in device memory (u64[ndocs+1], i64[ndocs])."""
        self._check(self._lib.cdb_build_resident(self._h, C.c_void_p(d_text_ptr), C.c_void_p(d_doc_start_ptr),
                                                 C.c_void_p(d_ids_ptr), int(ndocs)))

    def query(self, kw: bytes):
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        self._check(self._lib.cdb_query(self._h, kw, len(kw), C.byref(ids), C.byref(cnt), C.byref(n)))
        out = [(ids[i], cnt[i]) for i in range(n.value)]
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    def query_or(self, keywords):
        """Union over `keywords` by object id with summed counts, ascending id (interface.cpp:78-113)."""
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        self._check(self._lib.cdb_query_or(self._h, _ptr(blob) if len(blob) else None, _ptr(offs), len(keywords),
                                           C.byref(ids), C.byref(cnt), C.byref(n)))
        out = [(ids[i], cnt[i]) for i in range(n.value)]
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    def query_ranked(self, keywords, lo=1, hi=(1 << 62), limit=0):
        """query_or, filtered to lo <= count < hi, ranked by descending count (ties: ascending id), first `limit` rows."""
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        self._check(self._lib.cdb_query_ranked(self._h, _ptr(blob) if len(blob) else None, _ptr(offs), len(keywords),
                                               int(lo), int(hi), int(limit), C.byref(ids), C.byref(cnt), C.byref(n)))
        out = [(ids[i], cnt[i]) for i in range(n.value)]
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    def query_ranked_arrays(self, blob, offsets, lo=1, hi=(1 << 62), limit=0, rows=False):
        """cdb_query_ranked over an already packed keyword list (bench: 10^5 keywords); returns the number of rows
        (rows=True: the (ids, counts) arrays)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        self._check(self._lib.cdb_query_ranked(self._h, _ptr(blob), _ptr(offsets), len(offsets) - 1, int(lo), int(hi), int(limit),
                                               C.byref(ids), C.byref(cnt), C.byref(n)))
        out = int(n.value)
        if rows:
            out = (np.array(ids[:n.value], dtype=np.int64), np.array(cnt[:n.value], dtype=np.int64))
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    def query_spans(self, keywords):
        """{object id: [(begin, end_inclusive), ...]} — merged highlight spans (database.cpp:58-76)."""
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        r = CdbSpans()
        self._check(self._lib.cdb_query_spans(self._h, _ptr(blob) if len(blob) else None, _ptr(offs), len(keywords),
                                              C.byref(r)))
        try:
            out = []
            for d in range(r.ndocs):
                a, b = r.span_ptr[d], r.span_ptr[d + 1]
                out.append((r.ids[d], [(r.begin[k], r.end[k]) for k in range(a, b)]))
            return out
        finally:
            self._lib.cdb_spans_free(C.byref(r))

    # ---- batched
    @staticmethod
    def _adopt(ptr, n, dtype, owner):
        """numpy view of a C result array of n elements; large arrays are adopted without a copy (the view
        keeps `owner` alive, whose finaliser hands the pinned blocks back to the library)."""
        n = int(n)
        if n == 0:
            return np.empty(0, dtype=dtype)
        nbytes = n * np.dtype(dtype).itemsize
        buf = (C.c_uint8 * nbytes).from_address(C.addressof(ptr.contents))
        if nbytes < (1 << 20):
            return np.frombuffer(buf, dtype=dtype).copy()
        buf._owner = owner
        return np.frombuffer(buf, dtype=dtype)

    class _Owner:
        def __init__(self, lib, *frees):
            self._frees = [(getattr(lib, fn), C.byref(obj), obj) for fn, obj in frees]

        def __del__(self):
            for fn, ref, _obj in self._frees:
                fn(ref)

    def query_batch(self, blob, offsets):
        """Returns (row_ptr uint64[npat+1], ids int64[nrows], counts int64[nrows], nhits)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npat = len(offsets) - 1
        r = CdbResult()
        self._check(self._lib.cdb_query_batch(self._h, _ptr(blob), _ptr(offsets), npat, C.byref(r)))
        own = self._Owner(self._lib, ("cdb_result_free", r))
        nrows = int(r.nrows)
        return (self._adopt(r.row_ptr, npat + 1, np.uint64, own), self._adopt(r.ids, nrows, np.int64, own),
                self._adopt(r.counts, nrows, np.int64, own), int(r.nhits))

    def query_batch_offsets(self, blob, offsets):
        """query_batch plus (hit_ptr uint64[nrows+1], occurrence offsets uint64[nhits])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npat = len(offsets) - 1
        r, hx = CdbResult(), CdbHits()
        self._check(self._lib.cdb_query_batch_offsets(self._h, _ptr(blob), _ptr(offsets), npat, C.byref(r), C.byref(hx)))
        own = self._Owner(self._lib, ("cdb_result_free", r), ("cdb_hits_free", hx))
        nrows, nhits = int(r.nrows), int(r.nhits)
        return (self._adopt(r.row_ptr, npat + 1, np.uint64, own), self._adopt(r.ids, nrows, np.int64, own),
                self._adopt(r.counts, nrows, np.int64, own), self._adopt(hx.hit_ptr, nrows + 1, np.uint64, own),
                self._adopt(hx.offsets, nhits, np.uint64, own))

    def query_batch_device(self, d_blob_ptr, d_offsets_ptr, npat, blob_bytes):
        r = CdbDeviceResult()
        self._check(self._lib.cdb_query_batch_device(self._h, C.c_void_p(d_blob_ptr), C.c_void_p(d_offsets_ptr), npat,
                                                     blob_bytes, C.byref(r)))
        return r

    def query_batch_offsets_device(self, d_blob_ptr, d_offsets_ptr, npat, blob_bytes):
        r, hx = CdbDeviceResult(), CdbDeviceHits()
        self._check(self._lib.cdb_query_batch_offsets_device(self._h, C.c_void_p(d_blob_ptr), C.c_void_p(d_offsets_ptr), npat,
                                                             blob_bytes, C.byref(r), C.byref(hx)))
        return r, hx

    # ---- introspection / options / measurements
    size = property(lambda s: s._lib.cdb_size(s._h))
    bits = property(lambda s: s._lib.cdb_bits(s._h))
    mask = property(lambda s: s._lib.cdb_mask(s._h))
    sa_width = property(lambda s: s._lib.cdb_sa_width(s._h))

    def sa(self):
        n, w = self.size, self.sa_width
        out = np.empty(n, dtype=np.uint32 if w == 4 else np.uint64)
        if n:
            self._check(self._lib.cdb_sa_copy(self._h, _ptr(out), out.nbytes))
        return out

    def verify(self):
        """GPU-side structural check of the suffix array (see cdb_debug_verify)."""
        out = (C.c_uint64 * 5)()
        self._check(self._lib.cdb_debug_verify(self._h, out))
        return {"inversions": out[0], "tie_violations": out[1], "entry_sum": out[2], "invalid_entries": out[3],
                "expected_entry_sum": out[4]}

    def verify_reference(self):
        """GPU-side check of the REFERENCE's order (signed child order inside radix nodes; cdb_debug_verify_reference)."""
        out = (C.c_uint64 * 4)()
        self._check(self._lib.cdb_debug_verify_reference(self._h, out))
        return {"violations": out[0], "mixed_pairs": out[1], "radix_node_pairs": out[2], "tie_violations": out[3]}

    def proof_wait(self, timeout_ms=-1.0):
        """State of the order proof behind the last build / load (cdb_proof_wait): 2 proved, 3 damage found and repaired, ..."""
        return int(self._lib.cdb_proof_wait(self._h, float(timeout_ms)))

    def self_check(self, full=False):
        """(pairs out of order, invalid entries) among 2^15 random adjacent pairs, or among all of them (cdb_debug_self_check)."""
        out = (C.c_uint64 * 2)()
        self._check(self._lib.cdb_debug_self_check(self._h, 1 if full else 0, out))
        return int(out[0]), int(out[1])

    def query_latency_us(self, keywords, reps=32):
        """median microseconds per cdb_query call of every keyword, measured inside the library (no binding overhead)"""
        blob = b"".join(keywords)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(k) for k in keywords])
        out = np.zeros(len(keywords), dtype=np.float64)
        self._lib.cdb_debug_query_latency.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self._check(self._lib.cdb_debug_query_latency(self._h, blob, _ptr(offs), len(keywords), reps, _ptr(out)))
        return out

    def set_option(self, name, value):
        self._check(self._lib.cdb_set_option(self._h, name.encode(), int(value)))

    def stat(self, name):
        v = C.c_double(0)
        if self._lib.cdb_get_stat(self._h, name.encode(), C.byref(v)) != 0:
            raise KeyError(name)
        return v.value

    def profile(self):
        buf = C.create_string_buffer(1 << 16)
        self._lib.cdb_profile_dump(self._h, buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, launches, nbytes = line.split()
            out[name] = {"ms": float(ms), "launches": int(launches), "bytes": int(nbytes)}
        return out

    def profile_reset(self):
        self._lib.cdb_profile_reset(self._h)


class _BorrowedIndex(GpuStringIndex):
    """A shard handle owned by its cdb_shards object: never destroyed from here."""

    def close(self):
        self._h = None

    __del__ = close


class GpuShards:
    """cdb_shards: the string index spread over several GPUs of one process (devices may repeat on a one-GPU box)."""

    def __init__(self, devices):
        self._lib = load_library()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self._lib.cdb_shards_create(C.byref(h), arr, len(devices))
        if rc != 0:
            raise RuntimeError(f"cdb_shards_create failed (code {rc}): no usable gfx950 device — there is no CPU fallback")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cdb_shards_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.cdb_shards_last_error(self._h).decode(errors="replace"))

    def add(self, id_, value: bytes):
        self._check(self._lib.cdb_shards_add(self._h, int(id_), value, len(value)))

    def add_bulk(self, ids, blob, doc_start):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
        assert len(doc_start) == len(ids) + 1
        if len(doc_start) and int(doc_start[-1]) > len(blob):   # (the C ABI takes plain pointers: a short blob would be read past its end)
            raise ValueError(f"blob holds {len(blob)} bytes, doc_start[-1] = {int(doc_start[-1])}")
        self._check(self._lib.cdb_shards_add_bulk(self._h, _ptr(ids), _ptr(blob), _ptr(doc_start), len(ids)))

    def set_option(self, name, value):
        self._check(self._lib.cdb_shards_set_option(self._h, name.encode(), int(value)))

    def build(self):
        self._check(self._lib.cdb_shards_build(self._h))

    def build_views(self, ids, docs):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        ptrs = (C.c_char_p * len(docs))(*docs)
        lens = np.array([len(d) for d in docs], dtype=np.uint64)
        self._check(self._lib.cdb_shards_build_views(self._h, _ptr(ids), C.cast(ptrs, C.c_void_p), _ptr(lens), len(docs)))

    count = property(lambda s: s._lib.cdb_shards_count(s._h))
    transport = property(lambda s: s._lib.cdb_shards_transport(s._h).decode())

    def first_doc(self, i):
        return int(self._lib.cdb_shards_first_doc(self._h, i))

    def shard(self, i):
        """Borrowed GpuStringIndex view of shard i (do not close it)."""
        g = _BorrowedIndex.__new__(_BorrowedIndex)
        g._lib = self._lib
        g._h = C.c_void_p(self._lib.cdb_shards_get(self._h, i))
        return g

    def query(self, kw: bytes):
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        self._check(self._lib.cdb_shards_query(self._h, kw, len(kw), C.byref(ids), C.byref(cnt), C.byref(n)))
        out = [(ids[i], cnt[i]) for i in range(n.value)]
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    @staticmethod
    def _pack(keywords):
        blob = np.frombuffer(b"".join(keywords), dtype=np.uint8)
        offs = np.zeros(len(keywords) + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keywords], out=offs[1:])
        return blob, offs

    def query_or(self, keywords, ranked=False, lo=1, hi=(1 << 62), limit=0):
        blob, offs = self._pack(keywords)
        ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
        bp = _ptr(blob) if len(blob) else None
        if ranked:
            self._check(self._lib.cdb_shards_query_ranked(self._h, bp, _ptr(offs), len(keywords), int(lo), int(hi), int(limit),
                                                          C.byref(ids), C.byref(cnt), C.byref(n)))
        else:
            self._check(self._lib.cdb_shards_query_or(self._h, bp, _ptr(offs), len(keywords), C.byref(ids), C.byref(cnt), C.byref(n)))
        out = [(ids[i], cnt[i]) for i in range(n.value)]
        self._lib.cdb_free(ids)
        self._lib.cdb_free(cnt)
        return out

    def query_spans(self, keywords):
        blob, offs = self._pack(keywords)
        r = CdbSpans()
        self._check(self._lib.cdb_shards_query_spans(self._h, _ptr(blob) if len(blob) else None, _ptr(offs), len(keywords), C.byref(r)))
        try:
            return [(r.ids[d], [(r.begin[k], r.end[k]) for k in range(r.span_ptr[d], r.span_ptr[d + 1])]) for d in range(r.ndocs)]
        finally:
            self._lib.cdb_spans_free(C.byref(r))

    def query_batch(self, blob, offsets):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npat = len(offsets) - 1
        r = CdbResult()
        self._check(self._lib.cdb_shards_query_batch(self._h, _ptr(blob), _ptr(offsets), npat, C.byref(r)))
        own = GpuStringIndex._Owner(self._lib, ("cdb_result_free", r))
        nrows = int(r.nrows)
        ad = GpuStringIndex._adopt
        return (ad(r.row_ptr, npat + 1, np.uint64, own), ad(r.ids, nrows, np.int64, own), ad(r.counts, nrows, np.int64, own),
                int(r.nhits))

    def query_batch_offsets(self, blob, offsets):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npat = len(offsets) - 1
        r, hx = CdbResult(), CdbHits()
        self._check(self._lib.cdb_shards_query_batch_offsets(self._h, _ptr(blob), _ptr(offsets), npat, C.byref(r), C.byref(hx)))
        own = GpuStringIndex._Owner(self._lib, ("cdb_result_free", r), ("cdb_hits_free", hx))
        nrows, nhits = int(r.nrows), int(r.nhits)
        ad = GpuStringIndex._adopt
        return (ad(r.row_ptr, npat + 1, np.uint64, own), ad(r.ids, nrows, np.int64, own), ad(r.counts, nrows, np.int64, own),
                ad(hx.hit_ptr, nrows + 1, np.uint64, own), ad(hx.offsets, nhits, np.uint64, own))

    def add_raw_dir(self, directory, key: bytes):
        nrec, nadd = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.cdb_shards_add_raw_dir(self._h, os.fsencode(directory), key, C.byref(nrec), C.byref(nadd)))
        return nrec.value, nadd.value

    def save(self, path):
        self._check(self._lib.cdb_shards_save(self._h, os.fsencode(path)))

    def load(self, path):
        self._check(self._lib.cdb_shards_load(self._h, os.fsencode(path)))


class ShardComm:
    """cdb_comm: one rank of a one-process-per-GPU group; merge() is collective."""

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = np.zeros(128, dtype=np.uint8)
        if lib.cdb_comm_unique_id(_ptr(buf)) != 0:
            raise RuntimeError("cdb_comm_unique_id failed: RCCL is not available")
        return buf

    def __init__(self, uid, rank, world, device):
        self._lib = load_library()
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        h = C.c_void_p()
        rc = self._lib.cdb_comm_create(C.byref(h), _ptr(uid), rank, world, device)
        if rc != 0:
            raise RuntimeError(f"cdb_comm_create failed (code {rc})")
        self._h = h

    @classmethod
    def group(cls, devices):
        """The ranks of ONE process (cdb_comm_create_group): one ShardComm per entry of `devices`; a host thread per rank
        calls the collectives.  Ranks sharing a device exchange through device copies."""
        lib = load_library()
        devs = (C.c_int * len(devices))(*devices)
        hs = (C.c_void_p * len(devices))()
        rc = lib.cdb_comm_create_group(hs, len(devices), devs)
        if rc != 0:
            raise RuntimeError(f"cdb_comm_create_group failed (code {rc})")
        out = []
        for h in hs:
            c = cls.__new__(cls)
            c._lib, c._h = lib, C.c_void_p(h)
            out.append(c)
        return out

    def merge(self, local: "CdbDeviceResult"):
        out = CdbDeviceResult()
        if self._lib.cdb_comm_merge(self._h, C.byref(local), C.byref(out)) != 0:
            raise RuntimeError(self._lib.cdb_comm_last_error(self._h).decode(errors="replace"))
        return out

    def merge_counts(self, local: "CdbDeviceResult"):
        """Counts-only collective: merged row_ptr + this rank's first merged row per pattern (device arrays)."""
        out = CdbShardSlice()
        if self._lib.cdb_comm_merge_counts(self._h, C.byref(local), C.byref(out)) != 0:
            raise RuntimeError(self._lib.cdb_comm_last_error(self._h).decode(errors="replace"))
        return out

    world = property(lambda s: s._lib.cdb_comm_world(s._h))
    transport = property(lambda s: s._lib.cdb_comm_transport(s._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cdb_comm_destroy(self._h)
            self._h = None

    __del__ = close


def memory_stats():
    """(in_use, peak, cached) bytes of device memory as the library's allocator sees the process (cdb_memory_stats)."""
    lib = load_library()
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    lib.cdb_memory_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def memory_reset_peak():
    load_library().cdb_memory_reset_peak()


def query_and(keys, ranked=False, lo=1, hi=(1 << 62), limit=0):
    """AND across keys on the device (cdb_query_and / cdb_shards_query_and).  keys: (GpuStringIndex or GpuShards,
    [keywords]) for a string column or (None, [(id, count), ...]) for rows resolved elsewhere (ascending id).  Returns
    [(id, summed count), ...]."""
    lib = load_library()
    sharded = any(isinstance(ix, GpuShards) for ix, _ in keys)
    arr = (CdbKeyQuery * len(keys))()
    keep = []
    lead = None
    for k, (ix, data) in enumerate(keys):
        if ix is not None and not isinstance(ix, (GpuShards, GpuStringIndex)):
            raise TypeError(f"query_and: key {k} is neither a GpuStringIndex, a GpuShards nor None")
        if sharded and isinstance(ix, GpuStringIndex):
            # a plain index beside sharded keys: cdb_shards_key_query.shards must be a cdb_shards* — handing it a cdb_index*
            # would be reinterpreted.  Resolve the key with its own OR (ascending id) and pass the rows.
            if not data:
                raise RuntimeError("The constraint list cannot be empty")
            ix, data = None, ix.query_or(list(data))
        if ix is not None:
            if lead is None or (sharded and not isinstance(lead, GpuShards)):
                lead = ix
            blob = np.frombuffer(b"".join(data), dtype=np.uint8)
            offs = np.zeros(len(data) + 1, dtype=np.uint64)
            np.cumsum([len(x) for x in data], out=offs[1:])
            keep += [blob, offs]
            arr[k].index = ix._h
            arr[k].blob = blob.ctypes.data if len(blob) else None
            arr[k].offsets = offs.ctypes.data
            arr[k].nkw = len(data)
        else:
            ri = np.ascontiguousarray([r[0] for r in data], dtype=np.int64)
            rc = np.ascontiguousarray([r[1] for r in data], dtype=np.int64)
            keep += [ri, rc]
            arr[k].ids = ri.ctypes.data if len(ri) else None
            arr[k].counts = rc.ctypes.data if len(rc) else None
            arr[k].nrows = len(ri)
    ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
    fn = lib.cdb_shards_query_and if sharded else lib.cdb_query_and  # (cdb_shards_key_query has cdb_key_query's layout)
    rc_ = fn(arr, len(keys), 1 if ranked else 0, int(lo), int(hi), int(limit), C.byref(ids), C.byref(cnt), C.byref(n))
    if rc_ != 0:
        err = lib.cdb_shards_last_error if isinstance(lead, GpuShards) else lib.cdb_last_error
        raise RuntimeError(err(lead._h).decode(errors="replace") if lead is not None else "cdb_query_and: no string key")
    out = [(ids[i], cnt[i]) for i in range(n.value)]
    lib.cdb_free(ids)
    lib.cdb_free(cnt)
    return out


def debug_radix_sort(d_keys_ptr, d_vals_ptr, n, val_bytes, key_bits, variant=0, device=-1):
    """In-place device sort through the library's radix primitive; returns (onesweep_ms, passes)."""
    lib = load_library()
    ms, passes = C.c_double(0), C.c_int(0)
    rc = lib.cdb_debug_radix_sort(device, C.c_void_p(d_keys_ptr), C.c_void_p(d_vals_ptr) if d_vals_ptr else None, n,
                                  val_bytes, key_bits, variant, C.byref(ms), C.byref(passes))
    if rc != 0:
        raise RuntimeError(f"cdb_debug_radix_sort failed ({rc})")
    return ms.value, passes.value


def layout_rule(ndocs, longest):
    """(bits, mask, width, off_bits) of the reference's entry layout, or RuntimeError with the reference's message."""
    lib = load_library()
    lib.cdb_layout_rule.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    bits, mask, width, off = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_int(0)
    err = C.create_string_buffer(256)
    if lib.cdb_layout_rule(ndocs, longest, C.byref(bits), C.byref(mask), C.byref(width), C.byref(off), err, 256) != 0:
        raise RuntimeError(err.value.decode())
    return bits.value, mask.value, width.value, off.value


def raw_record_find_string(record: bytes, key: bytes):
    """(id, value) of the string stored under `key` in one CoffeeDB raw record, or None."""
    lib = load_library()
    id_, val, n = C.c_int64(0), C.c_char_p(), C.c_size_t(0)
    buf = C.create_string_buffer(record, len(record))
    r = lib.cdb_raw_record_find_string(buf, len(record), key, C.byref(id_), C.byref(val), C.byref(n))
    if r < 0:
        raise ValueError("malformed raw record")
    if r == 0:
        return None
    off = C.cast(val, C.c_void_p).value - C.addressof(buf)
    return id_.value, record[off:off + n.value]
