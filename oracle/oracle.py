"""ctypes binding of the CPU oracle (oracle/libcoltt_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product package (coltt_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcoltt_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libcoltt_ref_simd.so")

ORDER_AVX, ORDER_SSE, ORDER_NATIVE = 0, 1, 2
COSINE, L2 = 0, 1
Q_NONE, Q_F16, Q_F8, Q_BF16 = 0, 1, 2, 3
QUANT_DTYPE = {Q_NONE: np.float32, Q_F16: np.uint16, Q_F8: np.uint8, Q_BF16: np.uint16}


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "coltt_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "libcoltt_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/pkg/distance/simd/cpp/avx.cpp") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class HnswCfg(C.Structure):
    _fields_ = [("m", C.c_int32), ("mMax", C.c_int32), ("mMax0", C.c_int32), ("ef", C.c_int32),
                ("efConstruction", C.c_int32), ("algo", C.c_int32), ("levelMultiplier", C.c_float),
                ("extendCandidates", C.c_int32), ("keepPruned", C.c_int32)]


def default_cfg(**kw):
    """newHnswConfig defaults (core/vectorindex/hnsw_config.go:135-162)."""
    c = HnswCfg(16, -1, -1, 20, 200, 0, -1.0, 0, 1)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


_lib = None
_ref = None
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
_vp = C.c_void_p


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_l2.restype = C.c_float
        L.orc_cosine.restype = C.c_float
        L.orc_l2sq.restype = C.c_float
        L.orc_manhattan.restype = C.c_float
        for n in ("orc_pq_dot", "orc_pq_l2sq", "orc_pq_dot_pure", "orc_pq_l2sq_pure", "orc_pq_hamming",
                  "orc_pq_jaccard"):
            getattr(L, n).restype = C.c_float
        L.orc_shard_vertex.restype = C.c_uint64
        L.orc_shard_vertex.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_flat_create.restype = _vp
        L.orc_flat_len.restype = C.c_uint64
        L.orc_hnsw_create.restype = _vp
        L.orc_hnsw_len.restype = C.c_uint64
        L.orc_hnsw_slots.restype = C.c_int64
        L.orc_hnsw_entry.restype = C.c_int32
        L.orc_hnsw_export.restype = C.c_int64
        L.orc_hnsw_graph_hash.restype = C.c_uint64
        L.orc_level.argtypes = [C.c_uint64, C.c_uint64, C.c_float]
        L.orc_level_from_u.argtypes = [C.c_float, C.c_float]
        L.orc_fill_normal.argtypes = [C.c_uint64, C.c_uint64, _vp, C.c_size_t]
        _lib = L
    return _lib


def ref():
    """The reference's own avx.cpp / sse.cpp (oracle/_ref), or None when it was never built."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(_REF):
            return None
        _ref = C.CDLL(_REF)
    return _ref


def _p(a):
    return a.ctypes.data_as(_vp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ kernels
def l2(a, b, order=ORDER_AVX):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_l2(order, _p(a), _p(b), C.c_size_t(a.size)))


def manhattan(a, b, order=ORDER_AVX):
    """Manhattan.Distance (pkg/distance/space.go:77-79) in the avx.cpp / sse.cpp / native order"""
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_manhattan(order, _p(a), _p(b), C.c_size_t(a.size)))


def cosine(a, b, order=ORDER_AVX):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_cosine(order, _p(a), _p(b), C.c_size_t(a.size)))


def l2sq(a, b, order=ORDER_AVX):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_l2sq(order, _p(a), _p(b), C.c_size_t(a.size)))


def cosine_parts(a, b, order=ORDER_AVX):
    a, b = _f32(a), _f32(b)
    o = (C.c_float * 3)()
    lib().orc_cosine_parts(order, _p(a), _p(b), C.c_size_t(a.size), C.byref(o, 0), C.byref(o, 4), C.byref(o, 8))
    return np.float32(o[0]), np.float32(o[1]), np.float32(o[2])


def dist_rows(metric, q, rows, order=ORDER_AVX):
    q, rows = _f32(q), _f32(rows)
    out = np.empty(rows.shape[0], np.float32)
    lib().orc_dist_rows(metric, order, _p(q), _p(rows), C.c_size_t(rows.shape[0]), C.c_size_t(rows.shape[1]), _p(out))
    return out


def normalize(v):
    v = _f32(v)
    out = np.empty_like(v)
    if v.ndim == 1:
        lib().orc_normalize(_p(v), _p(out), C.c_size_t(v.size))
    else:
        for i in range(v.shape[0]):
            lib().orc_normalize(_p(v[i]), _p(out[i]), C.c_size_t(v.shape[1]))
    return out


def f16_encode(x):
    x = _f32(x); o = np.empty(x.shape, np.uint16); lib().orc_f16_encode(_p(x), _p(o), C.c_size_t(x.size)); return o


def f16_decode(x):
    x = np.ascontiguousarray(x, np.uint16); o = np.empty(x.shape, np.float32)
    lib().orc_f16_decode(_p(x), _p(o), C.c_size_t(x.size)); return o


def f8_encode(x):
    x = _f32(x); o = np.empty(x.shape, np.uint8); lib().orc_f8_encode(_p(x), _p(o), C.c_size_t(x.size)); return o


def f8_decode(x):
    x = np.ascontiguousarray(x, np.uint8); o = np.empty(x.shape, np.float32)
    lib().orc_f8_decode(_p(x), _p(o), C.c_size_t(x.size)); return o


def lower(quant, v):
    v = _f32(v); o = np.empty(v.shape, QUANT_DTYPE[quant])
    if v.ndim == 1:
        lib().orc_lower(quant, _p(v), C.c_size_t(v.size), _p(o))
    else:
        for i in range(v.shape[0]):
            lib().orc_lower(quant, _p(v[i]), C.c_size_t(v.shape[1]), _p(o[i]))
    return o


def shard_vertex(i, c=16):
    return int(lib().orc_shard_vertex(int(i), int(c)))


def pq_dot(x, y, pure=False):
    x, y = _f32(x), _f32(y)
    return np.float32((lib().orc_pq_dot_pure if pure else lib().orc_pq_dot)(_p(x), _p(y), C.c_size_t(x.size)))


def pq_l2sq(x, y, pure=False):
    x, y = _f32(x), _f32(y)
    return np.float32((lib().orc_pq_l2sq_pure if pure else lib().orc_pq_l2sq)(_p(x), _p(y), C.c_size_t(x.size)))


def pq_hamming(x, y):
    x, y = np.ascontiguousarray(x, np.uint64), np.ascontiguousarray(y, np.uint64)
    return np.float32(lib().orc_pq_hamming(_p(x), _p(y), C.c_size_t(x.size)))


def pq_jaccard(x, y):
    x, y = np.ascontiguousarray(x, np.uint64), np.ascontiguousarray(y, np.uint64)
    return np.float32(lib().orc_pq_jaccard(_p(x), _p(y), C.c_size_t(x.size)))


def heap_trace(is_max, prios, ops):
    prios = _f32(prios); ops = np.ascontiguousarray(ops, np.int32)
    pops = np.empty(len(ops), np.int32); fin = np.empty(len(ops), np.int32); nf = C.c_int32(0)
    n = lib().orc_heap_trace(int(is_max), _p(prios), _p(ops), C.c_size_t(len(ops)), _p(pops), _p(fin), C.byref(nf))
    return pops[:n].copy(), fin[:nf.value].copy()


def fill_normal(seed, shape, first=0):
    out = np.empty(shape, np.float32)
    lib().orc_fill_normal(int(seed), int(first), _p(out), C.c_size_t(out.size))
    return out


def level(seed, i, mult):
    return int(lib().orc_level(int(seed), int(i), C.c_float(mult)))


def level_from_u(u, mult):
    """Hnsw.RandomLevel (hnsw.go:280-282) for the uniform draw u that rand.Float32() returned."""
    return int(lib().orc_level_from_u(C.c_float(u), C.c_float(mult)))


def levels(seed, n, m=16):
    mult = np.float32(1.0) / np.float32(np.log(np.float64(np.float32(m))))
    return np.array([level(seed, i, mult) for i in range(n)], np.int32)


# ------------------------------------------------------------------ FLAT
class Flat:
    """edge.{none,f16,f8,bf16}VecSpace restated (edge/none_vectorstore.go etc.)."""

    def __init__(self, dim, metric=COSINE, quant=Q_NONE, order=ORDER_AVX):
        self.dim, self.metric, self.quant = dim, metric, quant
        self.h = _vp(lib().orc_flat_create(dim, metric, quant, order))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_flat_destroy(self.h); self.h = None

    def upsert(self, ids, vecs):
        ids = np.ascontiguousarray(ids, np.uint64); vecs = _f32(vecs)
        assert vecs.shape == (len(ids), self.dim)
        lib().orc_flat_upsert(self.h, _p(ids), _p(vecs), C.c_size_t(len(ids)))

    def remove(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64)
        lib().orc_flat_remove(self.h, _p(ids), C.c_size_t(len(ids)))

    def __len__(self):
        return int(lib().orc_flat_len(self.h))

    def get(self, id_):
        o = np.empty(self.dim, QUANT_DTYPE[self.quant])
        return o if lib().orc_flat_get(self.h, C.c_uint64(int(id_)), _p(o)) == 0 else None

    def save_vertex(self):
        """SaveVertex (edge/none_vectorstore.go:308-423) -> bytes"""
        lib().orc_flat_save.restype = C.c_int64
        n = lib().orc_flat_save(self.h, None, C.c_uint64(0))
        buf = np.empty(n, np.uint8); lib().orc_flat_save(self.h, _p(buf), C.c_uint64(n))
        return buf.tobytes()

    def load_vertex(self, data):
        """LoadVertex (edge/none_vectorstore.go:425-516)"""
        b = np.frombuffer(data, np.uint8)
        return lib().orc_flat_load(self.h, _p(b), C.c_uint64(len(b)))

    def search(self, query, k, nearest=False, mode=2, cand=None):
        """mode 0 literal (highCpu=false), 1 literal (highCpu=true), 2 canonical (score,id) order."""
        q = _f32(query)
        ids = np.empty(max(k, 1), np.uint64); sc = np.empty(max(k, 1), np.float32)
        if cand is not None:
            cand = np.ascontiguousarray(cand, np.uint64)
            n = lib().orc_flat_search(self.h, _p(q), int(k), int(nearest), mode, _p(cand), C.c_size_t(len(cand)), 1, _p(ids), _p(sc))
        else:
            n = lib().orc_flat_search(self.h, _p(q), int(k), int(nearest), mode, None, C.c_size_t(0), 0, _p(ids), _p(sc))
        return ids[:n].copy(), sc[:n].copy()


class CFlat:
    """experimental multiVectorVertex restated (experimental/multi_vector_vertex.go:60-137)."""

    def __init__(self, dim, n_fields, metric=COSINE, order=ORDER_AVX):
        self.dim, self.nf = dim, n_fields
        lib().orc_cflat_create.restype = _vp
        self.h = _vp(lib().orc_cflat_create(dim, metric, n_fields, order))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cflat_destroy(self.h); self.h = None

    def upsert(self, ids, vecs):
        ids = np.ascontiguousarray(ids, np.uint64); vecs = _f32(vecs).reshape(len(ids), self.nf, self.dim)
        lib().orc_cflat_upsert(self.h, _p(ids), _p(vecs), C.c_size_t(len(ids)))

    def remove(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64); lib().orc_cflat_remove(self.h, _p(ids), C.c_size_t(len(ids)))

    def search(self, q, ratios, include, k):
        q = _f32(q).reshape(self.nf, self.dim); r = np.ascontiguousarray(ratios, np.uint32); inc = np.ascontiguousarray(include, np.uint8)
        ids = np.empty(max(k, 1), np.uint64); sc = np.empty(max(k, 1), np.float32)
        n = lib().orc_cflat_search(self.h, _p(q), _p(r), _p(inc), int(k), _p(ids), _p(sc))
        return ids[:n].copy(), sc[:n].copy()


# ------------------------------------------------------------------ HNSW
class Hnsw:
    """core/vectorindex.Hnsw restated (core/vectorindex/hnsw.go)."""

    def __init__(self, dim, metric=COSINE, cfg=None, order=ORDER_AVX, canonical_build=False):
        self.dim, self.metric = dim, metric
        cfg = cfg or default_cfg()
        self.h = _vp(lib().orc_hnsw_create(dim, metric, order, C.byref(cfg)))
        self.cfg = HnswCfg(); lib().orc_hnsw_get_cfg(self.h, C.byref(self.cfg))
        if canonical_build:
            lib().orc_hnsw_set_canonical(self.h, 1)
        self._keep = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hnsw_destroy(self.h); self.h = None

    def insert(self, id_, vec, level):
        v = _f32(vec)
        return lib().orc_hnsw_insert(self.h, C.c_uint64(int(id_)), _p(v), int(level))

    def insert_many(self, ids, vecs, lvls):
        vecs = _f32(vecs)
        for i in range(len(ids)):
            rc = lib().orc_hnsw_insert(self.h, C.c_uint64(int(ids[i])), _p(vecs[i]), int(lvls[i]))
            assert rc == 0, rc

    def insert_batched(self, ids, vecs, lvls, batch, schedule=None):
        """GPU-builder semantics: consecutive groups of `batch` vertices are linked against a frozen graph."""
        ids = np.ascontiguousarray(ids, np.uint64); vecs = _f32(vecs); lvls = np.ascontiguousarray(lvls, np.int32)
        i = 0
        while i < len(ids):
            b = min(batch if schedule is None else schedule(i), len(ids) - i)
            rc = lib().orc_hnsw_insert_batch(self.h, _p(ids[i:i + b]), _p(vecs[i:i + b]), _p(lvls[i:i + b]), C.c_size_t(b))
            assert rc == 0, rc
            i += b

    def remove(self, id_):
        return lib().orc_hnsw_remove(self.h, C.c_uint64(int(id_)))

    def __len__(self):
        return int(lib().orc_hnsw_len(self.h))

    @property
    def entry(self):
        return int(lib().orc_hnsw_entry(self.h))

    @property
    def slots(self):
        return int(lib().orc_hnsw_slots(self.h))

    def search(self, q, k, mode=1, ef=0, with_stats=False):
        q = _f32(q)
        ids = np.empty(max(k, 1), np.uint64); sc = np.empty(max(k, 1), np.float32); sl = np.empty(max(k, 1), np.int32)
        st = (C.c_uint64 * 3)()
        n = lib().orc_hnsw_search(self.h, _p(q), int(k), mode, int(ef), _p(ids), _p(sc), _p(sl), st)
        if with_stats:
            return ids[:n].copy(), sc[:n].copy(), {"n_dist": st[0], "n_exp": st[1], "n_hops": st[2]}
        return ids[:n].copy(), sc[:n].copy()

    def commit(self, header=True):
        """Hnsw.Commit (hnsw_commit.go:69-162) -> bytes"""
        lib().orc_hnsw_commit.restype = C.c_int64
        n = lib().orc_hnsw_commit(self.h, int(header), None, C.c_uint64(0))
        buf = np.empty(n, np.uint8)
        lib().orc_hnsw_commit(self.h, int(header), _p(buf), C.c_uint64(n))
        return buf.tobytes()

    def load_stream(self, data, header=True):
        """Hnsw.Load (hnsw_commit.go:164-278)"""
        b = np.frombuffer(data, np.uint8)
        rc = lib().orc_hnsw_load(self.h, int(header), _p(b), C.c_uint64(len(b)))
        lib().orc_hnsw_get_cfg(self.h, C.byref(self.cfg))
        return rc

    def graph_hash(self):
        return int(lib().orc_hnsw_graph_hash(self.h))

    def export(self, with_vectors=True):
        n = self.slots
        ml = C.c_int32(0)
        ne = lib().orc_hnsw_export(self.h, None, None, None, None, None, None, None, C.byref(ml))
        ids = np.empty(n, np.uint64); lv = np.empty(n, np.int32); dl = np.empty(n, np.uint8)
        vec = np.empty((n, self.dim), np.float32) if with_vectors else None
        lib().orc_hnsw_export(self.h, _p(ids), _p(lv), _p(dl), None, None, None, None, C.byref(ml))
        rows = int((lv.astype(np.int64) + 1).sum())
        off = np.empty(rows + 1, np.int64); nb = np.empty(ne, np.int32); nd = np.empty(ne, np.float32)
        lib().orc_hnsw_export(self.h, _p(ids), _p(lv), _p(dl), _p(vec) if with_vectors else None, _p(off), _p(nb), _p(nd),
                              C.byref(ml))
        return {"ids": ids, "levels": lv, "deleted": dl, "vectors": vec, "row_offsets": off, "nbr": nb,
                "nbr_dist": nd, "entry": self.entry, "max_level": ml.value}

    def load(self, g, view=False):
        """import a graph dict (export() layout).  view=True keeps a reference to g['vectors'] (no copy)."""
        vec = _f32(g["vectors"])
        ids = np.ascontiguousarray(g["ids"], np.uint64); lv = np.ascontiguousarray(g["levels"], np.int32)
        dl = np.ascontiguousarray(g["deleted"], np.uint8) if g.get("deleted") is not None else None
        off = np.ascontiguousarray(g["row_offsets"], np.int64); nb = np.ascontiguousarray(g["nbr"], np.int32)
        nd = np.ascontiguousarray(g["nbr_dist"], np.float32) if g.get("nbr_dist") is not None else None
        if view:
            self._keep = (vec,)
        lib().orc_hnsw_import(self.h, C.c_int64(len(ids)), _p(ids), _p(lv), _p(dl) if dl is not None else None, _p(vec),
                              _p(off), _p(nb), _p(nd) if nd is not None else None, int(g["entry"]), int(view))


# ------------------------------------------------------------------ cpu_baseline drivers (bench.py, tests)
def cpu_count():
    return int(lib().orc_cpu_count())


def set_pin_policy(policy):
    """1 = dense (thread t on the t-th allowed CPU), 2 = spread (thread t of T on allowed[t * n / T]); applies to every pinned driver"""
    lib().orc_set_pin_policy(int(policy))


def pin_map(threads, policy):
    out = (C.c_int * max(1, threads))()
    k = lib().orc_pin_map(int(threads), int(policy), out, int(threads))
    return [int(out[i]) for i in range(k)]


class NumaArray:
    """numpy view of a page-interleaved anonymous mapping (orc_numa_alloc): mbind(MPOL_INTERLEAVE) when the kernel accepts
    it, transparent huge pages advised, first touch by `threads` pinned threads.  .flags: bit0 mbind ok, bit1 THP advised."""

    def __init__(self, shape, dtype, threads=None):
        L = lib()
        L.orc_numa_alloc.restype = _vp
        L.orc_numa_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_int)]
        L.orc_numa_free.argtypes = [_vp, C.c_size_t]
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        fl = C.c_int(0)
        self.ptr = L.orc_numa_alloc(C.c_size_t(self.nbytes), int(threads or cpu_count()), C.byref(fl))
        if not self.ptr:
            raise MemoryError(f"orc_numa_alloc({self.nbytes}) failed")
        self.flags = fl.value
        buf = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.a = None
            lib().orc_numa_free(_vp(self.ptr), C.c_size_t(self.nbytes))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csr_search(rows, quant, adj0, upper_off, adjU, dim, metric, entry, entry_level, queries, k, ef, del_bits=None,
               threads=1, pin=True, order=ORDER_AVX):
    """canonical Hnsw.Search over the GPU's HBM-layout arrays (orc_csr_search_mt).  Returns slots, scores, counts,
    {n_dist, n_exp, n_hops}, wall seconds of the parallel region."""
    q = _f32(queries).reshape(-1, dim); nq = len(q)
    sl = np.empty((nq, k), np.int32); sc = np.empty((nq, k), np.float32); cn = np.empty(nq, np.int32)
    st = (C.c_uint64 * 3)(); wall = C.c_double(0)
    adj0 = np.ascontiguousarray(adj0, np.uint32); adjU = np.ascontiguousarray(adjU, np.uint32)
    lib().orc_csr_search_mt(_p(rows), int(quant), _p(adj0), _p(np.ascontiguousarray(upper_off, np.uint32)), _p(adjU),
                            _p(del_bits) if del_bits is not None else None, C.c_uint32(adj0.shape[1]), C.c_uint32(adjU.shape[1]),
                            C.c_uint32(dim), int(metric), int(order), C.c_int32(int(entry)), C.c_int32(int(entry_level)), _p(q),
                            C.c_size_t(nq), int(k), int(ef), _p(sl), _p(sc), _p(cn), st, int(threads), int(bool(pin)), C.byref(wall))
    return sl, sc, cn, {"n_dist": st[0], "n_exp": st[1], "n_hops": st[2]}, wall.value


def csr_search_pq(rows, quant, adj0, upper_off, adjU, dim, metric, entry, entry_level, codes, codebooks, pq_metric, queries, k, ef, rerank=0,
                  del_bits=None, threads=1, pin=True, order=ORDER_AVX):
    """Product-quantised Hnsw.Search (a DEFINITION, coltt_oracle.cpp "Product-quantised HNSW"): the canonical walk with table distances
    over row-major codes [n, m], then the exact re-rank.  Returns slots, exact scores, counts, {n_dist, n_exp, n_hops, n_exact}, wall s."""
    q = _f32(queries).reshape(-1, dim); nq = len(q)
    cb = _f32(codebooks); m, c = cb.shape[0], cb.shape[1]
    codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, m)
    sl = np.empty((nq, k), np.int32); sc = np.empty((nq, k), np.float32); cn = np.empty(nq, np.int32)
    st = (C.c_uint64 * 4)(); wall = C.c_double(0)
    adj0 = np.ascontiguousarray(adj0, np.uint32); adjU = np.ascontiguousarray(adjU, np.uint32)
    rc = lib().orc_csr_search_pq_mt(_p(rows), int(quant), _p(adj0), _p(np.ascontiguousarray(upper_off, np.uint32)), _p(adjU),
                                    _p(del_bits) if del_bits is not None else None, C.c_uint32(adj0.shape[1]), C.c_uint32(adjU.shape[1]),
                                    C.c_uint32(dim), int(metric), int(order), C.c_int32(int(entry)), C.c_int32(int(entry_level)), _p(codes), _p(cb),
                                    int(m), int(c), int(pq_metric), _p(q), C.c_size_t(nq), int(k), int(ef), int(rerank), _p(sl), _p(sc), _p(cn), st,
                                    int(threads), int(bool(pin)), C.byref(wall))
    if rc != 0:
        raise ValueError("orc_csr_search_pq_mt: dim is not a multiple of the number of sub-vectors")
    return sl, sc, cn, {"n_dist": st[0], "n_exp": st[1], "n_hops": st[2], "n_exact": st[3]}, wall.value


def flat_scan(rows, quant, dim, metric, queries, k, nearest=True, shape=0, split=1, threads=1, pin=True, order=ORDER_AVX):
    """VertexSearch over contiguous stored rows (orc_flat_scan_mt).  rows: [n, dim] stored codes.  Returns slots, scores,
    counts, wall seconds."""
    q = _f32(queries).reshape(-1, dim); nq = len(q); n = len(rows)
    sl = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cn = np.zeros(nq, np.int32); wall = C.c_double(0)
    rc = lib().orc_flat_scan_mt(_p(rows), int(quant), C.c_uint64(n), C.c_uint32(dim), int(metric), int(order), _p(q), C.c_size_t(nq),
                                int(k), int(bool(nearest)), int(shape), int(split), int(threads), int(bool(pin)), _p(sl), _p(sc), _p(cn),
                                C.byref(wall))
    if rc != 0:
        raise ValueError("orc_flat_scan_mt: split > 1 needs threads == split")
    return sl, sc, cn, wall.value


def membw(arr, threads=None, reps=1, max_bytes=8 << 30):
    """streaming-read GB/s over (a prefix of) arr on pinned threads — the DRAM ceiling quoted beside the CPU legs"""
    L = lib(); L.orc_membw.restype = C.c_double
    L.orc_membw.argtypes = [_vp, C.c_size_t, C.c_int, C.c_int]
    nbytes = int(min(arr.nbytes, max_bytes))
    return float(L.orc_membw(arr.ctypes.data_as(_vp), C.c_size_t(nbytes), int(threads or cpu_count()), int(reps)))


# ------------------------------------------------------------------ product quantiser (a DEFINITION, see coltt_oracle.cpp)
PQ_COSINE, PQ_EUCLIDEAN, PQ_DOT = 0, 1, 2


def _cb(codebooks):
    cb = _f32(codebooks)
    assert cb.ndim == 3, "codebooks are [m][C][dsub]"
    return cb, cb.shape[0], cb.shape[1], cb.shape[2]


def pq_lut(metric, codebooks, query):
    """lut[j][c] = distFn(q_j, centroid[j][c]) with the store's distancepq function"""
    cb, m, c, ds = _cb(codebooks)
    q = _f32(query).reshape(m * ds)
    out = np.empty((m, c), np.float32)
    lib().orc_pq_lut(int(metric), _p(cb), m, c, ds, _p(q), _p(out))
    return out


def pq_encode(codebooks, vecs):
    cb, m, c, ds = _cb(codebooks)
    v = _f32(vecs).reshape(-1, m * ds)
    out = np.empty((len(v), m), np.uint8)
    lib().orc_pq_encode(_p(cb), m, c, ds, _p(v), C.c_size_t(len(v)), _p(out))
    return out


def pq_adc(lut, codes):
    lut = _f32(lut); codes = np.ascontiguousarray(codes, np.uint8)
    out = np.empty(len(codes), np.float32)
    lib().orc_pq_adc(_p(lut), lut.shape[0], lut.shape[1], _p(codes), C.c_size_t(len(codes)), _p(out))
    return out


def pq_train(vecs, m, c, iters):
    v = _f32(vecs); n, dim = v.shape
    assert dim % m == 0
    cb = np.empty((m, c, dim // m), np.float32)
    rc = lib().orc_pq_train(_p(cb), int(m), int(c), dim // m, _p(v), C.c_size_t(n), int(iters))
    if rc != 0:
        raise ValueError("pq_train: fewer training vectors than centroids")
    return cb


def pq_search(metric, codebooks, codes, queries, k, ids=None, threads=1, pin=False):
    """ADC top-k (k smallest in the canonical (score, id) order) over contiguous codes [n][m].  Returns ids, scores, counts, wall s."""
    cb, m, c, ds = _cb(codebooks)
    codes = np.ascontiguousarray(codes, np.uint8); n = len(codes)
    q = _f32(queries).reshape(-1, m * ds); nq = len(q)
    oi = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cn = np.zeros(nq, np.int32); wall = C.c_double(0)
    idp = None if ids is None else np.ascontiguousarray(ids, np.uint64)
    lib().orc_pq_search_mt(int(metric), _p(cb), m, c, ds, _p(codes), _p(idp) if idp is not None else None, C.c_uint64(n), _p(q), C.c_size_t(nq),
                           int(k), int(threads), int(bool(pin)), _p(oi), _p(sc), _p(cn), C.byref(wall))
    return oi, sc, cn, wall.value
