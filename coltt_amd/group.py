"""Group — one collection partitioned (or replicated) over the GPUs of a node: ctypes binding of coltt_group_*
(include/coltt_gpu.h).  Routing rule sharding.ShardVertex (pkg/sharding/shard.go:34-41); merge shape of the reference's
`highCpu` local-queue-then-global-queue scan (edge/none_vectorstore.go:148-178)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .hnsw import HnswCfg

GROUP_FLAT, GROUP_HNSW = 0, 1
LAYOUT_SHARD, LAYOUT_REPLICA = 0, 1
EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_HOST, EXCHANGE_SHM = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128


class GroupOpts(C.Structure):
    _fields_ = [("kind", C.c_int32), ("layout", C.c_int32), ("exchange", C.c_int32), ("world_size", C.c_int32),
                ("rank_base", C.c_int32), ("unique_id", C.c_void_p)]


def unique_id():
    """128 opaque bytes (ncclGetUniqueId; random bytes when librccl / a device is absent — enough for the shared-memory exchange)
    one process creates and hands to every process of a multi-process group."""
    b = (C.c_uint8 * UNIQUE_ID_BYTES)()
    L.check(L.lib().coltt_group_unique_id(b))
    return bytes(b)


def shard_vertex_host(id_, count):
    f = L.lib().coltt_shard_vertex_host
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_uint64]
    return int(f(int(id_), int(count)))


def merge_host(recs, world, nq, k, nearest=True):
    """the host-side final merge on packed records (no device): recs = structured array [world, nq, k] of
    (id u64, score f32, valid u32)."""
    recs = np.ascontiguousarray(recs)
    ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
    L.check(L.lib().coltt_group_merge_host(L.vp(recs), int(world), C.c_size_t(nq), C.c_uint32(k), int(bool(nearest)), L.vp(ids), L.vp(sc), L.vp(cnt)))
    return ids, sc, cnt


REC_DTYPE = np.dtype([("id", np.uint64), ("score", np.float32), ("valid", np.uint32)])


class ShmExchange:
    """coltt_shm_*: all-gather between the processes of one box through POSIX shared memory (host-only) — the transport of
    EXCHANGE_SHM groups.  Every process passes the same uid / world / bytes_per_rank and hosts ranks rank_base .. +n_local-1."""

    def __init__(self, uid, world, n_local, rank_base, bytes_per_rank):
        self.world, self.n_local = int(world), int(n_local)
        u = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(uid)
        h = C.c_uint64(0)
        L.check(L.lib().coltt_shm_open(u, int(world), int(n_local), int(rank_base), C.c_uint64(int(bytes_per_rank)), C.byref(h)))
        self.h = h

    def allgather(self, local):
        """local: array whose first axis is n_local -> array [world, ...] (rank-major), same dtype"""
        a = np.ascontiguousarray(local)
        assert a.shape[0] == self.n_local
        per = a.nbytes // self.n_local
        out = np.empty((self.world,) + a.shape[1:], a.dtype)
        L.check(L.lib().coltt_shm_allgather(self.h, L.vp(a), C.c_uint64(per), L.vp(out)))
        return out

    def close(self):
        if getattr(self, "h", None) is not None:
            L.lib().coltt_shm_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    def __init__(self, devices, dim, distance=L.COSINE, quantization=L.Q_NONE, kind=GROUP_FLAT, layout=LAYOUT_SHARD, cfg=None,
                 exchange=EXCHANGE_AUTO, world_size=0, rank_base=0, uid=None):
        self.dim, self.kind, self.layout = int(dim), kind, layout
        devs = (C.c_int * len(devices))(*devices)
        self._uid = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(uid) if uid is not None else None
        o = GroupOpts(kind, layout, exchange, world_size, rank_base, C.cast(self._uid, C.c_void_p) if self._uid is not None else None)
        cfg = cfg or HnswCfg.default()
        h = C.c_uint64(0)
        L.check(L.lib().coltt_group_create(devs, len(devices), C.c_uint32(dim), distance, quantization, C.byref(cfg), C.byref(o), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) is not None:
            L.lib().coltt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        a, b, c, d = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(L.lib().coltt_group_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"n_local": a.value, "world": b.value, "exchange": {1: "rccl", 2: "host", 3: "shm"}.get(c.value, c.value), "rank_base": d.value}

    def member(self, i):
        m = C.c_uint64(0)
        L.check(L.lib().coltt_group_member(self.h, int(i), C.byref(m)))
        return m

    def shard_of(self, id_):
        s = C.c_int32(0)
        L.check(L.lib().coltt_group_shard_of(self.h, C.c_uint64(int(id_)), C.byref(s)))
        return s.value

    def Len(self):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_group_len(self.h, C.byref(n)))
        return n.value

    def ChangedVertex(self, ids, vectors):
        ids = np.ascontiguousarray(ids, np.uint64); v = np.ascontiguousarray(vectors, np.float32).reshape(len(ids), self.dim)
        kept = C.c_uint64(0)
        L.check(L.lib().coltt_group_upsert(self.h, L.vp(ids), L.vp(v), C.c_size_t(len(ids)), C.byref(kept)))
        return kept.value

    def Insert(self, ids, vectors, levels, batch=1):
        ids = np.ascontiguousarray(ids, np.uint64); v = np.ascontiguousarray(vectors, np.float32).reshape(len(ids), self.dim)
        lv = np.ascontiguousarray(levels, np.int32)
        kept = C.c_uint64(0)
        L.check(L.lib().coltt_group_insert(self.h, L.vp(ids), L.vp(v), L.vp(lv), C.c_size_t(len(ids)), C.c_uint32(batch), C.byref(kept)))
        return kept.value

    def Remove(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64)
        L.check(L.lib().coltt_group_remove(self.h, L.vp(ids), C.c_size_t(len(ids))))

    def Search(self, queries, k, select=L.SELECT_NEAREST, mode=L.MODE_EXACT, ef=0):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim); nq = len(q)
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        L.check(L.lib().coltt_group_search(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(k), select, mode, C.c_uint32(ef), L.vp(ids), L.vp(sc), L.vp(cnt)))
        return ids, sc, cnt

    def SearchDevice(self, d_queries_per_member, nq, k, select=L.SELECT_NEAREST, mode=L.MODE_EXACT, ef=0, out=None):
        """queries already on every local member's device (list of device pointers); merged answers on the host."""
        ptrs = (C.c_void_p * len(d_queries_per_member))(*d_queries_per_member)
        if out is None:
            out = (np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32))
        L.check(L.lib().coltt_group_search_device(self.h, ptrs, C.c_size_t(nq), C.c_uint32(k), select, mode, C.c_uint32(ef),
                                                  L.vp(out[0]), L.vp(out[1]), L.vp(out[2])))
        return out

    def SearchBegin(self, k, queries=None, d_queries_per_member=None, nq=None, select=L.SELECT_NEAREST, mode=L.MODE_EXACT, ef=0):
        """coltt_group_search_begin: stage A now, exchange + merge queued; returns (ticket, out arrays) — keep `out` alive until SearchEnd"""
        if queries is not None:
            q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim); nq = len(q); qp = L.vp(q); ptrs = None
        else:
            q = None; qp = None; ptrs = (C.c_void_p * len(d_queries_per_member))(*d_queries_per_member)
        out = (np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32))
        t = C.c_uint64(0)
        L.check(L.lib().coltt_group_search_begin(self.h, qp, ptrs, C.c_size_t(nq), C.c_uint32(k), select, mode, C.c_uint32(ef),
                                                 L.vp(out[0]), L.vp(out[1]), L.vp(out[2]), C.byref(t)))
        return t.value, out

    def SearchEnd(self, ticket):
        L.check(L.lib().coltt_group_search_end(self.h, C.c_uint64(int(ticket))))

    def Timing(self):
        """cumulative ms of the finished shard-search batches: stage A (members' searches), exchange (pack + all-gather + D2H), host merge"""
        a = (C.c_double * 4)()
        L.check(L.lib().coltt_group_timing(self.h, a))
        return {"batches": int(a[0]), "search_ms": a[1], "exchange_ms": a[2], "merge_ms": a[3]}
