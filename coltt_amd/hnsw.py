"""Hnsw — GPU-backed stand-in for *vectorindex.Hnsw (core/vectorindex/hnsw.go:43-54)."""
import ctypes as C

import numpy as np

from . import _lib as L


class HnswCfg(C.Structure):
    """hnswConfig (core/vectorindex/hnsw_config.go:135-162); -1 = derive the default."""
    _fields_ = [("m", C.c_int32), ("m_max", C.c_int32), ("m_max0", C.c_int32), ("ef", C.c_int32),
                ("ef_construction", C.c_int32), ("algo", C.c_int32), ("level_multiplier", C.c_float),
                ("extend_candidates", C.c_int32), ("keep_pruned", C.c_int32)]

    @staticmethod
    def default(**kw):
        c = HnswCfg(16, -1, -1, 20, 200, 0, -1.0, 0, 1)
        for k, v in kw.items():
            setattr(c, k, v)
        return c


class HnswStats(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_exp", C.c_uint64), ("n_hops", C.c_uint64), ("n_visit_resets", C.c_uint64)]


class Hnsw:
    def __init__(self, dim, distance=L.COSINE, cfg=None, quantization=L.Q_NONE):
        self.dim, self.distance, self.quantization = int(dim), distance, quantization
        cfg = cfg or HnswCfg.default()
        h = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_create(C.c_uint32(dim), distance, quantization, C.byref(cfg), C.byref(h)))
        self.h = h
        self.cfg = HnswCfg()
        L.check(L.lib().coltt_hnsw_get_cfg(self.h, C.byref(self.cfg)))

    @classmethod
    def from_handle(cls, h, dim, distance, quantization):
        """wrap a handle owned by someone else (a group member): close() will not destroy it"""
        o = cls.__new__(cls)
        o.h, o.dim, o.distance, o.quantization, o._borrowed = h, int(dim), distance, quantization, True
        o.cfg = HnswCfg()
        o.Config()
        return o

    def close(self):
        if getattr(self, "_borrowed", False):
            self.h = None
        if getattr(self, "h", None) is not None:
            L.lib().coltt_hnsw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Len(self):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_len(self.h, C.byref(n)))
        return n.value

    def Dim(self):
        return self.dim

    def Config(self):
        """Hnsw.Config() (hnsw.go:92-94): the live configuration, re-read from the library."""
        L.check(L.lib().coltt_hnsw_get_cfg(self.h, C.byref(self.cfg)))
        return self.cfg

    def Get(self, id_):
        """Hnsw.Get(id) / GetVertex(id) (hnsw.go:169-189): (stored vector, level) of a live vertex."""
        dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(self.quantization, np.uint16)
        o = np.empty(self.dim, dt); lv = C.c_int32(0)
        L.check(L.lib().coltt_hnsw_get(self.h, C.c_uint64(int(id_)), L.vp(o), C.byref(lv)))
        return o, lv.value

    def RandomLevel(self, u):
        """Hnsw.RandomLevel() (hnsw.go:280-282) for a uniform draw u in (0,1) supplied by the caller."""
        lv = C.c_int32(0)
        L.check(L.lib().coltt_hnsw_random_level(self.h, C.c_float(u), C.byref(lv)))
        return lv.value

    # -- Hnsw.Load-shaped bulk import (graph dict in the oracle's export layout; raw vectors in slot order)
    def BulkLoad(self, g, raw_vectors):
        v = np.ascontiguousarray(raw_vectors, np.float32)
        ids = np.ascontiguousarray(g["ids"], np.uint64) if g.get("ids") is not None else None
        lv = np.ascontiguousarray(g["levels"], np.int32)
        dl = np.ascontiguousarray(g["deleted"], np.uint8) if g.get("deleted") is not None else None
        off = np.ascontiguousarray(g["row_offsets"], np.int64)
        nb = np.ascontiguousarray(g["nbr"], np.int32)
        nd = np.ascontiguousarray(g["nbr_dist"], np.float32) if g.get("nbr_dist") is not None else None
        L.check(L.lib().coltt_hnsw_bulk_load(self.h, C.c_uint64(len(lv)), L.vp(ids), L.vp(lv), L.vp(dl), L.vp(v), L.vp(off),
                                             L.vp(nb), L.vp(nd), C.c_int32(int(g["entry"]))))

    # -- Hnsw.Insert (hnsw.go:104-167)
    def Insert(self, id_, vector, level):
        v = np.ascontiguousarray(vector, np.float32).reshape(-1)
        L.check(L.lib().coltt_hnsw_insert(self.h, C.c_uint64(int(id_)), L.vp(v), C.c_int32(int(level))))

    def InsertBatchDevice(self, d_vecs, n, levels, batch, first_id=0, ids=None):
        lv = np.ascontiguousarray(levels, np.int32)
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.uint64)
        L.check(L.lib().coltt_hnsw_insert_batch_device(self.h, L.vp(ids), C.c_uint64(first_id), C.c_void_p(d_vecs), L.vp(lv),
                                                       C.c_size_t(n), C.c_uint32(batch)))

    # -- Hnsw.Remove (hnsw.go:191-241)
    def Remove(self, id_):
        L.check(L.lib().coltt_hnsw_remove(self.h, C.c_uint64(int(id_))))

    # -- Hnsw.Search (hnsw.go:243-278) for a batch of queries
    def Search(self, queries, k, ef=0, with_stats=False):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        st = HnswStats()
        L.check(L.lib().coltt_hnsw_search(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(k), C.c_uint32(ef), L.vp(ids), L.vp(sc),
                                          L.vp(cnt), C.byref(st)))
        if with_stats:
            return ids, sc, cnt, {"n_dist": st.n_dist, "n_exp": st.n_exp, "n_hops": st.n_hops, "n_visit_resets": st.n_visit_resets}
        return ids, sc, cnt

    def SearchDevice(self, d_q, nq, k, d_ids, d_scores, d_counts, ef=0):
        st = HnswStats()
        L.check(L.lib().coltt_hnsw_search_device(self.h, C.c_void_p(d_q), C.c_size_t(nq), C.c_uint32(k), C.c_uint32(ef),
                                                 C.c_void_p(d_ids), C.c_void_p(d_scores), C.c_void_p(d_counts), C.byref(st)))
        return {"n_dist": st.n_dist, "n_exp": st.n_exp, "n_hops": st.n_hops, "n_visit_resets": st.n_visit_resets}

    def Export(self):
        ns, nr, ne, ent = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        L.check(L.lib().coltt_hnsw_export(self.h, C.byref(ns), C.byref(nr), C.byref(ne), None, None, None, None, None, None, C.byref(ent)))
        ids = np.empty(ns.value, np.uint64); lv = np.empty(ns.value, np.int32); dl = np.empty(ns.value, np.uint8)
        off = np.empty(nr.value + 1, np.int64); nb = np.empty(ne.value, np.int32); nd = np.empty(ne.value, np.float32)
        L.check(L.lib().coltt_hnsw_export(self.h, C.byref(ns), C.byref(nr), C.byref(ne), L.vp(ids), L.vp(lv), L.vp(dl), L.vp(off),
                                          L.vp(nb), L.vp(nd), C.byref(ent)))
        return {"ids": ids, "levels": lv, "deleted": dl, "row_offsets": off, "nbr": nb, "nbr_dist": nd, "entry": ent.value}

    # -- Hnsw.Commit / Hnsw.Load (core/vectorindex/hnsw_commit.go:69-278)
    def Commit(self, header=True):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_commit(self.h, int(header), None, None, C.c_uint64(0), None, C.c_uint64(0), C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        L.check(L.lib().coltt_hnsw_commit(self.h, int(header), None, None, C.c_uint64(0), L.vp(buf), C.c_uint64(n.value), C.byref(n)))
        return buf.tobytes()

    def Load(self, data, header=True):
        b = np.frombuffer(data, np.uint8)
        n = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_load(self.h, int(header), L.vp(b), C.c_uint64(len(b)), C.byref(n), None, None, None, C.c_uint64(0)))
        L.check(L.lib().coltt_hnsw_get_cfg(self.h, C.byref(self.cfg)))
        return n.value

    def ExportRaw(self):
        """adjacency in the HBM layout: adj0 [n, mMax0], upper_off [n], adjU [n_upper, mMax] (0xffffffff padded)."""
        ns, nu, ent, el = C.c_uint64(0), C.c_uint64(0), C.c_int32(0), C.c_int32(0)
        L.check(L.lib().coltt_hnsw_export_raw(self.h, C.byref(ns), C.byref(nu), C.byref(ent), C.byref(el), None, None, None))
        adj0 = np.empty((ns.value, self.cfg.m_max0), np.uint32); uo = np.empty(ns.value, np.uint32)
        adjU = np.empty((max(nu.value, 1), self.cfg.m_max), np.uint32)
        L.check(L.lib().coltt_hnsw_export_raw(self.h, C.byref(ns), C.byref(nu), C.byref(ent), C.byref(el), L.vp(adj0), L.vp(uo), L.vp(adjU)))
        return {"adj0": adj0, "upper_off": uo, "adjU": adjU, "entry": ent.value, "entry_level": el.value, "n": ns.value}

    def FetchRows(self, first=0, n=None, out=None):
        ns = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_export_raw(self.h, C.byref(ns), None, None, None, None, None, None))
        n = ns.value - first if n is None else n
        dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(self.quantization, np.uint16)
        if out is None:
            out = np.empty((n, self.dim), dt)
        L.check(L.lib().coltt_hnsw_fetch_rows(self.h, C.c_uint64(first), C.c_uint64(n), L.vp(out)))
        return out

    def Reserve(self, n_slots, n_upper_rows=0):
        L.check(L.lib().coltt_hnsw_reserve(self.h, C.c_uint64(int(n_slots)), C.c_uint64(int(n_upper_rows))))

    def Rows8(self):
        """(search launches served by the eight-lanes-per-row core, whether the line-transposed row copy is complete)"""
        a, f = C.c_uint64(0), C.c_int32(0)
        L.check(L.lib().coltt_hnsw_rows8_searches(self.h, C.byref(a), C.byref(f)))
        return a.value, bool(f.value)

    # -- product-quantised search (coltt_hnsw_pq_*; the reference's call shape: playground/hnswpq_verification.go:69-105)
    def PqAttach(self, pq):
        """snapshot the trained quantiser `pq` (a PQSpace) into the index and encode every stored row"""
        L.check(L.lib().coltt_hnsw_pq_attach(self.h, pq.h))

    def PqInfo(self):
        m, c, mt, n = C.c_uint32(0), C.c_uint32(0), C.c_int32(0), C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_pq_info(self.h, C.byref(m), C.byref(c), C.byref(mt), C.byref(n)))
        return {"m": m.value, "C": c.value, "metric": mt.value, "coded": n.value}

    def PqCodes(self, first=0, n=None):
        info = self.PqInfo()
        n = info["coded"] - first if n is None else n
        out = np.empty((n, info["m"]), np.uint8)
        L.check(L.lib().coltt_hnsw_pq_fetch_codes(self.h, C.c_uint64(first), C.c_uint64(n), L.vp(out)))
        return out

    def PqSearch(self, queries, k, ef=0, rerank=0, with_stats=False):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        st = HnswStats(); nx = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_pq_search(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(k), C.c_uint32(ef), C.c_uint32(rerank), L.vp(ids), L.vp(sc),
                                             L.vp(cnt), C.byref(st), C.byref(nx)))
        if with_stats:
            return ids, sc, cnt, {"n_dist": st.n_dist, "n_exp": st.n_exp, "n_hops": st.n_hops, "n_exact": nx.value}
        return ids, sc, cnt

    def PqSearchDevice(self, d_q, nq, k, d_ids, d_scores, d_counts, ef=0, rerank=0):
        st = HnswStats(); nx = C.c_uint64(0)
        L.check(L.lib().coltt_hnsw_pq_search_device(self.h, C.c_void_p(d_q), C.c_size_t(nq), C.c_uint32(k), C.c_uint32(ef), C.c_uint32(rerank),
                                                    C.c_void_p(d_ids), C.c_void_p(d_scores), C.c_void_p(d_counts), C.byref(st), C.byref(nx)))
        return {"n_dist": st.n_dist, "n_exp": st.n_exp, "n_hops": st.n_hops, "n_exact": nx.value}

    def last_kernel_ms(self):
        ms = C.c_float(0)
        L.check(L.lib().coltt_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value
