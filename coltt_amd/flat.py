"""FlatSpace — GPU-backed stand-in for edge.{none,f16,f8,bf16}VecSpace (edge/vectorstore.go:30-49).

Method names follow the Go interface; metadata maps stay on the host side of the boundary (the Go shim keeps
id -> Metadata and re-attaches it after the call), so only ids/vectors/scores cross it.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class FlatSpace:
    def __init__(self, dim, distance=L.COSINE, quantization=L.Q_NONE):
        self.dim, self.distance, self.quantization = int(dim), distance, quantization
        h = C.c_uint64(0)
        L.check(L.lib().coltt_flat_create(C.c_uint32(dim), distance, quantization, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "_borrowed", False):
            self.h = None
        if getattr(self, "h", None) is not None:
            L.lib().coltt_flat_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ChangedVertex (edge/none_vectorstore.go:66-103), vector half
    def ChangedVertex(self, ids, vectors):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        v = np.ascontiguousarray(vectors, np.float32).reshape(len(ids), -1)
        if v.shape[1] != self.dim:  # none_vectorstore.go:86-88
            raise ValueError(f"Dim Length UnmatchdError: expect dimension: [{self.dim}], but got [{v.shape[1]}]")
        L.check(L.lib().coltt_flat_upsert(self.h, L.vp(ids), L.vp(v), C.c_size_t(len(ids))))

    def ChangedVertexDevice(self, d_ptr, n, first_id=0, ids=None):
        """vectors already in HBM (int device address of an [n, dim] f32 matrix)."""
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.uint64)
        L.check(L.lib().coltt_flat_upsert_device(self.h, L.vp(ids), C.c_uint64(first_id), C.c_void_p(d_ptr), C.c_size_t(n)))

    def Reserve(self, n):
        L.check(L.lib().coltt_flat_reserve(self.h, C.c_uint64(n)))

    # -- RemoveVertex (edge/none_vectorstore.go:105-127), after the inverted index resolved the filter to ids
    def RemoveVertex(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        L.check(L.lib().coltt_flat_remove(self.h, L.vp(ids), C.c_size_t(len(ids))))

    def LoadSize(self):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_flat_len(self.h, C.byref(n)))
        return n.value

    def Dim(self):
        return self.dim

    def Distance(self):
        return self.distance

    def Quantization(self):
        return self.quantization

    def Stored(self, id_):
        dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(self.quantization, np.uint16)
        o = np.empty(self.dim, dt)
        L.check(L.lib().coltt_flat_get(self.h, C.c_uint64(int(id_)), L.vp(o)))
        return o

    def FetchRows(self, first=0, n=None, out=None, with_ids=False):
        """stored rows [first, first+n) in scan order (bulk coltt_flat_get); `out` may be a preallocated array."""
        n = self.LoadSize() - first if n is None else n
        dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(self.quantization, np.uint16)
        if out is None:
            out = np.empty((n, self.dim), dt)
        ids = np.empty(n, np.uint64) if with_ids else None
        L.check(L.lib().coltt_flat_fetch_rows(self.h, C.c_uint64(first), C.c_uint64(n), L.vp(out), L.vp(ids)))
        return (out, ids) if with_ids else out

    @classmethod
    def from_handle(cls, h, dim, distance, quantization):
        """wrap a handle owned by someone else (a group member): close() will not destroy it"""
        o = cls.__new__(cls)
        o.h, o.dim, o.distance, o.quantization, o._borrowed = h, int(dim), distance, quantization, True
        return o

    # -- VertexSearch (edge/none_vectorstore.go:129-180) for a batch of targets
    def VertexSearch(self, targets, topK, select=L.SELECT_REFERENCE, mode=L.MODE_EXACT):
        q = np.ascontiguousarray(targets, np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, max(topK, 1)), np.uint64); sc = np.zeros((nq, max(topK, 1)), np.float32); cnt = np.zeros(nq, np.uint32)
        L.check(L.lib().coltt_flat_search(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(topK), select, mode, L.vp(ids), L.vp(sc), L.vp(cnt)))
        return ids, sc, cnt

    def VertexSearchDevice(self, d_q, nq, topK, d_ids, d_scores, d_counts, select=L.SELECT_REFERENCE, mode=L.MODE_EXACT):
        L.check(L.lib().coltt_flat_search_device(self.h, C.c_void_p(d_q), C.c_size_t(nq), C.c_uint32(topK), select, mode,
                                                 C.c_void_p(d_ids), C.c_void_p(d_scores), C.c_void_p(d_counts)))

    # -- FilterableVertexSearch (edge/none_vectorstore.go:182-253): candidates = ids from the inverted index
    def FilterableVertexSearch(self, candidates, targets, topK, select=L.SELECT_REFERENCE, mode=L.MODE_EXACT):
        q = np.ascontiguousarray(targets, np.float32).reshape(-1, self.dim)
        cand = np.ascontiguousarray(candidates, np.uint64).reshape(-1)
        nq = q.shape[0]
        ids = np.zeros((nq, max(topK, 1)), np.uint64); sc = np.zeros((nq, max(topK, 1)), np.float32); cnt = np.zeros(nq, np.uint32)
        L.check(L.lib().coltt_flat_search_ids_mode(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(topK), select, mode, L.vp(cand),
                                                   C.c_size_t(len(cand)), L.vp(ids), L.vp(sc), L.vp(cnt)))
        return ids, sc, cnt

    # -- SaveVertex / LoadVertex (edge/none_vectorstore.go:308-516)
    def SaveVertex(self):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_flat_save_vertex(self.h, None, None, None, C.c_uint64(0), None, C.c_uint64(0), C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        L.check(L.lib().coltt_flat_save_vertex(self.h, None, None, None, C.c_uint64(0), L.vp(buf), C.c_uint64(n.value), C.byref(n)))
        return buf.tobytes()

    def LoadVertex(self, data):
        b = np.frombuffer(data, np.uint8)
        n = C.c_uint64(0)
        L.check(L.lib().coltt_flat_load_vertex(self.h, L.vp(b), C.c_uint64(len(b)), C.byref(n), None, None, None, C.c_uint64(0)))
        return n.value

    def Stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        L.check(L.lib().coltt_flat_stats(self.h, C.byref(a), C.byref(b)))
        return {"mfma_groups": a.value, "mfma_fallbacks": b.value}

    def NormBounds(self):
        """(min, max) of the stored ||row||^2 over everything ever stored, and whether the cosine matrix-core path is open"""
        a, b_, o = C.c_float(0), C.c_float(0), C.c_int32(0)
        L.check(L.lib().coltt_flat_norm_bounds(self.h, C.byref(a), C.byref(b_), C.byref(o)))
        return a.value, b_.value, bool(o.value)

    def OneLaunchSearches(self):
        a = C.c_uint64(0)
        L.check(L.lib().coltt_flat_one_launch_searches(self.h, C.byref(a)))
        return a.value

    def last_kernel_ms(self):
        ms = C.c_float(0)
        L.check(L.lib().coltt_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value
