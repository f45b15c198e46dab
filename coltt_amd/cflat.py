"""MultiVectorSpace — GPU-backed stand-in for experimental.multiVectorVertex (experimental/multi_vector_vertex.go)."""
import ctypes as C

import numpy as np

from . import _lib as L


class MultiVectorSpace:
    def __init__(self, dim, n_fields, distance=L.COSINE):
        self.dim, self.nf = int(dim), int(n_fields)
        h = C.c_uint64(0)
        L.check(L.lib().coltt_cflat_create(C.c_uint32(dim), distance, C.c_uint32(n_fields), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) is not None:
            L.lib().coltt_cflat_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ChangedVertex (multi_vector_vertex.go:60-75)
    def ChangedVertex(self, ids, multi_vectors):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        v = np.ascontiguousarray(multi_vectors, np.float32).reshape(len(ids), self.nf, -1)
        if v.shape[2] != self.dim:
            raise ValueError(f"index expect dimension: [{self.dim}], but got [{v.shape[2]}]")
        L.check(L.lib().coltt_cflat_upsert(self.h, L.vp(ids), L.vp(v), C.c_size_t(len(ids))))

    def RemoveVertex(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        L.check(L.lib().coltt_cflat_remove(self.h, L.vp(ids), C.c_size_t(len(ids))))

    def Len(self):
        n = C.c_uint64(0); L.check(L.lib().coltt_cflat_len(self.h, C.byref(n))); return n.value

    # MultiVertexSearch (multi_vector_vertex.go:85-137)
    def MultiVertexSearch(self, topK, multi_vectors, ratios, include=None):
        q = np.ascontiguousarray(multi_vectors, np.float32).reshape(-1, self.nf, self.dim)
        r = np.ascontiguousarray(ratios, np.uint32); inc = np.ones(self.nf, np.uint8) if include is None else np.ascontiguousarray(include, np.uint8)
        nq = q.shape[0]
        ids = np.zeros((nq, topK), np.uint64); sc = np.zeros((nq, topK), np.float32); cnt = np.zeros(nq, np.uint32)
        L.check(L.lib().coltt_cflat_search(self.h, L.vp(q), L.vp(r), L.vp(inc), C.c_size_t(nq), C.c_uint32(topK), L.vp(ids), L.vp(sc), L.vp(cnt)))
        return ids, sc, cnt
