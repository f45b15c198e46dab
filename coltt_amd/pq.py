"""PQSpace — product-quantised store on the GPU: ctypes binding of coltt_pq_* (include/coltt_gpu.h).

Parameters follow models.ProductQuantizerParameters (pkg/models/hnsw_common.go:20-33: NumCentroids <= 256, NumSubVectors >= 2),
the arithmetic is pkg/distancepq's (distance.go:30-42 over asm/dot.s, asm/euclidean.s); method names follow the call shape of
playground/hnswpq_verification.go:90-105,154 (the package it drives, pkg/hnswpq, is not in the reference's tree — see pq.hip)."""
import ctypes as C

import numpy as np

from . import _lib as L

PQ_COSINE, PQ_EUCLIDEAN, PQ_DOT = 0, 1, 2


class PQSpace:
    def __init__(self, dim, distance=PQ_EUCLIDEAN, num_subvectors=8, num_centroids=256):
        self.dim, self.distance, self.m, self.c = int(dim), distance, int(num_subvectors), int(num_centroids)
        h = C.c_uint64(0)
        L.check(L.lib().coltt_pq_create(C.c_uint32(dim), distance, C.c_uint32(num_subvectors), C.c_uint32(num_centroids), C.byref(h)))
        self.h = h
        self.dsub = self.dim // self.m

    def close(self):
        if getattr(self, "h", None) is not None:
            L.lib().coltt_pq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- codebooks
    def SetCodebooks(self, codebooks):
        cb = np.ascontiguousarray(codebooks, np.float32)
        if cb.shape != (self.m, self.c, self.dsub):
            raise ValueError(f"codebooks must be [{self.m}][{self.c}][{self.dsub}], got {cb.shape}")
        L.check(L.lib().coltt_pq_set_codebooks(self.h, L.vp(cb)))

    def Codebooks(self):
        cb = np.empty((self.m, self.c, self.dsub), np.float32)
        L.check(L.lib().coltt_pq_get_codebooks(self.h, L.vp(cb)))
        return cb

    def Fit(self, vectors, iterations=8):
        """train the quantiser on a sample (PreTrainProductQuantizer / Fit, playground/hnswpq_verification.go:97-98,154)"""
        v = np.ascontiguousarray(vectors, np.float32).reshape(-1, self.dim)
        L.check(L.lib().coltt_pq_train(self.h, L.vp(v), C.c_size_t(len(v)), C.c_uint32(iterations)))

    def Encode(self, vectors):
        v = np.ascontiguousarray(vectors, np.float32).reshape(-1, self.dim)
        out = np.empty((len(v), self.m), np.uint8)
        L.check(L.lib().coltt_pq_encode(self.h, L.vp(v), C.c_size_t(len(v)), L.vp(out)))
        return out

    def Lut(self, query):
        q = np.ascontiguousarray(query, np.float32).reshape(self.dim)
        out = np.empty((self.m, self.c), np.float32)
        L.check(L.lib().coltt_pq_lut(self.h, L.vp(q), L.vp(out)))
        return out

    # -- rows
    def Insert(self, ids, vectors):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        v = np.ascontiguousarray(vectors, np.float32).reshape(len(ids), -1)
        if v.shape[1] != self.dim:
            raise ValueError(f"Dim Length UnmatchdError: expect dimension: [{self.dim}], but got [{v.shape[1]}]")
        L.check(L.lib().coltt_pq_upsert(self.h, L.vp(ids), L.vp(v), C.c_size_t(len(ids))))

    def InsertDevice(self, d_ptr, n, first_id=0, ids=None):
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.uint64)
        L.check(L.lib().coltt_pq_upsert_device(self.h, L.vp(ids), C.c_uint64(first_id), C.c_void_p(d_ptr), C.c_size_t(n)))

    def InsertCodes(self, ids, codes):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        c = np.ascontiguousarray(codes, np.uint8).reshape(len(ids), self.m)
        L.check(L.lib().coltt_pq_upsert_codes(self.h, L.vp(ids), L.vp(c), C.c_size_t(len(ids))))

    def Remove(self, ids):
        ids = np.ascontiguousarray(ids, np.uint64).reshape(-1)
        L.check(L.lib().coltt_pq_remove(self.h, L.vp(ids), C.c_size_t(len(ids))))

    def Len(self):
        n = C.c_uint64(0)
        L.check(L.lib().coltt_pq_len(self.h, C.byref(n)))
        return n.value

    def FetchCodes(self, first=0, n=None):
        n = self.Len() - first if n is None else n
        codes = np.empty((n, self.m), np.uint8); ids = np.empty(n, np.uint64)
        L.check(L.lib().coltt_pq_fetch_codes(self.h, C.c_uint64(first), C.c_uint64(n), L.vp(codes), L.vp(ids)))
        return codes, ids

    # -- search
    def Search(self, queries, k):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim); nq = len(q)
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        L.check(L.lib().coltt_pq_search(self.h, L.vp(q), C.c_size_t(nq), C.c_uint32(k), L.vp(ids), L.vp(sc), L.vp(cnt)))
        return ids, sc, cnt

    def SearchDevice(self, d_q, nq, k, d_ids, d_sc, d_cnt):
        L.check(L.lib().coltt_pq_search_device(self.h, C.c_void_p(d_q), C.c_size_t(nq), C.c_uint32(k), C.c_void_p(d_ids), C.c_void_p(d_sc),
                                               C.c_void_p(d_cnt)))

    def last_kernel_ms(self):
        """(whole search, the scan launch over the last segment alone) — hipEvent pairs on the search stream"""
        a, b, r = C.c_float(0), C.c_float(0), C.c_uint64(0)
        L.check(L.lib().coltt_pq_last_kernel_ms(self.h, C.byref(a), C.byref(b), C.byref(r)))
        self.last_scan_rows = r.value
        return a.value, b.value
