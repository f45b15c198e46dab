"""One-to-one bindings of the leaf kernels (pkg/distance, pkg/compresshelper, pkg/sharding, pkg/distancepq)."""
import ctypes as C

import numpy as np

from . import _lib as L


def distance_pairs(metric, a, b, order=0):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    a = a.reshape(-1, a.shape[-1]); b = b.reshape(-1, b.shape[-1])
    out = np.empty(a.shape[0], np.float32)
    L.check(L.lib().coltt_distance_pairs(metric, order, L.vp(a), L.vp(b), C.c_size_t(a.shape[0]), C.c_uint32(a.shape[1]), L.vp(out)))
    return out


def normalize(v):
    v = np.ascontiguousarray(v, np.float32); m = v.reshape(-1, v.shape[-1])
    out = np.empty_like(m)
    L.check(L.lib().coltt_normalize(L.vp(m), C.c_size_t(m.shape[0]), C.c_uint32(m.shape[1]), L.vp(out)))
    return out.reshape(v.shape)


def quant_lower(quant, x):
    x = np.ascontiguousarray(x, np.float32)
    dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(quant, np.uint16)
    out = np.empty(x.shape, dt)
    L.check(L.lib().coltt_quant_lower(quant, L.vp(x), C.c_size_t(x.size), L.vp(out)))
    return out


def quant_raise(quant, codes):
    dt = {L.Q_NONE: np.float32, L.Q_F8: np.uint8}.get(quant, np.uint16)
    c = np.ascontiguousarray(codes, dt)
    out = np.empty(c.shape, np.float32)
    L.check(L.lib().coltt_quant_raise(quant, L.vp(c), C.c_size_t(c.size), L.vp(out)))
    return out


def shard_vertex(ids, shard_count=16):
    ids = np.ascontiguousarray(ids, np.uint64)
    out = np.empty_like(ids)
    L.check(L.lib().coltt_shard_vertex(L.vp(ids), C.c_size_t(ids.size), C.c_uint64(shard_count), L.vp(out)))
    return out


def pq_float_scan(kind, query, rows):
    q = np.ascontiguousarray(query, np.float32); r = np.ascontiguousarray(rows, np.float32).reshape(-1, q.size)
    out = np.empty(r.shape[0], np.float32)
    L.check(L.lib().coltt_pq_float_scan(kind, L.vp(q), L.vp(r), C.c_size_t(r.shape[0]), C.c_uint32(q.size), L.vp(out)))
    return out


def pq_bit_scan(kind, query, rows):
    q = np.ascontiguousarray(query, np.uint64); r = np.ascontiguousarray(rows, np.uint64).reshape(-1, q.size)
    out = np.empty(r.shape[0], np.float32)
    L.check(L.lib().coltt_pq_bit_scan(kind, L.vp(q), L.vp(r), C.c_size_t(r.shape[0]), C.c_uint32(q.size), L.vp(out)))
    return out
