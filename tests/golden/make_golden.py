"""Generates tests/golden/*.npz from the CPU oracle (and pins the oracle's distance kernels against the
reference's own avx.cpp / sse.cpp when oracle/_ref is available).  Inputs are regenerated from seeds by
orc_fill_normal (integer-exact), so the fixtures hold only seeds' outputs.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

DIST_DIMS = [1, 7, 8, 9, 31, 32, 33, 128, 768, 1536]


def fnv64(b, h=14695981039346656037):
    for x in b:
        h ^= x; h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def kernels():
    out = {}
    for d in DIST_DIMS:
        a = O.fill_normal(100 + d, (8, d)); b = O.fill_normal(200 + d, (8, d))
        for mname, f in (("cos", O.cosine), ("l2", O.l2)):
            for order in (0, 1, 2):
                out[f"dist_{mname}_{order}_{d}"] = np.array([f(a[i], b[i], order) for i in range(8)], np.float32).view(np.uint32)
        out[f"norm_{d}"] = O.normalize(a).view(np.uint32)
        out[f"pqdot_{d}"] = np.array([O.pq_dot(a[i], b[i]) for i in range(8)], np.float32).view(np.uint32)
        out[f"pql2_{d}"] = np.array([O.pq_l2sq(a[i], b[i]) for i in range(8)], np.float32).view(np.uint32)
    codes = np.arange(65536, dtype=np.uint16)
    out["f16_decode_hash"] = np.array([fnv64(O.f16_decode(codes).tobytes())], np.uint64)
    x = O.fill_normal(7, 1000) * np.exp2(np.arange(1000, dtype=np.float32) % 40 - 25).astype(np.float32)
    edge = np.array([0.0, -0.0, np.inf, -np.inf, 65504.0, 65520.0, 6e-8, 5.96e-8, 2.98e-8, 6.1035156e-05, 1e-45, 1.0009766,
                     1.0004883, 1.0014648], np.float32)
    x = np.concatenate([x, edge, -edge]).astype(np.float32)
    out["enc_in"] = x.view(np.uint32)
    out["f16_encode"] = O.f16_encode(x)
    out["f8_encode"] = O.f8_encode(x)
    out["f8_lut"] = O.f8_decode(np.arange(256, dtype=np.uint8)).view(np.uint32)
    ids = (np.arange(1000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
    out["shard16"] = np.array([O.shard_vertex(int(i), 16) for i in ids], np.uint8)
    rng = np.random.default_rng(0)
    q = rng.integers(0, 2**63, 12, dtype=np.uint64); rows = rng.integers(0, 2**63, (32, 12), dtype=np.uint64)
    out["bit_q"] = q; out["bit_rows"] = rows
    out["hamming"] = np.array([O.pq_hamming(q, r) for r in rows], np.float32)
    out["jaccard"] = np.array([O.pq_jaccard(q, r) for r in rows], np.float32).view(np.uint32)
    # Go container/heap trace with deliberate ties
    pr = np.array([3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3], np.float32)
    ops = np.array(list(range(16)) + [-1] * 5 + list(range(8)) + [-1] * 4, np.int32)
    out["heap_prios"] = pr; out["heap_ops"] = ops
    for mx in (0, 1):
        pops, fin = O.heap_trace(mx, pr, ops)
        out[f"heap_pops_{mx}"] = pops; out[f"heap_final_{mx}"] = fin
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)


def flat():
    out = {}
    n, d = 2048, 128
    X = O.fill_normal(1, (n, d)); Q = O.fill_normal(99, (16, d))
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(1000003)) % np.uint64(1 << 40)
    for metric in (0, 1):
        for quant in (0, 1, 2, 3):
            f = O.Flat(d, metric, quant); f.upsert(ids, X)
            for k in (1, 10, 100):
                for nearest in (0, 1):
                    I = np.zeros((16, k), np.uint64); S = np.zeros((16, k), np.uint32)
                    for qi in range(16):
                        i, s = f.search(Q[qi], k, bool(nearest), 2)
                        I[qi] = i; S[qi] = s.view(np.uint32)
                    out[f"ids_{metric}_{quant}_{k}_{nearest}"] = I
                    out[f"sc_{metric}_{quant}_{k}_{nearest}"] = S
    np.savez_compressed(os.path.join(HERE, "flat_2048x128.npz"), **out)


def hnsw():
    out = {}
    for tag, n, d, metric in (("1000x128_cos", 1000, 128, 0), ("3000x64_l2", 3000, 64, 1), ("1500x768_cos", 1500, 768, 0)):
        X = O.fill_normal(40 + d, (n, d)); lv = O.levels(41 + d, n)
        ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(11)
        h = O.Hnsw(d, metric); h.insert_many(ids, X, lv)
        out[f"{tag}_graph_hash"] = np.array([h.graph_hash()], np.uint64)
        Q = O.fill_normal(123, (40, d))
        for ef in (20, 128):
            I = np.zeros((40, 10), np.uint64); S = np.zeros((40, 10), np.uint32); st = np.zeros((40, 3), np.uint64)
            for qi in range(40):
                i, s, c = h.search(Q[qi], 10, mode=1, ef=ef, with_stats=True)
                I[qi] = i; S[qi] = s.view(np.uint32); st[qi] = (c["n_dist"], c["n_exp"], c["n_hops"])
            out[f"{tag}_ids_{ef}"] = I; out[f"{tag}_sc_{ef}"] = S; out[f"{tag}_stats_{ef}"] = st
        if n == 1000:  # removals
            rng = np.random.default_rng(40 + d)
            for i in rng.choice(n, 200, replace=False): h.remove(ids[i])
            out[f"{tag}_graph_hash_removed"] = np.array([h.graph_hash()], np.uint64)
    np.savez_compressed(os.path.join(HERE, "hnsw.npz"), **out)


def pin_against_reference():
    r = O.ref()
    if r is None:
        print("oracle/_ref not built: distance kernels NOT re-pinned against the reference sources"); return
    bad = 0
    for d in DIST_DIMS + [5, 100, 384]:
        a = O.fill_normal(100 + d, (8, d)); b = O.fill_normal(200 + d, (8, d))
        for i in range(8):
            for order in (0, 1):
                res = C.c_float(); dot = C.c_float(); ns = C.c_float()
                r.ref_l2sq(order, C.c_size_t(d), a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p), C.byref(res))
                r.ref_cos_dot_norm(order, C.c_size_t(d), a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p), C.byref(dot), C.byref(ns))
                p = O.cosine_parts(a[i], b[i], order)
                bad += np.float32(res.value).view(np.uint32) != O.l2sq(a[i], b[i], order).view(np.uint32)
                bad += np.float32(dot.value).view(np.uint32) != p[0].view(np.uint32)
                bad += np.float32(ns.value).view(np.uint32) != np.float32(p[1] * p[2]).view(np.uint32)
    assert bad == 0, bad
    print("oracle distance kernels == reference avx.cpp/sse.cpp (bit-exact)")


if __name__ == "__main__":
    pin_against_reference()
    kernels(); flat(); hnsw()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"): print(f, os.path.getsize(os.path.join(HERE, f)))
