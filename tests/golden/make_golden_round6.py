#!/usr/bin/env python3
"""Generates tests/golden/round6_definitions.npz FROM oracle/pyref.py (pure Python, NOT the C++ oracle) for the two DEFINITIONS of round 6 — neither is reference
behaviour, so committed vectors are what pins them:
  diverse   COLTT_HNSW_DIVERSE (pyref.DiverseHnsw): inputs + the graph (levels, edge lists, stored edge distances, entrypoint) after inserts and removals
  pqwalk    the walk over product-quantiser codes (pyref.csr_search_pq): table distance = two half-row sums, bounded visiting once the result set is full —
            inputs (graph arrays, stored rows, codebooks, codes, queries) + slots, exact score bits and the four counters, for a set that fills (ef < n)
The C++ oracle (CPU suite) and the HIP path (GPU suite) must both reproduce them bit for bit.  Run: python tests/golden/make_golden_round6.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyref as P  # noqa: E402
from oracle import oracle as O  # noqa: E402  (ONLY for the deterministic input generators fill_normal / levels and the padded-array export helper)


def diverse_case(seed, n, d, metric, m, mmax0, efc, keep, n_remove):
    X = O.fill_normal(seed, (n, d)); lv = O.levels(seed + 1, n, m=m); ids = np.arange(n, dtype=np.uint64) * np.uint64(5) + np.uint64(3)
    h = P.DiverseHnsw(d, metric, keep_pruned=bool(keep), m=m, m_max0=mmax0, ef_construction=efc)
    rng = np.random.default_rng(seed + 2)
    rem = []     # (after which Insert, which vertex): the Removes are interleaved with the Inserts
    for i in range(n):
        assert h.insert(int(ids[i]), X[i], int(lv[i])) is None
        if i > 40 and len(rem) < n_remove and rng.random() < 0.2:
            v = int(rng.integers(0, i))
            if h.remove(int(ids[v])) is None:
                rem.append((i, v))
    g = h.export()
    return dict(X=X, ids=ids, levels=lv, removes=np.array(rem, np.int64).reshape(-1, 2), cfg=np.array([d, metric, m, mmax0, efc, keep], np.int32),
                g_levels=g["levels"], g_deleted=g["deleted"], g_row_offsets=g["row_offsets"], g_nbr=g["nbr"], g_nbr_dist=g["nbr_dist"], g_entry=np.int32(g["entry"]))


def pq_case(seed, n, d, metric, m, c, ef, k, rr, nq):
    from test_pq_hnsw_oracle import _padded
    X = O.fill_normal(seed, (n, d)); lv = O.levels(seed + 1, n)
    seen = np.array([P.normalize(x) for x in X], np.float32) if metric == 0 else X
    h = P.Hnsw(d, metric, m=6, ef=16, ef_construction=30)
    for i in range(n):
        assert h.insert(i, X[i], int(lv[i])) is None
    g = h.export()
    adj0, upper_off, adjU = _padded(g, h.m_max0, h.m_max)
    entry = int(g["entry"]); entry_level = int(g["levels"][entry])
    cb = O.pq_train(seen[:150], m, c, iters=2)            # (k-means itself is pinned elsewhere: tests/golden/pq.npz)
    codes = O.pq_encode(cb, seen)
    Q = O.fill_normal(seed + 3, (nq, d))
    sl = np.full((nq, k), -1, np.int64); sc = np.zeros((nq, k), np.float32); cn = np.zeros(nq, np.int32)
    tot = {"n_dist": 0, "n_exp": 0, "n_hops": 0, "n_exact": 0}
    for qi in range(nq):
        q = P.normalize(Q[qi]) if metric == 0 else Q[qi]
        s_, c_, cnt = P.csr_search_pq(seen, adj0, upper_off, adjU, metric, entry, entry_level, codes, cb, 1, q, k, ef, rr)
        cn[qi] = len(s_); sl[qi, :len(s_)] = s_; sc[qi, :len(s_)] = np.array(c_, np.float32)
        for kk in tot: tot[kk] += cnt[kk]
    return dict(X=X, seen=seen, adj0=adj0, upper_off=upper_off, adjU=adjU, entry=np.int32(entry), entry_level=np.int32(entry_level), cb=cb, codes=codes, Q=Q,
                g_levels=g["levels"], g_deleted=g["deleted"], g_row_offsets=g["row_offsets"], g_nbr=g["nbr"], g_nbr_dist=g["nbr_dist"], g_ids=g["ids"],
                cfg=np.array([d, metric, m, c, ef, k, rr], np.int32), slots=sl, scores=sc, counts=cn,
                counters=np.array([tot["n_dist"], tot["n_exp"], tot["n_hops"], tot["n_exact"]], np.int64))


def main():
    out = {}
    dcases = [diverse_case(9301, 240, 16, 0, 6, 10, 30, 0, 25), diverse_case(9311, 200, 12, 1, 8, -1, 40, 1, 15)]
    for ci, cse in enumerate(dcases):
        for k, v in cse.items():
            out[f"d{ci}_{k}"] = v
    pcases = [pq_case(9401, 300, 32, 1, 8, 16, 40, 5, 12, 6),      # 8 codes: one 16-byte piece (S_hi = 0)
              pq_case(9411, 260, 48, 0, 24, 16, 30, 5, 0, 6)]      # 24 codes: two pieces (16 + 8), cosine index with euclidean tables
    for ci, cse in enumerate(pcases):
        for k, v in cse.items():
            out[f"p{ci}_{k}"] = v
    out["n_diverse"] = np.int32(len(dcases)); out["n_pq"] = np.int32(len(pcases))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "round6_definitions.npz"), **out)
    print("wrote round6_definitions.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("p0_")})


if __name__ == "__main__":
    main()
