#!/usr/bin/env python3
"""debug: pair-owned core over line-transposed f32 rows vs the eight-lane core: all distances of a tiny index"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coltt_amd as G
from oracle import oracle as O
assert G.lib().coltt_init(0) == 0
os.environ["COLTT_ROWS8"] = "2"; os.environ["COLTT_MW_MAX_NQ"] = "0"
for quant, d in ((G.Q_NONE, 32), (G.Q_F16, 64), (G.Q_NONE, 64)):
    n = 40
    X = np.zeros((n, d), np.float32)
    for i in range(n):
        X[i, i % d] = 1.0; X[i, (i * 7 + 3) % d] += 0.5
    h = G.Hnsw(d, G.EUCLIDEAN, G.HnswCfg.default(m=8, ef=64, ef_construction=64), quantization=quant)
    for i in range(n): h.Insert(i, X[i], 0)
    q = np.zeros((1, d), np.float32); q[0, 5] = 1.0; q[0, 9] = 2.0
    a = h.Search(q, n, ef=64)
    os.environ["COLTT_EV8"] = "0"; G.lib().coltt_policy_reload()
    b = h.Search(q, n, ef=64)
    del os.environ["COLTT_EV8"]; G.lib().coltt_policy_reload()
    want = np.sqrt(((X - q) ** 2).sum(1))
    oa = np.argsort(a[0][0]); ob = np.argsort(b[0][0])
    print("quant", quant, "d", d, "rows8", h.Rows8())
    print(" ev8  by id:", np.round(a[1][0][oa][:12], 3))
    print(" pair by id:", np.round(b[1][0][ob][:12], 3), "count", b[2])
    print(" numpy     :", np.round(want[:12], 3))
