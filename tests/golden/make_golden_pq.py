"""tests/golden/pq.npz — fixtures of the product-quantiser scan.

    python tests/golden/make_golden_pq.py

Part A comes from the INDEPENDENT pure-Python restatement (oracle/pyref.py: exact-rational FMA), part B from the C++ oracle at a
size the Python cannot reach.  Inputs are regenerated from seeds (orc_fill_normal is integer-exact); the file holds outputs only.
The reference has no golden vectors for this row (pkg/hnswpq is absent from its tree): the scan is a definition, see
oracle/coltt_oracle.cpp "Product quantiser"."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O, pyref as P  # noqa: E402

A = dict(dim=24, m=6, c=17, n=150, nq=3, k=10, seed=9100)          # m = 6 -> rows padded to 8 codes; 17 centroids
B = dict(dim=768, m=96, c=256, n=3000, nq=4, k=10, seed=9200)      # the bench leg's shape: 96 one-byte codes per row


def inputs(cfg):
    X = O.fill_normal(cfg["seed"], (cfg["n"], cfg["dim"]))
    Q = O.fill_normal(cfg["seed"] + 1, (cfg["nq"], cfg["dim"]))
    T = O.fill_normal(cfg["seed"] + 2, (max(cfg["c"], 300), cfg["dim"]))       # training sample
    ids = (np.arange(cfg["n"], dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 40)
    return X, Q, T, ids


def main():
    out = {}
    X, Q, T, ids = inputs(A)
    cb = O.pq_train(T, A["m"], A["c"], 2)
    out["a_codebooks_bits"] = cb.view(np.uint32)
    for metric in (0, 1, 2):
        for qi in range(A["nq"]):
            codes, lut, ti, ts = P.pq_search(metric, cb, X, ids, Q[qi], A["k"])
            out[f"a_lut_{metric}_{qi}"] = lut.view(np.uint32); out[f"a_ids_{metric}_{qi}"] = ti; out[f"a_scores_{metric}_{qi}"] = ts.view(np.uint32)
    out["a_codes"] = codes
    X, Q, T, ids = inputs(B)
    cb = O.pq_train(T, B["m"], B["c"], 1)
    codes = O.pq_encode(cb, X)
    out["b_codes"] = codes
    for metric in (0, 1, 2):
        i, s, c, _ = O.pq_search(metric, cb, codes, Q, B["k"], ids=ids)
        out[f"b_ids_{metric}"] = i; out[f"b_scores_{metric}"] = s.view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "pq.npz"), **out)
    print("wrote pq.npz:", {k: v.shape for k, v in out.items() if k.startswith(("a_ids_0_0", "b_ids_0", "b_codes"))})


if __name__ == "__main__":
    main()
