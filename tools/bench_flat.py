#!/usr/bin/env python3
"""FLAT scan measurements for BASELINE.json configs[1] (1M x 768 f32, batch 64) and configs[2] (10M x 768 f16-quantised,
batch 256): exact-order scan vs matrix-core candidate generation (COLTT_MODE_MFMA).  Not the bench.py contract — these
are the secondary roofline numbers recorded in DESIGN.md / profiles/.  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(G, torch, n, dim, quant, batch, mode, k=10, reps=3, check=None):
    dev = torch.device("cuda", 0)
    fl = G.FlatSpace(dim, G.COSINE, quant); fl.Reserve(n)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    done = 0
    while done < n:
        c = min(1 << 20, n - done)
        x = torch.randn((c, dim), device=dev, dtype=torch.float32, generator=gen)
        torch.cuda.synchronize()  # the library reads on its own stream: producer must be done
        fl.ChangedVertexDevice(x.data_ptr(), c, first_id=done); done += c; del x
    q = torch.randn((batch, dim), device=dev, dtype=torch.float32, generator=gen)
    torch.cuda.synchronize()
    oi = torch.empty((batch, k), device=dev, dtype=torch.int64); osc = torch.empty((batch, k), device=dev, dtype=torch.float32)
    oc = torch.empty((batch,), device=dev, dtype=torch.int32)
    ms = []
    for r in range(reps + 1):
        fl.VertexSearchDevice(q.data_ptr(), batch, k, oi.data_ptr(), osc.data_ptr(), oc.data_ptr(), select=G.SELECT_NEAREST, mode=mode)
        if r: ms.append(fl.last_kernel_ms())
    t = float(np.mean(ms)) / 1e3
    s = {0: 4, 1: 2, 2: 1, 3: 2}[quant]
    res = {"case": f"FLAT cosine {n}x{dim} {'f32' if quant == 0 else 'f16 codes'} batch {batch} k={k} mode={'mfma' if mode else 'exact'}",
           "ms_per_batch": t * 1e3, "queries_per_s": batch / t, "algorithmic_GBps": n * dim * s / t / 1e9,
           "frac_of_hbm_peak": n * dim * s / t / 8e12, "TFLOPs": 2.0 * n * dim * batch / t / 1e12,
           "frac_of_f16_mfma_peak" if quant else "frac_of_f32_peak": 2.0 * n * dim * batch / t / (2.5e15 if quant else 157.3e12)}
    ids = oi.cpu().numpy(); sc = osc.cpu().numpy()
    fl.close()
    return res, ids, sc


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    cases = [(1_000_000, 768, 0, 64), (10_000_000, 768, 1, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for n, dim, quant, batch in cases:
        e, ei, es = run(G, torch, n, dim, quant, batch, G.MODE_EXACT)
        print(json.dumps(e), flush=True)
        if quant in (0, 1, 3):
            m, mi, msc = run(G, torch, n, dim, quant, batch, G.MODE_MFMA)
            m["identical_to_exact_mode"] = bool(np.array_equal(ei, mi) and np.array_equal(es.view(np.uint32), msc.view(np.uint32)))
            print(json.dumps(m), flush=True)


if __name__ == "__main__":
    main()
