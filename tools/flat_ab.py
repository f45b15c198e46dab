#!/usr/bin/env python3
"""A/B timing of the FLAT matrix-core path only (no exact-mode pass): `COLTT_LIB=<variant.so> python tools/flat_ab.py [n,dim,quant,batch ...]`
prints ms per batch (hipEvent pair around the whole search on its stream).  Variants are built with
`COLTT_OBJ=/tmp/obj_X COLTT_OUT=coltt_amd/variants/lib_X.so COLTT_EXTRA_FLAGS=-D... python -m coltt_amd.build`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    cases = [(1_000_000, 768, 0, 64), (10_000_000, 768, 1, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    dev = torch.device("cuda", 0)
    for case in cases:
        n, dim, quant, batch = case[:4]
        metric = case[4] if len(case) > 4 else G.COSINE   # 5th field: 0 cosine, 1 euclidean
        fl = G.FlatSpace(dim, metric, quant); fl.Reserve(n)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        done = 0
        while done < n:
            c = min(1 << 20, n - done)
            x = torch.randn((c, dim), device=dev, dtype=torch.float32, generator=gen)
            if os.environ.get("FLAT_AB_CONST"):   # power experiment: constant data (same bytes moved, matrix cores toggle little)
                x = torch.full((c, dim), 0.01, device=dev, dtype=torch.float32)
            torch.cuda.synchronize()
            fl.ChangedVertexDevice(x.data_ptr(), c, first_id=done); done += c; del x
        q = torch.randn((batch, dim), device=dev, dtype=torch.float32, generator=gen); torch.cuda.synchronize()
        k = 10
        oi = torch.empty((batch, k), device=dev, dtype=torch.int64); osc = torch.empty((batch, k), device=dev, dtype=torch.float32)
        oc = torch.empty((batch,), device=dev, dtype=torch.int32)
        ms = []
        for r in range(6):
            fl.VertexSearchDevice(q.data_ptr(), batch, k, oi.data_ptr(), osc.data_ptr(), oc.data_ptr(), select=G.SELECT_NEAREST, mode=G.MODE_MFMA)
            if r: ms.append(fl.last_kernel_ms())
        s = {0: 4, 1: 2, 2: 1, 3: 2}[quant]
        t = float(np.median(ms))
        print(f"{os.path.basename(os.environ.get('COLTT_LIB', 'default'))} gen={os.environ.get('COLTT_MFMA_GEN', 'default(3)')} {n}x{dim} q{quant} b{batch} {'cos' if metric == 0 else 'l2'} {fl.Stats()}: {t:.3f} ms  {n * dim * s / t / 1e9:.3f} TB/s "
              f"{2.0 * n * dim * batch / t / 1e9:.0f} TFLOP/s  (min {min(ms):.3f})", flush=True)
        fl.close()


if __name__ == "__main__":
    main()
