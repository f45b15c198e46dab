#!/usr/bin/env python3
"""Is the per-process bimodality of the f32 HBM-visited walk (profiles/r06av_stream_of_bursts.md) about WHERE the lazily allocated byte map lands?
`python tools/bimodal_probe.py <lazy|hole> [n] [ef] [quant] [dataset]`: `hole` reserves a contiguous block before the index is built and frees it right before the first
large-ef search, so that the byte map's allocation finds it; `lazy` is what every other tool does.  One JSON line: ms per 10 000 queries."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    import bench as B
    mode = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000; ef = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    quant = int(sys.argv[4]) if len(sys.argv) > 4 else 0; dataset = sys.argv[5] if len(sys.argv) > 5 else "normal"
    dim, k, nq = 768, 10, 10_000
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)
    hole = torch.empty((8 << 30,), dtype=torch.uint8, device=dev) if mode == "hole" else None

    class A: m = 16; ef = 128; efc = 200; build_batch = 16384
    ds = B.Dataset(torch, dev, dim, dataset)
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, 0xC0177, quant)
    gen = torch.Generator(device=dev); gen.manual_seed(0x5EED5)
    q = ds.rows(nq, gen)
    out = B.Out(torch, dev, nq, k)
    if hole is not None:
        del hole; torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info()
    res = {"mode": mode, "n": n, "quant": quant, "free_GiB_before_first_large_ef_search": round(free_b / 2**30, 1)}
    for e in (128, ef):
        ms = []
        for r in range(4):
            h.SearchDevice(q.data_ptr(), nq, k, *out.ptrs(), ef=e)
            if r: ms.append(h.last_kernel_ms())
        res[f"ms_ef{e}"] = round(float(np.median(ms)), 3)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
