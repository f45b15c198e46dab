#!/usr/bin/env python3
"""Yardstick for the C3 candidate GEMM: a LIBRARY f16 GEMM of the same shape (M = 10 M rows, N = 256 queries, K = 768) on the same
box under the same power probe as tools/power_probe.py.  Measurement only — nothing here is in the product path.
`python tools/c3_yardstick.py [n,dim,batch]`: one JSON line per variant (rows @ q^T with f16 / f32 output; q @ rows^T)."""
import json
import os
import re
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_probe import smi  # noqa: E402


def num(v):
    m = re.search(r"[-+]?\d+(\.\d+)?", str(v)); return float(m.group(0)) if m else float("nan")


def main():
    import numpy as np
    import torch
    n, dim, batch = (int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (10_000_000, 768, 256)
    dev = torch.device("cuda", 0)
    rows = torch.empty((n, dim), device=dev, dtype=torch.float16)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    for i in range(0, n, 1 << 20):
        c = min(1 << 20, n - i)
        rows[i:i + c] = torch.randn((c, dim), device=dev, dtype=torch.float32, generator=gen).to(torch.float16)
    q = torch.randn((batch, dim), device=dev, dtype=torch.float32, generator=gen).to(torch.float16)
    qt = q.t().contiguous()
    out16 = torch.empty((n, batch), device=dev, dtype=torch.float16)
    out16t = torch.empty((batch, n), device=dev, dtype=torch.float16)
    variants = {"rows@qT->f16": lambda: torch.mm(rows, qt, out=out16),
                "rows@q.t()(view)->f16": lambda: torch.mm(rows, q.t(), out=out16),
                "q@rowsT->f16": lambda: torch.mm(q, rows.t(), out=out16t)}
    try:
        out32 = torch.empty((n, batch), device=dev, dtype=torch.float32)
        torch.mm(rows[:1024], qt, out_dtype=torch.float32)
        variants["rows@qT->f32"] = lambda: torch.mm(rows, qt, out_dtype=torch.float32, out=out32)
    except Exception as e:  # noqa
        print(json.dumps({"variant": "rows@qT->f32", "unsupported": str(e)[:200]}), flush=True)
    flops = 2.0 * n * dim * batch
    for name, fn in variants.items():
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            print(json.dumps({"variant": name, "failed": str(e)[:200]}), flush=True)
            continue
        ms = []
        stop = [False]

        def work():
            while not stop[0]:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); e1.synchronize()
                ms.append(e0.elapsed_time(e1))
        t = threading.Thread(target=work); t.start()
        t0 = time.time(); samples = []
        while time.time() - t0 < 4.0:
            s_ = smi(); s_["t"] = round(time.time() - t0, 2); samples.append(s_)
        stop[0] = True; t.join()
        late = [x for x in samples if x["t"] > 1.5]
        pw = [num(v) for x in late for k, v in x.items() if "power" in k.lower()]
        sc = [num(v) for x in late for k, v in x.items() if k.startswith("sclk clock speed")]
        med = float(np.median(ms[len(ms) // 2:]))
        wr = n * batch * (4 if name.endswith("f32") else 2)
        print(json.dumps({"variant": name, "case": [n, dim, batch], "launches": len(ms), "ms_median_steady": med, "ms_min": float(min(ms)),
                          "tflops": flops / med / 1e9, "frac_of_2500TF": flops / med / 1e9 / 2500.0,
                          "read_GB": n * dim * 2 / 1e9, "write_GB": wr / 1e9, "hbm_frac_read_only": n * dim * 2 / med / 1e6 / 8000.0,
                          "hbm_frac_read_write": (n * dim * 2 + wr) / med / 1e6 / 8000.0,
                          "power_w_mean_after_1.5s": float(np.mean(pw)) if pw else None, "power_w_max": float(np.max(pw)) if pw else None,
                          "sclk_mhz_mean_after_1.5s": float(np.mean(sc)) if sc else None}), flush=True)


if __name__ == "__main__":
    main()
