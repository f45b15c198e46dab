#!/usr/bin/env python3
"""Board power / clocks while the FLAT matrix-core search runs back to back (is the kernel power-throttled?).
`python tools/power_probe.py [n,dim,quant,batch]` — loops the search for ~4 s in a thread, polls rocm-smi meanwhile."""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk", "junction", "hotspot"))}
        return keep
    except Exception as e:  # noqa
        return {"error": str(e)}


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    case = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (10_000_000, 768, 1, 256)
    n, dim, quant, batch = case
    dev = torch.device("cuda", 0)
    fl = G.FlatSpace(dim, G.COSINE, quant); fl.Reserve(n)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    done = 0
    while done < n:
        c = min(1 << 20, n - done)
        x = torch.randn((c, dim), device=dev, dtype=torch.float32, generator=gen); torch.cuda.synchronize()
        fl.ChangedVertexDevice(x.data_ptr(), c, first_id=done); done += c; del x
    q = torch.randn((batch, dim), device=dev, dtype=torch.float32, generator=gen); torch.cuda.synchronize()
    k = 10
    oi = torch.empty((batch, k), device=dev, dtype=torch.int64); osc = torch.empty((batch, k), device=dev, dtype=torch.float32)
    oc = torch.empty((batch,), device=dev, dtype=torch.int32)
    print("idle:", smi(), flush=True)
    stop = [False]; ms = []

    def work():
        while not stop[0]:
            fl.VertexSearchDevice(q.data_ptr(), batch, k, oi.data_ptr(), osc.data_ptr(), oc.data_ptr(), select=G.SELECT_NEAREST, mode=G.MODE_MFMA)
            ms.append(fl.last_kernel_ms())
    t = threading.Thread(target=work); t.start()
    t0 = time.time(); samples = []
    while time.time() - t0 < 4.0:
        s_ = smi(); s_["t"] = round(time.time() - t0, 2); s_["searches"] = len(ms); samples.append(s_)
    stop[0] = True; t.join()
    import numpy as np
    import re

    def num(v):
        m = re.search(r"[-+]?\d+(\.\d+)?", str(v)); return float(m.group(0)) if m else float("nan")
    if os.environ.get("POWER_PROBE_RAW"):   # every rocm-smi sample (power, clocks, temperatures) and every search time, as taken
        with open(os.environ["POWER_PROBE_RAW"], "w") as f:
            json.dump({"lib": os.path.basename(os.environ.get("COLTT_LIB", "default")), "case": list(case), "samples": samples, "search_ms": ms}, f)
    late = [x for x in samples if x["t"] > 1.5]
    pw = [num(v) for x in late for k, v in x.items() if "power" in k.lower()]
    sc = [num(v) for x in late for k, v in x.items() if k.startswith("sclk clock speed")]
    print(json.dumps({"lib": os.path.basename(os.environ.get("COLTT_LIB", "default")), "gen": os.environ.get("COLTT_MFMA_GEN", "default"),
                      "case": list(case), "searches": len(ms), "ms_median_steady": float(np.median(ms[len(ms) // 2:])), "ms_first": ms[:3],
                      "power_w_mean_after_1.5s": float(np.mean(pw)) if pw else None, "power_w_max": float(np.max(pw)) if pw else None,
                      "sclk_mhz_mean_after_1.5s": float(np.mean(sc)) if sc else None, "samples": len(samples), "last_sample": samples[-1] if samples else None}))


if __name__ == "__main__":
    main()
