#!/usr/bin/env python3
"""HBM traffic of hnsw_search_kernel from a rocprofv3 PMC pass -> profiles/pmc_traffic.json (what bench.py reports as
roofline.traffic).  Procedure (MI355X_MICROARCH.md, HBM / rocprofv3 section: counters in their OWN pass, no trace domains):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "hnsw_search|flat_scan_kernel" -f csv -d /tmp/pmc -o p -- \\
        python $REPO/bench.py --no-cpu-baseline
    python $REPO/tools/pmc_traffic.py /tmp/pmc/p_counter_collection.csv --bench-json <the JSON line bench.py printed>

FETCH_SIZE is reported in KiB; on gfx950 it counts this library's 16 B/lane row streams at half their size, so the factor is
CALIBRATED in the same pass on flat_scan_kernel, whose byte count is known exactly (rows x stride per launch; bench.py's recall
leg runs it over the same index): factor = known bytes / reported bytes, expected 2.0 +- 1 %.  The search kernel's launches are
picked by grid size (one 64-thread workgroup per resident wave, 10 000-query launches use the full persistent grid)."""
import argparse
import collections
import csv
import re
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_counter(path, counter="FETCH_SIZE"):
    """-> {kernel short name: [(grid_size, value), ...]} summed over the counter's dimensions per dispatch"""
    per_dispatch = collections.OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = (r["Dispatch_Id"], r["Kernel_Name"], int(r["Grid_Size"]))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
    out = collections.defaultdict(list)
    for (_, name, grid), v in per_dispatch.items():
        if "hnsw_search2_kernel" in name:   # round 3: hnsw_walk2.hpp — <METRIC, QUANT, PROFILE, OPT[, VISMODE]>; VISMODE 1 = the LDS hash (ef <= 128)
            m = re.search(r"hnsw_search2_kernel<([^>]*)>", name)
            t = [x.strip() for x in m.group(1).split(",")] if m else ["0", "0"]
            q = t[1]
            lds = len(t) >= 5 and t[4].endswith("1")
            short = ("hnsw_search_kernel/lds" if lds else "hnsw_search_kernel/hbm") + ("" if q == "0" else f"/q{q}")
        elif "hnsw_search_kernel" in name:   # the two visited-set variants are different kernels: <.., false> LDS hash, <.., true> HBM byte map
            q = name.split("hnsw_search_kernel<")[1].split(",")[1].strip() if "hnsw_search_kernel<" in name else "0"   # <METRIC, QUANT, VISG>
            short = ("hnsw_search_kernel/hbm" if "true>" in name else "hnsw_search_kernel/lds") + ("" if q == "0" else f"/q{q}")
        elif "pq_scan1_kernel" in name:  # pq.hip, round 6: ONE scan launch per single-query search (all rows of the store)
            short = "pq_scan1_kernel"
        elif "pq_scan_kernel" in name:   # pq.hip: the ADC scan (every query-group / piece-width instance)
            short = "pq_scan_kernel"
        elif "flat_scan_kernel" in name:   # flat_scan_kernel<METRIC, QUANT, ...>: the calibration launches of each row format apart
            q = name.split("flat_scan_kernel<")[1].split(",")[1].strip() if "flat_scan_kernel<" in name else "0"
            short = "flat_scan_kernel" if q == "0" else f"flat_scan_kernel/q{q}"
        else:
            short = name[:40]
        out[short].append((grid, v))
    return out


def calibrate(flat_launches, rows, stride):
    """flat_scan_kernel launches of the recall leg: every 16-query group scans the store in a few segments (512 rows, then 32x what has
    been seen, ...) that together cover each of the `rows` rows exactly once, so
        factor = groups x rows x stride / (sum of the counter over ALL the launches).
    groups = launches per distinct segment shape (the smallest grid belongs to the first segment only).  (Round 2 divided the last
    segment alone; since the 16 Ki / 512 Ki segments share the capped grid with it that reading is off by the segment count.)"""
    if not flat_launches:
        return None, None
    small = min(g for g, _ in flat_launches)
    groups = sum(1 for g, _ in flat_launches if g == small)
    total = sum(v for _, v in flat_launches)
    known_kib = groups * rows * stride / 1024.0
    return known_kib / total, {"launches": len(flat_launches), "groups": groups, "FETCH_SIZE_KiB_sum": total, "known_KiB": known_kib}


def flat_searches(path, counter="FETCH_SIZE"):
    """FLAT matrix-core searches in a PMC pass over tools/flat_ab.py: every search starts with mfma_prep_queries_kernel and ends
    with flat_select_kernel; returns [(query-tile width BN, summed counter over the dispatches of the search)]."""
    per = collections.OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    out, cur = [], None
    for (_, name), v in sorted(per.items()):
        if "mfma_prep_queries" in name:
            cur = {"bn": None, "sum": 0.0}; out.append(cur)
        if cur is None or not any(t in name for t in ("mfma", "flat_pick", "flat_rescore", "flat_select")):
            continue
        cur["sum"] += v
        for bn in (64, 128, 256):
            if f"kernelILi{bn}E" in name:
                cur["bn"] = bn
    return [(c["bn"], c["sum"]) for c in out if c["bn"]]


def main_flat(a):
    """`--flat n,dim,quant,batch [...]`: one table entry per case of the tools/flat_ab.py run profiled in `csv` (same order)."""
    table = json.load(open(a.out)) if os.path.exists(a.out) else {}
    found = flat_searches(a.csv)
    for case in a.flat:
        n, dim, quant, batch = (int(v) for v in case.split(","))
        bn = 64 if batch <= 64 else (128 if batch <= 128 else 256)
        vals = [v for b, v in found if b == bn]
        if not vals:
            sys.exit(f"no matrix-core search with a {bn}-wide query tile in {a.csv}")
        mean_kib = sum(vals) / len(vals)
        algorithmic = n * dim * (4 if quant == 0 else 2)
        traffic = mean_kib * 1024.0 * 2.0
        key = f"flat n={n} dim={dim} quant={quant} batch={batch}"
        table[key] = {"hbm_bytes_per_batch": traffic, f"FETCH_SIZE_KiB_mean_of_{len(vals)}_searches": mean_kib,
                      "correction": "x2: gfx950 FETCH_SIZE under-counts this library's 16 B/lane streams (factor 1.997 calibrated on flat_scan_kernel in the "
                                    "hnsw pass; the LDS-DMA rows are 16 B/lane loads as well); summed over every dispatch of a search (scan segments, picks, re-score, select)",
                      "algorithmic_bytes_per_batch": algorithmic, "traffic_over_algorithmic": traffic / algorithmic, "source": os.path.basename(a.csv),
                      "searches_used": len(vals)}
        print(key, "->", json.dumps(table[key], indent=1))
    json.dump(table, open(a.out, "w"), indent=1)


def main_pq(a):
    """`--pq n,dim,m`: HBM bytes of the DOMINANT pq_scan_kernel launch of a single-query search (the launch over the last, largest
    segment: rows [262144, n) at n = 10 M) in a pass over `bench.py --legs pq` / tools/pq_probe.py.  Those launches are the largest
    cluster of dispatches whose counter values agree within 3 % among the values above half the maximum of the one-pass launches
    (the 64-query call streams the codes 64 times in one dispatch and is excluded by its size)."""
    n, dim, m = (int(v) for v in a.pq.split(","))
    counters = read_counter(a.csv)
    one_launch = bool(counters.get("pq_scan1_kernel"))     # round 6: a single-query search is ONE scan launch over all n rows (pq_scan1_kernel)
    c = counters.get("pq_scan1_kernel") or counters.get("pq_scan_kernel", [])
    if not c:
        sys.exit("no pq_scan_kernel dispatches with FETCH_SIZE in " + a.csv)
    one_pass_cap = 1.5 * n * m / 1024.0 / 2.0           # KiB a one-pass launch can report at most (the counter halves 16 B/lane streams)
    vals = [v for _, v in c if v <= one_pass_cap]
    vals = [v for v in vals if v >= 0.5 * max(vals)]
    best = []
    for v0 in vals:
        grp = [v for v in vals if abs(v - v0) <= 0.03 * v0]
        if len(grp) > len(best):
            best = grp
    mean_kib = sum(best) / len(best)
    s0 = 4096
    while s0 * 64 < n:
        s0 *= 64
    rows = n if one_launch else n - s0                     # pq.hip: segments 4 Ki, x64, ... — the last one starts at the largest 4096 * 64^i below n
    algorithmic = rows * m
    traffic = mean_kib * 1024.0 * 2.0
    table = json.load(open(a.out)) if os.path.exists(a.out) else {}
    key = f"pq n={n} dim={dim} m={m}"
    table[key] = {"hbm_bytes_per_launch": traffic, f"FETCH_SIZE_KiB_mean_of_{len(best)}_launches": mean_kib,
                  "correction": "x2: gfx950 FETCH_SIZE under-counts 16 B/lane streams (factor 1.997 calibrated on flat_scan_kernel in the hnsw passes; the code pieces are 16 B/lane loads)",
                  "algorithmic_bytes_per_launch": algorithmic, "rows_of_the_launch": rows, "traffic_over_algorithmic": traffic / algorithmic,
                  "source": os.path.basename(a.csv), "dispatches_used": len(best), "kernel": "pq_scan1_kernel" if one_launch else "pq_scan_kernel"}
    print(key, "->", json.dumps(table[key], indent=1))
    json.dump(table, open(a.out, "w"), indent=1)


def fetch_cal(path, known):
    """FETCH_SIZE of tools/micro/fetch_cal.hip's three kernels -> {pattern: factor / bytes per load}.  `known` = the JSON line the tool printed."""
    per = collections.defaultdict(list); per_d = collections.OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            key = (r["Dispatch_Id"], r["Kernel_Name"])
            per_d[key] = per_d.get(key, 0.0) + float(r["Counter_Value"])
    for (_, name), v in per_d.items():
        for k in ("block_kernel", "probe_kernel", "stream_kernel"):
            if k in name:
                per[k].append(v)
    out = {}
    for k, vals in per.items():
        vals = vals[1:] if len(vals) > 1 else vals      # (the first launch touches cold pages)
        kib = sum(vals) / len(vals)
        if k == "block_kernel": out["block_factor"] = known["block_kernel_known_bytes_per_launch"] / (kib * 1024.0)
        if k == "stream_kernel": out["stream_factor"] = known["stream_kernel_known_bytes_per_launch"] / (kib * 1024.0)
        if k == "probe_kernel": out["probe_reported_bytes_per_load"] = kib * 1024.0 / known["probe_kernel_loads_per_launch"]
    return out


def main_hnswpq(a):
    """`--hnswpq bench_full.json --cal-csv fetch_cal.csv --cal-json fetch_cal.out`: HBM bytes of the product-quantised walk's launches (hnsw_pq_search_kernel) in a
    PMC pass over `bench.py --legs op`, with the counter calibrated PER ACCESS PATTERN on known byte counts (tools/micro/fetch_cal.hip):
      neighbourhood blocks + adjacency rows   n_exp x (mMax0 x row + mMax0 x 4) bytes, reported at 1 / block_factor
      visited probes                          the rest of what the kernel reports, at the probe pattern's reported bytes per load — one memory request per
                                              4-byte probe, i.e. a whole sector of HBM traffic for one useful byte (granularity of random probes, not re-reads)."""
    full = json.load(open(a.hnswpq))
    pw = full["operating_point"]["pq_walk"]
    nq = full["config"]["queries_per_step"]; n = full["config"]["n"]; dim = full["config"]["dim"]
    pqv = pw["per_query"]; m = pw["m"]; row = (m + 15) // 16 * 16; width = 32
    cal = fetch_cal(a.cal_csv, json.loads(open(a.cal_json).read().strip().splitlines()[-1]))
    per_d = collections.OrderedDict()
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == "FETCH_SIZE" and "hnsw_pq_search_kernel" in r["Kernel_Name"]:
                key = (r["Dispatch_Id"], int(r["Grid_Size"]))
                per_d[key] = per_d.get(key, 0.0) + float(r["Counter_Value"])
    if not per_d:
        sys.exit("no hnsw_pq_search_kernel dispatches with FETCH_SIZE in " + a.csv)
    vals = [v for (_, g), v in per_d.items() if g >= 64 * 1024]                 # the 10 000-query launches fill the persistent grid (>= 1024 waves); single-query calls do not
    best = []
    for v0 in vals:     # the timed steps: the largest group of launches whose counter values agree within 2 % (the ef sweep uses other ef)
        grp = [v for v in vals if abs(v - v0) <= 0.02 * v0]
        if len(grp) > len(best):
            best = grp
    raw = sum(best) / len(best) * 1024.0                                       # reported bytes per launch
    blocks_true = pqv["n_exp"] * (width * row + width * 4) * nq                # what the walk must read of blocks + adjacency rows
    blocks_raw = blocks_true / cal["block_factor"]
    probes_raw = max(0.0, raw - blocks_raw)
    probes = pqv["n_exp"] * width * nq                                          # one per LISTED neighbour (upper bound: rows are not all full); round 6: n_dist counts marks, not probes
    algorithmic = blocks_true + probes * 1 + pqv["n_dist"] * nq
    # a probe is ONE request of at least the reported size; taken at what is reported (a lower bound of the bytes moved)
    traffic = blocks_true + probes_raw
    table = json.load(open(a.out)) if os.path.exists(a.out) else {}
    key = f"hnswpq n={n} dim={dim} m={m} centroids={pw['centroids']} ef={pw['ef']} queries={nq}"
    table[key] = {"hbm_bytes_per_launch": traffic, f"FETCH_SIZE_bytes_reported_mean_of_{len(best)}_launches": raw,
                  "calibration": dict(cal, source=os.path.basename(a.cal_csv)),
                  "model": {"blocks_and_adjacency_bytes": blocks_true, "reported_for_them": blocks_raw, "reported_for_probes_and_the_rest": probes_raw,
                            "probes": probes, "reported_bytes_per_probe": probes_raw / probes},
                  "algorithmic_bytes_per_launch": algorithmic, "traffic_over_algorithmic": traffic / algorithmic,
                  "traffic_over_algorithmic_blocks_only": 1.0,
                  "note": "the walk kernel alone (the re-rank streams rows: x2 as every 16 B/lane stream); blocks are read once each, contiguous; what exceeds the "
                          "algorithmic bytes is the visited byte map: one memory request per probed byte",
                  "source": os.path.basename(a.csv), "dispatches_used": len(best)}
    print(key, "->", json.dumps(table[key], indent=1))
    json.dump(table, open(a.out, "w"), indent=1)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--hnswpq", help="HNSW-over-PQ mode: bench_full.json of the profiled run (operating_point.pq_walk holds the counters)")
    ap.add_argument("--cal-csv", help="--hnswpq: counter CSV of a pass over tools/micro/fetch_cal")
    ap.add_argument("--cal-json", help="--hnswpq: the JSON line fetch_cal printed in that pass")
    ap.add_argument("--pq", help="PQ mode: n,dim,m of the product-quantised store whose single-query scans were profiled")
    ap.add_argument("--bench-json", help="file holding the JSON line bench.py printed in the same run (hnsw mode)")
    ap.add_argument("--leg", default="headline", choices=["headline", "op"], help="which HNSW leg of the bench line (hnsw mode)")
    ap.add_argument("--flat", nargs="*", help="FLAT mode: the n,dim,quant,batch cases of the tools/flat_ab.py run that was profiled")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    a = ap.parse_args(argv)
    if a.hnswpq:
        return main_hnswpq(a)
    if a.flat:
        return main_flat(a)
    if a.pq:
        return main_pq(a)
    if not a.bench_json:
        ap.error("--bench-json or --flat is required")
    b = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
    cfg = b["config"]
    n, dim, nq, ef = cfg["n"], cfg["dim"], cfg["queries_per_step"], cfg["ef"]
    quant = 0 if b["dtype"] == "f32" else 1
    m = int(cfg["workload"].split("M=")[1].split()[0])
    if a.leg == "op":   # the recall >= 0.98 leg of the same run: f16 codes, its own ef, its own dataset
        op = b["operating_point"]
        b = dict(op, dataset=op["workload"].split("dataset ")[1].split(" ")[0])
        quant, ef = 1, op["ef"]
    stride = ((dim * (4 if quant == 0 else 2) + 15) // 16) * 16
    c = read_counter(a.csv)
    # the timed steps: `nq` queries at efSearch `ef` -> the LDS-visited variant up to ef 128, the HBM-visited one above; among its
    # dispatches the steps are the ones sharing the most frequent grid size (the recall pass and the ef curve use other shapes)
    hs = c.get(("hnsw_search_kernel/hbm" if ef > 128 else "hnsw_search_kernel/lds") + ("" if quant == 0 else f"/q{quant}"), [])
    if not hs:
        sys.exit("no hnsw_search_kernel dispatches with FETCH_SIZE in " + a.csv)
    # (the grid that moved the most bytes: single-query latency probes — grid 64 — can outnumber the timed steps)
    by_grid = collections.defaultdict(float)
    for g, v in hs:
        by_grid[g] += v
    full = max(by_grid, key=by_grid.get)
    vals = [v for g, v in hs if g == full]
    # launches of the same grid may still differ in efSearch (the ef sweeps use the persistent grid too): the timed steps are the
    # largest group of launches whose counter values agree within 3 %
    best = []
    for v0 in vals:
        grp = [v for v in vals if abs(v - v0) <= 0.03 * v0]
        if len(grp) > len(best):
            best = grp
    vals = best
    factor, cal = calibrate(c.get("flat_scan_kernel" if quant == 0 else f"flat_scan_kernel/q{quant}", []), n, stride)
    if factor is None or not (1.9 < factor < 2.1):
        print(f"warning: calibration factor {factor} outside 2.0 +- 5 % — using it anyway; check the flat_scan launches", file=sys.stderr)
    factor = factor or 2.0
    mean_kib = sum(vals) / len(vals)
    traffic = mean_kib * 1024.0 * factor
    algorithmic = b["per_query"]["bytes"] * nq
    key = f"hnsw n={n} dim={dim} quant={quant} ef={ef} m={m} queries={nq} dataset={b.get('dataset', 'normal')}"
    rec = {"hbm_bytes_per_launch": traffic, f"FETCH_SIZE_KiB_mean_of_{len(vals)}_launches": mean_kib,
           "correction": f"x{factor:.4f}: gfx950 FETCH_SIZE under-counts 16 B/lane streams; calibrated in this pass on flat_scan_kernel ({cal})",
           "algorithmic_bytes_per_launch": algorithmic, "traffic_over_algorithmic": traffic / algorithmic, "source": os.path.basename(a.csv), "dispatches_used": len(vals), "grid_size": full}
    table = {}
    if os.path.exists(a.out):
        try:
            table = json.load(open(a.out))
        except Exception:
            table = {}
    table[key] = rec
    json.dump(table, open(a.out, "w"), indent=1)
    print(key, "->", json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
