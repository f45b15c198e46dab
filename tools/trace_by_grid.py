#!/usr/bin/env python3
"""rocprofv3 kernel trace CSV -> per (kernel, grid size) launch statistics.  `--stats` averages every launch of a kernel name; the
bench line's `roofline.avg_launch_ms` is about ONE launch shape (e.g. the 10 000-query steps of hnsw_search_kernel, grid = resident
waves x 64), while other legs of the same run launch the same kernel on other shapes (300 single-query calls, recall samples, ef
sweeps).  Usage: python tools/trace_by_grid.py <..._kernel_trace.csv> [min_total_ms] > profiles/<name>_by_grid.csv"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    floor_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    acc = collections.OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            for cut in ("(coltt::dev::GraphView", "(unsigned char const*", "(float const*", "(unsigned long long"):
                if cut in name:
                    name = name.split(cut)[0]
            name = name.replace("(anonymous namespace)::", "").replace("void ", "")[:100]
            key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            acc.setdefault(key, []).append(d)
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel", "Grid_Size_X", "Calls", "TotalMs", "AverageMs", "MedianMs", "MinMs", "MaxMs", "LargestClusterCalls", "LargestClusterAverageMs"])
    for (name, grid), ds in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        tot = sum(ds)
        if tot < floor_ms:
            continue
        ds = sorted(ds)
        best = []   # the largest group of launches within 5 % of one another: one launch shape of one leg
        for d0 in ds:
            grp = [d for d in ds if abs(d - d0) <= 0.05 * d0]
            if len(grp) > len(best):
                best = grp
        w.writerow([name, grid, len(ds), f"{tot:.3f}", f"{tot / len(ds):.4f}", f"{ds[len(ds) // 2]:.4f}", f"{ds[0]:.4f}", f"{ds[-1]:.4f}",
                    len(best), f"{sum(best) / len(best):.4f}"])


if __name__ == "__main__":
    main()
