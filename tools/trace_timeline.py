#!/usr/bin/env python3
"""rocprofv3 kernel trace CSV -> the LAST n launches in time order: start offset, duration and the gap to the previous kernel (us).
`python tools/trace_timeline.py <..._kernel_trace.csv> [n]` — what a chain of short launches spends where."""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0][:70]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
    rows.sort()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = rows[-n:]
    t0, prev_end = rows[0][0], rows[0][0]
    print(f"{'start_us':>10} {'dur_us':>8} {'gap_us':>8}  kernel @ grid")
    for s, e, name, grid in rows:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {name} @ {grid}")
        prev_end = e


if __name__ == "__main__":
    main()
