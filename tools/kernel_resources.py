#!/usr/bin/env python3
"""Register / spill / occupancy table of every kernel in the library, from hipcc's own remarks:
    COLTT_EXTRA_FLAGS="-Rpass-analysis=kernel-resource-usage" COLTT_OBJ=/tmp/obj_rpass COLTT_OUT=/tmp/libcoltt_rpass.so python -m coltt_amd.build > /tmp/rpass.log 2>&1
    python tools/kernel_resources.py /tmp/rpass.log > profiles/rNN_kernel_resources.md
(a scratch build: the remarks do not change the code, the shipped library is not touched)."""
import re
import subprocess
import sys


def demangle(names):
    try:
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
                if len(out) >= len(names):
                    return dict(zip(names, out))
            except OSError:
                continue
    except Exception:
        pass
    return {n: n for n in names}


def main():
    rows = []; cur = None
    for line in open(sys.argv[1]):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}; rows.append(cur); continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?):\s+(\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    dm = demangle([r["name"] for r in rows])
    seen = set()
    print("| kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD (registers) |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        n = dm.get(r["name"], r["name"])
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = n.split("(")[0]
        if n in seen or "Occupancy [waves/SIMD]" not in r:
            continue
        seen.add(n)
        print(f"| `{n}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | {r.get('ScratchSize [bytes/lane]')} | {r.get('Occupancy [waves/SIMD]')} |")


if __name__ == "__main__":
    main()
