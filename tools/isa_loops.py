#!/usr/bin/env python3
"""Static profile of gfx950 assembly (hipcc -S --cuda-device-only), per kernel and per LOOP DEPTH:
instruction classes, scalar spills parked in VGPR lanes (v_writelane / v_readlane on the compiler's spill registers), hazard nops, waits —
outside every loop (prologue / epilogue: paid once per launch) and inside loops (paid per query, per expansion, per neighbour chunk).

    python tools/isa_loops.py file.s [substring of a (demangled) kernel name ...]

A loop = the span from a label to the last backward branch to it.  Spill registers =
the VGPRs a `v_writelane_b32 vN, sM, <constant lane>` writes (the walk kernels hold no v_writelane of their own); a spill reload =
`v_readlane_b32 sM, vN, <constant lane>` on one of those."""
import collections
import re
import subprocess
import sys


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
            if len(out) >= len(names):
                return dict(zip(names, out))
        except OSError:
            continue
    return {n: n for n in names}


def kernels(path):
    cur = None; body = []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".") and cur is None and not m.group(1).startswith("BB"):
            cur = m.group(1); body = []; continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                yield cur, body; cur = None; continue
            body.append(line)


def classify(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"


def analyse(body):
    labels = {}; ins = []
    for line in body:
        s = line.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins); continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        ins.append(s.split(";")[0].strip())
    hdr = {}   # one loop per header label: [header, the last backward branch to it]
    for i, s in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            a = labels[m.group(1)]
            hdr[a] = max(hdr.get(a, i), i)
    loops = sorted(hdr.items())
    depth = [0] * len(ins)
    for a, b in loops:
        for i in range(a, b + 1):
            depth[i] += 1
    spill_regs = set()
    for s in ins:
        m = re.match(r"^v_writelane_b32\s+(v\d+),\s*s\d+,\s*(\d+)$", s)
        if m:
            spill_regs.add(m.group(1))
    out = collections.defaultdict(lambda: collections.Counter())
    for i, s in enumerate(ins):
        op = s.split()[0]
        c = classify(op)
        out[depth[i]][c] += 1
        if op.startswith("v_writelane"):
            m = re.match(r"^v_writelane_b32\s+(v\d+),\s*s\d+,\s*(\d+)$", s)
            if m: out[depth[i]]["spill_st"] += 1
        if op.startswith("v_readlane"):
            m = re.match(r"^v_readlane_b32\s+s\d+,\s*(v\d+),\s*(\d+)$", s)
            if m and m.group(1) in spill_regs: out[depth[i]]["spill_ld"] += 1
    big = []   # the large loops (the work loop over queries, the expansion loop, ...) with what they hold
    for a, b in loops:
        if b - a < 120: continue
        c = collections.Counter()
        for i in range(a, b + 1):
            op = ins[i].split()[0]; c[classify(op)] += 1
            m = re.match(r"^v_readlane_b32\s+s\d+,\s*(v\d+),\s*(\d+)$", ins[i])
            if m and m.group(1) in spill_regs: c["spill_ld"] += 1
            if re.match(r"^v_writelane_b32\s+(v\d+),\s*s\d+,\s*(\d+)$", ins[i]): c["spill_st"] += 1
        big.append((a, b, c))
    return out, len(ins), len(loops), big


def main():
    path = sys.argv[1]; want = sys.argv[2:]
    ks = list(kernels(path))
    dm = demangle([k for k, _ in ks])
    cols = ["valu", "salu", "lane", "spill_st", "spill_ld", "nop", "wait", "lds", "vmem", "smem", "branch", "mfma"]
    for name, body in ks:
        n = re.sub(r"\(anonymous namespace\)::", "", dm.get(name, name))
        n = re.sub(r"^void ", "", n).split("(")[0]
        if want and not any(w in n for w in want):
            continue
        prof, total, nloops, big = analyse(body)
        if total < 20:
            continue
        print(f"## `{n}` — {total} instructions, {nloops} loops")
        print("| where | " + " | ".join(cols) + " |")
        print("|---|" + "---|" * len(cols))
        print(f"| outside loops | " + " | ".join(str(prof[0][c]) for c in cols) + " |")
        tot = collections.Counter()
        for d in prof:
            if d >= 1: tot.update(prof[d])
        print(f"| in loops | " + " | ".join(str(tot[c]) for c in cols) + " |")
        for a, b, c in big:
            print(f"| loop [{a}, {b}] | " + " | ".join(str(c[x]) for x in cols) + " |")
        print()


if __name__ == "__main__":
    main()
