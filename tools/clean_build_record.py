#!/usr/bin/env python3
"""Clean-tree build of libcoltt_gpu.so (objects, library and manifest deleted first) -> profiles/<name>.json: wall seconds, flags,
compiler, content hashes of every object's inputs, library size and the number of gfx950 kernels it carries.
`python tools/clean_build_record.py profiles/r03_clean_build.json`"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "clean_build.json")
    from coltt_amd import build as B
    shutil.rmtree(B.OBJ, ignore_errors=True)
    for f in (B.OUT, B.manifest_path()):
        if os.path.exists(f):
            os.remove(f)
    t0 = time.time()
    B.build(force=True)
    secs = time.time() - t0
    m = B.verify()
    kernels = None
    try:   # kernel descriptors (<name>.kd) in the symbol table of the embedded gfx950 code object
        sy = subprocess.run(["strings", "-n", "8", B.OUT], capture_output=True, text=True).stdout
        kernels = sum(1 for ln in sy.splitlines() if ln.endswith(".kd"))
    except OSError:
        pass
    rec = {"what": "clean-tree build of libcoltt_gpu.so (objects, library and manifest deleted first); coltt_amd.build.verify() recomputes these hashes "
                   "from the sources, tests/test_cabi.py::test_library_was_compiled_from_exactly_these_sources asserts it",
           "wall_seconds": round(secs, 1), "library_bytes": os.path.getsize(B.OUT), "gfx950_kernel_descriptors": kernels,
           "sources": B.sources(), **{k: m[k] for k in ("built_at", "flags", "hipcc", "objects", "library_sha256")}}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("wall_seconds", "library_bytes", "gfx950_kernel_descriptors")}))


if __name__ == "__main__":
    main()
