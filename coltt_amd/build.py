"""Build libcoltt_gpu.so (hipcc, gfx950) in-tree.  `python -m coltt_amd.build [--force]`.

Staleness is decided by CONTENT, not mtime: every object is keyed by sha256(source + every header + flags + hipcc version) and
the key is kept in <lib>.manifest.json next to the library, together with the sha256 of the library itself.  `verify()` (used by
tests/test_cabi.py and __graft_entry__.build()) recomputes the keys from the tree: a library that was not compiled from exactly the
sources beside it fails."""
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("COLTT_OUT", os.path.join(HERE, "libcoltt_gpu.so"))
OBJ = os.environ.get("COLTT_OBJ", os.path.join(HERE, "csrc", "_obj"))
# -ffp-contract=off : the exact-order kernels must never fuse a*b+c (the reference's AVX code has no FMA).
# denormals are kept (no -fgpu-flush-denormals-to-zero): the "f8" codec decodes to an f32 denormal.
EXTRA = os.environ.get("COLTT_EXTRA_FLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.sep not in c or os.path.exists(c):
            return c
    return "hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    hs.append(os.path.normpath(os.path.join(HERE, "..", "include", "coltt_gpu.h")))
    return hs


_HIPCC_VERSION = None


def _hipcc_version():
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip()
        except OSError:
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def source_keys():
    """{object name: sha256 over (its source, every header, the flags, the compiler version)} for the tree as it is now."""
    base = hashlib.sha256()
    for h in _headers():
        base.update(os.path.basename(h).encode()); base.update(_sha(h).encode())
    base.update(" ".join(FLAGS).encode()); base.update(_hipcc_version().encode())
    keys = {}
    for s in sources():
        k = base.copy(); k.update(s.encode()); k.update(_sha(os.path.join(CSRC, s)).encode())
        keys[s[:-4] + ".o"] = k.hexdigest()
    return keys


def manifest_path(out=None):
    return (out or OUT) + ".manifest.json"


def read_manifest(out=None):
    try:
        with open(manifest_path(out)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def verify(out=None):
    """Raise unless the library beside the sources was compiled from exactly these sources (content hashes)."""
    out = out or OUT
    m = read_manifest(out)
    if not m:
        raise RuntimeError(f"{manifest_path(out)} missing: run `python -m coltt_amd.build`")
    if not os.path.exists(out) or _sha(out) != m.get("library_sha256"):
        raise RuntimeError(f"{out} is not the library its manifest describes")
    want = source_keys()
    if m.get("objects") != want:
        changed = sorted(k for k in want if m.get("objects", {}).get(k) != want[k])
        raise RuntimeError(f"{out} is stale: sources/headers/flags changed since it was built ({changed})")
    return m


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    want = source_keys()
    have = read_manifest().get("objects", {}) if os.path.exists(OUT) else {}
    jobs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or not os.path.exists(obj) or have.get(s[:-4] + ".o") != want[s[:-4] + ".o"]:
            jobs.append((src, obj))
    t0 = time.time()

    def cc(job):
        cmd = [_hipcc()] + FLAGS + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if force or jobs or not os.path.exists(OUT) or read_manifest().get("library_sha256") != _sha(OUT):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        prev = read_manifest()
        full = force or len(jobs) == len(objs)
        with open(manifest_path(), "w") as f:
            json.dump({"objects": want, "library_sha256": _sha(OUT), "flags": FLAGS, "hipcc": _hipcc_version().splitlines()[:2],
                       "compiled_this_run": sorted(os.path.basename(j[1]) for j in jobs), "seconds": round(time.time() - t0, 1),
                       "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                       "last_full_rebuild_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()) if full else prev.get("last_full_rebuild_at")},
                      f, indent=1)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
