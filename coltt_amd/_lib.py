"""ctypes loader for libcoltt_gpu.so — the C-ABI declared in include/coltt_gpu.h."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
COSINE, EUCLIDEAN = 0, 1
Q_NONE, Q_F16, Q_F8, Q_BF16 = 0, 1, 2, 3
SELECT_REFERENCE, SELECT_NEAREST = 0, 1
MODE_EXACT, MODE_MFMA = 0, 1

_lib = None
_env_seen = None
# the knobs the library snapshots (coltt_amd/csrc/common.hpp: Policy); when the test / tool changes one of them in os.environ the
# binding asks the library to re-read them before the next call
_KNOBS = ("COLTT_FLAT_ONE", "COLTT_STAGING", "COLTT_EV8", "COLTT_ROWS8", "COLTT_F8_MFMA", "COLTT_VISG", "COLTT_WALK2", "COLTT_WALK2_LDS", "COLTT_BLOOM_KB",
          "COLTT_WAVES_PER_CU", "COLTT_ROWS_NT", "COLTT_ROWS_NT_MIN_MB", "COLTT_PQ_WAVES", "COLTT_PQ_NBR", "COLTT_LAT_SEQ", "COLTT_LAT_HELPERS", "COLTT_LAT_MAX_NQ", "COLTT_MW_MAX_NQ", "COLTT_VISG_BUDGET_MB")


class ColttError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"coltt_gpu error {code}: {msg}")
        self.code = code


def lib_path():
    return os.environ.get("COLTT_LIB", os.path.join(HERE, "libcoltt_gpu.so"))


def declared_symbols():
    """Every function name include/coltt_gpu.h declares."""
    hdr = open(os.path.join(HERE, "..", "include", "coltt_gpu.h")).read()
    return sorted(set(re.findall(r"\b(coltt_[a-z0-9_]+)\s*\(", hdr)))


def _sync_policy():
    global _env_seen
    get = os.environ.get
    now = tuple(get(k) for k in _KNOBS)
    if now != _env_seen:
        if _env_seen is not None:
            _lib.coltt_policy_reload()
        _env_seen = now


def lib():
    """Load the HIP extension.  There is no fallback: a missing .so is an error."""
    global _lib
    if _lib is not None:
        _sync_policy()
        return _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HIP runtimes in one process cannot both own
        # the GPU ("No HIP GPUs are available"), so when torch is installed it is imported FIRST: the dynamic linker then
        # resolves this library's NEEDED libamdhip64.so.7 to the copy that is already loaded (SONAME match) and torch
        # tensors, RCCL and these kernels share one runtime.  Without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        p = lib_path()
        if not os.path.exists(p):
            raise ColttError(-5, f"{p} is missing — build it with `python -m coltt_amd.build` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(p)
        L.coltt_last_error.restype = C.c_char_p
        L.coltt_version.restype = C.c_char_p
        _lib = L
        _sync_policy()
    return _lib


def check(rc):
    if rc != 0:
        raise ColttError(rc, lib().coltt_last_error().decode("utf-8", "replace"))


def vp(a):
    """numpy array / int address / None -> c_void_p"""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)
