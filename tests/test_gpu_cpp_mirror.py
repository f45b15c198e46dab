"""The C++ host-side mirror of the reference's Go interfaces (include/coltt_gpu.hpp) as a compiled consumer of the C-ABI:
tests/cpp/mirror_test.cpp restates the reference's own hnsw_commit_test.go / edge store checks in C++ and runs on the GPU."""
import os
import shutil
import subprocess

import pytest

import coltt_amd

pytestmark = pytest.mark.gpu


def _build_and_run(tmp_path, src, marker, timeout=600):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / src.replace(".cpp", "")
    libdir = os.path.dirname(coltt_amd.lib_path())
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", src), "-o", str(exe),
                           "-L", libdir, "-lcoltt_gpu", f"-Wl,-rpath,{libdir}"])
    import torch
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert marker in out.stdout
    return out.stdout


def test_cpp_mirror_program(gpu, tmp_path):
    _build_and_run(tmp_path, "mirror_test.cpp", "mirror ok")


def test_cpp_batcher_over_real_backends_and_concurrent_single_query_callers(gpu, tmp_path, capsys):
    """coltt::Batcher driving coltt_hnsw_search AND coltt_flat_search on the GPU from 48 threads (answers == unbatched), and
    64 single-query callers overlapping under the shared lock (SURVEY §8f.4, §8b threading).  Prints the measured rates."""
    out = _build_and_run(tmp_path, "batcher_gpu_test.cpp", "batcher gpu ok")
    with capsys.disabled():
        print("\n" + "\n".join(l for l in out.splitlines() if "q/s" in l or "batches" in l))
