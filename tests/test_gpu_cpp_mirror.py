"""The C++ host-side mirror of the reference's Go interfaces (include/coltt_gpu.hpp) as a compiled consumer of the C-ABI:
tests/cpp/mirror_test.cpp restates the reference's own hnsw_commit_test.go / edge store checks in C++ and runs on the GPU."""
import os
import shutil
import subprocess

import pytest

import coltt_amd

pytestmark = pytest.mark.gpu


def test_cpp_mirror_program(gpu, tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "mirror_test"
    libdir = os.path.dirname(coltt_amd.lib_path())
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "mirror_test.cpp"), "-o", str(exe),
                           "-L", libdir, "-lcoltt_gpu", f"-Wl,-rpath,{libdir}"])
    import torch
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert "mirror ok" in out.stdout
