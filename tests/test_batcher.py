"""The RPC micro-batcher as compiled C++ (include/coltt_batcher.hpp — same semantics as the Go source
go/colttgpu/batcher.go): tests/cpp/batcher_test.cpp drives it from 48 threads against a mock backend.  No GPU needed."""
import os
import shutil
import subprocess

import pytest


def test_cpp_batcher_program(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "batcher_test"
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "batcher_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert "batcher ok" in out.stdout
