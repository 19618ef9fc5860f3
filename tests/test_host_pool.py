"""The host-side packing pool of hg_host_pack.hpp (CPU only): concurrency, fork, and equality with the one-thread packing."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packing_pool_under_concurrent_callers_and_fork(tmp_path):
    exe = str(tmp_path / "host_pool_check")
    src = os.path.join(ROOT, "tests", "host_pool_check.cpp")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "hashgan_amd", "csrc"), "-o", exe, src],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert " bad 0" in run.stdout and "child exit 0" in run.stdout and "pack equal 1" in run.stdout, run.stdout
