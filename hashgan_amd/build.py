"""Build libhashgan_amd.so (gfx950) in-tree with hipcc.

    python -m hashgan_amd.build          # rebuild if sources are newer than the .so
    python -m hashgan_amd.build --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the working tree.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libhashgan_amd.so")
PROBE_LIB_PATH = os.path.join(LIB_DIR, "libhashgan_amd_probe.so")   # measurement probes compiled in (-DHG_PROBES=1)
SOURCES = [os.path.join(CSRC, "hg_engine.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")] + [
    os.path.join(ROOT, "include", "hashgan_amd.h")]
# -ffp-contract=off: k_ap reproduces NumPy's float64 rounding; no fused multiply-adds.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-ldl"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def stale(path=LIB_PATH):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, probes=False):
    """probes=True builds the measurement variant next to the production library (load it with HG_LIBRARY=...)."""
    out = PROBE_LIB_PATH if probes else LIB_PATH
    if not force and not stale(out):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = out + ".tmp%d" % os.getpid()                    # concurrent builders (pytest-xdist, N ranks) never see half a file
    cmd = [hipcc()] + FLAGS + (["-DHG_PROBES=1"] if probes else []) + ["-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("hipcc failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))
