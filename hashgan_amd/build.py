"""Build libhashgan_amd.so (gfx950) in-tree with hipcc.

    python -m hashgan_amd.build          # rebuild what is older than its sources
    python -m hashgan_amd.build --force

The library is several translation units (csrc/*.hip, mapped in csrc/hg_ctx.hpp) compiled in parallel to objects under
_lib/obj/ and linked into _lib/libhashgan_amd.so; an object is rebuilt when any file it included last time (its .d
file) is newer.  hipcc cross-compiles without a GPU; objects and the .so are git-ignored but travel to the GPU box with
the working tree.
"""
import concurrent.futures
import os
import shlex
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libhashgan_amd.so")
PROBE_LIB_PATH = os.path.join(LIB_DIR, "libhashgan_amd_probe.so")   # measurement probes compiled in (-DHG_PROBES=1)
UNITS = ["hg_core", "hg_seq", "hg_pairs_valu", "hg_pairs_mx", "hg_pairs_mx1", "hg_real", "hg_comm"]
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")] + [
    os.path.join(ROOT, "include", "hashgan_amd.h")]
# -ffp-contract=off: k_ap reproduces NumPy's float64 rounding; no fused multiply-adds.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]   # only the C ABI of include/hashgan_amd.h is exported
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl", "-lpthread"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _obj_dir(probes, extra):
    tag = "probe" if probes else "prod"
    if extra:
        tag += "_" + "".join(ch if ch.isalnum() else "_" for ch in " ".join(extra))[:80]
    return os.path.join(LIB_DIR, "obj", tag)


def _deps_of(dfile):
    """Paths a make-style .d file lists (None if unreadable)."""
    try:
        text = open(dfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    lhs, colon, rhs = text.partition(":")
    if not colon or not lhs.strip().endswith(".o"):       # not (yet) a rule for an object: treat as unreadable
        return None
    try:
        deps = shlex.split(rhs)
    except ValueError:
        return None
    return deps or None


def _obj_stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = _deps_of(obj[:-2] + ".d")
    if deps is None:
        deps = DEPS
    try:
        return os.path.getmtime(src) > t or any(os.path.getmtime(d) > t for d in deps)
    except OSError:
        return True


def stale(path=LIB_PATH, probes=False, extra=()):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    if any(os.path.getmtime(d) > t for d in DEPS):
        return True
    od = _obj_dir(probes, list(extra))
    return any(_obj_stale(os.path.join(od, u + ".o"), s) for u, s in zip(UNITS, SOURCES))


def build(force=False, verbose=False, probes=False, extra_flags=(), out=None):
    """probes=True builds the measurement variant next to the production library (load it with HG_LIBRARY=...).
    extra_flags: more -D switches (A/B builds of tools/ab_build.sh); such a build names its own `out`."""
    extra = list(extra_flags)
    out = out or (PROBE_LIB_PATH if probes else LIB_PATH)
    if not force and not stale(out, probes, extra):
        return out
    od = _obj_dir(probes, extra)
    os.makedirs(od, exist_ok=True)
    cc = hipcc()
    flags = CFLAGS + (["-DHG_PROBES=1"] if probes else []) + extra
    jobs = []
    for u, s in zip(UNITS, SOURCES):
        obj = os.path.join(od, u + ".o")
        if force or _obj_stale(obj, s):
            tmp = obj + ".tmp%d" % os.getpid()             # concurrent builders (pytest-xdist, N ranks) never see half a file
            # (the dependency file too: written next to the object under a per-process name, moved into place with it)
            jobs.append((u, obj, tmp, [cc] + flags + ["-c", s, "-MD", "-MF", tmp + ".d", "-MT", obj, "-o", tmp]))

    def run(job):
        u, obj, tmp, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            for f in (tmp, tmp + ".d"):
                if os.path.exists(f):
                    os.unlink(f)
            return u, r
        if os.path.exists(tmp + ".d"):
            os.replace(tmp + ".d", obj[:-2] + ".d")
        os.replace(tmp, obj)
        return u, r

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        results = list(ex.map(run, jobs))
    failed = [(u, r) for u, r in results if r.returncode != 0]
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join("[%s]\n%s\n%s" % (u, r.stdout, r.stderr) for u, r in failed))
    if verbose:
        for u, r in results:
            if r.stderr.strip():
                print("[%s]\n%s" % (u, r.stderr))
    tmp = out + ".tmp%d" % os.getpid()
    cmd = [cc] + LDFLAGS + ["-o", tmp] + [os.path.join(od, u + ".o") for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))
