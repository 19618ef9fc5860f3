"""CPU-side checks of the boundary: the library builds/loads, exports every
symbol include/hashgan_amd.h declares, fails loudly without a GPU, and the
product package never reaches into oracle/."""
import ast
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hashgan_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hashgan_amd import _native, build
    build.build()
    lib = _native.load()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_native.EXPORTS)
    assert lib.hg_version() >= 100


def test_pack_sign_matches_python_packing():
    from hashgan_amd import _native, metric
    from oracle import hamming_map as O
    rng = np.random.default_rng(0)
    for b in (1, 31, 32, 33, 64, 65, 100, 128):
        x = rng.standard_normal((37, b)).astype(np.float32)
        x[0, 0] = 0.0                                  # sign(0) -> bit 0
        w = _native.pack_sign_f32(x)
        assert np.array_equal(w, metric.pack_codes(x))
        assert np.array_equal(w, O.pack_bits((x > 0).astype(np.uint8)))
        assert np.array_equal(metric.pack_codes((x > 0).astype(np.uint8)), w)   # {0,1} and +-1 spellings agree


def test_label_packing_and_validation():
    from hashgan_amd import metric
    lab = np.zeros((3, 81), np.int64)
    lab[0, 0] = lab[1, 63] = lab[1, 64] = lab[2, 80] = 1
    w = metric.pack_labels(lab)
    assert w.shape == (3, 2)
    assert w[0, 0] == 1 and w[1, 0] == 1 << 63 and w[1, 1] == 1 and w[2, 1] == 1 << 16
    with pytest.raises(ValueError):
        metric.pack_labels(np.array([[0, 2]]))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_fails_loudly_without_gpu():
    from hashgan_amd import _native, MAP
    with pytest.raises(_native.HashganNativeError):
        _native.Context(0)
    q = np.ones((2, 8)); d = np.ones((5, 8)); ql = np.ones((2, 3), int); dl = np.ones((5, 3), int)
    with pytest.raises(_native.HashganNativeError):
        MAP(q, d, ql, dl, 3)                           # no silent CPU fallback


def test_python_argument_errors_need_no_gpu():
    from hashgan_amd import MAP
    q = np.ones((2, 8)); d = np.ones((5, 8)); ql = np.ones((2, 3), int); dl = np.ones((5, 3), int)
    with pytest.raises(ValueError):
        MAP(q, d, ql, dl, 6)                           # R > N, like metric.py:21
    with pytest.raises(ValueError):
        MAP(q, d[:, :4], ql, dl, 3)                    # code lengths differ
    with pytest.raises(ValueError):
        MAP(q, d, ql[:1], dl, 3)                       # rows of codes and labels differ


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hashgan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                assert not any(m.split(".")[0] in ("oracle", "tests") for m in mods), (f, mods)
