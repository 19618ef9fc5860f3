"""Array plumbing around the metric, mirroring the reference's evaluation glue.

    main.py:151-158  forward_all   -> stack_batches
    main.py:161-164  evaluate      -> evaluate
    lib/dataloader.py:39,44-45     -> read_label_list   ("relpath l0 l1 ... l{C-1}" per line)
    lib/config.py:10,22-27 + config/*.yaml -> load_eval_config (HASH_DIM, LABEL_DIM, DB_SIZE, TEST_SIZE, MAP_R)
    main.py:199,239  the `map_val` print -> report

The network forward pass itself (session.run, main.py:155) is out of scope: these helpers take
the per-batch outputs a caller already has.
"""
import types

import numpy as np

from .metric import MAPs

# lib/config.py:10,22-27 defaults (YAML files override DATA.*; HASH_DIM is not overridden by any of them)
_DEFAULTS = {"HASH_DIM": 64, "LABEL_DIM": 10, "DB_SIZE": 54000, "TEST_SIZE": 1000, "MAP_R": 54000}


def read_label_list(path, with_paths=False):
    """Parse a data_list file: one `relative/path l0 l1 ...` line per item (dataloader.py:39,44-45).
    -> int64 [n, C] (and the list of paths)."""
    paths, rows = [], []
    with open(path) as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            paths.append(parts[0])
            rows.append([int(v) for v in parts[1:]])
    lab = np.asarray(rows, dtype=np.int64)
    return (lab, paths) if with_paths else lab


def stack_batches(outputs, labels, size, hash_dim, label_dim):
    """main.py:157-158: np.array(batches).reshape([-1, dim])[:size] for outputs and labels.  The
    generator pads its last batch by wrapping around (dataloader.py:99-104); the [:size] cut drops
    exactly that padding."""
    out = np.array(outputs).reshape([-1, hash_dim])[:size, :]
    lab = np.array(labels).reshape([-1, label_dim])[:size, :]
    return types.SimpleNamespace(output=out, label=lab)


def evaluate(db, test, map_r, device=0, binarize=False):
    """main.py:161-164 with the arrays already in hand: database first, then queries.  Like the reference it
    ranks the features as they are (main.py:164 hands the raw tanh outputs to np.dot); binarize=True is the
    hashing evaluation proper -- sign() first, Hamming ranking."""
    m = MAPs(map_r, device=device, binarize=binarize)
    try:
        return m.get_maps_by_feature(db, test)
    finally:
        m.close()


def load_eval_config(yaml_path=None):
    """The five keys the evaluation needs, with lib/config.py's defaults under a YAML override."""
    cfg = dict(_DEFAULTS)
    if yaml_path:
        import yaml
        with open(yaml_path) as f:
            y = yaml.safe_load(f) or {}
        cfg["HASH_DIM"] = (y.get("MODEL") or {}).get("HASH_DIM", cfg["HASH_DIM"])
        for k in ("LABEL_DIM", "DB_SIZE", "TEST_SIZE", "MAP_R"):
            cfg[k] = (y.get("DATA") or {}).get(k, cfg[k])
    return types.SimpleNamespace(**cfg)


def report(map_val):
    """main.py:199,239."""
    print("map_val: {}".format(map_val))
    return map_val
