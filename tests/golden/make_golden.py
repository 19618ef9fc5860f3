#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (run in the build
container only: it imports /root/reference/lib/metric.py, which never ships).

For every case in tests/cases.py the inputs are regenerated from the seed, the
+-1 features get one extra tie-breaking coordinate (oracle.tie_free_features,
SURVEY.md section 8c) so that the reference's np.argsort(-ips, 1) has no ties
to break, and `MAPs(R).get_maps_by_feature(database, query)` is called
  * once with all queries      -> `map`   (float64; nan if every query skipped)
  * once per single query row  -> `ap[i]` (np.mean of a 1-element list is the
    AP itself; nan marks a query the reference skips because rel == 0).
For the small cases the first R entries of np.argsort(-ips', 1) are stored too
(`idx`), pinning the canonical order itself.

Also extracts the CIFAR-10 class vectors from the reference's label lists
(data_list/cifar10/{database,test}.txt -- data files, stored as class indices).

Usage:  python tests/golden/make_golden.py [case ...]
"""
import os
import sys
import types
import warnings
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from lib.metric import MAPs            # the reference itself  (noqa: E402)
from oracle import hamming_map as O    # only tie_free_features is used here  (noqa: E402)


def cifar_labels():
    out = {}
    for split in ("database", "test"):
        rows = [l.split()[1:] for l in open(os.path.join(REF, "data_list/cifar10/%s.txt" % split))]
        lab = np.array(rows, dtype=np.int64)
        assert (lab.sum(1) == 1).all()
        out[split + "_cls"] = lab.argmax(1).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "cifar10_labels.npz"), **out)


def run_case(name):
    from tests import cases
    c = cases.build_case(name)
    N, R = c["dbbits"].shape[0], c["R"]
    database = types.SimpleNamespace(output=O.tie_free_features(c["dbbits"], False, N),
                                     label=c["dblab"].astype(np.int64))
    qfeat = O.tie_free_features(c["qbits"], True, N)
    qlab = c["qlab"].astype(np.int64)
    Q = qfeat.shape[0]
    ap = np.empty(Q, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(Q):
            one = types.SimpleNamespace(output=qfeat[i:i + 1], label=qlab[i:i + 1])
            ap[i] = MAPs(R).get_maps_by_feature(database, one)
        if N * Q <= 60_000_000:
            allq = types.SimpleNamespace(output=qfeat, label=qlab)
            m = MAPs(R).get_maps_by_feature(database, allq)
        else:                                   # memory: same value by metric.py:24
            m = np.mean(np.array([a for a in ap if not np.isnan(a)]))
    out = dict(ap=ap, map=np.float64(m))
    if name in cases.SMALL:
        ips = np.dot(qfeat, database.output.T)
        out["idx"] = np.argsort(-ips, 1)[:, :R].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-16s Q=%-5d N=%-9d R=%-6d map=%.17g skipped=%d" % (name, Q, N, R, m, int(np.isnan(ap).sum())))


def run_real_case(name):
    """Real-valued features on the 1/64 grid (exact float arithmetic in any order): the unmodified
    reference on float64 copies + the tie-breaking coordinate (oracle.real_map.tie_free_real_features)."""
    from tests import cases
    from oracle import real_map as RM
    c = cases.build_real_case(name)
    N, R = c["dbf"].shape[0], c["R"]
    database = types.SimpleNamespace(output=RM.tie_free_real_features(c["dbf"], False, N), label=c["dblab"].astype(np.int64))
    qfeat = RM.tie_free_real_features(c["qf"], True, N)
    qlab = c["qlab"].astype(np.int64)
    Q = qfeat.shape[0]
    ap = np.empty(Q, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(Q):
            one = types.SimpleNamespace(output=qfeat[i:i + 1], label=qlab[i:i + 1])
            ap[i] = MAPs(R).get_maps_by_feature(database, one)
        m = MAPs(R).get_maps_by_feature(database, types.SimpleNamespace(output=qfeat, label=qlab))
    out = dict(ap=ap, map=np.float64(m))
    if N * Q <= 2_000_000:
        out["idx"] = np.argsort(-np.dot(qfeat, database.output.T), 1)[:, :R].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-16s Q=%-5d N=%-9d R=%-6d map=%.17g skipped=%d" % (name, Q, N, R, m, int(np.isnan(ap).sum())))


if __name__ == "__main__":
    cifar_labels()
    from tests import cases
    names = sys.argv[1:] or (list(cases.CASES) + list(cases.REAL_CASES))
    for nm in names:
        (run_real_case if nm in cases.REAL_CASES else run_case)(nm)
