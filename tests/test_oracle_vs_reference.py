"""Live comparison of the oracle with the imported reference (build container
only -- /root/reference does not exist on the GPU box, where this is skipped)."""
import os
import sys
import types
import warnings
import numpy as np
import pytest
from oracle import hamming_map as O
from hashgan_amd import synth

REF = "/root/reference"
pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lib/metric.py")), reason="no /root/reference")]


def _ref_maps():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from lib.metric import MAPs
    return MAPs


@pytest.mark.parametrize("Q,N,b,R,C,multi", [(30, 5000, 32, 5000, 10, False), (25, 8000, 64, 1000, 10, False),
                                             (20, 6000, 48, 900, 81, True), (10, 3000, 128, 300, 10, False),
                                             (9, 700, 7, 50, 3, False)])
def test_oracle_equals_unmodified_reference(Q, N, b, R, C, multi):
    MAPs = _ref_maps()
    seed = Q * 1000 + b
    if multi:
        dl, ql = synth.multihot_labels(seed, N, C), synth.multihot_labels(seed + 1, Q, C)
    else:
        dl, ql = synth.onehot_labels(seed, N, C)[0], synth.onehot_labels(seed + 1, Q, C)[0]
    db, qb = synth.planted_codes(seed, dl, b, 0.3), synth.planted_codes(seed, ql, b, 0.3)
    database = types.SimpleNamespace(output=O.tie_free_features(db, False, N), label=dl.astype(np.int64))
    query = types.SimpleNamespace(output=O.tie_free_features(qb, True, N), label=ql.astype(np.int64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = MAPs(R).get_maps_by_feature(database, query)
        m, *_ = O.map_from_codes(qb, db, ql, dl, R)
        aw = O.reference_as_written(database.output, database.label, query.output, query.label, R)
    assert m == ref
    assert aw == ref


def test_reference_rejects_r_gt_n():
    MAPs = _ref_maps()
    d = types.SimpleNamespace(output=np.ones((5, 4)), label=np.ones((5, 2), np.int64))
    q = types.SimpleNamespace(output=np.ones((2, 4)), label=np.ones((2, 2), np.int64))
    with pytest.raises(ValueError):
        MAPs(6).get_maps_by_feature(d, q)
