"""Other retrieval metrics on the ranked lists the path already produces.
Not in the reference (lib/metric.py only has mAP); these are the standard companions in the hashing
literature HashGAN's paper reports: precision/recall at the top k, and precision within Hamming
radius r.  The ranking, label matching and histograms run on the GPU (hashgan_amd._native); the
host only reduces small per-query vectors.
"""
import numpy as np

from . import metric


def _load(eng, q_codes, db_codes, q_labels, db_labels):
    """Binary codes only ({0,1} bits or +-1, the same spelling on both sides), like metric.MAP."""
    metric._load_database(eng, np.asarray(db_codes), np.asarray(db_labels), "codes")
    qbad = eng.ctx.set_queries_f32(np.asarray(q_codes), np.asarray(q_labels))
    if qbad[1]:
        raise ValueError("labels must be {0,1} indicator matrices")
    qk, dk = metric._kind(eng.ctx, 1), eng.db_kind
    if not (qk == dk and qk in ("pm1", "bits")):
        raise ValueError("codes must be binary: all {-1,+1} or all {0,1} (found %s queries, %s database)" % (qk, dk))
    return eng.ctx


def precision_recall_at_k(q_codes, db_codes, q_labels, db_labels, ks, device=0):
    """Mean precision@k and recall@k over the queries, Hamming ranking with the canonical tie order.
    recall uses the number of relevant rows in the WHOLE database; queries without any are skipped
    for recall.  -> (precision [len(ks)], recall [len(ks)])"""
    ks = np.asarray(sorted(int(k) for k in ks), dtype=np.int64)
    N = np.asarray(db_codes).shape[0]
    if ks[0] < 1 or ks[-1] > N:
        raise ValueError("every k must be in 1..N")
    eng = metric._Shared.get(device)
    with eng.lock:
        ctx = _load(eng, q_codes, db_codes, q_labels, db_labels)
        ctx.topr(int(ks[-1]))
        match = ctx.get_match()
    cum = np.cumsum(match.astype(np.int64), axis=1)                      # [Q, kmax]
    hits = cum[:, ks - 1]
    precision = (hits / ks[None, :]).mean(0)
    total_rel = ((np.asarray(q_labels) != 0).astype(np.int64) @ (np.asarray(db_labels) != 0).astype(np.int64).T > 0).sum(1)
    ok = total_rel > 0
    recall = (hits[ok] / total_rel[ok, None]).mean(0) if ok.any() else np.full(len(ks), np.nan)
    return precision, recall


def precision_within_radius(q_codes, db_codes, q_labels, db_labels, radius=2, device=0):
    """Mean precision of Hamming-ball lookups: for every query, the fraction of database rows within
    `radius` that share a label with it; a query whose ball is empty contributes 0 (the usual
    convention).  -> (mean precision, per-query ball sizes)"""
    eng = metric._Shared.get(device)
    with eng.lock:
        ctx = _load(eng, q_codes, db_codes, q_labels, db_labels)
        ctx.hist()
        ball = ctx.get_hist()[:radius + 1].astype(np.int64).sum(0)      # rows within the radius, per query
        if ball.max() == 0:
            return 0.0, ball
        ctx.topr(int(ball.max()))                                        # every ball is a prefix of its ranked list
        idx, dist = ctx.get_topr()
        match = ctx.get_match()
    inside = dist <= radius
    hits = (match.astype(bool) & inside).sum(1)
    prec = np.where(ball > 0, hits / np.maximum(ball, 1), 0.0)
    return float(prec.mean()), ball
