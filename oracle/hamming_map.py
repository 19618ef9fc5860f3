"""CPU oracle for HashGAN's retrieval-evaluation path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  hashgan_amd/ must never import
anything under oracle/ (tests/test_capi.py::test_product_never_imports_oracle enforces that).

It restates /root/reference/lib/metric.py:4-24 (class MAPs) in NumPy:

  * `reference_as_written`   -- metric.py:12-24 line for line (float32 np.dot,
    default np.argsort(-ips, 1), per-query Python loop).  Its tie order is
    NumPy's unstable default sort, i.e. implementation defined.
  * `map_from_codes` & co.   -- the binary-code specialisation the build
    accelerates: for c in {-1,+1}^b, <q,d> = b - 2*hamming(q,d), so descending
    inner product == ascending Hamming distance; ties are broken by ascending
    database index (the CANONICAL ORDER).  AP/mAP use the
    very expressions of metric.py:20-24 so the float64 rounding is NumPy's own.

Parity pin: tests/golden/*.npz hold outputs of the UNMODIFIED reference run on
tie-free inputs (one extra tie-breaking coordinate, see make_golden.py);
tests/test_oracle_golden.py checks this oracle against them bit for bit, and
tests/test_oracle_vs_reference.py re-runs the comparison live wherever
/root/reference exists.
"""
import numpy as np


# ---------------------------------------------------------------- bit packing
def pack_bits(bits01):
    """{0,1} matrix [n, b] -> uint64 [n, ceil(b/64)]; bit j of a code is bit
    (j % 64) of word j // 64 (little endian), pad bits zero."""
    bits01 = np.ascontiguousarray(bits01, dtype=np.uint8)
    n, b = bits01.shape
    W = (b + 63) // 64
    padded = np.zeros((n, W * 64), dtype=np.uint8)
    padded[:, :b] = bits01
    return np.packbits(padded, axis=1, bitorder="little").view(np.uint64).reshape(n, W)


def hamming_matrix(qwords, dbwords):
    """uint64 [Q, W] x uint64 [N, W] -> int64 [Q, N] Hamming distances.
    (metric.py:13 for +-1 codes: ips = b - 2*d.)"""
    d = np.zeros((qwords.shape[0], dbwords.shape[0]), dtype=np.int64)
    for w in range(qwords.shape[1]):
        d += np.bitwise_count(qwords[:, w][:, None] ^ dbwords[:, w][None, :])
    return d


def canonical_order(dist_row, R):
    """First R database indices by (distance asc, index asc).
    (metric.py:14 + the [0:R] slice at :19, with the tie order fixed.)"""
    return np.argsort(dist_row, kind="stable")[:R]


# ------------------------------------------------------------------ AP / mAP
def label_match(q_label_row, db_label_rows):
    """metric.py:17-19 verbatim semantics (labels must be a signed dtype)."""
    label = q_label_row.copy()
    label[label == 0] = -1
    return np.sum(db_label_rows == label, 1) > 0


def average_precision(imatch, R):
    """metric.py:20-23.  Returns (ap, rel); ap is None when rel == 0 (the
    reference skips such queries)."""
    rel = np.sum(imatch)
    px = np.cumsum(imatch).astype(float) / np.arange(1, R + 1, 1)
    if rel != 0:
        return np.sum(px * imatch) / rel, rel
    return None, rel


def mean_ap(ap_list):
    """metric.py:24."""
    return np.mean(np.array(ap_list))


def topr_from_codes(qbits, dbbits, R, chunk=64):
    """Canonical top-R of every query: (idx int64 [Q, R], dist int64 [Q, R])."""
    qw, dw = pack_bits(qbits), pack_bits(dbbits)
    Q, N = qw.shape[0], dw.shape[0]
    if R > N:
        raise ValueError("R=%d exceeds database size N=%d" % (R, N))
    idx = np.empty((Q, R), dtype=np.int64)
    dist = np.empty((Q, R), dtype=np.int64)
    for s in range(0, Q, chunk):
        d = hamming_matrix(qw[s:s + chunk], dw)
        for i in range(d.shape[0]):
            o = canonical_order(d[i], R)
            idx[s + i] = o
            dist[s + i] = d[i, o]
    return idx, dist


def map_from_codes(qbits, dbbits, qlabels, dblabels, R, chunk=64):
    """mAP of {0,1}-bit codes under the canonical order.

    Returns (mAP float64, ap float64 [Q] with nan where the reference skips the
    query, imatch bool [Q, R], idx int64 [Q, R], dist int64 [Q, R]).
    """
    idx, dist = topr_from_codes(qbits, dbbits, R, chunk)
    Q = idx.shape[0]
    ap = np.full(Q, np.nan)
    imatch = np.empty((Q, R), dtype=bool)
    kept = []
    for i in range(Q):
        imatch[i] = label_match(qlabels[i, :], dblabels[idx[i], :])
        a, _ = average_precision(imatch[i], R)
        if a is not None:
            ap[i] = a
            kept.append(a)
    return mean_ap(kept), ap, imatch, idx, dist


# --------------------------------------------- the reference, as it is written
class _Rows:
    def __init__(self, output, label):
        self.output, self.label = output, label


def reference_as_written(db_output, db_label, q_output, q_label, R):
    """metric.py:12-24 restated line for line; database first, like main.py:164.
    Used as bench.py's cpu_baseline ("port") and for the tie-envelope test."""
    database, query = _Rows(db_output, db_label), _Rows(q_output, q_label)
    ips = np.dot(query.output, database.output.T)
    ids = np.argsort(-ips, 1)
    apx = []
    for i in range(ips.shape[0]):
        label = query.label[i, :].copy()
        label[label == 0] = -1
        imatch = np.sum(database.label[ids[i, :][0:R], :] == label, 1) > 0
        rel = np.sum(imatch)
        px = np.cumsum(imatch).astype(float) / np.arange(1, R + 1, 1)
        if rel != 0:
            apx.append(np.sum(px * imatch) / rel)
    return np.mean(np.array(apx))


def tie_free_features(bits01, is_query, n_db):
    """+-1 features [n, b+1] whose inner products have no ties and sort as
    (Hamming distance asc, database index asc): query rows get +1 in the extra
    column, database row j gets -j*eps with eps = 2^-(ceil(log2 N)+1)
    (SURVEY.md section 8c).  float64 so every partial sum is exact."""
    x = bits01.astype(np.float64) * 2.0 - 1.0
    n = x.shape[0]
    extra = np.empty((n, 1), dtype=np.float64)
    if is_query:
        extra[:] = 1.0
    else:
        eps = 2.0 ** -(int(np.ceil(np.log2(max(n_db, 2)))) + 1)
        extra[:, 0] = -np.arange(n, dtype=np.float64) * eps
    return np.concatenate([x, extra], axis=1)
