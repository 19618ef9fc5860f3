"""Drop-in for HashGAN's retrieval metric on MI355X.

Mirrors /root/reference/lib/metric.py:4-24

    MAPs(R).get_maps_by_feature(database, query)        # main.py:26,164 -- database first

plus the spellings BASELINE.json's north star names (query first):

    MAP(query_codes, db_codes, query_labels, db_labels, R)
    calc_map(query_codes, db_codes, query_labels, db_labels, R)

Binary codes (+-1 features or {0,1} bits) are ranked by Hamming distance with ties broken by
ascending database index -- for +-1 codes exactly the order of np.argsort(-np.dot(q, db.T))
once ties are broken by index.  Real-valued features (HashGAN's tanh outputs, what main.py
feeds when nothing is binarised) are ranked by float32 inner product like metric.py:13-14
(MAPs only; `binarize=True` applies sign() first instead).  Labels are {0,1} matrices.  All
ranking work runs in the HIP kernels behind hashgan_amd._native; the host only checks
arguments and takes the final mean (metric.py:24).
"""
import warnings

import numpy as np

from . import _native


# ------------------------------------------------------------------ packing
def pack_codes(x):
    """[n, b] array -> uint64 [n, ceil(b/64)]; bit j = (x[:, j] > 0), little endian."""
    x = np.asarray(x)
    if x.ndim != 2 or x.shape[1] < 1:
        raise ValueError("codes must be a 2-D [n, b] array")
    n, b = x.shape
    W = (b + 63) // 64
    bits = np.zeros((n, W * 64), dtype=np.uint8)
    bits[:, :b] = x > 0
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(n, W)


def pack_labels(lab):
    """{0,1} label matrix [n, C] -> uint64 [n, ceil(C/64)]."""
    lab = np.asarray(lab)
    if lab.ndim != 2 or lab.shape[1] < 1:
        raise ValueError("labels must be a 2-D [n, C] array")
    if not np.isin(lab, (0, 1)).all():
        raise ValueError("labels must be {0,1} indicator matrices")
    n, C_ = lab.shape
    LW = (C_ + 63) // 64
    bits = np.zeros((n, LW * 64), dtype=np.uint8)
    bits[:, :C_] = lab != 0
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(n, LW)


def is_binary(x):
    """True when every entry is in {-1, +1} or every entry is in {0, 1}."""
    x = np.asarray(x)
    return bool(np.isin(x, (-1, 1)).all() or np.isin(x, (0, 1)).all())


# ------------------------------------------------------------------ engine
class RetrievalEngine:
    """A database shard resident on one GPU, evaluated against query batches."""

    def __init__(self, device=0):
        self.ctx = _native.Context(device)
        self.b = self.C = None

    def close(self):
        self.ctx.close()

    def set_database(self, codes, labels, idx_base=0, n_total=None, packed=False):
        if packed:
            cw, lw, b, C_ = codes
            lab = labels
        else:
            codes = np.asarray(codes)
            b = codes.shape[1]
            C_ = np.asarray(labels).shape[1]
            cw, lab = pack_codes(codes), pack_labels(labels)
        self.ctx.set_database(cw, lab, b, C_, idx_base, n_total)
        self.b, self.C = b, C_

    def set_database_packed(self, code_words, label_words, b, C_, idx_base=0, n_total=None):
        self.ctx.set_database(code_words, label_words, b, C_, idx_base, n_total)
        self.b, self.C = b, C_

    def set_queries(self, codes, labels):
        codes = np.asarray(codes)
        labels = np.asarray(labels)
        if codes.shape[1] != self.b or labels.shape[1] != self.C:
            raise ValueError("query codes/labels do not match the database (b=%s, C=%s)" % (self.b, self.C))
        self.ctx.set_queries(pack_codes(codes), pack_labels(labels))

    def set_queries_packed(self, code_words, label_words):
        self.ctx.set_queries(code_words, label_words)

    def average_precisions(self, R):
        """-> (ap float64 [Q] with nan where the query has no hit in its top R, rel int64 [Q])."""
        return self.ctx.map(R)

    def topr(self, R):
        self.ctx.topr(R)
        return self.ctx.get_topr()


def mean_over_hits(ap, rel):
    """metric.py:22-24: queries without a hit are skipped; mean of the rest
    (nan + RuntimeWarning if none is left, like np.mean of an empty array)."""
    if rel.size and rel.min() > 0:         # every query has a hit: the selection would be a copy of ap in the same order
        return np.mean(ap)
    return np.mean(np.array(ap[rel != 0]))


_engines = {}


def _engine(device):
    if device not in _engines:
        _engines[device] = RetrievalEngine(device)
    return _engines[device]


def _evaluate(q_codes, db_codes, q_labels, db_labels, R, device, binarize=False, real="rank"):
    db_codes, q_codes = np.asarray(db_codes), np.asarray(q_codes)
    db_labels, q_labels = np.asarray(db_labels), np.asarray(q_labels)
    if db_codes.ndim != 2 or q_codes.ndim != 2 or db_codes.shape[1] != q_codes.shape[1]:
        raise ValueError("query and database codes must be [n, b] with the same b")
    if db_labels.ndim != 2 or q_labels.ndim != 2 or db_labels.shape[1] != q_labels.shape[1]:
        raise ValueError("query and database labels must be [n, C] with the same C")
    if db_labels.shape[0] != db_codes.shape[0] or q_labels.shape[0] != q_codes.shape[0]:
        raise ValueError("codes and labels must have the same number of rows")
    N = db_codes.shape[0]
    if not 1 <= R <= N:
        # metric.py:21 fails the same way: a length-N px cannot be divided by arange(1, R+1)
        raise ValueError("R=%d must be in 1..N (N=%d database rows)" % (R, N))
    eng = _engine(device)
    # float32 features and int64 labels go to the GPU as they are; sign-binarise, bit packing and
    # the "is this really a binary code / indicator label" check all run there (k_pack_*)
    bad_c, bad_l = eng.ctx.set_database_f32(db_codes, db_labels)
    eng.b, eng.C = db_codes.shape[1], db_labels.shape[1]
    qbad_c, qbad_l = eng.ctx.set_queries_f32(q_codes, q_labels)
    if bad_l or qbad_l:
        raise ValueError("labels must be {0,1} indicator matrices")
    if (bad_c or qbad_c) and not binarize:
        # real-valued features (HashGAN's tanh outputs): rank by inner product like metric.py:13-14
        if real == "error":
            raise ValueError("features are not binary codes ({-1,+1} or {0,1}); binarise them first "
                             "(np.sign), use binarize=True, or allow the inner-product ranking")
        if db_codes.shape[1] > 128:
            raise ValueError("inner-product ranking supports up to 128 features (have %d)" % db_codes.shape[1])
        ap, rel = eng.ctx.map_real(R)
    else:
        ap, rel = eng.average_precisions(R)
    return mean_over_hits(ap, rel), ap, rel


# ------------------------------------------------------------------ reference surface
class MAPs:
    """Same constructor and method as lib/metric.py:4-24."""

    def __init__(self, r, device=0, binarize=False):
        self.R = r
        self.device = device
        self.binarize = binarize

    @staticmethod
    def distance(a, b):
        """lib/metric.py:8-10 (unused by the reference; kept for surface parity)."""
        return np.dot(a, b)

    def get_maps_by_feature(self, database, query):
        """database/query: objects with .output [n, b] and .label [n, C] (main.py:157)."""
        m, _, _ = _evaluate(query.output, database.output, query.label, database.label, int(self.R), self.device,
                            binarize=self.binarize)
        return m


def MAP(query_codes, db_codes, query_labels, db_labels, R, device=0):
    """mAP@R of binary codes, query-first argument order (BASELINE.json north star).  Codes must
    be binary ({0,1} or +-1); real-valued features belong to MAPs.get_maps_by_feature."""
    return _evaluate(query_codes, db_codes, query_labels, db_labels, int(R), device, real="error")[0]


calc_map = MAP


def MAP_per_query(query_codes, db_codes, query_labels, db_labels, R, device=0):
    """(mAP, ap [Q] with nan for skipped queries, rel [Q])."""
    return _evaluate(query_codes, db_codes, query_labels, db_labels, int(R), device)
