"""Drop-in for HashGAN's retrieval metric on MI355X.

Mirrors /root/reference/lib/metric.py:4-24

    MAPs(R).get_maps_by_feature(database, query)        # main.py:26,164 -- database first

plus the spellings BASELINE.json's north star names (query first):

    MAP(query_codes, db_codes, query_labels, db_labels, R)
    calc_map(query_codes, db_codes, query_labels, db_labels, R)

MAPs ranks what the reference ranks: np.argsort(-np.dot(q, db.T)) (metric.py:13-14), ties broken
by ascending database index.  For +-1 codes that is ascending Hamming distance and runs on the
Hamming kernels; anything else -- HashGAN's real-valued tanh outputs, {0,1} bits (whose np.dot
counts common ones: NOT a Hamming ranking), codes containing zeros -- is ranked by float32 inner
product (`binarize=True` applies sign() first instead).  MAP / calc_map take binary codes only,
{0,1} bits or +-1, and rank by Hamming distance.  Labels are {0,1} matrices.  All ranking work
runs in the HIP kernels behind hashgan_amd._native; the host only checks arguments and takes the
final mean (metric.py:24).
"""
import threading

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


# ------------------------------------------------------------------ engine
class RetrievalEngine:
    """A database shard resident on one GPU, evaluated against query batches."""

    def __init__(self, device=0):
        self.ctx = _native.Context(device)
        self.b = self.C = self.N = self.db_kind = self.db_src = None
        self.resident = None           # _evaluate's shortcut: (codes, labels, mode) of the arrays whose packed copy is on the GPU

    def close(self):
        self.ctx.close()

    def forget(self):
        """Before the engine goes back to the pool: nothing of the last owner's tables may be taken for resident."""
        self.b = self.C = self.N = self.db_kind = self.db_src = None
        self.resident = None

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


# ------------------------------------------------------------------ engines
class _Pool:
    """Process-wide pool of GPU contexts for MAPs objects.  main.py:164 writes `MAPs(cfg.DATA.MAP_R).get_maps_by_feature(...)`:
    a NEW object per evaluation, never closed.  A context of its own per object would mean a stream, ~0.3 GB of device buffers
    and the pinned staging blocks created and torn down around every call (hipMalloc / hipHostMalloc / hipFree: tens of
    milliseconds, and now and then a stall of a second -- bench.py's `drop_in_literal` leg).  So an object BORROWS a context on
    first use and hands it back when it is closed or collected; the context keeps its buffers (they only grow), the next
    object's tables are loaded into them.  An object that is alive keeps its context to itself (no two objects ever share
    device state at the same time); a context whose engine options were changed by its borrower is destroyed instead of
    recycled; at most `max_idle` contexts per device wait in the pool."""
    lock = threading.Lock()
    idle = {}
    max_idle = 2
    created = 0
    recycled = 0
    closed = 0

    @classmethod
    def acquire(cls, device):
        with cls.lock:
            lst = cls.idle.get(device)
            if lst:
                cls.recycled += 1
                return lst.pop()
            cls.created += 1
        eng = RetrievalEngine(device)
        eng.ctx.preload()          # (the process's first evaluation pays for every path's first use: no later one does)
        return eng

    @classmethod
    def release(cls, eng):
        keep = False
        try:
            touched = eng.ctx.options_touched - {"keep_floats"}         # (set by every load)
            pending = getattr(eng.ctx, "_in_flight", None)
            if not touched and not pending and eng.ctx._h and not getattr(eng, "poisoned", False):
                eng.forget()
                with cls.lock:
                    lst = cls.idle.setdefault(eng.ctx.device, [])
                    if len(lst) < cls.max_idle:
                        lst.append(eng)
                        keep = True
        finally:
            if not keep:
                with cls.lock:
                    cls.closed += 1
                eng.close()

    @classmethod
    def close_all(cls):
        with cls.lock:
            engines = [e for lst in cls.idle.values() for e in lst]
            cls.idle.clear()
            cls.closed += len(engines)
        for e in engines:
            e.close()

    @classmethod
    def stats(cls):
        with cls.lock:
            idle = [e for lst in cls.idle.values() for e in lst]
            return {"contexts_created": cls.created, "contexts_recycled": cls.recycled, "contexts_closed": cls.closed,
                    "contexts_idle": len(idle), "idle_device_bytes": sum(e.ctx.get_stat("device_bytes") for e in idle)}


def pool_stats():
    """Counters of the MAPs context pool: contexts created / recycled / closed / idle, device bytes the idle ones hold."""
    return _Pool.stats()


class _Shared:
    """One lazily created engine per device for the function spellings (MAP, calc_map, extra_metrics): a context
    is not thread safe, so every use holds its lock.  MAPs objects borrow theirs from _Pool instead."""
    lock = threading.Lock()
    engines = {}

    @classmethod
    def get(cls, device):
        with cls.lock:
            if device not in cls.engines:
                e = RetrievalEngine(device)
                e.lock = threading.RLock()
                cls.engines[device] = e
            return cls.engines[device]

    @classmethod
    def close_all(cls):
        with cls.lock:
            for e in cls.engines.values():
                e.close()
            cls.engines.clear()


def release_engines():
    """Free the GPU contexts (and their device memory) behind MAP / calc_map / extra_metrics and the idle ones of MAPs' pool,
    and hand the library's cache of device and pinned blocks back to the HIP runtime."""
    _Shared.close_all()
    _Pool.close_all()
    _native.release_cache()


def _check_shapes(q_codes, db_codes, q_labels, db_labels, R):
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


def _kind(ctx, which):
    """What hg_set_*_f32 found in a float table: 'ones' (every entry +1: reads as either spelling), 'pm1' (every
    entry +-1), 'bits' (every entry 0/1), 'ternary' (-1/0/+1 mixed) or 'real'."""
    other, zeros, neg, _ = ctx.census(which)
    if other:
        return "real"
    if not zeros:
        return "pm1" if neg else "ones"
    return "bits" if not neg else "ternary"


def _load_database(eng, db_codes, db_labels, mode="reference", floats=None):
    """Pack on the host, upload (hg_set_database_f32).  The float table itself goes to the GPU only when it may be ranked
    by inner product: never for the spellings that binarise or insist on binary codes, and in 'reference' mode only if
    the database is not a +-1 code (a +-1 database meeting real-valued queries is uploaded again with it, below)."""
    eng.b = eng.C = eng.N = eng.db_kind = eng.db_src = None      # nothing is resident until this load has succeeded
    eng.resident = None                               # whoever loads another database (extra_metrics, MAPs) ends _evaluate's shortcut
    eng.ctx.set_option("keep_floats", floats if floats is not None else (2 if mode == "reference" else 0))
    bad_c, bad_l = eng.ctx.set_database_f32(db_codes, db_labels)
    if bad_l:
        raise ValueError("labels must be {0,1} indicator matrices")
    eng.b, eng.C = db_codes.shape[1], db_labels.shape[1]
    eng.db_kind = _kind(eng.ctx, 0)
    eng.N = db_codes.shape[0]
    eng.db_src = (db_codes, db_labels)                # for that second upload


def _rank(eng, q_codes, q_labels, R, mode):
    """Queries against the engine's resident database.  mode:
       'reference'  MAPs: what np.dot ranks (metric.py:13-14).  +-1 codes on both sides -> Hamming kernels (the same
                    order, ties by index); anything else -- real-valued tanh outputs, {0,1} bits (np.dot counts common
                    ones, not a Hamming distance), codes with zeros -- goes through the float32 inner-product ranking;
       'sign'       MAPs(binarize=True): sign() first, then Hamming;
       'codes'      MAP / calc_map (north-star spelling): binary codes only, {0,1} bits or +-1, ranked by Hamming
                    distance; anything else is an error."""
    qbad_c, qbad_l = eng.ctx.set_queries_f32(q_codes, q_labels)
    if qbad_l:
        raise ValueError("labels must be {0,1} indicator matrices")
    qk, dk = _kind(eng.ctx, 1), eng.db_kind
    if mode == "sign" or {qk, dk} <= {"pm1", "ones"}:
        return eng.average_precisions(R)
    if mode == "codes":
        if {qk, dk} <= {"bits", "ones"}:
            return eng.average_precisions(R)
        raise ValueError("codes must be binary -- all {0,1} or all {-1,+1}, the same spelling for queries and "
                         "database (found %s queries, %s database); binarise first (np.sign), or hand real-valued "
                         "features to MAPs.get_maps_by_feature, which ranks them by inner product like metric.py:13" % (qk, dk))
    if q_codes.shape[1] > 255:                        # (the loaders take up to 255 columns)
        raise ValueError("inner-product ranking supports up to 255 features (have %d)" % q_codes.shape[1])
    if not eng.ctx.census(0)[3]:             # a +-1 database whose floats stayed on the host: bring them over now
        src = eng.db_src
        _load_database(eng, src[0], src[1], floats=1)
        eng.ctx.set_queries_f32(q_codes, q_labels)
    return eng.ctx.map_real(R)


def _evaluate(q_codes, db_codes, q_labels, db_labels, R, device, mode):
    db_codes, q_codes = np.asarray(db_codes), np.asarray(q_codes)
    db_labels, q_labels = np.asarray(db_labels), np.asarray(q_labels)
    _check_shapes(q_codes, db_codes, q_labels, db_labels, R)
    eng = _Shared.get(device)
    with eng.lock:
        # the same read-only arrays as the last call (an evaluation loop over one database): the packed copy on the GPU
        # is still theirs -- skip pack + upload, like MAPs does (a writable array may have changed in place: reload)
        # (every load goes through _load_database, which clears `resident`: another database loaded into this shared
        # engine in between -- extra_metrics -- can never be mistaken for this one)
        res = eng.resident
        same = (res is not None and res[0] is db_codes and res[1] is db_labels and res[2] == mode
                and eng.db_src is not None and eng.db_src[0] is db_codes and eng.db_src[1] is db_labels
                and not db_codes.flags.writeable and not db_labels.flags.writeable)
        if not same:
            eng.resident = None
            _load_database(eng, db_codes, db_labels, mode)
            eng.resident = (db_codes, db_labels, mode)
        try:
            ap, rel = _rank(eng, q_codes, q_labels, R, mode)
        except Exception:
            eng.resident = None
            raise
    return mean_over_hits(ap, rel), ap, rel


# ------------------------------------------------------------------ reference surface
class MAPs:
    """Same constructor and method as lib/metric.py:4-24.

    The object holds one GPU context from first use to close() / garbage collection -- borrowed from a process-wide pool
    (_Pool: main.py:164 builds a new MAPs per evaluation, and a context of its own each time would cost allocations worth
    many times the evaluation) -- and a lock, so several live MAPs objects -- different R, different threads -- never share
    device state.  main.py:237-240 evaluates
    the same database against fresh queries again and again; `set_database` keeps it packed on the GPU between
    calls (explicit), and a database whose arrays are the SAME read-only objects as last time is reused
    automatically (a writable array may have changed in place, so it is uploaded again)."""

    def __init__(self, r, device=0, binarize=False):
        self.R = r
        self.device = device
        self.binarize = binarize
        self._eng = None
        self._lock = threading.RLock()
        self._resident = None          # (output array, label array) the engine holds, or ("explicit",)

    # lib/metric.py:8-10 (unused by the reference; kept for surface parity)
    @staticmethod
    def distance(a, b):
        return np.dot(a, b)

    def _engine(self):
        if self._eng is None:
            self._eng = _Pool.acquire(self.device)
        return self._eng

    def close(self):
        with self._lock:
            if self._eng is not None:
                eng, self._eng = self._eng, None
                self._resident = None
                _Pool.release(eng)

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    def set_database(self, database):
        """Upload + pack `database` (.output [N, b], .label [N, C]) once; get_maps_by_feature(None, query) -- or with
        the same object -- then ranks against the resident copy."""
        out, lab = np.asarray(database.output), np.asarray(database.label)
        if out.ndim != 2 or lab.ndim != 2 or out.shape[0] != lab.shape[0]:
            raise ValueError("database.output must be [N, b] and database.label [N, C]")
        with self._lock:
            self._resident = None                      # a failed load leaves NO database (never the previous one half replaced)
            self._guard(_load_database, self._engine(), out, lab, "sign" if self.binarize else "reference")
            self._resident = ("explicit", database)

    def _guard(self, fn, *args):
        """A HIP-level failure (a fault, out of memory) may leave the context unusable: it is destroyed, not recycled."""
        try:
            return fn(*args)
        except _native.HashganNativeError as e:
            if e.code in (_native.HG_ERR_HIP, _native.HG_ERR_NOMEM) and self._eng is not None:
                self._eng.poisoned = True
            raise

    def _ensure_database(self, database):
        if database is None:
            if self._resident is None:
                raise ValueError("no resident database: call set_database first")
            return
        if self._resident is not None and self._resident[0] == "explicit" and self._resident[1] is database:
            return
        out, lab = np.asarray(database.output), np.asarray(database.label)
        if (self._resident is not None and self._resident[0] == "auto" and self._resident[1] is out
                and self._resident[2] is lab and not out.flags.writeable and not lab.flags.writeable):
            return                                     # the very same immutable arrays as last time
        if out.ndim != 2 or lab.ndim != 2 or out.shape[0] != lab.shape[0]:
            raise ValueError("database.output must be [N, b] and database.label [N, C]")
        self._resident = None                          # a failed load leaves NO database
        self._guard(_load_database, self._engine(), out, lab, "sign" if self.binarize else "reference")
        self._resident = ("auto", out, lab)

    def get_maps_by_feature(self, database, query):
        """database/query: objects with .output [n, b] and .label [n, C] (main.py:157); database first."""
        q_codes, q_labels = np.asarray(query.output), np.asarray(query.label)
        with self._lock:
            self._ensure_database(database)
            eng = self._engine()
            if eng.b is None:
                self._resident = None
                raise ValueError("no resident database: the last load failed")
            if q_codes.ndim != 2 or q_codes.shape[1] != eng.b:
                raise ValueError("query and database codes must be [n, b] with the same b")
            if q_labels.ndim != 2 or q_labels.shape[1] != eng.C or q_labels.shape[0] != q_codes.shape[0]:
                raise ValueError("query labels must be [Q, C] with the database's C")
            R = int(self.R)
            if not 1 <= R <= eng.N:
                raise ValueError("R=%d must be in 1..N (N=%d database rows)" % (R, eng.N))
            ap, rel = self._guard(_rank, eng, q_codes, q_labels, R, "sign" if self.binarize else "reference")
        return mean_over_hits(ap, rel)


def MAP(query_codes, db_codes, query_labels, db_labels, R, device=0):
    """mAP@R of binary codes, query-first argument order (BASELINE.json north star).  Codes must be binary --
    {0,1} bits or +-1, ranked by Hamming distance; real-valued features belong to MAPs.get_maps_by_feature."""
    return _evaluate(query_codes, db_codes, query_labels, db_labels, int(R), device, "codes")[0]


calc_map = MAP


def MAP_per_query(query_codes, db_codes, query_labels, db_labels, R, device=0):
    """(mAP, ap [Q] with nan for skipped queries, rel [Q]); binary codes like MAP."""
    return _evaluate(query_codes, db_codes, query_labels, db_labels, int(R), device, "codes")
