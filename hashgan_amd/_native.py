"""ctypes binding of libhashgan_amd.so (the C ABI in include/hashgan_amd.h).

There is no CPU implementation behind this module: if the library is missing,
fails to load, or finds no GPU, the error propagates (HashganNativeError).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HG_OK, HG_ERR_ARG, HG_ERR_HIP, HG_ERR_STATE, HG_ERR_NOMEM = 0, -1, -2, -3, -4
IDX_NONE = 0xFFFFFFFF
COMM_ID_BYTES = 128


class HashganNativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hashgan_amd native error %d: %s" % (code, msg))
        self.code = code


_p = C.c_void_p
_i64 = C.c_int64
# name -> (argtypes); every function returns int except hg_last_error
_SIGNATURES = {
    "hg_version": [],
    "hg_device_count": [C.POINTER(C.c_int)],
    "hg_init": [C.c_int, C.POINTER(_p)],
    "hg_destroy": [_p],
    "hg_pack_sign_f32": [_p, _i64, C.c_int, _p],
    "hg_set_database": [_p, _p, _p, _i64, C.c_int, C.c_int, _i64, _i64],
    "hg_set_queries": [_p, _p, _p, _i64],
    "hg_set_database_f32": [_p, _p, _p, _i64, C.c_int, C.c_int, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64)],
    "hg_set_queries_f32": [_p, _p, _p, _i64, C.POINTER(_i64), C.POINTER(_i64)],
    "hg_get_packed": [_p, C.c_int, _p, _p],
    "hg_hist": [_p],
    "hg_hist_buffer": [_p, C.POINTER(_p), C.POINTER(_i64)],
    "hg_plan": [_p, _i64, _p, C.c_int, C.c_int],
    "hg_select": [_p],
    "hg_bet_eligible": [_p, _i64, C.c_int, C.POINTER(C.c_int)],
    "hg_sample_hist": [_p, _i64],
    "hg_guess": [_p, _i64, _p, C.c_int, C.c_int],
    "hg_select_candidates": [_p],
    "hg_rank": [_p, _p, C.c_int, C.c_int, C.POINTER(C.c_int)],
    "hg_bet_verdict": [_p, C.POINTER(C.c_int)],
    "hg_select_ranked": [_p],
    "hg_merge_ranked": [_p, _p, _p, C.c_int, C.POINTER(C.c_int)],
    "hg_merge_ap_part": [_p, _p, _p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.POINTER(_p), C.POINTER(C.c_int64)],
    "hg_unpack_parts": [_p, _p, C.c_int, C.c_int64, _p, _p, C.POINTER(C.c_int)],
    "hg_pack_sample_by_owner": [_p, C.c_int, C.POINTER(_p), C.POINTER(C.c_int64)],
    "hg_guess_owned": [_p, _i64, _p, C.c_int, C.c_int, C.POINTER(_p), C.POINTER(C.c_int64)],
    "hg_guess_finish": [_p, _i64, _p, C.c_int, C.c_int],
    "hg_pack_ranked_by_owner": [_p, C.c_int, C.POINTER(_p), C.POINTER(C.c_int64)],
    "hg_merge_ap_owned": [_p, _p, C.c_int, C.c_int, C.POINTER(_p), C.POINTER(C.c_int64)],
    "hg_match": [_p],
    "hg_match_buffer": [_p, C.POINTER(_p), C.POINTER(_i64)],
    "hg_merge_match": [_p, _p, C.c_int],
    "hg_ap": [_p],
    "hg_topr_buffers": [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_i64)],
    "hg_merge_topr": [_p, _p, _p, C.c_int],
    "hg_topr": [_p, _i64],
    "hg_map": [_p, _i64, _p, _p],
    "hg_map_begin": [_p, _i64],
    "hg_map_end": [_p, _p, _p],
    "hg_map_real": [_p, _i64, _p, _p],
    "hg_topr_real": [_p, _i64],
    "hg_get_topr_real": [_p, _p, _p],
    "hg_get_topr": [_p, _p, _p],
    "hg_get_match": [_p, _p],
    "hg_get_ap": [_p, _p, _p],
    "hg_get_hist": [_p, _p],
    "hg_comm_unique_id": [_p],
    "hg_comm_init": [_p, _p, C.c_int, C.c_int],
    "hg_comm_destroy": [_p],
    "hg_comm_info": [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "hg_allgather": [_p, C.c_int, _p, _i64, C.POINTER(_p)],
    "hg_alltoall": [_p, C.c_int, _p, _i64, C.POINTER(_p)],
    "hg_allgather_topr": [_p],
    "hg_allreduce_max_f64": [_p, C.POINTER(C.c_double)],
    "hg_barrier": [_p],
    "hg_scratch": [_p, C.c_int, _i64, C.POINTER(_p)],
    "hg_memcpy_dtod": [_p, _p, _p, _i64],
    "hg_memcpy_dtoh": [_p, _p, _p, _i64],
    "hg_memcpy_htod": [_p, _p, _p, _i64],
    "hg_synchronize": [_p],
    "hg_set_stream": [_p, _p],
    "hg_set_option": [_p, C.c_char_p, _i64],
    "hg_get_stat": [_p, C.c_char_p, C.POINTER(_i64)],
    "hg_get_census": [_p, C.c_int, C.POINTER(_i64)],
    "hg_shard_step": [_p, _i64, C.c_int, _p, _p, C.POINTER(C.c_int)],
    "hg_trim": [_p],
    "hg_release_cache": [],
    "hg_preload": [_p],
    "hg_timing_enable": [_p, C.c_int],
    "hg_timing_reset": [_p],
    "hg_timing_read": [_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_int)],
}
EXPORTS = sorted(list(_SIGNATURES) + ["hg_last_error"])

_lib = None


def library_path():
    """The production library, or the one HG_LIBRARY names (e.g. the probe build)."""
    return os.environ.get("HG_LIBRARY") or _build.LIB_PATH


def load():
    """dlopen the library (no GPU needed for this step) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HashganNativeError(HG_ERR_STATE, "%s not built -- run `python -m hashgan_amd.build` "
                                 "(or __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.hg_last_error.argtypes = []
    lib.hg_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != HG_OK:
        raise HashganNativeError(rc, load().hg_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


def _carray(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _labels_i64(labels):
    """Labels as the int64 matrix the C ABI takes.  The {0,1} census runs on the GPU AFTER this conversion, so a
    conversion that changes a value (0.5 -> 0, 1.7 -> 1) would let a matrix through that lib/metric.py:19's
    `database.label == label` never matches: such labels are refused here."""
    a = np.asarray(labels)
    out = _carray(a, np.int64)
    if a.dtype.kind not in "iub" and not np.array_equal(out, a):
        raise ValueError("labels must be {0,1} indicator matrices")
    return out


class Context:
    """One GPU context (hg_ctx).  Thin, 1:1 with the C ABI."""

    def __init__(self, device=0):
        self._lib = load()
        h = _p()
        check(self._lib.hg_init(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.Q = self.N = self.R = self.b = self.C = None
        self.options_touched = set()       # keys set through set_option (metric's engine pool only recycles contexts on their defaults)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs ---------------------------------------------------------------
    def set_database(self, codes_u64, labels_u64, b, C_, idx_base=0, n_total=None):
        codes = _carray(codes_u64, np.uint64)
        labels = _carray(labels_u64, np.uint64)
        N = codes.shape[0]
        n_total = N if n_total is None else int(n_total)
        check(self._lib.hg_set_database(self._h, _ptr(codes), _ptr(labels), N, int(b), int(C_), int(idx_base), n_total))
        self.N, self.b, self.C = N, int(b), int(C_)

    def set_queries(self, codes_u64, labels_u64):
        codes = _carray(codes_u64, np.uint64)
        labels = _carray(labels_u64, np.uint64)
        check(self._lib.hg_set_queries(self._h, _ptr(codes), _ptr(labels), codes.shape[0]))
        self.Q = codes.shape[0]

    def set_database_f32(self, features, labels, idx_base=0, n_total=None):
        """float32 [N, b] features + int64 [N, C] labels; binarise + pack on the GPU.
        -> (entries outside {-1,0,+1}, label entries outside {0,1})"""
        x = _carray(features, np.float32)
        lab = _labels_i64(labels)
        N, b = x.shape
        n_total = N if n_total is None else int(n_total)
        bc, bl = _i64(), _i64()
        check(self._lib.hg_set_database_f32(self._h, _ptr(x), _ptr(lab), N, b, lab.shape[1], int(idx_base), n_total,
                                            C.byref(bc), C.byref(bl)))
        self.N, self.b, self.C = N, b, lab.shape[1]
        return bc.value, bl.value

    def set_queries_f32(self, features, labels):
        x = _carray(features, np.float32)
        lab = _labels_i64(labels)
        bc, bl = _i64(), _i64()
        check(self._lib.hg_set_queries_f32(self._h, _ptr(x), _ptr(lab), x.shape[0], C.byref(bc), C.byref(bl)))
        self.Q = x.shape[0]
        return bc.value, bl.value

    def get_packed(self, which):
        n = self.Q if which else self.N
        codes = np.empty((n, (self.b + 31) // 32), dtype=np.uint32)
        labels = np.empty((n, (self.C + 63) // 64), dtype=np.uint64)
        check(self._lib.hg_get_packed(self._h, int(which), _ptr(codes), _ptr(labels)))
        return codes, labels

    # -- stages -----------------------------------------------------------------
    def hist(self):
        check(self._lib.hg_hist(self._h))

    def hist_buffer(self):
        p, n = _p(), _i64()
        check(self._lib.hg_hist_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def plan(self, R, dev_hist_all=None, G=1, rank=0):
        check(self._lib.hg_plan(self._h, int(R), _p(dev_hist_all) if dev_hist_all else None, int(G), int(rank)))
        self.R = int(R)

    def select(self):
        check(self._lib.hg_select(self._h))

    def bet_eligible(self, R, world=1):
        v = C.c_int()
        check(self._lib.hg_bet_eligible(self._h, int(R), int(world), C.byref(v)))
        return bool(v.value)

    def sample_hist(self, R):
        check(self._lib.hg_sample_hist(self._h, int(R)))

    def guess(self, R, dev_hist_all=None, G=1, rank=0):
        check(self._lib.hg_guess(self._h, int(R), _p(dev_hist_all) if dev_hist_all else None, int(G), int(rank)))
        self.R = int(R)

    def select_candidates(self):
        check(self._lib.hg_select_candidates(self._h))

    def rank(self, dev_hist_all=None, G=1, rank=0):
        lost = C.c_int()
        check(self._lib.hg_rank(self._h, _p(dev_hist_all) if dev_hist_all else None, int(G), int(rank), C.byref(lost)))
        return None if lost.value < 0 else bool(lost.value)      # None: verdict deferred (option defer_verdict)

    def select_ranked(self):
        check(self._lib.hg_select_ranked(self._h))

    def merge_ranked(self, dev_hist_all=None, dev_bits_all=None, G=1):
        lost = C.c_int()
        check(self._lib.hg_merge_ranked(self._h, _p(dev_hist_all) if dev_hist_all else None,
                                        _p(dev_bits_all) if dev_bits_all else None, int(G), C.byref(lost)))
        return None if lost.value < 0 else bool(lost.value)

    def merge_ap_part(self, dev_hist_all, dev_bits_all, G, q0, nq, width):
        """-> (device address, bytes) of this rank's part: hg_merge_ap_part."""
        p, n = _p(), _i64()
        check(self._lib.hg_merge_ap_part(self._h, _p(dev_hist_all) if dev_hist_all else None, _p(dev_bits_all) if dev_bits_all else None,
                                         int(G), int(q0), int(nq), int(width), C.byref(p), C.byref(n)))
        return p.value, n.value

    def unpack_parts(self, dev_parts_all, G, width):
        """-> (ap [Q] float64, rel [Q] int64, bet lost?) from the gathered parts: hg_unpack_parts."""
        ap = np.empty(self.Q, dtype=np.float64)
        rel = np.empty(self.Q, dtype=np.int64)
        lost = C.c_int()
        check(self._lib.hg_unpack_parts(self._h, _p(dev_parts_all), int(G), int(width), _ptr(ap), _ptr(rel), C.byref(lost)))
        return ap, rel, bool(lost.value)

    # -- the same bet with its exchanges routed by query owner (hg_pack_sample_by_owner ... hg_merge_ap_owned)
    def pack_sample_by_owner(self, G):
        p, n = _p(), _i64()
        check(self._lib.hg_pack_sample_by_owner(self._h, int(G), C.byref(p), C.byref(n)))
        return p.value, n.value                      # (device address of [G] blocks, bytes per block)

    def guess_owned(self, R, dev_recv, G, rank):
        p, n = _p(), _i64()
        check(self._lib.hg_guess_owned(self._h, int(R), _p(dev_recv), int(G), int(rank), C.byref(p), C.byref(n)))
        return p.value, n.value

    def guess_finish(self, R, dev_answers, G, rank):
        check(self._lib.hg_guess_finish(self._h, int(R), _p(dev_answers), int(G), int(rank)))
        self.R = int(R)

    def pack_ranked_by_owner(self, G):
        p, n = _p(), _i64()
        check(self._lib.hg_pack_ranked_by_owner(self._h, int(G), C.byref(p), C.byref(n)))
        return p.value, n.value

    def merge_ap_owned(self, dev_recv, G, rank):
        p, n = _p(), _i64()
        check(self._lib.hg_merge_ap_owned(self._h, _p(dev_recv), int(G), int(rank), C.byref(p), C.byref(n)))
        return p.value, n.value

    def bet_verdict(self):
        lost = C.c_int()
        check(self._lib.hg_bet_verdict(self._h, C.byref(lost)))
        return bool(lost.value)

    def match(self):
        check(self._lib.hg_match(self._h))

    def match_buffer(self):
        p, n = _p(), _i64()
        check(self._lib.hg_match_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def merge_match(self, dev_bits_all, G):
        check(self._lib.hg_merge_match(self._h, _p(dev_bits_all), int(G)))

    def ap(self):
        check(self._lib.hg_ap(self._h))

    def topr_buffers(self):
        a, b, n = _p(), _p(), _i64()
        check(self._lib.hg_topr_buffers(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def merge_topr(self, dev_idx_all, dev_dist_all, G):
        check(self._lib.hg_merge_topr(self._h, _p(dev_idx_all), _p(dev_dist_all), int(G)))

    # -- one-shot ---------------------------------------------------------------
    def topr(self, R):
        check(self._lib.hg_topr(self._h, int(R)))
        self.R = int(R)

    def map(self, R):
        ap = np.empty(self.Q, dtype=np.float64)
        rel = np.empty(self.Q, dtype=np.int64)
        check(self._lib.hg_map(self._h, int(R), _ptr(ap), _ptr(rel)))
        self.R = int(R)
        return ap, rel

    def map_begin(self, R):
        """First half of map(): enqueue a step and return (at most two in flight; see hg_map_begin in include/hashgan_amd.h)."""
        check(self._lib.hg_map_begin(self._h, int(R)))
        self.R = int(R)
        self._in_flight = getattr(self, "_in_flight", []) + [self.Q]      # (a step's results have the length of ITS query table)

    def map_end(self):
        """Second half of map(): wait for the oldest step in flight, return its (ap, rel)."""
        pending = getattr(self, "_in_flight", [])
        nq = pending[0] if pending else self.Q
        ap = np.empty(nq, dtype=np.float64)
        rel = np.empty(nq, dtype=np.int64)
        check(self._lib.hg_map_end(self._h, _ptr(ap), _ptr(rel)))
        self._in_flight = pending[1:]
        return ap, rel

    # -- real-valued features ---------------------------------------------------
    def map_real(self, R):
        ap = np.empty(self.Q, dtype=np.float64)
        rel = np.empty(self.Q, dtype=np.int64)
        check(self._lib.hg_map_real(self._h, int(R), _ptr(ap), _ptr(rel)))
        self.R = int(R)
        return ap, rel

    def topr_real(self, R):
        check(self._lib.hg_topr_real(self._h, int(R)))
        self.R = int(R)
        idx = np.empty((self.Q, self.R), dtype=np.uint32)
        score = np.empty((self.Q, self.R), dtype=np.float32)
        check(self._lib.hg_get_topr_real(self._h, _ptr(idx), _ptr(score)))
        return idx, score

    # -- results ----------------------------------------------------------------
    def get_topr(self):
        idx = np.empty((self.Q, self.R), dtype=np.uint32)
        dist = np.empty((self.Q, self.R), dtype=np.uint8)
        check(self._lib.hg_get_topr(self._h, _ptr(idx), _ptr(dist)))
        return idx, dist

    def get_match(self):
        m = np.empty((self.Q, self.R), dtype=np.uint8)
        check(self._lib.hg_get_match(self._h, _ptr(m)))
        return m

    def get_ap(self):
        ap = np.empty(self.Q, dtype=np.float64)
        rel = np.empty(self.Q, dtype=np.int64)
        check(self._lib.hg_get_ap(self._h, _ptr(ap), _ptr(rel)))
        return ap, rel

    def get_hist(self):
        h = np.empty((self.b + 1, self.Q), dtype=np.uint32)
        check(self._lib.hg_get_hist(self._h, _ptr(h)))
        return h

    # -- collectives (RCCL) -------------------------------------------------------
    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(self._lib.hg_comm_init(self._h, buf, int(rank), int(world)))

    def comm_destroy(self):
        check(self._lib.hg_comm_destroy(self._h))

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        check(self._lib.hg_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allgather(self, slot, dev_ptr, nbytes):
        """-> device address of [world][nbytes], owned by the context (slot 0..3)."""
        out = _p()
        check(self._lib.hg_allgather(self._h, int(slot), _p(dev_ptr), int(nbytes), C.byref(out)))
        return out.value

    def alltoall(self, slot, dev_ptr, nbytes_per_peer):
        """dev_ptr = [world][nbytes_per_peer], block r to rank r -> device address of [world][nbytes_per_peer] received."""
        out = _p()
        check(self._lib.hg_alltoall(self._h, int(slot), _p(dev_ptr), int(nbytes_per_peer), C.byref(out)))
        return out.value

    def shard_step(self, R, replica_world=0):
        """One rank's whole step of the database-sharded bet in one call (hg_shard_step) -> (ap [Q] float64, rel [Q] int64, lost):
        lost False / True, or None when nothing was enqueued (the bet is not eligible: use the staged sequences)."""
        ap = np.empty(self.Q, dtype=np.float64)
        rel = np.empty(self.Q, dtype=np.int64)
        lost = C.c_int()
        check(self._lib.hg_shard_step(self._h, int(R), int(replica_world), _ptr(ap), _ptr(rel), C.byref(lost)))
        self.R = int(R)
        return ap, rel, (None if lost.value < 0 else bool(lost.value))

    def allgather_topr(self):
        check(self._lib.hg_allgather_topr(self._h))

    def allreduce_max(self, x):
        v = C.c_double(float(x))
        check(self._lib.hg_allreduce_max_f64(self._h, C.byref(v)))
        return v.value

    def barrier(self):
        check(self._lib.hg_barrier(self._h))

    def scratch(self, slot, nbytes):
        out = _p()
        check(self._lib.hg_scratch(self._h, int(slot), int(nbytes), C.byref(out)))
        return out.value

    def memcpy_dtod(self, dst, src, nbytes):
        check(self._lib.hg_memcpy_dtod(self._h, _p(dst), _p(src), int(nbytes)))

    def memcpy_dtoh(self, host_array, src, nbytes):
        check(self._lib.hg_memcpy_dtoh(self._h, _ptr(host_array), _p(src), int(nbytes)))

    def memcpy_htod(self, dst, host_array, nbytes):
        check(self._lib.hg_memcpy_htod(self._h, _p(dst), _ptr(host_array), int(nbytes)))

    def synchronize(self):
        check(self._lib.hg_synchronize(self._h))

    # -- tuning / timing ----------------------------------------------------------
    def set_stream(self, stream_handle):
        """Run on the caller's HIP stream (an int handle: a hipStream_t); None = back to a private one."""
        check(self._lib.hg_set_stream(self._h, _p(stream_handle) if stream_handle else None))
        self.options_touched.add("stream")

    def set_option(self, key, value):
        check(self._lib.hg_set_option(self._h, key.encode(), int(value)))
        self.options_touched.add(key)

    def preload(self):
        """One-time costs of every path now instead of on first use (hg_preload)."""
        check(self._lib.hg_preload(self._h))

    def trim(self):
        """Free the work buffers (they only grow); the loaded tables stay."""
        check(self._lib.hg_trim(self._h))

    def get_stat(self, key):
        v = _i64()
        check(self._lib.hg_get_stat(self._h, key.encode(), C.byref(v)))
        return v.value

    def census(self, queries):
        """(entries outside {-1,0,+1}, zeros, minus ones, floats resident on the device) of the float table hg_set_*_f32 was handed."""
        out = (_i64 * 4)()
        check(self._lib.hg_get_census(self._h, 1 if queries else 0, out))
        return out[0], out[1], out[2], bool(out[3])

    def timing_enable(self, on=True):
        """on: False/0 off, True/2 every kernel, 1 only the select pass over the pairs (k_select, k_select_mx*) and the step's span."""
        level = 2 if on is True else (0 if on is False else int(on))
        check(self._lib.hg_timing_enable(self._h, level))

    def timing_reset(self):
        check(self._lib.hg_timing_reset(self._h))

    def timing_read(self):
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (_i64 * cap)()
        n = C.c_int()
        check(self._lib.hg_timing_read(self._h, cap, names, ms, cnt, C.byref(n)))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}


def host_phase_timers(ctx):
    """Process-wide host-side cost of what a context does outside its kernels (hg_ctx.hpp, HostPhase): {phase: (total ms, calls,
    longest call ms)} for init, devmalloc, devfree, hostmalloc, hostfree, destroy, stream, event, sync (waiting for the GPU), pack (the host
    packing pass) and thread (starting the float table's staging thread)."""
    out = {}
    for ph in ("init", "devmalloc", "devfree", "hostmalloc", "hostfree", "destroy", "stream", "event", "sync", "pack", "thread"):
        out[ph] = (ctx.get_stat("host_us_" + ph) / 1e3, ctx.get_stat("host_n_" + ph), ctx.get_stat("host_max_us_" + ph) / 1e3)
    return out


def release_cache():
    """Return the process-wide cache of device / pinned blocks and streams to the HIP runtime (hg_release_cache)."""
    check(load().hg_release_cache())


def comm_unique_id():
    """128 opaque bytes (ncclUniqueId) for rank 0 to hand to the other ranks; loads RCCL."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(load().hg_comm_unique_id(buf))
    return bytes(buf)


def device_count():
    n = C.c_int()
    check(load().hg_device_count(C.byref(n)))
    return n.value


def pack_sign_f32(x):
    """float32 [n, b] -> uint64 [n, ceil(b/64)], bit = (x > 0) (hg_pack_sign_f32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, b = x.shape
    out = np.empty((n, (b + 63) // 64), dtype=np.uint64)
    check(load().hg_pack_sign_f32(_ptr(x), n, b, _ptr(out)))
    return out
