#!/usr/bin/env python3
"""One-off stress run on the GPU: random option combinations (kernel choices, record formats, segment geometry, thin safety
margins that make bets fail) on random shapes; hg_map and hg_topr against the vector-ALU exact sequence of the same library."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

OPTS = {"select_mfma": [0, 1], "compact_records": [0, 1], "rank_lds": [0, 1, 2], "select_packed": [0, 3],
        "hist_mfma": [0, 1, 2], "second_bet": [0, 1], "guess_sigma": [0, 1, 2, 5], "max_segments": [7, 64, 2048],
        "target_units": [64, 4096, 16384], "sample_stride": [0, 3, 24, 200], "optimistic": [1, 1, 1, 0], "cand_budget_x10": [11, 15, 40], "all_rows_shortcut": [0, 1],
        "staged_lists": [0, 1], "min_segment": [16, 4096, 65536], "step_graph": [0, 1], "ap_recip": [0, 1]}
DEFAULT = {"select_mfma": 1, "compact_records": 1, "rank_lds": 2, "select_packed": 3, "hist_mfma": 2,
           "second_bet": 1, "guess_sigma": 5, "max_segments": 2048, "target_units": 16384, "sample_stride": 0, "optimistic": 1, "cand_budget_x10": 40, "all_rows_shortcut": 1,
           "staged_lists": 1, "min_segment": 256, "step_graph": 0, "ap_recip": 1}

def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([8, 24, 32, 48, 64, 64, 100, 128, 200]))
    N = int(rng.integers(3000, 250000)); Q = int(rng.integers(1, 500))
    frac = float(rng.choice([0.001, 0.005, 0.02, 0.08, 0.3, 1.0])); R = min(N, max(1, int(N * frac)))
    if Q * R > 30_000_000: R = max(1, 30_000_000 // Q)
    C = int(rng.choice([3, 10, 81, 130]))
    dl, _ = synth.onehot_labels(seed * 3 + 1, N, C); ql, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
    if rng.random() < 0.5: db = synth.planted_codes(seed, dl, b, 0.25); qb = synth.planted_codes(seed, ql, b, 0.25)
    else: db = (rng.random((N, b)) < 0.5).astype(np.uint8); qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        def run(opts):
            for k, v in DEFAULT.items(): ctx.set_option(k, v)
            for k, v in opts.items(): ctx.set_option(k, v)
            ap, rel = ctx.map(R)
            ctx.topr(R)
            idx, dist = ctx.get_topr()
            ap2, rel2 = ctx.map(R)                    # and once more: state left behind by the list call
            return ap, rel, idx, dist, ap2, rel2
        if "-v" in sys.argv: print("  b=%d N=%d Q=%d R=%d C=%d" % (b, N, Q, R, C), flush=True)
        ref = run({"optimistic": 0, "hist_mfma": 0, "select_mfma": 0, "ap_recip": 0})
        for trial in range(4):
            opts = {k: int(rng.choice(v)) for k, v in OPTS.items() if rng.random() < 0.5}
            if "-v" in sys.argv: print("   trial", trial, opts, flush=True)
            got = run(opts)
            ok = (np.array_equal(got[0], ref[0], equal_nan=True) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
                  and np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[0], equal_nan=True) and np.array_equal(got[5], ref[1]))
            if not ok:
                return "MISMATCH seed=%d b=%d N=%d Q=%d R=%d C=%d opts=%s" % (seed, b, N, Q, R, C, opts)
        return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d" % (seed, b, N, Q, R, C)
    finally:
        ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; t = time.time()
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 10 == 0: print(r, flush=True)
    print("done: %d shapes, %d mismatches, %.0f s" % (n, bad, time.time() - t))
