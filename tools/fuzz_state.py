#!/usr/bin/env python3
"""One-off stress run on the GPU: ONE long-lived context through random sequences of operations -- new databases of other
sizes / code lengths / label widths, new queries, other R, lists, options, hg_trim, binary and real-valued rankings in turn --
every result against a fresh context's exact sequence (cached images, flags and work buffers must never go stale)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric

def fresh_exact(db, dl, qb, ql, R, want_lists):
    ctx = _native.Context(0)
    try:
        for k, v in (("optimistic", 0), ("hist_mfma", 0), ("exact_mfma", 0), ("select_mfma", 0)): ctx.set_option(k, v)
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), db.shape[1], dl.shape[1])
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        ap, rel = ctx.map(R)
        lists = None
        if want_lists:
            ctx.topr(R); lists = ctx.get_topr()
        return ap, rel, lists
    finally:
        ctx.close()

def fresh_real(dbf, dl, qf, ql, R):
    ctx = _native.Context(0)
    try:
        ctx.set_option("real_mfma", 1 if dbf.shape[1] <= 128 else 2)
        ctx.set_database_f32(dbf, dl); ctx.set_queries_f32(qf, ql)
        return ctx.map_real(R)
    finally:
        ctx.close()

def run(seed, steps):
    rng = np.random.default_rng(seed)
    ctx = _native.Context(0)
    state = {}
    def new_db():
        b = int(rng.choice([8, 32, 48, 64, 64, 100, 128, 200])); N = int(rng.integers(2000, 150000)); C = int(rng.choice([3, 10, 130]))
        state.update(b=b, N=N, C=C, db=(rng.random((N, b)) < 0.5).astype(np.uint8), dl=(rng.random((N, C)) < 0.2).astype(np.int64), real=False)
        ctx.set_database(metric.pack_codes(state["db"]), metric.pack_labels(state["dl"]), b, C)
        new_q()
    def new_db_real():
        b = int(rng.choice([16, 48, 64, 129, 200])); N = int(rng.integers(2000, 150000)); C = int(rng.choice([3, 10, 130]))
        state.update(b=b, N=N, C=C, dbf=np.tanh(rng.standard_normal((N, b))).astype(np.float32), dl=(rng.random((N, C)) < 0.2).astype(np.int64), real=True)
        ctx.set_database_f32(state["dbf"], state["dl"])
        new_q()
    def new_q():
        Q = int(rng.integers(1, 300)); state["Q"] = Q
        state["ql"] = (rng.random((Q, state["C"])) < 0.2).astype(np.int64)
        if state["real"]:
            state["qf"] = np.tanh(rng.standard_normal((Q, state["b"]))).astype(np.float32)
            ctx.set_queries_f32(state["qf"], state["ql"])
        else:
            state["qb"] = (rng.random((Q, state["b"])) < 0.5).astype(np.uint8)
            ctx.set_queries(metric.pack_codes(state["qb"]), metric.pack_labels(state["ql"]))
    new_db()
    log = []
    try:
        for step in range(steps):
            op = rng.choice(["db", "dbreal", "q", "map", "map", "map", "topr", "opt", "trim"])
            log.append(str(op))
            if op == "db": new_db()
            elif op == "dbreal": new_db_real()
            elif op == "q": new_q()
            elif op == "trim": ctx.trim()
            elif op == "opt":
                k = str(rng.choice(["compact_records", "select_mfma", "rank_cnt", "hist_mfma", "guess_sigma", "max_segments", "real_mfma", "real_sort_lds"]))
                v = int(rng.choice({"compact_records": [0, 1], "select_mfma": [0, 1], "rank_cnt": [0, 1], "hist_mfma": [0, 1, 2], "guess_sigma": [1, 5],
                                    "max_segments": [16, 2048], "real_mfma": [2, 2, 1], "real_sort_lds": [0, 1]}[k]))
                if k == "real_mfma" and state["b"] > 128: v = 2
                ctx.set_option(k, v); log[-1] += ":%s=%d" % (k, v)
            else:
                N = state["N"]; R = max(1, min(N, int(N * float(rng.choice([0.002, 0.02, 0.2, 1.0])))))
                if state["Q"] * R > 20_000_000: R = max(1, 20_000_000 // state["Q"])
                if state["real"]:
                    ap, rel = ctx.map_real(R)
                    wap, wrel = fresh_real(state["dbf"], state["dl"], state["qf"], state["ql"], R)
                    ok = np.array_equal(ap, wap, equal_nan=True) and np.array_equal(rel, wrel)
                else:
                    wap, wrel, wl = fresh_exact(state["db"], state["dl"], state["qb"], state["ql"], R, op == "topr")
                    if op == "topr":
                        ctx.topr(R); idx, dist = ctx.get_topr()
                        ok = np.array_equal(idx, wl[0]) and np.array_equal(dist, wl[1])
                    else:
                        ap, rel = ctx.map(R)
                        ok = np.array_equal(ap, wap, equal_nan=True) and np.array_equal(rel, wrel)
                if not ok:
                    return "MISMATCH seed=%d step=%d op=%s real=%s b=%d N=%d Q=%d R=%d C=%d log=%s" % (seed, step, op, state["real"], state["b"], state["N"], state["Q"], R, state["C"], log[-12:])
        return "ok seed=%d" % seed
    finally:
        ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; t = time.time()
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = run(seed, 25)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 5 == 0: print(r, flush=True)
    print("done: %d sequences, %d mismatches, %.0f s" % (n, bad, time.time() - t))
