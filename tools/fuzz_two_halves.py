#!/usr/bin/env python3
"""One-off stress run on the GPU: hg_map_begin / hg_map_end in random interleavings with hg_map, new queries, new R and option
changes, every result compared with a second context that only ever runs the synchronous hg_map on the same tables."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric, synth


def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([16, 32, 48, 64, 64, 100, 128]))
    N = int(rng.integers(40000, 300000))
    Q = int(rng.integers(8, 600))
    C = int(rng.choice([5, 10, 81, 130]))
    planted = bool(rng.integers(0, 2))
    dl, _ = synth.onehot_labels(seed, N, C)
    ql, _ = synth.onehot_labels(seed + 1, Q, C)
    if planted:
        db, qb = synth.planted_codes(seed + 2, dl, b, 0.25), synth.planted_codes(seed + 2, ql, b, 0.25)
    else:
        db, qb = synth.random_bits(seed + 2, N, b), synth.random_bits(seed + 3, Q, b)
    if rng.integers(0, 3) == 0:                                  # a class-sorted database: bets get lost, slices widened
        order = np.argsort(dl.argmax(1), kind="stable")
        db, dl = db[order], dl[order]
    Rs = sorted({max(1, int(N * f)) for f in rng.choice([0.001, 0.004, 0.01, 0.03, 0.2], 2)})
    a, ref = _native.Context(0), _native.Context(0)
    try:
        for c in (a, ref):
            c.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
            c.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        want = {}                                                 # (queries version, R) -> (ap, rel) of the reference context
        ver = 0

        def expect(R):
            if (ver, R) not in want:
                want[(ver, R)] = ref.map(R)
            return want[(ver, R)]
        flying = []
        for _ in range(int(rng.integers(8, 30))):
            op = int(rng.integers(0, 10))
            if op <= 3 and len(flying) < 2:
                R = int(rng.choice(Rs))
                if rng.integers(0, 4) == 0:
                    a.set_option("handicap_next_bet", 12)         # (round 6: a blind step that loses -- hg_map_end redoes it, or refuses if its tables are gone)
                a.map_begin(R)
                flying.append((ver, R))
            elif op <= 6 and flying:
                v, R = flying.pop(0)
                try:
                    ap, rel = a.map_end()
                except _native.HashganNativeError as e:
                    if e.code == _native.HG_ERR_STATE and v != ver and "replaced" in str(e):
                        continue                                  # a lost step whose queries were replaced: refused, as the header says
                    return "map_end failed: %s" % e
                e = want[(v, R)] if (v, R) in want else None
                if e is None:
                    assert v == ver
                    e = expect(R)
                if not (np.array_equal(ap, e[0], equal_nan=True) and np.array_equal(rel, e[1])):
                    return "map_end differs (R=%d)" % R
            elif op == 7:
                R = int(rng.choice(Rs))
                ap, rel = a.map(R)
                e = expect(R)
                if not (np.array_equal(ap, e[0], equal_nan=True) and np.array_equal(rel, e[1])):
                    return "map differs (R=%d)" % R
            elif op == 8:
                for (v, R) in flying:                             # the reference must see the old queries first
                    if v == ver:
                        expect(R)
                perm = rng.permutation(Q)
                qb, ql = qb[perm], ql[perm]
                ver += 1
                for c in (a, ref):
                    c.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
            else:
                a.set_option(str(rng.choice(["guess_sigma", "sample_stride"])), int(rng.choice([3, 5, 24, 40])))
        while flying:
            v, R = flying.pop(0)
            try:
                ap, rel = a.map_end()
            except _native.HashganNativeError as e:
                if e.code == _native.HG_ERR_STATE and v != ver and "replaced" in str(e):
                    continue
                return "final map_end failed: %s" % e
            e = want.get((v, R)) or expect(R)
            if not (np.array_equal(ap, e[0], equal_nan=True) and np.array_equal(rel, e[1])):
                return "final map_end differs (R=%d)" % R
        return None
    finally:
        a.close(); ref.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, bad = time.time(), 0
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed)
        if r:
            bad += 1
            print("MISMATCH seed=%d: %s" % (seed, r), flush=True)
    print("done: %d sequences, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
