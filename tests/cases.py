"""Seeded parity cases shared by the golden generator, the oracle tests (CPU)
and the HIP parity tests (GPU).  Inputs are regenerated from the seed with
hashgan_amd.synth; only expected outputs live in tests/golden/.

Shapes follow BASELINE.json `configs` / SURVEY.md section 8 (C1..C5), on query
subsets where the unmodified reference needs 16 B per (query, db) pair.
"""
import os
import numpy as np
from hashgan_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> dict(Q, N, b, R, C, kind, seed, [flip], [q_take])
CASES = {
    # --- BASELINE configs (query subsets of the big ones) -------------------
    "c1_cifar_full":   dict(Q=1000, N=54000, b=32, R=54000, C=10, kind="cifar", seed=0xC1, flip=0.25),
    "c2_q64":          dict(Q=10000, q_take=64, N=1000000, b=64, R=5000, C=10, kind="planted", seed=0xC2, flip=0.30),
    "c3_nus_q64":      dict(Q=2100, q_take=64, N=190000, b=48, R=5000, C=81, kind="multihot", seed=0xC3, flip=0.20),
    "c4_n10m_q8":      dict(Q=10000, q_take=8, N=10000000, b=64, R=5000, C=10, kind="iid", seed=0xC4),
    "c5_b128_q32":     dict(Q=10000, q_take=32, N=1000000, b=128, R=5000, C=10, kind="planted", seed=0xC5, flip=0.35),
    # BASELINE.json configs[1] as literally written: "synthetic random codes" -- C2's shape on i.i.d. Bernoulli(1/2) bits (bench.py's c2_iid leg)
    "c2_iid_q64":      dict(Q=10000, q_take=64, N=1000000, b=64, R=5000, C=10, kind="iid", seed=0x2C2),
    # --- edge cases ----------------------------------------------------------
    "e_r_eq_n":        dict(Q=70, N=3000, b=32, R=3000, C=10, kind="planted", seed=0xE1, flip=0.25),
    "e_r_1":           dict(Q=70, N=3000, b=32, R=1, C=10, kind="planted", seed=0xE2, flip=0.25),
    "e_b1":            dict(Q=33, N=2500, b=1, R=500, C=4, kind="planted", seed=0xE3, flip=0.10),
    "e_b8":            dict(Q=64, N=4096, b=8, R=1000, C=10, kind="planted", seed=0xE4, flip=0.20),
    "e_b65_pad":       dict(Q=65, N=5001, b=65, R=777, C=10, kind="planted", seed=0xE5, flip=0.30),
    "e_b100":          dict(Q=129, N=7777, b=100, R=2000, C=81, kind="multihot", seed=0xE6, flip=0.25),
    "e_dups_alleq":    dict(Q=10, N=2000, b=16, R=300, C=5, kind="alleq", seed=0xE7),
    "e_q1_n1":         dict(Q=1, N=1, b=64, R=1, C=3, kind="iid", seed=0xE8),
    "e_some_skipped":  dict(Q=40, N=1500, b=24, R=20, C=50, kind="iid", seed=0xE9),
    "e_all_skipped":   dict(Q=5, N=300, b=16, R=10, C=10, kind="disjoint", seed=0xEA),
    "e_ragged":        dict(Q=191, N=10007, b=48, R=4999, C=10, kind="planted", seed=0xEB, flip=0.30),
    "e_big_r":         dict(Q=16, N=200000, b=64, R=150000, C=10, kind="planted", seed=0xEC, flip=0.30),
}

SMALL = [k for k in CASES if k.startswith("e_") and k != "e_big_r"]

# Real-valued (tanh-like) feature cases for the float ranking path.  Features sit on
# the grid k/64, where float32 inner products are exact in any summation order (oracle/real_map.py).
REAL_CASES = {
    "real_small":  dict(Q=50, N=3000, b=32, R=500, C=10, seed=0xF1, labels="onehot"),
    "real_b64":    dict(Q=70, N=20000, b=64, R=2000, C=10, seed=0xF2, labels="onehot"),
    "real_multi":  dict(Q=33, N=9000, b=48, R=900, C=81, seed=0xF3, labels="multihot"),
    "real_dups":   dict(Q=20, N=4000, b=16, R=1500, C=5, seed=0xF4, labels="onehot", dups=True),
    "real_b128":   dict(Q=16, N=5000, b=128, R=5000, C=10, seed=0xF5, labels="onehot"),
    "real_bet":    dict(Q=96, N=150000, b=64, R=3000, C=10, seed=0xF6, labels="onehot"),
    # codes that are NOT +-1: np.dot (metric.py:13) does not rank them by Hamming distance -- {0,1} bits count common
    # ones, zeros in a +-1 code contribute nothing -- so MAPs must route them through the inner-product ranking
    "real_bits01":  dict(Q=40, N=6000, b=32, R=800, C=10, seed=0xF7, labels="onehot", feat="bits01"),
    "real_ternary": dict(Q=40, N=6000, b=48, R=1500, C=10, seed=0xF8, labels="onehot", feat="ternary"),
}


def build_real_case(name):
    """-> dict(qf, dbf float32 on the 1/64 grid; qlab, dblab int8; R, b)."""
    from oracle import real_map as RM
    c = REAL_CASES[name]
    Q, N, b, C, seed = c["Q"], c["N"], c["b"], c["C"], c["seed"]
    if c["labels"] == "onehot":
        dblab, dcls = synth.onehot_labels(seed * 3 + 1, N, C)
        qlab, qcls = synth.onehot_labels(seed * 3 + 2, Q, C)
    else:
        dblab = synth.multihot_labels(seed * 3 + 1, N, C)
        qlab = synth.multihot_labels(seed * 3 + 2, Q, C)
    # class-correlated real features: a per-class prototype plus noise, squashed to [-1, 1], snapped to the grid
    proto = RM.quantised_features(seed ^ 0x77, C, b).astype(np.float64)
    def feats(lab, s):
        base = lab.astype(np.float64) @ proto
        noise = RM.quantised_features(s, lab.shape[0], b).astype(np.float64)
        x = np.tanh(0.8 * base + 0.9 * noise)
        return (np.round(x * 64) / 64).astype(np.float32)
    dbf, qf = feats(dblab, seed + 11), feats(qlab, seed + 12)
    if c.get("feat") in ("bits01", "ternary"):
        dbits, qbits = synth.planted_codes(seed, dblab, b, 0.25), synth.planted_codes(seed, qlab, b, 0.25, noise_seed=seed + 5)
        if c["feat"] == "bits01":
            dbf, qf = dbits.astype(np.float32), qbits.astype(np.float32)
        else:                                           # +-1 codes with a quarter of the entries zeroed
            dz = synth.random_bits(seed + 21, N, b) & synth.random_bits(seed + 22, N, b)
            qz = synth.random_bits(seed + 23, Q, b) & synth.random_bits(seed + 24, Q, b)
            dbf = ((dbits.astype(np.float32) * 2 - 1) * (1 - dz)).astype(np.float32)
            qf = ((qbits.astype(np.float32) * 2 - 1) * (1 - qz)).astype(np.float32)
    if c.get("dups"):
        dbf[N // 2:] = dbf[:N - N // 2]               # the second half duplicates the first: exact ties
    return dict(qf=qf, dbf=dbf, qlab=qlab, dblab=dblab, R=c["R"], b=b, name=name)


def _cifar_labels():
    z = np.load(os.path.join(GOLDEN_DIR, "cifar10_labels.npz"))
    def onehot(cls):
        lab = np.zeros((cls.shape[0], 10), dtype=np.int8)
        lab[np.arange(cls.shape[0]), cls] = 1
        return lab
    return onehot(z["database_cls"]), onehot(z["test_cls"])


def build_case(name):
    """-> dict(qbits, dbbits uint8 {0,1}; qlab, dblab int8 {0,1}; R, b, spec)."""
    c = CASES[name]
    Q, N, b, C, seed, kind = c["Q"], c["N"], c["b"], c["C"], c["seed"], c["kind"]
    if kind == "cifar":
        dblab, qlab = _cifar_labels()
        dbbits = synth.planted_codes(seed, dblab, b, c["flip"])
        qbits = synth.planted_codes(seed, qlab, b, c["flip"])       # same prototypes, own noise below
        qbits = qbits ^ (synth.random_bits(seed + 17, Q, b) & synth.random_bits(seed + 18, Q, b) & synth.random_bits(seed + 19, Q, b))
    elif kind == "planted":
        shard = c.get("shard", 0)          # extra database shards: same prototypes, own labels and noise
        dblab, _ = synth.onehot_labels(seed * 3 + 1 + 7919 * shard, N, C)
        qlab, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
        dbbits = synth.planted_codes(seed, dblab, b, c["flip"], noise_seed=seed + 104729 * shard if shard else None)
        qbits = synth.planted_codes(seed, qlab, b, c["flip"]) ^ (
            synth.random_bits(seed + 17, Q, b) & synth.random_bits(seed + 18, Q, b))
    elif kind == "multihot":
        dblab = synth.multihot_labels(seed * 3 + 1, N, C)
        qlab = synth.multihot_labels(seed * 3 + 2, Q, C)
        dbbits = synth.planted_codes(seed, dblab, b, c["flip"])
        qbits = synth.planted_codes(seed, qlab, b, c["flip"]) ^ (
            synth.random_bits(seed + 17, Q, b) & synth.random_bits(seed + 18, Q, b))
    elif kind == "iid":
        dblab, _ = synth.onehot_labels(seed * 3 + 1, N, C)
        qlab, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
        dbbits = synth.random_bits(seed, N, b)
        qbits = synth.random_bits(seed + 7, Q, b)
    elif kind == "alleq":          # every database code identical: one giant tie group
        dblab, _ = synth.onehot_labels(seed * 3 + 1, N, C)
        qlab, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
        dbbits = np.repeat(synth.random_bits(seed, 1, b), N, axis=0)
        qbits = synth.random_bits(seed + 7, Q, b)
    elif kind == "disjoint":       # queries use classes the database never has
        dcls = (synth.splitmix64(seed, N) % np.uint64(C // 2)).astype(np.int64)
        qcls = (synth.splitmix64(seed + 1, Q) % np.uint64(C // 2)).astype(np.int64) + C // 2
        dblab = np.zeros((N, C), np.int8); dblab[np.arange(N), dcls] = 1
        qlab = np.zeros((Q, C), np.int8); qlab[np.arange(Q), qcls] = 1
        dbbits = synth.random_bits(seed, N, b)
        qbits = synth.random_bits(seed + 7, Q, b)
    else:
        raise KeyError(kind)
    take = c.get("q_take", Q)
    return dict(qbits=np.ascontiguousarray(qbits[:take]), dbbits=dbbits,
                qlab=np.ascontiguousarray(qlab[:take]), dblab=dblab,
                R=c["R"], b=b, spec=c, name=name)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}
