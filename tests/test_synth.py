"""Seeded inputs: the packed, offset-capable generators bench.py's sharded leg uses draw exactly the rows of the
global arrays tests/cases.py builds (so every rank of a C4 run holds its slice of ONE database)."""
import numpy as np
from hashgan_amd import metric, sharded, synth


def test_packed_generators_equal_packing_the_bit_matrices():
    for b in (1, 32, 48, 64, 100, 128):
        bits = synth.random_bits(5, 777, b)
        assert np.array_equal(metric.pack_codes(bits), synth.random_code_words(5, 777, b))
        assert np.array_equal(metric.pack_codes(bits[300:700]), synth.random_code_words(5, 400, b, row_offset=300))
    for C in (3, 10, 81, 150):
        lab, _ = synth.onehot_labels(7, 500, C)
        assert np.array_equal(metric.pack_labels(lab), synth.onehot_label_words(7, 500, C))
        assert np.array_equal(metric.pack_labels(lab[123:456]), synth.onehot_label_words(7, 333, C, row_offset=123))


def test_bench_shards_tile_the_c4_database():
    import bench
    spec = dict(bench.WORKLOADS["c4"], N=40000, Q=50)
    qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
    parts = [bench.build_packed(spec, base, rows) for base, rows in sharded.shard_bounds(spec["N"], 3)]
    assert np.array_equal(np.concatenate([p[2] for p in parts]), dw)
    assert np.array_equal(np.concatenate([p[3] for p in parts]), dl)
    for p in parts:
        assert np.array_equal(p[0], qw) and np.array_equal(p[1], ql)
    # and they are the arrays of tests/cases.py's iid kind (the golden c4_n10m_q8 is generated from those)
    from tests import cases
    cases.CASES["_t"] = dict(Q=50, N=40000, b=64, R=100, C=10, kind="iid", seed=0xC4)
    try:
        c = cases.build_case("_t")
    finally:
        del cases.CASES["_t"]
    assert np.array_equal(metric.pack_codes(c["dbbits"]), dw) and np.array_equal(metric.pack_labels(c["dblab"]), dl)
    assert np.array_equal(metric.pack_codes(c["qbits"]), qw) and np.array_equal(metric.pack_labels(c["qlab"]), ql)
