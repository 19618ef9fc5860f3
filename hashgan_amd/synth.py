"""Seeded synthetic inputs for the retrieval-evaluation path.

Everything here is integer arithmetic on a splitmix64 stream, so the same
(seed, shape) gives the same arrays in this container, on the GPU box and in
the golden-fixture generator (tests/golden/make_golden.py) -- only seeds and
expected outputs have to be committed, never the inputs.

Shapes follow SURVEY.md section 8 / BASELINE.json `configs`:
codes are {0,1}^b bit matrices (uint8 [n, b]); labels are {0,1}^C (int8 [n, C]),
one-hot (CIFAR-10 like, /root/reference/data_list/cifar10/*.txt) or multi-hot
(NUS-WIDE-81 like, /root/reference/data_list/nuswide_81/*.txt).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """n uint64 draws of the splitmix64 generator started at `seed`, skipping the first `offset` (so a shard can
    draw exactly its own rows of a global array)."""
    with np.errstate(over="ignore"):
        i = np.arange(1 + int(offset), int(offset) + int(n) + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform01(seed, n):
    """n doubles in [0, 1) with 53 random bits each."""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def random_code_words(seed, n, b, row_offset=0):
    """random_bits(seed, ...)[row_offset : row_offset + n] already packed: uint64 [n, ceil(b/64)], bit j of a code =
    bit j % 64 of word j // 64, pad bits zero (hashgan_amd.metric.pack_codes of the bit matrix, without the detour)."""
    words = (b + 63) // 64
    raw = splitmix64(seed, n * words, offset=int(row_offset) * words).reshape(n, words)
    if b % 64:
        raw[:, -1] &= np.uint64((1 << (b % 64)) - 1)
    return raw


def onehot_label_words(seed, n, C, row_offset=0):
    """onehot_labels(seed, ...)[row_offset : row_offset + n] packed like hashgan_amd.metric.pack_labels."""
    cls = (splitmix64(seed, n, offset=row_offset) % np.uint64(C)).astype(np.int64)
    out = np.zeros((n, (C + 63) // 64), dtype=np.uint64)
    out[np.arange(n), cls // 64] = np.uint64(1) << (cls % 64).astype(np.uint64)
    return out


def random_bits(seed, n, b):
    """i.i.d. Bernoulli(1/2) bit matrix uint8 [n, b]."""
    words = (b + 63) // 64
    raw = splitmix64(seed, n * words).reshape(n, words)
    bits = np.unpackbits(raw.view(np.uint8).reshape(n, words * 8), axis=1, bitorder="little")
    return np.ascontiguousarray(bits[:, :b])


def onehot_labels(seed, n, C):
    """Uniform one-hot labels int8 [n, C] and the class index vector."""
    cls = (splitmix64(seed, n) % np.uint64(C)).astype(np.int64)
    lab = np.zeros((n, C), dtype=np.int8)
    lab[np.arange(n), cls] = 1
    return lab, cls


def multihot_labels(seed, n, C, mean_pos=2.43):
    """Multi-hot labels int8 [n, C], >= 1 positive per row.

    Class frequencies fall off geometrically (a few frequent classes, a long
    tail) and are scaled so a row has `mean_pos` positives on average -- the
    statistics SURVEY.md section 2 row 3 reports for nuswide_81/train.txt.
    """
    freq = 0.85 ** np.arange(C)
    freq = freq * (mean_pos / freq.sum())
    freq = np.minimum(freq, 0.9)
    u = _uniform01(seed, n * C).reshape(n, C)
    lab = (u < freq[None, :]).astype(np.int8)
    empty = lab.sum(1) == 0
    if empty.any():
        forced = (splitmix64(seed ^ 0x5DEECE66D, n) % np.uint64(C)).astype(np.int64)
        rows = np.nonzero(empty)[0]
        lab[rows, forced[rows]] = 1
    return lab


def planted_codes(seed, label, b, flip_p, noise_seed=None):
    """Codes correlated with labels: class prototype XOR Bernoulli(flip_p) noise.

    A row's prototype is the XOR of the prototypes of its positive classes, so
    items sharing labels are close in Hamming distance and mAP is not chance.
    """
    n, C = label.shape
    proto = random_bits(seed ^ 0xA5A5A5A5, C, b)            # [C, b]
    base = (label.astype(np.int64) @ proto.astype(np.int64)) & 1
    noise = _uniform01(seed if noise_seed is None else noise_seed, n * b).reshape(n, b) < flip_p
    return (base.astype(np.uint8) ^ noise.astype(np.uint8)).astype(np.uint8)
