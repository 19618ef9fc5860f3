#!/usr/bin/env python3
"""Benchmark of the retrieval-evaluation hot path (BASELINE.json metric: queries/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5|c1|c4]

One step = one full pass of the path over the query batch with codes and labels already resident in HBM:
sampled histogram -> threshold guess -> select -> verify + order + label match -> AP (all HIP) -> per-query AP to
the host -> mean.  `value` = Q / step time: queries per second, whole job.

  --gpus 1 (default)   C2 = BASELINE.json configs[1]: Q=10k, N=1M, b=64, R=5000 on one MI355X -- the configuration
                       the metric is quoted on.
  --gpus G > 1         C4 = configs[3]: the FIXED N=10M database sharded over the G GPUs of one node (contiguous
                       index ranges, shard_bounds(10M, G)), Q=10k queries replicated; strong scaling.  One process per
                       GPU: either launched as `python -m torch.distributed.run --nproc-per-node G ... bench.py --gpus G`
                       (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment), or -- with no
                       WORLD_SIZE in the environment -- `python bench.py --gpus G` spawns the G rank processes itself and
                       prints rank 0's line.  The exchanges are native RCCL all-gathers through the library's C ABI
                       (hashgan_amd/sharded.py) -- no torch in the process.
                       HG_BENCH_FORCE_SHARDED=1 runs this leg with one rank (`--workload c2` keeps its database small).
  --workload c4 --gpus 1   C4 on a single GPU: the reference point for the scaling curve.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer VALU peak: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz.  Measured on this part
# (profiles/r01_ubench2_valu_rates.txt): v_xor_b32, v_bcnt_u32_b32, v_add_u32 and v_cmp all issue at
# 4.1-4.2 cycles per wave64 instruction per SIMD (37-38 T lane-ops/s sustained); the 78.6 T figure of
# MI355X_MICROARCH.md (32 lanes/clk) is the packed-FP32 rate and is not reachable by these ops.
VALU_PEAK_TLANEOPS = 39.3
VALU_NOMINAL_FP32_TLANEOPS = 78.6
HBM_PEAK_GBS = 8000.0         # HBM3E spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
SETTLE_STEPS = 25             # untimed steps before the timed region, at least (see main)
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MX-fp4 MFMA peak (MI355X_MICROARCH.md: ~10 PF dense; 9.1 PF measured)
METRIC = "queries/sec (mAP@R of Q queries vs N-code database, Hamming ranking)"
DTYPE = "fp4 (E2M1 0/+-1) x fp4 -> f32 exact distances; u32 xor+popcount elsewhere; f64 AP"

WORKLOADS = {
    # name: the case in tests/cases.py whose seeds/shape we reuse, at full Q; `golden` = its committed reference outputs
    "c2": dict(Q=10000, N=1000000, b=64, R=5000, C=10, kind="planted", seed=0xC2, flip=0.30, golden="c2_q64"),
    "c5": dict(Q=10000, N=1000000, b=128, R=5000, C=10, kind="planted", seed=0xC5, flip=0.35, golden="c5_b128_q32"),
    "c3": dict(Q=2100, N=190000, b=48, R=5000, C=81, kind="multihot", seed=0xC3, flip=0.20, golden="c3_nus_q64"),
    "c1": dict(Q=1000, N=54000, b=32, R=54000, C=10, kind="cifar", seed=0xC1, flip=0.25, golden="c1_cifar_full"),
    "c4": dict(Q=10000, N=10000000, b=64, R=5000, C=10, kind="iid", seed=0xC4, golden="c4_n10m_q8"),
    # BASELINE.json configs[1] as literally written ("synthetic random codes"): the timed shape on i.i.d. Bernoulli(1/2) bits
    # (SURVEY.md 8d); the reference's golden covers its first 64 queries.
    "c2_iid": dict(Q=10000, N=1000000, b=64, R=5000, C=10, kind="iid", seed=0x2C2, golden="c2_iid_q64"),
}


def build_inputs(spec):
    from tests import cases
    s = {k: v for k, v in spec.items() if k != "golden"}
    cases.CASES["_bench"] = s
    try:
        return cases.build_case("_bench")
    finally:
        del cases.CASES["_bench"]


def build_packed(spec, base, rows):
    """Packed query tables and rows [base, base + rows) of the packed database tables.  The iid workload (C4) draws
    exactly its own rows from the seeded stream (hashgan_amd.synth, same arrays as tests/cases.py `iid`)."""
    from hashgan_amd import metric, synth
    if spec["kind"] == "iid":
        seed, b, C = spec["seed"], spec["b"], spec["C"]
        qw = synth.random_code_words(seed + 7, spec["Q"], b)
        ql = synth.onehot_label_words(seed * 3 + 2, spec["Q"], C)
        dw = synth.random_code_words(seed, rows, b, row_offset=base)
        dl = synth.onehot_label_words(seed * 3 + 1, rows, C, row_offset=base)
        return qw, ql, dw, dl
    c = build_inputs(spec)
    qw, ql = metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"])
    sl = slice(base, base + rows)
    return qw, ql, metric.pack_codes(c["dbbits"][sl]), metric.pack_labels(c["dblab"][sl])


def unpack_bits(words, b):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8).reshape(words.shape[0], -1), axis=1, bitorder="little")[:, :b]


def cpu_baseline(spec, packed, budget_s=12.0, chunk=128):
    """metric.py:12-24 as written (float32 np.dot -> np.argsort(-ips,1) -> Python
    loop; oracle.reference_as_written) on the GPU box's host cores, over query
    chunks (rows are independent; a chunk bounds the reference's 16 B/pair) until
    ~budget_s seconds of CPU work are done."""
    from oracle import hamming_map as O
    qw, ql, dw, dl = packed
    b, C, R = spec["b"], spec["C"], spec["R"]
    Q = qw.shape[0]
    dbf = unpack_bits(dw, b).astype(np.float32) * 2 - 1
    dlab = unpack_bits(dl, C).astype(np.int64)
    qbits, qlab = unpack_bits(qw, b), unpack_bits(ql, C).astype(np.int64)
    chunk = max(1, min(chunk, int(2e9 // (16 * dbf.shape[0])) or 1))
    done, t0 = 0, time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while done < Q and time.perf_counter() - t0 < budget_s:
            sl = slice(done, min(Q, done + chunk))
            O.reference_as_written(dbf, dlab, qbits[sl].astype(np.float32) * 2 - 1, qlab[sl], R)
            done = sl.stop
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        blas = [(p.get("internal_api"), p.get("num_threads")) for p in threadpoolctl.threadpool_info()]
    except Exception:      # noqa: BLE001
        blas = None
    quota = None
    try:                                             # the box may cap the process far below its core count (cgroup v2 cpu.max: "<quota> <period>")
        qs, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if qs == "max" else float(qs) / float(per)
    except (OSError, ValueError):
        pass
    used = max([n for _, n in (blas or []) if isinstance(n, int)] or [os.cpu_count() or 1])      # threads the run really used: np.dot's BLAS pool (argsort and the loop: one)
    return {"value": done / dt, "unit": "queries/s", "cores": used, "host_hardware_threads": os.cpu_count(), "cgroup_cpu_quota": quota, "kind": "port",
            "sample": "%d of %d queries (chunks of %d) x full N=%d database in %.1f s; float32 +-1 features; np.dot on "
                      "BLAS threads %s, np.argsort and the per-query loop on 1 core; numpy %s"
                      % (done, Q, chunk, dbf.shape[0], dt, blas, np.__version__)}


def kernel_sources_sha():
    """Fingerprint of the device code of the path this benchmark's headline step runs (the kernel headers; the .hip units,
    hg_ctx.hpp and hg_host_pack.hpp are host code, hg_real_*.hpp is the real-valued path): the committed PMC traffic figure
    is only quoted for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hashgan_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if not f.endswith(".hpp") or f in ("hg_host_pack.hpp", "hg_ctx.hpp") or f.startswith("hg_real_"):
            continue
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def traffic_from_profiles(kernel, workload=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass (profiles/latest_traffic.json, written by
    tools/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE passes of this same command) -- quoted only if that pass measured
    the device code this run executes (same source fingerprint); else None plus the reason.  workload: one of the other
    configurations' passes (`_workloads` in the file: C5's, collected with `--workload c5`); None: the headline step's."""
    path = os.path.join(ROOT, "profiles", "latest_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC pass committed"
    if workload:
        d = d.get("_workloads", {}).get(workload)
        if not d:
            return None, "no PMC pass of workload %s committed" % workload
    sha = d.get("_kernel_sources_sha")
    if sha != kernel_sources_sha():
        return None, "profiles/latest_traffic.json was measured on other kernel sources (%s); rerun tools/pmc_traffic.py" % sha
    # the timing slot "k_select_mx" covers the kernel variants k_select_mx / k_select_mx3 / k_select_mx4: take the one the pass saw most
    cands = [k for k in d if k.startswith(kernel) and isinstance(d[k], dict)]
    if not cands:
        return None, "the PMC pass holds no %s* kernel" % kernel
    best = max(cands, key=lambda k: d[k].get("launches_sampled", 0))
    return d[best].get("hbm_bytes_per_launch"), "rocprofv3 PMC pass of these sources (%s), kernel %s" % (d.get("_collected", "?"), best)


SELECT_VARIANTS = {1: "k_select", 2: "k_select_dense", 3: "k_select_mx", 5: "k_select_mx3", 6: "k_select_mx4"}
RANK_VARIANTS = {1: "k_rank_fused", 3: "k_rank_cnt", 6: "k_rank_lean", 7: "k_rank_dense", 8: "k_rank_dense<slices>"}


def kernel_names(ctx, spec):
    """The kernels behind the timing slots of the last step, by the names a rocprofv3 trace shows (the slots `k_select_mx`,
    `k_rank_cnt`, `k_hist` each cover several kernels; the library reports which one it launched)."""
    NW, LW = (spec["b"] + 31) // 32, (spec["C"] + 63) // 64
    sel = SELECT_VARIANTS.get(ctx.get_stat("select_variant"))
    rank = RANK_VARIANTS.get(ctx.get_stat("rank_variant"))
    names = {}
    if sel == "k_select_mx3":
        names["k_select_mx"] = "k_select_mx3<%d,%d>" % (2 if NW == 2 else 1, LW)
    elif sel == "k_select_mx4":
        names["k_select_mx"] = "k_select_mx4<%d,%d>" % (NW, LW)
    elif sel == "k_select_mx":
        names["k_select_mx"] = "k_select_mx<%d,%d,%d,compact>" % (NW, LW if LW <= 2 else 0, 2 if NW <= 4 else 1)
    elif sel:
        names["k_select"] = "%s<%d,%d>" % (sel, NW, LW if LW <= 2 else 0)
    if rank == "k_rank_dense":
        names["k_select"] = "k_dense_bytes<%d,%d>" % (NW, LW)
    if rank:
        names["k_rank_cnt" if rank in ("k_rank_cnt", "k_rank_lean", "k_rank_dense<slices>") else "k_rank_fused"] = rank
    names["k_hist"] = "k_hist_i8<%d>" % NW if NW <= 4 else "k_hist_mx<%d>" % NW
    names["k_guess"] = "k_guess_direct"
    return names


def kernel_rooflines(timing, spec, rows, step_s, rec_bytes=1, names=None):
    """Per-kernel average launch time -> roofline of the dominant kernel.  `rows` = database rows this GPU holds;
    rec_bytes: bytes per record the select pass writes (1: compact {match, dist}, hg_map's default; 8: with index)."""
    Q, b, C, R = spec["Q"], spec["b"], spec["C"], spec["R"]
    NW, LW = (b + 31) // 32, (C + 63) // 64
    pairs = Q * rows
    code_bytes = (Q + rows) * NW * 4
    alg_bytes = {   # algorithmic (compulsory) HBM bytes per launch, DESIGN.md section 4
        "k_hist": code_bytes + (b + 1) * ((Q + 63) // 64 * 64) * 4,
        "k_select": code_bytes + (Q + rows) * LW * 8 + Q * R * 8,       # codes + labels in, >= R records of 8 B out per query
        # fp4 images of the codes (4 bits per code bit, 64-bit granules) + packed codes + labels in, records out
        "k_select_mx": (Q + rows) * ((NW + 1) // 2) * 32 + code_bytes + (Q + rows) * LW * 8 + Q * R * rec_bytes,
    }
    out = {name: {"avg_ms": ms / max(cnt, 1), "launches": cnt} for name, (ms, cnt) in timing.items()}
    pair_passes = {k: v for k, v in out.items() if k in alg_bytes}
    dom = max(pair_passes or out, key=lambda k: out[k]["avg_ms"] * out[k]["launches"])
    t = out[dom]["avg_ms"] * 1e-3
    ab = alg_bytes.get(dom, 0)
    hbm = {"algorithmic_bytes": ab, "achieved": ab / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / t / 1e9 / HBM_PEAK_GBS}
    traffic, traffic_note = traffic_from_profiles(dom)
    # SURVEY.md 8(d)'s own yardstick: the xor+popcount formulation's 4 W lane-ops per pair (W = ceil(b/64)) against the
    # guide's 78.6 T lane-op/s, for the dominant kernel and for the whole step
    W = (b + 63) // 64
    valu_equiv = {"definition": "pairs * 4 * ceil(b/64) lane-ops / (t * 78.6e12)  (SURVEY.md 8d)",
                  "kernel": pairs * 4 * W / t / (VALU_NOMINAL_FP32_TLANEOPS * 1e12),
                  "step": pairs * 4 * W / step_s / (VALU_NOMINAL_FP32_TLANEOPS * 1e12)}
    if dom == "k_select_mx":
        # matrix-core select: one v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) per 64 code bits and 32x32 tile of pairs
        K = 64 * ((NW + 1) // 2)
        flops = 2.0 * pairs * K
        roof = {"bound": "mfma", "kernel": (names or {}).get(dom, dom), "timing_slot": dom, "achieved": flops / t / 1e12, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / MFMA_FP4_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                "avg_launch_ms": out[dom]["avg_ms"], "algorithmic_flops": flops,
                "note": "fp4 MFMA inner product over K=%d code bits per pair (2 flops per bit), against the dense fp4 peak; the "
                        "kernel's other per-pair cost is the harvest of the accumulator signs on the vector ALU, which on gfx950 "
                        "overlaps the MFMAs only in part (about 4 vector instructions hide under one MFMA: tools/mfma_valu_overlap.hip, HISTORY.md): see 'valu'" % K,
                "valu": {"algorithmic_laneops": pairs, "achieved": pairs / t / 1e12, "peak": VALU_PEAK_TLANEOPS,
                         "unit": "Tlaneop/s", "frac": pairs / t / 1e12 / VALU_PEAK_TLANEOPS},
                "valu_equiv_frac": valu_equiv, "hbm": hbm}
        return roof, out
    laneops = pairs * 2 * NW                      # one v_xor_b32 + one v_bcnt_u32_b32 per 32-bit word per pair
    roof = {"bound": "valu", "kernel": (names or {}).get(dom, dom), "timing_slot": dom, "achieved": laneops / t / 1e12, "peak": VALU_PEAK_TLANEOPS,
            "unit": "Tlaneop/s", "frac": laneops / t / 1e12 / VALU_PEAK_TLANEOPS, "traffic": traffic, "traffic_source": traffic_note,
            "avg_launch_ms": out[dom]["avg_ms"], "algorithmic_laneops": laneops,
            "note": "integer bit-count path: xor+popcount lane-ops (2 per 32-bit code word per pair) against the "
                    "integer VALU issue peak (16 lanes/clk/SIMD)",
            "valu_equiv_frac": valu_equiv, "hbm": hbm}
    return roof, out


def h2d_inclusive(spec, packed, reps=10):
    """The drop-in call itself, from HOST arrays as forward_all() returns them (float32 +-1 features, int64 labels):
    MAPs(R).get_maps_by_feature(database, query) packs them with a pool of host threads (hg_host_pack.hpp), uploads
    the packed tables over PCIe, then runs the step.  Never `value`."""
    import types
    from hashgan_amd import MAPs
    qw, ql, dw, dl = packed
    b, C, R = spec["b"], spec["C"], spec["R"]
    db = types.SimpleNamespace(output=unpack_bits(dw, b).astype(np.float32) * 2 - 1, label=unpack_bits(dl, C).astype(np.int64))
    q = types.SimpleNamespace(output=unpack_bits(qw, b).astype(np.float32) * 2 - 1, label=unpack_bits(ql, C).astype(np.int64))
    m = MAPs(R)
    try:
        for _ in range(3):                              # first calls: allocations, the packing threads, host clocks
            val = m.get_maps_by_feature(db, q)
        each = []
        for _ in range(reps):
            t0 = time.perf_counter()
            val = m.get_maps_by_feature(db, q)
            each.append(time.perf_counter() - t0)
        full = sum(each) / reps
        m.set_database(db)                              # main.py:237-240 re-evaluates one database: keep it resident
        t0 = time.perf_counter()
        for _ in range(reps):
            val2 = m.get_maps_by_feature(None, q)
        resident = (time.perf_counter() - t0) / reps
    finally:
        m.close()
    Q = qw.shape[0]
    host_bytes = db.output.nbytes + db.label.nbytes + q.output.nbytes + q.label.nbytes
    return {"call": "MAPs(R).get_maps_by_feature(database, query) from host float32 features + int64 labels",
            "calls_timed": reps, "ms_per_call": full * 1e3, "median_ms_per_call": float(np.median(each)) * 1e3, "min_ms_per_call": min(each) * 1e3,
            "ms_each_call": [round(x * 1e3, 3) for x in each],
            "queries_per_sec": Q / full, "host_array_bytes": host_bytes,
            "bytes_over_pcie": int(dw.nbytes + dl.nbytes + qw.nbytes + ql.nbytes),
            "with_resident_database": {"call": "MAPs.set_database(database) once, then get_maps_by_feature(None, query)",
                                       "ms_per_call": resident * 1e3, "queries_per_sec": Q / resident},
            "map_equal_to_resident_path": bool(val == val2)}, float(val)


def drop_in_literal(spec, packed, reps=12):
    """main.py:164 LITERALLY: `MAPs(cfg.DATA.MAP_R).get_maps_by_feature(db, test)` -- a NEW MAPs object per evaluation, never closed,
    host arrays as forward_all() returns them.  `reps` calls per shape, each on a fresh object, NONE discarded (the first one pays
    whatever a process pays once: the pooled context's creation, its allocations, the packing threads).  Shapes: the timed
    workload on +-1 codes; the reference's own CIFAR-10 evaluation on tanh features (config/cifar_evaluation.yaml:9-12: TEST_SIZE
    1000, DB_SIZE 54000, MAP_R 54000, HASH_DIM 64 of lib/config.py:10); its NUS-WIDE setting (config/nuswide_step_1.yaml:10-13:
    5000 x 168692, MAP_R 5000, 81 labels) on tanh features.  Never `value`."""
    import gc
    import types
    from hashgan_amd import MAPs, pool_stats, _native
    qw, ql, dw, dl = packed
    b, C, R = spec["b"], spec["C"], spec["R"]
    rng = np.random.default_rng(0xD1)

    def tanh_case(Q, N, bb, CC, multi):
        if multi:
            lab = (rng.random((N, CC)) < 0.03).astype(np.int64)
            lab[np.arange(N), rng.integers(0, CC, N)] = 1
            qlab = (rng.random((Q, CC)) < 0.03).astype(np.int64)
            qlab[np.arange(Q), rng.integers(0, CC, Q)] = 1
        else:
            eye = np.eye(CC, dtype=np.int64)
            lab, qlab = eye[rng.integers(0, CC, N)], eye[rng.integers(0, CC, Q)]
        return (types.SimpleNamespace(output=np.tanh(rng.standard_normal((N, bb), dtype=np.float32)), label=lab),
                types.SimpleNamespace(output=np.tanh(rng.standard_normal((Q, bb), dtype=np.float32)), label=qlab))
    shapes = {
        "c2_pm1_codes": (R, types.SimpleNamespace(output=unpack_bits(dw, b).astype(np.float32) * 2 - 1, label=unpack_bits(dl, C).astype(np.int64)),
                         types.SimpleNamespace(output=unpack_bits(qw, b).astype(np.float32) * 2 - 1, label=unpack_bits(ql, C).astype(np.int64))),
        "cifar10_tanh": (54000,) + tanh_case(1000, 54000, 64, 10, False),
        "nuswide81_tanh": (5000,) + tanh_case(5000, 168692, 64, 81, True),
    }
    out = {"call": "MAPs(R).get_maps_by_feature(database, query): a NEW MAPs object per call, never closed (main.py:164), host float32 / int64 arrays; "
                   "%d calls per shape, none discarded" % reps}
    probe = _native.Context(0)
    try:
        for name, (r_, db, q) in shapes.items():
            gc.collect()
            p0, h0 = pool_stats(), _native.host_phase_timers(probe)
            each, vals = [], []
            for _ in range(reps):
                t0 = time.perf_counter()
                v = MAPs(r_).get_maps_by_feature(db, q)      # the object is collected with the statement: its context goes back to the pool
                each.append(time.perf_counter() - t0)
                vals.append(v)
            p1, h1 = pool_stats(), _native.host_phase_timers(probe)
            ms = [round(x * 1e3, 3) for x in each]
            out[name] = {"shape": "Q=%d N=%d b=%d R=%d C=%d" % (q.output.shape[0], db.output.shape[0], db.output.shape[1], r_, db.label.shape[1]),
                         "ms_each_call": ms, "min_ms": min(ms), "median_ms": float(np.median(ms)), "max_ms": max(ms),
                         "max_over_median": round(max(ms) / float(np.median(ms)), 3),
                         "max_over_median_after_first": round(max(ms[1:]) / float(np.median(ms)), 3),
                         "same_result_every_call": bool(all(v == vals[0] for v in vals)), "map": float(vals[0]),
                         "contexts_created": p1["contexts_created"] - p0["contexts_created"], "contexts_recycled": p1["contexts_recycled"] - p0["contexts_recycled"],
                         "host_phase_ms": {k: round(h1[k][0] - h0[k][0], 3) for k in h0 if h1[k][0] - h0[k][0] > 0.0005},
                         "host_phase_calls": {k: h1[k][1] - h0[k][1] for k in h0 if h1[k][1] - h0[k][1] > 0},
                         "host_array_bytes": int(db.output.nbytes + db.label.nbytes + q.output.nbytes + q.label.nbytes)}
        out["pool"] = pool_stats()
    finally:
        probe.close()
    return out


def real_valued(spec, reps=5):
    """The caller's REAL input: main.py:157,164 hands tanh outputs (lib/architecture.py:147,192,384-389), not codes, to the
    metric, which ranks them by float32 inner product (metric.py:13-14 as written).  The same call on such features at
    this workload's shape: from host arrays, and with the database kept resident.  Never `value`."""
    import types
    from hashgan_amd import MAPs
    Q, N, b, C, R = spec["Q"], spec["N"], spec["b"], spec["C"], spec["R"]
    rng = np.random.default_rng(spec["seed"])
    eye = np.eye(C, dtype=np.int64)
    db = types.SimpleNamespace(output=np.tanh(rng.standard_normal((N, b), dtype=np.float32)), label=eye[rng.integers(0, C, N)])
    q = types.SimpleNamespace(output=np.tanh(rng.standard_normal((Q, b), dtype=np.float32)), label=eye[rng.integers(0, C, Q)])
    m = MAPs(R)
    try:
        for _ in range(2):
            val = m.get_maps_by_feature(db, q)
        t0 = time.perf_counter()
        for _ in range(reps):
            val = m.get_maps_by_feature(db, q)
        full = (time.perf_counter() - t0) / reps
        m.set_database(db)
        m.get_maps_by_feature(None, q)
        t0 = time.perf_counter()
        for _ in range(reps):
            val2 = m.get_maps_by_feature(None, q)
        resident = (time.perf_counter() - t0) / reps
        # per-kernel times of the resident call (HIP events around every kernel: a pass of its own) -> the call's roofline
        roof = None
        try:
            ctx = m._eng.ctx
            ctx.set_option("timing_every", 1)
            ctx.timing_enable(2)
            m.get_maps_by_feature(None, q)
            ctx.timing_reset()
            for _ in range(3):
                m.get_maps_by_feature(None, q)
            timing = ctx.timing_read()
            ctx.timing_enable(False)
            timing.pop("step_gpu_span", None)
            real_names = {"k_real_select": "k_real_select_bf (bf16 MFMA filter)", "k_radix_pass": "k_real_rank_lds (select R-th key + order, in LDS)",
                          "k_real_sample": "k_real_sample_h + k_real_sample_count (16-bit sample scores; the second, counting sample)", "k_real_guess": "k_real_guess_lds + k_real_guess2 + k_real_thr2"}
            kern = {real_names.get(k_, k_): round(ms / max(cnt, 1) * (cnt / 3.0), 5) for k_, (ms, cnt) in timing.items()}    # ms per call
            flops = 2.0 * Q * N * b
            filt = timing.get("k_real_select", (0.0, 1))
            t_f = filt[0] / max(filt[1], 1) * 1e-3
            dom = max(kern, key=kern.get)
            roof = {"bound": "mfma (bf16)", "algorithmic_flops": flops, "definition": "2 * Q * N * b flop of the inner products (metric.py:13) / time / 2.5 PF dense bf16",
                    "call": {"achieved": flops / resident / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / resident / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                    "filter_kernel": {"kernel": "k_real_select_bf", "avg_launch_ms": t_f * 1e3, "achieved": flops / t_f / 1e12 if t_f else None,
                                      "frac": flops / t_f / 1e12 / MFMA_BF16_PEAK_TFLOPS if t_f else None},
                    "dominant_kernel": dom, "ms_per_call_by_kernel": kern,
                    "rescore_bytes": "k_real_rescore gathers 4 * %d B of float32 row per kept (query, row) pair out of the L2; the filter keeps ~1.3-2 R rows per query" % (((b + 15) // 16) * 16)}
        except Exception as e:      # noqa: BLE001
            roof = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        m.close()
    return {"call": "MAPs(R).get_maps_by_feature(database, query) on tanh(N(0,1)) float32 features, one-hot labels, Q=%d N=%d b=%d R=%d" % (Q, N, b, R),
            "roofline": roof,
            "ranking": "float32 inner product (bf16 MFMA filter with a rigorous margin + exact float32 fma-chain rescoring), ties by index",
            "from_host": {"ms_per_call": full * 1e3, "queries_per_sec": Q / full,
                          "host_array_bytes": int(db.output.nbytes + db.label.nbytes + q.output.nbytes + q.label.nbytes)},
            "with_resident_database": {"ms_per_call": resident * 1e3, "queries_per_sec": Q / resident},
            "map": float(val), "map_equal_to_resident_path": bool(val == val2), "calls_timed": reps}


def c4_reference(opts, steps=5, warmup=2):
    """The one-GPU point of the scaling curve: C4 (the fixed N = 10M database, configs[3]) unsharded on this GPU -- what
    `bench.py --gpus G` (G > 1) strong-scales from.  Reported beside the C2 line so that a 1/2/4/8 series has its origin."""
    from hashgan_amd import _native, metric
    spec = WORKLOADS["c4"]
    qw, ql, dw, dl = build_packed(spec, 0, spec["N"])
    ctx = _native.Context(0)
    try:
        for k_, v_ in opts:
            ctx.set_option(k_, v_)
        ctx.set_database(dw, dl, spec["b"], spec["C"])
        ctx.set_queries(qw, ql)
        for _ in range(warmup):
            a, r = ctx.map(spec["R"])
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            a, r = ctx.map(spec["R"])
            m = metric.mean_over_hits(a, r)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        from tests import cases
        g = cases.load_golden(spec["golden"])
        k = g["ap"].shape[0]
        return {"workload": "C4: Q=%d N=%d b=%d R=%d, one GPU, unsharded" % (spec["Q"], spec["N"], spec["b"], spec["R"]),
                "n_gpus": 1, "steps": steps, "ms_per_step": dt * 1e3, "value": spec["Q"] / dt, "unit": "queries/s",
                "pairs_per_sec": spec["Q"] * spec["N"] / dt, "map": float(m),
                "parity_vs_reference_golden": bool(np.array_equal(a[:k], g["ap"], equal_nan=True)),
                "optimistic_fallbacks": ctx.get_stat("optimistic_fallbacks")}
    finally:
        ctx.close()


L2_PEAK_GBS = 34500.0         # aggregate L2 bandwidth, MI355X_MICROARCH.md section L2


def config_leg(name, opts, steps=20, untimed=25, packed=None):
    """One of the other BASELINE.json configurations (C1 = configs[0], C3 = configs[2], C5 = configs[4]) on this GPU, timed
    like the headline step (inputs resident, K steps between synchronisations, untimed steps first), its per-kernel
    breakdown from HIP events in a second short pass, parity against the reference's golden on the first queries, and the
    roofline of ITS dominant kernel with the bound that applies to it.  Never `value`."""
    from hashgan_amd import _native, metric
    from tests import cases
    spec = WORKLOADS[name]
    Q, N, b, R, C = spec["Q"], spec["N"], spec["b"], spec["R"], spec["C"]
    NW, LW = (b + 31) // 32, (C + 63) // 64
    qw, ql, dw, dl = packed or build_packed(spec, 0, N)
    ctx = _native.Context(0)
    try:
        for k_, v_ in opts:
            ctx.set_option(k_, v_)
        ctx.set_database(dw, dl, b, C)
        ctx.set_queries(qw, ql)
        for _ in range(untimed):
            a, r = ctx.map(R)
        ctx.synchronize()
        each = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            a, r = ctx.map(R)
            m = metric.mean_over_hits(a, r)
            each.append(time.perf_counter() - t1)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kept = ctx.get_stat("records_kept")                # (a download of the slice counts, outside the timed region)
        ctx.set_option("timing_every", 1)
        ctx.timing_enable(2)                               # every kernel between HIP events: a pass of its own (the events cost time)
        for _ in range(4):
            ctx.map(R)
        ctx.timing_reset()
        for _ in range(6):
            ctx.map(R)
        timing = ctx.timing_read()
        ctx.timing_enable(False)
        span = timing.pop("step_gpu_span", None)
        names = kernel_names(ctx, spec)
        kern = {names.get(k_, k_): {"avg_ms": round(ms / max(cnt, 1), 5), "launches_per_step": cnt / 6.0} for k_, (ms, cnt) in timing.items()}
        slot = max(timing, key=lambda k_: timing[k_][0])
        dom = names.get(slot, slot)
        t = timing[slot][0] / max(timing[slot][1], 1) * 1e-3
        pairs = Q * N
        W = (b + 63) // 64
        if slot == "k_select_mx":
            flops = 2.0 * pairs * 64 * W
            roof = {"bound": "mfma", "kernel": dom, "avg_launch_ms": t * 1e3, "algorithmic_flops": flops, "achieved": flops / t / 1e12,
                    "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / t / 1e12 / MFMA_FP4_PEAK_TFLOPS,
                    "note": "fp4 MFMA inner product, 2 flops per code bit (K = %d) and pair; the kernel's other per-pair cost is the "
                            "vector-ALU harvest of the accumulators (hides under the MFMAs only in part on gfx950)" % (64 * W)}
        elif dom == "k_rank_dense":
            # N/8 < R <= N: k_dense_bytes wrote one byte {match, dist} per pair; one block per query streams its N bytes twice
            # (count, then place) and writes R match bits -- each row costs two LDS atomics on its thread's counter column
            ab = 2 * Q * N + Q * ((R + 63) // 64) * 8
            roof = {"bound": "hbm", "kernel": dom, "avg_launch_ms": t * 1e3, "algorithmic_bytes": ab, "achieved": ab / t / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / t / 1e9 / HBM_PEAK_GBS,
                    "note": "two streams over the query's row of the byte matrix (2 * Q * N bytes; at this size it is cache "
                            "resident) + R match bits; what bounds the kernel is the latency of its chain "
                            "of LDS atomics (two per row on the thread's own counter column) at one to four wavefronts per SIMD, not this stream"}
        else:
            ab = (Q + N) * (NW * 4 + LW * 8) + Q * R
            roof = {"bound": "hbm", "kernel": dom, "avg_launch_ms": t * 1e3, "algorithmic_bytes": ab, "achieved": ab / t / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / t / 1e9 / HBM_PEAK_GBS,
                    "note": "latency chain of a short kernel: neither the HBM nor an ALU bound is near"}
        roof["valu_equiv_frac"] = {"definition": "pairs * 4 * ceil(b/64) lane-ops / (t * 78.6e12)  (SURVEY.md 8d)",
                                   "kernel": pairs * 4 * W / t / (VALU_NOMINAL_FP32_TLANEOPS * 1e12),
                                   "step": pairs * 4 * W / dt / (VALU_NOMINAL_FP32_TLANEOPS * 1e12)}
        if name in ("c5",):                                # HBM bytes per launch of the leg's dominant kernel, from ITS committed PMC passes
            traffic, traffic_note = traffic_from_profiles(slot, name)
            roof["traffic"], roof["traffic_source"] = traffic, traffic_note
        g = cases.load_golden(spec["golden"])
        k = g["ap"].shape[0]
        out = {"workload": "%s: Q=%d N=%d b=%d R=%d C=%d, %s codes" % (name.upper(), Q, N, b, R, C, spec["kind"]),
               "steps": steps, "untimed_steps": untimed, "ms_per_step": dt * 1e3, "ms_per_step_min": min(each) * 1e3,
               "ms_per_step_median": float(np.median(each)) * 1e3, "queries_per_sec": Q / dt, "pairs_per_sec": pairs / dt,
               "map": float(m), "parity_vs_reference_golden": bool(np.array_equal(a[:k], g["ap"][:Q], equal_nan=True)), "golden_queries": int(min(k, Q)),
               "bet": bool(ctx.get_stat("last_optimistic")), "optimistic_fallbacks": ctx.get_stat("optimistic_fallbacks"),
               "dominant_kernel": dom, "roofline": roof, "kernels": kern}
        if kept >= 0:
            out["records_kept_over_R"] = round(kept / float(Q * R), 4)
        span_ms = span[0] / max(span[1], 1) if span else None
        if span_ms is not None and span_ms > dt * 1e3:
            # a step this short is a chain of launches: bracketing every kernel with HIP events costs more than the gaps it would
            # measure (the bracketed span exceeds the un-bracketed step), so no span is reported -- the per-kernel times above are
            # the bracketed pass's, the step time is the un-bracketed loop's; profiles/ holds the rocprofv3 trace of the same step
            out["gpu_span_ms_with_events"] = None
            out["event_bracketing"] = "span dropped: %.3f ms bracketed > %.3f ms un-bracketed step" % (span_ms, dt * 1e3)
        else:
            out["gpu_span_ms_with_events"] = round(span_ms, 5) if span_ms is not None else None
        return out
    finally:
        ctx.close()


def class_sorted(spec, packed, steps=10):
    """The same workload with the database STORED CLASS BY CLASS (rows stably sorted by label): a query's near rows then
    crowd into its class's share of the segments, the first bet's slices overflow, the engine widens them (cap_boost) and
    keeps the width -- the timed steps are the steady state after that.  Ties break by index, so the APs differ from the
    unsorted order's; parity here is the bet against the engine's own exact sequence on a sample of the queries."""
    from hashgan_amd import _native, metric
    qw, ql, dw, dl = packed
    key = unpack_bits(dl, spec["C"]).astype(np.int64) @ (1 << np.arange(spec["C"], dtype=np.int64))
    order = np.argsort(key, kind="stable")
    dws, dls = np.ascontiguousarray(dw[order]), np.ascontiguousarray(dl[order])
    R = spec["R"]
    firsts = []
    for _ in range(2):                     # the first call on a NEW context, twice (a box's one-off stall -- 123 ms once -- is not the library's)
        ctx = _native.Context(0)
        try:
            ctx.set_database(dws, dls, spec["b"], spec["C"])
            ctx.set_queries(qw, ql)
            t0 = time.perf_counter()
            ctx.map(R)
            firsts.append(time.perf_counter() - t0)
        finally:
            ctx.close()
    ctx = _native.Context(0)
    try:
        ctx.set_database(dws, dls, spec["b"], spec["C"])
        ctx.set_queries(qw, ql)
        t0 = time.perf_counter()
        a, r = ctx.map(R)
        firsts.append(time.perf_counter() - t0)
        first = min(firsts)
        ctx.map(R)
        t0 = time.perf_counter()
        for _ in range(steps):
            a, r = ctx.map(R)
        dt = (time.perf_counter() - t0) / steps
        out = {"workload": "the timed workload, database rows stably sorted by label", "ms_per_step": dt * 1e3, "first_call_ms": first * 1e3,
               "first_call_ms_each_new_context": [round(x * 1e3, 3) for x in firsts], "queries_per_sec": spec["Q"] / dt, "bet_held": bool(ctx.get_stat("last_optimistic")), "cap_boost": ctx.get_stat("cap_boost"),
               "lost_bets": ctx.get_stat("optimistic_rebets"), "exact_fallbacks": ctx.get_stat("optimistic_fallbacks"), "map": float(metric.mean_over_hits(a, r))}
        k = min(256, qw.shape[0])
        ctx.set_option("optimistic", 0)
        ctx.set_queries(qw[:k], ql[:k])
        a2, _ = ctx.map(R)
        out["equal_to_exact_sequence"] = {"queries": k, "equal": bool(np.array_equal(a[:k], a2, equal_nan=True))}
        return out
    finally:
        ctx.close()


def large_r(spec, packed, steps=3):
    """The timed workload's database and queries at LARGE R (not a BASELINE configuration; lib/metric.py:14,19 with MAP_R of the
    order of the database): R = N/20 -- the bet with long lists, ranked slice by slice (k_rank_dense<slices>) -- and R = N/2 --
    the byte matrix (k_dense_bytes + k_rank_dense).  Parity: against the engine's older sequences (k_rank_cnt's tiles; k_hist +
    k_select + k_rank_fused on 8-byte records) on a sample of the queries."""
    from hashgan_amd import _native
    qw, ql, dw, dl = packed
    N, Q = dw.shape[0], qw.shape[0]
    out = {}
    ctx = _native.Context(0)
    try:
        ctx.set_database(dw, dl, spec["b"], spec["C"])
        for name, R in (("r_n_over_20", N // 20), ("r_n_over_2", N // 2)):
            ctx.set_queries(qw, ql)
            ctx.map(R)
            ctx.map(R)
            t0 = time.perf_counter()
            for _ in range(steps):
                a, r = ctx.map(R)
            dt = (time.perf_counter() - t0) / steps
            leg = {"R": R, "ms_per_step": dt * 1e3, "queries_per_sec": Q / dt, "bet": bool(ctx.get_stat("last_optimistic")),
                   "rank_kernel": RANK_VARIANTS.get(ctx.get_stat("rank_variant"))}
            k = min(48, Q)
            ctx.set_option("rank_dense", 0)
            ctx.set_option("rank_slices", 0)
            ctx.set_queries(qw[:k], ql[:k])
            a2, _ = ctx.map(R)
            leg["equal_to_older_sequence"] = {"queries": k, "rank_kernel": RANK_VARIANTS.get(ctx.get_stat("rank_variant")),
                                              "equal": bool(np.array_equal(a[:k], a2, equal_nan=True))}
            ctx.set_option("rank_dense", 1)
            ctx.set_option("rank_slices", 7000)
            out[name] = leg
            ctx.trim()
        return out
    finally:
        ctx.close()


def query_split_leg(args, spec, rank, world, local_rank, dry_dir, comm, qw, ql):
    """The same workload decomposed the other way (hashgan_amd.sharded.evaluate_query_split): the WHOLE database on every
    GPU, the queries split, no data-path collective -- timed like the main leg (barrier, K steps, barrier, max over ranks)
    on a second context per rank.  Reported beside `value`, which stays the database-sharded form the north star names."""
    from hashgan_amd import _native, metric, sharded
    N, R, b, C = spec["N"], spec["R"], spec["b"], spec["C"]
    _, _, dw, dl = build_packed(spec, 0, N)
    ctx2 = _native.Context(0 if dry_dir else local_rank)
    try:
        for kv in args.opt:
            k_, v_ = kv.split("=")
            ctx2.set_option(k_, int(v_))
        ctx2.set_database(dw, dl, b, C)
        del dw, dl
        steps, warm = max(1, args.steps), max(1, min(args.warmup, 3))
        for _ in range(warm):
            a, r = sharded.evaluate_query_split(ctx2, comm, qw, ql, R)
        comm.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            a, r = sharded.evaluate_query_split(ctx2, comm, qw, ql, R)
            m = sharded.mean_ap(a, r)
        comm.barrier()
        dt = comm.allreduce_max(time.perf_counter() - t0) / steps
        from tests import cases
        g = cases.load_golden(spec["golden"])
        k = g["ap"].shape[0]
        # the curve's ORIGIN, measured in this very run: the same workload unsharded on one GPU (this rank's own -- every rank
        # holds the whole database for this leg; all ranks run it at once, rank 0 reports its own)
        ctx2.set_queries(qw, ql)
        for _ in range(2):
            a1, r1 = ctx2.map(R)
        ctx2.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            a1, r1 = ctx2.map(R)
        ctx2.synchronize()
        one_gpu = (time.perf_counter() - t0) / steps
        comm.barrier()
        return {"one_gpu_ms": one_gpu * 1e3, "one_gpu_parity": bool(np.array_equal(a1[:k], g["ap"], equal_nan=True)),
                "decomposition": "whole database (%d rows, %d MB packed) on each of %d GPUs; queries split %s; the only exchange "
                                 "is the all-gather of 16 bytes per query" % (N, N * 16 // 1000000, world,
                                                                            [n for _, n in sharded.shard_bounds(spec["Q"], world)]),
                "steps": steps, "ms_per_step": dt * 1e3, "value": spec["Q"] / dt, "unit": "queries/s", "map": float(m),
                "parity_vs_reference_golden": bool(np.array_equal(a[:k], g["ap"], equal_nan=True))}
    finally:
        ctx2.close()


def error_line(msg, n_gpus, steps=0, warmup=0):
    """The one JSON line of a run that could not measure anything."""
    return json.dumps({"metric": METRIC, "value": None, "unit": "queries/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
                       "ms_per_step": None, "higher_is_better": True, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                       "error": msg})


def self_launch(args, argv):
    """`python bench.py --gpus G` with no launcher around it: start the G rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment, the RCCL id file inside a private fresh directory), pass rank 0's
    stdout through, and if a rank dies or the launch times out print a JSON line with an `error` key instead of hanging."""
    import shutil
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    d = tempfile.mkdtemp(prefix="hashgan_amd_bench_")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    limit = float(os.environ.get("HG_BENCH_LAUNCH_TIMEOUT", "1800"))
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HG_COMM_ID_FILE=os.path.join(d, "rccl.id"))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        t0 = time.time()
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            time.sleep(0.05)
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = "rank %d exited with code %d" % (r, p.returncode)
            if failed is None and time.time() - t0 > limit:
                failed = "launch of %d ranks not finished after %.0f s" % (n, limit)
        for r, p in enumerate(procs):
            if failed is None and p.returncode not in (None, 0):
                failed = "rank %d exited with code %d" % (r, p.returncode)
        if failed:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                # the processes this function started, by their own handles
        out = procs[0].stdout.read().decode() if procs[0].stdout else ""
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if lines and not failed:
            print(lines[-1])
            return 0
        if lines and '"error"' in lines[-1]:
            print(lines[-1])
        else:
            print(error_line(failed or "rank 0 printed no result line", n, args.steps, args.warmup))
        return 1
    finally:
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:      # noqa: BLE001
                pass
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive MAPs(...) timing")
    ap.add_argument("--no-sorted", action="store_true", help="skip the class-sorted database timing")
    ap.add_argument("--no-literal", action="store_true", help="skip the literal drop-in leg (a new MAPs object per call, main.py:164)")
    ap.add_argument("--no-pipeline-extras", action="store_true",
                    help="skip the pipeline's side legs (a fresh query table per step; two contexts alternating): a profiled run "
                         "(rocprofv3) should hold the headline step's launches only")
    ap.add_argument("--no-large-r", action="store_true", help="skip the large-R legs (R = N/20, R = N/2 on the timed workload's arrays)")
    ap.add_argument("--no-real", action="store_true", help="skip the real-valued (tanh features) call timing")
    ap.add_argument("--no-c4-ref", action="store_true", help="skip the one-GPU C4 point of the scaling curve")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2-iid / C1 / C3 / C5 legs (the timed shape on i.i.d. codes; the other BASELINE.json configurations)")
    ap.add_argument("--no-query-split", action="store_true", help="sharded runs: skip the replicated-database / split-queries leg")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (hg_set_option), repeatable")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one GPU: every timed step is a synchronous hg_map (default: hg_map_begin / hg_map_end with two steps in flight -- "
                         "every step still delivers its verdict, APs and hit counts to the host inside the timed region)")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="pair-passes timing brackets the select pass on every n-th step of the timed region (its events cost a "
                         "step ~0.025 ms: tools/gpu_event_cost.sh); the average is over those launches")
    ap.add_argument("--kernel-timing", default="pair-passes", choices=["pair-passes", "all", "none"],
                    help="HIP events around the select pass over the pairs only (the roofline kernel; default), around every "
                         "kernel (events keep kernels from being dispatched back to back), or none")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))           # no launcher: be one
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if rank == 0:
            print(error_line("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world), args.gpus, args.steps, args.warmup))
        sys.exit(2)
    force = os.environ.get("HG_BENCH_FORCE_SHARDED") == "1"
    sharded_leg = world > 1 or force
    if world > 1:
        # a rank that waits for ever inside a collective (a peer died, a mismatched exchange) must not hold the launcher for ever:
        # after HG_BENCH_WATCHDOG seconds (default 900) rank 0 prints the line with an `error` key and every rank leaves
        import threading

        def _give_up():
            if rank == 0:
                print(error_line("no result after %s s (HG_BENCH_WATCHDOG): a collective of the sharded step did not return" %
                                 os.environ.get("HG_BENCH_WATCHDOG", "900"), world, args.steps, args.warmup), flush=True)
            os._exit(5)
        wd = threading.Timer(float(os.environ.get("HG_BENCH_WATCHDOG", "900")), _give_up)
        wd.daemon = True
        wd.start()
    wl = args.workload or ("c4" if world > 1 else "c2")
    spec = WORKLOADS[wl]

    from hashgan_amd import _native, metric, sharded
    Q, N, R, b, C = spec["Q"], spec["N"], spec["R"], spec["b"], spec["C"]
    base, rows = sharded.shard_bounds(N, world)[rank]
    packed = build_packed(spec, base, rows)
    qw, ql, dw, dl = packed

    # HG_BENCH_FILECOMM=<dir>: functional dry run of the N > 1 leg on a ONE-GPU box -- every rank a process on GPU 0, the
    # exchanges through files (tests/file_comm.py).  Never a measurement; the JSON line says so.
    dry_dir = os.environ.get("HG_BENCH_FILECOMM")
    ctx = _native.Context(0 if dry_dir else local_rank)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.set_database(dw, dl, b, C, idx_base=base, n_total=N)    # inputs resident in HBM before the timed region
    ctx.set_queries(qw, ql)

    if sharded_leg:
        if dry_dir:
            from tests.file_comm import FileComm
            comm = FileComm(dry_dir, rank, world, ctx)
            eng = sharded.HipShardEngine(ctx, want_lists=False)
        else:
            try:
                comm = sharded.init_rccl(ctx, rank, world, timeout=float(os.environ.get("HG_COMM_TIMEOUT", "120")))
            except (TimeoutError, RuntimeError) as e:
                if rank == 0:
                    print(error_line("RCCL rendezvous failed: %s" % e, world, args.steps, args.warmup))
                sys.exit(3)
            eng = sharded.HipShardEngine(ctx, want_lists=False, async_stages=True)   # one stream, no host waits between stages

        def step():
            a, r = sharded.evaluate_shard(eng, comm, R, always_gather=force)
            return sharded.mean_ap(a, r), a

        def fence():
            comm.barrier()                              # hipStreamSynchronize + a tiny all-reduce on every rank
    else:
        def step():
            a, r = ctx.map(R)
            return metric.mean_over_hits(a, r), a

        def fence():
            ctx.synchronize()                           # hipStreamSynchronize (every one-shot call also ends synchronised)
    pipelined = not sharded_leg and not args.no_pipeline

    # Untimed steps first: the W the caller asked for, and at least SETTLE_STEPS in all -- a GPU that has idled (the inputs
    # were generated on the host for seconds) needs ~20 ms of load before its clocks are back up (the first steps run 5-20 %
    # slow: tools/step_probe.py), and the first steps after kernel timing is switched on pay a one-off ~8 ms inside the HIP
    # runtime (event pools), so timing is on -- every step -- from the first untimed step.
    every = max(1, min(args.timing_every, args.steps)) if args.kernel_timing == "pair-passes" else 1
    level = {"pair-passes": 1, "all": 2, "none": 0}[args.kernel_timing]
    ctx.set_option("timing_every", 1)
    ctx.timing_enable(level)
    untimed = args.warmup if dry_dir else max(args.warmup, SETTLE_STEPS)      # (a dry run through files measures nothing)
    for _ in range(untimed):
        m, a = step()
    ctx.set_option("timing_every", every)
    ctx.timing_enable(level)                            # restarts the every-n-th count
    ctx.timing_reset()
    fence()
    each = []
    t0 = time.perf_counter()
    if pipelined:
        # hg_map in two halves, two steps in flight: step i + 1 is enqueued before step i's results are waited for, so the GPU goes
        # from one step straight into the next; every step's verdict is checked and its APs are on the host when its map_end returns
        ctx.map_begin(R)
        t1 = t0
        for i in range(args.steps):
            if i + 1 < args.steps:
                ctx.map_begin(R)
            a, r = ctx.map_end()
            t2 = time.perf_counter()
            each.append(t2 - t1)                        # (from one step's results to the next one's)
            t1 = t2
            m = metric.mean_over_hits(a, r)
    else:
        for _ in range(args.steps):
            t1 = time.perf_counter()
            m, a = step()                               # (ends synchronised: the AP vector is on the host)
            each.append(time.perf_counter() - t1)
    fence()
    dt = time.perf_counter() - t0
    if sharded_leg:
        dt = comm.allreduce_max(dt)                     # the slowest rank's clock
    timing = ctx.timing_read()
    ctx.timing_enable(False)
    pipe = None
    if pipelined:
        # the same steps one at a time (hg_map: enqueue, wait, copy out, return), for the latency of a single call
        ns = max(1, min(args.steps, 20))
        fence()
        ts = time.perf_counter()
        for _ in range(ns):
            step()
        fence()
        pipe = {"steps_in_flight": 2, "api": "hg_map_begin / hg_map_end", "sync_ms_per_step": (time.perf_counter() - ts) / ns * 1e3,
                "sync_steps_timed": ns, "blind_steps": ctx.get_stat("map_async_steps"), "blind_steps_redone": ctx.get_stat("map_async_redone")}
        try:
            if args.no_pipeline_extras:
                raise StopIteration
            # what a caller with a FRESH query batch per step gets (lib/metric.py once per batch): hg_set_queries -- which does not wait
            # for the stream -- then hg_map_begin, the previous batch's hg_map_end after it; two alternating query tables, every
            # step's APs checked against the table it was enqueued on
            qw2, ql2 = np.ascontiguousarray(qw[::-1]), np.ascontiguousarray(ql[::-1])
            tabs = [(qw, ql), (qw2, ql2)]
            ctx.set_queries(*tabs[1])
            ref1, _ = ctx.map(R)
            ctx.set_queries(*tabs[0])
            ref0, _ = ctx.map(R)
            refs = [ref0, ref1]
            n0 = ctx.get_stat("map_async_steps")
            fence()
            ts = time.perf_counter()
            ctx.set_queries(*tabs[0])
            ctx.map_begin(R)
            ok = True
            for i in range(ns):
                if i + 1 < ns:
                    ctx.set_queries(*tabs[(i + 1) & 1])
                    ctx.map_begin(R)
                a_, r_ = ctx.map_end()
                ok = ok and np.array_equal(a_, refs[i & 1], equal_nan=True)
            fence()
            pipe["fresh_queries_per_step"] = {"ms_per_step": (time.perf_counter() - ts) / ns * 1e3, "steps": ns, "blind_steps": ctx.get_stat("map_async_steps") - n0,
                                              "every_step_equals_its_own_batch": bool(ok),
                                              "note": "hg_set_queries (80 KB of packed codes + 80 KB of labels from host memory) + hg_map_begin per step, the previous step's hg_map_end after it"}
            ctx.set_queries(qw, ql)
            a, _ = ctx.map(R)
        except StopIteration:
            pass
        except Exception as e:      # noqa: BLE001 -- a side measurement must never cost the main line
            pipe["fresh_queries_per_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            if args.no_pipeline_extras:
                raise StopIteration
            # two contexts (a stream and work buffers each) on the same tables, steps enqueued alternately: one step's tail -- rank + AP,
            # the download -- and the next one's sampled histogram and guess run under the other stream's select.  What a caller may do
            # with the C ABI as it is; never `value`: per-kernel times no longer add up to the step
            ctx_b = _native.Context(0 if dry_dir else local_rank)
            try:
                for kv in args.opt:
                    k_, v_ = kv.split("=")
                    ctx_b.set_option(k_, int(v_))
                ctx_b.set_database(dw, dl, b, C)
                ctx_b.set_queries(qw, ql)
                for _ in range(5):
                    ab, _ = ctx_b.map(R)
                pair = [ctx, ctx_b]
                fence(); ctx_b.synchronize()
                ns2 = max(2, ns)
                ts = time.perf_counter()
                pair[0].map_begin(R)
                for i in range(ns2):
                    if i + 1 < ns2:
                        pair[(i + 1) & 1].map_begin(R)
                    a_, r_ = pair[i & 1].map_end()
                fence(); ctx_b.synchronize()
                pipe["two_contexts_alternating"] = {"ms_per_step": (time.perf_counter() - ts) / ns2 * 1e3, "steps": ns2,
                                                    "equal_to_one_context": bool(np.array_equal(ab, a, equal_nan=True) and np.array_equal(a_, a, equal_nan=True))}
            finally:
                ctx_b.close()
        except StopIteration:
            pass
        except Exception as e:      # noqa: BLE001
            pipe["two_contexts_alternating"] = {"error": "%s: %s" % (type(e).__name__, e)}
    exchange = None
    if sharded_leg:
        # what crosses the wire per step and rank (the owner-routed form: three all-to-alls and one small all-gather), and how long the
        # exchanges take on the stream (HIP events around every kernel and collective: a short pass of its own, after the timed region)
        NBp, HC, width, RWw = b + 1, min((b + 1) // 2 + 2, b + 1), -(-Q // world), -(-R // 64)
        cw = (NBp * width + 1) & ~1
        by = {"alltoall_sampled_histograms": world * (4 + HC * width) * 4, "alltoall_guess_answers": world * width * 16,
              "alltoall_record_counts_and_local_bitmaps": world * (cw + 64 + 2 * width * RWw) * 4, "allgather_ap_parts": world * (width + 1) * 16}
        exchange = {"form": "owner-routed (hg_shard_step: every stage and exchange one enqueue)" if getattr(comm, "routed_verified", None) or (world == 1 and not dry_dir)
                            else "all-gather of whole tables (the owner-routed form was not verified on this communicator, or no RCCL)",
                    "rccl_ranks": world, "routed_verified_against_allgather": getattr(comm, "routed_verified", None),
                    "ingress_bytes_per_rank_and_step": by, "ingress_bytes_total": int(sum(by.values()))}
        try:
            if dry_dir:
                raise RuntimeError("dry run through files: nothing to time")
            ctx.set_option("timing_every", 1)
            ctx.timing_enable(2)
            step()
            ctx.timing_reset()
            for _ in range(3):
                step()
            fence()
            tx = ctx.timing_read()
            ctx.timing_enable(False)
            if "rccl_allgather" in tx:
                exchange["exchanges_ms_per_step"] = round(tx["rccl_allgather"][0] / 3.0, 5)
                exchange["exchange_calls_per_step"] = tx["rccl_allgather"][1] / 3.0
            exchange["kernels_ms_per_launch_with_events"] = {k_: round(v_[0] / max(v_[1], 1), 5) for k_, v_ in tx.items()}
        except Exception as e:      # noqa: BLE001 -- a side measurement must never cost the main line
            exchange["timing_error"] = "%s: %s" % (type(e).__name__, e)
    qsplit = None
    if sharded_leg and not args.no_query_split:
        try:
            qsplit = query_split_leg(args, spec, rank, world, local_rank, dry_dir, comm, qw, ql)
        except Exception as e:      # noqa: BLE001 -- a side measurement must never cost the main line
            qsplit = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank != 0:
        if sharded_leg:
            comm.barrier()
            if not dry_dir:
                ctx.comm_destroy()
        return

    # parity flag: the first queries are a golden case of the unmodified reference
    from tests import cases
    g = cases.load_golden(spec["golden"])
    k = g["ap"].shape[0]
    parity = bool(np.array_equal(a[:k], g["ap"], equal_nan=True))

    per_step = dt / args.steps
    out = {
        "metric": METRIC, "value": Q / per_step, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True,
        "untimed_steps": untimed,
        "ms_per_step_min": min(each) * 1e3, "ms_per_step_median": float(np.median(each)) * 1e3, "ms_per_step_max": max(each) * 1e3,
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "%s: Q=%d N=%d b=%d R=%d C=%d, %s codes" % (wl.upper(), Q, N, b, R, C, spec["kind"]),
                   "parallelism": ("database sharded over %d GPU%s (%d rows on rank 0); native RCCL all-gather of shard "
                                   "histograms / record counts and match bitmaps" % (world, "s" if world > 1 else "", rows))
                   if sharded_leg else "1 GPU"},
        "map": float(m), "parity_vs_reference_golden": parity,
        "optimistic_runs": ctx.get_stat("optimistic_runs"), "optimistic_fallbacks": ctx.get_stat("optimistic_fallbacks"),
        "pairs_per_sec": Q * N / per_step,
    }
    kept = ctx.get_stat("records_kept")              # the last step's slice counts, downloaded now (outside the timed region)
    if kept >= 0:
        out["records_kept_over_R"] = round(kept / float(Q * R), 4)      # what the guess's safety margin costs the select's drain (1.0 = no surplus)
    if sharded_leg or wl == "c4":
        out["scaling"] = "strong"                            # the fixed N = 10M database over the GPUs (a one-GPU C2 line scales nothing)
    if pipe is not None:
        out["pipeline"] = pipe
    if exchange is not None:
        out["exchange"] = exchange
    if qsplit is not None:
        out["query_split"] = qsplit
        if "one_gpu_ms" in qsplit:
            # one record that explains itself: the SAME workload (C4 unless --workload says otherwise) on one GPU of this run, on the
            # database-sharded form (`value`) and on the query-split form -- nobody has to join a --gpus 1 line of another workload
            one = qsplit.pop("one_gpu_ms")
            sh_ms, qs_ms = per_step * 1e3, qsplit["ms_per_step"]
            out["strong_scaling"] = {
                "workload": out["config"]["workload"], "n_gpus": world,
                "one_gpu_ms": round(one, 5), "one_gpu_parity_vs_reference_golden": qsplit.pop("one_gpu_parity"),
                "database_sharded": {"ms_per_step": round(sh_ms, 5), "speedup": round(one / sh_ms, 4), "efficiency": round(one / sh_ms / world, 4)},
                "query_split": {"ms_per_step": round(qs_ms, 5), "speedup": round(one / qs_ms, 4), "efficiency": round(one / qs_ms / world, 4)},
                "recommended": "query_split for databases that fit one GPU (16 bytes per row: 160 MB at N = 10M) -- no data-path collective, modelled "
                               "~7.5x at 8 GPUs; database_sharded (the north star's form, `value`) when the database does not fit or must stay partitioned -- "
                               "modelled ~5x at 8 GPUs before the wire (DESIGN.md section 6)"}
    if sharded_leg:
        out["value_definition"] = "Q queries ranked against the WHOLE %d-row database per step / step time (max over ranks)" % N
        out["weak_scaling_equivalent"] = {"definition": "query x per-GPU-shard evaluations per second = value * n_gpus",
                                          "value": Q * world / per_step}
        out["scaling_note"] = ("strong scaling of ONE %d-row database over the GPUs (BASELINE config 4).  The --gpus 1 line times C2 (N = 1M, the "
                               "configuration the metric is quoted on), so value(--gpus 1) is NOT this curve's one-GPU point: that is "
                               "`scaling_reference_c4_one_gpu` in the --gpus 1 line (the same %d rows on one GPU, ~4.8 ms per step, ~2.1 M queries/s)" % (N, N))
    if timing:
        span = timing.pop("step_gpu_span", None)
        rec_bytes = 8 if any(kv.replace(" ", "") == "compact_records=0" for kv in args.opt) else 1
        roof, per_kernel = kernel_rooflines(timing, spec, rows, per_step, rec_bytes, kernel_names(ctx, spec))
        roof["launches_timed"] = per_kernel[roof["timing_slot"]]["launches"]
        if pipe is not None:       # the step one call at a time (hg_map: enqueue, wait, copy out, return): the figure comparable across rounds
            roof["sync_ms_per_step"] = round(pipe["sync_ms_per_step"], 5)
            roof["step_valu_equiv_frac_sync"] = round(roof["valu_equiv_frac"]["step"] * per_step * 1e3 / pipe["sync_ms_per_step"], 5)
        roof["timed"] = "HIP events around the kernel on every %s step of the timed region" % ("" if every == 1 else "%d-th" % every)
        out["roofline"] = roof
        out["kernels"] = {k_: {"avg_ms": round(v["avg_ms"], 5), "launches": v["launches"]} for k_, v in per_kernel.items()}
        if span:            # HIP events around the whole step on the GPU: what is left of the wall time is the host's
            busy = span[0] / max(span[1], 1)
            out["step_accounting"] = {"gpu_span_ms": round(busy, 5), "host_and_launch_ms": round(per_step * 1e3 - busy, 5),
                                      "kernels_timed_ms": round(sum(v["avg_ms"] * v["launches"] for v in per_kernel.values()) / max(span[1], 1), 5),
                                      "graph_replays": ctx.get_stat("graph_replays")}
    if dry_dir:
        out["dry_run_not_a_measurement"] = "ranks share GPU 0 and exchange through files (HG_BENCH_FILECOMM)"
    if sharded_leg:
        comm.barrier()
        if not dry_dir:
            ctx.comm_destroy()
    ctx.close()
    if world == 1 and not force:
        def side(key, fn):           # a side measurement must never cost the main line
            try:
                out[key] = fn()
            except Exception as e:      # noqa: BLE001
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}

        def h2d():
            d, m2 = h2d_inclusive(spec, packed)
            d["map_equal_to_timed_path"] = bool(m2 == out["map"])
            return d
        if not args.no_h2d:
            side("h2d_inclusive", h2d)
        if not args.no_literal:
            side("drop_in_literal", lambda: drop_in_literal(spec, packed))
        if not args.no_sorted and spec["kind"] == "planted":
            side("class_sorted_database", lambda: class_sorted(spec, packed))
        if not args.no_real:
            side("real_valued", lambda: real_valued(spec))
        if not args.no_large_r and wl == "c2":
            side("large_r", lambda: large_r(spec, packed))
        if not args.no_configs:
            copts = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt]
            out["configs"] = {}
            for nm in ("c2_iid", "c1", "c3", "c5"):
                if nm == wl:
                    continue
                try:
                    out["configs"][nm] = config_leg(nm, copts)
                except Exception as e:      # noqa: BLE001
                    out["configs"][nm] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_c4_ref and wl == "c2":
            side("scaling_reference_c4_one_gpu", lambda: c4_reference([(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt]))
        if not args.no_cpu_baseline:
            side("cpu_baseline", lambda: cpu_baseline(spec, packed))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
