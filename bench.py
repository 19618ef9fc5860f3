#!/usr/bin/env python3
"""Benchmark of the retrieval-evaluation hot path (BASELINE.json metric:
queries/sec at Q=10k, N=1M, b=64, R=5000 -- config C2 -- on MI355X).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5|c1]

One step = one full pass of the path over the query batch with codes and labels
already resident in HBM: distance histogram -> threshold plan -> select ->
order -> label match -> AP (all HIP) -> per-query AP to the host -> mean.
N > 1 (launched by torch.distributed.run, one rank per GPU): the database is
sharded, every rank holds `N` rows of it (weak scaling in database size), the
shards exchange histograms and match bits over RCCL (see hashgan_amd/sharded.py).
Prints ONE JSON line on rank 0.
"""
import argparse
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
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MX-fp4 MFMA peak (MI355X_MICROARCH.md: ~10 PF dense; 9.1 PF measured)

WORKLOADS = {
    # name: (case in tests/cases.py whose seeds/shape we reuse, Q, golden anchor)
    "c2": dict(Q=10000, N=1000000, b=64, R=5000, C=10, kind="planted", seed=0xC2, flip=0.30, golden="c2_q64"),
    "c5": dict(Q=10000, N=1000000, b=128, R=5000, C=10, kind="planted", seed=0xC5, flip=0.35, golden="c5_b128_q32"),
    "c3": dict(Q=2100, N=190000, b=48, R=5000, C=81, kind="multihot", seed=0xC3, flip=0.20, golden="c3_nus_q64"),
    "c1": dict(Q=1000, N=54000, b=32, R=54000, C=10, kind="cifar", seed=0xC1, flip=0.25, golden="c1_cifar_full"),
}


def build_inputs(spec):
    from tests import cases
    s = {k: v for k, v in spec.items() if k != "golden"}
    cases.CASES["_bench"] = s
    try:
        return cases.build_case("_bench")
    finally:
        del cases.CASES["_bench"]


def cpu_baseline(c, spec, budget_s=12.0, chunk=128):
    """metric.py:12-24 as written (float32 np.dot -> np.argsort(-ips,1) -> Python
    loop; oracle.reference_as_written) on the GPU box's host cores, over query
    chunks (rows are independent; a chunk bounds the reference's 16 B/pair) until
    ~budget_s seconds of CPU work are done."""
    from oracle import hamming_map as O
    Q = c["qbits"].shape[0]
    dbf = c["dbbits"].astype(np.float32) * 2 - 1
    dl = c["dblab"].astype(np.int64)
    chunk = max(1, min(chunk, int(2e9 // (16 * dbf.shape[0])) or 1))
    done, t0 = 0, time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while done < Q and time.perf_counter() - t0 < budget_s:
            sl = slice(done, min(Q, done + chunk))
            O.reference_as_written(dbf, dl, c["qbits"][sl].astype(np.float32) * 2 - 1, c["qlab"][sl].astype(np.int64), c["R"])
            done = sl.stop
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        blas = [(p.get("internal_api"), p.get("num_threads")) for p in threadpoolctl.threadpool_info()]
    except Exception:      # noqa: BLE001
        blas = None
    return {"value": done / dt, "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d of %d queries (chunks of %d) x full N=%d database in %.1f s; float32 +-1 features; np.dot on "
                      "BLAS threads %s, np.argsort and the per-query loop on 1 core; numpy %s"
                      % (done, Q, chunk, c["dbbits"].shape[0], dt, blas, np.__version__)}


def traffic_from_profiles(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass
    (profiles/latest_traffic.json, written by tools/pmc_traffic.py), or None."""
    path = os.path.join(ROOT, "profiles", "latest_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def kernel_rooflines(timing, steps, spec, geo_bytes):
    """Per-kernel average launch time -> VALU and HBM fractions of the dominant one."""
    Q, N, b = spec["Q"], spec["N"], spec["b"]
    NW = (b + 31) // 32
    pairs = Q * N
    out = {}
    for name, (ms, cnt) in timing.items():
        out[name] = {"avg_ms": ms / max(cnt, 1), "launches": cnt}
    dom = max(out, key=lambda k: out[k]["avg_ms"] * out[k]["launches"])
    t = out[dom]["avg_ms"] * 1e-3
    alg_bytes = geo_bytes.get(dom, 0)
    hbm = {"algorithmic_bytes": alg_bytes, "achieved": alg_bytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / t / 1e9 / HBM_PEAK_GBS}
    if dom == "k_select_mx":
        # matrix-core select: one v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) per 64 code bits and 32x32 tile of pairs
        K = 64 * ((NW + 1) // 2)
        flops = 2.0 * pairs * K
        valu = pairs                                   # one v_alignbit_b32 per pair (sign of the accumulator -> hit mask)
        roof = {"bound": "mfma", "kernel": dom, "achieved": flops / t / 1e12, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / MFMA_FP4_PEAK_TFLOPS, "traffic": traffic_from_profiles(dom),
                "avg_launch_ms": out[dom]["avg_ms"], "algorithmic_flops": flops,
                "note": "fp4 MFMA inner product over K=%d code bits per pair (2 flops per bit), against the dense fp4 peak; the "
                        "kernel's other per-pair cost is one v_alignbit_b32, which on gfx950 does not overlap the MFMAs "
                        "(profiles/r01_ubench_mx.txt): see 'valu'" % K,
                "valu": {"algorithmic_laneops": valu, "achieved": valu / t / 1e12, "peak": VALU_PEAK_TLANEOPS,
                         "unit": "Tlaneop/s", "frac": valu / t / 1e12 / VALU_PEAK_TLANEOPS},
                "hbm": hbm}
        return roof, out
    laneops = pairs * 2 * NW                      # one v_xor_b32 + one v_bcnt_u32_b32 per 32-bit word per pair
    roof = {"bound": "valu", "kernel": dom, "achieved": laneops / t / 1e12, "peak": VALU_PEAK_TLANEOPS,
            "unit": "Tlaneop/s", "frac": laneops / t / 1e12 / VALU_PEAK_TLANEOPS, "traffic": traffic_from_profiles(dom),
            "avg_launch_ms": out[dom]["avg_ms"],
            "algorithmic_laneops": laneops,
            "note": "integer bit-count path: xor+popcount lane-ops (2 per 32-bit code word per pair) against the "
                    "integer VALU issue peak (16 lanes/clk/SIMD); frac vs the 78.6 T packed-FP32 figure is %.3f"
                    % (laneops / t / 1e12 / VALU_NOMINAL_FP32_TLANEOPS),
            "hbm": hbm}
    return roof, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--target-units", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (hg_set_option), repeatable")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP events (overhead probe)")
    ap.add_argument("--kernel-timing", default="pair-passes", choices=["pair-passes", "all"],
                    help="HIP events around the passes over the pairs only (the roofline kernel; default) or around every "
                         "kernel (+0.04 ms per step: events keep kernels from being dispatched back to back)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))

    spec = WORKLOADS[args.workload]
    from hashgan_amd import _native, metric
    c = build_inputs(spec)
    Q, N, R, b = c["qbits"].shape[0], c["dbbits"].shape[0], c["R"], c["b"]
    qw, ql = metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"])
    dw, dl = metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"])

    if world > 1 or os.environ.get("HG_BENCH_FORCE_SHARDED") == "1":
        from bench_sharded import run_sharded
        return run_sharded(args, spec, c, (qw, ql, dw, dl), rank, local_rank, world)

    ctx = _native.Context(0)
    if args.target_units:
        ctx.set_option("target_units", args.target_units)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.set_database(dw, dl, b, spec["C"])           # inputs resident in HBM before the timed region
    ctx.set_queries(qw, ql)

    def step():
        a, r = ctx.map(R)
        return metric.mean_over_hits(a, r), a

    for _ in range(args.warmup):
        m, a = step()
    ctx.timing_enable(0 if args.no_kernel_timing else (2 if args.kernel_timing == "all" else 1))
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m, a = step()
    dt = time.perf_counter() - t0
    timing = ctx.timing_read() or {"k_select": (dt * 1e3, args.steps)}
    ctx.timing_enable(False)

    # parity flag: the first queries are a golden case of the unmodified reference
    from tests import cases
    g = cases.load_golden(spec["golden"])
    k = g["ap"].shape[0]
    parity = bool(np.array_equal(a[:k], g["ap"], equal_nan=True))

    NW = (b + 31) // 32
    NB = b + 1
    Qpad = (Q + 63) // 64 * 64
    code_bytes = (Q + N) * NW * 4
    LW = (spec["C"] + 63) // 64
    geo_bytes = {  # algorithmic (compulsory) HBM bytes per launch, DESIGN.md section 4
        "k_hist": code_bytes + NB * Qpad * 4,
        "k_select": code_bytes + (Q + N) * LW * 8 + Q * R * 8,      # codes + labels in, >= R records of 8 B out per query
        # fp4 images of the codes (4 bits per code bit, 64-bit granules) + packed codes + labels in, records out
        "k_select_mx": (Q + N) * ((NW + 1) // 2) * 32 + code_bytes + (Q + N) * LW * 8 + Q * R * 8,
    }
    roof, per_kernel = kernel_rooflines(timing, args.steps, spec, geo_bytes)
    ms = dt / args.steps * 1e3
    out = {
        "metric": "queries/sec (mAP@R of Q queries vs N-code database, Hamming ranking)",
        "value": Q / (dt / args.steps), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp4 (E2M1 0/+-1) x fp4 -> f32 exact distances; u32 xor+popcount elsewhere; f64 AP", "data": "synthetic",
        "config": {"workload": "%s: Q=%d N=%d b=%d R=%d C=%d, planted codes" % (args.workload.upper(), Q, N, b, R, spec["C"]),
                   "parallelism": "1 GPU"},
        "map": float(m), "parity_vs_reference_golden": parity,
        "optimistic_runs": ctx.get_stat("optimistic_runs"), "optimistic_fallbacks": ctx.get_stat("optimistic_fallbacks"),
        "pairs_per_sec": Q * N / (dt / args.steps),
        "roofline": roof,
        "kernels": {k_: {"avg_ms": round(v["avg_ms"], 5), "launches": v["launches"]} for k_, v in per_kernel.items()},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(c, spec)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
