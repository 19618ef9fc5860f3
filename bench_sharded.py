"""Multi-GPU leg of bench.py: launched by torch.distributed.run, one rank per GPU.

Weak scaling in database size: every rank holds its own N-row shard of a G*N-row
database (distinct rows per shard: own labels and noise), all ranks hold the Q
queries and evaluate mAP@R over the WHOLE database.  Exchanges per step: two
all-gathers over RCCL (shard histograms, match-bit rows) -- hashgan_amd/sharded.py.
"""
import json
import os
import time

import numpy as np


def run_sharded(args, spec, c0, packed0, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    from hashgan_amd import _native, metric, sharded
    from bench import build_inputs

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # HG_BENCH_DRYRUN=1: functional dry run of this leg on a ONE-GPU box (all ranks on GPU 0, gloo
    # rendezvous, exchanges staged through the host) -- never a measurement.
    dry = os.environ.get("HG_BENCH_DRYRUN") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dry:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    comm = sharded.TorchComm(host_sync=dry)                 # real runs: one stream, no host syncs between stages

    qw, ql, _, _ = packed0
    shard_spec = dict(spec)
    shard_spec["shard"] = rank
    c = build_inputs(shard_spec) if rank else c0           # rank 0 already holds shard 0
    Q, N, R, b = c["qbits"].shape[0], c["dbbits"].shape[0], c["R"], c["b"]
    dw, dl = metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"])

    ctx = _native.Context(local_rank)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.set_database(dw, dl, b, spec["C"], idx_base=rank * N, n_total=world * N)
    ctx.set_queries(qw, ql)
    eng = sharded.HipShardEngine(ctx, want_lists=False, share_stream=not dry)

    force = os.environ.get("HG_BENCH_FORCE_SHARDED") == "1"   # one-rank dry run of this leg over real RCCL

    def step():
        ap, rel = sharded.evaluate_shard(eng, comm, R, always_gather=force)
        return sharded.mean_ap(ap, rel)

    for _ in range(args.warmup):
        m = step()
    ctx.timing_enable(1)                                # HIP events around the passes over the pairs only
    ctx.timing_reset()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else "cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    timing = ctx.timing_read()
    maps = [None] * world
    dist.all_gather_object(maps, float(m))
    if rank == 0:
        per_step = dt / args.steps
        out = {
            "metric": "queries/sec (mAP@R of Q queries vs N-code database, Hamming ranking)",
            "value": Q * world / per_step, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp4 (E2M1 0/+-1) x fp4 -> f32 exact distances; u32 xor+popcount elsewhere; f64 AP", "data": "synthetic",
            "dry_run_not_a_measurement": dry,
            "config": {"workload": "%s sharded: Q=%d, %d shards x N=%d rows (database of %d rows), b=%d R=%d C=%d"
                                   % (args.workload.upper(), Q, world, N, world * N, b, R, spec["C"]),
                       "parallelism": "database sharded over %d GPUs; RCCL all-gather of histograms and match bits" % world},
            "value_definition": "Q queries x %d shards of %d rows per step / time: query-vs-1M-row-shard evaluations per "
                                "second (weak scaling: the database grows with the GPU count)" % (world, N),
            "queries_per_sec_raw": Q / per_step,
            "pairs_per_sec": Q * N * world / per_step,
            "map": maps[0], "map_identical_on_all_ranks": bool(all(x == maps[0] for x in maps)),
            "kernels_rank0": {k_: {"avg_ms": round(ms / max(cnt, 1), 5), "launches": cnt} for k_, (ms, cnt) in timing.items()},
        }
        if timing:
            import bench as B                          # rank 0's dominant kernel against its roofline (one shard's pairs)
            NW, LW = (b + 31) // 32, (spec["C"] + 63) // 64
            code_bytes = (Q + N) * NW * 4
            geo = {"k_hist": code_bytes + (b + 1) * ((Q + 63) // 64 * 64) * 4,
                   "k_select": code_bytes + (Q + N) * LW * 8 + Q * R * 8,
                   "k_select_mx": (Q + N) * ((NW + 1) // 2) * 32 + code_bytes + (Q + N) * LW * 8 + Q * R * 8}
            pair_passes = {k_: v for k_, v in timing.items() if k_ in geo}
            if pair_passes:
                out["roofline"], _ = B.kernel_rooflines(pair_passes, args.steps, dict(spec, Q=Q, N=N, b=b), geo)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
