#!/bin/bash
# usage: tools/gpu_ab_opts.sh <tag> "<label>|<HG_LIBRARY or ->|<bench args>" ... : the headline step under option / library variants,
# one line each: ms per step (mean of 3 runs), parity, step accounting, per-kernel averages when timed
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --no-configs"
for spec in "$@"; do
  IFS='|' read -r label lib args <<< "$spec"
  for i in 1 2 3; do
    if [ "$lib" = "-" ]; then $B $args > $OUT/${label}_$i.json 2> $OUT/${label}_$i.err
    else HG_LIBRARY=$lib $B $args > $OUT/${label}_$i.json 2> $OUT/${label}_$i.err; fi
  done
  python - "$OUT" "$label" <<'PY'
import json, sys
out, label = sys.argv[1:3]
ms, last = [], None
for i in (1, 2, 3):
    try:
        d = json.loads(open("%s/%s_%d.json" % (out, label, i)).read().strip().splitlines()[-1])
        ms.append(d["ms_per_step"]); last = d
    except Exception as e:
        print(label, "ERR", e, open("%s/%s_%d.err" % (out, label, i)).read()[-800:])
if last:
    print("%-14s %s mean %.4f parity %s fallbacks %s kept/R %s span %s kernels %s" % (label, ["%.4f" % m for m in ms], sum(ms) / len(ms),
          last.get("parity_vs_reference_golden"), last.get("optimistic_fallbacks"), last.get("records_kept_over_R"), last.get("step_accounting", {}).get("gpu_span_ms"),
          {k: round(v["avg_ms"], 4) for k, v in last.get("kernels", {}).items()}))
PY
done
