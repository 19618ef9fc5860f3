#!/bin/bash
# usage: tools/gpu_quick.sh <tag> : the fast check after a kernel change -- parity tests of the bet path, two headline bench lines
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py -m gpu -q -x -k "golden or stages or lost or bursts or far or long_codes or odd or ap_through or full or interleaved or fused_step" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref > $OUT/bench_$i.json 2> $OUT/bench_$i.err; done
python - $OUT <<'PY'
import json, sys
for i in (1, 2):
    try:
        d = json.loads(open("%s/bench_%d.json" % (sys.argv[1], i)).read().strip().splitlines()[-1])
        print(round(d["ms_per_step"], 4), round(d["ms_per_step_min"], 4), d["parity_vs_reference_golden"], d["step_accounting"], d["kernels"])
        print({k: (round(v["ms_per_step"], 4), v["parity_vs_reference_golden"]) for k, v in d.get("configs", {}).items()})
    except Exception as e:
        print("bench", i, "ERR", e, open("%s/bench_%d.err" % (sys.argv[1], i)).read()[-1500:])
PY
