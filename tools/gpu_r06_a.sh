#!/bin/bash
# round 6, first GPU call: the new tests, the new-context probe (where do the first-call stalls come from), one bench line, the suite
OUT=gpurun_out/r06a; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "two_halves or blind or batch_after or recycles or surface or independent" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $OUT/pytest_new.log
timeout 300 python tools/new_context_probe.py --contexts 8 > $OUT/probe_plain.jsonl 2> $OUT/probe_plain.err; echo "probe rc=$?"
timeout 300 python tools/new_context_probe.py --contexts 8 --sorted > $OUT/probe_sorted.jsonl 2> $OUT/probe_sorted.err
timeout 300 python tools/new_context_probe.py --contexts 8 --sorted --keep-one > $OUT/probe_sorted_keep.jsonl 2> $OUT/probe_sorted_keep.err
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $OUT/pytest_all.log
