#!/bin/bash
# round 6, second GPU call: block cache in place -- new-context probe again, the literal call's breakdown, two-stream overlap bound, suite
OUT=gpurun_out/r06b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "two_halves or blind or batch_after or recycles or surface or independent" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $OUT/pytest_new.log
timeout 300 python tools/new_context_probe.py --contexts 12 --sorted --keep-one > $OUT/probe_sorted_keep.jsonl 2> $OUT/probe_sorted_keep.err; echo "probe rc=$?"
timeout 300 python tools/new_context_probe.py --contexts 12 > $OUT/probe_plain.jsonl 2> $OUT/probe_plain.err
timeout 300 python tools/literal_breakdown.py cifar nus c2 > $OUT/literal_breakdown.txt 2>&1; echo "breakdown rc=$?"
timeout 300 python tools/two_context_overlap.py > $OUT/two_context_overlap.txt 2>&1; echo "overlap rc=$?"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $OUT/pytest_all.log
