#!/bin/bash
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 300 python tools/literal_trace.py --bench-leg > $OUT/literal_trace.txt 2> $OUT/literal_trace.err; echo "trace rc=$?"
timeout 300 python tools/new_context_probe.py --contexts 8 --sorted --keep-one > $OUT/probe_sorted_keep2.jsonl 2> $OUT/probe_sorted_keep2.err; echo "probe rc=$?"
