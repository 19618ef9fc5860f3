#!/bin/bash
# round 6's stress runs (each tool prints one line per mismatch and a summary)
OUT=gpurun_out/fuzz6; mkdir -p $OUT
timeout 1500 python tools/fuzz_two_halves.py 300 6000 > $OUT/two_halves.txt 2>&1; tail -1 $OUT/two_halves.txt
timeout 1500 python tools/fuzz_real.py 400 6000 > $OUT/real.txt 2>&1; tail -1 $OUT/real.txt
timeout 900 python tools/fuzz_surface.py 300 6000 > $OUT/surface.txt 2>&1; tail -1 $OUT/surface.txt
timeout 900 python tools/fuzz_bet_vs_exact.py 300 6000 > $OUT/bet.txt 2>&1; tail -1 $OUT/bet.txt
timeout 900 python tools/fuzz_options.py 200 6000 > $OUT/options.txt 2>&1; tail -1 $OUT/options.txt
timeout 900 python tools/fuzz_sharded.py 200 6000 > $OUT/sharded.txt 2>&1; tail -1 $OUT/sharded.txt
timeout 900 python tools/fuzz_dense.py 100 6000 > $OUT/dense.txt 2>&1; tail -1 $OUT/dense.txt
