#!/bin/bash
# end-of-round soak: the whole gpu suite under HG_EFENCE=1 (every device buffer ends at an unmapped page), then longer stress runs on new seeds
OUT=gpurun_out/soak6; mkdir -p $OUT
HG_EFENCE=1 timeout 3000 python -m pytest tests -m gpu -q -x > $OUT/efence1_all.log 2>&1; echo "efence1 suite rc=$?"; tail -2 $OUT/efence1_all.log
timeout 1500 python tools/fuzz_two_halves.py 600 7000 > $OUT/two_halves.txt 2>&1; tail -1 $OUT/two_halves.txt
timeout 1500 python tools/fuzz_real.py 800 7000 > $OUT/real.txt 2>&1; tail -1 $OUT/real.txt
timeout 900 python tools/fuzz_surface.py 600 7000 > $OUT/surface.txt 2>&1; tail -1 $OUT/surface.txt
timeout 1200 python tools/fuzz_bet_vs_exact.py 600 7000 > $OUT/bet.txt 2>&1; tail -1 $OUT/bet.txt
timeout 900 python tools/fuzz_sharded.py 400 7000 > $OUT/sharded.txt 2>&1; tail -1 $OUT/sharded.txt
