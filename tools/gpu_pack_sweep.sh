#!/bin/bash
# host packing threads under the box's CPU quota: the literal C2 call (tools/literal_outliers.py, 300 calls back to back) per thread count, twice
for i in 1 2; do for t in 16 24 32 40 48 64; do echo "HG_PACK_THREADS=$t"; HG_PACK_THREADS=$t python tools/literal_outliers.py c2 300 | head -2; done; done
