#!/bin/bash
# usage: tools/gpu_ab_shape.sh "Q N b R [opts]" ... : tools/shape_sweep.py for every hashgan_amd/_lib/ab_*.so, per shape
for sh in "$@"; do for so in hashgan_amd/_lib/ab_*.so; do echo -n "$(basename $so .so) "; HG_LIBRARY=$PWD/$so python tools/shape_sweep.py $sh | cut -c1-210; done; done
