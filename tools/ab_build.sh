#!/bin/bash
# usage: tools/ab_build.sh name "-DHG_X=1" [name2 "-D..."] ... : experiment builds into hashgan_amd/_lib/ab_<name>.so (parallel)
while [ $# -gt 0 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -ldl $f \
     -o hashgan_amd/_lib/ab_$n.so hashgan_amd/csrc/hg_engine.hip 2> /tmp/ab_$n.log &
done
wait; ls -la hashgan_amd/_lib/
