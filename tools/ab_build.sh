#!/bin/bash
# usage: tools/ab_build.sh name "-DHG_X=1" [name2 "-D..."] ... : experiment builds into hashgan_amd/_lib/ab_<name>.so
# (every translation unit of hashgan_amd/build.py with the extra switches; objects under _lib/obj/prod_<switches>/)
while [ $# -gt 0 ]; do
  n=$1; f=$2; shift 2
  python3 - "$n" $f <<'PY' 2> /tmp/ab_$n.log &
import sys
sys.path.insert(0, ".")
from hashgan_amd import build
print(build.build(force=True, extra_flags=sys.argv[2:], out=build.LIB_DIR + "/ab_%s.so" % sys.argv[1]))
PY
done
wait; ls -la hashgan_amd/_lib/
