#!/usr/bin/env python3
"""Register / scratch / LDS usage of the kernels whose mangled name contains a pattern, from the device assembly
(hipcc -S --cuda-device-only -o /tmp/hg_unit.s hashgan_amd/csrc/<unit>.hip)."""
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "k_select_mx"
s = open(sys.argv[2] if len(sys.argv) > 2 else "/tmp/hg_unit.s").read()
for blk in s.split("  - .agpr_count:")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m or pat not in m.group(1):
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print("%-70s vgpr %s sgpr %s scratch %s spill %s" % (m.group(1)[:70], g(r"\.vgpr_count"), g(r"\.sgpr_count"),
          g(r"\.private_segment_fixed_size"), g(r"\.vgpr_spill_count")))
