import sys, time, numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(1, "/root/repo")
import hashgan_amd
from hashgan_amd import _native, metric
from tests import cases
c = cases.build_case("c1_cifar_full")
ctx = _native.Context(0)
ctx.set_database(metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"]), c["b"], 10)
ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))
for _ in range(3): ctx.map(c["R"])
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(10): ctx.map(c["R"])
print(root, hashgan_amd.__file__, {k: round(v[0]/v[1],3) for k,v in ctx.timing_read().items()})
