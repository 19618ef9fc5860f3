import numpy as np, sys
sys.path.insert(0,"/root/repo")
from hashgan_amd import _native
rng=np.random.default_rng(0)
import os
Q,N,b=(int(x) for x in os.environ.get("HG_PROF_SHAPE","10000,1000000,64").split(","))     # HG_PROF_SHAPE=1000,54000,64 HG_PROF_R=54000: the reference's CIFAR-10 evaluation
R=int(os.environ.get("HG_PROF_R","5000"))
dbf=np.tanh(rng.standard_normal((N,b))).astype(np.float32); qf=np.tanh(rng.standard_normal((Q,b))).astype(np.float32)
dl=np.zeros((N,10),np.int64); dl[np.arange(N),rng.integers(0,10,N)]=1
ql=np.zeros((Q,10),np.int64); ql[np.arange(Q),rng.integers(0,10,Q)]=1
import os
ctx=_native.Context(0); ctx.set_option('real_mfma', int(os.environ.get('HG_REAL_MFMA','2'))); [ctx.set_option(kv.split('=')[0], int(kv.split('=')[1])) for kv in os.environ.get('HG_PROF_OPTS','').split(',') if kv]; ctx.set_database_f32(dbf,dl); ctx.set_queries_f32(qf,ql)
def run():
    try: ctx.map_real(R)
    except Exception as e: pass
for _ in range(2): run()
import time
ctx.timing_enable(2); ctx.timing_reset()
t=time.perf_counter()
for _ in range(3): run()
dt=(time.perf_counter()-t)/3
print("real_mfma=%s %.2f ms per call" % (os.environ.get("HG_REAL_MFMA","2"), dt*1e3), {k: round(v[0]/max(v[1],1),3) for k,v in ctx.timing_read().items()}, {k: v[1] for k,v in ctx.timing_read().items()})
print("segments", ctx.get_stat("segments"), "records_kept", ctx.get_stat("records_kept"))
