python bench.py --steps 140 --warmup 7 --cpu-iters-per-stage 0 --no-kernel-timing 2>&1 | tail -1 | cut -c1-230
python bench.py --steps 140 --warmup 7 --cpu-iters-per-stage 0 2>&1 | tail -1 | cut -c1-230
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0,'.')
from linearcorex_amd.backend import HipBackend
be = HipBackend(10000, 5000, 32, np.float64, 0)
be.generate_x(1,0,1,0)
w=(np.random.RandomState(0).randn(32,5000)*0.003)
be.set_ws(w); be.moments_a(0); be.moments_b(0,0.6,0); be.moments_c(0); print(be.read_state(0)[:3])
def direction():
    be.update_a(); be.update_b(0.6); be.update_c(0.6); be.update_d()
def trial():
    be.trial_linear_a(1e-3); be.trial_linear_b(0.6,1e-3); be.moments_c(1)
for name,fn,sync in (("direction",direction,0),("trial",trial,1),("direction+trial",lambda:(direction(),trial()),1)):
    for rep in range(2):
        be.synchronize(); t=time.perf_counter()
        for i in range(50):
            fn()
            be.read_state(sync)
        dt=(time.perf_counter()-t)/50
    print(name,"with sync each: %.1f us"%(dt*1e6))
    be.synchronize(); t=time.perf_counter()
    for i in range(50): fn()
    t1=time.perf_counter()-t
    be.synchronize(); dt=(time.perf_counter()-t)/50
    print(name,"async 50x: host enqueue %.1f us, total %.1f us per call"%(t1/50*1e6, dt*1e6))
PY
