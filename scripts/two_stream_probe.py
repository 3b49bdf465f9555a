"""Would running the 64 restarts as two half-batches on two streams (kernels of one half overlapping the per-trajectory
step kernel of the other) pay?  Probe with independent plans of 32 / 16 driven from host threads."""
import sys, threading, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems

eng = engine.Engine()
p = problems.wam_restarts(B=64)
r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
args = lambda sl: [np.ascontiguousarray(a[sl]) for a in (p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)]


def timed(plans, streams, reps=20):
    def run(pl, st):
        for _ in range(reps):
            pl.optimize(stream=st.cuda_stream)
    for pl, st in zip(plans, streams):
        pl.optimize(stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(pl, st)) for pl, st in zip(plans, streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


one = eng.plan(r, s, p.setting, 64); one.set_problem(*args(slice(0, 64)))
print("one plan of 64: %.3f ms per batch" % timed([one], [torch.cuda.Stream()]))
for parts in (2, 4):
    n = 64 // parts
    plans = []
    for k in range(parts):   # interleave the restarts so that each part gets a similar mix of iteration counts
        pl = eng.plan(r, s, p.setting, n); pl.set_problem(*args(slice(k, 64, parts))); plans.append(pl)
    print("%d plans of %d on %d streams: %.3f ms per 64" % (parts, n, parts, timed(plans, [torch.cuda.Stream() for _ in range(parts)])))
