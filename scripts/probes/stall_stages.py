"""which stage of a one-shot plan call waits for ANOTHER (stalled, non-blocking) stream?"""
import ctypes as C, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from gpmp2_amd import engine as E, problems
eng = E.Engine()
p = problems.wam_restarts(B=4, total_step=20, obs_check_inter=3, sdf="40")
args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
def stream():
    st = C.c_void_p(); eng._ck(eng.lib.gpmp2mi_debug_stream_create(C.byref(st))); return st
mine, other = stream(), stream()
n = 0
T0 = time.perf_counter()
def one(tag):
    t = [time.perf_counter()]
    pl = eng.plan(r, s, p.setting, p.B); t.append(time.perf_counter())
    pl.set_problem(*args, p.init); t.append(time.perf_counter())
    pl.optimize(stream=mine.value); t.append(time.perf_counter())
    pl.result(); t.append(time.perf_counter())
    pl.close(); t.append(time.perf_counter())
    global n
    n += 1
    if t[-1] - t[0] > 5e-3 or n % 100 == 0:
        print(tag, n, "at %.3f s:" % (t[0] - T0), "create %.2f set %.2f optimize %.2f result %.2f close %.2f ms" % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])), flush=True)
for _ in range(3): one("idle ")
tok = C.c_void_p()
eng._ck(eng.lib.gpmp2mi_debug_stall_begin(other, 400, C.byref(tok)))
T0 = t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3: one("busy ")
eng._ck(eng.lib.gpmp2mi_debug_stall_release(tok))
