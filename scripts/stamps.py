"""In-kernel s_memtime stamps of the shipped kernels (diagnostic build: `make -C gpmp2_amd/csrc stamps`).  Every kernel
stamps only while a trajectory is in its SECOND iteration (G2_STAMP_ITER), so the 64 slots of a trajectory hold one
pass of every kernel and no difference mixes passes.
usage: python scripts/stamps.py gpmp2_amd/csrc/build/stamps/libgpmp2mi_stamps.so"""
import ctypes as C
import subprocess
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems

path = sys.argv[1]
e = engine.Engine(path)
p = problems.wam_restarts(B=64)
r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
st = p.setting
st.fixed_iterations = 3
pl = e.plan(r, s, st, p.B)
pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
for _ in range(3):
    pl.optimize()
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = "?"
print(f"# s_memtime stamps (shader cycles), WAM N=100 I=5, 64 restarts, second Gauss-Newton iteration; commit {head or '?'}")


def d(a):
    a = np.asarray(a, dtype=np.float64)
    return [int(x) for x in np.diff(a[a > 0])]


for b in (0, 33):
    out = (C.c_ulonglong * 64)()
    e._ck(e.lib.gpmp2mi_plan_debug_stamps(pl.h.ptr, b, out))
    raw = np.array(list(out), dtype=np.float64)
    print(f"trajectory {b}")
    print("   linearize wavefront 0 (chunk 1), k_linearize_arm: fused finish (levels 4, 2, 1 of the previous step for its states) + state loads + interpolation + sin/cos + barrier | chain walk (wavefront 0) + barrier | "
          "its 4 spheres | wait for the others + tree sum | record store | (gp prior: last wavefront)", d(raw[48:64]))
    print("   assemble wave i=1 (odd block): stage / build / misc / eliminate / store", d(raw[32:38]))
    print("   build_tiles i=1: owner rows / constants + unary / sub-step loop / replanner priors + shuffles", d(raw[24:29]))
    t2 = raw[40:48]
    print("   assemble wave i=2 (level 2): stage / build / ... / wait for the odd blocks / products + eliminate + store", d(t2))
    if raw[47] > 0 and raw[32] > 0:
        print("   assemble i=1 start -> i=2 end:", int(raw[47] - raw[32]))
    if raw[4] > raw[0] > 0:
        print("   step kernel: control", int(raw[1] - raw[0]), "forward", int(raw[2] - raw[1]), "backward", int(raw[3] - raw[2]),
              "hand-over", int(raw[4] - raw[3]), "total", int(raw[4] - raw[0]))
    out2 = (C.c_ulonglong * 64)()
    e._ck(e.lib.gpmp2mi_plan_debug_stamps(pl.h.ptr, p.B + b, out2))
    task = np.array(list(out2), dtype=np.float64)
    for nm, o in (("level 4", 0), ("level 8", 8), ("level 16", 16)):
        v = task[o:o + 6]
        if np.all(v > 0):
            print(f"   one elimination task at {nm} (wave 0): loads / tile products / eliminate + W products / stores / wait at the barrier", d(v))
    fw = [raw[1]] + [raw[5 + k] for k in range(1, 9) if raw[5 + k] > 0]
    print("   forward levels h = 4, 8, ..:", [int(x) for x in np.diff(fw)])
    bw = [raw[2]] + [raw[16 + k] for k in range(7, -1, -1) if raw[16 + k] > 0]
    print("   backward levels h = final, ..:", [int(x) for x in np.diff(bw)])
