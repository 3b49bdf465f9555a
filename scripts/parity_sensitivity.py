"""Measures, per sweep case, what tests/test_gpu_parity_at_size.py asserts: GPU-vs-oracle trajectory difference per
trajectory and -- where it exceeds the 1e-6 contract -- the oracle's own sensitivity to a 2-ulp perturbation of its
initial values (tests/parity_bound.py).
usage: python scripts/parity_sensitivity.py [cases] [only]     (GPMP2MI_LIB / GPMP2MI_WIDE_DENSE select A/B builds)"""
import os
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine
from oracle import Oracle
from parity_bound import CONTRACT, oracle_self_sensitivity, per_traj_diff, solve_both
from sweep_cases import robot_sweep_cases

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
only = int(sys.argv[2]) if len(sys.argv) > 2 else None
eng, orc = engine.Engine(), Oracle()
tag = os.environ.get("GPMP2MI_LIB", "HEAD") + (" dense" if os.environ.get("GPMP2MI_WIDE_DENSE") == "1" else "")
n_over = 0
for case, name, opt, p in robot_sweep_cases(cases):
    if only is not None and case != only:
        continue
    res, ref, handles = solve_both(eng, orc, p)
    same = list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    d = per_traj_diff(res, ref)
    rel = np.abs(res["final_error"] / ref["final_error"] - 1.0)
    over = np.nonzero(d > CONTRACT)[0]
    line = (f"[{tag}] case {case:3d} {name:18s} N={p.setting.total_step:3d} I={p.setting.obs_check_inter} {opt:6s} B={p.B:2d} "
            f"flow {'same' if same else 'DIFF'} max|dtraj| {d.max():.1e} rel err {rel.max():.1e}")
    if over.size:
        n_over += 1
        ds = oracle_self_sensitivity(orc, handles, p, ref, over)
        line += " | over 1e-6: " + ", ".join(f"b{b}: gpu {d[b]:.1e} self {s:.1e} ratio {d[b] / s if s > 0 else np.inf:.2f} relerr {rel[b]:.1e}"
                                             for b, s in zip(over, ds))
    print(line, flush=True)
print(f"{n_over} cases above the 1e-6 contract")
