"""Per-kernel HIP-event times of the headline workload (config 3, GN to tolerance) for the current
GPMP2MI_LIN_PIPE setting; run once with =0 and once with =1."""
import os
import sys

sys.path.insert(0, '.')
from gpmp2_amd import engine, problems

e = engine.Engine()
for B in (64, 1024):
    p = problems.wam_restarts(B=B)
    r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = e.plan(r, s, p.setting, p.B)
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    pl.optimize()
    pl.enable_timing(True)
    pl.optimize()
    print('PIPE', os.environ.get('GPMP2MI_LIN_PIPE', 'auto'), 'B', B,
          {k: round(v['ms'] / v['launches'] * 1e3, 1) for k, v in pl.timing().items()}, flush=True)
