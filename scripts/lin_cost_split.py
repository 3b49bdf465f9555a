import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems
e = engine.Engine()
for eps in (0.2, -5.0):
    p = problems.wam_restarts(B=64)
    p.setting.set_epsilon(eps)
    p.setting.fixed_iterations = 3
    r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = e.plan(r, s, p.setting, p.B)
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    pl.optimize(); pl.enable_timing(True); pl.optimize()
    print('epsilon', eps, {k: round(v['ms'] / v['launches'] * 1e3, 1) for k, v in pl.timing().items()})
