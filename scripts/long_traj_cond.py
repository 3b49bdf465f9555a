"""Error-trace agreement with the oracle on long, ill-conditioned trajectories for a given library build."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems
from oracle import Oracle
o = Oracle()
for path in sys.argv[1:]:
    e = engine.Engine(path)
    for N in (100, 300, 600):
        p = problems.wam_restarts(B=2, total_step=N, obs_check_inter=1, opt="GN", sdf="40", max_iter=4)
        r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
        ro, so = o.robot(p.model), o.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
        args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
        res = e.batch_optimize(r, s, p.setting, *args, p.init)
        ref = o.batch_optimize(ro, so, p.setting, *args, p.init)
        m = ~np.isnan(ref["error_trace"])
        rel = np.abs(res["error_trace"][m] - ref["error_trace"][m]) / np.abs(ref["error_trace"][m])
        print(path.split('/')[-1], 'N', N, 'iters', res["iters"], ref["iters"], 'max rel trace diff %.2e' % rel.max(),
              'max traj diff %.2e' % np.abs(res["traj"] - ref["traj"]).max(), flush=True)
