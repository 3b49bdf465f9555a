"""The two Dogleg cases of the round-1 robot sweep (profiles/r01_stress_parity_robots.txt, cases 6 and 42) whose final
trajectories differ from the oracle's by more than the 1e-6 gate although iteration counts, status and error traces
agree.  What the per-iteration probes show (scripts/dogleg_cases.py, profiles/r02_dogleg_cases.txt):

  * at the FIRST iteration every trust-region scalar (g.g, g^T H g, g.dx_n) agrees with the oracle to <= 1e-14
    relative -- the order of the reductions is not the cause;
  * the Newton step itself differs by ~1e-11 relative (|dx_n|^2): two backward-stable factorisations (cyclic
    reduction here, natural-order block Cholesky in the oracle) of a matrix with cond(H) ~ 4e7;
  * that seed is amplified by one to two orders of magnitude per iteration by the problem itself (hinge switches,
    trust-region branches), with or without a dogleg blend (case 42 takes only full Gauss-Newton steps after k = 3).

The test pins this: the GPU may differ from the oracle by no more than a small multiple of what the ORACLE differs
from ITSELF when its initial values are perturbed in the last bits, and the first-iteration quantities must agree
to the bound the condition number allows."""
import copy

import numpy as np
import pytest

from parity_bound import K_SELF
from sweep_cases import robot_sweep_cases

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -52


def _dense_H(Hd, Ho):
    nb, n = Hd.shape[0], Hd.shape[1]
    H = np.zeros((nb * n, nb * n))
    for i in range(nb):
        H[i * n:(i + 1) * n, i * n:(i + 1) * n] = Hd[i]
        if i + 1 < nb:
            H[(i + 1) * n:(i + 2) * n, i * n:(i + 1) * n] = Ho[i]          # block (i+1, i)
            H[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = Ho[i].T
    return H


@pytest.mark.parametrize("which", [6, 42])
def test_dogleg_sweep_misses_are_the_problems_own_sensitivity(engine, oracle, which):
    case = next(c for c in robot_sweep_cases(which + 1) if c[0] == which)
    _, name, opt, p = case
    assert opt == "DOGLEG"
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    res = engine.batch_optimize(r, s, p.setting, *args, p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    # control flow and error traces: the usual gates hold
    assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    # the final error is E(returned values): held to the oracle's E at the GPU's own trajectory (no amplification in it)
    np.testing.assert_allclose(res["final_error"], oracle.graph_error(ro, so, p.setting, *args, res["traj"]), rtol=1e-9)
    d_gpu = np.abs(res["traj"] - ref["traj"]).reshape(p.B, -1).max(axis=1)

    # (1) the oracle against itself: initial values perturbed by +-2 ulp
    rng = np.random.default_rng(5)
    d_self = np.zeros(p.B)
    for _ in range(4):
        init2 = p.init * (1.0 + 2 * EPS * rng.choice([-1.0, 1.0], size=p.init.shape))
        alt = oracle.batch_optimize(ro, so, p.setting, *args, init2)
        same = (alt["iters"] == ref["iters"])
        dd = np.abs(alt["traj"] - ref["traj"]).reshape(p.B, -1).max(axis=1)
        d_self = np.maximum(d_self, np.where(same, dd, np.inf))       # a flipped iteration count is "infinitely" sensitive
    bound = np.maximum(1e-6, K_SELF * d_self)
    assert np.all(d_gpu <= bound), (d_gpu, d_self)
    assert d_gpu.max() < 2e-3

    # (2) first iteration of the worst trajectory: scalars to 1e-12, Newton step to cond(H) * eps
    b = int(np.argmax(d_gpu))
    one = [a[b:b + 1] for a in args]
    st = copy.copy(p.setting)
    st.fixed_iterations = 1
    pl = engine.plan(r, s, st, 1)
    pl.set_problem(*one, p.init[b:b + 1])
    pl.optimize()
    sc = pl.debug_scalars(0)
    with oracle.dogleg_probe() as pr:
        oracle.batch_optimize(ro, so, st, *one, p.init[b:b + 1])
    gg, ghg, gn, nn = pr.rows[0][:4]
    for mine, theirs in ((sc["gg"], gg), (sc["ghg"], ghg), (sc["gn"], gn)):
        assert abs(mine - theirs) <= 1e-12 * abs(theirs)
    Hd, Ho, _, _ = engine.linearize(r, s, p.setting, *one, p.init[b:b + 1])
    kappa = np.linalg.cond(_dense_H(Hd[0], Ho[0]))
    assert kappa > 1e6                                                  # these sweep settings are ill-conditioned
    assert abs(sc["nn"] - nn) <= 50.0 * kappa * EPS * nn
    print(f"case {which} {name}: cond(H) {kappa:.2e}, |dtraj| gpu-vs-oracle {d_gpu.max():.2e}, oracle-vs-perturbed-oracle "
          f"{np.max(d_self[np.isfinite(d_self)]) if np.any(np.isfinite(d_self)) else np.inf:.2e}")
