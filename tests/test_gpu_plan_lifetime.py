"""Lifetime of a gpmp2mi_plan (ownership notes SURVEY.md 8(b); gpmp2/planner/ISAM2TrajOptimizer.h:68-74 owns copies of
everything it plans with, BatchTrajOptimize* builds and drops its graph per call):

  * a gpmp2mi_plan_create that fails half-way (out of memory, here injected with GPMP2MI_FAIL_ALLOC_AT) or on an
    invalid description returns every arena chunk and the pass-flag buffer;
  * gpmp2mi_plan_destroy waits for the plan's own streams only: a one-shot gpmp2mi_batch_optimize on one stream is
    not held up by work another stream still has in flight;
  * a pass that does not finish within GPMP2MI_WAIT_TIMEOUT_MS returns GPMP2MI_ERR_TIMEOUT instead of hanging the
    caller, also in the calls that follow and in the destroy (the plan is poisoned, its memory leaked on purpose)."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from gpmp2_amd import problems
from gpmp2_amd.engine import Gpmp2miError, Plan

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _counts(engine):
    v = [C.c_long() for _ in range(5)]
    assert engine.lib.gpmp2mi_debug_resource_counts(*[C.byref(x) for x in v]) == 0
    return dict(zip(("live_chunks", "pooled_chunks", "live_flagbufs", "pooled_flagbufs", "leaked_plans"), (x.value for x in v)))


def _args(p):
    return p.start_conf, p.start_vel, p.end_conf, p.end_vel


def _stream(engine):
    st = C.c_void_p()
    engine._ck(engine.lib.gpmp2mi_debug_stream_create(C.byref(st)))
    return st


def test_failed_create_returns_every_chunk_and_flag_buffer(engine, monkeypatch):
    p = problems.wam_restarts(B=4, total_step=20, obs_check_inter=3, sdf="40")
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = engine.plan(r, s, p.setting, p.B)      # a plan that works: counts its allocations, fills the pools on close
    before_live = _counts(engine)
    pl.close()
    base = _counts(engine)
    assert base["live_chunks"] < before_live["live_chunks"] and base["live_flagbufs"] == before_live["live_flagbufs"] - 1
    # plans with extra factors allocate after the main buffers: fail early, in the middle and at the very end
    p.setting.add_workspace_prior(0, 6, np.eye(4), 1e-2, 1, 5)
    failed = 0
    for k in (1, 2, 3, 5, 8, 13, 21, 34, 44, 46, 48, 50, 52, 90):
        monkeypatch.setenv("GPMP2MI_FAIL_ALLOC_AT", str(k))
        try:
            q = engine.plan(r, s, p.setting, p.B)
        except Gpmp2miError as e:
            assert e.code == 5 and "injected" in str(e)
            failed += 1
        else:
            q.close()                            # k beyond the plan's last allocation: the create succeeds
        now = _counts(engine)
        assert now["live_chunks"] == base["live_chunks"] and now["live_flagbufs"] == base["live_flagbufs"], (k, now, base)
    assert failed >= 10
    monkeypatch.delenv("GPMP2MI_FAIL_ALLOC_AT")
    # an invalid description is rejected before anything is allocated
    p.setting.add_workspace_prior(0, 99, np.eye(4), 1e-2, 1, 5)          # link out of range
    pooled = _counts(engine)["pooled_chunks"]
    with pytest.raises(Gpmp2miError) as ei:
        engine.plan(r, s, p.setting, p.B)
    assert ei.value.code == 1
    now = _counts(engine)
    assert now["live_chunks"] == base["live_chunks"] and now["pooled_chunks"] == pooled
    # and the library still plans correctly afterwards (pooled chunks are zero-filled again on reuse)
    p2 = problems.wam_restarts(B=4, total_step=20, obs_check_inter=3, sdf="40")
    a = engine.batch_optimize(r, s, p2.setting, *_args(p2), p2.init)
    b = engine.batch_optimize(r, s, p2.setting, *_args(p2), p2.init)
    np.testing.assert_array_equal(a["traj"], b["traj"])


def test_unsupported_dof_is_rejected_at_create(engine):
    """dof 12..16 have no normal-equation export / dense-solve instantiation: refused when the plan is created, not
    inside optimize"""
    import gpmp2_amd as g
    from gpmp2_amd.settings import TrajOptimizerSetting
    arm = g.Arm(13, [0.2] * 13, [0.0] * 13, [0.0] * 13)
    model = g.ArmModel(arm, [g.BodySphere(l, 0.05, (0, 0, 0)) for l in range(13)])
    st = TrajOptimizerSetting(13)
    st.set_total_step(8); st.set_total_time(1.0); st.set_obs_check_inter(0); st.set_Qc_model(np.eye(13))
    origin, cell, data = problems.small3d_sdf(40)
    r, s = engine.robot(model), engine.sdf(origin, cell, data)
    with pytest.raises(Gpmp2miError) as ei:
        engine.plan(r, s, st, 1)
    assert ei.value.code == 4 and "dof" in str(ei.value)


def test_destroy_waits_for_the_plans_own_streams_only(engine):
    """a stream held busy for 400 ms by a stall kernel (another host thread's work) must not hold up one-shot plans on
    another stream: create -> set_problem -> optimize -> result -> destroy stays at its idle latency"""
    p = problems.wam_restarts(B=4, total_step=20, obs_check_inter=3, sdf="40")
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    mine, other = _stream(engine), _stream(engine)

    def one_shot():
        t0 = time.perf_counter()
        pl = engine.plan(r, s, p.setting, p.B)
        pl.set_problem(*_args(p), p.init)
        pl.optimize(stream=mine.value)
        res = pl.result()
        pl.close()
        return time.perf_counter() - t0, res

    one_shot()
    idle = min(one_shot()[0] for _ in range(5))
    ref = one_shot()[1]
    # (robot / SDF handles of earlier tests that the garbage collector happens to finalise inside the loop would call
    # hipFree, which waits for the whole device by itself: collect them now, keep the collector out of the loop)
    import gc
    gc.collect()
    gc.disable()
    try:
        tok = C.c_void_p()
        engine._ck(engine.lib.gpmp2mi_debug_stall_begin(other, 400, C.byref(tok)))
        t0 = time.perf_counter()
        busy = []
        while time.perf_counter() - t0 < 0.25:           # well inside the stall
            dt, res = one_shot()
            busy.append(dt)
            np.testing.assert_array_equal(res["traj"], ref["traj"])
        held = time.perf_counter() - t0
        engine._ck(engine.lib.gpmp2mi_debug_stall_release(tok))
    finally:
        gc.enable()
    assert held < 0.39, ("the loop itself outlasted the stall", [round(x * 1e3, 2) for x in busy if x > 2e-3], len(busy))
    assert len(busy) >= 5 and max(busy) < 0.1, (idle, busy)      # a device-wide wait would have cost up to 400 ms
    print(f"one-shot plan: {idle * 1e3:.2f} ms idle, median {np.median(busy) * 1e3:.2f} ms / max {max(busy) * 1e3:.2f} ms "
          f"beside a stalled stream ({len(busy)} calls)")


_TIMEOUT_SCRIPT = r"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ["GPMP2MI_WAIT_TIMEOUT_MS"] = "300"
from gpmp2_amd import engine as E, problems
eng = E.Engine()
p = problems.wam_restarts(B=4, total_step=20, obs_check_inter=3, sdf="40")
args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
good = eng.batch_optimize(r, s, p.setting, *args, p.init)
st = C.c_void_p()
eng._ck(eng.lib.gpmp2mi_debug_stream_create(C.byref(st)))
def counts():
    v = [C.c_long() for _ in range(5)]
    eng.lib.gpmp2mi_debug_resource_counts(*[C.byref(x) for x in v])
    return [x.value for x in v]
pl = eng.plan(r, s, p.setting, p.B)
pl.set_problem(*args, p.init)
tok = C.c_void_p()
eng._ck(eng.lib.gpmp2mi_debug_stall_begin(st, 4000, C.byref(tok)))
t0 = time.perf_counter()
try:
    pl.optimize(stream=st.value)
    print("FAIL: optimize returned although its stream is stalled"); sys.exit(1)
except E.Gpmp2miError as e:
    dt = time.perf_counter() - t0
    assert e.code == 6 and "timed out" in str(e), e
    assert 0.25 < dt < 2.0, dt
for call in (lambda: pl.optimize(stream=st.value), pl.result, lambda: pl.set_problem(*args, p.init)):
    t1 = time.perf_counter()
    try:
        call(); print("FAIL: a poisoned plan accepted work"); sys.exit(1)
    except E.Gpmp2miError as e:
        assert e.code == 6 and time.perf_counter() - t1 < 0.05, e
before = counts()
t1 = time.perf_counter()
pl.close()
assert time.perf_counter() - t1 < 0.05, "destroy waited for the hung stream"
after = counts()
assert after[4] == before[4] + 1 and after[1] == before[1] and after[3] == before[3], (before, after)   # leaked, not pooled
assert time.perf_counter() - t0 < 3.0                # all of this happened while the stall kernel was still running
eng._ck(eng.lib.gpmp2mi_debug_stall_release(tok))    # the stream drains: the poisoned plan's passes run into leaked memory
pl2 = eng.plan(r, s, p.setting, p.B)
pl2.set_problem(*args, p.init)
pl2.optimize(stream=st.value)
res = pl2.result()
assert np.array_equal(res["traj"], good["traj"]) and list(res["iters"]) == list(good["iters"])
print("timeout path ok: optimize gave up after %.2f s" % dt)
"""


def test_timed_out_pass_poisons_the_plan_and_nothing_hangs():
    """in a process of its own: the pass driver really runs into GPMP2MI_WAIT_TIMEOUT_MS on a stalled stream"""
    out = subprocess.run([sys.executable, "-c", _TIMEOUT_SCRIPT.format(root=ROOT)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "timeout path ok" in out.stdout
