"""`python bench.py --gpus 2` as the driver invokes it (no WORLD_SIZE in the environment): the parent starts the
ranks itself, the HIP plan runs in each rank, the gathered batch equals a single-rank solve of the same restarts.
On a one-GPU box the two ranks share device 0 (GPMP2MI_BENCH_REHEARSAL=1: gloo collectives on host copies); the
sharding, barrier, gather, MAX-reduced step time and rank-0 JSON line are the code the RCCL run uses."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gpmp2_amd import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_ranks_and_gathers_the_hip_results(engine, tmp_path):
    dump = str(tmp_path / "gathered.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GPMP2MI_BENCH_REHEARSAL="1", GPMP2MI_BENCH_DUMP=dump, OMP_NUM_THREADS="2", GPMP2MI_BENCH_WINDOWS="6")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["restarts_per_gpu"] == 4 and line["config"]["total_restarts"] == 8
    assert line["scaling"] == "weak" and line["value"] > 0
    got = np.load(dump)
    p = problems.wam_restarts(B=8)                      # the 8 restarts the two ranks shared 4 + 4
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = engine.batch_optimize(r, s, p.setting, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    np.testing.assert_array_equal(got["iters"], ref["iters"])
    np.testing.assert_allclose(got["traj"], ref["traj"], rtol=0, atol=1e-12)
    assert sum(line["status_counts"].values()) == 8
    # per-rank clocks and the gather's own time ride on the line, so that a scaling run can be read in one shot
    assert len(line["rank_ms_per_step"]) == 2 and line["rank_ms_spread"] >= 0 and line["gather_ms_per_step"] > 0
    assert max(line["rank_ms_per_step"]) == pytest.approx(line["ms_per_step"], rel=1e-6)
    # ... and so does BASELINE config 4 (here 6 windows per rank instead of 128): receding-horizon windows sharded like the
    # restarts, 3 fixed Gauss-Newton iterations, gather inside the timed step
    w = line["windows"]
    assert w["unit"] == "windows/sec" and w["value"] > 0
    assert w["config"]["windows_per_gpu"] == 6 and w["config"]["total_windows"] == 12 and w["config"]["fixed_iterations"] == 3
    assert w["iters"] == {"min": 3, "max": 3}
    assert len(w["rank_ms_per_step"]) == 2 and w["gather_ms_per_step"] > 0
    base = problems.wam_restarts(B=1)
    sol = engine.batch_optimize(r, s, base.setting, base.start_conf, base.start_vel, base.end_conf, base.end_vel, base.init)["traj"][0]
    pw = problems.wam_windows(sol, B=12)
    wref = engine.batch_optimize(r, s, pw.setting, pw.start_conf, pw.start_vel, pw.end_conf, pw.end_vel, pw.init)
    np.testing.assert_allclose(got["windows_traj"], wref["traj"], rtol=0, atol=1e-12)
