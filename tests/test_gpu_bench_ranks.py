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
    env.update(GPMP2MI_BENCH_REHEARSAL="1", GPMP2MI_BENCH_DUMP=dump, OMP_NUM_THREADS="2")
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
