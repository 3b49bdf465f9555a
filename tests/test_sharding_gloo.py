"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): the batch is sharded by trajectory,
every rank solves its own slice with no data-path collective, one all-gather at the end; the
gathered batch must equal the single-process solve.  The compute stand-in here is the CPU oracle
(allowed in tests only); on GPUs bench.py runs the HIP plan in its place with the same sharding and
gather code (gpmp2_amd/sharding.py)."""
import os
import subprocess
import sys

import numpy as np

from gpmp2_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["G2_ROOT"]); sys.path.insert(0, os.path.join(os.environ["G2_ROOT"], "tests"))
from gpmp2_amd import problems, sharding
from oracle import Oracle
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
B = 5                                   # uneven on purpose: 3 + 2
p = problems.wam_restarts(B=B, total_step=8, obs_check_inter=2, opt="GN", sdf="24")
local = sharding.shard_problem(dict(sc=p.start_conf, sv=p.start_vel, ec=p.end_conf, ev=p.end_vel, init=p.init), world, rank)
orc = Oracle()
r, s = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
res = orc.batch_optimize(r, s, p.setting, local["sc"], local["sv"], local["ec"], local["ev"], local["init"])
traj = sharding.gather_results(torch.from_numpy(res["traj"]), B)
iters = sharding.gather_results(torch.from_numpy(res["iters"]), B)
if rank == 0:
    np.savez(os.environ["G2_OUT"], traj=traj.numpy(), iters=iters.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_partitions_exactly():
    for total in (1, 5, 64, 1024, 7):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_gather_matches_single_process(tmp_path, oracle):
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, G2_ROOT=ROOT, G2_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531",
               OMP_NUM_THREADS="1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)], env=env,
                          timeout=600)
    got = np.load(out)
    from gpmp2_amd import problems
    p = problems.wam_restarts(B=5, total_step=8, obs_check_inter=2, opt="GN", sdf="24")
    r, s = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(r, s, p.setting, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    np.testing.assert_array_equal(got["iters"], ref["iters"])
    np.testing.assert_array_equal(got["traj"], ref["traj"])


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts 2 ranks itself (child torchrun, parent never touches
    the GPU).  Without GPUs the ranks refuse loudly instead of reporting a 1-GPU number as n_gpus 2."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs to observe the refusal")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPMP2MI_BENCH_REHEARSAL")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode != 0
    assert out.stdout.strip() == ""                       # no JSON line from a run that did not happen
    assert out.stderr.count("2 ranks requested but only") >= 1
