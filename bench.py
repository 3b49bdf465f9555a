#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GPMP2 linearize-and-solve hot path on MI355X.

Workload (BASELINE.json metric, config 3 "WAMFactorGraphExample"): 7-DOF WAM arm, 100 time steps x 5
GP-interpolated sub-steps, 200^3 fp64 SDF ("Synth200"), 64 random-init restarts batched per GPU,
Gauss-Newton run to the reference's stopping rule (rel 1e-2 / abs 1e-5 / max 50).

A "step" = one pass of the hot path over one batch: all 64 restarts optimised to tolerance, inputs
already resident in HBM.  One process per GPU (torch.distributed / RCCL); the path shards by
trajectory (weak scaling: 64 restarts per GPU, no data-path collective), the only collective is the
final all-gather of the results, which is inside the timed step.

Prints ONE JSON line (see the driver contract): metric trajectories/sec, plus
  roofline      -- dominant kernel, algorithmic bytes (638 048 B per trajectory-iteration,
                   SURVEY.md 8d) / its average launch duration measured with HIP events on the launch
                   stream inside the library, vs the 8 TB/s HBM peak
  cpu_baseline  -- the CPU oracle (a port of the reference algorithm; the reference itself needs
                   GTSAM and cannot be built here) timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_TRAJ_ITER = 638_048       # SURVEY.md section 8(d) contract figure
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="restarts per GPU (BASELINE config 3: 64)")
    ap.add_argument("--opt", default="GN", choices=["GN", "LM", "DOGLEG"])
    ap.add_argument("--workload", default="restarts", choices=["restarts", "windows"],
                    help="restarts = BASELINE config 3 (default, the metric's config); windows = config 4 "
                         "(receding-horizon windows warm-started from the solved trajectory, 3 fixed GN iterations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="restarts in the CPU sample (0 = auto)")
    return ap.parse_args()


def cpu_baseline(p, sample, threads):
    """Times the oracle (tests/oracle.py -> oracle/liboracle.so) on `sample` restarts of the same
    workload.  This is the ONLY place bench.py touches the oracle; it is never the thing measured
    as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import Oracle
    orc = Oracle()
    ro, so = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    sel = slice(0, sample)
    t0 = time.perf_counter()
    res = orc.batch_optimize(ro, so, p.setting, p.start_conf[sel], p.start_vel[sel], p.end_conf[sel],
                             p.end_vel[sel], p.init[sel], nthreads=threads)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="trajectories/sec", cores=threads, kind="port",
                sample=f"first {sample} of the {p.B} restarts, run to tolerance, {threads} OpenMP threads "
                       f"over trajectories, {dt:.1f} s wall",
                seconds=dt, traj_iters_per_sec=float(np.sum(res["iters"] + 1)) / dt), res


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of `kernel_prefix` from the newest committed PMC summary under profiles/
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE collected in separate passes of this same
    command, gfx950 correction applied; see profiles/README.md).  None when no summary matches."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix):
                best = (v["hbm_bytes_per_launch_corrected"], os.path.basename(f))
    return best


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    # GPMP2MI_BENCH_REHEARSAL=1: exercise the N > 1 control flow on a ONE-GPU box (all ranks on device 0, gloo
    # collectives on host copies).  Only for checking the multi-rank code path; its numbers mean nothing.
    rehearsal = os.environ.get("GPMP2MI_BENCH_REHEARSAL") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gpmp2_amd import engine, problems

    from gpmp2_amd import sharding

    B = args.batch
    eng = engine.Engine()
    if args.workload == "restarts":
        p = problems.wam_restarts(B=B * world, opt=args.opt)     # weak scaling: B restarts per rank
    else:
        base = problems.wam_restarts(B=1, opt="GN")
        r0, s0 = eng.robot(base.model), eng.sdf(base.sdf_origin, base.sdf_cell, base.sdf_data)
        sol = eng.batch_optimize(r0, s0, base.setting, base.start_conf, base.start_vel, base.end_conf,
                                 base.end_vel, base.init)["traj"][0]
        p = problems.wam_windows(sol, B=B * world)
    lo, hi = sharding.shard_range(B * world, world, rank)
    r = eng.robot(p.model)
    s = eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    plan = eng.plan(r, s, p.setting, B)
    t_in = [torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev) for a in
            (p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)]
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    plan.set_problem_dev(*[t.data_ptr() for t in t_in], stream=stream)
    N, D = p.setting.total_step, p.setting.dof
    out_traj = torch.empty((B, N + 1, 2 * D), dtype=torch.float64, device=dev)
    gathered = torch.empty((world * B, N + 1, 2 * D), dtype=torch.float64, device=dev) if world > 1 else None

    import ctypes as C

    def step():
        plan.optimize(stream=stream)
        if world > 1:     # final gather of the results: the only collective on the path
            eng._ck(eng.lib.gpmp2mi_plan_get_result_dev(plan.h.ptr, C.c_void_p(out_traj.data_ptr()), None, None,
                                                        None, C.c_void_p(stream)))
            gathered[...] = (sharding.gather_results(out_traj.cpu(), B * world).to(dev) if rehearsal
                             else sharding.gather_results(out_traj, B * world))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Per-kernel HIP events (one per kernel boundary, on the launch stream, inside the library) are recorded
    # on every EVENT_EVERY-th step of the timed region: each event is a barrier packet between two dependent
    # kernels (~3.4 us, 40 of them per step = 12 % of a 1 ms step when recorded on every step), so sampling keeps
    # the instrumentation from distorting `value` while the averages still come from >= 50 launches per kernel.
    EVENT_EVERY = 1 if os.environ.get("GPMP2MI_BENCH_EVENTS_EVERY_STEP") else 4
    kern = {}
    sampled = 0
    fence()
    t0 = time.perf_counter()
    for it in range(args.steps):
        timed = (it % EVENT_EVERY == 0)
        plan.enable_timing(timed)
        step()
        if timed:
            sampled += 1
            for k, v in plan.timing().items():
                a = kern.setdefault(k, dict(ms=0.0, launches=0))
                a["ms"] += v["ms"]
                a["launches"] += v["launches"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    iters, status, ferr = plan.result_counts()

    # the boundary handing over HOST buffers (gpmp2mi_plan_set_problem / get_result): same work plus the
    # PCIe copies, reported next to `value`, never as `value`
    host_rate = None
    if world == 1:
        plan.enable_timing(False)
        h_in = [np.ascontiguousarray(a[lo:hi]) for a in (p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)]
        reps = max(3, min(args.steps, 10))
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(reps):
            plan.set_problem(*h_in)
            plan.optimize(stream=stream)
            plan.result()
        host_rate = B * reps / (time.perf_counter() - th)
        plan.set_problem_dev(*[t.data_ptr() for t in t_in], stream=stream)

    if rank == 0:
        total_traj = B * world * args.steps
        ms_per_step = dt / args.steps * 1e3
        passes = int(np.sum(iters + 1))                    # linearize+solve passes per step (this rank)
        dom = max(kern, key=lambda k: kern[k]["ms"]) if kern else None
        roof = None
        if dom:
            launches_per_step = kern[dom]["launches"] / sampled
            avg_ms = kern[dom]["ms"] / kern[dom]["launches"]
            units_per_launch = passes / launches_per_step      # trajectory-iterations per launch
            achieved = ALGO_BYTES_PER_TRAJ_ITER * units_per_launch / (avg_ms * 1e-3) / 1e9
            tr = pmc_traffic("k_" + dom) if (args.workload == "restarts" and B == 64 and args.opt == "GN") else None
            roof = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=(tr[0] if tr else None),
                        traffic_source=(tr[1] if tr else None),
                        algorithmic_bytes_per_launch=ALGO_BYTES_PER_TRAJ_ITER * units_per_launch,
                        avg_launch_ms=avg_ms, launches_per_step=launches_per_step,
                        units_per_launch=units_per_launch,
                        event_sampled_steps=sampled,
                        kernels={k: dict(avg_ms=v["ms"] / v["launches"], launches_per_step=v["launches"] / sampled)
                                 for k, v in kern.items()})
        try:
            baseline_metric = json.load(open(os.path.join(ROOT, "BASELINE.json"))).get("metric")
        except Exception:
            baseline_metric = None
        out = dict(metric="trajectories/sec", baseline_metric=baseline_metric, value=total_traj / dt,
                   unit="trajectories/sec", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=("WAMFactorGraphExample: 7-DOF WAM, 100 steps x 5 GP-interp, 200^3 SDF "
                                         f"(Synth200), {B} random-init restarts per GPU, {args.opt} to tolerance")
                               if args.workload == "restarts" else
                               ("WAMReplannerExample receding horizon: 7-DOF WAM, 100 steps x 5 GP-interp, 200^3 SDF, "
                                f"{B} warm-started windows per GPU, 3 fixed GN iterations"),
                               restarts_per_gpu=B, total_step=N, obs_check_inter=p.setting.obs_check_inter,
                               optimizer=args.opt, parallelism=f"trajectory-sharded x{world}"),
                   gn_iters_to_tol=dict(min=int(iters.min()), median=float(np.median(iters)), max=int(iters.max())),
                   traj_iters_per_sec=passes * world * args.steps / dt,
                   status_counts={int(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))},
                   roofline=roof)
        if host_rate is not None:
            out["pcie_inclusive_value"] = host_rate
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(os.cpu_count() or 1, 64))
            sample = args.cpu_sample or min(B, max(8, threads))
            sample = min(sample, 256)
            cb, ref = cpu_baseline(p, sample, threads)
            # parity gate before the timing counts: same iteration counts as the oracle on the sample
            cb["iters_match_gpu"] = bool(np.array_equal(ref["iters"], iters[:sample]))
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
