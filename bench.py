#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GPMP2 linearize-and-solve hot path on MI355X.

Workload (BASELINE.json metric, config 3 "WAMFactorGraphExample"): 7-DOF WAM arm, 100 time steps x 5
GP-interpolated sub-steps, 200^3 fp64 SDF ("Synth200"), 64 random-init restarts batched per GPU,
Gauss-Newton run to the reference's stopping rule (rel 1e-2 / abs 1e-5 / max 50).

A "step" = one pass of the hot path over one batch: all 64 restarts optimised to tolerance, inputs
already resident in HBM.  One process per GPU (torch.distributed / RCCL); the path shards by
trajectory (weak scaling: 64 restarts per GPU, no data-path collective), the only collective is the
final all-gather of the results, which is inside the timed step.

Launching: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N
ranks itself (a child `python -m torch.distributed.run`, never an exec; the parent does not touch the
GPU), relays rank 0's JSON line and exits with the children's status.  Under an external torchrun
(WORLD_SIZE set) the process is a rank.

Prints ONE JSON line (see the driver contract): metric trajectories/sec, plus
  roofline      -- whole Gauss-Newton pass (linearize + assemble + solve; the finish of a step is part of the next
                   linearization for fixed-base arms, a kernel of its own otherwise), SURVEY.md 8(d)'s
                   638 048 algorithmic bytes per trajectory-iteration / the summed average launch
                   durations of the pass's kernels (HIP events on the launch stream, inside the
                   library) vs the 8 TB/s HBM peak; `path_frac` = the same bytes over the wall time of
                   the step; `mfma_frac` = SURVEY's 5.5 MFLOP per trajectory-iteration over the wall
                   time vs the fp64 matrix peak; per-kernel averages next to it
  cpu_baseline  -- the CPU oracle (a port of the reference algorithm; the reference itself needs
                   GTSAM and cannot be built here) timed on this box's host cores on a bounded
                   sample: all host threads, and a single thread (what the reference is)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_TRAJ_ITER = 638_048       # SURVEY.md section 8(d) contract figure
ALGO_FLOP_PER_TRAJ_ITER = 5.5e6          # SURVEY.md section 8(d), structured count
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6                  # MI355X_MICROARCH.md: fp64 vector = matrix peak
STATUS_NAMES = {0: "converged", 1: "max_iter", 2: "rolled_back", 3: "not_spd", 4: "already_optimal"}


def parse(argv=None):
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
    ap.add_argument("--no-variants", action="store_true", help="skip the LM / Dogleg sub-results")
    ap.add_argument("--cpu-sample", type=int, default=0, help="restarts in the CPU sample (0 = auto)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """--gpus N > 1 without WORLD_SIZE: start the N ranks as a CHILD torchrun (this process has not
    initialised the GPU and never execs), relay rank 0's JSON line, return the children's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------ CPU side
def cpu_baseline(p, sample, threads, budget_s=6.0):
    """Times the oracle (tests/oracle.py -> oracle/liboracle.so) on `sample` restarts of the same
    workload, repeated until about `budget_s` seconds of wall time have been spent (a bounded sample of
    CPU work).  This is the ONLY place bench.py touches the oracle; it is never the thing measured as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import Oracle
    orc = Oracle()
    ro, so = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    sel = slice(0, sample)
    reps, dt, res = 0, 0.0, None
    while dt < budget_s and reps < 64:
        t0 = time.perf_counter()
        res = orc.batch_optimize(ro, so, p.setting, p.start_conf[sel], p.start_vel[sel], p.end_conf[sel],
                                 p.end_vel[sel], p.init[sel], nthreads=threads)
        dt += time.perf_counter() - t0
        reps += 1
    return dict(value=sample * reps / dt, unit="trajectories/sec", cores=threads, kind="port",
                sample=f"first {sample} of the {p.B} restarts, run to tolerance, {threads} OpenMP thread(s) "
                       f"over trajectories, {reps} repetition(s), {dt:.1f} s wall",
                seconds=dt, repetitions=reps, traj_iters_per_sec=float(np.sum(res["iters"] + 1)) * reps / dt), res


def parity_vs_oracle(ref, gpu, sample):
    """iteration counts / status / final error / trajectory of the sampled restarts, GPU vs oracle"""
    gt, gi, gs, gf = gpu
    fe_rel = np.abs(gf[:sample] - ref["final_error"]) / np.maximum(np.abs(ref["final_error"]), 1e-300)
    return dict(iters_match_gpu=bool(np.array_equal(ref["iters"], gi[:sample])),
                status_match_gpu=bool(np.array_equal(ref["status"], gs[:sample])),
                traj_max_abs_diff=float(np.max(np.abs(gt[:sample] - ref["traj"]))),
                final_error_max_rel=float(np.max(fe_rel)),
                restarts_compared=int(sample))


def head_commit():
    """the commit this tree is at: git when .git is present, else the file scripts/evidence.sh's caller writes (the GPU
    box gets the tree without .git)"""
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if out.returncode == 0 and out.stdout.strip():
            return out.stdout.strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, ".evidence_commit")).read().strip() or None
    except Exception:
        return None


def kernel_sources_digest():
    """sha1 over the device sources (gpmp2_amd/csrc/*.hip and *.h without api.hip, the host driver); scripts/pmc_to_json.py
    stores the same digest in the traffic profile it writes"""
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "gpmp2_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if (f.endswith(".hip") or f.endswith(".h")) and f != "api.hip":
            h.update(f.encode())
            h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def kernels_changed_since(commit, here, digest=None):
    """True when the kernel sources differ between the tree a stored profile was taken from and this one.  The profile
    carries a digest of the device sources (exact, and it needs no git -- the GPU box gets the tree without .git); older
    profiles only name their commit: then git decides where it is present, else only an identical commit id counts."""
    if digest:
        return digest != kernel_sources_digest()
    if not commit or commit == "unknown":
        return True
    if here and (commit.startswith(here) or here.startswith(commit)):
        return False
    try:
        out = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", commit, "HEAD", "--", "gpmp2_amd/csrc"], capture_output=True, timeout=20)
        if out.returncode in (0, 1):
            return out.returncode == 1
    except Exception:
        pass
    return True


def pmc_traffic():
    """HBM bytes per launch and kernel from the newest committed PMC summary under profiles/ (rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE collected in separate passes of this same command, gfx950
    correction applied; see profiles/README.md).  A STORED profile, not a live measurement: the summary records the
    commit it was taken at, and a line produced from another commit says so (`traffic_stale`)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))   # the B = 64 GN profile of each round
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        out = {}
        for name, v in d.get("kernels", {}).items():
            if name.startswith("k_"):
                key = name.split("<")[0][2:]
                key = {"linearize_arm": "linearize"}.get(key, key)   # the fixed-base-arm form of the same pass stage
                out[key] = out.get(key, 0.0) + v["hbm_bytes_per_launch_corrected"]
        if out:
            return out, os.path.basename(f), d.get("commit"), d.get("sources_digest")
    return None, None, None, None


def status_hist(status):
    return {STATUS_NAMES.get(int(k), str(int(k))): int(v) for k, v in zip(*np.unique(status, return_counts=True))}


# ------------------------------------------------------------------------------------------ a rank
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world > 1):   # a bare external torchrun may omit --gpus
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per requested GPU", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist

    # GPMP2MI_BENCH_REHEARSAL=1: exercise the N > 1 control flow on a ONE-GPU box (all ranks on device 0, gloo
    # collectives on host copies).  Only for checking the multi-rank code path; its numbers mean nothing.
    rehearsal = os.environ.get("GPMP2MI_BENCH_REHEARSAL") == "1"
    if not rehearsal and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks requested but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = "cpu" if rehearsal else dev      # where the collectives' tensors live

    from gpmp2_amd import engine, problems, sharding

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

    def make_step(pl, out_t, gath, Bp):
        """one step = the whole batch optimised; for world > 1 the final gather of the results (the only collective on the
        path) is part of it.  HIP events around the gather give its own time."""
        gather_events = []

        def step(record=False):
            pl.optimize(stream=stream)
            if world > 1:
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                eng._ck(eng.lib.gpmp2mi_plan_get_result_dev(pl.h.ptr, C.c_void_p(out_t.data_ptr()), None, None,
                                                            None, C.c_void_p(stream)))
                gath[...] = (sharding.gather_results(out_t.cpu(), Bp * world).to(dev) if rehearsal
                             else sharding.gather_results(out_t, Bp * world))
                if record:
                    e1.record()
                    gather_events.append((e0, e1))
        return step, gather_events

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(pl, step, steps, warmup, events=True):
        """W untimed steps, then exactly K steps between two fences; returns (local wall seconds, per-kernel sums, sampled)"""
        for _ in range(warmup):
            step()
        # Per-kernel HIP events (one per kernel boundary, on the launch stream, inside the library) are recorded
        # on every EVENT_EVERY-th step of the timed region: each event is a barrier packet between two dependent
        # kernels (~3.4 us, 40 of them per step = 12 % of a 1 ms step when recorded on every step), so sampling keeps
        # the instrumentation from distorting `value` while the averages still come from >= 50 launches per kernel.
        EVENT_EVERY = 1 if os.environ.get("GPMP2MI_BENCH_EVENTS_EVERY_STEP") else 4
        kern_, sampled_ = {}, 0
        fence()
        t0 = time.perf_counter()
        for it in range(steps):
            timed = events and (it % EVENT_EVERY == 0)
            pl.enable_timing(timed)
            step(record=True)
            if timed:
                sampled_ += 1
                for k, v in pl.timing().items():
                    a = kern_.setdefault(k, dict(ms=0.0, launches=0))
                    a["ms"] += v["ms"]
                    a["launches"] += v["launches"]
        fence()
        return time.perf_counter() - t0, kern_, sampled_

    def over_ranks(dt_local, gather_events):
        """MAX over ranks of the wall time (the contract's clock), every rank's own time, mean gather time per step"""
        gms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else 0.0
        if world == 1:
            return dt_local, [dt_local], gms
        mine = torch.tensor([dt_local, gms], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu().numpy()
        return float(every[:, 0].max()), [float(x) for x in every[:, 0]], float(every[:, 1].max())

    step, gather_events = make_step(plan, out_traj, gathered, B)
    dt_local, kern, sampled = timed_region(plan, step, args.steps, args.warmup)
    dt, rank_dts, gather_ms = over_ranks(dt_local, gather_events)
    iters, status, ferr = plan.result_counts()
    passes_local = int(np.sum(iters + 1))                 # linearize+solve passes per step on this rank
    hist_local = np.bincount(status, minlength=8)[:8].astype(np.int64)
    if world > 1:
        cnt = torch.tensor([passes_local] + hist_local.tolist(), dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        cnt = cnt.cpu().numpy()
        passes_all, hist_all = int(cnt[0]), cnt[1:]
        it_all = sharding.gather_results(torch.from_numpy(iters.astype(np.int64)).to(cdev), B * world).cpu().numpy()
    else:
        passes_all, hist_all, it_all = passes_local, hist_local, iters

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
        plan.optimize(stream=stream)

    # LM / Dogleg on the same restarts (the optimizers the reference's WAM scripts use), measured in the same
    # run OUTSIDE the timed Gauss-Newton region
    variants = None
    if world == 1 and args.workload == "restarts" and args.opt == "GN" and not args.no_variants:
        variants = {}
        for opt in ("LM", "DOGLEG"):
            pv = problems.wam_restarts(B=B, opt=opt)
            plv = eng.plan(r, s, pv.setting, B)
            plv.set_problem_dev(*[t.data_ptr() for t in t_in], stream=stream)
            plv.optimize(stream=stream)
            reps = 3
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(reps):
                plv.optimize(stream=stream)
            torch.cuda.synchronize()
            tv = time.perf_counter() - tv
            vi, vs, _ = plv.result_counts()
            variants[opt.lower()] = dict(value=B * reps / tv, unit="trajectories/sec", ms_per_step=tv / reps * 1e3,
                                         iters=dict(min=int(vi.min()), median=float(np.median(vi)), max=int(vi.max())),
                                         status_counts=status_hist(vs))
            plv.close()

    # BASELINE config 4 beside the headline (it is the config BASELINE.json quotes for 8 GPUs): 128 receding-horizon
    # windows per GPU, warm-started from the solved restart-0 trajectory, 3 fixed Gauss-Newton iterations, the final
    # gather inside the timed step; measured in the same run OUTSIDE the timed region of `value`.
    windows = None
    if args.workload == "restarts" and args.opt == "GN" and not args.no_variants:
        WB = int(os.environ.get("GPMP2MI_BENCH_WINDOWS", "128"))
        base = problems.wam_restarts(B=1, opt="GN")
        sol = eng.batch_optimize(r, s, base.setting, base.start_conf, base.start_vel, base.end_conf, base.end_vel,
                                 base.init)["traj"][0]
        pw = problems.wam_windows(sol, B=WB * world)
        wlo, whi = sharding.shard_range(WB * world, world, rank)
        plw = eng.plan(r, s, pw.setting, WB)
        w_in = [torch.from_numpy(np.ascontiguousarray(a[wlo:whi])).to(dev) for a in
                (pw.start_conf, pw.start_vel, pw.end_conf, pw.end_vel, pw.init)]
        torch.cuda.synchronize()
        plw.set_problem_dev(*[t.data_ptr() for t in w_in], stream=stream)
        w_out = torch.empty((WB, N + 1, 2 * D), dtype=torch.float64, device=dev)
        w_gath = torch.empty((world * WB, N + 1, 2 * D), dtype=torch.float64, device=dev) if world > 1 else None
        wstep, wev = make_step(plw, w_out, w_gath, WB)
        wsteps = max(5, min(args.steps, 20))
        wdt_local, _, _ = timed_region(plw, wstep, wsteps, 2, events=False)
        wdt, wrank, wgather = over_ranks(wdt_local, wev)
        wi, ws, _ = plw.result_counts()
        windows = dict(value=WB * world * wsteps / wdt, unit="windows/sec", ms_per_step=wdt / wsteps * 1e3, steps=wsteps,
                       config=dict(workload=("WAMReplannerExample receding horizon: 7-DOF WAM, 100 steps x 5 GP-interp, 200^3 "
                                             f"SDF, {WB} warm-started windows per GPU, 3 fixed GN iterations"),
                                   windows_per_gpu=WB, total_windows=WB * world, fixed_iterations=int(pw.setting.fixed_iterations)),
                       iters=dict(min=int(wi.min()), max=int(wi.max())),
                       rank_ms_per_step=[x / wsteps * 1e3 for x in wrank],
                       rank_ms_spread=(max(wrank) - min(wrank)) / wsteps * 1e3,
                       gather_ms_per_step=wgather)
        windows_traj = (w_gath if world > 1 else torch.from_numpy(plw.result()["traj"])).cpu().numpy()
        plw.close()

    dump = os.environ.get("GPMP2MI_BENCH_DUMP")     # tests: the gathered batch of the last step, for comparison
    if dump and rank == 0:                          # against a single-rank solve
        np.savez(dump, traj=(gathered if world > 1 else torch.from_numpy(plan.result()["traj"])).cpu().numpy(),
                 iters=it_all, **({"windows_traj": windows_traj} if windows else {}))

    if rank == 0:
        total_traj = B * world * args.steps
        ms_per_step = dt / args.steps * 1e3
        roof = None
        if kern:
            per = {k: dict(avg_ms=v["ms"] / v["launches"], launches_per_step=v["launches"] / sampled) for k, v in kern.items()}
            dom = max(kern, key=lambda k: kern[k]["ms"])
            launches_per_step = per[dom]["launches_per_step"]          # = passes enqueued per step
            units_per_launch = passes_local / launches_per_step          # trajectory-iterations per pass (this GPU)
            pass_ms = sum(v["avg_ms"] * v["launches_per_step"] for v in per.values()) / launches_per_step
            bytes_per_launch = ALGO_BYTES_PER_TRAJ_ITER * units_per_launch
            achieved = bytes_per_launch / (pass_ms * 1e-3) / 1e9
            path_gbs = ALGO_BYTES_PER_TRAJ_ITER * passes_local / (ms_per_step * 1e-3) / 1e9
            path_tflops = ALGO_FLOP_PER_TRAJ_ITER * passes_local / (ms_per_step * 1e-3) / 1e12
            tr, tr_src, tr_commit, tr_digest = pmc_traffic() if (args.workload == "restarts" and B == 64 and args.opt == "GN") else (None, None, None, None)
            traffic = sum(tr.get(k, 0.0) for k in per) if tr else None
            here = head_commit()
            stale = bool(tr) and kernels_changed_since(tr_commit, here, tr_digest)
            if stale:
                print(f"bench.py: the stored traffic profile {tr_src} was taken at commit {tr_commit}, this tree is at {here}: "
                      "roofline.traffic and the per-kernel own-bytes fractions describe that commit (re-run scripts/evidence.sh)",
                      file=sys.stderr)
            if tr:      # every kernel judged on its OWN measured traffic: PMC bytes per launch / its mean duration / peak
                for k, v in per.items():
                    if k in tr and v["avg_ms"] > 0:
                        v["own_bytes_per_launch"] = tr[k]
                        v["own_gbs"] = tr[k] / (v["avg_ms"] * 1e-3) / 1e9
                        v["own_frac"] = v["own_gbs"] / HBM_PEAK_GBS
            roof = dict(bound="hbm", kernel="pass: " + " + ".join(per.keys()), dominant_kernel=dom,
                        achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=traffic, traffic_source=(tr_src + " (stored rocprofv3 PMC profile, not live)") if tr else None,
                        traffic_commit=tr_commit if tr else None, traffic_stale=stale if tr else None,
                        algorithmic_bytes_per_launch=bytes_per_launch, avg_launch_ms=pass_ms,
                        launches_per_step=launches_per_step, units_per_launch=units_per_launch,
                        path_achieved=path_gbs, path_frac=path_gbs / HBM_PEAK_GBS,
                        mfma_achieved_tflops=path_tflops, mfma_peak_tflops=FP64_PEAK_TFLOPS,
                        mfma_frac=path_tflops / FP64_PEAK_TFLOPS,
                        note=("achieved = SURVEY 8(d) bytes per trajectory-iteration x trajectory-iterations per pass / "
                              "summed mean launch time of the pass's kernels (HIP events, every 4th step); path_* = the same "
                              "bytes / flops over the wall time of a step (per GPU)"),
                        event_sampled_steps=sampled, kernels=per)
        try:
            baseline_metric = json.load(open(os.path.join(ROOT, "BASELINE.json"))).get("metric")
        except Exception:
            baseline_metric = None
        hist = {STATUS_NAMES.get(k, str(k)): int(v) for k, v in enumerate(hist_all) if v}
        n_all = int(sum(hist_all))
        out = dict(metric="trajectories/sec", baseline_metric=baseline_metric, value=total_traj / dt,
                   unit="trajectories/sec", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=("WAMFactorGraphExample: 7-DOF WAM, 100 steps x 5 GP-interp, 200^3 SDF "
                                         f"(Synth200), {B} random-init restarts per GPU, {args.opt} to tolerance")
                               if args.workload == "restarts" else
                               ("WAMReplannerExample receding horizon: 7-DOF WAM, 100 steps x 5 GP-interp, 200^3 SDF, "
                                f"{B} warm-started windows per GPU, 3 fixed GN iterations"),
                               restarts_per_gpu=B, total_restarts=B * world, total_step=N,
                               obs_check_inter=p.setting.obs_check_inter,
                               optimizer=args.opt, parallelism=f"trajectory-sharded x{world}"),
                   gn_iters_to_tol=dict(min=int(it_all.min()), median=float(np.median(it_all)), max=int(it_all.max())),
                   traj_iters_per_sec=passes_all * args.steps / dt,
                   status_counts=hist,
                   rolled_back_share=float(hist.get("rolled_back", 0)) / max(n_all, 1),
                   status_note=("rolled_back = the last iteration increased the error, so gpmp2::optimize returns the values "
                                "before it (planner/BatchTrajOptimizer.cpp:297-307); iters-to-tol counts that last iteration"),
                   roofline=roof)
        if host_rate is not None:
            out["pcie_inclusive_value"] = host_rate
        if variants:
            out["variants"] = variants
        if windows:
            out["windows"] = windows
        out["rank_ms_per_step"] = [x / args.steps * 1e3 for x in rank_dts]
        out["rank_ms_spread"] = (max(rank_dts) - min(rank_dts)) / args.steps * 1e3
        out["gather_ms_per_step"] = gather_ms
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(os.cpu_count() or 1, 64))
            sample = args.cpu_sample or min(B, max(8, threads))
            sample = min(sample, 256)
            cb, ref = cpu_baseline(p, sample, threads)
            # parity gate before the timing counts: every sampled restart against the oracle
            res = plan.result()
            cb.update(parity_vs_oracle(ref, (res["traj"], res["iters"], res["status"], res["final_error"]), sample))
            c1, _ = cpu_baseline(p, sample, 1)  # what the reference is: one thread
            cb["single_thread"] = dict(value=c1["value"], unit=c1["unit"], cores=1, sample=c1["sample"], seconds=c1["seconds"])
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
