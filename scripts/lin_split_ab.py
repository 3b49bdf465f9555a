import os, sys, json, subprocess
# A/B the lane-split linearize kernel at several batch sizes (kernel avg ms from the library's HIP-event timer)
for B in (1, 8, 64):
    for split in ("0", "1"):
        env = dict(os.environ, GPMP2MI_LIN_SPLIT=split)
        out = subprocess.run([sys.executable, "bench.py", "--batch", str(B), "--steps", "10", "--warmup", "2", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(f"B={B:4d} split={split} traj/s={d['value']:10.0f} ms/step={d['ms_per_step']:.3f} " +
              " ".join(f"{n}={v['avg_ms']*1e3:.1f}us" for n, v in k.items()))
