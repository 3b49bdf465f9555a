"""Build-time resource table of every shipped kernel: VGPRs, SGPRs, spills, scratch, LDS, occupancy, as the compiler
reports them (hipcc -Rpass-analysis=kernel-resource-usage on the product sources with the product flags).
usage: python scripts/resources.py > profiles/rNN_resources.txt        (compiles all six .hip files, a few minutes)"""
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpmp2_amd", "csrc")
FILES = ["api.hip", "sdf_kernels.hip", "factor_kernels.hip", "plan_kernels.hip", "cr_kernels.hip", "dense_kernels.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c"]
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]",
        "Occupancy [waves/SIMD]"]


def compile_one(f):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", "/dev/null"], cwd=SRC, capture_output=True, text=True)
    return f, out.stderr


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            if out.returncode == 0:
                return out.stdout.splitlines()
        except OSError:
            pass
    return names


def main():
    only = sys.argv[1:]          # optional substrings: print only kernels whose name contains one of them
    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        logs = dict(ex.map(compile_one, FILES))
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "?"
    print(f"# kernel resource usage, hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage; commit {head}")
    print(f"# {'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'LDS':>6s} {'occ':>4s}")
    for f in FILES:
        rows, cur = [], None
        for line in logs[f].splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = dict(name=m.group(1))
                rows.append(cur)
                continue
            if cur is None:
                continue
            for k in KEYS:
                m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
                if m:
                    cur[k] = int(m.group(1))
        rows = [r for r in rows if "VGPRs" in r]
        names = demangle([r["name"] for r in rows])
        print(f"## {f}")
        for r, nm in zip(rows, names):
            nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("g2::", "")
            if only and not any(o in nm for o in only):
                continue
            print(f"  {nm[:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('SGPRs Spill', 0):6d} "
                  f"{r.get('VGPRs Spill', 0):6d} {r.get('ScratchSize [bytes/lane]', 0):7d} {r.get('LDS Size [bytes/block]', 0):6d} "
                  f"{r.get('Occupancy [waves/SIMD]', 0):4d}")


if __name__ == "__main__":
    main()
