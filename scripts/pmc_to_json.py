"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/rNN_pmc_traffic.json: per kernel, mean KB per launch as reported and the gfx950-corrected HBM bytes
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM / rocprofv3 section).

usage: python scripts/pmc_to_json.py <fetch_dir> <write_dir> <out.json> [commit]"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_digest(root=ROOT):
    """sha1 over the device sources (gpmp2_amd/csrc/*.hip and *.h, api.hip -- the host driver -- left out): what a stored
    traffic profile is valid for.  bench.py computes the same digest of the tree it runs from."""
    h = hashlib.sha1()
    src = os.path.join(root, "gpmp2_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if (f.endswith(".hip") or f.endswith(".h")) and f != "api.hip":
            h.update(f.encode())
            h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]



def per_kernel(d, counter):
    acc = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("g2::", "")
            a = acc.setdefault(name, [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row.get("Dispatch_Id", str(len(a[1]))))
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        fl, wl = fetch[k][1], write[k][1]
        f_kb, w_kb = fetch[k][0] / fl, write[k][0] / wl
        kernels[k] = dict(launches=fl, fetch_size_kb_raw=f_kb, write_size_kb=w_kb,
                          hbm_bytes_per_launch_raw=(f_kb + w_kb) * 1024,
                          hbm_bytes_per_launch_corrected=(2 * f_kb + w_kb) * 1024)
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (python bench.py --steps 3 --warmup 1 "
            "--no-cpu-baseline), averaged per launch over all launches incl. late passes with few active trajectories. "
            "Units: KB as reported; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md 'HBM' "
            "(gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths uncalibrated).")
    json.dump(dict(note=note, commit=commit, sources_digest=kernel_sources_digest(), kernels=kernels), open(out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 2) for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
