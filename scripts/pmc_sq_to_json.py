"""Fold rocprofv3 --pmc passes (any counters, one directory per pass) into one JSON: per kernel, the mean value per
launch of every counter, plus a few ratios when their inputs are present (quad-cycle units per MI355X_MICROARCH.md:
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles).

usage: python scripts/pmc_sq_to_json.py <out.json> <pass_dir> [<pass_dir> ...]"""
import csv
import glob
import json
import os
import sys


def fold(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("g2::", "")
                k = acc.setdefault(name, {})
                c = k.setdefault(row["Counter_Name"], [0.0, set()])
                c[0] += float(row["Counter_Value"])
                c[1].add((f, row.get("Dispatch_Id")))
                for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                              "Workgroup_Size", "Grid_Size"):
                    if extra in row and row[extra] not in (None, ""):
                        k.setdefault("_" + extra, row[extra])
    out = {}
    for name, k in sorted(acc.items()):
        e = {c: v[0] / max(len(v[1]), 1) for c, v in k.items() if not c.startswith("_")}
        e["launches"] = max((len(v[1]) for c, v in k.items() if not c.startswith("_")), default=0)
        for c, v in k.items():
            if c.startswith("_"):
                e[c[1:]] = v
        g = e.get
        if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
            e["quad_cycles_per_wave"] = g("SQ_WAVE_CYCLES") / g("SQ_WAVES")
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            e["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
        if g("SQ_WAVE_CYCLES"):
            for src, dst in (("SQ_WAIT_ANY", "frac_wait_any"), ("SQ_WAIT_INST_ANY", "frac_wait_inst_any"),
                             ("SQ_ACTIVE_INST_ANY", "frac_active_inst_any"), ("SQ_ACTIVE_INST_VALU", "frac_active_inst_valu"),
                             ("SQ_WAIT_INST_LDS", "frac_wait_inst_lds")):
                if g(src) is not None:
                    e[dst] = g(src) / g("SQ_WAVE_CYCLES")
        if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            e["mfma_busy_over_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
        if g("SQ_INSTS_LDS") and g("SQ_LDS_BANK_CONFLICT") is not None:
            e["lds_conflict_cycles_per_inst"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")
        out[name] = e
    return out


if __name__ == "__main__":
    json.dump(dict(note="rocprofv3 --pmc, mean per launch over all launches of the kernel in the run; one pass per "
                        "counter group (no trace domains mixed in)", kernels=fold(sys.argv[2:])),
              open(sys.argv[1], "w"), indent=1)
    print("wrote", sys.argv[1])
