"""Explanatory counters of the cycle ("VALU utilisation + launch count"): the per-kernel means of
the separate `rocprofv3 --pmc SQ_*` passes of tools/profile_round2.sh -> profiles/<tag>_pmc_sq.json.

    python tools/pmc_sq_summary.py gpurun_out/r02_a profiles/r02_a_cfg2
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short  # noqa: E402


def collect(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(src, out_prefix):
    for scene in ("survey", "mid", "open"):
        files = sorted(glob.glob(os.path.join(src, f"cfg2_{scene}_pmc_sq_*.csv")))
        if not files:
            continue
        kernels = defaultdict(dict)
        for f in files:
            for k, ctrs in collect(f).items():
                for c, vals in ctrs.items():
                    kernels[k][c] = sum(vals) / len(vals)
                    kernels[k]["launches"] = len(vals)
        for k, v in kernels.items():
            # derived, per launch: all counters are sums over the chip (256 CUs, 4 SIMDs each)
            if v.get("SQ_BUSY_CYCLES") and v.get("SQ_ACTIVE_INST_VALU"):
                # SQ_ACTIVE_INST_VALU counts cycles (x4 quad-cycles) a SIMD issues VALU work; BUSY_CYCLES per SE
                v["valu_active_per_wave_cycle"] = v["SQ_ACTIVE_INST_VALU"] * 4 / max(v.get("SQ_WAVE_CYCLES", 0) * 1.0, 1.0) \
                    if v.get("SQ_WAVE_CYCLES") else None
            if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
                v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
            if v.get("SQ_INSTS_LDS") and v.get("SQ_WAVES"):
                v["lds_insts_per_wave"] = v["SQ_INSTS_LDS"] / v["SQ_WAVES"]
            if v.get("SQ_LDS_BANK_CONFLICT") and v.get("SQ_ACTIVE_INST_LDS"):
                v["lds_bank_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"] * 4, 1.0)
        doc = {"command": "rocprofv3 --pmc <counters of one pass> --kernel-trace -- python3 bench.py --only-headline "
                          f"--no-cpu --steps 50 --warmup 5 --scene {scene} (one pass per counter group: "
                          "sq_a..sq_d of tools/profile_round2.sh)",
               "note": "means per launch, summed over the chip as rocprofv3 reports them",
               "kernels": kernels}
        out = f"{out_prefix}_{scene}_pmc_sq.json"
        with open(out, "w") as f:
            json.dump(doc, f, indent=1)
        print(out)
        for k, v in kernels.items():
            print(f"  {k:28s}", {c: round(x, 1) for c, x in v.items() if isinstance(x, float)})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
