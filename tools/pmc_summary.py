"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate
passes, tools/profile_round.sh) -> per-kernel means and the corrected HBM bytes
per launch, the file bench.py reads for `roofline.traffic`.

    python tools/pmc_summary.py gpurun_out/r01_f_cfg2_pmc_FETCH_SIZE.csv \
                                gpurun_out/r01_f_cfg2_pmc_WRITE_SIZE.csv profiles/r01_f_cfg2_pmc_hbm.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    if "CycleTail" in name:  # the single-launch cycle is an instantiation of rollout_collide_kernel
        return "cycle_kernel"
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*$", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("kc::", "")


def means(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out):
    fe, wr = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in list(fe) + [k for k in wr if k not in fe]:
        f = sum(fe[k]) / len(fe[k]) if fe.get(k) else 0.0
        w = sum(wr[k]) / len(wr[k]) if wr.get(k) else 0.0
        kernels[k] = {"FETCH_SIZE_KB_mean": f, "launches_FETCH_SIZE": len(fe.get(k, [])),
                      "WRITE_SIZE_KB_mean": w, "launches_WRITE_SIZE": len(wr.get(k, [])),
                      "hbm_bytes_per_launch_corrected": int(round((2.0 * f + w) * 1024))}
    doc = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 50 --warmup 5 "
                      "--no-cpu (two separate passes)",
           "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; corrected = (2 x FETCH + WRITE) x 1024 "
                        "(gfx950 FETCH_SIZE halves wide reads; upper bound for narrower ones)",
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:32s} {v['hbm_bytes_per_launch_corrected']:>10d} B/launch  ({v['launches_FETCH_SIZE']} launches)")


if __name__ == "__main__":
    main(*sys.argv[1:4])
