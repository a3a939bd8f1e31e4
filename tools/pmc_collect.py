"""rocprofv3 evidence for one workload of bench.py, run ON the GPU box from the repo root:

    python3 tools/pmc_collect.py OUT_DIR NAME -- <bench.py arguments>

e.g.  python3 tools/pmc_collect.py gpurun_out/r03_b cfg2_survey -- --only-headline --no-cpu --scene survey

Seven runs of the same command: `--kernel-trace --stats` (kernel times), then SEPARATE counter passes --
FETCH_SIZE, WRITE_SIZE and four SQ groups -- each with --kernel-trace only (never combined with another
trace domain; the program goes directly after `--`).  Writes OUT_DIR/NAME_kernel_stats.csv,
NAME_pmc_hbm.json (gfx950 correction: 2 x FETCH + WRITE, KB -> bytes) and NAME_pmc_sq.json (means per
launch, summed over the chip); copy what is to be judged into profiles/<round>_NAME_*.
This process never touches the GPU itself."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short  # noqa: E402

SQ_GROUPS = {
    "sq_a": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
    "sq_b": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"],
    "sq_c": ["SQ_INSTS_SALU", "SQ_INST_CYCLES_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    "sq_d": ["SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
}


def run(cmd, log):
    with open(log, "w") as f:
        return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=dict(os.environ, TMPDIR="/tmp")).returncode


def counter_means(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    out, name = sys.argv[1], sys.argv[2]
    bench_args = sys.argv[sys.argv.index("--") + 1:]
    os.makedirs(out, exist_ok=True)
    prog = ["python3", "bench.py"] + bench_args
    cmdline = " ".join(prog)
    # ---- kernel times
    d = os.path.join(out, f"_{name}_stats")
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "s", "--"] + prog,
             os.path.join(out, f"{name}_stats.log"))
    found = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if found:
        shutil.copy(found[0], os.path.join(out, f"{name}_kernel_stats.csv"))
    print(f"[{name}] kernel stats rc {rc}", flush=True)
    shutil.rmtree(d, ignore_errors=True)
    # ---- counters, one pass per group
    small = [a for a in bench_args]
    for flag, val in (("--steps", "50"), ("--warmup", "5")):
        if flag in small:
            small[small.index(flag) + 1] = val
        else:
            small += [flag, val]
    prog_small = ["python3", "bench.py"] + small
    per_kernel = defaultdict(dict)
    hbm = {}
    for pname, ctrs in [("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"])] + list(SQ_GROUPS.items()):
        d = os.path.join(out, f"_{name}_{pname}")
        rc = run(["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"]
                 + prog_small, os.path.join(out, f"{name}_pmc_{pname}.log"))
        found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        print(f"[{name}] pmc {pname} rc {rc} ({'ok' if found else 'no csv'})", flush=True)
        if found:
            for k, cs in counter_means(found[0]).items():
                for c, vals in cs.items():
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        hbm.setdefault(k, {})[c] = (sum(vals) / len(vals), len(vals))
                    else:
                        per_kernel[k][c] = sum(vals) / len(vals)
                        per_kernel[k]["launches"] = len(vals)
        shutil.rmtree(d, ignore_errors=True)
    kernels = {}
    for k, v in hbm.items():
        f, nf = v.get("FETCH_SIZE", (0.0, 0))
        w, nw = v.get("WRITE_SIZE", (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KB_mean": f, "launches_FETCH_SIZE": nf, "WRITE_SIZE_KB_mean": w,
                      "launches_WRITE_SIZE": nw, "hbm_bytes_per_launch_corrected": int(round((2.0 * f + w) * 1024))}
    json.dump({"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- {' '.join(prog_small)} (two separate passes)",
               "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; corrected = (2 x FETCH + WRITE) x 1024 "
                            "(gfx950 FETCH_SIZE halves wide reads; upper bound for narrower ones)",
               "kernels": kernels}, open(os.path.join(out, f"{name}_pmc_hbm.json"), "w"), indent=1)
    for k, v in per_kernel.items():
        if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
            v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        if v.get("SQ_LDS_BANK_CONFLICT") is not None and v.get("SQ_ACTIVE_INST_LDS"):
            v["lds_bank_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"] * 4, 1.0)
        if v.get("SQ_INSTS_SALU") and v.get("SQ_INSTS_VALU"):
            v["salu_per_valu"] = v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]
    json.dump({"command": f"rocprofv3 --pmc <counters of one pass> --kernel-trace -- {' '.join(prog_small)} "
                          f"(one pass per group: {', '.join(SQ_GROUPS)})",
               "note": "means per launch, summed over the chip as rocprofv3 reports them",
               "kernels": per_kernel}, open(os.path.join(out, f"{name}_pmc_sq.json"), "w"), indent=1)
    print(f"[{name}] {cmdline}: done", flush=True)


if __name__ == "__main__":
    main()
