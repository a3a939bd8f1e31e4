"""Phase clocks of the single-launch cycle kernel as JSON (bench.py reads profiles/*_<cfg>_<scene>_phase_stamps.json
for `roofline.critical_path_us`).  Needs the -DKC_PHASE_STAMPS build in kompass-core_amd/lib_stamps:

    make -C kompass-core_amd OUT=lib_stamps HIPFLAGS_EXTRA=-DKC_PHASE_STAMPS
    python tools/stamps_json.py cfg2 survey gpurun_out/r03_b/cfg2_survey_phase_stamps.json
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main(cfg, scene, out):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cycle_stamps.py"), cfg, scene],
                       capture_output=True, text=True, timeout=300)
    phases, block = {}, None
    for ln in p.stderr.splitlines():
        if ln.startswith("[kc stamps] roll-out kernel"):
            block = "roll"
            continue
        if ln.startswith("[kc stamps]") or ln.startswith("[kc host]"):
            block = None
        m = re.match(r"\s+(.+?)\s+([\d.]+) /\s+([\d.]+)\s+\((\d+) blocks\)", ln)
        if block == "roll" and m and int(m.group(4)) > 0:
            phases[m.group(1).strip()] = {"avg": float(m.group(2)), "max": float(m.group(3)), "workgroups": int(m.group(4))}
    doc = {"kernel": "cycle_kernel", "config": cfg, "scene": scene,
           "what": "s_memrealtime stamps (100 MHz) of a -DKC_PHASE_STAMPS build, microseconds since the first workgroup's "
                   "start, avg / max over the workgroups that reached the phase; the phases are a serial chain inside "
                   "every workgroup, the max of the last one is the kernel's critical path",
           "summary": p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "", "phases_us": phases}
    json.dump(doc, open(out, "w"), indent=1)
    print(out, {k: v["max"] for k, v in phases.items()})
    if not phases:
        sys.stderr.write(p.stderr[-2000:])
        sys.exit(1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
