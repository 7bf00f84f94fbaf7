#!/usr/bin/env python3
"""profiles/pmc_latest.json from the two rocprofv3 counter passes of tools/kernel_probe.py.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid
    python3 tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/pmc_latest.json

Counter values are KB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced streaming read:
MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.  Kernels are told apart by their template arguments at the
finest level (TAG = 1) and the 4097^2 launch geometry of the probe.
"""
import csv
import json
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixed_precision_multigrid_solvers_for_pdes_amd import _build          # noqa: E402

N = 4097
fetch_csv, write_csv, out = sys.argv[1:4]


def classify(name):
    # the spanning leg: <T, HALO, TX, TC, TAG, SM, W, RPT, SPAN> (SPAN 1: the iterate in between is stored, 2: not)
    m = re.search(r"mg::rb_span_kernel<(float|double), \d+, \w+, \w+, [12], 0, \d+, \d+, ([12])>", name)
    if m:
        return f"span_leg{'' if m.group(2) == '1' else '_nomid'}_{'f32' if m.group(1) == 'float' else 'f64'}_{N}"
    m = re.search(r"mg::jacobi_kernel<(float|double), 1,", name)
    if m:
        return f"jacobi_sweep_{'f32' if m.group(1) == 'float' else 'f64'}_{N}"
    m = re.search(r"mg::fused_jacobi_kernel<(float|double), (\d+), (true|false), (\d), (true|false), \w+, \w+, 1, 0(, \d+)?(, false)?>", name)
    if not m:     # the register-blocked legs: <T, HALO, PROLONG, POST, ZERO_INIT, TX, TC, TAG, SM, W, RPT>
        m = re.search(r"mg::rb_leg_kernel<(float|double), (\d+), (true|false), (\d), (true|false), \w+, \w+, [12], 0, \d+, \d+(, false)?>", name)
    if not m:
        return None
    dt = "f32" if m.group(1) == "float" else "f64"
    prolong, post = m.group(3) == "true", int(m.group(4))
    if not prolong and post == 0:
        return f"jacobi_2sweeps_{dt}_{N}"
    if not prolong and post == 1:
        return f"down_leg_{dt}_{N}"
    if prolong and post == 2:
        return f"up_leg_{dt}_{N}"
    return None


def collect(path, counter):
    vals = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        key = classify(row["Kernel_Name"])
        if key:
            vals.setdefault(key, []).append(float(row["Counter_Value"]))
    return vals


fetch, write = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
w = {"f32": 4, "f64": 8}
words = {"jacobi_sweep": 3.0, "jacobi_2sweeps": 3.0, "down_leg": 3.25, "up_leg": 3.25, "span_leg": 4.5, "span_leg_nomid": 3.5}
kernels = {}
for key in sorted(fetch):
    kind, dt = key.rsplit("_", 2)[0], key.rsplit("_", 2)[1]
    f_kb, w_kb = statistics.median(fetch[key]), statistics.median(write.get(key, [0.0]))
    hbm = (2.0 * f_kb + w_kb) * 1024.0
    comp = int(words[kind] * w[dt] * N * N)
    kernels[key] = {"FETCH_SIZE_KB_raw_median": f_kb, "WRITE_SIZE_KB_median": w_kb, "launches": len(fetch[key]),
                    "hbm_bytes_per_launch_corrected": hbm, "compulsory_bytes_per_launch": comp,
                    "ratio_to_compulsory": hbm / comp}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python3 "
                   "tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid` on 1x MI355X, summarised by tools/pmc_summary.py. "
                   "Values in KB as reported; FETCH_SIZE is doubled (gfx950 counts 1/2 of a wide coalesced streaming read: "
                   "MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact. compulsory = bytes a perfect launch must move "
                   "(fields once).",
           "round": 3, "source_hash": _build.source_hash(), "kernels": kernels}, open(out, "w"), indent=1)
for k, v in kernels.items():
    print(f"{k:28s} {v['hbm_bytes_per_launch_corrected'] / 1e6:8.1f} MB / launch  x{v['ratio_to_compulsory']:.3f} of compulsory ({v['launches']} launches)")
