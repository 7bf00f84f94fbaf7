#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv) -> text table.

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
        --output-format csv -d gpurun_out/pmc_sq -o sq -- python3 tools/kernel_probe.py 4097 5 down_leg up_leg
    python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq/sq_counter_collection.csv
"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    if "reduce" in k or "convert" in k or "inject" in k or "sumsq" in k:
        continue
    n = max(len(v) for v in cs.values())
    if n < 3:
        continue
    print(k[:150])
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} mean {sum(v) / len(v):16.1f}   (n={len(v)})")
