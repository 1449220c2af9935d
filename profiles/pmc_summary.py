#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters: python profiles/pmc_summary.py <counter_collection.csv> [...] -> JSON on stdout.
Per kernel name (and grid size class): number of dispatches, every counter summed over the dispatches, VGPR / SGPR / LDS of the kernel."""
import csv, json, sys
out = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not (k.startswith("k_") or "radix" in k or "scan" in k or k.startswith("chase")):       # chase: tools/gather_ceiling.hip (the calibration pass of tools/measure_session.sh)
            continue
        k = k.split("(")[0][:60]
        d = out.setdefault(k, {"dispatch_ids": set(), "counters": {}, "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")})
        d["dispatch_ids"].add((path, r["Dispatch_Id"]))
        d["counters"][r["Counter_Name"]] = d["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in out.items():
    d["dispatches"] = len(d.pop("dispatch_ids"))
print(json.dumps(out, indent=1, sort_keys=True))
