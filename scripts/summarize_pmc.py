"""Collapse rocprofv3 --pmc CSVs (gpurun_out/pmc/*/**/*counter_collection.csv) into one table per kernel dispatch."""
import csv, glob, os, sys, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
rows = {}
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "pathtrace" not in r.get("Kernel_Name", ""):
            continue
        key = (os.path.relpath(f, root).split(os.sep)[0], r["Dispatch_Id"])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        rows[key]["_kernel"] = r["Kernel_Name"][:60]
        for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
            if k in r:
                rows[key]["_" + k] = r[k]
out = {}
for (p, d), v in sorted(rows.items()):
    print(p, "dispatch", d, json.dumps(v))
