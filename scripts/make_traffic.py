"""profiles/traffic.json from the rocprofv3 --pmc passes of scripts/profile_pmc.sh (FETCH_SIZE, WRITE_SIZE in KiB).
hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 as the MI355X guide prescribes; on gfx950 FETCH_SIZE can under-count
wide coalesced reads by up to 2x (uncalibrated for this kernel's 4-byte scattered reads), so the read side is a lower bound."""
import csv, glob, json, os, sys
root, key = sys.argv[1], sys.argv[2]
vals = {}
for name in ("fetch", "write"):
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pathtrace" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": int((fetch + write) * 1024),
             "launches_averaged": len(vals["FETCH_SIZE"]), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
json.dump(data, open(out_path, "w"), indent=1)
print(key, data[key])
