"""HBM bytes per kernel from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh
(gfx950: FETCH_SIZE counts half, unit KiB) — mean per launch and launches, libanirec kernels only."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
name = sys.argv[1]
out = defaultdict(dict)
for cn, sub, mul in (("FETCH_SIZE", "fetch", 2048.0), ("WRITE_SIZE", "write", 1024.0)):
    acc = defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/pmc_%s_%s/pmc_counter_collection.csv" % (name, sub))):
        if "anirec" in r["Kernel_Name"] and r["Counter_Name"] == cn:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("anirec::", "")].append(float(r["Counter_Value"]) * mul)
    for k, v in acc.items():
        out[k]["launches"] = len(v)
        out[k][sub + "_bytes_per_launch"] = sum(v) / len(v)
for k in out:
    out[k]["total_bytes_per_launch"] = out[k].get("fetch_bytes_per_launch", 0) + out[k].get("write_bytes_per_launch", 0)
def blob(path):        # `git hash-object`: the counters describe exactly this kernel source
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


sources = {os.path.basename(f): blob(f) for f in sorted(glob.glob("anime_recommendations_amd/csrc/*.h*"))}
print(json.dumps({"workload": name, "kernels": out, "sources": sources,
                  "note": "two separate rocprofv3 --pmc passes; FETCH_SIZE x2 (gfx950), unit KiB"}, indent=1))
