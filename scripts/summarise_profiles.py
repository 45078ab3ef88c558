"""gpurun_out/<round>/<workload>_{stats,fetch,write}/ (scripts/collect_profiles.sh) -> profiles/<round>_*:
  <round>_<workload>_kernel_stats.csv   per-kernel Calls / TotalDurationNs / AverageNs / ... of the libanirec kernels
  <round>_pmc_traffic_<workload>.json   HBM bytes per launch and kernel: FETCH_SIZE x 2 (gfx950 counts half of a wide
                                        coalesced read) + WRITE_SIZE, unit KiB (MI355X_MICROARCH.md, HBM), stamped with
                                        the git blob hash of every kernel source so that bench.py can tell a stale file.
usage: python scripts/summarise_profiles.py r02 train topk100 ...   (bench.py reads `train` as train_s109m)"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, works = sys.argv[1], sys.argv[2:]
NAMES = {"train": "train_s109m", "train7m": "train_s7m", "topk100": "cosine_topk_k100", "topk10": "cosine_topk_k10", "topkall": "cosine_topk_allpairs_k100",
         "topkall10": "cosine_topk_allpairs_k10",
         "topk18k": "cosine_topk_18k_k100", "pgrid": "pgrid", "ptk": "ptk", "ingest": "ingest", "recs": "recs"}


NCALLS = {"topk100": 3, "topk10": 3, "topkall": 3, "topkall10": 3, "topk18k": 3, "pgrid": 2, "ptk": 2, "ingest": 3}   # op calls per script run


def blob(path):        # `git hash-object`
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def short(n):
    return n.split("(")[0].replace("void ", "").replace("anirec::", "")


sources = {os.path.basename(f): blob(f) for f in sorted(glob.glob(os.path.join(ROOT, "anime_recommendations_amd/csrc/*.h*")))}
# the host code that shapes the launches of an op (batch plans, buffer sizes) is part of what was measured
for f in ("ops.py", "engine.py", "ingest.py", "recs.py"):
    sources[f] = blob(os.path.join(ROOT, "anime_recommendations_amd", f))
for w in works:
    base = os.path.join(ROOT, "gpurun_out", rnd, w)
    name = NAMES.get(w, w)
    st = glob.glob(base + "_stats/*kernel_stats.csv")
    if st:
        rows = [r for r in csv.DictReader(open(st[0])) if "anirec" in r["Name"]]
        if rows:
            with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (rnd, name)), "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
    # what the profiled script printed, and per-kernel sums of the LAST call it made (the earlier ones warm up)
    tr = glob.glob(base + "_stats/*kernel_trace.csv")
    log = base + "_stats.log"
    if tr and w in NCALLS:
        rows = sorted((r for r in csv.DictReader(open(tr[0])) if "anirec" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        by = defaultdict(list)
        for r in rows:
            by[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        last = {k: {"launches_per_call": len(v) // NCALLS[w], "sum_us_last_call": round(sum(v[-(len(v) // NCALLS[w]):]), 1)}
                for k, v in by.items() if len(v) % NCALLS[w] == 0}
        lines = [l.rstrip() for l in open(log)] if os.path.exists(log) else []
        lines = [l for l in lines if l and not l.startswith(("W2", "E2", "I2", "/opt/amdgpu"))]
        json.dump({"workload": name, "calls_in_script": NCALLS[w], "kernel_trace_last_call": last, "script_stdout": lines},
                  open(os.path.join(ROOT, "profiles", "%s_%s_last_call.json" % (rnd, name)), "w"), indent=1)
    out = defaultdict(dict)
    for cn, sub, mul in (("FETCH_SIZE", "fetch", 2048.0), ("WRITE_SIZE", "write", 1024.0)):
        f = glob.glob(base + "_%s/*counter_collection.csv" % sub)
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "anirec" in r["Kernel_Name"] and r["Counter_Name"] == cn:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * mul)
        for k, v in acc.items():
            out[k]["launches"] = len(v)
            out[k][sub + "_bytes_per_launch"] = sum(v) / len(v)
            out[k][sub + "_bytes_all_launches"] = sum(v)
    if out:
        for k in out:
            out[k]["total_bytes_per_launch"] = out[k].get("fetch_bytes_per_launch", 0) + out[k].get("write_bytes_per_launch", 0)
            out[k]["total_bytes_all_launches"] = out[k].get("fetch_bytes_all_launches", 0) + out[k].get("write_bytes_all_launches", 0)
        json.dump({"workload": name, "kernels": out, "sources": sources,
                   "note": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; FETCH_SIZE x2 "
                           "(gfx950), unit KiB; `sources` = git blob hashes of the kernel sources measured"},
                  open(os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, name)), "w"), indent=1)
    print(w, "->", name, sorted(out))
