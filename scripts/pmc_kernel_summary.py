"""Per-launch means of the SQ counters scripts/pmc_kernel.sh collected, for the kernels whose name contains a substring:
    python scripts/pmc_kernel_summary.py <tag> <substring> [clock_ghz]"""
import csv, os, sys
from collections import defaultdict
tag, sub = sys.argv[1], sys.argv[2]
clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for s in "ab":
    base = "gpurun_out/pmc_kernel_%s_%s" % (tag, s)
    for root, _, files in os.walk(base):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if sub in r["Kernel_Name"]:
                        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if f.endswith("kernel_trace.csv") and s == "a":
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if sub in r["Kernel_Name"]:
                        dur[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    t = sum(dur[k]) / max(1, len(dur[k]))
    print(k, "launches", len(next(iter(cs.values()))), "mean %.1f us (under the counters)" % (t / 1e3))
    for c in sorted(m):
        print("  %-28s %.4g" % (c, m[c]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA"):
            if c in m:
                print("  %-28s %.3f of the waves' resident time" % (c + " / WAVE_CYCLES", m[c] / wc))
    if t and "SQ_ACTIVE_INST_VALU" in m:
        print("  valu_pipe_busy %.3f (4 x ACTIVE_INST_VALU / (time x %.1f GHz x 1024 SIMDs))" %
              (4 * m["SQ_ACTIVE_INST_VALU"] / (t * clock * 1024), clock))
    if "SQ_WAVES" in m:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
            if c in m:
                print("  %-28s %.1f per wave" % (c, m[c] / m["SQ_WAVES"]))
