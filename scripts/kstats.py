"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count/avg/min/max (us) for anirec kernels
and the gaps between consecutive kernels inside the steady-state region."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "anirec" in n:
        short = n.split("(")[0].replace("void ", "").replace("anirec::", "")
        d[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v2 = sorted(v)
    print("%-22s n=%5d avg=%9.2f med=%9.2f min=%9.2f max=%9.2f us" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], v2[-1]))
# gaps between consecutive anirec kernels (graph region = most of them)
an = [r for r in rows if "anirec" in r["Kernel_Name"]]
gaps = defaultdict(list)
for a, b in zip(an, an[1:]):
    ka = a["Kernel_Name"].split("(")[0].split("::")[-1]
    kb = b["Kernel_Name"].split("(")[0].split("::")[-1]
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[ka + "->" + kb].append(g)
for k, v in sorted(gaps.items()):
    v2 = sorted(v)
    print("gap %-28s n=%5d med=%8.2f min=%8.2f max=%9.2f us" % (k, len(v), v2[len(v2) // 2], v2[0], v2[-1]))
