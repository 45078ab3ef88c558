"""Instruction mix per k_cand instantiation from one rocprofv3 --pmc pass
(SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS):  python scripts/pmc_insts_by_kernel.py <counter_collection.csv>"""
import csv, sys
from collections import defaultdict
by = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_cand" in n:
        by[n.split("(")[0].replace("void anirec::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(by.items()):
    m = c.get("SQ_INSTS_MFMA", 0.0) or 1.0
    print("%-34s MFMA %.3e  VALU/MFMA %.2f  SALU/MFMA %.2f  LDS/MFMA %.2f" % (
        k, m, (c.get("SQ_INSTS_VALU", 0) - m) / m, c.get("SQ_INSTS_SALU", 0) / m, c.get("SQ_INSTS_LDS", 0) / m))
