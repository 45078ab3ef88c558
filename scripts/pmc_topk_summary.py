"""Summarise the top-k PMC passes (gpurun_out/pmc_topk/g*/, gpurun_out/pmc_clk/) for the LAST (largest)
k_cand launch of the timed call and for all k_cand launches of that call -> JSON on stdout."""
import csv, glob, json, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
cnt_last, cnt_all = {}, defaultdict(float)
for f in sorted(glob.glob(root + "/pmc_topk/g*/pmc_counter_collection.csv")):
    by = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_cand" in r["Kernel_Name"]:
            by[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(by)
    half = ids[len(ids) // 2:]          # the timed call (the first half is the warm-up call)
    for c, v in by[half[-1]].items():
        cnt_last[c] = v
    for d in half:
        for c, v in by[d].items():
            cnt_all[c] += v
clk = {}
rows = list(csv.DictReader(open(root + "/pmc_clk/pmc_counter_collection.csv")))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(root + "/pmc_clk/pmc_kernel_trace.csv"))}
cands = [r for r in rows if "k_cand" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
last = cands[-1]
k = kt[last["Dispatch_Id"]]
dur_ns = int(k["End_Timestamp"]) - int(k["Start_Timestamp"])
ghz = float(last["Counter_Value"]) / 8 / dur_ns
n_simd = 1024
busy = cnt_last["SQ_VALU_MFMA_BUSY_CYCLES"] / n_simd
out = {
    "workload": "cosine_topk_mfma 350000 keys x 65536 queries, k=10; counters of the last k_cand launch (tiles 1536..2734) of the timed call",
    "last_launch": {kk: vv for kk, vv in sorted(cnt_last.items())},
    "last_launch_duration_us_under_pmc": dur_ns / 1e3,
    "effective_clock_ghz": ghz,
    "mfma_busy_cycles_per_simd": busy,
    "mfma_pipe_utilisation": busy / (dur_ns * ghz),
    "mfma_insts_per_valu_inst": cnt_last["SQ_INSTS_MFMA"] / max(cnt_last["SQ_INSTS_VALU"] - cnt_last["SQ_INSTS_MFMA"], 1),
    "lds_bank_conflict_cycles": cnt_last["SQ_LDS_BANK_CONFLICT"],
    "lds_active_fraction_of_cu_cycles": cnt_last["SQ_LDS_IDX_ACTIVE"] / (dur_ns * ghz * 256),
    "note": "effective clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md 'DVFS give-back'); "
            "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock)",
}
print(json.dumps(out, indent=1))
