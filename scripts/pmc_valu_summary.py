"""Per-kernel SQ counters of the training step from the three passes of scripts/pmc_valu.sh:

    python scripts/pmc_valu_summary.py <tag> > profiles/r04_pmc_valu_train_s109m.json

For every libanirec training kernel: launches, mean duration (kernel trace of pass a), the counters' per-launch means
and the ratios that say what a kernel is bound by.  Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES,
SQ_BUSY_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles), summed over the chip;
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
  valu_active_share   = ACTIVE_INST_VALU / WAVE_CYCLES   share of its resident time a wave spends issuing vector ALU work
  wait_any_share      = WAIT_ANY / WAVE_CYCLES           ... parked on s_waitcnt / a barrier (memory latency)
  wait_inst_share     = WAIT_INST_ANY / WAVE_CYCLES      ... ready but not issued (issue port / dependency stall)
  valu_pipe_busy      = 4 * ACTIVE_INST_VALU / (kernel time * clock * 1024 SIMDs)   share of the chip's VALU issue time
  waves_per_simd      = WAVE_CYCLES / (BUSY_CYCLES per SE summed ...) is not reliable; occupancy is quoted from the
                        register allocation instead (VGPR_Count of the trace).
"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
KEEP = ("k_lazy_flush", "k_lazy_adam", "k_lazy_catchup", "k_lazy_reduce", "k_fwd", "k_head", "k_bwd", "k_adam", "k_prep")


def short(n):
    return n.split("(")[0].replace("void ", "").replace("anirec::", "")


def find(sub, name):
    base = "gpurun_out/pmc_valu_%s_%s" % (tag, sub)
    for root, _, files in os.walk(base):
        for f in files:
            if f.endswith(name):
                return os.path.join(root, f)
    return None


out = defaultdict(lambda: {"counters": {}})
for sub in "abc":
    p = find(sub, "counter_collection.csv")
    if not p:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if not k.startswith(KEEP):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                   "lds": r.get("LDS_Block_Size"), "grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size")}
    for k, cs in acc.items():
        for c, v in cs.items():
            out[k]["counters"][c] = sum(v) / len(v)
            out[k]["launches"] = len(v)
        out[k].update({kk: vv for kk, vv in meta[k].items() if vv is not None})
    t = find(sub, "kernel_trace.csv")
    if t and sub == "a":
        dur = defaultdict(list)
        for r in csv.DictReader(open(t)):
            k = short(r["Kernel_Name"])
            if k.startswith(KEEP):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            out[k]["avg_us_under_pmc"] = sum(v) / len(v)

for k, d in out.items():
    c = d["counters"]
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for name, key in (("valu_active_share", "SQ_ACTIVE_INST_VALU"), ("any_active_share", "SQ_ACTIVE_INST_ANY"),
                          ("wait_any_share", "SQ_WAIT_ANY"), ("wait_inst_share", "SQ_WAIT_INST_ANY")):
            if key in c:
                d[name] = c[key] / wc
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_INSTS_VALU" in c and c["SQ_INSTS_VALU"]:
        d["quad_cycles_per_valu_inst"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
    if "GRBM_GUI_ACTIVE" in c and d.get("avg_us_under_pmc"):
        d["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (d["avg_us_under_pmc"] * 1e3)
    if "SQ_ACTIVE_INST_VALU" in c and d.get("avg_us_under_pmc"):
        clk = d.get("effective_clock_ghz", 2.4)
        d["valu_pipe_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (d["avg_us_under_pmc"] * 1e3 * clk * 1024)


def blob(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


print(json.dumps({"workload": "bench.py --no-also --no-cpu-baseline --steps 96 --warmup 32 (S109M, lazy dense Adam)",
                  "tag": tag, "kernels": out,
                  "sources": {"anirec_train.hip": blob("anime_recommendations_amd/csrc/anirec_train.hip")},
                  "note": "three separate rocprofv3 --pmc passes (scripts/pmc_valu.sh); SQ cycle counters in quad-cycles, "
                          "summed over the chip; per-launch means"}, indent=1))
