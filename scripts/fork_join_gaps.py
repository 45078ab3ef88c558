"""Cost of a cross-stream fork / join on this stack, from a rocprofv3 kernel trace of the one-rank rehearsal of the
multi-GPU step (mode sharded: k_bwd -> [side stream: k_adam over the user rows] || k_densify -> join -> k_adam over
the anime rows).  usage: fork_join_gaps.py <kernel_trace.csv> [out.json]"""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anirec" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("anirec::", "")
ev = [(name(r), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
gaps = {"bwd_end_to_forked_adam_start": [], "bwd_end_to_densify_start_same_stream": [],
        "forked_adam_end_to_joined_adam_start": []}
for i, (n, s, e, q) in enumerate(ev):
    if n == "k_bwd":
        nxt = [x for x in ev[i + 1:i + 6]]
        ad = [x for x in nxt if x[0].startswith("k_adam")]
        dz = [x for x in nxt if x[0].startswith("k_densify")]
        if ad and dz and ad[0][3] != dz[0][3]:
            gaps["bwd_end_to_forked_adam_start"].append((ad[0][1] - e) / 1e3)
            gaps["bwd_end_to_densify_start_same_stream"].append((dz[0][1] - e) / 1e3)
            if len(ad) > 1:
                gaps["forked_adam_end_to_joined_adam_start"].append((ad[1][1] - max(ad[0][2], dz[0][2])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2] if v else None
out = {k: {"n": len(v), "median_us": med(v), "min_us": min(v) if v else None, "max_us": max(v) if v else None} for k, v in gaps.items()}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump({"what": __doc__, "gaps": out}, open(sys.argv[2], "w"), indent=1)
