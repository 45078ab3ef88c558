"""Timeline of the libanirec kernels of the LAST op call in a rocprofv3 kernel trace:
   python scripts/trace_timeline.py <kernel_trace.csv> [n_calls_in_script]
prints start (us from the first kernel of the call), duration, queue / stream and name, then overlap statistics:
how much of the k_refresh / k_rerank time ran while a k_cand kernel of ANOTHER chain was running."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anirec" in r["Kernel_Name"]]
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into calls at k_rownorm / the first k_to_f16 of a job (one memset-less marker: the key conversion has the largest grid)
marks = [i for i, r in enumerate(rows) if "k_rownorm" in r["Kernel_Name"]]
if marks:
    rows = rows[marks[-1]:]
else:
    rows = rows[-(len(rows) // ncalls):]
t0 = int(rows[0]["Start_Timestamp"])
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
scol = "Stream_Id" if "Stream_Id" in rows[0] else None
def short(n): return n.split("(")[0].replace("void ", "").replace("anirec::", "")
ev = []
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    ev.append((s, e, r.get(qcol, "?") if qcol else "?", r.get(scol, "?") if scol else "?", short(r["Kernel_Name"])))
verbose = len(ev) < 400
for s, e, qd, sd, nm in ev:
    if verbose or e - s > 200:
        print("%10.1f %9.1f  q=%s s=%s  %s" % (s, e - s, qd, sd, nm))
end = max(e for _, e, *_ in ev)
tot = {}
for s, e, qd, sd, nm in ev:
    tot[nm] = tot.get(nm, 0) + (e - s)
print("call span %.1f us; kernel time by name:" % end, {k: round(v, 1) for k, v in tot.items()})
cand = [(s, e, sd) for s, e, qd, sd, nm in ev if nm.startswith("k_cand")]
# union of k_cand intervals
cand.sort()
un, cur = 0.0, None
for s, e, _ in cand:
    if cur is None: cur = [s, e]
    elif s <= cur[1]: cur[1] = max(cur[1], e)
    else: un += cur[1] - cur[0]; cur = [s, e]
if cur: un += cur[1] - cur[0]
print("union of k_cand intervals: %.1f us (%.1f %% of the span); sum of k_cand durations %.1f us" % (un, 100 * un / end, sum(e - s for s, e, _ in cand)))
for kind in ("k_refresh", "k_rerank"):
    ov = tt = 0.0
    for s, e, qd, sd, nm in ev:
        if not nm.startswith(kind): continue
        tt += e - s
        for cs, ce, csd in cand:
            if csd != sd: ov += max(0.0, min(e, ce) - max(s, cs))
    print("%s: %.1f us in total, %.1f us of it beside a k_cand of another chain" % (kind, tt, ov))
