"""Timeline of one filter call on ONE pair (configs 2 and 5) from rocprofv3's kernel trace: per kernel the median
duration and the median gap to the previous kernel of the same call, over the timed calls."""
import csv
import json
import os
import statistics as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cfg in (2, 5):
    d = os.path.join(ROOT, "gpurun_out", "r03_lat_cfg%d" % cfg)
    f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith("kernel_trace.csv")][0]
    ks = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "adf::" not in n:
            continue
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("adf::(anonymous namespace)::")[-1].split("(")[0], r.get("Stream_Id", "")))
    ks.sort()
    # a call = the kernels from one outside_kernel / weights kernel to the last column pass
    calls, cur = [], []
    for k in ks:
        cur.append(k)
        if k[2].startswith("wave_vpass_kernel") and k[2].rstrip(">").endswith(", 1"):
            calls.append(cur); cur = []
    calls = [c for c in calls if len(c) == len(calls[-1])][-200:]
    txt = open(os.path.join(ROOT, "gpurun_out", "r03_lat_cfg%d.txt" % cfg)).read().strip().splitlines()
    print("config %d, one pair per call, %d calls in the trace; untraced / traced runs of the same loop:" % (cfg, len(calls)))
    for t in txt:
        print("    " + t)
    print("  %-34s %10s %10s %10s" % ("kernel", "start us", "dur us", "end us"))
    # kernels of a call keyed by (name, occurrence): the two streams make the issue order vary from call to call
    keyed = []
    for c in calls:
        seen, d = {}, {}
        for k in c:
            seen[k[2]] = seen.get(k[2], 0) + 1
            d[(k[2], seen[k[2]])] = ((k[0] - c[0][0]) / 1e3, (k[1] - k[0]) / 1e3)
        keyed.append(d)
    keys = sorted(keyed[0], key=lambda q: st.median(kd[q][0] for kd in keyed if q in kd))
    for q in keys:
        start = st.median(kd[q][0] for kd in keyed if q in kd)
        dur = st.median(kd[q][1] for kd in keyed if q in kd)
        print("  %-34s %10.2f %10.2f %10.2f" % ((q[0] + (" #%d" % q[1] if q[1] > 1 else ""))[:34], start, dur, start + dur))
    span = [(max(x[1] for x in c) - c[0][0]) / 1e3 for c in calls]
    busy = [sum(x[1] - x[0] for x in c) / 1e3 for c in calls]
    inter = [(calls[j + 1][0][0] - max(x[1] for x in calls[j])) / 1e3 for j in range(len(calls) - 1)]
    print("  first kernel start -> last kernel end: median %.2f us; sum of kernel durations %.2f us; gap between calls %.2f us" % (
        st.median(span), st.median(busy), st.median(inter)))
