"""Table for profiles/r03_mall_kernels.txt: per-kernel average durations (rocprofv3) of the 4K filter call by pairs per
call, per pair, next to 1/64 of the 64-pair figures."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = (1, 2, 4, 8, 64)
rows = {}
for n in NS:
    path = os.path.join(ROOT, "gpurun_out", "r03_mall_p%d" % n)
    f = [os.path.join(dp, x) for dp, _, fs in os.walk(path) for x in fs if x.endswith("kernel_stats.csv")][0]
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if "adf::" not in name:
            continue
        short = name.split("adf::(anonymous namespace)::")[-1].split("(")[0]
        rows.setdefault(short, {})[n] = float(r["AverageNs"]) / 1e3
print("per-kernel average duration in microseconds PER PAIR (rocprofv3 --kernel-trace --stats), 3840x2160, ROI (256,0,3584,2160)")
print("%-36s" % "kernel" + "".join("%12s" % ("%d pair%s" % (n, "s" if n > 1 else "")) for n in NS))
tot = {n: 0.0 for n in NS}
for k in sorted(rows, key=lambda k: -rows[k].get(64, 0)):
    line = "%-36s" % k[:36]
    for n in NS:
        v = rows[k].get(n)
        line += "%12s" % ("-" if v is None else "%.1f" % (v / n))
    print(line)
step = {}
for n in NS:
    d = json.loads(open(os.path.join(ROOT, "gpurun_out", "r03_mall_p%d.json" % n)).read().strip().splitlines()[-1])
    step[n] = d["ms_per_step"] * 1e3 / n
print("%-36s" % "whole call per pair (bench, us)" + "".join("%12.1f" % step[n] for n in NS))
