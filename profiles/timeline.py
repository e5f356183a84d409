"""Prints the kernel timeline (start / end / duration, queue) of the last two steps of a rocprofv3 --kernel-trace run of
bench.py:   python profiles/timeline.py gpurun_out/<dir>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "adf::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-18:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].split("adf::(anonymous namespace)::")[-1].split("(")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-40s start %9.1f us  end %9.1f us  dur %8.1f  queue %s" % (name, s, e, e - s, r.get("Queue_Id")))
