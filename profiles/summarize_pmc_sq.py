"""Table for profiles/r03_pmc_sq.txt: per kernel (last dispatch of each), SQ / TA counters of profiles/collect_pmc_sq.sh."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for i in (1, 2, 3):
    d = os.path.join(ROOT, "gpurun_out", "%s_%d" % (sys.argv[1] if len(sys.argv) > 1 else "r03_sq", i))
    f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith("counter_collection.csv")]
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "adf::" not in name:
            continue
        short = name.split("adf::(anonymous namespace)::")[-1].split("(")[0]
        vals.setdefault(short, {})[r["Counter_Name"]] = float(r["Counter_Value"])   # later dispatches overwrite earlier ones
cols = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
        "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD",
        "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_INST_LEVEL_VMEM",
        "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_BUFFER_TOTAL_CYCLES_sum"]
print("counters of the LAST dispatch of each kernel, one sequential 8-pair 4K %s (ADF_NO_OVERLAP=1); SQ_* in the units rocprofv3 reports"
      % (sys.argv[2] if len(sys.argv) > 2 else "step"))
for k in sorted(vals):
    v = vals[k]
    print("\n" + k)
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    for c in cols:
        if c in v:
            frac = ("  (%.1f %% of SQ_WAVE_CYCLES)" % (100.0 * v[c] / wc)) if wc and c.startswith("SQ_") and c not in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INST_LEVEL_VMEM") else ""
            print("  %-36s %16.0f%s" % (c, v[c], frac))
