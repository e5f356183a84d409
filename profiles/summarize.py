"""Turns what profiles/collect.sh left under gpurun_out/ into the committed summaries:

  profiles/<round>_<solver>_bench_n1.json          the bench line
  profiles/<round>_<solver>_bench_kernel_stats.csv rocprofv3 --kernel-trace --stats (kernel_stats), our kernels first
  profiles/<round>_<solver>_pmc_8pairs.csv         HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
  profiles/pmc_traffic.json                    bytes per stereo pair per kernel (bench.py's roofline.traffic)
  profiles/<round>_matcher_kernel_stats.csv        rocprofv3 --kernel-trace --stats of tools/bm_time.py (block matcher, N4)
  profiles/<round>_matcher_times.txt               its timings and the oracle's CPU time

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so reads are doubled.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = 8
ROUND = os.environ.get("ADF_ROUND", "r04")


def kernel_sources_sha16():
    """Same digest bench.py prints as roofline.traffic_source.kernel_sources_sha16_now."""
    import hashlib
    d = os.path.join(ROOT, "addingdisparityfiltering_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    m = re.search(r"adf::\(anonymous namespace\)::([\w]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def counters(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sorted(f, key=os.path.getmtime)[-1])):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            agg[k].append(float(r["Counter_Value"]))
    return agg


def main(solver):
    tag = os.path.join(ROOT, "gpurun_out", "%s_%s" % (ROUND, solver))
    out = os.path.join(ROOT, "profiles", "%s_%s" % (ROUND, solver))
    shutil.copy(tag + "_bench_n1.json", out + "_bench_n1.json")
    st = glob.glob(os.path.join(tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.reader(open(sorted(st, key=os.path.getmtime)[-1])))
    ours = [r for r in rows[1:] if "adf::" in r[0]]
    rest = [r for r in rows[1:] if "adf::" not in r[0]]
    with open(out + "_bench_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(rows[0]); w.writerows(ours); w.writerows(rest)
    fe, wr = counters(tag + "_pmc_fetch", "FETCH_SIZE"), counters(tag + "_pmc_write", "WRITE_SIZE")
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    per = {}
    with open(out + "_pmc_8pairs.csv", "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0 --pairs 8 (config 3, %s solver)\n" % solver)
        f.write("# FETCH_SIZE is in KB and on gfx950 reports 1/2 of a coalesced streaming read (MI355X_MICROARCH.md, HBM): corrected_read = 2*FETCH_SIZE*1024\n")
        f.write("kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,corrected_read_MB,write_MB,total_MB_per_launch,total_bytes_per_ROI_pixel\n")
        for k in fe:
            fk, wk = sum(fe[k]) / len(fe[k]), sum(wr[k]) / max(1, len(wr[k]))
            rd, wt = 2 * fk * 1024, wk * 1024
            px = 3584 * 2160 * PAIRS
            f.write("%s,%d,%.1f,%.1f,%.1f,%.1f,%.1f,%.2f\n" % (k, len(fe[k]), fk, wk, rd / 1e6, wt / 1e6, (rd + wt) / 1e6, (rd + wt) / px))
            per[k] = {"bytes_per_pair": (rd + wt) / PAIRS}
    traffic["%s_cfg3" % solver] = per
    import datetime
    traffic["_comment"] = ("HBM bytes per stereo pair and launch from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 "
                           "correction + WRITE_SIZE); sources: profiles/%s_wave_pmc_8pairs.csv (exact solver: the round it was "
                           "last collected in)" % ROUND)
    coll = traffic.get("_collected")
    coll = coll if isinstance(coll, dict) else {}
    coll["%s_cfg3" % solver] = "%s %s" % (ROUND, datetime.date.today().isoformat())
    traffic["_collected"] = coll
    traffic["_kernel_sources_sha16"] = kernel_sources_sha16()
    traffic["_method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, bench.py --pairs 8 "
                          "--steps 1; read bytes = 2 * FETCH_SIZE KB * 1024 (gfx950 correction), write bytes = WRITE_SIZE KB * 1024; "
                          "bench.py multiplies bytes_per_pair by the pairs per launch -- the figure is NOT measured by the bench run")
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(open(out + "_pmc_8pairs.csv").read())
    # the block matcher's own run (collect.sh step 4)
    ms = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_matcher_stats" % ROUND, "**", "*kernel_stats.csv"), recursive=True)
    sg = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_sgbm_stats" % ROUND, "**", "*kernel_stats.csv"), recursive=True)
    if solver == "wave" and sg:
        shutil.copy(sorted(sg, key=os.path.getmtime)[-1], os.path.join(ROOT, "profiles", "%s_sgbm_kernel_stats.csv" % ROUND))
        lines = [ln for ln in open(os.path.join(ROOT, "gpurun_out", "%s_sgbm_times.txt" % ROUND)) if ln.startswith("semi-global")]
        open(os.path.join(ROOT, "profiles", "%s_sgbm_times.txt" % ROUND), "w").write(
            "# tools/sgbm_time.py (profiles/collect.sh step 5): StereoSGBM MODE_SGBM_3WAY, P1 = 24*w*w, P2 = 96*w*w, preFilterCap 63,\n"
            "# left view + the right-view matcher of createRightMatcher; W x H x channels, numDisparities, blockSize, pairs per call\n" + "".join(lines))
    if solver == "wave" and ms:
        shutil.copy(sorted(ms, key=os.path.getmtime)[-1], os.path.join(ROOT, "profiles", "%s_matcher_kernel_stats.csv" % ROUND))
        lines = [ln for ln in open(os.path.join(ROOT, "gpurun_out", "%s_matcher_times.txt" % ROUND)) if ln.startswith(("matcher", "oracle"))]
        open(os.path.join(ROOT, "profiles", "%s_matcher_times.txt" % ROUND), "w").write(
            "# tools/bm_time.py / tools/bm_cpu_time.py (profiles/collect.sh step 4); the last matcher line runs cv::StereoBM's default\n"
            "# uniqueness (15) and texture (10) tests, the others the filter factory's setting (both off)\n" + "".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "wave")
