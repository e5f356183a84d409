#!/usr/bin/env python3
"""bench.py -- DisparityWLSFilter throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one DisparityWLSFilter::filter call over this rank's batch of synthetic stereo pairs
(BASELINE config 3: 64 pairs of 3840x2160 per GPU, ROI (256,0,3584,2160), 8UC3 guide, lambda 8000,
sigma 1.5, 3 FGS iterations, LRC confidence on), inputs and outputs resident in HBM.  Pairs are
independent, so N GPUs filter N x 64 pairs (weak scaling); RCCL is used only to scatter the batch
from rank 0 before, and gather the filtered maps after, the timed region.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement):
  value     = whole-job filtered Mpixels/s (full-frame W*H per pair), max-over-ranks time
  roofline  = dominant kernel (the Thomas-solve pass) priced in ALGORITHMIC bytes:
              (4+8R)*P bytes per launch / mean launch duration from HIP events on the launch stream
  cpu_baseline = the CPU oracle (a port of the reference path) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (geometry)")
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per GPU and step")
    ap.add_argument("--solver", choices=["exact", "wave"], default=os.environ.get("ADF_BENCH_SOLVER", "wave"))
    ap.add_argument("--distribution", choices=["scatter", "local"], default="local",
                    help="N>1: each rank builds its own contiguous shard of the batch (default: the path shards "
                         "with no data-path collective), or rank 0 builds the whole batch and scatters it / gathers "
                         "the results over RCCL point-to-point groups (outside the timed region either way)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of pair 0")
    ap.add_argument("--matcher-pairs", type=int, default=4,
                    help="pairs of the extra views -> matcher -> filter leg (SURVEY 8f N4; N=1 only, 0 = skip)")
    return ap.parse_args()


def cpu_baseline(view, dl, dr, roi, radius, seconds):
    """Time the CPU oracle like perf_disparity_wls_filter.cpp:86-90 (filter built inside the loop).

    The stripe count (= thread count, DF.cpp:158) that is fastest on this host is not known in
    advance, so a few candidates share the time budget and the best one is reported."""
    import oracle

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({t for t in (8, 16, 32, 64, avail) if t <= avail} or {1})
    n = view.shape[0]
    H, W = dl.shape[1:]
    best, tried = None, []
    for threads in cands:
        p = oracle.default_params(sigma_color=1.5, disc_radius=radius, threads=threads)
        p.lambda_ = 8000.0
        oracle.wls_filter(dl[0], view[0], dr[0], roi, p, want_conf=True)  # warm-up cycle
        cycles, t0 = 0, time.perf_counter()
        while True:
            k = cycles % n
            oracle.wls_filter(dl[k], view[k], dr[k], roi, p, want_conf=True)
            cycles += 1
            el = time.perf_counter() - t0
            if el >= seconds / len(cands) or cycles >= 200:
                break
        rate = cycles * W * H / el / 1e6
        tried.append("%d thr: %.1f Mpx/s (%d calls, %.1f s)" % (threads, rate, cycles, el))
        if best is None or rate > best[0]:
            best = (rate, threads)
    return {
        "value": round(best[0], 3), "unit": "Mpixels/s", "cores": best[1], "kind": "port",
        "sample": "DisparityWLSFilter on %d pair(s) of %dx%d, same inputs/params as the GPU run; oracle/adf_oracle.c "
                  "(scalar order, pthread stripes); %d logical CPUs visible; tried %s"
                  % (n, W, H, avail, "; ".join(tried)),
    }


def views_to_filtered(xi, view, roi, radius, n, num_disp, block):
    """Extra leg, outside the timed region and not part of `value`: the device block matcher (both views) feeding the
    filter, all stages on torch's stream, inputs resident."""
    import torch
    left = view[:n, :, :, 0].contiguous()
    right = torch.roll(left, -min(num_disp // 3, 60), 2).contiguous()
    lm = xi.StereoBM.create(num_disp, block)
    wls = xi.createDisparityWLSFilter(lm)                    # DF.cpp:386-414 (forces texture / uniqueness tests off)
    rm = xi.createRightMatcher(lm)                           # DF.cpp:417-431
    wls.setLambda(8000.0); wls.setSigmaColor(1.5)
    H, W = left.shape[1:]
    dl = torch.empty((n, H, W), dtype=torch.int16, device=left.device); dr = torch.empty_like(dl); out = torch.empty_like(dl)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(2):                                      # first round warms the workspaces up
        ev[0].record()
        lm.computeBoth(left, right, dl, dr)                  # = lm.compute(left, right), rm.compute(right, left)
        ev[1].record()
        wls.filter(dl, view[:n], out, dr)
        ev[2].record()
    torch.cuda.synchronize()
    return {"pairs": n, "num_disparities": num_disp, "block_size": block, "roi": list(wls.getROI()),
            "matcher_ms_per_pair": round(ev[0].elapsed_time(ev[1]) / n, 4),
            "filter_ms_per_pair": round(ev[1].elapsed_time(ev[2]) / n, 4),
            "Mpixels_per_s": round(n * H * W / (ev[0].elapsed_time(ev[2]) * 1e-3) / 1e6, 1),
            "note": "block matcher (left + right view from one launch) then the filter, each one call for the batch; not part of `value`"}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import addingdisparityfiltering_amd as adf
    from addingdisparityfiltering_amd import parallel, synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # ADF_BENCH_BACKEND=gloo is a rehearsal mode for one-GPU boxes: the ranks share the visible GPUs
    # (rank r uses device r % device_count) and the two scalar all-reduces run on CPU tensors, so the whole
    # N > 1 flow executes except RCCL itself.  The driver's runs use the default, nccl (= RCCL).
    backend = os.environ.get("ADF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            if args.distribution == "scatter":
                raise SystemExit("the scatter distribution moves device tensors: it needs the nccl backend")
            dist.init_process_group(backend=backend)

    cfg = synthetic.CONFIGS[args.config]
    W, H, roi, ch, radius = cfg["W"], cfg["H"], cfg["roi"], cfg["channels"], cfg["radius"]
    pairs = args.pairs
    n_total = pairs * world

    # ---- inputs, resident in HBM before the timed region ----
    scatter_ms = None
    base_seed = synthetic.seed_for(args.config, 0)
    if world > 1 and args.distribution == "scatter":
        full = [None, None, None]
        if rank == 0:
            full = list(synthetic.make_artificial_batch_torch(n_total, W, H, ch, base_seed, cfg["rect_disparity"], dev))
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        vshape = (H, W, ch) if ch > 1 else (H, W)
        view = parallel.scatter_batch(full[0], n_total, vshape, torch.uint8, dev)
        dl = parallel.scatter_batch(full[1], n_total, (H, W), torch.int16, dev)
        dr = parallel.scatter_batch(full[2], n_total, (H, W), torch.int16, dev)
        torch.cuda.synchronize(); dist.barrier()
        scatter_ms = (time.perf_counter() - t0) * 1e3
        view, dl, dr = view.contiguous().clone(), dl.contiguous().clone(), dr.contiguous().clone()
        del full
        torch.cuda.empty_cache()
    else:
        view, dl, dr = synthetic.make_artificial_batch_torch(pairs, W, H, ch, base_seed + rank * pairs,
                                                             cfg["rect_disparity"], dev)
    out = torch.empty((pairs, H, W), dtype=torch.int16, device=dev)

    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0)
    f.setSigmaColor(1.5)
    f.setDepthDiscontinuityRadius(radius)
    f.setSolver(adf.SOLVER_WAVE if args.solver == "wave" else adf.SOLVER_EXACT)

    f.enableProfiling(True)   # also during warm-up, so that the event pool exists before the timed region
    for _ in range(max(args.warmup, 0)):
        f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()

    # ---- correctness of what is being timed: pair 0 of rank 0 against the CPU oracle ----
    checked = None
    if rank == 0 and not args.no_check:
        import oracle
        if args.warmup <= 0:
            f.filter(dl, view, out, dr, roi); torch.cuda.synchronize()
        p = oracle.default_params(sigma_color=1.5, disc_radius=radius,
                                  threads=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)
        p.lambda_ = 8000.0
        exp, exp_conf = oracle.wls_filter(dl[0].cpu().numpy(), view[0].cpu().numpy(), dr[0].cpu().numpy(), roi, p)
        got = out[0].cpu().numpy().astype(np.int64)
        conf_ok = bool(np.array_equal(f.getConfidenceMap(0).cpu().numpy(), exp_conf))
        diff = np.abs(got - exp)
        checked = {"pair": 0, "confidence_bit_exact": conf_ok, "disparity_max_abs_lsb": int(diff.max()),
                   "disparity_mean_abs_lsb": float(diff.mean())}
        ok = conf_ok and (diff.max() == 0 if args.solver == "exact" else (diff.max() <= 1 and diff.mean() <= 1 / 256))
        if not ok:
            raise SystemExit("bench: GPU result differs from the oracle: %s" % checked)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    f.enableProfiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, coll_dev)
    prof = f.readProfile()
    f.enableProfiling(False)

    # ---- gather the filtered maps (outside the timed region) ----
    gather_ms = None
    if world > 1 and args.distribution == "scatter":
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        full_out = parallel.gather_batch(out, n_total)
        torch.cuda.synchronize(); dist.barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del full_out
    checksum = parallel.sum_over_ranks(float(out.to(torch.int64).sum().item()), coll_dev)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    mpx = n_total * W * H * args.steps / elapsed / 1e6
    P = roi[2] * roi[3]
    alg_per_launch = 20.0 * P * pairs          # (4 + 8R) bytes per ROI pixel, R = 2

    # kernel names as rocprofv3 prints them (chunk lengths follow the launchers' buckets)
    def bucket(v, bs):
        return next(b for b in bs if v <= b)
    if args.solver == "wave":
        mh = bucket((roi[2] + 63) // 64, (4, 8, 16, 20, 28, 40, 56, 60, 64))
        mv = bucket((roi[3] + 63) // 64, (2, 4, 8, 12, 18, 26, 34))
        names = {"pass_h": "wave_hpass_kernel<%d, 2, false>" % mh, "pass_v": "wave_vpass_kernel<%d, 2, 0>" % mv}
    else:
        names = {"pass_h": "exact_pass_kernel<2, 0>", "pass_v": "exact_pass_kernel<2, 0>"}

    def pass_stats(classes):
        d = [prof[k] for k in classes if k in prof]
        n = sum(x["launches"] for x in d)
        ms = sum(x["total_ms"] for x in d) / max(n, 1)
        return n, ms
    # dominant kernel = the solve-pass kernel with the largest share of the step; both passes of the
    # exact solver are one kernel, the wave solver has a row kernel and a column kernel
    if names["pass_h"] == names["pass_v"]:
        dom_classes = ["pass_h", "pass_v"]
    else:
        dom_classes = [max(("pass_h", "pass_v"), key=lambda k: prof.get(k, {}).get("total_ms", 0.0))]
    launches, avg_ms = pass_stats(dom_classes)
    achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath)).get("%s_cfg%d" % (args.solver, args.config), {})
            per_pair = t.get(names[dom_classes[0]], {}).get("bytes_per_pair")
            traffic = None if per_pair is None else per_pair * pairs
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": names[dom_classes[0]], "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launches": launches, "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": alg_per_launch}
    for name in ("pass_h", "pass_v"):
        if name in prof and prof[name]["launches"]:
            d = prof[name]
            ms = d["total_ms"] / d["launches"]
            roofline["row_pass" if name == "pass_h" else "col_pass"] = {
                "kernel": names[name], "launches": d["launches"], "avg_launch_ms": round(ms, 4),
                "achieved": round(alg_per_launch / (ms * 1e-3) / 1e9, 1),
                "frac": round(alg_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "moved_GBs": round(d["moved_bytes"] / d["launches"] / (ms * 1e-3) / 1e9, 1)}
    kernels = {k: {"launches": v["launches"], "ms_per_step": round(v["total_ms"] / args.steps, 4),
                   "alg_GBs": round(v["alg_bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1),
                   "moved_GBs": round(v["moved_bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1)} for k, v in prof.items()}

    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        ncpu = min(pairs, 2)
        cpu = cpu_baseline(view[:ncpu].cpu().numpy(), dl[:ncpu].cpu().numpy(), dr[:ncpu].cpu().numpy(), roi, radius,
                           args.cpu_seconds)

    pipeline = None
    if world == 1 and args.matcher_pairs > 0:
        try:
            nd = max(16, (roi[0] + 15) // 16 * 16)           # the config's ROI x is its numDisparities
            pipeline = views_to_filtered(adf, view, roi, radius, min(args.matcher_pairs, pairs), min(nd, 256), 15)
        except Exception as e:                               # the extra leg must never cost the bench line
            pipeline = {"error": str(e)}

    F = W * H
    b_alg_pair = 10.0 * F + (ch + 8 + 120) * P  # I/O + weights + 6 passes
    line = {
        "metric": "filtered Mpixels/s (+ achieved HBM GB/s) on 4K disparity, 1/2/4/8 GPUs vs CPU ref",
        "value": round(mpx, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %d: %d pairs/GPU of %dx%d, ROI %s, 8UC%d guide, lambda 8000 sigma 1.5, "
                               "3 FGS iterations, LRC confidence on" % (args.config, pairs, W, H, list(roi), ch),
                   "pairs_per_gpu": pairs, "total_pairs": n_total, "solver": args.solver,
                   "distribution": args.distribution if world > 1 else "resident",
                   "parallelism": "batch-sharded x%d" % world},
        "roofline": roofline, "cpu_baseline": cpu,
        "whole_call_alg_GBs": round(b_alg_pair * n_total * args.steps / elapsed / 1e9, 1),
        "kernels": kernels, "checked": checked, "checksum": checksum,
        "scatter_ms": None if scatter_ms is None else round(scatter_ms, 2),
        "gather_ms": None if gather_ms is None else round(gather_ms, 2),
        "workspace_GB": round(f.workspaceBytes() / 1e9, 2),
        "views_to_filtered": pipeline,
    }
    if cpu:
        line["speedup_vs_cpu"] = round(mpx / cpu["value"], 1)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
