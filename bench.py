#!/usr/bin/env python3
"""bench.py -- DisparityWLSFilter throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N worker processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine)

One "step" = one DisparityWLSFilter::filter call over this rank's batch of synthetic stereo pairs
(BASELINE config 3: 64 pairs of 3840x2160 per GPU, ROI (256,0,3584,2160), 8UC3 guide, lambda 8000,
sigma 1.5, 3 FGS iterations, LRC confidence on), inputs and outputs resident in HBM.  Pairs are
independent, so N GPUs filter N x 64 pairs (weak scaling) with no collective on the data path; RCCL
carries a few scalar all-reduces (max time, checksum, check flags) and, with --distribution scatter, the
batch scatter / gather around the timed region plus an extra pipelined scatter -> filter -> gather leg.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement):
  value     = whole-job filtered Mpixels/s (full-frame W*H per pair), max-over-ranks time
  roofline  = dominant kernel (the Thomas-solve pass) priced in ALGORITHMIC bytes:
              (4+8R)*P bytes per launch / mean launch duration from HIP events on the launch stream
  cpu_baseline = the CPU oracle (a port of the reference path) timed on this box's host cores

Launching.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a worker.
A plain `python bench.py --gpus N` with N > 1 becomes a launcher instead: it starts N fresh worker
processes (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT set) BEFORE anything in it
has touched a GPU -- the launcher never imports torch --, relays rank 0's single JSON line and exits
non-zero if any worker does.  Nothing is ever exec'ed from a process that has initialised the GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args(argv=None):
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
    ap.add_argument("--sub-batches", type=int, default=4,
                    help="--distribution scatter: sub-batches of the extra pipelined scatter -> filter -> gather leg")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of the first and last pair")
    ap.add_argument("--matcher-pairs", type=int, default=4,
                    help="pairs of the extra views -> matcher -> filter leg (SURVEY 8f N4; N=1 only, 0 = skip)")
    ap.add_argument("--natural-pairs", type=int, default=8,
                    help="pairs of the extra leg on a natural-image guide (0 = skip; N = 1 only)")
    ap.add_argument("--next-rows", type=int, default=16,
                    help="pairs of the extra legs on SURVEY 8(f)'s rows N1 (down-scaled path) and N2 (generic FGS as the "
                         "reference's perf test calls it); 0 = skip; N = 1 only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal without a GPU: gloo, no filter call, value 0 (tests)")
    ap.add_argument("--roi", default="config",
                    help="ROI of the filter call: 'config' (the BASELINE config's SGBM-factory ROI), 'bm' (what "
                         "createDisparityWLSFilter derives from the sample's StereoBM, block 15: x = numDisparities + 7, "
                         "y = 7, radius 5 -- DF.cpp:401-402), or x,y,w,h")
    ap.add_argument("--radius", type=int, default=None, help="depth-discontinuity radius (default: the ROI choice's)")
    ap.add_argument("--rccl-legs", choices=["auto", "off"], default="auto",
                    help="N>1 with --distribution local: after the timed region also run the batch scatter / gather and "
                         "the pipelined scatter -> filter -> gather leg over RCCL (never part of `value`)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launch: seconds before workers are stopped (below the driver's own limit, so that a hung "
                         "rendezvous is diagnosed here)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# self-launch (N > 1 without torchrun)
# ---------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_workers(args, argv):
    """Start args.gpus worker processes of this script, one per GPU; relay rank 0's stdout; propagate failure.

    This process stays GPU-free (no torch import, no HIP call), so starting children is an ordinary spawn, not
    a re-exec of a process that owns a GPU context."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                ADF_BENCH_WORKER="1")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this pool (RCCL needs it)
    env0.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cap = tempfile.TemporaryFile(mode="w+")                # rank 0's stdout = the JSON line
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=cap if r == 0 else sys.stderr, stderr=sys.stderr, cwd=os.getcwd()))
    deadline = time.time() + args.launch_timeout
    failed = None
    live = list(range(n))
    while live and failed is None:
        time.sleep(0.2)
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.remove(r)
            if rc != 0 and failed is None:
                failed = (r, rc)
        if failed is None and live and time.time() > deadline:
            failed = (live[0], 124)
    if failed is not None:                                 # stop exactly the processes started above
        for r in live:
            procs[r].terminate()
        t_end = time.time() + 15
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write("bench launcher: rank %d exited with code %d\n" % failed)
    cap.seek(0)
    sys.stdout.write(cap.read())
    sys.stdout.flush()
    return 0 if failed is None else (failed[1] if failed[1] > 0 else 1)


# ---------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------
def cpu_baseline(view, dl, dr, roi, radius, seconds):
    """Time the CPU oracle like perf_disparity_wls_filter.cpp:86-90 (filter built inside the loop).

    Reported: the 1-thread rate and the rate at the fastest stripe count (= thread count, DF.cpp:158; not known
    in advance, so a few candidates share the budget), each for the scalar evaluation order (process_row,
    FGS.cpp:439-464) and for the order the reference's default SIMD build takes (FGS.cpp:251-437, 516-548)."""
    import oracle

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = [1] + sorted({t for t in (8, 16, 32, 64, 128, avail) if 1 < t <= avail})
    n = view.shape[0]
    H, W = dl.shape[1:]
    orders = (("scalar", oracle.ORDER_SCALAR), ("ref_simd", oracle.ORDER_REF_SIMD))
    slot = seconds / (len(cands) * len(orders))
    best, one, tried = None, {}, []
    for oname, order in orders:
        for threads in cands:
            p = oracle.default_params(sigma_color=1.5, disc_radius=radius, threads=threads, order=order)
            p.lambda_ = 8000.0
            if threads > 1:
                oracle.wls_filter(dl[0], view[0], dr[0], roi, p, want_conf=True)  # warm-up cycle
            cycles, t0 = 0, time.perf_counter()
            while True:
                k = cycles % n
                oracle.wls_filter(dl[k], view[k], dr[k], roi, p, want_conf=True)
                cycles += 1
                el = time.perf_counter() - t0
                if el >= slot or cycles >= 200:
                    break
            rate = cycles * W * H / el / 1e6
            tried.append("%s/%d thr: %.1f" % (oname, threads, rate))
            if threads == 1:
                one[oname] = round(rate, 2)
            if best is None or rate > best[0]:
                best = (rate, threads, oname)
    return {
        "value": round(best[0], 3), "unit": "Mpixels/s", "cores": best[1], "kind": "port", "order": best[2],
        "one_thread": one,
        "sample": "DisparityWLSFilter on %d pair(s) of %dx%d, same inputs/params as the GPU run; oracle/adf_oracle.c "
                  "(pooled pthread stripes end to end like ParallelLoopBody); %d logical CPUs visible; Mpixels/s by "
                  "evaluation order / threads: %s" % (n, W, H, avail, "; ".join(tried)),
    }


def views_to_filtered(xi, view, n, num_disp, block, matcher="bm", check=True, half=False):
    """Extra leg, outside the timed region and not part of `value`: a device matcher (both views) feeding the
    filter, all stages on torch's stream, inputs resident (EXPERIMENTS.md section 10).

    check: pair 0 against the oracle's pipeline -- the matcher's maps on the top rows of the frame (a block / 3-way
    semi-global match of row y reads no row below y + blockSize/2 + 2, so the oracle runs on a crop and finishes in
    seconds), then the filter of the device maps against the oracle's filter of the same maps on the whole frame."""
    import torch
    left = view[:n, :, :, 0].contiguous() if view.dim() == 4 else view[:n].contiguous()
    if half:
        # the sample's DEFAULT pipeline (samples/disparity_filtering.cpp:130-141,151-189): the matcher runs on HALF-size
        # views with half the disparity range (window 7), the filter on the full view.  (The 8U resize / gray conversion
        # of the views is OpenCV's and upstream of the path: the half-size views are made here, outside the timing.)
        lf = left.to(torch.float32)
        left = ((lf[:, 0::2, 0::2] + lf[:, 0::2, 1::2] + lf[:, 1::2, 0::2] + lf[:, 1::2, 1::2] + 2.0) * 0.25).floor().to(torch.uint8).contiguous()
        num_disp = max(16, (num_disp // 2 + 15) // 16 * 16)
    right = torch.roll(left, -min(num_disp // 3, 60), 2).contiguous()
    if matcher == "sgbm":
        lm = xi.StereoSGBM.create(0, num_disp, block)
        lm.setP1(24 * block * block); lm.setP2(96 * block * block)          # samples/disparity_filtering.cpp:166-170
        lm.setMode(xi.StereoSGBM.MODE_SGBM_3WAY)
    else:
        lm = xi.StereoBM.create(num_disp, block)
    wls = xi.createDisparityWLSFilter(lm)                    # DF.cpp:386-414 (forces texture / uniqueness tests off)
    rm = xi.createRightMatcher(lm)                           # DF.cpp:417-449
    wls.setLambda(8000.0); wls.setSigmaColor(1.5)
    H, W = left.shape[1:]
    VH, VW = view.shape[1:3]
    dl = torch.empty((n, H, W), dtype=torch.int16, device=left.device); dr = torch.empty_like(dl)
    out = torch.empty((n, VH, VW), dtype=torch.int16, device=left.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(2):                                      # first round warms the workspaces up
        ev[0].record()
        if matcher == "sgbm":
            lm.compute(left, right, dl)
            rm.compute(right, left, dr)
        else:
            lm.computeBoth(left, right, dl, dr)              # = lm.compute(left, right), rm.compute(right, left)
        ev[1].record()
        wls.filter(dl, view[:n], out, dr)
        ev[2].record()
    torch.cuda.synchronize()
    fl = wls.getLastPath()
    res = {"matcher": matcher, "pairs": n, "num_disparities": num_disp, "block_size": block, "roi": list(wls.getROI()),
           "radius": wls.getDepthDiscontinuityRadius(),
           "path": {"conf_band_kernel": bool(fl & xi.PATH_CONF_BAND), "fused_first_row_pass": bool(fl & xi.PATH_FUSED_FIRST_PASS),
                    "first_pass_interpolates_low_resolution_maps": bool(fl & xi.PATH_SCALED_FUSED)},
           "maps": "%dx%d" % (W, H), "view": "%dx%d" % (VW, VH),
           "matcher_ms_per_pair": round(ev[0].elapsed_time(ev[1]) / n, 4),
           "filter_ms_per_pair": round(ev[1].elapsed_time(ev[2]) / n, 4),
           "Mpixels_per_s": round(n * VH * VW / (ev[0].elapsed_time(ev[2]) * 1e-3) / 1e6, 1),
           "note": "left + right view matcher then the filter, each one call for the batch; not part of `value`"}
    if check:
        import numpy as np
        import oracle
        rows, crop = min(64, H), min(H, 96)
        L, R = left[0, :crop].cpu().numpy(), right[0, :crop].cpu().numpy()
        if matcher == "sgbm":
            el = oracle.sgbm_compute(L, R, num_disp, block, 0, 24 * block * block, 96 * block * block, lm.getPreFilterCap(), 0)
            er = oracle.sgbm_compute(R, L, num_disp, block, -num_disp + 1, 24 * block * block, 96 * block * block, rm.getPreFilterCap(), 0)
        else:
            el = oracle.bm_compute(L, R, num_disp, block, 0)
            er = oracle.bm_compute(R, L, num_disp, block, -num_disp + 1)
        gl, gr = dl[0].cpu().numpy(), dr[0].cpu().numpy()
        maps_ok = bool(np.array_equal(gl[:rows], el[:rows]) and np.array_equal(gr[:rows], er[:rows]))
        p = oracle.default_params(sigma_color=1.5, disc_radius=wls.getDepthDiscontinuityRadius(),
                                  threads=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)
        p.lambda_ = 8000.0
        if half:
            exp, exp_conf = oracle.wls_filter_scaled(gl, view[0].cpu().numpy(), gr, wls.getROI(), p)
        else:
            exp, exp_conf = oracle.wls_filter(gl, view[0].cpu().numpy(), gr, wls.getROI(), p)
        diff = np.abs(out[0].cpu().numpy().astype(np.int64) - exp)
        conf_ok = bool(np.array_equal(wls.getConfidenceMap(0).cpu().numpy(), exp_conf))
        res["checked"] = bool(maps_ok and conf_ok and diff.max() <= 1 and diff.mean() <= 1 / 256)
        res["check"] = {"pair": 0, "matcher_maps_bit_exact_rows": [0, rows] if maps_ok else False, "confidence_bit_exact": conf_ok,
                        "disparity_max_abs_lsb": int(diff.max()), "disparity_mean_abs_lsb": float(diff.mean()),
                        "valid_fraction_left_map": float((gl[:, num_disp:] >= 0).mean())}
        # (half: the filter's ROI is in the maps' coordinates, DF.cpp:229-230; the oracle's scaled call takes it the same way)
    else:
        res["checked"] = None
    return res


def check_pairs(f, out, view, dl, dr, roi, radius, solver, which, threads):
    """Pairs `which` of this rank's batch against the CPU oracle (the checker, never the thing measured)."""
    import numpy as np
    import oracle
    p = oracle.default_params(sigma_color=1.5, disc_radius=radius, threads=threads)
    p.lambda_ = 8000.0
    res, ok = [], True
    for k in which:
        exp, exp_conf = oracle.wls_filter(dl[k].cpu().numpy(), view[k].cpu().numpy(), dr[k].cpu().numpy(), roi, p)
        got = out[k].cpu().numpy().astype(np.int64)
        conf_ok = bool(np.array_equal(f.getConfidenceMap(k).cpu().numpy(), exp_conf))
        diff = np.abs(got - exp)
        res.append({"pair": int(k), "confidence_bit_exact": conf_ok, "disparity_max_abs_lsb": int(diff.max()),
                    "disparity_mean_abs_lsb": float(diff.mean())})
        ok = ok and conf_ok and (diff.max() == 0 if solver == "exact" else (diff.max() <= 1 and diff.mean() <= 1 / 256))
    return ok, res


def worker(args):
    # Libraries under us write to stdout (gloo's connection banner, RCCL with NCCL_DEBUG set): the one JSON line
    # goes to a private copy of the original stdout and everything else any library prints lands on stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from addingdisparityfiltering_amd import parallel, synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != args.gpus:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d: using the launched world size\n" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # ADF_BENCH_BACKEND=gloo is a rehearsal mode for one-GPU boxes: the ranks share the visible GPUs
    # (rank r uses device r % device_count) and the scalar all-reduces run on CPU tensors, so the whole
    # N > 1 flow executes except RCCL itself.  The driver's runs use the default, nccl (= RCCL).
    backend = "gloo" if dry else os.environ.get("ADF_BENCH_BACKEND", "nccl")
    if dry:
        dev = torch.device("cpu")
    else:
        dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            if args.distribution == "scatter" and not dry:
                raise SystemExit("the scatter distribution moves device tensors: it needs the nccl backend")
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == world and dist.get_rank() == rank

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if dry and os.environ.get("ADF_BENCH_TEST_FAIL_RANK") == str(rank):     # tests/test_bench_launcher.py: a rank that dies
        sys.stderr.write("bench: rank %d: failing on request (test hook, --dry-run only)\n" % rank)
        os._exit(7)
    cfg = synthetic.CONFIGS[args.config]
    W, H, roi, ch, radius = cfg["W"], cfg["H"], cfg["roi"], cfg["channels"], cfg["radius"]
    roi_kind = args.roi
    if args.roi == "bm":
        # createDisparityWLSFilter on a StereoBM with the sample's full-size block 15 (samples/disparity_filtering.cpp:71,
        # DF.cpp:401-402): ROI = (minD + numD + wsize/2, wsize/2, ...), radius ceil(0.33 * wsize)
        nd, half = roi[0], 7
        roi = (nd + half, half, W - nd - 2 * half, H - 2 * half)
        radius = 5
    elif args.roi != "config":
        roi = tuple(int(v) for v in args.roi.split(","))
        if len(roi) != 4 or roi[0] < 0 or roi[1] < 0 or roi[2] <= 0 or roi[3] <= 0 or roi[0] + roi[2] > W or roi[1] + roi[3] > H:
            raise SystemExit("--roi x,y,w,h must lie inside the %dx%d frame" % (W, H))
        roi_kind = "custom"
    if args.radius is not None:
        radius = args.radius
    pairs = args.pairs
    n_total = pairs * world
    vshape = (H, W, ch) if ch > 1 else (H, W)

    adf = f = view = dl = dr = None
    full = [None, None, None]
    scatter_ms = None
    if dry:
        out = torch.full((pairs, 4, 4), rank + 1, dtype=torch.int16)
    else:
        import addingdisparityfiltering_amd as adf
        # ---- inputs, resident in HBM before the timed region ----
        base_seed = synthetic.seed_for(args.config, 0)
        if world > 1 and args.distribution == "scatter":
            if rank == 0:
                full = list(synthetic.make_artificial_batch_torch(n_total, W, H, ch, base_seed, cfg["rect_disparity"], dev))
            sync(); dist.barrier()
            t0 = time.perf_counter()
            view = parallel.scatter_batch(full[0], n_total, vshape, torch.uint8, dev)
            dl = parallel.scatter_batch(full[1], n_total, (H, W), torch.int16, dev)
            dr = parallel.scatter_batch(full[2], n_total, (H, W), torch.int16, dev)
            sync(); dist.barrier()
            scatter_ms = (time.perf_counter() - t0) * 1e3
            view, dl, dr = view.contiguous().clone(), dl.contiguous().clone(), dr.contiguous().clone()
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

    def step():
        if dry:
            time.sleep(0.01)
        else:
            f.filter(dl, view, out, dr, roi)

    for _ in range(max(args.warmup, 0)):
        step()
    sync()

    # ---- correctness of what is being timed: first and last pair of EVERY rank against the CPU oracle ----
    checked = None
    if not args.no_check and not dry:
        import oracle
        if rank == 0:
            oracle.build()            # one rank compiles the checker if its .so is stale; the others wait
        if world > 1:
            dist.barrier()
        if args.warmup <= 0:
            step(); sync()
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
        ok, checked = check_pairs(f, out, view, dl, dr, roi, radius, args.solver,
                                  sorted({0, pairs - 1}), max(1, cores // world))
        all_ok = parallel.min_over_ranks(1.0 if ok else 0.0, coll_dev) > 0.5
        if not ok:
            sys.stderr.write("bench: rank %d: GPU result differs from the oracle: %s\n" % (rank, checked))
        if not all_ok:
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    if f is not None:
        f.enableProfiling(True)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    my_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, coll_dev)
    per_rank_ms = [round(v / args.steps * 1e3, 3) for v in parallel.gather_scalars(my_elapsed, coll_dev)]
    prof = f.readProfile() if f is not None else {}
    if f is not None:
        f.enableProfiling(False)

    # ---- gather the filtered maps (outside the timed region), then the pipelined serving-shaped leg ----
    gather_ms, pipelined = None, None
    if world > 1 and args.distribution == "scatter" and not dry:
        sync(); dist.barrier()
        t1 = time.perf_counter()
        full_out = parallel.gather_batch(out, n_total)
        sync(); dist.barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del full_out
        pipelined, _full_out = parallel.pipelined_scatter_filter_gather(
            full if rank == 0 else None, n_total, [vshape, (H, W), (H, W)], [torch.uint8, torch.int16, torch.int16],
            (H, W), torch.int16, dev, lambda v, a, b, o: f.filter(a, v, o, b, roi), args.sub_batches)
        del _full_out
        pipelined["resident_compute_ms"] = round(elapsed / args.steps * 1e3, 3)
        pipelined["blocking_scatter_plus_gather_ms"] = round(scatter_ms + gather_ms, 3)
        pipelined["exposed_transfer_ms"] = round(pipelined["total_ms"] - pipelined["resident_compute_ms"], 3)
        pipelined["hidden_transfer_ms"] = round(max(0.0, scatter_ms + gather_ms - pipelined["exposed_transfer_ms"]), 3)
    full = None
    checksum = parallel.sum_over_ranks(float(out.to(torch.int64).sum().item()), coll_dev)
    # one flat list for the line: {"rank", "pair", ...} for the first and last pair of every rank
    checked_all = None
    if checked is not None:
        per_rank = parallel.gather_objects(checked) if world > 1 else [checked]
        checked_all = [dict(c, rank=r) for r, lst in enumerate(per_rank) for c in (lst or [])]
    # who ran where: one record per rank, so that N ranks on N DISTINCT devices is visible in the line itself
    ident = device_identity(torch, dev, dry, rank, local_rank)
    ranks_info = parallel.gather_objects(ident) if world > 1 else [ident]
    path = None
    if f is not None:
        fl = f.getLastPath()
        path = {"conf_band_kernel": bool(fl & adf.PATH_CONF_BAND), "fused_first_row_pass": bool(fl & adf.PATH_FUSED_FIRST_PASS)}

    line = None
    if rank == 0:
        line = build_line(args, dry, world, backend, n_total, pairs, W, H, ch, roi, roi_kind, radius, elapsed, per_rank_ms,
                          prof, checked_all, checksum, ranks_info, path, scatter_ms, gather_ms, pipelined,
                          None if f is None else round(f.workspaceBytes() / 1e9, 2))

    import threading

    emit_lock, emitted = threading.Lock(), []

    def emit():
        with emit_lock:                                      # the watchdog thread and the main thread may both get here:
            if emitted or line is None:                      # the line is written once
                return
            emitted.append(True)
            json_out.write(json.dumps(line) + "\n")
            json_out.flush()

    # ---- extra legs.  None of them is part of `value`, none may cost the line: every one runs inside try/except, and
    # at N > 1 -- where a stuck transfer would hang all ranks -- under a watchdog that prints the line as it stands
    # (marked "extra_legs_hung": true) and ends the rank with status 5, so that a hung transfer is never reported as a
    # clean run: the launcher relays the line AND returns non-zero.
    watchdog = None
    if world > 1:
        def bail():
            with emit_lock:
                already = bool(emitted)
                if line is not None and not already:
                    line["extra_legs_hung"] = True
                    if not isinstance(line.get("rccl_legs"), dict):
                        line["rccl_legs"] = {}
                    line["rccl_legs"]["watchdog"] = "extra legs did not finish in %.0f s: line printed without them" % limit
            emit()
            sys.stderr.write("bench: rank %d: extra-leg watchdog fired\n" % rank)
            sys.stderr.flush()
            os._exit(0 if already else 5)
        limit = float(os.environ.get("ADF_BENCH_WATCHDOG_S", 240.0 + max(0.0, args.cpu_seconds) * 2))
        watchdog = threading.Timer(limit, bail)
        watchdog.daemon = True
        watchdog.start()

    if dry and world > 1 and os.environ.get("ADF_BENCH_TEST_HANG_RANK") == str(rank):   # test hook: a transfer that never ends
        time.sleep(3600)
    if world > 1 and args.distribution == "local" and args.rccl_legs == "auto":
        legs = None
        try:
            if dry:
                # rehearsal with CPU tensors over gloo: same code path, tiny items, "filter" = a copy with a twist
                item = (4, 4)
                vals = lambda r: (10 * r, 11 * r + 1, 0)              # a + b - v = r + 1 = what this rank's `out` holds
                mine = [torch.full((pairs,) + item, v, dtype=torch.int16) for v in vals(rank)]
                legs = rccl_legs(parallel, torch, dist, rank, world, dev, coll_dev, n_total, pairs, [item] * 3, [torch.int16] * 3,
                                 item, torch.int16, mine, out,
                                 lambda r: [torch.full((pairs,) + item, v, dtype=torch.int16) for v in vals(r)],
                                 lambda v, a, b, o: o.copy_(a + b - v), args.sub_batches, elapsed / args.steps * 1e3, lambda: None)
            elif backend != "nccl":
                legs = {"skipped": "backend %s cannot move device tensors" % backend}
            else:
                legs = rccl_legs(parallel, torch, dist, rank, world, dev, coll_dev, n_total, pairs,
                                 [vshape, (H, W), (H, W)], [torch.uint8, torch.int16, torch.int16], (H, W), torch.int16,
                                 [view, dl, dr], out,
                                 lambda r: synthetic.make_artificial_batch_torch(pairs, W, H, ch, base_seed + r * pairs,
                                                                                 cfg["rect_disparity"], dev),
                                 lambda v, a, b, o: f.filter(a, v, o, b, roi), args.sub_batches, elapsed / args.steps * 1e3, sync)
        except Exception as e:                               # (a failure on one rank only would leave the others waiting:
            legs = {"error": "%s: %s" % (type(e).__name__, e)}    #  the watchdog ends that)
        if line is not None:
            line["rccl_legs"] = legs

    if rank == 0 and args.cpu_seconds > 0 and not dry:
        try:
            ncpu = min(pairs, 2)
            cpu = cpu_baseline(view[:ncpu].cpu().numpy(), dl[:ncpu].cpu().numpy(), dr[:ncpu].cpu().numpy(), roi, radius,
                               args.cpu_seconds)
            line["cpu_baseline"] = cpu
            line["speedup_vs_cpu"] = round(line["value"] / cpu["value"], 1)
            # which CPU figure the ratio divides by: the BEST of the thread sweep (cpu_baseline.cores threads, order in
            # cpu_baseline.order) -- not the rate at getNumThreads() = all logical CPUs, which the sweep in `sample` also lists
            line["speedup_vs_cpu_basis"] = "value / cpu_baseline.value = best of the thread sweep: %s order on %d threads" % (cpu.get("order"), cpu.get("cores"))
        except Exception as e:
            line["cpu_baseline"] = None
            line["cpu_baseline_error"] = str(e)
    if world > 1:
        dist.barrier()                                       # the other ranks wait here while rank 0 times the CPU port

    if rank == 0 and world == 1 and args.matcher_pairs > 0 and not dry:
        nd = max(16, (cfg["roi"][0] + 15) // 16 * 16)       # the config's ROI x is its numDisparities
        pipeline = {}
        for m, blk in (("bm", 15), ("sgbm", 3), ("bm_half", 7)):
            try:
                pipeline[m] = views_to_filtered(adf, view, min(args.matcher_pairs, pairs), min(nd, 256), blk, m.split("_")[0],
                                                not args.no_check, half=m.endswith("_half"))
            except Exception as e:                           # the extra leg must never cost the bench line
                pipeline[m] = {"error": str(e)}
        line["views_to_filtered"] = pipeline

    if rank == 0 and world == 1 and not dry and args.natural_pairs > 0:
        try:
            line["natural_guide"] = natural_guide_leg(adf, torch, dev, dl, dr, roi, radius, ch, min(args.natural_pairs, pairs),
                                                      args.solver, not args.no_check)
        except Exception as e:                               # the extra leg must never cost the bench line
            line["natural_guide"] = {"error": str(e)}

    if rank == 0 and world == 1 and not dry and args.next_rows > 0:
        try:
            line["next_rows"] = next_rows_leg(adf, torch, dev, synthetic, cfg, W, H, ch, base_seed, min(args.next_rows, pairs),
                                              radius, not args.no_check)
        except Exception as e:                               # the extra leg must never cost the bench line
            line["next_rows"] = {"error": str(e)}

    if watchdog is not None:
        watchdog.cancel()
    emit()
    if world > 1:
        dist.destroy_process_group()


def next_rows_leg(adf, torch, dev, synthetic, cfg, W, H, ch, seed, n, radius, check):
    """SURVEY 8(f) rows N1 and N2 in the driver's own record (never part of `value`).
    N1, the down-scaled path (DF.cpp:224-227, 239-247, 268-277; perf_disparity_wls_filter.cpp:76-83 and the sample's
    default): `n` views of the config's size with HALF-size disparity maps, ROI halved.
    N2, fastGlobalSmootherFilter as the reference's perf test calls it (perf_fgs_filter.cpp:55-76): 1280x720, guide 8UC3,
    source 32FC3, a filter made AND one filter call per cycle with a fresh lambda / sigma every cycle.
    One result of each is compared with the oracle."""
    import time

    import numpy as np

    res = {}
    # ---- N1 ----
    dW, dH = W // 2, H // 2
    view = synthetic.make_artificial_batch_torch(n, W, H, ch, seed + 7, cfg["rect_disparity"], dev)[0]
    _, dl, dr = synthetic.make_artificial_batch_torch(n, dW, dH, ch, seed + 7, cfg["rect_disparity"] // 2, dev)
    r = cfg["roi"]
    roi = (r[0] // 2, r[1] // 2, r[2] // 2, r[3] // 2)
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    out = None
    for _ in range(2):
        out = f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    steps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f.filter(dl, view, out, dr, roi)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n1 = {"views": "%d x %dx%d, %d channel(s)" % (n, W, H, ch), "maps": "%dx%d" % (dW, dH), "roi_in_map_coordinates": list(roi),
          "radius": radius, "ms_per_call": round(ms, 3), "Mpixels_per_s_view_pixels": round(n * W * H / ms / 1e3, 1)}
    if check:
        import oracle
        p = oracle.default_params(threads=min(32, os.cpu_count() or 1), sigma_color=1.5, disc_radius=radius)
        exp, exp_conf = oracle.wls_filter_scaled(dl[0].cpu().numpy(), view[0].cpu().numpy(), dr[0].cpu().numpy(), roi, p)
        d = np.abs(out[0].cpu().numpy().astype(np.int32) - exp.astype(np.int32))
        n1["checked"] = {"pair": 0, "confidence_bit_exact": bool(np.array_equal(f.getConfidenceMap(0).cpu().numpy(), exp_conf)),
                         "disparity_max_abs_lsb": int(d.max()), "disparity_mean_abs_lsb": float(d.mean())}
    res["down_scaled_path"] = n1
    del f, view, dl, dr, out
    # ---- N2 ----
    w2, h2, cycles = 1280, 720, 10
    rng = np.random.default_rng(seed)
    guide = torch.from_numpy(rng.integers(0, 256, (h2, w2, 3), dtype=np.uint8)).to(dev)
    src = torch.from_numpy((rng.random((h2, w2, 3), dtype=np.float32) * 255).astype(np.float32)).to(dev)
    dst = torch.empty_like(src)
    lam = [float(rng.uniform(500.0, 10000.0)) for _ in range(cycles + 2)]
    sig = [float(rng.uniform(1.0, 100.0)) for _ in range(cycles + 2)]
    for k in range(2):
        adf.fastGlobalSmootherFilter(guide, src, lam[k], sig[k], dst=dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(cycles):
        adf.fastGlobalSmootherFilter(guide, src, lam[2 + k], sig[2 + k], dst=dst)
    torch.cuda.synchronize()
    one_shot = (time.perf_counter() - t0) / cycles * 1e3
    ff = adf.createFastGlobalSmootherFilter(guide, lam[-1], sig[-1])
    ff.filter(src, dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(cycles):
        ff.filter(src, dst)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / cycles * 1e3
    n2 = {"image": "%dx%d, guide 8UC3, source 32FC3" % (w2, h2), "cycles": cycles,
          "create_plus_filter_ms_per_call": round(one_shot, 3), "filter_alone_ms_per_call": round(alone, 3),
          "Mpixels_per_s_create_plus_filter": round(w2 * h2 / one_shot / 1e3, 1)}
    if check:
        import oracle
        got = adf.fastGlobalSmootherFilter(guide, src, lam[-1], sig[-1], solver=adf.SOLVER_EXACT).cpu().numpy()
        exp = oracle.fgs_filter(guide.cpu().numpy(), src.cpu().numpy(), lam[-1], sig[-1], threads=min(16, os.cpu_count() or 1))
        n2["checked"] = {"exact_solver_bit_exact": bool(np.array_equal(got, exp)),
                         "wave_solver_max_abs_diff": float(np.abs(dst.cpu().numpy() - exp).max())}
    res["fgs_one_shot"] = n2
    # ---- BASELINE config 5 as the stream it is (tools/stream_cfg5.py): one 1242x375 frame per call on K handles / streams,
    # sustained rate and per-frame latency, and the same frames as micro-batches
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("adf_stream_cfg5", os.path.join(ROOT, "tools", "stream_cfg5.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        res["config5_stream"] = mod.measure(adf, torch, dev, synthetic, frames=1024, check=check)
    except Exception as e:
        res["config5_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ... and issued from C++ through the C-ABI (tools/stream_cfg5.cpp, built by __graft_entry__.build()): a child process,
    # with HIP's default hardware-queue count and with one queue per stream
    res["config5_stream_cpp"] = config5_stream_cpp()
    res["note"] = "SURVEY 8(f) rows N1 / N2 on the final kernels, and config 5 one frame per call; not part of `value`"
    return res


def config5_stream_cpp(frames=1024):
    import subprocess
    exe = os.path.join(ROOT, "tools", "stream_cfg5_cpp")
    if not os.path.exists(exe):
        return {"skipped": "tools/stream_cfg5_cpp is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {"frames": frames, "columns": "sustained Mpixels/s (best of 3), frames/s, host issue us/frame, wall us/frame",
           "note": "one frame per adf_wls_filter_device call from ONE C++ host thread, K handles on K streams; never part of `value`"}
    # (HIP maps streams of one priority onto 4 hardware queues by default, and two streams then share one: either raise
    # the queue count in the environment or spread the streams over the three priority levels)
    for label, extra, prio in (("default_hw_queues", {}, "0"), ("default_hw_queues_stream_priorities_cycling", {}, "2"),
                               ("GPU_MAX_HW_QUEUES=6", {"GPU_MAX_HW_QUEUES": "6"}, "0")):
        env = dict(os.environ); env.update(extra)
        try:
            p = subprocess.run([exe, str(frames), "8", prio], env=env, capture_output=True, text=True, timeout=120)
        except Exception as e:
            out[label] = {"error": "%s: %s" % (type(e).__name__, e)}
            continue
        if p.returncode != 0:
            out[label] = {"error": "exit code %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
            continue
        rows = {}
        for ln in p.stdout.splitlines():
            f = [q.strip() for q in ln.split("|")]
            if len(f) == 6 and f[0].isdigit():
                rows["K=%s %s" % (f[0], f[1])] = [float(f[2]), float(f[3]), float(f[4]), float(f[5])]
        out[label] = rows
    return out


def natural_guide_leg(adf, torch, dev, dl, dr, roi, radius, ch, n, solver, check):
    """The same filter call with a NATURAL-IMAGE guide: the reference's KITTI fixture (modules/stereo/testdata/imgKittyl.bmp,
    a data fixture under tests/golden/) tiled to the config's frame, as `ch` channels (the gray value plus a little
    per-channel noise).  `value` is quoted on MakeArtificialExample's scene, as the reference's perf test is
    (perf_disparity_wls_filter.cpp:95-167); that scene's only strong colour edges are one rectangle's border, while
    natural images send a few per cent of the edge-weight look-ups past the table head cached in LDS -- this leg shows
    the rate there."""
    import numpy as np
    from PIL import Image

    H, W = dl.shape[1], dl.shape[2]
    im = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "kitti_left.bmp")))
    if im.ndim == 3:
        im = im[:, :, 0]
    tiled = np.ascontiguousarray(np.tile(im, ((H + im.shape[0] - 1) // im.shape[0], (W + im.shape[1] - 1) // im.shape[1]))[:H, :W])
    g1 = torch.from_numpy(tiled).to(dev)[None].expand(n, H, W).contiguous()
    if ch == 3:
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        guide = (g1[..., None].float() + 2.0 * torch.randn((n, H, W, 3), generator=gen, device=dev)).round_().clamp_(0, 255).to(torch.uint8)
        dh = ((guide[0, :, 1:].int() - guide[0, :, :-1].int()) ** 2).sum(-1)
    else:
        guide = g1
        dh = (guide[0, :, 1:].int() - guide[0, :, :-1].int()) ** 2
    beyond = float((dh >= 2048).float().mean().item())
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    f.setSolver(adf.SOLVER_WAVE if solver == "wave" else adf.SOLVER_EXACT)
    dln, drn = dl[:n].contiguous(), dr[:n].contiguous()
    out = None
    for _ in range(2):
        out = f.filter(dln, guide, out, drn, roi)
    torch.cuda.synchronize()
    steps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f.filter(dln, guide, out, drn, roi)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    res = {"guide": "tests/golden/kitti_left.bmp tiled to %dx%d, %d channel(s)" % (W, H, ch), "pairs": n,
           "table_indices_beyond_lds_head": round(beyond, 4), "ms_per_step": round(ms, 3),
           "Mpixels_per_s": round(n * W * H / ms / 1e3, 1),
           "note": "same call as `value` on a natural-image guide; not part of `value`"}
    if check:
        import oracle
        p = oracle.default_params(threads=min(32, os.cpu_count() or 1), sigma_color=1.5, disc_radius=radius)
        exp, exp_conf = oracle.wls_filter(dln[0].cpu().numpy(), guide[0].cpu().numpy(), drn[0].cpu().numpy(), roi, p)
        got = out[0].cpu().numpy()
        d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
        res["checked"] = {"pair": 0, "confidence_bit_exact": bool(np.array_equal(f.getConfidenceMap(0).cpu().numpy(), exp_conf)),
                          "disparity_max_abs_lsb": int(d.max()), "disparity_mean_abs_lsb": float(d.mean())}
    return res


def device_identity(torch, dev, dry, rank, local_rank):
    """What a reader needs to see that N ranks ran on N distinct devices."""
    info = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
    if dry:
        info.update(device="cpu", name=None, uuid=None, pci_bus_id=None)
        return info
    props = torch.cuda.get_device_properties(dev)
    info.update(device="cuda:%d" % dev.index, name=torch.cuda.get_device_name(dev),
                uuid=str(getattr(props, "uuid", None)), multi_processor_count=getattr(props, "multi_processor_count", None),
                total_memory_GB=round(props.total_memory / 1e9, 1))
    pci = None
    try:
        from addingdisparityfiltering_amd import _lib
        pci = _lib.device_pci_bus_id(dev.index)              # hipDeviceGetPCIBusId through the C-ABI
    except Exception:
        pci = None
    if pci is None and hasattr(props, "pci_bus_id"):
        pci = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    info["pci_bus_id"] = pci
    return info


def rccl_legs(parallel, torch, dist, rank, world, dev, coll_dev, n_total, pairs, in_shapes, in_dtypes, out_shape, out_dtype,
              mine, out, make_shard, process, n_sub, resident_ms, sync):
    """The data-path exchange of SURVEY 8(e) over the process group (RCCL on a GPU node), run AFTER the timed region of a
    `--distribution local` bench so that the first multi-GPU record exercises it: rank 0 builds the whole batch (every
    rank's shard with that rank's seed), scatters it with point-to-point groups, every rank compares what arrived with the
    shard it generated itself, the filtered maps are gathered on rank 0 and compared with the all-reduced checksum, and
    the pipelined scatter -> filter -> gather leg runs on the same data."""
    full, ok_build = None, 1.0
    if rank == 0:
        try:
            shards = [mine if r == 0 else list(make_shard(r)) for r in range(world)]
            full = [torch.cat([sh[k] for sh in shards]) for k in range(len(mine))]
            del shards
        except Exception as e:                               # e.g. out of memory: agree on skipping, never a half-run exchange
            sys.stderr.write("bench: rank 0 could not build the whole batch: %s\n" % e)
            full, ok_build = None, 0.0
    if parallel.min_over_ranks(ok_build, coll_dev) < 0.5:
        return {"skipped": "rank 0 could not build the whole batch"}
    # the first point-to-point call between two ranks sets the connection up (xGMI channels, buffers): not transfer time
    warm = parallel.scatter_batch(torch.zeros((world, 64), dtype=torch.uint8, device=dev) if rank == 0 else None,
                                  world, (64,), torch.uint8, dev)
    parallel.gather_batch(warm.contiguous(), world)
    del warm
    sync(); dist.barrier()
    t0 = time.perf_counter()
    got = [parallel.scatter_batch(None if full is None else full[k], n_total, in_shapes[k], in_dtypes[k], dev)
           for k in range(len(mine))]
    sync(); dist.barrier()
    scatter_ms = (time.perf_counter() - t0) * 1e3
    same = all(bool(torch.equal(g, m)) for g, m in zip(got, mine))
    scatter_ok = parallel.min_over_ranks(1.0 if same else 0.0, coll_dev) > 0.5
    del got
    sync(); dist.barrier()
    t1 = time.perf_counter()
    full_out = parallel.gather_batch(out, n_total)
    sync(); dist.barrier()
    gather_ms = (time.perf_counter() - t1) * 1e3
    total = parallel.sum_over_ranks(float(out.to(torch.int64).sum().item()), coll_dev)
    gather_ok = None
    if rank == 0:
        gather_ok = float(full_out.to(torch.int64).sum().item()) == total
    del full_out
    pipelined, piped_out = parallel.pipelined_scatter_filter_gather(full, n_total, in_shapes, in_dtypes, out_shape, out_dtype,
                                                                    dev, process, n_sub)
    piped_ok = None
    if rank == 0:
        piped_ok = float(piped_out.to(torch.int64).sum().item()) == total
    del piped_out, full
    in_bytes = sum(int(torch.empty((), dtype=d).element_size()) * int(torch.Size(sh).numel()) for sh, d in zip(in_shapes, in_dtypes))
    out_bytes = int(torch.empty((), dtype=out_dtype).element_size()) * int(torch.Size(out_shape).numel())
    moved_in, moved_out = in_bytes * pairs * (world - 1), out_bytes * pairs * (world - 1)
    pipelined["resident_compute_ms"] = round(resident_ms, 3)
    pipelined["exposed_transfer_ms"] = round(pipelined["total_ms"] - resident_ms, 3)
    pipelined["hidden_transfer_ms"] = round(max(0.0, scatter_ms + gather_ms - pipelined["exposed_transfer_ms"]), 3)
    pipelined["output_checksum_matches"] = piped_ok
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
            "scatter_ms": round(scatter_ms, 2), "scatter_GBs_from_root": round(moved_in / max(scatter_ms, 1e-9) / 1e6, 1),
            "scattered_shards_equal_locally_generated": scatter_ok,
            "gather_ms": round(gather_ms, 2), "gather_GBs_into_root": round(moved_out / max(gather_ms, 1e-9) / 1e6, 1),
            "gathered_checksum_matches": gather_ok,
            "pipelined_scatter_filter_gather": pipelined,
            "note": "after the timed region, never part of `value`: root <-> peers point-to-point groups (7 xGMI links at N = 8)"}


def build_line(args, dry, world, backend, n_total, pairs, W, H, ch, roi, roi_kind, radius, elapsed, per_rank_ms, prof,
               checked_all, checksum, ranks_info, path, scatter_ms, gather_ms, pipelined, workspace_gb):
    """Everything of the JSON line that is known when the timed region and the pair checks are done (rank 0)."""
    mpx = 0.0 if dry else n_total * W * H * args.steps / elapsed / 1e6
    P = roi[2] * roi[3]
    alg_per_launch = 20.0 * P * pairs          # (4 + 8R) bytes per ROI pixel, R = 2

    # kernel names as rocprofv3 prints them (chunk lengths follow the launchers' buckets)
    def bucket(v, bs):
        return next(b for b in bs if v <= b)
    if args.solver == "wave":
        # (rows beyond 4096 columns: two wavefronts per row; columns beyond 2176 rows: half strips of 128 chunks)
        if roi[2] > 4096:
            hname = "wave_hpass_kernel<%d, 2, 0, 2>" % bucket((roi[2] + 127) // 128, (40, 48, 56, 60, 64))
        else:
            hname = "wave_hpass_kernel<%d, 2, 0, 1>" % bucket((roi[2] + 63) // 64, (4, 8, 16, 20, 28, 40, 56, 60, 64))
        if roi[3] > 2176:
            vname = "wave_vpass_kernel<%d, 2, 0, 8, 128, false>" % bucket((roi[3] + 127) // 128, (20, 26, 34))
        else:
            vname = "wave_vpass_kernel<%d, 2, 0, 16, 64, false>" % bucket((roi[3] + 63) // 64, (2, 4, 8, 12, 18, 26, 34))
        names = {"pass_h": hname, "pass_v": vname}
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
    # HBM traffic per launch is NOT measured by this run (PMC counters need their own rocprofv3 passes): it is the
    # committed figure of profiles/pmc_traffic.json (bytes per pair x pairs); its provenance rides along so that a
    # figure older than the kernels is visible
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            t = tj.get("%s_cfg%d" % (args.solver, args.config), {})
            per_pair = t.get(names[dom_classes[0]], {}).get("bytes_per_pair")
            # the figure was collected on the config's own ROI: scale by the ROI area of this run
            cfg_roi = t.get("_roi")
            if per_pair is not None and cfg_roi and list(cfg_roi) != list(roi):
                per_pair = per_pair * P / float(cfg_roi[2] * cfg_roi[3])
            traffic = None if per_pair is None else per_pair * pairs
            traffic_src = {"file": "profiles/pmc_traffic.json", "collected": tj.get("_collected"),
                           "kernel_sources_sha16": tj.get("_kernel_sources_sha16"),
                           "kernel_sources_sha16_now": _kernel_sources_sha16(), "method": tj.get("_method")}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": names[dom_classes[0]], "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
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

    F = W * H
    b_alg_pair = 10.0 * F + (ch + 8 + 120) * P  # I/O + weights + 6 passes
    pcis = [r.get("pci_bus_id") or r.get("uuid") for r in ranks_info]
    line = {
        "metric": "filtered Mpixels/s (+ achieved HBM GB/s) on 4K disparity, 1/2/4/8 GPUs vs CPU ref",
        "value": round(mpx, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %d: %d pairs/GPU of %dx%d, ROI %s (%s), depth-discontinuity radius %d, 8UC%d "
                               "guide, lambda 8000 sigma 1.5, 3 FGS iterations, LRC confidence on"
                               % (args.config, pairs, W, H, list(roi), roi_kind, radius, ch),
                   "pairs_per_gpu": pairs, "total_pairs": n_total, "solver": args.solver, "roi": list(roi), "roi_kind": roi_kind,
                   "radius": radius, "distribution": args.distribution if world > 1 else "resident",
                   "parallelism": "batch-sharded x%d" % world},
        "roofline": roofline, "cpu_baseline": None,
        "whole_call_alg_GBs": round(b_alg_pair * n_total * args.steps / elapsed / 1e9, 1),
        "kernels": kernels, "path": path, "checked": checked_all, "checksum": checksum,
        "world_size": world, "backend": backend if world > 1 else None, "per_rank_ms_per_step": per_rank_ms,
        "ranks": ranks_info,
        "distinct_devices": None if dry else (len(set(pcis)) == world and None not in pcis),
        "launcher": "self" if os.environ.get("ADF_BENCH_WORKER") else ("torchrun" if world > 1 else "single"),
        "scatter_ms": None if scatter_ms is None else round(scatter_ms, 2),
        "gather_ms": None if gather_ms is None else round(gather_ms, 2),
        "pipelined_scatter_filter_gather": pipelined,
        "rccl_legs": None,
        "workspace_GB": workspace_gb,
        "views_to_filtered": None,
        "natural_guide": None,
    }
    if dry:
        line["dry_run"] = True
    return line


def _kernel_sources_sha16():
    """First 16 hex digits of the SHA-256 over the HIP sources: lets a reader see whether profiles/pmc_traffic.json
    (which records the same digest when it is collected) was measured on the kernels that just ran."""
    import hashlib
    d = os.path.join(ROOT, "addingdisparityfiltering_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    under_launcher = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_workers(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
