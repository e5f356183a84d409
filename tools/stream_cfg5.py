"""BASELINE config 5 measured as the STREAM it is ("per-frame latency and sustained Mpx/s"): frames of
1242x375 (8UC1 guide, ROI (128,0,1114,375), LRC confidence on, 3 FGS iterations), ONE frame per `filter` call, the calls
dealt round-robin to K handles, each on its own stream, K in {1, 2, 4, 8}.  Per K: sustained Mpixels/s over the whole
run (wall clock, first call issued -> last frame done) and the per-frame latency on the device, event to event (an
event pair brackets each call on its stream: what the frame's own kernels took, including whatever the other streams'
frames cost it).  Issue modes: "calls" = the C-ABI call per frame from ONE host thread (seven to nine launches issued by the
host per frame), "threads" = the same with one host thread per handle, and
"graphs" = the same call captured once per handle into a HIP graph and replayed (one launch per frame; the frame is
copied into the handle's fixed input buffers first, as a camera driver would; "graphs_inplace": the producer writes those
buffers itself, so a frame is one replay + the two timing events; "replay_only": the same without the events, ONE
runtime call per frame, throughput only).  One frame per K is compared with the oracle when `check` is set.

    python tools/stream_cfg5.py [frames]          prints the table (profiles/r04_stream_cfg5.txt)
bench.py's next_rows leg calls `measure()` (never part of `value`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _percentile(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(q * (len(v) - 1))))]


def measure(adf, torch, dev, synthetic, frames=256, ks=(1, 2, 4, 8), check=True, graphs=True, reps=3):
    cfg = synthetic.CONFIGS[5]
    W, H, roi, ch, radius = cfg["W"], cfg["H"], cfg["roi"], cfg["channels"], cfg["radius"]
    view, dl, dr = synthetic.make_artificial_batch_torch(frames, W, H, ch, synthetic.seed_for(5, 0), cfg["rect_disparity"], dev)
    out = torch.empty((frames, H, W), dtype=torch.int16, device=dev)
    res = {"frames": frames, "frame": "%dx%d, %d channel(s), ROI %s, radius %d, 3 iterations" % (W, H, ch, list(roi), radius), "by_streams": {}}

    def make():
        f = adf.createDisparityWLSFilterGeneric(True)
        f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius); f.setFGSParams(0.25, 3)
        return f

    def run(K, mode):
        handles = [make() for _ in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
        gr, bufs = [], []
        for k in range(K):                                   # warm-up: workspaces exist, kernels loaded
            with torch.cuda.stream(streams[k]):
                for _ in range(2):
                    handles[k].filter(dl[k], view[k], out[k], dr[k], roi)
        torch.cuda.synchronize()
        if mode in ("graphs", "graphs_inplace", "replay_only"):
            for k in range(K):
                b = (dl[k].clone(), view[k].clone(), dr[k].clone(), torch.empty((H, W), dtype=torch.int16, device=dev))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[k]):
                    handles[k].filter(b[0], b[1], b[3], b[2], roi)
                gr.append(g); bufs.append(b)
            torch.cuda.synchronize()
        def issue(i, k):
            s = streams[k]
            with torch.cuda.stream(s):
                if mode == "replay_only":                    # ... and nobody times the frames: ONE runtime call per frame
                    gr[k].replay()
                elif mode == "graphs_inplace":               # the producer writes the handle's input buffers itself: one replay per frame
                    ev[i][0].record(s)
                    gr[k].replay()
                    ev[i][1].record(s)
                elif mode == "graphs":
                    b = bufs[k]
                    b[0].copy_(dl[i], non_blocking=True); b[1].copy_(view[i], non_blocking=True); b[2].copy_(dr[i], non_blocking=True)
                    ev[i][0].record(s)
                    gr[k].replay()
                    ev[i][1].record(s)
                    out[i].copy_(b[3], non_blocking=True)
                else:
                    ev[i][0].record(s)
                    handles[k].filter(dl[i], view[i], out[i], dr[i], roi)
                    ev[i][1].record(s)

        t0 = time.perf_counter()
        if mode == "threads":                                # one host thread per handle (the C-ABI calls release the GIL)
            import threading

            def worker(k):
                torch.cuda.set_device(dev)
                for i in range(k, frames, K):
                    issue(i, k)
            th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            for i in range(frames):
                issue(i, i % K)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        lat = [a.elapsed_time(b) * 1e3 for a, b in ev] if mode != "replay_only" else [0.0]       # microseconds
        r = {"sustained_Mpixels_per_s": round(frames * W * H / wall / 1e6, 1), "frames_per_s": round(frames / wall, 1),
             "latency_us_median": round(_percentile(lat, 0.5), 1), "latency_us_p99": round(_percentile(lat, 0.99), 1),
             "host_issue_us_per_frame": round(t_issue / frames * 1e6, 1)}
        if check and mode not in ("graphs_inplace", "replay_only"):               # (in place every replay filters the handle's own frame k: nothing lands in `out`)
            import oracle
            i = frames - 1 - (K // 2)
            p = oracle.default_params(threads=min(16, os.cpu_count() or 1), sigma_color=1.5, disc_radius=radius)
            exp, _ = oracle.wls_filter(dl[i].cpu().numpy(), view[i].cpu().numpy(), dr[i].cpu().numpy(), roi, p)
            d = np.abs(out[i].cpu().numpy().astype(np.int32) - exp.astype(np.int32))
            r["checked"] = {"frame": i, "disparity_max_abs_lsb": int(d.max()), "disparity_mean_abs_lsb": round(float(d.mean()), 6)}
        del handles, gr, bufs
        return r

    def run3(K, mode):
        # the host's issue rate bounds most of these figures and jitters from run to run: three runs, the median one
        # reported, the best rate beside it
        rs = sorted((run(K, mode) for _ in range(reps)), key=lambda r: r["sustained_Mpixels_per_s"])
        r = rs[len(rs) // 2]
        r["best_of_%d_Mpixels_per_s" % reps] = rs[-1]["sustained_Mpixels_per_s"]
        return r

    for K in ks:
        entry = {"calls": run3(K, "calls")}
        if K > 1:
            out.zero_()
            entry["threads"] = run3(K, "threads")
        if graphs:
            try:
                out.zero_()
                entry["graphs"] = run3(K, "graphs")
                entry["graphs_inplace"] = run3(K, "graphs_inplace")
                entry["replay_only"] = run3(K, "replay_only")
            except Exception as e:                           # (a capture the runtime refuses must not cost the table)
                entry["graphs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        res["by_streams"][str(K)] = entry
    # Micro-batches: B consecutive frames per call (n_pairs = B) on one stream -- what a pipeline does when the per-call
    # launches, not the device, bound the frame rate.  Latency = event to event around the call (all B frames finish
    # together), so a frame waits for up to B - 1 later arrivals on top of it.
    res["micro_batches"] = {}
    f = make()
    for B in (1, 2, 4, 8, 16, 32):
        nb = frames // B
        for _ in range(2):
            f.filter(dl[:B], view[:B], out[:B], dr[:B], roi)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]
        t0 = time.perf_counter()
        for j in range(nb):
            a, b = j * B, (j + 1) * B
            ev[j][0].record()
            f.filter(dl[a:b], view[a:b], out[a:b], dr[a:b], roi)
            ev[j][1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        lat = [x.elapsed_time(y) * 1e3 for x, y in ev]
        res["micro_batches"][str(B)] = {"sustained_Mpixels_per_s": round(nb * B * W * H / wall / 1e6, 1), "frames_per_s": round(nb * B / wall, 1),
                                        "call_latency_us_median": round(_percentile(lat, 0.5), 1), "call_latency_us_p99": round(_percentile(lat, 0.99), 1)}
    res["note"] = ("one frame per filter call, K handles on K streams, frames dealt round-robin; latency = event to event around "
                   "each call on its stream; sustained = all frames / wall clock; never part of `value`")
    return res


def main():
    import torch

    import addingdisparityfiltering_amd as adf
    from addingdisparityfiltering_amd import synthetic

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    r = measure(adf, torch, dev, synthetic, frames)
    print("# config 5 as a stream: %d frames of %s" % (r["frames"], r["frame"]))
    print("# K streams | mode           | sustained Mpx/s: median of 3 runs (best) | frames/s | latency median / p99 (us) | host issue (us/frame) | checked (max / mean LSB)")
    for K, e in r["by_streams"].items():
        for mode, v in e.items():
            if "error" in v:
                print("  %2s        | %-14s | %s" % (K, mode, v["error"]))
                continue
            c = v.get("checked", {})
            best = [v[k] for k in v if k.startswith("best_of_")]
            print("  %2s        | %-14s | %9.1f (%7.1f) | %8.1f | %10.1f / %-10.1f | %8.1f              | %s / %s" % (
                K, mode, v["sustained_Mpixels_per_s"], best[0] if best else v["sustained_Mpixels_per_s"], v["frames_per_s"],
                v["latency_us_median"], v["latency_us_p99"], v["host_issue_us_per_frame"], c.get("disparity_max_abs_lsb"),
                c.get("disparity_mean_abs_lsb")))
    print("# micro-batches (B frames per call, one stream): B | sustained Mpx/s | frames/s | call latency median / p99 (us)")
    for B, v in r["micro_batches"].items():
        print("  %3s | %10.1f | %9.1f | %8.1f / %.1f" % (B, v["sustained_Mpixels_per_s"], v["frames_per_s"], v["call_latency_us_median"], v["call_latency_us_p99"]))


if __name__ == "__main__":
    main()
