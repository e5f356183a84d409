"""Measurement helper: one call on N pairs of 7680x4320 (ROI (512,0,7168,4320), radius 2), wave solver (two wavefronts per
row, half strips of 128 chunks per column: round 3) against the exact solver such ROIs used to fall back to."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import addingdisparityfiltering_amd as adf  # noqa: E402
from addingdisparityfiltering_amd import synthetic  # noqa: E402


def main(n=4):
    dev = torch.device("cuda:0")
    W, H, roi = 7680, 4320, (512, 0, 7168, 4320)
    view, dl, dr = synthetic.make_artificial_batch_torch(n, W, H, 3, 5, 512, dev)
    for name, solver in (("wave", adf.SOLVER_WAVE), ("exact", adf.SOLVER_EXACT)):
        f = adf.createDisparityWLSFilterGeneric(True)
        f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(2); f.setSolver(solver)
        out = None
        for _ in range(2):
            out = f.filter(dl, view, out, dr, roi)
        torch.cuda.synchronize()
        f.enableProfiling(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        steps = 5
        for _ in range(steps):
            f.filter(dl, view, out, dr, roi)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        prof = f.readProfile()
        print("%-6s %d x 8K: %8.3f ms per call  %8.1f Mpx/s  (ran: %s)  " % (name, n, ms, n * W * H / ms / 1e3, "wave" if f.getLastSolver() == adf.SOLVER_WAVE else "exact") +
              "  ".join("%s %.2f" % (k, v["total_ms"] / steps) for k, v in prof.items()))
        del f, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
