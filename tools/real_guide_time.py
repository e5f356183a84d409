"""Measurement helper (not part of the product): per-kernel times of the 4K filter call on a NATURAL-IMAGE guide next to
the synthetic scene of the benchmark.  The guide is the reference's KITTI fixture (tests/golden/kitti_left.bmp, gray)
tiled to 3840x2160, as one channel and as three (the gray value plus a little per-channel noise), so that the
edge-weight table sees what real images give it: a few per cent of indices beyond the head cached in LDS.

  python tools/real_guide_time.py [pairs]        (ADF_NO_OVERLAP=1 for sequential kernel times)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import addingdisparityfiltering_amd as adf  # noqa: E402
from addingdisparityfiltering_amd import synthetic  # noqa: E402


def kitti_tiled(h, w):
    from PIL import Image
    im = np.array(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kitti_left.bmp")))
    if im.ndim == 3:
        im = im[:, :, 0]
    reps = (h + im.shape[0] - 1) // im.shape[0], (w + im.shape[1] - 1) // im.shape[1]
    return np.ascontiguousarray(np.tile(im, reps)[:h, :w])


def run(name, view, dl, dr, roi, radius, steps=10):
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius); f.setSolver(adf.SOLVER_WAVE)
    for _ in range(2):
        f.filter(dl, view, None, dr, roi)
    torch.cuda.synchronize()
    f.enableProfiling(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f.filter(dl, view, None, dr, roi)
    e1.record(); torch.cuda.synchronize()
    prof = f.readProfile()
    ms = e0.elapsed_time(e1) / steps
    px = view.shape[0] * view.shape[1] * view.shape[2]
    print("%-28s %7.3f ms/step  %7.1f Mpx/s  " % (name, ms, px / ms / 1e3) +
          "  ".join("%s %.3f" % (k, v["total_ms"] / steps) for k, v in prof.items()))


def main(pairs=16):
    dev = torch.device("cuda:0")
    cfg = synthetic.CONFIGS[3]
    W, H, roi, radius = cfg["W"], cfg["H"], cfg["roi"], cfg["radius"]
    sview, dl, dr = synthetic.make_artificial_batch_torch(pairs, W, H, 3, synthetic.seed_for(3, 0), cfg["rect_disparity"], dev)
    run("synthetic, 3 channels", sview, dl, dr, roi, radius)
    run("synthetic, 1 channel", sview[..., 0].contiguous(), dl, dr, roi, radius)
    g = torch.from_numpy(kitti_tiled(H, W)).to(dev)
    g1 = g[None].expand(pairs, H, W).contiguous()
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    g3 = (g1[..., None].float() + 2.0 * torch.randn((pairs, H, W, 3), generator=gen, device=dev)).round_().clamp_(0, 255).to(torch.uint8)
    dh = ((g3[0, :, 1:].int() - g3[0, :, :-1].int()) ** 2).sum(-1)
    print("natural guide: %.2f %% of the horizontal table indices are >= 2048 (3 channels), %.2f %% (1 channel)" % (
        100.0 * (dh >= 2048).float().mean().item(),
        100.0 * (((g1[0, :, 1:].int() - g1[0, :, :-1].int()) ** 2) >= 2048).float().mean().item()))
    run("KITTI tiled, 3 channels", g3, dl, dr, roi, radius)
    run("KITTI tiled, 1 channel", g1, dl, dr, roi, radius)
    # the worst case for the table: white noise -- nearly every index lies beyond the head cached in LDS
    noise = torch.randint(0, 256, (pairs, H, W, 3), generator=gen, device=dev, dtype=torch.uint8)
    dn = ((noise[0, :, 1:].int() - noise[0, :, :-1].int()) ** 2).sum(-1)
    print("white-noise guide: %.1f %% of the horizontal table indices are >= 2048" % (100.0 * (dn >= 2048).float().mean().item()))
    run("white noise, 3 channels", noise, dl, dr, roi, radius)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
