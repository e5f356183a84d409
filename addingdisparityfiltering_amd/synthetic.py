"""Synthetic stereo-pair generator: a clone of the reference perf test's MakeArtificialExample
(modules/ximgproc/perf/perf_disparity_wls_filter.cpp:95-167) with this repo's own PRNG.

A uniform background with a centred foreground rectangle; the left disparity is 0 outside and
16*rect_disparity inside the rectangle; the right disparity is -16*rect_disparity inside the
rectangle shifted left by the disparity; Gaussian noise (sigma 6) on every guide byte and on both
disparity maps (saturating, round-half-even); ROI = (d, 0, w-d, h).

`make_artificial_example` (numpy, host) is what the parity tests feed to both the oracle and the
HIP path; `make_artificial_batch_torch` builds the same kind of scene directly in device memory for
the benchmark (no PCIe, no host PRNG cost).
"""
import numpy as np

# BASELINE.json configs: (W, H, ROI, guide channels, depth-discontinuity radius, rect disparity).
# ROI x = numDisparities of an SGBM matcher with minDisparity 0 (DF.cpp:407); radius = ceil(0.5*3).
CONFIGS = {
    1: dict(W=640, H=480, roi=(64, 0, 576, 480), channels=3, radius=2, rect_disparity=48),
    2: dict(W=1920, H=1080, roi=(160, 0, 1760, 1080), channels=3, radius=2, rect_disparity=120),
    3: dict(W=3840, H=2160, roi=(256, 0, 3584, 2160), channels=3, radius=2, rect_disparity=192),
    4: dict(W=3840, H=2160, roi=(256, 0, 3584, 2160), channels=3, radius=2, rect_disparity=192),
    5: dict(W=1242, H=375, roi=(128, 0, 1114, 375), channels=1, radius=2, rect_disparity=96),
}


def seed_for(config_id, pair_index):
    return 1000 * int(config_id) + int(pair_index)


def _sat(a, lo, hi, dtype):
    return np.clip(np.rint(a), lo, hi).astype(dtype)


def make_artificial_example(w, h, channels=3, seed=0, rect_disparity=None, sigma=6.0):
    """Returns (left_view uint8 (h,w[,3]), disp_left int16, disp_right int16, roi)."""
    rng = np.random.default_rng(seed)
    bg = int(rng.uniform(0.0, 255.0))
    fg = int(rng.uniform(0.0, 255.0))
    rect_w = int(rng.uniform(w // 16, w // 2))
    rect_h = int(rng.uniform(h // 16, h // 2))
    d = int(0.15 * w) if rect_disparity is None else int(rect_disparity)
    x0, y0 = (w - rect_w) // 2, (h - rect_h) // 2
    view = np.full((h, w, channels), float(bg))
    view[y0:y0 + rect_h, x0:x0 + rect_w] = fg
    dl = np.zeros((h, w))
    dr = np.zeros((h, w))
    dl[y0:y0 + rect_h, x0:x0 + rect_w] = 16 * d
    xr = max(x0 - d, 0)
    dr[y0:y0 + rect_h, xr:xr + rect_w] = -16 * d
    view = _sat(view + rng.normal(0.0, sigma, view.shape), 0, 255, np.uint8)
    dl = _sat(dl + rng.normal(0.0, sigma, dl.shape), -32768, 32767, np.int16)
    dr = _sat(dr + rng.normal(0.0, sigma, dr.shape), -32768, 32767, np.int16)
    if channels == 1:
        view = view[:, :, 0]
    return np.ascontiguousarray(view), dl, dr, (d, 0, w - d, h)


def make_config_example(config_id, pair_index=0):
    """One pair of a BASELINE.json config; returns (view, dl, dr, roi, radius)."""
    c = CONFIGS[config_id]
    view, dl, dr, _ = make_artificial_example(c["W"], c["H"], c["channels"], seed_for(config_id, pair_index),
                                              rect_disparity=c["rect_disparity"])
    return view, dl, dr, c["roi"], c["radius"]


def make_artificial_batch_torch(n, w, h, channels, seed, rect_disparity, device, sigma=6.0):
    """Same scene family, generated on `device` with torch's PRNG: (view, dl, dr) with a leading batch dim."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    cpu = np.random.default_rng(int(seed))
    view = torch.empty((n, h, w, channels) if channels > 1 else (n, h, w), dtype=torch.uint8, device=device)
    dl = torch.empty((n, h, w), dtype=torch.int16, device=device)
    dr = torch.empty((n, h, w), dtype=torch.int16, device=device)
    d = int(rect_disparity)
    for k in range(n):
        bg, fg = int(cpu.uniform(0, 255)), int(cpu.uniform(0, 255))
        rect_w, rect_h = int(cpu.uniform(w // 16, w // 2)), int(cpu.uniform(h // 16, h // 2))
        x0, y0 = (w - rect_w) // 2, (h - rect_h) // 2
        v = torch.full((h, w, channels), float(bg), device=device)
        v[y0:y0 + rect_h, x0:x0 + rect_w] = float(fg)
        v += sigma * torch.randn(v.shape, generator=g, device=device)
        v = v.round_().clamp_(0, 255).to(torch.uint8)
        view[k] = v if channels > 1 else v[:, :, 0]
        a = torch.zeros((h, w), device=device)
        a[y0:y0 + rect_h, x0:x0 + rect_w] = 16.0 * d
        a += sigma * torch.randn(a.shape, generator=g, device=device)
        dl[k] = a.round_().clamp_(-32768, 32767).to(torch.int16)
        b = torch.zeros((h, w), device=device)
        xr = max(x0 - d, 0)
        b[y0:y0 + rect_h, xr:xr + rect_w] = -16.0 * d
        b += sigma * torch.randn(b.shape, generator=g, device=device)
        dr[k] = b.round_().clamp_(-32768, 32767).to(torch.int16)
    return view, dl, dr
