"""Measurement helper (not part of the product): phase timeline of the wave column pass, per workgroup and CU.

  python -c "from addingdisparityfiltering_amd import build; build.build_variant('vphase', ['ADF_V_PHASE_TIMING'])"
  ADF_WLS_LIB=$PWD/addingdisparityfiltering_amd/libadf_wls_vphase.so python tools/vphase.py [pairs]

The instrumented kernel stamps the 100 MHz clock at the phase boundaries of every workgroup of the plain
(EPI_PLANES) column pass and records the CU it ran on; the stamps of the last such launch are read back.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import addingdisparityfiltering_amd as adf  # noqa: E402
from addingdisparityfiltering_amd import _lib, synthetic  # noqa: E402


def main(pairs=64, width=None):
    dev = torch.device("cuda:0")
    cfg = synthetic.CONFIGS[3]
    roi, radius = cfg["roi"], cfg["radius"]
    if width:   # narrower ROI = fewer strips = fewer CUs loading at once
        roi = (roi[0], roi[1], int(width), roi[3])
    view, dl, dr = synthetic.make_artificial_batch_torch(pairs, cfg["W"], cfg["H"], cfg["channels"], synthetic.seed_for(3, 0),
                                                         cfg["rect_disparity"], dev)
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    f.setSolver(adf.SOLVER_WAVE)
    for _ in range(3):
        f.filter(dl, view, None, dr, roi)
    torch.cuda.synchronize()
    lib = _lib.lib()
    n_wg = pairs * ((roi[2] + 63) // 64 * 64 // 16)
    buf = (ctypes.c_ulonglong * (n_wg * 8))()
    lib.adf_debug_read_vphase.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.adf_debug_read_vphase(buf, n_wg * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
    t = a[:, :7] * 10e-3  # 100 MHz ticks -> microseconds
    names = ["load", "boundary", "exchange+pcr", "solve", "store issue", "store drain"]
    d = np.diff(t, axis=1)
    print("workgroups", n_wg, "kernel span %.1f us" % (t[:, 6].max() - t[:, 0].min()))
    for k, nm in enumerate(names):
        print("  %-14s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (nm, d[:, k].mean(), *np.percentile(d[:, k], [10, 50, 90])))
    print("  (stamps are wave 0's: its 'exchange+pcr' includes waiting at the barrier for the other waves' loads)")
    hw = a[:, 7] & 0xFFFFFFFF
    xcc = (a[:, 7] >> 32) & 0xF
    key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xF)
    uniq = np.unique(key)
    wb = (ctypes.c_ulonglong * (n_wg * 16))()
    lib.adf_debug_read_vwave.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.adf_debug_read_vwave(wb, n_wg * 16) == 0
    w = np.frombuffer(wb, dtype=np.uint64).reshape(n_wg, 8, 2).astype(np.int64) * 10e-3
    wg_start, wg_end = w[:, :, 0].min(axis=1), w[:, :, 1].max(axis=1)
    print("per workgroup: first wave start -> last wave end %.2f us (last wave ends %.2f us after the first)" % (
        (wg_end - wg_start).mean(), (wg_end - w[:, :, 1].min(axis=1)).mean()))
    gaps = []
    for u in uniq:
        m = key == u
        o = np.argsort(wg_start[m])
        s_, e_ = wg_start[m][o], wg_end[m][o]
        if len(s_) > 1:
            gaps.append((s_[1:] - e_[:-1]).mean())
    print("distinct CUs %d; hand-over (last wave of a workgroup ends -> next workgroup starts on that CU): %.2f us" % (len(uniq), np.mean(gaps)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else None)
