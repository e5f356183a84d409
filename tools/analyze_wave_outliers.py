"""Analysis of fuzz draws where the wave solver is more than 1 LSB from the scalar order (tools/fuzz_filter_big.py):
for each seed, the wave solver, the oracle's scalar order, the oracle's restatement of the REFERENCE's OWN SIMD order
(FGS.cpp:305-314, 526-534) and a float64 banded solve of the same systems, pairwise.
    python tools/analyze_wave_outliers.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import addingdisparityfiltering_amd as adf  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _case  # noqa: E402


def main(seeds):
    for seed in seeds:
        c = _case(np.random.default_rng(seed))
        dr = c["dr"] if c["use_conf"] else None

        def params(order):
            p = oracle.default_params(threads=4, use_confidence=int(c["use_conf"]), sigma_color=c["sigma"], disc_radius=c["radius"],
                                      lrc_thresh=c["thresh"], num_iter=c["num_iter"], lambda_attenuation=c["atten"], order=order)
            p.lambda_ = c["lam"]
            return p
        scal, conf = oracle.wls_filter(c["dl"], c["view"], dr, c["roi"], params(oracle.ORDER_SCALAR))
        simd, _ = oracle.wls_filter(c["dl"], c["view"], dr, c["roi"], params(oracle.ORDER_REF_SIMD))
        f = adf.createDisparityWLSFilterGeneric(c["use_conf"])
        f.setLambda(c["lam"]); f.setSigmaColor(c["sigma"]); f.setDepthDiscontinuityRadius(c["radius"])
        f.setLRCthresh(c["thresh"]); f.setFGSParams(c["atten"], c["num_iter"]); f.setSolver(adf.SOLVER_WAVE)
        wave = f.filter(c["dl"], c["view"], None, dr, c["roi"])
        x, y, w, h = c["roi"]
        sl = (slice(y, y + h), slice(x, x + w))
        # float64 solve of the same separable systems (oracle/banded_f64.py), rounded like the epilogue
        import banded_f64
        guide = np.ascontiguousarray(c["view"][sl])
        if c["use_conf"]:
            planes = np.stack([conf[sl].astype(np.float64) * c["dl"][sl].astype(np.float64), conf[sl].astype(np.float64)])
        else:
            planes = c["dl"][sl].astype(np.float64)[None]
        sol = [banded_f64.fgs_f64(guide, pl, c["lam"], c["sigma"], c["atten"], c["num_iter"]) for pl in planes]
        ref64 = sol[0] / (sol[1] + 1e-43) if c["use_conf"] else sol[0]

        def dist(a, b):
            d = np.abs(a[sl].astype(np.int64) - b[sl].astype(np.int64))
            return "max %d, pixels > 1 LSB: %d, differing %.4f %%" % (d.max(), int((d > 1).sum()), (d > 0).mean() * 100)

        def dist64(a):
            d = np.abs(a[sl].astype(np.float64) - ref64)
            return "max %.3f, mean %.4f LSB" % (d.max(), d.mean())
        print("seed %d: %dx%d ROI %s ch %d radius %d lambda %.0f sigma %.2f atten %.2f iters %d confidence %s" % (
            seed, c["w"], c["h"], c["roi"], c["ch"], c["radius"], c["lam"], c["sigma"], c["atten"], c["num_iter"], c["use_conf"]))
        print("   wave vs scalar order:            ", dist(wave, scal))
        print("   reference-SIMD vs scalar order:  ", dist(simd, scal))
        print("   wave vs reference-SIMD order:    ", dist(wave, simd))
        print("   vs float64:  scalar", dist64(scal), "| reference-SIMD", dist64(simd), "| wave", dist64(wave))


if __name__ == "__main__":
    main([int(v) for v in sys.argv[1:]] or [7299, 7525])
