"""Stress run of tests/test_gpu_fuzz.py's generator over many seeds (not part of the suite): exact solver and confidence
map bit-exact, wave solver within 1 LSB, for every draw; prints the count of draws above the mean bar (expected for badly
conditioned parameters, tests/test_gpu_fuzz.py).    python tools/fuzz_filter_big.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import addingdisparityfiltering_amd as adf  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _case  # noqa: E402


def main(first=5000, count=400):
    bad, above = 0, 0
    for seed in range(first, first + count):
        c = _case(np.random.default_rng(seed))
        p = oracle.default_params(threads=8, use_confidence=int(c["use_conf"]), sigma_color=c["sigma"], disc_radius=c["radius"],
                                  lrc_thresh=c["thresh"], num_iter=c["num_iter"], lambda_attenuation=c["atten"])
        p.lambda_ = c["lam"]
        dr = c["dr"] if c["use_conf"] else None
        exp, exp_conf = oracle.wls_filter(c["dl"], c["view"], dr, c["roi"], p)
        f = adf.createDisparityWLSFilterGeneric(c["use_conf"])
        f.setLambda(c["lam"]); f.setSigmaColor(c["sigma"]); f.setDepthDiscontinuityRadius(c["radius"])
        f.setLRCthresh(c["thresh"]); f.setFGSParams(c["atten"], c["num_iter"])
        f.setSolver(adf.SOLVER_EXACT)
        got = f.filter(c["dl"], c["view"], None, dr, c["roi"])
        ok = np.array_equal(got, exp) and (not c["use_conf"] or np.array_equal(f.getConfidenceMap(), exp_conf))
        f.setSolver(adf.SOLVER_WAVE)
        got2 = f.filter(c["dl"], c["view"], None, dr, c["roi"])
        d = np.abs(got2.astype(np.int64) - exp)
        ok = ok and d.max() <= 1 and (not c["use_conf"] or np.array_equal(f.getConfidenceMap(), exp_conf))
        above += d.mean() > 1 / 256
        if not ok:
            bad += 1
            print("FAIL seed", seed, {k: c[k] for k in ("w", "h", "ch", "roi", "radius", "lam", "sigma", "use_conf")}, "wave max", d.max(), flush=True)
        if (seed - first) % 50 == 49:
            print("seeds %d..%d done, %d failures so far" % (first, seed, bad), flush=True)
    print("%d draws, %d failures, %d above the mean bar (badly conditioned parameters)" % (count, bad, above))
    return bad


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(1 if main(*a) else 0)
