import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import addingdisparityfiltering_amd as adf, oracle
from test_gpu_conf_band import _random_aligned_case, _check
bad = 0
for seed in range(400):
    rng = np.random.default_rng(90000 + seed)
    W, H, roi, radius, kind, thresh = _random_aligned_case(rng)
    try:
        _check(adf, oracle, W, H, roi, radius, kind, 90000 + seed, thresh)
    except AssertionError as e:
        bad += 1; print("FAIL", seed, W, H, roi, radius, kind, thresh, str(e)[:200], flush=True)
    if seed % 50 == 0: print("progress", seed, flush=True)
print("done, failures:", bad)
