"""Soak run (not part of the suite): the filter called a few thousand times while the geometry, the batch size and the call
form keep changing (full-resolution, down-scaled with and without getConfidenceMap, no-confidence mode, generic FGS, both
solvers), checking that (a) every call of a given case returns the SAME bits as its first call (the kernels have no
run-to-run nondeterminism: no atomics, no order-dependent reductions), (b) device memory in use stops growing once every
case has run once (workspaces are reused or returned, nothing leaks).    python tools/soak.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")


def digest(t):
    return int(t.to(torch.int64).sum().item()) ^ (int((t.to(torch.int64) * torch.arange(1, t.numel() + 1, device=t.device).reshape(t.shape) % 1000003).sum().item()) << 1)


def batch(cfg_id, n, seed):
    c = synthetic.CONFIGS[cfg_id]
    v, dl, dr = synthetic.make_artificial_batch_torch(n, c["W"], c["H"], c["channels"], seed, c["rect_disparity"], dev)
    return c, v, dl, dr


cases = []


def add_case(name, fn):
    cases.append([name, fn, None, 0])


# full-resolution calls: configs 5 / 2 / 3 geometry, several batch sizes, both solvers, two handles reused throughout
wls = adf.createDisparityWLSFilterGeneric(True); wls.setSigmaColor(1.5)
wls_exact = adf.createDisparityWLSFilterGeneric(True); wls_exact.setSigmaColor(1.5); wls_exact.setSolver(adf.SOLVER_EXACT)
noconf = adf.createDisparityWLSFilterGeneric(False); noconf.setSigmaColor(1.5)
for cfg_id, n in ((5, 1), (5, 7), (2, 1), (2, 3), (3, 2), (3, 9), (5, 32)):
    c, v, dl, dr = batch(cfg_id, n, 100 + cfg_id * 10 + n)
    def full(c=c, v=v, dl=dl, dr=dr, f=wls):
        f.setDepthDiscontinuityRadius(c["radius"])
        out = f.filter(dl, v, None, dr, c["roi"])
        conf = f.getConfidenceMap(0)
        return digest(out) ^ digest((conf * 1).to(torch.int32))
    add_case("cfg%d x%d wave" % (cfg_id, n), full)
    if n <= 3:
        add_case("cfg%d x%d exact" % (cfg_id, n), lambda c=c, v=v, dl=dl, dr=dr: full(c, v, dl, dr, wls_exact))
        def nc(c=c, v=v, dl=dl):
            noconf.setDepthDiscontinuityRadius(c["radius"])
            return digest(noconf.filter(dl, v, None, None, c["roi"]))
        add_case("cfg%d x%d no confidence" % (cfg_id, n), nc)

# down-scaled calls (half-size maps), with the confidence map asked for every other time
for cfg_id, n in ((2, 2), (3, 3)):
    c, v, dl, dr = batch(cfg_id, n, 500 + cfg_id)
    dlo = dl[:, ::2, ::2].contiguous(); dro = dr[:, ::2, ::2].contiguous()
    roi_lo = (c["roi"][0] // 2, c["roi"][1] // 2, c["roi"][2] // 2, c["roi"][3] // 2)
    state = {"k": 0}
    def scaled(c=c, v=v, dlo=dlo, dro=dro, roi_lo=roi_lo, state=state):
        wls.setDepthDiscontinuityRadius(2)
        out = wls.filter(dlo, v, None, dro, roi_lo)
        state["k"] += 1
        d = digest(out)
        if state["k"] % 2 == 0:
            wls.getConfidenceMap(0)          # materialises the view-sized map now and then; the output does not depend on it
        return d
    add_case("cfg%d x%d down-scaled" % (cfg_id, n), scaled)

# generic FGS: a new filter object per call (guide given at creation, as the reference's one-shot function does)
g = torch.randint(0, 256, (720, 1280, 3), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
src = torch.randint(0, 256, (720, 1280, 3), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(8))
def fgs():
    f = adf.createFastGlobalSmootherFilter(g, 1000.0, 10.0)
    return digest(f.filter(src))
add_case("generic FGS 1280x720x3 one-shot", fgs)

torch.cuda.synchronize()
free0 = None
t0 = time.time()
calls = 0
used_after = []
for r in range(rounds):
    order = np.random.default_rng(r).permutation(len(cases))
    for i in order:
        name, fn, first, cnt = cases[i]
        d = fn()
        calls += 1
        if first is None:
            cases[i][2] = d
        elif d != first:
            print("NONDETERMINISTIC: %s call %d differs from its first call" % (name, cnt + 1))
            sys.exit(1)
        cases[i][3] = cnt + 1
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(dev)
    used_after.append((total - free) / 2**30)
    if r % 5 == 0 or r == rounds - 1:
        print("round %3d: %5d calls, %.1f s, device memory in use %.2f GiB" % (r, calls, time.time() - t0, used_after[-1]), flush=True)
grow = used_after[-1] - used_after[min(2, len(used_after) - 1)]
print("%d cases x %d rounds = %d calls, every call bit-identical to its case's first; memory in use after round 2: %.2f GiB, after the last: %.2f GiB (growth %.3f GiB)"
      % (len(cases), rounds, calls, used_after[min(2, len(used_after) - 1)], used_after[-1], grow))
sys.exit(0 if grow < 0.25 else 2)
