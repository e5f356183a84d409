"""Thread-safety soak (not part of the suite): K host threads, each with its OWN handles and stream (the contract of
INTEGRATION.md: a handle is single-caller, concurrency = one handle per thread), hammer the library at the same time --
filter calls of different geometries and sigmas, down-scaled calls, one-shot generic FGS objects created and destroyed
(the shared block cache), parameter changes -- and every result must equal the bits the same case gave single-threaded.
python tools/soak_threads.py [threads] [rounds]"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0")


def digest(t):
    t = t.to(torch.int64).flatten()
    return int(t.sum().item()) ^ (int((t * (torch.arange(t.numel(), device=t.device) % 8191 + 1)).sum().item()) << 1)


def make_cases(k):
    """cases of thread k: closures over their own inputs and handles"""
    sigma = (0.8, 1.5, 2.5, 4.0, 1.0, 3.0)[k % 6]
    wls = adf.createDisparityWLSFilterGeneric(True); wls.setSigmaColor(sigma)
    ex = adf.createDisparityWLSFilterGeneric(True); ex.setSigmaColor(sigma); ex.setSolver(adf.SOLVER_EXACT)
    out = []
    for cfg_id, n in ((5, 1 + k), (2, 1), (3, 1 + k % 2)):
        c = synthetic.CONFIGS[cfg_id]
        v, dl, dr = synthetic.make_artificial_batch_torch(n, c["W"], c["H"], c["channels"], 900 + 10 * k + cfg_id, c["rect_disparity"], dev)
        def full(f=wls, c=c, v=v, dl=dl, dr=dr):
            f.setDepthDiscontinuityRadius(c["radius"])
            o = f.filter(dl, v, None, dr, c["roi"])
            return digest(o) ^ digest(f.getConfidenceMap(0).to(torch.int32))
        out.append(("cfg%d x%d" % (cfg_id, n), full))
        if cfg_id != 3:
            out.append(("cfg%d x%d exact" % (cfg_id, n), lambda c=c, v=v, dl=dl, dr=dr: full(ex, c, v, dl, dr)))
        if cfg_id == 2:
            dlo, dro = dl[:, ::2, ::2].contiguous(), dr[:, ::2, ::2].contiguous()
            rlo = tuple(q // 2 for q in c["roi"])
            def scaled(v=v, dlo=dlo, dro=dro, rlo=rlo):
                wls.setDepthDiscontinuityRadius(2)
                return digest(wls.filter(dlo, v, None, dro, rlo))
            out.append(("cfg2 down-scaled", scaled))
    gen = torch.Generator(device=dev).manual_seed(40 + k)
    g = torch.randint(0, 256, (480 + 16 * k, 640, 3), dtype=torch.uint8, device=dev, generator=gen)
    s = torch.randint(-2000, 2000, (480 + 16 * k, 640), dtype=torch.int16, device=dev, generator=gen)
    def fgs():
        f = adf.createFastGlobalSmootherFilter(g, 500.0 + 100 * k, 5.0 + k)
        return digest(f.filter(s))
    out.append(("one-shot FGS", fgs))
    return out


all_cases = [make_cases(k) for k in range(K)]
# single-threaded reference digests, each case on its thread's stream
streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
ref = []
for k in range(K):
    with torch.cuda.stream(streams[k]):
        ref.append([fn() for _, fn in all_cases[k]])
torch.cuda.synchronize()

errors = []
def worker(k):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[k]):
        for r in range(rounds):
            for i, (name, fn) in enumerate(all_cases[k]):
                d = fn()
                if d != ref[k][i]:
                    errors.append("thread %d round %d: %s differs from its single-threaded result" % (k, r, name))
                    return

t0 = time.time()
th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
n = sum(len(c) for c in all_cases) * rounds
if errors:
    print("\n".join(errors)); sys.exit(1)
print("%d threads x %d rounds: %d calls in %.1f s, every result bit-identical to the single-threaded run of its case" % (K, rounds, n, time.time() - t0))
