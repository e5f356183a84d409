"""Filters made and destroyed per call (EF.hpp:413 fastGlobalSmootherFilter; perf_fgs_filter.cpp:70-76): the library hands a
destroyed filter's device block to the next one behind an event and shares weight tables by sigma (adf_api.hip:
BlockCache, LutStore).  Results must not depend on any of that."""
import numpy as np
import pytest


def _case(rng, w, h, gch, depth, cn):
    guide = rng.integers(0, 256, (h, w, gch) if gch > 1 else (h, w), dtype=np.uint8)
    shape = (h, w, cn) if cn > 1 else (h, w)
    if depth == np.float32:
        src = (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
    else:
        src = rng.integers(0, 256, shape).astype(depth)
    return guide, src


@pytest.mark.gpu
def test_one_shot_calls_equal_the_oracle_through_block_reuse(adf, oracle):
    """Same size again and again (every create after the first takes the previous filter's block), then other sizes,
    fresh lambda / sigma every call as the reference's perf test draws them: exact solver bit for bit."""
    rng = np.random.default_rng(11)
    sizes = [(160, 96)] * 4 + [(131, 77), (160, 96), (320, 48), (131, 77)]
    for k, (w, h) in enumerate(sizes):
        gch, cn = (1, 3)[k % 2], (1, 3)[(k // 2) % 2]
        depth = (np.uint8, np.int16, np.float32)[k % 3]
        guide, src = _case(rng, w, h, gch, depth, cn)
        lam, sig = float(rng.uniform(500.0, 10000.0)), float(rng.uniform(1.0, 100.0))
        got = adf.fastGlobalSmootherFilter(guide, src, lam, sig, solver=adf.SOLVER_EXACT)
        exp = oracle.fgs_filter(guide, src, lam, sig)
        assert np.array_equal(got, exp), (k, w, h, gch, cn, depth)


@pytest.mark.gpu
def test_block_handed_over_between_streams(adf):
    """Filter A works on stream 1 and is destroyed while its kernels are queued; filter B is created on stream 2 and gets
    A's block: B's clearing of the block must wait for A's last copy-out (the event that travels with the block)."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    w, h = 1280, 720
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    cases = []
    for k in range(6):
        guide, src = _case(rng, w, h, 3, np.float32, 3)
        cases.append((torch.from_numpy(guide).to(dev), torch.from_numpy(src).to(dev), 3000.0 + 500 * k, 5.0 + k))
    # reference results: one filter at a time, synchronised
    expect = []
    for g, s, lam, sig in cases:
        f = adf.createFastGlobalSmootherFilter(g, lam, sig)
        expect.append(f.filter(s).clone())
        torch.cuda.synchronize()
        del f
    adf.releaseCachedMemory()
    torch.cuda.synchronize()
    outs = []
    for k, (g, s, lam, sig) in enumerate(cases):
        with torch.cuda.stream(s1 if k % 2 == 0 else s2):
            f = adf.createFastGlobalSmootherFilter(g, lam, sig)
            outs.append(f.filter(s))
            del f                                   # destroyed with its work still queued
    torch.cuda.synchronize()
    for k in range(len(cases)):
        assert torch.equal(outs[k], expect[k]), k


@pytest.mark.gpu
def test_more_sigmas_than_shared_tables_and_release(adf, oracle):
    rng = np.random.default_rng(3)
    guide, src = _case(rng, 96, 64, 3, np.uint8, 1)
    sigmas = [1.0 + 0.75 * k for k in range(20)]          # more than the store keeps
    first = {}
    for sig in sigmas + sigmas[:4]:
        got = adf.fastGlobalSmootherFilter(guide, src, 4000.0, sig, solver=adf.SOLVER_EXACT)
        if sig in first:
            assert np.array_equal(got, first[sig])
        else:
            first[sig] = got
            if len(first) in (1, 10, 20):
                assert np.array_equal(got, oracle.fgs_filter(guide, src, 4000.0, sig))
    adf.releaseCachedMemory()
    got = adf.fastGlobalSmootherFilter(guide, src, 4000.0, sigmas[0], solver=adf.SOLVER_EXACT)
    assert np.array_equal(got, first[sigmas[0]])


@pytest.mark.gpu
@pytest.mark.parametrize("depth,cn", [(np.uint8, 1), (np.uint8, 3), (np.int16, 3), (np.float32, 4), (np.int16, 1)])
def test_device_filter_where_the_images_are(adf, depth, cn):
    """adf_fgs_filter_device reads `src` and writes `dst` directly (no staging copy): rows at any stride and alignment,
    dst == src in place, overlapping buffers refused."""
    import ctypes as C

    import torch
    from addingdisparityfiltering_amd import _lib

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    w, h, pad_l, pad_r = 203, 57, 3, 6
    guide, src = _case(rng, w, h, 3, depth, cn)
    g = torch.from_numpy(guide).to(dev)
    s = torch.from_numpy(src.reshape(h, w, cn)).to(dev)
    f = adf.createFastGlobalSmootherFilter(g, 2500.0, 12.0)
    want = f.filter(s.reshape(src.shape).contiguous()).reshape(h, w, cn).clone()      # dense tensors (the wrapper's path)
    torch.cuda.synchronize()
    big_s = torch.zeros((h, pad_l + w + pad_r, cn), dtype=s.dtype, device=dev)
    big_d = torch.full((h, pad_r + w + pad_l, cn), 7, dtype=s.dtype, device=dev)
    big_s[:, pad_l:pad_l + w] = s
    vs, vd = big_s[:, pad_l:pad_l + w], big_d[:, pad_r:pad_r + w]
    dcode = {np.uint8: _lib.DEPTH_8U, np.int16: _lib.DEPTH_16S, np.float32: _lib.DEPTH_32F}[depth]
    esz = s.element_size()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    call = lambda a, sa, b, sb: _lib.lib().adf_fgs_filter_device(f._h, C.c_void_p(a.data_ptr()), sa, C.c_void_p(b.data_ptr()), sb, dcode, cn, st)
    assert call(vs, big_s.shape[1] * cn * esz, vd, big_d.shape[1] * cn * esz) == 0
    torch.cuda.synchronize()
    assert torch.equal(vd, want)
    assert int((big_d[:, :pad_r] != 7).sum()) == 0 and int((big_d[:, pad_r + w:] != 7).sum()) == 0   # padding untouched
    assert torch.equal(vs, s)                                                                        # source untouched
    assert call(vs, big_s.shape[1] * cn * esz, vs, big_s.shape[1] * cn * esz) == 0                   # in place
    torch.cuda.synchronize()
    assert torch.equal(vs, want)
    # overlapping but not identical: refused
    assert call(vs, big_s.shape[1] * cn * esz, big_s[:, pad_l + 1:pad_l + 1 + w], big_s.shape[1] * cn * esz) == _lib.ADF_EBADARG


@pytest.mark.gpu
def test_one_shot_calls_from_several_threads(adf):
    """Four threads, each on its own stream, make and destroy filters of two sizes back to back: blocks and weight tables
    change hands between threads and streams; every result must equal the one computed alone."""
    import threading

    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    cases = []
    for k in range(6):
        w, h = ((640, 360), (481, 203))[k % 2]
        guide, src = _case(rng, w, h, (1, 3)[k % 2], (np.uint8, np.float32, np.int16)[k % 3], (3, 1)[k % 2])
        g, s = torch.from_numpy(guide).to(dev), torch.from_numpy(src).to(dev)
        lam, sig = 1000.0 + 700 * k, 2.0 + 3 * (k % 3)
        f = adf.createFastGlobalSmootherFilter(g, lam, sig)
        want = f.filter(s).clone()
        torch.cuda.synchronize()
        del f
        cases.append((g, s, lam, sig, want))
    errors = []

    def work(tid):
        try:
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                for it in range(25):
                    g, s, lam, sig, want = cases[(tid + it) % len(cases)]
                    got = adf.fastGlobalSmootherFilter(g, s, lam, sig)
                    if it % 5 == 4:
                        st.synchronize()
                        if not torch.equal(got, want):
                            errors.append((tid, it))
                st.synchronize()
        except Exception as e:                      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors


@pytest.mark.gpu
def test_device_filter_call_captured_into_a_graph(adf):
    """adf_fgs_filter_device queues kernels only: it can be captured and replayed; a call that is being captured leaves
    the handle's ordering event alone (adf_api.hip: fgs_begin / fgs_end), later ordinary calls use it again, and a handle
    that was ever captured is freed, not cached, at destruction."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(13)
    guide, src = _case(rng, 640, 360, 3, np.float32, 3)
    g, s = torch.from_numpy(guide).to(dev), torch.from_numpy(src).to(dev)
    f = adf.createFastGlobalSmootherFilter(g, 4000.0, 9.0)
    want = f.filter(s).clone()
    torch.cuda.synchronize()
    dst = torch.zeros_like(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        f.filter(s, dst)
    for _ in range(3):
        dst.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dst, want)
    s2 = s.flip(0).contiguous()
    want2 = adf.createFastGlobalSmootherFilter(g, 4000.0, 9.0).filter(s2).clone()
    s.copy_(s2)                                   # the graph reads the same buffer
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(dst, want2)
    # Round 4: the capture state is not latched -- ordinary calls on the captured handle record and wait for
    # its event again, so two streams sharing the handle stay ordered: a long call on one stream, then a call on another
    # that overwrites the same planes, must both come out right.
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    s1 = s.flip(1).contiguous()
    want1 = adf.createFastGlobalSmootherFilter(g, 4000.0, 9.0).filter(s1).clone()
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(sa):
            da = f.filter(s1)
        with torch.cuda.stream(sb):
            db = f.filter(s)
        torch.cuda.synchronize()
        assert torch.equal(da, want1) and torch.equal(db, want2)
    del graph, f
    got = adf.fastGlobalSmootherFilter(g, s, 4000.0, 9.0)      # the library goes on working after the captured handle is gone
    torch.cuda.synchronize()
    assert torch.equal(got, want2)
