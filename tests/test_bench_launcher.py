"""`python bench.py --gpus N` without torchrun starts its own N workers.

Dry mode: the launcher, the rendezvous (gloo, 127.0.0.1), the barriers and scalar all-reduces and the single JSON
line run for real; the filter call is skipped, so no GPU is needed and `value` is 0 / `dry_run` true."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ADF_BENCH_WORKER"):
        env.pop(k, None)
    return env


def test_self_launch_two_ranks_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--pairs", "2"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["launcher"] == "self" and d["dry_run"] is True
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert len(d["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in d["per_rank_ms_per_step"])
    assert d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) - 1e-6      # max over ranks, barrier to barrier
    assert d["config"]["total_pairs"] == 4 and d["config"]["parallelism"] == "batch-sharded x2"
    # checksum all-reduce: rank r contributes pairs * 16 * (r + 1)
    assert d["checksum"] == 2 * 16 * (1 + 2)
    # round 3: who ran where, per rank, and the data-path exchange (scatter / gather / pipelined leg) runs by default at
    # N > 1 after the timed region -- here over gloo with CPU tensors, on a GPU node over RCCL
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and len({r["pid"] for r in d["ranks"]}) == 2
    assert all(set(("device", "name", "uuid", "pci_bus_id", "host", "local_rank")) <= set(r) for r in d["ranks"])
    assert d["backend"] == "gloo" and "distinct_devices" in d
    legs = d["rccl_legs"]
    assert legs["backend"] == "gloo" and legs["world_size"] == 2
    assert legs["scattered_shards_equal_locally_generated"] is True and legs["gathered_checksum_matches"] is True
    assert legs["scatter_ms"] > 0 and legs["gather_ms"] > 0
    pl = legs["pipelined_scatter_filter_gather"]
    assert pl["sub_batches"] >= 1 and pl["total_ms"] > 0 and pl["output_checksum_matches"] is True
    assert "cpu_baseline" in d and "path" in d


def test_rccl_legs_can_be_switched_off():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--rccl-legs", "off"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip())
    assert d["rccl_legs"] is None and d["n_gpus"] == 2


def test_already_under_a_launcher_is_a_worker():
    """With RANK / WORLD_SIZE in the environment (torch.distributed.run) bench.py must not spawn again."""
    env = _clean_env()
    port = __import__("socket").socket()
    port.bind(("127.0.0.1", 0))
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port.getsockname()[1]), WORLD_SIZE="2")
    port.close()
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"],
                                      cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    assert outs[1][0].strip() == ""
    d = json.loads(outs[0][0].strip())
    assert d["n_gpus"] == 2 and d["launcher"] == "torchrun"


def test_worker_failure_propagates():
    """A worker that dies takes the launcher's exit code with it (the driver must see rc != 0)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--config", "99"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "bench launcher: rank" in p.stderr


def test_self_launch_eight_ranks_rehearsal():
    """BASELINE config 4's world size (8 ranks, one node) on gloo: eight workers, one line, eight identities, the
    data-path legs (7-peer point-to-point groups, 4 pipelined sub-batches) run and verify themselves."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--pairs", "5"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and d["launcher"] == "self" and d["dry_run"] is True
    assert len(d["per_rank_ms_per_step"]) == 8
    assert [r["rank"] for r in d["ranks"]] == list(range(8)) and len({r["pid"] for r in d["ranks"]}) == 8
    assert [r["local_rank"] for r in d["ranks"]] == list(range(8))
    assert "distinct_devices" in d and d["config"]["total_pairs"] == 40 and d["config"]["parallelism"] == "batch-sharded x8"
    assert d["checksum"] == 5 * 16 * sum(range(1, 9))
    legs = d["rccl_legs"]
    assert legs["backend"] == "gloo" and legs["world_size"] == 8
    assert legs["scattered_shards_equal_locally_generated"] is True and legs["gathered_checksum_matches"] is True
    pl = legs["pipelined_scatter_filter_gather"]
    assert pl["sub_batches"] == 4 and pl["output_checksum_matches"] is True
    assert "extra_legs_hung" not in d


def test_rank_five_of_eight_failing_propagates():
    env = _clean_env()
    env["ADF_BENCH_TEST_FAIL_RANK"] = "5"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "0", "--pairs", "2",
                        "--launch-timeout", "120"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert "bench launcher: rank 5 exited with code 7" in p.stderr
    assert p.stdout.strip() == ""                                      # no line from a job that lost a rank


def test_hung_extra_leg_prints_the_line_and_fails():
    """a transfer that never finishes must not end as rc 0.  The watchdog prints the timed line once, marked,
    and the job exits non-zero."""
    env = _clean_env()
    env.update(ADF_BENCH_TEST_HANG_RANK="1", ADF_BENCH_WATCHDOG_S="6")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--pairs", "2",
                        "--launch-timeout", "120"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 5, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["extra_legs_hung"] is True and d["n_gpus"] == 2 and "watchdog" in d["rccl_legs"]
