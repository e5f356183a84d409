"""Short forms of the soak runs (tools/soak.py, tools/soak_threads.py; the long runs are profiles/r04_soak*.txt): every call of
a case returns the bits of its first call, memory in use does not grow, and host threads with their own handles get the
single-threaded results."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.gpu
def test_mixed_calls_are_deterministic_and_do_not_leak():
    out = _run(["tools/soak.py", "6"])
    assert "bit-identical" in out and "growth 0.000" in out


@pytest.mark.gpu
def test_threads_with_their_own_handles_get_the_single_threaded_bits():
    out = _run(["tools/soak_threads.py", "3", "12"])
    assert "bit-identical to the single-threaded run" in out
