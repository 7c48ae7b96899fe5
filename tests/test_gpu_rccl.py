"""Runs tests/rccl_cases.py -- the RCCL-native exchange step over a one-rank communicator that sends to itself -- in ONE child
process that imports torch before libehyb.so is loaded (one HIP runtime for torch's RCCL and the library: the order bench.py
has).  The pytest process itself has usually loaded libehyb.so (and with it /opt/rocm's HIP runtime) long before."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_rccl_native_step_over_a_loopback_communicator(gpu):
    code = "import sys, torch, pytest; sys.exit(pytest.main(['-x', '-q', '-s', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/rccl_cases.py']))"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=dict(os.environ))
    tail = p.stdout[-4000:] + p.stderr[-2000:]
    assert p.returncode == 0, tail
    m = re.search(r"(\d+) passed", p.stdout)
    assert m and int(m.group(1)) >= 10, tail
    m = re.search(r"host_us_per_step \(C step, 2 chunks, world 1\): ([0-9.]+)", p.stdout)
    assert m, tail
    for line in p.stdout.splitlines():
        if "cover:" in line:
            print(line)
    print(f"host_us_per_step of the C step: {m.group(1)}")
