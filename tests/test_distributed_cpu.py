"""N > 1 on CPU: world_size-2 (and 3) gloo runs of the sharding + exchange logic."""
import os

import pytest

from util import torchrun

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = torchrun(world, os.path.join(HERE, "dist_worker.py"), env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK" in p.stdout, p.stdout[-3000:]
    assert p.stdout.count("DIST_CASE") == 3


@pytest.mark.parametrize("world", [1, 2, 3])
def test_halo_path_gloo(world):
    """Rank-local build + halo exchange: each rank holds only its own rows and receives only the x
    entries they reference (ghost slots), checked against the oracle on the rank's rows."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = torchrun(world, os.path.join(HERE, "halo_worker.py"), env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "HALO_OK" in p.stdout, p.stdout[-3000:]
    assert p.stdout.count("HALO_CASE") == 9 + 8    # nine structured cases (two of them the "cover" exchange) + eight fuzz seeds
