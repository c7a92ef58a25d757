"""CPU, world_size 2 and 4 over gloo: env-range sharding with no step-path collective, and the
observation all-gather / gather-to-root reproduce the single-process batch exactly."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.parallel import shard_range, shard_sizes
from basilisk_env_amd.simulators.dynamics.config import default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_ranges_partition():
    for n, w in [(1048576, 8), (65536, 8), (10, 3), (7, 8), (131, 2)]:
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    assert shard_sizes(1048576, 8) == [131072] * 8
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_total", [(2, 128), (2, 131), (4, 130)])     # equal and ragged shards; four ranks (33, 33, 32, 32)
def test_two_rank_sharded_step_and_gather(tmp_path, world, n_total):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py"), str(n_total), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    # single-process reference on the whole batch
    cfg = default_config(4, GRAV_PM_J2)
    st = sample_ic_batch(n_total, 4, seed=42)
    steps, ticks = np.zeros(n_total, np.int32), np.zeros(n_total, np.int32)
    actions = (np.arange(n_total) % 3).astype(np.int32)
    for k in (10, 7):
        obs, rew, done, why = oracle.step(cfg, st, steps, ticks, actions, k)
    full = np.load(tmp_path / "obs_full.npy")
    assert full.shape == (5, n_total)
    assert np.array_equal(full, obs)                                   # env-index order preserved
    assert np.array_equal(np.load(tmp_path / "obs_root.npy"), obs)
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / ("obs_rank%d.npy" % r)), obs)    # all-gather: every rank has it
    assert abs(np.load(tmp_path / "rew_sum.npy")[0] - rew.sum()) < 1e-12
