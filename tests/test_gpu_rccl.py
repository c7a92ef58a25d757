"""GPU: the observation gather and the reward all-reduce over the real RCCL backend (backend "nccl"), with the
HIP propagator's zero-copy device views.  One rank (the GPU box has one card); world_size 2 runs on CPU over
gloo in tests/test_parallel_gloo.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics.config import default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
from oracle import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_gather_single_rank(tmp_path):
    n_total = 4096
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "_rccl_worker.py"), str(n_total), str(tmp_path)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    cfg = default_config(4, GRAV_PM_J2)
    st = sample_ic_batch(n_total, 4, seed=42)
    steps, ticks = np.zeros(n_total, np.int32), np.zeros(n_total, np.int32)
    actions = (np.arange(n_total) % 3).astype(np.int32)
    for k in (10, 7):
        obs, rew, done, why = oracle.step(cfg, st, steps, ticks, actions, k)
    full = np.load(tmp_path / "obs_full.npy")
    assert full.shape == (5, n_total)
    assert np.abs(full - obs).max() < 1e-11
    assert np.array_equal(np.load(tmp_path / "obs_root.npy"), full)
    assert abs(np.load(tmp_path / "rew_sum.npy")[0] - rew.sum()) < 1e-10
