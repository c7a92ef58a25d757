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
    # the direct leg, rank-major root layout: the same observations, rewards and reasons, one block per rank
    import json
    assert np.array_equal(np.load(tmp_path / "rm_obs.npy"), full)
    assert np.abs(np.load(tmp_path / "rm_rew.npy") - rew).max() < 1e-12 and np.array_equal(np.load(tmp_path / "rm_why.npy"), why)
    meta = json.load(open(tmp_path / "meta.json"))
    assert meta["comm_count"] == 1 and meta["messages_on_root"] == 0          # one rank: RCCL says so; nothing crosses the fabric
    # ... and the batch scalars' all-reduce twice between two steps: the same sums both times (in place it doubled)
    assert meta["sums"][0] == meta["sums"][1]
    assert abs(meta["sums"][0][0] - rew.sum()) < 1e-10 and meta["sums"][0][1] == float((why != 0).sum())


def test_bench_collective_legs_on_rccl_single_rank(tmp_path):
    """bench.py under torch.distributed.run with ONE rank: the process group is RCCL (backend nccl), so the
    observation-exchange legs (gather to rank 0, all-gather, direct D2H) and the configs[3] / strong-scaling extras
    run on the real backend — the N > 1 control flow itself is rehearsed over gloo (BENCH_REHEARSAL) and in the
    CPU tests."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "5",
           "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BENCH_REHEARSAL", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out_lines = [l for l in res.stdout.splitlines() if l.strip()]
    # the LAST stdout line is the compact one the driver reads; the whole record rides on the EXTRA line before it
    assert out_lines[-1].startswith("{") and len(out_lines[-1]) < 4096 and out_lines[-2].startswith("EXTRA {"), [l[:80] for l in out_lines[-3:]]
    h = json.loads(out_lines[-1])
    d = json.loads(out_lines[-2][len("EXTRA "):])
    assert h["n_gpus"] == 1 and h["value"] == pytest.approx(d["value"], rel=1e-5) and h["value_with_join"] > 0 and isinstance(h.get("extra", ""), str)
    assert h["ranks"] == 1 and h["distinct_devices"] == 1 and h["gather"]["nccl_comm_count"] == 1 and h["gather_ms"] > 0
    assert {"torch_gather_ms", "torch_all_gather_ms", "direct_7row_ms", "direct_rank_major_ms", "all_reduce_stats_ms"} <= set(h["gather"]), h["gather"]
    assert h["strong_65536_total"]["k1_env_steps_per_s"] > 0 and h["strong_65536_total"]["full_k1800_env_steps_per_s"] > 0 and h["config3_env_steps_per_s"] > 0
    assert d["n_gpus"] == 1 and d["value"] > 0
    g = d["gather"]
    assert g["gather_to_rank0_ms"] > 0 and g["all_gather_ms"] > 0 and g["direct_d2h_per_gpu_ms"] > 0
    assert g["shard_bytes"] == 5 * 65536 * 8
    # the direct librccl leg (its own communicator from ncclCommInitRank, gather on the handle's stream) ran and agrees
    assert g.get("direct_rccl_matches_torch_gather") is True and g["direct_rccl_gather_to_rank0_ms"] > 0, g
    # both root layouts of the seven-row form, their message counts, RCCL's own rank count, the all-reduce after repeated calls
    assert g["direct_rccl_gather7_to_rank0_ms"] > 0 and g["direct_rccl_gather7_rank_major_ms"] > 0, g
    assert g.get("direct_rccl_rank_major_matches_own_shard") is True and g.get("all_reduce_stats_matches_host_sums") is True, g
    assert g["nccl_comm_count"] == 1 and set(g["messages_on_root"]) == {"direct_rccl_gather_to_rank0", "direct_rccl_gather7_to_rank0", "direct_rccl_gather7_rank_major"}
    assert len(d["ranks"]) == 1 and d["ranks"][0]["process_group_size"] == 1 and d["ranks"][0]["backend"] == "nccl" and d["distinct_devices"] == 1
    assert d["config"]["batch_stats"] == h["config"]["batch_stats"] == "per-wave sums in the step launch; join on demand"
    x = d["extra"]
    assert "1048576" not in x["config3"]["workload"] and "131072 per GPU" in x["config3"]["workload"]
    assert x["config3"]["gather"]["shard_bytes"] == 5 * 131072 * 8
    assert x["strong_65536_total"]["k1800_env_steps_per_s"] > 1e6


def test_a_hung_leg_under_the_launcher_still_yields_the_line_and_a_failure():
    """The watchdog's failure path end to end (rehearsal: two ranks on this one card over gloo, the auxiliary leg replaced by one that
    never returns): rank 0 prints the ONE JSON line with the timeout recorded in it, both ranks end with status 3, and the launcher -
    hence `python bench.py --gpus 2` as typed - reports a failure instead of a clean exit."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BENCH_REHEARSAL="1", BENCH_FAULT_HANG_LEG="1", BENCH_LEG_DEADLINE_S="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "5", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode != 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and res.stdout.strip().splitlines()[-1] == lines[0], (lines, res.stderr[-2000:])
    h = json.loads(lines[0])
    assert h["n_gpus"] == 2 and h["value"] > 0 and "timeout after 2 s" in h["gather"]["direct_rccl"] and h["ranks"] == 2
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("EXTRA {")][-1][len("EXTRA "):])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "timeout after 2 s" in d["gather"]["direct_rccl"]
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["process_group_size"] == 2 for r in d["ranks"])


def test_rehearsal_of_gpus_4_ends_in_one_compact_line_and_status_0():
    """VERDICT r05 #4b: `python bench.py --gpus 4` as typed, rehearsed on this one card (BENCH_REHEARSAL=1: every rank on device 0,
    gloo instead of RCCL - numbers from it mean nothing): the self-launch reaches four ranks, the weak-scaling step region, the
    torch exchange legs, configs[3]'s per-GPU share and the strong-scaling points of the literal target all run, rank 0 prints the
    EXTRA line and - LAST - the one compact line carrying what the first real multi-GPU record has to state."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BENCH_FAULT_HANG_LEG", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "50", "--warmup", "5"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out_lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len([l for l in out_lines if l.startswith("{")]) == 1 and out_lines[-1].startswith("{") and len(out_lines[-1]) < 4096
    h = json.loads(out_lines[-1])
    assert h["n_gpus"] == 4 and h["ranks"] == 4 and h["scaling"] == "weak" and h["value"] > 0 and h["value_with_join"] > 0
    assert h["distinct_devices"] == 1                        # the rehearsal's four ranks share the one card - a real run must say 4
    assert h["gather_ms"] > 0 and h["gather"]["torch_gather_ms"] > 0 and h["gather"]["torch_all_gather_ms"] > 0
    st = h["strong_65536_total"]
    assert st["envs_per_gpu"] == 16384 and st["k1_env_steps_per_s"] > 0 and st["full_k1800_env_steps_per_s"] > 0
    assert h["config3_env_steps_per_s"] > 0 and "cpu_baseline" not in h     # (the CPU baseline is rank 0 at N = 1 only)
    d = json.loads([l for l in out_lines if l.startswith("EXTRA {")][-1][len("EXTRA "):])
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3] and all(r["process_group_size"] == 4 and r["backend"] == "gloo" for r in d["ranks"])
    assert d["extra"]["config3"]["gather"]["shard_bytes"] == 5 * 131072 * 8
