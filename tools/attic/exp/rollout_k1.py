#!/usr/bin/env python3
"""(needs profiles/r06/rejected/rollout_k1_overlap.diff applied and `make tunables`; expects to sit in tools/exp/)
Rollout kernel, K = 1, constant action: the ordinary kernel against the overlapped-epilogue form (round 6 experiment; `make tunables`
library, BSKGPU_ROLLOUT_K1=0|1 read by bsk_step_n).  One process per form, alternating on one box; wall time per env step and the
buffers' hash (the two forms must leave identical bits).  usage (GPU box): python3 tools/exp/rollout_k1.py [N [T [ROUNDS]]]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 541
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if os.environ.get("ROLLOUT_K1_CHILD"):
    import numpy as np
    import torch
    from basilisk_env_amd._lib import GRAV_PM_J2
    from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
    cfg = default_config(4, GRAV_PM_J2)
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=6))
    ob = torch.empty((T, 5, n), dtype=torch.float64, device="cuda")
    rw = torch.empty((T, n), dtype=torch.float64, device="cuda")
    wy = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    p.step_n(T, 1, None, 0, ob.data_ptr(), rw.data_ptr(), wy.data_ptr())
    p.sync()
    h = hashlib.sha256(ob.cpu().numpy().tobytes() + rw.cpu().numpy().tobytes() + wy.cpu().numpy().tobytes() + p.get_state().tobytes()).hexdigest()[:16]
    reps = 30 if n <= 1 << 20 else 3
    for _ in range(reps):
        p.step_n(T, 1, None, 0, ob.data_ptr(), rw.data_ptr(), wy.data_ptr())
    p.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.step_n(T, 1, None, 0, ob.data_ptr(), rw.data_ptr(), wy.data_ptr())
    p.sync()
    dt = (time.perf_counter() - t0) / reps
    print("K1=%s  n %d T %d: %.3f us per env step  (%s, first launch's buffers %s)" % (os.environ.get("BSKGPU_ROLLOUT_K1"), n, T, dt / T * 1e6, p.kernel_info()["name"], h))
    sys.exit(0)
for r in range(rounds):
    for k1 in ("0", "1"):
        env = dict(os.environ, ROLLOUT_K1_CHILD="1", BSKGPU_ROLLOUT_K1=k1, BSKGPU_LIB=os.path.join(ROOT, "basilisk_env_amd", "variants", "tunables.so"))
        res = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
        sys.stdout.write(res.stdout if res.returncode == 0 else "K1=%s failed: %s\n" % (k1, res.stderr[-800:]))
        sys.stdout.flush()
