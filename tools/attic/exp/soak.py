"""Soak: many back-to-back launches of the wave-split forms (three-wave form at 16 384 spacecraft and for ONE spacecraft, pair form at
8 192), random actions, device-side auto-reset on; every synchronising call checks the handle's error word (exchange time-outs),
the state must stay finite.  usage: tools/exp/soak.py [seconds per configuration]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd._lib import FLAG_AUTO_RESET, FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for name, n, env in (("tri 16384", 16384, {"BSKGPU_TRI": "1", "BSKGPU_PAIR": "0"}), ("tri 1", 1, {"BSKGPU_TRI": "1", "BSKGPU_PAIR": "0"}),
                     ("pair 8192", 8192, {"BSKGPU_TRI": "0", "BSKGPU_PAIR": "1"})):
    os.environ.update(env)
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT | FLAG_AUTO_RESET
    cfg.max_length = 50
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=1))
    p.set_ic_pool(sample_ic_batch(256, 4, seed=2))
    rng = np.random.default_rng(0)
    t0 = time.perf_counter(); launches = 0
    while time.perf_counter() - t0 < budget:
        for _ in range(20):
            p.step(rng.integers(0, 3, n).astype(np.int32), int(rng.choice((180, 600, 1800))))
            launches += 1
        s = p.get_state()                       # synchronises: BSK_EHIP here if a kernel raised the error word
        assert np.isfinite(s).all()
    _, eps = p.get_terminal_obs()
    print("%-10s %5d launches in %.1f s, kernel %s, episodes finished %d, state finite, no device error" % (name, launches, time.perf_counter() - t0, p.kernel_info()["name"], int(eps.sum())), flush=True)
    p.close()
