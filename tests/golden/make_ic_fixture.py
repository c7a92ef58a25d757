#!/usr/bin/env python3
"""Generate tests/golden/ic_random_tumble.json from the reference's own sampler.

``basilisk_env/simulators/initial_conditions/sc_attitudes.py`` is the one module of the
reference's hot-path neighbourhood that imports without Basilisk (numpy only).  This script
loads THAT FILE from /root/reference at generation time (it is not copied anywhere), calls
``random_tumble`` under a few legacy-RNG seeds and stores inputs and outputs as data.
Run in the build container only:  python tests/golden/make_ic_fixture.py
"""
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference/basilisk_env/simulators/initial_conditions/sc_attitudes.py"
spec = importlib.util.spec_from_file_location("ref_sc_attitudes", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

cases = []
for seed, max_rate in [(0, 1e-5), (12345, 1e-5), (7, 0.001), (2021, 0.05)]:
    np.random.seed(seed)
    sigma, omega = mod.random_tumble(maxSpinRate=max_rate)
    sigma2, omega2 = mod.random_tumble(maxSpinRate=max_rate)   # second draw from the same stream
    cases.append({"seed": seed, "maxSpinRate": max_rate, "sigma": sigma.tolist(), "omega": omega.tolist(),
                  "sigma_2": sigma2.tolist(), "omega_2": omega2.tolist()})
s, w = mod.static_inertial()
out = {"source": "reference basilisk_env/simulators/initial_conditions/sc_attitudes.py:3-23 run under numpy %s" % np.__version__,
       "random_tumble": cases, "static_inertial": {"sigma": s.tolist(), "omega": w.tolist()}}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ic_random_tumble.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
