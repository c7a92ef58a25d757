"""LEOPowerAttitudeSimulator — the scenario object the gym env owns, with the Basilisk engine
replaced by the batched HIP propagator.

Mirrors the inner surface of reference ``simulators/leoPowerAttitudeSimulator.py``:
``LEOPowerAttitudeSimulator(dynRate, fswRate, step_duration, initial_conditions=None)`` (:67),
``.initial_conditions`` dict with the schema of ``set_ICs`` (:119-193), ``.obs`` (5,1) (:105,
filled from the ICs at :348-351), ``.run_sim(action) -> (obs, sim_states, sim_over)`` (:535-644)
and ``.close_gracefully()`` (:646-652).  One instance drives ONE spacecraft (a batch of 1); the
batched surface is ``envs.LeoPowerAttVecEnv``.
"""
import math

import numpy as np

from .._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM
from .dynamics import config as _config
from .dynamics.effectorPrimatives import actuatorPrimatives as ap
from .dynamics.propagator import BatchedPropagator, pack_ic
from .initial_conditions import leo_orbit, sc_attitudes

RPM = _config.RPM


def ic_dict_to_block(ic, n_rw):
    """One reference-style IC dict -> SoA block ``[n_fields, 1]`` for the propagator."""
    wheel = np.zeros(n_rw)
    ws = np.asarray(ic.get("wheelSpeeds", np.zeros(3)), dtype=float) * RPM  # dict holds RPM (:155,303-305)
    wheel[:min(n_rw, ws.size)] = ws[:n_rw]
    lext = float(ic.get("disturbance_magnitude", 0.0)) * np.asarray(ic.get("disturbance_vector", np.zeros(3)), dtype=float)
    return pack_ic(n_rw, np.asarray(ic["rN"], float).reshape(1, 3), np.asarray(ic["vN"], float).reshape(1, 3),
                   np.asarray(ic["sigma_init"], float).reshape(1, 3), np.asarray(ic["omega_init"], float).reshape(1, 3),
                   wheelSpeeds=wheel.reshape(1, -1) if n_rw else None, lext=lext.reshape(1, 3),
                   charge=[float(ic.get("storedCharge_Init", 0.0))])


# The reference deletes and rebuilds its simulator on every reset (envs/leoPowerAttitudeEnvironment.py:184-185).
# A propagator handle is 11 device allocations and a stream, none of which depend on the initial conditions, so
# released handles wait here and the next simulator with the same configuration takes one over (reset = upload
# of one IC block).  Keyed by the full bsk_config bytes, batch size, device and factory; a few entries at most.
_IDLE_PROPAGATORS = {}
_IDLE_MAX = 4


def _prop_key(cfg, n_envs, device, factory):
    import ctypes
    return (ctypes.string_at(ctypes.addressof(cfg), ctypes.sizeof(cfg)), int(n_envs), int(device), factory)


def acquire_propagator(cfg, n_envs, device, factory):
    key = _prop_key(cfg, n_envs, device, factory)
    idle = _IDLE_PROPAGATORS.get(key)
    if idle:
        prop = idle.pop()
        # handle-level state that reset(ic) does not touch goes back to its create-time value: the Sun epoch offset
        # and an armed profile (handles carrying harmonics tables or a device IC pool are never parked, see release)
        prop.set_sim_time(0.0)
        if hasattr(prop, "profile_end"):
            prop.profile_end()
        return prop, key
    return factory(cfg, n_envs, device=device), key


def release_propagator(prop, key):
    """Park a propagator for re-use, or close it when enough are parked already."""
    if prop is None:
        return
    from .._lib import FLAG_AUTO_RESET, GRAV_SH
    cfg = getattr(prop, "cfg", None)
    stateful = cfg is not None and (cfg.gravity_model == GRAV_SH or (cfg.flags & FLAG_AUTO_RESET))
    if stateful or sum(len(v) for v in _IDLE_PROPAGATORS.values()) >= _IDLE_MAX:
        prop.close()
        return
    _IDLE_PROPAGATORS.setdefault(key, []).append(prop)


def drain_idle_propagators():
    """Close every parked propagator (process exit, tests)."""
    for v in _IDLE_PROPAGATORS.values():
        while v:
            v.pop().close()
    _IDLE_PROPAGATORS.clear()


class LEOPowerAttitudeSimulator(object):
    """Drop-in for the reference simulator class; see the module docstring for the surface.

    Extra keyword-only knobs (not in the reference): ``n_rw`` (3 = the reference's triad),
    ``gravity_model``, ``device`` and ``propagator_factory`` (dependency injection for tests).
    """

    def __init__(self, dynRate, fswRate, step_duration, initial_conditions=None, *, n_rw=3,
                 gravity_model=GRAV_PM, device=0, propagator_factory=None):
        self.dynRate = dynRate
        self.fswRate = fswRate
        self.step_duration = step_duration
        self.simTime = 0.0
        self.n_rw = n_rw

        if initial_conditions is None:
            self.initial_conditions = self.set_ICs()
        else:
            self.initial_conditions = initial_conditions
        self.mass = self.initial_conditions.get("mass")
        self.powerDraw = self.initial_conditions.get("powerDraw")

        self.obs = np.zeros([5, 1])
        self.sim_states = np.zeros([11, 1])
        self.sim_over = False
        self.modeRequest = None

        ic = self.initial_conditions
        cfg = _config.default_config(n_rw=n_rw, gravity_model=gravity_model, mass=ic.get("mass"), width=ic.get("width"),
                                     depth=ic.get("depth"), height=ic.get("height"))
        cfg.dt = float(dynRate)
        k = float(fswRate) / float(dynRate)
        if abs(k - round(k)) > 1e-9 or round(k) < 1:
            raise ValueError("fswRate must be a positive integer multiple of dynRate")
        cfg.fsw_every = int(round(k))
        n_sub = float(step_duration) / float(dynRate)
        if abs(n_sub - round(n_sub)) > 1e-9 or round(n_sub) < 1:
            raise ValueError("step_duration must be a positive integer multiple of dynRate")
        self.substeps = int(round(n_sub))
        cfg.K, cfg.P = float(ic.get("K")), float(ic.get("P"))
        for j in range(3):
            cfg.sigma_R0N[j] = float(ic.get("sigma_R0N")[j])
        for j in range(9):
            cfg.ctrl_axes[j] = float(ic.get("controlAxes_B")[j])
        cfg.power_draw = float(ic.get("powerDraw"))
        cfg.storage_capacity = float(ic.get("storageCapacity"))
        cfg.panel_area = float(ic.get("panelArea"))
        cfg.panel_efficiency = float(ic.get("panelEfficiency"))
        for j in range(3):
            cfg.panel_normal[j] = float(ic.get("nHat_B")[j])
        # the reference scenario: eclipse + solar panel + battery + sink (:286-288, 326-345), Sun as a
        # third body (:227-232), exponential atmosphere + facet drag (:265-284)
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        if n_rw:
            # action 2: momentum dumping with the Monarc-1 octet (:313-318, 452-478, 574-588)
            cfg.flags |= FLAG_DESAT
            cfg.hs_min = float(ic.get("hs_min"))
            cfg.thr_max_counter = int(ic.get("maxCounterValue"))
            cfg.thr_min_fire_time = float(ic.get("thrMinFireTime"))
        cfg.base_density = float(ic.get("baseDensity"))
        cfg.scale_height = float(ic.get("scaleHeight"))
        self.cfg = cfg

        factory = propagator_factory or BatchedPropagator
        self.propagator, self._prop_key = acquire_propagator(cfg, 1, device, factory)
        self.propagator.reset(ic_dict_to_block(ic, n_rw))

        # initial observation exactly as the reference fills it (:348-351): |sigma_BN| (not
        # sigma_BR), |omega|, |wheel speeds| in RPM (every later step reports rad/s), charge in W h
        self.sim_states[0:3, 0] = np.asarray(ic["sigma_init"]).flatten()
        self.sim_states[3:6, 0] = np.asarray(ic["rN"]).flatten()
        self.sim_states[6:9, 0] = np.asarray(ic["vN"]).flatten()
        self.sim_states[9, 0] = ic.get("storedCharge_Init")
        self.obs[0, 0] = np.linalg.norm(ic["sigma_init"])
        self.obs[1, 0] = np.linalg.norm(ic["omega_init"])
        self.obs[2, 0] = np.linalg.norm(ic["wheelSpeeds"])
        self.obs[3, 0] = ic.get("storedCharge_Init") / 3600.0

    def set_ICs(self):
        """Random initial conditions with the reference's schema, distributions and legacy-RNG draw
        order (:119-193): orbit, tumble, disturbance vector, wheel speeds, battery charge."""
        oe, rN, vN = leo_orbit.sampled_400km()
        sigma_init, omega_init = sc_attitudes.random_tumble(maxSpinRate=0.00001)
        initial_conditions = {
            "mass": 330,
            "oe": oe, "rN": rN, "vN": vN,
            "width": 1.38, "depth": 1.04, "height": 1.58,
            "sigma_init": sigma_init, "omega_init": omega_init,
            "planetRadius": _config.REQ_EARTH_KM * 1000., "baseDensity": 1.22, "scaleHeight": 8e3,
            "disturbance_magnitude": 2e-4,
            "disturbance_vector": np.random.standard_normal(3),
            "wheelSpeeds": np.random.uniform(-800, 800, 3),  # RPM
            "nHat_B": np.array([0, -1, 0]), "panelArea": 0.2 * 0.3, "panelEfficiency": 0.20,
            "powerDraw": -5.0,
            "storageCapacity": 20.0 * 3600.,
            "storedCharge_Init": np.random.uniform(8. * 3600., 20. * 3600., 1)[0],
            "sigma_R0N": [1, 0, 0],
            "controlAxes_B": [1, 0, 0, 0, 1, 0, 0, 0, 1],
            "K": 7, "Ki": -1.0, "P": 35,
            "hs_min": 4.,
            "thrForceSign": 1,
            "maxCounterValue": 4,
            "thrMinFireTime": 0.002,
        }
        # the reference's set_dynamics draws (and discards) one more wheel-speed triple inside
        # balancedHR16Triad(useRandom=True) (:301, actuatorPrimatives.py:18); keep the stream aligned
        ap.balancedHR16Triad(useRandom=True, randomBounds=(-800, 800))
        return initial_conditions

    def run_sim(self, action):
        """Advance ``step_duration`` seconds under mode ``action`` (0 nadir, 1 sun-point, 2 desat)
        and return ``(obs (5,1), sim_states, sim_over)`` like the reference (:535-644)."""
        self.modeRequest = str(action)
        if self.modeRequest not in ("0", "1", "2"):
            raise ValueError("action must be 0, 1 or 2, got %r" % (action,))
        self.sim_over = False
        self.simTime += self.step_duration
        self.propagator.step(np.array([int(self.modeRequest)], dtype=np.int32), self.substeps)
        if hasattr(self.propagator, "get_obs_state"):
            dev_obs, st = self.propagator.get_obs_state()      # one synchronisation for both read-backs
        else:
            dev_obs, _, _, _ = self.propagator.get_obs()
            st = self.propagator.get_state()
        omega = st[9:12, 0]
        wheels = st[12:12 + min(self.n_rw, 3), 0]  # the reference logs wheelSpeeds[0:3] (:606,614)
        charge = st[12 + self.n_rw + 7, 0]
        obs = np.hstack([dev_obs[0, 0], np.linalg.norm(omega), np.linalg.norm(wheels), charge / 3600., dev_obs[4, 0]])
        self.obs = obs.reshape(len(obs), 1)
        self.sim_states = []
        if np.linalg.norm(st[0:3, 0]) < (_config.REQ_EARTH_KM / 1000.):
            self.sim_over = True
        return self.obs, self.sim_states, self.sim_over

    def release(self):
        """Hand the propagator back for the next simulator (the env calls this where the reference does
        ``del self.simulator``); the object must not be stepped afterwards."""
        prop, self.propagator = self.propagator, None
        release_propagator(prop, self._prop_key)

    def close_gracefully(self):
        """The reference unloads SPICE kernels here (:646-652); this build has none to unload.
        The device buffers stay valid until the object is deleted (reset_init reads
        ``initial_conditions`` after this call)."""
        return


def create_leoPowerAttSimulator():
    return LEOPowerAttitudeSimulator(0.1, 0.1, 60.)


def demo(steps=2 * 180, action=0, step_duration=60., plot=False, **simulator_kwargs):
    """What the reference module does when run as a script (:657-694): one simulator with 60 s steps, ``steps`` calls of
    ``run_sim(action)`` (the reference draws none at random either: ``act = 0``), the observation history kept and - with
    ``plot=True`` and matplotlib at hand - drawn with the reference's labels.  -> observations (steps, 5)."""
    sim = LEOPowerAttitudeSimulator(0.1, 1.0, step_duration, **simulator_kwargs)
    obs = []
    for _ in range(steps):
        ob, _, over = sim.run_sim(action)
        obs.append(ob[:, 0].copy())
        if over:
            break
    sim.close_gracefully()
    if hasattr(sim, "release"):
        sim.release()
    obs = np.asarray(obs)
    print("simulator demo: %d steps of %.0f s under action %d, last observation %s" % (len(obs), step_duration, action, np.array2string(obs[-1], precision=5)))
    if plot:
        from matplotlib import pyplot as plt
        plt.figure()
        for k, name in enumerate(("sigma_BR", "omega_BN", "omega_rw", "J_bat", "eclipse_ind")):
            plt.plot(range(len(obs)), obs[:, k], label=name)
        plt.legend()
        plt.show()
    return obs


if __name__ == "__main__":
    import sys
    demo(plot="--plot" in sys.argv)
