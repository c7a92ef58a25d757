/*
 * bskgpu.h — C-ABI of libbskgpu.so: batched MI355X (gfx950) spacecraft propagator.
 *
 * This is the drop-in boundary for ONE hot path of atharris/basilisk_env: the per-env-step
 * call into the Basilisk engine,
 *     LEOPowerAttitudeSimulator.run_sim -> ConfigureStopTime + ExecuteSimulation
 *     (reference basilisk_env/simulators/leoPowerAttitudeSimulator.py:535-644, hot call :594-595)
 * for N independent spacecraft at once.  Plain pointers and sizes only; no torch / numpy types.
 * Every entry point returns 0 on success or a negative BSK_E* code and never throws.
 * `bsk_last_error()` returns a thread-local message for the last failure on this thread.
 *
 * Ownership: the caller owns every host buffer; the library owns every device buffer.
 * One handle <-> one device <-> one HIP stream.  A handle is not thread-safe; distinct handles are.
 * There is NO CPU fallback: bsk_create fails with BSK_ENODEV when no gfx950 device is usable.
 *
 * Layouts are structure-of-arrays, fp64: field f of env i lives at buf[f * n_envs + i]
 * on the host side of this ABI (the device side pads the env stride to a multiple of 256).
 */
#ifndef BSKGPU_H
#define BSKGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSK_ABI_VERSION 4u
#define BSK_MAX_RW 4
#define BSK_MAX_THR 8
#define BSK_MAX_SH_DEGREE 70

/* error codes */
#define BSK_OK 0
#define BSK_EINVAL (-1)   /* bad argument / config */
#define BSK_ENODEV (-2)   /* no usable gfx950 device */
#define BSK_ENOMEM (-3)   /* device allocation failed */
#define BSK_EHIP (-4)     /* HIP runtime error (see bsk_last_error) */
#define BSK_EABI (-5)     /* bsk_config abi_version / struct_size mismatch */

/* gravity models (reference: gravBodyFactory, leoPowerAttitudeSimulator.py:217-232; the only
 * spherical-harmonics call site is opNav_models/BSK_OpNavDynamics.py:211-214) */
enum { BSK_GRAV_PM = 0, BSK_GRAV_PM_J2 = 1, BSK_GRAV_SH = 2 };

/* flags */
#define BSK_FLAG_SUN_THIRD_BODY 0x1u /* Sun third-body gravity (leoPowerAttitudeSimulator.py:227) */
#define BSK_FLAG_POWER 0x2u          /* eclipse + solar panel + battery (…Simulator.py:286-288,326-345) */
#define BSK_FLAG_DESAT 0x4u          /* action 2 fires the thruster octet (…Simulator.py:574-588) */
#define BSK_FLAG_DRAG 0x8u           /* exponential atmosphere + facet drag (…Simulator.py:265-284) */
#define BSK_FLAG_AUTO_RESET 0x10u    /* device-side masked auto-reset from a staged IC pool */
#define BSK_FLAG_EPISODE_STATS 0x40u /* device-resident episode statistics (the Monitor convention of envs/leoPowerAttitudeEnvironment.py:130-135
                                        without the host): running return per env, return / length of the episode that just ended,
                                        a 0 / 1 done byte; bsk_get_episode_device.  16 B more HBM traffic per env-step */
#define BSK_FLAG_OBS_ROWMAJOR 0x80u  /* the step kernel also writes the observation row-major, f64[n_envs][5] (what the reference's
                                        (5,1) observation stacks to, envs/leoPowerAttitudeEnvironment.py:43-45): a policy on the GPU
                                        reshapes it without a copy kernel.  40 B more per env-step */
#define BSK_FLAG_LDS_SCRATCH 0x20u   /* bare propagator only: stage the RK4 accumulator (12 doubles per spacecraft) in
                                        LDS between the stages instead of VGPRs: 3 waves per SIMD instead of 2.
                                        Same results bit for bit; timings in DESIGN.md §4 */

/* State field indices of the SoA state block, bsk_get_state / bsk_set_state / bsk_reset `ic`.
 * n_fields = BSK_NF_BASE + n_rw (wheel speeds) + BSK_NF_TAIL.                                   */
enum {
    BSK_F_R = 0,      /* r_BN_N   [m]      3 fields  (scObject.hub.r_CN_NInit,  …Simulator.py:252) */
    BSK_F_V = 3,      /* v_BN_N   [m/s]    3 fields  (…Simulator.py:253)                            */
    BSK_F_SIGMA = 6,  /* sigma_BN [-]      3 fields  (…Simulator.py:258)                            */
    BSK_F_OMEGA = 9,  /* omega_BN_B [rad/s] 3 fields (…Simulator.py:259)                            */
    BSK_NF_BASE = 12, /* wheel speeds Omega_i [rad/s] follow: n_rw fields (…Simulator.py:303-305)   */
};
/* tail fields after the wheel speeds */
enum {
    BSK_T_LEXT = 0,    /* external disturbance torque L_B [N m] 3 fields (…Simulator.py:291-298)    */
    BSK_T_UCMD = 3,    /* held wheel motor torque u_s [N m]: BSK_MAX_RW fields (zero-order hold)    */
    BSK_T_CHARGE = 7,  /* battery stored charge [W s] 1 field (…Simulator.py:343)                   */
    /* desaturation state (BSK_FLAG_DESAT; zero otherwise) */
    BSK_T_THR_REM = 8,  /* thrMomentumDumping: on-time still owed per thruster [s], BSK_MAX_THR fields */
    BSK_T_THR_LIM = 16, /* current burst: on-time per thruster in half dyn steps (integer-valued)     */
    BSK_T_THR_T0 = 24,  /* RK4 tick at which the current burst started                               */
    BSK_T_THR_CNT = 25, /* thrMomentumDumping counter (control periods until the next burst)         */
    /* FSW task order (bsk_config.fsw_lag): the wheel torque that the NEXT FSW tick will apply, i.e.
     * rwMotorTorque(MRP_Feedback(att_guidance of the last tick)); zero after a reset (empty message) */
    BSK_T_UPEND = 26,   /* BSK_MAX_RW fields                                                           */
    BSK_T_SBR = 30,     /* |sigma_BR| of the att_guidance message the last FSW tick wrote (bsk_config.nav_lag) */
    BSK_NF_TAIL = 31,
};

typedef struct bsk_config {
    uint32_t abi_version; /* = BSK_ABI_VERSION */
    uint32_t struct_size; /* = sizeof(bsk_config) */

    /* integrator / schedule (reference: dynRate .1, fswRate 1.0 — leoPowerAttitudeEnvironment.py:185) */
    double dt;         /* RK4 step [s]                                                       */
    int32_t fsw_every; /* RK4 steps per FSW update (fswRate / dynRate)                       */
    int32_t gravity_model;
    int32_t sh_degree; /* used when gravity_model == BSK_GRAV_SH                             */
    int32_t n_rw;      /* 0..BSK_MAX_RW                                                      */
    uint32_t flags;
    int32_t max_length; /* episode length in env steps (leoPowerAttitudeEnvironment.py:25)   */
    /* 1 (default, reference order): mrpControlTask runs MRP_Feedback BEFORE attTrackingError
     * (AddModelToTask order, …Simulator.py:484-486), so the controller consumes the att_guidance
     * message of the PREVIOUS FSW tick: the wheel torque lags the guidance by one FSW period and
     * the first tick after a reset commands zero (empty message).  0: guidance and control on the
     * same tick (the order the module names suggest).                                           */
    int32_t fsw_lag;
    /* 1 (default, reference priorities): the FSW tasks are created with priorities 100 / 50 (…Simulator.py:383-386),
     * the dynamics tasks with the default (:101-103), and Basilisk runs higher priorities first at equal time: the
     * FSW tick at time k*fsw_every*dt executes BEFORE the dynamics task integrates to that time — on the navigation
     * and wheel-speed messages of one integrator step earlier — and its commands (wheel torque, thruster burst) are
     * latched when the dynamics task runs, i.e. act from that time on.  The tick at t = 0 finds messages nobody has
     * written yet (zeros); a tick that coincides with the end of an env step belongs to that step, with its mode
     * (ExecuteSimulation runs the tasks scheduled at its stop time); obs[0] is the att_guidance message as the last
     * FSW tick left it.  0: an FSW tick works on the state of its own time, belongs to the env step that starts
     * there, and obs[0] is the tracking error of the end-of-step state.                                          */
    int32_t nav_lag;

    /* gravity constants (leo_orbit.py:30; REQ_EARTH at …Simulator.py:146) */
    double mu;      /* m^3/s^2 */
    double req;     /* m       */
    double j2;      /* -        */
    double planet_rate; /* rad/s, planet-fixed frame rotation about inertial z (SH only)     */

    /* hub (…Simulator.py:245-250) */
    double inertia[9]; /* I_sc about B, body frame, row-major [kg m^2] */
    double mass;       /* kg */

    /* reaction wheels (actuatorPrimatives.py:7-63; 4-wheel pyramid BSK_OpNavDynamics.py:278-291) */
    double gs[BSK_MAX_RW][3]; /* spin axes, body frame, unit */
    double js[BSK_MAX_RW];    /* spin-axis inertia [kg m^2]   */
    double u_max;             /* motor torque saturation [N m] */
    double u_min;             /* motor torque dead-band [N m]  */
    double f_coulomb;         /* Coulomb friction torque [N m] */

    /* FSW (…Simulator.py:170-180, 407-449) */
    double K, P;            /* MRP_Feedback gains (Ki < 0: integral feedback off) */
    double sigma_R0N[3];    /* inertial3D reference                               */
    double ctrl_axes[9];    /* rwMotorTorque controlAxes_B, row-major             */

    /* env constants (leoPowerAttitudeEnvironment.py:36-42; …Simulator.py:641) */
    double wheel_limit;     /* rad/s */
    double power_max;       /* W h   */
    double reward_mult;
    double failure_penalty;
    double r_min;           /* |r| below this ends the episode ("orbit decayed") */

    /* power system (…Simulator.py:158-167) */
    double panel_normal[3]; /* nHat_B */
    double panel_area, panel_efficiency;
    double power_draw;       /* W, negative = sink */
    double storage_capacity; /* W s */
    double solar_flux;       /* W/m^2 at 1 AU */

    /* Sun (replaces spice_interface, …Simulator.py:219-225): position at t = 0 and velocity,
       inertial, Earth-centred; advanced linearly once per env step like the 180 s SPICE task */
    double sun_r0[3];
    double sun_v[3];
    double mu_sun;

    /* desaturation (actuatorPrimatives.py:66-161; …Simulator.py:183-190, 452-478) */
    int32_t n_thr;
    int32_t thr_max_counter;
    double thr_pos[BSK_MAX_THR][3];
    double thr_dir[BSK_MAX_THR][3];
    double thr_max_thrust;     /* N   (MOOG Monarc-1: 0.9)                                   */
    double thr_min_fire_time;  /* s   thrMomentumDumping.thrMinFireTime (…Simulator.py:190)   */
    double thr_min_on_time;    /* s   thruster MinOnTime (MOOG Monarc-1: 0.02)               */
    double hs_min;             /* N m s  thrMomentumManagement.hs_min (…Simulator.py:183)     */

    /* drag (…Simulator.py:147-148, 272-281) */
    double base_density, scale_height;
    int32_t n_facets;
    int32_t pad0_;
    double facet_area[8], facet_cd[8];
    double facet_normal[8][3], facet_pos[8][3];
} bsk_config;

typedef struct bsk_handle bsk_handle;

/* Fill `cfg` with the reference scenario's constants (…Simulator.py:127-191) for `n_rw` wheels
 * (3 = actuatorPrimatives.py triad, 4 = BSK_OpNavDynamics.py:278-291 pyramid, 0 = none). */
int bsk_default_config(bsk_config* cfg, int n_rw, int gravity_model);

/* Replaces LEOPowerAttitudeSimulator.__init__ (…Simulator.py:67-117) for n_envs spacecraft.
 * `stream` may be NULL (the library creates its own hipStream_t) or an existing hipStream_t. */
int bsk_create(const bsk_config* cfg, int n_envs, int device_id, void* stream, bsk_handle** out);
void bsk_destroy(bsk_handle* h);

/* Normalised spherical-harmonic coefficients Cbar/Sbar, index l*(l+1)/2+m, 0<=m<=l<=degree. */
int bsk_set_gravity_sh(bsk_handle* h, int degree, const double* cbar, const double* sbar);

/* Replaces set_ICs + hub/wheel/battery initialisation (…Simulator.py:119-193, 249-259, 303-305, 343)
 * and reset_init's IC replay (leoPowerAttitudeEnvironment.py:202-216).
 * mask: NULL = all envs, else uint8[n_envs] (non-zero = reset this env).
 * ic: host SoA [n_fields][n_envs] as in the field enums above; step counters are zeroed. */
int bsk_reset(bsk_handle* h, const uint8_t* mask, const double* ic);

/* Replaces run_sim(action) (…Simulator.py:535-644) + the env's reward/done logic
 * (leoPowerAttitudeEnvironment.py:98-127,161-170) for every env: mode switch, `substeps` RK4
 * steps with the FSW chain every fsw_every steps, observation/reward/done.  Asynchronous on
 * the handle's stream.  actions: int32[n_envs] in {0,1,2} (host pointer; copied H2D). */
int bsk_step(bsk_handle* h, const int32_t* actions, int substeps);
/* Same, actions already resident in device memory (no PCIe traffic on the step path). */
int bsk_step_device(bsk_handle* h, const int32_t* d_actions, int substeps);
/* Same, actions as int64 in device memory (what torch's argmax returns: no conversion kernel between policy and step);
 * the kernel reads the low word of each little-endian element. */
int bsk_step_device_i64(bsk_handle* h, const int64_t* d_actions, int substeps);

/* Open-loop rollout: `n_steps` env steps of `substeps` RK4 sub-steps each, enqueued by ONE call - at the bare level (point mass / J2, no
 * BSK_FLAG_POWER) as ONE launch with the spacecraft's state kept in registers across the steps; at the scenario levels and with the
 * harmonics, where an env step is milliseconds of arithmetic, as one step launch + one history-row launch per env step.  What the reference's own mains do - whole episodes under one constant action
 * (envs/leoPowerAttitudeEnvironment.py:218-231; ...Simulator.py:657-694: 360 steps of action 0) - and what evaluating a fixed action
 * sequence does, without a launch, a state round trip through memory and a host visit per env step.
 *   d_actions   int32[n_steps][n_envs] in DEVICE memory, or NULL: `constant_action` at every step
 *   d_obs_hist  f64[n_steps][5][n_envs], d_reward_hist f64[n_steps][n_envs], d_reason_hist u8[n_steps][n_envs] in DEVICE memory
 *               (each may be NULL): row t = what bsk_get_obs would have returned after step t of n_steps calls of bsk_step_device -
 *               where the device-side auto-reset restarted an env, its observation row is the NEW episode's first observation (as
 *               in the observation buffers), reward and reason are the finished step's.
 * Afterwards every buffer of the handle - state, counters, observation / reward / reason / done mask, terminal observations,
 * episode counts and statistics - holds, bit for bit, what `n_steps` calls of bsk_step_device with the same actions leave.
 * Per env step the fused launch reads 4 bytes (none for a constant action) and writes 49.  Asynchronous on the handle's stream (a
 * constant action at the unfused levels goes through the handle's own action buffer, the one bsk_step stages host actions in). */
int bsk_step_n(bsk_handle* h, const int32_t* d_actions, int32_t constant_action, int substeps, int n_steps,
               double* d_obs_hist, double* d_reward_hist, uint8_t* d_reason_hist);

/* Host copies of the last step's outputs (synchronises the stream).  Any pointer may be NULL.
 * obs: f64[5][n_envs] = [|sigma_BR|, |omega_BN_B|, |Omega|/wheel_limit, charge/3600/power_max,
 * shadow factor] (…Simulator.py:636-638 + leoPowerAttitudeEnvironment.py:107-108);
 * reward f64[n_envs]; done uint8[n_envs]; done_reason uint8[n_envs] bit-or of BSK_DONE_*. */
#define BSK_DONE_LENGTH 0x1
#define BSK_DONE_WHEELS 0x2
#define BSK_DONE_BATTERY 0x4
#define BSK_DONE_ORBIT 0x8
int bsk_get_obs(bsk_handle* h, double* obs, double* reward, uint8_t* done, uint8_t* done_reason);

/* bsk_get_obs + bsk_get_state behind ONE stream synchronisation (the single-env mirror reads both after every step:
 * …Simulator.py:598-619 pulls the same quantities from the message logs).  Any pointer may be NULL. */
int bsk_get_obs_state(bsk_handle* h, double* obs, double* reward, uint8_t* done_reason, double* state);

/* The same read-back with the observation as the row-major f64[n_envs][5] block the kernel writes under BSK_FLAG_OBS_ROWMAJOR (one
 * contiguous copy; the layout a VecEnv hands out as (N, 5, 1): no transposition on the host).  BSK_EINVAL without the flag.  Any
 * pointer may be NULL.  Synchronises. */
int bsk_get_obs_rowmajor(bsk_handle* h, double* obs_n5, double* reward, uint8_t* done_reason);
/* Device pointers of the output buffers (for the RCCL gather / zero-copy hand-off).
 * obs stride (envs per field) is returned in *stride; done_mask is uint64[ceil(n/64)]. */
int bsk_get_obs_device(bsk_handle* h, double** d_obs, double** d_reward, uint64_t** d_done_mask,
                       uint8_t** d_done_reason, int64_t* stride);

/* Device-resident stepping ("so training loops stay on-GPU"; replaces the host read-back of
 * …Simulator.py:598-619 for a policy that lives on the same GPU).
 * bsk_get_stream: the hipStream_t the handle launches on, so that a caller can order its own work against the step
 * kernel with events / stream waits instead of a host synchronisation (or hand its own stream to bsk_create).
 * bsk_get_terminal_obs_device: device pointers of the terminal observations f64[5][stride] and the per-env
 * finished-episode counts int32[stride] of the device-side auto-reset (NULL until a pool is staged).
 * bsk_get_state_device: the state slab f64[n_fields][*stride] itself (read-only for the caller between steps).  ITS stride need not be
 * the observation buffers' (every other [stride] on this page is bsk_get_obs_device's: n_envs rounded up to 256): for batches of
 * 65 536 ... 98 304 spacecraft the slab's rows carry 256 B of padding each.  That is an EMPIRICAL constant - the K = 1 launch measured
 * 2.6 % shorter at 65 536 and 0.5 % at 98 304 with the rows an odd multiple of 256 B apart, nothing either way at 32 768 / 131 072,
 * slightly longer at 4 Mi (profiles/r06/stride_pad.txt); the mechanism is not established - so always take the stride from here. */
/* bsk_get_episode_device (BSK_FLAG_EPISODE_STATS / BSK_FLAG_OBS_ROWMAJOR; pointers are NULL where the flag is off):
 *   ep_return   f64[stride]  return of the running episode, this step's reward included; 0 where a device-side reset fired
 *   term_return f64[stride], term_len int32[stride]: 'r' and 'l' of info['episode'] for the envs whose done byte is set at this
 *               step ('l' = env steps taken before this one, as the reference counts: ...Environment.py:133)
 *   done        u8[stride]   0 / 1 (a torch.bool view needs no kernel)
 *   obs_rowmajor f64[n_envs][5]
 * Every reset leaves the NEW episode's first observation [|sigma_BN|, |omega|, |Omega|/limit, charge/3600/power_max, 1] in the
 * observation buffers of the envs it restarts, so that a device-resident loop never has to compute or upload a reset observation.
 *  - The step kernel's own auto-reset (BSK_FLAG_AUTO_RESET) KEEPS that step's reward, reason and done byte - the finished episode's
 *    last transition is what the step reports - and zeroes only ep_return (the new episode's running return).
 *  - The explicit entry points (bsk_reset, bsk_reset_from_pool, bsk_reset_from_pool_device) ALSO zero reward, reason, done and
 *    ep_return of the envs they restart: a consumer that reads those buffers on the device must have consumed the last step's
 *    values - or be ordered before the reset on the handle's stream - before it calls a masked reset entry point, or the
 *    terminal reward and penalty of the restarted envs are gone (bsk_get_batch_stats* still report the last step's sums: the
 *    snapshot is taken before the zeroing).
 * HIP graphs: the device-resident entry points (bsk_step_device*, bsk_reset_from_pool_device, bsk_get_batch_stats_device) may be
 * captured.  The library notices the capture and from then on evaluates nothing at enqueue time for that handle (batch scalars
 * are re-formed on every request, the bare levels read the battery charge in every launch), so replays stay correct after a later
 * bsk_set_state / bsk_set_ic_pool / bsk_reset; launch GEOMETRY (substeps, kernel form) is what was captured. */
int bsk_get_episode_device(bsk_handle* h, double** d_ep_return, double** d_term_return, int32_t** d_term_len, uint8_t** d_done,
                           double** d_obs_rowmajor);
int bsk_get_stream(bsk_handle* h, void** stream);
int bsk_get_terminal_obs_device(bsk_handle* h, double** d_term_obs, int32_t** d_episodes);
int bsk_get_state_device(bsk_handle* h, double** d_state, int64_t* stride);

/* Batch scalars of the LAST STEP: sum of its rewards and number of done envs, formed in a fixed order (bitwise
 * reproducible: per 64 envs an xor butterfly, the wave sums w = t mod 256 added in ascending order by thread t, a halving tree
 * over the 256 partials) from the reward buffer and the per-wave done ballots by two small launches of their own (a
 * multi-workgroup first level: one sum per 64 envs; a single workgroup joins them) - once, when first asked for, or just before
 * a reset entry point zeroes the restarted envs' rewards (up to 65 536 spacecraft the join is one wave making the same additions).
 * A step does NOT produce them unless bsk_set_step_stats says so: its epilogue carries the done ballot (one 64-bit mask per wave) and
 * no reward reduction.  After a reset entry point the scalars stay the last STEP's until the next step launch - also on a handle whose
 * launches replay from a HIP graph: the reset seals the snapshot on the device (env 0's counter word and episode number, which every
 * step launch changes) and a later request leaves a sealed snapshot alone.  (Synchronises.) */
int bsk_get_batch_stats(bsk_handle* h, double* reward_sum, int64_t* n_done);
/* The same two scalars left ON the device as f64[2] = {sum of rewards, number of done envs}, enqueued on the handle's stream
 * without synchronising: the operand of the one all-reduce a sharded batch needs. */
int bsk_get_batch_stats_device(bsk_handle* h, double** d_stats2);
/* For consumers that ask for the batch scalars after EVERY step (the per-step all-reduce of a sharded batch, a monitor): on != 0
 * makes every step launch (bsk_step*, not bsk_step_n) form the first level - the sum per 64 envs, same butterfly - in its own
 * epilogue, behind its stores, so that a request costs the join launch alone (batches of up to 2 Mi spacecraft; larger ones keep the
 * two-level form, which is as fast there).  Same bits either way.  Default off: a step that nobody asks about pays nothing.  (Takes effect with the next launch; a handle whose launches have been captured into a HIP graph keeps
 * the two-level form - replays step without the host's knowledge.) */
int bsk_set_step_stats(bsk_handle* h, int on);

/* Full state read-back / injection, host SoA [n_fields][n_envs] (parity tests, reset_init). */
int bsk_n_fields(const bsk_handle* h);
int bsk_get_state(bsk_handle* h, double* state);
int bsk_set_state(bsk_handle* h, const double* state);
/* per-env counters: env steps and RK4 ticks since reset, int32[n_envs] each (may be NULL) */
int bsk_get_counters(bsk_handle* h, int32_t* steps, int32_t* ticks);
/* Restore the per-env counters (checkpoint / resume together with bsk_set_state): steps in 0..2^20-1,
 * ticks >= 0, int32[n_envs] each, both required.  The FSW phase follows as ticks mod fsw_every. */
int bsk_set_counters(bsk_handle* h, const int32_t* steps, const int32_t* ticks);

/* Device-side auto-reset (BSK_FLAG_AUTO_RESET; reference reset semantics,
 * envs/leoPowerAttitudeEnvironment.py:172-191, without the host round trip).  Stage a pool of
 * initial conditions, host SoA [n_fields][n_pool].  When an env finishes, the step kernel itself
 * reloads it from pool slot  ((env_base + env) * 2654435761 + episode * 40503 + 12345) mod 2^32 mod n_pool
 * (episode = that env's count of finished episodes; env_base: bsk_set_env_base, 0 unless the batch is sharded), zeroes its counters, writes the NEW episode's
 * initial observation [|sigma_BN|, |omega|, |Omega|/limit, charge/3600/power_max, 1] to obs and keeps
 * the finished episode's last observation in the terminal-observation buffer. */
int bsk_set_ic_pool(bsk_handle* h, int n_pool, const double* ic_pool);
/* On-device IC sampler: fill the pool with `n_pool` initial conditions drawn on the GPU with the
 * reference's distributions (set_ICs, …Simulator.py:119-193; leo_orbit.py:25-40; sc_attitudes.py:3-13):
 * a = 6 871 km, e~U[0,.05), i~U[-90,90) deg, Omega, omega, f~U[0,360) deg -> elem2rv; sigma~U[0,1)^3;
 * omega~U(+-1e-5)^3 rad/s; L_ext = 2e-4 N(0,1)^3 N m; wheel speeds U(-800,800) RPM; charge U(8,20) W h.
 * Random numbers: Philox4x32-10, key = seed, counter = (slot, draw index): slot k is reproducible
 * on its own.  Needs BSK_FLAG_AUTO_RESET.  bsk_reset_from_pool then (re)starts every env (mask NULL)
 * or the masked envs from the pool with the slot rule above, without any host data. */
int bsk_sample_ic_pool(bsk_handle* h, int n_pool, uint64_t seed);
int bsk_reset_from_pool(bsk_handle* h, const uint8_t* mask);
/* Same with the mask (or NULL) in DEVICE memory, asynchronous on the handle's stream: no host data, no copy, no synchronisation. */
int bsk_reset_from_pool_device(bsk_handle* h, const uint8_t* d_mask);
/* Host copy of the staged pool, SoA [n_fields][n_pool] (n_pool as staged; the caller sizes the buffer):
 * lets the host replay a device-side reset (reset_init, leoPowerAttitudeEnvironment.py:202-216). */
int bsk_get_ic_pool(bsk_handle* h, double* ic_pool);
/* terminal observations f64[5][n_envs] (valid for envs whose done flag is set) and per-env
 * finished-episode counts int32[n_envs]; either pointer may be NULL.  Synchronises. */
int bsk_get_terminal_obs(bsk_handle* h, double* term_obs, int32_t* episodes);

/* Sharded batches (one handle per GPU): the GLOBAL index of this handle's env 0.  The device-side
 * reset's slot rule hashes env_base + local index, so a batch split over several handles restarts its envs from
 * exactly the pool slots the unsplit batch would use.  Default 0. */
int bsk_set_env_base(bsk_handle* h, int64_t env_base);

/* Epoch offset [s] added to every spacecraft's own clock (ticks * dt) when the Sun position
 * sun_r0 + sun_v * t is evaluated at the start of an env step (default 0). */
int bsk_set_sim_time(bsk_handle* h, double t_seconds);

/* Synchronises the handle's stream.  Like every synchronising entry point (bsk_get_obs*, bsk_get_state, bsk_get_batch_stats,
 * bsk_get_terminal_obs) it then checks the handle's device error word and returns BSK_EHIP when a kernel raised it: the
 * three-wave form's barrier-free exchange gives up after 2^20 polls instead of hanging, and says so here. */
int bsk_sync(bsk_handle* h);

/* Process-wide counts of the host <-> device copies and stream synchronisations this library has issued (tests assert that the
 * device-resident entry points issue none).  Either pointer may be NULL. */
int bsk_debug_counters(int64_t* n_copies, int64_t* n_syncs);
/* Probe builds only (csrc/bsk_probes.hpp; all zeros from the product library): the word every wavefront of the last launch left
 * in the handle's debug buffer, uint64[ceil(n_envs / 64)].  Synchronises. */
int bsk_debug_words(bsk_handle* h, uint64_t* words);

/* Per-launch timing of the step kernel with hipEvents recorded on the handle's stream around
 * each launch.  begin() arms it (capacity launches); end() synchronises and reports the mean
 * kernel duration in milliseconds over the launches seen since begin(). */
int bsk_profile_begin(bsk_handle* h, int capacity);
/* stride 1 (default): every launch is stamped.  stride > 1: a pair of launches is stamped every
 * `stride` launches and only the second of the pair is counted (the first absorbs the transition
 * from un-stamped back-to-back launches).  Stamping costs ~5 us of launch throughput per stamped
 * launch on MI355X, so a timed region samples instead of stamping every launch. */
int bsk_profile_set_stride(bsk_handle* h, int stride);
int bsk_profile_end(bsk_handle* h, double* mean_kernel_ms, int* n_launches);
/* Same, also copying the individual kernel durations [ms] of the first min(cap, n) counted launches
 * into samples_ms (so that a caller can report median / min / max beside the mean). */
int bsk_profile_end_samples(bsk_handle* h, double* mean_kernel_ms, int* n_launches, float* samples_ms, int cap);

/* Measurement aid: the fp64 FMA rate device `device_id` sustains with `waves_per_simd` waves of independent v_fma_f64 chains per
 * SIMD (median of `repeats` timed launches of ~2-4 ms after three warm-up launches): TFLOP/s of the whole device and
 * nanoseconds per FMA wave-instruction and SIMD.  bench.py prints it beside its fp64 rooflines, whose `peak` stays the nominal
 * 78.6 TFLOP/s. */
int bsk_calibrate_fp64(int device_id, int waves_per_simd, int repeats, double* tflops, double* ns_per_fma_per_simd);

/* Kernel resource facts for DESIGN.md / bench: name of the kernel variant the handle's LAST launch ran, its VGPR count, static
 * LDS bytes and the launch geometry.  The variant is chosen per launch: batches of <= 16 384 spacecraft run launches of >= 16
 * sub-steps in a wave-split form of the same arithmetic (bit-identical results: "...,pair" at the power level - a dynamics and
 * an FSW + environment wave per 64 spacecraft, 128-thread workgroups - and "...,tri" at the full-scenario level - a
 * translational, a rotational and the FSW + environment wave, 192-thread workgroups), everything else the
 * single-wave form.  BSKGPU_PAIR / BSKGPU_TRI = 0 | 1 in the environment of bsk_create switch a form off / on for every launch. */
int bsk_kernel_info(bsk_handle* h, char* name, int name_cap, int* vgprs, int* lds_bytes,
                    int* block, int* grid);

const char* bsk_last_error(void);
const char* bsk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BSKGPU_H */
