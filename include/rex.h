/*
 * rex.h -- C-ABI of the MI355X-native batched domain-randomised locomotion simulator.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json:north_star:
 * the per-instance forward-dynamics step() and the reset()-time xi sampling of the
 * random_envs environments, batched one environment per GPU lane.
 *
 * The reference (gabrieletiboni/random-envs) has no FFI of its own: its native boundary is
 * mujoco-py's Cython binding (MjSim.step / forward / reset / get_state / set_state and raw
 * views into MjModel / MjData).  Each entry point below cites the reference interface it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no C++ / torch types; every pointer marked [dev] is a caller-owned DEVICE
 *     pointer (e.g. torch.Tensor.data_ptr()), every pointer marked [host] is host memory.
 *   - all calls return 0 on success or a negative rex_status; rex_last_error() gives the
 *     message (thread-local).  No C++ exception crosses this boundary.  (The reference raises
 *     Python exceptions, or drops into pdb on MujocoException: jinja_mujoco_env.py:153-164.)
 *   - kernels are enqueued asynchronously on the caller's HIP stream (`stream`, a hipStream_t
 *     passed as void*, NULL = default stream).  No allocation or synchronisation in
 *     rex_step / rex_reset.
 *   - internal state is SoA [field][env]; I/O buffers are SoA too: obs is [obs_dim][batch],
 *     action is [act_dim][batch], xi is [task_dim][batch] (a torch [batch, dim] view is the
 *     zero-copy transpose).  reward is float[batch], done / truncated are uint8[batch].
 *   - a handle is bound to one device; calls on one handle are not re-entrant.  Every entry point that enqueues work, copies or
 *     synchronises makes the handle's device the calling thread's current device first, so one thread may drive one handle per GPU.
 *   - environment: REX_LANES / REX_PAIR / REX_ROLLED / REX_HUM_PAIR / REX_HUM_FUSED_RESET / REX_FUSED_DERIVE (launch shape) and
 *     REX_FAST / REX_LS_MAX / REX_LS_FREE / REX_WARM / REX_CORR (solver schedule) are A/B and test knobs: rex_create honours them only
 *     with REX_ALLOW_TUNING=1 and otherwise REFUSES to create a handle while one is set (REX_ERR_STATE).  Nothing else is read from
 *     the environment.
 */
#ifndef REX_H_
#define REX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rex_env rex_t;

/* env_kind: one per kinematic chain of the reference. */
enum rex_env_kind {
  REX_CARTPOLE    = 0, /* random_envs/random_cartpole.py:19            */
  REX_HOPPER      = 1, /* random_envs/jinja/random_hopper.py:16        */
  REX_HALFCHEETAH = 2, /* random_envs/jinja/random_half_cheetah.py:17  */
  REX_WALKER2D    = 3, /* random_envs/jinja/random_walker2d.py:19      */
  REX_HUMANOID    = 4  /* random_envs/jinja/random_humanoid.py:27      */
};

/* dr_type of RandomEnv.set_dr_distribution (random_envs/random_env.py:72-90). */
enum rex_dr_type {
  REX_DR_NONE         = 0,
  REX_DR_UNIFORM      = 1, /* params = [lo0,hi0,lo1,hi1,...]     random_env.py:102-107 */
  REX_DR_TRUNCNORM    = 2, /* params = [mean0,std0,...]          random_env.py:109-114 */
  REX_DR_GAUSSIAN     = 3, /* params = [mean0,std0,...]          random_env.py:116-121 */
  REX_DR_FULLGAUSSIAN = 4  /* params = [mean(d), chol(cov)(d*d, row-major lower), lo(d), hi(d)]
                              random_env.py:123-127,192-198,205-220 */
};

enum rex_status {
  REX_OK            =  0,
  REX_ERR_ARG       = -1, /* bad argument (unknown env kind / dr type / sizes)   */
  REX_ERR_HIP       = -2, /* a HIP runtime call failed                           */
  REX_ERR_STATE     = -3, /* call sequence error (e.g. sampling before set_dr)   */
  REX_ERR_UNSUPPORTED = -4
};

/* Static description of an env kind (dims of SURVEY.md section 8 table). */
typedef struct rex_dims {
  int nq, nv, act_dim, obs_dim, task_dim;
  int frame_skip;
  int max_episode_steps; /* 500 for all 13 ids, e.g. random_hopper.py:155-166 */
  int discrete_action;   /* 1 for CartPole (Discrete(2), random_cartpole.py:96) */
  float dt;              /* model timestep * frame_skip, jinja_mujoco_env.py:166-168 */
  float act_low, act_high; /* actuator_ctrlrange, jinja_mujoco_env.py:99-103 */
  int n_info;            /* per-term reward rows rex_set_info_buffer exposes: 2 for the planar chains
                            (reward_run, reward_ctrl: random_half_cheetah.py:105-110), 4 for the humanoid
                            (reward_linvel, reward_quadctrl, reward_alive, reward_impact: random_humanoid.py:182-187), 0 CartPole */
  int n_aux;             /* rows of sim data that outlive a step besides (qpos, qvel): the humanoid's data.xipos[:,0]
                            (14 bodies) left by the last mj_forward, which mass_center() reads BEFORE the next
                            do_simulation (random_humanoid.py:22-25,162); 0 for the other chains */
} rex_dims;

/* variant: 0 = regular id, 1 = the "Unmodeled" id of the same chain (e.g. random_hopper_unmodeled.py:16-43): a
 * prefix of xi is frozen at 0.8x nominal and leaves the task vector, so task_dim shrinks (3 / 5 / 9). */
int rex_get_dims(int env_kind, int variant, rex_dims* out);

/* Replaces MujocoEnv.__init__ / build_model (jinja_mujoco_env.py:43-97): load_model_from_xml +
 * MjSim for `batch` environments at once.  `env_offset` is the global index of this handle's
 * first env: RNG streams are keyed by the GLOBAL env index so results do not depend on how a
 * batch is sharded over GPUs.  `variant` 0 = regular id, 1 = "Unmodeled" id. */
int rex_create(int env_kind, int variant, int64_t batch, int device_id, uint64_t seed,
               int64_t env_offset, rex_t** out);
int rex_destroy(rex_t* h);

/* RandomEnv.set_dr_distribution (random_env.py:72-127).  `params` [host] layout per rex_dr_type;
 * `lower_bounds` [host, task_dim] = get_task_lower_bound(i) (e.g. random_hopper.py:60-72), used by
 * the truncnorm resampling rule (random_env.py:153-171). May be NULL for the other types. */
int rex_set_dr(rex_t* h, int dr_type, const float* params, int n_params, const float* lower_bounds);
/* RandomEnv.set_dr_training (random_env.py:41-46). */
int rex_set_dr_training(rex_t* h, int flag);
/* set_endless (random_env.py:51-60); `noisy` ctor kwarg + noise_level (random_hopper.py:17-28);
 * noise_var < 0 keeps the env's reference default. */
int rex_set_flags(rex_t* h, int endless, int noisy, float noise_var);
/* auto-reset of finished lanes inside rex_step (SB3 VecEnv convention used by the reference's
 * downstream, README.md:68); time-limit truncation at max_episode_steps (gym TimeLimit). */
int rex_set_autoreset(rex_t* h, int autoreset, int time_limit);
/* seed(): MujocoEnv.seed (jinja_mujoco_env.py:109-111). Re-keys the Philox streams. */
int rex_seed(rex_t* h, uint64_t seed);

/* MujocoEnv.reset + reset_model (jinja_mujoco_env.py:141-144, random_hopper.py:112-120):
 * lanes with mask[i]!=0 (all lanes if mask==NULL) get qpos0/qvel0 + init noise, a fresh xi when
 * dr_training is on, and their observation written to obs_out [dev, obs_dim*batch] (may be NULL). */
int rex_reset(rex_t* h, const uint8_t* mask, float* obs_out, void* stream);

/* step(): RandomHopperEnv.step etc. (random_hopper.py:83-98) = do_simulation
 * (jinja_mujoco_env.py:170-173: ctrl <- a; frame_skip x sim.step()) + reward + done + _get_obs.
 * action [dev]: float[act_dim*batch] (CartPole: int32[batch], values 0/1).
 * Optional outputs (NULL to skip): truncated_out (TimeLimit.truncated), terminal_obs_out
 * (observation before auto-reset). */
int rex_step(rex_t* h, const void* action, float* obs_out, float* reward_out, uint8_t* done_out,
             uint8_t* truncated_out, float* terminal_obs_out, void* stream);

/* Offline replay of logged transitions under candidate xi (get_full_mjstate + set_sim_state + step: random_hopper.py:128-152,
 * random_half_cheetah.py:136-158, random_walker2d.py:161-185, random_humanoid.py:244-270, and the Unmodeled task files'
 * copies of them): one env.step per lane from the CALLER's qpos [dev, nq*batch], qvel [dev, nv*batch], xi [dev, task_dim*batch]
 * (the reduced task for the Unmodeled ids), action [dev, act_dim*batch] into obs_out / reward_out / done_out.  Nothing of the
 * handle is read back or changed: state, task, step / episode counters, diagnostic counters, RNG position.  Hopper and
 * half-cheetah: ONE launch.  Walker2d: its per-env geometry follows the xi lengths (set_task rebuilds the model,
 * random_walker2d.py:106-113), so a derive launch into replay scratch precedes the step launch.  Humanoid: the forward
 * launch of set_state (data.xipos for mass_center(), jinja_mujoco_env.py:154) precedes it.  Unmodeled ids: one scatter
 * launch places the reduced task over the handle's frozen rows in a scratch copy of the full xi block.  The scratch is
 * allocated by the first such call (which therefore synchronises the device) and belongs to the handle: issue the replays of one
 * handle on one stream at a time.  RandomCartPole: REX_ERR_UNSUPPORTED (the reference has no such helpers for it). */
int rex_replay(rex_t* h, const float* qpos, const float* qvel, const float* xi, const float* action,
               float* obs_out, float* reward_out, uint8_t* done_out, void* stream);

/* get_sim_state / set_sim_state (random_hopper.py:148-152), MujocoEnv.set_state
 * (jinja_mujoco_env.py:146-154), state_vector (:231-235). qpos [dev, nq*batch], qvel [dev, nv*batch].
 * CartPole: qpos = (x, theta), qvel = (x_dot, theta_dot). */
int rex_get_state(rex_t* h, float* qpos, float* qvel, void* stream);
int rex_set_state(rex_t* h, const float* qpos, const float* qvel, void* stream);
/* get_task / set_task (random_hopper.py:75-80 etc.). xi [dev, task_dim*batch]. */
int rex_get_task(rex_t* h, float* xi, void* stream);
int rex_set_task(rex_t* h, const float* xi, void* stream);
/* set_random_task (random_env.py:37-39) for the masked lanes, without touching their state. */
int rex_set_random_task(rex_t* h, const uint8_t* mask, void* stream);
/* current observation of every lane (_get_obs, random_hopper.py:100-110), without noise. */
int rex_get_obs(rex_t* h, float* obs_out, void* stream);

/* The rest of the sim state get_sim_state returns (random_hopper.py:148-152: MjSimState incl. time) and the
 * TimeLimit wrapper's elapsed-step count: per-lane step counter t [dev, int32 batch], episode index
 * [dev, uint32 batch] (together they key the Philox streams, so restoring them makes a resume RNG-exact and
 * time-limit-exact) and the done flags [dev, uint8 batch]. */
int rex_get_counters_state(rex_t* h, int32_t* t, uint32_t* episode, uint8_t* done, void* stream);
int rex_set_counters_state(rex_t* h, const int32_t* t, const uint32_t* episode, const uint8_t* done, void* stream);
/* The remaining MjData a bit-exact resume needs (rex_dims.n_aux rows, [dev, n_aux*batch]); rex_set_state refreshes it
 * with sim.forward() like the reference's set_state, so restore it AFTER the state. */
int rex_get_aux(rex_t* h, float* aux, void* stream);
int rex_set_aux(rex_t* h, const float* aux, void* stream);
/* RandomEnv.sample_task / sample_tasks (random_env.py:145-203) WITHOUT set_task: one xi per lane into
 * xi_out [dev, task_dim*batch]; `draw_index` selects the draw (stream family separate from the reset streams);
 * neither the current task nor the episode counters change. */
int rex_sample_task(rex_t* h, float* xi_out, uint64_t draw_index, void* stream);
/* The `info` dict of step(): per-term rewards (random_half_cheetah.py:110 reward_run / reward_ctrl,
 * random_humanoid.py:182-187 reward_linvel / reward_quadctrl / reward_alive / reward_impact) written by every
 * later rex_step into info [dev, n_info*batch] (rex_dims.n_info rows; NULL switches it off). */
int rex_set_info_buffer(rex_t* h, float* info);
/* One lane's (qpos, qvel, xi) to HOST memory for an external viewer: the data MujocoEnv.render reads from the
 * sim (jinja_mujoco_env.py:175-226; CartPole: random_cartpole.py:231-283). Synchronises the device. */
int rex_export_lane(rex_t* h, int64_t lane, float* qpos /*[host, nq]*/, float* qvel /*[host, nv]*/, float* xi /*[host, task_dim]*/);

/* number of env-steps executed by this handle (host counter; the only quantity the multi-GPU
 * path reduces across ranks). */
int64_t rex_step_count(const rex_t* h);
/* device-side diagnostics accumulated since creation: [0] lanes that went non-finite,
 * [1] gaussian-DR draws that failed the reference's 3-attempt rule (random_env.py:173-190),
 * [2] constraint solves that hit the iteration cap. Copies 4 int64 to `out` [host]; synchronises. */
int rex_get_counters(rex_t* h, int64_t* out);

/* the launch shape rex_create picked for this handle from its batch and the GPU's SIMD count (rex_set_launch_shape and,
 * under REX_ALLOW_TUNING=1, the REX_LANES / REX_PAIR / REX_ROLLED / REX_HUM_PAIR knobs override): out[0] lanes per workgroup of the step launch (two lanes per env: 64; 32 / 16 for a
 * walker2d / half-cheetah batch of 8 .. 16 envs per SIMD, where narrower waves still all get a SIMD; one lane
 * per env: 32, 64 past 32 768 envs), out[1] 1 = the planar step runs
 * two lanes per env, out[2] 1 = hopper step on the 256-register kernel with the rolled general solver (two waves per SIMD), out[3] 1 = the
 * humanoid step runs two lanes per env.  No reference counterpart (the reference steps one MjSim on one core); bench.py names the launched
 * kernel from it.  Writes 4 int32 to `out` [host]. */
int rex_get_launch_shape(const rex_t* h, int32_t* out);
/* Pins the launch shape of a handle (same four int32 as rex_get_launch_shape, [host]; -1 keeps a field): every shape runs the same solver to
 * the same minimiser, but which solver instantiation a wave enters depends on the shape, so two runs agree bit for bit only under the same
 * shape.  sharding.shard_strong pins every shard to the shape the GLOBAL batch would get on one GPU, which makes an index-sharded run reproduce the
 * single-GPU trajectories exactly.  Pure host bookkeeping (the shape is read at launch time); REX_ERR_ARG for a shape the env kind has no kernel
 * for.  No reference counterpart. */
int rex_set_launch_shape(rex_t* h, const int32_t* shape);

/* duration in ms of the sampled rex_step kernel launches since the last enable / read (at most the last 8192), measured
 * with HIP events on the launch stream; returns the number of samples written.  The two event packets of a bracketed launch
 * cost about 8 us of stream time, so throughput runs sample every n-th launch.  rex_enable_timing(1) creates
 * the event pool (the only allocation of the timing path: rex_step itself never allocates). */
int rex_enable_timing(rex_t* h, int every);   /* 0 = off, n >= 1 = bracket every n-th rex_step launch */
int rex_read_timing(rex_t* h, float* ms_out, int max_n);

const char* rex_last_error(void);
const char* rex_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REX_H_ */
