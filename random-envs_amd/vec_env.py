"""VecRandomEnv: the reference's RandomEnv / MujocoEnv surface for a batch of environments.

Mirrors, method for method, ``random_envs/random_env.py`` (DR state: set_dr_distribution,
get_dr_distribution, set_dr_training, sample_task(s), set_random_task, search bounds,
denormalize_parameters, load_dr_distribution_from_file) and the gym protocol of the task files
(reset, step -> (obs, reward, done, info), get_task, set_task, seed).  All tensors are
torch ROCm tensors; obs is ``[batch, obs_dim]`` (a zero-copy transposed view of the kernels'
SoA ``[obs_dim][batch]`` buffer), reward ``[batch]`` f32, done ``[batch]`` bool.
"""
import ctypes

import numpy as np

from . import _native
from .dr import DRConfig
from .specs import MAX_EPISODE_STEPS, SPECS, UNMODELED_SPECS


class _Box:
    """Stand-in with the attributes callers read from ``gym.spaces.Box`` -- used only when neither gymnasium nor gym is
    importable (this image); with either installed the real classes are built (:func:`make_spaces`)."""
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0); hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)


class _Discrete:
    def __init__(self, n):
        self.n = n
        self.shape, self.dtype = (), np.dtype(np.int64)

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))


def spaces_module():
    """``gymnasium.spaces`` or ``gym.spaces``, whichever is importable (gymnasium first: what current stable-baselines3
    checks ``isinstance(space, spaces.Box)`` against), else None."""
    for name in ("gymnasium", "gym"):
        try:
            return __import__(name + ".spaces", fromlist=["spaces"])
        except Exception:
            continue
    return None


def make_spaces(dims):
    """(observation_space, action_space) of one env as the reference builds them: action bounds from
    ``actuator_ctrlrange`` (jinja_mujoco_env.py:99-103; Discrete(2) for the cart-pole, random_cartpole.py:96), observation
    ``Box(-inf, inf)`` over the observation vector (jinja_mujoco_env.py:23-36) -- float32, the dtype the batched env
    returns.  Real gymnasium / gym spaces when one of them is importable, the duck-typed stand-ins otherwise."""
    sp = spaces_module()
    obs_shape, act_shape = (int(dims.obs_dim),), (int(dims.act_dim),)
    if sp is not None:
        obs = sp.Box(low=-np.inf, high=np.inf, shape=obs_shape, dtype=np.float32)
        act = sp.Discrete(2) if dims.discrete_action else sp.Box(low=np.float32(dims.act_low), high=np.float32(dims.act_high),
                                                                 shape=act_shape, dtype=np.float32)
        return obs, act
    return (_Box(-np.inf, np.inf, obs_shape),
            _Discrete(2) if dims.discrete_action else _Box(dims.act_low, dims.act_high, act_shape))


class VecRandomEnv(DRConfig):
    def __init__(self, kind, batch=1, device=0, seed=0, env_offset=0, noisy=False, env_id=None,
                 autoreset=True, time_limit=True, unmodeled=False):
        import torch
        self._torch = torch
        self.unmodeled = bool(unmodeled)
        if self.unmodeled and noisy:
            raise TypeError("the Unmodeled ids take no `noisy` argument (random_hopper_unmodeled.py:17)")
        self.spec = UNMODELED_SPECS[kind] if self.unmodeled else SPECS[kind]
        self.kind, self.env_id = kind, env_id
        self.batch, self.num_envs = int(batch), int(batch)
        self.device = torch.device("cuda", device)
        L = _native.lib()   # raises loudly when the HIP library is missing
        dims = _native.RexDims()
        _native.check(L.rex_get_dims(_native.ENV_KINDS[kind], int(self.unmodeled), ctypes.byref(dims)))
        self.dims = dims
        self.task_dim = dims.task_dim
        self._h = ctypes.c_void_p()
        _native.check(L.rex_create(_native.ENV_KINDS[kind], int(self.unmodeled), self.batch, device, seed, env_offset,
                                     ctypes.byref(self._h)))
        self._L = L
        DRConfig.__init__(self, self.spec)   # RandomEnv.__init__ state + per-env tables
        self.noisy = bool(noisy)
        self.max_episode_steps = MAX_EPISODE_STEPS
        self.autoreset, self.time_limit = bool(autoreset), bool(time_limit)
        self.observation_space, self.action_space = make_spaces(dims)       # gymnasium / gym spaces when importable
        f32 = dict(dtype=torch.float32, device=self.device)
        B = self.batch
        self._obs = torch.zeros(dims.obs_dim, B, **f32)
        self._term_obs = torch.zeros(dims.obs_dim, B, **f32)
        self._reward = torch.zeros(B, **f32)
        self._done = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self._trunc = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self._act = torch.zeros(dims.act_dim, B, dtype=torch.int32 if dims.discrete_action else torch.float32,
                                device=self.device)
        self._info = torch.zeros(max(dims.n_info, 1), B, **f32) if dims.n_info else None
        if self._info is not None:
            _native.check(L.rex_set_info_buffer(self._h, ctypes.c_void_p(self._info.data_ptr())))
        self._draws = 0
        self._push_flags()

    @property
    def dt(self):                             # jinja_mujoco_env.py:166-168: model.opt.timestep * frame_skip
        return float(self.dims.dt)

    #: names of the per-term reward rows returned in ``info`` (random_half_cheetah.py:110, random_humanoid.py:182-187)
    INFO_TERMS = {"hopper": ("reward_run", "reward_ctrl"), "walker2d": ("reward_run", "reward_ctrl"),
                  "halfcheetah": ("reward_run", "reward_ctrl"),
                  "humanoid": ("reward_linvel", "reward_quadctrl", "reward_alive", "reward_impact"), "cartpole": ()}

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _push_flags(self):
        _native.check(self._L.rex_set_flags(self._h, int(self.endless), int(self.noisy), float(self.noise_level)))
        _native.check(self._L.rex_set_autoreset(self._h, int(self.autoreset), int(self.time_limit)))
        _native.check(self._L.rex_set_dr_training(self._h, int(self.dr_training)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.rex_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ RandomEnv surface (device side)
    def _push_dr(self):
        d = self.task_dim
        lower = np.array(self.spec.lower_bounds, dtype=np.float32)
        if self.sampling == 'uniform':
            p = np.stack([self.min_task, self.max_task], 1).ravel()
        elif self.sampling in ('truncnorm', 'gaussian'):
            p = np.stack([self.mean_task, self.stdev_task], 1).ravel()
        else:
            lo, hi = self.get_task_search_bounds()
            chol = np.linalg.cholesky(np.asarray(self.cov_task, dtype=np.float64))
            p = np.concatenate([self.mean_task, chol.ravel(), lo, hi])
        p = np.ascontiguousarray(p, dtype=np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        _native.check(self._L.rex_set_dr(self._h, _native.DR_TYPES[self.sampling], p.ctypes.data_as(fp), p.size,
                                         lower.ctypes.data_as(fp)))

    def _set_dr_training_native(self):
        _native.check(self._L.rex_set_dr_training(self._h, int(self.dr_training)))

    def set_random_task(self, mask=None):     # random_env.py:37-39, for every (masked) env of the batch
        if self.sampling is None:
            raise ValueError('sampling value of random env needs to be set before using sample_task() or '
                             'set_random_task(). Set it by uploading a DR distr.')
        m = self._mask_ptr(mask)
        _native.check(self._L.rex_set_random_task(self._h, m, self._stream()))

    def sample_task(self):
        """One xi per env, WITHOUT applying it: [batch, task_dim] (the reference returns one vector)."""
        return self.sample_tasks(1)[0]

    def sample_tasks(self, num_tasks=1):      # random_env.py:145-146
        """[num_tasks, batch, task_dim] draws from the DR distribution.  Side-effect free like the reference's:
        neither the current task nor the episode counters (which key the reset streams) change."""
        if self.sampling is None:
            raise ValueError('sampling value of random env needs to be set before using sample_task() or '
                             'set_random_task(). Set it by uploading a DR distr.')
        t = self._torch
        out = t.empty(num_tasks, self.task_dim, self.batch, dtype=t.float32, device=self.device)
        for k in range(num_tasks):
            _native.check(self._L.rex_sample_task(self._h, ctypes.c_void_p(out[k].data_ptr()), self._draws, self._stream()))
            self._draws += 1
        return out.transpose(1, 2)

    # ------------------------------------------------------------------ task / state
    def get_task(self):
        xi = self._torch.empty(self.task_dim, self.batch, dtype=self._torch.float32, device=self.device)
        _native.check(self._L.rex_get_task(self._h, ctypes.c_void_p(xi.data_ptr()), self._stream()))
        return xi.t()

    def set_task(self, *task):
        """set_task(xi[batch, task_dim]) or set_task(*xi_scalars) broadcast to every env.  A tensor already on the
        env's device stays there (no host round trip, no synchronisation): the replay workload sets a fresh xi per
        call."""
        t = self._torch
        if len(task) == 1 and isinstance(task[0], t.Tensor):
            x = task[0].to(device=self.device, dtype=t.float32)
        elif len(task) == 1 and hasattr(task[0], "__len__"):
            x = t.as_tensor(np.asarray(task[0], dtype=np.float32)).to(self.device)
        else:
            x = t.as_tensor(np.asarray(task, dtype=np.float32)).to(self.device)
        if x.dim() == 1:
            x = x.unsqueeze(0).expand(self.batch, -1)
        assert tuple(x.shape) == (self.batch, self.task_dim), "task must be [batch, task_dim]"
        self._xi_in = x.t().contiguous()      # kept alive until the next call (the copy is stream-ordered)
        _native.check(self._L.rex_set_task(self._h, ctypes.c_void_p(self._xi_in.data_ptr()), self._stream()))

    def get_state(self):
        t = self._torch
        q = t.empty(self.dims.nq, self.batch, dtype=t.float32, device=self.device)
        v = t.empty(self.dims.nv, self.batch, dtype=t.float32, device=self.device)
        _native.check(self._L.rex_get_state(self._h, ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(v.data_ptr()), self._stream()))
        return q.t(), v.t()

    def set_state(self, qpos, qvel):          # jinja_mujoco_env.py:146-154
        t = self._torch
        q = t.as_tensor(qpos, dtype=t.float32).reshape(-1, self.dims.nq)
        v = t.as_tensor(qvel, dtype=t.float32).reshape(-1, self.dims.nv)
        assert q.shape[0] in (1, self.batch)
        self._q_in = q.to(self.device).expand(self.batch, -1).t().contiguous()
        self._v_in = v.to(self.device).expand(self.batch, -1).t().contiguous()
        _native.check(self._L.rex_set_state(self._h, ctypes.c_void_p(self._q_in.data_ptr()), ctypes.c_void_p(self._v_in.data_ptr()), self._stream()))

    def get_full_state(self):
        """Everything a bit-exact resume needs (the reference's MjSimState carries time as well, random_hopper.py:
        148-152; gym's TimeLimit its elapsed steps): qpos, qvel, xi and the per-lane step / episode counters that
        key the Philox streams, plus the done flags."""
        t = self._torch
        q, v = self.get_state()
        B = self.batch
        tt = t.empty(B, dtype=t.int32, device=self.device)
        ep = t.empty(B, dtype=t.int32, device=self.device)      # uint32 bit pattern
        dn = t.empty(B, dtype=t.uint8, device=self.device)
        _native.check(self._L.rex_get_counters_state(self._h, ctypes.c_void_p(tt.data_ptr()), ctypes.c_void_p(ep.data_ptr()),
                                                     ctypes.c_void_p(dn.data_ptr()), self._stream()))
        st = dict(qpos=q.clone(), qvel=v.clone(), task=self.get_task().clone(), t=tt, episode=ep, done=dn)
        if self.dims.n_aux:   # humanoid: data.xipos[:, 0] of the last mj_forward (mass_center() "before", random_humanoid.py:162)
            aux = t.empty(self.dims.n_aux, B, dtype=t.float32, device=self.device)
            _native.check(self._L.rex_get_aux(self._h, ctypes.c_void_p(aux.data_ptr()), self._stream()))
            st["aux"] = aux
        return st

    def set_full_state(self, st):
        """Inverse of :meth:`get_full_state`.  Order matters for the humanoid: set_state runs sim.forward() with the
        masses in force, so the task goes first (the same order get_full_state observed them in)."""
        self.set_task(st["task"])
        self.set_state(st["qpos"], st["qvel"])
        keep = [st[k].to(self.device).contiguous() for k in ("t", "episode", "done")]
        if self.dims.n_aux and "aux" in st:   # after set_state (whose sim.forward() refreshes it)
            keep.append(st["aux"].to(self.device).contiguous())
            _native.check(self._L.rex_set_aux(self._h, ctypes.c_void_p(keep[3].data_ptr()), self._stream()))
        self._ctr_in = keep
        _native.check(self._L.rex_set_counters_state(self._h, ctypes.c_void_p(keep[0].data_ptr()), ctypes.c_void_p(keep[1].data_ptr()),
                                                     ctypes.c_void_p(keep[2].data_ptr()), self._stream()))

    def export_lane(self, k=0):
        """Host snapshot of ONE env for an external viewer -- the data MujocoEnv.render / RandomCartPoleEnv.render
        read from the sim (jinja_mujoco_env.py:175-226, random_cartpole.py:231-283): numpy qpos, qvel, xi (+ names).
        Rendering itself stays off the GPU path."""
        fp = ctypes.POINTER(ctypes.c_float)
        q = np.zeros(self.dims.nq, dtype=np.float32); v = np.zeros(self.dims.nv, dtype=np.float32)
        xi = np.zeros(self.task_dim, dtype=np.float32)
        _native.check(self._L.rex_export_lane(self._h, int(k), q.ctypes.data_as(fp), v.ctypes.data_as(fp), xi.ctypes.data_as(fp)))
        return dict(env_id=self.env_id, kind=self.kind, lane=int(k), qpos=q, qvel=v, task=xi,
                    task_names=list(self.dyn_ind_to_name.values()) if hasattr(self.dyn_ind_to_name, "values") else list(self.dyn_ind_to_name))

    # ---- state (de)serialisation helpers used by offline-replay callers (DROPO-style), SURVEY section 8 f2 ----
    def get_sim_state(self):                  # random_hopper.py:151-152 (MjSimState -> (qpos, qvel) tensors)
        q, v = self.get_state()
        return q.clone(), v.clone()

    def set_sim_state(self, mjstate):         # random_hopper.py:148-149
        q, v = mjstate
        return self.set_state(q, v)

    def get_full_mjstate(self, state, template=None):
        """Observation -> full (qpos, qvel) with the root x (and y for the humanoid) zeroed
        (random_hopper.py:128-136, random_half_cheetah.py:136-146, random_walker2d.py:161-171,
        random_humanoid.py:244-253).  `state` is [batch or n, obs_dim] (only its qpos/qvel part is read)."""
        t = self._torch
        st = t.as_tensor(state, dtype=t.float32, device=self.device).reshape(-1, self.dims.obs_dim)
        nq, nv = self.dims.nq, self.dims.nv
        if self.kind == "cartpole":
            raise NotImplementedError("RandomCartPoleEnv has no get_full_mjstate (random_cartpole.py)")
        skip = 2 if self.kind == "humanoid" else 1
        q = t.zeros(st.shape[0], nq, dtype=t.float32, device=self.device)
        q[:, skip:] = st[:, :nq - skip]
        v = st[:, nq - skip:nq - skip + nv].clone()
        return q, v

    get_initial_mjstate = get_full_mjstate    # identical bodies upstream (random_hopper.py:138-146)

    def replay_transitions(self, obs, action, xi=None):
        """One logged transition per env under this env's (or the given) xi: set_sim_state(get_full_mjstate(obs)),
        step(action) without auto-reset side effects -> next observation.  The massively parallel inner loop of
        offline system identification (many candidate xi x one transition each).  Every MuJoCo chain and id goes
        through `rex_replay` (caller buffers in, caller buffers out, the env untouched)."""
        if self.kind == "cartpole":
            raise NotImplementedError("RandomCartPoleEnv has no get_full_mjstate / set_sim_state (random_cartpole.py)")
        return self._replay_fused(obs, action, self.get_task() if xi is None else xi)

    def replay_three_call(self, obs, action, xi=None):
        """The same transition through the reference's own call sequence (set_task + set_sim_state + step on the env
        itself, auto-reset off): what `rex_replay` is checked against."""
        keep = self.autoreset
        self.autoreset = False; self._push_flags()
        try:
            if xi is not None:
                self.set_task(xi)
            q, v = self.get_full_mjstate(obs)
            self.set_state(q, v)
            nxt, r, d, _ = self.step(action)
            return nxt.clone(), r.clone(), d.clone()
        finally:
            self.autoreset = keep; self._push_flags()

    def replay_soa(self, qpos_soa, qvel_soa, xi_soa, action_soa):
        """Zero-copy replay through `rex_replay`: reads the caller's contiguous SoA device tensors -- qpos [nq, batch] (root x
        zeroed, as get_full_mjstate gives it), qvel [nv, batch], xi [task_dim, batch], action [act_dim, batch] -- and writes
        next observation / reward / done into buffers of this object (valid until the next replay call); the env itself
        (state, task, counters, RNG position) is not touched.  One launch for hopper / half-cheetah; walker2d, the humanoid
        and the Unmodeled ids add the launch their set_task / set_state needs (include/rex.h)."""
        t = self._torch
        if not hasattr(self, "_rp_obs"):
            self._rp_obs = t.empty(self.dims.obs_dim, self.batch, dtype=t.float32, device=self.device)
            self._rp_r = t.empty(self.batch, dtype=t.float32, device=self.device)
            self._rp_d = t.empty(self.batch, dtype=t.uint8, device=self.device)
        p = lambda z: ctypes.c_void_p(z.data_ptr())
        _native.check(self._L.rex_replay(self._h, p(qpos_soa), p(qvel_soa), p(xi_soa), p(action_soa), p(self._rp_obs), p(self._rp_r),
                                         p(self._rp_d), self._stream()))
        return self._rp_obs, self._rp_r, self._rp_d

    def _replay_fused(self, obs, action, xi):
        """[batch, dim] convenience form of :meth:`replay_soa`: returns fresh [batch, ...] tensors."""
        t = self._torch
        q, v = self.get_full_mjstate(obs)
        x = t.as_tensor(xi, dtype=t.float32, device=self.device).reshape(self.batch, self.task_dim)
        a = t.as_tensor(action, dtype=t.float32, device=self.device).reshape(self.batch, self.dims.act_dim)
        keep = (q.t().contiguous(), v.t().contiguous(), x.t().contiguous(), a.t().contiguous())
        self._replay_keep = keep              # alive until the launch has consumed them (stream-ordered)
        o, r, d = self.replay_soa(*keep)
        return o.t().clone(), r.clone(), d.bool()

    def set_model_args(self, size):           # jinja_mujoco_env.py:89-90 (the template's `size` list; walker2d: xi[7:11])
        """Walker2d: the four link lengths of the Jinja template, for every env (the rest of the task is kept); the other
        chains have no per-env template arguments here (their `size` lists are compile-time constants of the kernels)."""
        if self.kind != "walker2d" or self.unmodeled:
            raise NotImplementedError("set_model_args: only RandomWalker2d-v0 rebuilds its model from template arguments")
        xi = self.get_task().clone()
        xi[:, 7:11] = self._torch.as_tensor(np.asarray(size, dtype=np.float32), device=self.device).reshape(-1, 4)
        self.set_task(xi)

    def state_vector(self):                   # jinja_mujoco_env.py:231-235
        q, v = self.get_state()
        return self._torch.cat([q, v], 1)

    def seed(self, seed=None):                # jinja_mujoco_env.py:109-111
        seed = 0 if seed is None else int(seed)
        _native.check(self._L.rex_seed(self._h, seed))
        return [seed]

    # ------------------------------------------------------------------ gym protocol
    def _mask_ptr(self, mask):
        if mask is None:
            return None
        t = self._torch
        self._mask = t.as_tensor(mask).to(device=self.device, dtype=t.uint8).contiguous()
        assert self._mask.numel() == self.batch
        return ctypes.c_void_p(self._mask.data_ptr())

    def reset(self, mask=None):
        m = self._mask_ptr(mask)
        _native.check(self._L.rex_reset(self._h, m, ctypes.c_void_p(self._obs.data_ptr()), self._stream()))
        return self._obs.t()

    def step(self, action):
        t = self._torch
        a = t.as_tensor(action, device=self.device)
        if self.dims.discrete_action:
            a = a.reshape(self.batch).to(t.int32)
            self._act.view(-1).copy_(a)
        else:
            a = a.reshape(self.batch, self.dims.act_dim).to(t.float32)
            self._act.copy_(a.t())
        _native.check(self._L.rex_step(
            self._h, ctypes.c_void_p(self._act.data_ptr()), ctypes.c_void_p(self._obs.data_ptr()),
            ctypes.c_void_p(self._reward.data_ptr()), ctypes.c_void_p(self._done.data_ptr()),
            ctypes.c_void_p(self._trunc.data_ptr()), ctypes.c_void_p(self._term_obs.data_ptr()), self._stream()))
        info = {"TimeLimit.truncated": self._trunc.bool(), "terminal_observation": self._term_obs.t()}
        for k, name in enumerate(self.INFO_TERMS[self.kind]):    # per-term rewards, [batch] each
            info[name] = self._info[k]
        return self._obs.t(), self._reward, self._done.bool(), info

    def step_soa(self, action_soa):
        """Zero-copy hot path: ``action_soa`` is a contiguous [act_dim, batch] tensor on the env's
        device; returns the SoA obs/reward/done buffers (no transposes, no info dict)."""
        _native.check(self._L.rex_step(
            self._h, ctypes.c_void_p(action_soa.data_ptr()), ctypes.c_void_p(self._obs.data_ptr()),
            ctypes.c_void_p(self._reward.data_ptr()), ctypes.c_void_p(self._done.data_ptr()), None, None, self._stream()))
        return self._obs, self._reward, self._done

    def step_count(self):
        return int(self._L.rex_step_count(self._h))

    def counters(self):
        out = (ctypes.c_int64 * 4)()
        _native.check(self._L.rex_get_counters(self._h, out))
        return dict(nonfinite=out[0], gaussian_fail=out[1], solver_capped=out[2], overflow=out[3])

    def launch_shape(self):
        """What rex_create picked for this handle from its batch and the GPU's SIMD count, or what set_launch_shape pinned."""
        out = (ctypes.c_int32 * 4)()
        _native.check(self._L.rex_get_launch_shape(self._h, out))
        return dict(lanes=int(out[0]), pair=bool(out[1]), rolled=bool(out[2]), hum_pair=bool(out[3]))

    def set_launch_shape(self, lanes=None, pair=None, rolled=None, hum_pair=None):
        """Pin the launch shape of this handle (rex_set_launch_shape; None keeps a field).  Two runs agree bit for bit only under the
        same shape, so `sharding.pin_global_shape` gives every shard of a split batch the shape its GLOBAL batch gets on one GPU."""
        f = lambda v: -1 if v is None else int(v)
        arr = (ctypes.c_int32 * 4)(f(lanes), f(pair), f(rolled), f(hum_pair))
        _native.check(self._L.rex_set_launch_shape(self._h, arr))
        return self.launch_shape()

    def enable_timing(self, every=1):
        """HIP-event duration of every `every`-th step kernel launch (True / 1: all of them, 0 / False: off)."""
        _native.check(self._L.rex_enable_timing(self._h, int(every)))

    def read_timing(self, max_n=65536):
        buf = (ctypes.c_float * max_n)()
        n = self._L.rex_read_timing(self._h, buf, max_n)
        return np.frombuffer(buf, dtype=np.float32, count=max(n, 0)).copy()
