"""ctypes bindings of the ORACLE (oracle/_build/libmjo.so) for tests, smoke() and bench.py's
cpu_baseline leg.  Never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "_build", "libmjo.so")
KINDS = {"cartpole": 0, "hopper": 1, "halfcheetah": 2, "walker2d": 3, "humanoid": 4}
DIMS = {"humanoid": dict(nq=24, nv=23, nu=17, nx=30, nobs=376, frame_skip=5),
        "hopper": dict(nq=6, nv=6, nu=3, nx=4, nobs=11, frame_skip=4),
        "walker2d": dict(nq=9, nv=9, nu=6, nx=13, nobs=17, frame_skip=4),
        "halfcheetah": dict(nq=9, nv=9, nu=6, nx=8, nobs=17, frame_skip=5)}
_D = ctypes.POINTER(ctypes.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        try:
            _lib = ctypes.CDLL(_LIB)
        except OSError:   # built on another host: rebuild here
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "clean"], stdout=subprocess.DEVNULL)
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
            _lib = ctypes.CDLL(_LIB)
        _lib.mjo_set_tolerance.argtypes = [ctypes.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(_D) if a is not None else None


def _soa(x, dim):
    """[n, dim] -> contiguous float64 [dim, n]"""
    x = np.asarray(x, dtype=np.float64).reshape(-1, dim)
    return np.ascontiguousarray(x.T)


UNMODELED_NX = {"hopper": 3, "halfcheetah": 5, "walker2d": 9, "humanoid": 23}


def oracle_batch_step(kind, qpos, qvel, action, xi, nthreads=8, tolerance=1e-12, variant=0):
    """One env.step() per row from (qpos, qvel, action, xi); all arrays [n, dim].
    variant=1: the Unmodeled id (xi is the reduced task; a NaN xi row keeps the freshly built model)."""
    d = DIMS[kind]; L = lib(); L.mjo_set_tolerance(tolerance)
    nx = UNMODELED_NX[kind] if variant else d["nx"]
    q, v, a, x = _soa(qpos, d["nq"]), _soa(qvel, d["nv"]), _soa(action, d["nu"]), _soa(xi, nx)
    n = q.shape[1]
    qo = np.zeros_like(q); vo = np.zeros_like(v); obs = np.zeros((d["nobs"], n)); r = np.zeros(n)
    dn = np.zeros(n, dtype=np.uint8)
    rc = L.mjo_batch_step(KINDS[kind], int(variant), n, _p(q), _p(v), _p(a), _p(x), _p(qo), _p(vo), _p(obs), _p(r),
                          dn.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), nthreads)
    assert rc == 0
    return dict(qpos=qo.T.copy(), qvel=vo.T.copy(), obs=obs.T.copy(), reward=r, done=dn.astype(bool))


def oracle_rollout(kind, qpos, qvel, actions, xi, nthreads=8, tolerance=0.0):
    """`steps` env-steps per env without reset; actions [steps, n, nu]. tolerance 0 = the model's 1e-8."""
    d = DIMS[kind]; L = lib(); L.mjo_set_tolerance(tolerance)
    q, v, x = _soa(qpos, d["nq"]), _soa(qvel, d["nv"]), _soa(xi, d["nx"])
    n = q.shape[1]
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).transpose(0, 2, 1))   # [steps][nu][n]
    steps = acts.shape[0]
    qo = np.zeros_like(q); vo = np.zeros_like(v); r = np.zeros(n)
    rc = L.mjo_batch_rollout(KINDS[kind], 0, n, steps, _p(q), _p(v), _p(acts), _p(x), _p(qo), _p(vo), _p(r), nthreads)
    assert rc == 0
    return dict(qpos=qo.T.copy(), qvel=vo.T.copy(), reward_sum=r)


def oracle_rollout_autoreset(kind, qpos, qvel, actions, xi, qpos_reset, qvel_reset, nthreads=8, tolerance=0.0):
    """oracle_rollout with the batched env's auto-reset: a lane whose step returns done restarts from its row of
    (qpos_reset, qvel_reset).  Returns the end state, the reward sums and the number of resets per lane."""
    d = DIMS[kind]; L = lib(); L.mjo_set_tolerance(tolerance)
    q, v, x = _soa(qpos, d["nq"]), _soa(qvel, d["nv"]), _soa(xi, d["nx"])
    qr, vr = _soa(qpos_reset, d["nq"]), _soa(qvel_reset, d["nv"])
    n = q.shape[1]
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).transpose(0, 2, 1))   # [steps][nu][n]
    qo = np.zeros_like(q); vo = np.zeros_like(v); r = np.zeros(n); resets = np.zeros(n, dtype=np.int64)
    rc = L.mjo_batch_rollout_autoreset(KINDS[kind], 0, n, acts.shape[0], _p(q), _p(v), _p(acts), _p(x), _p(qr), _p(vr), _p(qo), _p(vo),
                                       _p(r), resets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), nthreads)
    assert rc == 0
    return dict(qpos=qo.T.copy(), qvel=vo.T.copy(), reward_sum=r, resets=resets)


def oracle_forward(kind, qpos, qvel, action, xi, tolerance=1e-12):
    d = DIMS[kind]; L = lib(); L.mjo_set_tolerance(tolerance)
    nv = d["nv"]
    q = np.ascontiguousarray(qpos, dtype=np.float64); v = np.ascontiguousarray(qvel, dtype=np.float64)
    a = np.ascontiguousarray(action, dtype=np.float64); x = np.ascontiguousarray(xi, dtype=np.float64)
    qacc = np.zeros(nv); qs = np.zeros(nv); M = np.zeros((nv, nv)); b = np.zeros(nv); f = np.zeros(256)
    nc, ne, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = L.mjo_probe_forward(KINDS[kind], _p(q), _p(v), _p(a), _p(x), _p(qacc), _p(qs), _p(M), _p(b),
                             ctypes.byref(nc), ctypes.byref(ne), _p(f), ctypes.byref(it))
    assert rc == 0
    return dict(qacc=qacc, qacc_smooth=qs, M=M, bias=b, ncon=nc.value, nefc=ne.value, iters=it.value, force=f[:ne.value])


def oracle_contacts(kind, qpos, qvel, xi):
    """rows: geom1, geom2, dist, pos(3), normal(3), dim"""
    L = lib()
    q = np.ascontiguousarray(qpos, dtype=np.float64); v = np.ascontiguousarray(qvel, dtype=np.float64)
    x = np.ascontiguousarray(xi, dtype=np.float64); out = np.zeros(10 * 32)
    n = L.mjo_probe_contacts(KINDS[kind], _p(q), _p(v), _p(x), _p(out), 32)
    return out.reshape(32, 10)[:n].copy()


def oracle_humanoid_step(qpos, qvel, action, xi, xipos_x_prev=None, nthreads=8):
    """RandomHumanoidEnv.step from (qpos[n,24], qvel[n,23], action[n,17], xi[n,30]); xipos_x_prev [n,14] or None
    (None: the state was just set -> sim.forward())."""
    L = lib()
    q, v, a, x = _soa(qpos, 24), _soa(qvel, 23), _soa(action, 17), _soa(xi, 30)
    n = q.shape[1]
    xp = None if xipos_x_prev is None else _soa(xipos_x_prev, 14)
    qo = np.zeros_like(q); vo = np.zeros_like(v); obs = np.zeros((376, n)); r = np.zeros(n); dn = np.zeros(n, dtype=np.uint8)
    xo = np.zeros((14, n))
    rc = L.mjo_humanoid_batch_step(n, _p(q), _p(v), _p(a), _p(x), _p(xp), _p(qo), _p(vo), _p(obs), _p(r),
                                   dn.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _p(xo), nthreads)
    assert rc == 0
    return dict(qpos=qo.T.copy(), qvel=vo.T.copy(), obs=obs.T.copy(), reward=r, done=dn.astype(bool), xipos_x=xo.T.copy())


def oracle_humanoid_reset_obs(qpos, qvel, xi):
    """Observation of reset() / set_state at (qpos[n,24], qvel[n,23]) with the task xi[n,30] in force
    (random_humanoid.py:219-234): (obs[n,376], xipos_x[n,14])."""
    L = lib()
    q, v, x = _soa(qpos, 24), _soa(qvel, 23), _soa(xi, 30)
    n = q.shape[1]
    obs = np.zeros((376, n)); xo = np.zeros((14, n))
    rc = L.mjo_humanoid_batch_reset_obs(n, _p(q), _p(v), _p(x), _p(obs), _p(xo))
    assert rc == 0
    return obs.T.copy(), xo.T.copy()


def oracle_sensitivity(step_fn, inputs, keys, rel=2.0 ** -22, trials=3, seed=0):
    """How far the ORACLE's own outputs move when its inputs move by fp32-rounding-sized amounts: per lane, the largest
    |out(x + dx) - out(x)| over `trials` random perturbations dx ~ +-rel * (1 + |x|) of every input array.
    A lane whose GPU-vs-oracle error exceeds the stated tolerance is `explained` only if the oracle itself is that
    ill-conditioned there (a contact / limit / solver active-set switch within rounding of the inputs)."""
    rng = np.random.RandomState(seed)
    base = step_fn(*inputs)
    sens = {k: np.zeros(np.asarray(base[k]).shape[0]) for k in keys}
    for _ in range(trials):
        pert = [x + rel * (1 + np.abs(x)) * rng.choice([-1.0, 1.0], size=x.shape) for x in inputs]
        out = step_fn(*pert)
        for k in keys:
            d = np.abs(np.asarray(out[k], dtype=np.float64) - np.asarray(base[k], dtype=np.float64))
            d = d.reshape(d.shape[0], -1).max(1)
            sens[k] = np.maximum(sens[k], np.where(np.isfinite(d), d, np.inf))
    return base, sens


def oracle_constants(kind, size=None):
    L = lib()
    bm = np.zeros(16); bi = np.zeros(16 * 9); ip = np.zeros(16 * 3); iw = np.zeros(32); dw = np.zeros(24); q0 = np.zeros(26)
    dims = np.zeros(6, dtype=np.int32)
    s = None if size is None else np.ascontiguousarray(size, dtype=np.float64)
    rc = L.mjo_model_constants(KINDS[kind], _p(s), _p(bm), _p(bi), _p(ip), _p(iw), _p(dw), _p(q0),
                               dims.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    nb, nq, nv = dims[0], dims[1], dims[2]
    return dict(body_mass=bm[:nb], body_inertia=bi.reshape(16, 9)[:nb], body_ipos=ip.reshape(16, 3)[:nb],
                body_invweight0=iw.reshape(16, 2)[:nb], dof_invweight0=dw[:nv], qpos0=q0[:nq],
                nbody=int(nb), nq=int(nq), nv=int(nv), ngeom=int(dims[3]), nu=int(dims[4]), npair=int(dims[5]))


def oracle_energy_drift(kind, steps, qpos, qvel, keep_springs=True):
    L = lib()
    e0, e1 = ctypes.c_double(), ctypes.c_double()
    q = np.ascontiguousarray(qpos, dtype=np.float64); v = np.ascontiguousarray(qvel, dtype=np.float64)
    rc = L.mjo_test_energy_drift(KINDS[kind], steps, _p(q), _p(v), int(keep_springs), ctypes.byref(e0), ctypes.byref(e1))
    assert rc == 0
    return e0.value, e1.value


def oracle_cartpole_step(state, action, xi):
    """states [n,4] (x, x_dot, theta, theta_dot), action [n] int, xi [n,4]"""
    L = lib()
    s = _soa(state, 4); x = _soa(xi, 4); n = s.shape[1]
    a = np.ascontiguousarray(action, dtype=np.int32)
    o = np.zeros_like(s); r = np.zeros(n); d = np.zeros(n, dtype=np.uint8)
    L.mjo_cartpole_batch_step(n, _p(s), a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(x), _p(o), _p(r),
                              d.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    return o.T.copy(), r, d.astype(bool)


def rollout_states(kind, n, steps_max=60, seed=0, xi_scale=(0.7, 1.3), nthreads=8):
    """States sampled along oracle rollouts under random actions from the reset distribution
    (physically reachable states: contacts at realistic penetration, joint limits being hit)."""
    from random_envs_amd.specs import SPECS
    d = DIMS[kind]; rng = np.random.RandomState(seed)
    spec = SPECS[kind]
    nom = np.array(spec.nominal_task)
    xi = np.tile(nom, (n, 1)) * rng.uniform(xi_scale[0], xi_scale[1], (n, d["nx"]))
    q = rng.uniform(-0.005, 0.005, (n, d["nq"])); v = rng.uniform(-0.005, 0.005, (n, d["nv"]))
    if kind != "halfcheetah":
        q[:, 1] += 1.25
    else:
        q = rng.uniform(-0.1, 0.1, (n, d["nq"])); v = 0.1 * rng.randn(n, d["nv"])
    steps = rng.randint(0, steps_max, n)
    order = np.argsort(steps)
    q, v, xi, steps = q[order], v[order], xi[order], steps[order]
    done_mask = np.zeros(n, bool)
    for s in range(steps.max()):
        act = rng.uniform(-1, 1, (n, d["nu"]))
        live = steps > s
        if not live.any():
            break
        idx = np.where(live)[0]
        out = oracle_batch_step(kind, q[idx], v[idx], act[idx], xi[idx], nthreads=nthreads, tolerance=0.0)
        q[idx] = out["qpos"]; v[idx] = out["qvel"]
    return q, v, xi
