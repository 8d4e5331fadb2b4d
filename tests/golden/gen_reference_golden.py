#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (run in the build container only).

The reference (`/root/reference`, gabrieletiboni/random-envs) cannot travel to the GPU box, so
the vectors produced here are committed as plain data under tests/golden/ and this script is
committed next to them as their provenance.

What is importable here: `random_envs/random_env.py` (DR base class) and
`random_envs/random_cartpole.py` (analytic cart-pole).  They need `gym`, which this image lacks,
so a throw-away ~30-line stand-in for the few gym symbols they touch (Env, spaces.Box/Discrete,
logger.warn, utils.seeding.np_random, envs.register) is created in a temp dir.  The two
reference files are loaded *by path* (never copied).  The MuJoCo-backed envs are NOT importable
(mujoco_py / libmujoco210 absent) -> no golden vectors exist for them ("parity unpinned").

Outputs (tests/golden/):
  cartpole_step.json     -- (state, action, xi) -> (state', reward, done), incl. step-after-done
  cartpole_rollout.json  -- seeded reset + 200-step rollouts with default and randomised xi
  dr_sampler.json        -- set_dr_distribution / sample_task / denormalize_parameters I/O
"""
import importlib.util, json, os, sys, tempfile, types
import numpy as np

REF = os.environ.get("REX_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_gym_standin():
    d = tempfile.mkdtemp(prefix="gymshim_")
    os.makedirs(os.path.join(d, "gym", "utils"))
    os.makedirs(os.path.join(d, "gym", "envs"))
    open(os.path.join(d, "gym", "__init__.py"), "w").write(
        "from . import spaces, logger, utils, envs\n"
        "class Env(object):\n    pass\n")
    open(os.path.join(d, "gym", "spaces.py"), "w").write(
        "import numpy as np\n"
        "class Box:\n"
        "    def __init__(self, low, high, dtype=np.float32):\n"
        "        self.low, self.high, self.dtype, self.shape = low, high, dtype, np.shape(low)\n"
        "class Discrete:\n"
        "    def __init__(self, n):\n        self.n = n\n"
        "    def contains(self, x):\n        return int(x) == x and 0 <= int(x) < self.n\n")
    open(os.path.join(d, "gym", "logger.py"), "w").write("def warn(*a, **k):\n    pass\n")
    open(os.path.join(d, "gym", "utils", "__init__.py"), "w").write("from . import seeding\n")
    open(os.path.join(d, "gym", "utils", "seeding.py"), "w").write(
        "import numpy as np\n"
        "def np_random(seed=None):\n"
        "    return np.random.RandomState(seed), seed\n")
    open(os.path.join(d, "gym", "envs", "__init__.py"), "w").write(
        "registry = {}\n"
        "def register(id, **kw):\n    registry[id] = kw\n")
    sys.path.insert(0, d)


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    _install_gym_standin()
    pkg = types.ModuleType("random_envs")
    pkg.__path__ = []  # synthetic package: bypasses random_envs/__init__.py (-> mujoco_py)
    sys.modules["random_envs"] = pkg
    _load("random_envs.random_env", "random_envs/random_env.py")
    cp = _load("random_envs.random_cartpole", "random_envs/random_cartpole.py")
    import gym
    registry = dict(gym.envs.registry)

    rng = np.random.RandomState(1234)

    # ---------------- single-step vectors -----------------
    cases = []
    env = cp.RandomCartPoleEnv()
    lo = np.array([2.0, 0.5, 0.05, 0.1]); hi = np.array([20.0, 3.0, 0.3, 1.0])
    for k in range(400):
        xi = env.original_task.copy() if k % 4 == 0 else rng.uniform(lo, hi)
        env.set_task(*xi)
        if k % 5 == 0:   # near the termination thresholds
            st = np.array([rng.choice([-1, 1]) * rng.uniform(2.3, 2.5), rng.uniform(-3, 3),
                           rng.choice([-1, 1]) * rng.uniform(0.19, 0.23), rng.uniform(-3, 3)])
        else:
            st = np.array([rng.uniform(-2.4, 2.4), rng.uniform(-3, 3),
                           rng.uniform(-0.21, 0.21), rng.uniform(-3, 3)])
        a = int(rng.randint(2))
        env.state = tuple(st); env.steps_beyond_done = None
        s1, r1, d1, _ = env.step(a)
        # one more step from the new state: exercises the reward-0-after-done branch
        a2 = int(rng.randint(2))
        s2, r2, d2, _ = env.step(a2)
        cases.append(dict(xi=list(map(float, xi)), state=list(map(float, st)), action=a,
                          next_state=list(map(float, s1)), reward=float(r1), done=bool(d1),
                          action2=a2, next_state2=list(map(float, s2)), reward2=float(r2),
                          done2=bool(d2), polemass_length=float(env.polemass_length)))
    json.dump(dict(source="random_envs/random_cartpole.py:172-224 (step), :162-166 (set_task)",
                   cases=cases), open(os.path.join(OUT, "cartpole_step.json"), "w"))

    # ---------------- seeded rollouts -----------------
    rollouts = []
    for seed, xi in [(0, None), (1, [15.0, 2.0, 0.2, 0.8]), (2, [3.0, 0.6, 0.06, 0.15])]:
        env = cp.RandomCartPoleEnv()
        env.seed(seed)
        if xi is not None:
            env.set_task(*xi)
        s0 = env.reset()
        arng = np.random.RandomState(100 + seed)
        acts, states, rews, dones = [], [list(map(float, s0))], [], []
        for t in range(200):
            a = int(arng.randint(2)); s, r, d, _ = env.step(a)
            acts.append(a); states.append(list(map(float, s))); rews.append(float(r)); dones.append(bool(d))
        rollouts.append(dict(seed=seed, xi=list(map(float, env.get_task())), actions=acts,
                             states=states, rewards=rews, dones=dones))
    json.dump(dict(source="random_envs/random_cartpole.py:172-229", rollouts=rollouts),
              open(os.path.join(OUT, "cartpole_rollout.json"), "w"))

    # ---------------- DR sampler -----------------
    env = cp.RandomCartPoleEnv()
    out = dict(source="random_envs/random_env.py:72-220", registry=registry)
    env.set_dr_distribution("uniform", [5, 15, 0.8, 1.2, 0.08, 0.12, 0.4, 0.6])
    out["uniform_get"] = [list(map(float, x)) for x in env.get_dr_distribution()]
    np.random.seed(7); out["uniform_seed7"] = env.sample_tasks(5).tolist()
    env.set_dr_distribution("gaussian", [9.8, 1.0, 1.0, 0.1, 0.2, 0.02, 0.5, 0.05])
    np.random.seed(7); out["gaussian_seed7"] = env.sample_tasks(5).tolist()
    np.random.seed(7); out["gaussian_seed7_randn"] = np.random.randn(20).tolist()
    mean = [2.0, 1.0, 3.0, 2.5]; cov = (np.diag([0.5, 0.2, 0.3, 0.1]) + 0.05).tolist()
    env.set_dr_distribution("fullgaussian", dict(mean=mean, cov=cov))
    np.random.seed(7); out["fullgaussian_seed7"] = env.sample_tasks(5).tolist()
    np.random.seed(7); out["fullgaussian_seed7_raw"] = np.stack(
        [np.random.multivariate_normal(mean, cov) for _ in range(5)]).tolist()
    out["fullgaussian_mean"] = mean; out["fullgaussian_cov"] = cov
    out["search_bounds"] = [x.tolist() for x in env.get_task_search_bounds()]
    out["denorm_in"] = [0.0, 1.0, 2.5, 4.0]
    out["denorm_out"] = env.denormalize_parameters(np.array(out["denorm_in"])).tolist()
    # behaviours of the reference that the build documents
    quirks = {}
    env.set_dr_distribution("truncnorm", [9.8, 1.0, 1.0, 0.1, 0.2, 0.02, 0.5, 0.05])
    out["truncnorm_get"] = [list(map(float, x)) for x in env.get_dr_distribution()]
    try:
        env.sample_task(); quirks["truncnorm_sample"] = "ok"
    except Exception as e:
        quirks["truncnorm_sample"] = type(e).__name__
    try:
        env.load_dr_distribution_from_file("/nonexistent"); quirks["load_file"] = "ok"
    except Exception as e:
        quirks["load_file"] = type(e).__name__
    env2 = cp.RandomCartPoleEnv(); env2.seed(0)
    env2.set_dr_distribution("uniform", [5, 15, 0.8, 1.2, 0.08, 0.12, 0.4, 0.6]); env2.set_dr_training(True)
    t0 = env2.get_task().tolist(); env2.reset(); quirks["cartpole_reset_resamples"] = env2.get_task().tolist() != t0
    env2.set_random_task(); quirks["polemass_length_after_set_task"] = float(env2.polemass_length)
    try:
        env2.get_endless(); quirks["cartpole_get_endless"] = "ok"
    except Exception as e:
        quirks["cartpole_get_endless"] = type(e).__name__
    try:
        env2.set_dr_distribution("bogus", []); quirks["unknown_dr_type"] = "ok"
    except Exception as e:
        quirks["unknown_dr_type"] = type(e).__name__
    env3 = cp.RandomCartPoleEnv()
    try:
        env3.sample_task(); quirks["sample_before_set"] = "ok"
    except Exception as e:
        quirks["sample_before_set"] = type(e).__name__
    out["quirks"] = quirks
    out["cartpole_bounds"] = dict(
        search=[list(env.get_search_bounds_mean(i)) for i in range(4)],
        lower=[env.get_task_lower_bound(i) for i in range(4)],
        names=[env.dyn_index_to_name(i) for i in range(4)],
        reward_threshold=env.get_reward_threshold(), task_dim=env.task_dim,
        original_task=env.original_task.tolist(),
        theta_threshold=env.theta_threshold_radians, x_threshold=env.x_threshold)
    json.dump(out, open(os.path.join(OUT, "dr_sampler.json"), "w"), indent=0)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
