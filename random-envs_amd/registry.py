"""gym.make-style factory under the reference's ids (random_envs/__init__.py + the
``gym.envs.register`` calls at the bottom of every task file)."""
from .specs import IDS, SPECS, UNMODELED_SPECS


def registered_ids():
    return sorted(IDS)


def spec(env_id):
    if env_id not in IDS:
        raise KeyError("No registered env with id: %s" % env_id)
    kind, kw = IDS[env_id]
    return UNMODELED_SPECS[kind] if kw.get("unmodeled") else SPECS[kind]


def make(env_id, batch=1, device=0, seed=0, env_offset=0, **kwargs):
    """``gym.make(env_id)`` for a batch of ``batch`` independent environments on GPU ``device``.

    ``env_offset`` is the global index of this shard's first env (multi-GPU sharding keeps RNG
    streams keyed by global index)."""
    from .vec_env import VecRandomEnv
    if env_id not in IDS:
        raise KeyError("No registered env with id: %s" % env_id)
    kind, kw = IDS[env_id]
    kw = dict(kw); kw.update(kwargs)
    return VecRandomEnv(kind, batch=batch, device=device, seed=seed, env_offset=env_offset, env_id=env_id, **kw)


def register_with_gym(batch=1, **kwargs):
    """The import side effect of ``random_envs/__init__.py`` (``gym.envs.register`` at the bottom of every task
    file, e.g. random_hopper.py:155-166) for whichever of ``gym`` / ``gymnasium`` is importable: the 13 ids with
    ``max_episode_steps=500`` resolve to :func:`make` with ``batch`` environments.  Returns the names of the
    packages registered with (empty when neither is installed -- nothing here depends on them)."""
    done = []
    for name in ("gym", "gymnasium"):
        try:
            mod = __import__(name)
            register = mod.envs.registration.register if hasattr(mod.envs, "registration") else mod.envs.register
        except Exception:
            continue
        for env_id in IDS:
            try:
                register(id=env_id, entry_point="random_envs_amd.registry:_gym_entry",
                         kwargs=dict(env_id=env_id, batch=batch, **kwargs))
            except Exception:   # already registered
                pass
        done.append(name)
    return done


def _gym_entry(env_id, **kwargs):
    return make(env_id, **kwargs)
