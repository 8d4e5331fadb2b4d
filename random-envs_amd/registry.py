"""gym.make-style factory under the reference's ids (random_envs/__init__.py + the
``gym.envs.register`` calls at the bottom of every task file)."""
from .specs import IDS, PENDING_IDS, SPECS, UNMODELED_SPECS


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
    if env_id in PENDING_IDS:
        raise NotImplementedError("%s: kernel not built yet (SURVEY.md section 8 rows a7 / f1)" % env_id)
    if env_id not in IDS:
        raise KeyError("No registered env with id: %s" % env_id)
    kind, kw = IDS[env_id]
    kw = dict(kw); kw.update(kwargs)
    return VecRandomEnv(kind, batch=batch, device=device, seed=seed, env_offset=env_offset, env_id=env_id, **kw)
