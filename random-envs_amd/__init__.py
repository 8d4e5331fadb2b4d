"""rex -- MI355X-native batched domain-randomised locomotion environments.

Drop-in for the hot path of gabrieletiboni/random-envs: ``make(id, batch=...)`` returns a
:class:`VecRandomEnv` with the reference's gym surface (``reset``, ``step``,
``set_dr_distribution``, ``set_dr_training``, ``get_task``/``set_task`` ...) whose step and
reset run as hand-written HIP kernels through the C-ABI of ``include/rex.h``.
There is no CPU fallback: importing works anywhere, creating an env needs ``librex_hip.so``
and a GPU.
"""
from .registry import make, registered_ids, spec  # noqa: F401
from .vec_env import VecRandomEnv  # noqa: F401
from . import _native  # noqa: F401

__all__ = ["make", "registered_ids", "spec", "VecRandomEnv"]
