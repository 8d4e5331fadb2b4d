"""Per-environment spec tables (pure data; usable without a GPU).

Everything here is transcribed from the reference's task-definition files; the cited lines are
paths inside gabrieletiboni/random-envs.
"""
from collections import namedtuple

EnvSpec = namedtuple("EnvSpec", [
    "kind",            # key of _native.ENV_KINDS
    "names",           # dyn_ind_to_name
    "search_bounds",   # get_search_bounds_mean(i)
    "lower_bounds",    # get_task_lower_bound(i)
    "nominal_task",    # get_task() of a freshly built env
    "reward_threshold", "preferred_lr",
    "noise_level",     # variance of the obs noise when noisy=True
    "dr_on_reset",     # does reset() call set_random_task() when dr_training?
])

_M = (0.5, 10.0)

CARTPOLE = EnvSpec(                                   # random_envs/random_cartpole.py
    kind="cartpole",
    names=["gravity", "cart_mass", "pole_mass", "pole_length"],                 # :104-107
    search_bounds=[(2., 20.0), (0.5, 3.0), (0.05, 0.3), (0.1, 1.)],            # :127-132
    lower_bounds=[0.1, 0.1, 0.1, 0.1],                                          # :140-145
    nominal_task=[9.8, 1.0, 0.1, 0.5],                                          # :74-78
    reward_threshold=500, preferred_lr=None, noise_level=0.0,                  # :120
    dr_on_reset=False)                                                          # :226-229

HOPPER = EnvSpec(                                     # random_envs/jinja/random_hopper.py
    kind="hopper",
    names=["torsomass", "thighmass", "legmass", "footmass"],                    # :42
    search_bounds=[_M] * 4,                                                     # :52-57
    lower_bounds=[0.1] * 4,                                                     # :65-70
    # body_mass[1:] of the compiled hopper.xml under MuJoCo 2.1.0 (1000*pi*r^2*(L+r))
    nominal_task=[3.5342917352885186, 3.9269908169872427, 2.7143360527015816, 5.0893800988154645],
    reward_threshold=1750, preferred_lr=0.0005, noise_level=1e-4,              # :44-45,28
    dr_on_reset=True)                                                           # :117-118

HALFCHEETAH = EnvSpec(                                # random_envs/jinja/random_half_cheetah.py
    kind="halfcheetah",
    names=["torso", "bthigh", "bshin", "bfoot", "fthigh", "fshin", "ffoot", "friction"],   # :46
    search_bounds=[_M] * 7 + [(0.1, 2.0)],                                      # :55-64
    lower_bounds=[0.1] * 7 + [0.02],                                            # :72-81
    nominal_task=[6.360313315926893, 1.5352480417754566, 1.5809399477806787, 1.069190600522193,
                  1.425587467362924, 1.1788511749347257, 0.8498694516971277, 0.4],   # settotalmass=14; friction :37
    reward_threshold=4500, preferred_lr=0.0005, noise_level=1e-4,              # :48-49,30
    dr_on_reset=True)                                                           # :128-129

WALKER2D = EnvSpec(                                   # random_envs/jinja/random_walker2d.py
    kind="walker2d",
    names=["torso", "thigh", "leg", "foot", "thigh_left", "leg_left", "foot_left",
           "torsosize", "thighsize", "legsize", "footsize", "friction_right", "friction_left"],   # :46
    search_bounds=[_M] * 7 + [(0.15, 1.0)] * 4 + [(0.1, 3.0)] * 2,              # :56-73
    lower_bounds=[0.1] * 7 + [0.1] * 4 + [0.05] * 2,                            # :80-96
    nominal_task=[3.5342917352885186, 3.9269908169872427, 2.7143360527015816, 2.9405307237600464,
                  3.9269908169872427, 2.7143360527015816, 2.9405307237600464,
                  0.4, 0.45, 0.6, 0.2, 0.9, 1.9],                               # :21,37
    reward_threshold=2200, preferred_lr=0.0005, noise_level=1e-3,              # :48-49,30
    dr_on_reset=True)                                                           # :145-146

HUMANOID = EnvSpec(                                   # random_envs/jinja/random_humanoid.py
    kind="humanoid",
    names=["mass%d" % i for i in range(13)] + ["damp%d" % i for i in range(1, 18)],   # :55-61
    search_bounds=[_M] * 13 + [(1, 10.0)] * 6 + [(.2, 5.0)] + [(1, 10.0)] * 3 + [(.2, 5.0)] * 7,   # :72-105 (damp7, damp11-17 -> (.2,5))
    lower_bounds=[0.2] * 13 + [0.8] * 6 + [.15] + [0.8] * 3 + [.15] * 7,          # :113-146
    # body_mass[1:] of humanoid.xml under MuJoCo 2.1.0 (capsules 1000*pi*r^2*(L+r)) + dof_damping[6:] (humanoid.xml:38-86)
    nominal_task=[8.322078939359361, 2.035752039526186, 5.852787113637785, 4.525556257747776, 2.6324944224829134,
                  1.7671458676442582, 4.525556257747776, 2.6324944224829134, 1.7671458676442582, 1.5940598415616263,
                  1.1983431305833825, 1.5940598415616263, 1.1983431305833825,
                  5, 5, 5, 5, 5, 5, 1, 5, 5, 5, 1, 1, 1, 1, 1, 1, 1],
    reward_threshold=2200, preferred_lr=0.0001, noise_level=1e-3,              # :63-64,39
    dr_on_reset=True)                                                           # :231-232

# ---- "Unmodeled" ids: a prefix of xi is frozen at 0.8x nominal and leaves the task vector ----------
HOPPER_UNMODELED = EnvSpec(                           # random_envs/jinja/random_hopper_unmodeled.py
    kind="hopper",
    names=["thighmass", "legmass", "footmass"],                                 # :39
    search_bounds=[_M] * 3,                                                     # :49-53
    lower_bounds=[0.001] * 3,                                                   # :61-65
    nominal_task=HOPPER.nominal_task[1:],                                       # :29 body_mass[2:]
    reward_threshold=1750, preferred_lr=0.0005, noise_level=0.0,               # :41-42 (no noisy option)
    dr_on_reset=True)                                                           # :110-111

HALFCHEETAH_UNMODELED = EnvSpec(                      # random_envs/jinja/random_half_cheetah_unmodeled.py
    kind="halfcheetah",
    names=["bfoot", "fthigh", "fshin", "ffoot", "friction"],                    # :43
    search_bounds=[_M] * 4 + [(0.1, 2.0)],                                      # :52-61
    lower_bounds=[0.1] * 4 + [0.02],                                            # :69-78
    nominal_task=HALFCHEETAH.nominal_task[3:],                                  # :33-34
    reward_threshold=4500, preferred_lr=0.0005, noise_level=0.0,               # :45-46
    dr_on_reset=True)

WALKER2D_UNMODELED = EnvSpec(                         # random_envs/jinja/random_walker2d_unmodeled.py
    kind="walker2d",
    names=["foot", "thigh_left", "leg_left", "foot_left", "thighsize", "legsize", "footsize",
           "friction_right", "friction_left"],                                  # :49
    search_bounds=[_M] * 4 + [(0.3, 1.0), (0.3, 1.0), (0.15, 0.8)] + [(0.1, 3.0)] * 2,   # :59-76
    lower_bounds=[0.1] * 4 + [0.25, 0.25, 0.12] + [0.05] * 2,                   # :84-100
    nominal_task=WALKER2D.nominal_task[3:7] + [0.45, 0.6, 0.2, 0.9, 1.9],       # :38-40
    reward_threshold=2200, preferred_lr=0.0005, noise_level=0.0,               # :51-52
    dr_on_reset=True)

HUMANOID_UNMODELED = EnvSpec(                         # random_envs/jinja/random_humanoid_unmodeled.py
    kind="humanoid",
    names=["mass%d" % i for i in range(4, 13)] + ["damp%d" % i for i in range(4, 18)],      # :63-69
    search_bounds=HUMANOID.search_bounds[4:13] + HUMANOID.search_bounds[16:],               # :88-120
    lower_bounds=HUMANOID.lower_bounds[4:13] + HUMANOID.lower_bounds[16:],                  # :129-162
    nominal_task=HUMANOID.nominal_task[4:13] + HUMANOID.nominal_task[16:],                  # :52-53
    reward_threshold=2200, preferred_lr=0.0001, noise_level=0.0,                            # :71-72
    dr_on_reset=True)

SPECS = {"cartpole": CARTPOLE, "hopper": HOPPER, "halfcheetah": HALFCHEETAH, "walker2d": WALKER2D, "humanoid": HUMANOID}
UNMODELED_SPECS = {"hopper": HOPPER_UNMODELED, "halfcheetah": HALFCHEETAH_UNMODELED, "walker2d": WALKER2D_UNMODELED,
                   "humanoid": HUMANOID_UNMODELED}

# gym ids registered by the reference: id -> (kind, kwargs)
IDS = {
    "RandomCartPole-v0": ("cartpole", {}),                       # random_cartpole.py:291-296
    "RandomHopper-v0": ("hopper", {}),                           # random_hopper.py:155-159
    "RandomHopperNoisy-v0": ("hopper", {"noisy": True}),         # random_hopper.py:161-166
    "RandomHalfCheetah-v0": ("halfcheetah", {}),                 # random_half_cheetah.py:161-165
    "RandomHalfCheetahNoisy-v0": ("halfcheetah", {"noisy": True}),   # :167-172
    "RandomWalker2d-v0": ("walker2d", {}),                       # random_walker2d.py:188-192
    "RandomWalker2dNoisy-v0": ("walker2d", {"noisy": True}),     # :194-199
    "RandomHumanoid-v0": ("humanoid", {}),                                   # random_humanoid.py:273-277
    "RandomHumanoidNoisy-v0": ("humanoid", {"noisy": True}),                 # :279-284
    "RandomHopperUnmodeled-v0": ("hopper", {"unmodeled": True}),             # random_hopper_unmodeled.py:146-150
    "RandomHalfCheetahUnmodeled-v0": ("halfcheetah", {"unmodeled": True}),   # random_half_cheetah_unmodeled.py:155-159
    "RandomWalker2dUnmodeled-v0": ("walker2d", {"unmodeled": True}),         # random_walker2d_unmodeled.py:187-191
    "RandomHumanoidUnmodeled-v0": ("humanoid", {"unmodeled": True}),         # random_humanoid_unmodeled.py:275-279
}
# ids of the reference not built yet: none (all 13 ids of SURVEY.md Appendix A are registered)
MAX_EPISODE_STEPS = 500
