"""Host-side DR configuration: RandomEnv's distribution state and bookkeeping methods
(random_envs/random_env.py:10-143,205-259), independent of any GPU handle so that it is testable
on CPU against golden vectors generated from the reference."""
import csv

import numpy as np


class DRConfig:
    def __init__(self, spec):
        self.spec = spec
        self.task_dim = len(spec.names)
        # RandomEnv.__init__ (random_env.py:10-15)
        self.sampling = None
        self.dr_training = False
        self.endless = False
        self.noise_level = spec.noise_level
        self.preferred_lr = spec.preferred_lr
        self.reward_threshold = spec.reward_threshold
        self.dyn_ind_to_name = dict(enumerate(spec.names))
        self.min_task = np.zeros(self.task_dim); self.max_task = np.zeros(self.task_dim)
        self.mean_task = np.zeros(self.task_dim); self.stdev_task = np.zeros(self.task_dim)
        self.cov_task = None
        self.original_task = np.array(spec.nominal_task, dtype=np.float64)

    # hooks overridden by VecRandomEnv
    def _push_dr(self):
        pass

    def _push_flags(self):
        pass

    def _set_dr_training_native(self):
        pass

    def get_search_bounds_mean(self, index):
        return self.spec.search_bounds[index]

    def get_task_lower_bound(self, index):
        return self.spec.lower_bounds[index]

    def set_dr_training(self, flag):          # random_env.py:41-46
        self.dr_training = bool(flag)
        self._set_dr_training_native()

    def get_dr_training(self):
        return self.dr_training

    def set_endless(self, flag):              # random_env.py:51-60
        self.endless = bool(flag)
        self._push_flags()

    def get_endless(self):
        return self.endless

    def get_reward_threshold(self):
        return self.reward_threshold

    def dyn_index_to_name(self, index):
        assert self.dyn_ind_to_name is not None
        return self.dyn_ind_to_name[index]

    def set_dr_distribution(self, dr_type, distr):   # random_env.py:72-90
        if dr_type == 'uniform':
            self._set_interleaved('uniform', distr, self.min_task, self.max_task)
        elif dr_type == 'truncnorm':
            self._set_interleaved('truncnorm', distr, self.mean_task, self.stdev_task)
        elif dr_type == 'gaussian':
            self._set_interleaved('gaussian', distr, self.mean_task, self.stdev_task)
        elif dr_type == 'fullgaussian':
            self.sampling = 'fullgaussian'
            self.mean_task[:] = distr['mean']
            self.cov_task = np.copy(distr['cov'])
        else:
            raise Exception('Unknown dr_type:' + str(dr_type))
        self._push_dr()

    def _set_interleaved(self, name, bounds, a, b):  # random_env.py:102-121
        self.sampling = name
        for i in range(len(bounds) // 2):
            a[i] = bounds[i * 2]
            b[i] = bounds[i * 2 + 1]

    def get_dr_distribution(self):            # random_env.py:92-100
        if self.sampling == 'uniform':
            return self.min_task, self.max_task
        elif self.sampling == 'truncnorm':
            return self.mean_task, self.stdev_task
        elif self.sampling == 'gaussian':
            raise ValueError('Not implemented')
        return None

    def set_task_search_bounds(self):         # random_env.py:129-134
        for i in range(self.task_dim):
            self.min_task[i], self.max_task[i] = self.get_search_bounds_mean(i)

    def get_task_search_bounds(self):         # random_env.py:136-143
        b = np.array(self.spec.search_bounds, dtype=np.float64)
        return b[:, 0].copy(), b[:, 1].copy()

    def denormalize_parameters(self, parameters):   # random_env.py:205-220
        parameters = np.asarray(parameters)
        assert parameters.shape[0] == self.task_dim
        lo, hi = self.get_task_search_bounds()
        return (parameters * (hi - lo)) / 4 + lo

    def load_dr_distribution_from_file(self, filename):   # random_env.py:222-259 (with the missing `import csv`)
        with open(filename, 'r', encoding='utf-8') as file:
            reader = csv.reader(file, delimiter=',')
            dr_type = str(next(reader)[0])
            bounds = [float(col) for col in next(reader)]
        if len(bounds) != self.task_dim * 2:
            raise Exception('The file did not contain the right number of column values')
        if dr_type not in ('uniform', 'truncnorm', 'gaussian'):
            raise Exception('Filename is wrongly formatted: ' + str(filename))
        self.set_dr_distribution(dr_type, bounds)

