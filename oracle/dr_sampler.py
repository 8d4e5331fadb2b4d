"""ORACLE (test infrastructure, NOT product code).

numpy restatement of RandomEnv.sample_task (random_envs/random_env.py:148-203).  For `uniform`,
`gaussian` and `fullgaussian` it makes the same numpy calls in the same order as the reference,
so under a fixed np.random.seed it is bit-identical to the golden vectors in
tests/golden/dr_sampler.json.  `truncnorm` restates the INTENDED semantics (the reference raises
NameError there).
"""
import numpy as np
from scipy.stats import truncnorm


def sample_task(sampling, min_task=None, max_task=None, mean_task=None, stdev_task=None, cov_task=None,
                lower_bounds=None, search_bounds=None):
    if sampling == 'uniform':                                   # :150-151
        return np.random.uniform(min_task, max_task, np.shape(min_task))
    if sampling == 'truncnorm':                                 # :153-171
        a, b = -2, 2
        sample = []
        for i, (mean, std) in enumerate(zip(mean_task, stdev_task)):
            lower_bound = lower_bounds[i]
            attempts = 0
            obs = truncnorm.rvs(a, b, loc=mean, scale=std)
            while obs < lower_bound:
                obs = truncnorm.rvs(a, b, loc=mean, scale=std)
                attempts += 1
                if attempts > 2:
                    obs = lower_bound
            sample.append(obs)
        return np.array(sample)
    if sampling == 'gaussian':                                  # :173-190
        sample = []
        for mean, std in zip(mean_task, stdev_task):
            attempts = 0
            obs = np.random.randn() * std + mean
            while obs < 0.1:
                obs = np.random.randn() * std + mean
                attempts += 1
                if attempts > 2:
                    raise Exception('Not all samples were above > 0.1 after 2 attempts')
            sample.append(obs)
        return np.array(sample)
    if sampling == 'fullgaussian':                              # :192-198
        sample = np.random.multivariate_normal(mean_task, cov_task)
        sample = np.clip(sample, 0, 4)
        lo, hi = np.asarray(search_bounds[0]), np.asarray(search_bounds[1])
        return (sample * (hi - lo)) / 4 + lo                    # denormalize_parameters :205-220
    raise ValueError('sampling value of random env needs to be set before using sample_task() or '
                     'set_random_task(). Set it by uploading a DR distr.')
