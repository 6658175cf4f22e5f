"""Minimal gym-space look-alikes (the reference uses gym.spaces.Discrete / Box,
environment.py:47-79; gym is not a dependency here)."""
import numpy as np


class Space(object):
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def __repr__(self):
        return 'Discrete(%d)' % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = dtype

    def __repr__(self):
        return 'Box%s' % (self.shape,)

    def __eq__(self, other):
        return isinstance(other, Box) and other.shape == self.shape


class MultiDiscrete(Space):
    """multiagent/multi_discrete.py:9-44: a series of discrete spaces given as [[min, max], ...]
    (a non-silent agent's physical x comm action, environment.py:69)."""

    def __init__(self, array_of_param_array):
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.dtype = np.int64

    @property
    def shape(self):                       # multi_discrete.py:38-40 (an int, not a tuple)
        return self.num_discrete_space

    @property
    def n_entries(self):
        """Length of the flat soft one-hot action vector (environment.py:161-168)."""
        return int(np.sum(self.high - self.low + 1))

    def __repr__(self):
        return 'MultiDiscrete' + str(self.num_discrete_space)

    def __eq__(self, other):
        return (isinstance(other, MultiDiscrete) and np.array_equal(self.low, other.low)
                and np.array_equal(self.high, other.high))
