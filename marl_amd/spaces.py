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
