"""CPU restatement of the reference's replay buffer (maddpg/maddpg/trainer/replay_buffer.py), the first
slice of SURVEY 8f row f4.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).

*** parity unpinned ***  The reference holds no test or fixture for this class, and importing the file to
generate golden vectors was refused in this container.  The class below follows the
reference text method by method (line numbers cited); the HIP path is checked against it.
"""
import random

import numpy as np


class ReplayBuffer(object):
    def __init__(self, size):                       # replay_buffer.py:5-16
        self._storage = []
        self._maxsize = int(size)
        self._next_idx = 0

    def __len__(self):                              # :18-19
        return len(self._storage)

    def clear(self):                                # :21-23
        self._storage = []
        self._next_idx = 0

    def add(self, obs_t, action, reward, obs_tp1, done):        # :25-32
        data = (obs_t, action, reward, obs_tp1, done)
        if self._next_idx >= len(self._storage):
            self._storage.append(data)
        else:
            self._storage[self._next_idx] = data
        self._next_idx = (self._next_idx + 1) % self._maxsize

    def add_batch(self, obs_t, action, reward, obs_tp1, done):
        """A step of B envs = B add() calls in env order (what MADDPGAgentTrainer.experience, maddpg.py:154-156,
        would do if the training loop ran the envs one after the other)."""
        for e in range(len(reward)):
            self.add(obs_t[e], action[e], reward[e], obs_tp1[e], float(done[e]))

    def _encode_sample(self, idxes):                # :34-44
        obses_t, actions, rewards, obses_tp1, dones = [], [], [], [], []
        for i in idxes:
            obs_t, action, reward, obs_tp1, done = self._storage[i]
            obses_t.append(np.asarray(obs_t))
            actions.append(np.asarray(action))
            rewards.append(reward)
            obses_tp1.append(np.asarray(obs_tp1))
            dones.append(done)
        return np.array(obses_t), np.array(actions), np.array(rewards), np.array(obses_tp1), np.array(dones)

    def make_index(self, batch_size):               # :46-47
        return [random.randint(0, len(self._storage) - 1) for _ in range(batch_size)]

    def make_latest_index(self, batch_size):        # :49-52
        idx = [(self._next_idx - 1 - i) % self._maxsize for i in range(batch_size)]
        np.random.shuffle(idx)
        return idx

    def sample_index(self, idxes):                  # :54-55
        return self._encode_sample(idxes)

    def sample(self, batch_size):                   # :57-83
        if batch_size > 0:
            idxes = self.make_index(batch_size)
        else:
            idxes = range(0, len(self._storage))
        return self._encode_sample(idxes)

    def collect(self):                              # :85-86
        return self.sample(-1)
