"""Throughput of the replay-buffer kernels on one GPU (run on the GPU box from the repo root):
    python tools/replay_bench.py
add: one step of B envs of simple_tag's adversary 0 (record = 16 obs + 5 act + 1 + 16 obs + 1 = 39 columns)
appended from SoA row views; gather: a batch of rows by random index.  Bytes = read + written once."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.replay_buffer import ReplayBuffer  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    B, od, ad = 1 << 20, 16, 5
    ld = B + 256
    obs = torch.rand(od, ld, device='cuda')[:, :B].t()
    nxt = torch.rand(od, ld, device='cuda')[:, :B].t()
    act = torch.rand(ad, ld, device='cuda')[:, :B].t()
    rew, done = torch.rand(B, device='cuda'), torch.zeros(B, device='cuda')
    width = 2 * od + ad + 2
    rb = ReplayBuffer(8 * B, dtype=torch.float32)
    t = timed(lambda: rb.add(obs, act, rew, nxt, done), 20)
    print('add    %d transitions x %d columns: %8.1f us  %6.0f GB/s (read + write)' % (B, width, t * 1e6, 2 * B * width * 4 / t / 1e9))
    for batch in (1024, 1 << 16, 1 << 20):
        idx = rb.make_index(batch)
        t = timed(lambda: rb.sample_index(idx), 20)
        print('gather %8d rows x %d columns: %8.1f us  %6.0f GB/s (read + write)' % (batch, width, t * 1e6, 2 * batch * width * 4 / t / 1e9))


if __name__ == '__main__':
    main()
