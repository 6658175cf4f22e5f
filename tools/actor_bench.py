"""Time of the fused actor forward kernel on one GPU (run on the GPU box from the repo root):
    python tools/actor_bench.py
simple_tag adversary: 16 observation rows in, 5 action rows out, 1 Mi envs; flop = 2 (16 64 + 64 64 + 64 8) per env
(the kernel evaluates 8 or 16 output columns)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.maddpg import actor_forward, mlp_model, pack_actor  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    ld = B + 256
    for in_dim, sizes in ((16, [5]), (34, [5, 4]), (28, [5, 7])):
        net = mlp_model(in_dim, sum(sizes)).cuda()
        packed = pack_actor(net)
        obs = torch.randn(in_dim, ld, device='cuda')[:, :B].t()
        for kernel, name in ((1, 'vector FMA'), (2, 'bf16x6 MFMA'), (3, 'f16x3 MFMA ')):
            for c in range(3):
                actor_forward(packed, obs, sizes, seed=1, call_counter=c, kernel=kernel)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in range(20):
                actor_forward(packed, obs, sizes, seed=1, call_counter=c, kernel=kernel)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e-3
            flop = 2.0 * B * (in_dim * 64 + 64 * 64 + 64 * sum(sizes))   # useful flop (the kernels evaluate 8 / 16 / 32 columns)
            print('actor %3d -> 64 -> 64 -> %-7s %d envs, %-12s: %7.1f us  %5.1f useful TFLOP/s f32  %5.0f GB/s' %
                  (in_dim, sizes, B, name, t * 1e6, flop / t / 1e12, 4.0 * B * (in_dim + sum(sizes)) / t / 1e9))


if __name__ == '__main__':
    main()
