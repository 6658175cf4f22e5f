"""Fused env step vs world_step + observe as two launches (register pressure vs re-reading the state), 1 Mi envs:
    python tools/split_vs_fused.py <fixture name for the spec | tag | wcomm> <f32|f64>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpe_oracle as O  # noqa: E402
from tests.gpu_util import HipSim  # noqa: E402


def main():
    which, dt = sys.argv[1], torch.float64 if sys.argv[2] == 'f64' else torch.float32
    if which == 'tag':
        spec = O.simple_tag_spec()
    elif which == 'wcomm':
        spec = O.simple_world_comm_spec()
    else:
        spec = O.hier_spec_from_golden(np.load(os.path.join('tests', 'golden', which + '.npz')))
    B = 1 << 20
    ld = B + (256 if dt == torch.float32 else 128)
    sim = HipSim(spec, B, dt, ld=ld)
    sim.pos.uniform_(-1, 1)
    act = torch.rand(sum(spec.act_dims), ld, dtype=dt, device='cuda')
    act = act.view(spec.n_agents, 5, ld) if spec.act_dims == [5] * spec.n_agents else act
    comm = torch.zeros(spec.n_agents, max(spec.dim_c, 1), ld, dtype=dt, device='cuda') if spec.dim_c and not all(spec.silent) else None

    def timed(fn, n=60):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def reset_state():
        sim.pos.uniform_(-1, 1)
        sim.vel.zero_()
    fused = timed(lambda: sim.env_step(act, comm=comm))
    reset_state()
    step = timed(lambda: sim.world_step(act, comm=comm))
    reset_state()
    obs = timed(lambda: sim.observe(comm=comm))
    print('%s %s: fused %.1f us   world_step %.1f + observe %.1f = %.1f us' % (which, sys.argv[2], fused, step, obs, step + obs))


if __name__ == '__main__':
    main()
