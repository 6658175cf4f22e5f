"""One actor shape, one kernel, 20 launches: the program to put under rocprofv3 (tools/actor_bench.py times them).
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU ... -- python3 tools/actor_prof.py [kernel=2] [in_dim=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.maddpg import actor_forward, mlp_model, pack_actor  # noqa: E402

kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 2
in_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B, ld = 1 << 20, (1 << 20) + 256
net = mlp_model(in_dim, 5).cuda()
packed = pack_actor(net)
obs = torch.randn(in_dim, ld, device='cuda')[:, :B].t()
for c in range(20):
    actor_forward(packed, obs, [5], seed=1, call_counter=c, kernel=kernel)
torch.cuda.synchronize()
