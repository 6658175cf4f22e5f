"""Register kernels at batch sizes whose buffers pass 2^31 and 2^32 bytes (9 Mi envs: 2-10 GB of observation rows; the
Infinity Cache plan streams every row class there): one fused step of the whole batch against the same step of three
1 500-env slices (first, middle, last) taken as batches of their own -- envs are independent, so every output must agree
bit for bit.  Run on the GPU box from the repo root: python tools/big_batch_check.py [envs]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.mpe_oracle as O
from tests.gpu_util import HipSim, dev, set_call_flags
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
def run(spec, B, dtype, flags=0):
    A, E = spec.n_agents, spec.n_entities
    g = torch.Generator(device='cuda').manual_seed(1)
    sim = HipSim(spec, B, dtype)
    sim.pos[..., :B] = torch.rand(E, 2, B, generator=g, device='cuda', dtype=dtype) * 0.8 - 0.4
    sim.vel[..., :B] = torch.rand(A, 2, B, generator=g, device='cuda', dtype=dtype) - 0.5
    if sim.lforce is not None:
        sim.lforce[..., :B] = (torch.rand(A, 2, B, generator=g, device='cuda', dtype=dtype) - 0.5) * torch.tensor(np.asarray(spec.forced_id) >= 0, device='cuda', dtype=dtype)[:, None, None]
    act = torch.rand(sum(spec.act_dims), B, generator=g, device='cuda', dtype=dtype)
    comm = torch.zeros(A, max(spec.dim_c, 1), B, dtype=dtype, device='cuda')[:, :spec.dim_c] if spec.dim_c else None
    p0, v0 = sim.pos.clone(), sim.vel.clone()
    lf0 = sim.lforce.clone() if sim.lforce is not None else None
    set_call_flags(flags)
    a3 = act.view(A, 5, B) if all(spec.silent) else act
    sim.env_step(a3, comm=comm)
    set_call_flags(0)
    torch.cuda.synchronize()
    bad = 0
    for lo in (0, B // 2 - 700, B - 1500):
        n = 1500
        s2 = HipSim(spec, n, dtype)
        s2.pos.copy_(p0[..., lo:lo + n]); s2.vel.copy_(v0[..., lo:lo + n])
        if lf0 is not None: s2.lforce.copy_(lf0[..., lo:lo + n])
        a2 = act[:, lo:lo + n].contiguous()
        c2 = torch.zeros(A, max(spec.dim_c, 1), n, dtype=dtype, device='cuda')[:, :spec.dim_c] if spec.dim_c else None
        set_call_flags(2)
        s2.env_step(a2.view(A, 5, n) if all(spec.silent) else a2, comm=c2)
        set_call_flags(0)
        for nm, x, y in zip(('pos', 'vel', 'obs', 'rew', 'info'), (sim.pos, sim.vel, sim.obs, sim.rew, sim.info), (s2.pos, s2.vel, s2.obs, s2.rew, s2.info)):
            xx = x[..., lo:lo + n]
            eq = torch.equal(torch.nan_to_num(xx, nan=7.0), torch.nan_to_num(y[..., :n], nan=7.0))
            if not eq:
                bad += 1
                print('   DIFF', nm, 'slice at', lo, float((xx - y[..., :n]).abs().nan_to_num().max()))
        if comm is not None and not torch.equal(comm[..., lo:lo + n], c2):
            bad += 1; print('   DIFF comm at', lo)
    return bad

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 9 * 1048576 + 77
    for name, spec, dtype in (('simple_tag f32', O.simple_tag_spec(), torch.float32), ('simple_spread f32', O.simple_spread_spec(), torch.float32),
                              ('simple_world_comm f32', O.simple_world_comm_spec(), torch.float32),
                              ('S1L2F2_10Targets f32', O.hier_spec_from_golden(np.load(gd + '/h_S1L2F2_10Targets_wide.npz')), torch.float32),
                              ('S1L2F4_MovingTarget f32', O.hier_spec_from_golden(np.load(gd + '/m_S1L2F4_MovingTarget.npz')), torch.float32),
                              ('simple_tag f64', O.simple_tag_spec(), torch.float64)):
        rows = sum(spec.obs_dims)
        print(name, 'B', B, 'obs bytes %.2f GB' % (rows * B * (4 if dtype == torch.float32 else 8) / 1e9), flush=True)
        print('   mismatching slices:', run(spec, B, dtype), flush=True)
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
