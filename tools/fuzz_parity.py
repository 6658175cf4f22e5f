"""Differential fuzz of the HIP step against the CPU oracle on worlds no scenario file builds: random agent / landmark counts
and random per-entity constants (sizes, masses, accel None or a value, max_speed None or a value, collide flags in
irregular patterns, u_noise on some agents, damping / dt / contact constants), dense starts so that contacts happen.
f64, every output of 3 steps within 1e-9 (NaN patterns equal).  Run on the GPU box from the repo root:
    python tools/fuzz_parity.py [n_worlds] [seed] [stock|moving] [grid]   (stock: only the shapes that have register kernels; moving: only random *_MovingTarget-like worlds;
    grid: entities start on a grid with sizes and radii on the same grid -- distances EXACTLY at the contact / visibility thresholds, exact ties in the
    target order, coincident entities whose NaN forces spread)
Prints one line per mismatch or refusal and a summary; exit status 1 if anything differs."""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpe_oracle as O  # noqa: E402
from marl_amd import _hip  # noqa: E402
from tests.gpu_util import HipSim, dev, set_call_flags  # noqa: E402
from tests.gpu_util_host import desc_from_spec  # noqa: E402,F401


def random_hier(rng):
    """A command hierarchy no scenario file builds: random counts, random who-commands-whom (several leaders on one
    forced_id, silent leaders, a super-leader or none -- never sharing a force_id with a plain leader, which both sides
    refuse), 3 or 7 commands, random radii / sizes / masses / speed limits / collide flags, both reward texts."""
    A, L = int(rng.randint(2, 9)), int(rng.randint(1, 12))
    pick = lambda p: rng.rand() < p
    n_lead = int(rng.randint(1, max(2, A // 2 + 1)))
    lead = [i >= A - n_lead for i in range(A)]
    sup = [lead[i] and i == A - 1 and n_lead > 1 and pick(0.5) for i in range(A)]
    ids = [1, 2, 3]
    force_id, forced_id = [-1] * A, [-1] * A
    for i in range(A):
        if lead[i] and not sup[i]:
            force_id[i] = int(rng.choice(ids[:2]))
            if pick(0.3):
                forced_id[i] = 3                     # a leader under the super-leader
        elif sup[i]:
            force_id[i] = 3
        elif pick(0.8):
            forced_id[i] = int(rng.choice(ids[:2]))
    silent = [not lead[i] or pick(0.15) for i in range(A)]
    if all(silent):
        silent[A - 1] = False
    return 'hier', O.ScenarioSpec(
        kind=O.HIER_PURSUIT, n_agents=A, n_landmarks=L, adversary=[pick(0.85) for _ in range(A)],
        size=[float(rng.uniform(0.04, 0.12)) for _ in range(A + L)],
        accel=[None if pick(0.2) else float(rng.uniform(2, 5)) for _ in range(A)],
        max_speed=[None if pick(0.2) else float(rng.uniform(0.5, 1.5)) for _ in range(A)],
        collide=[bool(pick(0.8)) for _ in range(A + L)], movable=[True] * A + [False] * L, u_noise=[None] * A,
        mass=[float(rng.choice([1.0, 0.5, 2.0])) for _ in range(A + L)], agent_pos_range=1.0, landmark_pos_range=1.0,
        dim_c=int(rng.choice([3, 7])), silent=silent, force_id=force_id, forced_id=forced_id, advleader=lead,
        advsuperleader=sup, obs_r=[float(rng.uniform(0.15, 0.8)) for _ in range(A)],
        reward_variant=int(rng.randint(0, 2)), n_target_obs=3)


_STOCK = ['h_L1F2_3Targets_100observe', 'h_L1F4_3Targets_wide', 'h_L1F1_10Targets_wide', 'h_L1F2_10Targets_wide',
          'h_S1L1F1_10Targets_wide', 'h_S1L2F2_3Targets_100observe', 'h_S1L2F2_10Targets_wide', 'm_L1F4_MovingTarget',
          'm_S1L2F4_MovingTarget']
_stock_cache = {}


def random_stock_hier(rng):
    """One of the nine shapes that have register kernels (mpe_launch.h: HIER_* / MOVING_*), with everything those
    kernels read at run time drawn at random: sizes, masses, accel, speed limits, observation radii, who commands whom
    (the followers' and leaders' forced_id, the leaders' force_id), the reward text, the world constants.  What the
    shape fixes stays: counts, who speaks, dim_c, every entity collides, every agent has a max_speed."""
    name = _STOCK[int(rng.randint(len(_STOCK)))]
    if name not in _stock_cache:
        g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', name + '.npz'))
        _stock_cache[name] = O.hier_spec_from_golden(g)
    spec = _stock_cache[name]
    A, E = spec.n_agents, spec.n_entities
    pick = lambda p: rng.rand() < p
    force_id, forced_id = list(spec.force_id), list(spec.forced_id)
    lead = [not spec.silent[i] for i in range(A)]
    sup = list(spec.advsuperleader)
    if pick(0.7):
        sup_id = [force_id[i] for i in range(A) if sup[i]]          # never shared with a plain leader (both sides refuse)
        pool = [k for k in (1, 2, 3, 4) if k not in sup_id][:2]
        for i in range(A):
            if lead[i] and not sup[i]:
                force_id[i] = int(rng.choice(pool))
                forced_id[i] = sup_id[0] if (sup_id and pick(0.6)) else -1
            elif not lead[i]:
                forced_id[i] = int(rng.choice([-1] + pool))
    rep = dict(force_id=force_id, forced_id=forced_id,
               size=[float(rng.uniform(0.03, 0.15)) for _ in range(E)],
               mass=[float(rng.choice([1.0, 1.0, 0.5, 2.5])) for _ in range(E)],
               accel=[None if pick(0.15) else float(rng.uniform(2, 5)) for _ in range(A)],
               max_speed=[float(rng.uniform(0.3, 1.5)) for _ in range(A)],
               obs_r=[float(rng.uniform(0.1, 1.2)) for _ in range(A)], reward_variant=int(rng.randint(0, 2)))
    if pick(0.4):
        rep.update(damping=float(rng.uniform(0.0, 0.5)), dt=float(rng.uniform(0.05, 0.2)),
                   contact_force=float(rng.uniform(20, 200)), contact_margin=float(10 ** rng.uniform(-3, -2)))
    return 'stock_hier', dataclasses.replace(spec, **rep)


def random_stock_small(rng):
    """A stock scenario in the shape its register kernel is built for (mpe_small_kernel.h / mpe_comm_kernel.h /
    mpe_wcomm_kernel.h) with the constants those kernels read at run time drawn at random: sizes, masses, accel, the
    values of the speed limits (who has one is part of the shape), u_noise, the world constants."""
    mk = [O.simple_tag_spec, O.simple_spread_spec, O.simple_spec, O.simple_adversary_spec, O.simple_push_spec,
          O.speaker_listener_spec, O.simple_reference_spec, O.simple_crypto_spec, O.simple_world_comm_spec]
    spec = mk[int(rng.randint(len(mk)))]()
    A, E = spec.n_agents, spec.n_entities
    pick = lambda p: rng.rand() < p
    rep = dict(size=[float(rng.uniform(0.02, 0.25)) for _ in range(E)],
               mass=[float(rng.choice([1.0, 1.0, 0.5, 2.5])) for _ in range(E)],
               accel=[None if pick(0.2) else float(rng.uniform(1, 6)) for _ in range(A)],
               max_speed=[None if m is None else float(rng.uniform(0.3, 1.5)) for m in spec.max_speed])
    if spec.kind in (O.SIMPLE_TAG, O.SIMPLE_SPREAD) and pick(0.4):
        rep['u_noise'] = [None if pick(0.4) else float(rng.uniform(0.05, 0.5)) for _ in range(A)]
    if pick(0.4):
        rep.update(damping=float(rng.uniform(0.0, 0.5)), dt=float(rng.uniform(0.05, 0.2)),
                   contact_force=float(rng.uniform(20, 200)), contact_margin=float(10 ** rng.uniform(-3, -2)))
    return 'stock_small', dataclasses.replace(spec, **rep)


def random_moving(rng):
    """A *_MovingTarget-like world no scenario file builds (kind HIER_MOVING): silent followers, speaking leaders (one of
    them possibly a super-leader), one prey as the last agent, no landmarks; random counts, links, radii, constants,
    irregular collide flags, 3 or 7 commands, both files' observation / reward texts."""
    A = int(rng.randint(3, 10))
    pick = lambda p: rng.rand() < p
    n_lead = int(rng.randint(1, max(2, (A - 1) // 2 + 1)))
    G = A - 1
    lead = [G - n_lead <= i < G for i in range(A)]
    sup = [lead[i] and i == G - 1 and n_lead > 1 and pick(0.5) for i in range(A)]
    force_id, forced_id = [-1] * A, [-1] * A
    for i in range(G):
        if lead[i] and not sup[i]:
            force_id[i] = int(rng.choice([1, 2]))
            if any(sup) and pick(0.5):
                forced_id[i] = 3
        elif sup[i]:
            force_id[i] = 3
        elif pick(0.8):
            forced_id[i] = int(rng.choice([1, 2]))
    return 'moving', O.ScenarioSpec(
        kind=O.HIER_MOVING, n_agents=A, n_landmarks=0, adversary=[i < G for i in range(A)],
        size=[float(rng.uniform(0.04, 0.15)) for _ in range(A)],
        accel=[None if pick(0.2) else float(rng.uniform(2, 5)) for _ in range(A)],
        max_speed=[None if pick(0.2) else float(rng.uniform(0.5, 1.5)) for _ in range(A)],
        collide=[bool(pick(0.85)) for _ in range(A)], movable=[True] * A, u_noise=[None] * A,
        mass=[float(rng.choice([1.0, 0.5, 2.0])) for _ in range(A)], agent_pos_range=1.0, landmark_pos_range=1.0,
        dim_c=int(rng.choice([3, 7])), silent=[not lead[i] for i in range(A)], force_id=force_id, forced_id=forced_id,
        advleader=lead, advsuperleader=sup, obs_r=[float(rng.uniform(0.15, 0.9)) for _ in range(A)],
        reward_variant=int(rng.randint(0, 2)), n_target_obs=3)


def random_spec(rng, stock_only=False):
    if stock_only == 'moving':
        return random_moving(rng)
    if stock_only and rng.rand() < 0.5:
        return random_stock_small(rng)
    kind = 'stock_hier' if stock_only else rng.choice(['tag', 'spread', 'simple', 'adversary', 'push', 'hier', 'hier',
                                                       'wcomm', 'big', 'stock_hier', 'moving'])
    if kind == 'hier':
        return random_hier(rng)
    if kind == 'moving':
        return random_moving(rng)
    if kind == 'stock_hier':
        return random_stock_hier(rng)
    if kind == 'tag':
        spec = O.simple_tag_spec(int(rng.randint(1, 7)), int(rng.randint(1, 5)), int(rng.randint(0, 6)))
    elif kind == 'spread':
        spec = O.simple_spread_spec(int(rng.randint(1, 7)), int(rng.randint(1, 7)))
    elif kind == 'big':       # up to the library's limits (64 agents, 128 entities): several agent chunks per tile
        A = int(rng.randint(20, 65))
        L = int(rng.randint(0, 129 - A))
        n_good = int(rng.randint(1, A))
        spec = O.simple_tag_spec(A - n_good, n_good, L) if rng.rand() < 0.6 else O.simple_spread_spec(A, max(L, 1))
    elif kind == 'wcomm':     # simple_world_comm.py: tag physics, food, two forests, a speaking leader (agent 0)
        spec = O.simple_world_comm_spec(int(rng.randint(1, 6)), int(rng.randint(1, 4)), int(rng.randint(0, 3)),
                                        int(rng.randint(1, 4)), 2)
    elif kind == 'simple':
        spec = O.simple_spec()
    elif kind == 'adversary':
        spec = O.simple_adversary_spec(int(rng.randint(2, 7)))
    else:
        n = int(rng.randint(2, 7))
        spec = O.simple_push_spec(n, int(rng.randint(1, n)), int(rng.randint(1, 3)))
    A, E = spec.n_agents, spec.n_entities
    pick = lambda p: rng.rand() < p
    rep = {}
    if pick(0.7):
        rep['size'] = [float(rng.uniform(0.02, 0.3)) for _ in range(E)]
    if pick(0.5):
        rep['mass'] = [float(rng.uniform(0.5, 3.0)) for _ in range(E)]
    if pick(0.6):
        rep['accel'] = [None if pick(0.3) else float(rng.uniform(1.0, 6.0)) for _ in range(A)]
    if pick(0.6):
        rep['max_speed'] = [None if pick(0.4) else float(rng.uniform(0.2, 2.0)) for _ in range(A)]
    if pick(0.6):
        rep['collide'] = [bool(pick(0.6)) for _ in range(E)]
    if pick(0.4):
        rep['u_noise'] = [None if pick(0.5) else float(rng.uniform(0.05, 0.5)) for _ in range(A)]
    if kind != 'wcomm' and spec.n_landmarks and pick(0.3):
        # movable landmarks (core.py:211-222; no scenario file has one): some of them, some with a max_speed
        rep['movable'] = [True] * A + [bool(pick(0.6)) for _ in range(E - A)]
        rep['landmark_max_speed'] = [None if pick(0.5) else float(rng.uniform(0.05, 1.0)) for _ in range(E - A)]
    if pick(0.4):
        rep.update(damping=float(rng.uniform(0.0, 0.5)), dt=float(rng.uniform(0.05, 0.2)),
                   contact_force=float(rng.uniform(20, 200)), contact_margin=float(10 ** rng.uniform(-3, -2)))
    return kind, dataclasses.replace(spec, **rep)


def run(n_worlds, seed, perturb=0.0, stock_only=False, progress=0, grid=False):
    """-> (mismatching runs, worlds refused by mpe_desc_check, worlds per kind).  perturb != 0: the oracle steps a world
    whose damping differs by that much (negative control: the comparison must notice)."""
    rng = np.random.RandomState(seed)
    B, T = int(os.environ.get('FUZZ_ENVS', 257)), 3      # FUZZ_ENVS=4099: 16x the envs per world (rare events: exact f32 ties, ...)
    bad = refused = 0
    kernels = {}
    for w in range(n_worlds):
        kind, spec = random_spec(rng, stock_only)
        d = desc_from_spec_or_none(spec)
        if d is None:
            refused += 1
            continue
        A, E = spec.n_agents, spec.n_entities
        if kind == 'stock_hier':  # the random constants must not push the world off its register kernels
            rows = [_hip.lib.mpe_stats_rows(d, B, 8, f) for f in (0, 2)]
            assert rows == [-(-B // 64), -(-B // 256)], (rows, spec)
        if kind == 'stock_small':
            splits = spec.kind in (O.SIMPLE_TAG, O.SIMPLE_SPREAD)
            on_tile = spec.kind == O.SIMPLE_PUSH     # 10 agents in f64: more than 256 VGPRs, the tile kernel takes it
            rows = [_hip.lib.mpe_stats_rows(d, B, 8, f) for f in (0, 2)]
            assert on_tile or rows == [-(-B // (64 if splits else 256)), -(-B // 256)], (rows, spec)
        pos = rng.uniform(-0.45, 0.45, (E, 2, B))
        vel = rng.uniform(-0.8, 0.8, (A, 2, B))
        if grid:   # cells of 1/8 (1/32 in a crowded world); sizes in half cells, so that centre distances hit size sums exactly
            cell = 0.125 if E <= 12 else 0.03125
            n = int(round(0.5 / cell))
            pos = rng.randint(-n, n + 1, (E, 2, B)) * cell
            vel = rng.randint(-6, 7, (A, 2, B)) * 0.125
            rep = dict(size=[float(rng.randint(1, 4)) * cell / 2 for _ in range(E)])
            if spec.obs_r is not None:
                rep['obs_r'] = [float(rng.randint(1, 9)) * cell if r else r for r in spec.obs_r]
            spec = dataclasses.replace(spec, **rep)
            d = desc_from_spec_or_none(spec)
            assert d is not None
        lm_moves = any(spec.movable[A:])
        lvel0 = rng.uniform(-0.8, 0.8, (E - A, 2, B)) * np.asarray(spec.movable[A:], float)[:, None, None] if lm_moves else None
        gr = spec.goal_rows                  # per-env goal landmark(s): one row, or one per agent (simple_reference)
        goal = (rng.randint(0, max(spec.n_landmarks, 1), B) if spec.kind in O.GOAL_KINDS or gr == 0
                else rng.randint(0, spec.n_landmarks, (gr, B)))
        goal_arg = goal if gr else None
        hier = spec.kind in (O.HIER_PURSUIT, O.HIER_MOVING)
        ragged = hier or not all(spec.silent)
        lf0 = rng.uniform(-1, 1, (A, 2, B)) if hier and spec.has_leader_force else None
        if lf0 is not None:   # nobody writes the leader_force of an agent no leader commands: it keeps make_world's zeros
            lf0 = lf0 * (np.asarray(spec.forced_id) >= 0)[:, None, None]
        acts = [rng.rand(sum(spec.act_dims), B) if ragged else rng.dirichlet(np.ones(5), (A, B)).transpose(0, 2, 1)
                for _ in range(T)]
        noises = [rng.randn(A, 2, B) for _ in range(T)]
        has_noise = any(spec.u_noise)
        bo = O.BatchedOracle(dataclasses.replace(spec, damping=spec.damping + perturb) if perturb else spec)
        # 0: the kernel the library picks (a register kernel for the stock shapes: one agent per wavefront at this batch
        # size), 1: MPE_FLAG_FORCE_TILE, 2: MPE_FLAG_WIDE (one env per lane; the stock shapes only)
        for flags in ((0, 1, 2) if kind in ('stock_hier', 'stock_small') else (0, 1)):
            set_call_flags(flags)
            try:
                sim = HipSim(spec, B, torch.float64)
                sim.set_state(pos, vel, goal, lf0)
                if lm_moves:
                    sim.lvel.copy_(dev(lvel0, torch.float64))
                p, v, lv = pos, vel, lvel0
                if grid:   # the gathers ON the grid state (a step leaves it): thresholds met exactly, ties in the target order
                    comm0 = np.zeros((A, spec.dim_c, B)) if spec.dim_c else None
                    wobs, wrew, winfo = bo.observe(pos, vel, goal=goal_arg, lforce=lf0, comm=comm0)
                    sim.observe(comm=dev(comm0, torch.float64) if spec.dim_c else None)
                    for nm, x, y in zip(('obs', 'rew', 'info'), sim.host()[2:], (np.concatenate(wobs, 0), wrew, winfo)):
                        if y is None or x.size == 0:
                            continue
                        y = np.asarray(y, np.float64).reshape(x.shape)
                        if not (np.array_equal(np.isnan(x), np.isnan(y)) and np.nanmax(np.abs(x - y), initial=0.0) <= 1e-9):
                            bad += 1
                            ne = ~(np.isclose(x, y, rtol=0, atol=1e-9) | (np.isnan(x) & np.isnan(y))).reshape(-1, B)
                            print('GRID OBSERVE MISMATCH world %d (%s A=%d L=%d) flags=%d %s: rows %s, %d envs   spec=%r'
                                  % (w, kind, A, spec.n_landmarks, flags, nm, np.nonzero(ne.any(1))[0].tolist(), int(ne.any(0).sum()), spec), flush=True)
                            raise StopIteration
                for t in range(T):
                    sim.env_step(dev(acts[t], torch.float64), dev(noises[t], torch.float64) if has_noise else None)
                    res = bo.env_step(p, v, acts[t], noises[t] if has_noise else None,
                                      goal=goal_arg, lvel=lv)
                    p, v, obs, rew, info = res[:5]
                    lv = res[5] if lm_moves else None
                    got = sim.host() + ((sim.lvel.cpu().numpy(),) if lm_moves else ())
                    want = (p, v, np.concatenate(obs, 0), rew, info) + ((lv,) if lm_moves else ())
                    for nm, x, y in zip(('pos', 'vel', 'obs', 'rew', 'info', 'lvel'), got, want):
                        if y is None or x.size == 0:
                            continue
                        y = np.asarray(y, np.float64).reshape(x.shape)
                        ok = np.array_equal(np.isnan(x), np.isnan(y)) and np.nanmax(np.abs(x - y), initial=0.0) <= 1e-9
                        if not ok:
                            bad += 1
                            err = np.nanmax(np.abs(x - y), initial=0.0)
                            if not perturb:
                                print('MISMATCH world %d (%s A=%d L=%d) flags=%d step %d %s: max err %.3e nan %d/%d   spec=%r'
                                      % (w, kind, A, spec.n_landmarks, flags, t, nm, err, np.isnan(x).sum(), np.isnan(y).sum(), spec),
                                      flush=True)
                            raise StopIteration
                # the split entry points on the same world: mpe_world_step + mpe_observe == mpe_env_step bit for bit
                parts = HipSim(spec, B, torch.float64)
                parts.set_state(pos, vel, goal, lf0)
                if lm_moves:
                    parts.lvel.copy_(dev(lvel0, torch.float64))
                comm = torch.zeros(A, max(spec.dim_c, 1), B, dtype=torch.float64, device='cuda')[:, :spec.dim_c]
                for t in range(T):
                    parts.world_step(dev(acts[t], torch.float64), dev(noises[t], torch.float64) if has_noise else None,
                                     comm=comm if spec.dim_c else None)
                    parts.observe(comm=comm if spec.dim_c else None)
                for nm, x, y in zip(('pos', 'vel', 'obs', 'rew', 'info'), sim.host(), parts.host()):
                    if not np.array_equal(x, y, equal_nan=True):
                        bad += 1
                        if not perturb:
                            print('SPLIT ENTRY POINTS DIFFER world %d (%s A=%d L=%d) flags=%d %s   spec=%r'
                                  % (w, kind, A, spec.n_landmarks, flags, nm, spec), flush=True)
                        break
            except StopIteration:
                pass
            finally:
                set_call_flags(0)
        # f32 (its own arithmetic and, in the tile kernel, its own pair-loop shape): ONE step from the same state; the
        # continuous outputs within 1e-4 of the f64 oracle (observations only where they carry no visibility masks)
        p1, v1, obs1 = bo.env_step(pos, vel, acts[0], noises[0] if has_noise else None,
                                   goal=goal_arg, lvel=lvel0)[:3]
        if not perturb and np.isfinite(p1).all():
            sim = HipSim(spec, B, torch.float32)
            sim.set_state(pos, vel, goal, lf0)
            if lm_moves:
                sim.lvel.copy_(dev(lvel0, torch.float32))
            sim.env_step(dev(acts[0], torch.float32), dev(noises[0], torch.float32) if has_noise else None)
            gp, gv, go, _, _ = sim.host()
            checks = [('pos', gp, p1), ('vel', gv, v1)]
            if not ragged:
                checks.append(('obs', go, np.concatenate(obs1, 0)))
            # 1e-4 relative to the scale of the step: in a crowd of 100 overlapping entities an agent's force is a sum
            # of dozens of contact terms of either sign, and f32 rounds each of them relative to ITS size, not to the
            # net velocity that is left (the f64 leg above holds 1e-9 on the same worlds)
            scale = max(1.0, float(np.abs(v1).max()))
            # ... and no worse than a few times what the same operations in numpy float32 lose on this world (the speed
            # clamp turns a cancellation error of the force into an error of the direction, whatever |v| is)
            p32, v32 = O.BatchedOracle(spec, np.float32).env_step(
                pos, vel, acts[0], noises[0] if has_noise else None, goal=goal_arg,
                lvel=lvel0)[:2]
            slack = 8.0 * max(float(np.nanmax(np.abs(np.asarray(v32, np.float64) - v1))),
                              float(np.nanmax(np.abs(np.asarray(p32, np.float64) - p1))))
            for nm, x, y in checks:
                if not np.allclose(x, np.asarray(y).reshape(x.shape), rtol=1e-4, atol=1e-4 * scale + slack):
                    bad += 1
                    print('F32 MISMATCH world %d (%s A=%d L=%d) %s: max err %.3e   spec=%r'
                          % (w, kind, A, spec.n_landmarks, nm, np.nanmax(np.abs(x - np.asarray(y).reshape(x.shape))), spec), flush=True)
                    break
        # f32 again, free-running: every kernel form steps the world T times from the same state and actions; the forms
        # keep one operation order, so all outputs agree bit for bit (the f64 leg pins that order to the oracle)
        if not perturb:
            forms = (0, 1, 2) if kind in ('stock_hier', 'stock_small') else (0, 1)
            sims = []
            for flags in forms:
                sim = HipSim(spec, B, torch.float32)
                sim.set_state(pos, vel, goal, lf0)
                if lm_moves:
                    sim.lvel.copy_(dev(lvel0, torch.float32))
                sims.append(sim)
            names = ('pos', 'vel', 'obs', 'rew', 'info') + (('lvel',) if lm_moves else ()) + (('lforce',) if lf0 is not None else ())
            differ = False
            for t in range(T):
                outs = []
                for flags, sim in zip(forms, sims):
                    set_call_flags(flags)
                    try:
                        sim.env_step(dev(acts[t], torch.float32), dev(noises[t], torch.float32) if has_noise else None)
                    finally:
                        set_call_flags(0)
                    outs.append(sim.host() + ((sim.lvel.cpu().numpy(),) if lm_moves else ())
                                + ((sim.lforce[..., :B].cpu().numpy(),) if lf0 is not None else ()))
                for flags, other in zip(forms[1:], outs[1:]):
                    for nm, x, y in zip(names, outs[0], other):
                        if not np.array_equal(x, y, equal_nan=True):
                            bad += 1
                            differ = True
                            x2, y2 = x.reshape(-1, B), y.reshape(-1, B)
                            ne = ~((x2 == y2) | (np.isnan(x2) & np.isnan(y2)))
                            rows = np.nonzero(ne.any(1))[0]
                            r0 = rows[0]
                            e0 = np.nonzero(ne[r0])[0][0]
                            print('F32 KERNEL FORMS DIFFER world %d (%s A=%d L=%d) step %d flags=%d vs %d %s: rows %s, %d envs; row %d env %d: %r vs %r   spec=%r'
                                  % (w, kind, A, spec.n_landmarks, t, forms[0], flags, nm, rows.tolist(), int(ne.any(0).sum()), r0, e0,
                                     float(x2[r0, e0]), float(y2[r0, e0]), spec), flush=True)
                            break
                if differ:
                    break
        kernels[kind] = kernels.get(kind, 0) + 1
        if progress and (w + 1) % progress == 0:
            print('... %d worlds, %d mismatching runs so far' % (w + 1, bad), flush=True)
    return bad, refused, {str(k): v for k, v in kernels.items()}


def main():
    n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    mode = sys.argv[3:]
    bad, refused, kinds = run(n_worlds, seed, stock_only=('moving' if 'moving' in mode else 'stock' in mode), progress=500 if int(os.environ.get('FUZZ_ENVS', 257)) <= 1000 else 50, grid='grid' in mode)
    print('worlds %d (refused by mpe_desc_check: %d), by kind %s, mismatching (world, kernel) runs: %d'
          % (n_worlds, refused, kinds, bad))
    sys.exit(1 if bad else 0)


def desc_from_spec_or_none(spec):
    try:
        return desc_from_spec(spec)
    except AssertionError:
        return None


if __name__ == '__main__':
    main()
