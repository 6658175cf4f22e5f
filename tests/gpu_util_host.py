"""desc_from_spec without touching CUDA (shared by CPU and GPU tests)."""
from marl_amd import _hip


def desc_from_spec(spec):
    d = _hip.ScenarioDesc()
    d.kind = spec.kind
    d.n_agents, d.n_landmarks, d.dim_c = spec.n_agents, spec.n_landmarks, spec.dim_c
    d.dt, d.damping = spec.dt, spec.damping
    d.contact_force, d.contact_margin = spec.contact_force, spec.contact_margin
    d.agent_pos_range, d.landmark_pos_range = spec.agent_pos_range, spec.landmark_pos_range
    for e in range(spec.n_entities):
        d.size[e], d.mass[e] = spec.size[e], spec.mass[e]
        d.movable[e], d.collide[e] = int(spec.movable[e]), int(spec.collide[e])
    for a in range(spec.n_agents):
        d.accel[a] = 5.0 if spec.accel[a] is None else spec.accel[a]
        d.max_speed[a] = -1.0 if spec.max_speed[a] is None else spec.max_speed[a]
        d.u_noise[a] = spec.u_noise[a] if spec.u_noise[a] else 0.0
        d.adversary[a] = int(spec.adversary[a])
        d.silent[a] = int(spec.silent[a])
        d.force_id[a], d.forced_id[a] = spec.force_id[a], spec.forced_id[a]
        d.advleader[a], d.advsuperleader[a] = int(spec.advleader[a]), int(spec.advsuperleader[a])
        d.obs_r[a] = spec.obs_r[a]
        d.advfollower[a] = int(spec.advfollower[a])
        d.leader[a] = int(spec.leader[a])
        d.c_noise[a] = spec.c_noise[a] if spec.c_noise[a] else 0.0
    for l in range(spec.n_landmarks):
        ms = spec.landmark_max_speed[l]
        d.lm_max_speed[spec.n_agents + l] = -1.0 if ms is None else ms
    d.n_food, d.n_forests = spec.n_food, spec.n_forests
    d.reward_variant, d.n_target_obs = spec.reward_variant, spec.n_target_obs
    assert _hip.lib.mpe_desc_check(d) == 0
    return d
