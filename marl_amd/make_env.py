"""make_env (reference: multiagent-particle-envs/make_env.py:15-44) with batch geometry.

    env = make_env('simple_tag', benchmark=True, num_envs=65536, dtype=torch.float64)
"""


def make_env(scenario_name, benchmark=False, num_envs=1, device=None, dtype=None, seed=0, **scenario_kwargs):
    from marl_amd.environment import MultiAgentEnv
    import marl_amd.scenarios as scenarios

    # load scenario from script
    scenario = scenarios.load(scenario_name + ".py").Scenario(num_envs=num_envs, device=device, dtype=dtype,
                                                             seed=seed, **scenario_kwargs)
    # create world
    world = scenario.make_world()
    # create multiagent environment
    if benchmark:
        env = MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation,
                            scenario.benchmark_data)
    else:
        env = MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation)
    return env
