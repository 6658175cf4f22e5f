"""MADDPG learner on the batches the on-device replay buffers hand out (reference:
maddpg/maddpg/trainer/maddpg.py, maddpg/maddpg/common/distributions.py, experiments/train_and_eval.py:94-101).
SURVEY 8f row f4: observations, actions and rewards go from the env kernels into the replay rings and from
there into the networks without leaving HBM.

PyTorch is used here for what it is good at (small dense layers through rocBLAS / hipBLASLt, autograd, Adam);
the data path around it (env step, replay append / gather) is the HIP library.  The reference is TensorFlow 1.x
graph code; this module restates it function by function:

    mlp_model                  train_and_eval.py:94-101   two hidden ReLU layers of num_units, linear output,
                                                           Xavier-uniform weights / zero biases (tf.contrib.layers)
    soft_sample                distributions.py:199-201,247-251   Gumbel-softmax relaxation per sub-space
    p_train / q_train          maddpg.py:28-110           losses, per-variable clip_by_norm(0.5), Adam(lr)
    make_update_exp            maddpg.py:20-26            target <- 0.99 target + 0.01 online
    MADDPGAgentTrainer         maddpg.py:112-196          action / experience / preupdate / update

*** parity unpinned ***  TensorFlow is not installed here and the reference's learner code is not run
(DESIGN.md section 4); tests check the restated arithmetic against closed forms and that learning reduces the
losses.  Known deviation: torch.optim.Adam adds eps outside the bias-corrected root, TF inside (eps = 1e-8).
"""
import torch
from torch import nn

from .replay_buffer import ReplayBuffer
from .spaces import Discrete, MultiDiscrete


def mlp_model(num_inputs, num_outputs, num_units=64):
    """train_and_eval.py:94-101."""
    net = nn.Sequential(nn.Linear(num_inputs, num_units), nn.ReLU(), nn.Linear(num_units, num_units), nn.ReLU(),
                        nn.Linear(num_units, num_outputs))
    for m in net:
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
    return net


def param_sizes(act_space):
    """Logit widths per sub-space (distributions.py:77-110): Discrete(n) -> [n]; MultiDiscrete -> high - low + 1."""
    if isinstance(act_space, Discrete):
        return [act_space.n]
    if isinstance(act_space, MultiDiscrete):
        return [int(h - l + 1) for l, h in zip(act_space.low, act_space.high)]
    raise NotImplementedError('action space %r' % (act_space,))


def soft_sample(flat, sizes, generator=None):
    """SoftCategoricalPd.sample / SoftMultiCategoricalPd.sample (distributions.py:199-201,247-251):
    softmax(logits - log(-log(u))), u ~ U[0, 1), per sub-space, concatenated (MPE's sub-spaces have low = 0)."""
    u = torch.rand(flat.shape, device=flat.device, dtype=flat.dtype, generator=generator)
    noisy = flat - torch.log(-torch.log(u))
    return torch.cat([torch.softmax(part, dim=-1) for part in torch.split(noisy, sizes, dim=-1)], dim=-1)


ACTOR_HIDDEN, ACTOR_MAX_OUT = 64, 16       # MPE_ACTOR_HIDDEN / MPE_ACTOR_MAX_OUT (include/mpe_hip.h)


def pack_actor(net, out=None):
    """The packed f32 weight array mpe_actor_forward_f32 reads (include/mpe_hip.h): W1t [in][64], b1 [64],
    W2t [64][64], b2 [64], W3t [64][16], b3 [16]; or None when ``net`` is not the stock 2 x 64 model.
    ``out``: an earlier result to refill in place (its address may be recorded in a HIP graph, marl_amd/graph.py)."""
    lin = [m for m in net if isinstance(m, nn.Linear)] if isinstance(net, nn.Sequential) else []
    if (len(lin) != 3 or len(net) != 5 or lin[0].out_features != ACTOR_HIDDEN or lin[1].out_features != ACTOR_HIDDEN
            or lin[2].out_features > ACTOR_MAX_OUT or lin[0].weight.dtype != torch.float32):
        return None
    w3 = torch.zeros(ACTOR_HIDDEN, ACTOR_MAX_OUT, device=lin[2].weight.device)
    b3 = torch.zeros(ACTOR_MAX_OUT, device=lin[2].weight.device)
    w3[:, :lin[2].out_features] = lin[2].weight.detach().t()
    b3[:lin[2].out_features] = lin[2].bias.detach()
    packed = torch.cat([lin[0].weight.detach().t().reshape(-1), lin[0].bias.detach(), lin[1].weight.detach().t().reshape(-1),
                        lin[1].bias.detach(), w3.reshape(-1), b3]).contiguous()
    if out is not None and out.shape == packed.shape and out.device == packed.device:
        return out.copy_(packed)
    return packed


def actor_forward(packed, obs, sizes, u=None, seed=0, call_counter=0, env_id0=0, kernel=0):
    """One launch of the fused actor (libmpe_hip.so: mpe_actor_forward_multi_f32 with one job).  ``obs``: ``[B, in_dim]``
    view of SoA rows (strides (1, ld)), as env.step returns it.  Returns the action as the same kind of view
    (``[B, sum(sizes)]`` over SoA rows with the same ld); env.step copies it row-wise into the env's action buffer (a
    per-agent list is never adopted in place: only a whole [act_total, B] buffer with the world's stride is; see
    ``act_all(..., out=env.action_views)`` for the copy-free form).  ``u``: optional ``[B, sum(sizes)]`` uniforms
    (same layout) instead of the kernel's own Philox draw.  ``kernel``: _hip.ACTOR_AUTO / ACTOR_VALU / ACTOR_MFMA."""
    from . import _hip
    B, in_dim = obs.shape
    ld = obs.stride(1) if in_dim > 1 else B
    nout = sum(sizes)
    act = torch.empty(nout, ld, dtype=torch.float32, device=obs.device)
    if u is not None and (u.stride(0) != 1 or (nout > 1 and u.stride(1) != ld)):
        uu = torch.empty(nout, ld, dtype=torch.float32, device=obs.device)
        uu[:, :B] = u.t()
        u = uu[:, :B].t()
    s0, s1 = sizes[0], (sizes[1] if len(sizes) > 1 else 0)
    job = _hip.ActorJob(obs.data_ptr(), packed.data_ptr(), None if u is None else u.data_ptr(), act.data_ptr(), in_dim,
                        s0, s1, seed)
    with _hip.on_device(obs.device):
        _hip.check(_hip.lib.mpe_actor_forward_multi_f32(1, job, B, ld, ld, call_counter, env_id0, kernel,
                                                        _hip.stream_of(obs.device)), 'mpe_actor_forward')
    return act[:, :B].t()


@torch.no_grad()
def act_all(trainers, obs_n, out=None, kernel=0, counter_dev=None):
    """``[tr.action(obs) for tr, obs in zip(trainers, obs_n)]`` (experiments/simple_train.py:260) with every agent's
    network + Gumbel-softmax sample in ONE launch (libmpe_hip.so: mpe_actor_forward_multi_f32).  Falls back to the
    per-agent calls when some agent's observation / model is not what the fused kernel takes.

    ``out``: ``env.action_views`` -- the kernel then writes straight into the env's SoA action rows and ``env.step``
    of the returned list copies nothing (without it each agent's action lands in a fresh buffer that ``env.step``
    copies row-wise: 4 x 9 us per step for simple_tag at 1 Mi envs).

    ``counter_dev``: an int32 device tensor whose word 0 is added to the call counter that keys the Gumbel noise
    (mpe_actor_forward_multi_ctr_f32): marl_amd.graph.CapturedTrainSteps records acting steps in a HIP graph and moves
    that word between replays.  Needs the fused path (raises otherwise)."""
    from . import _hip
    t0, o0 = trainers[0], obs_n[0]
    fused = o0.dim() == 2 and o0.shape[1] > 1
    B, ld = (o0.shape[0], o0.stride(1)) if fused else (0, 0)
    for tr, obs in zip(trainers, obs_n):
        sizes = tr.sizes_n[tr.agent_index]
        fused = (fused and obs.dim() == 2 and obs.dtype == torch.float32 and obs.is_cuda and obs.stride(0) == 1
                 and obs.shape[1] > 1 and (obs.shape[0], obs.stride(1)) == (B, ld) and len(sizes) <= 2
                 and sum(sizes) <= ACTOR_MAX_OUT and tr._act_calls == t0._act_calls and tr.env_id0 == t0.env_id0)
        if fused and tr._packed is None:
            tr._packed = pack_actor(tr.p)
        fused = fused and tr._packed is not None
    if not fused:
        if counter_dev is not None:
            raise RuntimeError('act_all(counter_dev=...) needs the fused actor kernel for every agent')
        return [tr.action(obs) for tr, obs in zip(trainers, obs_n)]
    nouts = [sum(tr.sizes_n[tr.agent_index]) for tr in trainers]
    # action rows: the caller's SoA views when they share one row stride, else fresh buffers with the observations' stride
    in_place = out is not None and len(out) == len(trainers) and all(
        torch.is_tensor(v) and v.dim() == 2 and v.dtype == torch.float32 and v.device == o0.device
        and tuple(v.shape) == (B, n) and v.stride(0) == 1 for v, n in zip(out, nouts))
    strides = {v.stride(1) for v in out if v.shape[1] > 1} if in_place else set()
    if in_place and len(strides) == 1:
        acts, ld_act = list(out), strides.pop()
    else:
        acts, ld_act = [torch.empty(n, ld, dtype=torch.float32, device=o0.device)[:, :B].t() for n in nouts], ld
    for tr in trainers:
        tr._act_calls += 1
    jobs = (_hip.ActorJob * len(trainers))(*[
        _hip.ActorJob(obs.data_ptr(), tr._packed.data_ptr(), None, act.data_ptr(), obs.shape[1],
                      tr.sizes_n[tr.agent_index][0], (tr.sizes_n[tr.agent_index] + [0])[1], tr.seed)
        for tr, obs, act in zip(trainers, obs_n, acts)])
    with _hip.on_device(o0.device):
        _hip.check(_hip.lib.mpe_actor_forward_multi_ctr_f32(
            len(trainers), jobs, B, ld, ld_act, t0._act_calls, None if counter_dev is None else counter_dev.data_ptr(),
            t0.env_id0, kernel, _hip.stream_of(o0.device)), 'mpe_actor_forward_multi')
    return acts


def clip_by_norm_(params, clip_val):
    """tf_util.py:137-149: every variable's gradient clipped to norm clip_val on its own (tf.clip_by_norm)."""
    for p in params:
        if p.grad is not None:
            norm = p.grad.norm()
            p.grad.mul_(clip_val / torch.clamp(norm, min=clip_val))


def polyak_update_(online, target, polyak=1.0 - 1e-2):
    """maddpg.py:20-26."""
    with torch.no_grad():
        for v, vt in zip(online.parameters(), target.parameters()):
            vt.mul_(polyak).add_(v, alpha=1.0 - polyak)


class MADDPGAgentTrainer(object):
    """maddpg.py:112-196, for batched envs: ``action`` maps ``[B, obs_dim]`` to ``[B, act_dim]``; ``experience``
    appends one step of all B envs to this agent's replay ring."""

    def __init__(self, name, obs_shape_n, act_space_n, agent_index, args, local_q_func=False, model=mlp_model,
                 device='cuda', dtype=torch.float32, replay_size=1e6, seed=0, env_id0=0, replay_compact=False):
        """``seed``: the run's seed (keys the exploration noise of the fused actor kernel together with the agent
        index); ``env_id0``: global id of this process's env 0 (``world.env_id0`` of an env-sharded multi-GPU run), so
        that the noise of env ``i`` does not depend on how the batch is sharded -- like the reset RNG."""
        self.name = name
        self.n = len(obs_shape_n)
        self.agent_index = agent_index
        self.args = args
        self.local_q_func = local_q_func
        self.device, self.dtype = torch.device(device), dtype
        obs_dims = [int(s[0]) for s in obs_shape_n]
        self.sizes_n = [param_sizes(sp) for sp in act_space_n]
        act_dims = [sum(s) for s in self.sizes_n]
        i = agent_index
        q_in = obs_dims[i] + act_dims[i] if local_q_func else sum(obs_dims) + sum(act_dims)
        mk = lambda a, b: model(a, b, num_units=args.num_units).to(device=self.device, dtype=dtype)
        self.p, self.target_p = mk(obs_dims[i], act_dims[i]), mk(obs_dims[i], act_dims[i])
        self.q, self.target_q = mk(q_in, 1), mk(q_in, 1)
        # tf.global_variables_initializer gives the target nets their own random init (they are separate
        # variables, maddpg.py:66,104); only the polyak updates pull them towards the online nets.
        self.p_opt = torch.optim.Adam(self.p.parameters(), lr=args.lr, eps=1e-8)
        self.q_opt = torch.optim.Adam(self.q.parameters(), lr=args.lr, eps=1e-8)
        self.grad_norm_clipping = 0.5                                     # maddpg.py:130,142
        self.replay_buffer = ReplayBuffer(replay_size, device=self.device, dtype=dtype, compact=replay_compact)   # maddpg.py:147
        self.max_replay_buffer_len = args.batch_size * args.max_episode_len
        self.replay_sample_index = None
        self._packed = None            # packed actor weights for the fused kernel; dropped by every update
        self._act_calls = 0
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x5EED0000 + agent_index) & 0xFFFFFFFFFFFFFFFF
        self.env_id0 = int(env_id0)

    # -- acting ------------------------------------------------------------------------------------
    @torch.no_grad()
    def action(self, obs):
        """maddpg.py:151-152 (act = a sample of the soft categorical, :45,62).  Observations that are views of
        the env kernels' SoA rows go through the fused HIP kernel (network + Gumbel-softmax in one launch,
        mpe_actor_forward_f32); anything else (other layouts, models, dtypes) through the same arithmetic in
        torch on the GPU."""
        sizes = self.sizes_n[self.agent_index]
        if (obs.dim() == 2 and obs.dtype == torch.float32 and obs.is_cuda and obs.stride(0) == 1
                and (obs.shape[1] == 1 or obs.stride(1) >= obs.shape[0]) and len(sizes) <= 2
                and sum(sizes) <= ACTOR_MAX_OUT):
            if self._packed is None:
                self._packed = pack_actor(self.p)
            if self._packed is not None:
                self._act_calls += 1
                return actor_forward(self._packed, obs, sizes, seed=self.seed, call_counter=self._act_calls,
                                     env_id0=self.env_id0)
        return soft_sample(self.p(obs), sizes)

    @torch.no_grad()
    def target_action(self, obs):       # p_debug['target_act'], maddpg.py:70-71
        return soft_sample(self.target_p(obs), self.sizes_n[self.agent_index])

    def experience(self, obs, act, rew, new_obs, done, terminal):
        """maddpg.py:154-156."""
        self.replay_buffer.add(obs, act, rew, new_obs, done, terminal if self.replay_buffer.compact else None)

    def preupdate(self):                # maddpg.py:158-159
        self.replay_sample_index = None

    # -- learning ----------------------------------------------------------------------------------
    def _q_input(self, obs_n, act_n):
        if self.local_q_func:           # maddpg.py:51-52,86-87
            return torch.cat([obs_n[self.agent_index], act_n[self.agent_index]], dim=1)
        return torch.cat(list(obs_n) + list(act_n), dim=1)

    def update(self, agents, t):
        """maddpg.py:161-196.  Returns [q_loss, p_loss, mean(target_q), mean(rew), mean(target_q_next),
        std(target_q)] as device scalars, or None when no update is due."""
        if len(self.replay_buffer) < self.max_replay_buffer_len:          # :162
            return None
        if not t % 100 == 0:                                              # :164
            return None
        self.replay_sample_index = self.replay_buffer.make_index(self.args.batch_size)
        return self.update_from_index(agents, self.replay_sample_index)

    def update_from_index(self, agents, index):
        """maddpg.py:166-196: everything of ``update`` behind its two gates, for the batch ``index`` (a device int64
        tensor) -- the part that ``marl_amd.graph.CapturedUpdates`` records."""
        obs_n, obs_next_n, act_n = [], [], []
        for i in range(self.n):                                           # :173-177
            obs, act, rew, obs_next, done = agents[i].replay_buffer.sample_index(index)
            obs_n.append(obs)
            obs_next_n.append(obs_next)
            act_n.append(act)
        obs, act, rew, obs_next, done = self.replay_buffer.sample_index(index)

        with torch.no_grad():                                             # :181-187 (num_sample = 1)
            target_act_next_n = [agents[i].target_action(obs_next_n[i]) for i in range(self.n)]
            target_q_next = self.target_q(self._q_input(obs_next_n, target_act_next_n))[:, 0]
            target_q = rew + self.args.gamma * (1.0 - done) * target_q_next

        # q_train (:75-110): mean squared Bellman error
        q = self.q(self._q_input(obs_n, act_n))[:, 0]
        q_loss = torch.mean(torch.square(q - target_q))
        self.q_opt.zero_grad(set_to_none=True)
        q_loss.backward()
        clip_by_norm_(self.q.parameters(), self.grad_norm_clipping)
        self.q_opt.step()

        # p_train (:28-73): -mean Q with this agent's action replaced by a fresh sample, + 1e-3 mean(logits^2)
        logits = self.p(obs_n[self.agent_index])
        act_input_n = list(act_n)
        act_input_n[self.agent_index] = soft_sample(logits, self.sizes_n[self.agent_index])
        pg_loss = -torch.mean(self.q(self._q_input(obs_n, act_input_n))[:, 0])
        p_loss = pg_loss + torch.mean(torch.square(logits)) * 1e-3
        self.p_opt.zero_grad(set_to_none=True)
        p_loss.backward()                                                 # only p's variables are stepped (:58)
        clip_by_norm_(self.p.parameters(), self.grad_norm_clipping)
        self.p_opt.step()
        if self._packed is not None:          # refilled in place: a captured acting step keeps reading this buffer
            pack_actor(self.p, out=self._packed)
        self.q_opt.zero_grad(set_to_none=True)

        polyak_update_(self.p, self.target_p)                             # :193-194
        polyak_update_(self.q, self.target_q)
        return [q_loss.detach(), p_loss.detach(), target_q.mean(), rew.mean(), target_q_next.mean(), target_q.std(unbiased=False)]
