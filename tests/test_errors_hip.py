"""GPU suite: the library fails loudly — integer status + ilsx_last_error() surfaced as RuntimeError by the adapters — instead
of computing something else (no silent fallback anywhere on the product path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shape_and_state_errors_are_reported(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy
    with pytest.raises(NotImplementedError, match="widest kernel"):
        ia.FlattenMlp([400, 300], 1, 14, ctx=ctx)            # any widths up to 256 run (embedded as structural zeros); wider has no kernel
    assert ia.FlattenMlp([100, 100], 1, 14, ctx=ctx).num_params == 14 * 100 + 100 + 100 * 100 + 100 + 100 + 1
    assert ia.FlattenMlp([128, 256], 1, 14, ctx=ctx).kernel_width == 256
    # (round 5: PPO takes any list of widths up to 256 like every other trainer — g7e; what it still refuses is a value net of another shape)
    tr_ppo = PPO(ReparamMultivariateGaussianPolicy([128, 64], 11, 3, conditioned_std=False, hidden_activation="tanh", ctx=ctx),
                 ia.Mlp([128, 64], 1, 11, hidden_activation="tanh", ctx=ctx))
    assert tr_ppo.get_flat_params(1).size == 11 * 128 + 128 + 128 * 64 + 64 + 64 + 1
    with pytest.raises(ValueError, match="share net_size"):
        PPO(ReparamMultivariateGaussianPolicy([128, 64], 11, 3, conditioned_std=False, hidden_activation="tanh", ctx=ctx),
            ia.Mlp([64, 64], 1, 11, hidden_activation="tanh", ctx=ctx))
    net = ia.FlattenMlp([64, 64], 1, 14, ctx=ctx)
    with pytest.raises(RuntimeError, match="libilsx error"):
        net.set_flat_params(np.zeros(net.num_params + 1, np.float32))
    rb = SimpleReplayBuffer(128, 11, 3, random_seed=0, ctx=ctx)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], 11, 3, ctx=ctx)
    q1, q2 = ia.FlattenMlp([64, 64], 1, 14, ctx=ctx), ia.FlattenMlp([64, 64], 1, 14, ctx=ctx)
    bad_q = ia.FlattenMlp([64, 64], 1, 15, ctx=ctx)
    with pytest.raises(RuntimeError, match="qf1/qf2 must be identical|qf input"):
        ia.SoftActorCritic(pol, q1, bad_q, max_batch=64)
    tr = ia.SoftActorCritic(pol, q1, q2, max_batch=64)
    with pytest.raises(RuntimeError, match="already belongs to an agent"):
        ia.SoftActorCritic(pol, ia.FlattenMlp([64, 64], 1, 14, ctx=ctx), ia.FlattenMlp([64, 64], 1, 14, ctx=ctx), max_batch=64)
    with pytest.raises(RuntimeError, match="empty"):
        tr.train_from_replay(rb, 1, 32)
    rng = np.random.default_rng(0)
    rb.add_rows(rng.normal(0, 1, (64, 11)), rng.normal(0, 1, (64, 3)), np.zeros(64), np.zeros(64, bool), rng.normal(0, 1, (64, 11)))
    with pytest.raises(RuntimeError, match="not in 1..max_batch"):
        tr.train_from_replay(rb, 1, 128)
    wrong = SimpleReplayBuffer(128, 17, 6, random_seed=0, ctx=ctx)
    wrong.add_rows(rng.normal(0, 1, (8, 17)), rng.normal(0, 1, (8, 6)), np.zeros(8), np.zeros(8, bool), rng.normal(0, 1, (8, 17)))
    with pytest.raises(RuntimeError, match="replay dims"):
        tr.train_from_replay(wrong, 1, 8)
    tr.train_from_replay(rb, 2, 32)     # and the good call still works afterwards
    two_head = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], 11, 3, ctx=ctx)
    two_head.noise, two_head.noise_clip, two_head.max_act = 0.2, 0.5, 1.0
    with pytest.raises(RuntimeError, match="single-head"):
        TD3(two_head, ia.FlattenMlp([64, 64], 1, 14, ctx=ctx), ia.FlattenMlp([64, 64], 1, 14, ctx=ctx), max_batch=64)
    tpol = MlpGaussianNoisePolicy([64, 64], 11, 3, output_activation="tanh", ctx=ctx)
    with pytest.raises(RuntimeError, match="no log-probability"):
        from ilswiss_amd import _lib
        from ilswiss_amd.device import as_dev
        k, p = as_dev(ctx, np.zeros((4, 11), np.float32))
        act, lp = ctx.empty((4, 3)), ctx.empty((4,))
        _lib.check(ctx.lib.ilsx_policy_act(tpol.h, p, 4, 0, None, act.ptr, lp.ptr))
    ppol = ReparamMultivariateGaussianPolicy([64, 64], 11, 3, conditioned_std=False, hidden_activation="tanh", ctx=ctx)
    vf = ia.FlattenMlp([64, 64], 1, 11, hidden_activation="tanh", ctx=ctx)
    ppo = PPO(ppol, vf, mini_batch_size=16, update_epoch=1, max_samples=64)
    trajs = [dict(observations=np.zeros((100, 11), np.float32), actions=np.zeros((100, 3), np.float32), rewards=np.zeros((100, 1), np.float32))]
    with pytest.raises(RuntimeError, match="max_samples"):
        ppo.train_step(trajs)
    with pytest.raises(ValueError, match="tanh hidden units"):
        PPO(ppol, ia.FlattenMlp([64, 64], 1, 11, ctx=ctx), max_samples=64)
