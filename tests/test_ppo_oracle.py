"""CPU suite: the PPO oracle (GAE + clipped-surrogate minibatch updates) against the golden vectors produced
by the reference (ppo.py:57-170)."""
import numpy as np

from conftest import load_golden
from oracle.ppo import PPOOracle, gae_one_traj

KW = dict(reward_scale=1.0, discount=0.99, clip_eps=0.2, policy_lr=3e-4, value_lr=3e-4, gae_tau=0.95,
          value_l2_reg=1e-3, mini_batch_size=16, update_epoch=2)


def _trajs(g):
    return [dict(observations=g[f"t{i}_observations"], actions=g[f"t{i}_actions"], rewards=g[f"t{i}_rewards"])
            for i in range(len(g["lens"]))]


def test_gae_and_fixed_log_probs():
    g = load_golden("g7_ppo")
    o, a = int(g["dims"][0]), int(g["dims"][1])
    orc = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **KW)
    obs, act, R, A, V = orc.calc_adv(_trajs(g))
    np.testing.assert_allclose(V, g["values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R, g["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(A, g["advantages"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(orc.log_prob(obs, act)[0], g["fixed_log_probs"], rtol=1e-5, atol=1e-5)
    # per-trajectory standardisation with the unbiased std (ppo.py:86): each segment has mean 0, std(ddof=1) 1
    off = 0
    for L in g["lens"]:
        seg = A[off:off + L]
        assert abs(seg.mean()) < 1e-5 and abs(seg.std(ddof=1) - 1) < 1e-4
        off += L


def test_gae_zero_bootstrap_closed_form():
    v = np.array([[1.0], [2.0], [3.0]], np.float32)
    r = np.array([[0.5], [0.5], [0.5]], np.float32)
    R, _, adv = gae_one_traj(v, r, 0.9, 0.8)
    d2 = 0.5 + 0.9 * 0 - 3.0
    d1 = 0.5 + 0.9 * 3.0 - 2.0
    d0 = 0.5 + 0.9 * 2.0 - 1.0
    a2 = d2; a1 = d1 + 0.72 * a2; a0 = d0 + 0.72 * a1
    np.testing.assert_allclose(adv.ravel(), [a0, a1, a2], rtol=1e-6)
    np.testing.assert_allclose(R.ravel(), [1 + a0, 2 + a1, 3 + a2], rtol=1e-6)


def test_train_step_two_epochs():
    g = load_golden("g7_ppo")
    o, a = int(g["dims"][0]), int(g["dims"][1])
    orc = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **KW)
    orc.train_step(_trajs(g), list(g["perms"]))
    np.testing.assert_allclose(orc.vf, g["vf_final"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(orc.pi, g["pi_final"], rtol=0, atol=5e-5)
    assert np.abs(orc.pi - g["pi0"]).max() > 1e-4  # the policy really moved
