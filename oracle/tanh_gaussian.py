"""tanh-Gaussian policy head, restating rlkit/torch/common/distributions.py:11-104 and
rlkit/torch/common/policies.py:248-307,329-345 (LOG_SIG_MIN/MAX = -20/2, policies.py:15-16).
numpy fp32 (dtype argument lets tests evaluate the same formulas in float64).  Test infrastructure.
"""
import numpy as np

LOG_SIG_MIN, LOG_SIG_MAX = -20.0, 2.0
EPS = 1e-6
HALF_LOG_2PI = 0.5 * np.log(2.0 * np.pi)  # added ONCE, not per action dim (distributions.py:45-49)


def head_forward(mu, log_std_raw, eps, dtype=np.float32):
    """policies.py:262-283 + distributions.py:23-28,43-50,74-97.
    Returns dict(action, log_std, std, z, log_prob[rows,1])."""
    T = dtype
    mu = mu.astype(T)
    ls = np.clip(log_std_raw.astype(T), T(LOG_SIG_MIN), T(LOG_SIG_MAX))
    std = np.exp(ls)
    z = eps.astype(T) * std + mu
    a = np.tanh(z)
    cov = np.exp(T(2.0) * ls)
    lp = T(-0.5) * np.sum((mu - z) ** 2 / cov, axis=1, keepdims=True)
    lp = lp - (np.sum(ls, axis=1, keepdims=True) + T(HALF_LOG_2PI))
    lp = lp - np.sum(np.log(T(1) - a * a + T(EPS)), axis=1, keepdims=True)
    return dict(action=a.astype(T), log_std=ls, std=std, z=z, log_prob=lp.astype(T))


def head_backward(fw, eps, log_std_raw, g_action, g_logp, g_mu_extra=None, g_ls_extra=None,
                  dtype=np.float32):
    """Analytic backward.  g_action [rows,a], g_logp [rows,1] upstream grads;
    g_mu_extra / g_ls_extra are direct grads on mean / CLAMPED log_std (SAC regularisers).
    Returns (d_mu, d_log_std_raw).  (mu - z)^2/sigma^2 == eps^2 carries no gradient."""
    T = dtype
    a = fw["action"].astype(T)
    one_m = T(1) - a * a
    dz = g_action.astype(T) * one_m + g_logp.astype(T) * (T(2) * a * one_m / (one_m + T(EPS)))
    d_mu = dz.copy()
    d_ls = dz * fw["std"].astype(T) * eps.astype(T) - g_logp.astype(T)
    if g_mu_extra is not None:
        d_mu = d_mu + g_mu_extra.astype(T)
    if g_ls_extra is not None:
        d_ls = d_ls + g_ls_extra.astype(T)
    raw = log_std_raw.astype(T)
    gate = (raw >= T(LOG_SIG_MIN)) & (raw <= T(LOG_SIG_MAX))  # torch.clamp passes grad on the closed interval
    return d_mu.astype(T), (d_ls * gate).astype(T)


def log_prob_of_action(mu, log_std_raw, action, dtype=np.float32):
    """policies.py:329-345 -> distributions.py:74-97 with pre_tanh_value=None:
    z = 0.5*(log(1+a+eps) - log(1-a+eps))."""
    T = dtype
    mu = mu.astype(T)
    a = action.astype(T)
    ls = np.clip(log_std_raw.astype(T), T(LOG_SIG_MIN), T(LOG_SIG_MAX))
    z = T(0.5) * (np.log(T(1) + a + T(EPS)) - np.log(T(1) - a + T(EPS)))
    cov = np.exp(T(2.0) * ls)
    lp = T(-0.5) * np.sum((mu - z) ** 2 / cov, axis=1, keepdims=True)
    lp = lp - (np.sum(ls, axis=1, keepdims=True) + T(HALF_LOG_2PI))
    lp = lp - np.sum(np.log(T(1) - a * a + T(EPS)), axis=1, keepdims=True)
    return lp.astype(T)


def gaussian_log_prob(mu, log_std, value, dtype=np.float32):
    """distributions.py:43-50 (un-squashed Gaussian; PPO policy, policies.py:348-478)."""
    T = dtype
    cov = np.exp(T(2.0) * log_std.astype(T))
    lp = T(-0.5) * np.sum((mu.astype(T) - value.astype(T)) ** 2 / cov, axis=1, keepdims=True)
    lp = lp - (np.sum(log_std.astype(T), axis=1, keepdims=True) + T(HALF_LOG_2PI))
    return lp.astype(T)


def logp_fp32_tolerance(action, ulps=4.0):
    """Per-row absolute tolerance for comparing two fp32 evaluations of log_prob: the Jacobian term
    log(1 - a^2 + 1e-6) turns a 1-ulp difference in tanh(z) into 2|a|*ulp/(1 - a^2 + 1e-6)
    (the reference's own fp32 value moves by this much between tanh implementations)."""
    a = np.asarray(action, dtype=np.float64)
    ulp = 6.0e-8
    sens = np.sum(2.0 * np.abs(a) * ulp / (1.0 - a * a + 1e-6), axis=1, keepdims=True)
    return 1e-5 + ulps * sens
