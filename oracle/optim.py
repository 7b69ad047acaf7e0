"""Adam + Polyak, restating torch.optim.Adam (torch==1.9.0 pinned by requirements.txt:26; call sites
sac_alpha.py:65-76, td3.py:56-67, ppo.py:47-55, adv_irl.py:75-77) and
rlkit/torch/utils/pytorch_util.py:10-12 (soft_update_from_to).  numpy.  Test infrastructure.
"""
import numpy as np


class AdamState:
    def __init__(self, n, dtype=np.float32):
        self.m = np.zeros(n, dtype=dtype)
        self.v = np.zeros(n, dtype=dtype)
        self.t = 0

    def copy(self):
        s = AdamState(self.m.size, self.m.dtype)
        s.m, s.v, s.t = self.m.copy(), self.v.copy(), self.t
        return s


def adam_step(p, g, st, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch 1.9 `_functional.adam` (no amsgrad / weight decay):
        m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2
        p <- p - (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
    The step counter starts at 1 on the first call.  In-place on p and st."""
    T = p.dtype.type
    st.t += 1
    g = g.astype(p.dtype)
    st.m *= T(beta1)
    st.m += T(1.0 - beta1) * g
    st.v *= T(beta2)
    st.v += T(1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** st.t
    bc2 = 1.0 - beta2 ** st.t
    step_size = T(lr / bc1)
    denom = np.sqrt(st.v) / T(np.sqrt(bc2)) + T(eps)
    p -= step_size * (st.m / denom)
    return p


def polyak(target, source, tau):
    """pytorch_util.py:10-12: target <- target*(1-tau) + source*tau.  In-place on target."""
    T = target.dtype.type
    target *= T(1.0 - tau)
    target += source * T(tau)
    return target
