"""MLP forward / backward / init, restating rlkit/torch/common/networks.py:23-115 (Mlp, FlattenMlp)
and rlkit/torch/utils/pytorch_util.py:20-29 (fanin_init).  numpy fp32.  Test infrastructure.

Flat parameter layout (the libilsx ABI uses the same one, include/ilsx.h):
    fc0.W [H0,in] row-major | fc0.b [H0] | fc1.W [H1,H0] | fc1.b [H1] | ... |
    head0.W [out,Hlast] | head0.b [out] [| head1.W | head1.b]
which is torch's `parameters()` order for Mlp (head0 = last_fc) and for the tanh-Gaussian policy
(head0 = last_fc, head1 = last_fc_log_std; policies.py:231-239).  y = x @ W.T + b (nn.Linear).
"""
import numpy as np

F32 = np.float32
RELU, TANH = 0, 1


def layer_shapes(in_dim, hidden, out_dim, n_heads=1):
    shapes = []
    d = in_dim
    for h in hidden:
        shapes.append((h, d))
        d = h
    for _ in range(n_heads):
        shapes.append((out_dim, d))
    return shapes


def n_params(in_dim, hidden, out_dim, n_heads=1):
    return sum(o * i + o for o, i in layer_shapes(in_dim, hidden, out_dim, n_heads))


def unpack(flat, in_dim, hidden, out_dim, n_heads=1):
    """flat fp32 vector -> list of (W, b) views."""
    out, off = [], 0
    for o, i in layer_shapes(in_dim, hidden, out_dim, n_heads):
        W = flat[off : off + o * i].reshape(o, i)
        off += o * i
        b = flat[off : off + o]
        off += o
        out.append((W, b))
    assert off == flat.size
    return out


def pack(layers):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in layers]).astype(F32)


def init_mlp(rng, in_dim, hidden, out_dim, init_w=3e-3, b_init=0.1, n_heads=1, last_scale=None):
    """networks.py:57-83 + pytorch_util.py:20-29.

    Hidden W ~ U(+-1/sqrt(size[0])) where size[0] is nn.Linear.weight.size(0) == OUT features
    (pytorch_util.py:21-24 — fan_in is read from the wrong axis; we reproduce it), hidden b = b_init,
    heads W,b ~ U(+-init_w) (networks.py:82-83, policies.py:236-237).
    `last_scale=(w_mul, b_mul)` reproduces policies.py:378-379 (PPO policy: W*0.1, b*0).
    """
    layers = []
    shapes = layer_shapes(in_dim, hidden, out_dim, n_heads)
    for li, (o, i) in enumerate(shapes):
        if li < len(hidden):
            bound = 1.0 / np.sqrt(o)
            W = rng.uniform(-bound, bound, size=(o, i)).astype(F32)
            b = np.full((o,), b_init, dtype=F32)
        else:
            W = rng.uniform(-init_w, init_w, size=(o, i)).astype(F32)
            b = rng.uniform(-init_w, init_w, size=(o,)).astype(F32)
            if last_scale is not None and li == len(hidden):
                W = (W * F32(last_scale[0])).astype(F32)
                b = (b * F32(last_scale[1])).astype(F32)
        layers.append((W, b))
    return pack(layers)


def _act(z, act):
    if act == RELU:
        return np.maximum(z, F32(0))
    return np.tanh(z).astype(F32)


def forward(flat, x, in_dim, hidden, out_dim, n_heads=1, act=RELU):
    """networks.py:85-101.  Returns (list of head outputs [rows,out], cache)."""
    layers = unpack(flat, in_dim, hidden, out_dim, n_heads)
    h = np.ascontiguousarray(x, dtype=F32)
    hs = [h]
    for W, b in layers[: len(hidden)]:
        h = _act(h @ W.T + b, act).astype(F32)
        hs.append(h)
    outs = [(h @ W.T + b).astype(F32) for W, b in layers[len(hidden) :]]
    return outs, hs


def backward(flat, hs, douts, in_dim, hidden, out_dim, n_heads=1, act=RELU, need_dx=True):
    """Manual backprop of `forward`.  douts: list (per head) of dL/dout [rows,out].
    Returns (flat grad in the same layout, dL/dx or None)."""
    layers = unpack(flat, in_dim, hidden, out_dim, n_heads)
    nh = len(hidden)
    grads = [None] * len(layers)
    hl = hs[-1]
    dh = np.zeros_like(hl)
    for k in range(n_heads):
        W, _ = layers[nh + k]
        d = np.ascontiguousarray(douts[k], dtype=F32)
        grads[nh + k] = ((d.T @ hl).astype(F32), d.sum(0).astype(F32))
        dh = dh + d @ W
    for li in range(nh - 1, -1, -1):
        W, _ = layers[li]
        h_out, h_in = hs[li + 1], hs[li]
        if act == RELU:
            dz = dh * (h_out > 0)
        else:
            dz = dh * (F32(1) - h_out * h_out)
        dz = dz.astype(F32)
        grads[li] = ((dz.T @ h_in).astype(F32), dz.sum(0).astype(F32))
        if li > 0 or need_dx:
            dh = (dz @ W).astype(F32)
    return pack(grads), (dh if need_dx else None)
