"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and the two things libilsx
makes of it — the N(0,1) draws of every stochastic policy epilogue (csrc/kernels.h: philox_normal4) and the replay index draw
(replay_draw; the device stand-in for `np.random.RandomState.randint` of simple_replay_buffer.py:242 in the fused loop).  The
reference's noise comes from torch.randn (distributions.py:24), which no other generator can reproduce; what is pinned here is that
the library's stream IS the published Philox (Random123 known-answer vectors, tests/test_philox.py) and that its normals are the
Box-Muller transform of it.  numpy; test infrastructure only."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr [..., 4] uint32, key [..., 2] uint32 (broadcastable) -> [..., 4] uint32."""
    c = [np.asarray(ctr[..., i], np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], np.uint64)
    k1 = np.asarray(key[..., 1], np.uint64)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, -1).astype(np.uint32)


def _ctr_key(seed, step, stream, row, quad):
    row = np.asarray(row, np.uint32)
    ctr = np.stack([row, np.full_like(row, quad), np.full_like(row, step & 0xFFFFFFFF),
                    np.full_like(row, ((step >> 32) ^ ((stream * W0) & 0xFFFFFFFF)) & 0xFFFFFFFF)], -1)
    key = np.array([seed & 0xFFFFFFFF, ((seed >> 32) ^ stream) & 0xFFFFFFFF], np.uint32)
    return ctr, key


def u01_open(x):
    """(0,1) from the top 24 bits: ((x >> 8) + 0.5) / 2^24"""
    return ((x >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0


def normals(seed, step, stream, n_rows, a):
    """[n_rows, a] float64: dim j of row r is element j & 3 of the block with counter (r, j >> 2): Box-Muller pairs
    (sqrt(-2 ln u0) cos 2 pi u1, .. sin .., sqrt(-2 ln u2) cos 2 pi u3, .. sin ..)."""
    out = np.empty((n_rows, a))
    rows = np.arange(n_rows, dtype=np.uint32)
    for q in range((a + 3) // 4):
        c = philox4x32_10(*_ctr_key(seed, step, stream, rows, q))
        r0, r1 = np.sqrt(-2.0 * np.log(u01_open(c[:, 0]))), np.sqrt(-2.0 * np.log(u01_open(c[:, 2])))
        t0, t1 = 2.0 * np.pi * u01_open(c[:, 1]), 2.0 * np.pi * u01_open(c[:, 3])
        z = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], 1)
        out[:, 4 * q:4 * q + 4] = z[:, : min(4, a - 4 * q)]
    return out


def replay_draw(seed, step, stream, n, size):
    """Index of batch row r = 0..n-1: word r & 3 of the block with counter (r >> 2, 'RBUF'), scaled to [0, size) by a 32x64 multiply-high."""
    r = np.arange(n, dtype=np.uint32)
    c = philox4x32_10(*_ctr_key(seed, step, stream, r >> np.uint32(2), 0x52425546))
    u = c[np.arange(n), r & np.uint32(3)].astype(np.uint64)
    return ((u * np.uint64(size)) >> np.uint64(32)).astype(np.int64)


def disc_eps(seed, step, stream, B):
    """The U[0,1) interpolation weights of the discriminator's gradient penalty (ptu.rand(B, 1), adv_irl.py:184) as k_disc_prep draws
    them: row r takes word r & 3 of the block with counter (r >> 2, 'DISC'); w = (word >> 8) * 2^-24 (exact in fp32)."""
    r = np.arange(B, dtype=np.uint32)
    c = philox4x32_10(*_ctr_key(seed, step, stream, r >> np.uint32(2), 0x44495343))
    u = c[np.arange(B), r & np.uint32(3)]
    return ((u >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(B, 1)
