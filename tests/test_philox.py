"""The library's random stream is the published Philox4x32-10: oracle/philox.py against the Random123 known-answer vectors (CPU),
the device generator against oracle/philox.py word for word, its N(0,1) draws against the float64 Box-Muller transform of those words,
and the fused loop's replay indices against the same stream (GPU).  VERDICT r2 weak #10."""
import ctypes as C

import numpy as np
import pytest

from oracle import philox

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
KAT = [((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox4x32_10_known_answers():
    for ctr, key, out in KAT:
        got = philox.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert [int(v) for v in got] == list(out), ([hex(int(v)) for v in got], [hex(v) for v in out])


def test_normals_are_box_muller_of_the_stream():
    z = philox.normals(seed=0x1234_5678_9ABC, step=7, stream=3, n_rows=4096, a=6)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and np.abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.05
    # dims 0..3 come from one block, 4..5 from the next (quad 1): different blocks, both functions of (row, quad) alone
    z2 = philox.normals(seed=0x1234_5678_9ABC, step=7, stream=3, n_rows=16, a=3)
    np.testing.assert_array_equal(z[:16, :3], z2)


@pytest.mark.gpu
def test_device_philox_words_and_normals_known_answer(ctx):
    from ilswiss_amd import _lib
    seed, step, stream, n, a = 0xDEADBEEF12345678, (5 << 32) | 17, 9, 64, 6
    raw = ctx.empty((n, 4))      # 4 x uint32 per row, read back as raw 32-bit words
    nor = ctx.empty((n, a))
    _lib.check(ctx.lib.ilsx_debug_philox(ctx.h, C.c_uint64(seed), C.c_uint64(step), stream, n, a, raw.ptr, nor.ptr))
    words = raw.numpy().view(np.uint32)
    ref_c, ref_k = philox._ctr_key(seed, step, stream, np.arange(n, dtype=np.uint32), 0)
    np.testing.assert_array_equal(words, philox.philox4x32_10(ref_c, ref_k))        # bit for bit, 256 words
    # the device evaluates log2 / sqrt / sin / cos with the hardware approximations (v_log_f32, v_sin_f32 in revolutions: ~1e-6
    # absolute on the unit circle, amplified by |r| <= 5.8): 2e-5 absolute against the float64 transform of the same words
    np.testing.assert_allclose(nor.numpy(), philox.normals(seed, step, stream, n, a), rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_fused_replay_indices_are_the_philox_draw(ctx):
    """ilsx_replay_sample(idx=NULL) — the draw the fused SAC step makes in-kernel — against oracle.philox.replay_draw."""
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    c = ia.Context(0, seed=99)     # fresh context: the ring below owns its first Philox stream (ids start at 1, host_common.h)
    try:
        cap, o, a, B = 1000, 3, 2, 64
        rb = ia.SimpleReplayBuffer(cap, o, a, random_seed=4242, ctx=c)
        rng = np.random.default_rng(0)
        rb.add_rows(rng.normal(0, 1, (700, o)).astype(np.float32), rng.normal(0, 1, (700, a)).astype(np.float32), np.zeros(700, np.float32),
                    np.zeros(700, np.uint8), rng.normal(0, 1, (700, o)).astype(np.float32))
        bufs = [c.empty((B, o)), c.empty((B, a)), c.empty((B,)), c.empty((B,)), c.empty((B, o))]
        idx = c.empty((B,), np.int64)
        for call in (1, 2, 3):      # the k-th call draws with counter k
            _lib.check(c.lib.ilsx_replay_sample(rb.h, B, None, *[b.ptr for b in bufs], idx.ptr))
            np.testing.assert_array_equal(idx.numpy(), philox.replay_draw(4242, call, 1, B, 700))
    finally:
        c.close()
