"""CPU suite, world_size 2 over gloo: the N>1 plumbing (ilswiss_amd/parallel.py).  The compute engine behind
the SplitRunStep interface here is the numpy oracle (no GPU in this container); on the GPU box the same
orchestration drives libilsx (tests/test_hip_parity.py::test_sac_two_way_batch_split_matches_single checks
the arithmetic of the split on the device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ilswiss_amd.parallel import SplitRunStep, shard_batch, shard_items
from oracle import mlp as omlp
from oracle.sac_alpha import SacAlphaOracle

KW = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005,
          alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
O, A, HID, B = 11, 3, [32, 32], 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_items_and_batch():
    seeds = list(range(32))
    parts = [shard_items(seeds, 8, r) for r in range(8)]
    assert sum(parts, []) == seeds and all(len(p) == 4 for p in parts)      # config 5: 32 seeds on 8 GPUs
    parts = [shard_items(range(10), 4, r) for r in range(4)]
    assert [len(p) for p in parts] == [3, 3, 2, 2] and sum(parts, []) == list(range(10))
    b = dict(x=np.arange(12).reshape(6, 2))
    assert np.array_equal(shard_batch(b, 3, 1)["x"], b["x"][2:4])
    with pytest.raises(ValueError):
        shard_batch(b, 4, 0)


class OracleTrainer:
    """SplitRunStep's trainer interface over the numpy oracle; the gradient arenas are torch views."""

    def __init__(self, world):
        rng = np.random.default_rng(5)
        self.o = SacAlphaOracle(O, A, HID, omlp.init_mlp(rng, O, HID, A, init_w=1e-3, n_heads=2),
                                omlp.init_mlp(rng, O + A, HID, 1), omlp.init_mlp(rng, O + A, HID, 1), **KW)
        self.o.grad_world = world

    def set_batch(self, batch, e1, e2):
        self.batch, self.e1, self.e2 = batch, e1, e2

    def critic_backward(self):
        self.o.critic_backward(self.batch, self.e1)

    def critic_update(self):
        self.o.critic_update()

    def actor_backward(self):
        self.o.actor_backward(self.e2)

    def actor_update(self):
        self.o.actor_update()

    def grad_tensor(self, seg):
        return torch.from_numpy(self.o.g_critic if seg == 0 else self.o.g_actor)  # shares memory


def _data(steps):
    rng = np.random.default_rng(9)
    out = []
    for _ in range(steps):
        out.append((dict(observations=rng.normal(0, 1, (B, O)).astype(np.float32),
                         actions=np.tanh(rng.normal(0, 1, (B, A))).astype(np.float32),
                         rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
                         terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                         next_observations=rng.normal(0, 1, (B, O)).astype(np.float32)),
                    rng.normal(0, 1, (B, A)).astype(np.float32), rng.normal(0, 1, (B, A)).astype(np.float32)))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = OracleTrainer(world)
    step = SplitRunStep(tr)
    for batch, e1, e2 in _data(3):
        sl = slice(rank * B // world, (rank + 1) * B // world)
        step.train_step(shard_batch(batch, world, rank), e1[sl], e2[sl])
    q.put((rank, tr.o.pi.copy(), tr.o.q1.copy(), tr.o.tq2.copy(), float(tr.o.log_alpha[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_split_run_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = OracleTrainer(1)
    for batch, e1, e2 in _data(3):
        single.o.train_step(batch, e1, e2)
    for r in res:  # every replica took the identical optimiser steps, == the un-split run up to fp32 summation order
        np.testing.assert_array_equal(r[1], res[0][1])
        np.testing.assert_allclose(r[1], single.o.pi, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[2], single.o.q1, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[3], single.o.tq2, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[4], single.o.log_alpha[0], rtol=0, atol=1e-7)


# ---------------------------------------------------------------------------------------------- split run FROM REPLAY SHARDS
# SURVEY §8e / parallel.py: in a split run every rank draws its B/G rows from ITS OWN replay shard; the union of the ranks' draws is a
# stratified-uniform B-row sample of the union of the shards, and the step on it is the single-process step on the concatenated batch.
SHARD_ROWS = 500


class ShardOracle:
    """random_batch over oracle.replay.ReplayOracle (the reference's RandomState.randint draw, simple_replay_buffer.py:242)."""

    def __init__(self, rank):
        from oracle.replay import ReplayOracle
        rng = np.random.default_rng(1000 + rank)
        self.rb = ReplayOracle(SHARD_ROWS, O, A, random_seed=77 + rank)
        n = SHARD_ROWS
        self.rb.add_rows(rng.normal(rank, 1, (n, O)).astype(np.float32), np.tanh(rng.normal(0, 1, (n, A))).astype(np.float32),
                         rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 0.1).astype(np.uint8), rng.normal(rank, 1, (n, O)).astype(np.float32))
        self.noise = np.random.default_rng(2000 + rank)      # per-rank policy noise (a shared stream would correlate the shards' rows)

    def random_batch(self, batch_size):
        b = self.rb.gather(self.rb.draw_indices(batch_size))
        b["terminals"] = b["terminals"].astype(np.float32)
        return b

    def draw_noise(self, n):
        return self.noise.normal(0, 1, (n, A)).astype(np.float32), self.noise.normal(0, 1, (n, A)).astype(np.float32)


def _replay_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = OracleTrainer(world)
    shard = ShardOracle(rank)
    tr.draw_noise = shard.draw_noise
    SplitRunStep(tr).train_from_replay(shard, 4, B // world)
    q.put((rank, tr.o.pi.copy(), tr.o.q1.copy(), tr.o.tq2.copy(), float(tr.o.log_alpha[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_from_replay_shards_is_the_stratified_batch_step():
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_replay_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one process replays both shards' draws and steps on the concatenation: B/2 rows of shard 0, then B/2 rows of shard 1
    shards = [ShardOracle(r) for r in range(world)]
    single = OracleTrainer(1)
    seen0, seen1 = [], []
    for _ in range(4):
        parts = [(s.random_batch(B // world), s.draw_noise(B // world)) for s in shards]
        batch = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
        e1, e2 = np.concatenate([p[1][0] for p in parts]), np.concatenate([p[1][1] for p in parts])
        seen0.append(parts[0][0]["observations"].mean()), seen1.append(parts[1][0]["observations"].mean())
        single.o.train_step(batch, e1, e2)
    assert abs(np.mean(seen0)) < 0.3 and abs(np.mean(seen1) - 1.0) < 0.3        # stratified: exactly half the rows from each shard's distribution
    for r in res:
        np.testing.assert_array_equal(r[1], res[0][1])                            # replicas took identical optimiser steps
        np.testing.assert_allclose(r[1], single.o.pi, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[2], single.o.q1, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[3], single.o.tq2, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[4], single.o.log_alpha[0], rtol=0, atol=1e-7)


# ---------------------------------------------------------------------------------------------- the run scripts' split plumbing (rl_alg_params.split_ranks)
def _agree_worker(rank, world, port, q):
    from ilswiss_amd.parallel import SplitInfo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sp = SplitInfo(world, rank, dist)
    q.put((rank, sp.agree(True), sp.agree(rank == 0), sp.agree(False)))
    dist.barrier()
    dist.destroy_process_group()


def test_split_info_scales_the_loop_and_ranks_agree_on_training():
    """rl_alg_params of a split run: every count stated for the whole run is this rank's share, the batch is B / G rows; a train call happens
    only when every rank can train (one rank alone in the gradient all-reduce would wait for ever)."""
    from ilswiss_amd.parallel import SplitInfo
    alg = dict(batch_size=256, num_steps_per_epoch=10240, num_steps_between_train_calls=1024, min_steps_before_training=10240,
               replay_buffer_size=1000000, num_train_steps_per_train_call=250, num_epochs=3, split_ranks=2, max_path_length=1000)
    out = SplitInfo(2, 1, None).scale(alg)
    assert (out["batch_size"], out["num_steps_per_epoch"], out["num_steps_between_train_calls"], out["min_steps_before_training"],
            out["replay_buffer_size"]) == (128, 5120, 512, 5120, 500000)
    assert out["num_train_steps_per_train_call"] == 250 and out["num_epochs"] == 3 and out["max_path_length"] == 1000   # not counts of rows / env steps
    assert "split_ranks" not in out and out["split_world"] == 2 and callable(out["split_agree"])
    with pytest.raises(ValueError):
        SplitInfo(3, 0, None).scale(alg)
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, False, False), (1, True, False, False)]


# ---------------------------------------------------------------------------------------------- PPO minibatch gradients and the discriminator, split
# "PPO split: same, per-minibatch grads.  Disc: same": every rank holds mini_batch / G (B / G) rows, scales its mean-loss gradients by
# 1 / (rows * G), the flat gradient is summed over the ranks before the L2 term / norm clip / Adam.  The engine here is the numpy oracle
# (grad_world + allreduce hook); on the GPU box the same orchestration runs inside libilsx (ilsx_ppo_cfg.grad_world / ilsx_disc_cfg.grad_world,
# tests/test_ppo_oracle.py and tests/test_disc.py check the device path on a one-rank communicator).
PPO_HID, PPO_MB = [32, 32], 48


def _gloo_sum(flat):
    t = torch.from_numpy(flat)      # shares memory
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return flat


def _ppo_oracle(world):
    from oracle.ppo import PPOOracle
    rng = np.random.default_rng(11)
    pi = np.concatenate([omlp.init_mlp(rng, O, PPO_HID, A, init_w=1e-2), np.zeros(A, np.float32)])   # mean net | action_log_std
    o = PPOOracle(O, A, PPO_HID, pi, omlp.init_mlp(rng, O, PPO_HID, 1), mini_batch_size=PPO_MB, update_epoch=1, value_l2_reg=1e-3,
                  use_value_clip=True)
    o.grad_world = world
    return o


def _ppo_minibatches(steps):
    rng = np.random.default_rng(21)
    n = PPO_MB
    for _ in range(steps):
        yield dict(ob=rng.normal(0, 1, (n, O)).astype(np.float32), ac=rng.normal(0, 1, (n, A)).astype(np.float32),
                   R=rng.normal(0, 1, (n, 1)).astype(np.float32), V=rng.normal(0, 1, (n, 1)).astype(np.float32),
                   A=rng.normal(0, 1, (n, 1)).astype(np.float32), lp=rng.normal(-3, 0.3, (n, 1)).astype(np.float32))


def _ppo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = _ppo_oracle(world)
    o.allreduce = _gloo_sum
    norms = []
    for mb in _ppo_minibatches(4):
        sl = slice(rank * PPO_MB // world, (rank + 1) * PPO_MB // world)
        o.value_step(mb["ob"][sl], mb["R"][sl], mb["V"][sl])
        norms.append(o.policy_step(mb["ob"][sl], mb["ac"][sl], mb["A"][sl], mb["lp"][sl])[2])
    q.put((rank, o.pi.copy(), o.vf.copy(), norms))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(worker, world=2):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_ppo_minibatch_split_equals_single_process():
    """value step (clipped value loss + L2, applied once on the summed gradient) and policy step (clipped surrogate, the action_log_std
    gradient in the arena, clip_grad_norm_ on the SUMMED gradient) of four minibatches, each rank on its half of the rows"""
    res = _spawn(_ppo_worker)
    single = _ppo_oracle(1)
    norms = []
    for mb in _ppo_minibatches(4):
        single.value_step(mb["ob"], mb["R"], mb["V"])
        norms.append(single.policy_step(mb["ob"], mb["ac"], mb["A"], mb["lp"])[2])
    for r in res:
        np.testing.assert_array_equal(r[1], res[0][1])            # replicas took identical optimiser steps
        np.testing.assert_array_equal(r[2], res[0][2])
        np.testing.assert_allclose(r[1], single.pi, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[2], single.vf, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r[3], norms, rtol=1e-5)        # the norm that is clipped is the whole minibatch's, not the shard's


DISC_D, DISC_H, DISC_B = O + A, 32, 24


def _disc_oracle(world, blocks):
    from oracle.disc import DiscOracle
    rng = np.random.default_rng(31)
    flat = omlp.init_mlp(rng, DISC_D, [DISC_H] * blocks, 1)
    d = DiscOracle(DISC_D, DISC_H, flat, use_grad_pen=True, grad_pen_weight=8.0, num_layer_blocks=blocks)
    d.grad_world = world
    return d


def _disc_batches(steps):
    rng = np.random.default_rng(41)
    for _ in range(steps):
        yield (rng.normal(0.3, 1, (DISC_B, DISC_D)).astype(np.float32), rng.normal(-0.3, 1, (DISC_B, DISC_D)).astype(np.float32),
               rng.random((DISC_B, 1)).astype(np.float32))


def _disc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = [rank]
    for blocks in (2, 3):
        d = _disc_oracle(world, blocks)
        d.allreduce = _gloo_sum
        for xe, xp, eps in _disc_batches(3):
            sl = slice(rank * DISC_B // world, (rank + 1) * DISC_B // world)
            (d.train_step if blocks == 2 else d.train_step_blocks)(xe[sl], xp[sl], eps[sl])
        out.append(d.p.copy())
    q.put(tuple(out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_discriminator_split_equals_single_process():
    """BCE over 2B rows + WGAN-GP over B interpolates, each rank on B / 2 rows per class (and its own B / 2 interpolates): the two-block
    closed form and the any-depth chain"""
    res = _spawn(_disc_worker)
    for i, blocks in enumerate((2, 3)):
        single = _disc_oracle(1, blocks)
        for xe, xp, eps in _disc_batches(3):
            (single.train_step if blocks == 2 else single.train_step_blocks)(xe, xp, eps)
        for r in res:
            np.testing.assert_array_equal(r[1 + i], res[0][1 + i])
            np.testing.assert_allclose(r[1 + i], single.p, rtol=0, atol=2e-6)


def test_split_info_scales_trainer_rows():
    from ilswiss_amd.parallel import SplitInfo
    sp = SplitInfo(4, 2, None)
    assert sp.scale_rows(dict(mini_batch_size=32768, update_epoch=10), ("mini_batch_size",)) == dict(mini_batch_size=8192, update_epoch=10)
    out = sp.scale_rows(dict(disc_optim_batch_size=256, policy_optim_batch_size=256, policy_optim_batch_size_from_expert=0, mode="gail2"),
                        ("disc_optim_batch_size", "policy_optim_batch_size", "policy_optim_batch_size_from_expert"))
    assert out == dict(disc_optim_batch_size=64, policy_optim_batch_size=64, policy_optim_batch_size_from_expert=0, mode="gail2")
    with pytest.raises(ValueError):
        sp.scale_rows(dict(mini_batch_size=30), ("mini_batch_size",))
