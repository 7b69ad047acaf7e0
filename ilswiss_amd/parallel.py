"""Multi-GPU plumbing.  One process per GPU, `torch.distributed` (backend "nccl" == RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  torch is used for the process group and the collective only.

Two ways the path shards:
  1. independent replicas (seeds / runs): no data-path collective at all — `shard_items`;
  2. one run split over G ranks: every rank holds a full replica of the parameters, processes B/G rows of
     the batch with mean-loss gradients pre-scaled by 1/(B*G) (`grad_world`), and the flat gradient arena is
     all-reduced (sum) between backward and the optimiser step — `SplitRunStep`.  Critic and actor
     gradients go in separate messages because the actor loss uses the post-update critics
     (sac_alpha.py:142-146); the alpha gradient rides in the actor message (arena slot).
"""


def shard_items(items, world_size, rank):
    """Contiguous, balanced partition of independent work items (seeds, runs) over ranks."""
    items = list(items)
    n, base, extra = len(items), len(items) // world_size, len(items) % world_size
    start = rank * base + min(rank, extra)
    return items[start:start + base + (1 if rank < extra else 0)]


def shard_batch(batch, world_size, rank):
    """Rows [rank*B/G, (rank+1)*B/G) of every array of a batch dict (B must divide evenly)."""
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        if B % world_size:
            raise ValueError(f"batch of {B} rows does not split over {world_size} ranks")
        n = B // world_size
        out[k] = v[rank * n:(rank + 1) * n]
    return out


def comm_init(ctx, group=None):
    """Give `ctx` (an ilswiss_amd Context) an RCCL communicator spanning the torch.distributed group: rank 0 draws the id
    (ilsx_comm_unique_id), torch.distributed carries its 128 bytes (any backend — this is the only thing the process group
    is used for), every rank calls ilsx_comm_init.  From then on the library's own all-reduces run on the ctx stream."""
    import ctypes as C

    import torch.distributed as dist

    from . import _lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        _lib.check(ctx.lib.ilsx_comm_unique_id(buf))
    box = [bytes(buf)]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    buf = (C.c_uint8 * 128).from_buffer_copy(box[0])
    _lib.check(ctx.lib.ilsx_comm_init(ctx.h, buf, world, rank))
    return world, rank


def ensure_comm(ctx, group=None):
    """comm_init unless the ctx already carries a communicator (trainers of one run share their ctx — and its communicator)."""
    import ctypes as C
    n = C.c_int()
    ctx.lib.ilsx_comm_info(ctx.h, C.byref(n), None)
    if n.value == 0:
        comm_init(ctx, group)


class SplitInfo:
    """What a run script needs to build its share of a split run: `world`, `rank`, and the `rl_alg_params` / trainer kwargs scaled to it."""

    def __init__(self, world, rank, dist):
        self.world, self.rank, self.dist = int(world), int(rank), dist

    def scale(self, alg):
        """This rank's share of the loop: counts that the reference states in env steps / rows of the whole run are divided by G (every rank
        steps env_num / G envs and counts ITS env steps); the batch is B / G rows; the keys are removed that only rank 0 acts on."""
        G, out = self.world, dict(alg)
        out.pop("split_ranks", None)
        for k in ("batch_size", "num_steps_per_epoch", "num_steps_between_train_calls", "min_steps_before_training", "replay_buffer_size"):
            if k in out:
                if int(out[k]) % G:
                    raise ValueError(f"rl_alg_params.{k}={out[k]} does not split over split_ranks={G}")
                out[k] = int(out[k]) // G
        out["split_world"] = G
        out["split_agree"] = self.agree
        return out

    def scale_rows(self, params, keys):
        """`params` with the row counts under `keys` (batch sizes of a trainer: ppo_params.mini_batch_size, adv_irl_params.*_batch_size)
        divided by G: every rank processes its share of each batch."""
        G, out = self.world, dict(params)
        for k in keys:
            if k in out and out[k]:
                if int(out[k]) % G:
                    raise ValueError(f"{k}={out[k]} does not split over split_ranks={G}")
                out[k] = int(out[k]) // G
        return out

    def agree(self, flag):
        """True only if every rank says so (a rank that trained alone would wait in the all-reduce for ever)."""
        if self.dist is None or self.world == 1:
            return bool(flag)
        import torch
        t = torch.tensor([1.0 if flag else 0.0], device="cuda" if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)


class SplitRunStep:
    """Drives one split-run SAC step on this rank.

    Two engines behind the same call:
      * a libilsx trainer (`ilswiss_amd.SoftActorCritic(grad_world=G)`) whose ctx carries an RCCL communicator
        (`comm_init`): `train_step` / `train_from_replay` are ONE library call — critic-backward -> ncclAllReduce -> critic
        Adam -> actor-backward -> ncclAllReduce -> actor Adam, all enqueued on the ctx stream (no host synchronisation, no
        second stream);
      * any object with the four phases (`set_batch`, `critic_backward`, `critic_update`, `actor_backward`,
        `actor_update`) and `grad_tensor(segment)` -> a torch tensor ALIASING the gradient arena segment (0 = critics,
        1 = actor + alpha slot): the collective is torch.distributed's.  This is what the CPU tests drive (gloo + the numpy
        oracle).  For a device trainer on this route the trainer's stream is drained before the collective reads the arena
        and torch's stream is drained before the update phase is enqueued (`sync` defaults to the trainer's `ctx.sync`)."""

    def __init__(self, trainer, group=None, sync=None, use_library_comm=None):
        import torch.distributed as dist
        self.dist, self.trainer, self.group = dist, trainer, group
        ctx = getattr(trainer, "ctx", None)
        if use_library_comm is None:
            use_library_comm = ctx is not None and hasattr(ctx, "lib") and hasattr(trainer, "train_from_replay")
        self.library = bool(use_library_comm)
        if self.library:
            import ctypes as C
            n = C.c_int()
            ctx.lib.ilsx_comm_info(ctx.h, C.byref(n), None)
            if n.value == 0:
                comm_init(ctx, group)
        if sync is None and ctx is not None and hasattr(ctx, "sync"):
            sync = ctx.sync
        self.sync = sync or (lambda: None)

    def _allreduce(self, seg):
        t = self.trainer.grad_tensor(seg)
        self.sync()                                   # the trainer's stream has produced the arena
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()   # ... and the collective is done before the update is enqueued

    def train_step(self, local_batch, eps_next=None, eps_cur=None):
        tr = self.trainer
        if self.library:
            return tr.train_step(local_batch, eps_next, eps_cur)
        tr.set_batch(local_batch, eps_next, eps_cur)
        tr.critic_backward()
        self._allreduce(0)
        tr.critic_update()
        tr.actor_backward()
        self._allreduce(1)
        tr.actor_update()

    def train_from_replay(self, replay_shard, n_steps, local_batch_size):
        """n fused steps, each rank drawing its B/G rows from ITS replay shard (stratified-uniform over the union, §8e).
        The policy noise (eps_next / eps_cur) of a rank's rows is keyed by its ctx seed: give every rank its OWN ctx seed (the networks'
        init seeds stay identical) — with one ctx seed for all ranks the shards draw identical eps rows and the B-row batch carries
        G-fold correlated noise (bench.py:split_run_leg mixes the rank into the ctx seed)."""
        if self.library:
            return self.trainer.train_from_replay(replay_shard, n_steps, local_batch_size)
        # host engines (the CPU tests: gloo + the numpy oracle): the same loop, one random_batch of the local shard per step
        for _ in range(int(n_steps)):
            b = replay_shard.random_batch(int(local_batch_size))
            noise = getattr(self.trainer, "draw_noise", None)
            e1, e2 = noise(int(local_batch_size)) if noise else (None, None)
            self.train_step(b, e1, e2)
