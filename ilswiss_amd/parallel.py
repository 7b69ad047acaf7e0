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


class SplitRunStep:
    """Drives one split-run SAC step on this rank.

    `trainer` exposes the four phases (`set_batch`, `critic_backward`, `critic_update`, `actor_backward`,
    `actor_update`) and `grad_tensor(segment)` -> a torch tensor ALIASING the gradient arena segment
    (0 = critics, 1 = actor + alpha slot).  `sync()` (optional) drains the trainer's own stream before the
    collective reads the arena and is called again after it."""

    def __init__(self, trainer, group=None, sync=None):
        import torch.distributed as dist
        self.dist, self.trainer, self.group = dist, trainer, group
        self.sync = sync or (lambda: None)

    def _allreduce(self, seg):
        t = self.trainer.grad_tensor(seg)
        self.sync()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def train_step(self, local_batch, eps_next=None, eps_cur=None):
        tr = self.trainer
        tr.set_batch(local_batch, eps_next, eps_cur)
        tr.critic_backward()
        self._allreduce(0)
        tr.critic_update()
        tr.actor_backward()
        self._allreduce(1)
        tr.actor_update()
