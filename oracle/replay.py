"""Replay ring buffer semantics, restating rlkit/data_management/simple_replay_buffer.py
(ctor :17-68, add_sample :78-108, terminate_episode :125-132, _advance :228-237,
random_batch :239-253, _get_batch_using_indices :255-293, _get_segment :325-332,
sample_all_trajs :374-395, clear :397-442) and env_replay_buffer.py:7-49.
numpy; storage fp32 (the reference stores float64 on host and casts to fp32 at use,
rlkit/torch/core.py:124-127).  Test infrastructure.
"""
import numpy as np


class ReplayOracle:
    def __init__(self, cap, obs_dim, act_dim, random_seed=1995):
        self.cap, self.o, self.a = int(cap), obs_dim, act_dim
        self._rs = np.random.RandomState(random_seed)  # :20
        self.clear()

    def clear(self):  # :397-442
        c = self.cap
        self.obs = np.zeros((c, self.o), np.float32)
        self.next_obs = np.zeros((c, self.o), np.float32)
        self.act = np.zeros((c, self.a), np.float32)
        self.rew = np.zeros((c, 1), np.float32)
        self.term = np.zeros((c, 1), np.uint8)
        self.absorbing = np.zeros((c, 2), np.float32)   # :66-67
        self.top = 0
        self.size = 0
        self.cur_start = 0
        self.traj_endpoints = {}  # start -> end, [start, end), insertion-ordered

    def add_sample(self, obs, act, rew, terminal, next_obs, absorbing=None):  # :78-108
        t = self.top
        self.act[t] = act
        self.rew[t] = rew
        self.term[t] = terminal
        if absorbing is not None:   # :91-92
            self.absorbing[t] = absorbing
        if terminal:
            nxt = (t + 1) % self.cap
            self.traj_endpoints[self.cur_start] = nxt
            self.cur_start = nxt
        self.obs[t] = obs
        self.next_obs[t] = next_obs
        self._advance()

    def _advance(self):  # :228-237
        if self.top in self.traj_endpoints:
            del self.traj_endpoints[self.top]
        self.top = (self.top + 1) % self.cap
        if self.size < self.cap:
            self.size += 1

    def terminate_episode(self):  # :125-132
        if self.cur_start != self.top:
            self.traj_endpoints[self.cur_start] = self.top
            self.cur_start = self.top

    def add_rows(self, obs, act, rew, term, next_obs, ep_end=None):
        """n x add_sample, plus terminate_episode() after rows flagged in ep_end
        (what BaseAlgorithm._handle_path does per finished path, base_algorithm.py:399-423,509-519)."""
        n = len(rew)
        for i in range(n):
            self.add_sample(obs[i], act[i], np.ravel(rew)[i], int(np.ravel(term)[i]), next_obs[i])
            if ep_end is not None and ep_end[i]:
                self.terminate_episode()

    def add_path(self, path, absorbing=False, sample_action=None):
        """:134-216.  absorbing=True (the wrap_absorbing branch of adversarial IRL, adv_irl_exp_script.py:135-138): every stored
        terminal flag is False; a terminal transition is followed by (next_ob -> 0-state, absorbing [0,1]) and
        (0-state -> 0-state, absorbing [1,1]), both with a freshly sampled action and the terminal transition's reward."""
        for ob, act, rew, nob, term in zip(path["observations"], path["actions"], path["rewards"], path["next_observations"],
                                           path["terminals"]):
            if not absorbing:
                self.add_sample(ob, act, np.ravel(rew)[0], int(np.ravel(term)[0]), nob)
                continue
            self.add_sample(ob, act, np.ravel(rew)[0], 0, nob, absorbing=[0.0, 0.0])
            if np.ravel(term)[0]:
                self.add_sample(nob, sample_action(), np.ravel(rew)[0], 0, np.zeros_like(nob), absorbing=[0.0, 1.0])
                self.add_sample(np.zeros_like(nob), sample_action(), np.ravel(rew)[0], 0, np.zeros_like(nob), absorbing=[1.0, 1.0])
        self.terminate_episode()

    def num_steps_can_sample(self):  # :371-372
        return self.size

    def draw_indices(self, batch_size):  # :242
        return self._rs.randint(0, self.size, batch_size)

    def gather(self, idx):  # :255-293
        idx = np.asarray(idx)
        return dict(observations=self.obs[idx], actions=self.act[idx], rewards=self.rew[idx],
                    terminals=self.term[idx], next_observations=self.next_obs[idx], absorbing=self.absorbing[idx])

    def segment_indices(self, start, end):  # :325-332
        if start < end or end == 0:
            if end == 0:
                end = self.cap
            return list(range(start, end))
        return list(range(start, self.cap)) + list(range(0, end))

    def traj_sample_indices(self, start, end, samples_per_traj):  # _get_samples_from_traj, :334-347 (draws from the buffer's RandomState)
        inds = self.segment_indices(start, end)
        return list(self._rs.choice(inds, size=samples_per_traj, replace=len(inds) < samples_per_traj))

    def sample_all_trajs(self, samples_per_traj=None):  # :374-395
        if samples_per_traj is None:
            return [self.gather(self.segment_indices(s, e)) for s, e in self.traj_endpoints.items()]
        return [self.gather(self.traj_sample_indices(s, e, samples_per_traj)) for s, e in list(self.traj_endpoints.items())]

    def sample_trajs(self, num_trajs, samples_per_traj=None):  # :349-369
        keys = list(self.traj_endpoints.keys())
        starts = self._rs.choice(keys, size=num_trajs, replace=len(keys) < num_trajs)
        ends = [self.traj_endpoints[k] for k in starts]
        if samples_per_traj is None:
            return [self.gather(self.segment_indices(s, e)) for s, e in zip(starts, ends)]
        return [self.gather(self.traj_sample_indices(s, e, samples_per_traj)) for s, e in zip(starts, ends)]

    def get_all(self):  # :219-226
        return self.gather(np.arange(self.size))
