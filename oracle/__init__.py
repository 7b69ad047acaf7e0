"""oracle/ — CPU restatement of the ILSwiss hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Every function here restates (in plain numpy fp32, or plain C for the env stepper) the algorithm of
one reference function and cites the reference file:line it follows.  The restatement is pinned
against golden vectors produced by running the reference itself (tools/make_golden.py →
tests/golden/*.npz; see tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import anything from
this package, and only as the checker / the timed CPU baseline.  The product path
(ilswiss_amd/) never imports it and has no CPU fallback: it raises if libilsx.so is missing.

Pinning status
  * trainer / model / buffer arithmetic (SAC-alpha, SAC-V, TD3, PPO, disc, replay, RunningMeanStd,
    action map): PINNED by tests/golden/*.npz generated from the reference in the survey container.
  * env physics (oracle/planar_env.c): **parity unpinned** — the reference delegates dynamics to
    MuJoCo 2.1 (mujoco-py, requirements.txt:12) + gym 0.22 XML models, none of which exist here.
    Only the in-tree reward / termination / reset-noise formulas (rlkit/envs/mujoco/*.py) and the
    README random-policy known answers (README.md:158-169) anchor it.
"""
