"""CPU suite: exp_spec -> variant grid (run_experiment.py:25-45 contract) and the spec files shipped in exp_specs/."""
import glob
import os

import yaml

from ilswiss_amd.launcher import variants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nested_variables_expand_to_the_cartesian_grid():
    spec = dict(meta_data=dict(script_path="run_scripts/x.py", exp_name="x", num_workers=2),
                variables=dict(seed=[0, 1, 2], sac_params=dict(reward_scale=[2.0, 4.0]), adv_irl_params=dict(grad_pen_weight=[8.0])),
                constants=dict(net_size=256, sac_params=dict(discount=0.99), adv_irl_params=dict(mode="gail2")))
    vs = list(variants(spec))
    assert len(vs) == 6 and [v["exp_id"] for v in vs] == list(range(6))
    assert {(v["seed"], v["sac_params"]["reward_scale"]) for v in vs} == {(s, r) for s in (0, 1, 2) for r in (2.0, 4.0)}
    for v in vs:
        assert v["sac_params"]["discount"] == 0.99 and v["adv_irl_params"] == dict(mode="gail2", grad_pen_weight=8.0)
        assert v["script_path"] == "run_scripts/x.py" and v["net_size"] == 256
    assert spec["constants"]["sac_params"] == dict(discount=0.99)   # the spec itself is left alone


def test_shipped_specs_name_existing_scripts_and_expand():
    specs = glob.glob(os.path.join(ROOT, "exp_specs", "*", "*.yaml"))
    assert len(specs) >= 5
    for path in specs:
        spec = yaml.safe_load(open(path))
        assert os.path.exists(os.path.join(ROOT, spec["meta_data"]["script_path"])), path
        v = next(variants(spec))
        assert "env_specs" in v and "seed" in v, path
