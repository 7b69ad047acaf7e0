"""Variant-grid expansion of an exp_spec (rlkit/launchers/launcher_util.py build_nested_variant_generator as used by
run_experiment.py:25-45): `variables` holds (possibly nested) value lists, `constants` the fixed keys, `meta_data` is
merged in and every grid point gets its `exp_id`."""
import copy
import itertools


def _grid(variables):
    """[(key path, values)] for every list-valued leaf of the (possibly nested) `variables` mapping."""
    out = []
    for k, v in (variables or {}).items():
        if isinstance(v, dict):
            out += [((k,) + path, vals) for path, vals in _grid(v)]
        else:
            out.append(((k,), v if isinstance(v, list) else [v]))
    return out


def variants(spec):
    leaves = _grid(spec.get("variables"))
    for i, combo in enumerate(itertools.product(*[vals for _, vals in leaves])):
        v = copy.deepcopy(spec.get("constants") or {})
        for (path, _), val in zip(leaves, combo):
            d = v
            for k in path[:-1]:
                d = d.setdefault(k, {})
            d[path[-1]] = val
        v.update(copy.deepcopy(spec.get("meta_data") or {}))
        v["exp_id"] = i
        yield v
