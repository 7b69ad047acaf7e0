#!/usr/bin/env python
"""Normalise expert demonstrations with the statistics of their own observations — the job of the reference's
scripts/normalize_exp_demos.py:24-131 (`get_normalized`: mean / std over the first `size` rows, std == 0 -> 1; observations and
next_observations of every split get the TRAIN split's statistics; actions untouched, NORMALIZE_ACTS = False), on the PKL demo
format the run scripts read today (adv_irl_exp_script.py:51-60: a pickled list of path dicts; the reference script itself still
expects the older {"train": buffer, "test": buffer} layout, which is why it is restated here on the list format).

    python scripts/normalize_exp_demos.py <listing key>  [--listing demos_listing.yaml] [--train-frac 1.0] [--out-dir demos]

Writes <out-dir>/norm_<key>.pkl = dict(train=[paths], test=[paths], obs_mean, obs_std, acts_mean=None, acts_std=None) and
prints the listing entry to add.  The statistics are what ScaledEnv(obs_mean, obs_std) takes (rlkit/envs/wrappers.py:53-131).
"""
import argparse
import os
import pickle

import numpy as np
import yaml


def get_normalized(data, size, mean=None, std=None, return_stats=False):   # normalize_exp_demos.py:24-38
    if mean is None:
        mean = np.mean(data[:size], axis=0, keepdims=True)
    if std is None:
        std = np.std(data[:size], axis=0, keepdims=True)
        std = np.where(std == 0, np.ones(std.shape), std)   # a constant axis must not divide by zero
    if return_stats:
        return (data - mean) / std, mean, std
    return (data - mean) / std


def normalize_paths(train, test=()):
    """-> (train', test', obs_mean [1,o], obs_std [1,o]); paths are copied, only the two observation keys change."""
    obs = np.vstack([p["observations"] for p in train])
    _, mean, std = get_normalized(obs, len(obs), return_stats=True)

    def apply(paths):
        out = []
        for p in paths:
            q = dict(p)
            q["observations"] = get_normalized(np.asarray(p["observations"]), 0, mean=mean, std=std)
            q["next_observations"] = get_normalized(np.asarray(p["next_observations"]), 0, mean=mean, std=std)
            out.append(q)
        return out
    return apply(train), apply(test), mean, std


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("key")
    ap.add_argument("--listing", default="demos_listing.yaml")
    ap.add_argument("--train-frac", type=float, default=1.0)
    ap.add_argument("--out-dir", default="demos")
    args = ap.parse_args()
    with open(args.listing) as f:
        listing = yaml.safe_load(f)
    with open(listing[args.key]["file_paths"][0], "rb") as f:
        paths = pickle.load(f)
    n_train = max(1, int(round(len(paths) * args.train_frac)))
    train, test, mean, std = normalize_paths(paths[:n_train], paths[n_train:])
    os.makedirs(args.out_dir, exist_ok=True)
    out = os.path.join(args.out_dir, f"norm_{args.key}.pkl")
    with open(out, "wb") as f:
        pickle.dump(dict(train=train, test=test, obs_mean=mean, obs_std=std, acts_mean=None, acts_std=None), f)
    print("Observations:\nMean:\n", mean, "\nStd:\n", std)
    chk = np.vstack([p["observations"] for p in train])
    print("Post normalisation check (train obs): mean", np.mean(chk, 0), "std", np.std(chk, 0))
    print(f"\nwrote {out}\nRemember to add the new normalized demos to your expert listings:\n"
          f"norm_{args.key}:\n  description: \"{args.key}, observations normalised\"\n  file_paths: [\n    ./{out}\n  ]")


if __name__ == "__main__":
    main()
