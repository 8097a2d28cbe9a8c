#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the repo's own CPU oracle (the reference is pure R with no
fixtures and cannot be run here: parity is unpinned w.r.t. R).  The vectors pin
the stream spec: any change to the oracle or the HIP engine that alters a bit shows up here.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def chain(prior, learning_rank, K, G, N, R_true, data_seed, seed, iters, temp=None):
    M, _, _ = synth_counts(K, G, R_true, data_seed)
    o = O.Oracle(M, N, prior=prior, learning_rank=learning_rank, seed=seed, temperature=temp, save_Z=True, nthreads=4)
    apply_hyperprior_params(o, prior, M, N)
    rows = [o.init()]
    rows += list(o.run(iters))
    names = ["P", "E", "A", "R", "ZsumK", "ZsumG"] + (["Alpha_p", "Beta_p", "Alpha_e", "Beta_e"] if prior == "gamma" else ["Lambda_p", "Lambda_e"])
    out = {nm: o.get(nm) for nm in names}
    out["metrics"] = np.array(rows)
    out["M"] = M
    out["Zsum_check"] = o.get("Z").sum(1)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "pg_k8_g6_n3.npz"), **chain("gamma", False, 8, 6, 3, 2, 11, 7, 50))
    np.savez_compressed(os.path.join(HERE, "pe_k8_g6_n3.npz"), **chain("exponential", False, 8, 6, 3, 2, 11, 7, 50))
    temp = np.concatenate([np.zeros(3), 10.0 ** np.linspace(-6, 0, 20), np.ones(40)])
    out = chain("gamma", True, 12, 10, 4, 2, 12, 9, 40, temp=temp)
    out["temperature"] = temp
    np.savez_compressed(os.path.join(HERE, "pg_sbfi_k12_g10_n4.npz"), **out)
    # known-answer vectors of the primitives
    rng = np.random.default_rng(1)
    x = np.concatenate([10 ** rng.uniform(-10, 10, 40), [1.0, 2.0, 0.5]])
    p = rng.uniform(0, 1, 40)
    np.savez_compressed(os.path.join(HERE, "math_kat.npz"), x=x, log=O.vec("log", x), exp=O.vec("exp", np.log(x)), xin_exp=np.log(x),
                        lgamma=O.vec("lgamma", x), digamma=O.vec("digamma", x), p=p, qnorm=O.vec("qnorm", p),
                        rgamma=O.rgamma(np.full(16, 6.5), 10.0, var=2, it=3), rtnorm0=O.rtnorm0(np.linspace(-3, 3, 16), 1.0, var=3, it=4),
                        ralpha=O.ralpha(np.full(16, 65.0), 6.0, 6.5, var=5, it=5)[0],
                        ralpha_fast=O.ralpha(np.full(16, 65.0), 6.0, 6.5, var=5, it=5, fast=True)[0])
    print("golden fixtures written to", HERE)
