"""Host-side one-off setup that the reference does in R/setup.R (defaults) plus the
synthetic-count generator of SURVEY.md §8(d)."""
import numpy as np


def default_hyperprior_params(prior, M, N):
    """Defaults of get_default_*_hyperprior_params_ (R/setup.R:123-181), lower-case scalars."""
    mean_m = float(np.mean(M))
    if prior == "truncnormal":
        s = np.sqrt(mean_m / N)
        return dict(m_p=0.0, s_p=s, a_p=N + 1.0, b_p=np.sqrt(N), m_e=0.0, s_e=s, a_e=N + 1.0, b_e=np.sqrt(N))
    if prior == "exponential":
        return dict(a_p=10 * np.sqrt(N), b_p=10 * np.sqrt(mean_m), a_e=10 * np.sqrt(N), b_e=10 * np.sqrt(mean_m))
    if prior == "gamma":
        return dict(a_p=10 * np.sqrt(N), b_p=10.0, c_p=10 * np.sqrt(mean_m), d_p=10.0,
                    a_e=10 * np.sqrt(N), b_e=10.0, c_e=10 * np.sqrt(mean_m), d_e=10.0)
    raise ValueError(f"unknown prior {prior!r}")


def apply_hyperprior_params(chain, prior, M, N, user=None):
    """fill_hyperprior_params_ (R/setup.R:15-88): defaults, overridden by user entries; a
    lower-case scalar `a_p` is broadcast unless the upper-case matrix `A_p` is supplied."""
    hp = default_hyperprior_params(prior, M, N)
    user = dict(user or {})
    hp.update({k: v for k, v in user.items() if k[0].islower()})
    mats = {k: v for k, v in user.items() if k[0].isupper()}
    for k, v in hp.items():
        name = k[0].upper() + k[1:]
        if name in mats:
            chain.set(name, mats[name])
        else:
            chain.set(name, [float(v)])
    return hp


def synth_counts(K, G, R_true, seed, mean_total=4000.0):
    """Synthetic mutation-count matrix shaped like inst/extdata/example_data.rds
    (SURVEY.md §8d): Dirichlet(0.1) signatures, Dirichlet(1) mixing, Poisson totals."""
    rng = np.random.default_rng(seed)
    P = rng.dirichlet(0.1 * np.ones(K), size=R_true).T
    w = rng.dirichlet(np.ones(R_true), size=G).T
    tot = rng.poisson(mean_total * K / 96.0, size=G)
    E = np.stack([rng.multinomial(tot[g], w[:, g]) for g in range(G)], axis=1)
    M = rng.poisson(P @ E).astype(np.int32)
    return np.asfortranarray(M), P, E
