"""bayesnmf_amd — MI355X-native Gibbs engine for Bayesian (Poisson) NMF.

Drop-in for the sampling loop of jennalandy/bayesNMF: the host mirror of the R interface
lives in sampler.py (bayesNMF(), bayesNMF_sampler, new_convergence_control()); engine.py binds
the C ABI of libbnmf.so (include/bnmf.h), whose kernels are hand-written HIP for gfx950.
"""
from .engine import Engine, BnmfError, lib, device_count, device_info  # noqa: F401
from .setup import default_hyperprior_params, synth_counts  # noqa: F401
