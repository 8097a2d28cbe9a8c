"""Does a third wave per SIMD pay?  At K = 64 the per-wave slab is small enough for 12 waves per CU: allocation kernel alone
(bnmf_profile) and whole iteration (bnmf_run), 8 against 12 waves per CU, interleaved in one process.  K from env OCC_K."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
K = int(os.environ.get("OCC_K", "64"))
M, _, _ = synth_counts(K, 10000, 8, 20250218)
def mk(env):
    for k in ("BNMF_ZGRID", "BNMF_ZW"): os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(50, metrics=False)
    return e
variants = [("8 waves/CU (zw8 g256)", {"BNMF_ZW": "8", "BNMF_ZGRID": "256"}), ("8 waves/CU (zw4 g512)", {"BNMF_ZW": "4", "BNMF_ZGRID": "512"}),
            ("12 waves/CU (zw4 g768)", {"BNMF_ZW": "4", "BNMF_ZGRID": "768"}), ("12 waves/CU (zw6 g512)", {"BNMF_ZW": "6", "BNMF_ZGRID": "512"}),
            ("16 waves/CU (zw4 g1024)", {"BNMF_ZW": "4", "BNMF_ZGRID": "1024"})]
eng = [(n, mk(env)) for n, env in variants]
alone = {n: [] for n, _ in eng}; e2e = {n: [] for n, _ in eng}
for rnd in range(4):
    for n, e in eng:
        alone[n].append(e.profile(30)["k_zalloc"] * 1e3)
        e.run(50, metrics=False)
        t0 = time.perf_counter(); e.run(500, metrics=True); e2e[n].append((time.perf_counter() - t0) / 500 * 1e6)
for n, _ in eng: print(f"K={K} {n:26s} kernel alone {np.median(alone[n]):7.1f} us   iteration {np.median(e2e[n]):7.1f} us", flush=True)
