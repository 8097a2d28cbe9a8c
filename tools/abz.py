"""A/B of two builds of the library on one box for configs 4 and 5: ms per iteration and us per allocation-kernel launch, alternating processes.
usage: python tools/abz.py libA.so libB.so   (CFGS=4,5  G5=columns of config 5)"""
import os, sys, subprocess
libs = sys.argv[1:]
code = '''
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bayesnmf_amd.engine as E
E.LIB_PATH = os.path.abspath(sys.argv[1])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
for cfg in os.environ.get("CFGS", "4,5").split(","):
    if cfg == "4":
        K, G, N, R, seed, iters, kw = 96, 10000, 50, 12, 20250222, 100, dict(learning_rank=True, temperature=np.ones(8000))
    else:
        K, G, N, R, seed, iters, kw = 1536, int(os.environ.get("G5", "25000")), 100, 30, 20250223, 6, {}
    M, _, _ = synth_counts(K, G, R, seed)
    e = Engine(M, N, prior="gamma", seed=1, window=2, **kw)
    apply_hyperprior_params(e, "gamma", M, N); e.init()
    e.run(max(3, iters // 3), metrics=False)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); e.run(iters, metrics=False); ts.append((time.perf_counter() - t0) / iters)
    prof = e.profile(3)
    print("%-32s cfg %s: ms/iter %.3f  k_zalloc %.1f us  k_rank %.1f us" % (sys.argv[1], cfg, min(ts) * 1e3, prof["k_zalloc"] * 1e3, prof.get("k_rank", 0) * 1e3), flush=True)
    e.close()
'''
for rnd in range(2):
    for l in libs:
        subprocess.run([sys.executable, "-c", code, l], check=True)
