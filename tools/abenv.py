"""End-to-end A/B of environment switches on ONE library, alternating processes on the same box.
usage: tools/abenv.py "A=1 B=2" "A=0" ...   (each argument: space-separated assignments; "-" = none; ABCFG=4 in the
environment: BASELINE config 4 instead of the metric configuration)"""
import os, sys, subprocess
code = '''
import os, sys, time
sys.path.insert(0, ".")
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
import numpy as np
if os.environ.get("ABCFG") == "4":      # BASELINE config 4: learned rank, N = 50
    M, _, _ = synth_counts(96, 10000, 8, 20250218)
    e = Engine(M, 50, prior="gamma", seed=1, learning_rank=True, rank_method="SBFI", temperature=np.ones(4000)); apply_hyperprior_params(e, "gamma", M, 50); n_it = 100
else:
    M, _, _ = synth_counts(96, 10000, 8, 20250218)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); n_it = 1000
e.init(); e.run(n_it // 3, metrics=False)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); e.run(n_it, metrics=True); ts.append((time.perf_counter() - t0) / n_it * 1e6)
print("%-30s min %.1f  median %.1f us/iter" % (sys.argv[1], min(ts), sorted(ts)[len(ts) // 2]), flush=True)
'''
for rnd in range(2):
    for a in sys.argv[1:]:
        env = dict(os.environ)
        if a != "-":
            for kv in a.split(): k, v = kv.split("="); env[k] = v
        subprocess.run([sys.executable, "-c", code, a], check=True, env=env)
