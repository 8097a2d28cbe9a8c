"""A/B of two builds of the library on the same box, alternating processes: us per 20-iteration bnmf_run call (the driver's bench shape:
record window 1000, metrics on) and us per iteration in 1000-iteration calls."""
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
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(300, metrics=False)
t20 = []
for _ in range(40):
    t0 = time.perf_counter(); e.run(20, metrics=True); t20.append((time.perf_counter() - t0) * 1e6)
t1k = []
for _ in range(3):
    t0 = time.perf_counter(); e.run(1000, metrics=True); t1k.append((time.perf_counter() - t0) / 1000 * 1e6)
print("%-34s run(20): min %.1f median %.1f us   run(1000): %.2f us/iter" % (sys.argv[1], min(t20), float(np.median(t20)), min(t1k)))
'''
for rnd in range(3):
    for l in libs:
        subprocess.run([sys.executable, "-c", code, l], check=True)
