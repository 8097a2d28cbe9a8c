"""A/B of builds of the library on one box, steady state only: us per iteration over 4,000-iteration calls, builds alternating (each its own process), five rounds."""
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
e = Engine(M, 20, prior="gamma", seed=1, window=int(os.environ.get("WINDOW", "1000"))); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(1500, metrics=False)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); e.run(4000, metrics=True); ts.append((time.perf_counter() - t0) / 4000 * 1e6)
print("%-34s %s  min %.2f" % (sys.argv[1], " ".join("%.2f" % t for t in ts), min(ts)))
'''
# a build may carry environment settings: path@NAME=value@NAME2=value2
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for l in libs:
        parts = l.split("@")
        env = dict(os.environ, **dict(p.split("=", 1) for p in parts[1:]))
        print(("  ".join(parts[1:]) + "  ") if len(parts) > 1 else "", end="", flush=True)
        subprocess.run([sys.executable, "-c", code, parts[0]], check=True, env=env)
