"""End-to-end A/B of two builds (separate processes would differ by box; here: same box, alternating)."""
import os, sys, subprocess
libs = sys.argv[1:]
code = '''
import os, sys, time
sys.path.insert(0, ".")
import bayesnmf_amd.engine as E
E.LIB_PATH = os.path.abspath(sys.argv[1])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(300, metrics=False)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); e.run(1000, metrics=True); ts.append((time.perf_counter() - t0) / 1000 * 1e6)
print("%s  min %.1f  median %.1f us/iter" % (sys.argv[1], min(ts), sorted(ts)[len(ts) // 2]))
'''
for rnd in range(2):
    for l in libs:
        subprocess.run([sys.executable, "-c", code, l], check=True)
