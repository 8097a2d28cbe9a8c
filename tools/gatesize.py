"""Where the merged draw kernel + gated allocation kernel starts to pay: µs per iteration with BNMF_GATE=0 and 1 at several G
(K = 96, N = 20), alternating processes on the same box."""
import os, sys, subprocess
code = '''
import os, sys, time
sys.path.insert(0, ".")
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
G = int(sys.argv[1])
M, _, _ = synth_counts(96, G, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(300, metrics=False)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); e.run(1500, metrics=True); ts.append((time.perf_counter() - t0) / 1500 * 1e6)
print("G=%6d gate=%s  min %.1f  median %.1f us/iter" % (G, os.environ["BNMF_GATE"], min(ts), sorted(ts)[len(ts) // 2]), flush=True)
'''
for G in [int(g) for g in sys.argv[1:]] or [3000, 5000, 7000, 10000]:
    for gate in ("0", "1"):
        subprocess.run([sys.executable, "-c", code, str(G)], check=True, env=dict(os.environ, BNMF_GATE=gate))
