"""What the bench's brackets cost: bnmf_run(20) alone, with torch.cuda.synchronize() behind it (the contract's bracket), and the synchronize of an idle device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
torch.cuda.set_device(0)
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(600, metrics=False)
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))
print(f"run(20)                          {med(lambda: e.run(20, metrics=True)):8.1f} us")
def both():
    e.run(20, metrics=True); torch.cuda.synchronize()
print(f"run(20) + torch.cuda.synchronize {med(both):8.1f} us")
print(f"torch.cuda.synchronize (idle)    {med(torch.cuda.synchronize):8.1f} us")
def bracket():
    torch.cuda.synchronize(); t0 = time.perf_counter(); e.run(20, metrics=True); torch.cuda.synchronize(); return time.perf_counter() - t0
print(f"sync; run(20); sync (timed part) {1e6 * float(np.median([bracket() for _ in range(15)])):8.1f} us")
print(f"run(2000) per 20                 {med(lambda: e.run(2000, metrics=True), 3) / 100:8.1f} us")
