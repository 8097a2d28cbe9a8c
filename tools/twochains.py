"""Aggregate throughput of C independent chains sharing ONE GPU (one host thread per chain; ctypes releases the GIL)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
for C in (1, 2, 3, 4):
    es = []
    for c in range(C):
        e = Engine(M, 20, prior="gamma", seed=1, chain_id=c); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(200, metrics=False); es.append(e)
    n = 1500
    ths = [threading.Thread(target=lambda e=e: e.run(n, metrics=True)) for e in es]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"{C} chain(s) on one GPU: {C * n / dt:8.1f} it/s aggregate, {n / dt:8.1f} it/s per chain", flush=True)
    for e in es: e.close()
