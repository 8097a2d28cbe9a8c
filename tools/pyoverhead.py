"""What the Python wrapper adds to a bnmf_run(20) call: Engine.run against the bare ctypes call with prebuilt arguments."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.engine import lib
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(600, metrics=False)
def med(f, n=31):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))
L = lib(); out = np.empty((20, 11)); p = out.ctypes.data_as(C.POINTER(C.c_double)); h = e._h
a = med(lambda: e.run(20, metrics=True)); b = med(lambda: L.bnmf_run(h, 20, 0, p)); c = med(lambda: e.run(20, metrics=True)); d = med(lambda: L.bnmf_run(h, 20, 0, p))
print(f"Engine.run(20) {a:.1f} / {c:.1f} us   bare ctypes call {b:.1f} / {d:.1f} us")
print(f"np.empty((20, 11)) {med(lambda: np.empty((20, 11))):.2f} us; data_as {med(lambda: out.ctypes.data_as(C.POINTER(C.c_double))):.2f} us")
