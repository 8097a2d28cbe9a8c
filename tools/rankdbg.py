import os, sys, ctypes as C
os.environ["BNMF_RANKDBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine, engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 12, 20250222)
e = Engine(M, 50, prior="gamma", seed=1, learning_rank=True, temperature=np.ones(8000), window=10)
apply_hyperprior_params(e, "gamma", M, 50); e.init(); e.run(5, metrics=False)
L = engine.lib(); L.bnmf_debug_rank.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
n = int(os.environ.get('BNMF_RANKGRID', '250')) * 16 * 8   # rank_grid workgroups (250 = ceil(ceil(G / 8) / 5) at G = 10,000: five blocks per workgroup; 179 with BNMF_RANKHALF=0)
buf = np.zeros(n, dtype=np.uint64)
g = L.bnmf_debug_rank(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
d = buf[:g * 16 * 8].reshape(g, 16, 8).astype(np.float64) / 100.0   # us
t0 = d[:, :, 0].min()
# stamps per factor.  Decision wave: 0 start of its step, 2 after gather + sum of factor n, 3 after the decision.  Compute wave 0: 1 after the
# step-ahead evaluation of factor n+1, 4 end of its step (behind the barrier, and behind the redo when the factor flipped)
for n_ in range(1, 8):
    s = d[:, n_, :]
    print(f"factor {n_}: step ahead done {np.median(s[:,1]-s[:,0]):5.2f} (max {np.max(s[:,1]-s[:,0]):5.2f})  gather + sum done {np.median(s[:,2]-s[:,0]):5.2f} (min {np.min(s[:,2]-s[:,0]):5.2f})  "
          f"decision {np.median(s[:,3]-s[:,2]):5.2f}  end of step {np.median(s[:,4]-s[:,0]):5.2f}  (us after the decision wave's start of the step)")
print("arrival of the last compute wave at the barrier, factors 1..8 (median us after the step's start):", [round(float(np.median(d[:, n_, 7] - d[:, n_, 0])), 2) for n_ in range(1, 9)])
print("poll rounds of the gather, factors 1..8 (median / max over workgroups):", [(float(np.median(buf[:g * 128].reshape(g, 16, 8)[:, n_, 5])), int(buf[:g * 128].reshape(g, 16, 8)[:, n_, 5].max())) for n_ in range(1, 9)])
per = np.median(d[:, 9, 0] - d[:, 1, 0]) / 8.0
print(f"per factor (steps 1..8): {per:5.2f} us")
e.close()
