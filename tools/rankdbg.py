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
n = int(os.environ.get('BNMF_RANKGRID', '157')) * 16 * 8   # rank_grid workgroups (157 = ceil(ceil(G / 8) / 8) at G = 10,000)
buf = np.zeros(n, dtype=np.uint64)
g = L.bnmf_debug_rank(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
d = buf[:g * 16 * 8].reshape(g, 16, 8).astype(np.float64) / 100.0   # us
t0 = d[:, :, 0].min()
# stamps per factor: 0 start, 1 after the step-ahead evaluation of factor n+1, 2 after the gather of factor n, 3 after tree + decision, 4 end
for n_ in range(1, 8):
    s = d[:, n_, :]
    print(f"factor {n_}: step ahead {np.median(s[:,1]-s[:,0]):5.2f} (max {np.max(s[:,1]-s[:,0]):5.2f})  gather {np.median(s[:,2]-s[:,1]):5.2f} (min {np.min(s[:,2]-s[:,1]):5.2f})  "
          f"tree + decide {np.median(s[:,3]-s[:,2]):5.2f}  tail (redo when flipped) {np.median(s[:,4]-s[:,3]):5.2f}  whole {np.median(s[:,4]-s[:,0]):5.2f}")
e.close()
