"""Several chains on ONE GPU at the same time (one host thread each) must give the same bits as each chain alone."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
cases = [dict(K=96, G=3000, N=20, kw=dict(prior="gamma"), window=50), dict(K=96, G=900, N=8, kw=dict(prior="gamma", learning_rank=True, temperature=np.linspace(0.2, 1, 40)), window=0),
         dict(K=96, G=1200, N=6, kw=dict(prior="truncnormal", MH=True), window=20), dict(K=200, G=700, N=30, kw=dict(prior="exponential"), window=10)]
def make(c, cid):
    M, _, _ = synth_counts(c["K"], c["G"], 4, 77 + cid)
    e = Engine(M, c["N"], seed=5, chain_id=cid, window=c["window"], **c["kw"]); apply_hyperprior_params(e, c["kw"]["prior"], M, c["N"]); e.init(); return e
n = 120
alone = []
for cid, c in enumerate(cases):
    e = make(c, cid); m = e.run(n, converged=True); alone.append((m.copy(), e.get("P").copy(), e.get("E").copy())); e.close()
es = [make(c, cid) for cid, c in enumerate(cases)]
out = [None] * len(es)
def work(i): out[i] = es[i].run(n, converged=True)
ths = [threading.Thread(target=work, args=(i,)) for i in range(len(es))]
for t in ths: t.start()
for t in ths: t.join()
bad = 0
for i, e in enumerate(es):
    same = np.array_equal(out[i][:, :9].view(np.uint64), alone[i][0][:, :9].view(np.uint64)) and np.array_equal(e.get("P").view(np.uint64), alone[i][1].view(np.uint64)) and np.array_equal(e.get("E").view(np.uint64), alone[i][2].view(np.uint64))
    print("chain", i, "identical to its solo run:", same); bad += not same
    e.close()
sys.exit(1 if bad else 0)
