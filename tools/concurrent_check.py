"""Several chains on ONE GPU at the same time (one host thread each) must give the same bits as each chain alone.
usage: python tools/concurrent_check.py [repetitions of the concurrent part, default 1] — every repetition creates its handles anew (six
chains: gated fixed rank, rank learning, MH after and before convergence, exponential prior with K > 128, Normal likelihood; CONC_BIG=1: six
full-size chains — two gated fixed-rank ones, config 3 twice, config 4, N = 100)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("BNMF_TOOL_LIB"):                      # another build of the library (A/B)
    import bayesnmf_amd.engine as _E
    _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TOOL_LIB"])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
cases = [dict(K=96, G=3000, N=20, kw=dict(prior="gamma"), window=50, conv=True),
         dict(K=96, G=900, N=8, kw=dict(prior="gamma", learning_rank=True, temperature=np.linspace(0.2, 1, 40)), window=0, conv=True),
         dict(K=96, G=1200, N=6, kw=dict(prior="truncnormal", MH=True), window=20, conv=True),
         dict(K=200, G=700, N=30, kw=dict(prior="exponential"), window=10, conv=True),
         dict(K=96, G=1500, N=5, kw=dict(prior="truncnormal", MH=True), window=0, conv=False),
         dict(K=60, G=800, N=4, kw=dict(prior="truncnormal", likelihood="normal"), window=5, conv=False)]
if os.environ.get("CONC_BIG"):                             # full-size chains: two gated fixed-rank chains, config 3, config 4, N = 100
    cases = [dict(K=96, G=10000, N=20, kw=dict(prior="gamma"), window=50, conv=True),
             dict(K=96, G=10000, N=20, kw=dict(prior="exponential"), window=0, conv=True),
             dict(K=96, G=5000, N=20, kw=dict(prior="truncnormal", MH=True), window=20, conv=True),
             dict(K=96, G=10000, N=50, kw=dict(prior="gamma", learning_rank=True, temperature=np.ones(400)), window=0, conv=True),
             dict(K=96, G=5000, N=20, kw=dict(prior="truncnormal", MH=True), window=0, conv=False),
             dict(K=256, G=4000, N=100, kw=dict(prior="gamma"), window=0, conv=True)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def make(c, cid):
    M, _, _ = synth_counts(c["K"], c["G"], 4, 77 + cid)
    e = Engine(M, c["N"], seed=5, chain_id=cid, window=c["window"], **c["kw"]); apply_hyperprior_params(e, c["kw"]["prior"], M, c["N"]); e.init(); return e
n = 120
alone = []
for cid, c in enumerate(cases):
    e = make(c, cid); m = e.run(n, converged=c["conv"]); alone.append((m.copy(), e.get("P").copy(), e.get("E").copy())); e.close()
bad = 0
for rep in range(reps):
    es = [make(c, cid) for cid, c in enumerate(cases)]
    out = [None] * len(es)
    def work(i): out[i] = np.concatenate([es[i].run(n // 3, converged=cases[i]["conv"]), es[i].run(n - n // 3, converged=cases[i]["conv"])])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(es))]
    for t in ths: t.start()
    for t in ths: t.join()
    for i, e in enumerate(es):
        rows = [r for r in range(n) if not np.array_equal(out[i][r, :9].view(np.uint64), alone[i][0][r, :9].view(np.uint64))]
        same = not rows and np.array_equal(e.get("P").view(np.uint64), alone[i][1].view(np.uint64)) and np.array_equal(e.get("E").view(np.uint64), alone[i][2].view(np.uint64))
        if reps == 1 or not same: print(f"repetition {rep}: chain {i} identical to its solo run: {same}" + (f" (metric rows that differ: {rows[:6]})" if rows else ""), flush=True)
        bad += not same
        e.close()
print(f"{reps} repetition(s) of {len(cases)} chains at once: {bad} chain runs differ from their solo runs")
sys.exit(1 if bad else 0)
