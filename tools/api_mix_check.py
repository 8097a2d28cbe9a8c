"""Chains on ONE GPU at once, each host thread mixing the other entry points into its run blocks (bnmf_map with bounds, bnmf_window, bnmf_get_array,
bnmf_set_array of P mid-chain, bnmf_assign): every chain must give the bits of the same call sequence made alone.
usage: python tools/api_mix_check.py [repetitions, default 5]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
cases = [dict(K=96, G=3000, N=20, kw=dict(prior="gamma"), conv=True),
         dict(K=96, G=900, N=8, kw=dict(prior="gamma", learning_rank=True, temperature=np.linspace(0.2, 1, 40)), conv=True),
         dict(K=96, G=1200, N=6, kw=dict(prior="truncnormal", MH=True), conv=True),
         dict(K=200, G=700, N=30, kw=dict(prior="exponential"), conv=True),
         dict(K=60, G=800, N=4, kw=dict(prior="truncnormal", likelihood="normal"), conv=False)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ref = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cosmic_v3.3.1_sbs.npz"))["P"]
def sequence(c, cid):
    M, _, _ = synth_counts(c["K"], c["G"], 4, 77 + cid)
    e = Engine(M, c["N"], seed=5, chain_id=cid, window=40, **c["kw"]); apply_hyperprior_params(e, c["kw"]["prior"], M, c["N"]); e.init()
    out = [e.run(45, converged=c["conv"])]
    m = e.map(30, 0.9)
    out += [m["P"], m["E"], m["P_lower"], m["E_upper"]]
    out.append(np.stack(e.window("E", 7)))
    out.append(e.run(17, converged=c["conv"]))
    P = e.get("P"); out.append(P)
    e.set("P", P * 1.5)                                  # a user value mid-chain: the side streams' work is re-issued
    out.append(e.run(23, converged=c["conv"]))
    if c["K"] == 96:
        keep = np.ones(c["N"], dtype=np.int32)
        a = e.assign(25, ref, keep=keep, MAP_P=m["P"], credible_interval=0.9)
        out += [a["votes"], a["assigned"].astype(np.float64)]
    out += [e.get("P"), e.get("E")]
    e.close()
    return [np.ascontiguousarray(np.nan_to_num(np.asarray(x, dtype=np.float64), nan=-1.0)) for x in out]
alone = [sequence(c, cid) for cid, c in enumerate(cases)]
bad = 0
for rep in range(reps):
    got, err = [None] * len(cases), [None] * len(cases)
    def work(i):
        try: got[i] = sequence(cases[i], i)
        except BaseException as ex: err[i] = ex   # noqa: BLE001
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in ths: t.start()
    for t in ths: t.join()
    for i in range(len(cases)):
        if err[i] is not None: print(f"repetition {rep}: chain {i} raised {err[i]!r}", flush=True); bad += 1; continue
        diff = [j for j, (a, b) in enumerate(zip(alone[i], got[i])) if a.shape != b.shape or not np.array_equal(a.view(np.uint64), b.view(np.uint64))]
        if diff: print(f"repetition {rep}: chain {i} differs from its solo sequence in outputs {diff}", flush=True); bad += 1
print(f"{reps} repetition(s) of {len(cases)} chains with mixed calls at once: {bad} differ")
sys.exit(1 if bad else 0)
