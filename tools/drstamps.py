"""Time line of k_draw's E waves from in-kernel stamps (tools/bin/libbnmf_zslight.so: -DZSPROF -DZSLIGHT -DBNMF_FASTBUILD, the
builder's diagnostic build, never the product): when the waves start, how long each section takes, when they end."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E
E.LIB_PATH = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "tools/bin/libbnmf_zslight.so")
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = E.Engine(M, 20, prior="gamma", seed=1, window=1000)
apply_hyperprior_params(e, "gamma", M, 20)
print("created", flush=True); e.init(); print("init done", flush=True); e.run(20, metrics=False); print("20 iterations", flush=True); e.run(280, metrics=False); print("300 iterations", flush=True)
L = E.lib(); W = 4096
out = (C.c_ulonglong * (8 * W))()
L.bnmf_debug_draw.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
t0 = time.perf_counter(); e.run(400, metrics=False); dt = (time.perf_counter() - t0) / 400
L.bnmf_debug_draw(e._h, out, 0)
a = np.array(list(out), dtype=np.float64).reshape(W, 8)
a = a[(a[:, 7] > 0) & (a[:, 5] > 0) & (a[:, 4] > 0)]
# s_memtime counts shader-clock cycles PER XCD (the eight counters are not aligned): durations inside a wave come from it, the
# position of a wave in the launch from s_memrealtime (100 MHz, one counter for the device) taken at the wave's end
MHZ = 2100.0
TK = 1.0 / MHZ
a = a[a[:, 6] > a[:, 6].max() - 100 * 100]     # the last launch only (ends within 100 us of the newest)
end = (a[:, 6] - a[:, 6].min()) * 0.01
dur = lambda i, j: (a[:, i] - a[:, j]) * TK
st4 = end
st3 = st4 - dur(4, 3); st2 = st4 - dur(4, 2); st1 = st4 - dur(4, 1); stA = st4 - dur(4, 5); st0 = st4 - dur(4, 0)
sh = st0.min()
st0, stA, st1, st2, st3, st4 = [v - sh for v in (st0, stA, st1, st2, st3, st4)]
print(f"{dt * 1e6:.1f} us per iteration with the stamps build; {len(a)} E waves in the last launch")
def q(name, v):
    print(f"  {name:38s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
q("start of the wave (after the first)", st0)
q("loads + Gamma(shape,1) of E", stA - st0)
q("hyper_pre (Gamma of Beta) + table", st1 - stA)
q("wait for P (+ workgroup barrier)", st2 - st1)
q("divide + stores", st3 - st2)
q("Beta, tau, Alpha", st4 - st3)
q("end of the wave", st4)
# how many waves are inside [start, end) at each microsecond
T = min(int(np.ceil(st4.max())) + 1, 80)
print("  t(us): waves alive | before the wait | waiting | in the hyper sweep")
for t in range(0, T, 1):
    alive = ((st0 <= t) & (st4 > t)).sum(); pre = ((st0 <= t) & (st1 > t)).sum(); wt = ((st1 <= t) & (st2 > t)).sum(); hy = ((st2 <= t) & (st4 > t)).sum()
    print(f"  {t:4d}: {alive:5d} {pre:5d} {wt:5d} {hy:5d}")
