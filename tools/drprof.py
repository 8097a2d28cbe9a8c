"""In-kernel section timers of k_draw's E workgroups (libbnmf_zsprof.so, built with -DZSPROF): ticks per wave per section,
and the same with parts of the work switched off (BNMF_DRDIAG: 1 no Alpha draw, 2 no Beta draw, 4 no E draw; timing only)."""
import ctypes as C
import os
import subprocess
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import bayesnmf_amd.engine as E
    E.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libbnmf_zsprof.so")
    from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts
    G = 10000
    M, _, _ = synth_counts(96, G, 8, 20250218)
    e = E.Engine(M, 20, prior="gamma", seed=1, window=0)
    apply_hyperprior_params(e, "gamma", M, 20)
    e.init()
    e.run(300, metrics=False)
    L = E.lib()
    W = 4096
    out = (C.c_ulonglong * (8 * W))()
    L.bnmf_debug_draw.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    L.bnmf_debug_draw(e._h, out, 1)
    n = 200
    t0 = time.perf_counter(); e.run(n, metrics=False); dt = time.perf_counter() - t0
    L.bnmf_debug_draw(e._h, out, 1)
    a = np.array(list(out), dtype=np.float64).reshape(W, 8)
    fb = a[W - 1, 4]
    cnt = a[W - 1].copy()
    a = a[a[:, 7] > 0]
    v = a.sum(axis=0)
    names = ["Gamma(shape,1) of E", "wait for P", "divide + stores", "hyper sweep (Beta, Alpha)", "", "whole"]
    print(f"BNMF_DRDIAG={os.environ.get('BNMF_DRDIAG', '0')}: {dt / n * 1e6:.1f} us per iteration (profile build); {len(a)} waves per launch; "
          f"{fb / n:.1f} lanes per launch into the general Alpha sampler")
    for i in (0, 1, 2, 3, 5):
        print(f"  {names[i]:44s} {v[i] / v[7]:10.0f} ticks per wave  {100 * v[i] / v[5]:5.1f} %")
    nw = len(a)
    print(f"  per launch: Newton {cnt[0] / n / nw:.2f} wave-iterations per wave, {cnt[3] / n / (64 * nw):.2f} lane-iterations per element; "
          f"Alpha {cnt[1] / n / nw:.2f} wave-passes, {cnt[2] / n / (64 * nw):.3f} attempts per element; "
          f"rgamma (E draw + Beta + P side) {cnt[5] / n / nw:.2f} wave-passes per E wave, {cnt[6] / n / (64 * nw):.3f} attempts per E element")
    w = a[:, 5] / a[:, 7]
    print(f"  whole, per wave: min {w.min():.0f} median {np.median(w):.0f} max {w.max():.0f}; start stamps of the last launch span {a[:, 6].max() - a[:, 6].min():.0f} ticks")
else:
    for dg in ("0",):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, BNMF_DRDIAG=dg))
