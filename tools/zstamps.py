"""Wall-clock stamps (s_memrealtime) of the waves of three workgroups of one k_zalloc_sort launch (tools/bin/libbnmf_zsprof.so, -DZSPROF)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E
E.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libbnmf_zsprof.so")
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M, _, _ = synth_counts(96, G, 8, 20250218)
e = E.Engine(M, 20, prior="gamma", seed=1, window=0); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(50, metrics=False)
L = E.lib(); out = (C.c_ulonglong * (3 * 16 * 8))()
L.bnmf_debug_zstamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
(e.profile(1) if os.environ.get("ZSTAMP_PROFILE") else e.run(1, metrics=False))   # ZSTAMP_PROFILE=1: the kernel alone on the device (profile mode)
L.bnmf_debug_zstamps(e._h, out)
v = np.array(list(out), dtype=np.float64).reshape(3, 16, 8)
t0 = v[:, :, 0][v[:, :, 0] > 0].min()            # (stamps 6 / 7 of a wave without a task are stale: only values after t0 count)
names = ["start", "set-up done", "past set-up barrier", "tasks done", "past end barrier", "end", "first task: thresholds", "first task: quads"]
order = [0, 1, 2, 6, 7, 3, 4, 5]
print(f"G={G}: stamps in us from the first wave's start (block: waves' min / median / max)")
for b, bn in enumerate(("block 0", "block 128", "last block")):
    w = v[b]; live = w[:, 0] > 0
    for j in order:
        x = (w[live, j] - t0) / 100.0
        x = x[(w[live, j] > 0) & (x >= 0)]
        if len(x): print(f"  {bn:10s} {names[j]:26s} {x.min():7.2f} {np.median(x):7.2f} {x.max():7.2f}   ({len(x)} waves)")
