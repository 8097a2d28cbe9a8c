import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, save_Z=bool(int(os.environ.get("SAVE_Z", "0"))), window=int(os.environ.get("WINDOW", "1000")))
apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(int(os.environ.get("ITERS", "20")), metrics=False); e.close()
