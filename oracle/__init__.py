"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  bayesnmf_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NMETRIC = 11
METRIC_NAMES = ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC",
                "rank", "temp", "P_mean_acceptance_rate", "E_mean_acceptance_rate"]

# array ids (same numbering as include/bnmf.h)
IDS = dict(P=0, E=1, A=2, R=3, Z=4, ZsumK=5, ZsumG=6, sigmasq=7,
           Alpha_p=10, Beta_p=11, Alpha_e=12, Beta_e=13, Mu_p=14, Sigmasq_p=15, Mu_e=16,
           Sigmasq_e=17, Lambda_p=18, Lambda_e=19, Alpha=20, Beta=21,
           A_p=30, B_p=31, C_p=32, D_p=33, M_p=34, S_p=35,
           A_e=40, B_e=41, C_e=42, D_e=43, M_e=44, S_e=45,
           P_acceptance_rate=50, E_acceptance_rate=51, Mhat=60)
LIKELIHOOD = dict(poisson=0, normal=1)
PRIOR = dict(truncnormal=0, exponential=1, gamma=2)
RANK_METHOD = dict(SBFI=0, BFI=1)


class OrcConfig(C.Structure):
    _fields_ = [("K", C.c_int32), ("G", C.c_int32), ("N", C.c_int32),
                ("likelihood", C.c_int32), ("prior", C.c_int32), ("MH", C.c_int32),
                ("learning_rank", C.c_int32), ("rank_method", C.c_int32),
                ("save_Z", C.c_int32), ("nthreads", C.c_int32),
                ("seed", C.c_uint64), ("chain_id", C.c_uint32), ("_pad", C.c_uint32),
                ("temperature", C.POINTER(C.c_double)), ("n_temperature", C.c_int64)]


def build(force=False):
    if os.environ.get("ORACLE_ASAN") == "1":           # tests/test_oracle_asan.py: the sanitizer build (needs libasan preloaded)
        subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
        return os.path.join(_HERE, "liboracle_asan.so")
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bnmf_oracle.c", "orc_math.h", "orc_samplers.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig), ip]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_array.argtypes = [C.c_void_p, C.c_int, dp, C.c_long]
        L.orc_get_array.argtypes = [C.c_void_p, C.c_int, dp, C.c_long]
        L.orc_init.argtypes = [C.c_void_p, dp]
        L.orc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.orc_get_iter.argtypes = [C.c_void_p]
        L.orc_set_M.argtypes = [C.c_void_p, ip]
        L.orc_set_fresh_mhat.argtypes = [C.c_void_p, C.c_int]
        L.orc_t_step.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        for f in ("log", "exp", "lgamma", "digamma", "qnorm", "log_pnorm"):
            fn = getattr(L, "orc_t_" + f)
            fn.restype = C.c_double
            fn.argtypes = [C.c_double]
        L.orc_t_u52.restype = C.c_double
        L.orc_t_u52.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_t_philox.argtypes = [up, up, up]
        L.orc_t_philox7.argtypes = [up, up, up]
        L.orc_t_canon_sum.restype = C.c_double
        L.orc_t_canon_sum.argtypes = [dp, C.c_long, C.c_long, C.c_int]
        hdr = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_t_rgamma.argtypes = hdr + [dp, dp, dp, C.c_long]
        L.orc_t_rtnorm0.argtypes = hdr + [dp, dp, dp, C.c_long]
        L.orc_t_rnorm.argtypes = hdr + [dp, C.c_long]
        L.orc_t_ralpha.argtypes = hdr + [dp, dp, dp, dp, ip, C.c_long]
        L.orc_t_ralpha_fast.argtypes = hdr + [dp, dp, dp, dp, ip, C.c_long]
        L.orc_t_alpha_h.argtypes = [C.c_double, C.c_double, C.c_double, dp, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def vec(fname, x):
    """Apply a scalar oracle math function element-wise."""
    fn = getattr(lib(), "orc_t_" + fname)
    x = np.asarray(x, dtype=np.float64)
    return np.array([fn(float(v)) for v in x.ravel()]).reshape(x.shape)


def philox(ctr, key, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    (lib().orc_t_philox7 if rounds == 7 else lib().orc_t_philox)(c, k, o)
    return [int(v) for v in o]


def canon_sum(x, W, stride=1, L=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if L is None:
        L = (x.size + stride - 1) // stride
    return lib().orc_t_canon_sum(_dp(x), L, stride, W)


def rgamma(shape, rate, seed=1, chain=0, var=2, elem0=0, it=1):
    shape = np.ascontiguousarray(shape, dtype=np.float64)
    rate = np.ascontiguousarray(np.broadcast_to(rate, shape.shape), dtype=np.float64)
    out = np.empty_like(shape)
    lib().orc_t_rgamma(seed, chain, var, elem0, it, _dp(shape), _dp(rate), _dp(out), shape.size)
    return out


def rtnorm0(mu, sd, seed=1, chain=0, var=2, elem0=0, it=1):
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    sd = np.ascontiguousarray(np.broadcast_to(sd, mu.shape), dtype=np.float64)
    out = np.empty_like(mu)
    lib().orc_t_rtnorm0(seed, chain, var, elem0, it, _dp(mu), _dp(sd), _dp(out), mu.size)
    return out


def rnorm(n, seed=1, chain=0, var=8, elem0=0, it=1):
    out = np.empty(n)
    lib().orc_t_rnorm(seed, chain, var, elem0, it, _dp(out), n)
    return out


def ralpha(c, tau, xprev, seed=1, chain=0, var=5, elem0=0, it=1, fast=False):
    """fast=False: the general 3-tangent sampler; fast=True: the Gamma-envelope sampler the sweep uses (falls back to the general one)."""
    c = np.ascontiguousarray(c, dtype=np.float64)
    tau = np.ascontiguousarray(np.broadcast_to(tau, c.shape), dtype=np.float64)
    xprev = np.ascontiguousarray(np.broadcast_to(xprev, c.shape), dtype=np.float64)
    out = np.empty_like(c)
    att = np.empty(c.size, dtype=np.int32)
    (lib().orc_t_ralpha_fast if fast else lib().orc_t_ralpha)(seed, chain, var, elem0, it, _dp(c), _dp(tau), _dp(xprev), _dp(out),
                                                               att.ctypes.data_as(C.POINTER(C.c_int32)), c.size)
    return out, att


def alpha_h(x, c, tau):
    h, hp = C.c_double(), C.c_double()
    lib().orc_t_alpha_h(x, c, tau, C.byref(h), C.byref(hp))
    return h.value, hp.value


class Oracle:
    """One chain on the CPU oracle.  Mirrors bayesnmf_amd.engine.Engine's method names."""

    def __init__(self, M, N, likelihood="poisson", prior="gamma", MH=False, learning_rank=False,
                 rank_method="SBFI", seed=1, chain_id=0, temperature=None, save_Z=False, nthreads=1):
        M = np.asfortranarray(M, dtype=np.int32)
        self.K, self.G = M.shape
        self.N = int(N)
        self._temp = None if temperature is None else np.ascontiguousarray(temperature, dtype=np.float64)
        cfg = OrcConfig(self.K, self.G, self.N, LIKELIHOOD[likelihood], PRIOR[prior], int(MH),
                        int(learning_rank), RANK_METHOD[rank_method], int(save_Z), int(nthreads),
                        int(seed), int(chain_id), 0,
                        _dp(self._temp) if self._temp is not None else None,
                        0 if self._temp is None else self._temp.size)
        self._h = lib().orc_create(C.byref(cfg), M.ctypes.data_as(C.POINTER(C.c_int32)))
        self.M = M

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, name):
        K, G, N = self.K, self.G, self.N
        i = IDS[name]
        if name in ("A",):
            return (1, N)
        if name == "R":
            return (1,)
        if name == "Z":
            return (K, N, G)
        if name in ("sigmasq", "Alpha", "Beta"):
            return (G,)
        if name == "Mhat":
            return (K, G)
        if name in ("P", "ZsumG", "P_acceptance_rate") or name.endswith("_p"):
            return (K, N)
        return (N, G)

    def set(self, name, value):
        v = np.asarray(value, dtype=np.float64)
        flat = np.ascontiguousarray(v.ravel(order="F"))
        rc = lib().orc_set_array(self._h, IDS[name], _dp(flat), flat.size)
        if rc != 0:
            raise ValueError(f"orc_set_array({name}) failed rc={rc}")

    def get(self, name):
        shp = self._shape(name)
        n = int(np.prod(shp))
        out = np.empty(n)
        rc = lib().orc_get_array(self._h, IDS[name], _dp(out), n)
        if rc != 0:
            raise ValueError(f"orc_get_array({name}) failed rc={rc}")
        return out.reshape(shp, order="F")

    def init(self):
        row = np.empty(NMETRIC)
        rc = lib().orc_init(self._h, _dp(row))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error(self._h).decode())
        return row

    def run(self, n_iter, converged=False):
        out = np.empty((n_iter, NMETRIC))
        rc = lib().orc_run(self._h, n_iter, int(converged), _dp(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error(self._h).decode())
        return out

    STEPS = dict(hyper=0, P=1, E=2, R=3, A=4, Z=5, sigmasq=6)

    def step(self, what, t, converged=False):
        """One conditional of the sweep on the current state, with the random streams of iteration t."""
        rc = lib().orc_t_step(self._h, self.STEPS[what], int(t), int(converged))
        if rc != 0:
            raise ValueError(f"orc_t_step({what}) not defined for this model")

    def set_fresh_mhat(self, on=True):
        """Recompute P diag(A) E from scratch at every factor (what the R code does) instead of maintaining it (the stream spec)."""
        lib().orc_set_fresh_mhat(self._h, int(bool(on)))

    def set_M(self, M):
        M = np.asfortranarray(M, dtype=np.int32)
        assert M.shape == (self.K, self.G)
        lib().orc_set_M(self._h, M.ctypes.data_as(C.POINTER(C.c_int32)))
        self.M = M

    @property
    def iter(self):
        return lib().orc_get_iter(self._h)
