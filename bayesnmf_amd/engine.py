"""ctypes binding of the C ABI in include/bnmf.h (libbnmf.so, HIP/gfx950 only).

This is the Python equivalent of the R `.Call` shim in r/bnmf_shim.c: logic-free marshalling.
There is no CPU fallback: if the library is missing or no GPU is visible the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbnmf.so")
_LIB = None

NMETRIC = 11
NKERNEL = 8
METRIC_NAMES = ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC",
                "rank", "temp", "P_mean_acceptance_rate", "E_mean_acceptance_rate"]
IDS = dict(P=0, E=1, A=2, R=3, Z=4, ZsumK=5, ZsumG=6, sigmasq=7,
           Alpha_p=10, Beta_p=11, Alpha_e=12, Beta_e=13, Mu_p=14, Sigmasq_p=15, Mu_e=16,
           Sigmasq_e=17, Lambda_p=18, Lambda_e=19, Alpha=20, Beta=21,
           A_p=30, B_p=31, C_p=32, D_p=33, M_p=34, S_p=35,
           A_e=40, B_e=41, C_e=42, D_e=43, M_e=44, S_e=45,
           P_acceptance_rate=50, E_acceptance_rate=51, Mhat=60)
LIKELIHOOD = dict(poisson=0, normal=1)
PRIOR = dict(truncnormal=0, exponential=1, gamma=2)
RANK_METHOD = dict(SBFI=0, BFI=1)
MATH_FN = dict(log=0, exp=1, lgamma=2, digamma=3, qnorm=4, log_pnorm=5, sqrt=6, recip=7)
SAMPLER = dict(rgamma=0, rtnorm0=1, rnorm=2, ralpha=3, runif=4, rexp=5, ralpha_fast=6)


class BnmfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbnmf error {code}: {msg}")
        self.code = code


class BnmfConfig(C.Structure):
    _fields_ = [("K", C.c_int32), ("G", C.c_int32), ("N", C.c_int32),
                ("likelihood", C.c_int32), ("prior", C.c_int32), ("MH", C.c_int32),
                ("learning_rank", C.c_int32), ("rank_method", C.c_int32),
                ("save_Z", C.c_int32), ("window", C.c_int32),
                ("seed", C.c_uint64), ("chain_id", C.c_uint32), ("device", C.c_int32),
                ("temperature", C.POINTER(C.c_double)), ("n_temperature", C.c_int64)]


# every symbol include/bnmf.h declares (checked by tests/test_abi.py)
class BnmfMapInfo(C.Structure):
    _fields_ = [("n_used", C.c_int32), ("n_patterns", C.c_int32), ("top_counts", C.c_int32 * 5), ("_pad", C.c_int32),
                ("rmse", C.c_double), ("kl", C.c_double)]


class BnmfConvergenceControl(C.Structure):
    _fields_ = [("MAP_over", C.c_int32), ("MAP_every", C.c_int32), ("Ninarow_nochange", C.c_int32), ("Ninarow_nobest", C.c_int32),
                ("miniters", C.c_int32), ("maxiters", C.c_int32), ("metric", C.c_int32), ("_pad", C.c_int32), ("tol", C.c_double)]


class BnmfConvergenceState(C.Structure):
    _fields_ = [("converged", C.c_int32), ("why", C.c_int32), ("best_iter", C.c_int32), ("inarow_na", C.c_int32),
                ("inarow_no_change", C.c_int32), ("inarow_no_best", C.c_int32), ("have_prev", C.c_int32), ("n_checks", C.c_int32),
                ("prev_MAP_metric", C.c_double), ("best_MAP_metric", C.c_double), ("prev_percent_change", C.c_double)]


NMAPROW = 17
CC_METRICS = ["loglikelihood", "logposterior", "RMSE", "KL", "BIC"]
WHY = {0: None, 1: "no change", 2: "no best", 3: "max iters"}

ABI_SYMBOLS = ["bnmf_create", "bnmf_destroy", "bnmf_set_array", "bnmf_get_array", "bnmf_get_array_i32",
               "bnmf_init", "bnmf_run", "bnmf_window", "bnmf_map", "bnmf_run_until", "bnmf_run_post_warmup", "bnmf_assign", "bnmf_get_iter", "bnmf_profile",
               "bnmf_kernel_name", "bnmf_ubench", "bnmf_test_math", "bnmf_test_sampler", "bnmf_test_philox", "bnmf_test_philox7",
               "bnmf_device_info", "bnmf_device_count", "bnmf_last_error", "bnmf_version", "bnmf_probe_overlap", "bnmf_trim", "bnmf_get_stat"]


def lib():
    """Load libbnmf.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise BnmfError(-100, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        L.bnmf_create.argtypes = [C.POINTER(BnmfConfig), ip, C.POINTER(C.c_void_p)]
        L.bnmf_destroy.argtypes = [C.c_void_p]
        L.bnmf_set_array.argtypes = [C.c_void_p, C.c_int, dp, C.c_size_t]
        L.bnmf_get_array.argtypes = [C.c_void_p, C.c_int, dp, C.c_size_t]
        L.bnmf_get_array_i32.argtypes = [C.c_void_p, C.c_int, ip, C.c_size_t]
        L.bnmf_init.argtypes = [C.c_void_p, dp]
        L.bnmf_run.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.bnmf_window.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.bnmf_map.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp, ip, C.POINTER(BnmfMapInfo)]
        L.bnmf_run_until.argtypes = [C.c_void_p, C.POINTER(BnmfConvergenceControl), C.POINTER(BnmfConvergenceState), dp, C.c_int,
                                     C.POINTER(C.c_int), dp, C.c_int, C.POINTER(C.c_int)]
        L.bnmf_run_post_warmup.argtypes = [C.c_void_p, C.POINTER(BnmfConvergenceControl), C.POINTER(BnmfConvergenceState), C.c_int, dp, C.c_int,
                                           C.POINTER(C.c_int), dp, C.c_int, C.POINTER(C.c_int)]
        L.bnmf_assign.argtypes = [C.c_void_p, C.c_int, ip, dp, C.c_int, ip, dp, C.c_double, dp, ip, dp, dp, dp]
        L.bnmf_get_iter.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.bnmf_profile.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.bnmf_kernel_name.restype = C.c_char_p
        L.bnmf_kernel_name.argtypes = [C.c_int]
        L.bnmf_ubench.argtypes = [C.c_int, dp, dp]
        L.bnmf_test_math.argtypes = [C.c_int, C.c_int, dp, dp, C.c_size_t]
        L.bnmf_test_sampler.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, dp, dp, dp, dp, C.c_size_t]
        L.bnmf_test_philox.argtypes = [C.c_int, up, up, up]
        L.bnmf_test_philox7.argtypes = [C.c_int, up, up, up]
        L.bnmf_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.bnmf_device_count.restype = C.c_int
        L.bnmf_probe_overlap.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.bnmf_trim.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        L.bnmf_get_stat.argtypes = [C.c_void_p, C.c_int, dp]
        L.bnmf_last_error.restype = C.c_char_p
        L.bnmf_version.restype = C.c_int
        _LIB = L
    return _LIB


_RUN = None


def _run_fn():
    """bnmf_run with plain-integer argument types (handle and buffer as addresses): no per-call conversion objects"""
    global _RUN
    if _RUN is None:
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
        _RUN = proto(("bnmf_run", lib()))
    return _RUN


def _chk(rc):
    if rc != 0:
        raise BnmfError(rc, lib().bnmf_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count():
    return lib().bnmf_device_count()


def ubench(device=0):
    """Measured ceilings: (Philox4x32-10 words/s with nothing else in the loop, device-to-device copy GB/s)."""
    a, b = C.c_double(), C.c_double()
    _chk(lib().bnmf_ubench(device, C.byref(a), C.byref(b)))
    return a.value, b.value


def trim(device=0):
    """Release what destroyed handles left cached on the device (rings, streams); returns the bytes of device memory given back."""
    b = C.c_size_t(0)
    _chk(lib().bnmf_trim(device, C.byref(b)))
    return b.value


def device_info(device=0):
    buf = C.create_string_buffer(512)
    _chk(lib().bnmf_device_info(device, buf, 512))
    return buf.value.decode()


def test_math(fn, x, device=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _chk(lib().bnmf_test_math(device, MATH_FN[fn], _dp(x), _dp(out), x.size))
    return out


def test_sampler(which, a=None, b=None, c=None, n=None, seed=1, chain=0, var=2, elem0=0, it=1, device=0):
    arrs = []
    for v in (a, b, c):
        if v is None:
            arrs.append(None)
        else:
            arrs.append(np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64),
                                                             (n,) if n else np.shape(a)), dtype=np.float64))
    if n is None:
        n = arrs[0].size
    out = np.empty(n)
    _chk(lib().bnmf_test_sampler(device, SAMPLER[which], seed, chain, var, elem0, it,
                                 *[_dp(v) if v is not None else None for v in arrs], _dp(out), n))
    return out


def test_philox(ctr, key, device=0, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    _chk((lib().bnmf_test_philox7 if rounds == 7 else lib().bnmf_test_philox)(device, c, k, o))
    return [int(v) for v in o]


class Engine:
    """One chain on one MI355X.  Thin wrapper: every method is one C-ABI call."""

    def __init__(self, M, N, likelihood="poisson", prior="gamma", MH=False, learning_rank=False,
                 rank_method="SBFI", seed=1, chain_id=0, temperature=None, save_Z=False,
                 window=0, device=0):
        M = np.asfortranarray(M, dtype=np.int32)
        self.K, self.G = M.shape
        self.N = int(N)
        self._temp = None if temperature is None else np.ascontiguousarray(temperature, dtype=np.float64)
        cfg = BnmfConfig(self.K, self.G, self.N, LIKELIHOOD[likelihood], PRIOR[prior], int(MH),
                         int(learning_rank), RANK_METHOD[rank_method], int(save_Z), int(window),
                         int(seed), int(chain_id), int(device),
                         _dp(self._temp) if self._temp is not None else None,
                         0 if self._temp is None else self._temp.size)
        h = C.c_void_p()
        _chk(lib().bnmf_create(C.byref(cfg), M.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h)))
        self._h = h
        self._hv = h.value
        self._run = _run_fn()
        self.M = M

    def close(self):
        if getattr(self, "_h", None):
            lib().bnmf_destroy(self._h)
            self._h = None
            self._hv = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, name):
        K, G, N = self.K, self.G, self.N
        if name == "A":
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
        _chk(lib().bnmf_set_array(self._h, IDS[name], _dp(flat), flat.size))

    def get(self, name):
        shp = self._shape(name)
        n = int(np.prod(shp))
        if name in ("Z", "ZsumK", "ZsumG"):
            out = np.empty(n, dtype=np.int32)
            _chk(lib().bnmf_get_array_i32(self._h, IDS[name], out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        else:
            out = np.empty(n)
            _chk(lib().bnmf_get_array(self._h, IDS[name], _dp(out), n))
        return out.reshape(shp, order="F")

    def init(self):
        row = np.empty(NMETRIC)
        _chk(lib().bnmf_init(self._h, _dp(row)))
        return row

    def run(self, n_iter, converged=False, metrics=True):
        out = np.empty((n_iter, NMETRIC)) if metrics else None
        # (the hot call of every driver loop: the bound function and the handle's integer value are looked up once, the buffer goes in as
        # its address — the generic path cost 16 us per call, 1 % of a 20-iteration call at the metric configuration)
        rc = self._run(self._hv, n_iter, 1 if converged else 0, out.ctypes.data if metrics else None)
        if rc != 0:
            _chk(rc)
        return out

    def profile(self, n_iter, converged=False):
        out = np.zeros(NKERNEL)
        _chk(lib().bnmf_profile(self._h, n_iter, int(converged), _dp(out)))
        return {lib().bnmf_kernel_name(i).decode(): out[i] for i in range(NKERNEL)}

    def window(self, name, last_n):
        shp = self._shape(name)
        out = np.empty((last_n, int(np.prod(shp))))
        _chk(lib().bnmf_window(self._h, IDS[name], last_n, _dp(out)))
        return [out[i].reshape(shp, order="F") for i in range(last_n)]

    def run_until(self, cc, state=None):
        """Warm-up to convergence in one C-ABI call.  cc: new_convergence_control() dict; state: BnmfConvergenceState to
        continue from (None = fresh).  Returns (metrics rows, MAP rows [n_checks x NMAPROW], state)."""
        c = BnmfConvergenceControl(cc["MAP_over"], cc["MAP_every"], cc["Ninarow_nochange"], cc["Ninarow_nobest"], cc["miniters"],
                                   cc["maxiters"], CC_METRICS.index(cc["metric"]), 0, cc["tol"])
        st = state or BnmfConvergenceState()
        cap_rows = max(cc["maxiters"] - self.iter, 0) + 1
        cap_checks = cap_rows // cc["MAP_every"] + 2
        rows, maps = np.empty((cap_rows, NMETRIC)), np.empty((cap_checks, NMAPROW))
        nr, nc = C.c_int(), C.c_int()
        _chk(lib().bnmf_run_until(self._h, C.byref(c), C.byref(st), _dp(rows), cap_rows, C.byref(nr), _dp(maps), cap_checks, C.byref(nc)))
        return rows[:nr.value].copy(), maps[:nc.value].copy(), st

    def run_post_warmup(self, cc, state, post_warmup):
        """The MH models' post-warm-up tail in one C-ABI call.  Returns (metrics rows, MAP rows, state)."""
        c = BnmfConvergenceControl(cc["MAP_over"], cc["MAP_every"], cc["Ninarow_nochange"], cc["Ninarow_nobest"], cc["miniters"],
                                   cc["maxiters"], CC_METRICS.index(cc["metric"]), 0, cc["tol"])
        cap_rows = int(post_warmup) + 1
        cap_checks = cap_rows // cc["MAP_every"] + 3
        rows, maps = np.empty((cap_rows, NMETRIC)), np.empty((cap_checks, NMAPROW))
        nr, nc = C.c_int(), C.c_int()
        _chk(lib().bnmf_run_post_warmup(self._h, C.byref(c), C.byref(state), int(post_warmup), _dp(rows), cap_rows, C.byref(nr), _dp(maps),
                                        cap_checks, C.byref(nc)))
        return rows[:nr.value].copy(), maps[:nc.value].copy(), state

    def assign(self, last_n, reference_P, used=None, keep=None, MAP_P=None, credible_interval=0.95):
        """assign_signatures_ensemble_ over recorded samples: votes (N x R), assigned reference per signature (-1 = not
        kept), cosine of the MAP estimate and credible bounds of the per-sample cosines."""
        N = self.N
        ref = np.asfortranarray(reference_P, dtype=np.float64)
        R = ref.shape[1]
        ip = C.POINTER(C.c_int32)
        u = None if used is None else np.ascontiguousarray(used, dtype=np.int32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.int32)
        mp = None if MAP_P is None else np.asfortranarray(MAP_P, dtype=np.float64)
        votes, asg = np.zeros(N * R), np.empty(N, dtype=np.int32)
        mc, lo, hi = np.empty(N), np.empty(N), np.empty(N)
        _chk(lib().bnmf_assign(self._h, last_n, None if u is None else u.ctypes.data_as(ip), _dp(ref.ravel(order="F")), R,
                               None if kp is None else kp.ctypes.data_as(ip), None if mp is None else _dp(mp.ravel(order="F")),
                               float(credible_interval), _dp(votes), asg.ctypes.data_as(ip), _dp(mc), _dp(lo), _dp(hi)))
        return dict(votes=votes.reshape((N, R), order="F"), assigned=asg, MAP_cosine=mc, lower_cosine=lo, upper_cosine=hi)

    def map(self, last_n, credible_interval=0.95):
        """get_MAP_ over the last `last_n` recorded samples, on the device (one C-ABI call)."""
        K, G, N = self.K, self.G, self.N
        Pm, Em, Am, top = np.empty(K * N), np.empty(N * G), np.empty(N), np.empty(5 * N)
        ci = credible_interval is not None and credible_interval > 0
        Pl, Pu, El, Eu = (np.empty(K * N), np.empty(K * N), np.empty(N * G), np.empty(N * G)) if ci else (None,) * 4
        used = np.empty(last_n, dtype=np.int32)
        info = BnmfMapInfo()
        _chk(lib().bnmf_map(self._h, last_n, float(credible_interval) if ci else 0.0, _dp(Pm), _dp(Em), _dp(Am), _dp(top),
                            *[_dp(a) if a is not None else None for a in (Pl, Pu, El, Eu)],
                            used.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(info)))
        f = lambda a, shp: None if a is None else a.reshape(shp, order="F")   # noqa: E731
        npat = min(info.n_patterns, 5)
        return dict(P=f(Pm, (K, N)), E=f(Em, (N, G)), A=Am.reshape(1, N), used=used.astype(bool),
                    P_lower=f(Pl, (K, N)), P_upper=f(Pu, (K, N)), E_lower=f(El, (N, G)), E_upper=f(Eu, (N, G)),
                    top_A=top.reshape(5, N)[:npat], top_counts=[int(c) for c in info.top_counts][:npat],
                    n_used=info.n_used, n_patterns=info.n_patterns, rmse=info.rmse, kl=info.kl)

    def stat(self, what):
        """Sizes of the handle's per-iteration buffers (bnmf_get_stat)."""
        v = C.c_double()
        _chk(lib().bnmf_get_stat(self._h, int(what), C.byref(v)))
        return v.value

    @property
    def iter(self):
        it = C.c_int()
        _chk(lib().bnmf_get_iter(self._h, C.byref(it)))
        return it.value
