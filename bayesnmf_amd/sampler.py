"""Host-side mirror of the reference's R interface for the Gibbs path: `bayesNMF()`
(R/bayesNMF.R:24-138) and the R6 class `bayesNMF_sampler` (R/bayesNMF_sampler.R:8-747) — same names,
argument meaning, defaults, field names and error messages — driving the HIP engine through the C ABI
(one call per block of <= MAP_every iterations instead of R-level work per iteration).

Only the sampling path is mirrored.  Post-processing (reference assignment, plots, summary) is out of
scope.  New optional trailing arguments: seed, chain_id, device, save_Z.
"""
import datetime
import os
import pickle
import shutil
import time
from collections import Counter

import numpy as np
import pandas as pd

from .convergence import check_convergence, new_convergence_control
from .engine import Engine
from .setup import apply_hyperprior_params

_PRIOR_PARAM_NAMES = {"truncnormal": ["Mu_p", "Sigmasq_p", "Mu_e", "Sigmasq_e"],
                      "exponential": ["Lambda_p", "Lambda_e"],
                      "gamma": ["Alpha_p", "Beta_p", "Alpha_e", "Beta_e"]}


def get_temp_sched_(length, n_temp, rng=None):
    """get_temp_sched_ (R/utils.R:307-332): 0, 1e-9..1e-5, 8 x 1e-4, (1..9.9)e-4..e-1, then 1."""
    nX = max(int(round(n_temp / 374)), 1)
    sched = [0.0] * nX
    for x in range(9, 4, -1):
        sched += [10.0 ** (-x)] * nX
    sched += [10.0 ** (-4)] * int(round(8 * nX))
    for y in range(4, 0, -1):
        for x in np.arange(0, 8.9 + 1e-9, 0.1):
            sched += [(1 + x) * 10.0 ** (-y)] * nX
    sched = np.array(sched)
    if len(sched) > n_temp:
        rng = rng or np.random.default_rng(0)
        sched = np.sort(rng.choice(sched, size=n_temp, replace=False))
    return np.concatenate([sched, np.ones(max(length - len(sched), 0))])


def renormalize(P, E):
    """renormalize (R/helpers.R:35-49): columns of P sum to 1, product P E unchanged."""
    cs = P.sum(axis=0)
    return P / cs[None, :], E * cs[:, None]


def get_mode(matrix_list):
    """get_mode (R/helpers.R:63-79): most frequent matrix by its pasted string."""
    keys = ["".join(str(int(v)) if float(v).is_integer() else str(v) for v in np.ravel(m, order="F")) for m in matrix_list]
    # sort(table(.), decreasing = TRUE): table() orders the patterns alphabetically and the sort is stable
    counts = sorted(Counter(keys).items(), key=lambda kv: (-kv[1], kv[0]))
    mode = counts[0][0]
    idx = [i for i, k in enumerate(keys) if k == mode]
    return dict(matrix=matrix_list[idx[0]], top_counts=counts[:5], idx=idx)


class bayesNMF_sampler:
    """Python mirror of the R6 class `bayesNMF_sampler` (public fields of R/bayesNMF_sampler.R:11-67)."""

    def __init__(self, data, rank, likelihood="poisson", prior="truncnormal", rank_method="SBFI",
                 MH=None, convergence_control=None, prop_temp=0.2, post_warmup=None,
                 output_dir=None, overwrite=False, hyperprior_params=None, init_prior_params=None,
                 init_params=None, verbosity=1, periodic_save=True, save_all_samples=False,
                 seed=1, chain_id=0, device=0, save_Z=False, engine_factory=None, intermediate_credible_intervals=False,
                 engine_side_convergence=True):
        if MH is None:
            MH = likelihood == "poisson" and prior in ("truncnormal", "exponential")
        cc = dict(convergence_control) if convergence_control is not None else new_convergence_control()
        if post_warmup is None:
            post_warmup = 2 * cc["MAP_over"]
        if output_dir is None:
            output_dir = f"nmf_{likelihood}_{prior}"
        # output dir handling (R/bayesNMF_sampler.R:110-121)
        final_dir, tail = output_dir, 0
        while not overwrite and os.path.isdir(final_dir):
            tail += 1
            final_dir = f"{output_dir}_{tail}"
        if overwrite and os.path.isdir(final_dir):
            shutil.rmtree(final_dir)
        os.makedirs(final_dir, exist_ok=True)

        rank = np.atleast_1d(np.asarray(rank, dtype=int))
        learning_rank = rank.size > 1
        if learning_rank and rank.min() != 0:
            rank = np.arange(0, rank.max() + 1)
        display_rank = f"{rank.min()}:{rank.max()}" if learning_rank else int(rank[0])
        data = np.asarray(data)
        n_iters = cc["maxiters"] + (post_warmup if MH else 0)
        if learning_rank:
            self.temperature_schedule = get_temp_sched_(n_iters, int(round(prop_temp * cc["maxiters"])),
                                                        rng=np.random.default_rng(seed))
        else:
            self.temperature_schedule = np.ones(n_iters)
        self.data = data
        self.dims = dict(K=data.shape[0], N=int(rank.max()), G=data.shape[1])
        self.specs = dict(rank=rank, likelihood=likelihood, prior=prior, MH=bool(MH), learning_rank=learning_rank,
                          convergence_control=cc, output_dir=final_dir, overwrite=overwrite, verbosity=verbosity,
                          periodic_save=periodic_save, save_all_samples=save_all_samples, save_Z=bool(save_Z),
                          intermediate_credible_intervals=intermediate_credible_intervals,
                          engine_side_convergence=engine_side_convergence)
        if learning_rank:
            self.specs["prop_temp"] = prop_temp
            self.specs["rank_method"] = rank_method
        if MH:
            self.specs["post_warmup"] = post_warmup
        self.state = dict(iter=1, indent=0, converged=False)
        self.MAP = dict(assignment_res=None)
        self.credible_intervals = {}
        self.reference_comparison = dict(reference_P=None, assignments=None, keep_sigs=None, idxs=None, votes=None,
                                         summary=None, plots={}, label_switching_df=None)
        self.time = {}
        self._block_hook = None          # multi-chain launcher: called after every block (multichain.ChainSync.block)
        self.acceptance_rates = dict(P_acceptance_rate=None, E_acceptance_rate=None)
        self.log_con = open(os.path.join(final_dir, "log.txt"), "w")
        self.log("Initialized sampler", verbosity=1)
        self.state["indent"] = 1
        self.log(f"likelihood = {likelihood}, prior = {prior}, MH = {str(bool(MH)).upper()}", verbosity=1)
        self.log(f"learning_rank = {str(learning_rank).upper()}, rank = {display_rank}", verbosity=1)
        self.log(f"maxiters = {cc['maxiters']}", verbosity=1)
        self.log(f"MAP_over = {cc['MAP_over']}", verbosity=1)
        self.log(f"MAP_every = {cc['MAP_every']}", verbosity=1)
        cols = ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank"]
        extra = ["P_mean_acceptance_rate", "E_mean_acceptance_rate"] if MH else []
        self.state["MAP_metrics"] = pd.DataFrame(columns=cols + ["MAP_A_counts", "mean_temp"] + extra, dtype=float)
        self.state["sample_metrics"] = pd.DataFrame(columns=cols + ["temp"] + extra, dtype=float)
        self.state["MAP_idx"] = np.arange(1, cc["MAP_over"] + 1)
        self.state["indent"] = 0
        self.log("Setup", verbosity=1)
        self.state["indent"] = 1
        self.log("Setting hyperprior parameters", verbosity=1)
        self._check_model()
        self.log("Model check passed", verbosity=1)

        window = n_iters if save_all_samples else cc["MAP_over"]
        factory = engine_factory or Engine
        kw = dict(likelihood=likelihood, prior=prior, MH=bool(MH), learning_rank=learning_rank,
                  rank_method=rank_method if rank_method in ("SBFI", "BFI") else "SBFI",
                  seed=seed, chain_id=chain_id, temperature=self.temperature_schedule, save_Z=save_Z)
        if engine_factory is None:
            kw.update(window=window, device=device)
        self._chain = factory(np.asfortranarray(data, dtype=np.int32), self.dims["N"], **kw)
        self.hyperprior_params = apply_hyperprior_params(self._chain, prior, data, self.dims["N"], hyperprior_params)
        self.log("Initializing prior parameters and parameters", verbosity=1)
        for name, val in (init_prior_params or {}).items():
            self._chain.set(name, val)
        for name, val in (init_params or {}).items():
            self._chain.set(name, val)
        self.log("Sampling parameters from priors", verbosity=1)
        row = self._chain.init()
        self.log("Logging initial sample", verbosity=1)
        self.log("Logging initial sample metrics", verbosity=1)
        self._append_metrics(np.asarray(row)[None, :])
        self._sync_state()
        self.state["indent"] = 0

    # ---------------------------------------------------------------- helpers
    def _check_model(self):
        """check_model (R/bayesNMF_sampler.R:623-645): same messages."""
        lk, pr, MH = self.specs["likelihood"], self.specs["prior"], self.specs["MH"]
        if lk not in ("normal", "poisson"):
            self._error("likelihood must be one of normal, poisson")
        if lk == "normal":
            if pr not in ("truncnormal", "exponential"):
                self._error("prior must be one of c('truncnormal','exponential') with `likelihood = 'normal'`")
        else:
            if pr not in ("gamma", "exponential", "truncnormal"):
                self._error("prior must be one of c('gamma','exponential','truncnormal') with `likelihood = 'poisson'`")
            if pr == "gamma" and MH:
                self._error("gamma prior cannot be used in a MH-within-gibbs sampler")
            if pr == "truncnormal" and not MH:
                self._error("truncnormal prior can only be used in a MH-within-gibbs sampler")

    def _error(self, msg):
        msg = "ERROR: " + msg
        if self.log_con is not None:
            ts = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
            self.log_con.write(f"[{ts}] {msg}\n")
            self.log_con.flush()
        raise ValueError(msg)

    def log(self, msg, verbosity=5):
        """log (R/bayesNMF_sampler.R:423-455): `[timestamp] <tabs><msg>`."""
        if verbosity > self.specs["verbosity"] or msg is None or self.log_con is None:
            return
        lines = str(msg).split("\n")
        ts = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        indent = "\t" * self.state["indent"]
        out = [indent + lines[0]] + [indent + " " * (len(ts) + 1) + ln for ln in lines[1:]]
        out = [ln for ln in out if ln.strip() != ""]
        self.log_con.write(f"[{ts}] " + "\n".join(out) + "\n")
        self.log_con.flush()

    def _append_metrics(self, rows):
        cols = list(self.state["sample_metrics"].columns)
        names = ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank", "temp",
                 "P_mean_acceptance_rate", "E_mean_acceptance_rate"]
        df = pd.DataFrame(rows, columns=names)[cols]
        self.state["sample_metrics"] = df if self.state["sample_metrics"].empty else \
            pd.concat([self.state["sample_metrics"], df], ignore_index=True)

    def _sync_state(self):
        """Materialise the device state into the R6-style fields (params, prior_params)."""
        names = ["P", "E", "A", "R"] + (["sigmasq"] if self.specs["likelihood"] == "normal" else [])
        self.params = {n: self._chain.get(n) for n in names}
        self.prior_params = {n: self._chain.get(n) for n in _PRIOR_PARAM_NAMES[self.specs["prior"]]}
        if self.specs["MH"]:
            self.acceptance_rates = {n: self._chain.get(n) for n in ("P_acceptance_rate", "E_acceptance_rate")}

    @property
    def samples(self):
        """samples[[name]][[i]] of the last min(iter, window) iterations (record_sample, :651-672)."""
        cc = self.specs["convergence_control"]
        n = min(self.state["iter"], len(self.temperature_schedule) if self.specs["save_all_samples"] else cc["MAP_over"])
        names = ["P", "E", "A", "R"] + _PRIOR_PARAM_NAMES[self.specs["prior"]]
        if self.specs["MH"]:
            names += ["P_acceptance_rate", "E_acceptance_rate"]
        if self.specs["likelihood"] == "normal":
            names += ["sigmasq"]
        out = {nm: self._chain.window(nm, n) for nm in names}
        if self.specs.get("save_Z"):
            try:
                out["Z"] = self._chain.window("Z", n)
            except Exception:  # noqa: BLE001  (Z history not kept: too large for the device ring)
                pass
        return out

    # ---------------------------------------------------------------- public utilities (R/utils.R)
    def get_Mhat(self, P=None, A=None, E=None):
        P = self.params["P"] if P is None else P
        A = self.params["A"] if A is None else A
        E = self.params["E"] if E is None else E
        return (P * np.ravel(A)[None, :]) @ E

    def get_loglik(self, P=None, A=None, E=None):
        from scipy.stats import poisson
        Mhat = np.maximum(self.get_Mhat(P, A, E), 1e-6)
        return float(poisson.logpmf(self.data, Mhat).sum())

    def get_MAP(self, final=False, credible_interval=0.95):
        """get_MAP_ (R/utils.R:194-288) over the window state$MAP_idx of recorded samples."""
        cc = self.specs["convergence_control"]
        n = min(cc["MAP_over"], self.state["iter"])
        first_iter = self.state["iter"] - n + 1
        if hasattr(self._chain, "map"):
            # on-device window statistics: one C-ABI call returns K*N + N*G means (+ credible bounds), no window copy
            try:
                r = self._chain.map(n, credible_interval)
            except Exception as ex:     # e.g. a credible interval that needs more order statistics than the device keeps
                if "bnmf_window" not in str(ex):
                    raise
                r = None
            if r is not None:
                keep = np.where(np.ravel(r["A"]) == 1)[0] if final else np.arange(self.dims["N"])
                counts = [("".join(str(int(v)) for v in row), c) for row, c in zip(r["top_A"], r["top_counts"])]
                self.MAP = dict(P=r["P"][:, keep], A=r["A"][:, keep], E=r["E"][keep, :],
                                idx=[first_iter + i for i in np.where(r["used"])[0]], A_counts=counts, keep_sigs=keep,
                                RMSE=r["rmse"], KL=r["kl"])
                if r["P_lower"] is not None:
                    self.credible_intervals = dict(P=dict(lower=r["P_lower"][:, keep], upper=r["P_upper"][:, keep]),
                                                   E=dict(lower=r["E_lower"][keep, :], upper=r["E_upper"][keep, :]))
                return self.MAP
        A_list = self._chain.window("A", n)
        mode = get_mode(A_list)
        idx = mode["idx"]
        keep = np.where(np.ravel(mode["matrix"]) == 1)[0] if final else np.arange(self.dims["N"])
        Ps, Es = self._chain.window("P", n), self._chain.window("E", n)
        rs = [renormalize(Ps[i], Es[i]) for i in idx]
        Pm = np.mean([r[0][:, keep] for r in rs], axis=0)
        Em = np.mean([r[1][keep, :] for r in rs], axis=0)
        self.MAP = dict(P=Pm, A=np.asarray(mode["matrix"]).reshape(1, -1)[:, keep], E=Em,
                        idx=[first_iter + i for i in idx], A_counts=mode["top_counts"], keep_sigs=keep)
        if not credible_interval:
            return self.MAP
        probs = [0.5 - credible_interval / 2, 0.5 + credible_interval / 2]
        Parr = np.stack([r[0][:, keep] for r in rs], axis=2)
        Earr = np.stack([r[1][keep, :] for r in rs], axis=2)
        self.credible_intervals = dict(
            P=dict(lower=np.quantile(Parr, probs[0], axis=2), upper=np.quantile(Parr, probs[1], axis=2)),
            E=dict(lower=np.quantile(Earr, probs[0], axis=2), upper=np.quantile(Earr, probs[1], axis=2)))
        return self.MAP

    def _update_MAP_metrics(self, final=False):
        """update_MAP_metrics_ (R/utils.R:356-397)."""
        G, K = self.dims["G"], self.dims["K"]
        P, A, E = self.MAP["P"], self.MAP["A"], self.MAP["E"]
        if final:
            A = np.ones((1, P.shape[1]))
        sm = self.state["sample_metrics"]
        cc_ = self.specs["convergence_control"]
        n_win = min(cc_["MAP_over"], self.state["iter"])
        window = np.arange(self.state["iter"] - n_win + 1, self.state["iter"] + 1)   # state$MAP_idx: the whole window, not MAP$idx
        win = sm[sm["iter"].isin(window)]
        ll, lpost = float(win["loglikelihood"].mean()), float(win["logposterior"].mean())
        n_params = float(np.sum(A) * (G + K))
        if "RMSE" in self.MAP:                      # computed on the device with the window statistics
            rmse, kl = float(self.MAP["RMSE"]), float(self.MAP["KL"])
        else:
            Mhat = (P * np.ravel(A)[None, :]) @ E
            Mt, Mh = np.maximum(self.data, 1e-6), np.maximum(Mhat, 1e-6)
            rmse, kl = float(np.sqrt(np.mean((Mhat - self.data) ** 2))), float(np.sum(Mt * np.log(Mt / Mh)))
        row = dict(iter=self.state["iter"], RMSE=rmse, KL=kl, loglikelihood=ll, logposterior=lpost, n_params=n_params,
                   BIC=-2 * ll + n_params * np.log(G), rank=float(np.sum(self.MAP["A"])),
                   MAP_A_counts=float(self.MAP["A_counts"][0][1]),
                   mean_temp=float(np.mean(self.temperature_schedule[window - 1])))
        if self.specs["MH"]:                        # compute_metrics_ uses the CURRENT acceptance matrices (R/utils.R:444-452)
            row["P_mean_acceptance_rate"] = float(sm["P_mean_acceptance_rate"].iloc[-1])
            row["E_mean_acceptance_rate"] = float(sm["E_mean_acceptance_rate"].iloc[-1])
        df = pd.DataFrame([row])
        self.state["MAP_metrics"] = df if self.state["MAP_metrics"].empty else \
            pd.concat([self.state["MAP_metrics"], df], ignore_index=True)
        return row

    def _check(self, final=False):
        self.log(f"iter = {self.state['iter']}", verbosity=1)
        self.state["indent"] = 2
        self.log("Computing MAP", verbosity=1)
        # The reference recomputes the credible intervals at EVERY check (credible_intervals starts as list(), not NULL:
        # R/bayesNMF_sampler.R:49, R/utils.R:269) although nothing reads them before the final MAP.  Here they are
        # materialised at the final MAP and on every user call of get_MAP(); intermediate_credible_intervals=True
        # restores the reference's every-check behaviour.
        ci = 0.95 if (final or self.specs.get("intermediate_credible_intervals")) else None
        self.get_MAP(final=final, credible_interval=ci)
        if self.specs["learning_rank"]:
            self.log("\n".join(f"{k}  {v}" for k, v in self.MAP["A_counts"]), verbosity=1)
        self.log("Checking convergence", verbosity=1)
        row = self._update_MAP_metrics(final=final)
        cc = self.specs["convergence_control"]
        msg = check_convergence(self.state, cc, row[cc["metric"]], self.temperature_schedule)
        self.log(msg, verbosity=1)
        self.state["indent"] = 1
        return msg

    # ---------------------------------------------------------------- the sampling loop
    def run_gibbs_sampler(self):
        """run_gibbs_sampler (R/bayesNMF_sampler.R:265-408) with one engine call per block."""
        cc = self.specs["convergence_control"]
        self.log("Starting Gibbs sampler", verbosity=1)
        start = time.time()
        self.state["indent"] = 1
        if (hasattr(self._chain, "run_until") and self._block_hook is None and not self.specs["periodic_save"]
                and not self.specs["save_all_samples"] and self.specs.get("engine_side_convergence", True)):
            self._run_until_on_engine(cc)
        while not self.state["converged"] and self.state["iter"] < cc["maxiters"]:
            it = self.state["iter"]
            nxt = (it // cc["MAP_every"] + 1) * cc["MAP_every"]
            n = min(nxt, cc["maxiters"]) - it
            rows = self._chain.run(n, converged=False)
            self._append_metrics(rows)
            self.state["iter"] = it + n
            it = self.state["iter"]
            if (it % cc["MAP_every"] == 0 and it >= max(cc["MAP_over"], cc["MAP_every"])) or it >= cc["maxiters"]:
                if self.specs["save_all_samples"]:
                    self.state["MAP_idx"] = np.arange(it - cc["MAP_over"] + 1, it + 1)
                self._check()
                if self.state["converged"]:
                    self.state["converged_iter"] = it
                    self.log(f"Converged at {it} due to {self.state['why']}", verbosity=1)
                if self.specs["periodic_save"]:
                    self.log("Saving object", verbosity=1)
                    self.save_object()
            if self._block_hook is not None:
                self._block_hook(self, rows)
        if self.specs["MH"]:
            start_MH = time.time()
            self.time["warmup"] = (start_MH - start) / 60.0
            pw = self.specs["post_warmup"]
            self.log(f"Warmup done, sampling {pw} with MH for inference", verbosity=1)
            done = 0
            if (hasattr(self._chain, "run_post_warmup") and self._block_hook is None and not self.specs["periodic_save"]
                    and not self.specs["save_all_samples"] and self.specs.get("engine_side_convergence", True) and pw > 0):
                self._post_warmup_on_engine(cc, pw)      # the tail as one engine call (bnmf_run_post_warmup)
                done = pw
            while done < pw:
                it = self.state["iter"]
                nxt = (it // cc["MAP_every"] + 1) * cc["MAP_every"]
                n = min(nxt - it, pw - done)
                rows = self._chain.run(n, converged=True)
                self._append_metrics(rows)
                self.state["iter"] = it + n
                done += n
                if self.state["iter"] % cc["MAP_every"] == 0 or done == pw:
                    self._check(final=(done == pw))
                    if self.specs["periodic_save"]:
                        self.save_object()
                if self._block_hook is not None:
                    self._block_hook(self, rows)
            self.log(f"Additional {pw} MH samples done", verbosity=1)
            self.time["MH"] = (time.time() - start_MH) / 60.0
        else:
            self.get_MAP(final=True)
            self.log("Final MAP computed", verbosity=1)
        self._sync_state()
        self.log("Sampler done", verbosity=1)
        self.time["total"] = (time.time() - start) / 60.0
        self.time["per_iter"] = self.time["total"] / self.state["iter"]
        self.log(f"Total time: {round(self.time['total'], 2)} minutes", verbosity=1)
        self.log("Saving final object", verbosity=1)
        self.save_object()
        return self

    def _cc_state(self):
        """state$prev_MAP_metric ... as the C-ABI's bnmf_convergence_state."""
        from .engine import BnmfConvergenceState, WHY
        st = BnmfConvergenceState()
        have = "prev_MAP_metric" in self.state
        st.have_prev = 1 if have else 0
        if have:
            st.prev_MAP_metric = self.state["prev_MAP_metric"]; st.best_MAP_metric = self.state["best_MAP_metric"]
            st.prev_percent_change = self.state.get("prev_percent_change", float("nan"))
            st.inarow_na = int(self.state.get("inarow_na", 0)); st.inarow_no_change = int(self.state.get("inarow_no_change", 0))
            st.inarow_no_best = int(self.state.get("inarow_no_best", 0)); st.best_iter = int(self.state.get("best_iter", 0) or 0)
        st.converged = 1 if self.state["converged"] else 0
        st.why = {v: k for k, v in WHY.items()}.get(self.state.get("why"), 0)
        return st

    def _absorb_map_rows(self, maps, cc):
        """state$MAP_metrics rows and the log lines of the MAP checks an engine-side loop has made."""
        names = ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank", "MAP_A_counts", "mean_temp",
                 "P_mean_acceptance_rate", "E_mean_acceptance_rate"]
        flip = -1 if cc["metric"] in ("loglikelihood", "logposterior") else 1
        for r in maps:
            row = dict(zip(names, r[:12]))
            if not self.specs["MH"]:
                row.pop("P_mean_acceptance_rate"); row.pop("E_mean_acceptance_rate")
            df = pd.DataFrame([row])
            self.state["MAP_metrics"] = df if self.state["MAP_metrics"].empty else pd.concat([self.state["MAP_metrics"], df], ignore_index=True)
            m = flip * row[cc["metric"]]
            pcs = "NA" if np.isnan(r[12]) else f"{flip * round(r[12] * 100, 2)}"
            self.log(f"iter = {int(r[0])}", verbosity=1)
            self.state["indent"] = 2
            self.log("Computing MAP", verbosity=1)
            self.log("Checking convergence", verbosity=1)
            self.log(f"{cc['metric']} = {round(m, 2)} | {pcs}% change | {int(r[13])} no change | {int(r[14])} no best | {int(r[15])} NA", verbosity=1)
            self.state["indent"] = 1

    def _absorb_cc_state(self, st):
        if st.have_prev:
            self.state.update(prev_MAP_metric=st.prev_MAP_metric, best_MAP_metric=st.best_MAP_metric,
                              prev_percent_change=st.prev_percent_change, inarow_na=st.inarow_na,
                              inarow_no_change=st.inarow_no_change, inarow_no_best=st.inarow_no_best)
            if st.best_iter:
                self.state["best_iter"] = st.best_iter
        from .engine import WHY
        if st.why in WHY and WHY[st.why] is not None:      # check_convergence_ keeps updating `why` during the post-warm-up checks too
            self.state["why"] = WHY[st.why]

    def _post_warmup_on_engine(self, cc, pw):
        """The MH models' post-warm-up iterations (R/bayesNMF_sampler.R:332-384) as one engine call; the final MAP (kept
        signatures, credible intervals) is then taken from the device window as after the reference's last check."""
        # The reference's LAST check is get_MAP(final = TRUE) + check_convergence(final = TRUE): MAP metrics AND convergence bookkeeping on
        # the KEPT signatures (R/bayesNMF_sampler.R:364-375).  The engine's checks are over all N (bnmf_map), which is what every check
        # but the last wants — so the engine runs up to the last regular check before the end, the rest of the iterations are one plain
        # bnmf_run, and the final check is made here, exactly as the reference makes it (rounds 3-4 rebuilt the last
        # MAP-metrics row but kept the all-N bookkeeping).
        me = cc["MAP_every"]
        it0 = int(self.state["iter"])
        end = it0 + int(pw)
        n1 = max(((end - 1) // me) * me - it0, 0)
        if n1 > 0:
            rows, maps, st = self._chain.run_post_warmup(cc, self._cc_state(), n1)
            if len(rows):
                self._append_metrics(rows)
                self.state["iter"] = int(rows[-1, 0])
            self._absorb_map_rows(maps, cc)
            self._absorb_cc_state(st)
        n2 = end - int(self.state["iter"])
        if n2 > 0:
            self._append_metrics(self._chain.run(n2, converged=True))
            self.state["iter"] = end
        self._check(final=True)

    def _run_until_on_engine(self, cc):
        """The warm-up loop (blocks, MAP, MAP metrics, convergence bookkeeping) as one engine call;
        the R6-style state, both metric tables and the log lines are rebuilt from what it returns."""
        from .engine import WHY
        rows, maps, st = self._chain.run_until(cc)
        if len(rows):
            self._append_metrics(rows)
            self.state["iter"] = int(rows[-1, 0])
        self._absorb_map_rows(maps, cc)
        self._absorb_cc_state(st)
        if st.converged:
            self.state["converged"] = True
            self.state["why"] = WHY[st.why]
            self.state["converged_iter"] = self.state["iter"]
            self.log(f"Converged at {self.state['iter']} due to {self.state['why']}", verbosity=1)

    def assign_signatures_ensemble(self, reference_P, reference_names=None, idxs="MAP_idx", credible_interval=0.95):
        """assign_signatures_ensemble_ (R/postprocessing.R:175-341): every posterior sample of the MAP window votes, with
        its cosine similarity as weight, for the Hungarian assignment of its included signatures to the reference
        catalogue `reference_P` (K x R matrix; the reference's default is its bundled COSMIC v3.3.1 SBS table).
        Returns dict(assignments, votes) (data frames with the reference's columns) and stores them in
        self.reference_comparison (fields reference_P, idxs, keep_sigs, assignments, votes)."""
        ref = np.asarray(reference_P, dtype=float)
        if ref.shape[0] != self.dims["K"]:
            raise ValueError(f"Reference matrix has {ref.shape[0]} rows, but data has {self.dims['K']} rows.")
        names = list(reference_names) if reference_names is not None else list(range(1, ref.shape[1] + 1))
        cc = self.specs["convergence_control"]
        n = min(cc["MAP_over"], self.state["iter"])
        first_iter = self.state["iter"] - n + 1
        idx = self.MAP["idx"] if isinstance(idxs, str) else list(idxs)
        used = np.zeros(n, dtype=np.int32)
        used[np.asarray(idx) - first_iter] = 1
        N = self.dims["N"]
        A = np.ravel(self.MAP["A"])
        if len(self.MAP["keep_sigs"]) == N and (A == 0).any():      # get_MAP(final = FALSE): only included signatures
            keep_sigs = np.where(A == 1)[0]
            self.MAP["sig_idx"] = keep_sigs
        else:
            keep_sigs = np.asarray(self.MAP["keep_sigs"])
            self.MAP["sig_idx"] = np.arange(len(keep_sigs))
        keep = np.zeros(N, dtype=np.int32); keep[keep_sigs] = 1
        MAP_full = np.zeros((self.dims["K"], N)); MAP_full[:, keep_sigs] = np.asarray(self.MAP["P"])[:, self.MAP["sig_idx"]]
        r = self._chain.assign(n, ref, used=used, keep=keep, MAP_P=MAP_full, credible_interval=credible_interval)
        rows = []
        for i in keep_sigs:
            tot = r["votes"][i].sum()
            for j in np.argsort(-r["votes"][i], kind="stable"):
                if r["votes"][i, j] > 0:
                    rows.append(dict(sig_est=int(i) + 1, sig_ref=names[j], prop_votes=r["votes"][i, j] / tot))
        votes = pd.DataFrame(rows, columns=["sig_est", "sig_ref", "prop_votes"])
        assignments = pd.DataFrame([dict(sig_est=int(i) + 1, sig_ref=names[r["assigned"][i]], MAP_cosine=r["MAP_cosine"][i],
                                         lower_cosine=r["lower_cosine"][i], upper_cosine=r["upper_cosine"][i]) for i in keep_sigs])
        self.reference_comparison.update(reference_P=ref, idxs=idx, keep_sigs=keep_sigs, assignments=assignments, votes=votes)
        return dict(assignments=assignments, votes=votes)

    def save_object(self):
        """save_object (R/bayesNMF_sampler.R:414-416): sampler.rds -> sampler.pkl (fields, not the device handle)."""
        keep = {k: v for k, v in self.__dict__.items() if k not in ("_chain", "log_con", "_block_hook")}
        with open(os.path.join(self.specs["output_dir"], "sampler.pkl"), "wb") as f:
            pickle.dump(keep, f)

    def close(self):
        if getattr(self, "log_con", None) is not None:
            self.log_con.write("[INFO] Finalizing and closing log file\n")
            self.log_con.close()
            self.log_con = None
        if getattr(self, "_chain", None) is not None:
            self._chain.close()
            self._chain = None


def bayesNMF(data, rank, likelihood="poisson", prior="truncnormal", rank_method="SBFI", MH=None,
             convergence_control=None, prop_temp=0.2, post_warmup=None, output_dir=None, overwrite=False,
             hyperprior_params=None, init_prior_params=None, init_params=None, periodic_save=True,
             save_all_samples=True, seed=1, chain_id=0, device=0, save_Z=False, engine_factory=None,
             intermediate_credible_intervals=False, n_chains=1, devices=None, engine_side_convergence=True):
    """bayesNMF() (R/bayesNMF.R:24-138): build the sampler and run it; with rank_method = "BIC" run one
    fixed-rank sampler per rank and return dict(results, best_rank, sampler).

    New optional trailing arguments: seed, chain_id, device, save_Z, and n_chains / devices:
    n_chains > 1 runs independent replicas (chain_id = 0..n_chains-1 in the Philox key), chain c on
    devices[c % len(devices)], and returns the list of samplers (multichain.run_chains)."""
    if n_chains and n_chains > 1:
        from .multichain import run_chains
        kw = dict(likelihood=likelihood, prior=prior, rank_method=rank_method, MH=MH, convergence_control=convergence_control,
                  prop_temp=prop_temp, post_warmup=post_warmup, output_dir=output_dir, overwrite=overwrite,
                  hyperprior_params=hyperprior_params, init_prior_params=init_prior_params, init_params=init_params,
                  periodic_save=periodic_save, save_all_samples=save_all_samples, seed=seed, save_Z=save_Z,
                  engine_factory=engine_factory, intermediate_credible_intervals=intermediate_credible_intervals,
                  engine_side_convergence=engine_side_convergence)
        return run_chains(data, rank, n_chains=n_chains, devices=devices, **kw)
    if output_dir is None:
        output_dir = f"nmf_{likelihood}_{prior}"
    common = dict(likelihood=likelihood, prior=prior, rank_method=rank_method, MH=MH,
                  convergence_control=convergence_control, prop_temp=prop_temp, post_warmup=post_warmup,
                  overwrite=overwrite, hyperprior_params=hyperprior_params, init_prior_params=init_prior_params,
                  init_params=init_params, verbosity=1, periodic_save=periodic_save,
                  save_all_samples=save_all_samples, seed=seed, chain_id=chain_id, device=device, save_Z=save_Z,
                  engine_factory=engine_factory, intermediate_credible_intervals=intermediate_credible_intervals,
                  engine_side_convergence=engine_side_convergence)
    ranks = np.atleast_1d(np.asarray(rank, dtype=int))
    if ranks.size > 1 and rank_method == "BIC":
        # One fixed-rank sampler per rank (R/bayesNMF.R:66-126).  The reference runs them one after the other; they are
        # independent, so here they run concurrently, one host thread per sampler, spread over the visible GPUs
        # (SURVEY.md 8 f2); `devices` picks the GPUs, default all of them.
        import concurrent.futures as cf
        if devices is None:
            try:
                from .engine import device_count
                devices = list(range(max(1, device_count()))) if engine_factory is None else [device]
            except Exception:  # noqa: BLE001
                devices = [device]
        common.pop("device")

        def one(i_k):
            i, k = i_k
            s = bayesNMF_sampler(data, int(k), output_dir=os.path.join(output_dir, f"rank_{k}"), device=devices[i % len(devices)], **common)
            s.run_gibbs_sampler()
            bic = float(s.state["MAP_metrics"].iloc[-1]["BIC"])
            return dict(rank=int(k), dir=s.specs["output_dir"], BIC=bic, time=s.time["total"], sampler=s)

        with cf.ThreadPoolExecutor(max_workers=max(1, min(len(ranks), 4 * len(devices)))) as ex:
            results = list(ex.map(one, enumerate(ranks)))
        best = min(results, key=lambda r: r["BIC"])
        return dict(results=pd.DataFrame([{k: v for k, v in r.items() if k != "sampler"} for r in results]).sort_values("BIC"),
                    best_rank=best["rank"], sampler=best["sampler"])
    if ranks.size > 1 and rank_method not in ("SBFI", "BFI"):
        raise ValueError("Rank method must be SBFI, BFI, or BIC")
    sampler = bayesNMF_sampler(data, rank, output_dir=output_dir, **common)
    sampler.run_gibbs_sampler()
    return sampler
