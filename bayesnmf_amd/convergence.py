"""Convergence control and bookkeeping — host-side mirror of R/convergence.R (cheap; stays on the host)."""
import math
import warnings


def new_convergence_control(MAP_over=1000, MAP_every=100, tol=0.001, Ninarow_nochange=5,
                            Ninarow_nobest=10, miniters=1000, maxiters=5000, minA=0,
                            metric="logposterior"):
    """new_convergence_control() (R/convergence.R:16-45): same names, defaults and the
    `miniters >= maxiters` warning that resets miniters to 0."""
    if miniters >= maxiters:
        warnings.warn("miniters >= maxiters, setting miniters to 0.")
        miniters = 0
    return dict(MAP_over=MAP_over, MAP_every=MAP_every, tol=tol, Ninarow_nochange=Ninarow_nochange,
                Ninarow_nobest=Ninarow_nobest, miniters=miniters, maxiters=maxiters, minA=minA,
                metric=metric)


def check_convergence(state, cc, MAP_metric_value, temperature_schedule):
    """check_convergence_ (R/convergence.R:60-154) on the already computed MAP metric of the current
    iteration.  Mutates `state` exactly like the R function mutates self$state; returns the log line."""
    metric = cc["metric"]
    m = MAP_metric_value
    if metric in ("loglikelihood", "logposterior"):
        m = -1.0 * m
    if "prev_MAP_metric" not in state:
        state["prev_MAP_metric"] = m + 1
        state["best_MAP_metric"] = m + 1
        state["inarow_na"] = 0
        state["inarow_no_change"] = 0
        state["inarow_no_best"] = 0
    prev = state["prev_MAP_metric"]
    try:
        pc = (m - prev) / prev
    except ZeroDivisionError:
        pc = float("nan")
    state["prev_percent_change"] = pc
    state["prev_MAP_metric"] = m
    if pc is None or (isinstance(pc, float) and math.isnan(pc)):
        state["inarow_no_change"] = 0
        state["inarow_no_best"] = 0
        state["inarow_na"] += 1
    elif abs(pc) < cc["tol"]:
        state["inarow_no_change"] += 1
        state["inarow_na"] = 0
    else:
        state["inarow_no_change"] = 0
        state["inarow_na"] = 0
    it = state["iter"]
    # temperature_schedule[(iter - MAP_over):iter] in R (1-based, inclusive; index 0 is dropped by R)
    lo = max(it - cc["MAP_over"], 1)
    temps_one = all(t == 1 for t in temperature_schedule[lo - 1:it])
    if temps_one and it >= cc["miniters"]:
        if m < state["best_MAP_metric"]:
            state["best_MAP_metric"] = m
            state["best_iter"] = it
            state["inarow_no_best"] = 0
        else:
            state["inarow_no_best"] += 1
        if state["inarow_no_change"] >= cc["Ninarow_nochange"]:
            state["converged"] = True
            state["why"] = "no change"
        elif state["inarow_no_best"] >= cc["Ninarow_nobest"]:
            state["converged"] = True
            state["why"] = "no best"
        elif it >= cc["maxiters"]:
            state["converged"] = True
            state["why"] = "max iters"
    flip = -1 if metric in ("loglikelihood", "logposterior") else 1
    pcs = "NA" if (isinstance(pc, float) and math.isnan(pc)) else f"{flip * round(pc * 100, 2)}"
    return (f"{metric} = {round(m, 2)} | {pcs}% change | {state['inarow_no_change']} no change | "
            f"{state['inarow_no_best']} no best | {state['inarow_na']} NA")
