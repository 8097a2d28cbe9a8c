#!/usr/bin/env python3
"""bench.py — Gibbs iterations/sec of the HIP engine at BASELINE.json's metric config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one whole Gibbs iteration of the reference's loop body (R/bayesNMF_sampler.R:273-285):
hyper sweep -> P -> E -> Z allocation -> record_sample into the MAP_over = 1000 deep device ring ->
metrics row, for the Poisson-Gamma model at K=96, G=10,000, N=20 on synthetic counts already resident
in HBM.  One independent chain per GPU (chain_id = rank); no data-path collective; RCCL only gathers
the chains' final metrics rows.  Rank 0 prints ONE JSON line.

`--gpus N` without WORLD_SIZE in the environment starts the N ranks itself (fresh child processes,
before anything here touches the GPU).  The timed region is repeated `--reps` times (default 9), each
repetition timing EXACTLY `--steps` iterations between barriers; `value` is the median repetition
(SURVEY.md 8d protocol) and every repetition is listed under `rep_values`.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_, G_, N_, R_TRUE, DATA_SEED = 96, 10000, 20, 8, 20250218
MAP_OVER = 1000                # new_convergence_control() default: depth of the record_sample ring
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_TAG = "r05"                         # profiles/<tag>_pmc_*.json: the committed counter passes the roofline quotes (falls back to r03)


def z_bytes(K, G, N, save_Z):
    """Algorithmic bytes one k_zalloc launch must move (state is fp64):
    M int32 + E fp64 + P fp64 read; ZsumK int32 + ZsumG int32 (+ Z int32) written."""
    b = 4 * K * G + 8 * N * G + 8 * K * N + 4 * N * G + 4 * K * N
    if save_Z:
        b += 4 * K * N * G
    return b


def pmc_valu_busy(save_Z, simds=1024):
    """Fraction of the SIMDs' cycles the hot kernel's VALU pipe is busy, from the COMMITTED SQ counters (rocprofv3 --pmc,
    tools/pmc_r3.sh; not measured in this run): SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / SQ_WAVES x SIMDs) — both count
    quad-cycles, and the kernel's waves are persistent (one set per launch), so WAVE_CYCLES / WAVES is the launch's length.
    Returns (value, source file) or (None, None)."""
    if save_Z:
        return None, None                                                  # the save_Z kernel's waves are not persistent
    for tag in (PROFILE_TAG, "r03"):
        try:
            name = f"profiles/{tag}_pmc_counters.json"
            d = json.load(open(os.path.join(ROOT, name)))["p3"]
            key = next(k.split(":")[0] for k in d if k.startswith("k_zalloc_sort") and k.endswith(":SQ_WAVES"))
            waves, wc, act = (d[f"{key}:{c}"]["mean"] for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"))
            return act / (wc / waves * float(simds)), name
        except Exception:
            continue
    return None, None


def pmc_traffic(save_Z):
    """HBM bytes per k_zalloc launch from the COMMITTED PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    rocprofv3 --pmc passes, gfx950-corrected; not measured in this run).  Returns (bytes, source file) or (None, None)."""
    for tag in (PROFILE_TAG, "r03", "r01"):
        try:
            name = f"profiles/{tag}_pmc_traffic.json"
            d = json.load(open(os.path.join(ROOT, name)))
            return d["k_zalloc_full" if save_Z else "k_zalloc_stats"]["hbm_bytes_per_launch"], name
        except Exception:
            continue
    return None, None


_CEILINGS = {}


def ceilings(device):
    """Measured ceilings of this box: Philox4x32-7 words/s (the count-allocation stream's generator) with nothing else in the loop (one word per
    allocated count is the floor of any allocation kernel) and the device-to-device copy bandwidth, next to the nominal 8 TB/s."""
    if device not in _CEILINGS:
        from bayesnmf_amd.engine import ubench
        _CEILINGS[device] = ubench(device)
    return _CEILINGS[device]


def roofline_of(chain, K, G, N, save_Z, total_counts, n_iter, kernel=None, device=0):
    prof = chain.profile(n_iter)
    zb = z_bytes(K, G, N, save_Z)
    z_ms = prof["k_zalloc"]
    achieved = zb / (z_ms * 1e-3) / 1e9 if z_ms > 0 else 0.0
    philox_peak, copy_gbs = ceilings(device)
    draws = total_counts / (z_ms * 1e-3) if z_ms > 0 else 0.0
    if kernel is None:
        kernel = "k_zalloc_sort with save_Z (item records) + k_zexpand (records -> Z)" if save_Z else "k_zalloc_sort"
    traffic, traffic_src = pmc_traffic(save_Z)
    try:
        import torch
        simds = 4 * torch.cuda.get_device_properties(device).multi_processor_count
    except Exception:
        simds = 1024
    busy, busy_src = pmc_valu_busy(save_Z, simds)
    return prof, {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                  "traffic_source": f"{traffic_src} (committed rocprofv3 --pmc passes of this kernel, not collected in this run)" if traffic_src else None,
                  "peak_measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs > 0 else None,
                  "algorithmic_bytes_per_launch": zb, "avg_launch_ms": z_ms,
                  "draws_per_s": draws,
                  # the bound that actually binds in stats mode: one Philox word per allocated count
                  "alu": {"achieved_philox_words_per_s": draws, "peak_philox_words_per_s": philox_peak,
                          "frac": draws / philox_peak if philox_peak > 0 else None,
                          "valu_busy_frac": busy,
                          "valu_busy_source": f"{busy_src} (committed SQ counters, not collected in this run)" if busy_src else None,
                          "note": "peak = Philox4x32-7 (the generator of the count-allocation words) alone at 8 waves/SIMD, measured on this box (bnmf_ubench); the kernel also "
                                  "searches 19 thresholds and updates two tables per word"},
                  "column_terms_launch_ms": prof.get("other"),
                  "note": "stats mode moves 6.3 MB per 40 M categorical draws: the kernel is VALU/Philox-bound, not HBM-bound (DESIGN.md 5); since round 5 it leaves "
                          "Mhat (8KG bytes, not credited) for the per-column metric terms, which are summed beside the next allocation kernel (in this serialised "
                          "profile pass: a launch of their own, column_terms_launch_ms)"
                          if not save_Z else "full mode: Z (K x N x G int32) written every iteration: k_zalloc_sort leaves one packed record per item, k_zexpand turns the records of a column slab into Z with 16-byte stores (both in avg_launch_ms)"}


def full_mode(M, total_counts, args, device):
    """Full mode (save_Z: samples$Z, R/bayesNMF_sampler.R:245-252) AS A THROUGHPUT, same protocol as the headline (median of repetitions of
    --steps iterations, a metrics row every iteration, record_sample on — here into a 100-deep window: 1000 samples of Z do not fit any budget).
    Two forms, both reported:
      records       what the library does: the allocation kernel writes every sample of Z as its item records (two 16-bit counts per word
                    and item; `record_bytes` per iteration) into the sample's slot of a ring; Z[k,n,g] itself is expanded from the records when
                    it is read (bnmf_get_array / bnmf_window: `expand_ms` per sample).  Bytes credited: the stats-mode bytes + the records.
      materialised  BNMF_ZEAGER=1: k_zexpand runs every iteration and Z (K x N x G int32) is written every iteration — the form SURVEY 8(d)'s
                    B_full = B_stats + 4KNG describes; frac = B_full x rate / 8 TB/s."""
    import torch
    K, G = M.shape
    steps = max(args.steps, 200)
    res = {}
    for name, eager in (("records", False), ("materialised", True)):
        if eager:
            os.environ["BNMF_ZEAGER"] = "1"
        try:
            c = make_chain(M, N_, seed=1, chain_id=0, device=device, save_Z=True, window=100)
        finally:
            os.environ.pop("BNMF_ZEAGER", None)
        c.run(300, metrics=False)
        vals = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); c.run(steps, metrics=True); torch.cuda.synchronize()
            vals.append(steps / (time.perf_counter() - t0))
        rate = float(np.median(vals))
        rec_bytes = c.stat(0)
        t0 = time.perf_counter(); z = c.get("Z"); t_get = time.perf_counter() - t0
        assert (z.sum(axis=1) == M).all()                        # every count of the last iteration is in Z
        b_stats = z_bytes(K, G, N_, False)
        nbytes = b_stats + (4 * K * N_ * G if eager else rec_bytes)
        res[name] = {"value": rate, "unit": "Gibbs iterations/s", "steps": steps, "rep_values": vals, "record_bytes_per_iteration": rec_bytes,
                     "samples_Z_kept": "ring of records (window 100)" if c.stat(2) else "last iteration only",
                     "roofline": {"bound": "hbm", "kernel": "k_zalloc_sort (records)" + (" + k_zexpand (records -> Z) every iteration" if eager else ""),
                                  "algorithmic_bytes_per_iteration": nbytes, "achieved": nbytes * rate / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": nbytes * rate / 1e9 / HBM_PEAK_GBS},
                     "get_Z_ms": 1e3 * t_get}
        prof = c.profile(60)
        res[name]["kernel_ms"] = prof
        c.close()
    res["note"] = ("samples$Z is kept as item records and expanded on read; 'materialised' writes Z every iteration (the expansion cannot share a CU with "
                   "the allocation or the draw kernel: its slab is 150 KB of LDS, theirs 153 / 45 KB with 14 of 16 wave slots — DESIGN.md 5)")
    return res


def make_chain(M, N, seed, chain_id, device, save_Z=False, window=MAP_OVER, prior="gamma", **kw):
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import apply_hyperprior_params
    e = Engine(M, N, prior=prior, seed=seed, chain_id=chain_id, device=device, save_Z=save_Z, window=window, **kw)
    apply_hyperprior_params(e, prior, M, N)
    e.init()
    return e


def _time_oracle(M, N, cores, budget_s):
    import oracle as O
    from bayesnmf_amd.setup import apply_hyperprior_params
    o = O.Oracle(M, N, prior="gamma", seed=1, nthreads=cores)
    apply_hyperprior_params(o, "gamma", M, N)
    o.init()
    t0 = time.perf_counter(); o.run(1); dt = time.perf_counter() - t0
    n = int(max(2, min(200, budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter(); o.run(n); dt = time.perf_counter() - t0
    o.close()
    return n / dt, n


def probe_rscript(M, N):
    """the reference pure-R path can be timed only where the box has Rscript with the bayesNMF
    package installed.  Returns its it/s (config-1-sized sample, R is single-threaded) or the reason it cannot run."""
    rs = shutil.which("Rscript")
    if not rs:
        return {"available": False, "reason": "Rscript not on PATH"}
    try:
        r = subprocess.run([rs, "-e", "suppressMessages(library(bayesNMF))"], capture_output=True, timeout=60)
        if r.returncode != 0:
            return {"available": False, "reason": "library(bayesNMF) failed: " + r.stderr.decode()[-200:]}
        import tempfile
        d = tempfile.mkdtemp()
        np.savetxt(os.path.join(d, "M.csv"), M[:, :100], fmt="%d", delimiter=",")
        code = (f"suppressMessages(library(bayesNMF)); M <- as.matrix(read.csv('{d}/M.csv', header=FALSE)); "
                f"cc <- new_convergence_control(maxiters=20, MAP_over=10, MAP_every=10, miniters=0); t0 <- Sys.time(); "
                f"s <- bayesNMF(M, rank={N}, prior='gamma', likelihood='poisson', MH=FALSE, periodic_save=FALSE, "
                f"save_all_samples=FALSE, convergence_control=cc, output_dir='{d}/out', overwrite=TRUE); "
                f"cat(as.numeric(difftime(Sys.time(), t0, units='secs')))")
        r = subprocess.run([rs, "-e", code], capture_output=True, timeout=600)
        secs = float(r.stdout.decode().strip().split()[-1])
        return {"available": True, "value": 20.0 / secs, "unit": "Gibbs iterations/s", "cores": 1, "kind": "reference",
                "sample": f"20 iterations of bayesNMF() in R on the first 100 columns (K={M.shape[0]}, N={N})"}
    except Exception as ex:  # noqa: BLE001
        return {"available": False, "reason": repr(ex)[:200]}


def cpu_baseline(M, N, budget_s=10.0):
    """The CPU oracle ("port" of the reference sweep) timed on this box's host cores on a bounded sample of
    the same workload: all cores (OpenMP over columns) and one core.  The reference itself is pure R: it is
    probed for and timed only if this box happens to have it installed."""
    cores = min(os.cpu_count() or 1, 64)
    v_all, n_all = _time_oracle(M, N, cores, budget_s)
    v_one, n_one = _time_oracle(M, N, 1, budget_s)
    return {"value": v_all, "unit": "Gibbs iterations/s", "cores": cores, "kind": "port",
            "value_1core": v_one,
            "sample": f"{n_all} iterations on {cores} cores and {n_one} iterations on 1 core of the same K={M.shape[0]}, "
                      f"G={M.shape[1]}, N={N} Poisson-Gamma sweep (CPU oracle: scalar C, OpenMP over columns)",
            "reference_R": probe_rscript(M, N)}


def secondary_configs(device, quick=False):
    """BASELINE.json configs 2-5 at full size on one GPU (one chain each), each with the HBM roofline of its
    allocation / dominant kernel.  Parity for these shapes is in tests/; these are timings only."""
    from bayesnmf_amd.setup import synth_counts
    out = []

    def run(name, K, G, N, prior, iters, rtrue, seed_off, kernel, **kw):
        t0 = time.perf_counter()
        M, _, _ = synth_counts(K, G, rtrue, DATA_SEED + seed_off)
        c = make_chain(M, N, 1, 0, device, prior=prior, **kw)
        c.run(max(2, iters // 5), metrics=False)
        rec = {"config": name, "K": K, "G": G, "N": N, "window": kw.get("window", MAP_OVER)}
        phases = [("it_per_s", False)] + ([("it_per_s_after_convergence", True)] if kw.get("MH") else [])
        for key, conv in phases:
            vals = []
            for _ in range(3):                                # median of three timed blocks (a block is 10-300 ms)
                t1 = time.perf_counter(); c.run(iters, converged=conv, metrics=True); vals.append(iters / (time.perf_counter() - t1))
            rec[key] = sorted(vals)[1]
        if not kw.get("MH"):
            zb = z_bytes(K, G, N, False)
            prof = c.profile(max(3, iters // 10))
            zms = prof["k_zalloc"]
            rec["roofline"] = {"bound": "hbm", "kernel": kernel, "algorithmic_bytes_per_launch": zb, "avg_launch_ms": zms,
                               "achieved": zb / (zms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": zb / (zms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            rec["kernel_ms"] = prof
        else:
            # 2N sequential factor steps per iteration on data that stays in L2 (M, Mhat, E: a few MB): latency-bound, not an
            # HBM stream.  What is reported is the time per factor step and the CUs the P side can use (one workgroup per row).
            prof = c.profile(max(3, iters // 10))
            rec["kernel_ms"] = prof
            rec["roofline"] = {"bound": "latency (sequential-in-n sweeps on L2-resident data; no HBM or MFMA roofline applies)",
                               "kernel": "k_mh_prow + k_mh_ecol16", "us_per_factor_step": 1e6 / (rec["it_per_s"] * 2 * N),
                               "us_per_factor_step_after_convergence": 1e6 / (rec["it_per_s_after_convergence"] * 2 * N),
                               "P_side_workgroups": K, "CUs": 256}
        if kw.get("learning_rank") and "kernel_ms" in rec:
            # the persistent rank sweep: N sequential decisions, each one alternative Poisson log-likelihood per cell (fp64 log)
            rk_ms = rec["kernel_ms"].get("k_rank", 0.0)
            rec["roofline_k_rank_sweep"] = {"bound": "fp64 VALU (software log) + per-factor exchange", "avg_launch_ms": rk_ms,
                                            "log_evaluations_per_launch": N * K * G,
                                            "achieved_log_per_s": N * K * G / (rk_ms * 1e-3) if rk_ms > 0 else None,
                                            "us_per_factor": 1e3 * rk_ms / N}
        rec["setup_s"] = time.perf_counter() - t0
        c.close()
        out.append(rec)

    run("2: Poisson-Gamma fixed rank N=20, K=96 x G=2,000", 96, 2000, 20, "gamma", 1000, 8, 2, "k_zalloc_sort")
    run("3: Poisson-TruncNormal+MH fixed rank N=20, K=96 x G=5,000", 96, 5000, 20, "truncnormal", 60, 8, 3, "k_mh", MH=True)
    run("4: Poisson-Gamma SBFI learned rank 1:50, K=96 x G=10,000", 96, 10000, 50, "gamma", 60, 12, 4, "k_zalloc_step",
        learning_rank=True, rank_method="SBFI", temperature=np.ones(8000))
    if not quick:
        run("5: Poisson-Gamma fixed rank N=100, K=1,536 x G=50,000", 1536, 50000, 100, "gamma", 12, 30, 5, "k_zalloc_step",
            window=2)
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as fresh children (this process has not
    touched the GPU), pass rank 0's JSON line through."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=9, help="repetitions of the timed K-step region; value = median")
    ap.add_argument("--window", type=int, default=MAP_OVER, help="record_sample ring depth (MAP_over); 0 = recording off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the timings of BASELINE configs 2-5")
    ap.add_argument("--save-z", action="store_true", help="full mode: materialise Z every iteration")
    ap.add_argument("--G", type=int, default=G_)
    ap.add_argument("--force-dist", action="store_true", help="initialise the nccl (RCCL) process group and run the gather / all-reduce "
                                                               "legs even with one rank (exercises the collectives on a one-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # REHEARSAL ONLY (a one-GPU box): BNMF_BENCH_REHEARSE=1 puts every rank on GPU 0 and gathers over gloo, to exercise
    # the multi-rank code path; its numbers are meaningless and the line says so.
    rehearse = os.environ.get("BNMF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    tdev = "cpu" if rehearse else "cuda"
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1])); sk.close()
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on STDOUT when its first communicator is created: keep the contract (rank 0 prints ONE
        # JSON line) by pointing fd 1 at stderr while the group is set up and the first collective runs
        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd, 1)
            os.close(keep_fd)
    else:
        torch.cuda.set_device(local_rank)

    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(K_, args.G, R_TRUE, DATA_SEED)
    total_counts = int(M.sum())
    chain = make_chain(M, N_, seed=1, chain_id=rank, device=local_rank, save_Z=args.save_z, window=args.window)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    chain.run(args.warmup, metrics=False)
    n_prof = min(200, max(20, args.steps // 10))
    n_refill = max(args.warmup, 1500)   # (0.14 s at the metric configuration: the device clocks have settled by the first timed repetition)
    # roofline of the dominant kernel (k_zalloc): HIP events on the chain's own stream, one kernel at a time.
    # Done before the timed region (it advances the chain like any other iterations and keeps the clocks up).
    prof, roof = roofline_of(chain, K_, args.G, N_, args.save_z, total_counts, n_prof, device=local_rank)
    chain.run(n_refill, metrics=False)  # refill the stream pipeline after the serialised profile pass; keeps the clocks up
    n_shape = 2                         # ... and two untimed repetitions of exactly the timed shape (on some boxes the first calls after the
    for _ in range(n_shape):            # profile pass run 3 % slower than the ones behind them)
        chain.run(args.steps, metrics=True)

    rep_dt = []
    met = None
    for _ in range(max(1, args.reps)):
        barrier()
        t0 = time.perf_counter()
        met = chain.run(args.steps, metrics=True)       # synchronises the chain's streams at the end
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rep_dt.append(float(tmax.item()))
    gathered = None
    if dist is not None:
        from bayesnmf_amd.multichain import gather_rows
        gathered = gather_rows(met[-1:], dist, device=tdev)     # RCCL: gather the chains' last metrics rows
        # ... and (SURVEY 8e ii) the chains' MAP window statistics: mode of A and the renormalised window means of P and E from
        # bnmf_map on every rank (N + K N + N G doubles per chain), all-gathered like the launcher does at a MAP check
        mp = chain.map(min(args.window, 200), None) if args.window > 0 else None
        gathered_map = None
        if mp is not None:
            vec = np.concatenate([np.ravel(mp["A"]), np.ravel(mp["P"], order="F"), np.ravel(mp["E"], order="F")]).reshape(1, -1)
            gathered_map = gather_rows(vec, dist, device=tdev)
    tmed = float(np.median(rep_dt))

    out = None
    if rank == 0:
        out = {
            "metric": "Gibbs iters/sec at K=96, G=10k, N=20; per-chain scaling at 1/2/4/8 GPUs",
            "value": world * args.steps / tmed,
            "unit": "Gibbs iterations/s (aggregate over chains)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # what ran before the first timed repetition: --warmup iterations, the serialised per-kernel profile pass the roofline
            # comes from, a refill of the stream pipeline, and two untimed repetitions of the timed shape
            "warmup_effective": args.warmup + n_prof + n_refill + n_shape * args.steps,
            "rng": "philox4x32-7 (count-allocation words, variable Z) / philox4x32-10 (every other stream)",
            "ms_per_step": 1e3 * tmed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            **({"REHEARSAL": "all ranks on one GPU over gloo: not a measurement"} if rehearse else {}),
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Poisson-Gamma fixed rank N={N_}, K={K_} x G={args.G} synthetic counts, "
                                   f"one chain per GPU, {'full (save_Z)' if args.save_z else 'stats'} mode, "
                                   f"record_sample into a {args.window}-deep device ring "
                                   f"({'on' if args.window > 0 else 'OFF'}) and a metrics row every iteration",
                       "chains": world, "seed": 1, "sum_M": total_counts, "record_window": args.window},
            "reps": len(rep_dt), "rep_values": [world * args.steps / t for t in rep_dt],
            "roofline": roof,
            "kernel_ms": prof,
        }
        if world == 1 and not args.save_z:
            out["full_mode"] = full_mode(M, total_counts, args, local_rank)
            out["value_save_Z"] = out["full_mode"]["records"]["value"]
            out["roofline_save_Z"] = out["full_mode"]["materialised"]["roofline"]
        if gathered is not None:
            out["chains_final_logposterior"] = [float(g[0][4]) for g in gathered]
            out["collectives"] = {"backend": "gloo" if rehearse else "nccl (RCCL)", "world": world, "forced_on_one_rank": bool(args.force_dist and world == 1),
                                  "gathered": ["last metrics row per chain"] + ([f"MAP window statistics per chain ({gathered_map.shape[2]} doubles)"] if gathered_map is not None else [])}
    chain.close()
    if rank == 0:
        if world == 1 and not args.no_secondary:
            try:
                out["secondary"] = secondary_configs(local_rank)
            except Exception as ex:  # noqa: BLE001  (a secondary timing must not lose the headline line)
                out["secondary_error"] = repr(ex)[:300]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, N_)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
