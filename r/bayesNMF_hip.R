# r/bayesNMF_hip.R — reference-side binding: how bayesNMF_sampler is switched to the HIP engine.
#
# A subclass of the reference's R6 class that keeps every public field and method and replaces only the
# four private methods the loop body calls (R/bayesNMF_sampler.R:273-285: sample_prior_params,
# sample_params, record_sample, update_sample_metrics) by ONE `.Call` per block of iterations.
# Everything else (get_MAP, check_convergence, logging, save_object, summary/plot) is the reference's
# own code operating on fields that are refreshed from the device at block boundaries.
# EXPERIMENTAL: not runnable in this repository's container (no R).  The Python mirror bayesnmf_amd/sampler.py is
# the tested equivalent and performs the same C-ABI call sequence; keep this file logic-free.  The MAP checks use
# bnmf_map (get_MAP_ on the device: no window copy), the warm-up loop and the MH tail one call each (bnmf_run_until,
# bnmf_run_post_warmup) unless periodic_save asks for the block-by-block loop; the recorded samples are materialised
# into self$samples ONCE, at the end; assign_signatures_ensemble goes through bnmf_assign.  See INTEGRATION.md.

.bnmf_ids <- c(P = 0L, E = 1L, A = 2L, R = 3L, Z = 4L, sigmasq = 7L,
               Alpha_p = 10L, Beta_p = 11L, Alpha_e = 12L, Beta_e = 13L, Mu_p = 14L, Sigmasq_p = 15L,
               Mu_e = 16L, Sigmasq_e = 17L, Lambda_p = 18L, Lambda_e = 19L,
               A_p = 30L, B_p = 31L, C_p = 32L, D_p = 33L, M_p = 34L, S_p = 35L,
               A_e = 40L, B_e = 41L, C_e = 42L, D_e = 43L, M_e = 44L, S_e = 45L,
               P_acceptance_rate = 50L, E_acceptance_rate = 51L)
.bnmf_metric_names <- c("iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank", "temp",
                        "P_mean_acceptance_rate", "E_mean_acceptance_rate")

bayesNMF_sampler_hip <- R6::R6Class(
  "bayesNMF_sampler", inherit = bayesNMF::bayesNMF_sampler,
  public = list(
    handle = NULL,
    initialize = function(..., seed = 1, chain_id = 0L, device = 0L, save_Z = FALSE) {
      private$hip <- list(seed = seed, chain_id = chain_id, device = device, save_Z = save_Z)
      super$initialize(...)     # runs the reference constructor; its prior draws are redirected below
    },
    run_gibbs_sampler = function() {
      cc <- self$specs$convergence_control
      start_time <- Sys.time()
      if (!self$specs$periodic_save) {
        private$absorb(.Call("C_bnmf_run_until", self$handle, private$cc_int(), as.double(cc$tol), private$cc_state()), cc)
      }
      while (!self$state$converged & self$state$iter < cc$maxiters) {        # block-by-block (periodic_save)
        nxt <- (self$state$iter %/% cc$MAP_every + 1) * cc$MAP_every
        private$run_block(min(nxt, cc$maxiters) - self$state$iter, converged = FALSE)
        it <- self$state$iter
        if ((it %% cc$MAP_every == 0 & it >= max(cc$MAP_over, cc$MAP_every)) | it >= cc$maxiters) {
          self$get_MAP()
          msg <- private$check_convergence(); self$log(msg, verbosity = 1)
          if (self$specs$periodic_save) self$save_object()
        }
      }
      if (self$specs$MH) {
        if (!self$specs$periodic_save) {
          private$absorb(.Call("C_bnmf_run_post_warmup", self$handle, private$cc_int(), as.double(cc$tol), private$cc_state(),
                               as.integer(self$specs$post_warmup)), cc)
          self$get_MAP(final = TRUE)
        } else {
          done <- 0
          while (done < self$specs$post_warmup) {
            nxt <- (self$state$iter %/% cc$MAP_every + 1) * cc$MAP_every
            n <- min(nxt - self$state$iter, self$specs$post_warmup - done)
            private$run_block(n, converged = TRUE); done <- done + n
            if (self$state$iter %% cc$MAP_every == 0 | done == self$specs$post_warmup) {
              self$get_MAP(final = done == self$specs$post_warmup)
              private$check_convergence(final = done == self$specs$post_warmup)
              self$save_object()
            }
          }
        }
      } else self$get_MAP(final = TRUE)
      private$pull_state(); private$pull_window()      # self$params / self$samples for summary(), plot(), saveRDS: once
      self$time$total <- difftime(Sys.time(), start_time, units = "mins")
      self$time$per_iter <- self$time$total / self$state$iter
      self$save_object()
    },
    # get_MAP_ (R/utils.R:194-288) on the device: mode of A, renormalised means, 95 % bounds at the final MAP
    get_MAP = function(final = FALSE, credible_interval = 0.95) {
      n <- min(self$specs$convergence_control$MAP_over, self$state$iter)
      r <- .Call("C_bnmf_map", self$handle, as.integer(n), as.double(if (final) credible_interval else 0),
                 c(self$dims$K, self$dims$G, self$dims$N))
      keep <- if (final) which(r$A[1, ] == 1) else seq_len(self$dims$N)
      first <- self$state$iter - n + 1
      pats <- apply(r$top_A[seq_len(min(5, r$n_patterns)), , drop = FALSE], 1, paste, collapse = "")
      self$MAP <- list(P = r$P[, keep, drop = FALSE], A = r$A[, keep, drop = FALSE], E = r$E[keep, , drop = FALSE],
                       idx = first + which(r$used) - 1, A_counts = stats::setNames(r$top_counts[seq_along(pats)], pats),
                       keep_sigs = keep, RMSE = r$rmse, KL = r$kl)
      if (final) self$credible_intervals <- list(P = list(lower = r$P_lower[, keep, drop = FALSE], upper = r$P_upper[, keep, drop = FALSE]),
                                                 E = list(lower = r$E_lower[keep, , drop = FALSE], upper = r$E_upper[keep, , drop = FALSE]))
      invisible(self$MAP)
    },
    # assign_signatures_ensemble_ (R/postprocessing.R:175-341) on the recorded window: cosine matrices on the device
    assign_signatures_ensemble = function(reference_P, credible_interval = 0.95) {
      n <- min(self$specs$convergence_control$MAP_over, self$state$iter)
      used <- rep(FALSE, n); used[self$MAP$idx - (self$state$iter - n)] <- TRUE
      keep <- rep(FALSE, self$dims$N); keep[self$MAP$keep_sigs] <- TRUE
      Pfull <- matrix(0, self$dims$K, self$dims$N); Pfull[, self$MAP$keep_sigs] <- self$MAP$P
      r <- .Call("C_bnmf_assign", self$handle, as.integer(n), used, as.matrix(reference_P), keep, Pfull, as.double(credible_interval),
                 c(self$dims$K, self$dims$G, self$dims$N))
      self$reference_comparison$reference_P <- reference_P
      self$reference_comparison$votes <- r$votes[self$MAP$keep_sigs, , drop = FALSE]
      self$reference_comparison$assignments <- data.frame(sig = self$MAP$keep_sigs, ref = colnames(reference_P)[r$assigned[self$MAP$keep_sigs]],
                                                          cos_sim = r$MAP_cosine[self$MAP$keep_sigs], lower = r$lower[self$MAP$keep_sigs],
                                                          upper = r$upper[self$MAP$keep_sigs])
      self$reference_comparison$idxs <- self$MAP$idx
      invisible(self$reference_comparison)
    }
  ),
  private = list(
    hip = NULL,
    # the constructor's sample_params(from_prior = TRUE) + record_sample + update_sample_metrics
    sample_params = function(skip = c(), from_prior = FALSE) {
      if (!from_prior) stop("per-iteration sampling goes through run_block()")
      lk <- c(poisson = 0L, normal = 1L); pr <- c(truncnormal = 0L, exponential = 1L, gamma = 2L)
      spec <- c(lk[[self$specs$likelihood]], pr[[self$specs$prior]], as.integer(self$specs$MH),
                as.integer(self$specs$learning_rank),
                if (isTRUE(self$specs$rank_method == "BFI")) 1L else 0L, as.integer(private$hip$save_Z),
                as.integer(if (self$specs$save_all_samples) length(self$temperature_schedule)
                           else self$specs$convergence_control$MAP_over))
      storage.mode(self$data) <- "integer"
      self$handle <- .Call("C_bnmf_create", self$data, c(self$dims$K, self$dims$G, self$dims$N), spec,
                           as.double(self$temperature_schedule), as.double(private$hip$seed),
                           as.integer(private$hip$chain_id), as.integer(private$hip$device))
      for (nm in names(self$hyperprior_params)) if (nm %in% names(.bnmf_ids) && is.matrix(self$hyperprior_params[[nm]]))
        .Call("C_bnmf_set_array", self$handle, .bnmf_ids[[nm]], as.double(self$hyperprior_params[[nm]]))
      # prior parameters of iteration 1: the reference's constructor has already filled self$prior_params (user-supplied
      # init_prior_params verbatim, the rest drawn from the hyper-priors in R, R/sample_priors.R:15-141).  All of them go
      # to the device, which keeps supplied arrays verbatim (bnmf_set_array before bnmf_init), so samples$<name>[[1]]
      # equals what the constructor produced (vignettes/advanced.qmd:181-185, :245-249, :315-319).
      for (nm in names(self$prior_params)) if (nm %in% names(.bnmf_ids) && is.matrix(self$prior_params[[nm]]))
        .Call("C_bnmf_set_array", self$handle, .bnmf_ids[[nm]], as.double(self$prior_params[[nm]]))
      for (nm in skip) if (nm %in% names(.bnmf_ids)) .Call("C_bnmf_set_array", self$handle, .bnmf_ids[[nm]], as.double(self$params[[nm]]))
      row <- .Call("C_bnmf_init", self$handle)
      private$pull_state(); private$bind_metrics(matrix(row, ncol = 1))
    },
    cc_int = function() {
      cc <- self$specs$convergence_control
      as.integer(c(cc$MAP_over, cc$MAP_every, cc$Ninarow_nochange, cc$Ninarow_nobest, cc$miniters, cc$maxiters,
                   match(cc$metric, c("loglikelihood", "logposterior", "RMSE", "KL", "BIC")) - 1L))
    },
    cc_state = function() {
      st <- self$state; have <- !is.null(st$prev_MAP_metric)
      nz <- function(x) if (is.null(x)) 0 else x
      as.double(c(isTRUE(st$converged), match(nz(st$why), c("no change", "no best", "max iters"), nomatch = 0), nz(st$best_iter),
                  nz(st$inarow_na), nz(st$inarow_no_change), nz(st$inarow_no_best), have, 0,
                  if (have) st$prev_MAP_metric else 0, if (have) st$best_MAP_metric else 0, nz(st$prev_percent_change)))
    },
    # what an engine-side loop returns -> state$sample_metrics, state$MAP_metrics, the convergence counters
    absorb = function(res, cc) {
      if (ncol(res$metrics) > 0) { private$bind_metrics(res$metrics); self$state$iter <- res$metrics[1, ncol(res$metrics)] }
      mm_names <- c("iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank", "MAP_A_counts", "mean_temp",
                    "P_mean_acceptance_rate", "E_mean_acceptance_rate")
      for (j in seq_len(ncol(res$map_rows))) {
        row <- as.data.frame(t(res$map_rows[1:12, j])); names(row) <- mm_names
        self$state$MAP_metrics <- rbind(self$state$MAP_metrics, row[, names(self$state$MAP_metrics)])
      }
      st <- res$state
      if (st[7] == 1) {
        self$state$prev_MAP_metric <- st[9]; self$state$best_MAP_metric <- st[10]; self$state$prev_percent_change <- st[11]
        self$state$inarow_na <- st[4]; self$state$inarow_no_change <- st[5]; self$state$inarow_no_best <- st[6]
        if (st[3] > 0) self$state$best_iter <- st[3]
      }
      if (st[1] == 1 && !isTRUE(self$state$converged)) {
        self$state$converged <- TRUE; self$state$why <- c("no change", "no best", "max iters")[st[2]]
        self$state$converged_iter <- self$state$iter
      } else if (st[2] > 0) {
        self$state$why <- c("no change", "no best", "max iters")[st[2]]   # check_convergence_ keeps updating `why` during the post-warm-up checks
      }
    },
    record_sample = function() invisible(NULL),          # recorded on the device (bnmf_window)
    update_sample_metrics = function(update_trace = FALSE) invisible(NULL),
    run_block = function(n, converged) {
      met <- .Call("C_bnmf_run", self$handle, as.integer(n), as.logical(converged))
      self$state$iter <- self$state$iter + n
      private$bind_metrics(met); private$pull_state()
    },
    bind_metrics = function(met) {
      df <- as.data.frame(t(met)); names(df) <- .bnmf_metric_names
      self$state$sample_metrics <- rbind(self$state$sample_metrics, df[, names(self$state$sample_metrics)])
    },
    pull_state = function() {
      get <- function(nm, dim) { x <- .Call("C_bnmf_get_array", self$handle, .bnmf_ids[[nm]], as.double(prod(dim))); dim(x) <- dim; x }
      K <- self$dims$K; N <- self$dims$N; G <- self$dims$G
      self$params$P <- get("P", c(K, N)); self$params$E <- get("E", c(N, G)); self$params$A <- get("A", c(1, N))
      self$params$R <- get("R", 1)
      for (nm in names(self$prior_params)) if (nm %in% names(.bnmf_ids))
        self$prior_params[[nm]] <- get(nm, if (grepl("_p$", nm)) c(K, N) else c(N, G))
    },
    # samples[[name]][[i]] of the last n recorded iterations (record_sample, R/bayesNMF_sampler.R:651-672): every name the
    # reference records (params, prior_params, acceptance rates, sigmasq).  With save_all_samples the lists are indexed
    # by iteration (as in the reference) and MAP_idx is the last MAP_over iterations; otherwise positions 1..n.
    pull_window = function() {
      n <- min(self$specs$convergence_control$MAP_over, self$state$iter)
      K <- self$dims$K; N <- self$dims$N; G <- self$dims$G
      nms <- c("P", "E", "A", "R", names(self$prior_params))
      if (self$specs$MH) nms <- c(nms, "P_acceptance_rate", "E_acceptance_rate")
      if (self$specs$likelihood == "normal") nms <- c(nms, "sigmasq")
      first <- if (self$specs$save_all_samples) self$state$iter - n + 1 else 1
      for (nm in intersect(nms, names(.bnmf_ids))) {
        d <- if (nm == "A") c(1, N) else if (nm == "R") 1 else if (nm == "sigmasq") G else
             if (nm == "P" || grepl("_p$", nm) || nm == "P_acceptance_rate") c(K, N) else c(N, G)
        w <- .Call("C_bnmf_window", self$handle, .bnmf_ids[[nm]], as.integer(n), as.double(prod(d)))
        if (is.null(self$samples[[nm]])) self$samples[[nm]] <- list()
        for (i in seq_len(n)) self$samples[[nm]][[first + i - 1]] <- array(w[, i], dim = d)
      }
      self$state$MAP_idx <- seq(first, first + n - 1)
    }
  )
)
