// bayesnmf_amd/csrc/zalloc_reg.h — k_zalloc_reg: register-resident variant of the Z-allocation
// kernel for N <= 25 (the metric configuration has N = 20).
//
// Same stream spec and bit-identical results as k_zalloc (kernels.h), different machine mapping:
// the bucket of a count is found with a two-level search: the pivots of the cell's threshold row
// (every 4th threshold) live in registers (reloaded when the lane's contiguous quad range crosses
// into the next cell), compare+add-carry pairs pick the 4-block, ONE 128-bit LDS read fetches it and
// four more compares finish: half the VALU of a full register compare, one LDS read per count
// instead of the six dependent reads of a binary search.  Each count then costs two LDS atomics: zacc[n][k] (bank = k, so lanes on
// different cells never collide) and the lane's private packed 8-bit histogram used for ZsumK.
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZNMAX = 25;          // max N of the register path (<= 24 thresholds)

// b_j += (T <= u_j) for four counts at once.  Measured on gfx950 (tools/ubench.hip): the plain C++
// form (v_cmp -> VCC -> v_addc, with the compiler's hazard nops) issues at ~2.2 cycles per
// compare-accumulate per SIMD; hand-written e64 forms through SGPR pairs are 2x SLOWER (4.2).
BNMF_DEV void cmp_acc4(uint32_t T, uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3,
                       uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  b0 += (T <= u0) ? 1u : 0u;
  b1 += (T <= u1) ? 1u : 0u;
  b2 += (T <= u2) ? 1u : 0u;
  b3 += (T <= u3) ? 1u : 0u;
}

// v_cvt_u32_f64 saturates (x >= 2^32 -> 0xFFFFFFFF, x < 0 / NaN -> 0): C's (uint32_t)x is undefined there
BNMF_DEV uint32_t cvt_u32_sat(double x) {
  uint32_t r;
  asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips)
BNMF_DEV int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// only what this kernel needs (the full Dev by value costs ~60 SGPRs and made the hot loop spill SGPRs)
struct ZArgs {
  int K, G, N, maxM;
  uint32_t k0, k1;
  const int32_t* M;
  const double *P, *E, *A;
  int32_t *ZsumK, *ZsumG, *Z;
  double *colsse, *colll, *colkl;
  const double *lgfact, *logm;
};
template <bool SAVE_Z, int ZT, int TRC /* threshold registers: multiple of 4, >= N-1 */>
__global__ __launch_bounds__(ZT, 4) void k_zalloc_reg(ZArgs d, uint32_t t, ZGeom zg, int ablate) {
  constexpr int ZW = ZT / 64;
  constexpr int NC = TRC + 1;                            // factors covered by this instantiation
  constexpr int NMIN = TRC == 8 ? 1 : TRC == 16 ? 10 : TRC - 2;   // smallest N routed here (api.hip: zg.TR)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KP = zg.KP, HW = zg.HW;
  const int KR = (K + 63) >> 6;
  uint32_t* zacc = (uint32_t*)smem;                      // [N][KP] shared by the workgroup
  double* Pl = (double*)(zacc + zg.zacc_words);          // [N][K] P, shared by the workgroup
  uint32_t* slab = zacc + zg.zacc_words + zg.p_words + (size_t)wave * zg.slab_words;
  uint32_t* hist = slab;                                 // [HW][ZH] per-lane packed 8-bit bucket counts
  uint32_t* thr = hist + HW * ZH;                        // [K][TRC] threshold rows (16-B aligned)
  double* ae = (double*)(thr + (size_t)K * TRC);         // [N]  A[n] * E[n,g]
  uint32_t* qoff = (uint32_t*)(ae + N);                  // [K+1] quad offset (30 bits) | (-M mod 4) << 30
  uint32_t* zkt = qoff + K + 1;                          // [N] column totals
  uint32_t* zloc = zkt + N;                              // [N][KP]  (SAVE_Z only)
  for (int i = tid; i < N * KP; i += ZT) zacc[i] = 0;
  for (int i = tid; i < K * N; i += ZT) Pl[i] = d.P[i];
  for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
  for (int i = lane; i < N; i += 64) zkt[i] = 0;
  if (SAVE_Z) for (int i = lane; i < N * KP; i += 64) zloc[i] = 0;
  __syncthreads();
  const int nthr = N - 1;
  uint32_t* ztarget = SAVE_Z ? zloc : zacc;
  const int gw = blockIdx.x * ZW + wave, nw = gridDim.x * ZW;
  // Phase-synchronised: all waves of the workgroup run phase 1 together and phase 2 together
  // (ablate bit 512 turns the barriers off).  In phase 2 the SIMD then always has several waves in the
  // compare/Philox stream, which hides the VALU->VCC hazard slots a lone wave would stall on.
  const bool psync = (ablate & 512) != 0;      // measured: no gain, off by default
  const int nround = (G - (int)(blockIdx.x * ZW) + nw - 1) / nw;   // same for every wave of the workgroup
  for (int rr = 0; rr < nround; ++rr) {
    const int g = gw + rr * nw;
    const bool active = g < G;
    if (!active && !psync) break;
    if (active) {
    // ---------------- phase 1: one pass over the factors, thresholds written as 128-bit rows
    double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
    int carry = 0;
    const double* Eg = d.E + (size_t)N * g;
    for (int n = lane; n < N; n += 64) ae[n] = d.A[n] * Eg[n];
    wave_lds_fence();
    for (int r = 0; r < KR; ++r) {
      const int kk = (r << 6) + lane;
      int q = 0, m = 0;
      if (kk < K) {
        const double* Pk = Pl + kk;                       // P[kk, n] = Pl[kk + K n] (workgroup copy in LDS)
        m = d.M[kk + (size_t)K * g];
        double c = 0.0;
        double pv[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) {
          if (n < NMIN || n < N) c = c + Pk[(size_t)K * n] * ae[n];
          pv[n] = c;
        }
        if (c > 0.0 && m > 0) {
          // thr_n = floor(cum_n 2^32 / Mhat), saturating at 2^32-1 = "never".  Factors at/after the last
          // positive one have cum_n == Mhat, and fl(Mhat * fl(2^32/Mhat)) >= 2^32 (1 - 2^-52) > 2^32 - 1,
          // so they saturate by themselves (the oracle's explicit `n >= nlast` test is the same set).
          const double scale = 4294967296.0 / c;
          u4* row = (u4*)(thr + (size_t)kk * TRC);
#pragma unroll
          for (int j = 0; j < TRC / 4; ++j) {
            uint32_t tv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int n = 4 * j + i;
              uint32_t tj = 0xFFFFFFFFu;
              if (n < NMIN - 1 || n < nthr) tj = cvt_u32_sat(pv[n] * scale);
              tv[i] = tj;
            }
            row[j] = u4{tv[0], tv[1], tv[2], tv[3]};
          }
          q = (m + 3) >> 2;
        }
        const double dd = c - (double)m;
        const double mh = c < 1e-6 ? 1e-6 : c;
        const double lmh = dlog(mh);
        const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
        const double mt = m < 1 ? 1e-6 : (double)m;
        a_sse = a_sse + dd * dd;                          // canonical: lane l adds rows l, l+64, ...
        a_ll = a_ll + (((double)m * lmh - mh) - d.lgfact[mi]);
        a_kl = a_kl + mt * (d.logm[mi] - lmh);
      }
      const int incl = wave_incl_scan_dpp(q);
      if (kk < K) qoff[kk] = (uint32_t)(carry + incl - q) | ((uint32_t)((4 - (m & 3)) & 3) << 30);
      carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) qoff[K] = (uint32_t)carry;
    a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
    if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
    wave_lds_fence();
    }
    if (psync) __syncthreads();
    if (active) {
    const int Q = (int)qoff[K];
    // ---------------- phase 2: contiguous quad range per lane, thresholds in registers; chunks of
    // <= 63 quads so that the packed 8-bit per-lane histogram cannot overflow
    const int per = (Q + 63) >> 6;
    for (int cbase = 0; cbase < per; cbase += 63) {
      const int q0 = min(Q, lane * per + cbase);
      const int q1 = min(Q, min(lane * per + per, q0 + 63));
      if (q0 < q1 && !(ablate & 2)) {
        int cell;
        {  // upper_bound(qoff[0..K] & mask, q0) - 1, branch-free
          int b = 0, len = K + 1;
          while (len > 1) { const int half = len >> 1; b = ((int)(qoff[b + half - 1] & 0x3FFFFFFFu) <= q0) ? b + half : b; len -= half; }
          cell = b + ((int)(qoff[b] & 0x3FFFFFFFu) <= q0 ? 1 : 0) - 1;
        }
        int cstart = 0, cend = -1, mc = 0;
        constexpr int NPV = TRC / 4 - 1;                  // pivots: last threshold of every 4-block but the last
        uint32_t PV[NPV > 0 ? NPV : 1];
        const uint32_t* row = thr;
        --cell;
        uint32_t* hl = hist + lane;
        for (int qi = q0; qi < q1; ++qi) {
          if (qi >= cend) {                               // enter the cell that holds quad qi
            uint32_t qa, qb;
            do { ++cell; qa = qoff[cell]; qb = qoff[cell + 1]; cend = (int)(qb & 0x3FFFFFFFu); } while (qi >= cend);
            cstart = (int)(qa & 0x3FFFFFFFu);
            mc = ((cend - cstart) << 2) - (int)(qa >> 30);
            row = thr + (size_t)cell * TRC;
#pragma unroll
            for (int j = 0; j < NPV; ++j) PV[j] = row[4 * j + 3];
          }
          const int j0 = (qi - cstart) << 2;
          const int nd = mc - j0;                          // >= 1; counts of this quad = min(4, nd)
          u32x4 w;
          if (ablate & 16) w = u32x4{(uint32_t)qi * 2654435761u, (uint32_t)qi * 40503u, (uint32_t)qi, ~(uint32_t)qi};
          else w = philox4x32_10((uint32_t)(j0 >> 2), (uint32_t)(cell + (size_t)K * g), t, BNMF_V_Z, d.k0, d.k1);
          const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
          uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
          if (!(ablate & 8)) {
            // two-level search: pivots in registers pick the 4-block, one 128-bit LDS read fetches it
            uint32_t j0b = 0, j1b = 0, j2b = 0, j3b = 0;
#pragma unroll
            for (int j = 0; j < NPV; ++j) cmp_acc4(PV[j], u0, u1, u2, u3, j0b, j1b, j2b, j3b);
            const u4 k0 = *(const u4*)(row + 4 * j0b), k1 = *(const u4*)(row + 4 * j1b);
            const u4 k2 = *(const u4*)(row + 4 * j2b), k3 = *(const u4*)(row + 4 * j3b);
            b0 = 4 * j0b + (k0.x <= u0) + (k0.y <= u0) + (k0.z <= u0) + (k0.w <= u0);
            b1 = 4 * j1b + (k1.x <= u1) + (k1.y <= u1) + (k1.z <= u1) + (k1.w <= u1);
            b2 = 4 * j2b + (k2.x <= u2) + (k2.y <= u2) + (k2.z <= u2) + (k2.w <= u2);
            b3 = 4 * j3b + (k3.x <= u3) + (k3.y <= u3) + (k3.z <= u3) + (k3.w <= u3);
          }
          uint32_t* zc = ztarget + cell;
          if (ablate & 4) { if (b0 + b1 + b2 + b3 == 0xFFFFFFF0u) zc[0] = w.x; }
          else {
            atomicAdd(&zc[b0 * KP], 1u); atomicAdd(&hl[(b0 >> 2) * ZH], 1u << ((b0 & 3) << 3));
            if (nd > 1) { atomicAdd(&zc[b1 * KP], 1u); atomicAdd(&hl[(b1 >> 2) * ZH], 1u << ((b1 & 3) << 3)); }
            if (nd > 2) { atomicAdd(&zc[b2 * KP], 1u); atomicAdd(&hl[(b2 >> 2) * ZH], 1u << ((b2 & 3) << 3)); }
            if (nd > 3) { atomicAdd(&zc[b3 * KP], 1u); atomicAdd(&hl[(b3 >> 2) * ZH], 1u << ((b3 & 3) << 3)); }
          }
        }
      }
      wave_lds_fence();
      // flush the packed histograms: lane n sums byte (n&3) of word n>>2 over the 64 lanes
      for (int n = lane; n < N; n += 64) {
        const uint32_t* hr = hist + (n >> 2) * ZH;
        const int sh = (n & 3) << 3;
        uint32_t tot = 0;
#pragma unroll
        for (int l4 = 0; l4 < 64; l4 += 4) {
          const u4 v = *(const u4*)(hr + l4);
          tot += ((v.x >> sh) & 0xFFu) + ((v.y >> sh) & 0xFFu) + ((v.z >> sh) & 0xFFu) + ((v.w >> sh) & 0xFFu);
        }
        zkt[n] += tot;
      }
      wave_lds_fence();
      for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
      wave_lds_fence();
    }
    // ---------------- phase 3: ZsumK[:,g] (and Z[:,:,g])
    for (int n = lane; n < N; n += 64) { d.ZsumK[n + (size_t)N * g] = (int32_t)zkt[n]; zkt[n] = 0; }
    if (SAVE_Z) {
      for (int i = lane; i < K * N; i += 64) {            // i = kk + K*n: coalesced Z store
        const int kk = i % K, n = i / K;
        const size_t a = (size_t)n * KP + kk;
        const uint32_t z = zloc[a];
        d.Z[kk + (size_t)K * (n + (size_t)N * g)] = (int32_t)z;
        if (z) { atomicAdd(&zacc[a], z); zloc[a] = 0; }
      }
    }
    wave_lds_fence();
    }
    if (psync) __syncthreads();
  }
  __syncthreads();
  for (int i = tid; i < K * N; i += ZT) {
    const int kk = i % K, n = i / K;
    const uint32_t v = zacc[(size_t)n * KP + kk];
    if (v && !(ablate & 1)) atomicAdd(&d.ZsumG[kk + (size_t)K * n], (int32_t)v);
  }
}

}  // namespace bnmf
