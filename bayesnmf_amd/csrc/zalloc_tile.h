// bayesnmf_amd/csrc/zalloc_tile.h — k_zalloc_tile: the Z-allocation kernel for N > 25 (and for N <= 25 when K is
// too large for the register kernel's per-column list).
//
// Same stream spec and bit-identical results as k_zalloc (kernels.h); different machine mapping.  k_zalloc gives a
// wave a whole column: with N = 50..100 its threshold table plus a private count table is 24-60 KB of LDS per
// wave, i.e. 2-4 waves per CU, and every cell re-reads its row of P from global memory twice.  Here a WORKGROUP owns
// a chunk of ZTR = 32 rows of M for a slice of the columns:
//   * the chunk's rows of P are staged in LDS once per workgroup (fp64, [n][row]) and the chunk's ZsumG counts
//     accumulate in one LDS table shared by the workgroup (flushed once with global integer atomics);
//   * a wave takes one column at a time: a tile of 32 cells, threshold table (N-1) x 32 words;
//     phase 1 (lane = row) builds it from LDS, phase 2 (lane = contiguous quad range) is k_zalloc's search;
//   * ZsumK[:, g] is accumulated across the row chunks with coalesced global integer atomics (zeroed by the host before
//     the launch), Mhat[k, g] is written out and the per-column metric terms are formed by k_colmetrics afterwards in
//     the canonical order (lane l adds rows l, l + 64, ...; wave tree), exactly as the one-wave-per-column kernels do.
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZTR = 32;            // rows per tile
constexpr int ZTP = 33;            // pitch of the [n][row] LDS arrays of a tile
struct ZTGeom { int HW, nch, nslice, slab_words, zacc_words, p_words; };

// host and device agree on the slab layout through these
constexpr int ztile_ae_words(int N) { return 2 * ((N + 1) & ~1); }
inline size_t ztile_slab_words(int N, int HW, bool save_Z) {
  size_t w = (size_t)ztile_ae_words(N) + (size_t)HW * ZH + (size_t)(N - 1) * ZTP + (ZTR + 1) + ZTR + N + (save_Z ? (size_t)N * ZTP : 0);
  return (w + 3) & ~(size_t)3;
}

template <bool SAVE_Z, int ZT>
__global__ __launch_bounds__(ZT) void k_zalloc_tile(ZArgs d, double* __restrict__ Mhat, uint32_t t, ZTGeom zg) {
  constexpr int ZW = ZT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N, HW = zg.HW;
  double* Pc = (double*)smem;                             // [N][ZTR] rows k0.. of P, shared by the workgroup
  uint32_t* zacc = (uint32_t*)(Pc + (size_t)zg.p_words / 2);   // [N][ZTP] ZsumG counts of the chunk
  uint32_t* slab = zacc + zg.zacc_words + (size_t)wave * zg.slab_words;
  double* ae = (double*)slab;                             // [N]  A[n] * E[n,g]
  uint32_t* hist = slab + ztile_ae_words(N);              // [HW][ZH] per-lane packed 8-bit bucket counts (16-B aligned)
  uint32_t* thr = hist + HW * ZH;                         // [N-1][ZTP] thresholds of the tile
  uint32_t* qoff = thr + (size_t)(N - 1) * ZTP;           // [ZTR+1]  quad offset (22 bits) | nlast << 22
  int* mcnt = (int*)(qoff + ZTR + 1);                     // [ZTR]
  uint32_t* zkt = (uint32_t*)(mcnt + ZTR);                // [N] the tile's factor totals
  uint32_t* zloc = zkt + N;                               // [N][ZTP]  (SAVE_Z only)
  const int chunk = blockIdx.x % zg.nch, slice = blockIdx.x / zg.nch;
  const int k0 = chunk * ZTR, kc = min(ZTR, K - k0);
  const int gs0 = (int)((long)G * slice / zg.nslice), gs1 = (int)((long)G * (slice + 1) / zg.nslice);
  for (int i = tid; i < zg.zacc_words; i += ZT) zacc[i] = 0;
  for (int i = tid; i < N * ZTR; i += ZT) { const int cl = i & (ZTR - 1), n = i / ZTR; Pc[i] = cl < kc ? d.P[k0 + cl + (size_t)K * n] : 0.0; }
  for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
  for (int i = lane; i < N; i += 64) zkt[i] = 0;
  if (SAVE_Z) for (int i = lane; i < N * ZTP; i += 64) zloc[i] = 0;
  __syncthreads();
  const int nthr = N - 1;
  uint32_t* ztarget = SAVE_Z ? zloc : zacc;
  // counts and exposures of the NEXT column are requested while this one is processed
  int mpre = 0;
  double epre0 = 0.0, epre1 = 0.0;                         // N <= 128 rides in registers; larger N reloads below
  if (gs0 + wave < gs1) {
    const int g = gs0 + wave;
    if (lane < kc) mpre = d.M[k0 + lane + (size_t)K * g];
    if (lane < N) epre0 = d.E[lane + (size_t)N * g];
    if (64 + lane < N) epre1 = d.E[64 + lane + (size_t)N * g];
  }
  for (int g = gs0 + wave; g < gs1; g += ZW) {
    const int m = mpre;
    if (lane < N) ae[lane] = d.A[lane] * epre0;
    if (64 + lane < N) ae[64 + lane] = d.A[64 + lane] * epre1;
    for (int n = 128 + lane; n < N; n += 64) ae[n] = d.A[n] * d.E[n + (size_t)N * g];
    {
      const int gn = g + ZW;
      if (gn < gs1) {
        if (lane < kc) mpre = d.M[k0 + lane + (size_t)K * gn];
        if (lane < N) epre0 = d.E[lane + (size_t)N * gn];
        if (64 + lane < N) epre1 = d.E[64 + lane + (size_t)N * gn];
      }
    }
    wave_lds_fence();
    // ---------------- phase 1: lane = row of the tile.  Mhat, thresholds, quad counts
    int q = 0, nl = -1;
    if (lane < kc) {
      const double* Pl = Pc + lane;
      double c = 0.0;
      for (int n0 = 0; n0 < N; n0 += 8) {
        double pv[8], av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int n = min(n0 + j, N - 1); pv[j] = Pl[n * ZTR]; av[j] = ae[n]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (n0 + j < N) {
            const double p = pv[j] * av[j];
            c = c + p;
            if (p > 0.0) nl = n0 + j;
          }
        }
      }
      Mhat[k0 + lane + (size_t)K * g] = c;
      if (c > 0.0 && m > 0 && nl >= 0) {
        const double scale = 4294967296.0 / c;
        double cc = 0.0;
        for (int n0 = 0; n0 < nthr; n0 += 8) {
          double pv[8], av[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int n = min(n0 + j, N - 1); pv[j] = Pl[n * ZTR]; av[j] = ae[n]; }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (n0 + j < nthr) {
              cc = cc + pv[j] * av[j];
              const double tt = cc * scale;
              thr[(n0 + j) * ZTP + lane] = (n0 + j >= nl || tt >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)tt;
            }
          }
        }
        q = (m + 3) >> 2;
      }
      mcnt[lane] = q > 0 ? m : 0;
    }
    const int incl = wave_incl_scan_dpp(q);
    if (lane < kc) qoff[lane] = (uint32_t)(incl - q) | ((uint32_t)(nl < 0 ? 0 : nl) << 22);
    const int Q = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) qoff[kc] = (uint32_t)Q;
    wave_lds_fence();
    // ---------------- phase 2: lane takes quads [q0, q1) of the tile, in sub-chunks of <= 63 quads so that the packed
    // 8-bit per-lane histogram cannot overflow (<= 252 counts per flush)
    const int per = (Q + 63) >> 6;
    for (int cbase = 0; cbase < per; cbase += 63) {
      const int q0 = min(Q, lane * per + cbase);
      const int q1 = min(Q, min(lane * per + per, q0 + 63));
      if (q0 < q1) {
        int cell;
        {  // upper_bound(qoff[0..kc] & mask, q0) - 1, branch-free
          int b = 0, len = kc + 1;
          while (len > 1) { const int half = len >> 1; b = ((int)(qoff[b + half - 1] & 0x3FFFFFu) <= q0) ? b + half : b; len -= half; }
          cell = b + ((int)(qoff[b] & 0x3FFFFFu) <= q0 ? 1 : 0) - 1;
        }
        int cstart = (int)(qoff[cell] & 0x3FFFFFu);
        int cend = (int)(qoff[cell + 1] & 0x3FFFFFu);
        int mc = mcnt[cell];
        uint32_t* hl = hist + lane;
        for (int qi = q0; qi < q1; ++qi) {
          if (qi >= cend) {
            do { ++cell; cstart = cend; cend = (int)(qoff[cell + 1] & 0x3FFFFFu); } while (qi >= cend);
            mc = mcnt[cell];
          }
          const int j0 = (qi - cstart) << 2;
          const int nd = mc - j0;                          // >= 1; draws of this quad = min(4, nd)
          const u32x4 w = philox4x32_10((uint32_t)(j0 >> 2), (uint32_t)(k0 + cell) + (uint32_t)K * (uint32_t)g, t, BNMF_V_Z, d.k0, d.k1);
          const uint32_t* col = thr + cell;
          const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
          int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
          if (nthr > 0) {
            int len = nthr;
            while (len > 1) {                             // 4 interleaved branch-free searches
              const int half = len >> 1, off = half - 1;
              const uint32_t t0 = col[(b0 + off) * ZTP], t1 = col[(b1 + off) * ZTP], t2 = col[(b2 + off) * ZTP], t3 = col[(b3 + off) * ZTP];
              b0 = (t0 <= u0) ? b0 + half : b0;
              b1 = (t1 <= u1) ? b1 + half : b1;
              b2 = (t2 <= u2) ? b2 + half : b2;
              b3 = (t3 <= u3) ? b3 + half : b3;
              len -= half;
            }
            b0 += (col[b0 * ZTP] <= u0) ? 1 : 0;
            b1 += (col[b1 * ZTP] <= u1) ? 1 : 0;
            b2 += (col[b2 * ZTP] <= u2) ? 1 : 0;
            b3 += (col[b3 * ZTP] <= u3) ? 1 : 0;
          }
          uint32_t* zc = ztarget + cell;
          atomicAdd(&zc[b0 * ZTP], 1u); atomicAdd(&hl[(b0 >> 2) * ZH], 1u << ((b0 & 3) << 3));
          if (nd > 1) { atomicAdd(&zc[b1 * ZTP], 1u); atomicAdd(&hl[(b1 >> 2) * ZH], 1u << ((b1 & 3) << 3)); }
          if (nd > 2) { atomicAdd(&zc[b2 * ZTP], 1u); atomicAdd(&hl[(b2 >> 2) * ZH], 1u << ((b2 & 3) << 3)); }
          if (nd > 3) { atomicAdd(&zc[b3 * ZTP], 1u); atomicAdd(&hl[(b3 >> 2) * ZH], 1u << ((b3 & 3) << 3)); }
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
    // ---------------- the tile's share of ZsumK[:, g]; Z[k0:k0+kc, :, g]
    for (int n = lane; n < N; n += 64) { const uint32_t v = zkt[n]; if (v) atomicAdd(&d.ZsumK[n + (size_t)N * g], (int32_t)v); zkt[n] = 0; }
    if (SAVE_Z) {
      for (int i = lane; i < kc * N; i += 64) {            // i = cl + kc*n: coalesced Z store
        const int cl = i % kc, n = i / kc;
        const size_t a = (size_t)n * ZTP + cl;
        const uint32_t z = zloc[a];
        d.Z[k0 + cl + (size_t)K * (n + (size_t)N * g)] = (int32_t)z;
        if (z) { atomicAdd(&zacc[a], z); zloc[a] = 0; }
      }
    }
    wave_lds_fence();
  }
  __syncthreads();
  for (int i = tid; i < kc * N; i += ZT) {
    const int cl = i % kc, n = i / kc;
    const uint32_t v = zacc[(size_t)n * ZTP + cl];
    if (v) atomicAdd(&d.ZsumG[k0 + cl + (size_t)K * n], (int32_t)v);
  }
}

// per-column metric terms from Mhat (written by k_zalloc_tile) in the canonical order: wave per column, lane l adds
// rows l, l + 64, ..., then the wave tree (the same operations as the fused form in k_zalloc / k_zalloc_reg)
template <int T>
__global__ __launch_bounds__(T) void k_colmetrics(ZArgs d, const double* __restrict__ Mhat) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = d.K, g = blockIdx.x * (T / 64) + wave;
  if (g >= d.G) return;
  double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
  for (int kk = lane; kk < K; kk += 64) {
    const int m = d.M[kk + (size_t)K * g];
    const double c = Mhat[kk + (size_t)K * g];
    const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
    const double dd = c - (double)m;
    const double mh = c < 1e-6 ? 1e-6 : c;
    const double lmh = dlog(mh);
    const double mt = m < 1 ? 1e-6 : (double)m;
    a_sse = a_sse + dd * dd;
    a_ll = a_ll + (((double)m * lmh - mh) - d.lgfact[mi]);
    a_kl = a_kl + mt * (d.logm[mi] - lmh);
  }
  a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
  if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
}

}  // namespace bnmf
