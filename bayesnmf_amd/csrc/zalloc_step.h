// bayesnmf_amd/csrc/zalloc_step.h — k_zalloc_step: the Z-allocation kernel of the stats mode for 25 <= N <= 100 and any K
// (BASELINE configs 4 and 5: N = 50, K = 96 and N = 100, K = 1,536).  Same stream spec and bit-identical ZsumK / ZsumG /
// metric partials as k_zalloc_tile and k_zalloc (kernels.h); the machine mapping is k_zalloc_sort's (LANE = ITEM, static
// schedule built from M at bnmf_create) carried to the sizes where neither P nor a cell's thresholds fit beside 16 waves:
//
//   * a WORKGROUP owns a set of columns (dealt by total count at bnmf_create) and takes them in BATCHES of <= GBP columns:
//     A[n] E[n, g] of the batch is staged once ([n][column], fp64), ZsumK of the batch accumulates in LDS and leaves with
//     plain stores — E is read once and no column total crosses a workgroup;
//   * inside a batch the rows go by in CHUNKS of 32 (one STEP = chunk x batch, <= 32 x 40 cells): the chunk's rows of P are
//     staged ([n][row], fp64; prefetched into registers during the step before), the chunk's share of ZsumG accumulates in
//     LDS and is flushed with global integer atomics at the end of the step (exact, order-independent);
//   * the cells of a step are ITEMS sorted by their number of quads (a cell above 240 counts: several items), 64 per task;
//     a wave takes a task from an LDS ticket.  Pass A, lane = item: Mhat = sum_n P[k,n] (A[n] E[n,g]) in factor order, the
//     running sum recorded at every 25th factor.  Then the task's cells are taken CPT = 64 / L at a time with L lanes per
//     cell (L = 2 for N <= 50, 4 above: the thresholds of 64 cells do not fit beside 8 waves): pass B, lane = (cell,
//     quarter), continues the SAME sequential sum from the recorded value over its 25 factors and writes the thresholds as
//     a three-level table — per quarter one 128-bit block of the closers of its five 128-bit blocks of four thresholds,
//     the quarter's own closer a pivot in registers — in the cell's column of the wave's slab ([block][cell]: a wave's
//     128-bit reads are conflict-free by construction).  The quad loop, lane q of a cell taking quads q, q + L, ...:
//     Philox block, <= 3 pivot compares, a 128-bit read + 4 compares, a second 128-bit read + 4 compares, one LDS atomic
//     into the cell's packed 8-bit histogram; the histogram is flushed per cell into the step's zG[n][row] / zK[n][column];
//   * Mhat of every cell of the step (zero-count cells are items too) is left in an LDS tile; after the step's barrier
//     each wave adds the metric terms of its columns of the batch to accumulators it keeps IN REGISTERS across the chunks:
//     lane l holds accumulator l of the canonical W = 64 order (row mod 64), a chunk feeds the half-wave its rows belong
//     to, the rows arrive in ascending order — the same additions in the same order as one wave walking the column, without
//     Mhat ever leaving the chip (k_zalloc_tile wrote it out and k_colmetrics read it back: 1.2 GB per launch at config 5).
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZP_KC = 32;            // rows per chunk (= half a wave: a chunk feeds one half of the canonical accumulators)
constexpr int ZP_W = 8;              // waves per workgroup
constexpr int ZP_T = ZP_W * 64;
constexpr int ZP_MAXC = 5;           // metric columns per wave: a batch holds at most ZP_W * ZP_MAXC = 40 columns
constexpr int ZP_QMAX = 60;          // quads per item: 240 counts fit an 8-bit histogram field
constexpr int ZP_NMAX = 100;         // four quarters of 25 factors

struct ZPWg { int batch0, nbatch; };                  // batches [batch0, batch0 + nbatch) of the workgroup
struct ZPBatch { int col0, ncols; };                  // columns cols[col0 .. col0 + ncols)
struct ZPStep { long long item0; int ntask, pad; };   // items [item0, item0 + 64 ntask) of step (batch, chunk)
struct ZPGeom { int nch, nwg; unsigned long long* prof; };   // prof: -DZPPROF builds only (section ticks summed over the waves)
// -DZPPROF: section timers.  [0] staging + end of step, [1] pass A, [2] pass B, [3] quad loops, [4] histogram flush, [5] waiting at the
// step's barrier, [6] whole kernel, [7] waves.  Never defined in the product build.
#ifdef ZPPROF
#define ZPTIC(i) const uint64_t zptic_##i = __builtin_amdgcn_s_memtime()
#define ZPTOC(i) zpprof[i] += __builtin_amdgcn_s_memtime() - zptic_##i
#else
#define ZPTIC(i)
#define ZPTOC(i)
#endif
struct ZPArgs {
  ZArgs a;
  const uint32_t* items;             // row in chunk | column in batch << 5 | fragment << 11; 0xFFFFFFFF = empty lane
  const ZPWg* wgs;
  const ZPBatch* batches;
  const ZPStep* steps;               // [batch][chunk]
  const int* cols;
};
// host and device agree on the LDS layout through these.  NP = 25 NS rows of P / A E (zero beyond N: adding +0.0 is exact)
BNMF_HD size_t zstep_shared_bytes(int NS, int N, int GBP) {
  const size_t NP = 25 * (size_t)NS;
  size_t b = NP * ZP_KC * 8 + NP * GBP * 8 + (size_t)GBP * ZP_KC * 8;       // Pc, ae, mt (fp64)
  b += (size_t)N * ZP_KC * 4 + (size_t)N * GBP * 4;                          // zG, zK
  b += 2 * (size_t)GBP * ZP_KC * 4 + 2 * (size_t)GBP * 4 + 16;              // Ms x 2, colid x 2, ticket
  return (b + 15) & ~(size_t)15;
}
BNMF_HD size_t zstep_wave_bytes(int NS, int L, int N) { return (size_t)(64 / L) * (96 * (size_t)NS + 4 * (size_t)((N + 3) / 4)); }

typedef uint32_t __attribute__((ext_vector_type(4))) zp_uv4;
typedef __attribute__((address_space(3))) zp_uv4 lds_uv4;
BNMF_DEV u4 lds_ld4(uint32_t off) { const zp_uv4 v = *(lds_uv4*)(uintptr_t)off; return u4{v.x, v.y, v.z, v.w}; }   // ds_read_b128 by byte offset
BNMF_DEV uint32_t le4(const u4& v, uint32_t u) { return (v.x <= u ? 1u : 0u) + (v.y <= u ? 1u : 0u) + (v.z <= u ? 1u : 0u) + (v.w <= u ? 1u : 0u); }
BNMF_DEV double shfl_d(double v, int src) {
  const int lo = __shfl((int)__double_as_longlong(v), src, 64), hi = __shfl((int)(__double_as_longlong(v) >> 32), src, 64);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int NS /* quarters of 25 factors: N <= 25 NS */, int L /* lanes per cell */, int GBP /* column pitch: batch capacity */>
__global__ __launch_bounds__(ZP_T, 2) void k_zalloc_step(ZPArgs s, uint32_t t, ZPGeom zg) {
  static_assert(L == 1 || L == 2 || L == 4, "lanes per cell");
  static_assert(NS <= L || L == 1, "a quarter per lane of a cell");
  constexpr int CPT = 64 / L;                              // cells per sub-task
  constexpr int NPV = NS - 1;                              // pivots in registers: the closers of quarters 0 .. NS - 2
  constexpr int NP = 25 * NS;
  constexpr uint32_t ROW = CPT * 16;                       // bytes of one [block] row of the slab
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ZArgs& d = s.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, N = d.N;
  const int HWD = (N + 3) >> 2;
  double* Pc = (double*)smem;                              // [NP][32] rows of the chunk, zero beyond N / the last row
  double* ae = Pc + NP * ZP_KC;                            // [NP][GBP] A[n] E[n, column], zero beyond N
  double* mt = ae + NP * GBP;                              // [GBP][32] Mhat of the step's cells
  uint32_t* zG = (uint32_t*)(mt + GBP * ZP_KC);            // [N][32] the step's share of ZsumG
  uint32_t* zK = zG + (size_t)N * ZP_KC;                   // [N][GBP] ZsumK of the batch
  int* Ms = (int*)(zK + (size_t)N * GBP);                  // [2][GBP][32] counts of the step's cells (the next step's are staged beside)
  int* colid = Ms + 2 * GBP * ZP_KC;                       // [2][GBP]
  uint32_t* ticket = (uint32_t*)(colid + 2 * GBP);
  unsigned char* wbase = smem + zstep_shared_bytes(NS, N, GBP) + (size_t)wave * zstep_wave_bytes(NS, L, N);
  u4* tb = (u4*)wbase;                                     // [6 NS][CPT]: per quarter the closers' block, then its five blocks
  uint32_t* hist = (uint32_t*)(tb + 6 * NS * CPT);         // [HWD][CPT] packed 8-bit bucket counts of the sub-task's cells
  const ZPWg wg = s.wgs[blockIdx.x];
  const int j = lane & (CPT - 1), q = lane / CPT;          // cell of the sub-task, quarter
#ifdef ZPPROF
  uint64_t zpprof[8] = {0, 0, 0, 0, 0, 0, 0, 1};
#endif
  ZPTIC(6);

  // ---------------- staging
  constexpr int PRE_P = (NP * ZP_KC + ZP_T - 1) / ZP_T, PRE_M = (GBP * ZP_KC + ZP_T - 1) / ZP_T;
  double preP[PRE_P]; int preM[PRE_M];
  auto prefetch = [&](const ZPBatch& b, int ch) {          // the chunk's rows of P and the step's counts, into registers
    const int k0 = ch * ZP_KC, kc = min(ZP_KC, K - k0);
#pragma unroll
    for (int r = 0; r < PRE_P; ++r) {
      const int i = tid + r * ZP_T, n = i >> 5, kl = i & 31;
      preP[r] = (i < NP * ZP_KC && n < N && kl < kc) ? d.P[k0 + kl + (size_t)K * n] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < PRE_M; ++r) {
      const int i = tid + r * ZP_T, gl = i >> 5, kl = i & 31;
      preM[r] = (gl < b.ncols && kl < kc) ? d.M[k0 + kl + (size_t)K * s.cols[b.col0 + gl]] : 0;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int r = 0; r < PRE_P; ++r) { const int i = tid + r * ZP_T; if (i < NP * ZP_KC) Pc[i] = preP[r]; }
    int* Md = Ms + buf * GBP * ZP_KC;
#pragma unroll
    for (int r = 0; r < PRE_M; ++r) { const int i = tid + r * ZP_T; if (i < GBP * ZP_KC) Md[i] = preM[r]; }
  };
  auto stage_batch = [&](const ZPBatch& b, int par) {
    for (int i = tid; i < N * b.ncols; i += ZP_T) {
      const int gl = i / N, n = i - gl * N;
      ae[n * GBP + gl] = d.A[n] * d.E[n + (size_t)N * s.cols[b.col0 + gl]];
    }
    for (int i = tid; i < b.ncols; i += ZP_T) colid[par * GBP + i] = s.cols[b.col0 + i];
  };
  for (int i = tid; i < NP * GBP; i += ZP_T) ae[i] = 0.0;
  for (int i = tid; i < N * ZP_KC; i += ZP_T) zG[i] = 0;
  for (int i = tid; i < N * GBP; i += ZP_T) zK[i] = 0;
  for (int i = lane; i < HWD * CPT; i += 64) hist[i] = 0;
  if (tid == 0) *ticket = 0;
  __syncthreads();
  if (wg.nbatch > 0) {
    const ZPBatch b0 = s.batches[wg.batch0];
    stage_batch(b0, 0);
    prefetch(b0, 0);
    commit(0);
  }
  __syncthreads();

  auto next_task = [&]() -> int {
    int tk = 0;
    if (lane == 0) tk = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readlane(tk, 0);
  };
  const uint32_t tbj = lds_off(tb) + (uint32_t)j * 16u;     // the cell's column of the slab
  const uint32_t hbj = lds_off(hist) + (uint32_t)j * 4u;
  int stepno = 0;
  for (int bi = 0; bi < wg.nbatch; ++bi) {
    const ZPBatch bt = s.batches[wg.batch0 + bi];
    const int* cid = colid + (bi & 1) * GBP;
    double acc[ZP_MAXC][3];
#pragma unroll
    for (int cs = 0; cs < ZP_MAXC; ++cs) { acc[cs][0] = 0.0; acc[cs][1] = 0.0; acc[cs][2] = 0.0; }
    for (int ch = 0; ch < zg.nch; ++ch, ++stepno) {
      const ZPStep st = s.steps[(size_t)(wg.batch0 + bi) * zg.nch + ch];
      const int k0 = ch * ZP_KC, kc = min(ZP_KC, K - k0);
      const int* Msb = Ms + (stepno & 1) * GBP * ZP_KC;
      // the next step's rows of P and counts: requested now, written to LDS after this step's tasks
      const bool last_ch = ch + 1 == zg.nch, last_step = last_ch && bi + 1 == wg.nbatch;
      ZPBatch bn = bt;
      if (last_ch && !last_step) bn = s.batches[wg.batch0 + bi + 1];
      { ZPTIC(0); if (!last_step) prefetch(bn, last_ch ? 0 : ch + 1); ZPTOC(0); }
      for (int tk = next_task(); tk < st.ntask; tk = next_task()) {
        // ---------------- pass A: lane = item
        ZPTIC(1);
        const uint32_t it = s.items[(size_t)st.item0 + (size_t)tk * 64 + lane];
        const bool valid = it != 0xFFFFFFFFu;
        const int kl = valid ? (int)(it & 31u) : 0, gl = valid ? (int)((it >> 5) & 63u) : 0;
        const int frag = valid ? (int)(it >> 11) : 0;
        const int m = Msb[gl * ZP_KC + kl];
        double c = 0.0, ps[NS];
        {
          const double* Pk = Pc + kl;
          const double* ag = ae + gl;
#pragma unroll
          for (int qq = 0; qq < NS; ++qq) {
            ps[qq] = c;
#pragma unroll
            for (int i = 0; i < 25; ++i) c = c + Pk[(25 * qq + i) * ZP_KC] * ag[(25 * qq + i) * GBP];
          }
        }
        if (valid && frag == 0) mt[gl * ZP_KC + kl] = c;
        int nq = 0, npad = 0;
        const int q0 = frag * ZP_QMAX;
        if (valid && c > 0.0 && m > 0) {
          const int qt = (m + 3) >> 2;
          nq = min(ZP_QMAX, qt - q0);
          npad = (q0 + nq == qt) ? ((4 - (m & 3)) & 3) : 0;
        }
        ZPTOC(1);
        // ---------------- the task's cells, CPT at a time, L lanes per cell
#pragma unroll 1
        for (int sub = 0; sub < L; ++sub) {
          const int src = sub * CPT + j;
          const int nqS = L == 1 ? nq : __shfl(nq, src, 64);
          if (__builtin_amdgcn_ballot_w64(nqS > 0) == 0) continue;        // nothing to allocate (sorted: the tail of the step)
          ZPTIC(2);
          const uint32_t itS = L == 1 ? it : (uint32_t)__shfl((int)it, src, 64);
          const int npadS = L == 1 ? npad : __shfl(npad, src, 64);
          const double cS = L == 1 ? c : shfl_d(c, src);
          double psS = 0.0;
#pragma unroll
          for (int qq = 1; qq < NS; ++qq) { const double v = L == 1 ? ps[qq] : shfl_d(ps[qq], src); if (q == qq) psS = v; }
          const int klS = (int)(itS & 31u), glS = (int)((itS >> 5) & 63u), q0S = (int)(itS >> 11) * ZP_QMAX;
          // pass B: the quarter's 25 thresholds, continuing the sequential sum from the value pass A recorded
          uint32_t pvme = 0xFFFFFFFFu;
          if (q < NS) {
            const double scale = 4294967296.0 / cS;
            const double* Pk = Pc + (25 * q) * ZP_KC + klS;
            const double* ag = ae + (25 * q) * GBP + glS;
            double cc = psS;
            uint32_t l2[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
              uint32_t tv[5];
#pragma unroll
              for (int i = 0; i < 5; ++i) { cc = cc + Pk[(5 * b + i) * ZP_KC] * ag[(5 * b + i) * GBP]; tv[i] = cvt_u32_sat(cc * scale); }
              tb[(q * 6 + 1 + b) * CPT + j] = u4{tv[0], tv[1], tv[2], tv[3]};
              l2[b] = tv[4];
            }
            tb[(q * 6) * CPT + j] = u4{l2[0], l2[1], l2[2], l2[3]};
            pvme = l2[4];
          }
          uint32_t pv[NPV > 0 ? NPV : 1];
#pragma unroll
          for (int p = 0; p < NPV; ++p) pv[p] = (uint32_t)__shfl((int)pvme, p * CPT + j, 64);
          const uint32_t celem = (uint32_t)(k0 + klS) + (uint32_t)K * (uint32_t)cid[glS];
          auto quad = [&](int qidx, uint32_t inc0, uint32_t inc1, uint32_t inc2, uint32_t inc3) {
            const u32x4 w = philox4x32_7((uint32_t)qidx, celem, t, BNMF_V_Z, d.k0, d.k1);
            const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int p = 0; p < NPV; ++p) cmp_acc4(pv[p], u0, u1, u2, u3, a0, a1, a2, a3);
            const uint32_t A0 = mad24(a0, 6 * ROW, tbj), A1 = mad24(a1, 6 * ROW, tbj), A2 = mad24(a2, 6 * ROW, tbj), A3 = mad24(a3, 6 * ROW, tbj);
            const u4 x0 = lds_ld4(A0), x1 = lds_ld4(A1), x2 = lds_ld4(A2), x3 = lds_ld4(A3);
            const uint32_t s0 = le4(x0, u0), s1 = le4(x1, u1), s2 = le4(x2, u2), s3 = le4(x3, u3);
            const u4 y0 = lds_ld4(mad24(s0, ROW, A0 + ROW)), y1 = lds_ld4(mad24(s1, ROW, A1 + ROW)), y2 = lds_ld4(mad24(s2, ROW, A2 + ROW)), y3 = lds_ld4(mad24(s3, ROW, A3 + ROW));
            const uint32_t b0 = mad24(a0, 25, mad24(s0, 5, le4(y0, u0))), b1 = mad24(a1, 25, mad24(s1, 5, le4(y1, u1)));
            const uint32_t b2 = mad24(a2, 25, mad24(s2, 5, le4(y2, u2))), b3 = mad24(a3, 25, mad24(s3, 5, le4(y3, u3)));
            lds_add(mad24(b0 >> 2, CPT * 4, hbj), inc0 << ((b0 & 3u) << 3));
            lds_add(mad24(b1 >> 2, CPT * 4, hbj), inc1 << ((b1 & 3u) << 3));
            lds_add(mad24(b2 >> 2, CPT * 4, hbj), inc2 << ((b2 & 3u) << 3));
            lds_add(mad24(b3 >> 2, CPT * 4, hbj), inc3 << ((b3 & 3u) << 3));
          };
          wave_lds_fence();
          ZPTOC(2);
          ZPTIC(3);
          // lane q of the cell takes quads q, q + L, ...; the cell's last quad is the only one that can hold pads
          for (int i = q; __builtin_amdgcn_ballot_w64(i < nqS - 1) != 0; i += L)
            if (i < nqS - 1) quad(q0S + i, 1u, 1u, 1u, 1u);
          if (nqS > 0 && ((nqS - 1) % L) == q) quad(q0S + nqS - 1, 1u, npadS > 2 ? 0u : 1u, npadS > 1 ? 0u : 1u, npadS > 0 ? 0u : 1u);
          wave_lds_fence();
          ZPTOC(3);
          ZPTIC(4);
          // flush the cell's histogram into the step's tables (the cell's L lanes share its words)
          if (nqS > 0) {
            const uint32_t zgb = lds_off(zG) + ((uint32_t)klS << 2), zkb = lds_off(zK) + ((uint32_t)glS << 2);
            for (int w = q; w < HWD; w += L) {
              const uint32_t v = hist[w * CPT + j];
              if (v) {
                hist[w * CPT + j] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                  const uint32_t cnt = (v >> (8 * b)) & 0xFFu;
                  if (cnt) { lds_add(zgb + (uint32_t)(4 * w + b) * (ZP_KC * 4u), cnt); lds_add(zkb + (uint32_t)(4 * w + b) * (GBP * 4u), cnt); }
                }
              }
            }
          }
          wave_lds_fence();
          ZPTOC(4);
        }
      }
      { ZPTIC(5); __syncthreads(); ZPTOC(5); }
      ZPTIC(0);
      // ---------------- end of the step: the chunk's share of ZsumG, the metric terms of the chunk's rows, the next step's staging
      for (int i = tid; i < N * ZP_KC; i += ZP_T) {
        const uint32_t v = zG[i];
        if (v) { const int kl = i & 31, n = i >> 5; atomicAdd(&d.ZsumG[k0 + kl + (size_t)K * n], (int32_t)v); zG[i] = 0; }
      }
      {
        const int kl = lane & 31;
        const bool mine = (lane >> 5) == (ch & 1) && kl < kc;               // canonical accumulator of row k0 + kl: (k0 + kl) mod 64 = lane
#pragma unroll
        for (int cs = 0; cs < ZP_MAXC; ++cs) {
          const int gl = cs * ZP_W + wave;
          if (gl < bt.ncols && mine) {
            const int m = Msb[gl * ZP_KC + kl];
            const double cv = mt[gl * ZP_KC + kl];
            const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
            const double dd = cv - (double)m;
            const double mh = cv < 1e-6 ? 1e-6 : cv;
            const double lmh = dlog(mh);
            const double mtt = m < 1 ? 1e-6 : (double)m;
            acc[cs][0] = acc[cs][0] + dd * dd;
            acc[cs][1] = acc[cs][1] + (((double)m * lmh - mh) - d.lgfact[mi]);
            acc[cs][2] = acc[cs][2] + mtt * (d.logm[mi] - lmh);
          }
        }
      }
      if (last_ch) {
        // end of the batch: the columns' metric terms (wave tree over the 64 accumulators) and ZsumK
#pragma unroll
        for (int cs = 0; cs < ZP_MAXC; ++cs) {
          const int gl = cs * ZP_W + wave;
          if (gl < bt.ncols) {                                              // wave-uniform
            const double r0 = wave_tree64(acc[cs][0]), r1 = wave_tree64(acc[cs][1]), r2 = wave_tree64(acc[cs][2]);
            if (lane == 0) { const int g = cid[gl]; d.colsse[g] = r0; d.colll[g] = r1; d.colkl[g] = r2; }
          }
        }
        for (int i = tid; i < N * bt.ncols; i += ZP_T) {
          const int gl = i / N, n = i - gl * N;
          d.ZsumK[n + (size_t)N * cid[gl]] = (int32_t)zK[n * GBP + gl];
          zK[n * GBP + gl] = 0;
        }
        if (!last_step) stage_batch(bn, (bi + 1) & 1);
      }
      if (!last_step) commit((stepno + 1) & 1);
      if (tid == 0) *ticket = 0;
      __syncthreads();
      ZPTOC(0);
    }
  }
  ZPTOC(6);
#ifdef ZPPROF
  if (zg.prof && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&zg.prof[i], (unsigned long long)zpprof[i]);
#endif
}

}  // namespace bnmf
