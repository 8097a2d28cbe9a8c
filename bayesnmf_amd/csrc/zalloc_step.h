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
//   * only the INCLUDED factors (A[n] != 0) are staged and walked: an excluded factor adds +0.0 to the running sum and its
//     threshold repeats the one before it, so no count can land on it and every other count lands where it would have — with
//     the rank being learned (config 4) most of the N factors are excluded most of the time;
//   * the cells of a step are ITEMS (a cell above 240 counts: several items) sorted by their number of quads and dealt to
//     the 8 waves in snake order at bnmf_create: every wave has the same number of items of the same sizes, 64 per task —
//     no ticket, and the step's barrier finds the waves level.  Pass A, lane = item: Mhat = sum_n P[k,n] (A[n] E[n,g]) in
//     factor order (operands of the next five factors in flight while five are added), the running sum recorded where a
//     lane's range of factors starts.  Then the task's cells are taken CPT = 64 / L at a time with L lanes per cell (L = 2
//     for N <= 50, 4 above: the thresholds of 64 cells do not fit beside 8 waves): pass B, lane = (cell, range), continues
//     the SAME sequential sum from the recorded value over its range of <= 5 blocks of 5 factors and writes the thresholds
//     as a three-level table — per range one 128-bit block of the closers of its blocks, then the blocks' first four
//     thresholds as 128-bit blocks, the range's own closer a pivot in registers — in the cell's column of the wave's slab
//     ([block][cell]: a wave's 128-bit reads are conflict-free by construction).  The quad loop, lane q of a cell taking
//     quads q, q + L, ...: Philox block, <= 3 pivot compares, a 128-bit read + 4 compares, a second 128-bit read + 4
//     compares, one LDS atomic into the cell's packed 8-bit histogram; the words a lane touched are flushed per cell into
//     the step's zG[n][row] / zK[n][column] (an LDS exchange hands each word to one of the cell's lanes);
//   * Mhat of every cell of the step (zero-count cells are items too) is left in an LDS tile; after the step's barrier
//     each wave adds the metric terms of its columns of the batch to accumulators it keeps IN REGISTERS across the chunks:
//     lane l holds accumulator l of the canonical W = 64 order (row mod 64), a chunk feeds the half-wave its rows belong
//     to, the rows arrive in ascending order — the same additions in the same order as one wave walking the column, without
//     Mhat ever leaving the chip (k_zalloc_tile wrote it out and k_colmetrics read it back: 1.2 GB per launch at config 5).
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZP_KC = 32;            // rows per chunk (= half a wave: a chunk feeds one half of the canonical accumulators)
constexpr int ZP_WMAX = 12;          // most waves per workgroup (api.hip launches 8: two per SIMD)
constexpr int ZP_QMAX = 60;          // quads per item: 240 counts fit an 8-bit histogram field
constexpr int ZP_NMAX = 100;         // four lanes per cell, five blocks of five factors each

struct ZPWg { int batch0, nbatch; };                  // batches [batch0, batch0 + nbatch) of the workgroup
struct ZPBatch { int col0, ncols; };                  // columns cols[col0 .. col0 + ncols)
struct ZPStep { long long item0; int ntw, pad; };     // step (batch, chunk): wave w, task i reads items item0 + 64 (w ntw + i) + lane
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
  int it16;                          // the same fields as uint16 (no cell above 31 fragments = 7,440 counts): half the schedule's bytes per launch
  const ZPWg* wgs;
  const ZPBatch* batches;
  const ZPStep* steps;               // [batch][chunk]
  const int* cols;
};
// host and device agree on the LDS layout through these.  NP rows of P / A E: N rounded up to whole blocks of five (zero beyond the
// included factors)
BNMF_HD int zstep_rows(int N) { return 5 * ((N + 4) / 5); }
BNMF_HD size_t zstep_shared_bytes(int N, int GBP) {
  const size_t NP = (size_t)zstep_rows(N);
  size_t b = NP * ZP_KC * 8 + NP * GBP * 8 + (size_t)GBP * ZP_KC * 8;       // Pc, ae, mt (fp64)
  b += (size_t)N * (ZP_KC + 1) * 4 + (size_t)N * (GBP + 1) * 4;              // zG, zK (odd pitches: the lanes of a cell add to one column)
  b += 2 * (size_t)GBP * ZP_KC * 4 + 2 * (size_t)GBP * 4;                   // Ms x 2, colid x 2
  b += 2 * NP * 4 + 16;                                                      // act, inv, the number of included factors
  return (b + 15) & ~(size_t)15;
}
BNMF_HD size_t zstep_wave_bytes(int L) { return (size_t)(64 / L) * 96 * L; }   // the threshold tables of 64 / L cells

typedef uint32_t __attribute__((ext_vector_type(4))) zp_uv4;
typedef __attribute__((address_space(3))) zp_uv4 lds_uv4;
BNMF_DEV u4 lds_ld4(uint32_t off) { const zp_uv4 v = *(lds_uv4*)(uintptr_t)off; return u4{v.x, v.y, v.z, v.w}; }   // ds_read_b128 by byte offset
// One ds_read_b64 per operand, never merged: a wave's 64 reads of one row of Pc / ae (256 B) are conflict-free whatever rows and
// columns the lanes hold, at 256 B/clk; merged into ds_read2_b64 they go 16 lanes at a time over 32 banks (128 B/clk, and lanes
// 16 rows apart collide).  The asm statements keep the loads of the next five factors in front of the sums of the current ones.
typedef __attribute__((address_space(3))) volatile double lds_vf64;
BNMF_DEV double lds_ld8(uint32_t off) { return *(lds_vf64*)(uintptr_t)off; }
#define ZP_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
BNMF_DEV uint32_t le4(const u4& v, uint32_t u) { return (v.x <= u ? 1u : 0u) + (v.y <= u ? 1u : 0u) + (v.z <= u ? 1u : 0u) + (v.w <= u ? 1u : 0u); }
BNMF_DEV double shfl_d(double v, int src) {
  const int lo = __shfl((int)__double_as_longlong(v), src, 64), hi = __shfl((int)(__double_as_longlong(v) >> 32), src, 64);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int L /* lanes per cell: N <= 25 L */, int GBP /* column pitch: batch capacity */, int ZP_W /* waves */>   // api.hip uses L = 4 throughout
__global__ __launch_bounds__(ZP_W * 64, ZP_W / 4) void k_zalloc_step(ZPArgs s, uint32_t t, ZPGeom zg) {
  constexpr int ZP_T = ZP_W * 64;
  constexpr int ZP_MAXC = (GBP + ZP_W - 1) / ZP_W;         // metric columns per wave
  static_assert(L == 2 || L == 4, "lanes per cell");
  constexpr int CPT = 64 / L;                              // cells per sub-task
  constexpr int NPV = L - 1;                               // pivots in registers: the closers of ranges 0 .. L - 2
  constexpr int NPMAX = 25 * L;                            // most rows of the staged operands this instantiation serves
  constexpr uint32_t ROW = CPT * 16;                       // bytes of one [block] row of the slab
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ZArgs& d = s.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, N = d.N;
  const int NP = zstep_rows(N), NBMAX = NP / 5;            // rows of the staged operands; blocks of five factors
  double* Pc = (double*)smem;                              // [NP][32] rows of the chunk, included factors only, zero beyond
  double* ae = Pc + NP * ZP_KC;                            // [NP][GBP] A[n] E[n, column], included factors only, zero beyond
  double* mt = ae + NP * GBP;                              // [GBP][32] Mhat of the step's cells
  constexpr int GP = ZP_KC + 1, KQ = GBP + 1;              // odd pitches of zG and zK: the L lanes of a cell add to the same column of
                                                           // different rows — with an even pitch, to the same bank
  uint32_t* zG = (uint32_t*)(mt + GBP * ZP_KC);            // [N][GP] the step's share of ZsumG (rows: included factors)
  uint32_t* zK = zG + (size_t)N * GP;                      // [N][KQ] ZsumK of the batch (rows: included factors)
  int* Ms = (int*)(zK + (size_t)N * KQ);                   // [2][GBP][32] counts of the step's cells (the next step's are staged beside)
  int* colid = Ms + 2 * GBP * ZP_KC;                       // [2][GBP]
  int* act = colid + 2 * GBP;                              // [NP] the included factors in order
  int* inv = act + NP;                                     // [NP] factor -> its place among the included ones, or -1
  int* misc = inv + NP;                                    // [0] number of included factors
  unsigned char* wbase = smem + zstep_shared_bytes(N, GBP) + (size_t)wave * zstep_wave_bytes(L);
  u4* tb = (u4*)wbase;                                     // [6 L][CPT]: per range the closers' block, then its five blocks
  const ZPWg wg = s.wgs[blockIdx.x];
  const int j = lane & (CPT - 1), q = lane / CPT;          // cell of the sub-task, range of factors
#ifdef ZPPROF
  uint64_t zpprof[8] = {0, 0, 0, 0, 0, 0, 0, 1};
#endif
  ZPTIC(6);

  // ---------------- the included factors (R/sample_params.R:257: probs[n] = P[k,n] A[n] E[n,g])
  if (wave == 0) {
    int cnt = 0;
    for (int base = 0; base < N; base += 64) {
      const int n = base + lane;
      const bool on = n < N && d.A[n] != 0.0;
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(on);
      const int pos = cnt + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
      if (on) act[pos] = n;
      if (n < N) inv[n] = on ? pos : -1;
      cnt += __builtin_popcountll(mk);
    }
    if (lane == 0) misc[0] = cnt;
  }
  for (int i = tid; i < NP * GBP; i += ZP_T) ae[i] = 0.0;
  for (int i = tid; i < N * GP; i += ZP_T) zG[i] = 0;
  for (int i = tid; i < N * KQ; i += ZP_T) zK[i] = 0;
  __syncthreads();
  const int Na = __builtin_amdgcn_readfirstlane(misc[0]);
  const int nblk = (Na + 4) / 5;                           // blocks of five factors
  const int BPL = nblk > L ? (nblk + L - 1) / L : 1;       // blocks per lane of a cell (<= 5)

  // ---------------- staging
  constexpr int PRE_P = (NPMAX * ZP_KC + ZP_T - 1) / ZP_T, PRE_M = (GBP * ZP_KC + ZP_T - 1) / ZP_T;
  double preP[PRE_P]; int preM[PRE_M];
  auto prefetch = [&](const ZPBatch& b, int ch) {          // the chunk's rows of P and the step's counts, into registers
    const int k0 = ch * ZP_KC, kc = min(ZP_KC, K - k0);
#pragma unroll
    for (int r = 0; r < PRE_P; ++r) {
      const int i = tid + r * ZP_T, n = i >> 5, kl = i & 31;
      preP[r] = (n < Na && kl < kc) ? d.P[k0 + kl + (size_t)K * act[n]] : 0.0;
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
    if (Na > 0)
      for (int i = tid; i < Na * b.ncols; i += ZP_T) {
        const int gl = i / Na, ii = i - gl * Na, n = act[ii];
        ae[ii * GBP + gl] = d.A[n] * d.E[n + (size_t)N * s.cols[b.col0 + gl]];
      }
    for (int i = tid; i < b.ncols; i += ZP_T) colid[par * GBP + i] = s.cols[b.col0 + i];
  };
  if (wg.nbatch > 0) {
    const ZPBatch b0 = s.batches[wg.batch0];
    stage_batch(b0, 0);
    prefetch(b0, 0);
    commit(0);
  }
  __syncthreads();

  const uint32_t tbj = lds_off(tb) + (uint32_t)j * 16u;     // the cell's column of the slab
  const uint32_t bstride = 5u * (uint32_t)BPL;              // factors per lane range
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
      const bool last_ch = ch + 1 == zg.nch, last_step = last_ch && bi + 1 == wg.nbatch;
      ZPBatch bn = bt;
      if (last_ch && !last_step) bn = s.batches[wg.batch0 + bi + 1];
      for (int tk = 0; tk < st.ntw; ++tk) {
        // ---------------- pass A: lane = item
        ZPTIC(1);
        uint32_t it;
        {
          const size_t ix = (size_t)st.item0 + ((size_t)wave * st.ntw + tk) * 64 + lane;
          if (s.it16) { const uint32_t r = ((const uint16_t*)s.items)[ix]; it = r == 0xFFFFu ? 0xFFFFFFFFu : r; }   // wave-uniform
          else it = s.items[ix];
        }
        const bool valid = it != 0xFFFFFFFFu;
        if (__builtin_amdgcn_ballot_w64(valid) == 0) break;               // the wave's list is sorted: nothing behind an empty task
        const int kl = valid ? (int)(it & 31u) : 0, gl = valid ? (int)((it >> 5) & 63u) : 0;
        const int frag = valid ? (int)(it >> 11) : 0;
        const int m = Msb[gl * ZP_KC + kl];
        double c = 0.0, ps[L];
#pragma unroll
        for (int qq = 0; qq < L; ++qq) ps[qq] = 0.0;
        {
          const uint32_t Pk = lds_off(Pc) + ((uint32_t)kl << 3), ag = lds_off(ae) + ((uint32_t)gl << 3);
          double pa[5], aa[5], pb[5], ab[5];
          auto ld = [&](int blk, double* pv, double* av) {
            const uint32_t r0 = 5u * (uint32_t)min(blk, NBMAX - 1);
            const uint32_t po = mad24(r0, ZP_KC * 8, Pk), ao = mad24(r0, GBP * 8, ag);
#pragma unroll
            for (int i = 0; i < 5; ++i) { pv[i] = lds_ld8(po + i * ZP_KC * 8); av[i] = lds_ld8(ao + i * GBP * 8); }
          };
          auto ac = [&](int blk, const double* pv, const double* av) {
#pragma unroll
            for (int qq = 1; qq < L; ++qq) if (blk == qq * BPL) ps[qq] = c;    // where range qq starts (wave-uniform)
#pragma unroll
            for (int i = 0; i < 5; ++i) c = c + pv[i] * av[i];
          };
          ld(0, pa, aa);
          for (int blk = 0; blk < nblk; blk += 2) {
            ld(blk + 1, pb, ab);
            ZP_ORDER();
            ac(blk, pa, aa);
            ld(blk + 2, pa, aa);
            ZP_ORDER();
            if (blk + 1 < nblk) ac(blk + 1, pb, ab);
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
          const int nqS = __shfl(nq, src, 64);
          if (__builtin_amdgcn_ballot_w64(nqS > 0) == 0) continue;        // nothing to allocate (sorted: the tail of the list)
          ZPTIC(2);
          const uint32_t itS = (uint32_t)__shfl((int)it, src, 64);
          const int npadS = __shfl(npad, src, 64);
          const double cS = shfl_d(c, src);
          double psS = 0.0;
#pragma unroll
          for (int qq = 1; qq < L; ++qq) { const double v = shfl_d(ps[qq], src); if (q == qq) psS = v; }
          const int klS = (int)(itS & 31u), glS = (int)((itS >> 5) & 63u), q0S = (int)(itS >> 11) * ZP_QMAX;
          // pass B: the thresholds of the lane's range of factors, continuing the sequential sum from the value pass A recorded
          uint32_t l2[5] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
          {
            const double scale = 4294967296.0 / cS;
            const int gb0 = q * BPL;
            const uint32_t Pk = lds_off(Pc) + ((uint32_t)klS << 3), ag = lds_off(ae) + ((uint32_t)glS << 3);
            double cc = psS;
            double pa[5], aa[5], pb[5], ab[5];
            auto ld = [&](int b, double* pv, double* av) {
              const uint32_t r0 = 5u * (uint32_t)min(gb0 + b, NBMAX - 1);
              const uint32_t po = mad24(r0, ZP_KC * 8, Pk), ao = mad24(r0, GBP * 8, ag);
#pragma unroll
              for (int i = 0; i < 5; ++i) { pv[i] = lds_ld8(po + i * ZP_KC * 8); av[i] = lds_ld8(ao + i * GBP * 8); }
            };
            auto put = [&](int b, const double* pv, const double* av, uint32_t& closer) {
              const bool ok = gb0 + b < nblk;                               // beyond the included factors: "never"
              uint32_t tv[5];
#pragma unroll
              for (int i = 0; i < 5; ++i) { cc = cc + pv[i] * av[i]; tv[i] = ok ? cvt_u32_sat(cc * scale) : 0xFFFFFFFFu; }
              tb[(q * 6 + 1 + b) * CPT + j] = u4{tv[0], tv[1], tv[2], tv[3]};
              closer = tv[4];
            };
            ld(0, pa, aa);
            if (BPL > 1) ld(1, pb, ab);
            ZP_ORDER();
            put(0, pa, aa, l2[0]);
            if (BPL > 1) { if (BPL > 2) ld(2, pa, aa); ZP_ORDER(); put(1, pb, ab, l2[1]); }
            if (BPL > 2) { if (BPL > 3) ld(3, pb, ab); ZP_ORDER(); put(2, pa, aa, l2[2]); }
            if (BPL > 3) { if (BPL > 4) ld(4, pa, aa); ZP_ORDER(); put(3, pb, ab, l2[3]); }
            if (BPL > 4) put(4, pa, aa, l2[4]);
          }
          // the closers' block (entries at and beyond the range's own closer never decide: that one is a pivot, the rest "never")
          tb[(q * 6) * CPT + j] = u4{l2[0], l2[1], l2[2], l2[3]};
          const uint32_t pvme = BPL == 1 ? l2[0] : BPL == 2 ? l2[1] : BPL == 3 ? l2[2] : BPL == 4 ? l2[3] : l2[4];
          uint32_t pv[NPV];
#pragma unroll
          for (int p = 0; p < NPV; ++p) pv[p] = (uint32_t)__shfl((int)pvme, p * CPT + j, 64);
          const uint32_t celem = (uint32_t)(k0 + klS) + (uint32_t)K * (uint32_t)cid[glS];
          const uint32_t zgb = lds_off(zG) + ((uint32_t)klS << 2), zkb = lds_off(zK) + ((uint32_t)glS << 2);
          // one quad: Philox block -> 4 words -> search -> 4 x 2 LDS atomics straight into the step's zG[n][row] / zK[n][column]
          // (inc: 1, or 0 for a pad).  The L lanes of a cell add to one column of each table: with the odd pitches, to different
          // banks unless they picked the same factor.  A private histogram per cell (k_zalloc_sort keeps one per lane) would meet the
          // same conflicts among the cell's lanes, cost three more instructions per count to pack, and has to be flushed.
          auto quad = [&](int qidx, uint32_t inc1, uint32_t inc2, uint32_t inc3) {
            const u32x4 w = philox4x32_7((uint32_t)qidx, celem, t, BNMF_V_Z, d.k0, d.k1);
            const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int p = 0; p < NPV; ++p) cmp_acc4(pv[p], u0, u1, u2, u3, a0, a1, a2, a3);
            uint32_t A0 = mad24(a0, 6 * ROW, tbj + ROW), A1 = mad24(a1, 6 * ROW, tbj + ROW), A2 = mad24(a2, 6 * ROW, tbj + ROW), A3 = mad24(a3, 6 * ROW, tbj + ROW);
            uint32_t b0 = mul24(a0, (int)bstride), b1 = mul24(a1, (int)bstride), b2 = mul24(a2, (int)bstride), b3 = mul24(a3, (int)bstride);
            if (BPL > 1) {                                                  // wave-uniform: with one block per range the pivots have named the block
              const u4 x0 = lds_ld4(A0 - ROW), x1 = lds_ld4(A1 - ROW), x2 = lds_ld4(A2 - ROW), x3 = lds_ld4(A3 - ROW);
              const uint32_t s0 = le4(x0, u0), s1 = le4(x1, u1), s2 = le4(x2, u2), s3 = le4(x3, u3);
              A0 = mad24(s0, ROW, A0); A1 = mad24(s1, ROW, A1); A2 = mad24(s2, ROW, A2); A3 = mad24(s3, ROW, A3);
              b0 = mad24(s0, 5, b0); b1 = mad24(s1, 5, b1); b2 = mad24(s2, 5, b2); b3 = mad24(s3, 5, b3);
            }
            const u4 y0 = lds_ld4(A0), y1 = lds_ld4(A1), y2 = lds_ld4(A2), y3 = lds_ld4(A3);
            b0 += le4(y0, u0); b1 += le4(y1, u1); b2 += le4(y2, u2); b3 += le4(y3, u3);
            lds_add(mad24(b0, GP * 4, zgb), 1u); lds_add(mad24(b0, KQ * 4, zkb), 1u);
            lds_add(mad24(b1, GP * 4, zgb), inc1); lds_add(mad24(b1, KQ * 4, zkb), inc1);
            lds_add(mad24(b2, GP * 4, zgb), inc2); lds_add(mad24(b2, KQ * 4, zkb), inc2);
            lds_add(mad24(b3, GP * 4, zgb), inc3); lds_add(mad24(b3, KQ * 4, zkb), inc3);
          };
          wave_lds_fence();
          ZPTOC(2);
          ZPTIC(3);
          // lane q of the cell takes quads q, q + L, ...; the cell's last quad is the only one that can hold pads
          for (int i = q; __builtin_amdgcn_ballot_w64(i < nqS) != 0; i += L)
            if (i < nqS) {
              const bool lastq = i == nqS - 1;
              quad(q0S + i, (lastq && npadS > 2) ? 0u : 1u, (lastq && npadS > 1) ? 0u : 1u, (lastq && npadS > 0) ? 0u : 1u);
            }
          wave_lds_fence();
          ZPTOC(3);
        }
      }
      // the next step's rows of P and counts: requested here (not before the tasks: 17 registers less through the hot loops), in flight
      // through the barrier and the end-of-step work, written to LDS at its end
      // (with 12 waves the kernel has 168 registers: there the request waits until the metric terms, the widest stretch, are done)
      if (ZP_W < 12 && !last_step) prefetch(bn, last_ch ? 0 : ch + 1);
      { ZPTIC(5); __syncthreads(); ZPTOC(5); }
      ZPTIC(0);
      // ---------------- end of the step: the chunk's share of ZsumG, the metric terms of the chunk's rows, the next step's staging
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
      if (ZP_W >= 12 && !last_step) prefetch(bn, last_ch ? 0 : ch + 1);
      for (int i = tid; i < Na * ZP_KC; i += ZP_T) {
        const int kl = i & 31, ic = i >> 5;
        const uint32_t v = zG[ic * GP + kl];
        if (v) { atomicAdd(&d.ZsumG[k0 + kl + (size_t)K * act[ic]], (int32_t)v); zG[ic * GP + kl] = 0; }
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
          const int gl = i / N, n = i - gl * N, ic = inv[n];
          int32_t v = 0;
          if (ic >= 0) { v = (int32_t)zK[ic * KQ + gl]; zK[ic * KQ + gl] = 0; }
          d.ZsumK[n + (size_t)N * cid[gl]] = v;
        }
        if (!last_step) stage_batch(bn, (bi + 1) & 1);
      }
      if (!last_step) commit((stepno + 1) & 1);
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
