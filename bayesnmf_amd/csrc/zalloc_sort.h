// bayesnmf_amd/csrc/zalloc_sort.h — k_zalloc_sort: the Z-allocation kernel of the stats mode for N <= 25 (the metric
// configuration: K = 96, G = 10,000, N = 20).  Same stream spec, bit-identical ZsumK / ZsumG / metric partials as
// k_zalloc_reg and k_zalloc; different machine mapping.
//
// M never changes during a chain, so the work is scheduled ONCE, on the host, at bnmf_create (api.hip build_zsort):
//   * the columns are dealt into blocks of equal total count (one workgroup per block, one block per CU);
//   * inside a block every non-empty cell becomes an item (cells above 4*ZS_QMAX counts: several items, each a run of
//     ZS_QMAX Philox blocks of the cell's stream); the items are sorted by their number of quads and cut into tasks of
//     64: a task is 64 cells with (nearly) the same number of counts.
// A wavefront takes a task at a time (LDS ticket): LANE = ITEM.  The lane builds its cell's thresholds once — the
// pivots (every 5th threshold) stay in registers, the others go to the lane's own column of the wave's LDS slab as
// 128-bit blocks [block][lane] (conflict-free by construction) — and then runs its quads: Philox block, pivot compares
// in registers, ONE 128-bit LDS read, four compares, one LDS atomic into the lane's private packed 16-bit histogram.
// No cell switch, no prefix scan, no random LDS address inside the loop.  The histogram is flushed per item into the
// block's zG[n][k] / zK[n][column] (LDS atomics), those once per block into ZsumG (global integer atomics: exact,
// order-independent) and ZsumK (plain stores: a column belongs to one block).  The per-column metric terms need the
// canonical W = 64 order over the rows: rounds 3-4 formed Mhat a second time for them, lane = row ("metric tasks"); since round 5 the
// lane of a cell's first item stores the Mhat it has formed (s.mh) and colterms.h sums the terms from there, beside the next allocation kernel.
// save_Z (round 4; samples$Z, R/sample_params.R:80-84, R/bayesNMF_sampler.R:245-252): the same kernel with s.rec set.  A lane's
// histogram is its item's share of Z[k, ., g]; the flush also writes it — packed as the flush packs it, two 16-bit counts per
// word — to the record buffer rec[task][word][lane]: one coalesced 256-byte store per word and task.  k_zexpand then turns a
// block's records into its columns of Z: the records are added into an LDS slab [column][factor][row] (16-bit halves; the
// fragments of a cell above 128 counts meet there), which leaves as whole 16-byte stores of Z[k .. k + 3 + K (n + N g)].  No
// scattered global store, no zero fill: 38 MB of records written and read beside the 77 MB of Z at the metric configuration.
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZS_QMAX = 32;          // quads per item (128 counts): larger cells are split
constexpr int ZS_QMAX16 = 64;        // ... with 2-byte items (3 bits of fragment index: 8 x 256 counts)
constexpr int ZS_NMAX = 25;          // 5 blocks of 4 thresholds + 4 pivots = 24 thresholds

struct ZSBlock { int item0, ntask, col0, ncols; };   // items [item0, item0 + 64 ntask), columns cols[col0 .. col0 + ncols)
struct ZSGeom { int KP, GBc, nblocks; };              // pitch of zG rows, column capacity of a block
struct ZSArgs {
  ZArgs a;
  const uint32_t* items;             // k | gl << 10 | fragment << 16; 0xFFFFFFFF = empty lane.  it16 (round 4; K <= 127, <= 64 columns per block, no cell above
                                     // 2,048 counts — the metric configuration): 2-byte items k | gl << 7 | fragment << 13, 0xFFFF = empty lane: half the bytes
                                     // of the one array the kernel reads beside M
  const ZSBlock* blocks;
  const int* cols;
  const int32_t* Mblk;               // M with its columns in block order: column col0 + gl of Mblk = column cols[col0 + gl] of M
  int it16, qmax;                    // item format; quads per item (ZS_QMAX, or ZS_QMAX16 with 2-byte items)
  uint32_t* rec;                     // save_Z: [item slot / 64][(N + 1) / 2][64] the items' histograms, two 16-bit counts per word; else null
  double* mh;                        // [G][K] Mhat of every cell, left by the lane of the cell's first item for the per-column metric terms (colterms_pair)
  int prio;                          // raise the waves' issue priority (api.hip; BNMF_ZSPRIO=0: not)
  int shared;                        // large cells are spread over the blocks (api.hip build_zsort): a column's ZsumK has several writers
  unsigned long long* prof;          // -DZSPROF builds only: per-section s_memtime ticks summed over the waves (diagnostics)
};
// -DZSPROF: section timers.  [0] block set-up, [1] thresholds, [2] quad loops, [3] histogram flush, [4] metric tasks,
// [5] end barrier + epilogue, [6] whole kernel, [7] waves.  Never defined in the product build.
#ifdef ZSPROF
#define ZSTIC(i) const uint64_t zstic_##i = __builtin_amdgcn_s_memtime()
#define ZSTOC(i) zsprof[i] += __builtin_amdgcn_s_memtime() - zstic_##i
// wall-clock stamps (s_memrealtime, 100 MHz) of the waves of three blocks: prof[8 + (sel * 16 + wave) * 8 + j] (tools/zstamps.py)
#define ZSSTAMP(j) do { if (s.prof && zs_sel >= 0 && lane == 0) s.prof[8 + (zs_sel * 16 + wave) * 8 + (j)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ZSTIC(i)
#define ZSTOC(i)
#define ZSSTAMP(j)
#endif
// host and device agree on the LDS layout through these.  PK: zG / zK hold two factors per word (16-bit halves, as the
// lanes' histograms do), chosen at bnmf_create when no half can overflow
#define BNMF_HD __host__ __device__ inline
BNMF_HD int zsort_zrows(int N, bool pk) { return pk ? (N + 1) / 2 : N; }
BNMF_HD size_t zsort_shared_bytes(int K, int N, int KP, int GBc, bool pk) {
  size_t w = (size_t)zsort_zrows(N, pk) * (KP + GBc) + GBc + 4;   // zG, zK, colid, ticket (32-bit words).  (Round 5: the block's slab of M is no longer copied
                                                                 // to LDS — since the metric tasks are gone a count is read once, by its item's lane, straight from
                                                                 // the block-ordered copy of M: 15 KB less, which is what lets 14 waves fit beside the tables)
  w = (w + 3) & ~(size_t)3;
  return (w * 4 + ((size_t)K * N + (size_t)N * GBc) * 8 + 15) & ~(size_t)15;   // + Pl, ae (fp64); the waves' slabs are 16-byte aligned
}
BNMF_HD size_t zsort_wave_bytes(int NBLK, int N) { return (size_t)NBLK * 64 * 16 + (size_t)(2 * ((N + 1) / 2)) * 64 * 4; }   // threshold blocks + one histogram word per factor and lane

// <= 128 VGPRs (4 waves per SIMD by registers): three waves per SIMD of this kernel then leave the side streams' kernels
// (128 VGPRs) a wave slot on every SIMD — at 133 VGPRs they could not start before the first workgroups here had ended
template <int ZT, int NBLK /* threshold blocks per cell: covers N <= 5 NBLK */, bool PK>
__global__ __launch_bounds__(ZT) __attribute__((amdgpu_waves_per_eu(4))) void k_zalloc_sort(ZSArgs s, uint32_t t, ZSGeom zg) {
  constexpr int NPV = NBLK - 1;                           // pivots: threshold 5j + 4 closes block j
  constexpr int NC = 5 * NBLK;                            // factors covered
  constexpr int NMIN = NBLK == 1 ? 1 : 5 * (NBLK - 1) + 1; // smallest N routed here
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ZArgs& d = s.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, N = d.N, KP = zg.KP, GBc = zg.GBc;
  const int HW = (N + 1) >> 1;
  const int ZR = PK ? HW : N;                              // rows of zG / zK
  const ZSBlock bk = s.blocks[blockIdx.x];
  uint32_t* zG = (uint32_t*)smem;                          // [ZR][KP]
  uint32_t* zK = zG + (size_t)ZR * KP;                     // [ZR][GBc]
  int* colid = (int*)(zK + (size_t)ZR * GBc);              // [GBc]
  uint32_t* ticket = (uint32_t*)(colid + GBc);
  const size_t w32 = (((size_t)ZR * (KP + GBc) + GBc + 4) + 3) & ~(size_t)3;
  const int32_t* Mb = s.Mblk + (size_t)K * bk.col0;        // [ncols][K] counts of the block's columns (global memory, L2)
  double* Pl = (double*)(smem + w32 * 4);                  // [N][K]
  double* ae = Pl + (size_t)K * N;                         // [N][GBc]  A[n] E[n, column]
  unsigned char* wbase = smem + zsort_shared_bytes(K, N, KP, GBc, PK) + (size_t)wave * zsort_wave_bytes(NBLK, N);
  u4* tblk = (u4*)wbase;                                   // [NBLK][64]
  uint32_t* hist = (uint32_t*)(tblk + NBLK * 64);          // [2 HW][64] bucket counts of the lane's item: one word each (the update is an
                                                           // address and an add; packed in 16-bit halves it was three more instructions per count)
#ifdef ZSPROF
  uint64_t zsprof[8] = {0, 0, 0, 0, 0, 0, 0, 1};
#endif
#ifdef ZSPROF
  int zs_ntask = 0;
  const int zs_sel = blockIdx.x == 0 ? 0 : blockIdx.x == 128 ? 1 : blockIdx.x == gridDim.x - 1 ? 2 : -1;
#endif
  ZSSTAMP(0);
  ZSTIC(6);
  ZSTIC(0);
  // ---------------- block set-up.  Round 5: every lane requests ALL its loads before it uses the first.  Rounds 3-4 gave the three
  // global-memory chains (P; the block's slab of M; column ids -> E) to different waves, one element per round trip: the copy of P was 7.5
  // round trips in a row, an A E product two dependent ones, 3 products per lane — 9.2 us of the kernel's 55 at the set-up barrier
  // (section timers), in front of every block's first task since the metric tasks are gone.  Now: the column ids of the lane's products, then
  // its share of P (UP elements), then — the ids have arrived — E; the LDS writes follow.  Two round trips.
  for (int i = lane; i < 2 * HW * 64; i += 64) hist[i] = 0;
  {
    constexpr int UA = 2, UP = 4;                          // per lane and chunk: one chunk each at the metric configuration (800 / 1,920 elements, 768 or 896 lanes)
    const int nP = K * N, nA = N * bk.ncols;
    const int nchunk = max(max((nA + UA * ZT - 1) / (UA * ZT), (nP + UP * ZT - 1) / (UP * ZT)), 1);
    for (int ch = 0; ch < nchunk; ++ch) {
      int gcol[UA], an_i[UA], ax[UA];
      double an[UA], pv[UP], ev[UA];
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const int i = (ch * UA + u) * ZT + tid;
        const bool ok = i < nA;
        const int gl = ok ? i / N : 0, n = ok ? i - gl * N : 0;
        ax[u] = ok ? n * GBc + gl : -1; an_i[u] = n;
        gcol[u] = s.cols[bk.col0 + gl];
        an[u] = d.A[n];
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) { const int i = (ch * UP + u) * ZT + tid; pv[u] = i < nP ? d.P[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < UA; ++u) ev[u] = d.E[an_i[u] + (size_t)N * gcol[u]];
#pragma unroll
      for (int u = 0; u < UP; ++u) { const int i = (ch * UP + u) * ZT + tid; if (i < nP) Pl[i] = pv[u]; }
#pragma unroll
      for (int u = 0; u < UA; ++u) if (ax[u] >= 0) ae[ax[u]] = an[u] * ev[u];
    }
    for (int i = tid; i < bk.ncols; i += ZT) colid[i] = s.cols[bk.col0 + i];
    for (int i = tid; i < ZR * KP; i += ZT) zG[i] = 0;
    for (int i = tid; i < ZR * GBc; i += ZT) zK[i] = 0;
    if (tid == 0) *ticket = 0;
  }
  ZSSTAMP(1);
  __syncthreads();
  ZSTOC(0);
  ZSSTAMP(2);
  // The kernel is bound by instruction issue and the side streams' kernels run beside it on the same SIMDs: its waves take the issue
  // priority (timing only).  Small side launches raise theirs to 3 (kernels.h side_body): they are few and the next draw kernel waits for them.
  if (s.prio) __builtin_amdgcn_s_setprio(2);
  const int nthr = N - 1;
  const int ntot = bk.ntask;
  const uint32_t hlb = lds_off(hist + lane);
  // the next task of the block: lane 0 draws a ticket, every lane reads lane 0's (readlane: whatever EXEC is)
  auto next_task = [&]() -> int {
    int tk = 0;
    if (lane == 0) tk = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readlane(tk, 0);
  };
  // the item tasks in descending size: the block's tail is made of its smallest tasks (the last ones: the cells without counts, which
  // are items too since round 5 — their lanes form Mhat and nothing else).  Round 5: no metric tasks.  The per-column metric terms
  // (R/utils.R:412-455) need Mhat of every cell in the canonical W = 64 order over the rows; rounds 3-4 formed it a second time, lane = row
  // (14 % of the kernel's wave time; without those dot products the iteration was 4.5 us shorter).  Now the lane of a cell's first item
  // leaves the Mhat it has just formed in s.mh, and the terms are summed from there by colterms_pair (kernels.h): in idle slots of the
  // next draw kernel, or by k_colterms.
  for (int task = next_task(); task < ntot; task = next_task()) {
    {
    // ---------------- item task: lane = item
    ZSTIC(1);
    uint32_t it;
    if (s.it16) {                                          // wave-uniform: the same fields from half the bytes
      const uint32_t r = ((const uint16_t*)s.items)[(size_t)bk.item0 + (size_t)task * 64 + lane];
      it = r == 0xFFFFu ? 0xFFFFFFFFu : ((r & 127u) | (((r >> 7) & 63u) << 10) | ((r >> 13) << 16));
    } else it = s.items[(size_t)bk.item0 + (size_t)task * 64 + lane];
    const bool valid = it != 0xFFFFFFFFu;
    const int k = valid ? (int)(it & 1023u) : 0, gl = valid ? (int)((it >> 10) & 63u) : 0;
    const int q0 = valid ? (int)(it >> 16) * s.qmax : 0;
    const int m = Mb[k + (size_t)K * gl];
    const int g = colid[gl];
    int nq = 0, npad = 0;
    bool any = false;
    uint32_t pv[NPV > 0 ? NPV : 1];
    {
      // Mhat = sum_n P[k,n] (A[n] E[n,g]) in factor order; thr_n = floor(cum_n 2^32 / Mhat) saturating at 2^32 - 1 = "never"
      // (factors at/after the last positive one saturate by themselves, see zalloc_reg.h)
      // the running sums cum_n are kept (NC - 1 fp64 registers) between the pass that forms Mhat and the thresholds: the same
      // additions in the same order as a second accumulation from zero, without reading the row of P and the column of A E again
      const double* Pk = Pl + k;
      const double* ag = ae + gl;
      double cs[NC];
      double c = 0.0;
#pragma unroll
      for (int n = 0; n < NC; ++n) { if (n < NMIN || n < N) c = c + Pk[(size_t)K * n] * ag[(size_t)n * GBc]; cs[n] = c; }
      if (valid && c > 0.0 && m > 0) {
        const int qt = (m + 3) >> 2;
        nq = min(s.qmax, qt - q0);
        npad = (q0 + nq == qt) ? ((4 - (m & 3)) & 3) : 0;
      }
      if (valid && q0 == 0) s.mh[(size_t)k + (size_t)K * (size_t)g] = c;
      any = __builtin_amdgcn_ballot_w64(nq > 0) != 0;      // a task of cells without counts: no thresholds, no draws
      if (any) {
      const double scale = 4294967296.0 / c;
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        uint32_t tv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int n = 5 * j + i;
          tv[i] = 0xFFFFFFFFu;
          if (n < NC - 1 && (n < NMIN - 1 || n < nthr)) tv[i] = cvt_u32_sat(cs[n] * scale);
        }
        tblk[j * 64 + lane] = u4{tv[0], tv[1], tv[2], tv[3]};
        if (j < NPV) pv[j] = tv[4];
      }
      }
    }
    const uint32_t celem = (uint32_t)k + (uint32_t)K * (uint32_t)g;
    // one quad: Philox block q of the cell's stream -> 4 words -> 4 buckets -> 4 histogram increments (inc: 1, or 0 for a pad)
    auto quad = [&](int qidx, uint32_t inc0, uint32_t inc1, uint32_t inc2, uint32_t inc3) {
      const u32x4 w = philox4x32_7((uint32_t)qidx, celem, t, BNMF_V_Z, d.k0, d.k1);
      const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
      uint32_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
#pragma unroll
      for (int p = 0; p < NPV; ++p) cmp_acc4(pv[p], u0, u1, u2, u3, j0, j1, j2, j3);
      const u4 k0 = tblk[j0 * 64 + lane], k1 = tblk[j1 * 64 + lane], k2 = tblk[j2 * 64 + lane], k3 = tblk[j3 * 64 + lane];
      const uint32_t b0 = 5 * j0 + (k0.x <= u0) + (k0.y <= u0) + (k0.z <= u0) + (k0.w <= u0);
      const uint32_t b1 = 5 * j1 + (k1.x <= u1) + (k1.y <= u1) + (k1.z <= u1) + (k1.w <= u1);
      const uint32_t b2 = 5 * j2 + (k2.x <= u2) + (k2.y <= u2) + (k2.z <= u2) + (k2.w <= u2);
      const uint32_t b3 = 5 * j3 + (k3.x <= u3) + (k3.y <= u3) + (k3.z <= u3) + (k3.w <= u3);
      lds_add(mad24(b0, 256, hlb), inc0);
      lds_add(mad24(b1, 256, hlb), inc1);
      lds_add(mad24(b2, 256, hlb), inc2);
      lds_add(mad24(b3, 256, hlb), inc3);
    };
    wave_lds_fence();
    ZSTOC(1);
#ifdef ZSPROF
    if (zs_ntask == 0) ZSSTAMP(6);
#endif
    ZSTIC(2);
    // full quads, then each lane's last quad (the only one that can hold pads)
    if (any) for (int i = 0; __builtin_amdgcn_ballot_w64(i < nq - 1) != 0; ++i)
      if (i < nq - 1) quad(q0 + i, 1u, 1u, 1u, 1u);
    if (nq > 0) quad(q0 + nq - 1, 1u, npad > 2 ? 0u : 1u, npad > 1 ? 0u : 1u, npad > 0 ? 0u : 1u);
    wave_lds_fence();
    ZSTOC(2);
#ifdef ZSPROF
    if (zs_ntask == 0) ZSSTAMP(7);
    ++zs_ntask;
#endif
    ZSTIC(3);
    // flush the lane's histogram into the block's tables
    if (s.rec) {                       // save_Z: the item's record (lanes without counts write zeros: their histogram is clear)
      uint32_t* rt = s.rec + ((size_t)(bk.item0 >> 6) + (size_t)task) * (size_t)HW * 64 + lane;
      for (int w = 0; w < HW; ++w) rt[(size_t)w * 64] = hist[(2 * w) * 64 + lane] | (hist[(2 * w + 1) * 64 + lane] << 16);
    }
    if (nq > 0) {
      const uint32_t zgb = lds_off(zG) + ((uint32_t)k << 2), zkb = lds_off(zK) + ((uint32_t)gl << 2);
      for (int w = 0; w < HW; ++w) {
        const uint32_t v0 = hist[(2 * w) * 64 + lane], v1 = hist[(2 * w + 1) * 64 + lane];
        const uint32_t v = v0 | (v1 << 16);                // a cell holds at most 4 ZS_QMAX counts per item
        if (v) {
          hist[(2 * w) * 64 + lane] = 0; hist[(2 * w + 1) * 64 + lane] = 0;
          if (PK) {                      // the pair of factors 2w, 2w + 1 in one add (no half can overflow: checked at bnmf_create)
            lds_add(zgb + (uint32_t)w * (uint32_t)KP * 4u, v); lds_add(zkb + (uint32_t)w * (uint32_t)GBc * 4u, v);
          } else {
            const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
            if (lo) { lds_add(zgb + (uint32_t)(2 * w) * (uint32_t)KP * 4u, lo); lds_add(zkb + (uint32_t)(2 * w) * (uint32_t)GBc * 4u, lo); }
            if (hi) { lds_add(zgb + (uint32_t)(2 * w + 1) * (uint32_t)KP * 4u, hi); lds_add(zkb + (uint32_t)(2 * w + 1) * (uint32_t)GBc * 4u, hi); }
          }
        }
      }
    }
    wave_lds_fence();
    ZSTOC(3);
    }
  }
  ZSSTAMP(3);
  ZSTIC(5);
  __syncthreads();
  ZSTOC(5);
  ZSSTAMP(4);
  {
  ZSTIC(0);                                               // (profile builds: the epilogue is charged to the set-up slot)
  // ---------------- block epilogue: ZsumK of the block's columns (plain stores), ZsumG (global integer atomics)
  for (int i = tid; i < N * bk.ncols; i += ZT) {
    const int gl = i / N, n = i - gl * N;
    const uint32_t v = PK ? (zK[(size_t)(n >> 1) * GBc + gl] >> ((n & 1) << 4)) & 0xFFFFu : zK[(size_t)n * GBc + gl];
    if (s.shared) { if (v) atomicAdd(&d.ZsumK[n + (size_t)N * colid[gl]], (int32_t)v); }   // (zeroed by the draw kernel that consumed the last one)
    else d.ZsumK[n + (size_t)N * colid[gl]] = (int32_t)v;
  }
  for (int i = tid; i < K * N; i += ZT) {
    const int kk = i % K, n = i / K;
    const uint32_t v = PK ? (zG[(size_t)(n >> 1) * KP + kk] >> ((n & 1) << 4)) & 0xFFFFu : zG[(size_t)n * KP + kk];
    if (v) atomicAdd(&d.ZsumG[kk + (size_t)K * n], (int32_t)v);
  }
  ZSTOC(0);
  }
  ZSTOC(6);
  ZSSTAMP(5);
#ifdef ZSPROF
  if (s.prof && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&s.prof[i], (unsigned long long)zsprof[i]);
#endif
  if (d.gate0 && blockIdx.x == 0 && tid == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(d.gate0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.gate_epoch ||
           __hip_atomic_load(d.gate1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.gate_epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) { __hip_atomic_store(d.gate_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
}

// ---- k_zexpand: Z of a block's columns from the records of its items (save_Z on the sorted schedule, see the head of the file) ----
// One workgroup per block of the schedule.  The slab holds as many of the block's columns as fit the LDS (all of them at the
// metric configuration: 40 x 20 x 48 words = 150 KB); with more, the records are walked once per group of columns.
constexpr int ZX_T = 1024;
BNMF_HD int zexpand_cols(int K, int N, size_t lds_bytes) { return (int)(lds_bytes / ((size_t)N * (size_t)((K + 1) / 2) * 4)); }
__global__ __launch_bounds__(ZX_T) void k_zexpand(ZSArgs s, int cols_per_pass) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* slab = (uint32_t*)smem;                        // [column of the pass][N][KH]: rows 2 j, 2 j + 1 in the halves of word j
  const ZArgs& d = s.a;
  const int tid = threadIdx.x, K = d.K, N = d.N, KH = (K + 1) >> 1, HW = (N + 1) >> 1;
  const ZSBlock bk = s.blocks[blockIdx.x];
  const int nslots = bk.ntask * 64;
  for (int c0 = 0; c0 < bk.ncols; c0 += cols_per_pass) {
    const int nc = min(cols_per_pass, bk.ncols - c0);
    for (int i = tid; i < nc * N * KH; i += ZX_T) slab[i] = 0u;
    __syncthreads();
    for (int i = tid; i < nslots; i += ZX_T) {             // lane = item slot, as in the allocation kernel: the records are read as they were written
      uint32_t it;
      if (s.it16) { const uint32_t r = ((const uint16_t*)s.items)[(size_t)bk.item0 + i]; it = r == 0xFFFFu ? 0xFFFFFFFFu : ((r & 127u) | (((r >> 7) & 63u) << 10)); }
      else it = s.items[(size_t)bk.item0 + i];
      if (it == 0xFFFFFFFFu) continue;
      const int k = (int)(it & 1023u), gl = (int)((it >> 10) & 63u) - c0;
      if (gl < 0 || gl >= nc) continue;
      const uint32_t* rt = s.rec + ((size_t)((bk.item0 + i) >> 6)) * (size_t)HW * 64 + (i & 63);
      uint32_t* sc = slab + (size_t)gl * N * KH + (k >> 1);
      const int sh = (k & 1) << 4;
      for (int w = 0; w < HW; ++w) {
        const uint32_t v = rt[(size_t)w * 64];
        if (v & 0xFFFFu) atomicAdd(sc + (size_t)(2 * w) * KH, (v & 0xFFFFu) << sh);          // (2 w + 1 < N whenever the upper half is set)
        if (v >> 16) atomicAdd(sc + (size_t)(2 * w + 1) * KH, (v >> 16) << sh);
      }
    }
    __syncthreads();
    // Z[k + K (n + N g)]: a column's K N entries are contiguous; four rows (two slab words) per 16-byte store when K is a multiple of 4
    for (int gl = 0; gl < nc; ++gl) {
      int32_t* zc = d.Z + (size_t)K * N * (size_t)s.cols[bk.col0 + c0 + gl];
      const uint32_t* sc = slab + (size_t)gl * N * KH;
      if ((K & 3) == 0) {
        for (int i = tid; i < N * (K >> 2); i += ZX_T) {
          const int n = i / (K >> 2), j = i - n * (K >> 2);
          const uint32_t a = sc[(size_t)n * KH + 2 * j], b = sc[(size_t)n * KH + 2 * j + 1];
          *(int4*)(zc + (size_t)K * n + 4 * j) = int4{(int)(a & 0xFFFFu), (int)(a >> 16), (int)(b & 0xFFFFu), (int)(b >> 16)};
        }
      } else {
        for (int i = tid; i < N * K; i += ZX_T) {
          const int n = i / K, k = i - n * K;
          zc[i] = (int32_t)((sc[(size_t)n * KH + (k >> 1)] >> ((k & 1) << 4)) & 0xFFFFu);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace bnmf
