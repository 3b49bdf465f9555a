// cr_kernels.hip -- the linear-solve half of one Gauss-Newton iteration, restructured for
// latency (SURVEY.md "hard part: sequential chain"):
//
//   k_assemble   : one wavefront per (trajectory, support state).  Builds the block's 16x16 tiles
//                  from the per-point records (Kronecker weights, appendix A.6) fully in parallel:
//                  S_i = [D_i | -g_i] for every block and the two couplings H_{i,i-1}, H_{i,i+1}
//                  for odd blocks (even blocks never use their stride-1 couplings).
//   k_gn_step_cr : one 1024-thread workgroup (16 wavefronts) per trajectory.  Block cyclic
//                  reduction = block Cholesky in nested-dissection order: level h eliminates the
//                  blocks that are odd multiples of h, all in parallel across wavefronts, so the
//                  dependent chain is log2(N) levels instead of N blocks.  Each elimination is the
//                  register-resident tile Cholesky of tiles.h; Schur complements and fill-in
//                  couplings are A^T B tile products on v_mfma_f64_16x16x4_f64.  Also carries the
//                  gpmp2::optimize control flow (planner/BatchTrajOptimizer.cpp:273-307) and the
//                  retract.
#include "assembler.h"
#include "plan_device.h"

namespace g2 {

// Diagnostic build only (-DG2_STAMPS): s_memtime stamps of one workgroup's phases, written to a
// buffer nothing else reads (cdna_hip_programming.md section 7 "In-kernel stamps").
#ifdef G2_STAMPS
#define G2_STAMP(k)                                                          \
  do {                                                                       \
    if (tid == 0 && (k) < 64 && g2_son) pb.stamps[(size_t)b * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// (the gate is read once per function: a stamp then costs one s_memtime and one store, not a load as well)
#define G2_STAMP_DECL const bool g2_son = pb.iters[b] == G2_STAMP_ITER
#else
#define G2_STAMP(k) do {} while (0)
#define G2_STAMP_DECL do {} while (0)
#endif
#ifdef G2_TSTAMPS
// -DG2_TSTAMPS on top of -DG2_STAMPS (`make stamps STAMPFLAGS=-DG2_TSTAMPS`; these stamps and the waits that delimit their
// phases lengthen a level by ~30 %): phases of ONE elimination task (wave 0 of the workgroup) at levels 4, 8 and 16: second stamp region of the plan (row
// B + b), slots 0..5 / 8..13 / 16..21
#define G2_TSTAMP(q)                                                                                     \
  do {                                                                                                   \
    if (w == 0 && lane == 0 && idx == 0 && g2_son && (h == 4 || h == 8 || h == 16)) \
      pb.stamps[((size_t)gridDim.x + b) * 64 + (h == 4 ? 0 : h == 8 ? 8 : 16) + (q)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define G2_TSTAMP(q) do {} while (0)
#endif
#ifdef G2_STAMPS
#define G2_ASTAMP(k) do { if (i == 1 && lane == 0 && pb.iters[b] == G2_STAMP_ITER) pb.stamps[(size_t)b * 64 + 32 + (k)] = __builtin_amdgcn_s_memtime(); \
                          if (i == 2 && lane == 0 && pb.iters[b] == G2_STAMP_ITER) pb.stamps[(size_t)b * 64 + 40 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G2_ASTAMP(k) do {} while (0)
#endif

// S -= A^T A restricted to real rows (< n) and to the matrix + rhs columns
template <int n>
__device__ __forceinline__ void schur_sub(Tile& S, const Tile& A, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const Tile T = tile_atb(A, A);
#pragma unroll
  for (int k = 0; k < 4; k++)
    if ((g + 4 * k) < n && (c < n || c == RHSCOL)) S.r[k] -= T.r[k];
}
// A^T A restricted like schur_sub (real rows, matrix + rhs columns), as a tile of its own.
// The restriction is a v_cndmask on a LITERAL lane mask (inline asm): written in C++ as
// `rows && (c < n || c == RHSCOL) ? t : 0.0` -- also with non-short-circuit operators -- hipcc 7.2 lowered the select for
// n = 4, 6, 8 (registers whose row test is compile-time true) to exec-masked control flow that zeroed EVERY column below
// RHSCOL, i.e. the whole matrix part (s_and_saveexec on c < 15, the zero move, and only then the c < 8 test); found
// through the planar-arm cases of tests/test_gpu_plan.py, scripts/probes/fold_debug2.py shows the tile.
constexpr unsigned long long schur_mask(int n, int k) {
  unsigned long long m = 0;
  for (int g = 0; g < 4; g++)
    for (int c = 0; c < 16; c++)
      if (g + 4 * k < n && (c < n || c == RHSCOL)) m |= 1ull << (16 * g + c);
  return m;
}
template <unsigned long long M>
__device__ __forceinline__ double keep_lanes(double x) {   // x in the lanes whose bit is set in M, 0.0 elsewhere
  int lo = __double2loint(x), hi = __double2hiint(x);
  const unsigned long long m = M;
  asm("v_cndmask_b32_e64 %0, 0, %0, %2\n\tv_cndmask_b32_e64 %1, 0, %1, %2" : "+v"(lo), "+v"(hi) : "s"(m));
  return __hiloint2double(hi, lo);
}
template <int n>
__device__ __forceinline__ void schur_keep(Tile& T) {
  T.r[0] = keep_lanes<schur_mask(n, 0)>(T.r[0]);
  T.r[1] = keep_lanes<schur_mask(n, 1)>(T.r[1]);
  T.r[2] = keep_lanes<schur_mask(n, 2)>(T.r[2]);
  T.r[3] = keep_lanes<schur_mask(n, 3)>(T.r[3]);
}
template <int n>
__device__ __forceinline__ Tile schur_prod(const Tile& A, int lane) {
  Tile T = tile_atb(A, A);
  schur_keep<n>(T);
  return T;
}
// -(A^T B) restricted to the n x n matrix part
template <int n>
__device__ __forceinline__ Tile coupling(const Tile& A, const Tile& B, int lane) {
  const int c = lane & 15, g = lane >> 4;
  Tile T = tile_atb(A, B);
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = ((g + 4 * k) < n && c < n) ? -T.r[k] : 0.0;
  return T;
}

// =============================================================================== assemble
// One workgroup = 4 wavefronts = the 4 consecutive blocks 4q .. 4q+3 of one trajectory; wavefront r
// forms block i = 4q + r.  Levels 1 AND 2 of the cyclic reduction happen here, spread over the whole
// chip: odd blocks (r = 1, 3) are eliminated as soon as they are formed; block 4q + 2 then absorbs
// their Schur complements (handed over through LDS), gets its fill-in couplings to 4q and 4q + 4 and is
// eliminated too.  The per-trajectory solve kernels start at level 4 (cr_forward).
//
// Round 3: everything the eliminated blocks of the group owe to the surviving multiples of 4 is FOLDED here as well,
// on matrix cores that this (vector-issue bound) kernel leaves idle, instead of in the per-trajectory step kernel whose
// first two levels were bound by exactly these products (416 + 408 v_mfma_f64 per trajectory on ONE compute unit):
//   * S(4q)   -= W_l(4q+1)^T W_l(4q+1) + W_l(4q+2)^T W_l(4q+2)              written back as the block's diagonal tile
//   * pend[q]  = W_r(4q+3)^T W_r(4q+3) + W_r(4q+2)^T W_r(4q+2)              what block 4q + 4 (next group) still owes
//   * coup[q]  = the fill-in coupling between 4q and 4q + 4 through 4q + 2, in the orientation of whichever of the
//                two is eliminated at level 4 (the odd multiple of 4): rows 4q, cols 4q+4 for odd q, the transpose else
// so that a level-4 task is: S - pend, two ready-made couplings, eliminate -- no tile product in front of it.
constexpr int ASM_WAVES = 4;
// register budget of k_assemble: wavefronts per SIMD the kernel must leave room for (5 -> <= 96 VGPRs)
#ifndef G2_ASM_MINW
#define G2_ASM_MINW 5
#endif
template <int D, bool LIE>
__global__ __launch_bounds__(64 * ASM_WAVES, LIE ? 2 : G2_ASM_MINW) void k_assemble(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                              const double* __restrict__ traj, int bufsel,
                                                              double* __restrict__ tiles,
                                                              const int* __restrict__ active) {
  constexpr int n = 2 * D;
  using Asm = Assembler<D, LIE>;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int groups = (N + ASM_WAVES) / ASM_WAVES;  // ceil((N + 1) / 4)
  const int b = blockIdx.x / groups, q = blockIdx.x - b * groups;
  if (active && !active[b]) return;
  // Dogleg retries (phase 1: same linearization, smaller trust region) need no new factorisation
  if (P.opt_type == GPMP2MI_OPT_DOGLEG && active && pb.phase[b] != 0) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int i = ASM_WAVES * q + wv;
  const bool live = i <= N;             // the last group may be partly empty
  const int ic = live ? i : N;          // idle wavefronts shadow block N up to the barriers
  extern __shared__ __attribute__((aligned(16))) double asm_smem[];
  // the 4 blocks of the group need the 5 intervals 4q .. 4q+4; every interval is staged once, into a slot all
  // wavefronts can read: wavefront wv stages interval 4q + wv (the last one also 4q + 4) and then uses slots
  // wv (interval i) and wv + 1 (interval i + 1)
  // hand-over tiles: [0] W_l, [1] W_r of block 4q+1; [2] W_l, [3] W_r of block 4q+3; S of block 4q goes where only its
  // own wavefront has read -- record slot 0 -- when a slot holds a tile, otherwise into a fifth tile
  const int slotd = Asm::slot_doubles(P.I, P.RECS, P.GPS);
  double* xch = asm_smem + (size_t)(ASM_WAVES + 1) * slotd;
  double* xs0 = (slotd >= TILE_DBL) ? asm_smem : xch + 4 * TILE_DBL;
  const double* rec = rec_of(pb, pb.which[b], bufsel);
  const double* gpu = gpu_of(pb, pb.which[b], bufsel);
  Asm as(P, pb, rec, gpu, b, lane);
  G2_ASTAMP(0);
  const typename Asm::Slot slot0 = as.make_slot(asm_smem, wv), slot1 = as.make_slot(asm_smem, wv + 1);
  as.stage2(i, slot0, slot1, wv == ASM_WAVES - 1 ? 2 : 1);
  __syncthreads();
  G2_ASTAMP(1);
  const bool odd = (i & 1) != 0;
  const bool fuse2 = N >= 2;            // level 2 is fused here unless it is the final level (N < 2)
  const bool lvl2 = fuse2 && wv == 2 && live;
  Tile S, Cl, Cr;
  double err_acc = as.build_tiles(ic, slot0, slot1, traj + ((size_t)b * (N + 1) + ic) * n, S, Cl, Cr,
                                  odd || P.opt_type == GPMP2MI_OPT_DOGLEG);
  G2_ASTAMP(2);
  if (live) {
    err_acc = wave_sum(err_acc);
    if (lane == 0) pb.epart[(size_t)b * P.Npad + i] = 0.5 * err_acc;
    // gradient g_i (the rhs column holds -g_i), kept for the step-control scalars of LM / Dogleg
    if (c == RHSCOL) {
#pragma unroll
      for (int k = 0; k < 4; k++) pb.gvec[((size_t)b * (N + 1) + i) * 16 + g + 4 * k] = -S.r[k];
    }
    if (P.opt_type == GPMP2MI_OPT_DOGLEG) {  // un-eliminated blocks for g^T H g (k_ghg)
      double* ht = pb.htiles + ((size_t)b * (N + 1) + i) * 2 * TILE_DBL;
      tile_store(ht, S, lane);
      tile_store(ht + TILE_DBL, Cr, lane);
    }
    if (P.opt_type == GPMP2MI_OPT_LM) {  // LM damping: sqrt(lambda) I prior rows on every variable
      const double lam = pb.lambda[b];
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (g + 4 * k == c && c < n) S.r[k] += lam;
    }
    if (!odd) {
      if (!fuse2) tile_store_rows<n>(tiles + ((size_t)b * (N + 1) + i) * TILE_DBL, S, lane);
      else if (wv == 0) tile_store(xs0, S, lane);   // folded and written back by wavefront 2
    } else {
      // level h = 1: odd blocks only couple to their (even) neighbours
      Tile V;   // holds Vt = R^-1 (tiles.h: column-form elimination); stored transposed, as V
      G2_ASTAMP(3);
      const bool ok = tile_eliminate_cv<n>(S, Cl, Cr, V, lane);
      G2_ASTAMP(4);
      double* f = pb.fac + ((size_t)b * (N + 1) + i) * 3 * TILE_DBL;
      tile_store_rows<n>(f, Cl, lane);
      tile_store_rows<n>(f + TILE_DBL, Cr, lane);
      tile_store_transposed<n>(f + 2 * TILE_DBL, V, lane);
      if (fuse2) {  // hand W_l, W_r to wavefront 2
        double* x = xch + (size_t)(wv >> 1) * 2 * TILE_DBL;
        tile_store(x, Cl, lane);
        tile_store(x + TILE_DBL, Cr, lane);
      }
      if (!ok && lane == 0) pb.notspd[b] = 1;
      G2_ASTAMP(5);
    }
  }
  if (!fuse2) return;
  __syncthreads();
  G2_ASTAMP(6);
  if (wv != 2) return;
  // wavefront 2: level h = 2 for block j = 4q + 2 (an odd multiple of 2; the E task of cr_forward with the neighbours'
  // factor tiles taken from LDS) when that block exists, and the folding for block 4q / the next group either way
  const int j0 = ASM_WAVES * q;
  const bool next = j0 + 4 <= N;
  Tile Sp = tile_load(xs0, lane);                             // S of block 4q
  // what the odd blocks owe: W_l(4q+1)^T W_l(4q+1) to block 4q, W_r(4q+3)^T W_r(4q+3) to block 4q + 4.  The products are
  // issued here, in front of the level-2 elimination, and only masked / applied behind it: the matrix cores work through
  // them while the pivot loop runs on the vector ALU.
  Tile A1 = tile_zero(), R = tile_zero();
  if (j0 + 1 <= N) {
    const Tile Wl1 = tile_load(xch, lane);
    A1 = tile_atb(Wl1, Wl1);
  }
  if (next && j0 + 3 <= N) {
    const Tile Wr3 = tile_load(xch + 3 * TILE_DBL, lane);
    R = tile_atb(Wr3, Wr3);
  }
  if (live) {
    const int j = i;
    const Tile Wr_m = tile_load(xch + TILE_DBL, lane);       // block j - 1: W_r
    schur_sub<n>(S, Wr_m, lane);
    const Tile Wl_m = tile_load(xch, lane);                  // block j - 1: W_l
    Tile C2l = coupling<n>(Wr_m, Wl_m, lane);                // rows j, cols j - 2
    Tile C2r = tile_zero();
    if (j + 1 <= N) {
      const Tile Wl_p = tile_load(xch + 2 * TILE_DBL, lane); // block j + 1: W_l
      schur_sub<n>(S, Wl_p, lane);
      if (j + 2 <= N) {
        const Tile Wr_p = tile_load(xch + 3 * TILE_DBL, lane);
        C2r = coupling<n>(Wl_p, Wr_p, lane);                 // rows j, cols j + 2
      }
    }
    Tile V;
    const bool ok = tile_eliminate_cv<n>(S, C2l, C2r, V, lane);
    double* f = pb.fac + ((size_t)b * (N + 1) + j) * 3 * TILE_DBL;
    tile_store_rows<n>(f, C2l, lane);
    tile_store_rows<n>(f + TILE_DBL, C2r, lane);
    tile_store_transposed<n>(f + 2 * TILE_DBL, V, lane);
    if (!ok && lane == 0) pb.notspd[b] = 1;
    const Tile A2 = tile_atb(C2l, C2l);                      // W_l(j)^T W_l(j): owed by block 4q
#pragma unroll
    for (int k = 0; k < 4; k++) A1.r[k] += A2.r[k];
    if (next) {
      const Tile B2 = tile_atb(C2r, C2r);                    // W_r(j)^T W_r(j): owed by block 4q + 4
#pragma unroll
      for (int k = 0; k < 4; k++) R.r[k] += B2.r[k];
      // fill-in between 4q and 4q + 4: rows of the one that level 4 eliminates
      const Tile K = (q & 1) ? coupling<n>(C2l, C2r, lane) : coupling<n>(C2r, C2l, lane);
      tile_store_rows<n>(pb.coup + ((size_t)b * groups + q) * TILE_DBL, K, lane);
    }
    G2_ASTAMP(7);
  }
  schur_keep<n>(A1);                                          // real rows, matrix + rhs columns (as schur_sub)
  schur_keep<n>(R);
#pragma unroll
  for (int k = 0; k < 4; k++) Sp.r[k] -= A1.r[k];
  tile_store_rows<n>(tiles + ((size_t)b * (N + 1) + j0) * TILE_DBL, Sp, lane);
  if (next) tile_store_rows<n>(pb.pend + ((size_t)b * groups + q) * TILE_DBL, R, lane);
}

int launch_assemble(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                    const int* active, hipStream_t st) {
  const dim3 grid(hp.B * ((hp.N + ASM_WAVES) / ASM_WAVES)), block(64 * ASM_WAVES);
  const size_t slotd = (size_t)(hp.I + 1) * hp.RECS + hp.GPS + 24 * hp.I;
  const size_t shmem = ((ASM_WAVES + 1) * slotd + (slotd >= TILE_DBL ? 4 : 5) * TILE_DBL) * sizeof(double);
  switch (hp.D) {
#define G2_ASM_CASE(DD) \
  case DD:                                                                                              \
    if (hp.lie) k_assemble<DD, true><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, pb.tiles, active); \
    else k_assemble<DD, false><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, pb.tiles, active);       \
    break;
    G2_ASM_CASE(1) G2_ASM_CASE(2) G2_ASM_CASE(3) G2_ASM_CASE(4) G2_ASM_CASE(5) G2_ASM_CASE(6) G2_ASM_CASE(7)
#undef G2_ASM_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== GN step (CR)
constexpr int FIN_BLOCKS = 8;   // blocks per workgroup of k_finish_step / k_finish_trial (levels 4, 2, 1 run there)
#ifndef G2_CR_WAVES
#define G2_CR_WAVES 16
#endif
constexpr int CR_WAVES = G2_CR_WAVES;



// Forward elimination (levels h >= 2; level 1 was done by k_assemble) of one trajectory's system.
// All CR_WAVES wavefronts of the workgroup take part; returns false (per wavefront) on a bad pivot.
template <int n>
__device__ __forceinline__ bool cr_forward(const PlanBuffers& pb, int b, int N, int tid) {
  // (wavefront index as a scalar: the task selection below then compiles to scalar branches)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  // ---- forward: block cyclic reduction.  Level h = 1 (odd blocks) was done by k_assemble.
  // Phase h >= 2: every block that is still active (multiple of h) absorbs the Schur complements
  // of level h/2; odd multiples of h are then eliminated (E tasks), even multiples write their
  // updated diagonal tile back (U tasks, done by the wavefronts that have no E task).
  double* tiles = pb.tiles + (size_t)b * (N + 1) * TILE_DBL;  // S tile of every block
  double* fac = pb.fac + (size_t)b * (N + 1) * 3 * TILE_DBL;  // per block: Wl, Wr, V
  bool ok = true;
  G2_STAMP_DECL;
  int hfinal = 1;
  while (hfinal <= N) hfinal <<= 1;
  // levels 1 and 2 were done by k_assemble (level 2 only when it is not the final one, N >= 2), including
  // their Schur complements on the surviving blocks (multiples of 4) and the level-4 couplings
  const int h0 = (N >= 2) ? 4 : 2;
  // Level 4 would be 13 E + 13 U tasks on 16 wavefronts (two rounds) for N = 100: its U tasks (the blocks that
  // are multiples of 8) are deferred -- at level 8 every task absorbs the Schur complements of its level-1,
  // level-2 and level-4 neighbours in one go, so both levels take a single round.
  const bool defer4 = (h0 == 4) && (hfinal >= 8);
  for (int h = h0; h <= hfinal; h <<= 1) {
    const bool final = (h == hfinal);
    const int hh = h >> 1;
    const int countE = final ? 1 : ((N / h) + 1) / 2;
    const int countU = (final || (defer4 && h == 4)) ? 0 : (N / (2 * h)) + 1;  // multiples of 2h in [0, N]
    for (int idx = w; idx < countE + countU; idx += CR_WAVES) {
      const bool elim = idx < countE;
      const int j = elim ? (final ? 0 : h * (2 * idx + 1)) : 2 * h * (idx - countE);
      // every tile this task needs is requested before the first product: the level-1 / level-2 factors come from
      // the previous kernel (other XCDs' L2 -> Infinity Cache / HBM latency), and one latency is paid instead of
      // one per neighbour
      // k_assemble has folded everything the level-1 / level-2 blocks owe to the multiples of 4 (its header comment):
      // the first task that touches such a block here -- level 4 for the odd multiples of 4, level 8 for the multiples
      // of 8 -- only subtracts what the previous group left pending for it, and a level-4 task takes its two couplings
      // ready-made.  Every tile is requested before the first product (one latency per task, not one per neighbour).
      const bool first = h0 == 4 && (h == 4 || (defer4 && h == 8));
      const bool ready = h0 == 4 && h == 4;          // level-2 neighbours already absorbed, couplings precomputed
      const int jm = j - hh, jp = j + hh, qj = j >> 2;
      const bool em = jm >= 0, ep = jp <= N;
      const bool cm = em && elim && !final, cp = ep && elim && !final && j + h <= N;
      auto facp = [&](int blk, int which) { return fac + ((size_t)blk * 3 + which) * TILE_DBL; };
      const int groups = (N + 4) / 4;
      const double* pend = pb.pend + (size_t)b * groups * TILE_DBL;
      const double* coup = pb.coup + (size_t)b * groups * TILE_DBL;
      G2_TSTAMP(0);
      Tile S = tile_load_rows<n>(tiles + (size_t)j * TILE_DBL, lane);
      Tile Pd = tile_zero();
      Tile Wr_m = tile_zero(), Wl_m = tile_zero(), Wl_p = tile_zero(), Wr_p = tile_zero();
      Tile Cl = tile_zero(), Cr = tile_zero();
      if (first && qj >= 1) Pd = tile_load_rows<n>(pend + (size_t)(qj - 1) * TILE_DBL, lane);
      if (ready) {
        if (cm) Cl = tile_load_rows<n>(coup + (size_t)(qj - 1) * TILE_DBL, lane);
        if (cp) Cr = tile_load_rows<n>(coup + (size_t)qj * TILE_DBL, lane);
      } else {
        if (em) Wr_m = tile_load_rows<n>(facp(jm, 1), lane);
        if (cm) Wl_m = tile_load_rows<n>(facp(jm, 0), lane);
        if (ep) Wl_p = tile_load_rows<n>(facp(jp, 0), lane);
        if (cp) Wr_p = tile_load_rows<n>(facp(jp, 1), lane);
      }
#ifdef G2_TSTAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      G2_TSTAMP(1);
#pragma unroll
      for (int k = 0; k < 4; k++) S.r[k] -= Pd.r[k];
      if (!ready) {
        if (em) schur_sub<n>(S, Wr_m, lane);
        if (cm) Cl = coupling<n>(Wr_m, Wl_m, lane);  // rows j, cols j - h
        if (ep) schur_sub<n>(S, Wl_p, lane);
        if (cp) Cr = coupling<n>(Wl_p, Wr_p, lane);  // rows j, cols j + h
      }
      if (!elim) {
        tile_store_rows<n>(tiles + (size_t)j * TILE_DBL, S, lane);
        continue;
      }
#ifdef G2_TSTAMPS
      asm volatile("" : "+v"(S.r[0]), "+v"(Cl.r[0]), "+v"(Cr.r[0]));
#endif
      G2_TSTAMP(2);
      Tile V;
      ok = tile_eliminate_cv<n>(S, Cl, Cr, V, lane) && ok;
#ifdef G2_TSTAMPS
      asm volatile("" : "+v"(V.r[0]), "+v"(Cl.r[0]), "+v"(Cr.r[0]));
#endif
      G2_TSTAMP(3);
      double* f = fac + (size_t)j * 3 * TILE_DBL;
      tile_store_rows<n>(f, Cl, lane);
      tile_store_rows<n>(f + TILE_DBL, Cr, lane);
      tile_store_rows<n>(f + 2 * TILE_DBL, V, lane);   // levels >= 4: Vt, loaded transposed by the back-substitution
      G2_TSTAMP(4);
    }
    __syncthreads();
    { const int idx = 0; G2_TSTAMP(5); }
    G2_STAMP(5 + __builtin_ctz(h));   // 6.. : after level h = 2, 4, ...
  }
  return ok;
}

// Back-substitution down the same tree, levels hfinal .. hmin; leaves x of every block it reaches in
// xs[(N+1)][16] (LDS).
template <int n>
__device__ __forceinline__ void cr_backward(const PlanBuffers& pb, int b, int N, int tid, double* xs, int hmin = 1) {
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const double* fac = pb.fac + (size_t)b * (N + 1) * 3 * TILE_DBL;
  G2_STAMP_DECL;
  int hfinal = 1;
  while (hfinal <= N) hfinal <<= 1;
  // the factor tiles of a wavefront's first task of a level do not depend on the level above: they are requested
  // BEFORE the barrier that publishes that level's solutions, so their latency hides behind it
  auto block_of = [&](int h, int idx) { return (h == hfinal) ? 0 : h * (2 * idx + 1); };
  auto count_of = [&](int h) { return (h == hfinal) ? 1 : ((N / h) + 1) / 2; };
  Tile pWl = tile_zero(), pWr = tile_zero(), pV = tile_zero();
  auto prefetch = [&](int h) {
    if (h >= hmin && w < count_of(h)) {
      const double* f = fac + (size_t)block_of(h, w) * 3 * TILE_DBL;
      pWl = tile_load_rows<n>(f, lane);
      pWr = tile_load_rows<n>(f + TILE_DBL, lane);
      pV = load_v<n>(f + 2 * TILE_DBL, h, N, lane);
    }
  };
  prefetch(hfinal);
  for (int h = hfinal; h >= hmin; h >>= 1) {
    const bool final = (h == hfinal);
    const int count = count_of(h);
    for (int idx = w; idx < count; idx += CR_WAVES) {
      const int j = block_of(h, idx);
      Tile Wl = pWl, Wr = pWr, V = pV;
      if (idx != w) {
        const double* f = fac + (size_t)j * 3 * TILE_DBL;
        Wl = tile_load_rows<n>(f, lane);
        Wr = tile_load_rows<n>(f + TILE_DBL, lane);
        V = load_v<n>(f + 2 * TILE_DBL, h, N, lane);
      }
      const int jl = j - h, jr = j + h;
      const double xl = (!final && jl >= 0) ? xs[jl * 16 + c] : 0.0;
      const double xr = (!final && jr <= N) ? xs[jr * 16 + c] : 0.0;
      const double x = cr_backsolve<n>(Wl, Wr, V, xl, xr, lane);
      if (g == 0) xs[j * 16 + c] = (c < n) ? x : 0.0;
    }
    prefetch(h >> 1);
    __syncthreads();
    G2_STAMP(16 + __builtin_ctz(h));  // 16.. : after backward level h
  }

}

template <int D>
__device__ __forceinline__ void gn_step_body(const PlanParams& P, const PlanBuffers& pb, int pass) {
  constexpr int n = 2 * D;
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  if (!pb.active[b]) return;
  const int N = P.N;
  const size_t tsz = (size_t)(N + 1) * n;
  // fused finish (plan.h: fuse_finish): the states of pass k live in pb.cur for even k, pb.last for odd k, and the other
  // buffer holds the states the last step started from
  const bool odd = P.fuse_finish && (pass & 1);
  double* cur = (odd ? pb.last : pb.cur) + b * tsz;
  double* last = (odd ? pb.cur : pb.last) + b * tsz;
  double* result = pb.result + b * tsz;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;                       // [N+1][16] solution of each block
  double* red = smem + (size_t)(N + 1) * 16;  // [CR_WAVES] reduction scratch
  int* flags = reinterpret_cast<int*>(red + CR_WAVES);  // [0] decision, [1] not-spd
  G2_STAMP_DECL;

  G2_STAMP(0);
  // what the step control below needs from HBM is requested before the error sum, not after its barrier
  const int it = pb.iters[b];
  const double prev = pb.prev_err[b];
  // ---- graph error at `cur`: fixed-order sum of the per-block partials written by k_assemble,
  // then the gpmp2::optimize control flow
  if (w == 0) {
    double acc = 0.0;
    for (int i = lane; i <= N; i += 64) acc += pb.epart[(size_t)b * P.Npad + i];
    acc = wave_sum(acc);
    if (lane == 0) {
      red[0] = acc;
      flags[1] = 0;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const double new_err = red[0];
    int decision = 0;  // 0 iterate, 1 stop(result = cur), 2 stop(result = last)
    double* tr = pb.trace + (size_t)b * (P.max_iter + 1);
    if (it <= P.max_iter) tr[it] = new_err;
    if (pass == 0) {
      pb.prev_err[b] = new_err;
      if (P.fixed_iters > 0) decision = 0;
      else if (new_err <= P.err_tol) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_ALREADY_OPTIMAL; }
      else if (P.max_iter <= 0) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; }
    } else if (P.fixed_iters > 0) {
      if (it >= P.fixed_iters) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; }
    } else {
      const bool conv = check_convergence(P.rel_thresh, P.abs_tol, P.err_tol, prev, new_err);
      if (it < P.max_iter && !conv) {
        pb.prev_err[b] = new_err;
      } else if (new_err > prev && P.no_increase) {
        decision = 2;
        pb.status[b] = GPMP2MI_TRAJ_ROLLED_BACK;
        pb.final_err[b] = prev;
      } else {
        decision = 1;
        pb.status[b] = conv ? GPMP2MI_TRAJ_CONVERGED : GPMP2MI_TRAJ_MAX_ITER;
      }
    }
    if (decision == 1) pb.final_err[b] = new_err;
    pb.cur_err[b] = new_err;
    flags[0] = decision;
  }
  __syncthreads();
  const int decision = flags[0];
  if (decision != 0) {
    const double* src = (decision == 2) ? last : cur;
    for (size_t k = tid; k < tsz; k += blockDim.x) result[k] = src[k];
    if (tid == 0) pb.active[b] = 0;
    return;
  }

  G2_STAMP(1);
  const bool ok = cr_forward<n>(pb, b, N, tid);
  G2_STAMP(2);
  if ((!ok && lane == 0) || (tid == 0 && pb.notspd[b])) flags[1] = 1;
  __syncthreads();
  if (flags[1]) {
    if (tid == 0) pb.notspd[b] = 0;
    for (size_t k = tid; k < tsz; k += blockDim.x) result[k] = cur[k];
    if (tid == 0) {
      pb.status[b] = GPMP2MI_TRAJ_NOT_SPD;
      pb.final_err[b] = pb.cur_err[b];
      pb.active[b] = 0;
    }
    return;
  }

  if (P.split_back) {
    // the three widest back-substitution levels (88 of the 101 blocks) and the retract run chip-wide in
    // k_finish_step; this kernel only solves the blocks that are multiples of 8 and hands them over
    cr_backward<n>(pb, b, N, tid, xs, FIN_BLOCKS);
    G2_STAMP(3);
    double* xg = pb.xg + (size_t)b * (N + 1) * 16;
    for (int k = tid; k < (N / FIN_BLOCKS + 1) * 16; k += blockDim.x) {
      const size_t o = (size_t)(k >> 4) * FIN_BLOCKS * 16 + (k & 15);
      xg[o] = xs[o];
    }
    G2_STAMP(4);
    if (tid == 0) {
      pb.stepped[b] = pass + 1;
      pb.last_err[b] = pb.cur_err[b];
      pb.iters[b] += 1;
      atomicAdd(pb.n_active + pass, 1);
    }
    return;
  }
  cr_backward<n>(pb, b, N, tid, xs);
  G2_STAMP(3);
  // ---- last = cur ; cur = retract(cur, delta)   (Values::retract; Pose2 chart for mobile bases)
  for (size_t k = tid; k < tsz; k += blockDim.x) last[k] = cur[k];
  __syncthreads();
  for (size_t k = tid; k < tsz; k += blockDim.x) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double* zs = last + (size_t)i * n;
    const double* dz = xs + i * 16;
    cur[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
  }
  G2_STAMP(4);
  if (tid == 0) {
    pb.last_err[b] = pb.cur_err[b];
    pb.iters[b] += 1;
    atomicAdd(pb.n_active + pass, 1);
  }
}
template <int D>
__global__ __launch_bounds__(64 * CR_WAVES) void k_gn_step_cr(const PlanParams* __restrict__ pp,
                                                               PlanBuffers pb, int pass) {
  gn_step_body<D>(*pp, pb, pass);
  if (threadIdx.x == 0) publish_pass_count(pb, pass);
}

// Chip-wide tail of a Gauss-Newton pass (split path): one workgroup of 8 wavefronts per (trajectory, blocks
// 8q .. 8q+7).  Block 8q+4 is back-substituted from x_{8q}, x_{8q+8} (level 4), then 8q+2 / 8q+6 (level 2), then the
// odd blocks (level 1); every wavefront then retracts its own state: last = cur; cur = cur (+) x.
// (FIN_BLOCKS is declared next to CR_WAVES: the step kernels stop their back-substitution at the multiples of it.)
template <int D>
struct FinishGroup {
  static constexpr int n = 2 * D;
  Tile Wl, Wr, V;
  int b, q, wv, lane, c, g, i, N;
  bool live;
  // loads of the wavefront's factor tiles are requested right away, so they are in flight while the levels above
  // are being solved
  __device__ __forceinline__ FinishGroup(const PlanBuffers& pb, int N_, int b_, int q_, double (*xl_)[16])
      : b(b_), q(q_), N(N_) {
    wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lane = threadIdx.x & 63;
    c = lane & 15;
    g = lane >> 4;
    i = FIN_BLOCKS * q + wv;
    live = i <= N;
    const double* xg = pb.xg + (size_t)b * (N + 1) * 16;
    const double* fac = pb.fac + (size_t)b * (N + 1) * 3 * TILE_DBL;
    Wl = Wr = V = tile_zero();
    if (live && wv != 0) {
      const double* f = fac + (size_t)i * 3 * TILE_DBL;
      Wl = tile_load_rows<n>(f, lane);
      Wr = tile_load_rows<n>(f + TILE_DBL, lane);
      V = load_v<n>(f + 2 * TILE_DBL, (i & 3) ? 1 : 4, N, lane);   // blocks 8q + 4 were eliminated by the step kernel (level 4)
    }
    if (wv == 0 && lane < 16) xl_[0][lane] = xg[(size_t)(FIN_BLOCKS * q) * 16 + lane];
    if (wv == 1 && lane < 16)
      xl_[FIN_BLOCKS][lane] = (FIN_BLOCKS * q + FIN_BLOCKS <= N) ? xg[(size_t)(FIN_BLOCKS * q + FIN_BLOCKS) * 16 + lane] : 0.0;
  }
  __device__ __forceinline__ void solve(int h, double (*xl_)[16]) const {
    const int jl = i - h, jr = i + h;
    const double xl = (jl >= 0) ? xl_[jl - FIN_BLOCKS * q][c] : 0.0;
    const double xr = (jr <= N) ? xl_[jr - FIN_BLOCKS * q][c] : 0.0;
    const double x = cr_backsolve<n>(Wl, Wr, V, xl, xr, lane);
    if (g == 0) xl_[wv][c] = (c < n) ? x : 0.0;
  }
  // levels 4, 2, 1 (every wavefront of the workgroup must call this)
  __device__ __forceinline__ void solve_all(double (*xl_)[16]) const {
    __syncthreads();
    if (wv == 4 && live) solve(4, xl_);
    __syncthreads();
    if ((wv == 2 || wv == 6) && live) solve(2, xl_);
    __syncthreads();
    if ((wv & 1) && live) solve(1, xl_);
    __syncthreads();
  }
};

template <int D>
__global__ __launch_bounds__(64 * FIN_BLOCKS) void k_finish_step(const PlanParams* __restrict__ pp, PlanBuffers pb, int pass) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int groups = (N + FIN_BLOCKS) / FIN_BLOCKS;
  const int b = blockIdx.x / groups, q = blockIdx.x - b * groups;
  if (pb.stepped[b] != pass + 1) return;
  __shared__ double xl_[FIN_BLOCKS + 1][16];
  const FinishGroup<D> fg(pb, N, b, q, xl_);
  fg.solve_all(xl_);
  const int lane = fg.lane, i = fg.i;
  if (!fg.live || lane >= n) return;
  const size_t k = ((size_t)b * (N + 1) + i) * n + lane;
  const double* zs = pb.cur + ((size_t)b * (N + 1) + i) * n;
  const double zold = zs[lane];
  const double znew = (lane < D) ? retract_coord(P.lie != 0, lane, zs, xl_[fg.wv]) : zold + xl_[fg.wv][lane];
  pb.last[k] = zold;
  __builtin_amdgcn_wave_barrier();  // every lane has read the old state of this block before any lane overwrites it
  pb.cur[k] = znew;
}

int launch_finish_step(const PlanParams& hp, const PlanBuffers& pb, int pass, hipStream_t st) {
  const dim3 grid(hp.B * ((hp.N + FIN_BLOCKS) / FIN_BLOCKS)), block(64 * FIN_BLOCKS);
  switch (hp.D) {
#define G2_FIN_CASE(DD) \
  case DD: k_finish_step<DD><<<grid, block, 0, st>>>(pb.params, pb, pass); break;
    G2_FIN_CASE(1) G2_FIN_CASE(2) G2_FIN_CASE(3) G2_FIN_CASE(4) G2_FIN_CASE(5) G2_FIN_CASE(6) G2_FIN_CASE(7)
#undef G2_FIN_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// Chip-wide tail of an LM / GN trial step (split form of k_solve_step): as k_finish_step, but the step goes to
// `delta`, the trial point cur (+) delta to `trial` (cur stays), and every workgroup leaves its share of g.delta,
// |delta|^2, |g|^2 in spart for the step control (k_decide sums them in group order).
template <int D>
__global__ __launch_bounds__(64 * FIN_BLOCKS) void k_finish_trial(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int groups = (N + FIN_BLOCKS) / FIN_BLOCKS;
  const int b = blockIdx.x / groups, q = blockIdx.x - b * groups;
  if (!pb.active[b] || pb.stepped[b] != 1) return;
  __shared__ double xl_[FIN_BLOCKS + 1][16];
  __shared__ double psum[FIN_BLOCKS][3];
  const FinishGroup<D> fg(pb, N, b, q, xl_);
  fg.solve_all(xl_);
  const int lane = fg.lane, i = fg.i, wv = fg.wv;
  double gd = 0.0, dd = 0.0, gg = 0.0;
  if (fg.live && lane < n) {
    const size_t k = ((size_t)b * (N + 1) + i) * n + lane;
    const double* zs = pb.cur + ((size_t)b * (N + 1) + i) * n;
    const double x = xl_[wv][lane], gk = pb.gvec[((size_t)b * (N + 1) + i) * 16 + lane];
    pb.delta[k] = x;
    pb.trial[k] = (lane < D) ? retract_coord(P.lie != 0, lane, zs, xl_[wv]) : zs[lane] + x;
    gd = gk * x;
    dd = x * x;
    gg = gk * gk;
  }
  gd = wave_sum(gd);
  dd = wave_sum(dd);
  gg = wave_sum(gg);
  if (lane == 0) {
    psum[wv][0] = gd;
    psum[wv][1] = dd;
    psum[wv][2] = gg;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int t = threadIdx.x;
    double a = 0.0;
    for (int w = 0; w < FIN_BLOCKS; w++) a += psum[w][t];
    pb.spart[((size_t)b * groups + q) * 3 + t] = a;
  }
}

int launch_finish_trial(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B * ((hp.N + FIN_BLOCKS) / FIN_BLOCKS)), block(64 * FIN_BLOCKS);
  switch (hp.D) {
#define G2_FINT_CASE(DD) \
  case DD: k_finish_trial<DD><<<grid, block, 0, st>>>(pb.params, pb); break;
    G2_FINT_CASE(1) G2_FINT_CASE(2) G2_FINT_CASE(3) G2_FINT_CASE(4) G2_FINT_CASE(5) G2_FINT_CASE(6) G2_FINT_CASE(7)
#undef G2_FINT_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_gn_step_cr(const PlanParams& hp, const PlanBuffers& pb, int pass, hipStream_t st) {
  const dim3 grid(hp.B), block(64 * CR_WAVES);
  const size_t shmem = ((size_t)(hp.N + 1) * 16 + CR_WAVES + 2) * sizeof(double);
  if (shmem > 150 * 1024) {
    set_error("total_step too large for the LDS-resident solution buffer");
    return GPMP2MI_ERR_UNSUPPORTED;
  }
  switch (hp.D) {
#define G2_CR_CASE(DD) \
  case DD: k_gn_step_cr<DD><<<grid, block, shmem, st>>>(pb.params, pb, pass); break;
    G2_CR_CASE(1) G2_CR_CASE(2) G2_CR_CASE(3) G2_CR_CASE(4) G2_CR_CASE(5) G2_CR_CASE(6) G2_CR_CASE(7)
#undef G2_CR_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}


// =============================================================================== trial steps (LM / Dogleg)
// block-wide deterministic sum; every thread returns the total
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < CR_WAVES; k++) t += red[k];
  return t;
}

// Solve the current linearization of every active trajectory (cyclic reduction, factors from
// k_assemble) and form the trial point, without any accept / reject decision:
//   LM     : delta = -(H + lambda I)^-1 g,  trial = cur + delta; scalars g.delta, |delta|^2
//            (LevenbergMarquardtOptimizer::tryLambda up to the retract)
//   Dogleg : dx_n = -H^-1 g, dx_u = -(g.g / g^T H g) g, dogleg point for the trust radius,
//            trial = cur + dx_d, model decrease q(dx_d)  (DoglegOptimizerImpl::ComputeDoglegPoint /
//            ComputeBlend); phase 1 (radius halved after a rejected step) re-blends without solving
//   GN     : as LM with lambda = 0
template <int D>
__global__ __launch_bounds__(64 * CR_WAVES) void k_solve_step(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!pb.active[b]) return;
  const int N = P.N;
  const size_t tsz = (size_t)(N + 1) * n;
  const double* cur = pb.cur + b * tsz;
  double* trial = pb.trial + b * tsz;
  double* delta = pb.delta + b * tsz;
  double* sc = pb.scal + (size_t)b * SC_COUNT;
  const double* gv = pb.gvec + (size_t)b * (N + 1) * 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;
  double* red = smem + (size_t)(N + 1) * 16;
  int* flags = reinterpret_cast<int*>(red + CR_WAVES);
  const bool dogleg = P.opt_type == GPMP2MI_OPT_DOGLEG;
  const bool resolve = !(dogleg && pb.phase[b] != 0);
  if (tid == 0) {
    flags[1] = 0;
    pb.stepped[b] = 0;   // set again once the factorisation has succeeded (split form)
  }
  __syncthreads();
  if (resolve) {
    const bool ok = cr_forward<n>(pb, b, N, tid);
    if ((!ok && (tid & 63) == 0) || (tid == 0 && pb.notspd[b])) flags[1] = 1;
    __syncthreads();
    if (flags[1]) {
      if (tid == 0) pb.notspd[b] = 1;  // k_decide consumes and clears it
      return;
    }
    if (P.split_back && !dogleg) {
      // LM / GN: as on the Gauss-Newton fast path only the blocks that are multiples of 8 are back-substituted
      // here; levels 4, 2, 1, the step, the trial point and the step-control sums follow chip-wide in k_finish_trial
      cr_backward<n>(pb, b, N, tid, xs, FIN_BLOCKS);
      double* xg = pb.xg + (size_t)b * (N + 1) * 16;
      for (int k = tid; k < (N / FIN_BLOCKS + 1) * 16; k += blockDim.x) {
        const size_t o = (size_t)(k >> 4) * FIN_BLOCKS * 16 + (k & 15);
        xg[o] = xs[o];
      }
      if (tid == 0) pb.stepped[b] = 1;
      return;
    }
    cr_backward<n>(pb, b, N, tid, xs);
    double gd = 0.0, dd = 0.0, gg = 0.0;
    for (size_t k = tid; k < tsz; k += blockDim.x) {
      const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
      const double x = xs[i * 16 + rho], gk = gv[i * 16 + rho];
      delta[k] = x;
      gd = fma(gk, x, gd);
      dd = fma(x, x, dd);
      gg = fma(gk, gk, gg);
    }
    gd = block_sum(gd, red, tid);
    dd = block_sum(dd, red, tid);
    gg = block_sum(gg, red, tid);
    if (tid == 0) {
      sc[SC_GD] = gd;
      sc[SC_DD] = dd;
      sc[SC_GG] = gg;
      sc[SC_GN] = gd;
      sc[SC_NN] = dd;
    }
    if (dogleg) {
      double acc = 0.0;
      for (int i = tid; i <= N; i += blockDim.x) acc += pb.hgpart[(size_t)b * P.Npad + i];
      acc = block_sum(acc, red, tid);
      if (tid == 0) sc[SC_GHG] = acc;
    }
    __syncthreads();
  }
  if (!dogleg) {
    for (size_t k = tid; k < tsz; k += blockDim.x) {
      const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
      const double* zs = cur + (size_t)i * n;
      const double* dz = xs + i * 16;
      trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
    }
    return;
  }
  // ---- Powell dogleg point for trust radius pb.lambda[b]
  const double Delta = pb.lambda[b];
  const double gg = sc[SC_GG], gHg = sc[SC_GHG], gn = sc[SC_GN], nn = sc[SC_NN];
  const double step = -gg / gHg;          // dx_u = step * g   (optimizeGradientSearch)
  const double uu = step * step * gg, un = step * gn;
  const double DeltaSq = Delta * Delta;
  double cu, cn, q;                        // dx_d = cu * g + cn * dx_n
  if (DeltaSq < uu) {
    const double k = sqrt(DeltaSq / uu);
    cu = k * step;
    cn = 0.0;
    q = cu * gg + 0.5 * cu * cu * gHg;
  } else if (DeltaSq < nn) {
    const double a = uu - 2. * un + nn, bq = 2. * (un - uu), cq = uu - Delta * Delta;
    const double sq = sqrt(bq * bq - 4 * a * cq);
    const double tau1 = (-bq + sq) / (2. * a), tau2 = (-bq - sq) / (2. * a);
    const double tau = (0.0 <= tau1 && tau1 <= 1.0) ? tau1 : tau2;
    cu = (1. - tau) * step;
    cn = tau;
    // g^T x + 0.5 x^T H x with H dx_n = -g
    q = cu * gg + cn * gn + 0.5 * (cu * cu * gHg - 2.0 * cu * cn * gg - cn * cn * gn);
  } else {
    cu = 0.0;
    cn = 1.0;
    q = 0.5 * gn;
  }
  double xn = 0.0;
  __syncthreads();
  for (size_t k = tid; k < tsz; k += blockDim.x) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double x = cu * gv[i * 16 + rho] + cn * delta[k];
    xs[i * 16 + rho] = x;
    xn = fma(x, x, xn);
  }
  __syncthreads();
  for (size_t k = tid; k < tsz; k += blockDim.x) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double* zs = cur + (size_t)i * n;
    const double* dz = xs + i * 16;
    trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
  }
  xn = block_sum(xn, red, tid);
  if (tid == 0) {
    sc[SC_Q] = q;
    sc[SC_XNORM] = sqrt(xn);
  }
}

int launch_solve_step(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B), block(64 * CR_WAVES);
  const size_t shmem = ((size_t)(hp.N + 1) * 16 + CR_WAVES + 2) * sizeof(double);
  switch (hp.D) {
#define G2_SS_CASE(DD) \
  case DD: k_solve_step<DD><<<grid, block, shmem, st>>>(pb.params, pb); break;
    G2_SS_CASE(1) G2_SS_CASE(2) G2_SS_CASE(3) G2_SS_CASE(4) G2_SS_CASE(5) G2_SS_CASE(6) G2_SS_CASE(7)
#undef G2_SS_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// g^T H g, block by block: share_i = g_i^T D_i g_i + 2 g_i^T H_{i,i+1} g_{i+1} from the tiles saved by
// k_assemble (Dogleg only).  One wavefront per (trajectory, block).
template <int D>
__global__ __launch_bounds__(64) void k_ghg(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int b = blockIdx.x / (N + 1), i = blockIdx.x - b * (N + 1);
  if (!pb.active[b] || pb.phase[b] != 0) return;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const double* ht = pb.htiles + ((size_t)b * (N + 1) + i) * 2 * TILE_DBL;
  const Tile Dt = tile_load(ht, lane), Ht = tile_load(ht + TILE_DBL, lane);
  const double* gv = pb.gvec + ((size_t)b * (N + 1) + i) * 16;
  const double gi_c = (c < n) ? gv[c] : 0.0;
  const double gn_c = (c < n && i < N) ? gv[16 + c] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int rho = g + 4 * k;
    const double gi_r = (rho < n) ? gv[rho] : 0.0;
    if (c < n) acc += gi_r * (Dt.r[k] * gi_c + 2.0 * Ht.r[k] * gn_c);
  }
  acc = wave_sum(acc);
  if (lane == 0) pb.hgpart[(size_t)b * P.Npad + i] = acc;
}

int launch_ghg(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B * (hp.N + 1)), block(64);
  switch (hp.D) {
#define G2_GHG_CASE(DD) \
  case DD: k_ghg<DD><<<grid, block, 0, st>>>(pb.params, pb); break;
    G2_GHG_CASE(1) G2_GHG_CASE(2) G2_GHG_CASE(3) G2_GHG_CASE(4) G2_GHG_CASE(5) G2_GHG_CASE(6) G2_GHG_CASE(7)
#undef G2_GHG_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// diagnostic: exercises the cross-lane helpers so tests can pin their lane semantics on hardware
__global__ void k_debug_crosslane(const double* __restrict__ in, double* __restrict__ out) {
  const int l = threadIdx.x;
  const double v = in[l];
  out[0 * 64 + l] = bcast_row<0>(v);
  out[1 * 64 + l] = bcast_row<1>(v);
  out[2 * 64 + l] = bcast_row<2>(v);
  out[3 * 64 + l] = bcast_row<3>(v);
  out[4 * 64 + l] = bcast_in_row<5>(v);
  out[5 * 64 + l] = row_sum16_dpp(v);
  out[6 * 64 + l] = sum_rows(v);
  out[7 * 64 + l] = bcast_in_row<13>(v);
}
int launch_debug_crosslane(const double* in, double* out, hipStream_t st) {
  k_debug_crosslane<<<dim3(1), dim3(64), 0, st>>>(in, out);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

#include "wide_cr.h"

}  // namespace g2
