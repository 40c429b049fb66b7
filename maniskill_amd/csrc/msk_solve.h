/*
 * msk_solve.h — the TGS row solver in constraint space, one wavefront per env (gfx950, wave64).
 *
 * The Gauss-Seidel sweeps are serial in the rows, so the quantity to minimise is the dependent chain
 * of ONE row update.  Carrying a = J v per row (instead of v per coordinate) removes every cross-lane
 * reduction from that chain:
 *
 *   lane b  <->  constraint block b, MSK_MAX_BLOCKS = 64 per env, in this order: joints with an active limit or a force-limited
 *                drive (rows: drive, lower limit, upper limit), joints with friction (one row), contact points (rows: normal,
 *                tangent 1, tangent 2), torsional rows of one-point manifolds (one row, its cone sized by the point's normal row).
 *                The lane keeps a, b = J dq, lambda, sum(lambda), c0, 1/A_rr of its <= 3 rows in VGPRs.
 *   build        the lane assembles its rows J (S_k . F against the LDS-resident motion subspace
 *                columns), Y = W J^T (W in LDS), parks Y in LDS, then walks all columns r and stores
 *                A[i][r] = J_i . Y_r for its rows i (nine contiguous words per (column block, row block): a sweep
 *                step reads them from one address with immediate offsets).
 *   sweep step   every lane evaluates the clamp for its own slot s (5 VALU), the owner's impulse change
 *                is broadcast (DPP row_newbcast / v_readlane), every lane does a[s'] += A[s'][r] * dl
 *                (3 FMA on operands prefetched one block ahead).  Chain: fma, max, min, sub, bcast, fma.
 *   finish       v = v* + Y^T lambda and dq = h (Np v* + Y^T sum lambda) by the first NVP lanes,
 *                impulses back to the contact slots, q/qd/qacc, free bodies.
 *
 * One launch, four LDS capacity classes (k_csolve).  The median env has 6 blocks and a whole wave per env would be
 * issue-bound on idle lanes, so class 0 (<= 13 blocks for nv <= 16) packs FOUR envs into a wavefront (16 lanes each, the
 * broadcast is one DPP row_newbcast) sharing an LDS pool for Y and A, carved after counting.  Classes 1..3 (<= 20, <= 32,
 * <= 64 blocks: fingers or links lying on the table) get one wavefront per env with v_readlane broadcast; class 3 keeps
 * its A image in global memory (L2) behind a four-deep register prefetch ring, so every workgroup of the launch needs
 * the same 46.5 KB of LDS.  The envs are sorted into the classes by the narrowphase (classify_envs) — exactly, from the
 * same block count the solver computes —, the one-env-per-wave workgroups come first in the grid and walk the classes
 * from the largest down: the longest solves start first, and they are what bounds the launch.
 * Arithmetic is the oracle's (oracle/orc_sim.c), operation for operation.
 */
#ifndef MSK_SOLVE_H
#define MSK_SOLVE_H

#include <type_traits>

#include "msk_model.h"

#define MSK_PEN_RATE_COEF 2.0f   /* penetration recovery: bias = depth * 2 sqrt(1 / dt) (oracle: ORC_PEN_RATE_COEF) */
#define MSK_MAX_DEPEN_VEL 3.0f
#define MSK_LIMIT_BACKSTOP 0.01f          /* a joint coordinate is never integrated further than this past a limit (oracle: ORC_LIMIT_BACKSTOP) */
#define MSK_MAX_JOINT_VELOCITY 100.0f   /* PhysX's default maxJointVelocity (oracle: ORC_MAX_JOINT_VELOCITY) */
/* a row whose own response J W J^T is below this cannot be moved by an impulse (two links with no relative freedom along the normal:
 * fixed-jointed siblings whose hulls overlap; round-off leaves ~1e-8, 1 / that turned one env of UnitreeG1TransportBox-v1 into NaNs):
 * it takes no impulse -- PhysX's minimal-response test (recipResponse = 0).  Lightest response a real body gives: 1 / (1e6 kg). */
#define MSK_MIN_RESPONSE 1.0e-6f
/* and a limit / normal row hands over at most this per sweep (N s): overlapping hulls of links that can barely move relative to each
 * other (response 1e-5: fingers of a hand at their zero pose) asked for 3e5 and more, and that momentum came back through the other
 * rows.  Three orders of magnitude above anything a manipulation task produces (a 100 kg body stopped from 5 m/s: 500). */
#define MSK_MAX_ROW_IMPULSE 1.0e3f
#define MSK_SMALL_BLOCKS 16

/* LDS hand-off inside a wavefront: DS instructions of one wave execute in issue order, so only the compiler has to be
 * stopped from moving the accesses — a wavefront-scope fence would also drain the global loads / stores in flight */
MSK_DEV void wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
/* global memory written by some lanes of the wavefront is read by others behind this point (the pattern of k_pickcube_observe_kin and of
 * k_dynamics' broadphase tail: release, wave barrier, acquire at workgroup scope -- the workgroup shares one L1 --, plus the explicit wait
 * for the stores) */
MSK_DEV void wave_global_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  MSK_WAIT_VMCNT0();
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
MSK_DEV float readlane_f(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }

template <int CTRL>
MSK_DEV float dpp_mov(float x) { /* every source lane of a row broadcast exists: `old` is never used */
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xF, 0xF, false));
}
/* value of lane `blk` of my GL-lane group, in every lane of the group (blk is wave-uniform) */
template <int GL>
MSK_DEV float group_bcast(float x, int blk) {
  if (GL == 64) return readlane_f(x, blk);
  if (GL == 32) return __shfl(x, blk, 32);
  switch (blk) { /* row_newbcast:n — folds to one v_mov_dpp once the caller's loop is unrolled */
    case 0: return dpp_mov<0x150>(x); case 1: return dpp_mov<0x151>(x); case 2: return dpp_mov<0x152>(x);
    case 3: return dpp_mov<0x153>(x); case 4: return dpp_mov<0x154>(x); case 5: return dpp_mov<0x155>(x);
    case 6: return dpp_mov<0x156>(x); case 7: return dpp_mov<0x157>(x); case 8: return dpp_mov<0x158>(x);
    case 9: return dpp_mov<0x159>(x); case 10: return dpp_mov<0x15A>(x); case 11: return dpp_mov<0x15B>(x);
    case 12: return dpp_mov<0x15C>(x); case 13: return dpp_mov<0x15D>(x); case 14: return dpp_mov<0x15E>(x);
    default: return dpp_mov<0x15F>(x);
  }
}

/* inclusive prefix sum of x over my GL-lane group, and the group's total (in every lane) */
template <int GL>
MSK_DEV void group_scan(int x, int* incl, int* total) {
  if (GL == 16) { /* a DPP row: four shifted adds (lanes shifted in from outside the row read 0), the total is lane 15 of the row */
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   /* row_shr:8 */
    *incl = x;
    *total = __builtin_amdgcn_update_dpp(x, x, 0x15F, 0xF, 0xF, false);   /* row_newbcast:15 */
  } else {
    const int l = threadIdx.x % GL;
#pragma unroll
    for (int d = 1; d < GL; d <<= 1) {
      const int o = __shfl_up(x, d, GL);
      if (l >= d) x += o;
    }
    *incl = x;
    *total = __shfl(x, GL - 1, GL);
  }
}

#define MSK_CLASS0_REG_BLOCKS 16   /* packed class of the 16-coordinate kernels: every lane keeps its A entries (9 per column block) in registers */
#define MSK_CLASS1_BLOCKS 20   /* capacity classes of the one-env-per-wave images */
#define MSK_CLASS2_BLOCKS 32
#define MSK_CLASS3_BLOCKS 64   /* one lane per block: MSK_MAX_BLOCKS */
/* LDS image of one workgroup: 64/GL envs, each with a fixed part, plus one pool for Y (and A).
 * NVP = 16 (one Panda + <= 1 free body: the benchmark configurations): the packed class (GL = 16, four envs per wave) keeps A in
 * REGISTERS and only parks Y in LDS (a fixed region of GL x 3 rows per env); class 1 (<= 20 blocks, one env per wave) has its A image
 * in LDS; larger envs keep it in global memory.  All three need <= 22 KB per workgroup, so seven workgroups share a CU and the whole
 * launch is resident at once (it took two rounds at three workgroups per CU with the 46.5 KB image of round 2).
 * NVP = 32 (Fetch, two arms, cabinets): the round-2 layout, A in LDS up to MSK_CLASS2_BLOCKS blocks.  NVP = 64: the same, GL = 64. */
constexpr int cs_fix_words(int nvp, int gl) {
  return ((nvp * nvp + nvp * 8 + nvp + 2 * nvp + 6 * gl + 3 * (gl < MSK_MAX_BLOCKS ? gl : MSK_MAX_BLOCKS) + 3) / 4) * 4;
}
constexpr int cs_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
template <int NVP, int GL, int CAP>
struct CsLds {
  static constexpr bool AREG = (NVP == 16 && GL == 16);
  static constexpr int EPW = 64 / GL;               /* envs per wavefront                              */
  static constexpr int COLS = 3 * GL;
  static constexpr int NDESC = GL < MSK_MAX_BLOCKS ? GL : MSK_MAX_BLOCKS;   /* (an env of these classes has at most GL blocks) */
  static constexpr int W = 0;                       /* [NVP][NVP]                                   */
  static constexpr int SC = W + NVP * NVP;          /* [NVP][8]  motion subspace columns            */
  static constexpr int VF = SC + NVP * 8;           /* [NVP]     v*                                 */
  static constexpr int VD = VF + NVP;               /* [2 NVP]   v | dq for the integration         */
  static constexpr int LAMF = VD + 2 * NVP;         /* [COLS]    lambda per row (lambda_0, then final) */
  static constexpr int LAMS = LAMF + COLS;          /* [COLS]    sum of lambda over position sweeps */
  static constexpr int DESC = LAMS + COLS;          /* int [NDESC] contact blocks: pair*4 + point     */
  static constexpr int TDESC = DESC + NDESC;        /* int [NDESC] torsional blocks: pair             */
  static constexpr int TREF = TDESC + NDESC;        /* int [NDESC] ... and the contact block (index among the contact blocks) of the pair's point */
  static constexpr int FIX = cs_fix_words(NVP, GL);
  static_assert(FIX >= TREF + NDESC, "fixed part");
  /* pool (floats): per env  Y [3 nblk][NVP]  then  A [3 nblk][3][nblk]  (A[(lane, s')][col]) */
  static constexpr int MAXBLK = AREG ? MSK_CLASS0_REG_BLOCKS : ((GL < 64) ? GL : CAP);
  static constexpr int need(int nb) { return nb * 3 * NVP + 9 * nb * nb; }
  static constexpr int YFIX = 3 * GL * NVP;         /* AREG: the env's fixed Y region (rows of idle lanes are zero) */
  /* one LDS size for every kind of workgroup of a launch */
  static constexpr int TOTAL16 = cs_max3(4 * cs_fix_words(16, 16) + 4 * 3 * 16 * 16,                      /* packed, A in registers */
                                         cs_fix_words(16, 64) + MSK_CLASS1_BLOCKS * 3 * 16 + 9 * MSK_CLASS1_BLOCKS * MSK_CLASS1_BLOCKS, /* class 1 */
                                         cs_fix_words(16, 64) + MSK_CLASS3_BLOCKS * 3 * 16);               /* A in global memory */
  static constexpr int TOTAL32 = cs_fix_words(32, 64) + MSK_CLASS2_BLOCKS * 3 * 32 + 9 * MSK_CLASS2_BLOCKS * MSK_CLASS2_BLOCKS;
  /* NVP = 64 (more than 32 generalized velocities: several free bodies next to an arm, humanoids): one env per wavefront in every class,
   * the 32-coordinate layout with wider W and Y (82 KB: one workgroup per SIMD pair; these scenes are few and large) */
  static constexpr int TOTAL64 = cs_fix_words(64, 64) + MSK_CLASS2_BLOCKS * 3 * 64 + 9 * MSK_CLASS2_BLOCKS * MSK_CLASS2_BLOCKS;
  static constexpr int TOTAL = (NVP == 16) ? TOTAL16 : ((NVP == 32) ? TOTAL32 : TOTAL64);
  static constexpr int POOL = TOTAL - EPW * FIX;
  /* packed launch: the block count up to which EPW envs always fit together */
  static constexpr int fit() {
    if (AREG) return MSK_CLASS0_REG_BLOCKS;
    int nb = 0;
    while (nb + 1 <= GL && EPW * need(nb + 1) <= POOL) ++nb;
    return nb;
  }
};


#define MSK_TRIM_CANDIDATES 512   /* oracle: ORC_TRIM_CANDIDATES */
/* An env with more contact points than it can take (capc): the capc DEEPEST are kept -- smallest separation, ties in (pair, point) order --
 * of the first MSK_TRIM_CANDIDATES in (pair, point) order (oracle: collide()).  Whole wavefront, one env (such an env is never in a packed
 * class); tmp: MSK_TRIM_CANDIDATES words of LDS.  The slots are compacted in place (points, separations, warm-start impulses) and their counts
 * rewritten: what the sweeps, the contact reports and the next step's warm start see is the trimmed set. */
MSK_DEV void trim_deepest(int* cnts, float* recs, const int np, const int capc, float* tmp) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int p0 = 0; p0 < np; p0 += 64) {
    const int p = p0 + lane;
    const int cnt = (p < np) ? cnts[p] : 0;
    int incl, tot;
    group_scan<64>(cnt, &incl, &tot);
    const int first = base + incl - cnt;
    for (int kk = 0; kk < cnt; ++kk)
      if (first + kk < MSK_TRIM_CANDIDATES) tmp[first + kk] = recs[(size_t)p * MSK_CT_REC + 16 + kk];
    base += tot;
  }
  wave_sync();
  const int n = base < MSK_TRIM_CANDIDATES ? base : MSK_TRIM_CANDIDATES;
  unsigned keep = 0u;   /* bit t: candidate lane + 64 t stays */
  for (int t = 0; t * 64 < n; ++t) {
    const int i = lane + 64 * t;
    const float si = (i < n) ? tmp[i] : 0.0f;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = tmp[j];
      rank += (sj < si || (sj == si && j < i)) ? 1 : 0;
    }
    if (i < n && rank < capc) keep |= 1u << t;
  }
  wave_sync();
  for (int t = 0; t * 64 < n; ++t)
    if (lane + 64 * t < n) ((int*)tmp)[lane + 64 * t] = (int)((keep >> t) & 1u);
  wave_sync();
  base = 0;
  for (int p0 = 0; p0 < np; p0 += 64) {
    const int p = p0 + lane;
    const int cnt = (p < np) ? cnts[p] : 0;
    int incl, tot;
    group_scan<64>(cnt, &incl, &tot);
    const int first = base + incl - cnt;
    if (cnt > 0) {
      float* rec = recs + (size_t)p * MSK_CT_REC;
      int w = 0;
      for (int kk = 0; kk < cnt; ++kk) {
        if (!(first + kk < MSK_TRIM_CANDIDATES && ((const int*)tmp)[first + kk] != 0)) continue;
        if (w != kk) {
          for (int a = 0; a < 3; ++a) { rec[4 + 3 * w + a] = rec[4 + 3 * kk + a]; rec[20 + 3 * w + a] = rec[20 + 3 * kk + a]; }
          rec[16 + w] = rec[16 + kk];
        }
        ++w;
      }
      cnts[p] = w;
    }
    base += tot;
  }
  wave_global_handoff();   /* the compacted slots are read by the lanes that own their blocks */
}

/* Sweep-invariant part of a row update: bias / A_rr, with the bias of a limit / normal row (penetration
 * recovery or approach speed from c0 + J.dq) or of a friction row (drift J.dq).  b = J.dq only changes
 * between sweeps, so every lane evaluates this once per sweep for its own rows, off the serial chain. */
template <bool POSIT, bool FRICTION>
MSK_DEV float bias_over_arr(float b, float c0, float rinv, float inv_h, float inv_dt, float pen_rate, float rest = 0.0f, float vclose = 0.0f) {
  float bias;
  if (!FRICTION) {
    const float cur = c0 + b;
    if (POSIT) bias = (cur > 0.0f) ? cur * inv_h : fmaxf(cur * pen_rate, -MSK_MAX_DEPEN_VEL);
    else bias = (cur > 0.0f) ? cur * inv_dt : 0.0f;
    /* restitution: a normal row that came in faster than bounce_threshold aims at the rebound speed (rest = e * J.v* < 0) once the gap
     * is closed (position sweeps) or would be eaten by the approach allowance of the next step (velocity sweep) */
    if (rest < 0.0f) {
      if (POSIT) { if (!(cur > 0.0f)) bias = fminf(bias, rest); }
      else if (cur < vclose) bias = rest;
    }
  } else {
    bias = POSIT ? b * inv_h : 0.0f;
  }
  return bias * rinv;
}

/* GL lanes per env (16: four envs per wave, 64: one); the envs are entries first .. first + 64/GL - 1 of `list`.
 * AGLOB (class 3, one env per wave): the A image lives in this workgroup's slice of st.a_scratch instead of LDS; every
 * lane reads back only what it wrote itself, and the sweeps prefetch four blocks ahead to cover the L2 round trip. */
template <int NVP, int GL, int CAP, bool AGLOB>
MSK_DEV void solve_env(const DModel* __restrict__ m, const DState& st, const int* __restrict__ list, const int first, const int count,
                       float* lds_all, const int wg) {   /* wg: this workgroup's index among the context's solver workgroups */
  typedef CsLds<NVP, GL, CAP> LY;
  const int g = threadIdx.x / GL, lane = threadIdx.x % GL;   /* `lane` = my block / coordinate inside the env */
  const int gshift = g * GL;
  const unsigned long long gmask = (GL == 64) ? ~0ull : ((1ull << GL) - 1ull);
  const bool in_range = first + g < count;
  const int e = list[in_range ? first + g : first];
  float* lds = lds_all + g * LY::FIX;
  float* pool = lds_all + LY::EPW * LY::FIX;
#define GBALLOT(pred) ((__ballot(pred) >> gshift) & gmask)
#ifdef MSK_PROFILE_PHASES
  long long tph[8]; int nph = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   /* the 100 MHz clock: where in the launch this wave ran (tools/gpu_phase_probe.py) */
#define PHASE() tph[nph++] = (long long)__builtin_readcyclecounter()
#else
#define PHASE()
#endif
  PHASE();
  const int nv = m->nv, nd = m->nd, np = m->np, npp = m->npp;
  const float dt = m->cfg.timestep;
  /* sweep schedule (oracle: orc_step_env): the T = Np + Nv configured sweeps are spent as nsub sub-steps of (biased sweep, advance of
   * the rows' positions, relaxing sweep) and nfinal relaxing sweeps: 15 + 1 -> 7 x (1 + 1) + 2 */
  const int T = (m->cfg.solver_position_iterations > 0 ? m->cfg.solver_position_iterations : 1) +
                (m->cfg.solver_velocity_iterations > 0 ? m->cfg.solver_velocity_iterations : 0);
  const int nsub = T >= 4 ? T / 2 - 1 : 1;
  const int nfinal = T - 2 * nsub > 0 ? T - 2 * nsub : 0;
  const float h = dt / (float)nsub;
  const float inv_h = 1.0f / h, inv_dt = 1.0f / dt, pen_rate = MSK_PEN_RATE_COEF * sqrtf(inv_dt);
  float* E = EREC(st, m, e);
  int* cnts = st.ct_cnt + (size_t)e * npp;
  float* recs = st.ct_rec + (size_t)e * npp * MSK_CT_REC;
  float* Lw = lds + LY::W;
  float* Lsc = lds + LY::SC;
  float* Lvf = lds + LY::VF;
  float* Lvd = lds + LY::VD;
  float* Llamf = lds + LY::LAMF;
  float* Llams = lds + LY::LAMS;
  int* Ldesc = (int*)(lds + LY::DESC);
  int* Ltdesc = (int*)(lds + LY::TDESC);
  int* Ltref = (int*)(lds + LY::TREF);

  /* ---- joint blocks: lane d owns dof d; a joint has a block when it is near a limit or its drive is a solver row (k_dynamics) ---- */
  float c_lo = 3.0e38f, c_hi = 3.0e38f, reach_lo = 0.0f, reach_hi = 0.0f;
  if (lane < nd) {
    const float lo = m->dof_lo[lane], hi = m->dof_hi[lane], q = E[m->lay.q + lane];
    if (!(lo < -1e30f && hi > 1e30f)) {
      const float vf = st.vfree[(size_t)e * NVP + lane];   /* a limit row exists while the joint can reach the limit within this step */
      c_lo = q - lo; c_hi = hi - q;
      reach_lo = fmaf(2.0f * dt, fmaxf(0.0f, -vf), MSK_LIMIT_SLACK);
      reach_hi = fmaf(2.0f * dt, fmaxf(0.0f, vf), MSK_LIMIT_SLACK);
    }
  }
  const unsigned long long drvm = st.drv_mask[e];
  const unsigned long long blo = GBALLOT(c_lo < reach_lo), bhi = GBALLOT(c_hi < reach_hi);
  const unsigned long long bdrv = GBALLOT(lane < nd && ((drvm >> lane) & 1ull));
  const unsigned long long bjoint = blo | bhi | bdrv;
  const int njoint = __popcll(bjoint);
  const int nfix = njoint + m->njfric;   /* ... followed by the joint-friction blocks, one per joint with a friction coefficient */
  const int room = m->cap_blocks - nfix > 0 ? m->cap_blocks - nfix : 0;
  const int capc = room < m->cap_contacts ? room : m->cap_contacts;   /* contact points this env can take */

  /* ---- contact points in canonical (pair, point) order, capacity capc; torsional rows of one-point manifolds ----------- */
  int base = 0, ntors_pre = 0, ntors_all = 0, base0 = 0, ntors_pre0 = 0;
  const bool any_tors = m->has_tors != 0;
  constexpr int PCH = 8;   /* the counts of PCH x GL pairs are fetched side by side: one global round trip instead of one per GL pairs */
  for (int pass = 0;; ++pass) { /* (a second pass only behind trim_deepest: an env over its capacity) */
  base = 0; ntors_pre = 0; ntors_all = 0;
  for (int pc = 0; pc < np; pc += PCH * GL) {
    int cbuf[PCH];
#pragma unroll
    for (int u = 0; u < PCH; ++u) {
      const int p = pc + u * GL + lane;
      cbuf[u] = (p < np) ? cnts[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < PCH; ++u) {
      const int p0 = pc + u * GL;
      if (p0 >= np) break;
      const int p = p0 + lane;
      int cnt = cbuf[u];
      int incl, tot;
      group_scan<GL>(cnt, &incl, &tot);
      const int first = base + incl - cnt;
      const bool tors_pair = any_tors && p < np && (m->pinfo[p < np ? p : 0].patch_r > 0.0f || m->pinfo[p < np ? p : 0].min_patch_r > 0.0f);
      if (any_tors) ntors_pre += __popcll(GBALLOT(tors_pair && cnt == 1));
      if (GL != 64 && first + cnt > capc) { /* (packed classes: an env over its capacity is never sorted into them) */
        const int keep = max(0, capc - first);
        if (cnt > 0) cnts[p] = keep;
        cnt = keep;
      }
      if (first + cnt <= LY::NDESC)
        for (int kk = 0; kk < cnt; ++kk) Ldesc[first + kk] = p * 4 + kk;
      if (any_tors) { /* a pair left with exactly one point and a patch radius: a torsional block behind the contact blocks, in pair order */
        const bool tors = tors_pair && cnt == 1;
        const unsigned long long tm = GBALLOT(tors);
        const int trank = ntors_all + __popcll(tm & ((1ull << lane) - 1ull));
        if (tors && trank < LY::NDESC) { Ltdesc[trank] = p; Ltref[trank] = first; }
        ntors_all += __popcll(tm);
      }
      base += tot;
    }
  }
  if (pass == 0) { base0 = base; ntors_pre0 = ntors_pre; }
  if (GL != 64 || pass == 1 || !(base > capc)) break;   /* (GL = 64: one env per wavefront, the test is wave-uniform) */
  static_assert(GL != 64 || LY::POOL >= MSK_TRIM_CANDIDATES, "trim_deepest's scratch fits the pool");
  trim_deepest(cnts, recs, np, capc, pool);              /* the pool is not in use yet */
  }
  bool overflow = base0 > capc;
  int ncont = overflow ? capc : base;
  if (in_range && lane == 0) { /* the running total the classification read (msk_kernels.h) must be this row's block sum; trimmed rows follow */
    if (st.ct_total[e] != base0 + ntors_pre0) atomicOr(st.env_overflow, 4);
    if (overflow) st.ct_total[e] = ncont + ntors_all;
  }
  int ntors = ntors_all < room - ncont ? ntors_all : room - ncont;   /* torsional rows get what the points leave */
  if (ntors > LY::NDESC) ntors = LY::NDESC;
  int nblk = nfix + ncont + ntors;
  /* carve the pool: groups in order (class 0 admits only block counts that fit together: CsLds::fit) */
  if (!in_range) nblk = 0;
  bool active = in_range;
  int pbase = 0;
  {
    int off = 0;
#pragma unroll
    for (int j = 0; j < LY::EPW; ++j) {
      const int nbj = __builtin_amdgcn_readlane(nblk, j * GL);
      const int need = LY::AREG ? LY::YFIX : (AGLOB ? nbj * 3 * NVP : LY::need(nbj));
      const bool fits = nbj <= LY::MAXBLK && off + need <= LY::POOL;
      if (j == g) { pbase = off; if (!fits) active = false; }
      if (fits) off += need;
    }
  }
  if (in_range && !active && lane == 0) atomicOr(st.env_overflow, 2); /* misclassified env: never expected */
  if (!active) nblk = 0;
  if (GL == 64) nblk = __builtin_amdgcn_readfirstlane(nblk); /* one env per wave: let the loops below run on scalar counters */
  if (active && lane == 0) {
    st.env_ncontacts[e] = ncont;
    if (overflow) atomicOr(st.env_overflow, 1);
  }
  const int nbmax = (GL == 64) ? nblk : max(max(__builtin_amdgcn_readlane(nblk, 0), __builtin_amdgcn_readlane(nblk, 16 % 64)),
                                           max(__builtin_amdgcn_readlane(nblk, 32 % 64), __builtin_amdgcn_readlane(nblk, 48 % 64)));
  if (nbmax == 0 && __ballot(active) == 0ull) return;
  float* Ly = pool + (LY::AREG ? g * LY::YFIX : pbase);
  float* La;
  if constexpr (AGLOB) La = st.a_scratch + (size_t)wg * (9 * CAP * (CAP + 4));
  else La = Ly + nblk * 3 * NVP;   /* (unused when the A image lives in registers) */
  const int nb = nblk > 0 ? nblk : 1; /* row stride of my A image */

  PHASE();
  /* ---- loads first: the env's solver tables (for LDS) and everything my block needs from global memory are requested side by side,
   * then the tables are parked in LDS, then the rows are computed: two global round trips on the critical path instead of six ---- */
  constexpr int NW4 = (NVP * NVP / 4 + GL - 1) / GL, NS4 = (NVP * 2 + GL - 1) / GL;
  float4 wreg[NW4], sreg[NS4];
  {
    const float4* wsrc = (const float4*)(st.W + (size_t)e * NVP * NVP);
    const float4* ssrc = (const float4*)(st.Scol + (size_t)e * NVP * 8);
#pragma unroll
    for (int u = 0; u < NW4; ++u) { const int i = lane + u * GL; wreg[u] = (i < NVP * NVP / 4) ? wsrc[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < NS4; ++u) { const int i = lane + u * GL; sreg[u] = (i < NVP * 2) ? ssrc[i] : make_float4(0, 0, 0, 0); }
  }
  const float vfreg = (lane < NVP) ? st.vfree[(size_t)e * NVP + lane] : 0.0f;

  /* ---- my block: what it reads from global memory ---------------------------------------------------------------
   * slot 0: the drive row of a joint block, the row of a joint-friction or torsional block, the normal row of a contact block:
   *         clamp [lo0, hi0] (torsional blocks: +- mu_r x the normal impulse of their point, fetched when they are swept)
   * slots 1, 2: the limit rows of a joint block ([0, cap]) or the tangential rows of a contact block (+- flim x lam[0]) */
  float J[3][NVP];
  float c0[3] = {0.0f, 0.0f, 0.0f}, lam[3] = {0.0f, 0.0f, 0.0f};
  bool valid[3] = {false, false, false};
  float mu = 0.0f, erest = 0.0f;
  float lo0 = 0.0f, hi0 = 0.0f, cfm0 = 0.0f, vb0 = 0.0f, mu_r = 0.0f;
  bool fixed_ct = false;   /* my block is a contact point against a body that no coordinate moves (static, kinematic, an unjointed link) */
  int code = -1; /* contact blocks: pair * 4 + point; torsional blocks: pair * 4 */
  int tref = 0;  /* torsional blocks: the lane of their point's contact block */
  int jd = -1;   /* joint and joint-friction blocks: my dof */
  typedef typename std::conditional<(NVP > 32), unsigned long long, unsigned>::type cmask_t;   /* one bit per coordinate */
  cmask_t coordsA = 0, coordsB = 0;   /* contact and torsional blocks: coordinates that move the two bodies (bit k) */
  v3 cn = v3_make(0, 0, 1), cpt = v3_make(0, 0, 0);   /* ... their normal, my contact point */
  float csep = 0.0f;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int k = 0; k < NVP; ++k) J[s][k] = 0.0f;
  const bool is_joint = lane < njoint && lane < nblk;
  const bool is_jfric = lane >= njoint && lane < nfix && lane < nblk;
  const bool is_contact = lane >= nfix && lane < nfix + ncont && lane < nblk;
  const bool is_tors = lane >= nfix + ncont && lane < nblk;
  if (is_joint) {
    unsigned long long mk = bjoint;
    for (int t = 0; t < lane; ++t) mk &= mk - 1ull;
    jd = __ffsll((long long)mk) - 1;
    valid[0] = (bdrv >> jd) & 1ull;
    valid[1] = (blo >> jd) & 1ull;
    valid[2] = (bhi >> jd) & 1ull;
    const float q = E[m->lay.q + jd];
    c0[1] = valid[1] ? q - m->dof_lo[jd] : 0.0f;   /* (a row that does not exist is swept along in the packed class: everything about it stays finite) */
    c0[2] = valid[2] ? m->dof_hi[jd] - q : 0.0f;
    if (valid[0]) { /* force-limited drive as a soft row: compliance, velocity bias, impulse limit (k_dynamics) */
      const float4 dr = *(const float4*)(st.drv + ((size_t)e * NVP + jd) * 4);
      cfm0 = dr.x; vb0 = dr.y; hi0 = dr.z; lo0 = -dr.z;
    }
  } else if (is_jfric) { /* joint friction: holds the joint velocity at zero with at most coefficient x |transmitted wrench| x dt */
    unsigned long long mk = m->jfric_mask;   /* 64 coordinates: bit d = dof d */
    for (int t = 0; t < lane - njoint; ++t) mk &= mk - 1ull;
    jd = __ffsll((long long)mk) - 1;
    const int body = m->dof_body[jd];
    const float* x = st.jforce + ((size_t)e * m->nb + body) * 6;
    const float mag = sqrtf(fmaf(x[0], x[0], fmaf(x[1], x[1], fmaf(x[2], x[2], fmaf(x[3], x[3], fmaf(x[4], x[4], x[5] * x[5]))))));
    valid[0] = true;
    hi0 = m->bodies[body].jfriction * mag * dt;
    lo0 = -hi0;
  } else if (is_contact || is_tors) {
    int p, kk = 0;
    if (is_contact) { code = Ldesc[lane - nfix]; p = code >> 2; kk = code & 3; }
    else { p = Ltdesc[lane - nfix - ncont]; code = p * 4; tref = nfix + Ltref[lane - nfix - ncont]; }
    const DPairInfo pi = m->pinfo[p];
    const float* rec = recs + (size_t)p * MSK_CT_REC;
    const float4 r0 = *(const float4*)rec;   /* normal, impulse of the torsional row */
    cn = v3_make(r0.x, r0.y, r0.z);
    csep = rec[16 + kk];
    coordsA = pi.ca; coordsB = pi.cb;
    if constexpr (NVP > 32) { coordsA |= (cmask_t)pi.ca_hi << 32; coordsB |= (cmask_t)pi.cb_hi << 32; }
    /* static friction until the pair slides (narrowphase: ct_slip) */
    const float mu_eff = (m->has_static && st.ct_slip[(size_t)e * npp + p]) ? pi.mu : pi.mu_s;
    if (is_contact) {
      cpt = v3_make(rec[4 + 3 * kk], rec[4 + 3 * kk + 1], rec[4 + 3 * kk + 2]);
      mu = mu_eff;
      erest = pi.rest;
      lo0 = 0.0f; hi0 = MSK_MAX_ROW_IMPULSE;
      fixed_ct = coordsA == 0 || coordsB == 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) { valid[s] = true; c0[s] = csep; lam[s] = rec[20 + 3 * kk + s]; }
    } else { /* relative spin about the normal of a one-point manifold */
      const float rp = fmaxf(pi.min_patch_r, sqrtf(fmaxf(0.0f, -csep) * pi.patch_r));   /* PhysX: the patch grows with the penetration */
      mu_r = mu_eff * rp;
      valid[0] = true;
      lam[0] = r0.w;
    }
  }

  /* ---- the env's solver tables into LDS -------------------------------------------------------------------------- */
  {
#pragma unroll
    for (int u = 0; u < NW4; ++u) { const int i = lane + u * GL; if (i < NVP * NVP / 4) ((float4*)Lw)[i] = wreg[u]; }
#pragma unroll
    for (int u = 0; u < NS4; ++u) { const int i = lane + u * GL; if (i < NVP * 2) ((float4*)Lsc)[i] = sreg[u]; }
    if (lane < NVP) Lvf[lane] = vfreg;
  }
  wave_sync();

  /* ---- my block: rows J -----------------------------------------------------------------------------------------
   * ucoords: the coordinates some row of the wave touches (wave-uniform): every loop over coordinates below skips the others -- their
   * entries of J are zero in every lane, so the skipped terms are exact zeros */
  cmask_t ucoords = 0;
  {
    const cmask_t mine = (is_joint || is_jfric) ? ((cmask_t)1 << jd) : ((is_contact || is_tors) ? (coordsA | coordsB) : (cmask_t)0);
#pragma unroll
    for (int k = 0; k < NVP; ++k)
      if (__ballot((mine >> k) & 1u)) ucoords |= (cmask_t)1 << k;
  }
  if (is_joint) {
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
      if (k == jd) { J[0][k] = valid[0] ? 1.0f : 0.0f; J[1][k] = valid[1] ? 1.0f : 0.0f; J[2][k] = valid[2] ? -1.0f : 0.0f; }
    }
  } else if (is_jfric) {
#pragma unroll
    for (int k = 0; k < NVP; ++k)
      if (k == jd) J[0][k] = 1.0f;
  }
  {
    /* contact blocks: F_s = [p x d_s; d_s] for the normal and the two tangents; torsional blocks: the couple [n; 0].
     * J_k = (+1 if coordinate k moves body A) + (-1 if it moves body B) times S_k . F: the sign factor is exact */
    v3 t1 = v3_make(0, 0, 0), t2 = t1;
    if (is_contact) msk_tangents(cn, &t1, &t2);
    sv6 F0, F1, F2;
    if (is_contact) {
      F0.a = v3_cross(cpt, cn); F0.l = cn;
      F1.a = v3_cross(cpt, t1); F1.l = t1;
      F2.a = v3_cross(cpt, t2); F2.l = t2;
    } else {
      F0.a = cn; F0.l = v3_make(0, 0, 0);
      F1 = sv6_zero(); F2 = sv6_zero();
    }
    const bool rows3 = is_contact, rows1 = is_contact || is_tors;
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
      if (!((ucoords >> k) & 1u)) continue;   /* wave-uniform */
      const float* sc = Lsc + k * 8;
      sv6 Sk;
      Sk.a = v3_make(sc[0], sc[1], sc[2]);
      Sk.l = v3_make(sc[3], sc[4], sc[5]);
      const float sgn = (float)((coordsA >> k) & 1u) - (float)((coordsB >> k) & 1u);
      if (rows1) J[0][k] = sgn * sv6_dot(Sk, F0);
      if (rows3) { J[1][k] = sgn * sv6_dot(Sk, F1); J[2][k] = sgn * sv6_dot(Sk, F2); }
    }
  }
  /* The friction frame follows the motion (oracle: orc_step_env, contact rows): two tangential rows clamped to +-mu lam_n each are a
   * pyramid -- sqrt(2) mu along the diagonal --, so when the point's unconstrained tangential velocity (u1, u2) = (J_1 . v*, J_2 . v*)
   * exceeds MSK_FRICTION_ALIGN_SPEED the two rows are rotated in the tangent plane until row 1 points along it.  (cos, sin) are parked in
   * this lane's LAMS words (free until the finish), where the impulses are rotated back into the frame msk_tangents() gives. */
  if (is_contact) {
    float u1 = 0.0f, u2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
      if (!((ucoords >> k) & 1u)) continue;
      u1 = fmaf(J[1][k], Lvf[k], u1);
      u2 = fmaf(J[2][k], Lvf[k], u2);
    }
    const float n2 = fmaf(u1, u1, u2 * u2);
    float fc = 1.0f, fs = 0.0f;
    if (n2 > MSK_FRICTION_ALIGN_SPEED * MSK_FRICTION_ALIGN_SPEED) {
      const float inv = 1.0f / sqrtf(n2);
      fc = u1 * inv; fs = u2 * inv;
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        if (!((ucoords >> k) & 1u)) continue;
        const float j1 = J[1][k], j2 = J[2][k];
        J[1][k] = fmaf(fc, j1, fs * j2);
        J[2][k] = fmaf(fc, j2, -(fs * j1));
      }
      const float l1 = lam[1], l2 = lam[2];
      lam[1] = fmaf(fc, l1, fs * l2);
      lam[2] = fmaf(fc, l2, -(fs * l1));
    }
    Llams[lane * 3 + 1] = fc;
    Llams[lane * 3 + 2] = fs;
  }
  PHASE();
  /* Y = W J^T, parked in LDS; lambda_0 published for the warm start.  (A image in registers: every lane of the group parks its rows --
   * zero for the lanes without a block --, the unrolled build below reads all column blocks up to the wave's largest env) */
  if (LY::AREG || lane < nblk) {
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
      float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f;
#pragma unroll
      for (int j = 0; j < NVP; ++j) {
        const float w = Lw[k * NVP + j];
        y0 = fmaf(w, J[0][j], y0);
        y1 = fmaf(w, J[1][j], y1);
        y2 = fmaf(w, J[2][j], y2);
      }
      Ly[(lane * 3 + 0) * NVP + k] = y0;
      Ly[(lane * 3 + 1) * NVP + k] = y1;
      Ly[(lane * 3 + 2) * NVP + k] = y2;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) Llamf[lane * 3 + s] = lam[s];
  }
  wave_sync();
  const unsigned long long vm0 = GBALLOT(valid[0]), vm1 = GBALLOT(valid[1]), vm2 = GBALLOT(valid[2]);
  /* rows that exist in at least one env of the wave: bit blk*3+s (wave-uniform, tested with scalar ops) */
  unsigned long long wrows = 0ull;   /* the packed class only (16 blocks x 3 bits) */
  if (GL == 16) {
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
      if (blk >= nbmax) break;
      if (__ballot((vm0 >> blk) & 1ull)) wrows |= 1ull << (blk * 3);
      if (__ballot((vm1 >> blk) & 1ull)) wrows |= 1ull << (blk * 3 + 1);
      if (__ballot((vm2 >> blk) & 1ull)) wrows |= 1ull << (blk * 3 + 2);
    }
  }

  PHASE();
  /* ---- constraint-space operator: A[(me, s')][col] = J_(me,s') . Y_col, per (column block, row block) nine contiguous words in LDS -------------------- */
  float av[3], bv[3] = {0.0f, 0.0f, 0.0f}, ls[3] = {0.0f, 0.0f, 0.0f}, rinv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
      if (!((ucoords >> k) & 1u)) continue;
      acc = fmaf(J[s][k], Lvf[k], acc);
    }
    av[s] = acc;
  }
  /* restitution bias of my normal row (the oracle's rows[i].rest): e * J.v* if the approach beats bounce_threshold */
  const bool bounces = is_contact && erest > 0.0f && av[0] < -m->cfg.bounce_threshold;
  const float rest0 = bounces ? erest * av[0] : 0.0f, vclose0 = bounces ? -av[0] * dt : 0.0f;
  constexpr int NREG = LY::AREG ? LY::MAXBLK : 1;
  float Areg[NREG][9];   /* packed class: my rows' nine entries of every column block, in registers (the loops over blocks are unrolled) */
  if constexpr (LY::AREG) {
#pragma unroll
    for (int blk = 0; blk < NREG; ++blk) {
#pragma unroll
      for (int i = 0; i < 9; ++i) Areg[blk][i] = 0.0f;
      if (blk < nbmax) { /* wave-uniform */
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int col = blk * 3 + s;
          if (!((wrows >> col) & 1ull)) continue;   /* a row no env of the wave has: its column is zero everywhere */
          /* (in an env that lacks the row, or the whole block, the parked Y row is zero: the dot products are exactly zero) */
          const float* ycol = Ly + col * NVP;
          float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
#pragma unroll
          for (int k = 0; k < NVP; ++k) {
            const float y = ycol[k];
            d0 = fmaf(J[0][k], y, d0);
            d1 = fmaf(J[1][k], y, d1);
            d2 = fmaf(J[2][k], y, d2);
          }
          Areg[blk][s * 3 + 0] = d0; Areg[blk][s * 3 + 1] = d1; Areg[blk][s * 3 + 2] = d2;
          if (lane == blk) { /* my own diagonal; a drive row is soft: its compliance adds to the response */
            if (s == 0) { const float arr = d0 + cfm0; rinv[0] = arr > MSK_MIN_RESPONSE ? 1.0f / arr : 0.0f; }
            if (s == 1) rinv[1] = d1 > MSK_MIN_RESPONSE ? 1.0f / d1 : 0.0f;
            if (s == 2) rinv[2] = d2 > MSK_MIN_RESPONSE ? 1.0f / d2 : 0.0f;
          }
          /* warm start: a += A[:, col] * lambda_0[col] (rows ascending == columns ascending) */
          const float l0 = Llamf[col];
          av[0] = fmaf(d0, l0, av[0]);
          av[1] = fmaf(d1, l0, av[1]);
          av[2] = fmaf(d2, l0, av[2]);
        }
      }
    }
  } else
  for (int blk = 0; blk < nblk; ++blk) { /* per-group trip count: the groups of a wave diverge here, no cross-lane ops inside */
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const unsigned long long vm = (s == 0) ? vm0 : ((s == 1) ? vm1 : vm2);
      const int col = blk * 3 + s;
      if (!((vm >> blk) & 1ull)) { /* a row that does not exist is an all-zero row: its column is zero */
        if (lane < nblk) {
          float* a = La + ((size_t)blk * nb + lane) * 9 + s * 3;
          a[0] = 0.0f; a[1] = 0.0f; a[2] = 0.0f;
        }
        continue;
      }
      const float* ycol = Ly + col * NVP;
      float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        const float y = ycol[k];
        d0 = fmaf(J[0][k], y, d0);
        d1 = fmaf(J[1][k], y, d1);
        d2 = fmaf(J[2][k], y, d2);
      }
      if (lane < nblk) {
        float* a = La + ((size_t)blk * nb + lane) * 9 + s * 3;
        a[0] = d0; a[1] = d1; a[2] = d2;
      }
      if (lane == blk) { /* my own diagonal; a drive row is soft: its compliance adds to the response */
        if (s == 0) { const float arr = d0 + cfm0; rinv[0] = arr > MSK_MIN_RESPONSE ? 1.0f / arr : 0.0f; }
        if (s == 1) rinv[1] = d1 > MSK_MIN_RESPONSE ? 1.0f / d1 : 0.0f;
        if (s == 2) rinv[2] = d2 > MSK_MIN_RESPONSE ? 1.0f / d2 : 0.0f;
      }
      /* warm start: a += A[:, col] * lambda_0[col] (rows ascending == columns ascending) */
      const float l0 = Llamf[col];
      av[0] = fmaf(d0, l0, av[0]);
      av[1] = fmaf(d1, l0, av[1]);
      av[2] = fmaf(d2, l0, av[2]);
    }
  }
  wave_sync();
  const float keep0 = fmaf(-cfm0, rinv[0], 1.0f);   /* what a soft row keeps of its impulse in an update; exactly 1 for rigid rows */

  PHASE();
  /* ---- Gauss-Seidel sweeps ----------------------------------------------------------------------------------------- */
  /* my rows' entries of the three columns of block blk.  Lanes past their env's last block re-read its last
   * block: the impulse change they receive for such a step is exactly zero (zero rows), so any finite value does */
  const int lrow = lane < nblk ? lane : 0;
  auto load_cols = [&](int blk, float* dst) {
    /* one env per wave: a prefetch past the last block reads words nobody consumes (LDS: in range of the pool or zero;
     * global image: one block of slack behind every worker's slice), so the index needs no clamp */
    const int bc = (GL == 64) ? blk : (blk < nblk ? blk : (nblk > 0 ? nblk - 1 : 0));
#pragma unroll
    for (int i = 0; i < 9; ++i) dst[i] = La[(bc * nb + lrow) * 9 + i];   /* one address, nine immediate offsets */
  };
  /* blocks that are a torsional block in at least one env of the wave (bit = block index inside the env) */
  unsigned long long tblocks = 0ull;
  if (any_tors) {
    const unsigned long long tb = __ballot(is_tors);
    tblocks = (GL == 64) ? tb : ((GL == 32) ? ((tb | (tb >> 32)) & 0xFFFFFFFFull) : ((tb | (tb >> 16) | (tb >> 32) | (tb >> 48)) & 0xFFFFull));
  }
  /* clamp bounds of slots 1, 2 without selects: hi = fma(flim, lam_n, hi_c), lo = fma(-flim, lam_n, 0)
   *   contact block: flim = mu, hi_c = 0 -> +-mu*lam_n ; joint block (limit rows): flim = 0, hi_c = cap -> [0, cap] */
  const float flim = is_contact ? mu : 0.0f;
  const float hi_c = is_contact ? 0.0f : MSK_MAX_ROW_IMPULSE;
  auto sweep = [&](auto posit_tag, auto tors_tag) {
    constexpr bool POSIT = decltype(posit_tag)::value;
    constexpr bool TORS = decltype(tors_tag)::value;      /* some env of the wavefront has a torsional block; without one (most templates) the per-block test of `tblocks` -- a
                                                           * scalar branch and two moves in every block of the unrolled stream -- is not compiled into the sweep */
    /* sweep-invariant bias terms of my rows.  slot 0: a contact's normal row (position bias / restitution), or the constant velocity bias
     * of a drive row (0 for joint-friction and torsional rows); slots 1, 2: limit rows of a joint block, friction rows of a contact block */
    const float t0n = bias_over_arr<POSIT, false>(bv[0], c0[0], rinv[0], inv_h, inv_dt, pen_rate, rest0, vclose0);
    const float t0 = is_contact ? t0n : vb0 * rinv[0];
    const float t1f = bias_over_arr<POSIT, true>(bv[1], c0[1], rinv[1], inv_h, inv_dt, pen_rate);
    const float t1n = bias_over_arr<POSIT, false>(bv[1], c0[1], rinv[1], inv_h, inv_dt, pen_rate);
    const float t1 = is_contact ? t1f : t1n;
    const float t2f = bias_over_arr<POSIT, true>(bv[2], c0[2], rinv[2], inv_h, inv_dt, pen_rate);
    const float t2n = bias_over_arr<POSIT, false>(bv[2], c0[2], rinv[2], inv_h, inv_dt, pen_rate);
    const float t2 = is_contact ? t2f : t2n;
    /* one block step; Ac = my rows' nine entries of the three columns of block blk (already in registers) */
    /* clamp(x, lo, hi) as ONE instruction on the serial chain: v_med3_f32 == fminf(fmaxf(x, lo), hi) whenever lo <= hi (always: the bounds
     * are +-(a coefficient >= 0) x (a normal impulse >= 0), [0, cap] or +-(a limit >= 0)) and x is no NaN */
    auto clampf = [](float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); };
    auto block_steps = [&](const int blk, const float* Ac, auto all_rows_tag) {
      const bool owner = lane == blk;
      /* the packed class sweeps all three rows of every block: a row that does not exist has rinv = lam = 0 and a zero column, so its
       * update is an exact no-op -- cheaper than a scalar branch per row in the unrolled stream */
      unsigned rowbits;
      /* (measured, round 5 A/B on one box: skipping the rows no env of the wave has with a scalar bit test each -- fewer steps on the serial chain -- makes
       * the packed class SLOWER, k_csolve 58.0 -> 69.3 us: three scalar branches per block cost more than the no-op row updates they save;
       * profiles/r05_ab_head_vs_e1_vs_c73dab6.log) */
      if (decltype(all_rows_tag)::value || LY::AREG) rowbits = 7u;
      else if (GL == 64) rowbits = (unsigned)(((vm0 >> blk) & 1ull) | (((vm1 >> blk) & 1ull) << 1) | (((vm2 >> blk) & 1ull) << 2));
      else if (GL == 16) rowbits = (unsigned)((wrows >> (blk * 3)) & 7ull);
      else rowbits = (__ballot((vm0 >> blk) & 1ull) ? 1u : 0u) | (__ballot((vm1 >> blk) & 1ull) ? 2u : 0u) | (__ballot((vm2 >> blk) & 1ull) ? 4u : 0u);
      /* new impulse = clamp(lam * keep - (J.v + bias) / (A_rr + cfm)); a row that does not exist has rinv = lam = 0 -> stays 0 */
      /* the normal impulse that sizes the friction cone of rows 1, 2: the owner's is the value row 0 just took; the other lanes' candidates for this block are
       * never broadcast, so their bounds may come from anything finite -- their own candidate: no select on the way to the clamp */
      float l0 = lam[0];
      if (rowbits & 1u) {
        float lo = lo0, hi = hi0;
        if (TORS && !decltype(all_rows_tag)::value && ((tblocks >> blk) & 1ull)) { /* a torsional block: its cone is sized by its point's normal impulse */
          const float lref = __shfl(lam[0], tref, GL);
          if (is_tors) { hi = mu_r * lref; lo = -hi; }
        }
        const float nl = clampf(fmaf(-av[0], rinv[0], fmaf(lam[0], keep0, -t0)), lo, hi);
        const float dl = group_bcast<GL>(nl - lam[0], blk);
        if (owner) lam[0] = nl;
        l0 = nl;
        av[0] = fmaf(Ac[0], dl, av[0]); av[1] = fmaf(Ac[1], dl, av[1]); av[2] = fmaf(Ac[2], dl, av[2]);
      }
      if (rowbits & 2u) {
        const float hi = fmaf(flim, l0, hi_c), lo = fmaf(-flim, l0, 0.0f);
        const float nl = clampf(fmaf(-av[1], rinv[1], lam[1] - t1), lo, hi);
        const float dl = group_bcast<GL>(nl - lam[1], blk);
        if (owner) lam[1] = nl;
        av[0] = fmaf(Ac[3], dl, av[0]); av[1] = fmaf(Ac[4], dl, av[1]); av[2] = fmaf(Ac[5], dl, av[2]);
      }
      if (rowbits & 4u) {
        const float hi = fmaf(flim, l0, hi_c), lo = fmaf(-flim, l0, 0.0f);
        const float nl = clampf(fmaf(-av[2], rinv[2], lam[2] - t2), lo, hi);
        const float dl = group_bcast<GL>(nl - lam[2], blk);
        if (owner) lam[2] = nl;
        av[0] = fmaf(Ac[6], dl, av[0]); av[1] = fmaf(Ac[7], dl, av[1]); av[2] = fmaf(Ac[8], dl, av[2]);
      }
    };
    if constexpr (LY::AREG) { /* unrolled: the block index becomes the immediate of row_newbcast, A comes from registers */
#pragma unroll
      for (int blk = 0; blk < NREG; ++blk) {
        if (blk >= nbmax) break;
        block_steps(blk, Areg[blk], std::false_type{});
      }
    } else if (GL == 16) { /* unrolled: the block index becomes the immediate of row_newbcast, the copies are renames */
      float Ac[9], An[9];
      load_cols(0, Ac);
#pragma unroll
      for (int blk = 0; blk < 16; ++blk) {
        if (blk >= nbmax) break;
        load_cols((blk + 1 < nbmax) ? blk + 1 : 0, An);
        block_steps(blk, Ac, std::false_type{});
#pragma unroll
        for (int i = 0; i < 9; ++i) Ac[i] = An[i];
      }
    } else { /* one env: joint and joint-friction blocks first (rows as present), then the contact blocks, whose three rows all exist, then the
              * torsional blocks (slot 0 only) */
      /* [nl0, nc1): the blocks that are contact blocks in every env of the wave (two envs when GL = 32) */
      int nl0 = __builtin_amdgcn_readfirstlane(nfix < nbmax ? nfix : nbmax);
      int nc1 = __builtin_amdgcn_readfirstlane(nfix + ncont);
      if (GL == 32) {
        const int nfix1 = __builtin_amdgcn_readlane(nfix, 32), end1 = __builtin_amdgcn_readlane(nfix + ncont, 32);
        nl0 = max(nl0, nfix1 < nbmax ? nfix1 : nbmax);
        nc1 = min(nc1, end1);
      }
      nc1 = max(nl0, min(nc1, nbmax));
      {
        float Ac[9], An[9];
        load_cols(0, Ac);
        for (int blk = 0; blk < nl0; ++blk) {
          load_cols(blk + 1, An);
          block_steps(blk, Ac, std::false_type{});
#pragma unroll
          for (int i = 0; i < 9; ++i) Ac[i] = An[i];
        }
      }
      /* ring of D register sets: at step k the set of block k + D - 1 is requested, the set of block k consumed */
      constexpr int D = AGLOB ? 4 : 2;
      float R[D][9];
#pragma unroll
      for (int u = 0; u + 1 < D; ++u) load_cols(nl0 + u, R[u]);
      for (int blk = nl0; blk < nc1; blk += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          if (blk + u < nc1) {
            load_cols(blk + u + D - 1, R[(u + D - 1) % D]);
            block_steps(blk + u, R[u], std::true_type{});
          }
        }
      }
      if (nc1 < nbmax) {
        float Ac[9];
        for (int blk = nc1; blk < nbmax; ++blk) {
          load_cols(blk, Ac);
          block_steps(blk, Ac, std::false_type{});
        }
      }
    }
    if (POSIT) {
      /* Static geometry has the last word before positions move (oracle: ORC_STATIC_LAST_WORD).  The normal rows against bodies that nothing
       * moves are visited once more in block order; each may only ADD impulse, and only what stops the approach (gap / h for an open gap,
       * zero otherwise).  A row wants something iff its candidate from the current `a` exceeds its impulse; if no row of the wave does, none
       * would after the others' (absent) updates either, and the pass is skipped exactly. */
      const bool cand_row = fixed_ct && !(rest0 < 0.0f);
      auto last_word = [&]() {
        const float cur = c0[0] + bv[0];
        const float bias = (cur > 0.0f) ? cur * inv_h : 0.0f;
        return clampf(fmaf(-av[0], rinv[0], fmaf(lam[0], keep0, -(bias * rinv[0]))), 0.0f, MSK_MAX_ROW_IMPULSE);
      };
      if (__ballot(cand_row && last_word() > lam[0]) != 0ull) {
        const unsigned long long cand = __ballot(cand_row);
        const unsigned long long candg = (GL == 64) ? cand : ((GL == 32) ? ((cand | (cand >> 32)) & 0xFFFFFFFFull) : ((cand | (cand >> 16) | (cand >> 32) | (cand >> 48)) & 0xFFFFull));
        auto word_step = [&](const int blk, const float* Ac) {
          const float nl = last_word();
          const bool add = cand_row && nl > lam[0];
          const float dl = group_bcast<GL>(add ? nl - lam[0] : 0.0f, blk);
          if (lane == blk && add) lam[0] = nl;
          av[0] = fmaf(Ac[0], dl, av[0]); av[1] = fmaf(Ac[1], dl, av[1]); av[2] = fmaf(Ac[2], dl, av[2]);
        };
        if constexpr (LY::AREG || GL == 16) { /* unrolled like the sweeps: the block index is the immediate of row_newbcast, A a register name;
                                               * a block without a candidate row in any env of the wave costs one scalar bit test */
#pragma unroll
          for (int blk = 0; blk < (LY::AREG ? NREG : 16); ++blk) {
            if (blk >= nbmax) break;
            if (!((candg >> blk) & 1ull)) continue;
            if constexpr (LY::AREG) word_step(blk, Areg[blk]);
            else { float Ac[9]; load_cols(blk, Ac); word_step(blk, Ac); }
          }
        } else {
          for (int blk = 0; blk < nbmax; ++blk) {
            if (!((candg >> blk) & 1ull)) continue;      /* no env of the wave has such a row at this block */
            float Ac[9];
            load_cols(blk, Ac);
            word_step(blk, Ac);
          }
        }
      }
      /* the sub-step's advance: the rows' positions move on with the biased velocity, the sub-step's impulse is booked */
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        bv[s] = fmaf(h, av[s], bv[s]);
        ls[s] += lam[s];
      }
    }
  };
  if (tblocks != 0ull) {      /* (wave-uniform: a ballot) */
    for (int sb = 0; sb < nsub; ++sb) {
      sweep(std::true_type{}, std::true_type{});
      sweep(std::false_type{}, std::true_type{});
    }
    for (int it = 0; it < nfinal; ++it) sweep(std::false_type{}, std::true_type{});
  } else {
    for (int sb = 0; sb < nsub; ++sb) {
      sweep(std::true_type{}, std::false_type{});
      sweep(std::false_type{}, std::false_type{});
    }
    for (int it = 0; it < nfinal; ++it) sweep(std::false_type{}, std::false_type{});
  }

  PHASE();
  /* ---- back to generalized coordinates ----------------------------------------------------------------------------------- */
  if (lane < nblk) {
    const float fc = is_contact ? Llams[lane * 3 + 1] : 1.0f, fs = is_contact ? Llams[lane * 3 + 2] : 0.0f;   /* parked at the rows (friction frame) */
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      Llamf[lane * 3 + s] = lam[s];
      Llams[lane * 3 + s] = ls[s];
    }
    if (is_contact) { /* impulses back to the contact slot (reports + next step's warm start), the friction pair in the frame of msk_tangents() */
      float* rec = recs + (size_t)(code >> 2) * MSK_CT_REC + 20 + (code & 3) * 3;
      rec[0] = lam[0];
      rec[1] = fmaf(fc, lam[1], -(fs * lam[2]));
      rec[2] = fmaf(fs, lam[1], fc * lam[2]);
    }
    if (is_tors) recs[(size_t)(code >> 2) * MSK_CT_REC + 3] = lam[0];
  }
  wave_sync();
  float v = 0.0f, dq = 0.0f;
  if (lane < NVP) {
    const float vf = Lvf[lane];
    float vk = vf, sk = (float)nsub * vf;
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned long long vm = (s == 0) ? vm0 : ((s == 1) ? vm1 : vm2);
        if (!((vm >> blk) & 1ull)) continue;
        const int col = blk * 3 + s;
        const float y = Ly[col * NVP + lane];
        vk = fmaf(y, Llamf[col], vk);
        sk = fmaf(y, Llams[col], sk);
      }
    }
    v = vk;
    dq = h * sk;
    if (lane < nd && !m->dof_body_is_root[lane]) { /* PhysX's maxJointVelocity: joint coordinates only (a floating root's six are a body's velocity) */
      v = fminf(fmaxf(v, -MSK_MAX_JOINT_VELOCITY), MSK_MAX_JOINT_VELOCITY);
      dq = fminf(fmaxf(dq, -MSK_MAX_JOINT_VELOCITY * dt), MSK_MAX_JOINT_VELOCITY * dt);
      const float lo = m->dof_lo[lane], hi = m->dof_hi[lane];
      if (!(lo < -1e30f && hi > 1e30f)) { /* the backstop behind the limit rows (oracle: ORC_LIMIT_BACKSTOP) */
        const float q0 = E[m->lay.q + lane], qn = q0 + dq;
        if (qn > hi + MSK_LIMIT_BACKSTOP) { dq = (hi + MSK_LIMIT_BACKSTOP) - q0; v = fminf(v, 0.0f); }
        else if (qn < lo - MSK_LIMIT_BACKSTOP) { dq = (lo - MSK_LIMIT_BACKSTOP) - q0; v = fmaxf(v, 0.0f); }
      }
    }
    Lvd[lane] = v;
    Lvd[NVP + lane] = dq;
  }
  wave_sync();

  /* ---- integrate ------------------------------------------------------------------------------------------------------------ */
  if (active && lane < nd) {
    const float q = E[m->lay.q + lane], qd = E[m->lay.qd + lane];
    E[m->lay.qacc + lane] = (v - qd) / dt;
    E[m->lay.q + lane] = m->dof_body_is_root[lane] ? 0.0f : q + dq;   /* a floating root's six slots carry no position: its pose does */
    E[m->lay.qd + lane] = v;
  }
  const int fr = (active && lane < nd) ? m->coord_root[lane] : -1;
  if (fr >= 0) { /* floating root: integrated like the free bodies below (its six coordinates are v of its centre of mass, omega) */
    const int k = lane;
    const v3 dx = v3_make(Lvd[NVP + k], Lvd[NVP + k + 1], Lvd[NVP + k + 2]);
    const v3 dr = v3_make(Lvd[NVP + k + 3], Lvd[NVP + k + 4], Lvd[NVP + k + 5]);
    const v3 cw = v3_add(load_v3(E, m->lay.comw, fr), dx);
    pose T = load_pose(E, m->lay.bpose, fr);
    const quat qn = quat_normalize(quat_mul(quat_from_rotvec(dr), T.q));
    T.q = qn;
    T.p = v3_sub(cw, quat_rotate(qn, m->bodies[fr].com));
    store_pose(E, m->lay.bpose, fr, T);
  }
  const int fb = (active && lane < nv) ? m->coord_body[lane] : -1;
  if (fb >= 0) {
    const int k = lane;
    const DBody* b = &m->bodies[fb];
    const v3 dx = v3_make(Lvd[NVP + k], Lvd[NVP + k + 1], Lvd[NVP + k + 2]);
    const v3 dr = v3_make(Lvd[NVP + k + 3], Lvd[NVP + k + 4], Lvd[NVP + k + 5]);
    const v3 cw = v3_add(load_v3(E, m->lay.comw, fb), dx);
    pose T = load_pose(E, m->lay.bpose, fb);
    const quat qn = quat_normalize(quat_mul(quat_from_rotvec(dr), T.q));
    T.q = qn;
    T.p = v3_sub(cw, quat_rotate(qn, b->com));
    store_pose(E, m->lay.bpose, fb, T);
    store_v3(E, m->lay.blin, fb, v3_make(Lvd[k], Lvd[k + 1], Lvd[k + 2]));
    store_v3(E, m->lay.bang, fb, v3_make(Lvd[k + 3], Lvd[k + 4], Lvd[k + 5]));
  }
  PHASE();
#ifdef MSK_PROFILE_PHASES
  if (active && lane == 0) {
    for (int i = 0; i < 7; ++i) st.dbg[(size_t)e * 8 + i] = tph[i];
    st.dbg[(size_t)e * 8 + 7] = (long long)(((__builtin_amdgcn_s_memrealtime() & 0xffffffffull) << 32) | (rt0 & 0xffffffffull));
  }
#endif
#undef PHASE
#undef GBALLOT
}

#include "msk_solve_wide.h"

/* Every class in ONE launch (no cross-stream joins): workgroups 0 .. gm-1 walk the one-env-per-wave classes, 3 before 2
 * before 1 (longest solves first: they are dispatched first and bound the launch), the rest take 64/GL consecutive
 * class-0 envs each.  All kinds use the same LDS bytes; class 3 (more than MSK_CLASS2_BLOCKS blocks: rare) keeps its
 * A image in global memory. */
template <int NVP, int GL>
MSK_DEV void csolve_block(const DModel* __restrict__ m, const DState& st, const int gm, const int blk, float* lds) {
  constexpr int CAPL = (NVP == 16) ? MSK_CLASS1_BLOCKS : MSK_CLASS2_BLOCKS;   /* largest one-env-per-wave image that lives in LDS */
  static_assert(CsLds<NVP, GL, GL>::TOTAL == CsLds<NVP, 64, CAPL>::TOTAL, "one LDS size for both kinds of workgroup");
  static_assert(CsLds<NVP, 64, CAPL>::need(CAPL) <= CsLds<NVP, 64, CAPL>::POOL, "the LDS image of the largest in-LDS class fits the pool");
  static_assert(MSK_CLASS3_BLOCKS * 3 * NVP <= CsLds<NVP, 64, CAPL>::POOL, "Y of the largest env fits the pool");
  static_assert(!CsLds<NVP, GL, GL>::AREG || CsLds<NVP, GL, GL>::EPW * CsLds<NVP, GL, GL>::YFIX <= CsLds<NVP, GL, GL>::POOL, "the packed class's Y regions fit");
  if (blk == 0 && threadIdx.x == 0) *st.hq_count = 0;   /* the narrowphase has consumed the hull queue; the next broadphase refills it */
  if (blk < gm) {
    const int n3 = st.cls_count[3], n2 = st.cls_count[2], n1 = st.cls_count[1];
    const size_t N = (size_t)m->N;
    for (int i = blk; i < n3 + n2 + n1; i += gm) {
      if (i < n3) solve_env<NVP, 64, MSK_CLASS3_BLOCKS, true>(m, st, st.cls_list + 3 * N, i, n3, lds, blk);
      else if (i < n3 + n2) {
        if constexpr (NVP == 16) solve_env<NVP, 64, MSK_CLASS3_BLOCKS, true>(m, st, st.cls_list + 2 * N, i - n3, n2, lds, blk);   /* 21 .. 32 blocks: A in global memory too */
        else solve_env<NVP, 64, MSK_CLASS2_BLOCKS, false>(m, st, st.cls_list + 2 * N, i - n3, n2, lds, blk);
      } else solve_env<NVP, 64, CAPL, false>(m, st, st.cls_list + N, i - n3 - n2, n1, lds, blk);
      wave_sync();
    }
  } else {
    const int count = st.cls_count[0], first = (blk - gm) * (64 / GL);
    if (first >= count) return;
    solve_env<NVP, GL, GL, false>(m, st, st.cls_list, first, count, lds, blk);
  }
}
/* The wide class (msk_solve_wide.h), only for contexts created with msk_config.contact_capacity = 1: worker w of `workers` takes entries
 * w, w + workers, ... of the class list with its slice of DState::wide_scratch.  Two blocks per lane need twice the registers of the packed
 * class: with 32 and 64 coordinates that would halve the occupancy of every other class inside k_csolve (AGPRs, scratch), so there it is a
 * launch of its own behind k_csolve (k_csolve_wide).  With 16 coordinates it fits k_csolve's own budget (242 against 251 VGPRs, 8 against
 * 22 KB of LDS) and its workers are the FIRST `ww` workgroups of k_csolve's grid: an almost always empty launch per substep less, and when
 * there are wide envs -- the longest solves of all -- they start first. */
template <int NVP>
MSK_DEV void csolve_wide_block(const DModel* __restrict__ m, const DState& st, const int w, float* lds) {
  static_assert(CsWide<NVP>::TOTAL * sizeof(float) <= 48 * 1024, "the wide class's LDS image needs no opt-in");
  if (w >= st.wide_workers) return;
  const int n4 = st.cls_count[MSK_SOLVE_CLASSES - 1];
  for (int i = w; i < n4; i += st.wide_workers)
    solve_env_wide<NVP>(m, st, st.cls_list[(size_t)(MSK_SOLVE_CLASSES - 1) * m->N + i], lds, st.wide_scratch + (size_t)w * CsWide<NVP>::SCRATCH);
}
template <int NVP, int GL>
__global__ void __launch_bounds__(64) k_csolve(const DModel* __restrict__ m, DState st, const int gm, const int ww) {
  extern __shared__ __attribute__((aligned(16))) float lds_cs[];
  if constexpr (NVP == 16) {
    static_assert(CsWide<16>::TOTAL <= CsLds<16, GL, GL>::TOTAL, "the wide class's LDS image fits the launch's");
    if ((int)blockIdx.x < ww) { csolve_wide_block<NVP>(m, st, blockIdx.x, lds_cs); return; }
  }
  csolve_block<NVP, GL>(m, st, gm, (int)blockIdx.x - ww, lds_cs);
}
template <int NVP>
__global__ void __launch_bounds__(64) k_csolve_wide(const DModel* __restrict__ m, DState st) {
  extern __shared__ __attribute__((aligned(16))) float lds_cw[];
  csolve_wide_block<NVP>(m, st, blockIdx.x, lds_cw);
}

#endif
