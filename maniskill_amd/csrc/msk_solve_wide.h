/*
 * msk_solve_wide.h — the wide solver class (msk_config.contact_capacity = 1): envs with more than MSK_MAX_BLOCKS constraint blocks, up to
 * MSK_MAX_BLOCKS_WIDE = 128 (a hand closed around an object, loose parts lying in a fixture: PhysX sizes its contact buffers for the
 * whole scene, mani_skill/utils/structs/types.py:18-23).
 *
 * One wavefront per env like classes 1..3 of msk_solve.h, the same constraint-space operator and the same sweeps -- the oracle's
 * arithmetic (oracle/orc_sim.c: orc_step_env), operation for operation -- with TWO blocks per lane: lane l owns blocks l and l + 64.
 * What does not fit the launch's LDS image lives in this worker's slice of DState::wide_scratch (L2): Y = W J^T (a column is written
 * by its owner and read by every lane: the wave waits for its stores first; the workgroup shares one L1) and the A image (every lane
 * reads back only what it wrote).  LDS keeps the env's tables (W, motion subspace, v*) and the per-row impulses.
 * These envs are few and long; the point of the class is that no contact is dropped, not its speed.
 */
#ifndef MSK_SOLVE_WIDE_H
#define MSK_SOLVE_WIDE_H

template <int NVP>
struct CsWide {
  static constexpr int NB = MSK_MAX_BLOCKS_WIDE;
  static constexpr int COLS = 3 * NB;
  static constexpr int W = 0;                       /* [NVP][NVP]                                   */
  static constexpr int SC = W + NVP * NVP;          /* [NVP][8]  motion subspace columns            */
  static constexpr int VF = SC + NVP * 8;           /* [NVP]     v*                                 */
  static constexpr int VD = VF + NVP;               /* [2 NVP]   v | dq for the integration         */
  static constexpr int LAMF = VD + 2 * NVP;         /* [COLS]    lambda per row                     */
  static constexpr int LAMS = LAMF + COLS;          /* [COLS]    sum of lambda over position sweeps */
  static constexpr int DESC = LAMS + COLS;          /* int [NB]  contact blocks: pair*4 + point     */
  static constexpr int TDESC = DESC + NB;           /* int [NB]  torsional blocks: pair             */
  static constexpr int TREF = TDESC + NB;           /* int [NB]  ... and the contact block of the pair's point */
  static constexpr int TRIM = TREF + NB;            /* [MSK_TRIM_CANDIDATES] trim_deepest's scratch */
  static constexpr int TOTAL = TRIM + MSK_TRIM_CANDIDATES;
  /* this worker's slice of DState::wide_scratch (floats): Y [COLS][NVP], then A [NB + 1][NB][9] (a column block of slack: the sweeps
   * request one block ahead) */
  static constexpr size_t Y_WORDS = (size_t)COLS * NVP;
  static constexpr size_t SCRATCH = Y_WORDS + (size_t)(NB + 1) * NB * 9;
};

template <int NVP>
struct WideBlk { /* what a lane keeps of one of its two blocks */
  typedef typename std::conditional<(NVP > 32), unsigned long long, unsigned>::type cmask_t;
  float J[3][NVP];
  float c0[3], lam[3], av[3], bv[3], ls[3], rinv[3];
  bool valid[3];
  float mu, erest, lo0, hi0, cfm0, vb0, mu_r, keep0, flim, hi_c, rest0, vclose0, fc, fs, csep;
  bool fixed_ct, is_joint, is_jfric, is_contact, is_tors;
  int code, tref, jd;
  cmask_t coordsA, coordsB;
  v3 cn, cpt;
};

/* integration of one env from v (new generalized velocity) and dq (the step's displacement) of coordinate `lane`: joint-velocity clamp and
 * limit backstop, q / qd / qacc, floating roots and free bodies (oracle: orc_step_env, "integrate") */
template <int NVP>
MSK_DEV void wide_integrate(const DModel* __restrict__ m, float* E, const int lane, float v, float dq, float* Lvd) {
  const int nd = m->nd, nv = m->nv;
  const float dt = m->cfg.timestep;
  if (lane < NVP) {
    if (lane < nd && !m->dof_body_is_root[lane]) {
      v = fminf(fmaxf(v, -MSK_MAX_JOINT_VELOCITY), MSK_MAX_JOINT_VELOCITY);
      dq = fminf(fmaxf(dq, -MSK_MAX_JOINT_VELOCITY * dt), MSK_MAX_JOINT_VELOCITY * dt);
      const float lo = m->dof_lo[lane], hi = m->dof_hi[lane];
      if (!(lo < -1e30f && hi > 1e30f)) {
        const float q0 = E[m->lay.q + lane], qn = q0 + dq;
        if (qn > hi + MSK_LIMIT_BACKSTOP) { dq = (hi + MSK_LIMIT_BACKSTOP) - q0; v = fminf(v, 0.0f); }
        else if (qn < lo - MSK_LIMIT_BACKSTOP) { dq = (lo - MSK_LIMIT_BACKSTOP) - q0; v = fmaxf(v, 0.0f); }
      }
    }
    Lvd[lane] = v;
    Lvd[NVP + lane] = dq;
  }
  wave_sync();
  if (lane < nd) {
    const float q = E[m->lay.q + lane], qd = E[m->lay.qd + lane];
    E[m->lay.qacc + lane] = (v - qd) / dt;
    E[m->lay.q + lane] = m->dof_body_is_root[lane] ? 0.0f : q + dq;
    E[m->lay.qd + lane] = v;
  }
  const int fr = (lane < nd) ? m->coord_root[lane] : -1;
  if (fr >= 0) {
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
  const int fb = (lane < nv) ? m->coord_body[lane] : -1;
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
}

template <int NVP>
MSK_DEV void solve_env_wide(const DModel* __restrict__ m, const DState& st, const int e, float* lds, float* scratch) {
  typedef CsWide<NVP> LY;
  typedef WideBlk<NVP> Blk;
  typedef typename Blk::cmask_t cmask_t;
  const int lane = threadIdx.x & 63;
  const int nd = m->nd, np = m->np, npp = m->npp;
  const float dt = m->cfg.timestep;
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
  float* Yg = scratch;
  float* Ag = scratch + LY::Y_WORDS;

  /* ---- joint blocks (msk_solve.h: solve_env) ---- */
  float c_lo = 3.0e38f, c_hi = 3.0e38f, reach_lo = 0.0f, reach_hi = 0.0f;
  if (lane < nd) {
    const float lo = m->dof_lo[lane], hi = m->dof_hi[lane], q = E[m->lay.q + lane];
    if (!(lo < -1e30f && hi > 1e30f)) {
      const float vf = st.vfree[(size_t)e * NVP + lane];
      c_lo = q - lo; c_hi = hi - q;
      reach_lo = fmaf(2.0f * dt, fmaxf(0.0f, -vf), MSK_LIMIT_SLACK);
      reach_hi = fmaf(2.0f * dt, fmaxf(0.0f, vf), MSK_LIMIT_SLACK);
    }
  }
  const unsigned long long drvm = st.drv_mask[e];
  const unsigned long long blo = __ballot(c_lo < reach_lo), bhi = __ballot(c_hi < reach_hi);
  const unsigned long long bdrv = __ballot(lane < nd && ((drvm >> lane) & 1ull));
  const unsigned long long bjoint = blo | bhi | bdrv;
  const int njoint = __popcll(bjoint);
  const int nfix = njoint + m->njfric;
  const int room = m->cap_blocks - nfix > 0 ? m->cap_blocks - nfix : 0;
  const int capc = room < m->cap_contacts ? room : m->cap_contacts;

  /* ---- contact points in canonical (pair, point) order, capacity capc; torsional rows of one-point manifolds ---- */
  int base = 0, ntors_pre = 0, ntors_all = 0, base0 = 0, ntors_pre0 = 0;
  const bool any_tors = m->has_tors != 0;
  for (int pass = 0;; ++pass) { /* (a second pass only behind trim_deepest) */
  base = 0; ntors_pre = 0; ntors_all = 0;
  for (int p0 = 0; p0 < np; p0 += 64) {
    const int p = p0 + lane;
    const int cnt = (p < np) ? cnts[p] : 0;
    int incl, tot;
    group_scan<64>(cnt, &incl, &tot);
    const int first = base + incl - cnt;
    const bool tors_pair = any_tors && p < np && (m->pinfo[p < np ? p : 0].patch_r > 0.0f || m->pinfo[p < np ? p : 0].min_patch_r > 0.0f);
    if (any_tors) ntors_pre += __popcll(__ballot(tors_pair && cnt == 1));
    if (first + cnt <= LY::NB)
      for (int kk = 0; kk < cnt; ++kk) Ldesc[first + kk] = p * 4 + kk;
    if (any_tors) {
      const bool tors = tors_pair && cnt == 1;
      const unsigned long long tm = __ballot(tors);
      const int trank = ntors_all + __popcll(tm & ((1ull << lane) - 1ull));
      if (tors && trank < LY::NB) { Ltdesc[trank] = p; Ltref[trank] = first; }
      ntors_all += __popcll(tm);
    }
    base += tot;
  }
  if (pass == 0) { base0 = base; ntors_pre0 = ntors_pre; }
  if (pass == 1 || !(base > capc)) break;
  trim_deepest(cnts, recs, np, capc, lds + LY::TRIM);
  }
  const bool overflow = base0 > capc;
  const int ncont = overflow ? capc : base;
  if (lane == 0) {
    if (st.ct_total[e] != base0 + ntors_pre0) atomicOr(st.env_overflow, 4);
    if (overflow) st.ct_total[e] = ncont + ntors_all;
  }
  int ntors = ntors_all < room - ncont ? ntors_all : room - ncont;
  if (ntors > LY::NB) ntors = LY::NB;
  const int nblk = __builtin_amdgcn_readfirstlane(nfix + ncont + ntors);   /* <= cap_blocks <= NB */
  if (lane == 0) {
    st.env_ncontacts[e] = ncont;
    if (overflow) atomicOr(st.env_overflow, 1);
  }
  const int nb = nblk > 0 ? nblk : 1; /* row stride of the A image */

  /* ---- the env's solver tables into LDS ---- */
  {
    const float4* wsrc = (const float4*)(st.W + (size_t)e * NVP * NVP);
    const float4* ssrc = (const float4*)(st.Scol + (size_t)e * NVP * 8);
    for (int i = lane; i < NVP * NVP / 4; i += 64) ((float4*)Lw)[i] = wsrc[i];
    for (int i = lane; i < NVP * 2; i += 64) ((float4*)Lsc)[i] = ssrc[i];
    if (lane < NVP) Lvf[lane] = st.vfree[(size_t)e * NVP + lane];
  }
  wave_sync();

  /* ---- my two blocks: what they read from global memory ---- */
  Blk B[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    Blk& X = B[hf];
    const int b = lane + 64 * hf;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      X.c0[s] = 0.0f; X.lam[s] = 0.0f; X.valid[s] = false; X.bv[s] = 0.0f; X.ls[s] = 0.0f; X.rinv[s] = 0.0f; X.av[s] = 0.0f;
#pragma unroll
      for (int k = 0; k < NVP; ++k) X.J[s][k] = 0.0f;
    }
    X.mu = 0.0f; X.erest = 0.0f; X.lo0 = 0.0f; X.hi0 = 0.0f; X.cfm0 = 0.0f; X.vb0 = 0.0f; X.mu_r = 0.0f; X.fc = 1.0f; X.fs = 0.0f; X.csep = 0.0f;
    X.fixed_ct = false; X.code = -1; X.tref = 0; X.jd = -1; X.coordsA = 0; X.coordsB = 0;
    X.cn = v3_make(0, 0, 1); X.cpt = v3_make(0, 0, 0);
    X.is_joint = b < njoint && b < nblk;
    X.is_jfric = b >= njoint && b < nfix && b < nblk;
    X.is_contact = b >= nfix && b < nfix + ncont && b < nblk;
    X.is_tors = b >= nfix + ncont && b < nblk;
    if (X.is_joint) {
      unsigned long long mk = bjoint;
      for (int t = 0; t < b; ++t) mk &= mk - 1ull;
      X.jd = __ffsll((long long)mk) - 1;
      X.valid[0] = (bdrv >> X.jd) & 1ull;
      X.valid[1] = (blo >> X.jd) & 1ull;
      X.valid[2] = (bhi >> X.jd) & 1ull;
      const float q = E[m->lay.q + X.jd];
      X.c0[1] = X.valid[1] ? q - m->dof_lo[X.jd] : 0.0f;
      X.c0[2] = X.valid[2] ? m->dof_hi[X.jd] - q : 0.0f;
      if (X.valid[0]) {
        const float4 dr = *(const float4*)(st.drv + ((size_t)e * NVP + X.jd) * 4);
        X.cfm0 = dr.x; X.vb0 = dr.y; X.hi0 = dr.z; X.lo0 = -dr.z;
      }
    } else if (X.is_jfric) {
      unsigned long long mk = m->jfric_mask;   /* 64 coordinates: bit d = dof d */
      for (int t = 0; t < b - njoint; ++t) mk &= mk - 1ull;
      X.jd = __ffsll((long long)mk) - 1;
      const int body = m->dof_body[X.jd];
      const float* x = st.jforce + ((size_t)e * m->nb + body) * 6;
      const float mag = sqrtf(fmaf(x[0], x[0], fmaf(x[1], x[1], fmaf(x[2], x[2], fmaf(x[3], x[3], fmaf(x[4], x[4], x[5] * x[5]))))));
      X.valid[0] = true;
      X.hi0 = m->bodies[body].jfriction * mag * dt;
      X.lo0 = -X.hi0;
    } else if (X.is_contact || X.is_tors) {
      int p, kk = 0;
      if (X.is_contact) { X.code = Ldesc[b - nfix]; p = X.code >> 2; kk = X.code & 3; }
      else { p = Ltdesc[b - nfix - ncont]; X.code = p * 4; X.tref = nfix + Ltref[b - nfix - ncont]; }
      const DPairInfo pi = m->pinfo[p];
      const float* rec = recs + (size_t)p * MSK_CT_REC;
      const float4 r0 = *(const float4*)rec;
      X.cn = v3_make(r0.x, r0.y, r0.z);
      X.csep = rec[16 + kk];
      X.coordsA = pi.ca; X.coordsB = pi.cb;
      if constexpr (NVP > 32) { X.coordsA |= (cmask_t)pi.ca_hi << 32; X.coordsB |= (cmask_t)pi.cb_hi << 32; }
      const float mu_eff = (m->has_static && st.ct_slip[(size_t)e * npp + p]) ? pi.mu : pi.mu_s;
      if (X.is_contact) {
        X.cpt = v3_make(rec[4 + 3 * kk], rec[4 + 3 * kk + 1], rec[4 + 3 * kk + 2]);
        X.mu = mu_eff;
        X.erest = pi.rest;
        X.lo0 = 0.0f; X.hi0 = MSK_MAX_ROW_IMPULSE;
        X.fixed_ct = X.coordsA == 0 || X.coordsB == 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) { X.valid[s] = true; X.c0[s] = X.csep; X.lam[s] = rec[20 + 3 * kk + s]; }
      } else {
        const float rp = fmaxf(pi.min_patch_r, sqrtf(fmaxf(0.0f, -X.csep) * pi.patch_r));
        X.mu_r = mu_eff * rp;
        X.valid[0] = true;
        X.lam[0] = r0.w;
      }
    }
  }

  /* ---- rows J ---- */
  cmask_t ucoords = 0;
  {
    cmask_t mine = 0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const Blk& X = B[hf];
      mine |= (X.is_joint || X.is_jfric) ? ((cmask_t)1 << X.jd) : ((X.is_contact || X.is_tors) ? (X.coordsA | X.coordsB) : (cmask_t)0);
    }
#pragma unroll
    for (int k = 0; k < NVP; ++k)
      if (__ballot((mine >> k) & 1u)) ucoords |= (cmask_t)1 << k;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    Blk& X = B[hf];
    if (X.is_joint) {
#pragma unroll
      for (int k = 0; k < NVP; ++k)
        if (k == X.jd) { X.J[0][k] = X.valid[0] ? 1.0f : 0.0f; X.J[1][k] = X.valid[1] ? 1.0f : 0.0f; X.J[2][k] = X.valid[2] ? -1.0f : 0.0f; }
    } else if (X.is_jfric) {
#pragma unroll
      for (int k = 0; k < NVP; ++k)
        if (k == X.jd) X.J[0][k] = 1.0f;
    }
    {
      v3 t1 = v3_make(0, 0, 0), t2 = t1;
      if (X.is_contact) msk_tangents(X.cn, &t1, &t2);
      sv6 F0, F1, F2;
      if (X.is_contact) {
        F0.a = v3_cross(X.cpt, X.cn); F0.l = X.cn;
        F1.a = v3_cross(X.cpt, t1); F1.l = t1;
        F2.a = v3_cross(X.cpt, t2); F2.l = t2;
      } else {
        F0.a = X.cn; F0.l = v3_make(0, 0, 0);
        F1 = sv6_zero(); F2 = sv6_zero();
      }
      const bool rows3 = X.is_contact, rows1 = X.is_contact || X.is_tors;
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        if (!((ucoords >> k) & 1u)) continue;
        const float* sc = Lsc + k * 8;
        sv6 Sk;
        Sk.a = v3_make(sc[0], sc[1], sc[2]);
        Sk.l = v3_make(sc[3], sc[4], sc[5]);
        const float sgn = (float)((X.coordsA >> k) & 1u) - (float)((X.coordsB >> k) & 1u);
        if (rows1) X.J[0][k] = sgn * sv6_dot(Sk, F0);
        if (rows3) { X.J[1][k] = sgn * sv6_dot(Sk, F1); X.J[2][k] = sgn * sv6_dot(Sk, F2); }
      }
    }
    if (X.is_contact) { /* the friction frame follows the motion (msk_solve.h) */
      float u1 = 0.0f, u2 = 0.0f;
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        if (!((ucoords >> k) & 1u)) continue;
        u1 = fmaf(X.J[1][k], Lvf[k], u1);
        u2 = fmaf(X.J[2][k], Lvf[k], u2);
      }
      const float n2 = fmaf(u1, u1, u2 * u2);
      if (n2 > MSK_FRICTION_ALIGN_SPEED * MSK_FRICTION_ALIGN_SPEED) {
        const float inv = 1.0f / sqrtf(n2);
        X.fc = u1 * inv; X.fs = u2 * inv;
#pragma unroll
        for (int k = 0; k < NVP; ++k) {
          if (!((ucoords >> k) & 1u)) continue;
          const float j1 = X.J[1][k], j2 = X.J[2][k];
          X.J[1][k] = fmaf(X.fc, j1, X.fs * j2);
          X.J[2][k] = fmaf(X.fc, j2, -(X.fs * j1));
        }
        const float l1 = X.lam[1], l2 = X.lam[2];
        X.lam[1] = fmaf(X.fc, l1, X.fs * l2);
        X.lam[2] = fmaf(X.fc, l2, -(X.fs * l1));
      }
    }
    /* Y = W J^T to the scratch slice; lambda_0 published for the warm start */
    const int b = lane + 64 * hf;
    if (b < nblk) {
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NVP; ++j) {
          const float w = Lw[k * NVP + j];
          y0 = fmaf(w, X.J[0][j], y0);
          y1 = fmaf(w, X.J[1][j], y1);
          y2 = fmaf(w, X.J[2][j], y2);
        }
        Yg[(size_t)(b * 3 + 0) * NVP + k] = y0;
        Yg[(size_t)(b * 3 + 1) * NVP + k] = y1;
        Yg[(size_t)(b * 3 + 2) * NVP + k] = y2;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) Llamf[b * 3 + s] = X.lam[s];
    }
  }
  wave_global_handoff();   /* the Y columns are read by every lane below */
  unsigned long long vm[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) { vm[s][0] = __ballot(B[0].valid[s]); vm[s][1] = __ballot(B[1].valid[s]); }
  auto rowvalid = [&](const int s, const int blk) -> bool { return (((blk < 64) ? vm[s][0] : vm[s][1]) >> (blk & 63)) & 1ull; };

  /* ---- constraint-space operator: A[(my block, s')][col] = J_(me, s') . Y_col ---- */
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    Blk& X = B[hf];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < NVP; ++k) {
        if (!((ucoords >> k) & 1u)) continue;
        acc = fmaf(X.J[s][k], Lvf[k], acc);
      }
      X.av[s] = acc;
    }
    const bool bounces = X.is_contact && X.erest > 0.0f && X.av[0] < -m->cfg.bounce_threshold;
    X.rest0 = bounces ? X.erest * X.av[0] : 0.0f;
    X.vclose0 = bounces ? -X.av[0] * dt : 0.0f;
  }
  for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int col = blk * 3 + s;
      if (!rowvalid(s, blk)) { /* a row that does not exist is an all-zero row: its column is zero */
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int b = lane + 64 * hf;
          if (b < nblk) {
            float* a = Ag + ((size_t)blk * nb + b) * 9 + s * 3;
            a[0] = 0.0f; a[1] = 0.0f; a[2] = 0.0f;
          }
        }
        continue;
      }
      const float* ycol = Yg + (size_t)col * NVP;
      MSK_OPAQUE_VGPR(ycol);   /* written by vector stores of this launch: never a scalar load */
      const float l0 = Llamf[col];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        Blk& X = B[hf];
        const int b = lane + 64 * hf;
        float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < NVP; ++k) {
          const float y = ycol[k];
          d0 = fmaf(X.J[0][k], y, d0);
          d1 = fmaf(X.J[1][k], y, d1);
          d2 = fmaf(X.J[2][k], y, d2);
        }
        if (b < nblk) {
          float* a = Ag + ((size_t)blk * nb + b) * 9 + s * 3;
          a[0] = d0; a[1] = d1; a[2] = d2;
        }
        if (b == blk) {
          if (s == 0) { const float arr = d0 + X.cfm0; X.rinv[0] = arr > MSK_MIN_RESPONSE ? 1.0f / arr : 0.0f; }
          if (s == 1) X.rinv[1] = d1 > MSK_MIN_RESPONSE ? 1.0f / d1 : 0.0f;
          if (s == 2) X.rinv[2] = d2 > MSK_MIN_RESPONSE ? 1.0f / d2 : 0.0f;
        }
        X.av[0] = fmaf(d0, l0, X.av[0]);
        X.av[1] = fmaf(d1, l0, X.av[1]);
        X.av[2] = fmaf(d2, l0, X.av[2]);
      }
    }
  }
  MSK_WAIT_VMCNT0();
  wave_sync();
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    Blk& X = B[hf];
    X.keep0 = fmaf(-X.cfm0, X.rinv[0], 1.0f);
    X.flim = X.is_contact ? X.mu : 0.0f;
    X.hi_c = X.is_contact ? 0.0f : MSK_MAX_ROW_IMPULSE;
  }

  /* ---- Gauss-Seidel sweeps ---- */
  auto load_cols = [&](const int blk, float (*dst)[9]) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int b = lane + 64 * hf;
      const float* a = Ag + ((size_t)blk * nb + (b < nblk ? b : 0)) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) dst[hf][i] = a[i];
    }
  };
  const unsigned long long tb0 = any_tors ? __ballot(B[0].is_tors) : 0ull, tb1 = any_tors ? __ballot(B[1].is_tors) : 0ull;
  auto clampf = [](float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); };
  auto sweep = [&](auto posit_tag) {
    constexpr bool POSIT = decltype(posit_tag)::value;
    float t0[2], t1[2], t2[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const Blk& X = B[hf];
      const float t0n = bias_over_arr<POSIT, false>(X.bv[0], X.c0[0], X.rinv[0], inv_h, inv_dt, pen_rate, X.rest0, X.vclose0);
      t0[hf] = X.is_contact ? t0n : X.vb0 * X.rinv[0];
      const float t1f = bias_over_arr<POSIT, true>(X.bv[1], X.c0[1], X.rinv[1], inv_h, inv_dt, pen_rate);
      const float t1n = bias_over_arr<POSIT, false>(X.bv[1], X.c0[1], X.rinv[1], inv_h, inv_dt, pen_rate);
      t1[hf] = X.is_contact ? t1f : t1n;
      const float t2f = bias_over_arr<POSIT, true>(X.bv[2], X.c0[2], X.rinv[2], inv_h, inv_dt, pen_rate);
      const float t2n = bias_over_arr<POSIT, false>(X.bv[2], X.c0[2], X.rinv[2], inv_h, inv_dt, pen_rate);
      t2[hf] = X.is_contact ? t2f : t2n;
    }
    auto block_steps = [&](auto hb_tag, const int blk, const float (*Ac)[9]) {
      constexpr int HB = decltype(hb_tag)::value;
      Blk& O = B[HB];                 /* the half the owner's block is in */
      const int ol = blk - 64 * HB;   /* the owner's lane */
      const bool owner = lane == ol;
      const unsigned rowbits = (rowvalid(0, blk) ? 1u : 0u) | (rowvalid(1, blk) ? 2u : 0u) | (rowvalid(2, blk) ? 4u : 0u);
      if (rowbits & 1u) {
        float lo = O.lo0, hi = O.hi0;
        if (((HB ? tb1 : tb0) >> ol) & 1ull) { /* a torsional block: its cone is sized by its point's normal impulse */
          const float l0 = __shfl(B[0].lam[0], O.tref & 63, 64), l1 = __shfl(B[1].lam[0], O.tref & 63, 64);
          const float lref = O.tref < 64 ? l0 : l1;
          if (O.is_tors) { hi = O.mu_r * lref; lo = -hi; }
        }
        const float nl = clampf(fmaf(-O.av[0], O.rinv[0], fmaf(O.lam[0], O.keep0, -t0[HB])), lo, hi);
        const float dl = readlane_f(nl - O.lam[0], ol);
        if (owner) O.lam[0] = nl;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          B[hf].av[0] = fmaf(Ac[hf][0], dl, B[hf].av[0]); B[hf].av[1] = fmaf(Ac[hf][1], dl, B[hf].av[1]); B[hf].av[2] = fmaf(Ac[hf][2], dl, B[hf].av[2]);
        }
      }
      if (rowbits & 2u) {
        const float hi = fmaf(O.flim, O.lam[0], O.hi_c), lo = fmaf(-O.flim, O.lam[0], 0.0f);
        const float nl = clampf(fmaf(-O.av[1], O.rinv[1], O.lam[1] - t1[HB]), lo, hi);
        const float dl = readlane_f(nl - O.lam[1], ol);
        if (owner) O.lam[1] = nl;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          B[hf].av[0] = fmaf(Ac[hf][3], dl, B[hf].av[0]); B[hf].av[1] = fmaf(Ac[hf][4], dl, B[hf].av[1]); B[hf].av[2] = fmaf(Ac[hf][5], dl, B[hf].av[2]);
        }
      }
      if (rowbits & 4u) {
        const float hi = fmaf(O.flim, O.lam[0], O.hi_c), lo = fmaf(-O.flim, O.lam[0], 0.0f);
        const float nl = clampf(fmaf(-O.av[2], O.rinv[2], O.lam[2] - t2[HB]), lo, hi);
        const float dl = readlane_f(nl - O.lam[2], ol);
        if (owner) O.lam[2] = nl;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          B[hf].av[0] = fmaf(Ac[hf][6], dl, B[hf].av[0]); B[hf].av[1] = fmaf(Ac[hf][7], dl, B[hf].av[1]); B[hf].av[2] = fmaf(Ac[hf][8], dl, B[hf].av[2]);
        }
      }
    };
    {
      float Ac[2][9], An[2][9];
      load_cols(0, Ac);
      for (int blk = 0; blk < nblk; ++blk) {
        load_cols(blk + 1, An);   /* (one column block of slack behind the image) */
        if (blk < 64) block_steps(std::integral_constant<int, 0>{}, blk, Ac);
        else block_steps(std::integral_constant<int, 1>{}, blk, Ac);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int i = 0; i < 9; ++i) Ac[hf][i] = An[hf][i];
      }
    }
    if (POSIT) {
      /* static geometry has the last word before positions move (msk_solve.h; oracle: ORC_STATIC_LAST_WORD) */
      auto last_word = [&](const Blk& X) {
        const float cur = X.c0[0] + X.bv[0];
        const float bias = (cur > 0.0f) ? cur * inv_h : 0.0f;
        return clampf(fmaf(-X.av[0], X.rinv[0], fmaf(X.lam[0], X.keep0, -(bias * X.rinv[0]))), 0.0f, MSK_MAX_ROW_IMPULSE);
      };
      const bool cand0 = B[0].fixed_ct && !(B[0].rest0 < 0.0f), cand1 = B[1].fixed_ct && !(B[1].rest0 < 0.0f);
      if (__ballot((cand0 && last_word(B[0]) > B[0].lam[0]) || (cand1 && last_word(B[1]) > B[1].lam[0])) != 0ull) {
        const unsigned long long cm0 = __ballot(cand0), cm1 = __ballot(cand1);
        auto word_step = [&](auto hb_tag, const int blk, const float (*Ac)[9]) {
          constexpr int HB = decltype(hb_tag)::value;
          Blk& O = B[HB];
          const int ol = blk - 64 * HB;
          const float nl = last_word(O);
          const bool add = (HB ? cand1 : cand0) && nl > O.lam[0];
          const float dl = readlane_f(add ? nl - O.lam[0] : 0.0f, ol);
          if (lane == ol && add) O.lam[0] = nl;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            B[hf].av[0] = fmaf(Ac[hf][0], dl, B[hf].av[0]); B[hf].av[1] = fmaf(Ac[hf][1], dl, B[hf].av[1]); B[hf].av[2] = fmaf(Ac[hf][2], dl, B[hf].av[2]);
          }
        };
        for (int blk = 0; blk < nblk; ++blk) {
          if (!((((blk < 64) ? cm0 : cm1) >> (blk & 63)) & 1ull)) continue;
          float Ac[2][9];
          load_cols(blk, Ac);
          if (blk < 64) word_step(std::integral_constant<int, 0>{}, blk, Ac);
          else word_step(std::integral_constant<int, 1>{}, blk, Ac);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          B[hf].bv[s] = fmaf(h, B[hf].av[s], B[hf].bv[s]);
          B[hf].ls[s] += B[hf].lam[s];
        }
    }
  };
  for (int sb = 0; sb < nsub; ++sb) {
    sweep(std::true_type{});
    sweep(std::false_type{});
  }
  for (int it = 0; it < nfinal; ++it) sweep(std::false_type{});

  /* ---- back to generalized coordinates ---- */
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const Blk& X = B[hf];
    const int b = lane + 64 * hf;
    if (b < nblk) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        Llamf[b * 3 + s] = X.lam[s];
        Llams[b * 3 + s] = X.ls[s];
      }
      if (X.is_contact) {
        float* rec = recs + (size_t)(X.code >> 2) * MSK_CT_REC + 20 + (X.code & 3) * 3;
        rec[0] = X.lam[0];
        rec[1] = fmaf(X.fc, X.lam[1], -(X.fs * X.lam[2]));
        rec[2] = fmaf(X.fs, X.lam[1], X.fc * X.lam[2]);
      }
      if (X.is_tors) recs[(size_t)(X.code >> 2) * MSK_CT_REC + 3] = X.lam[0];
    }
  }
  wave_sync();
  float v = 0.0f, dq = 0.0f;
  if (lane < NVP) {
    const float vf = Lvf[lane];
    float vk = vf, sk = (float)nsub * vf;
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (!rowvalid(s, blk)) continue;
        const int col = blk * 3 + s;
        const float y = Yg[(size_t)col * NVP + lane];
        vk = fmaf(y, Llamf[col], vk);
        sk = fmaf(y, Llams[col], sk);
      }
    }
    v = vk;
    dq = h * sk;
  }
  wide_integrate<NVP>(m, E, lane, v, dq, Lvd);
  wave_sync();   /* the next env of this worker reuses the LDS image */
}

#endif
