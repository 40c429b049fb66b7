/*
 * msk_solve.h — the TGS row solver, lane-group form (gfx950, wave64).
 *
 * One lane per generalized coordinate: an env owns G = 16 (nv <= 16) or 32 consecutive lanes of a
 * wavefront, a 64-thread workgroup holds 64/G envs, the grid is N*G/64 single-wave workgroups
 * (4096 PickCube envs -> 1024 waves = one per SIMD of the chip).
 *
 *   build   every constraint row r (joint limit or contact normal / tangent) is a pair of
 *           G-vectors J_r, Y_r = W J_r^T; lane k computes J_r[k] = +-S_k . F from its own motion
 *           subspace column S_k and Y_r[k] from its own row of W (both in registers), and the row
 *           is parked in LDS as float2 {J,Y} per lane together with {c0, 1/(J.Y), mu, kind}.  The
 *           envs of a workgroup share one LDS pool of 64/G * MSK_ROWS_LDS rows, carved after a
 *           counting pass (most envs need ~16 rows, a few need > 100); what does not fit spills to HBM.
 *   sweep   Gauss-Seidel over the rows, Np + Nv times: lane k keeps v[k] and dq[k] in registers,
 *           J.v and J.dq are 4-step DPP butterflies inside the 16-lane row (+1 bpermute for G = 32),
 *           the clamp is computed redundantly by all G lanes, v[k] += Y[k] * dlambda.
 *           The next row's {J,Y} and scalars are fetched from LDS while the current one reduces.
 *   finish  impulses back to the contact slots, q/qd/qacc, free bodies integrated by the lane of
 *           their first coordinate.
 *
 * Arithmetic order is the oracle's (oracle/orc_sim.c): coordinate-wise fmaf chains and balanced
 * pairwise-tree dot products, which is exactly what the butterfly produces — results are bit-identical.
 */
#ifndef MSK_SOLVE_H
#define MSK_SOLVE_H

#include "msk_model.h"

#define MSK_PEN_BETA 0.8f
#define MSK_MAX_DEPEN_VEL 3.0f
#define MSK_LIMIT_DISTANCE 0.1f

enum { ROW_LIMLO = 0, ROW_LIMHI = 1, ROW_CN = 2, ROW_CT1 = 3, ROW_CT2 = 4 };

/* ---- cross-lane primitives ---------------------------------------------------------------- */
template <int CTRL>
MSK_DEV float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
/* sum over the G lanes of a group as a balanced tree over adjacent pairs; every lane gets the result */
template <int G>
MSK_DEV float group_sum(float x) {
  x = x + dpp_mov<0xB1>(x);   /* quad_perm [1,0,3,2]  : lane ^ 1                         */
  x = x + dpp_mov<0x4E>(x);   /* quad_perm [2,3,0,1]  : lane ^ 2                         */
  x = x + dpp_mov<0x141>(x);  /* row_half_mirror      : the other quad of the 8-lane half */
  x = x + dpp_mov<0x140>(x);  /* row_mirror           : the other half of the 16-lane row */
  if (G == 32) x = x + __shfl_xor(x, 16, 64);
  return x;
}
/* wave-synchronous LDS hand-off inside one single-wave workgroup: LDS operations of a wave execute
 * in order, the fence keeps the compiler from moving accesses across it */
MSK_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int G>
struct SolveLds {
  static constexpr int EPB = 64 / G;                           /* envs per single-wave workgroup        */
  static constexpr int POOL = EPB * MSK_ROWS_LDS;              /* rows of the block-shared LDS row pool */
  static constexpr int JY = 0;                                 /* float2 [POOL][G]                      */
  static constexpr int RS = 2 * POOL * G;                      /* float4 [POOL]                         */
  static constexpr int LAM = RS + 4 * POOL;                    /* float  [POOL]                         */
  static constexpr int JT = LAM + POOL;                        /* float  [EPB][G]  J of the row being built */
  static constexpr int VD = JT + EPB * G;                      /* float  [EPB][2G] v | dq for the finish    */
  static constexpr int CNT = VD + EPB * 2 * G;                 /* int    [EPB]     rows per env             */
  static constexpr int TOTAL = CNT + 4;
};

/* one Gauss-Seidel row update; all G lanes of the env hold the same scalars.
 * new impulse = clamp(lam - (J.v + bias) / (J.Y)); the bias part does not depend on v and is folded first */
template <int G, bool POSIT>
MSK_DEV float row_step(const float2 jy, const float4 rs, const float lam0, const float inv_h, const float inv_dt,
                       const float beta_dt, float& v, const float dq, float& lam_n) {
  const float jv = group_sum<G>(jy.x * v);
  const float jdq = group_sum<G>(jy.x * dq);
  const bool is_n = (__float_as_int(rs.w) & 7) <= ROW_CN;
  const float cur = rs.x + jdq;
  float bias_n;
  if (POSIT) bias_n = (cur > 0.0f) ? cur * inv_h : fmaxf(cur * beta_dt, -MSK_MAX_DEPEN_VEL);
  else bias_n = (cur > 0.0f) ? cur * inv_dt : 0.0f;
  const float bias_t = POSIT ? jdq * inv_h : 0.0f;
  const float bias = is_n ? bias_n : bias_t;
  const float lim = rs.z * lam_n;
  const float lo = is_n ? 0.0f : -lim, hi = is_n ? INFINITY : lim;
  const float t0 = lam0 - bias * rs.y;
  const float nl = fminf(fmaxf(fmaf(-jv, rs.y, t0), lo), hi);
  if (is_n) lam_n = nl;
  v = fmaf(jy.y, nl - lam0, v);
  return nl;
}

/* one sweep over the rows of an env: rl rows in the LDS pool (next row prefetched while the current one
 * reduces), the rest (rare: the block's pool is full) in the HBM spill area */
template <int G, bool POSIT>
MSK_DEV void sweep_once(const int nr, const int rl, const float inv_h, const float inv_dt, const float beta_dt,
                        const float2* Ljy_k, const float4* Lrs, float* Llam, const float2* ovjy_k, const float4* ovrs,
                        float* ovlam, const bool live, float& v, const float dq) {
  float lam_n = 0.0f;
  if (rl > 0) {
    float2 jy = Ljy_k[0];
    float4 rs = Lrs[0];
    for (int r = 0; r < rl; ++r) {
      const float lam0 = Llam[r];
      const int rn = (r + 1 < rl) ? r + 1 : 0;
      const float2 jy_n = Ljy_k[rn * G];
      const float4 rs_n = Lrs[rn];
      Llam[r] = row_step<G, POSIT>(jy, rs, lam0, inv_h, inv_dt, beta_dt, v, dq, lam_n);
      jy = jy_n;
      rs = rs_n;
    }
  }
  for (int r = rl; r < nr; ++r) {
    const float2 jy = ovjy_k[(r - rl) * G];
    const float4 rs = ovrs[r - rl];
    const float lam0 = ovlam[r - rl];
    const float nl = row_step<G, POSIT>(jy, rs, lam0, inv_h, inv_dt, beta_dt, v, dq, lam_n);
    if (live) ovlam[r - rl] = nl;
  }
}

template <int G>
__global__ void __launch_bounds__(64) k_solve(const DModel* __restrict__ m, DState st) {
  typedef SolveLds<G> LY;
  __shared__ __attribute__((aligned(16))) float lds[LY::TOTAL];
  const int N = m->N;
  const int lane = threadIdx.x;
  const int k = lane % G;              /* my generalized coordinate */
  const int le = lane / G;             /* env slot inside the wave  */
  const int e_raw = blockIdx.x * LY::EPB + le;
  const bool live = e_raw < N;
  const int e = live ? e_raw : N - 1;  /* surplus groups shadow the last env and store nothing */
  const int nv = m->nv, nd = m->nd, np = m->np, npp = m->npp;
  const float dt = m->cfg.timestep;
  const int Np = m->cfg.solver_position_iterations, Nv = m->cfg.solver_velocity_iterations;
  const float h = dt / (float)Np;
  const float inv_h = 1.0f / h, inv_dt = 1.0f / dt, beta_dt = MSK_PEN_BETA / dt;
  const unsigned gmask = (G == 32) ? 0xFFFFFFFFu : 0xFFFFu;

  float* Ljt = lds + LY::JT + le * G;
  float* Lvd = lds + LY::VD + le * 2 * G;
  int* Lcnt = (int*)(lds + LY::CNT);

  /* my coordinate's tables */
  float Wk[G];
  {
    const float4* wp = (const float4*)(st.W + ((size_t)e * G + k) * G);
#pragma unroll
    for (int j = 0; j < G / 4; ++j) {
      float4 w = wp[j];
      Wk[4 * j] = w.x; Wk[4 * j + 1] = w.y; Wk[4 * j + 2] = w.z; Wk[4 * j + 3] = w.w;
    }
  }
  sv6 Sk;
  {
    const float4* sp = (const float4*)(st.Scol + ((size_t)e * G + k) * 8);
    float4 a = sp[0], b = sp[1];
    Sk.a = v3_make(a.x, a.y, a.z);
    Sk.l = v3_make(a.w, b.x, b.y);
  }
  const unsigned long long moves = (k < nv) ? m->coord_moves[k] : 0ull;
  float v = st.vfree[(size_t)e * G + k];  /* zero for k >= nv */
  float dq = 0.0f;
  float* E = EREC(st, m, e);
  const float qk = (k < nd) ? E[m->lay.q + k] : 0.0f;
  const float qdk = (k < nd) ? E[m->lay.qd + k] : 0.0f;
  int* cnts = st.ct_cnt + (size_t)e * npp;
  float* recs = st.ct_rec + (size_t)e * npp * MSK_CT_REC;

  /* ---- count the rows of every env of the block, carve the LDS pool ------------------------------- */
  float c_lo = 3.0e38f, c_hi = 3.0e38f;   /* lane k owns dof k: distance to its limits */
  if (k < nd) {
    const float lo = m->dof_lo[k], hi = m->dof_hi[k];
    if (!(lo < -1e30f && hi > 1e30f)) { c_lo = qk - lo; c_hi = hi - qk; }
  }
  const unsigned bits_lo = (unsigned)(__ballot(c_lo < MSK_LIMIT_DISTANCE) >> (le * G)) & gmask;
  const unsigned bits_hi = (unsigned)(__ballot(c_hi < MSK_LIMIT_DISTANCE) >> (le * G)) & gmask;
  int my_points = 0;
  for (int p0 = 0; p0 < np; p0 += G) my_points += (p0 + k < np) ? cnts[p0 + k] : 0;
  const int points = (int)group_sum<G>((float)my_points);  /* <= 4 * MSK_MAX_PAIRS: exact in fp32 */
  const int nr_total = __popc(bits_lo) + __popc(bits_hi) + 3 * min(points, MSK_MAX_CONTACTS);
  if (k == 0) Lcnt[le] = nr_total;
  wave_sync();
  int base = 0;
#pragma unroll
  for (int j = 0; j < LY::EPB; ++j) base += (j < le) ? Lcnt[j] : 0;
  const int rl = max(0, min(nr_total, LY::POOL - base));  /* my rows in LDS: pool slots base .. base+rl-1 */
  float2* Ljy = (float2*)(lds + LY::JY) + (size_t)(base < LY::POOL ? base : 0) * G + k;
  float4* Lrs = (float4*)(lds + LY::RS) + (base < LY::POOL ? base : 0);
  float* Llam = lds + LY::LAM + (base < LY::POOL ? base : 0);
  float2* ovjy = st.ov_jy + (size_t)e * MSK_MAX_ROWS * G + k;
  float4* ovrs = st.ov_rs + (size_t)e * MSK_MAX_ROWS;
  float* ovlam = st.ov_lam + (size_t)e * MSK_MAX_ROWS;

  int nr = 0;
  /* stores one finished row: all lanes pass the same scalars */
  auto put_row = [&](float J, int kind, int code, float c0, float mu, float lam0) {
    Ljt[k] = J;
    wave_sync();
    float Y = 0.0f;
#pragma unroll
    for (int j = 0; j < G / 4; ++j) {
      const float4 jj = ((const float4*)Ljt)[j];
      Y = fmaf(Wk[4 * j], jj.x, Y);
      Y = fmaf(Wk[4 * j + 1], jj.y, Y);
      Y = fmaf(Wk[4 * j + 2], jj.z, Y);
      Y = fmaf(Wk[4 * j + 3], jj.w, Y);
    }
    wave_sync();
    const float d = group_sum<G>(J * Y);
    const float rinv = 1.0f / d;
    const float4 rs = make_float4(c0, rinv, mu, __int_as_float(kind | (code << 3)));
    if (nr < rl) {
      Ljy[nr * G] = make_float2(J, Y);
      Lrs[nr] = rs;
      Llam[nr] = lam0;
    } else if (live) {
      ovjy[(nr - rl) * G] = make_float2(J, Y);
      ovrs[nr - rl] = rs;
      ovlam[nr - rl] = lam0;
    }
    if (lam0 != 0.0f) v = fmaf(Y, lam0, v); /* warm start */
    nr++;
  };

  /* ---- joint-limit rows -------------------------------------------------------------------------- */
  {
    unsigned both = bits_lo | bits_hi;
    while (both) {
      const int d = __ffs(both) - 1;
      both &= both - 1;
      const float c0lo = __shfl(c_lo, d, G), c0hi = __shfl(c_hi, d, G);
      if ((bits_lo >> d) & 1) put_row((k == d) ? 1.0f : 0.0f, ROW_LIMLO, d, c0lo, 0.0f, 0.0f);
      if ((bits_hi >> d) & 1) put_row((k == d) ? -1.0f : 0.0f, ROW_LIMHI, d, c0hi, 0.0f, 0.0f);
    }
  }

  /* ---- contact rows, candidate pairs in canonical order ------------------------------------------- */
  int ncontacts = 0;
  bool overflow = false;
  for (int p0 = 0; p0 < np; p0 += G) {
    const int cnt_mine = (p0 + k < np) ? cnts[p0 + k] : 0;
    unsigned bits = (unsigned)(__ballot(cnt_mine > 0) >> (le * G)) & gmask;
    if (overflow) { /* capacity exhausted earlier: the remaining slots are emptied */
      if (live && p0 + k < np && cnt_mine > 0) cnts[p0 + k] = 0;
      continue;
    }
    while (bits) {
      const int j = __ffs(bits) - 1;
      bits &= bits - 1;
      const int p = p0 + j;
      const int cnt = __shfl(cnt_mine, j, G);
      if (overflow) {
        if (live && k == 0) cnts[p] = 0;
        continue;
      }
      const DPairInfo pi = m->pinfo[p];
      const float4* rec = (const float4*)(recs + (size_t)p * MSK_CT_REC);
      const float4 r0 = rec[0];
      const v3 n = v3_make(r0.x, r0.y, r0.z);
      v3 t1, t2;
      msk_tangents(n, &t1, &t2);
      const float4 p01 = rec[1], p12 = rec[2], p23 = rec[3], sp = rec[4];
      const float pos[12] = {p01.x, p01.y, p01.z, p01.w, p12.x, p12.y, p12.z, p12.w, p23.x, p23.y, p23.z, p23.w};
      const float sep[4] = {sp.x, sp.y, sp.z, sp.w};
      const float4 l0 = rec[5], l1 = rec[6], l2 = rec[7];
      const float lam[12] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w};
      const bool mvA = pi.ba >= 0 && ((moves >> pi.ba) & 1), mvB = pi.bb >= 0 && ((moves >> pi.bb) & 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk >= cnt) break;
        if (ncontacts >= MSK_MAX_CONTACTS) {
          overflow = true;
          if (live && k == 0) cnts[p] = kk;
          break;
        }
        const v3 pt = v3_make(pos[3 * kk], pos[3 * kk + 1], pos[3 * kk + 2]);
        const v3 dirs[3] = {n, t1, t2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          sv6 F;
          F.a = v3_cross(pt, dirs[a]);
          F.l = dirs[a];
          const float x = sv6_dot(Sk, F);
          float J = 0.0f;
          if (mvA) J = fmaf(1.0f, x, J);
          if (mvB) J = fmaf(-1.0f, x, J);
          put_row(J, ROW_CN + a, p * 4 + kk, sep[kk], pi.mu, lam[3 * kk + a]);
        }
        ncontacts++;
      }
    }
  }
  if (live && k == 0) {
    st.env_ncontacts[e] = ncontacts;
    if (overflow) atomicOr(st.env_overflow, 1);
  }
  wave_sync();

  /* ---- Gauss-Seidel sweeps: Np position iterations (dq advances by h*v after each), Nv velocity ones --- */
  for (int it = 0; it < Np; ++it) {
    sweep_once<G, true>(nr, rl, inv_h, inv_dt, beta_dt, Ljy, Lrs, Llam, ovjy, ovrs, ovlam, live, v, dq);
    dq = fmaf(h, v, dq);
  }
  for (int it = 0; it < Nv; ++it)
    sweep_once<G, false>(nr, rl, inv_h, inv_dt, beta_dt, Ljy, Lrs, Llam, ovjy, ovrs, ovlam, live, v, dq);
  wave_sync();

  /* ---- impulses back to the contact slots (reports + next step's warm start) ------------------------ */
  for (int r = k; r < nr; r += G) {
    const float4 rs = (r < rl) ? Lrs[r] : ovrs[r - rl];
    const int w = __float_as_int(rs.w), kind = w & 7, code = w >> 3;
    if (kind >= ROW_CN && live) {
      const float lam = (r < rl) ? Llam[r] : ovlam[r - rl];
      recs[(size_t)(code >> 2) * MSK_CT_REC + 20 + (code & 3) * 3 + (kind - ROW_CN)] = lam;
    }
  }

  /* ---- integrate ------------------------------------------------------------------------------------- */
  Lvd[k] = v;
  Lvd[G + k] = dq;
  wave_sync();
  if (live && k < nd) {
    E[m->lay.qacc + k] = (v - qdk) / dt;
    E[m->lay.q + k] = qk + dq;
    E[m->lay.qd + k] = v;
  }
  const int fb = (k < nv) ? m->coord_body[k] : -1;
  if (live && fb >= 0) {
    const DBody* b = &m->bodies[fb];
    const v3 dx = v3_make(Lvd[G + k], Lvd[G + k + 1], Lvd[G + k + 2]);
    const v3 dr = v3_make(Lvd[G + k + 3], Lvd[G + k + 4], Lvd[G + k + 5]);
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

#endif
