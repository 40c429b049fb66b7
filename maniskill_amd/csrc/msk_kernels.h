/*
 * msk_kernels.h — the HIP kernels of one physics substep (gfx950, wave64).
 *
 * Launch shapes (N envs, P candidate pairs):
 *   k_dynamics   <<<N/64, 64>>>          one lane per env: link frames, CRBA, RNEA, implicit-PD
 *                                         system matrix, Cholesky, A^-1, unconstrained velocity
 *   k_collide    <<<(N/64, P), 64>>>     one lane per (pair, env); the pair is wave-uniform
 *   k_solve      <<<N/64, 64>>>          one lane per env: row assembly, TGS sweeps, integration,
 *                                         final kinematics, impulse write-back
 *   k_apply / k_fetch / k_kinematics / k_query   memcpy-class layout converters (AoS rows <-> SoA)
 * All per-env data is SoA with env fastest (msk_model.h), so a wave's accesses coalesce.
 * Arithmetic order mirrors the CPU oracle statement for statement (bitwise parity target).
 */
#ifndef MSK_KERNELS_H
#define MSK_KERNELS_H

#include "msk_collide.h"

#define MSK_PEN_BETA 0.8f
#define MSK_MAX_DEPEN_VEL 3.0f
#define MSK_WARM_DIST 5.0e-3f
#define MSK_WARM_FACTOR 0.9f

#define AT(arr, k) (arr)[(size_t)(k) * (size_t)N + (size_t)e]

MSK_DEV pose load_pose(const float* bpose, int body, int N, int e) {
  pose p;
  p.p = v3_make(AT(bpose, body * 7 + 0), AT(bpose, body * 7 + 1), AT(bpose, body * 7 + 2));
  p.q = quat_make(AT(bpose, body * 7 + 3), AT(bpose, body * 7 + 4), AT(bpose, body * 7 + 5), AT(bpose, body * 7 + 6));
  return p;
}
MSK_DEV void store_pose(float* bpose, int body, int N, int e, pose p) {
  AT(bpose, body * 7 + 0) = p.p.x; AT(bpose, body * 7 + 1) = p.p.y; AT(bpose, body * 7 + 2) = p.p.z;
  AT(bpose, body * 7 + 3) = p.q.w; AT(bpose, body * 7 + 4) = p.q.x; AT(bpose, body * 7 + 5) = p.q.y; AT(bpose, body * 7 + 6) = p.q.z;
}
MSK_DEV v3 load_v3(const float* a, int k, int N, int e) { return v3_make(AT(a, k * 3 + 0), AT(a, k * 3 + 1), AT(a, k * 3 + 2)); }
MSK_DEV void store_v3(float* a, int k, int N, int e, v3 v) { AT(a, k * 3 + 0) = v.x; AT(a, k * 3 + 1) = v.y; AT(a, k * 3 + 2) = v.z; }

/* per-thread working set of the kinematics / dynamics phase */
struct KinScratch {
  pose bpose[MSK_MAX_BODIES];
  sv6 S[MSK_MAX_BODIES];
  sv6 V[MSK_MAX_BODIES];
  v3 comw[MSK_MAX_BODIES];
  float Iw[MSK_MAX_BODIES][6];
};

/* link frames, joint subspaces, spatial velocities, world COM / inertia; publishes link velocities */
MSK_DEV void kinematics(const DModel* m, const DState& st, int N, int e, const float* q, const float* qd,
                        KinScratch* s, bool publish) {
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    s->S[i] = sv6_zero();
    s->V[i] = sv6_zero();
    if (b->kind == MSK_BODY_LINK && b->parent >= 0) {
      pose Tj = pose_mul(s->bpose[b->parent], b->Xp);
      v3 axis = quat_rotate(Tj.q, v3_make(1, 0, 0));
      pose Jq;
      Jq.p = v3_make(0, 0, 0);
      Jq.q = quat_make(1, 0, 0, 0);
      if (b->jtype == MSK_JOINT_REVOLUTE) {
        float sn, cs;
        msk_sincos(0.5f * q[b->dof], &sn, &cs);
        Jq.q = quat_make(cs, sn, 0, 0);
        s->S[i].a = axis;
        s->S[i].l = v3_cross(Tj.p, axis);
      } else if (b->jtype == MSK_JOINT_PRISMATIC) {
        Jq.p = v3_make(q[b->dof], 0, 0);
        s->S[i].l = axis;
      }
      pose T = pose_mul(pose_mul(Tj, Jq), b->XcInv);
      T.q = quat_normalize(T.q);
      s->bpose[i] = T;
      s->V[i] = s->V[b->parent];
      if (b->dof >= 0) s->V[i] = sv6_madd(s->V[i], s->S[i], qd[b->dof]);
    } else {
      s->bpose[i] = load_pose(st.bpose, i, N, e);
    }
    m33 R = quat_to_m33(s->bpose[i].q);
    s->comw[i] = v3_add(s->bpose[i].p, m33_mulv(&R, b->com));
    sym6_rotate(&R, b->I6, s->Iw[i]);
    if (publish) {
      if (b->kind == MSK_BODY_LINK) {
        if (b->parent >= 0) store_pose(st.bpose, i, N, e, s->bpose[i]);
        store_v3(st.bang, i, N, e, s->V[i].a);
        store_v3(st.blin, i, N, e, v3_add(s->V[i].l, v3_cross(s->V[i].a, s->comw[i])));
      } else if (b->kind == MSK_BODY_KINEMATIC) {
        store_v3(st.blin, i, N, e, v3_make(0, 0, 0));
        store_v3(st.bang, i, N, e, v3_make(0, 0, 0));
      }
    }
  }
}

__global__ void __launch_bounds__(64) k_kinematics(const DModel* __restrict__ m, DState st) {
  const int N = m->N;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= N) return;
  float q[MSK_MAX_DOF], qd[MSK_MAX_DOF];
  for (int i = 0; i < m->nd; ++i) { q[i] = AT(st.q, i); qd[i] = AT(st.qd, i); }
  KinScratch s;
  kinematics(m, st, N, e, q, qd, &s, true);
}

/* ---- dynamics ------------------------------------------------------------------------ */
__global__ void __launch_bounds__(64) k_dynamics(const DModel* __restrict__ m, DState st) {
  const int N = m->N;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= N) return;
  const int nd = m->nd;
  const float dt = m->cfg.timestep;
  const v3 g = v3_make(m->cfg.gravity[0], m->cfg.gravity[1], m->cfg.gravity[2]);
  float q[MSK_MAX_DOF], qd[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i) { q[i] = AT(st.q, i); qd[i] = AT(st.qd, i); }
  KinScratch s;
  kinematics(m, st, N, e, q, qd, &s, true);

  sinertia Ic[MSK_MAX_BODIES];
  sv6 f[MSK_MAX_BODIES];
  sv6 acc[MSK_MAX_BODIES];
  float M[MSK_MAX_DOF][MSK_MAX_DOF];
  float bias[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i)
    for (int k = 0; k < nd; ++k) M[i][k] = 0.0f;
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    v3 cw = s.comw[i];
    float ms = b->mass;
    sinertia Isp;
    Isp.m = ms;
    Isp.h = v3_scale(cw, ms);
    float cc = v3_dot(cw, cw);
    Isp.I[0] = s.Iw[i][0] + ms * (cc - cw.x * cw.x);
    Isp.I[1] = s.Iw[i][1] + ms * (cc - cw.y * cw.y);
    Isp.I[2] = s.Iw[i][2] + ms * (cc - cw.z * cw.z);
    Isp.I[3] = s.Iw[i][3] - ms * (cw.x * cw.y);
    Isp.I[4] = s.Iw[i][4] - ms * (cw.x * cw.z);
    Isp.I[5] = s.Iw[i][5] - ms * (cw.y * cw.z);
    Ic[i] = Isp;
    if (b->parent < 0) {
      acc[i] = sv6_zero();
    } else {
      acc[i] = acc[b->parent];
      if (b->dof >= 0) {
        sv6 sq = {v3_scale(s.S[i].a, qd[b->dof]), v3_scale(s.S[i].l, qd[b->dof])};
        acc[i] = sv6_add(acc[i], sv6_crossm(s.V[b->parent], sq));
      }
    }
    sv6 Iv = sinertia_mul(&Isp, s.V[i]);
    f[i] = sv6_add(sinertia_mul(&Isp, acc[i]), sv6_crossf(s.V[i], Iv));
    if (!b->nograv) {
      v3 mg = v3_scale(g, ms);
      f[i].a = v3_sub(f[i].a, v3_cross(cw, mg));
      f[i].l = v3_sub(f[i].l, mg);
    }
  }
  for (int i = m->nb - 1; i >= 0; --i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    if (b->dof >= 0) bias[b->dof] = sv6_dot(s.S[i], f[i]);
    if (b->parent >= 0) {
      f[b->parent] = sv6_add(f[b->parent], f[i]);
      sinertia_acc(&Ic[b->parent], &Ic[i]);
    }
  }
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    sv6 F = sinertia_mul(&Ic[i], s.S[i]);
    M[b->dof][b->dof] = sv6_dot(s.S[i], F) + b->armature;
    int j = b->parent;
    while (j >= 0) {
      const DBody* bj = &m->bodies[j];
      if (bj->dof >= 0) {
        float v = sv6_dot(s.S[j], F);
        M[b->dof][bj->dof] = v;
        M[bj->dof][b->dof] = v;
      }
      j = bj->parent;
    }
    /* publish the joint subspace for the row assembly */
    AT(st.S, b->dof * 6 + 0) = s.S[i].a.x; AT(st.S, b->dof * 6 + 1) = s.S[i].a.y; AT(st.S, b->dof * 6 + 2) = s.S[i].a.z;
    AT(st.S, b->dof * 6 + 3) = s.S[i].l.x; AT(st.S, b->dof * 6 + 4) = s.S[i].l.y; AT(st.S, b->dof * 6 + 5) = s.S[i].l.z;
  }
  /* implicit PD drives / tendons folded into A */
  float Kd[MSK_MAX_DOF], Dd[MSK_MAX_DOF], fconst[MSK_MAX_DOF], fmaxd[MSK_MAX_DOF], err[MSK_MAX_DOF];
  float qt[MSK_MAX_DOF], qdt[MSK_MAX_DOF], qf[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i) { qt[i] = AT(st.qt, i); qdt[i] = AT(st.qdt, i); qf[i] = AT(st.qf, i); }
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    Kd[b->dof] = b->K; Dd[b->dof] = b->D; fmaxd[b->dof] = b->fmax; fconst[b->dof] = 0.0f;
    err[b->dof] = q[b->dof] - qt[b->dof];
  }
  float A[MSK_MAX_DOF][MSK_MAX_DOF], L[MSK_MAX_DOF][MSK_MAX_DOF], rhs[MSK_MAX_DOF], vfree[MSK_MAX_DOF];
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < nd; ++i) {
      float mv = 0.0f;
      for (int k = 0; k < nd; ++k) { A[i][k] = M[i][k]; mv = fmaf(M[i][k], qd[k], mv); }
      A[i][i] += dt * fmaf(dt, Kd[i], Dd[i]);
      float tau = qf[i] - bias[i] - Kd[i] * err[i] + Dd[i] * qdt[i] + fconst[i];
      rhs[i] = fmaf(dt, tau, mv);
    }
    for (int t = 0; t < m->nt; ++t) {
      const DTendon* tn = &m->tendons[t];
      float g2 = dt * fmaf(dt, tn->K, tn->D);
      float te = fmaf(tn->ca, q[tn->dof_a], tn->cb * q[tn->dof_b]) - tn->rest;
      A[tn->dof_a][tn->dof_a] += g2 * tn->ca * tn->ca;
      A[tn->dof_b][tn->dof_b] += g2 * tn->cb * tn->cb;
      A[tn->dof_a][tn->dof_b] += g2 * tn->ca * tn->cb;
      A[tn->dof_b][tn->dof_a] += g2 * tn->ca * tn->cb;
      rhs[tn->dof_a] -= dt * tn->K * te * tn->ca;
      rhs[tn->dof_b] -= dt * tn->K * te * tn->cb;
    }
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < nd; ++k) L[i][k] = 0.0f;
    for (int i = 0; i < nd; ++i) {
      for (int j = 0; j <= i; ++j) {
        float sum = A[i][j];
        for (int k = 0; k < j; ++k) sum = fmaf(-L[i][k], L[j][k], sum);
        if (i == j) L[i][i] = sqrtf(sum);
        else L[i][j] = sum / L[j][j];
      }
    }
    float y[MSK_MAX_DOF];
    for (int i = 0; i < nd; ++i) {
      float sum = rhs[i];
      for (int k = 0; k < i; ++k) sum = fmaf(-L[i][k], y[k], sum);
      y[i] = sum / L[i][i];
    }
    for (int i = nd - 1; i >= 0; --i) {
      float sum = y[i];
      for (int k = i + 1; k < nd; ++k) sum = fmaf(-L[k][i], vfree[k], sum);
      vfree[i] = sum / L[i][i];
    }
    if (pass == 1) break;
    int nsat = 0;
    for (int i = 0; i < nd; ++i) {
      if (Kd[i] == 0.0f && Dd[i] == 0.0f) continue;
      float F = -Kd[i] * fmaf(dt, vfree[i], err[i]) - Dd[i] * (vfree[i] - qdt[i]);
      if (fabsf(F) > fmaxd[i]) {
        fconst[i] = (F > 0.0f) ? fmaxd[i] : -fmaxd[i];
        Kd[i] = 0.0f; Dd[i] = 0.0f; err[i] = 0.0f;
        nsat++;
      }
    }
    if (nsat == 0) break;
  }
  for (int i = 0; i < nd; ++i) AT(st.vfree, i) = vfree[i];
  /* A^-1 column by column (reuse A as the output buffer) */
  for (int col = 0; col < nd; ++col) {
    float y[MSK_MAX_DOF], x[MSK_MAX_DOF];
    for (int i = 0; i < nd; ++i) {
      float sum = (i == col) ? 1.0f : 0.0f;
      for (int k = 0; k < i; ++k) sum = fmaf(-L[i][k], y[k], sum);
      y[i] = sum / L[i][i];
    }
    for (int i = nd - 1; i >= 0; --i) {
      float sum = y[i];
      for (int k = i + 1; k < nd; ++k) sum = fmaf(-L[k][i], x[k], sum);
      x[i] = sum / L[i][i];
    }
    for (int i = 0; i < nd; ++i) AT(st.Minv, i * nd + col) = x[i];
  }
  /* free bodies */
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_DYNAMIC) continue;
    v3 v = load_v3(st.blin, i, N, e), w = load_v3(st.bang, i, N, e);
    if (!b->nograv) v = v3_madd(v, g, dt);
    float kl = fmaxf(0.0f, 1.0f - dt * b->lin_damp);
    float ka = fmaxf(0.0f, 1.0f - dt * b->ang_damp);
    v = v3_scale(v, kl);
    w = v3_scale(w, ka);
    AT(st.vfree, b->vofs + 0) = v.x; AT(st.vfree, b->vofs + 1) = v.y; AT(st.vfree, b->vofs + 2) = v.z;
    AT(st.vfree, b->vofs + 3) = w.x; AT(st.vfree, b->vofs + 4) = w.y; AT(st.vfree, b->vofs + 5) = w.z;
    m33 R = quat_to_m33(s.bpose[i].q);
    float Iwi[6];
    sym6_rotate(&R, b->Iinv6, Iwi);
    for (int k = 0; k < 6; ++k) AT(st.Iwinv, i * 6 + k) = Iwi[k];
    store_v3(st.comw, i, N, e, s.comw[i]);
  }
}

/* ---- collision -------------------------------------------------------------------------- */
MSK_DEV pose shape_pose_dev(const DModel* m, const DState& st, const DShape* sh, int N, int e) {
  if (sh->body < 0) return sh->local;
  return pose_mul(load_pose(st.bpose, sh->body, N, e), sh->local);
}

__global__ void __launch_bounds__(64) k_collide(const DModel* __restrict__ m, DState st) {
  const int N = m->N;
  const int e = blockIdx.x * 64 + threadIdx.x;
  const int pi = blockIdx.y;
  if (e >= N) return;
  const DShape* A = &m->shapes[m->pairs[pi].sa];
  const DShape* B = &m->shapes[m->pairs[pi].sb];
  pose TA = shape_pose_dev(m, st, A, N, e), TB = shape_pose_dev(m, st, B, N, e);
  const float margin = 2.0f * m->cfg.contact_offset;
  DContactOut out[4];
  int n = 0;
  bool done = false;
  if (A->type == MSK_SHAPE_PLANE || B->type == MSK_SHAPE_PLANE) {
    const int pa = A->type == MSK_SHAPE_PLANE;
    const DShape* P = pa ? A : B;
    const DShape* C = pa ? B : A;
    const pose* TP = pa ? &TA : &TB;
    const pose* TC = pa ? &TB : &TA;
    if (C->type != MSK_SHAPE_PLANE) {
      v3 cc, ch;
      world_aabb(C, TC, &cc, &ch);
      v3 pn = quat_rotate(TP->q, v3_make(1, 0, 0));
      float lo = v3_dot(pn, cc) - v3_dot(pn, TP->p) - (fabsf(pn.x) * ch.x + fabsf(pn.y) * ch.y + fabsf(pn.z) * ch.z);
      if (!(lo > margin)) n = plane_convex(m, P, TP, C, TC, margin, pa, out);
    }
    done = true;
  }
  if (!done) {
    v3 ca, ha, cb, hb;
    world_aabb(A, &TA, &ca, &ha);
    world_aabb(B, &TB, &cb, &hb);
    bool overlap = !(fabsf(ca.x - cb.x) > ha.x + hb.x + margin) && !(fabsf(ca.y - cb.y) > ha.y + hb.y + margin) &&
                   !(fabsf(ca.z - cb.z) > ha.z + hb.z + margin);
    if (overlap) {
      v3 nrm, wa, wb;
      float sep;
      int hit;
      if (A->type == MSK_SHAPE_BOX && B->type == MSK_SHAPE_BOX) {
        hit = sat_box_box(A, &TA, B, &TB, margin, &nrm, &sep);
        if (hit) {
          wa = support(m, A, &TA, v3_neg(nrm));
          wb = support(m, B, &TB, nrm);
        }
      } else {
        hit = gjk_epa(m, A, &TA, B, &TB, ca, cb, margin, &nrm, &sep, &wa, &wb);
      }
      if (hit) n = build_manifold(m, A, &TA, B, &TB, nrm, margin, wa, wb, sep, out);
    }
  }
  /* warm start from the previous contents of this pair's slot, then overwrite it */
  const int nprev = AT(st.ct_cnt, pi);
  v3 ppos[4];
  float plam[4][3];
  for (int j = 0; j < 4; ++j) {
    if (j < nprev) {
      ppos[j] = v3_make(AT(st.ct_pos, pi * 12 + j * 3 + 0), AT(st.ct_pos, pi * 12 + j * 3 + 1), AT(st.ct_pos, pi * 12 + j * 3 + 2));
      for (int a = 0; a < 3; ++a) plam[j][a] = AT(st.ct_lam, pi * 12 + j * 3 + a);
    }
  }
  AT(st.ct_cnt, pi) = n;
  if (n > 0) {
    AT(st.ct_n, pi * 3 + 0) = out[0].n.x; AT(st.ct_n, pi * 3 + 1) = out[0].n.y; AT(st.ct_n, pi * 3 + 2) = out[0].n.z;
  }
  for (int k = 0; k < n; ++k) {
    float lam[3] = {0.0f, 0.0f, 0.0f};
    int best = -1;
    float bd = MSK_WARM_DIST * MSK_WARM_DIST;
    for (int j = 0; j < nprev; ++j) {
      float d2 = v3_len2(v3_sub(ppos[j], out[k].pos));
      if (d2 < bd) { bd = d2; best = j; }
    }
    if (best >= 0)
      for (int a = 0; a < 3; ++a) lam[a] = MSK_WARM_FACTOR * plam[best][a];
    AT(st.ct_pos, pi * 12 + k * 3 + 0) = out[k].pos.x;
    AT(st.ct_pos, pi * 12 + k * 3 + 1) = out[k].pos.y;
    AT(st.ct_pos, pi * 12 + k * 3 + 2) = out[k].pos.z;
    AT(st.ct_sep, pi * 4 + k) = out[k].sep - m->cfg.rest_offset * 2.0f;
    for (int a = 0; a < 3; ++a) AT(st.ct_lam, pi * 12 + k * 3 + a) = lam[a];
  }
}

/* ---- solver ------------------------------------------------------------------------------ */
/* J += sgn * d(velocity of the body-fixed point p along dir)/d(generalized velocity) */
MSK_DEV void jac_point(const DModel* m, const DState& st, int N, int e, int body, v3 p, v3 dir, float sgn, float* J) {
  if (body < 0) return;
  const DBody* b = &m->bodies[body];
  if (b->kind == MSK_BODY_LINK) {
    sv6 F = {v3_cross(p, dir), dir};
    int j = body;
    while (j >= 0) {
      const DBody* bj = &m->bodies[j];
      if (bj->dof >= 0) {
        sv6 Sj;
        Sj.a = v3_make(AT(st.S, bj->dof * 6 + 0), AT(st.S, bj->dof * 6 + 1), AT(st.S, bj->dof * 6 + 2));
        Sj.l = v3_make(AT(st.S, bj->dof * 6 + 3), AT(st.S, bj->dof * 6 + 4), AT(st.S, bj->dof * 6 + 5));
        J[bj->dof] = fmaf(sgn, sv6_dot(Sj, F), J[bj->dof]);
      }
      j = bj->parent;
    }
  } else if (b->kind == MSK_BODY_DYNAMIC) {
    v3 r = v3_cross(v3_sub(p, load_v3(st.comw, body, N, e)), dir);
    J[b->vofs + 0] += sgn * dir.x; J[b->vofs + 1] += sgn * dir.y; J[b->vofs + 2] += sgn * dir.z;
    J[b->vofs + 3] += sgn * r.x; J[b->vofs + 4] += sgn * r.y; J[b->vofs + 5] += sgn * r.z;
  }
}

/* Y = A^-1 J^T, d = J.Y ; writes the row to the workspace (row stride NV, zero padded) */
template <int NV>
MSK_DEV void finish_row(const DModel* m, const DState& st, int N, int e, int row, const float* J) {
  const int nd = m->nd, nv = m->nv;
  float Y[MSK_MAX_NV];
  for (int i = 0; i < nd; ++i) {
    float a = 0.0f;
    for (int k = 0; k < nd; ++k) a = fmaf(AT(st.Minv, i * nd + k), J[k], a);
    Y[i] = a;
  }
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_DYNAMIC) continue;
    float im = 1.0f / b->mass;
    Y[b->vofs + 0] = J[b->vofs + 0] * im;
    Y[b->vofs + 1] = J[b->vofs + 1] * im;
    Y[b->vofs + 2] = J[b->vofs + 2] * im;
    v3 ja = v3_make(J[b->vofs + 3], J[b->vofs + 4], J[b->vofs + 5]);
    float Iwi[6];
    for (int k = 0; k < 6; ++k) Iwi[k] = AT(st.Iwinv, i * 6 + k);
    v3 ya = sym6_mulv(Iwi, ja);
    Y[b->vofs + 3] = ya.x; Y[b->vofs + 4] = ya.y; Y[b->vofs + 5] = ya.z;
  }
  float d = 0.0f;
  for (int k = 0; k < nv; ++k) {
    d = fmaf(J[k], Y[k], d);
    AT(st.rw_J, row * NV + k) = J[k];
    AT(st.rw_Y, row * NV + k) = Y[k];
  }
  for (int k = nv; k < NV; ++k) {
    AT(st.rw_J, row * NV + k) = 0.0f;
    AT(st.rw_Y, row * NV + k) = 0.0f;
  }
  AT(st.rw_d, row) = d;
}

enum { ROW_LIMLO = 0, ROW_LIMHI = 1, ROW_CN = 2, ROW_CT1 = 3, ROW_CT2 = 4 };

/* NV = generalized-velocity size padded to a compile-time constant: v[] and dq[] then live in
 * registers (every k loop is fully unrolled) instead of scratch memory. */
template <int NV>
__global__ void __launch_bounds__(64) k_solve(const DModel* __restrict__ m, DState st) {
  const int N = m->N;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= N) return;
  const int nv = m->nv, nd = m->nd;
  const float dt = m->cfg.timestep;
  const int Np = m->cfg.solver_position_iterations, Nv = m->cfg.solver_velocity_iterations;
  const float h = dt / (float)Np;

  float q[MSK_MAX_DOF], qd[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i) { q[i] = AT(st.q, i); qd[i] = AT(st.qd, i); }

  /* row table: kind/idx (idx = body for limits, contact slot code pair*4+k for contacts), mu, c0, lam */
  unsigned short rkind[MSK_MAX_ROWS];
  unsigned short ridx[MSK_MAX_ROWS];
  float rc0[MSK_MAX_ROWS], rlam[MSK_MAX_ROWS], rmu[MSK_MAX_ROWS];
  int nr = 0;
  float J[MSK_MAX_NV];
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    if (b->lim_lo < -1e30f && b->lim_hi > 1e30f) continue;
    for (int kind = ROW_LIMLO; kind <= ROW_LIMHI; ++kind) {
      for (int k = 0; k < nv; ++k) J[k] = 0.0f;
      J[b->dof] = (kind == ROW_LIMHI) ? -1.0f : 1.0f;
      finish_row<NV>(m, st, N, e, nr, J);
      rkind[nr] = kind; ridx[nr] = i; rlam[nr] = 0.0f; rmu[nr] = 0.0f;
      rc0[nr] = (kind == ROW_LIMLO) ? (q[b->dof] - b->lim_lo) : (b->lim_hi - q[b->dof]);
      nr++;
    }
  }
  int ncontacts = 0;
  int overflow = 0;
  for (int p = 0; p < m->np; ++p) {
    const int cnt = AT(st.ct_cnt, p);
    if (cnt == 0) continue;
    const DShape* A = &m->shapes[m->pairs[p].sa];
    const DShape* B = &m->shapes[m->pairs[p].sb];
    v3 n = v3_make(AT(st.ct_n, p * 3 + 0), AT(st.ct_n, p * 3 + 1), AT(st.ct_n, p * 3 + 2));
    v3 t1, t2;
    msk_tangents(n, &t1, &t2);
    const float mu = 0.5f * (A->df + B->df);
    for (int k = 0; k < cnt; ++k) {
      if (ncontacts >= MSK_MAX_CONTACTS) { overflow = 1; AT(st.ct_cnt, p) = k; break; }
      v3 pos = v3_make(AT(st.ct_pos, p * 12 + k * 3 + 0), AT(st.ct_pos, p * 12 + k * 3 + 1), AT(st.ct_pos, p * 12 + k * 3 + 2));
      v3 dirs[3] = {n, t1, t2};
      for (int a = 0; a < 3; ++a) {
        for (int kk = 0; kk < nv; ++kk) J[kk] = 0.0f;
        jac_point(m, st, N, e, A->body, pos, dirs[a], 1.0f, J);
        jac_point(m, st, N, e, B->body, pos, dirs[a], -1.0f, J);
        finish_row<NV>(m, st, N, e, nr, J);
        rkind[nr] = ROW_CN + a; ridx[nr] = (unsigned short)(p * 4 + k);
        rlam[nr] = AT(st.ct_lam, p * 12 + k * 3 + a);
        rmu[nr] = mu;
        rc0[nr] = AT(st.ct_sep, p * 4 + k);
        nr++;
      }
      ncontacts++;
    }
    if (overflow) {
      for (int pp = p + 1; pp < m->np; ++pp) AT(st.ct_cnt, pp) = 0;
      break;
    }
  }
  st.env_ncontacts[e] = ncontacts;
  if (overflow) atomicOr(st.env_overflow, 1);

  float v[NV], dq[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) { v[k] = (k < nv) ? AT(st.vfree, k) : 0.0f; dq[k] = 0.0f; }
  for (int ri = 0; ri < nr; ++ri)
    if (rlam[ri] != 0.0f) {
      const float l = rlam[ri];
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = fmaf(AT(st.rw_Y, ri * NV + k), l, v[k]);
    }

  for (int it = 0; it < Np + Nv; ++it) {
    const int posit = it < Np;
    for (int ri = 0; ri < nr; ++ri) {
      float jv = 0.0f, jdq = 0.0f;
      float Jr[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) Jr[k] = AT(st.rw_J, ri * NV + k);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        jv = fmaf(Jr[k], v[k], jv);
        jdq = fmaf(Jr[k], dq[k], jdq);
      }
      const float d = AT(st.rw_d, ri);
      const int kind = rkind[ri];
      const float lam0 = rlam[ri];
      float dl, nl;
      if (kind <= ROW_CN) {
        float cur = rc0[ri] + jdq;
        float bias;
        if (posit) bias = (cur > 0.0f) ? cur / h : fmaxf(cur * (MSK_PEN_BETA / dt), -MSK_MAX_DEPEN_VEL);
        else bias = (cur > 0.0f) ? cur / dt : 0.0f;
        dl = -(jv + bias) / d;
        nl = fmaxf(lam0 + dl, 0.0f);
      } else {
        float bias = posit ? jdq / h : 0.0f;
        dl = -(jv + bias) / d;
        float lim = rmu[ri] * rlam[ri - (kind - ROW_CN)];
        nl = fminf(fmaxf(lam0 + dl, -lim), lim);
      }
      dl = nl - lam0;
      rlam[ri] = nl;
      if (dl != 0.0f) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = fmaf(AT(st.rw_Y, ri * NV + k), dl, v[k]);
      }
    }
    if (posit) {
#pragma unroll
      for (int k = 0; k < NV; ++k) dq[k] = fmaf(h, v[k], dq[k]);
    }
  }

  /* impulse write-back (contact reports + next step's warm start) */
  for (int ri = 0; ri < nr; ++ri)
    if (rkind[ri] >= ROW_CN) {
      int code = ridx[ri];
      AT(st.ct_lam, (code >> 2) * 12 + (code & 3) * 3 + (rkind[ri] - ROW_CN)) = rlam[ri];
    }

  /* integrate (vs/dqs: dynamically indexed copies; v/dq themselves stay in registers) */
  float vs[NV], dqs[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) { vs[k] = v[k]; dqs[k] = dq[k]; }
  for (int i = 0; i < nd; ++i) {
    AT(st.qacc, i) = (vs[i] - qd[i]) / dt;
    q[i] += dqs[i];
    qd[i] = vs[i];
    AT(st.q, i) = q[i];
    AT(st.qd, i) = qd[i];
  }
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_DYNAMIC) continue;
    v3 dx = v3_make(dqs[b->vofs + 0], dqs[b->vofs + 1], dqs[b->vofs + 2]);
    v3 dr = v3_make(dqs[b->vofs + 3], dqs[b->vofs + 4], dqs[b->vofs + 5]);
    v3 cw = v3_add(load_v3(st.comw, i, N, e), dx);
    pose T = load_pose(st.bpose, i, N, e);
    quat qn = quat_normalize(quat_mul(quat_from_rotvec(dr), T.q));
    T.q = qn;
    T.p = v3_sub(cw, quat_rotate(qn, b->com));
    store_pose(st.bpose, i, N, e, T);
    store_v3(st.blin, i, N, e, v3_make(vs[b->vofs + 0], vs[b->vofs + 1], vs[b->vofs + 2]));
    store_v3(st.bang, i, N, e, v3_make(vs[b->vofs + 3], vs[b->vofs + 4], vs[b->vofs + 5]));
  }
  KinScratch s;
  kinematics(m, st, N, e, q, qd, &s, true);
}

/* ---- AoS <-> SoA converters ------------------------------------------------------------------ */
struct DBuffers { float* buf[MSK_BUF_COUNT]; int max_dof; };

__global__ void __launch_bounds__(256) k_apply(const DModel* __restrict__ m, DState st, DBuffers bf, unsigned mask, const int* __restrict__ art_dof0,
                                               const int* __restrict__ art_ndof) {
  const int N = m->N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  const float ox = AT(st.offsets, 0), oy = AT(st.offsets, 1), oz = AT(st.offsets, 2);
  for (int i = 0; i < m->nb; ++i) {
    const DBody* b = &m->bodies[i];
    const float* r = bf.buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * m->nb + i) * 13;
    const bool is_root = b->kind == MSK_BODY_LINK && b->parent < 0;
    if ((b->kind != MSK_BODY_LINK && (mask & MSK_APPLY_RIGID_DATA)) || (is_root && (mask & MSK_APPLY_ART_ROOT_POSE))) {
      /* rows the caller did not touch since the last fetch are left alone: (p + off) - off and
       * re-normalisation are not exact in fp32, and apply must not perturb untouched envs */
      const pose cur = load_pose(st.bpose, i, N, e);
      const bool same = (r[0] == cur.p.x + ox) && (r[1] == cur.p.y + oy) && (r[2] == cur.p.z + oz) &&
                        (r[3] == cur.q.w) && (r[4] == cur.q.x) && (r[5] == cur.q.y) && (r[6] == cur.q.z);
      if (!same) {
        pose T;
        T.p = v3_make(r[0] - ox, r[1] - oy, r[2] - oz);
        T.q = quat_normalize(quat_make(r[3], r[4], r[5], r[6]));
        store_pose(st.bpose, i, N, e, T);
      }
      if (b->kind == MSK_BODY_DYNAMIC) {
        store_v3(st.blin, i, N, e, v3_make(r[7], r[8], r[9]));
        store_v3(st.bang, i, N, e, v3_make(r[10], r[11], r[12]));
      }
    }
  }
  for (int a = 0; a < m->na; ++a)
    for (int j = 0; j < art_ndof[a]; ++j) {
      const int d = art_dof0[a] + j;
      const size_t row = ((size_t)e * m->na + a) * bf.max_dof + j;
      if (mask & MSK_APPLY_ART_QPOS) AT(st.q, d) = bf.buf[MSK_BUF_ART_QPOS][row];
      if (mask & MSK_APPLY_ART_QVEL) AT(st.qd, d) = bf.buf[MSK_BUF_ART_QVEL][row];
      if (mask & MSK_APPLY_ART_QF) AT(st.qf, d) = bf.buf[MSK_BUF_ART_QF][row];
      if (mask & MSK_APPLY_ART_TARGET_QPOS) AT(st.qt, d) = bf.buf[MSK_BUF_ART_TARGET_QPOS][row];
      if (mask & MSK_APPLY_ART_TARGET_QVEL) AT(st.qdt, d) = bf.buf[MSK_BUF_ART_TARGET_QVEL][row];
    }
}

__global__ void __launch_bounds__(256) k_fetch(const DModel* __restrict__ m, DState st, DBuffers bf, unsigned mask, const int* __restrict__ art_dof0,
                                               const int* __restrict__ art_ndof) {
  const int N = m->N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  const float ox = AT(st.offsets, 0), oy = AT(st.offsets, 1), oz = AT(st.offsets, 2);
  if (mask & MSK_FETCH_RIGID_DATA)
    for (int i = 0; i < m->nb; ++i) {
      float* r = bf.buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * m->nb + i) * 13;
      pose T = load_pose(st.bpose, i, N, e);
      v3 lv = load_v3(st.blin, i, N, e), av = load_v3(st.bang, i, N, e);
      r[0] = T.p.x + ox; r[1] = T.p.y + oy; r[2] = T.p.z + oz;
      r[3] = T.q.w; r[4] = T.q.x; r[5] = T.q.y; r[6] = T.q.z;
      r[7] = lv.x; r[8] = lv.y; r[9] = lv.z; r[10] = av.x; r[11] = av.y; r[12] = av.z;
    }
  for (int a = 0; a < m->na; ++a)
    for (int j = 0; j < art_ndof[a]; ++j) {
      const int d = art_dof0[a] + j;
      const size_t row = ((size_t)e * m->na + a) * bf.max_dof + j;
      if (mask & MSK_FETCH_ART_QPOS) bf.buf[MSK_BUF_ART_QPOS][row] = AT(st.q, d);
      if (mask & MSK_FETCH_ART_QVEL) bf.buf[MSK_BUF_ART_QVEL][row] = AT(st.qd, d);
      if (mask & MSK_FETCH_ART_QACC) bf.buf[MSK_BUF_ART_QACC][row] = AT(st.qacc, d);
      if (mask & MSK_FETCH_ART_TARGETS) {
        bf.buf[MSK_BUF_ART_TARGET_QPOS][row] = AT(st.qt, d);
        bf.buf[MSK_BUF_ART_TARGET_QVEL][row] = AT(st.qdt, d);
      }
    }
}

/* sum of contact impulses applied on body x by body y, per env, per queried pair */
__global__ void __launch_bounds__(256) k_query(const DModel* __restrict__ m, DState st, const int* __restrict__ qpairs, int nq, float* __restrict__ out) {
  const int N = m->N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  for (int qi = 0; qi < nq; ++qi) {
    const int x = qpairs[2 * qi], y = qpairs[2 * qi + 1];
    v3 sum = v3_make(0, 0, 0);
    for (int p = 0; p < m->np; ++p) {
      const int ba = m->shapes[m->pairs[p].sa].body, bb = m->shapes[m->pairs[p].sb].body;
      float sgn;
      if (ba == x && bb == y) sgn = 1.0f;
      else if (ba == y && bb == x) sgn = -1.0f;
      else continue;
      const int cnt = AT(st.ct_cnt, p);
      if (cnt == 0) continue;
      v3 n = v3_make(AT(st.ct_n, p * 3 + 0), AT(st.ct_n, p * 3 + 1), AT(st.ct_n, p * 3 + 2));
      v3 t1, t2;
      msk_tangents(n, &t1, &t2);
      for (int k = 0; k < cnt; ++k) {
        float l0 = AT(st.ct_lam, p * 12 + k * 3 + 0), l1 = AT(st.ct_lam, p * 12 + k * 3 + 1), l2 = AT(st.ct_lam, p * 12 + k * 3 + 2);
        v3 imp = v3_madd(v3_madd(v3_scale(n, l0), t1, l1), t2, l2);
        sum = v3_madd(sum, imp, sgn);
      }
    }
    float* o = out + ((size_t)e * nq + qi) * 3;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
  }
}

#endif
