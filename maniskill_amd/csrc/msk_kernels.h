/*
 * msk_kernels.h — the HIP kernels of one physics substep (gfx950, wave64).
 *
 * Launch shapes (N envs, P candidate pairs):
 *   k_dynamics   <<<N/64, 64>>>          one lane per env: link frames, CRBA, RNEA, implicit-PD
 *                                         system matrix, Cholesky, A^-1, unconstrained velocity
 *   k_collide    <<<(N/64, P), 64>>>     one lane per (pair, env); the pair is wave-uniform
 *   k_solve<G>   <<<N*G/64, 64>>>        G lanes per env (one per generalized coordinate): msk_solve.h
 *   k_apply / k_fetch / k_kinematics / k_query   memcpy-class layout converters (AoS rows <-> SoA)
 * All per-env data is SoA with env fastest (msk_model.h), so a wave's accesses coalesce.
 * Arithmetic order mirrors the CPU oracle statement for statement (bitwise parity target).
 */
#ifndef MSK_KERNELS_H
#define MSK_KERNELS_H

#include "msk_collide.h"
#include "msk_solve.h"

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
  const int G = m->G;
  const float dt = m->cfg.timestep;
  const v3 g = v3_make(m->cfg.gravity[0], m->cfg.gravity[1], m->cfg.gravity[2]);
  float q[MSK_MAX_DOF], qd[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i) { q[i] = AT(st.q, i); qd[i] = AT(st.qd, i); }
  KinScratch s;
  kinematics(m, st, N, e, q, qd, &s, true);
  /* W: block-diagonal inverse mass matrix, row k for lane k of the solver */
  float* Wenv = st.W + (size_t)e * G * G;   /* entries outside the blocks stay zero from allocation */
  float* vfenv = st.vfree + (size_t)e * G;

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
    /* motion subspace column of coordinate dof, for the row assembly */
    float* sc = st.Scol + ((size_t)e * G + b->dof) * 8;
    sc[0] = s.S[i].a.x; sc[1] = s.S[i].a.y; sc[2] = s.S[i].a.z; sc[3] = s.S[i].l.x; sc[4] = s.S[i].l.y; sc[5] = s.S[i].l.z;
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
  for (int i = 0; i < nd; ++i) vfenv[i] = vfree[i];
  /* A^-1 column by column */
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
    for (int i = 0; i < nd; ++i) Wenv[i * G + col] = x[i];
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
    const int o = b->vofs;
    vfenv[o + 0] = v.x; vfenv[o + 1] = v.y; vfenv[o + 2] = v.z;
    vfenv[o + 3] = w.x; vfenv[o + 4] = w.y; vfenv[o + 5] = w.z;
    m33 R = quat_to_m33(s.bpose[i].q);
    float Ii[6];
    sym6_rotate(&R, b->Iinv6, Ii);
    const float im = 1.0f / b->mass;
    const float Im[3][3] = {{Ii[0], Ii[3], Ii[4]}, {Ii[3], Ii[1], Ii[5]}, {Ii[4], Ii[5], Ii[2]}};
    const v3 ex[3] = {v3_make(1, 0, 0), v3_make(0, 1, 0), v3_make(0, 0, 1)};
    for (int a = 0; a < 3; ++a) {
      Wenv[(o + a) * G + o + a] = im;
      for (int j = 0; j < 3; ++j) Wenv[(o + 3 + a) * G + o + 3 + j] = Im[a][j];
      float* sl = st.Scol + ((size_t)e * G + o + a) * 8;       /* v_com */
      sl[0] = 0.0f; sl[1] = 0.0f; sl[2] = 0.0f; sl[3] = ex[a].x; sl[4] = ex[a].y; sl[5] = ex[a].z;
      float* sa = st.Scol + ((size_t)e * G + o + 3 + a) * 8;   /* omega: point velocity = w x (p - c) */
      const v3 cl = v3_cross(s.comw[i], ex[a]);
      sa[0] = ex[a].x; sa[1] = ex[a].y; sa[2] = ex[a].z; sa[3] = cl.x; sa[4] = cl.y; sa[5] = cl.z;
    }
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
  int* cntp = st.ct_cnt + (size_t)e * m->npp + pi;
  float* rec = st.ct_rec + ((size_t)e * m->npp + pi) * MSK_CT_REC;
  const int nprev = *cntp;
  v3 ppos[4];
  float plam[4][3];
  for (int j = 0; j < 4; ++j) {
    if (j < nprev) {
      ppos[j] = v3_make(rec[4 + j * 3 + 0], rec[4 + j * 3 + 1], rec[4 + j * 3 + 2]);
      for (int a = 0; a < 3; ++a) plam[j][a] = rec[20 + j * 3 + a];
    }
  }
  if (n == 0 && nprev == 0) return;
  *cntp = n;
  if (n > 0) { rec[0] = out[0].n.x; rec[1] = out[0].n.y; rec[2] = out[0].n.z; }
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
    rec[4 + k * 3 + 0] = out[k].pos.x;
    rec[4 + k * 3 + 1] = out[k].pos.y;
    rec[4 + k * 3 + 2] = out[k].pos.z;
    rec[16 + k] = out[k].sep - m->cfg.rest_offset * 2.0f;
    for (int a = 0; a < 3; ++a) rec[20 + k * 3 + a] = lam[a];
  }
}

/* ---- AoS <-> SoA converters ------------------------------------------------------------------ */
struct DBuffers { float* buf[MSK_BUF_COUNT]; int max_dof; };

__global__ void __launch_bounds__(256) k_apply(const DModel* __restrict__ m, DState st, DBuffers bf, unsigned mask, const int* __restrict__ art_dof0,
                                               const int* __restrict__ art_ndof) {
  const int N = m->N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  const float ox = AT(st.offsets, 0), oy = AT(st.offsets, 1), oz = AT(st.offsets, 2);
  bool teleported = false; /* a pose or joint position was overwritten: the env's contact cache is stale */
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
        teleported = true;
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
      if (mask & MSK_APPLY_ART_QPOS) {
        const float nq = bf.buf[MSK_BUF_ART_QPOS][row];
        if (nq != AT(st.q, d)) teleported = true;
        AT(st.q, d) = nq;
      }
      if (mask & MSK_APPLY_ART_QVEL) AT(st.qd, d) = bf.buf[MSK_BUF_ART_QVEL][row];
      if (mask & MSK_APPLY_ART_QF) AT(st.qf, d) = bf.buf[MSK_BUF_ART_QF][row];
      if (mask & MSK_APPLY_ART_TARGET_QPOS) AT(st.qt, d) = bf.buf[MSK_BUF_ART_TARGET_QPOS][row];
      if (mask & MSK_APPLY_ART_TARGET_QVEL) AT(st.qdt, d) = bf.buf[MSK_BUF_ART_TARGET_QVEL][row];
    }
  if (teleported) { /* no warm start across a teleport: replays from a state are reproducible */
    int* cnts = st.ct_cnt + (size_t)e * m->npp;
    for (int p = 0; p < m->np; ++p) cnts[p] = 0;
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
  const int* cnts = st.ct_cnt + (size_t)e * m->npp;
  const float* recs = st.ct_rec + (size_t)e * m->npp * MSK_CT_REC;
  for (int qi = 0; qi < nq; ++qi) {
    const int x = qpairs[2 * qi], y = qpairs[2 * qi + 1];
    v3 sum = v3_make(0, 0, 0);
    for (int p = 0; p < m->np; ++p) {
      const int ba = m->pinfo[p].ba, bb = m->pinfo[p].bb;
      float sgn;
      if (ba == x && bb == y) sgn = 1.0f;
      else if (ba == y && bb == x) sgn = -1.0f;
      else continue;
      const int cnt = cnts[p];
      if (cnt == 0) continue;
      const float* rec = recs + (size_t)p * MSK_CT_REC;
      v3 n = v3_make(rec[0], rec[1], rec[2]);
      v3 t1, t2;
      msk_tangents(n, &t1, &t2);
      for (int k = 0; k < cnt; ++k) {
        float l0 = rec[20 + k * 3 + 0], l1 = rec[20 + k * 3 + 1], l2 = rec[20 + k * 3 + 2];
        v3 imp = v3_madd(v3_madd(v3_scale(n, l0), t1, l1), t2, l2);
        sum = v3_madd(sum, imp, sgn);
      }
    }
    float* o = out + ((size_t)e * nq + qi) * 3;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
  }
}

#endif
