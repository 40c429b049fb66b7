/*
 * orc_sim.c — one simulation substep of one env, scalar CPU oracle.
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product path.
 *
 * Stands in for `PhysxGpuSystem.step()` (mani_skill/envs/scene.py:379-380) under the
 * parameters of mani_skill/utils/structs/types.py:35-90 (TGS, 15 position + 1 velocity
 * iterations, contact_offset 0.02, friction every iteration).  Pipeline:
 *   1. kinematics      link frames from (root pose, q); joint axes; spatial velocities
 *   2. dynamics        CRBA joint-space inertia M, RNEA bias forces; joint PD drives and
 *                      tendon springs are integrated implicitly by folding them into the
 *                      system matrix A = M + dt*D + dt^2*K (+ tendon terms), Cholesky,
 *                      A^-1, unconstrained velocity v* = A^-1 (M v + dt*(tau - bias - K e + D vt));
 *                      a drive whose predicted force exceeds its limit leaves the system matrix and
 *                      becomes a row of the constraint solver (soft row, impulse clamped to
 *                      +-force_limit * dt: PhysX limits a drive's impulse inside the solve)
 *   3. collision       orc_collide.c over the static candidate pair table
 *   4. rows            joint limits (only while the joint can reach the limit within the step, the joint
 *                      counterpart of contact_offset), contact normal + 2 friction rows per
 *                      point, in generalized coordinates: J_k = (+-) S_k . F for every coordinate k
 *                      that moves the body (S_k = motion subspace column of coordinate k, free
 *                      bodies included), response Y = W J^T with W = block-diag(A^-1, 1/m, Iw^-1)
 *   5. TGS             in constraint space: A = J W J^T (row-by-row sequential dot products),
 *                      a = J v is carried per row instead of v, b = J dq likewise.  The
 *                      Np + Nv Gauss-Seidel sweeps the scene configures are spent as Nsub sub-steps
 *                      of dt / Nsub -- each one sweep WITH the position bias (errors re-linearised
 *                      from b; a row update adds its column of A, scaled by the impulse change, to a),
 *                      the advance of the rows' positions b += h a, and one sweep WITHOUT the bias that
 *                      takes the push-out back out of the velocity -- and the remaining (at least two)
 *                      sweeps without bias: 15 + 1 -> 7 x (1 + 1) + 2.  (One biased sweep per
 *                      iteration and a single velocity sweep left stacks and impacts unconverged:
 *                      DESIGN.md §2.)  v and dq are recovered at the end from the impulses:
 *                      v = v* + Y^T lambda, dq = h (Nsub v* + Y^T sum_sub lambda_sub)
 *   6. integrate       q += dq, free bodies x += dq_lin, R = exp(dq_ang) R; kinematics
 */
#include <stdio.h>
#include "orc_sim.h"
#include <string.h>

#ifndef ORC_PEN_RATE_COEF
#define ORC_PEN_RATE_COEF 2.0f   /* penetration recovery: bias = depth * 2 sqrt(1 / dt), the bias coefficient of PhysX's TGS contact
                                  * preparation [ext] (20 / s at 100 Hz; 0.8 / dt pumped the sway of a stack)           */
#endif
#define ORC_MAX_DEPEN_VEL 3.0f   /* m/s cap on the penetration-recovery bias            */
#define ORC_MAX_ROW_IMPULSE 1.0e3f /* N s per sweep on a limit / normal row (msk_solve.h MSK_MAX_ROW_IMPULSE) */
#define ORC_MIN_RESPONSE 1.0e-6f /* J W J^T below this: the row takes no impulse (msk_solve.h MSK_MIN_RESPONSE) */
#ifndef ORC_STATIC_LAST_WORD
#define ORC_STATIC_LAST_WORD 1   /* the extra pass over the normal rows against static bodies before each sub-step's advance (below) */
#endif
#define ORC_WARM_DIST 5.0e-3f    /* contact matching radius for warm starting             */
#ifndef ORC_FRICTION_ALIGN_SPEED
#define ORC_FRICTION_ALIGN_SPEED 1.0e-2f   /* m/s: below it the friction frame is the one orc_tangents() derives from the normal */
#endif
#ifndef ORC_WARM_NORMAL
#define ORC_WARM_NORMAL 1.0f     /* fraction of last step's normal impulses applied up front */
#define ORC_WARM_TANGENT 0.0f    /* ... and of its friction (tangential, torsional) impulses: none -- carried over, the tangential impulses of a manifold's
                                  * redundant points build up against each other from step to step and shake a stack (measured:
                                  * 1.0 / 0.9 leaves a five-cube stack swaying at 1.2 rad/s, 1.0 / 0.0 at rest to 1e-4)      */
#endif
#define ORC_LIMIT_SLACK 5.0e-3f   /* a joint-limit row exists while the joint can reach the limit in this step: distance < slack + twice what
                                  * its unconstrained velocity covers towards it in dt (the joint counterpart of the speculative contact rule, orc_collide.c) */
#ifndef ORC_MAX_JOINT_VELOCITY
#define ORC_LIMIT_BACKSTOP 0.01f /* a joint coordinate is never integrated further than this (rad, m) past a limit: the second line behind the limit rows.
                                  * Gauss-Seidel does not converge on a light chain that a stiff drive jams against the table (round 3's chain fuzzer:
                                  * limits overshot by up to 0.6 rad with the limit row present and unsaturated); the backstop puts the coordinate back
                                  * on the limit + backstop and takes the velocity into the limit away.  The Panda of the benchmarked tasks stays within
                                  * 0.0002 rad (arm) / 4 mm (fingers) of its limits: it never gets here. */
#define ORC_MAX_JOINT_VELOCITY 100.0f /* PhysX's default maxJointVelocity of a reduced-coordinate articulation joint (rad/s, m/s): the second line behind the drive rows */
#endif

enum { ROW_DRIVE, ROW_LIMLO, ROW_LIMHI, ROW_JFRIC, ROW_CN, ROW_CT1, ROW_CT2, ROW_TORS };

typedef struct {
  int kind, idx;
  int nref;     /* friction rows: the normal row whose impulse sizes the cone */
  float J[MSK_MAX_NV], Y[MSK_MAX_NV];
  float c0;     /* position-level error at the start of the step (separation / distance to the limit) */
  float rinv;   /* 1 / (J . Y + cfm) */
  float keep;   /* 1 - cfm * rinv: what a soft row (drive) keeps of its impulse; exactly 1 for rigid rows */
  float vbias;  /* drive rows: the velocity the row asks for with no impulse on it */
  float hi_c;   /* drive rows: force_limit * dt; joint-friction rows: coefficient * transmitted force * dt */
  float mu;     /* friction rows: the coefficient of this step (torsional row: times the patch radius) */
  float rest;   /* normal rows: restitution bias e * (J . v*) when the approach is faster than bounce_threshold, else 0 */
  float vclose; /* ... and the distance that approach covers in one step */
  float lam;    /* accumulated impulse */
  float a, b;   /* J . v and J . dq, carried in constraint space */
  float lsum;   /* sum of lam over the position iterations */
} orc_row;

typedef struct {
  sv6 S[MSK_MAX_BODIES];      /* joint motion subspace about the env origin (zero if fixed) */
  sv6 V[MSK_MAX_BODIES];      /* spatial velocity of links */
  v3 comw[MSK_MAX_BODIES];    /* world COM */
  float Iw[MSK_MAX_BODIES][6];/* world inertia about the COM */
  float Minv[MSK_MAX_DOF][MSK_MAX_DOF];
  float Iwinv[MSK_MAX_BODIES][6];
  float vfree[MSK_MAX_NV];
  /* per generalized coordinate k: motion subspace column about the env origin, row k of the
   * block-diagonal inverse mass matrix W (zero padded to npad), bodies moved by k (bit mask) */
  sv6 Scol[MSK_MAX_NV];
  float W[MSK_MAX_NV][MSK_MAX_NV];
  uint64_t moves[MSK_MAX_NV];
  int npad;
  /* drives taken out of the implicit system because their predicted force exceeds the limit: soft rows of the solver */
  int drv_on[MSK_MAX_DOF];
  float drv_cfm[MSK_MAX_DOF], drv_vbias[MSK_MAX_DOF], drv_hi[MSK_MAX_DOF];
} orc_scratch;

/* ---- 1. kinematics ---------------------------------------------------------------- */
/* Link frames down the tree.  What a link hands to its children is the composed frame U = U_parent * L with the rotation AS COMPOSED (not normalised), L = Xp * J(q) *
 * XcInv being the joint-local transform with a normalised rotation: one pose product per tree level is all that depends on the parent (the HIP kernels run the levels
 * one after the other; L, the normalisation of the published frame, the joint axis and the velocity sums are per-body work beside that chain).  The published frame
 * is (U.p, normalised U.q); |U.q| stays within a few ulp of 1 (a product of unit quaternions), so the un-normalised rotation costs nothing measurable.
 * Velocities: V = V_parent + S qd, acc = acc_parent + V_parent x (S qd), added for every link (S = 0, qd = 0 where the joint is fixed). */
static void kinematics(const orc_ctx* c, orc_env* e, orc_scratch* s) {
  pose U[MSK_MAX_BODIES];
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    s->S[i] = sv6_zero();
    s->V[i] = sv6_zero();
    U[i] = e->bpose[i];
    if (b->kind == MSK_BODY_LINK && b->parent >= 0) {
      pose Jq;
      Jq.p = v3_make(0, 0, 0);
      Jq.q = quat_make(1, 0, 0, 0);
      if (b->jtype == MSK_JOINT_REVOLUTE) {
        float sn, cs;
        orc_sincos(0.5f * e->q[b->dof], &sn, &cs);
        Jq.q = quat_make(cs, sn, 0, 0);
      } else if (b->jtype == MSK_JOINT_PRISMATIC) {
        Jq.p = v3_make(e->q[b->dof], 0, 0);
      }
      pose L = pose_mul(pose_mul(b->Xp, Jq), b->XcInv);
      L.q = quat_normalize(L.q);
      U[i] = pose_mul(U[b->parent], L);
      pose T;
      T.p = U[i].p;
      T.q = quat_normalize(U[i].q);
      e->bpose[i] = T;
      /* joint frame and axis from the parent's PUBLISHED frame */
      const pose Tj = pose_mul(e->bpose[b->parent], b->Xp);
      const v3 axis = quat_rotate(Tj.q, v3_make(1, 0, 0));
      if (b->jtype == MSK_JOINT_REVOLUTE) {
        s->S[i].a = axis;
        s->S[i].l = v3_cross(Tj.p, axis);
      } else if (b->jtype == MSK_JOINT_PRISMATIC) {
        s->S[i].l = axis;
      }
      s->V[i] = sv6_madd(s->V[b->parent], s->S[i], b->dof >= 0 ? e->qd[b->dof] : 0.0f);
    }
    m33 R = quat_to_m33(e->bpose[i].q);
    s->comw[i] = v3_add(e->bpose[i].p, m33_mulv(&R, b->com));
    if (b->kind == MSK_BODY_LINK && b->parent < 0 && b->root_dof >= 0) { /* floating root: (v of its centre of mass, omega) is state, as for a free body */
      const float* qr = e->qd + b->root_dof;
      const v3 vc = v3_make(qr[0], qr[1], qr[2]), w = v3_make(qr[3], qr[4], qr[5]);
      s->V[i].a = w;
      s->V[i].l = v3_add(vc, v3_cross(s->comw[i], w));   /* Pluecker: velocity of the point at the env origin */
    }
    sym6_rotate(&R, b->I6, s->Iw[i]);
    if (b->kind == MSK_BODY_LINK) {
      /* published velocities: angular, and linear velocity of the COM */
      e->bang[i] = s->V[i].a;
      e->blin[i] = v3_add(s->V[i].l, v3_cross(s->V[i].a, s->comw[i]));
    } else if (b->kind == MSK_BODY_DYNAMIC) {
      if (c->xb_slot[i] >= 0) { /* per-env instance: principal inverse inertia from the env's record */
        const float* x = c->xbody + ((size_t)(e - c->envs) * c->nxb + c->xb_slot[i]) * 8;
        const float Ii[6] = {x[1], x[2], x[3], 0.0f, 0.0f, 0.0f};
        sym6_rotate(&R, Ii, s->Iwinv[i]);
      } else sym6_rotate(&R, b->Iinv6, s->Iwinv[i]);
    } else {
      e->blin[i] = v3_make(0, 0, 0);
      e->bang[i] = v3_make(0, 0, 0);
    }
  }
}

void orc_forward_kinematics(const orc_ctx* c, orc_env* e) {
  orc_scratch s;
  kinematics(c, e, &s);
}

/* The six coordinates of a floating root are those of a free body: velocity of its centre of mass c (0..2), angular velocity (3..5).
 * Unit motion a as a spatial vector about the env origin: translation (0; e), rotation about the axis through c (e; c x e);
 * S_a . F for a spatial force F = (moment about the origin; force): the force component, the moment about c. */
static sv6 root_unit(int a, v3 c) {
  sv6 u = sv6_zero();
  const v3 ex = v3_make(a % 3 == 0 ? 1.0f : 0.0f, a % 3 == 1 ? 1.0f : 0.0f, a % 3 == 2 ? 1.0f : 0.0f);
  if (a < 3) u.l = ex;
  else { u.a = ex; u.l = v3_cross(c, ex); }
  return u;
}
static void root_project(sv6 F, v3 c, float out[6]) {
  const v3 mc = v3_add(F.a, v3_cross(F.l, c));
  out[0] = F.l.x; out[1] = F.l.y; out[2] = F.l.z; out[3] = mc.x; out[4] = mc.y; out[5] = mc.z;
}

/* ---- 2. dynamics ------------------------------------------------------------------ */
/* MSK_VP_GUARD (a CANDIDATE, not compiled by default: `make liborc_vpguard.so` = -DMSK_VP_GUARD=1.3f; the HIP side does not have it yet, so the default
 * oracle must not either): kinetic-energy gain of one step's velocity-product terms above which they are scaled back, 1.3 <=> (omega dt)^2 > 0.3. */
static void dynamics(const orc_ctx* c, orc_env* e, orc_scratch* s) {
  const float dt = c->cfg.timestep;
  const v3 g = v3_make(c->cfg.gravity[0], c->cfg.gravity[1], c->cfg.gravity[2]);
  const int nd = c->ndof;
  sinertia Isp[MSK_MAX_BODIES], Ic[MSK_MAX_BODIES];
  sv6 f[MSK_MAX_BODIES], fvp[MSK_MAX_BODIES];
  sv6 acc[MSK_MAX_BODIES];
  float M[MSK_MAX_DOF][MSK_MAX_DOF];
  float bias[MSK_MAX_DOF], bias_vp[MSK_MAX_DOF];
  memset(M, 0, sizeof(M));
  /* spatial inertias about the env origin, RNEA forward pass with zero joint accelerations */
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    v3 cw = s->comw[i];
    float m = b->mass;
    Isp[i].m = m;
    Isp[i].h = v3_scale(cw, m);
    float cc = v3_dot(cw, cw);
    Isp[i].I[0] = s->Iw[i][0] + m * (cc - cw.x * cw.x);
    Isp[i].I[1] = s->Iw[i][1] + m * (cc - cw.y * cw.y);
    Isp[i].I[2] = s->Iw[i][2] + m * (cc - cw.z * cw.z);
    Isp[i].I[3] = s->Iw[i][3] - m * (cw.x * cw.y);
    Isp[i].I[4] = s->Iw[i][4] - m * (cw.x * cw.z);
    Isp[i].I[5] = s->Iw[i][5] - m * (cw.y * cw.z);
    Ic[i] = Isp[i];
    if (b->parent < 0) {
      acc[i] = sv6_zero();
      if (b->root_dof >= 0) { /* the angular unit motions turn about the moving centre of mass: d/dt (c x e) = v_c x e */
        const float* qr = e->qd + b->root_dof;
        acc[i].l = v3_cross(v3_make(qr[0], qr[1], qr[2]), v3_make(qr[3], qr[4], qr[5]));
      }
    } else {
      const float qdi = b->dof >= 0 ? e->qd[b->dof] : 0.0f;      /* (every link adds its term: kinematics()) */
      const sv6 sq = {v3_scale(s->S[i].a, qdi), v3_scale(s->S[i].l, qdi)};
      acc[i] = sv6_add(acc[b->parent], sv6_crossm(s->V[b->parent], sq));
    }
    sv6 Iv = sinertia_mul(&Isp[i], s->V[i]);
    f[i] = sv6_add(sinertia_mul(&Isp[i], acc[i]), sv6_crossf(s->V[i], Iv));
    fvp[i] = f[i];   /* the velocity-product part alone (Coriolis, centrifugal, gyroscopic): what the energy guard below looks at */
    if (!b->nograv) {
      v3 mg = v3_scale(g, m);
      f[i].a = v3_sub(f[i].a, v3_cross(cw, mg));
      f[i].l = v3_sub(f[i].l, mg);
    }
    if (c->wrench_pending) { /* external force / torque at the link's centre of mass, this step only (cuda_rigid_body_force / _torque rows of a link) */
      const float* wr = c->wrench + ((size_t)(e - c->envs) * c->nb + i) * 8;
      if (wr[0] != 0.0f || wr[1] != 0.0f || wr[2] != 0.0f || wr[4] != 0.0f || wr[5] != 0.0f || wr[6] != 0.0f) {
        const v3 F = v3_make(wr[0], wr[1], wr[2]), Tq = v3_make(wr[4], wr[5], wr[6]);
        f[i].a = v3_sub(f[i].a, v3_add(v3_cross(cw, F), Tq));
        f[i].l = v3_sub(f[i].l, F);
      }
    }
  }
  /* backward pass: bias torques, composite inertias.  A link's accumulated wrench / composite inertia is its OWN term plus its descendants' own terms, added one by
   * one in DESCENDING body index (a flat sum per link: every link can do its own, no hand-off from level to level -- the HIP kernel's lanes each walk their
   * descendants' list; the nested form "parents absorb their children's sums" was a chain of tree-depth barriers there) */
  {
    sv6 fown[MSK_MAX_BODIES], fvpown[MSK_MAX_BODIES];
    for (int i = 0; i < c->nb; ++i) { fown[i] = f[i]; fvpown[i] = fvp[i]; }   /* (Isp keeps the links' own inertias) */
    for (int i = c->nb - 1; i >= 0; --i) {
      const orc_body* b = &c->bodies[i];
      if (b->kind != MSK_BODY_LINK) continue;
      for (int d = c->nb - 1; d > i; --d) {
        if (c->bodies[d].kind != MSK_BODY_LINK) continue;
        int a = c->bodies[d].parent;
        while (a > i) a = c->bodies[a].parent;      /* (a link's index is above its parent's) */
        if (a != i) continue;
        f[i] = sv6_add(f[i], fown[d]);
        fvp[i] = sv6_add(fvp[i], fvpown[d]);
        sinertia_acc(&Ic[i], &Isp[d]);
      }
      if (b->dof >= 0) { bias[b->dof] = sv6_dot(s->S[i], f[i]); bias_vp[b->dof] = sv6_dot(s->S[i], fvp[i]); }
      if (b->root_dof >= 0) { /* the accumulated wrench: force, moment about the root's centre of mass */
        root_project(f[i], s->comw[i], bias + b->root_dof);
        root_project(fvp[i], s->comw[i], bias_vp + b->root_dof);
      }
    }
  }
  /* CRBA */
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    sv6 F = sinertia_mul(&Ic[i], s->S[i]);
    M[b->dof][b->dof] = sv6_dot(s->S[i], F) + b->armature;
    int j = b->parent;
    while (j >= 0) {
      const orc_body* bj = &c->bodies[j];
      if (bj->dof >= 0) {
        float v = sv6_dot(s->S[j], F);
        M[b->dof][bj->dof] = v;
        M[bj->dof][b->dof] = v;
      }
      if (bj->root_dof >= 0) { /* coupling with the floating root's six unit motions */
        float Fc[6];
        root_project(F, s->comw[j], Fc);
        for (int a = 0; a < 6; ++a) { M[b->dof][bj->root_dof + a] = Fc[a]; M[bj->root_dof + a][b->dof] = Fc[a]; }
      }
      j = bj->parent;
    }
  }
  for (int i = 0; i < c->nb; ++i) { /* root blocks: the composite spatial inertia of the whole tree, column by column */
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->root_dof < 0) continue;
    for (int a = 0; a < 6; ++a) {
      const sv6 F = sinertia_mul(&Ic[i], root_unit(a, s->comw[i]));
      float Fc[6];
      root_project(F, s->comw[i], Fc);
      for (int r = 0; r < 6; ++r) M[b->root_dof + r][b->root_dof + a] = Fc[r];
    }
  }
  /* implicit PD drives / tendons: A = M + dt*D + dt^2*K.  A drive whose predicted force exceeds its limit is taken out of the
   * system again (second pass) and handed to the constraint solver as a soft row with its impulse clamped to +-fmax*dt: the
   * implicit spring-damper   lambda = -g v - dt K e + dt D vt,  g = dt (dt K + D)   written as   v + lambda / g + vbias = 0 */
  float Kd[MSK_MAX_DOF], Dd[MSK_MAX_DOF], fmaxd[MSK_MAX_DOF], err[MSK_MAX_DOF];
  for (int i = 0; i < nd; ++i) { Kd[i] = 0.0f; Dd[i] = 0.0f; fmaxd[i] = 0.0f; err[i] = 0.0f; s->drv_on[i] = 0; }   /* root coordinates: no drive */
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    Kd[b->dof] = b->K; Dd[b->dof] = b->D; fmaxd[b->dof] = b->fmax;
    if (b->drive_accel) { /* acceleration drive: gains are per unit of the joint's own inertia (the diagonal of the joint-space inertia) */
      Kd[b->dof] = b->K * M[b->dof][b->dof]; Dd[b->dof] = b->D * M[b->dof][b->dof];
    }
    err[b->dof] = e->q[b->dof] - e->qt[b->dof];
  }
  float A[MSK_MAX_DOF][MSK_MAX_DOF], L[MSK_MAX_DOF][MSK_MAX_DOF], rhs[MSK_MAX_DOF];
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < nd; ++i) {
      float mv = 0.0f;
      for (int k = 0; k < nd; ++k) { A[i][k] = M[i][k]; mv = fmaf(M[i][k], e->qd[k], mv); }
      A[i][i] += dt * fmaf(dt, Kd[i], Dd[i]);
      float tau = e->qf[i] - bias[i] - Kd[i] * err[i] + Dd[i] * e->qdt[i];
      rhs[i] = fmaf(dt, tau, mv);
    }
    for (int t = 0; t < c->nt; ++t) {
      const orc_tendon* tn = &c->tendons[t];
      float g2 = dt * fmaf(dt, tn->K, tn->D);
      float te = fmaf(tn->ca, e->q[tn->dof_a], tn->cb * e->q[tn->dof_b]) - tn->rest;
      A[tn->dof_a][tn->dof_a] += g2 * tn->ca * tn->ca;
      A[tn->dof_b][tn->dof_b] += g2 * tn->cb * tn->cb;
      A[tn->dof_a][tn->dof_b] += g2 * tn->ca * tn->cb;
      A[tn->dof_b][tn->dof_a] += g2 * tn->ca * tn->cb;
      rhs[tn->dof_a] -= dt * tn->K * te * tn->ca;
      rhs[tn->dof_b] -= dt * tn->K * te * tn->cb;
    }
    /* Cholesky A = L L^T (lower) */
    memset(L, 0, sizeof(L));
    for (int i = 0; i < nd; ++i) {
      for (int j = 0; j <= i; ++j) {
        float sum = A[i][j];
        for (int k = 0; k < j; ++k) sum = fmaf(-L[i][k], L[j][k], sum);
        if (i == j) L[i][i] = sqrtf(sum);
        else L[i][j] = sum / L[j][j];
      }
    }
    /* v* = A^-1 rhs */
    float y[MSK_MAX_DOF];
    for (int i = 0; i < nd; ++i) {
      float sum = rhs[i];
      for (int k = 0; k < i; ++k) sum = fmaf(-L[i][k], y[k], sum);
      y[i] = sum / L[i][i];
    }
    for (int i = nd - 1; i >= 0; --i) {
      float sum = y[i];
      for (int k = i + 1; k < nd; ++k) sum = fmaf(-L[k][i], s->vfree[k], sum);
      s->vfree[i] = sum / L[i][i];
    }
    if (pass == 1) break;
    /* drive force limits: predict the PD force at v*; where it exceeds the limit the drive becomes a solver row */
    int nsat = 0;
    for (int i = 0; i < nd; ++i) {
      if (Kd[i] == 0.0f && Dd[i] == 0.0f) continue;
      float F = -Kd[i] * fmaf(dt, s->vfree[i], err[i]) - Dd[i] * (s->vfree[i] - e->qdt[i]);
      /* ... and the force of the stalled joint (v = 0: a link that a contact holds back loses the damping term the free prediction counts on:
       * K err alone may be several times the limit while the prediction at v* stays under it) */
      const float Fstall = fmaf(Dd[i], e->qdt[i], -(Kd[i] * err[i]));
      if (fabsf(F) > fmaxd[i] || fabsf(Fstall) > fmaxd[i]) {
        const float cfm = 1.0f / (dt * fmaf(dt, Kd[i], Dd[i]));
        s->drv_on[i] = 1;
        s->drv_cfm[i] = cfm;
        s->drv_vbias[i] = (dt * fmaf(Kd[i], err[i], -(Dd[i] * e->qdt[i]))) * cfm;
        s->drv_hi[i] = fmaxd[i] * dt;
        Kd[i] = 0.0f; Dd[i] = 0.0f; err[i] = 0.0f;
        nsat++;
      }
    }
    if (nsat == 0) break;
  }
  /* Energy guard on the velocity-product terms.  They do no work (d/dt of v'Mv/2 is v'tau), but taken explicitly over a whole step they ADD
   * |dt M^-1 c|^2 -- a relative gain of (omega dt)^2: 1e-4 for an arm at 1 rad/s, 4 for the limbs of a floating humanoid thrashing at 200 rad/s, whose
   * root then doubles its speed every substep until it leaves fp32 (UnitreeG1Stand-v1 under full-range random actions).  v0 = A^-1 M v is the step
   * without them, vb = v0 - dt A^-1 c what they turn it into: where vb carries more than MSK_VP_GUARD times v0's kinetic energy, vb is scaled back to
   * that energy (the turn of the velocity stays, its growth goes) and v* moves by the difference.  Below the threshold nothing is touched: v* is the
   * single solve above, bit for bit.  Measured on the CPU suite (-DORC_VP_TRACE): deliberately fast test scenes reach a gain of 1.27 (a spinning chain in free
   * flight), the jammed light chains of the fuzzer 2.04; UnitreeG1Stand-v1 passes 1.6 -> 2.2 -> 3 -> 4.8 -> 10 -> 100 in its last six steps before it leaves fp32. */
#ifdef MSK_VP_GUARD
  if (nd > 0) {
    float r0[MSK_MAX_DOF], rc[MSK_MAX_DOF], v0[MSK_MAX_DOF], dv[MSK_MAX_DOF], y[MSK_MAX_DOF];
    for (int i = 0; i < nd; ++i) {
      float mv = 0.0f;
      for (int k = 0; k < nd; ++k) mv = fmaf(M[i][k], e->qd[k], mv);
      r0[i] = mv;
      rc[i] = -(dt * bias_vp[i]);
    }
    for (int which = 0; which < 2; ++which) {
      const float* r = which ? rc : r0;
      float* out = which ? dv : v0;
      for (int i = 0; i < nd; ++i) {
        float sum = r[i];
        for (int k = 0; k < i; ++k) sum = fmaf(-L[i][k], y[k], sum);
        y[i] = sum / L[i][i];
      }
      for (int i = nd - 1; i >= 0; --i) {
        float sum = y[i];
        for (int k = i + 1; k < nd; ++k) sum = fmaf(-L[k][i], out[k], sum);
        out[i] = sum / L[i][i];
      }
    }
    /* kinetic energies (twice) of v0 and vb = v0 + dv.  The whole energy, a floating tree's translation included: measured relative to the tree's centre
     * of mass the guard left the linear momentum alone, and the momentum error of the same explicit terms then walked the humanoid's root up to
     * 1e3 m/s and out of fp32 after 115 control steps (tried); bounding the whole energy bounds that too.  The price: where the guard acts on a flying
     * tree it takes linear momentum away -- at (omega dt)^2 > 0.5, where the step is not an integration any more anyway. */
    float vb[MSK_MAX_DOF];
    float T0 = 0.0f, Tb = 0.0f;
    for (int i = 0; i < nd; ++i) vb[i] = v0[i] + dv[i];
    for (int i = 0; i < nd; ++i) {
      float m0 = 0.0f, mb = 0.0f;
      for (int k = 0; k < nd; ++k) { m0 = fmaf(M[i][k], v0[k], m0); mb = fmaf(M[i][k], vb[k], mb); }
      T0 = fmaf(v0[i], m0, T0);
      Tb = fmaf(vb[i], mb, Tb);
    }
#ifdef ORC_VP_TRACE
    { static float worst = 0.0f; if (T0 > 1e-6f && Tb / T0 > worst) { worst = Tb / T0; fprintf(stderr, "VPTRACE %g (T0 %g)\n", worst, T0); } }
#endif
    if (Tb > MSK_VP_GUARD * T0 && T0 > 0.0f) {
      const float sc = sqrtf(T0 / Tb) - 1.0f;
      for (int i = 0; i < nd; ++i) s->vfree[i] = fmaf(sc, vb[i], s->vfree[i]);
    }
  }
#else
  (void)bias_vp;
#endif
  /* A^-1 column by column */
  for (int col = 0; col < nd; ++col) {
    float y[MSK_MAX_DOF];
    for (int i = 0; i < nd; ++i) {
      float sum = (i == col) ? 1.0f : 0.0f;
      for (int k = 0; k < i; ++k) sum = fmaf(-L[i][k], y[k], sum);
      y[i] = sum / L[i][i];
    }
    for (int i = nd - 1; i >= 0; --i) {
      float sum = y[i];
      for (int k = i + 1; k < nd; ++k) sum = fmaf(-L[k][i], s->Minv[k][col], sum);
      s->Minv[i][col] = sum / L[i][i];
    }
  }
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_DYNAMIC) continue;
    v3 v = e->blin[i], w = e->bang[i];
    if (!b->nograv) v = v3_madd(v, g, dt);
    if (c->wrench_pending) { /* external force / torque at the centre of mass, this step only (msk_apply FORCE / TORQUE) */
      const float* wr = c->wrench + ((size_t)(e - c->envs) * c->nb + i) * 8;
      const float mass_i = (c->xb_slot[i] >= 0) ? c->xbody[((size_t)(e - c->envs) * c->nxb + c->xb_slot[i]) * 8] : b->mass;
      v = v3_madd(v, v3_make(wr[0], wr[1], wr[2]), dt / mass_i);
      w = v3_madd(w, sym6_mulv(s->Iwinv[i], v3_make(wr[4], wr[5], wr[6])), dt);
    }
    float kl = fmaxf(0.0f, 1.0f - dt * b->lin_damp);
    float ka = fmaxf(0.0f, 1.0f - dt * b->ang_damp);
    v = v3_scale(v, kl);
    w = v3_scale(w, ka);
    if (b->lock) { /* locked world axes carry no velocity (PhysxRigidDynamicComponent.set_locked_motion_axes) */
      if (b->lock & 1u) v.x = 0.0f;
      if (b->lock & 2u) v.y = 0.0f;
      if (b->lock & 4u) v.z = 0.0f;
      if (b->lock & 8u) w.x = 0.0f;
      if (b->lock & 16u) w.y = 0.0f;
      if (b->lock & 32u) w.z = 0.0f;
    }
    s->vfree[b->vofs + 0] = v.x; s->vfree[b->vofs + 1] = v.y; s->vfree[b->vofs + 2] = v.z;
    s->vfree[b->vofs + 3] = w.x; s->vfree[b->vofs + 4] = w.y; s->vfree[b->vofs + 5] = w.z;
  }
}

/* ---- link incoming joint forces (px.cuda_articulation_link_incoming_joint_forces, structs/articulation.py:596-620) ----
 * Inverse dynamics of the state the last step left behind: with the link accelerations that go with qacc, the wrench a
 * link's parent transmits through the link's inbound joint is
 *     f_i = I_i a_i + v_i x* I_i v_i - (gravity + contact wrenches on link i) + sum over children f_c
 * (Featherstone RNEA backward pass; contact forces = the last step's impulses / dt, so drives, limits and contacts are all
 * accounted for).  Reported per link as [force | torque] at the origin of the joint's child frame, in that frame's axes
 * (x = joint axis); the root link's row is the wrench the world applies to the fixed base, in the root link's frame.
 * out: [nb][6], rows of bodies that are not links stay zero. */
void orc_link_joint_forces(const orc_ctx* c, orc_env* e, float* out) {
  orc_scratch s;
  const float inv_dt = 1.0f / c->cfg.timestep;
  const v3 g = v3_make(c->cfg.gravity[0], c->cfg.gravity[1], c->cfg.gravity[2]);
  sv6 f[MSK_MAX_BODIES], acc[MSK_MAX_BODIES];
  kinematics(c, e, &s);
  memset(out, 0, sizeof(float) * 6 * (size_t)c->nb);
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    sinertia Isp;
    const v3 cw = s.comw[i];
    const float m = b->mass, cc = v3_dot(cw, cw);
    Isp.m = m;
    Isp.h = v3_scale(cw, m);
    Isp.I[0] = s.Iw[i][0] + m * (cc - cw.x * cw.x);
    Isp.I[1] = s.Iw[i][1] + m * (cc - cw.y * cw.y);
    Isp.I[2] = s.Iw[i][2] + m * (cc - cw.z * cw.z);
    Isp.I[3] = s.Iw[i][3] - m * (cw.x * cw.y);
    Isp.I[4] = s.Iw[i][4] - m * (cw.x * cw.z);
    Isp.I[5] = s.Iw[i][5] - m * (cw.y * cw.z);
    if (b->parent < 0) {
      acc[i] = sv6_zero();
    } else {
      acc[i] = acc[b->parent];
      if (b->dof >= 0) {
        sv6 sq = {v3_scale(s.S[i].a, e->qd[b->dof]), v3_scale(s.S[i].l, e->qd[b->dof])};
        acc[i] = sv6_add(acc[i], sv6_crossm(s.V[b->parent], sq));
        acc[i] = sv6_madd(acc[i], s.S[i], e->qacc[b->dof]);
      }
    }
    sv6 Iv = sinertia_mul(&Isp, s.V[i]);
    f[i] = sv6_add(sinertia_mul(&Isp, acc[i]), sv6_crossf(s.V[i], Iv));
    if (!b->nograv) {
      v3 mg = v3_scale(g, m);
      f[i].a = v3_sub(f[i].a, v3_cross(cw, mg));
      f[i].l = v3_sub(f[i].l, mg);
    }
  }
  for (int k = 0; k < e->ncontacts; ++k) {   /* contact wrenches about the env origin: +F on body A, -F on body B */
    const orc_contact* ct = &e->contacts[k];
    v3 F = v3_scale(ct->n, ct->lam[0]);
    F = v3_madd(F, ct->t1, ct->lam[1]);
    F = v3_madd(F, ct->t2, ct->lam[2]);
    F = v3_scale(F, inv_dt);
    const v3 T = v3_madd(v3_cross(ct->pos, F), ct->n, ct->lam_t * inv_dt);   /* ... plus the couple of the torsional row */
    if (ct->ba >= 0 && c->bodies[ct->ba].kind == MSK_BODY_LINK) { f[ct->ba].a = v3_sub(f[ct->ba].a, T); f[ct->ba].l = v3_sub(f[ct->ba].l, F); }
    if (ct->bb >= 0 && c->bodies[ct->bb].kind == MSK_BODY_LINK) { f[ct->bb].a = v3_add(f[ct->bb].a, T); f[ct->bb].l = v3_add(f[ct->bb].l, F); }
  }
  for (int i = c->nb - 1; i >= 0; --i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    if (b->parent >= 0) f[b->parent] = sv6_add(f[b->parent], f[i]);
  }
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    const pose C = (b->parent >= 0) ? pose_mul(e->bpose[i], pose_inv(b->XcInv)) : e->bpose[i];
    const v3 torque = v3_sub(f[i].a, v3_cross(C.p, f[i].l));   /* moved from the env origin to the frame's origin */
    const v3 fl = quat_rotate(quat_conj(C.q), f[i].l), tl = quat_rotate(quat_conj(C.q), torque);
    float* o = out + 6 * i;
    o[0] = fl.x; o[1] = fl.y; o[2] = fl.z; o[3] = tl.x; o[4] = tl.y; o[5] = tl.z;
  }
}

/* ---- 3. collision ----------------------------------------------------------------- */
#define ORC_TRIM_CANDIDATES 512   /* device: MSK_TRIM_CANDIDATES */
/* capc: the contact points this env can take (the capacity in points, and what the joint blocks leave of the capacity in blocks).  More
 * than that: the capc DEEPEST points are kept (smallest separation; ties in (pair, point) order), of the first ORC_TRIM_CANDIDATES in
 * (pair, point) order -- a body whose contacts were dropped sinks, is the deepest at the next step and gets them back; dropping in pair
 * order instead let the last body of a crowded scene fall through the table (tests/test_contact_trimming.py). */
static void collide(const orc_ctx* c, orc_env* e, const int capc) {
  /* previous step's contacts, for warm starting: a new point inherits the impulses of the nearest
   * old point of the same shape pair if it lies within ORC_WARM_DIST */
  orc_contact prev[MSK_MAX_CONTACTS_WIDE];
  static _Thread_local orc_contact cand[ORC_TRIM_CANDIDATES];
  const int nprev = e->ncontacts;
  memcpy(prev, e->contacts, sizeof(orc_contact) * (size_t)nprev);
  int ncand = 0, total = 0;
  for (int p = 0; p < c->npairs; ++p) {
    orc_contact tmp[4];
    int n = orc_collide_pair(c, e, p, tmp);
    /* static or dynamic friction (PhysxMaterial.static_friction / dynamic_friction): the pair slides when the friction impulses of its
     * points, summed, ended the last step on the cone of the coefficient that step used -- then this step's cone is the dynamic one;
     * it sticks again when they end inside it.  (Per pair and from the impulses, not per point by position: a sliding point is not
     * where it was, and a lightly loaded corner of a resting box saturates without anything sliding.) */
    int pair_slip = 0;
    if (n > 0) {
      float T1 = 0.0f, T2 = 0.0f, Nn = 0.0f, mu_used = 0.0f;
      for (int j = 0; j < nprev; ++j)
        if (prev[j].sa == tmp[0].sa && prev[j].sb == tmp[0].sb) {
          T1 += prev[j].lam[1]; T2 += prev[j].lam[2]; Nn += prev[j].lam[0];
          mu_used = prev[j].slip ? prev[j].mu : prev[j].mu_s;
        }
      const float lim = 0.999f * mu_used * Nn;
      pair_slip = Nn > 0.0f && fmaf(T1, T1, T2 * T2) >= lim * lim;   /* (the friction frame turns with the motion: the length, not the components) */
    }
    total += n;
    for (int k = 0; k < n; ++k) {
      if (ncand >= ORC_TRIM_CANDIDATES) break;
      orc_contact* ct = &tmp[k];
      ct->lam[0] = ct->lam[1] = ct->lam[2] = 0.0f;
      ct->lam_t = 0.0f;
      ct->slip = pair_slip;
      int best = -1;
      float bd = ORC_WARM_DIST * ORC_WARM_DIST;
      for (int j = 0; j < nprev; ++j) {
        if (prev[j].sa != ct->sa || prev[j].sb != ct->sb) continue;
        float d2 = v3_len2(v3_sub(prev[j].pos, ct->pos));
        if (d2 < bd) { bd = d2; best = j; }
      }
      if (best >= 0)
      {
        ct->lam[0] = ORC_WARM_NORMAL * prev[best].lam[0];
        ct->lam[1] = ORC_WARM_TANGENT * prev[best].lam[1];
        ct->lam[2] = ORC_WARM_TANGENT * prev[best].lam[2];
        ct->lam_t = ORC_WARM_TANGENT * prev[best].lam_t;
      }
      cand[ncand++] = *ct;
    }
  }
  e->ncontacts = 0;
  if (total > capc) e->overflow = 1;
  for (int i = 0; i < ncand; ++i) {
    int rank = 0;
    if (total > capc)
      for (int j = 0; j < ncand; ++j) rank += cand[j].sep < cand[i].sep || (cand[j].sep == cand[i].sep && j < i);
    if (rank < capc) e->contacts[e->ncontacts++] = cand[i];
  }
}

/* ---- 4. rows ---------------------------------------------------------------------- */
/* sequential dot product over the padded width (entries >= nv are zero) */
static float dot_seq(const float* a, const float* b, int nv, int npad) {
  float acc = 0.0f;
  for (int k = 0; k < npad; ++k) acc = fmaf((k < nv) ? a[k] : 0.0f, (k < nv) ? b[k] : 0.0f, acc);
  return acc;
}

/* per-coordinate tables: Scol, W, moves */
static void coordinate_tables(const orc_ctx* c, const orc_env* e, orc_scratch* s) {
  const int nd = c->ndof, nv = c->nv;
  s->npad = (nv <= 16) ? 16 : ((nv <= 32) ? 32 : 64);
  for (int k = 0; k < nv; ++k) {
    s->Scol[k] = sv6_zero();
    s->moves[k] = 0;
    for (int j = 0; j < MSK_MAX_NV; ++j) s->W[k][j] = 0.0f;
  }
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind == MSK_BODY_LINK) {
      if (b->dof >= 0) s->Scol[b->dof] = s->S[i];
      if (b->root_dof >= 0)
        for (int a = 0; a < 6; ++a) s->Scol[b->root_dof + a] = root_unit(a, s->comw[i]);
      for (int j = i; j >= 0; j = c->bodies[j].parent) {
        if (c->bodies[j].dof >= 0) s->moves[c->bodies[j].dof] |= (uint64_t)1 << i;
        if (c->bodies[j].root_dof >= 0)
          for (int a = 0; a < 6; ++a) s->moves[c->bodies[j].root_dof + a] |= (uint64_t)1 << i;
      }
    } else if (b->kind == MSK_BODY_DYNAMIC) {
      const v3 ex[3] = {v3_make(1, 0, 0), v3_make(0, 1, 0), v3_make(0, 0, 1)};
      const float mass_i = (c->xb_slot[i] >= 0) ? c->xbody[((size_t)(e - c->envs) * c->nxb + c->xb_slot[i]) * 8] : b->mass;
      const float im = 1.0f / mass_i;
      const float* Ii = s->Iwinv[i];
      const float Im[3][3] = {{Ii[0], Ii[3], Ii[4]}, {Ii[3], Ii[1], Ii[5]}, {Ii[4], Ii[5], Ii[2]}};
      for (int a = 0; a < 3; ++a) {
        s->Scol[b->vofs + a].l = ex[a];                          /* v_com */
        s->Scol[b->vofs + 3 + a].a = ex[a];                      /* omega: point velocity = w x (p - c) */
        s->Scol[b->vofs + 3 + a].l = v3_cross(s->comw[i], ex[a]);
        s->moves[b->vofs + a] = s->moves[b->vofs + 3 + a] = (uint64_t)1 << i;
        /* ... and no response: the rows and columns of a locked axis are zero */
        s->W[b->vofs + a][b->vofs + a] = ((b->lock >> a) & 1u) ? 0.0f : im;
        for (int j = 0; j < 3; ++j) s->W[b->vofs + 3 + a][b->vofs + 3 + j] = (((b->lock >> (3 + a)) | (b->lock >> (3 + j))) & 1u) ? 0.0f : Im[a][j];
      }
    }
  }
  for (int i = 0; i < nd; ++i)
    for (int j = 0; j < nd; ++j) s->W[i][j] = s->Minv[i][j];
}

/* J[k] += sgn * S_k . F for every coordinate k that moves `body`; F = [p x dir; dir] */
static void jac_point(const orc_ctx* c, const orc_scratch* s, int body, v3 p, v3 dir, float sgn, float* J) {
  if (body < 0) return;
  sv6 F = {v3_cross(p, dir), dir};
  for (int k = 0; k < c->nv; ++k)
    if ((s->moves[k] >> body) & 1) J[k] = fmaf(sgn, sv6_dot(s->Scol[k], F), J[k]);
}

/* the same for a pure couple about `dir`: F = [dir; 0] */
static void jac_couple(const orc_ctx* c, const orc_scratch* s, int body, v3 dir, float sgn, float* J) {
  if (body < 0) return;
  sv6 F = {dir, v3_make(0, 0, 0)};
  for (int k = 0; k < c->nv; ++k)
    if ((s->moves[k] >> body) & 1) J[k] = fmaf(sgn, sv6_dot(s->Scol[k], F), J[k]);
}

/* Y = W J^T (full padded width, in coordinate order) */
static void finish_row(const orc_ctx* c, const orc_scratch* s, orc_row* r) {
  for (int i = 0; i < c->nv; ++i) {
    float a = 0.0f;
    for (int k = 0; k < s->npad; ++k) a = fmaf(s->W[i][k], (k < c->nv) ? r->J[k] : 0.0f, a);
    r->Y[i] = a;
  }
}

/* ---- 5./6. solve and integrate ----------------------------------------------------- */
void orc_step_env(const orc_ctx* c, orc_env* e) {
  static _Thread_local orc_row rows[4 * MSK_MAX_DOF + 4 * MSK_MAX_CONTACTS_WIDE];
  orc_scratch s;
  const int nv = c->nv, nd = c->ndof;
  const float dt = c->cfg.timestep;
  /* sweep schedule (header, item 5): T = Np + Nv sweeps = Nsub x (biased, relaxing) + the rest relaxing */
  const int T = (c->cfg.solver_position_iterations > 0 ? c->cfg.solver_position_iterations : 1) +
                (c->cfg.solver_velocity_iterations > 0 ? c->cfg.solver_velocity_iterations : 0);
  const int nsub = T >= 4 ? T / 2 - 1 : 1;
  const int nfinal = T - 2 * nsub > 0 ? T - 2 * nsub : 0;
  const float h = dt / (float)nsub;
  const float inv_h = 1.0f / h, inv_dt = 1.0f / dt, pen_rate = ORC_PEN_RATE_COEF * sqrtf(inv_dt);

  /* joint friction (PhysxArticulationJoint.friction, agents/controllers/pd_joint_pos.py:44-53): a row that holds the joint velocity at
   * zero with at most coefficient x |wrench the joint transmitted in the last step| x dt (PhysX: "the friction coefficient is unitless
   * and relates the magnitude of the spatial force transmitted from parent to child link to the maximal friction force") */
  float jf_lim[MSK_MAX_DOF];
  int any_jf = 0;
  for (int i = 0; i < c->nb; ++i)
    if (c->bodies[i].kind == MSK_BODY_LINK && c->bodies[i].dof >= 0 && c->bodies[i].jfriction > 0.0f) any_jf = 1;
  if (any_jf) {
    float w[MSK_MAX_BODIES * 6];
    orc_link_joint_forces(c, e, w);   /* of the state, acceleration and contact impulses the last step left */
    for (int i = 0; i < c->nb; ++i) {
      const orc_body* b = &c->bodies[i];
      if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
      const float* x = w + 6 * i;
      const float mag = sqrtf(fmaf(x[0], x[0], fmaf(x[1], x[1], fmaf(x[2], x[2], fmaf(x[3], x[3], fmaf(x[4], x[4], x[5] * x[5]))))));
      jf_lim[b->dof] = b->jfriction * mag * dt;
    }
  }

  kinematics(c, e, &s);
  dynamics(c, e, &s);
  coordinate_tables(c, e, &s);

  static _Thread_local float A[4 * MSK_MAX_DOF + 4 * MSK_MAX_CONTACTS_WIDE][4 * MSK_MAX_DOF + 4 * MSK_MAX_CONTACTS_WIDE];
  int nr = 0;
  /* joint blocks: force-limited drive (dynamics()), lower limit, upper limit -- whichever exist, joint by joint; the limits come
   * after the drive so that within a sweep they have the last word */
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    const int limited = !(b->lim_lo < -1e30f && b->lim_hi > 1e30f);
    for (int kind = ROW_DRIVE; kind <= ROW_LIMHI; ++kind) {
      float c0 = 0.0f;
      if (kind == ROW_DRIVE) {
        if (!s.drv_on[b->dof]) continue;
      } else {
        if (!limited) continue;
        c0 = (kind == ROW_LIMLO) ? (e->q[b->dof] - b->lim_lo) : (b->lim_hi - e->q[b->dof]);
        const float toward = (kind == ROW_LIMLO) ? -s.vfree[b->dof] : s.vfree[b->dof];
        if (!(c0 < fmaf(2.0f * dt, fmaxf(0.0f, toward), ORC_LIMIT_SLACK))) continue;
      }
      orc_row* r = &rows[nr++];
      memset(r, 0, sizeof(*r));
      r->kind = kind;
      r->idx = i;
      r->c0 = c0;
      r->J[b->dof] = (kind == ROW_LIMHI) ? -1.0f : 1.0f;
      finish_row(c, &s, r);
    }
  }
  if (any_jf) /* joint-friction blocks, joint by joint */
    for (int i = 0; i < c->nb; ++i) {
      const orc_body* b = &c->bodies[i];
      if (b->kind != MSK_BODY_LINK || b->dof < 0 || !(b->jfriction > 0.0f)) continue;
      orc_row* r = &rows[nr++];
      memset(r, 0, sizeof(*r));
      r->kind = ROW_JFRIC;
      r->idx = i;
      r->J[b->dof] = 1.0f;
      r->hi_c = jf_lim[b->dof];
      finish_row(c, &s, r);
    }
  /* Capacity: a block is a joint (drive, limits), a joint with friction, a contact point (normal, two tangents) or a torsional row, and
   * an env has cap_blocks of them (MSK_MAX_BLOCKS: one lane each on the device; MSK_MAX_BLOCKS_WIDE with msk_config.contact_capacity = 1: two per
   * lane).  Of more contact points than cap_contacts or than the joint blocks leave the deepest are kept (collide()), torsional rows get what
   * the points leave. */
  int nblocks = 0;
  for (int i = 0, last = -1; i < nr; ++i)
    if (rows[i].kind == ROW_JFRIC || rows[i].idx != last) { nblocks++; last = rows[i].idx; }
  {
    const int room = c->cap_blocks - nblocks > 0 ? c->cap_blocks - nblocks : 0;
    collide(c, e, room < c->cap_contacts ? room : c->cap_contacts);
  }
  nblocks += e->ncontacts;
  int first_contact_row = nr;
  for (int k = 0; k < e->ncontacts; ++k) {
    orc_contact* ct = &e->contacts[k];
    orc_tangents(ct->n, &ct->t1, &ct->t2);
    v3 dirs[3] = {ct->n, ct->t1, ct->t2};
    for (int a = 0; a < 3; ++a) {
      orc_row* r = &rows[nr++];
      memset(r, 0, sizeof(*r));
      r->kind = ROW_CN + a;
      r->idx = k;
      r->nref = first_contact_row + 3 * k;
      r->c0 = ct->sep;
      r->mu = ct->slip ? ct->mu : ct->mu_s;   /* static friction until the pair slides (PhysxMaterial.static_friction / dynamic_friction) */
      jac_point(c, &s, ct->ba, ct->pos, dirs[a], 1.0f, r->J);
      jac_point(c, &s, ct->bb, ct->pos, dirs[a], -1.0f, r->J);
      r->lam = ct->lam[a]; /* warm start */
    }
    /* The friction frame follows the motion.  Two tangential rows clamped to +-mu lam_n each are a friction PYRAMID: a body sliding
     * along the frame's diagonal is braked with sqrt(2) mu (and pulled off its course towards the diagonal).  PhysX's patch friction takes
     * the first friction direction along the relative tangential velocity; so here: if the point's unconstrained tangential velocity
     * (u1, u2) = (J_t1 . v*, J_t2 . v*) is more than ORC_FRICTION_ALIGN_SPEED, the two rows are rotated in the tangent plane so that row 1
     * points along it (J_1' = c J_1 + s J_2, J_2' = c J_2 - s J_1, (c, s) = (u1, u2) / |u|): sliding friction then acts along row 1, against
     * the motion, with mu lam_n whatever the direction.  The impulses are reported in the frame orc_tangents() gives (rotated back below). */
    {
      orc_row* r1 = &rows[nr - 2];
      orc_row* r2 = &rows[nr - 1];
      const float u1 = dot_seq(r1->J, s.vfree, nv, s.npad), u2 = dot_seq(r2->J, s.vfree, nv, s.npad);
      const float n2 = fmaf(u1, u1, u2 * u2);
      float fc = 1.0f, fs = 0.0f;
      if (n2 > ORC_FRICTION_ALIGN_SPEED * ORC_FRICTION_ALIGN_SPEED) {
        const float inv = 1.0f / sqrtf(n2);
        fc = u1 * inv; fs = u2 * inv;
        for (int q = 0; q < nv; ++q) {
          const float j1 = r1->J[q], j2 = r2->J[q];
          r1->J[q] = fmaf(fc, j1, fs * j2);
          r2->J[q] = fmaf(fc, j2, -(fs * j1));
        }
        const float l1 = r1->lam, l2 = r2->lam;
        r1->lam = fmaf(fc, l1, fs * l2);
        r2->lam = fmaf(fc, l2, -(fs * l1));
      }
      ct->fc = fc; ct->fs = fs;
      finish_row(c, &s, &rows[nr - 3]);
      finish_row(c, &s, r1);
      finish_row(c, &s, r2);
    }
  }
  for (int k = 0; k < e->ncontacts; ++k) { /* torsional rows: relative spin about the normal, blocks of their own behind the points'.  PhysX gives
                                             * one to a friction patch with a single anchor: here a pair with exactly one point (of those kept) */
    orc_contact* ct = &e->contacts[k];
    int same = 0;
    for (int j = 0; j < e->ncontacts; ++j) same += e->contacts[j].sa == ct->sa && e->contacts[j].sb == ct->sb;
    if (same != 1 || !(ct->patch_r > 0.0f || ct->min_patch_r > 0.0f)) { ct->lam_t = 0.0f; continue; }
    const float rp = fmaxf(ct->min_patch_r, sqrtf(fmaxf(0.0f, -ct->sep) * ct->patch_r));   /* PhysX: the patch grows with the penetration (0: the row is there and idle) */
    if (nblocks >= c->cap_blocks) { ct->lam_t = 0.0f; continue; }
    nblocks++;
    orc_row* r = &rows[nr++];
    memset(r, 0, sizeof(*r));
    r->kind = ROW_TORS;
    r->idx = k;
    r->nref = first_contact_row + 3 * k;
    r->mu = (ct->slip ? ct->mu : ct->mu_s) * rp;
    jac_couple(c, &s, ct->ba, ct->n, 1.0f, r->J);
    jac_couple(c, &s, ct->bb, ct->n, -1.0f, r->J);
    finish_row(c, &s, r);
    r->lam = ct->lam_t;
  }
  /* constraint-space operator and initial state: a = J (v* + Y^T lambda_0) */
  for (int i = 0; i < nr; ++i) {
    for (int r = 0; r < nr; ++r) A[i][r] = dot_seq(rows[i].J, rows[r].Y, nv, s.npad);
    /* a row without response of its own (two links with no relative freedom along the direction: PhysX's minimal-response test) takes
     * no impulse: rinv = 0 keeps lambda at its clamp of 0.  A drive row is soft: its compliance 1 / g adds to the response. */
    float arr = A[i][i];
    rows[i].keep = 1.0f;
    if (rows[i].kind == ROW_DRIVE) {
      const int d = c->bodies[rows[i].idx].dof;
      arr = A[i][i] + s.drv_cfm[d];
      rows[i].vbias = s.drv_vbias[d];
      rows[i].hi_c = s.drv_hi[d];
    }
    rows[i].rinv = arr > ORC_MIN_RESPONSE ? 1.0f / arr : 0.0f;
    if (rows[i].kind == ROW_DRIVE) rows[i].keep = fmaf(-s.drv_cfm[c->bodies[rows[i].idx].dof], rows[i].rinv, 1.0f);
    float a = dot_seq(rows[i].J, s.vfree, nv, s.npad);
    /* restitution (PhysxMaterial.restitution, scene bounce_threshold: structs/types.py:35-67): a normal row approaching faster
     * than the threshold aims at the rebound speed -e * (approach speed) instead of zero */
    if (rows[i].kind == ROW_CN) {
      const float er = e->contacts[rows[i].idx].rest;
      if (er > 0.0f && a < -c->cfg.bounce_threshold) { rows[i].rest = er * a; rows[i].vclose = -a * dt; }
    }
    for (int r = 0; r < nr; ++r) a = fmaf(A[i][r], rows[r].lam, a);
    rows[i].a = a;
  }

  for (int sw = 0; sw < 2 * nsub + nfinal; ++sw) {
    const int posit = sw < 2 * nsub && (sw & 1) == 0;   /* the biased sweep of a sub-step */
    for (int ri = 0; ri < nr; ++ri) {
      orc_row* r = &rows[ri];
      /* new impulse = clamp(lam * keep - (J.v + bias) / (J.Y + cfm)); the bias part does not depend on v and is folded first */
      float bias, lo, hi;
      if (r->kind == ROW_DRIVE || r->kind == ROW_JFRIC) {
        bias = r->vbias;
        hi = r->hi_c; lo = -hi;
      } else if (r->kind <= ROW_CN) {
        const float cur = r->c0 + r->b;
        if (posit) bias = (cur > 0.0f) ? cur * inv_h : fmaxf(cur * pen_rate, -ORC_MAX_DEPEN_VEL);
        else bias = (cur > 0.0f) ? cur * inv_dt : 0.0f;
        /* bounce: once the gap is closed (biased sweeps) or would be eaten by the approach allowance of the next step (relaxing sweeps) */
        if (r->rest < 0.0f) {
          if (posit) { if (!(cur > 0.0f)) bias = fminf(bias, r->rest); }
          else if (cur < r->vclose) bias = r->rest;
        }
        lo = 0.0f; hi = ORC_MAX_ROW_IMPULSE;
      } else { /* friction: tangential rows hold their anchor within the step, the torsional row only brakes */
        bias = (posit && r->kind != ROW_TORS) ? r->b * inv_h : 0.0f;
        hi = r->mu * rows[r->nref].lam; lo = -hi; /* == fma(+-mu, lam_n, 0) on the device; lam_n: the point's normal row, already swept */
      }
      const float t0 = fmaf(r->lam, r->keep, -(bias * r->rinv));
      const float nl = fminf(fmaxf(fmaf(-r->a, r->rinv, t0), lo), hi);
      const float dl = nl - r->lam;
      r->lam = nl;
      for (int i = 0; i < nr; ++i) rows[i].a = fmaf(A[i][ri], dl, rows[i].a);
    }
#if ORC_STATIC_LAST_WORD
    /* Static geometry has the last word before positions move.  One Gauss-Seidel sweep per sub-step leaves a chain static - light body -
     * heavy / driven body unconverged (contraction m_heavy / (m_heavy + m_light) per sweep), and advancing with that velocity pushes the
     * light body INTO the static one: a squeeze leaks through the table.  So after the biased sweep the normal rows against static and
     * kinematic bodies are visited once more, in row order: each may only ADD impulse, and only what stops the approach (gap / h for an
     * open gap, zero otherwise -- no recovery push on top).  What is left of the leak then sits between the movable bodies, where it is
     * pushed out again instead of being lost through the floor.  Almost always no row wants anything (PickCube under random actions:
     * 0.65 % of the passes), and then the pass is exactly a no-op: the device skips it behind one ballot. */
    if (posit)
      for (int ri = 0; ri < nr; ++ri) {
        orc_row* r = &rows[ri];
        if (r->kind != ROW_CN || r->rest < 0.0f) continue;
        const orc_contact* ct = &e->contacts[r->idx];
        const int fixed_a = ct->ba < 0 || !c->bodies[ct->ba].movable, fixed_b = ct->bb < 0 || !c->bodies[ct->bb].movable;   /* static, kinematic, or a link no joint moves */
        if (!(fixed_a || fixed_b)) continue;
        const float cur = r->c0 + r->b;
        const float bias = (cur > 0.0f) ? cur * inv_h : 0.0f;
        const float nl = fminf(fmaxf(fmaf(-r->a, r->rinv, fmaf(r->lam, r->keep, -(bias * r->rinv))), 0.0f), ORC_MAX_ROW_IMPULSE);
        if (!(nl > r->lam)) continue;
        const float dl = nl - r->lam;
        r->lam = nl;
        for (int i = 0; i < nr; ++i) rows[i].a = fmaf(A[i][ri], dl, rows[i].a);
      }
#endif
    if (posit) /* the sub-step's advance: the rows' positions move on with the biased velocity, the sub-step's impulse is booked */
      for (int i = 0; i < nr; ++i) {
        rows[i].b = fmaf(h, rows[i].a, rows[i].b);
        rows[i].lsum += rows[i].lam;
      }
  }
  /* back to generalized coordinates */
  float v[MSK_MAX_NV], dq[MSK_MAX_NV];
  for (int k = 0; k < nv; ++k) {
    float vk = s.vfree[k], sk = (float)nsub * s.vfree[k];
    for (int r = 0; r < nr; ++r) {
      vk = fmaf(rows[r].Y[k], rows[r].lam, vk);
      sk = fmaf(rows[r].Y[k], rows[r].lsum, sk);
    }
    v[k] = vk;
    dq[k] = h * sk;
  }

  /* contact impulses for the reports */
  for (int ri = 0; ri < nr; ++ri)
  {
    if (rows[ri].kind == ROW_CN) e->contacts[rows[ri].idx].lam[0] = rows[ri].lam;
    if (rows[ri].kind == ROW_CT1) { /* the two friction impulses back into the frame of orc_tangents() (the rows were rotated by (fc, fs)) */
      orc_contact* ct = &e->contacts[rows[ri].idx];
      const float l1 = rows[ri].lam, l2 = rows[ri + 1].lam;
      ct->lam[1] = fmaf(ct->fc, l1, -(ct->fs * l2));
      ct->lam[2] = fmaf(ct->fs, l1, ct->fc * l2);
    }
    if (rows[ri].kind == ROW_TORS) e->contacts[rows[ri].idx].lam_t = rows[ri].lam;
  }


  /* integrate */
  for (int i = 0; i < c->nb; ++i) { /* PhysX's maxJointVelocity: joint coordinates only (a floating root's six are a body's velocity) */
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0) continue;
    v[b->dof] = fminf(fmaxf(v[b->dof], -ORC_MAX_JOINT_VELOCITY), ORC_MAX_JOINT_VELOCITY);
    dq[b->dof] = fminf(fmaxf(dq[b->dof], -ORC_MAX_JOINT_VELOCITY * dt), ORC_MAX_JOINT_VELOCITY * dt);
  }
  for (int i = 0; i < c->nb; ++i) { /* the backstop behind the limit rows (ORC_LIMIT_BACKSTOP) */
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->dof < 0 || (b->lim_lo < -1e30f && b->lim_hi > 1e30f)) continue;
    const float qn = e->q[b->dof] + dq[b->dof];
    if (qn > b->lim_hi + ORC_LIMIT_BACKSTOP) { dq[b->dof] = (b->lim_hi + ORC_LIMIT_BACKSTOP) - e->q[b->dof]; v[b->dof] = fminf(v[b->dof], 0.0f); }
    else if (qn < b->lim_lo - ORC_LIMIT_BACKSTOP) { dq[b->dof] = (b->lim_lo - ORC_LIMIT_BACKSTOP) - e->q[b->dof]; v[b->dof] = fmaxf(v[b->dof], 0.0f); }
  }
  for (int i = 0; i < nd; ++i) {
    e->qacc[i] = (v[i] - e->qd[i]) / dt;
    e->q[i] += dq[i];
    e->qd[i] = v[i];
  }
  for (int i = 0; i < c->nb; ++i) { /* floating roots: integrated like the free bodies below */
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK || b->root_dof < 0) continue;
    const int rd = b->root_dof;
    const v3 dx = v3_make(dq[rd + 0], dq[rd + 1], dq[rd + 2]), dr = v3_make(dq[rd + 3], dq[rd + 4], dq[rd + 5]);
    const v3 cw = v3_add(s.comw[i], dx);
    const quat qn = quat_normalize(quat_mul(quat_from_rotvec(dr), e->bpose[i].q));
    e->bpose[i].q = qn;
    e->bpose[i].p = v3_sub(cw, quat_rotate(qn, b->com));
    for (int a = 0; a < 6; ++a) e->q[rd + a] = 0.0f;   /* these slots carry no position: the pose does */
  }
  for (int i = 0; i < c->nb; ++i) {
    const orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_DYNAMIC) continue;
    v3 dx = v3_make(dq[b->vofs + 0], dq[b->vofs + 1], dq[b->vofs + 2]);
    v3 dr = v3_make(dq[b->vofs + 3], dq[b->vofs + 4], dq[b->vofs + 5]);
    v3 cw = v3_add(s.comw[i], dx);
    quat qn = quat_normalize(quat_mul(quat_from_rotvec(dr), e->bpose[i].q));
    e->bpose[i].q = qn;
    e->bpose[i].p = v3_sub(cw, quat_rotate(qn, b->com));
    e->blin[i] = v3_make(v[b->vofs + 0], v[b->vofs + 1], v[b->vofs + 2]);
    e->bang[i] = v3_make(v[b->vofs + 3], v[b->vofs + 4], v[b->vofs + 5]);
  }
  kinematics(c, e, &s);
}
