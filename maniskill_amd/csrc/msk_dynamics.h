/*
 * msk_dynamics.h — kinematics + articulation dynamics, one wavefront per env (gfx950, wave64).
 *
 * (two envs share a wavefront, 32 lanes each, when the template has <= 32 bodies and <= 15 dofs; the per-lane rows of the
 *  joint-space matrices are register arrays of MD = 16 entries then, of MD = 32 for up to 31 dofs with a wavefront per env, of MD = 64
 *  for up to 63 -- humanoids; those arrays live in scratch memory then)
 *
 *   lane i  <-> body i   link frames / velocities / RNEA, level-synchronous over the tree depth
 *                        (a lane reads its parent's pose, V, acc from LDS; parents gather the
 *                        children's forces and composite inertias on the way back)
 *   lane i  <-> dof i    CRBA row, implicit-PD system matrix row, Cholesky row
 *   lane c  <-> column c of A^-1 (c < nd) and lane 16 <-> the unconstrained velocity: 17 independent
 *                        triangular solves run side by side, each in the oracle's sequential order
 *
 * Everything a lane owns stays in registers; LDS carries only what crosses lanes (parent/child
 * hand-offs, the matrices M and L).  Every per-element fmaf chain is the oracle's (oracle/orc_sim.c
 * kinematics()/dynamics()), so results are bit-identical while the serial depth drops from
 * O(bodies + dof^3) per lane to O(tree depth + dof) per wave.
 *
 * Outputs: link poses / velocities and world COMs in the env record, and the solver tables
 * W (block-diagonal inverse mass matrix), Scol (motion subspace columns), vfree.
 */
#ifndef MSK_DYNAMICS_H
#define MSK_DYNAMICS_H

#include "msk_model.h"
#include "msk_broadphase.h"

MSK_DEV void dyn_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

/* dynamic LDS carve (floats) for a template with nb bodies */
struct DynLds {
  int pose, S, V, acc, meta, Ic, M, L, vec, total;
  __host__ __device__ DynLds(int nb, int md) {
    int o = 0;
    pose = o; o += nb * 8;
    S = o; o += nb * 6;
    V = o; o += nb * 6;
    acc = o; o += nb * 6;            /* acc on the way down (16-byte aligned: nb * 20 floats precede it) ... */
    Ic = o; o += nb * 10;            /* ... the cross terms of the forward pass; on the way back the two regions together are ONE array of 16 floats per body, */
                                     /* [f (6) | composite inertia (10)]: a link's record is four ds_read_b128 for whoever sums it up (phase 3) */
    meta = o; o += nb * 2;           /* per body: its joint's dof, its root_dof (int bits): what the CRBA rows ask of their ancestors */
    M = o; o += md * (md + 1);
    L = o; o += md * (md + 1);
#ifdef MSK_VP_GUARD
    fvp = o; o += nb * 6;            /* the velocity-product part of f on the way back */
    vec = o; o += 14 * md;           /* ... | bias_vp | M qd | v0 | vb | M v0 | M vb */
#else
    vec = o; o += 8 * md;            /* qd | bias | Kd | Dd | fconst | err | rhs | vfree */
#endif
    total = (o + 3) & ~3;            /* (every env's carve starts 16-byte aligned) */
  }
#ifdef MSK_VP_GUARD
  int fvp;
#endif
};
enum { DV_QD = 0, DV_BIAS = 1, DV_KD = 2, DV_DD = 3, DV_FC = 4, DV_ERR = 5, DV_RHS = 6, DV_VF = 7, DV_BVP = 8, DV_R0 = 9, DV_V0 = 10, DV_VB = 11, DV_M0 = 12, DV_MB = 13 };

MSK_DEV sv6 lds_sv6(const float* p) { sv6 r = {v3_make(p[0], p[1], p[2]), v3_make(p[3], p[4], p[5])}; return r; }
MSK_DEV void lds_put_sv6(float* p, sv6 v) { p[0] = v.a.x; p[1] = v.a.y; p[2] = v.a.z; p[3] = v.l.x; p[4] = v.l.y; p[5] = v.l.z; }

/* The six coordinates of a floating root are those of a free body: velocity of its centre of mass c (0..2), angular velocity (3..5).
 * Unit motion a as a spatial vector about the env origin: translation (0; e), rotation about the axis through c (e; c x e);
 * S_a . F for a spatial force F = (moment about the origin; force): the force component, the moment about c.  (oracle: root_unit / root_project) */
MSK_DEV sv6 root_unit(int a, v3 c) {
  sv6 u = sv6_zero();
  const v3 ex = v3_make(a % 3 == 0 ? 1.0f : 0.0f, a % 3 == 1 ? 1.0f : 0.0f, a % 3 == 2 ? 1.0f : 0.0f);
  if (a < 3) u.l = ex;
  else { u.a = ex; u.l = v3_cross(c, ex); }
  return u;
}
MSK_DEV void root_project(sv6 F, v3 c, float out[6]) {
  const v3 mc = v3_add(F.a, v3_cross(F.l, c));
  out[0] = F.l.x; out[1] = F.l.y; out[2] = F.l.z; out[3] = mc.x; out[4] = mc.y; out[5] = mc.z;
}

/* Forward pass shared by k_dynamics and k_kinematics: link frames, joint subspaces, spatial velocities (and accelerations with zero joint
 * acceleration).  Returns this lane's body frame; S / V of every body are in LDS afterwards.
 *
 * Only ONE pose product per tree level depends on the parent (oracle: kinematics()): U = U_parent * L, the rotation as composed.  The level loop -- the
 * serial part: one lane per env works, the wavefront issues every instruction of a level anyway -- is that product and its LDS hand-off (~60
 * instructions a level; the loop that also normalised, built S, V and acc inside was ~300).  Everything else is lane-parallel work around it: L
 * before, the published frame (normalised), the joint axis and S after; V and acc are sums along each body's own path from its root (DModel::path),
 * which every lane walks by itself in the oracle's order of additions, no hand-offs.
 * EARLY (k_dynamics): the frames go to the env record as soon as they exist and the workgroup's barrier lets the broadphase wave start on them. */
template <bool WITH_ACC, bool EARLY>
MSK_DEV pose forward_pass(const DModel* m, float* E, float* lds, const DynLds& ly, int i, bool has, sv6* Sout, sv6* Vout, sv6* Aout) {
  const DBody* b = &m->bodies[has ? i : 0];
  pose T;
  T.p = v3_make(0, 0, 0);
  T.q = quat_make(1, 0, 0, 0);
  sv6 S = sv6_zero(), V = sv6_zero(), A = sv6_zero();
  const bool child_link = has && b->kind == MSK_BODY_LINK && b->parent >= 0;
  const int mydepth = has ? m->depth[i] : -1;
  const int maxdepth = m->maxdepth;
  /* everything a lane needs from the template and the env record is fetched here, side by side: inside the level loop only the parent's LDS image is read */
  int parent = 0, dof = -1, jtype = MSK_JOINT_FIXED;
  pose L, Xp;
  L.p = Xp.p = v3_make(0, 0, 0);
  L.q = Xp.q = quat_make(1, 0, 0, 0);
  float qdi = 0.0f;
  uint4 pk = make_uint4(0, 0, 0, 0);
  if (has) {
    T = load_pose(E, m->lay.bpose, i);
    if (b->kind == MSK_BODY_LINK && b->parent < 0 && b->root_dof >= 0) { /* floating root: (v of its centre of mass, omega) is state, as for a free body */
      const float* qr = E + m->lay.qd + b->root_dof;
      const v3 vc = v3_make(qr[0], qr[1], qr[2]), w = v3_make(qr[3], qr[4], qr[5]);
      const m33 Rr = quat_to_m33(T.q);
      const v3 cr = v3_add(T.p, m33_mulv(&Rr, b->com));
      V.a = w;
      V.l = v3_add(vc, v3_cross(cr, w));   /* Pluecker: velocity of the point at the env origin */
      if (WITH_ACC) A.l = v3_cross(vc, w);  /* the angular unit motions turn about the moving centre of mass: d/dt (c x e) = v_c x e */
    }
    if (child_link) {
      parent = b->parent; dof = b->dof; jtype = b->jtype;
      Xp = b->Xp;
      float qi = 0.0f;
      if (dof >= 0) { qi = E[m->lay.q + dof]; qdi = E[m->lay.qd + dof]; }
      pose Jq;
      Jq.p = v3_make(0, 0, 0);
      Jq.q = quat_make(1, 0, 0, 0);
      if (jtype == MSK_JOINT_REVOLUTE) {
        float sn, cs;
        msk_sincos(0.5f * qi, &sn, &cs);
        Jq.q = quat_make(cs, sn, 0, 0);
      } else if (jtype == MSK_JOINT_PRISMATIC) {
        Jq.p = v3_make(qi, 0, 0);
      }
      L = pose_mul(pose_mul(Xp, Jq), b->XcInv);
      L.q = quat_normalize(L.q);
      pk = *(const uint4*)(m->path[i]);   /* root, ancestors at depth 1 .. 15 */
    }
    float* pp = lds + ly.pose + i * 8;
    pp[0] = T.p.x; pp[1] = T.p.y; pp[2] = T.p.z; pp[3] = T.q.w; pp[4] = T.q.x; pp[5] = T.q.y; pp[6] = T.q.z; pp[7] = qdi;
    lds_put_sv6(lds + ly.S + i * 6, S);
    lds_put_sv6(lds + ly.V + i * 6, V);
    if (WITH_ACC) lds_put_sv6(lds + ly.acc + i * 6, A);
  }
  dyn_sync();
  /* ---- the chain: one pose product per level ---- */
  pose U = T;
  for (int d = 1; d <= maxdepth; ++d) {
    if (child_link && mydepth == d) {
      const float* pp = lds + ly.pose + parent * 8;
      pose Up;
      Up.p = v3_make(pp[0], pp[1], pp[2]);
      Up.q = quat_make(pp[3], pp[4], pp[5], pp[6]);
      U = pose_mul(Up, L);
      float* po = lds + ly.pose + i * 8;
      po[0] = U.p.x; po[1] = U.p.y; po[2] = U.p.z; po[3] = U.q.w; po[4] = U.q.x; po[5] = U.q.y; po[6] = U.q.z;
    }
    dyn_sync();
  }
  /* ---- published frames: the composed rotation normalised (every lane for itself; the parents' images in LDS are replaced) ---- */
  if (child_link) {
    T.p = U.p;
    T.q = quat_normalize(U.q);
    float* po = lds + ly.pose + i * 8;
    po[3] = T.q.w; po[4] = T.q.x; po[5] = T.q.y; po[6] = T.q.z;
    if (EARLY) store_pose(E, m->lay.bpose, i, T);
  }
  if (EARLY) { if (blockDim.x > 64) __syncthreads(); }   /* the link frames are in the env records: the workgroup's broadphase wave may go */
  dyn_sync();
  /* ---- joint axes from the parents' published frames ---- */
  if (child_link) {
    const float* pp = lds + ly.pose + parent * 8;
    pose Tp;
    Tp.p = v3_make(pp[0], pp[1], pp[2]);
    Tp.q = quat_make(pp[3], pp[4], pp[5], pp[6]);
    const pose Tj = pose_mul(Tp, Xp);
    const v3 axis = quat_rotate(Tj.q, v3_make(1, 0, 0));
    if (jtype == MSK_JOINT_REVOLUTE) {
      S.a = axis;
      S.l = v3_cross(Tj.p, axis);
    } else if (jtype == MSK_JOINT_PRISMATIC) {
      S.l = axis;
    }
    lds_put_sv6(lds + ly.S + i * 6, S);
  }
  dyn_sync();
  /* ---- V = V_root + sum over the path of S qd, in path order ----
   * The first 15 steps in chunks of four without branches inside a chunk (a step past the body's own depth reads the body's own entry -- the table's filler --
   * and is dropped by a select), so that a chunk's 28 LDS reads are in flight together; deeper trees go on one step at a time. */
  const uint4 pk0 = pk;
  auto path_walk = [&](auto step) {
    const unsigned w4[4] = {pk0.x, pk0.y, pk0.z, pk0.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (4 * c + 1 <= maxdepth) {
#pragma unroll
        for (int u = 1; u <= 4; ++u) {
          const int d = 4 * c + u;
          if (d <= 15) step((int)((w4[d >> 2] >> ((d & 3) * 8)) & 0xffu), d <= mydepth);
        }
      }
    }
    for (int d = 16; d <= mydepth; ++d) step((int)m->path[i][d], true);
  };
  if (child_link) {
    V = lds_sv6(lds + ly.V + (pk0.x & 0xffu) * 6);
    path_walk([&](const int a, const bool on) {
      const sv6 Vn = sv6_madd(V, lds_sv6(lds + ly.S + a * 6), lds[ly.pose + a * 8 + 7]);
      V.a.x = on ? Vn.a.x : V.a.x; V.a.y = on ? Vn.a.y : V.a.y; V.a.z = on ? Vn.a.z : V.a.z;
      V.l.x = on ? Vn.l.x : V.l.x; V.l.y = on ? Vn.l.y : V.l.y; V.l.z = on ? Vn.l.z : V.l.z;
    });
    lds_put_sv6(lds + ly.V + i * 6, V);
  }
  if (WITH_ACC) {
    dyn_sync();
    /* ---- acc = acc_root + sum over the path of V_parent x (S qd) ---- */
    float* cx = lds + ly.Ic;   /* (free until the RNEA phase) */
    if (child_link) {
      const sv6 Vp = lds_sv6(lds + ly.V + parent * 6);
      const sv6 sq = {v3_scale(S.a, qdi), v3_scale(S.l, qdi)};
      lds_put_sv6(cx + i * 6, sv6_crossm(Vp, sq));
    }
    dyn_sync();
    if (child_link) {
      A = lds_sv6(lds + ly.acc + (pk0.x & 0xffu) * 6);
      path_walk([&](const int a, const bool on) {
        const sv6 An = sv6_add(A, lds_sv6(cx + a * 6));
        A.a.x = on ? An.a.x : A.a.x; A.a.y = on ? An.a.y : A.a.y; A.a.z = on ? An.a.z : A.a.z;
        A.l.x = on ? An.l.x : A.l.x; A.l.y = on ? An.l.y : A.l.y; A.l.z = on ? An.l.z : A.l.z;
      });
    }
    dyn_sync();   /* (everybody has read the cross terms: the RNEA phase may write its inertias there) */
  }
  *Sout = S; *Vout = V; *Aout = A;
  return T;
}

/* publishes what the reference exposes per body: link pose, COM linear velocity, angular velocity */
MSK_DEV void publish_body(const DModel* m, float* E, int i, const DBody* b, pose T, sv6 V, v3 comw) {
  if (b->kind == MSK_BODY_LINK) {
    if (b->parent >= 0) store_pose(E, m->lay.bpose, i, T);
    store_v3(E, m->lay.bang, i, V.a);
    store_v3(E, m->lay.blin, i, v3_add(V.l, v3_cross(V.a, comw)));
  } else if (b->kind == MSK_BODY_KINEMATIC) {
    store_v3(E, m->lay.blin, i, v3_make(0, 0, 0));
    store_v3(E, m->lay.bang, i, v3_make(0, 0, 0));
  }
}

/* PhysxGpuSystem.gpu_update_articulation_kinematics: frames and velocities only */
template <int LPE>
MSK_DEV void kinematics_block(const DModel* __restrict__ m, const DState& st, float* lds_all, const int blk) {
  const DynLds ly(m->nb, 0);
  const int sub = threadIdx.x / LPE, i = threadIdx.x % LPE;
  const int e_raw = blk * (64 / LPE) + sub;
  const bool live = e_raw < m->N;
  const int e = live ? e_raw : m->N - 1;
  float* lds = lds_all + sub * ly.total;
  const bool has = live && i < m->nb;
  float* E = EREC(st, m, e);
  sv6 S, V, A;
  pose T = forward_pass<false, false>(m, E, lds, ly, i, has, &S, &V, &A);
  if (!has) return;
  const DBody* b = &m->bodies[i];
  m33 R = quat_to_m33(T.q);
  v3 comw = v3_add(T.p, m33_mulv(&R, b->com));
  publish_body(m, E, i, b, T, V, comw);
}
template <int LPE>
__global__ void __launch_bounds__(64) k_kinematics(const DModel* __restrict__ m, DState st) {
  extern __shared__ __attribute__((aligned(16))) float lds_kin[];
  kinematics_block<LPE>(m, st, lds_kin, blockIdx.x);
}

/* blk: which block of the env range this workgroup is (its blockIdx.x when the launch serves one context: k_dynamics; its index inside
 * the context's share of a launch over several contexts: k_multi_dynamics, msk_kernels.h) */
/* broadphase of the block's envs (msk_broadphase.h), the whole wavefront on one env at a time.  It reads the body poses in the env
 * record and nothing else: in k_dynamics it is the workgroup's SECOND wavefront, released as soon as the first one has published the
 * link frames (the forward pass), and runs beside the rest of the dynamics on another SIMD of the CU (it used to be the kernel's tail:
 * ~25 k cycles of the ~105 k of a block). */
template <int LPE>
MSK_DEV void broadphase_block(const DModel* __restrict__ m, const DState& st, const int blk, const BpConst& bk) {
  __shared__ float bp_aabb[MSK_MAX_SHAPES][6];
  __shared__ float bp_obb[MSK_MAX_SHAPES][13];   /* rotation columns (9), local half extents (3); odd stride: no bank conflicts */
  if (blk == 0 && (threadIdx.x & 63) < MSK_SOLVE_CLASSES) st.cls_count[threadIdx.x & 63] = 0;   /* this substep's solver lists (filled by the narrowphase) */
#pragma unroll 1
  for (int k = 0; k < 64 / LPE; ++k) {
    const int eb = blk * (64 / LPE) + k;
    if (eb < m->N) broadphase_env(m, st, eb, bp_aabb, bp_obb, bk);
  }
}

template <int LPE, int MD>   /* MD: capacity of the per-lane joint-space rows (16 or 32 dofs) */
MSK_DEV void dynamics_block(const DModel* __restrict__ m, const DState& st, float* lds_all, const int blk) {
  const DynLds ly(m->nb, MD);
  const int sub = (threadIdx.x & 63) / LPE, i = threadIdx.x % LPE;   /* (lds_all: this wavefront's own carve) */
  const int e_raw = blk * (64 / LPE) + sub;
  const bool live = e_raw < m->N;      /* a surplus half-wave shadows the last env and stores nothing */
  const int e = live ? e_raw : m->N - 1;
  float* lds = lds_all + sub * ly.total;
  const int nb = m->nb, nd = m->nd, G = m->G;
  const bool has = live && i < nb;
  const float dt = m->cfg.timestep;
  const v3 g = v3_make(m->cfg.gravity[0], m->cfg.gravity[1], m->cfg.gravity[2]);
  float* E = EREC(st, m, e);
  float* Wenv = st.W + (size_t)e * G * G;     /* entries outside the blocks stay zero from allocation */
  float* Senv = st.Scol + (size_t)e * G * 8;
  float* vfenv = st.vfree + (size_t)e * G;
  float* Lm = lds + ly.M;
  float* Ll = lds + ly.L;
  float* vec = lds + ly.vec;
  const int LD = MD + 1;

#ifdef MSK_PROFILE_PHASES
  long long* dstamp = st.dbg + (size_t)m->N * 8 + 64 + (size_t)e * 8;
  int dsi = 0;
#define DPHASE() do { if (live && i == 0) dstamp[dsi] = (long long)__builtin_readcyclecounter(); dsi++; } while (0)
  const unsigned long long drt0 = __builtin_amdgcn_s_memrealtime();   /* the 100 MHz clock: where in the launch this wave ran (tools/gpu_phase_probe.py) */
#else
#define DPHASE()
#endif
  /* what phase 5 wants from the template and the env record, asked for now: two dependent loads (the joint's link, its force limit) that would otherwise be
   * waited for behind the barriers of phases 1-4 (the compiler does not move loads across them) */
  const bool rowlane = i < nd;   /* surplus half-waves compute along (LDS only) and store nothing */
  const float fmax_i = rowlane ? m->bodies[m->dof_body[i]].fmax : 0.0f;
  const float qdt_i = rowlane ? E[m->lay.qdt + i] : 0.0f;
  const float qf_i = rowlane ? E[m->lay.qf + i] : 0.0f;
  DPHASE();
  /* ---- 1. frames, velocities, bias accelerations (down the tree) ------------------------------------ */
  sv6 S, V, acc;
  const pose T = forward_pass<true, true>(m, E, lds, ly, i, has, &S, &V, &acc);
  const DBody* b = &m->bodies[has ? i : 0];
  const bool link = has && b->kind == MSK_BODY_LINK;
  const uint4 pk = (link && b->parent >= 0) ? *(const uint4*)(m->path[i]) : make_uint4(0, 0, 0, 0);   /* root, ancestors at depth 1 .. 15: the CRBA rows (phase 4) */
  if (has) {
    int* meta = (int*)(lds + ly.meta);
    meta[i * 2] = b->kind == MSK_BODY_LINK ? b->dof : -1;
    meta[i * 2 + 1] = b->kind == MSK_BODY_LINK ? b->root_dof : -1;
  }
  v3 comw = v3_make(0, 0, 0);
  m33 R;
  sinertia Ic;
  sv6 f = sv6_zero();
#ifdef MSK_VP_GUARD
  sv6 fvp = sv6_zero();
#endif
  if (has) {
    R = quat_to_m33(T.q);
    comw = v3_add(T.p, m33_mulv(&R, b->com));
    publish_body(m, E, i, b, T, V, comw);   /* (the frames went out inside the forward pass, and the workgroup's broadphase wave with them) */
  }
  /* zero M while the forward results settle */
  for (int k = i; k < MD * LD; k += LPE) Lm[k] = 0.0f;
  if (i < nd) vec[DV_QD * MD + i] = E[m->lay.qd + i];
  DPHASE();
  /* ---- 2. RNEA body forces, spatial inertias about the env origin ---------------------------------------- */
  if (link) {
    float Iw[6];
    sym6_rotate(&R, b->I6, Iw);
    const v3 cw = comw;
    const float ms = b->mass;
    Ic.m = ms;
    Ic.h = v3_scale(cw, ms);
    const float cc = v3_dot(cw, cw);
    Ic.I[0] = Iw[0] + ms * (cc - cw.x * cw.x);
    Ic.I[1] = Iw[1] + ms * (cc - cw.y * cw.y);
    Ic.I[2] = Iw[2] + ms * (cc - cw.z * cw.z);
    Ic.I[3] = Iw[3] - ms * (cw.x * cw.y);
    Ic.I[4] = Iw[4] - ms * (cw.x * cw.z);
    Ic.I[5] = Iw[5] - ms * (cw.y * cw.z);
    sv6 Iv = sinertia_mul(&Ic, V);
    f = sv6_add(sinertia_mul(&Ic, acc), sv6_crossf(V, Iv));
#ifdef MSK_VP_GUARD
    fvp = f;   /* Coriolis, centrifugal, gyroscopic alone: what the energy guard behind the solves looks at (oracle: dynamics(), MSK_VP_GUARD) */
#endif
    if (!b->nograv) {
      v3 mg = v3_scale(g, ms);
      f.a = v3_sub(f.a, v3_cross(cw, mg));
      f.l = v3_sub(f.l, mg);
    }
    { /* external force / torque at the link's centre of mass, this step only (the link's rows of cuda_rigid_body_force / _torque): consumed and
       * cleared like a free body's below */
      float* wr = st.ext_wrench + ((size_t)e * m->nb + i) * 8;
      const float4 wf = *(const float4*)wr, wt = *(const float4*)(wr + 4);
      if (wf.x != 0.0f || wf.y != 0.0f || wf.z != 0.0f || wt.x != 0.0f || wt.y != 0.0f || wt.z != 0.0f) {
        const v3 F = v3_make(wf.x, wf.y, wf.z), Tq = v3_make(wt.x, wt.y, wt.z);
        f.a = v3_sub(f.a, v3_add(v3_cross(cw, F), Tq));
        f.l = v3_sub(f.l, F);
        *(float4*)wr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *(float4*)(wr + 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  }
  dyn_sync(); /* everybody has read its parent's acc: the two regions now carry [f | inertia] records of 16 floats */
  float* rec = lds + ly.acc + (has ? i : 0) * 16;
  if (link) {
    *(float4*)(rec) = make_float4(f.a.x, f.a.y, f.a.z, f.l.x);
    *(float4*)(rec + 4) = make_float4(f.l.y, f.l.z, Ic.m, Ic.h.x);
    *(float4*)(rec + 8) = make_float4(Ic.h.y, Ic.h.z, Ic.I[0], Ic.I[1]);
    *(float4*)(rec + 12) = make_float4(Ic.I[2], Ic.I[3], Ic.I[4], Ic.I[5]);
#ifdef MSK_VP_GUARD
    lds_put_sv6(lds + ly.fvp + i * 6, fvp);
#endif
  }
  const int mydepth = has ? m->depth[i] : -1;
  const int maxdepth = m->maxdepth;
  /* the link's descendants (descending body index, DModel::desc): the first sixteen in registers, fetched before the barrier */
  const int ndesc = link ? (int)m->ndesc[i] : 0;
  uint4 dk = ndesc > 0 ? *(const uint4*)(m->desc[i]) : make_uint4(0, 0, 0, 0);
  dyn_sync();
  DPHASE();
  /* ---- 3. back up the tree: every link adds its descendants' OWN records to its own, one by one in descending body index (oracle: dynamics(), the
   * backward pass) -- lane-parallel, no hand-off between levels: the loop is as long as the root's list, a step is four wide LDS reads and sixteen adds ---- */
  for (int k = 0; k < ndesc; ++k) {
    if (k >= 16 && (k & 15) == 0) dk = *(const uint4*)(m->desc[i] + k);
    const unsigned w = (k & 8) ? ((k & 4) ? dk.w : dk.z) : ((k & 4) ? dk.y : dk.x);
    const int d = (int)((w >> ((k & 3) * 8)) & 0xffu);
    const float* rd = lds + ly.acc + d * 16;
    const float4 r0 = *(const float4*)(rd), r1 = *(const float4*)(rd + 4), r2 = *(const float4*)(rd + 8), r3 = *(const float4*)(rd + 12);
    f.a = v3_add(f.a, v3_make(r0.x, r0.y, r0.z));
    f.l = v3_add(f.l, v3_make(r0.w, r1.x, r1.y));
    Ic.m += r1.z;
    Ic.h = v3_add(Ic.h, v3_make(r1.w, r2.x, r2.y));
    Ic.I[0] += r2.z; Ic.I[1] += r2.w; Ic.I[2] += r3.x; Ic.I[3] += r3.y; Ic.I[4] += r3.z; Ic.I[5] += r3.w;
#ifdef MSK_VP_GUARD
    fvp = sv6_add(fvp, lds_sv6(lds + ly.fvp + d * 6));
#endif
  }
  dyn_sync();
  DPHASE();
  /* ---- 4. bias torques and CRBA rows (lane = body with a dof) ------------------------------------------------ */
  if (link && b->dof >= 0) {
    const int di = b->dof;
    vec[DV_BIAS * MD + di] = sv6_dot(S, f);
#ifdef MSK_VP_GUARD
    vec[DV_BVP * MD + di] = sv6_dot(S, fvp);
#endif
    const sv6 F = sinertia_mul(&Ic, S);
    Lm[di * LD + di] = sv6_dot(S, F) + b->armature;
    /* the row's entries with its ancestors' joints (the entries are disjoint: any order).  Trees of depth <= 15: the ancestors are the bytes of the body's path
     * (DModel::path, already in registers) and what is asked of an ancestor -- its dof, its root_dof -- is in LDS: no walk up the parent pointers, whose every
     * hop was a dependent load from the template (~10 k cycles of this phase on a 12-deep arm) */
    auto with_ancestor = [&](const int j, const int dofj, const int rootj) {
      if (dofj >= 0) {
        const float v = sv6_dot(lds_sv6(lds + ly.S + j * 6), F);
        Lm[di * LD + dofj] = v;
        Lm[dofj * LD + di] = v;
      }
      if (rootj >= 0) { /* coupling with the floating root's six unit motions */
        float Fc[6];
        const float* pj = lds + ly.pose + j * 8;
        const m33 Rj = quat_to_m33(quat_make(pj[3], pj[4], pj[5], pj[6]));
        root_project(F, v3_add(v3_make(pj[0], pj[1], pj[2]), m33_mulv(&Rj, m->bodies[j].com)), Fc);
#pragma unroll
        for (int a = 0; a < 6; ++a) { Lm[di * LD + rootj + a] = Fc[a]; Lm[(rootj + a) * LD + di] = Fc[a]; }
      }
    };
    if (maxdepth <= 15) {
      const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
      const int* meta = (const int*)(lds + ly.meta);
#pragma unroll
      for (int d = 0; d < 15; ++d) {
        if (d < mydepth) {
          const int j = (int)((w4[d >> 2] >> ((d & 3) * 8)) & 0xffu);
          with_ancestor(j, meta[j * 2], meta[j * 2 + 1]);
        }
      }
    } else {
      int j = b->parent;
      while (j >= 0) {
        const DBody* bj = &m->bodies[j];
        with_ancestor(j, bj->dof, bj->root_dof);
        j = bj->parent;
      }
    }
    float* sc = Senv + di * 8;   /* motion subspace column of coordinate di, for the row assembly */
    sc[0] = S.a.x; sc[1] = S.a.y; sc[2] = S.a.z; sc[3] = S.l.x; sc[4] = S.l.y; sc[5] = S.l.z;
    /* drive of this joint; an acceleration drive's gains are per unit of the joint's own inertia (the diagonal just computed) */
    const float Mdd = Lm[di * LD + di];
    vec[DV_KD * MD + di] = b->drive_accel ? b->K * Mdd : b->K;
    vec[DV_DD * MD + di] = b->drive_accel ? b->D * Mdd : b->D;
    vec[DV_FC * MD + di] = 0.0f;
    vec[DV_ERR * MD + di] = E[m->lay.q + di] - E[m->lay.qt + di];
  }
  if (link && b->root_dof >= 0) { /* floating root: bias = the accumulated wrench, root block = the tree's composite inertia column by column */
    const int rd = b->root_dof;
    if (live) store_v3(E, m->lay.comw, i, comw);   /* the solver integrates the root like a free body: about its centre of mass */
    float fc[6];
    root_project(f, comw, fc);
#ifdef MSK_VP_GUARD
    float fcv[6];
    root_project(fvp, comw, fcv);
#pragma unroll
    for (int a = 0; a < 6; ++a) vec[DV_BVP * MD + rd + a] = fcv[a];
#endif
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const sv6 u = root_unit(a, comw);
      const sv6 F = sinertia_mul(&Ic, u);
      float Fc[6];
      root_project(F, comw, Fc);
#pragma unroll
      for (int r = 0; r < 6; ++r) Lm[(rd + r) * LD + rd + a] = Fc[r];
      vec[DV_BIAS * MD + rd + a] = fc[a];
      vec[DV_KD * MD + rd + a] = 0.0f;
      vec[DV_DD * MD + rd + a] = 0.0f;
      vec[DV_FC * MD + rd + a] = 0.0f;
      vec[DV_ERR * MD + rd + a] = 0.0f;
      float* sc = Senv + (rd + a) * 8;
      sc[0] = u.a.x; sc[1] = u.a.y; sc[2] = u.a.z; sc[3] = u.l.x; sc[4] = u.l.y; sc[5] = u.l.z;
    }
  }
  dyn_sync();

  DPHASE();
  /* ---- 5. implicit PD: A = M + dt D + dt^2 K (+ tendons), Cholesky, solves; second pass without the drives whose predicted force
   * exceeds their limit: those become soft rows of the solver, clamped to +-fmax dt (st.drv_mask, st.drv; oracle: dynamics()) ---- */
  const float bias_i = rowlane ? vec[DV_BIAS * MD + i] : 0.0f;
  float Kd = rowlane ? vec[DV_KD * MD + i] : 0.0f, Dd = rowlane ? vec[DV_DD * MD + i] : 0.0f;
  float err = rowlane ? vec[DV_ERR * MD + i] : 0.0f;
  bool drive_row = false;   /* my joint's drive goes to the solver */
  for (int pass = 0; pass < 2; ++pass) {
    float Arow[MD];
    float rhs = 0.0f;
    if (rowlane) {
      float mv = 0.0f;
#pragma unroll
      for (int k = 0; k < MD; ++k) {
        const float mk = (k < nd) ? Lm[i * LD + k] : 0.0f;
        Arow[k] = mk;
        if (k < nd) mv = fmaf(mk, vec[DV_QD * MD + k], mv);
      }
#ifdef MSK_VP_GUARD
      vec[DV_R0 * MD + i] = mv;
#endif
      const float dadd = dt * fmaf(dt, Kd, Dd);
#pragma unroll
      for (int k = 0; k < MD; ++k)
        if (k == i) Arow[k] += dadd;
      const float tau = qf_i - bias_i - Kd * err + Dd * qdt_i;
      rhs = fmaf(dt, tau, mv);
      for (int t = 0; t < m->nt; ++t) {
        const DTendon* tn = &m->tendons[t];
        const float g2 = dt * fmaf(dt, tn->K, tn->D);
        const float te = fmaf(tn->ca, E[m->lay.q + tn->dof_a], tn->cb * E[m->lay.q + tn->dof_b]) - tn->rest;
        if (i == tn->dof_a) {
#pragma unroll
          for (int k = 0; k < MD; ++k) {
            if (k == tn->dof_a) Arow[k] += g2 * tn->ca * tn->ca;
            if (k == tn->dof_b) Arow[k] += g2 * tn->ca * tn->cb;
          }
          rhs -= dt * tn->K * te * tn->ca;
        }
        if (i == tn->dof_b) {
#pragma unroll
          for (int k = 0; k < MD; ++k) {
            if (k == tn->dof_b) Arow[k] += g2 * tn->cb * tn->cb;
            if (k == tn->dof_a) Arow[k] += g2 * tn->ca * tn->cb;
          }
          rhs -= dt * tn->K * te * tn->cb;
        }
      }
      vec[DV_RHS * MD + i] = rhs;
    }
    /* The 16-row form (two envs per wavefront, an env's rows in ONE DPP row of 16 lanes): L stays in the row lanes' registers and an entry of another row is
     * a row_newbcast of that lane's register -- no LDS image of L, no hand-off barriers, no loads inside the factorisation and the substitutions (they were
     * one ds_read per multiply-add and two barriers per column).  Same operations in the same order as the LDS form below: the same bits.  The unconstrained
     * velocity is solved by lane 15 of the row (nd <= 15 in this form), whose right-hand side is the row lanes' own `rhs`.  Every lane runs along (a DPP
     * source lane must be active); only the lanes with a result store. */
#ifdef MSK_VP_GUARD
    constexpr bool DPP_ROWS = false;   /* (the guard's two extra right-hand sides ride on lanes of the LDS form) */
#else
    constexpr bool DPP_ROWS = (LPE == 32 && MD == 16);
#endif
    if constexpr (DPP_ROWS) {
      float Lrow[MD];
#pragma unroll
      for (int k = 0; k < MD; ++k) { Lrow[k] = 0.0f; if (!rowlane) Arow[k] = 0.0f; }
      if (!rowlane) rhs = 0.0f;
#pragma unroll
      for (int j = 0; j < MD; ++j) {
        if (j < nd) {
          float sum = Arow[j];
#pragma unroll
          for (int k = 0; k < j; ++k) sum = fmaf(-Lrow[k], group_bcast<16>(Lrow[k], j), sum);
          const float dj = sqrtf(sum);                      /* (meant for lane j) */
          const float djj = group_bcast<16>(dj, j);
          const float below = sum / djj;
          Lrow[j] = (i == j) ? dj : ((i > j) ? below : 0.0f);
        }
      }
      constexpr int VFL = 15;
      const bool col = i < nd, vf = i == VFL;
      float y[MD], x[MD];
#pragma unroll
      for (int r = 0; r < MD; ++r) {
        y[r] = 0.0f;
        if (r < nd) {
          const float rr = group_bcast<16>(rhs, r);
          float sum = vf ? rr : ((r == i) ? 1.0f : 0.0f);
#pragma unroll
          for (int k = 0; k < r; ++k) sum = fmaf(-group_bcast<16>(Lrow[k], r), y[k], sum);
          y[r] = sum / group_bcast<16>(Lrow[r], r);
        }
      }
#pragma unroll
      for (int r = MD - 1; r >= 0; --r) {
        x[r] = 0.0f;
        if (r < nd) {
          float sum = y[r];
#pragma unroll
          for (int k = r + 1; k < MD; ++k)
            if (k < nd) sum = fmaf(-group_bcast<16>(Lrow[r], k), x[k], sum);
          x[r] = sum / group_bcast<16>(Lrow[r], r);
        }
      }
      if (vf) {
#pragma unroll
        for (int r = 0; r < MD; ++r)
          if (r < nd) { vec[DV_VF * MD + r] = x[r]; if (live) vfenv[r] = x[r]; }
      } else if (col && live) {
#pragma unroll
        for (int r = 0; r < MD; ++r)
          if (r < nd) Wenv[r * G + i] = x[r];
      }
    } else {
      /* Cholesky A = L L^T, lane i owns row i; column j is finished at step j */
      float Lrow[MD];
  #pragma unroll
      for (int k = 0; k < MD; ++k) Lrow[k] = 0.0f;
  #pragma unroll
      for (int j = 0; j < MD; ++j) {
        if (j < nd) {
          float sum = 0.0f;
          if (rowlane && i >= j) {
            sum = Arow[j];
  #pragma unroll
            for (int k = 0; k < j; ++k) sum = fmaf(-Lrow[k], Ll[j * LD + k], sum);
            if (i == j) {
              Lrow[j] = sqrtf(sum);
              Ll[j * LD + j] = Lrow[j];
            }
          }
          dyn_sync();
          if (rowlane && i > j) {
            Lrow[j] = sum / Ll[j * LD + j];
            Ll[i * LD + j] = Lrow[j];
          }
          dyn_sync();
        }
      }
      /* triangular solves: lane c < nd -> column c of A^-1, lane VFL -> vfree = A^-1 rhs (VFL = MD, or the wavefront's last lane in the
       * 64-row form: nd <= 63 leaves it free) */
      constexpr int VFL = (MD < LPE) ? MD : LPE - 1;
      const bool col = i < nd, vf = i == VFL;
  #ifdef MSK_VP_GUARD
      /* the guard's two right-hand sides (M qd and -dt bias_vp) ride along as two more columns where the half-wave has lanes to spare (every form but the
       * 64-row one, which solves them in a phase of its own below) */
      constexpr bool GUARD_RIDES = (MD < LPE) && (MD + 2 < LPE);
      const bool g0 = GUARD_RIDES && i == VFL + 1, g1 = GUARD_RIDES && i == VFL + 2;
  #else
      constexpr bool g0 = false, g1 = false;
  #endif
      float y[MD], x[MD];
      if (col || vf || g0 || g1) {
  #pragma unroll
        for (int r = 0; r < MD; ++r) {
          y[r] = 0.0f;
          if (r < nd) {
            float sum = vf ? vec[DV_RHS * MD + r] : ((r == i) ? 1.0f : 0.0f);
  #ifdef MSK_VP_GUARD
            if (g0) sum = vec[DV_R0 * MD + r];
            if (g1) sum = -(dt * vec[DV_BVP * MD + r]);
  #endif
  #pragma unroll
            for (int k = 0; k < r; ++k) sum = fmaf(-Ll[r * LD + k], y[k], sum);
            y[r] = sum / Ll[r * LD + r];
          }
        }
  #pragma unroll
        for (int r = MD - 1; r >= 0; --r) {
          x[r] = 0.0f;
          if (r < nd) {
            float sum = y[r];
  #pragma unroll
            for (int k = r + 1; k < MD; ++k)
              if (k < nd) sum = fmaf(-Ll[k * LD + r], x[k], sum);
            x[r] = sum / Ll[r * LD + r];
          }
        }
        if (vf) {
  #pragma unroll
          for (int r = 0; r < MD; ++r)
            if (r < nd) { vec[DV_VF * MD + r] = x[r]; if (live) vfenv[r] = x[r]; }
  #ifdef MSK_VP_GUARD
        } else if (g0 || g1) {
  #pragma unroll
          for (int r = 0; r < MD; ++r)
            if (r < nd) vec[(g0 ? DV_V0 : DV_VB) * MD + r] = x[r];      /* (the vb row holds dv until the guard adds v0) */
  #endif
        } else if (live) {
  #pragma unroll
          for (int r = 0; r < MD; ++r)
            if (r < nd) Wenv[r * G + i] = x[r];
        }
      }
    }
    dyn_sync();
    if (pass == 1) break;
    /* drive force limits: predict the PD force at v*; where it exceeds the limit the drive becomes a solver row */
    bool sat = false;
    if (rowlane && !(Kd == 0.0f && Dd == 0.0f)) {
      const float vfi = vec[DV_VF * MD + i];
      const float F = -Kd * fmaf(dt, vfi, err) - Dd * (vfi - qdt_i);
      /* ... and the force of the stalled joint (v = 0: a link that a contact holds back loses the damping term the free prediction counts on) */
      const float Fstall = fmaf(Dd, qdt_i, -(Kd * err));
      if (fabsf(F) > fmax_i || fabsf(Fstall) > fmax_i) {
        const float cfm = 1.0f / (dt * fmaf(dt, Kd, Dd));
        if (live) {
          float4 rec;
          rec.x = cfm;
          rec.y = (dt * fmaf(Kd, err, -(Dd * qdt_i))) * cfm;
          rec.z = fmax_i * dt;
          rec.w = 0.0f;
          *(float4*)(st.drv + ((size_t)e * G + i) * 4) = rec;
        }
        Kd = 0.0f; Dd = 0.0f; err = 0.0f;
        sat = true;
        drive_row = true;
      }
    }
    /* wave-uniform decision: an env that did not saturate recomputes the identical pass */
    if (__ballot(sat) == 0ull) break;
  }
  { /* which drives are solver rows in this substep: one word per env (the classification and the solver read it) */
    const unsigned long long dm = __ballot(drive_row);
    if (live && i == 0) st.drv_mask[e] = (LPE == 64) ? dm : ((dm >> (sub * LPE)) & 0xFFFFFFFFull);
  }

#ifdef MSK_VP_GUARD
  { /* ---- 5b. energy guard on the velocity-product terms (oracle: dynamics(), MSK_VP_GUARD; a candidate, not compiled by default).  v0 = A^-1 M qd,
     * vb = v0 + A^-1 (-dt bias_vp) by two more triangular solves (two spare lanes of the solves above; the 64-row form: lanes 0 and 1 here), M v0 and M vb by
     * the row lanes, the two energies by lane 0
     * in the oracle's order; where vb carries more than MSK_VP_GUARD times v0's energy, v* moves by (sqrt(T0 / Tb) - 1) vb */
    dyn_sync();
    constexpr bool GUARD_RIDES = (MD < LPE) && (MD + 2 < LPE);
    if (!GUARD_RIDES && i < 2) {
      float y[MD], x[MD];
#pragma unroll
      for (int r = 0; r < MD; ++r) {
        y[r] = 0.0f;
        if (r < nd) {
          float sum = (i == 0) ? vec[DV_R0 * MD + r] : -(dt * vec[DV_BVP * MD + r]);
#pragma unroll
          for (int k = 0; k < r; ++k) sum = fmaf(-Ll[r * LD + k], y[k], sum);
          y[r] = sum / Ll[r * LD + r];
        }
      }
#pragma unroll
      for (int r = MD - 1; r >= 0; --r) {
        x[r] = 0.0f;
        if (r < nd) {
          float sum = y[r];
#pragma unroll
          for (int k = r + 1; k < MD; ++k)
            if (k < nd) sum = fmaf(-Ll[k * LD + r], x[k], sum);
          x[r] = sum / Ll[r * LD + r];
        }
      }
#pragma unroll
      for (int r = 0; r < MD; ++r)
        if (r < nd) vec[(i == 0 ? DV_V0 : DV_VB) * MD + r] = x[r];      /* (the vb row holds dv for a moment) */
    }
    dyn_sync();
    const float vb_i = rowlane ? vec[DV_V0 * MD + i] + vec[DV_VB * MD + i] : 0.0f;
    dyn_sync();
    if (rowlane) vec[DV_VB * MD + i] = vb_i;
    dyn_sync();
    if (rowlane) {
      float m0 = 0.0f, mb = 0.0f;
      for (int k = 0; k < nd; ++k) {
        const float mk = Lm[i * LD + k];
        m0 = fmaf(mk, vec[DV_V0 * MD + k], m0);
        mb = fmaf(mk, vec[DV_VB * MD + k], mb);
      }
      vec[DV_M0 * MD + i] = m0;
      vec[DV_MB * MD + i] = mb;
    }
    dyn_sync();
    if (i == 0) {
      float T0 = 0.0f, Tb = 0.0f;
      for (int r = 0; r < nd; ++r) {
        T0 = fmaf(vec[DV_V0 * MD + r], vec[DV_M0 * MD + r], T0);
        Tb = fmaf(vec[DV_VB * MD + r], vec[DV_MB * MD + r], Tb);
      }
      const bool trig = nd > 0 && Tb > MSK_VP_GUARD * T0 && T0 > 0.0f;
      vec[DV_R0 * MD + 0] = trig ? sqrtf(T0 / Tb) - 1.0f : 0.0f;      /* (M qd is not needed any more; a scale of exactly 0 = leave v* alone) */
    }
    dyn_sync();
    const float sc = vec[DV_R0 * MD + 0];
    if (sc != 0.0f && rowlane) {
      const float nv = fmaf(sc, vb_i, vec[DV_VF * MD + i]);
      vec[DV_VF * MD + i] = nv;
      if (live) { MSK_WAIT_VMCNT0(); vfenv[i] = nv; }      /* behind the solve lane's store of the same word */
    }
    dyn_sync();
  }
#endif
  DPHASE();
  /* ---- 6. free bodies: unconstrained velocity, world inverse inertia, subspace columns -------------------- */
  if (has && b->kind == MSK_BODY_DYNAMIC) {
    v3 v = load_v3(E, m->lay.blin, i), w = load_v3(E, m->lay.bang, i);
    if (!b->nograv) v = v3_madd(v, g, dt);
    float Iinv[6];
    float mass_i = b->mass;
    const int xb = m->xb_slot[i];
    if (xb >= 0) { /* per-env instance: mass and principal inverse inertia from the env record */
      const float* x = E + m->lay.xbody + xb * 8;
      const float Ii[6] = {x[1], x[2], x[3], 0.0f, 0.0f, 0.0f};
      mass_i = x[0];
      sym6_rotate(&R, Ii, Iinv);
    } else sym6_rotate(&R, b->Iinv6, Iinv);
    { /* external force / torque at the centre of mass, this step only (msk_apply FORCE / TORQUE).  Data-driven so that a captured
       * step graph sees forces applied between replays: the rows are always read (32 bytes per dynamic body), a non-zero row is
       * consumed and cleared here */
      float* wr = st.ext_wrench + ((size_t)e * m->nb + i) * 8;
      const float4 wf = *(const float4*)wr, wt = *(const float4*)(wr + 4);
      if (wf.x != 0.0f || wf.y != 0.0f || wf.z != 0.0f || wt.x != 0.0f || wt.y != 0.0f || wt.z != 0.0f) {
        v = v3_madd(v, v3_make(wf.x, wf.y, wf.z), dt / mass_i);
        w = v3_madd(w, sym6_mulv(Iinv, v3_make(wt.x, wt.y, wt.z)), dt);
        *(float4*)wr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *(float4*)(wr + 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
    const float kl = fmaxf(0.0f, 1.0f - dt * b->lin_damp);
    const float ka = fmaxf(0.0f, 1.0f - dt * b->ang_damp);
    v = v3_scale(v, kl);
    w = v3_scale(w, ka);
    const unsigned lk = b->lock;   /* locked world axes: no velocity, no response (oracle: dynamics(), assemble) */
    if (lk) {
      if (lk & 1u) v.x = 0.0f;
      if (lk & 2u) v.y = 0.0f;
      if (lk & 4u) v.z = 0.0f;
      if (lk & 8u) w.x = 0.0f;
      if (lk & 16u) w.y = 0.0f;
      if (lk & 32u) w.z = 0.0f;
    }
    const int o = b->vofs;
    vfenv[o + 0] = v.x; vfenv[o + 1] = v.y; vfenv[o + 2] = v.z;
    vfenv[o + 3] = w.x; vfenv[o + 4] = w.y; vfenv[o + 5] = w.z;
    const float im = 1.0f / mass_i;
    const float Im[3][3] = {{Iinv[0], Iinv[3], Iinv[4]}, {Iinv[3], Iinv[1], Iinv[5]}, {Iinv[4], Iinv[5], Iinv[2]}};
    const v3 ex[3] = {v3_make(1, 0, 0), v3_make(0, 1, 0), v3_make(0, 0, 1)};
    for (int a = 0; a < 3; ++a) {
      Wenv[(o + a) * G + o + a] = ((lk >> a) & 1u) ? 0.0f : im;
      for (int j = 0; j < 3; ++j) Wenv[(o + 3 + a) * G + o + 3 + j] = (((lk >> (3 + a)) | (lk >> (3 + j))) & 1u) ? 0.0f : Im[a][j];
      float* sl = Senv + (o + a) * 8;       /* v_com */
      sl[0] = 0.0f; sl[1] = 0.0f; sl[2] = 0.0f; sl[3] = ex[a].x; sl[4] = ex[a].y; sl[5] = ex[a].z;
      float* sa = Senv + (o + 3 + a) * 8;   /* omega: point velocity = w x (p - c) */
      const v3 cl = v3_cross(comw, ex[a]);
      sa[0] = ex[a].x; sa[1] = ex[a].y; sa[2] = ex[a].z; sa[3] = cl.x; sa[4] = cl.y; sa[5] = cl.z;
    }
    store_v3(E, m->lay.comw, i, comw);
  }
  DPHASE();
#ifdef MSK_PROFILE_PHASES
  if (live && i == 0) dstamp[7] = (long long)(((__builtin_amdgcn_s_memrealtime() & 0xffffffffull) << 32) | (drt0 & 0xffffffffull));
#endif
#undef DPHASE
}
/* 64 (DW + 1) threads, DW = 1 or 2: wavefronts 0 .. DW-1 = the dynamics of DW consecutive env blocks (a block = the 64 / LPE envs of one
 * wavefront), the last wavefront = the broadphase of the same blocks, released by the workgroup barrier that follows the publication of the
 * link frames; it runs beside the rest of the dynamics on another SIMD.  The host picks the widest form whose wavefronts are all resident at
 * once (3 per SIMD at this kernel's register count = 3072 on the chip): 4096 envs of two per wavefront are 1024 workgroups of 192 threads --
 * with 128-thread workgroups they were 4096 wavefronts that queued, so the broadphase stayed the ~25 k-cycle tail of the one wavefront.
 * 64 threads: that form (broadphase as the tail), for launches larger still.
 * Round 6 (profiles/r06_launch_position_probe.log): at 154 VGPRs three wavefronts fit a SIMD, so the 1024 x 3 wavefronts of 4096 envs fill the chip EXACTLY -- the 100 MHz
 * stamps of the profiling build showed workgroups starting 21-23 us after the first (one that finds no CU with three free slots on the right SIMDs waits for a dynamics
 * wavefront to end: a 25-30 us kernel became 38-47 us, and which CUs overflowed depended on what the launch before had left behind).  Holding the kernel to 128 VGPRs (four per
 * SIMD) starts every wavefront at once but spills and makes each 11 % longer (profiles/r06_ab_dynamics_four_waves_per_simd.log: + 1 % at 4096 envs, - 2 % at 8192 and more).
 * 256 threads = three dynamics wavefronts + the broadphase wavefront of their three blocks: one wavefront per SIMD per workgroup, 683 workgroups for 4096 envs = 89 % of the slots. */
template <int LPE, int MD>
__global__ void __launch_bounds__(256) k_dynamics(const DModel* __restrict__ m, DState st) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int DW = nwaves > 1 ? nwaves - 1 : 1;
  const int nblk = (m->N + (64 / LPE) - 1) / (64 / LPE);
  if (wave < DW) {
    const int blk = blockIdx.x * DW + wave;
    if (blk < nblk) dynamics_block<LPE, MD>(m, st, lds_dyn + wave * (64 / LPE) * DynLds(m->nb, MD).total, blk);
    else if (nwaves > 1) __syncthreads();      /* a surplus dynamics wavefront of the last workgroup only meets the barrier */
    if (nwaves == 1 && m->np > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the other half-wave's stores to its env record */
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const BpConst bk = bp_const(m);
      broadphase_block<LPE>(m, st, blk, bk);
    }
  } else {
    const BpConst bk = bp_const(m);      /* what the broadphase asks of the template, fetched while the dynamics wavefronts compute the link frames */
    __syncthreads();
    if (m->np > 0)
      for (int w = 0; w < DW; ++w)
        if (blockIdx.x * DW + w < nblk) broadphase_block<LPE>(m, st, blockIdx.x * DW + w, bk);
  }
}

#endif
