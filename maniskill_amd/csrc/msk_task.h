/*
 * msk_task.h — fused PickCube-v1 task kernels (include/msk_task.h), one lane per env.
 * They read the simulator's env records directly; arithmetic follows maniskill_amd/envs/pick_cube.py
 * (the torch mirror of the reference task code) statement by statement.
 */
#ifndef MSK_TASK_KERNELS_H
#define MSK_TASK_KERNELS_H

#include "../../include/msk_task.h"
#include "msk_model.h"

__global__ void __launch_bounds__(256) k_pickcube_set_action(const DModel* __restrict__ m, DState st, msk_pickcube_desc d,
                                                             const float* __restrict__ actions) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= m->N) return;
  float* E = EREC(st, m, e);
  const float* a = actions + (size_t)e * (d.arm_dofs + 1);
  for (int j = 0; j < d.arm_dofs; ++j) {
    const float aj = fminf(fmaxf(a[j], -1.0f), 1.0f);
    E[m->lay.qt + j] = E[m->lay.q + j] + d.arm_delta * aj;
  }
  const float ag = fminf(fmaxf(a[d.arm_dofs], -1.0f), 1.0f);
  /* _clip_and_scale_action: 0.5 (high + low) + 0.5 (high - low) a */
  const float g = d.gripper_mid + d.gripper_half * ag;
  E[m->lay.qt + d.arm_dofs] = g;
  E[m->lay.qt + d.arm_dofs + 1] = g;
}

/* pd_ee_delta_pos / pd_ee_delta_pose (agents/controllers/pd_ee_pose.py:224-262; utils/kinematics.py:229-245): the action is a
 * delta pose of the tcp in the root frame; one Levenberg-Marquardt step (J^T J + lambda I) dq = J^T delta on the geometric
 * Jacobian of the arm's revolute joints (column k = [z_k x (p_ee - o_k); z_k]); arm target = q + dq.  One thread per env. */
struct EeCtl { int root, adim; float pos_bound, rot_scale, damping; };
__global__ void __launch_bounds__(256) k_pickcube_set_action_ee(const DModel* __restrict__ m, DState st, msk_pickcube_desc d, EeCtl c,
                                                                const float* __restrict__ actions) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= m->N) return;
  float* E = EREC(st, m, e);
  const float* a = actions + (size_t)e * c.adim;
  /* delta pose: translation clipped and scaled; rotation clipped by norm, scaled by rot_lower (as the reference does) */
  float del[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 3; ++k) del[k] = c.pos_bound * fminf(fmaxf(a[k], -1.0f), 1.0f);
  if (c.adim == 7) {
    float rx = a[3], ry = a[4], rz = a[5];
    const float nrm = sqrtf(rx * rx + ry * ry + rz * rz);
    if (nrm > 1.0f) { const float inv = 1.0f / nrm; rx = rx * inv; ry = ry * inv; rz = rz * inv; }
    del[3] = rx * c.rot_scale; del[4] = ry * c.rot_scale; del[5] = rz * c.rot_scale;
  }
  const pose root = load_pose(E, m->lay.bpose, c.root);
  const v3 pee = load_pose(E, m->lay.bpose, d.tcp).p;
  float J[6][7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const DBody* b = &m->bodies[m->dof_body[k]];
    const pose Tj = pose_mul(load_pose(E, m->lay.bpose, b->parent), b->Xp);
    const v3 z = quat_rotate_inv(root.q, quat_rotate(Tj.q, v3_make(1, 0, 0)));
    const v3 r = quat_rotate_inv(root.q, v3_sub(pee, Tj.p));
    const v3 jv = v3_cross(z, r);
    J[0][k] = jv.x; J[1][k] = jv.y; J[2][k] = jv.z; J[3][k] = z.x; J[4][k] = z.y; J[5][k] = z.z;
  }
  /* One Levenberg-Marquardt step in its dual form: dq = J^T (J J^T + lambda I)^-1 delta.  The reference's primal 7 x 7 system
   * (J^T J + lambda I) dq = J^T delta (kinematics.py:233-242) has the same solution, but is rank 6 up to lambda = 1e-4: rounding in the
   * null-space direction comes back 1e4-fold (1e-2 rad between two fp32 solvers).  The 6 x 6 dual is well conditioned and its solution
   * lies in J's row space by construction.  Cholesky, two triangular solves. */
  float A[6][6], y[6], rhs[7];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    y[i] = del[i];
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float s = (i == j) ? c.damping : 0.0f;
#pragma unroll
      for (int k = 0; k < 7; ++k) s = fmaf(J[i][k], J[j][k], s);
      A[i][j] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fmaf(-A[i][k], A[j][k], s);
      A[i][j] = (i == j) ? sqrtf(s) : s / A[j][j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = y[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fmaf(-A[i][k], y[k], s);
    y[i] = s / A[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s = fmaf(-A[k][i], y[k], s);
    y[i] = s / A[i][i];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) s = fmaf(J[r][k], y[r], s);
    rhs[k] = s;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) E[m->lay.qt + j] = E[m->lay.q + j] + rhs[j];
  const float ag = fminf(fmaxf(a[c.adim - 1], -1.0f), 1.0f);
  const float g = d.gripper_mid + d.gripper_half * ag;
  E[m->lay.qt + d.arm_dofs] = g;
  E[m->lay.qt + d.arm_dofs + 1] = g;
}

/* Kinematics.compute_ik(delta, q0, is_delta_pose) for a chain given by coordinates (include/msk_task.h msk_compute_ik_delta): one thread per
 * env.  n >= 6: dq = J^T (J J^T + lambda I)^-1 delta, the arithmetic of k_pickcube_set_action_ee (a 7-joint chain gives its bits); n < 6:
 * (J^T J + lambda I) dq = J^T delta.  Cholesky of an s x s matrix, s = min(n, 6). */
struct IkCtl { int ee_body, root_body, njoints, dofs[MSK_IK_MAX_JOINTS]; float damping, alpha; };
__global__ void __launch_bounds__(256) k_ik_delta(const DModel* __restrict__ m, DState st, IkCtl d, const float* __restrict__ delta,
                                                  float* __restrict__ out, int commit) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= m->N) return;
  float* E = EREC(st, m, e);
  const int n = d.njoints;
  const float* del = delta + (size_t)e * 6;
  const pose root = load_pose(E, m->lay.bpose, d.root_body);
  const v3 pee = load_pose(E, m->lay.bpose, d.ee_body).p;
  float J[6][MSK_IK_MAX_JOINTS];
#pragma unroll
  for (int k = 0; k < MSK_IK_MAX_JOINTS; ++k) {
    v3 jv = v3_make(0, 0, 0), jw = v3_make(0, 0, 0);
    if (k < n) {
      const DBody* b = &m->bodies[m->dof_body[d.dofs[k]]];
      const pose Tj = pose_mul(load_pose(E, m->lay.bpose, b->parent), b->Xp);
      const v3 z = quat_rotate_inv(root.q, quat_rotate(Tj.q, v3_make(1, 0, 0)));
      if (b->jtype == MSK_JOINT_PRISMATIC) jv = z;
      else {
        const v3 r = quat_rotate_inv(root.q, v3_sub(pee, Tj.p));
        jv = v3_cross(z, r);
        jw = z;
      }
    }
    J[0][k] = jv.x; J[1][k] = jv.y; J[2][k] = jv.z; J[3][k] = jw.x; J[4][k] = jw.y; J[5][k] = jw.z;
  }
  const bool dual = n >= 6;
  const int s = dual ? 6 : n;
  float A[6][6], y[6], dq[MSK_IK_MAX_JOINTS];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (dual) y[i] = del[i];
    else { /* J^T delta */
      float r = 0.0f;
#pragma unroll
      for (int t = 0; t < 6; ++t) r = fmaf(J[t][i], del[t], r);
      y[i] = r;
    }
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float acc = (i == j) ? d.damping : 0.0f;
      if (dual) {
#pragma unroll
        for (int k = 0; k < MSK_IK_MAX_JOINTS; ++k) acc = fmaf(J[i][k], J[j][k], acc);      /* columns k >= n are zero */
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) acc = fmaf(J[t][i], J[t][j], acc);
      }
      A[i][j] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (i >= s) continue;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float acc = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) acc = fmaf(-A[i][k], A[j][k], acc);
      A[i][j] = (i == j) ? sqrtf(acc) : acc / A[j][j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (i >= s) continue;
    float acc = y[i];
#pragma unroll
    for (int k = 0; k < i; ++k) acc = fmaf(-A[i][k], y[k], acc);
    y[i] = acc / A[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    if (i >= s) continue;
    float acc = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k)
      if (k < s) acc = fmaf(-A[k][i], y[k], acc);
    y[i] = acc / A[i][i];
  }
#pragma unroll
  for (int k = 0; k < MSK_IK_MAX_JOINTS; ++k) {
    float acc = 0.0f;
    if (dual) {
#pragma unroll
      for (int r = 0; r < 6; ++r) acc = fmaf(J[r][k], y[r], acc);
    } else if (k < 6) acc = y[k];
    dq[k] = acc;
  }
#pragma unroll
  for (int k = 0; k < MSK_IK_MAX_JOINTS; ++k) {
    if (k >= n) continue;
    const float t = (d.alpha == 1.0f) ? E[m->lay.q + d.dofs[k]] + dq[k] : fmaf(d.alpha, dq[k], E[m->lay.q + d.dofs[k]]);
    if (out) out[(size_t)e * n + k] = t;
    if (commit) E[m->lay.qt + d.dofs[k]] = t;
  }
}

/* sum of the contact impulses applied on body x by body y (the pair-impulse query of scene.py:771-781) */
MSK_DEV v3 pair_impulse(const DModel* m, const DState& st, int e, int x, int y) {
  const int* cnts = st.ct_cnt + (size_t)e * m->npp;
  const float* recs = st.ct_rec + (size_t)e * m->npp * MSK_CT_REC;
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
    const v3 n = v3_make(rec[0], rec[1], rec[2]);
    v3 t1, t2;
    msk_tangents(n, &t1, &t2);
    for (int k = 0; k < cnt; ++k) {
      const float l0 = rec[20 + k * 3 + 0], l1 = rec[20 + k * 3 + 1], l2 = rec[20 + k * 3 + 2];
      const v3 imp = v3_madd(v3_madd(v3_scale(n, l0), t1, l1), t2, l2);
      sum = v3_madd(sum, imp, sgn);
    }
  }
  return sum;
}

/* the same sum over a pair list prepared on the host (the candidate pairs between two bodies, ascending: the order of the scan
 * above): the observe kernels ask for finger-object impulses every step and need not walk the whole pair table for them */
struct PairSel { int n; int idx[14]; float sgn[14]; };
MSK_DEV v3 pair_impulse_sel(const DModel* m, const DState& st, int e, const PairSel& ps) {
  const int* cnts = st.ct_cnt + (size_t)e * m->npp;
  const float* recs = st.ct_rec + (size_t)e * m->npp * MSK_CT_REC;
  v3 sum = v3_make(0, 0, 0);
  for (int i = 0; i < ps.n; ++i) {
    const int p = ps.idx[i];
    const int cnt = cnts[p];
    if (cnt == 0) continue;
    const float* rec = recs + (size_t)p * MSK_CT_REC;
    const v3 n = v3_make(rec[0], rec[1], rec[2]);
    v3 t1, t2;
    msk_tangents(n, &t1, &t2);
    for (int k = 0; k < cnt; ++k) {
      const float l0 = rec[20 + k * 3 + 0], l1 = rec[20 + k * 3 + 1], l2 = rec[20 + k * 3 + 2];
      const v3 imp = v3_madd(v3_madd(v3_scale(n, l0), t1, l1), t2, l2);
      sum = v3_madd(sum, imp, ps.sgn[i]);
    }
  }
  return sum;
}

MSK_DEV float v3_norm_plain(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

/* Panda.is_grasping for one finger: force >= min_force and angle(finger opening direction, force) <= max_angle */
MSK_DEV bool finger_grasps(v3 force, v3 dir, float min_force, float cos_max_angle) {
  const float fn = v3_norm_plain(force);
  const float dn = v3_norm_plain(dir);
  const float fi = 1.0f / fmaxf(fn, 1e-8f), di = 1.0f / fmaxf(dn, 1e-8f);
  float c = (dir.x * di) * (force.x * fi) + (dir.y * di) * (force.y * fi) + (dir.z * di) * (force.z * fi);
  c = fminf(fmaxf(c, -1.0f), 1.0f);
  return fn >= min_force && c >= cos_max_angle;
}

MSK_DEV void pickcube_observe_env(const DModel* __restrict__ m, const DState& st, const msk_pickcube_desc& d, const PairSel& lsel, const PairSel& rsel,
                                  float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags,
                                  int* __restrict__ elapsed, int advance, float cos_max_angle, const int e) {
  const float* E = EREC(st, m, e);
  const int nq = d.arm_dofs + 2;
  float* o = obs + (size_t)e * 42;
  float qv2 = 0.0f, qvmax = 0.0f;
  for (int j = 0; j < nq; ++j) {
    const float q = E[m->lay.q + j], qd = E[m->lay.qd + j];
    o[j] = q;
    o[nq + j] = qd;
    if (j < d.arm_dofs) { qv2 += qd * qd; qvmax = fmaxf(qvmax, fabsf(qd)); }
  }
  const pose cube = load_pose(E, m->lay.bpose, d.cube), tcp = load_pose(E, m->lay.bpose, d.tcp);
  const v3 goal = load_pose(E, m->lay.bpose, d.goal).p;
  const pose lf = load_pose(E, m->lay.bpose, d.left_finger), rf = load_pose(E, m->lay.bpose, d.right_finger);
  /* is_grasping: contact forces = impulses of the last substep / dt */
  const float inv_dt = 1.0f / m->cfg.timestep;
  const v3 lforce = v3_scale(pair_impulse_sel(m, st, e, lsel), inv_dt);
  const v3 rforce = v3_scale(pair_impulse_sel(m, st, e, rsel), inv_dt);
  const m33 Rl = quat_to_m33(lf.q), Rr = quat_to_m33(rf.q);
  const v3 ldir = m33_col(&Rl, 1), rdir = v3_neg(m33_col(&Rr, 1));
  const bool grasped = finger_grasps(lforce, ldir, d.min_force, cos_max_angle) && finger_grasps(rforce, rdir, d.min_force, cos_max_angle);
  const v3 c2g = v3_sub(goal, cube.p), t2c = v3_sub(cube.p, tcp.p);
  const float dist_goal = v3_norm_plain(c2g), dist_tcp = v3_norm_plain(t2c);
  const bool placed = dist_goal <= d.goal_thresh;
  const bool is_static = qvmax <= d.static_thresh;
  const bool success = placed && is_static;
  /* observation: qpos, qvel, is_grasped, tcp_pose, goal_pos, obj_pose, tcp_to_obj_pos, obj_to_goal_pos */
  int k = 2 * nq;
  o[k++] = grasped ? 1.0f : 0.0f;
  o[k++] = tcp.p.x; o[k++] = tcp.p.y; o[k++] = tcp.p.z; o[k++] = tcp.q.w; o[k++] = tcp.q.x; o[k++] = tcp.q.y; o[k++] = tcp.q.z;
  o[k++] = goal.x; o[k++] = goal.y; o[k++] = goal.z;
  o[k++] = cube.p.x; o[k++] = cube.p.y; o[k++] = cube.p.z; o[k++] = cube.q.w; o[k++] = cube.q.x; o[k++] = cube.q.y; o[k++] = cube.q.z;
  o[k++] = t2c.x; o[k++] = t2c.y; o[k++] = t2c.z;
  o[k++] = c2g.x; o[k++] = c2g.y; o[k++] = c2g.z;
  /* compute_normalized_dense_reward */
  float r = 1.0f - tanhf(5.0f * dist_tcp);
  r += grasped ? 1.0f : 0.0f;
  r += (1.0f - tanhf(5.0f * dist_goal)) * (grasped ? 1.0f : 0.0f);
  r += (1.0f - tanhf(5.0f * sqrtf(qv2))) * (placed ? 1.0f : 0.0f);
  if (success) r = 5.0f;
  reward[e] = r / 5.0f;
  int el = elapsed[e] + (advance ? 1 : 0);
  elapsed[e] = el;
  uint8_t* f = flags + (size_t)e * 8;
  f[0] = success; f[1] = placed; f[2] = is_static; f[3] = grasped;
  f[4] = success;                       /* terminated */
  f[5] = el >= d.max_episode_steps;     /* truncated (TimeLimitWrapper) */
  f[6] = 0; f[7] = 0;
}
__global__ void __launch_bounds__(64) k_pickcube_observe(const DModel* __restrict__ m, DState st, msk_pickcube_desc d, PairSel lsel, PairSel rsel,
                                                         float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags,
                                                         int* __restrict__ elapsed, int advance, float cos_max_angle) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= m->N) return;
  pickcube_observe_env(m, st, d, lsel, rsel, obs, reward, flags, elapsed, advance, cos_max_angle, e);
}
/* the same behind the link frames of the post-step (q, qd): k_kinematics and the observation in one launch (a control step ends with both;
 * one dependent launch less).  LPE lanes per env do the frames (msk_dynamics.h), then the env's first lane the observation. */
template <int LPE>
__global__ void __launch_bounds__(64) k_pickcube_observe_kin(const DModel* __restrict__ m, DState st, msk_pickcube_desc d, PairSel lsel, PairSel rsel,
                                                             float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags,
                                                             int* __restrict__ elapsed, int advance, float cos_max_angle) {
  extern __shared__ __attribute__((aligned(16))) float lds_ok[];
  kinematics_block<LPE>(m, st, lds_ok, blockIdx.x);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the frames just stored are read back by the env's first lane */
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int e = blockIdx.x * (64 / LPE) + threadIdx.x / LPE;
  if (threadIdx.x % LPE == 0 && e < m->N) pickcube_observe_env(m, st, d, lsel, rsel, obs, reward, flags, elapsed, advance, cos_max_angle, e);
}

/* ---- PegInsertionSide-v1 ------------------------------------------------------------------------------------------- */
struct PegTables { const float* half; const float* hole; const float* radius; };   /* [N][3], [N][3], [N] */

MSK_DEV pose pose_from_p(float x, float y, float z) {
  pose r;
  r.p = v3_make(x, y, z);
  r.q = quat_make(1.0f, 0.0f, 0.0f, 0.0f);
  return r;
}

MSK_DEV void peg_observe_env(const DModel* __restrict__ m, const DState& st, const msk_pickcube_desc& d, const PairSel& lsel, const PairSel& rsel, const PegTables& tb,
                             float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags, int* __restrict__ elapsed,
                             float* __restrict__ head_at_hole, const int advance, const float cos_max_angle, const int e) {
  const float* E = EREC(st, m, e);
  const int nq = d.arm_dofs + 2;
  float* o = obs + (size_t)e * 43;
  for (int j = 0; j < nq; ++j) {
    o[j] = E[m->lay.q + j];
    o[nq + j] = E[m->lay.qd + j];
  }
  const pose peg = load_pose(E, m->lay.bpose, d.cube), tcp = load_pose(E, m->lay.bpose, d.tcp), box = load_pose(E, m->lay.bpose, d.goal);
  const pose lf = load_pose(E, m->lay.bpose, d.left_finger), rf = load_pose(E, m->lay.bpose, d.right_finger);
  const float hl = tb.half[e * 3], hr1 = tb.half[e * 3 + 1], hr2 = tb.half[e * 3 + 2], rr = tb.radius[e];
  const pose hole = pose_mul(box, pose_from_p(tb.hole[e * 3], tb.hole[e * 3 + 1], tb.hole[e * 3 + 2]));
  const pose head = pose_mul(peg, pose_from_p(hl, 0.0f, 0.0f));
  const v3 inside = pose_mul(pose_inv(hole), head).p;                       /* peg head in the hole's frame */
  const bool success = -0.015f <= inside.x && -rr <= inside.y && inside.y <= rr && -rr <= inside.z && inside.z <= rr;
  /* is_grasping(max_angle=20): contact forces = impulses of the last substep / dt */
  const float inv_dt = 1.0f / m->cfg.timestep;
  const v3 lforce = v3_scale(pair_impulse_sel(m, st, e, lsel), inv_dt);      /* (the finger-peg candidate pairs, prepared on the host: the same sum in the same order as a scan of the pair table) */
  const v3 rforce = v3_scale(pair_impulse_sel(m, st, e, rsel), inv_dt);
  const m33 Rl = quat_to_m33(lf.q), Rr = quat_to_m33(rf.q);
  const v3 ldir = m33_col(&Rl, 1), rdir = v3_neg(m33_col(&Rr, 1));
  const bool grasped = finger_grasps(lforce, ldir, d.min_force, cos_max_angle) && finger_grasps(rforce, rdir, d.min_force, cos_max_angle);
  int k = 2 * nq;
  o[k++] = tcp.p.x; o[k++] = tcp.p.y; o[k++] = tcp.p.z; o[k++] = tcp.q.w; o[k++] = tcp.q.x; o[k++] = tcp.q.y; o[k++] = tcp.q.z;
  o[k++] = peg.p.x; o[k++] = peg.p.y; o[k++] = peg.p.z; o[k++] = peg.q.w; o[k++] = peg.q.x; o[k++] = peg.q.y; o[k++] = peg.q.z;
  o[k++] = hl; o[k++] = hr1; o[k++] = hr2;
  o[k++] = hole.p.x; o[k++] = hole.p.y; o[k++] = hole.p.z; o[k++] = hole.q.w; o[k++] = hole.q.x; o[k++] = hole.q.y; o[k++] = hole.q.z;
  o[k++] = rr;
  /* compute_normalized_dense_reward (peg_insertion_side.py:279-337) */
  const float g = grasped ? 1.0f : 0.0f;
  const v3 grip_target = pose_mul(peg, pose_from_p(-0.06f, 0.0f, 0.0f)).p;
  float r = 1.0f - tanhf(4.0f * v3_norm_plain(v3_sub(tcp.p, grip_target)));
  r += g;
  const pose ginv = pose_inv(pose_mul(hole, pose_from_p(-hl, 0.0f, 0.0f)));   /* goal pose of the peg: head at the hole's centre */
  const v3 head_g = pose_mul(ginv, head).p, peg_g = pose_mul(ginv, peg).p;
  const float head_yz = sqrtf(head_g.y * head_g.y + head_g.z * head_g.z), peg_yz = sqrtf(peg_g.y * peg_g.y + peg_g.z * peg_g.z);
  r += 3.0f * (1.0f - tanhf(0.5f * (head_yz + peg_yz) + 4.5f * fmaxf(head_yz, peg_yz))) * g;
  const bool pre_inserted = head_yz < 0.01f && peg_yz < 0.01f;
  r += 5.0f * (1.0f - tanhf(5.0f * v3_norm_plain(inside))) * ((grasped && pre_inserted) ? 1.0f : 0.0f);
  if (success) r = 10.0f;
  reward[e] = r / 10.0f;
  head_at_hole[e * 3] = inside.x; head_at_hole[e * 3 + 1] = inside.y; head_at_hole[e * 3 + 2] = inside.z;
  const int el = elapsed[e] + (advance ? 1 : 0);
  elapsed[e] = el;
  uint8_t* f = flags + (size_t)e * 8;
  f[0] = success; f[1] = 0; f[2] = 0; f[3] = grasped;
  f[4] = success;                       /* terminated */
  f[5] = el >= d.max_episode_steps;     /* truncated (TimeLimitWrapper) */
  f[6] = 0; f[7] = 0;
}

/* lane = env: the link frames are current (a reset's observation behind msk_update_kinematics) */
__global__ void __launch_bounds__(64) k_peg_observe(const DModel* __restrict__ m, DState st, msk_pickcube_desc d, PairSel lsel, PairSel rsel, PegTables tb,
                                                    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags,
                                                    int* __restrict__ elapsed, float* __restrict__ head_at_hole, int advance,
                                                    float cos_max_angle) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e < m->N) peg_observe_env(m, st, d, lsel, rsel, tb, obs, reward, flags, elapsed, head_at_hole, advance, cos_max_angle, e);
}

/* behind a control step: the link frames of the post-step (q, qd) and the observation in one launch, as k_pickcube_observe_kin (round 6: the observation was a launch
 * of its own behind k_kinematics, one lane per env over 16 workgroups, scanning the whole pair table twice for the finger impulses: 13 + 46 us) */
template <int LPE>
__global__ void __launch_bounds__(64) k_peg_observe_kin(const DModel* __restrict__ m, DState st, msk_pickcube_desc d, PairSel lsel, PairSel rsel, PegTables tb,
                                                        float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ flags,
                                                        int* __restrict__ elapsed, float* __restrict__ head_at_hole, int advance,
                                                        float cos_max_angle) {
  extern __shared__ __attribute__((aligned(16))) float lds_ok[];      /* (the name k_pickcube_observe_kin gives its carve: one symbol for the emulation to back) */
  kinematics_block<LPE>(m, st, lds_ok, blockIdx.x);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the frames just stored are read back by the env's first lane */
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int e = blockIdx.x * (64 / LPE) + threadIdx.x / LPE;
  if (threadIdx.x % LPE == 0 && e < m->N) peg_observe_env(m, st, d, lsel, rsel, tb, obs, reward, flags, elapsed, head_at_hole, advance, cos_max_angle, e);
}

/* ---- PushT-v1 ------------------------------------------------------------------------------------------------------ */
struct PushTTables {
  const unsigned short* src;   /* [nsrc] masked source pixels of the T in its own frame: row << 8 | column (row-major order) */
  int nsrc;
  const unsigned* hit;         /* [128] bit (ix * 64 + iy): a mark at (ix, iy) lands on the goal T's mask (permute + flip folded in) */
};

__global__ void __launch_bounds__(256) k_pusht_set_action(const DModel* __restrict__ m, DState st, msk_pusht_desc d,
                                                          const float* __restrict__ actions) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= m->N) return;
  float* E = EREC(st, m, e);
  const float* a = actions + (size_t)e * d.arm_dofs;
  for (int j = 0; j < d.arm_dofs; ++j) {
    const float aj = fminf(fmaxf(a[j], -1.0f), 1.0f);
    E[m->lay.qt + j] = E[m->lay.q + j] + d.arm_delta * aj;
  }
}

/* quat_to_z_euler (push_t.py:322-327): 2 acos(clamp(w * sign(z))) */
MSK_DEV float pusht_z_euler(quat q) {
  const float sg = (q.z < 0.0f) ? -1.0f : 1.0f;
  return 2.0f * acosf(fminf(fmaxf(q.w * sg, -1.0f), 1.0f));
}

/* one wavefront per env: the T's mask pixels are pushed through goal-from-world x world-from-T and marked in a 64 x 64 bit
 * image in LDS; the marks that fall on the goal mask are counted */
/* KIN: behind a control step -- the env's link frames first (the wavefront is the env's: the 64-lane form of the forward pass), then the observation: one launch
 * instead of k_kinematics + this one */
template <bool KIN>
__global__ void __launch_bounds__(64) k_pusht_observe(const DModel* __restrict__ m, DState st, msk_pusht_desc d, PushTTables tb,
                                                      float* __restrict__ obs, int obs_dim, float* __restrict__ reward,
                                                      uint8_t* __restrict__ flags, int* __restrict__ elapsed, int advance) {
  __shared__ unsigned img[128];
  if (KIN) {
    extern __shared__ __attribute__((aligned(16))) float lds_ok[];
    kinematics_block<64>(m, st, lds_ok, blockIdx.x);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the frames just stored are read back below */
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const int e = blockIdx.x, lane = threadIdx.x;
  const float* E = EREC(st, m, e);
  img[lane] = 0u; img[lane + 64] = 0u;
  const pose tee = load_pose(E, m->lay.bpose, d.tee);
  const float a = pusht_z_euler(tee.q);
  const float ca = cosf(a), sa = sinf(a);
  /* T = world_to_goal @ [[c, -s, x], [s, c, y], [0, 0, 1]] */
  const float* W = d.world_to_goal;
  const float t00 = W[0] * ca + W[1] * sa, t01 = W[0] * -sa + W[1] * ca, t02 = W[0] * tee.p.x + W[1] * tee.p.y + W[2];
  const float t10 = W[3] * ca + W[4] * sa, t11 = W[3] * -sa + W[4] * ca, t12 = W[3] * tee.p.x + W[4] * tee.p.y + W[5];
  __syncthreads();
  for (int j = lane; j < tb.nsrc; j += 64) {
    const int rc = tb.src[j], r = rc >> 8, c = rc & 255;
    const float u = ((float)(c - 32) + 0.5f) / d.uv_scale, v = ((float)(32 - r) + 0.5f) / d.uv_scale;
    const float x = t00 * u + t01 * v + t02, y = t10 * u + t11 * v + t12;
    int ix = (int)(x * d.uv_scale + 32.0f), iy = (int)(y * d.uv_scale + 32.0f);   /* .long(): towards zero */
    if (ix < 0 || ix >= 64 || iy < 0 || iy >= 64) { ix = 0; iy = 0; }
    const int bit = ix * 64 + iy;
    atomicOr(&img[bit >> 5], 1u << (bit & 31));
  }
  __syncthreads();
  int cnt = __popc(img[lane] & tb.hit[lane]) + __popc(img[lane + 64] & tb.hit[lane + 64]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane != 0) return;
  const float inter = (float)cnt / (float)tb.nsrc;
  const bool success = inter >= d.intersection_thresh;
  /* observation */
  float* o = obs + (size_t)e * obs_dim;
  const int nq = d.arm_dofs;
  for (int j = 0; j < nq; ++j) { o[j] = E[m->lay.q + j]; o[nq + j] = E[m->lay.qd + j]; }
  const pose tcp = load_pose(E, m->lay.bpose, d.tcp);
  float* ot = o + 2 * nq;
  ot[0] = tcp.p.x; ot[1] = tcp.p.y; ot[2] = tcp.p.z; ot[3] = tcp.q.w; ot[4] = tcp.q.x; ot[5] = tcp.q.y; ot[6] = tcp.q.z;
  const pose goal = load_pose(E, m->lay.bpose, d.goal);
  if (obs_dim >= 2 * nq + 17) {
    float* og = ot + 7;
    og[0] = goal.p.x; og[1] = goal.p.y; og[2] = goal.p.z;
    og[3] = tee.p.x; og[4] = tee.p.y; og[5] = tee.p.z; og[6] = tee.q.w; og[7] = tee.q.x; og[8] = tee.q.y; og[9] = tee.q.z;
  }
  /* compute_dense_reward / 3 (push_t.py:511-540) */
  const float rot_rew = cosf(a - d.goal_z_rot);
  const float h = (rot_rew + 1.0f) / 2.0f;
  float rew = (h * h) / 2.0f;
  const float dgx = tee.p.x - goal.p.x, dgy = tee.p.y - goal.p.y;
  const float d_goal = sqrtf(dgx * dgx + dgy * dgy);
  const float g1 = 1.0f - tanhf(5.0f * d_goal);
  rew = rew + (g1 * g1) / 2.0f;
  const float dx = tee.p.x - tcp.p.x, dy = tee.p.y - tcp.p.y, dz = tee.p.z - tcp.p.z;
  const float d_tcp = sqrtf(dx * dx + dy * dy + dz * dz);
  rew = rew + sqrtf(1.0f - tanhf(5.0f * d_tcp)) / 20.0f;
  if (success) rew = 3.0f;
  reward[e] = rew / 3.0f;
  int el = elapsed[e] + (advance ? 1 : 0);
  elapsed[e] = el;
  uint8_t* f = flags + (size_t)e * 8;
  f[0] = success; f[1] = 0; f[2] = 0; f[3] = 0;
  f[4] = success;
  f[5] = el >= d.max_episode_steps;
  f[6] = 0; f[7] = 0;
}

#endif
