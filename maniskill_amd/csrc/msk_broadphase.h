/* Broadphase of one env by one wavefront: shape AABBs and oriented boxes, candidate-pair culling, the three narrowphase work
 * lists.  It used to be a launch of its own (9 us + the ~10 us a dependent launch costs behind its predecessor); it only needs
 * the body poses of the previous substep, so it now runs as the tail of k_dynamics (msk_dynamics.h), whose blocks own the same envs. */
#ifndef MSK_BROADPHASE_H
#define MSK_BROADPHASE_H

#include "msk_collide.h"

/* ---- collision -------------------------------------------------------------------------- */
/* half sizes / local position of a shape as this env instantiates it (declared boxes: from the env record) */
MSK_DEV v3 shape_half_dev(const DModel* m, const float* E, const DShape* sh) {
  const int xs = m->xs_slot[sh - m->shapes];
  if (xs < 0) return sh->aabb_h;
  const float* x = E + m->lay.xshape + xs * 8;
  return v3_make(x[0], x[1], x[2]);
}
MSK_DEV pose shape_pose_dev(const DModel* m, const float* E, const DShape* sh) {
  pose L = sh->local;
  const int xs = m->xs_slot[sh - m->shapes];
  if (xs >= 0) { const float* x = E + m->lay.xshape + xs * 8; L.p = v3_make(x[4], x[5], x[6]); }
  if (sh->body < 0) return L;
  return pose_mul(load_pose(E, m->lay.bpose, sh->body), L);
}
MSK_DEV CShape cshape_env(const DModel* m, const float* E, const DShape* sh) {
  CShape c = cshape_of(sh);
  if (m->xs_slot[sh - m->shapes] >= 0) { const v3 h = shape_half_dev(m, E, sh); c.par[0] = h.x; c.par[1] = h.y; c.par[2] = h.z; }
  return c;
}

/* Collision runs in two kernels.
 *   broadphase    (the tail of k_dynamics, one wavefront per env in turn: the poses it reads are the ones the previous substep's
 *                 integration left, k_dynamics does not move bodies) lane s computes the world AABB and the oriented box of shape s (LDS), lane p
 *                 tests candidate pair p (AABBs; for hull pairs also the six face normals of the two oriented boxes);
 *                 survivors are appended to the env's three work lists (one per narrowphase type) by ballot rank — no
 *                 atomics, deterministic order —, culled pairs get their contact slot emptied.
 *   k_narrowphase contact generation + warm-start matching, see the comment at the kernel.
 * The cull tests are the oracle's, so the set of pairs that reach the narrowphase is identical. */
enum { NP_PLANE = 0, NP_BOXBOX = 1, NP_GJK = 2, NP_TYPES = 3 };

/* the whole wavefront works on env e; aabb / obb: LDS scratch of the calling block, free to be overwritten */
MSK_DEV void broadphase_env(const DModel* __restrict__ m, const DState& st, const int e, float (*aabb)[6], float (*obb)[13]) {
  const int lane = threadIdx.x & 63;   /* (the broadphase wave of k_dynamics is the workgroup's second one) */
  const float* E = EREC(st, m, e);
  const float margin = 2.0f * m->cfg.contact_offset;
  if (lane < m->ns) {
    const DShape* sh = &m->shapes[lane];
    if (sh->type != MSK_SHAPE_PLANE) {
      const pose T = shape_pose_dev(m, E, sh);
      v3 c, h;
      const v3 hl = shape_half_dev(m, E, sh);
      world_aabb(sh->aabb_c, hl, &T, &c, &h);
      aabb[lane][0] = c.x; aabb[lane][1] = c.y; aabb[lane][2] = c.z;
      aabb[lane][3] = h.x; aabb[lane][4] = h.y; aabb[lane][5] = h.z;
      const m33 R = quat_to_m33(T.q);
#pragma unroll
      for (int j = 0; j < 3; ++j) { obb[lane][j * 3] = R.m[0][j]; obb[lane][j * 3 + 1] = R.m[1][j]; obb[lane][j * 3 + 2] = R.m[2][j]; }
      obb[lane][9] = hl.x; obb[lane][10] = hl.y; obb[lane][11] = hl.z;
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  int* cnts = st.ct_cnt + (size_t)e * m->npp;
  int base[NP_TYPES] = {0, 0, 0};
  int dropped = 0;
  for (int p0 = 0; p0 < m->np; p0 += 64) {   /* uniform trip count: the ballots below need the whole wave */
    const int pi = p0 + lane;
    const bool valid = pi < m->np;
    const int sa = m->pairs[valid ? pi : 0].sa, sb = m->pairs[valid ? pi : 0].sb;
    const DShape* A = &m->shapes[sa];
    const DShape* B = &m->shapes[sb];
    bool keep = false;
    int type;
    if (A->type == MSK_SHAPE_PLANE || B->type == MSK_SHAPE_PLANE) {
      type = NP_PLANE;
      const int pa = A->type == MSK_SHAPE_PLANE;
      const DShape* P = pa ? A : B;
      const int sc = pa ? sb : sa;
      if (m->shapes[sc].type != MSK_SHAPE_PLANE) {
        const pose TP = shape_pose_dev(m, E, P);
        const v3 cc = v3_make(aabb[sc][0], aabb[sc][1], aabb[sc][2]), ch = v3_make(aabb[sc][3], aabb[sc][4], aabb[sc][5]);
        const v3 pn = quat_rotate(TP.q, v3_make(1, 0, 0));
        const float lo = v3_dot(pn, cc) - v3_dot(pn, TP.p) - (fabsf(pn.x) * ch.x + fabsf(pn.y) * ch.y + fabsf(pn.z) * ch.z);
        keep = valid && !(lo > margin);
      }
    } else {
      type = (A->type == MSK_SHAPE_BOX && B->type == MSK_SHAPE_BOX) ? NP_BOXBOX : NP_GJK;
      keep = valid && !(fabsf(aabb[sa][0] - aabb[sb][0]) > aabb[sa][3] + aabb[sb][3] + margin) &&
             !(fabsf(aabb[sa][1] - aabb[sb][1]) > aabb[sa][4] + aabb[sb][4] + margin) &&
             !(fabsf(aabb[sa][2] - aabb[sb][2]) > aabb[sa][5] + aabb[sb][5] + margin);
      if (keep && type == NP_GJK) { /* second stage (oracle: obb_separated): the two oriented boxes along their six face normals */
        const v3 d = v3_make(aabb[sa][0] - aabb[sb][0], aabb[sa][1] - aabb[sb][1], aabb[sa][2] - aabb[sb][2]);
        v3 au[3], bu[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          au[j] = v3_make(obb[sa][j * 3], obb[sa][j * 3 + 1], obb[sa][j * 3 + 2]);
          bu[j] = v3_make(obb[sb][j * 3], obb[sb][j * 3 + 1], obb[sb][j * 3 + 2]);
        }
        const float hax = obb[sa][9], hay = obb[sa][10], haz = obb[sa][11], hbx = obb[sb][9], hby = obb[sb][10], hbz = obb[sb][11];
        bool sep = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const v3 L = (k < 3) ? au[k] : bu[k - 3];
          const float ra = fmaf(hax, fabsf(v3_dot(au[0], L)), fmaf(hay, fabsf(v3_dot(au[1], L)), haz * fabsf(v3_dot(au[2], L))));
          const float rb = fmaf(hbx, fabsf(v3_dot(bu[0], L)), fmaf(hby, fabsf(v3_dot(bu[1], L)), hbz * fabsf(v3_dot(bu[2], L))));
          if (fabsf(v3_dot(d, L)) > ra + rb + margin) sep = true;
        }
        keep = !sep;
      }
    }
    /* append to this env's per-type list, in pair order (ballot ranks: no atomics, deterministic) */
#pragma unroll
    for (int t = 0; t < NP_TYPES; ++t) {
      const unsigned long long mask = __ballot(keep && type == t);
      if (t == NP_GJK) { /* hull pairs go to the launch-wide queue: their contact slots are addressed by (env, pair), so the order of the
                          * queue (whichever wave's atomic comes first) does not reach the results */
        if (mask != 0ull) {
          int pos = 0;
          if (lane == 0) pos = atomicAdd(st.hq_count, __popcll(mask));
          pos = __builtin_amdgcn_readfirstlane(pos);
          if (keep && type == t) st.hq_items[pos + __popcll(mask & ((1ull << lane) - 1ull))] = e * m->np + pi;
        }
      } else if (keep && type == t) {
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        st.np_items[((size_t)e * NP_TYPES + t) * m->np + base[t] + rank] = pi;
      }
      base[t] += __popcll(mask);
    }
    int gone = 0;
    if (!keep && pi < m->np) { gone = cnts[pi]; if (gone != 0) { cnts[pi] = 0; gone = ct_blocks(m, pi, gone); } }
    dropped += gone;
  }
  if (__ballot(dropped != 0)) { /* keep the env's contact total in step with its row (this wave is the only writer in this launch) */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o);
    if (lane == 0) st.ct_total[e] -= dropped;
  }
  if (lane == 0 && base[NP_GJK] > 0) atomicSub(&st.np_done[e / 64], base[NP_GJK]);   /* the chunk's classification waits for these items too */
  if (lane < NP_TYPES) st.np_count[(size_t)e * 4 + lane] = (lane == 0) ? base[0] : ((lane == 1) ? base[1] : base[2]);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();   /* the next env of this block reuses aabb / obb */
}


#endif
