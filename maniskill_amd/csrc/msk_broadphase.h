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

/* What the lanes of a broadphase wavefront ask of the TEMPLATE -- lane s its shape, lane p the candidate pairs p and 64 + p --: the same for every env, so it is fetched
 * once per wavefront (in k_dynamics: while the wavefront waits for the link frames) instead of once per env.  Per env that was a chain of dependent loads (pair -> its two
 * shapes -> their types; shape -> local pose, box, instance slot) in front of the first use of the env record: ~10 k cycles per env, of which the arithmetic is a small part
 * (round 6, call 16: the wavefront that runs the broadphase of three env blocks ended 9 us after their dynamics). */
struct BpConst {
  int stype, xs, body;   /* shape `lane`: type (-1: no such shape), instance slot, body */
  pose local;
  v3 ac, ah;
  int sa[2], sb[2], ta[2], tb[2];   /* pairs lane, 64 + lane: shapes and their types (the first pair's for a lane without one: it takes part in the ballots) */
};
MSK_DEV BpConst bp_const(const DModel* __restrict__ m) {
  const int lane = threadIdx.x & 63;
  BpConst k;
  k.stype = -1; k.xs = -1; k.body = -1;
  k.local.p = v3_make(0, 0, 0); k.local.q.w = 1.0f; k.local.q.x = k.local.q.y = k.local.q.z = 0.0f;
  k.ac = k.ah = v3_make(0, 0, 0);
  if (lane < m->ns) {
    const DShape* sh = &m->shapes[lane];
    k.stype = sh->type; k.xs = m->xs_slot[lane]; k.body = sh->body; k.local = sh->local; k.ac = sh->aabb_c; k.ah = sh->aabb_h;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pi = j * 64 + lane;
    const DPair pr = m->pairs[pi < m->np ? pi : 0];
    k.sa[j] = pr.sa; k.sb[j] = pr.sb;
    k.ta[j] = m->shapes[pr.sa].type; k.tb[j] = m->shapes[pr.sb].type;
  }
  return k;
}

/* the whole wavefront works on env e; aabb / obb: LDS scratch of the calling block, free to be overwritten */
MSK_DEV void broadphase_env(const DModel* __restrict__ m, const DState& st, const int e, float (*aabb)[6], float (*obb)[13], const BpConst& k) {
  const int lane = threadIdx.x & 63;   /* (the broadphase wave of k_dynamics is the workgroup's last one) */
  const float* E = EREC(st, m, e);
  const float margin = 2.0f * m->cfg.contact_offset;
  int* cnts = st.ct_cnt + (size_t)e * m->npp;
  /* the contact counts of my two pairs (what a culled pair gives back), asked for now: the answer arrives behind the shape pass instead of being waited for inside the pair pass */
  const int cn0 = lane < m->np ? cnts[lane] : 0, cn1 = 64 + lane < m->np ? cnts[64 + lane] : 0;
  if (k.stype >= 0 && k.stype != MSK_SHAPE_PLANE) {   /* shape_pose_dev / shape_half_dev of shape `lane` on the fetched constants */
    pose T = k.local;
    v3 hl = k.ah;
    if (k.xs >= 0) { const float* x = E + m->lay.xshape + k.xs * 8; hl = v3_make(x[0], x[1], x[2]); T.p = v3_make(x[4], x[5], x[6]); }
    if (k.body >= 0) T = pose_mul(load_pose(E, m->lay.bpose, k.body), T);
    v3 c, h;
    world_aabb(k.ac, hl, &T, &c, &h);
    aabb[lane][0] = c.x; aabb[lane][1] = c.y; aabb[lane][2] = c.z;
    aabb[lane][3] = h.x; aabb[lane][4] = h.y; aabb[lane][5] = h.z;
    const m33 R = quat_to_m33(T.q);
#pragma unroll
    for (int j = 0; j < 3; ++j) { obb[lane][j * 3] = R.m[0][j]; obb[lane][j * 3 + 1] = R.m[1][j]; obb[lane][j * 3 + 2] = R.m[2][j]; }
    obb[lane][9] = hl.x; obb[lane][10] = hl.y; obb[lane][11] = hl.z;
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  int base[NP_TYPES] = {0, 0, 0};
  int dropped = 0;
  /* one pass of 64 candidate pairs: lane's pair is pi = p0 + lane, shapes sa, sb of types ta, tb (the whole wave runs it: ballots) */
  auto pass = [&](const int p0, const int sa, const int sb, const int ta, const int tb, const int cn) {   /* cn: the pair's contact count, < 0: not fetched yet */
    const int pi = p0 + lane;
    const bool valid = pi < m->np;
    bool keep = false;
    int type;
    if (ta == MSK_SHAPE_PLANE || tb == MSK_SHAPE_PLANE) {
      type = NP_PLANE;
      const int pa = ta == MSK_SHAPE_PLANE;
      const DShape* P = &m->shapes[pa ? sa : sb];
      const int sc = pa ? sb : sa;
      if ((pa ? tb : ta) != MSK_SHAPE_PLANE) {
        const pose TP = shape_pose_dev(m, E, P);
        const v3 cc = v3_make(aabb[sc][0], aabb[sc][1], aabb[sc][2]), ch = v3_make(aabb[sc][3], aabb[sc][4], aabb[sc][5]);
        const v3 pn = quat_rotate(TP.q, v3_make(1, 0, 0));
        const float lo = v3_dot(pn, cc) - v3_dot(pn, TP.p) - (fabsf(pn.x) * ch.x + fabsf(pn.y) * ch.y + fabsf(pn.z) * ch.z);
        keep = valid && !(lo > margin);
      }
    } else {
      type = (ta == MSK_SHAPE_BOX && tb == MSK_SHAPE_BOX) ? NP_BOXBOX : NP_GJK;
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
        for (int kk = 0; kk < 6; ++kk) {
          const v3 L = (kk < 3) ? au[kk] : bu[kk - 3];
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
    if (!keep && pi < m->np) { gone = cn >= 0 ? cn : cnts[pi]; if (gone != 0) { cnts[pi] = 0; gone = ct_blocks(m, pi, gone); } }
    dropped += gone;
  };
  pass(0, k.sa[0], k.sb[0], k.ta[0], k.tb[0], cn0);
  if (m->np > 64) pass(64, k.sa[1], k.sb[1], k.ta[1], k.tb[1], cn1);
  for (int p0 = 128; p0 < m->np; p0 += 64) {   /* (templates with more than 128 candidate pairs: the rest is fetched per env, as all of it used to be) */
    const DPair pr = m->pairs[p0 + lane < m->np ? p0 + lane : 0];
    pass(p0, pr.sa, pr.sb, m->shapes[pr.sa].type, m->shapes[pr.sb].type, -1);
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
