/*
 * msk_kernels.h — collision, layout-converter and query kernels of one physics substep (gfx950, wave64).
 *
 * Launches of a substep (N envs):
 *   k_dynamics    <<<N/2, 64>>>                half a wavefront per env, lane = body / dof / A^-1 column: msk_dynamics.h; its tail is
 *                                              the broadphase of the block's envs (msk_broadphase.h): AABBs, oriented boxes, pair culling, work lists
 *   k_narrowphase <<<(N/16, 1 + 1 + 4), 64>>>  per 16-env group: plane list, box-box list, four blocks sharing the hull list;
 *                                              the last block of a 64-env chunk sorts its envs into the solver lists
 *   k_csolve      <<<768 + N/4, 64>>>          constraint-space TGS, every capacity class in one launch: msk_solve.h
 *   k_apply / k_fetch / k_kinematics / k_query / k_classify   on demand: layout converters (AoS rows <-> env records), queries
 * All per-env data is env-major (msk_model.h: EnvLayout), so a wave's accesses to its env(s) are contiguous.
 * Arithmetic order mirrors the CPU oracle statement for statement (bitwise parity target).
 */
#ifndef MSK_KERNELS_H
#define MSK_KERNELS_H

#include "msk_collide.h"
#include "msk_collide_lane.h"
#include "msk_solve.h"
#include "msk_broadphase.h"
#include "msk_dynamics.h"

#define MSK_WARM_DIST 5.0e-3f
#define MSK_SPECULATIVE_SLACK 5.0e-3f   /* oracle: ORC_SPECULATIVE_SLACK */
#define MSK_WARM_NORMAL 1.0f    /* fraction of last step's normal (and torsional) impulses applied up front (oracle: ORC_WARM_NORMAL) */
#define MSK_WARM_TANGENT 0.0f   /* ... and of its friction impulses: none (carried over they shake a stack: DESIGN.md §2) */

/* Solver capacity class of env e: constraint blocks = joints that can reach a limit within the step or have a drive row + joints with
 * friction + the blocks of the contact slots (the same count solve_env makes), against the per-template class capacities. */
MSK_DEV int solver_class_of(const DModel* __restrict__ m, const DState& st, const int e, const int contacts) {
  const float* E = EREC(st, m, e);
  int nblk = 0;
  const unsigned long long dm = st.drv_mask[e];   /* joints whose force-limited drive is a solver row in this substep (k_dynamics) */
  for (int d = 0; d < m->nd; ++d) {
    const float lo = m->dof_lo[d], hi = m->dof_hi[d], q = E[m->lay.q + d];
    bool limit = false;
    if (!(lo < -1e30f && hi > 1e30f)) { /* the solver's rule (msk_solve.h): the joint can reach the limit within this step */
      const float vf = st.vfree[(size_t)e * m->G + d], dt = m->cfg.timestep;
      limit = (q - lo < fmaf(2.0f * dt, fmaxf(0.0f, -vf), MSK_LIMIT_SLACK)) || (hi - q < fmaf(2.0f * dt, fmaxf(0.0f, vf), MSK_LIMIT_SLACK));
    }
    if (((dm >> d) & 1ull) || limit) nblk++;
  }
  nblk += m->njfric;
  const int room = m->cap_blocks - nblk > 0 ? m->cap_blocks - nblk : 0;
  nblk += contacts < room ? contacts : room;
  return nblk <= m->cls_cap[0] ? 0 : (nblk <= m->cls_cap[1] ? 1 : (nblk <= m->cls_cap[2] ? 2 : (nblk <= m->cls_cap[3] ? 3 : 4)));
}

/* lanes 0 .. n-1 of the calling wave append envs e0 .. e0+n-1 to their class lists (one atomic per class) */
MSK_DEV void classify_envs(const DModel* __restrict__ m, const DState& st, const int e0, const int n) {
  const int lane = threadIdx.x & 63;
  const bool mine = lane < n && e0 + lane < m->N;
  /* the contact total is kept current by device-scope atomics (narrowphase writers): an atomic load sees what every block of this
   * launch has added before it signed off, without the L2 write-back / invalidate a fence across the XCDs would cost */
  const int cls = mine ? solver_class_of(m, st, e0 + lane, __hip_atomic_load(&st.ct_total[e0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -1;
#pragma unroll
  for (int c = 0; c < MSK_SOLVE_CLASSES; ++c) {
    const unsigned long long mask = __ballot(cls == c);
    if (mask == 0ull) continue;
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&st.cls_count[c], __popcll(mask));
    base = __builtin_amdgcn_readlane(base, __ffsll((long long)mask) - 1);
    if (cls == c) st.cls_list[(size_t)c * m->N + base + __popcll(mask & ((1ull << lane) - 1ull))] = e0 + lane;
  }
}

/* scenes without collision pairs: the narrowphase is not launched, the lists are built here */
__global__ void __launch_bounds__(64) k_classify(const DModel* __restrict__ m, DState st) {
  classify_envs(m, st, blockIdx.x * 64, 64);
}

#define NP_GROUP_MAX 16
#define MSK_DBG_NP_BLOCKS 8192   /* DState::dbg, profiling builds: behind the per-env stamps, a word per narrowphase workgroup (launch position) and, for the first 4096, a word of its phases */
/* One wavefront per (env group, narrowphase list).
 *   plane and box-box lists (blockIdx.y = 0, 1): one lane per surviving pair of `group` envs (msk_collide_lane.h) — the
 *     lists are long, a wave holds up to 64 pairs and fetches the per-pair code once for all of them;
 *   hull list (blockIdx.y = 2 ..): a 16-lane group per pair (msk_collide.h), four pairs in flight, four blocks sharing
 *     the group's list — few pairs, each a long chain of support scans over up to 64 vertices.
 * `group` = 16 at 4096 envs, fewer when there are few envs (then the launch is bound by its slowest wave). */
/* number of plane / box-box blocks whose sign-off a 64-env chunk waits for */
MSK_DEV int chunk_fixed_blocks(const DModel* __restrict__ m, const int chunk, const int group, const int per_group) {
  const int first_blk = (chunk * 64 + group - 1) / group, end_env = min(chunk * 64 + 64, m->N);
  return ((end_env + group - 1) / group - first_blk) * per_group;
}

template <int TYPE, int LPI>   /* LPI = lanes per item: 1 or NPG */
MSK_DEV void narrowphase_body(const DModel* __restrict__ m, const DState& st, const int e0, const int group, const int part,
                              const int nparts, int* pref, float* s_ws, float* s_we, const v3* verts, const int fixed_per_group) {
  constexpr int type = TYPE;
  constexpr bool GLOBALQ = TYPE == NP_GJK;   /* hull items: one queue for the whole launch (part / nparts count over all hull blocks) */
  const int lane = threadIdx.x;
  const float margin = 2.0f * m->cfg.contact_offset;
#ifdef MSK_PROFILE_PHASES   /* where a workgroup's time goes, 100 MHz clock: first pass only: [0] entry, [2] operands fetched, [3] contacts computed, [1] speculative points filtered, [5] old record read / count and total updated, [4] slots written (tools/gpu_phase_probe.py) */
  unsigned long long brt[6];
  brt[0] = brt[1] = brt[2] = brt[3] = brt[4] = brt[5] = __builtin_amdgcn_s_memrealtime();
  bool brt_first = true;
#endif
  int count;
  if constexpr (GLOBALQ) {
    count = *st.hq_count;
  } else { /* exclusive prefix of the group's list lengths: one load per lane, a 16-lane scan */
    int c = (lane < group && e0 + lane < m->N) ? st.np_count[(size_t)(e0 + lane) * 4 + type] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < NP_GROUP_MAX; d <<= 1) {
      const int o = __shfl_up(incl, d, NP_GROUP_MAX);
      if ((lane & (NP_GROUP_MAX - 1)) >= d) incl += o;
    }
    if (lane < NP_GROUP_MAX) pref[lane] = incl - c;
    if (lane == NP_GROUP_MAX - 1) pref[NP_GROUP_MAX] = incl;
    __syncthreads();
    count = pref[NP_GROUP_MAX];
  }
  /* the list is dealt out pass by pass to the `nparts` blocks that share it (part = which one I am): the blocks stay balanced
   * even when one env owns most of the pairs (an arm lying on the table).  Uniform trip count: the sign-off below needs the wave. */
  constexpr int STEP = (LPI == 1) ? PL_LANES : 64 / LPI;   /* pairs per pass */
  for (int it = GLOBALQ ? 0 : part * STEP; it < count; it += STEP * nparts) {
  /* hull queue: item i goes to block i % nparts, so up to nparts items get a wave each (no divergence between the four lane groups of
   * a wave, no taking turns at the wave's EPA workspace) and only a longer queue doubles up */
  const int idx = GLOBALQ ? it + (lane / LPI) * nparts + part : it + ((LPI == 1) ? (lane < PL_LANES ? lane : count) : lane / LPI);
  const bool act = idx < count;
  int e = 0;
  if (act) do {
  int pi;
  if constexpr (GLOBALQ) {
    const int item = st.hq_items[idx];
    e = item / m->np;
    pi = item - e * m->np;
  } else {
    int j = 0;
#pragma unroll
    for (int k = 1; k < NP_GROUP_MAX; ++k) j += (k < group && idx >= pref[k]) ? 1 : 0;
    e = e0 + j;
    pi = st.np_items[((size_t)e * NP_TYPES + type) * m->np + (idx - pref[j])];
  }
  float* E = EREC(st, m, e);
  const DShape* dA = &m->shapes[m->pairs[pi].sa];
  const DShape* dB = &m->shapes[m->pairs[pi].sb];
  pose TA = shape_pose_dev(m, E, dA), TB = shape_pose_dev(m, E, dB);
  CShape cA = cshape_env(m, E, dA), cB = cshape_env(m, E, dB);
  if (TYPE == NP_BOXBOX) cA.type = cB.type = MSK_SHAPE_BOX;   /* known: lets the compiler drop the hull paths */
  const CShape* A = &cA;
  const CShape* B = &cB;
#ifdef MSK_PROFILE_PHASES
  long long tq[6]; tq[0] = (long long)__builtin_readcyclecounter(); tq[1] = tq[2] = tq[5] = tq[0]; tq[3] = tq[4] = 0;
  if (brt_first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); brt[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
  v3 opos[4], onrm = v3_make(0, 0, 1);
  float osep[4];
  int n = 0;
  bool writer = true;
  if constexpr (LPI == 1) { /* box-box: one lane per pair, its polygon arrays interleaved in LDS */
    CCtx cx;
    cx.verts = verts; cx.ws = nullptr; cx.we = nullptr; cx.gl = 0; cx.grp = 0; cx.dbg = nullptr;
    perlane::DContactOut out[4];
    v3 nrm;
    float sep;
    if (sat_box_box(A, &TA, B, &TB, margin, &nrm, &sep)) {
      const v3 wa = support(cx, A, &TA, v3_neg(nrm)), wb = support(cx, B, &TB, nrm);
#ifdef MSK_PROFILE_PHASES
      tq[1] = (long long)__builtin_readcyclecounter();
#endif
      n = perlane::build_manifold(s_ws + lane, A, &TA, B, &TB, nrm, margin, wa, wb, sep, out);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { opos[k] = out[k].pos; osep[k] = out[k].sep; }
    onrm = out[0].n;
  } else {
    CCtx cx;
    cx.verts = verts;
    cx.ws = s_ws + (lane / NPG) * WS_TOTAL;
    cx.we = s_we;
    cx.gl = lane % NPG;
    cx.grp = lane / NPG;
    cx.dbg = nullptr;
#ifdef MSK_PROFILE_PHASES
    cx.gjk_iters = 0; cx.epa_cycles = 0;
#endif
    DContactOut out[4];
    if (type == NP_PLANE) {
      const int pa = A->type == MSK_SHAPE_PLANE;
      n = plane_convex(cx, pa ? A : B, pa ? &TA : &TB, pa ? B : A, pa ? &TB : &TA, margin, pa, out);
    } else {
      v3 nrm, wa, wb;
      float sep;
      int hit;
      if (type == NP_BOXBOX) {
        hit = sat_box_box(A, &TA, B, &TB, margin, &nrm, &sep);
        if (hit) {
          wa = support(cx, A, &TA, v3_neg(nrm));
          wb = support(cx, B, &TB, nrm);
        }
      } else {
        v3 ca, ha, cb, hb;
        const v3 hla = shape_half_dev(m, E, dA), hlb = shape_half_dev(m, E, dB);
        world_aabb(dA->aabb_c, hla, &TA, &ca, &ha);
        world_aabb(dB->aabb_c, hlb, &TB, &cb, &hb);
        /* third cull stage: each shape's vertices against the other's oriented local box (msk_collide.h verts_beyond_obb) */
        hit = !(verts_beyond_obb(cx, A, &TA, &TB, dB->aabb_c, hlb, margin) || verts_beyond_obb(cx, B, &TB, &TA, dA->aabb_c, hla, margin));
#ifdef MSK_PROFILE_PHASES
        tq[5] = (long long)__builtin_readcyclecounter();
#endif
        if (hit) hit = gjk_epa(cx, A, &TA, B, &TB, ca, cb, margin, &nrm, &sep, &wa, &wb, st.gjk_cache + (size_t)e * m->npp + pi);
      }
#ifdef MSK_PROFILE_PHASES
      tq[1] = (long long)__builtin_readcyclecounter();
#endif
      if (hit) n = build_manifold(cx, A, &TA, B, &TB, nrm, margin, wa, wb, sep, out);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { opos[k] = out[k].pos; osep[k] = out[k].sep; }
    onrm = out[0].n;
    writer = cx.gl == 0;   /* the group's first lane owns the contact slot */
#ifdef MSK_PROFILE_PHASES
    tq[3] = cx.epa_cycles; tq[4] = cx.gjk_iters;
#endif
  }
#ifdef MSK_PROFILE_PHASES
  tq[2] = (long long)__builtin_readcyclecounter();
  if (brt_first) brt[3] = __builtin_amdgcn_s_memrealtime();
#ifndef MSK_PROFILE_NO_NPSTATS   /* (the counters below are same-address atomics from every workgroup: ~14 us of a box-box block; -DMSK_PROFILE_NO_NPSTATS leaves the stamps alone) */
  if (writer) { /* per type: [0] items, [1] sum primary, [2] max primary, [3] sum manifold, [4] max manifold, [5] hits */
    unsigned long long* d = (unsigned long long*)st.dbg + (size_t)m->N * 8 + type * 8;
    atomicAdd(&d[0], 1ull);
    atomicAdd(&d[1], (unsigned long long)(tq[1] - tq[0])); atomicMax(&d[2], (unsigned long long)(tq[1] - tq[0]));
    atomicAdd(&d[3], (unsigned long long)(tq[2] - tq[1])); atomicMax(&d[4], (unsigned long long)(tq[2] - tq[1]));
    if (n > 0) atomicAdd(&d[5], 1ull);
    if (type == NP_GJK) { /* row 3: [0] items through EPA, [1] sum / [2] max EPA cycles, [3] sum / [4] max GJK iterations; row 4: histogram of primary */
      unsigned long long* x = (unsigned long long*)st.dbg + (size_t)m->N * 8 + 3 * 8;
      if (tq[3] > 0) { atomicAdd(&x[0], 1ull); atomicAdd(&x[1], (unsigned long long)tq[3]); atomicMax(&x[2], (unsigned long long)tq[3]); }
      atomicAdd(&x[3], (unsigned long long)tq[4]); atomicMax(&x[4], (unsigned long long)tq[4]);
      atomicAdd(&x[5], (unsigned long long)(tq[5] - tq[0])); if (tq[4] > 0) atomicAdd(&x[6], 1ull);   /* cull stage cycles; items that entered GJK */
      const long long pc = tq[1] - tq[0];
      const int bk = pc < 25000 ? 0 : (pc < 50000 ? 1 : (pc < 100000 ? 2 : (pc < 150000 ? 3 : (pc < 200000 ? 4 : (pc < 300000 ? 5 : (pc < 400000 ? 6 : 7))))));
      atomicAdd(&x[8 + bk], 1ull);
    }
  }
#endif
#endif
  if (!writer) break;
  /* Speculative points that cannot touch within this step are no contacts (oracle: orc_collide_pair): a point further apart than
   * MSK_SPECULATIVE_SLACK plus twice what the two bodies' present velocities close along the normal in one step is dropped. */
  if (n > 0) {
    const float dt = m->cfg.timestep;
    const int ba = dA->body, bb = dB->body;
    v3 cwa = v3_make(0, 0, 0), cwb = v3_make(0, 0, 0), la = cwa, lb = cwa, wa_ = cwa, wb_ = cwa;
    if (ba >= 0) {
      const pose Tb = load_pose(E, m->lay.bpose, ba);
      cwa = v3_add(Tb.p, quat_rotate(Tb.q, m->bodies[ba].com));
      la = load_v3(E, m->lay.blin, ba); wa_ = load_v3(E, m->lay.bang, ba);
    }
    if (bb >= 0) {
      const pose Tb = load_pose(E, m->lay.bpose, bb);
      cwb = v3_add(Tb.p, quat_rotate(Tb.q, m->bodies[bb].com));
      lb = load_v3(E, m->lay.blin, bb); wb_ = load_v3(E, m->lay.bang, bb);
    }
    int kept = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= n) continue;
      const v3 va = ba >= 0 ? v3_add(la, v3_cross(wa_, v3_sub(opos[k], cwa))) : v3_make(0, 0, 0);
      const v3 vb = bb >= 0 ? v3_add(lb, v3_cross(wb_, v3_sub(opos[k], cwb))) : v3_make(0, 0, 0);
      const float vn = v3_dot(onrm, v3_sub(va, vb));                          /* < 0: approaching */
      const float reach = fmaf(2.0f * dt, fmaxf(0.0f, -vn), MSK_SPECULATIVE_SLACK);
      if (osep[k] - m->cfg.rest_offset * 2.0f > reach) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == kept && j != k) { opos[j] = opos[k]; osep[j] = osep[k]; }
      kept++;
    }
    n = kept;
  }
#ifdef MSK_PROFILE_PHASES
  if (brt_first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); brt[1] = __builtin_amdgcn_s_memrealtime(); }   /* (the speculative-point filter done) */
#endif
  /* warm start from the previous contents of this pair's slot, then overwrite it */
  int* cntp = st.ct_cnt + (size_t)e * m->npp + pi;
  float* rec = st.ct_rec + ((size_t)e * m->npp + pi) * MSK_CT_REC;
  const int nprev = *cntp;
  v3 ppos[4];
  float plam[4][3];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    if (jj < nprev) {
      ppos[jj] = v3_make(rec[4 + jj * 3 + 0], rec[4 + jj * 3 + 1], rec[4 + jj * 3 + 2]);
#pragma unroll
      for (int a = 0; a < 3; ++a) plam[jj][a] = rec[20 + jj * 3 + a];
    }
  }
  if (n == 0 && nprev == 0) break;
  *cntp = n;
  {
    const int blocks = ct_blocks(m, pi, n), blocks_prev = ct_blocks(m, pi, nprev);
    if (blocks != blocks_prev) atomicAdd(&st.ct_total[e], blocks - blocks_prev);   /* device scope: read by whichever block classifies the env's chunk */
  }
  if (m->has_static) { /* static or dynamic friction: the pair slides when its friction impulses of the last step, summed, ended on the cone
                        * of the coefficient that step used (oracle: collide(), pair_slip) */
    unsigned char* slip = st.ct_slip + (size_t)e * m->npp + pi;
    float T1 = 0.0f, T2 = 0.0f, Nn = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (jj < nprev) { T1 += plam[jj][1]; T2 += plam[jj][2]; Nn += plam[jj][0]; }
    const float mu_used = (nprev > 0 && *slip) ? m->pinfo[pi].mu : m->pinfo[pi].mu_s;
    const float lim = 0.999f * mu_used * Nn;   /* (the friction frame turns with the motion: the length, not the components) */
    *slip = (nprev > 0 && Nn > 0.0f && fmaf(T1, T1, T2 * T2) >= lim * lim) ? 1 : 0;
  }
  const float prev_lam_t = rec[3];
#ifdef MSK_PROFILE_PHASES
  if (brt_first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); brt[5] = __builtin_amdgcn_s_memrealtime(); }   /* (the old record read, count and total updated, slip flag written) */
#endif
  if (n > 0) { rec[0] = onrm.x; rec[1] = onrm.y; rec[2] = onrm.z; rec[3] = 0.0f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= n) continue;
    float lam[3] = {0.0f, 0.0f, 0.0f};
    int best = -1;
    float bd = MSK_WARM_DIST * MSK_WARM_DIST;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (jj >= nprev) continue;
      float d2 = v3_len2(v3_sub(ppos[jj], opos[k]));
      if (d2 < bd) { bd = d2; best = jj; }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (best == jj) { lam[0] = MSK_WARM_NORMAL * plam[jj][0]; lam[1] = MSK_WARM_TANGENT * plam[jj][1]; lam[2] = MSK_WARM_TANGENT * plam[jj][2]; }
    if (k == 0 && n == 1 && nprev == 1 && best == 0) rec[3] = MSK_WARM_TANGENT * prev_lam_t;   /* the torsional row of a one-point manifold */
    rec[4 + k * 3 + 0] = opos[k].x;
    rec[4 + k * 3 + 1] = opos[k].y;
    rec[4 + k * 3 + 2] = opos[k].z;
    rec[16 + k] = osep[k] - m->cfg.rest_offset * 2.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) rec[20 + k * 3 + a] = lam[a];
  }
  } while (0);
#ifdef MSK_PROFILE_PHASES
  if (brt_first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); brt[4] = __builtin_amdgcn_s_memrealtime(); brt_first = false; }
#endif
  if constexpr (GLOBALQ) { /* sign the pass's items off with their chunks (see k_narrowphase); a wave may finish several chunks */
    /* the writer's atomic on the contact total must have been performed before the sign-off is: a workgroup-scope release emits no
     * wait on this target (two global atomics back to back), so the memory counter is drained explicitly -- still no L2 write-back */
    MSK_WAIT_VMCNT0();
    MSK_WAVE_REJOIN();   /* the lane groups rejoin here: the sign-off below is the whole wavefront's */
    const int chunk = e / 64;
    const bool signer = act && (lane % LPI) == 0;
    int old = -1;
    if (signer) old = atomicAdd(&st.np_done[chunk], 1);
    unsigned long long fin = __ballot(signer && old == chunk_fixed_blocks(m, chunk, group, fixed_per_group) - 1);
    while (fin != 0ull) {
      const int L = __ffsll((long long)fin) - 1;
      fin &= fin - 1ull;
      const int ch = __builtin_amdgcn_readlane(chunk, L);
      if (lane == 0) st.np_done[ch] = 0;
      classify_envs(m, st, ch * 64, 64);
    }
  }
  }
#ifdef MSK_PROFILE_PHASES
  {
    const int nb_ = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && nb_ < MSK_DBG_NP_BLOCKS / 2) {
      unsigned long long w = 0ull;
      for (int k = 1; k < 6; ++k) { unsigned long long d = brt[k] - brt[0]; if (d > 4095ull) d = 4095ull; w |= d << (12 * (k - 1)); }
      st.dbg[(size_t)m->N * 16 + 64 + MSK_DBG_NP_BLOCKS / 2 + nb_] = (long long)w;
    }
  }
#endif
}

/* blockIdx.y walks [plane | box-box | hull] blocks of env group blockIdx.x: cfg.nplane / nbox / nhull of them, the
 * blocks of one kind share the group's list of that kind.
 * Per 64 consecutive envs, the block that finishes last sorts them into the solver lists (one wave, one env per lane:
 * a handful of same-address atomics per 64 envs). */
struct NpCfg { int nplane, nbox, nhull; int lds_words; /* dynamic LDS of the launch, in floats (np_lds_words) */ };
#define NP_LDS_BASE ((64 / NPG) * WS_TOTAL + WE_TOTAL)   /* four group workspaces + the wave's EPA workspace; the staged hull vertices follow */
/* LDS a narrowphase block needs for a template with `nverts` hull vertices: the box-box image (PL_LANES x PL_WORDS) or the lane-group
 * workspaces plus the staged vertex pool, whichever is larger; a pool beyond MSK_NP_MAX_STAGED vertices is read from global memory
 * instead.  (Measured, round 3: the size is NOT what limits this kernel -- 19.5 KB per block with 24 box lanes was 4 us slower than
 * 26.6 KB with 32, because groups of 25-32 pairs then need a second pass.) */
#define MSK_NP_MAX_STAGED 1280
static inline int np_lds_words(int nverts) {
  const int box = PL_WORDS * PL_LANES;
  const int grp = NP_LDS_BASE + (nverts <= MSK_NP_MAX_STAGED ? nverts * 3 : 0);
  return box > grp ? box : grp;
}
MSK_DEV void narrowphase_block(const DModel* __restrict__ m, const DState& st, const int group, const NpCfg cfg, const int bx, const int by, const int gx,
                               const int gy) {
  __shared__ int pref[NP_GROUP_MAX + 1];
  /* one LDS image for both kinds of block: PL_LANES lanes x 208 words of per-pair arrays (box-box), or four group workspaces
   * and the wave's EPA workspace (plane, hull) followed by the staged vertex pool; sized by the launch (np_lds_words) */
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int LDS_WORDS = cfg.lds_words;
  float* s_ws = s_lds;
  float* s_we = s_lds + (64 / NPG) * WS_TOTAL;
  const int e0 = bx * group;
  const int fixed_per_group = cfg.nplane + cfg.nbox;
  /* y: [plane | box-box | hull] as everything below counts; by: the row of the grid, which is also the ORDER OF DISPATCH -- hull rows first, then box-box, plane last.  With one
   * wavefront per SIMD the 1792 workgroups of 4096 envs queue for 1024 slots, and the longest ones are hull items (GJK / EPA: 40-60 us): behind the plane and box-box rows the last
   * hull rows of PegInsertionSide started 21-24 us late and ended the launch at 72-75 us, its workgroups lasting 51 at most (round 6, call 23: profiles/r06_launch_position_probe.log) */
  const int y = (by < cfg.nhull) ? fixed_per_group + by : ((by - cfg.nhull < cfg.nbox) ? cfg.nplane + (by - cfg.nhull) : by - cfg.nhull - cfg.nbox);
  const bool lane_kind = (y >= cfg.nplane) && (y - cfg.nplane < cfg.nbox);   /* box-box, one lane per pair: boxes only */
  /* plane and hull blocks scan hull vertices over and over (support points, features): the template's vertex pool (<= 12 KB) is
   * staged in the part of the LDS image that only the lane-per-pair kind uses, so every scan is an LDS read instead of an L2 hit */
  const v3* verts = m->verts;
  if (!lane_kind) {
    float* s_verts = s_lds + (64 / NPG) * WS_TOTAL + WE_TOTAL;
    const int nw = m->nverts_total * 3;
    if (nw <= LDS_WORDS - ((64 / NPG) * WS_TOTAL + WE_TOTAL)) {
      const float* src = (const float*)m->verts;
      for (int i = threadIdx.x; i < nw; i += 64) s_verts[i] = src[i];
      verts = (const v3*)s_verts;
    }
  }
  if (y >= fixed_per_group) { /* hull blocks: the launch-wide queue; their items sign off one by one */
    narrowphase_body<NP_GJK, NPG>(m, st, e0, group, (y - fixed_per_group) * gx + bx, cfg.nhull * gx, pref, s_ws, s_we,
                                  verts, fixed_per_group);
    return;
  }
  if (y < cfg.nplane) narrowphase_body<NP_PLANE, NPG>(m, st, e0, group, y, cfg.nplane, pref, s_ws, s_we, verts, fixed_per_group);
  else narrowphase_body<NP_BOXBOX, 1>(m, st, e0, group, y - cfg.nplane, cfg.nbox, pref, s_ws, s_we, verts, fixed_per_group);
  /* Sign-off.  The only data of this launch the classifying wave reads are the contact totals, and those are device-scope
   * atomics; so all that is needed is that this wave's atomics have been performed before its sign-off is (a wait on the
   * memory counter), not __threadfence(): on this part an agent-scope fence writes back and
   * invalidates the XCD's L2 (an empty launch of these 1536 blocks took 25 us with it, 4 us without: profiles/r02_np_floor.md).
   * np_done[chunk] starts at minus the chunk's hull items (broadphase); plane / box-box blocks and hull items add one each. */
  MSK_WAIT_VMCNT0();   /* (the workgroup-scope release alone emits no wait: the atomics could overtake each other) */
  const int chunk = e0 / 64;
  int done = 0;
  if (threadIdx.x == 0) done = atomicAdd(&st.np_done[chunk], 1);
  done = __builtin_amdgcn_readfirstlane(done);
  if (done == chunk_fixed_blocks(m, chunk, group, fixed_per_group) - 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x == 0) st.np_done[chunk] = 0;
    classify_envs(m, st, chunk * 64, 64);
  }
}

MSK_DEV void narrowphase_launch_body(const DModel* __restrict__ m, const DState& st, const int group, const NpCfg& cfg) {
#ifdef MSK_PROFILE_PHASES
  const unsigned long long nrt0 = __builtin_amdgcn_s_memrealtime();   /* the 100 MHz clock: where in the launch this workgroup ran (tools/gpu_phase_probe.py) */
#endif
  narrowphase_block(m, st, group, cfg, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
#ifdef MSK_PROFILE_PHASES
  const int nb_ = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0 && nb_ < MSK_DBG_NP_BLOCKS / 2)
    st.dbg[(size_t)m->N * 16 + 64 + nb_] = (long long)(((__builtin_amdgcn_s_memrealtime() & 0xffffffffull) << 32) | (nrt0 & 0xffffffffull));
#endif
}
/* Two register budgets of the same code (round 6, profiles/r06_launch_position_probe.log, r06_ab_narrowphase_two_waves_per_simd.log).  k_narrowphase: 256 VGPRs + 45 AGPRs = ONE wavefront
 * per SIMD, 1024 slots for the 1792 workgroups of 4096 envs -- workgroups queue, but each runs as fast as it can, and at 4096 PickCube envs the launch is as long as its longest workgroup
 * (a box-box block, 32-40 us; a hull item through EPA).  k_narrowphase_w2: 256 VGPRs, no AGPRs = two per SIMD, 183 scratch instructions instead of 42: every workgroup ~5 % slower
 * (4096 PickCube envs: 45.2 -> 47.5 us), but launches whose workgroups are mostly busy, or many times the slots, stop queueing: 16384 PegInsertionSide envs 191 -> 154 us, 65536 PickCube
 * envs 213 -> 166 us.  The host picks by env count (step_part). */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_narrowphase(const DModel* __restrict__ m, DState st, const int group, const NpCfg cfg) {
  narrowphase_launch_body(m, st, group, cfg);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_narrowphase_w2(const DModel* __restrict__ m, DState st, const int group, const NpCfg cfg) {
  narrowphase_launch_body(m, st, group, cfg);
}

/* ---- AoS <-> SoA converters ------------------------------------------------------------------ */
struct DBuffers { float* buf[MSK_BUF_COUNT]; int max_dof; int pitch; /* floats between articulation rows (>= max_dof) */ };

/* gpu_apply_rigid_dynamic_force / _torque: the buffer rows become the pending wrench of the next substep */
__global__ void k_apply_wrench(float* __restrict__ wrench, const float* __restrict__ force, const float* __restrict__ torque,
                               int rows, unsigned mask) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  if (mask & MSK_APPLY_RIGID_FORCE) {
    const float4 f = *(const float4*)(force + (size_t)r * 4);
    *(float4*)(wrench + (size_t)r * 8) = make_float4(f.x, f.y, f.z, 0.0f);
  }
  if (mask & MSK_APPLY_RIGID_TORQUE) {
    const float4 t = *(const float4*)(torque + (size_t)r * 4);
    *(float4*)(wrench + (size_t)r * 8 + 4) = make_float4(t.x, t.y, t.z, 0.0f);
  }
}

/* NL = 1: the calling lane does the whole env (k_apply: lane = env); NL = 64: the wavefront shares one env, lane = body / joint / candidate pair (k_reset_masked) */
template <int NL = 1>
MSK_DEV void apply_env(const DModel* __restrict__ m, const DState& st, const DBuffers& bf, unsigned mask, const int* __restrict__ art_dof0,
                       const int* __restrict__ art_ndof, const int e, const int lane = 0) {
  float* E = EREC(st, m, e);
  const float ox = E[m->lay.off + 0], oy = E[m->lay.off + 1], oz = E[m->lay.off + 2];
  bool teleported = false; /* a pose or joint position was overwritten: the env's contact cache is stale */
  for (int i = lane; i < m->nb; i += NL) {
    const DBody* b = &m->bodies[i];
    const float* r = bf.buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * m->nb + i) * 13;
    const bool is_root = b->kind == MSK_BODY_LINK && b->parent < 0;
    if ((b->kind != MSK_BODY_LINK && (mask & MSK_APPLY_RIGID_DATA)) || (is_root && (mask & MSK_APPLY_ART_ROOT_POSE))) {
      /* rows the caller did not touch since the last fetch are left alone: (p + off) - off and
       * re-normalisation are not exact in fp32, and apply must not perturb untouched envs */
      const pose cur = load_pose(E, m->lay.bpose, i);
      const bool same = (r[0] == cur.p.x + ox) && (r[1] == cur.p.y + oy) && (r[2] == cur.p.z + oz) &&
                        (r[3] == cur.q.w) && (r[4] == cur.q.x) && (r[5] == cur.q.y) && (r[6] == cur.q.z);
      if (!same) {
        pose T;
        T.p = v3_make(r[0] - ox, r[1] - oy, r[2] - oz);
        T.q = quat_normalize(quat_make(r[3], r[4], r[5], r[6]));
        store_pose(E, m->lay.bpose, i, T);
        teleported = true;
      }
      if (b->kind == MSK_BODY_DYNAMIC) {
        store_v3(E, m->lay.blin, i, v3_make(r[7], r[8], r[9]));
        store_v3(E, m->lay.bang, i, v3_make(r[10], r[11], r[12]));
      }
    }
    if (is_root && b->root_dof >= 0 && (mask & MSK_APPLY_ART_ROOT_VELOCITY)) { /* the root's six coordinates ARE (v_com, omega) */
      float* qd = E + m->lay.qd + b->root_dof;
#pragma unroll
      for (int k = 0; k < 6; ++k) qd[k] = r[7 + k];
    }
  }
  for (int a = 0; a < m->na; ++a)
    for (int j = lane; j < art_ndof[a]; j += NL) {
      const int d = art_dof0[a] + j;
      const size_t row = ((size_t)e * m->na + a) * bf.pitch + j;
      if (mask & MSK_APPLY_ART_QPOS) {
        const float nq = bf.buf[MSK_BUF_ART_QPOS][row];
        if (nq != E[m->lay.q + (d)]) teleported = true;
        E[m->lay.q + (d)] = nq;
      }
      if (mask & MSK_APPLY_ART_QVEL) E[m->lay.qd + (d)] = bf.buf[MSK_BUF_ART_QVEL][row];
      if (mask & MSK_APPLY_ART_QF) E[m->lay.qf + (d)] = bf.buf[MSK_BUF_ART_QF][row];
      if (mask & MSK_APPLY_ART_TARGET_QPOS) E[m->lay.qt + (d)] = bf.buf[MSK_BUF_ART_TARGET_QPOS][row];
      if (mask & MSK_APPLY_ART_TARGET_QVEL) E[m->lay.qdt + (d)] = bf.buf[MSK_BUF_ART_TARGET_QVEL][row];
    }
  if (NL > 1) teleported = __any(teleported);
  if (teleported) { /* no warm start across a teleport: replays from a state are reproducible */
    int* cnts = st.ct_cnt + (size_t)e * m->npp;
    unsigned long long* gc = st.gjk_cache + (size_t)e * m->npp;
    for (int p = lane; p < m->np; p += NL) { cnts[p] = 0; gc[p] = 0ull; }
    if (lane == 0) st.ct_total[e] = 0;
  }
}
MSK_DEV void apply_block(const DModel* __restrict__ m, const DState& st, const DBuffers& bf, unsigned mask, const int* __restrict__ art_dof0,
                         const int* __restrict__ art_ndof, const int blk) {
  const int e = blk * 256 + threadIdx.x;
  if (e >= m->N) return;
  apply_env(m, st, bf, mask, art_dof0, art_ndof, e);
}

__global__ void __launch_bounds__(256) k_apply(const DModel* __restrict__ m, DState st, DBuffers bf, unsigned mask, const int* __restrict__ art_dof0,
                                               const int* __restrict__ art_ndof) {
  apply_block(m, st, bf, mask, art_dof0, art_ndof, blockIdx.x);
}

/* gpu_fetch_articulation_link_incoming_joint_forces: inverse dynamics of the state the last step left (see the statement in
 * oracle/orc_sim.c orc_link_joint_forces, mirrored here operation by operation).  On demand, not on the step path: one
 * lane per env, serial over the tree, per-lane arrays in scratch.  out: [N][na][max_links][6]; link_slot[body] = row of a
 * link within its articulation. */
struct LinkSlots { signed char slot[MSK_MAX_BODIES]; int max_links; };

/* out_body (optional): the same wrenches as [N][nb][6] by body id, for the joint-friction rows of the next substep (DState::jforce) */
__global__ void __launch_bounds__(64) k_link_forces(const DModel* __restrict__ m, DState st, LinkSlots ls, float* __restrict__ out, float* __restrict__ out_body) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= m->N) return;
  const float* E = EREC(st, m, e);
  const float inv_dt = 1.0f / m->cfg.timestep;
  const v3 g = v3_make(m->cfg.gravity[0], m->cfg.gravity[1], m->cfg.gravity[2]);
  pose T[MSK_MAX_BODIES], U[MSK_MAX_BODIES];
  sv6 V[MSK_MAX_BODIES], f[MSK_MAX_BODIES], acc[MSK_MAX_BODIES];
  const int nb = m->nb;
  for (int i = 0; i < nb; ++i) {
    const DBody* b = &m->bodies[i];
    T[i] = U[i] = load_pose(E, m->lay.bpose, i);
    V[i] = sv6_zero();
    f[i] = sv6_zero();
    acc[i] = sv6_zero();
    if (b->kind != MSK_BODY_LINK) continue;
    sv6 S = sv6_zero();
    if (b->parent >= 0) {   /* the oracle's kinematics(): frame (composed rotation down the tree, normalised where published), joint subspace and spatial velocity */
      pose Jq;
      Jq.p = v3_make(0, 0, 0);
      Jq.q = quat_make(1, 0, 0, 0);
      if (b->jtype == MSK_JOINT_REVOLUTE) {
        float sn, cs;
        msk_sincos(0.5f * E[m->lay.q + b->dof], &sn, &cs);
        Jq.q = quat_make(cs, sn, 0, 0);
      } else if (b->jtype == MSK_JOINT_PRISMATIC) {
        Jq.p = v3_make(E[m->lay.q + b->dof], 0, 0);
      }
      pose Lq = pose_mul(pose_mul(b->Xp, Jq), b->XcInv);
      Lq.q = quat_normalize(Lq.q);
      U[i] = pose_mul(U[b->parent], Lq);
      pose Ti;
      Ti.p = U[i].p;
      Ti.q = quat_normalize(U[i].q);
      T[i] = Ti;
      const pose Tj = pose_mul(T[b->parent], b->Xp);
      const v3 axis = quat_rotate(Tj.q, v3_make(1, 0, 0));
      if (b->jtype == MSK_JOINT_REVOLUTE) {
        S.a = axis;
        S.l = v3_cross(Tj.p, axis);
      } else if (b->jtype == MSK_JOINT_PRISMATIC) {
        S.l = axis;
      }
      V[i] = sv6_madd(V[b->parent], S, b->dof >= 0 ? E[m->lay.qd + b->dof] : 0.0f);
    }
    const m33 R = quat_to_m33(T[i].q);
    const v3 cw = v3_add(T[i].p, m33_mulv(&R, b->com));
    float Iw[6];
    sym6_rotate(&R, b->I6, Iw);
    sinertia Isp;
    const float mass = b->mass, cc = v3_dot(cw, cw);
    Isp.m = mass;
    Isp.h = v3_scale(cw, mass);
    Isp.I[0] = Iw[0] + mass * (cc - cw.x * cw.x);
    Isp.I[1] = Iw[1] + mass * (cc - cw.y * cw.y);
    Isp.I[2] = Iw[2] + mass * (cc - cw.z * cw.z);
    Isp.I[3] = Iw[3] - mass * (cw.x * cw.y);
    Isp.I[4] = Iw[4] - mass * (cw.x * cw.z);
    Isp.I[5] = Iw[5] - mass * (cw.y * cw.z);
    if (b->parent >= 0) {
      acc[i] = acc[b->parent];
      if (b->dof >= 0) {
        const float qd = E[m->lay.qd + b->dof];
        const sv6 sq = {v3_scale(S.a, qd), v3_scale(S.l, qd)};
        acc[i] = sv6_add(acc[i], sv6_crossm(V[b->parent], sq));
        acc[i] = sv6_madd(acc[i], S, E[m->lay.qacc + b->dof]);
      }
    }
    const sv6 Iv = sinertia_mul(&Isp, V[i]);
    f[i] = sv6_add(sinertia_mul(&Isp, acc[i]), sv6_crossf(V[i], Iv));
    if (!b->nograv) {
      const v3 mg = v3_scale(g, mass);
      f[i].a = v3_sub(f[i].a, v3_cross(cw, mg));
      f[i].l = v3_sub(f[i].l, mg);
    }
  }
  /* contact wrenches about the env origin: +F on body A, -F on body B (impulses of the last step / dt) */
  const int* cnts = st.ct_cnt + (size_t)e * m->npp;
  const float* recs = st.ct_rec + (size_t)e * m->npp * MSK_CT_REC;
  for (int p = 0; p < m->np; ++p) {
    const int cnt = cnts[p];
    if (cnt == 0) continue;
    const int ba = m->pinfo[p].ba, bb = m->pinfo[p].bb;
    const bool la = ba >= 0 && m->bodies[ba].kind == MSK_BODY_LINK, lb = bb >= 0 && m->bodies[bb].kind == MSK_BODY_LINK;
    if (!la && !lb) continue;
    const float* rec = recs + (size_t)p * MSK_CT_REC;
    const v3 n = v3_make(rec[0], rec[1], rec[2]);
    v3 t1, t2;
    msk_tangents(n, &t1, &t2);
    const float lam_t = (cnt == 1 && ct_blocks(m, p, 1) == 2) ? rec[3] : 0.0f;   /* the torsional row of a one-point manifold */
    for (int k = 0; k < cnt; ++k) {
      const v3 pos = v3_make(rec[4 + k * 3 + 0], rec[4 + k * 3 + 1], rec[4 + k * 3 + 2]);
      v3 F = v3_scale(n, rec[20 + k * 3 + 0]);
      F = v3_madd(F, t1, rec[20 + k * 3 + 1]);
      F = v3_madd(F, t2, rec[20 + k * 3 + 2]);
      F = v3_scale(F, inv_dt);
      const v3 Tq = v3_madd(v3_cross(pos, F), n, lam_t * inv_dt);   /* ... plus the couple of the torsional row */
      if (la) { f[ba].a = v3_sub(f[ba].a, Tq); f[ba].l = v3_sub(f[ba].l, F); }
      if (lb) { f[bb].a = v3_add(f[bb].a, Tq); f[bb].l = v3_add(f[bb].l, F); }
    }
  }
  for (int i = nb - 1; i >= 0; --i) {
    const DBody* b = &m->bodies[i];
    if (b->kind == MSK_BODY_LINK && b->parent >= 0) f[b->parent] = sv6_add(f[b->parent], f[i]);
  }
  for (int i = 0; i < nb; ++i) {
    const DBody* b = &m->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    const pose C = (b->parent >= 0) ? pose_mul(T[i], pose_inv(b->XcInv)) : T[i];
    const v3 torque = v3_sub(f[i].a, v3_cross(C.p, f[i].l));   /* moved from the env origin to the frame's origin */
    const v3 fl = quat_rotate(quat_conj(C.q), f[i].l), tl = quat_rotate(quat_conj(C.q), torque);
    if (out) {
      float* o = out + (((size_t)e * m->na + b->art) * ls.max_links + ls.slot[i]) * 6;
      o[0] = fl.x; o[1] = fl.y; o[2] = fl.z; o[3] = tl.x; o[4] = tl.y; o[5] = tl.z;
    }
    if (out_body) {
      float* o = out_body + ((size_t)e * m->nb + i) * 6;
      o[0] = fl.x; o[1] = fl.y; o[2] = fl.z; o[3] = tl.x; o[4] = tl.y; o[5] = tl.z;
    }
  }
}

template <int NL = 1>
MSK_DEV void fetch_env(const DModel* __restrict__ m, const DState& st, const DBuffers& bf, unsigned mask, const int* __restrict__ art_dof0,
                       const int* __restrict__ art_ndof, const int e, const int lane = 0) {
  float* E = EREC(st, m, e);
  const float ox = E[m->lay.off + 0], oy = E[m->lay.off + 1], oz = E[m->lay.off + 2];
  if (mask & MSK_FETCH_RIGID_DATA)
    for (int i = lane; i < m->nb; i += NL) {
      float* r = bf.buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * m->nb + i) * 13;
      pose T = load_pose(E, m->lay.bpose, i);
      v3 lv = load_v3(E, m->lay.blin, i), av = load_v3(E, m->lay.bang, i);
      r[0] = T.p.x + ox; r[1] = T.p.y + oy; r[2] = T.p.z + oz;
      r[3] = T.q.w; r[4] = T.q.x; r[5] = T.q.y; r[6] = T.q.z;
      r[7] = lv.x; r[8] = lv.y; r[9] = lv.z; r[10] = av.x; r[11] = av.y; r[12] = av.z;
    }
  for (int a = 0; a < m->na; ++a)
    for (int j = lane; j < art_ndof[a]; j += NL) {
      const int d = art_dof0[a] + j;
      const size_t row = ((size_t)e * m->na + a) * bf.pitch + j;
      if (mask & MSK_FETCH_ART_QPOS) bf.buf[MSK_BUF_ART_QPOS][row] = E[m->lay.q + (d)];
      if (mask & MSK_FETCH_ART_QVEL) bf.buf[MSK_BUF_ART_QVEL][row] = E[m->lay.qd + (d)];
      if (mask & MSK_FETCH_ART_QACC) bf.buf[MSK_BUF_ART_QACC][row] = E[m->lay.qacc + (d)];
      if (mask & MSK_FETCH_ART_TARGETS) {
        bf.buf[MSK_BUF_ART_TARGET_QPOS][row] = E[m->lay.qt + (d)];
        bf.buf[MSK_BUF_ART_TARGET_QVEL][row] = E[m->lay.qdt + (d)];
      }
    }
}
MSK_DEV void fetch_block(const DModel* __restrict__ m, const DState& st, const DBuffers& bf, unsigned mask, const int* __restrict__ art_dof0,
                         const int* __restrict__ art_ndof, const int blk) {
  const int e = blk * 256 + threadIdx.x;
  if (e >= m->N) return;
  fetch_env(m, st, bf, mask, art_dof0, art_ndof, e);
}

/* msk_reset_masked (include/msk_physx.h): the partial reset of the envs a device-side mask names, without the host.  For such an env: the sapien buffers are
 * brought up to date (fetch of this env: entries the episode image does not name keep their current values), the image's entries are written over them -- the
 * rows a host-side reset would have written through the torch views (BaseEnv.reset -> _clear_sim_state, _initialize_episode, controller.reset:
 * envs/sapien_env.py:857-978,1023-1036) --, and the env is applied exactly as scene._gpu_apply_all() applies it (rows that did not change are left alone, a teleport
 * drops the env's contact cache).  image: [N][slots][nent] floats, slot = episode[e] % slots; entry t of an image goes to word ent[t] of the env's buffer image
 * (MSK_RESET_*: a buffer id in the high bits, the word inside the env's rows of that buffer in the low ones). */
struct ResetPlan { const float* image; const int* ent; int nent, slots; const unsigned char* mask; int* episode; int* elapsed; };
/* one wavefront per env (grid = N): a wavefront whose env is not named leaves at once; the others share their env's bodies, joints, image entries and candidate
 * pairs among the 64 lanes (lane = env, one thread walking its env's ~500 words alone, made a reset of 82 of 4096 envs a 60 us chain of dependent memory
 * operations).  The phases hand over through global memory inside the wavefront: a barrier (one wavefront: the wait for its own stores) between them. */
__global__ void __launch_bounds__(64) k_reset_masked(const DModel* __restrict__ m, DState st, DBuffers bf, ResetPlan rp, unsigned fetch_mask, unsigned apply_mask,
                                                     const int* __restrict__ art_dof0, const int* __restrict__ art_ndof) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= m->N || !rp.mask[e]) return;
  fetch_env<64>(m, st, bf, fetch_mask, art_dof0, art_ndof, e, lane);
  __syncthreads();
  const int ep = rp.episode[e];
  const float* img = rp.image + ((size_t)e * rp.slots + (size_t)(ep % rp.slots)) * rp.nent;
  for (int t = lane; t < rp.nent; t += 64) {
    const int code = rp.ent[t], which = code >> 24, word = code & 0xFFFFFF;
    float* base;
    if (which == 0) base = bf.buf[MSK_BUF_RIGID_BODY_DATA] + (size_t)e * m->nb * 13;
    else {
      const int id = which == 1 ? MSK_BUF_ART_QPOS : (which == 2 ? MSK_BUF_ART_QVEL : (which == 3 ? MSK_BUF_ART_TARGET_QPOS : MSK_BUF_ART_TARGET_QVEL));
      base = bf.buf[id] + (size_t)e * m->na * bf.pitch;
    }
    base[word] = img[t];
  }
  __syncthreads();
  apply_env<64>(m, st, bf, apply_mask, art_dof0, art_ndof, e, lane);
  __syncthreads();   /* (every lane has read the counter) */
  if (lane == 0) {
    rp.episode[e] = ep + 1;
    if (rp.elapsed) rp.elapsed[e] = 0;
  }
}

/* msk_episode_book_step (include/msk_physx.h): one workgroup walks all sub-scenes (a dozen bytes per env: 4096 envs are four rounds of its 1024 lanes), so that
 * `dones.any()` needs no second launch and no atomics: the lanes' verdicts meet in LDS and lane 0 stores the flag */
__global__ void __launch_bounds__(1024) k_episode_book(const int n, const msk_episode_book b) {
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  bool mine = false;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const bool term = !b.ignore_terminations && b.terminated[(size_t)e * b.terminated_stride] != 0;
    const bool done = term || b.truncated[(size_t)e * b.truncated_stride] != 0;
    if (b.record_metrics) {
      const float ret = b.returns[e] + b.reward[e];
      const int len = b.elapsed[e];
      b.out_return[e] = ret;
      b.out_episode_len[e] = len;
      b.out_reward[e] = ret / (float)len;
      b.returns[e] = (done && b.clear_done) ? 0.0f : ret;
      if (b.success) {
        const bool now = b.success[(size_t)e * b.success_stride] != 0, once = b.success_once[e] != 0 || now;
        b.out_success_once[e] = once ? 1 : 0;
        if (b.ignore_terminations) b.out_success_at_end[e] = now ? 1 : 0;
        b.success_once[e] = (once && !(done && b.clear_done)) ? 1 : 0;
      }
      if (b.fail) {
        const bool now = b.fail[(size_t)e * b.fail_stride] != 0, once = b.fail_once[e] != 0 || now;
        b.out_fail_once[e] = once ? 1 : 0;
        if (b.ignore_terminations) b.out_fail_at_end[e] = now ? 1 : 0;
        b.fail_once[e] = (once && !(done && b.clear_done)) ? 1 : 0;
      }
    }
    b.out_terminated[e] = term ? 1 : 0;
    b.out_done[e] = done ? 1 : 0;
    mine = mine || done;
  }
  if (mine) s_any = 1;      /* (every writer stores the same value) */
  __syncthreads();
  if (threadIdx.x == 0) b.any_done[0] = s_any;
}

__global__ void __launch_bounds__(256) k_fetch(const DModel* __restrict__ m, DState st, DBuffers bf, unsigned mask, const int* __restrict__ art_dof0,
                                               const int* __restrict__ art_ndof) {
  fetch_block(m, st, bf, mask, art_dof0, art_ndof, blockIdx.x);
}

/* ---- one launch over several contexts (msk_batch) -------------------------------------------------------------------------------
 * A scene whose sub-scenes differ in structure runs as one context per structural group (a few dozen envs each): per group the
 * kernels above are latency chains on a handful of workgroups, and 25 groups x 15 launches per control step, spread over streams,
 * still run four at a time.  Here the groups of one template variant share each launch: a workgroup finds its group from the
 * table (first workgroup of every group in this launch), then runs the same block function on that group's model and state. */
struct GroupRef {
  const DModel* m;
  DState st;
  DBuffers bufs;
  const int* art_dof0;
  const int* art_ndof;
  int b_dyn, b_np, b_cs, b_af;   /* first workgroup of this group in the dynamics (= kinematics) / narrowphase / solver / apply-fetch launches */
  int np_group, np_gx, np_gy, gm;
  NpCfg np_cfg;
};
enum { MG_DYN = 0, MG_NP = 1, MG_CS = 2, MG_AF = 3 };
template <int WHICH>
MSK_DEV int multi_find(const GroupRef* __restrict__ refs, const int n, const int blk, int* local) {
  int g = 0;
  for (int k = 1; k < n; ++k) {
    const int b = WHICH == MG_DYN ? refs[k].b_dyn : (WHICH == MG_NP ? refs[k].b_np : (WHICH == MG_CS ? refs[k].b_cs : refs[k].b_af));
    if (blk >= b) g = k;
  }
  const int b0 = WHICH == MG_DYN ? refs[g].b_dyn : (WHICH == MG_NP ? refs[g].b_np : (WHICH == MG_CS ? refs[g].b_cs : refs[g].b_af));
  *local = blk - b0;
  return g;
}
template <int LPE, int MD>
__global__ void __launch_bounds__(128) k_multi_dynamics(const GroupRef* __restrict__ refs, const int n) {
  extern __shared__ __attribute__((aligned(16))) float lds_md[];
  int blk;
  const GroupRef& r = refs[multi_find<MG_DYN>(refs, n, blockIdx.x, &blk)];
  if (blk * (64 / LPE) >= r.m->N) return;   /* (the whole workgroup) */
  if (threadIdx.x < 64) {
    dynamics_block<LPE, MD>(r.m, r.st, lds_md, blk);
    if (blockDim.x == 64 && r.m->np > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const BpConst bk = bp_const(r.m);
      broadphase_block<LPE>(r.m, r.st, blk, bk);
    }
  } else {
    const BpConst bk = bp_const(r.m);
    __syncthreads();
    if (r.m->np > 0) broadphase_block<LPE>(r.m, r.st, blk, bk);
  }
}
template <int LPE>
__global__ void __launch_bounds__(64) k_multi_kinematics(const GroupRef* __restrict__ refs, const int n) {
  extern __shared__ __attribute__((aligned(16))) float lds_mk[];
  int blk;
  const GroupRef& r = refs[multi_find<MG_DYN>(refs, n, blockIdx.x, &blk)];
  if (blk * (64 / LPE) >= r.m->N) return;
  kinematics_block<LPE>(r.m, r.st, lds_mk, blk);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_multi_narrowphase(const GroupRef* __restrict__ refs, const int n) {
  int blk;
  const GroupRef& r = refs[multi_find<MG_NP>(refs, n, blockIdx.x, &blk)];
  if (blk >= r.np_gx * r.np_gy) return;
  narrowphase_block(r.m, r.st, r.np_group, r.np_cfg, blk % r.np_gx, blk / r.np_gx, r.np_gx, r.np_gy);
}
template <int NVP, int GL>
__global__ void __launch_bounds__(64) k_multi_csolve(const GroupRef* __restrict__ refs, const int n) {
  extern __shared__ __attribute__((aligned(16))) float lds_mc[];
  int blk;
  const GroupRef& r = refs[multi_find<MG_CS>(refs, n, blockIdx.x, &blk)];
  csolve_block<NVP, GL>(r.m, r.st, r.gm, blk, lds_mc);
}
template <int NVP>
__global__ void __launch_bounds__(64) k_multi_csolve_wide(const GroupRef* __restrict__ refs, const int n, const int workers) { /* grid: n x workers */
  extern __shared__ __attribute__((aligned(16))) float lds_mw[];
  const GroupRef& r = refs[blockIdx.x / workers];
  csolve_wide_block<NVP>(r.m, r.st, blockIdx.x % workers, lds_mw);
}
__global__ void __launch_bounds__(256) k_multi_apply(const GroupRef* __restrict__ refs, const int n, const unsigned mask) {
  int blk;
  const GroupRef& r = refs[multi_find<MG_AF>(refs, n, blockIdx.x, &blk)];
  apply_block(r.m, r.st, r.bufs, mask, r.art_dof0, r.art_ndof, blk);
}
__global__ void __launch_bounds__(256) k_multi_fetch(const GroupRef* __restrict__ refs, const int n, const unsigned mask) {
  int blk;
  const GroupRef& r = refs[multi_find<MG_AF>(refs, n, blockIdx.x, &blk)];
  fetch_block(r.m, r.st, r.bufs, mask, r.art_dof0, r.art_ndof, blk);
}

/* sum of contact impulses applied on body x by body y, per env, per queried pair */
__global__ void __launch_bounds__(256) k_query(const DModel* __restrict__ m, DState st, const int* __restrict__ qpairs, int nq, float* __restrict__ out) {
  const int N = m->N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  float* E = EREC(st, m, e);
  const int* cnts = st.ct_cnt + (size_t)e * m->npp;
  const float* recs = st.ct_rec + (size_t)e * m->npp * MSK_CT_REC;
  for (int qi = 0; qi < nq; ++qi) {
    const int x = qpairs[2 * qi], y = qpairs[2 * qi + 1];
    v3 sum = v3_make(0, 0, 0);
    for (int p = 0; p < m->np; ++p) {
      const int ba = m->pinfo[p].ba, bb = m->pinfo[p].bb;
      float sgn;
      if (ba == x && (bb == y || y == MSK_ANY_BODY)) sgn = 1.0f;
      else if (bb == x && (ba == y || y == MSK_ANY_BODY)) sgn = -1.0f;
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
