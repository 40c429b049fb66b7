/*
 * msk_render.h — batched depth + segmentation rasteriser (include/msk_render.h), gfx950.
 *
 *   k_render_env     one 256-thread workgroup per env: camera-from-shape transforms (body poses come straight from
 *                    the simulator's env record), then per triangle: camera-frame corners, near-plane clip,
 *                    projection, back-face cull, edge + 1/depth plane equations, pixel bounding box -> 64-byte
 *                    TriSetup records in LDS, binned into 16x4-pixel tiles (LDS counters, scan, lists of record
 *                    numbers); then the workgroup's wavefronts walk the tiles, lane = pixel: per record 3 edge
 *                    evaluations + 1/depth compare; the winner's camera-space position (mm, int16) and
 *                    segmentation id leave as one 8-byte store per pixel (a tile row = one 128-byte line).
 *   k_render_splat   the same workgroup-per-env pipeline with the small triangles (pixel bounding box <= 16 x 16: four out of five
 *                    records of the benchmarked scenes, median box 8 pixels) taken off the tile lists: four lanes per record walk the
 *                    record's own box, one row each, and put (1/depth, primitive, record) as a 64-bit ds_max into the wavefront's
 *                    own key buffer of one segment (4 tiles of a tile row) in LDS -- no workgroup barrier after the lists are built;
 *                    the tile walk starts from the keys, tests the large and medium triangles lane = pixel as before, and reads
 *                    segmentation id and colour from the winning record.  A 16 x 4 tile costs ~45 instructions per record whatever
 *                    the record's size; a splatted record costs its own box.
 *
 * The output (N x H x W x 8 bytes = 512 MiB at 4096 envs, 128x128, + 2 x 2 bytes for the depth and
 * segmentation planes) is the algorithmic traffic of this path.  Arithmetic follows oracle/orc_render.c
 * statement by statement (the images are compared bit for bit).
 */
#ifndef MSK_RENDER_KERNELS_H
#define MSK_RENDER_KERNELS_H

#include "../../include/msk_render.h"
#include "msk_model.h"
#include <type_traits>

#define MSK_TW 16                   /* tile = 16 x 4 pixels, lane = (x = lane & 15, y = lane >> 4) */
#define MSK_TH 4
#define MSK_RENDER_THREADS 256
#define MSK_MAX_TILES 4096         /* 16 x 4 pixel tiles per picture: up to 512 x 512 (the human-render cameras, sapien_env.py _default_human_render_camera_configs);
                                    * the workgroup's LDS is sized by the camera's own tile count (RCamera::tile_cap) */
#define MSK_BIG_TILES 16           /* a triangle over more tiles than this goes to the env's list of large triangles: every tile tests its mask bit */
#define MSK_MAX_BIG 16
#define MSK_SEG_BIG 0x40000000      /* flag in TriSetup::seg: the record is in the env's list of large triangles */
#define MSK_SETUP_WORDS 16
#define MSK_RSHAPE_WORDS 12          /* LDS image of a render shape: camera-from-shape pose (7), pad, per-env scale (3), pad */

struct RShape { int body, seg; pose local; float color[4]; int xs; /* per-env box instance it follows (slot in the env record), -1: none */
                int tex; /* texture id (msk_render_set_texture) or -1 */ int v0; /* first vertex of the shape */ };
struct RTexture { int w, h, ofs; };   /* texels [ofs, ofs + w * h) of RModel::texels, row 0 first */
#define MSK_MAX_LIGHTS 4
struct RTri { int v0, v1, v2, shape; };
struct RModel {
  int nv, nt, ns;
  RShape shapes[MSK_MAX_RENDER_SHAPES];
  v3 verts[MSK_MAX_RENDER_VERTS];
  unsigned char vshape[MSK_MAX_RENDER_VERTS];
  RTri tris[MSK_MAX_RENDER_TRIS];
  /* Color: flat Lambert shading by the scene's ambient + directional lights (directions in the env frame, normalised) */
  float ambient[3];
  int nlights;
  float ldir[MSK_MAX_LIGHTS][3], lcol[MSK_MAX_LIGHTS][3];
  /* point and spot lights (msk_render_set_local_lights): position and unit axis in the env frame, colour, cosines of the half cone angles */
  int nlocal;
  float ppos[MSK_MAX_LOCAL_LIGHTS][3], pdir[MSK_MAX_LOCAL_LIGHTS][3], pcol[MSK_MAX_LOCAL_LIGHTS][3], pcone[MSK_MAX_LOCAL_LIGHTS][2];
  /* base-colour textures (msk_render_set_texture): per-vertex texture coordinates, the textures, the texels (device memory, r in the low byte) */
  int ntex, ntexels;
  RTexture tex[MSK_MAX_TEXTURES];
  const unsigned* texels;
  float vuv[MSK_MAX_RENDER_VERTS][2];
};
#define MSK_LIGHT_WORDS (MSK_MAX_LIGHTS * 3 + MSK_MAX_LOCAL_LIGHTS * 6)   /* LDS: directions, then positions, then axes, all in the camera frame */
struct RCamera {
  int W, H, mount, tiles_x, tiles_y;
  int tile_cap;                    /* tiles_x * tiles_y */
  float fx, fy, cx, cy, near_, far_;
  pose local;
  int ns;                          /* render shapes of the model (LDS carve) */
  int rcap;                        /* screen-triangle records the workgroup keeps in LDS */
  int icap;                        /* entries of the tiles' record-number lists (LDS) */
  int spill_cap;                   /* records beyond rcap, per env, in global memory: rcap + spill_cap = 2 x triangles, every record has a place */
  float* setups;                   /* [N][spill_cap][16] */
  short* out;                      /* [N][H][W][4]        */
  unsigned* color;                 /* [N][H][W]: Color r8g8b8a8unorm (r in the low byte), 0 = background; null until asked for */
  short* depth;                    /* [N][H][W]: -z of out (Camera.get_obs's depth), written by the same store */
  short* seg;                      /* [N][H][W]: w of out                                                      */
  int* overflow;                   /* [1] a tile list ran over icap (the picture may miss triangles)          */
  int dbg_cut;                     /* MSK_PROFILE_PHASES builds (tools/gpu_render_probe.py): the workgroup returns after phase dbg_cut */
  int mode;                        /* 0: k_render_env (every record through the tile lists), 1: k_render_splat */
  int uvcap;                       /* k_render_splat: screen triangles of textured shapes whose u/depth, v/depth planes the workgroup keeps (LDS); 0: no textures */
  unsigned* uvt;                   /* [N][H][W]: 1 + index of the texel under the pixel (0: none), written next to Color, resolved by k_render_texture */
  int bcap;                        /* k_render_splat: entries of the tile rows' lists of small records (LDS) */
  int want_tex;                    /* 1: `out` is written; 0: only the planes (msk_camera_set_outputs): camera-space x, y are not computed, 8 of 12 bytes per pixel not stored */
};
#define MSK_SEG_SMALL 0x20000000    /* flag in TriSetup::seg (k_render_splat): pixel box <= 16 x 16, bb = x0 | y0 << 10 | (x1 - x0) << 20 | (y1 - y0) << 24 */
#define MSK_SPLAT_MAX 16
/* LDS words of k_render_splat */
#define MSK_SEG_TILES 4             /* a segment = 4 tiles of one tile row (64 x 4 pixels): the unit a wavefront splats and then walks */
#define MSK_SEG_PX (MSK_SEG_TILES * MSK_TW * MSK_TH)
static inline __host__ __device__ int render_segments(int tiles_x, int tiles_y) { return tiles_y * ((tiles_x + MSK_SEG_TILES - 1) / MSK_SEG_TILES); }
static inline __host__ __device__ size_t render_splat_lds_words(int ns, int rcap, int icap, int ntiles, int nseg, int bcap, int uvcap = 0) {
  return (size_t)rcap * 16 + (size_t)uvcap * 8 + (size_t)(MSK_RENDER_THREADS / 64) * 2 * MSK_SEG_PX + (size_t)ns * 12 + MSK_LIGHT_WORDS + (size_t)(ntiles + 1) + (size_t)ntiles + 16 + 16 +
         2 * ((size_t)((ntiles + 1) & ~1) / 2) + (size_t)(icap + 1) / 2 + (size_t)(nseg + 1) + (size_t)nseg + (size_t)(bcap + 1) / 2 + 4;
}
/* LDS words of k_render_env (the carve at its top) */
static inline __host__ __device__ size_t render_lds_words(int ns, int rcap, int icap, int ntiles) {
  return (size_t)rcap * 16 + (size_t)ns * 12 + MSK_LIGHT_WORDS + (size_t)(ntiles + 1) + (size_t)ntiles + 16 + 8 + 2 * ((size_t)((ntiles + 1) & ~1) / 2) + (size_t)(icap + 1) / 2 + 4;
}

/* One screen triangle: A,B,C of the three edge functions (inside = all >= 0), the 1/depth plane,
 * segmentation id, primitive id (tie break), pixel bounding box. */
struct TriSetup {   /* (word order = what the tile kernel loads: the first two edges' coefficients side by side for packed FMAs) */
  float A0, A1, B0, B1, C0, C1, A2, B2, C2, Aw, Bw, Cw;
  int seg, prim, bb;               /* bb = tx0 | tx1 << 8 | ty0 << 16 | ty1 << 24: the TILES (16 x 4 pixels) its pixel bounding box reaches */
  unsigned color;                  /* shaded r8g8b8a8 of the (flat) triangle */
};
#define BB_X0(bb) ((bb) & 0xFF)
#define BB_X1(bb) (((bb) >> 8) & 0xFF)
#define BB_Y0(bb) (((bb) >> 16) & 0xFF)
#define BB_Y1(bb) (((unsigned)(bb)) >> 24)

/* flat shading of one triangle (camera-frame corners p0 p1 p2, counter-clockwise seen from outside): per channel
 * base * min(1, ambient + sum_l light_l * max(0, n . -dir_l) + sum_p light_p * cone_p * max(0, n . l_p) / |x_p - centre|^2), rounded to 8 bits;
 * alpha = 255 (oracle/orc_render.c shade_triangle, statement by statement) */
MSK_DEV unsigned shade_triangle(v3 p0, v3 p1, v3 p2, const float* base, const float* ambient, int nl, const float* ldir_cam, const float* lcol,
                                int np, const float* ppos_cam, const float* pdir_cam, const float* pcol, const float* pcone) {
  v3 n = v3_cross(v3_sub(p1, p0), v3_sub(p2, p0));
  const float l = v3_len(n);
  n = (l > 0.0f) ? v3_scale(n, 1.0f / l) : v3_make(0, 0, 0);
  float lit[3] = {ambient[0], ambient[1], ambient[2]};
  for (int k = 0; k < nl; ++k) {
    const float d = fmaxf(0.0f, -(n.x * ldir_cam[k * 3] + n.y * ldir_cam[k * 3 + 1] + n.z * ldir_cam[k * 3 + 2]));
    lit[0] = fmaf(lcol[k * 3], d, lit[0]); lit[1] = fmaf(lcol[k * 3 + 1], d, lit[1]); lit[2] = fmaf(lcol[k * 3 + 2], d, lit[2]);
  }
  if (np > 0) {
    const float third = 1.0f / 3.0f;
    const v3 cen = v3_make((p0.x + p1.x + p2.x) * third, (p0.y + p1.y + p2.y) * third, (p0.z + p1.z + p2.z) * third);
    for (int k = 0; k < np; ++k) {
      const v3 L = v3_make(ppos_cam[k * 3] - cen.x, ppos_cam[k * 3 + 1] - cen.y, ppos_cam[k * 3 + 2] - cen.z);
      const float d2 = fmaf(L.x, L.x, fmaf(L.y, L.y, L.z * L.z));
      if (!(d2 > 1e-12f)) continue;
      const float inv = 1.0f / sqrtf(d2);
      float a = fmaxf(0.0f, fmaf(n.x, L.x, fmaf(n.y, L.y, n.z * L.z)) * inv) / d2;
      if (pcone[k * 2] > -1.5f) { /* spot: the light looks along its axis */
        const float cs = -fmaf(pdir_cam[k * 3], L.x, fmaf(pdir_cam[k * 3 + 1], L.y, pdir_cam[k * 3 + 2] * L.z)) * inv;
        const float span = pcone[k * 2] - pcone[k * 2 + 1];
        const float f = span > 1e-6f ? fminf(fmaxf((cs - pcone[k * 2 + 1]) / span, 0.0f), 1.0f) : (cs >= pcone[k * 2] ? 1.0f : 0.0f);
        a = a * f;
      }
      lit[0] = fmaf(pcol[k * 3], a, lit[0]); lit[1] = fmaf(pcol[k * 3 + 1], a, lit[1]); lit[2] = fmaf(pcol[k * 3 + 2], a, lit[2]);
    }
  }
  unsigned out = 0xFF000000u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float v = fminf(fmaxf(base[ch] * fminf(lit[ch], 1.0f), 0.0f), 1.0f);
    out |= ((unsigned)rintf(v * 255.0f) & 0xFFu) << (8 * ch);
  }
  return out;
}

/* projects a camera-frame point (x forward, y left, z up): pixel coordinates and 1/depth */
MSK_DEV void project_point(const RCamera& cam, v3 p, float* u, float* v, float* w) {
  const float iw = 1.0f / p.x;
  *u = fmaf(cam.fx, -p.y * iw, cam.cx);
  *v = fmaf(-cam.fy, p.z * iw, cam.cy);
  *w = iw;
}

/* The decisions of setup_triangle's cull, and nothing else: 1 if the screen triangle (p0, p1, p2), all in front of the near plane, would get a record.  The
 * same expressions in the same order as there (projection, signed area, pixel box), so it never disagrees with it; k_render_splat's first pass asks it for every
 * triangle of the template and runs the full set-up -- which repeats these tests -- only for the ~30 % that pass. */
MSK_DEV int triangle_survives_cull(const RCamera& cam, v3 p0, v3 p1, v3 p2) {
  float u0, v0, w0, u1, v1, w1, u2, v2, w2;
  project_point(cam, p0, &u0, &v0, &w0);
  project_point(cam, p1, &u1, &v1, &w1);
  project_point(cam, p2, &u2, &v2, &w2);
  const float area = fmaf(u1 - u0, v2 - v0, -((v1 - v0) * (u2 - u0)));
  if (!(area < -1e-12f)) return 0;
  const float umin = fminf(u0, fminf(u2, u1)), umax = fmaxf(u0, fmaxf(u2, u1));
  const float vmin = fminf(v0, fminf(v2, v1)), vmax = fmaxf(v0, fmaxf(v2, v1));
  int x0 = (int)ceilf(umin - 0.5f), x1 = (int)floorf(umax - 0.5f);
  int y0 = (int)ceilf(vmin - 0.5f), y1 = (int)floorf(vmax - 0.5f);
  if (!(umin < 1e9f && umax > -1e9f && vmin < 1e9f && vmax > -1e9f)) return 0;
  x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, cam.W - 1); y1 = min(y1, cam.H - 1);
  return !(x0 > x1 || y0 > y1);
}

/* sets up the screen triangle (p0, p1, p2), all in front of the near plane; returns 0 if it is culled */
/* uv (optional): texture coordinates of the corners (u0 v0 u1 v1 u2 v2); uvp gets the planes of u / depth and v / depth (Au Bu Cu Av Bv Cv) */
MSK_DEV int setup_triangle(const RCamera& cam, v3 p0, v3 p1, v3 p2, int seg, int prim, TriSetup* t, int* box = nullptr, const float* uv = nullptr,
                           float* uvp = nullptr) {
  float u0, v0, w0, u1, v1, w1, u2, v2, w2;
  project_point(cam, p0, &u0, &v0, &w0);
  project_point(cam, p1, &u1, &v1, &w1);
  project_point(cam, p2, &u2, &v2, &w2);
  /* image rows grow downwards: a triangle that is counter-clockwise seen from outside has negative area here */
  const float area = fmaf(u1 - u0, v2 - v0, -((v1 - v0) * (u2 - u0)));
  if (!(area < -1e-12f)) return 0;
  /* swap 1 <-> 2: positive orientation, interior = all edge functions >= 0 */
  float t_; t_ = u1; u1 = u2; u2 = t_; t_ = v1; v1 = v2; v2 = t_; t_ = w1; w1 = w2; w2 = t_;
  const float a2 = -area;
  const float umin = fminf(u0, fminf(u1, u2)), umax = fmaxf(u0, fmaxf(u1, u2));
  const float vmin = fminf(v0, fminf(v1, v2)), vmax = fmaxf(v0, fmaxf(v1, v2));
  /* pixel centres at +0.5 */
  int x0 = (int)ceilf(umin - 0.5f), x1 = (int)floorf(umax - 0.5f);
  int y0 = (int)ceilf(vmin - 0.5f), y1 = (int)floorf(vmax - 0.5f);
  if (!(umin < 1e9f && umax > -1e9f && vmin < 1e9f && vmax > -1e9f)) return 0;
  x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, cam.W - 1); y1 = min(y1, cam.H - 1);
  if (x0 > x1 || y0 > y1) return 0;
  /* edge a->b: E(x, y) = (b.u - a.u)(y - a.v) - (b.v - a.v)(x - a.u) = A x + B y + C */
  t->A0 = -(v1 - v0); t->B0 = u1 - u0; t->C0 = -fmaf(t->A0, u0, t->B0 * v0);
  t->A1 = -(v2 - v1); t->B1 = u2 - u1; t->C1 = -fmaf(t->A1, u1, t->B1 * v1);
  t->A2 = -(v0 - v2); t->B2 = u0 - u2; t->C2 = -fmaf(t->A2, u2, t->B2 * v2);
  /* 1/depth is affine on the screen: w = (E12 w0 + E20 w1 + E01 w2) / area */
  const float ia = 1.0f / a2;
  t->Aw = fmaf(t->A1, w0, fmaf(t->A2, w1, t->A0 * w2)) * ia;
  t->Bw = fmaf(t->B1, w0, fmaf(t->B2, w1, t->B0 * w2)) * ia;
  t->Cw = fmaf(t->C1, w0, fmaf(t->C2, w1, t->C0 * w2)) * ia;
  t->seg = seg; t->prim = prim;
  t->bb = (x0 / MSK_TW) | ((x1 / MSK_TW) << 8) | ((y0 / MSK_TH) << 16) | ((y1 / MSK_TH) << 24);
  t->color = 0u;
  if (box) { box[0] = x0; box[1] = x1; box[2] = y0; box[3] = y1; }
  if (uv) { /* (corners 1 and 2 were exchanged above) */
    const float a0 = uv[0] * w0, a1 = uv[4] * w1, a2 = uv[2] * w2, b0 = uv[1] * w0, b1 = uv[5] * w1, b2 = uv[3] * w2;
    uvp[0] = fmaf(t->A1, a0, fmaf(t->A2, a1, t->A0 * a2)) * ia;
    uvp[1] = fmaf(t->B1, a0, fmaf(t->B2, a1, t->B0 * a2)) * ia;
    uvp[2] = fmaf(t->C1, a0, fmaf(t->C2, a1, t->C0 * a2)) * ia;
    uvp[3] = fmaf(t->A1, b0, fmaf(t->A2, b1, t->A0 * b2)) * ia;
    uvp[4] = fmaf(t->B1, b0, fmaf(t->B2, b1, t->B0 * b2)) * ia;
    uvp[5] = fmaf(t->C1, b0, fmaf(t->C2, b1, t->C0 * b2)) * ia;
  }
  return 1;
}

MSK_DEV v3 lerp_near(v3 a, v3 b, float near_) { /* point of segment a-b on the plane x = near */
  const float s = (near_ - a.x) / (b.x - a.x);
  return v3_make(near_, fmaf(s, b.y - a.y, a.y), fmaf(s, b.z - a.z, a.z));
}
MSK_DEV void lerp_near_uv(v3 a, v3 b, float near_, const float* ua, const float* ub, float* out) { /* the same point's texture coordinates */
  const float s = (near_ - a.x) / (b.x - a.x);
  out[0] = fmaf(s, ub[0] - ua[0], ua[0]); out[1] = fmaf(s, ub[1] - ua[1], ua[1]);
}
/* the texel of a w x h texture under (u, v) whose one-pixel footprint is rho level-0 texels: mip level floor(log2 rho), nearest, repeating;
 * (0, 0) is the top-left corner of texel (0, 0); -> index into the texture's mip chain (oracle/orc_render.c texel_index) */
MSK_DEV int texel_index(int w, int h, float u, float v, float rho) {
  int level = 0;
  if (rho >= 2.0f && rho < 3.0e38f) level = (int)((__float_as_uint(rho) >> 23) & 0xFFu) - 127;
  else if (!(rho < 2.0f)) level = 30;
  int ofs = 0;
  for (int l = 0; l < level && !(w == 1 && h == 1); ++l) { ofs += w * h; w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1; }
  const float fu = u * (float)w, fv = v * (float)h;
  int iu = (fabsf(fu) < 1.0e9f) ? (int)floorf(fu) : 0, iv = (fabsf(fv) < 1.0e9f) ? (int)floorf(fv) : 0;
  iu %= w; if (iu < 0) iu += w;
  iv %= h; if (iv < 0) iv += h;
  return ofs + iv * w + iu;
}
#define MSK_SEG_UV_SHIFT 16          /* TriSetup::seg bits 16..28: 1 + the record's slot in the workgroup's table of texture planes (0: untextured) */
#define MSK_SEG_UV_MASK 0x1FFF

/* ---- one workgroup per env: setup, binning and rasterisation without a round trip through HBM -------------------------------------
 *
 * Rounds 1-3 ran two launches: k_render_setup wrote every env's screen triangles and a per-tile COPY of them to global memory (485 MB of
 * writes per picture of 4096 envs), k_render_tiles (one wave per four 8 x 8 tiles: 262 144 waves) read them back through two dependent
 * loads per wave and stored 8 x 64-byte row pieces per tile.  Counters: 1.24 GB written for 805 MB of picture, 937 us per picture; the
 * waves did ~5 triangle tests per pixel -- the launch was the sum of per-wave latencies, not arithmetic and not HBM.
 *
 * Now the env's workgroup keeps its screen triangles in LDS (rcap records of 64 bytes; a scene with more spills the rest to global memory
 * and reads them back through the L2), bins them into 16 x 4-pixel tiles as lists of 16-bit record numbers (LDS), and its wavefronts then
 * walk the tiles, lane = pixel: a tile row is 16 pixels x 8 bytes = one full 128-byte line of the PositionSegmentation texture (32-byte
 * sectors of the depth and segmentation planes).  A pixel's winner is the record of largest 1/depth, ties to the smaller primitive id, so
 * neither the binning nor the order of a list can change a picture: bit-equal to oracle/orc_render.c's scan in primitive order. */

/* Can the triangle cover a pixel centre of tile (tx, ty)?  An edge function fma(A, x, fma(B, y, C)) is monotone in x and
 * in y (rounding is monotone), so over the tile's pixel centres it peaks at one of the four corner centres: if that
 * peak is negative for some edge, no centre of the tile passes the inside test.  Exact, not just conservative in R. */
MSK_DEV bool tile_touches_wh(float A0, float B0, float C0, float A1, float B1, float C1, float A2, float B2, float C2, int tx, int ty) {
  const float x0 = (float)(tx * MSK_TW) + 0.5f, x1 = (float)(tx * MSK_TW + MSK_TW - 1) + 0.5f;
  const float y0 = (float)(ty * MSK_TH) + 0.5f, y1 = (float)(ty * MSK_TH + MSK_TH - 1) + 0.5f;
  const float m0 = fmaxf(fmaxf(fmaf(A0, x0, fmaf(B0, y0, C0)), fmaf(A0, x1, fmaf(B0, y0, C0))), fmaxf(fmaf(A0, x0, fmaf(B0, y1, C0)), fmaf(A0, x1, fmaf(B0, y1, C0))));
  const float m1 = fmaxf(fmaxf(fmaf(A1, x0, fmaf(B1, y0, C1)), fmaf(A1, x1, fmaf(B1, y0, C1))), fmaxf(fmaf(A1, x0, fmaf(B1, y1, C1)), fmaf(A1, x1, fmaf(B1, y1, C1))));
  const float m2 = fmaxf(fmaxf(fmaf(A2, x0, fmaf(B2, y0, C2)), fmaf(A2, x1, fmaf(B2, y0, C2))), fmaxf(fmaf(A2, x0, fmaf(B2, y1, C2)), fmaf(A2, x1, fmaf(B2, y1, C2))));
  return m0 >= 0.0f && m1 >= 0.0f && m2 >= 0.0f;
}

/* Does the triangle cover EVERY pixel centre of tile (tx, ty)?  The same monotonicity: an edge function's minimum over the tile's pixel
 * centres is at one of the four corner centres, so if all three minima are >= 0 every centre passes the inside test -- exactly. */
MSK_DEV bool tile_covered_wh(float A0, float B0, float C0, float A1, float B1, float C1, float A2, float B2, float C2, int tx, int ty) {
  const float x0 = (float)(tx * MSK_TW) + 0.5f, x1 = (float)(tx * MSK_TW + MSK_TW - 1) + 0.5f;
  const float y0 = (float)(ty * MSK_TH) + 0.5f, y1 = (float)(ty * MSK_TH + MSK_TH - 1) + 0.5f;
  const float m0 = fminf(fminf(fmaf(A0, x0, fmaf(B0, y0, C0)), fmaf(A0, x1, fmaf(B0, y0, C0))), fminf(fmaf(A0, x0, fmaf(B0, y1, C0)), fmaf(A0, x1, fmaf(B0, y1, C0))));
  const float m1 = fminf(fminf(fmaf(A1, x0, fmaf(B1, y0, C1)), fmaf(A1, x1, fmaf(B1, y0, C1))), fminf(fmaf(A1, x0, fmaf(B1, y1, C1)), fmaf(A1, x1, fmaf(B1, y1, C1))));
  const float m2 = fminf(fminf(fmaf(A2, x0, fmaf(B2, y0, C2)), fmaf(A2, x1, fmaf(B2, y0, C2))), fminf(fmaf(A2, x0, fmaf(B2, y1, C2)), fmaf(A2, x1, fmaf(B2, y1, C2))));
  return m0 >= 0.0f && m1 >= 0.0f && m2 >= 0.0f;
}

__global__ void __launch_bounds__(MSK_RENDER_THREADS) k_render_env(const DModel* __restrict__ m, DState st, const RModel* __restrict__ rm, RCamera cam) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = cam.tile_cap, rcap = cam.rcap, icap = cam.icap;
  float* Lrec = lds;                                                /* [rcap][16] screen triangles (first: 16-byte aligned) */
  float* Lshape = Lrec + (size_t)rcap * MSK_SETUP_WORDS;            /* [ns][12] camera-from-shape pose, per-env scale */
  float* Llight = Lshape + cam.ns * MSK_RSHAPE_WORDS;               /* light directions, point / spot positions and axes in the camera frame */
  int* Lcnt = (int*)(Llight + MSK_LIGHT_WORDS);                     /* [ntiles + 1] per-tile counts, then list starts */
  int* Lfill = Lcnt + ntiles + 1;                                   /* [ntiles] */
  int* Lbig = Lfill + ntiles;                                       /* [MSK_MAX_BIG] record numbers of the large triangles */
  int* Lmisc = Lbig + MSK_MAX_BIG;                                  /* [8]: 0 records, 1 large ones, 2.. wave sums of the scan */
  unsigned short* Lmask = (unsigned short*)(Lmisc + 8);             /* [ntiles] large triangles that reach the tile */
  unsigned short* Lcover = Lmask + ((ntiles + 1) & ~1);             /* [ntiles] ... and those that cover every pixel centre of it */
  unsigned short* Lidx = Lcover + ((ntiles + 1) & ~1);              /* [icap] the tiles' lists of record numbers */
  const float* E = EREC(st, m, e);
  TriSetup* spill = (TriSetup*)(cam.setups + (size_t)e * cam.spill_cap * MSK_SETUP_WORDS);   /* records rcap, rcap + 1, ... */
  for (int i = tid; i <= ntiles; i += MSK_RENDER_THREADS) { Lcnt[i] = 0; if (i < ntiles) Lfill[i] = 0; }
  if (tid < 8) Lmisc[tid] = 0;
  /* camera-from-shape transforms */
  pose Tc = cam.local;
  if (cam.mount >= 0) Tc = pose_mul(load_pose(E, m->lay.bpose, cam.mount), cam.local);
  const pose Tci = pose_inv(Tc);
  if (tid < rm->nlights) {
    const v3 dcam = quat_rotate(Tci.q, v3_make(rm->ldir[tid][0], rm->ldir[tid][1], rm->ldir[tid][2]));
    Llight[tid * 3] = dcam.x; Llight[tid * 3 + 1] = dcam.y; Llight[tid * 3 + 2] = dcam.z;
  }
  float* Lppos = Llight + MSK_MAX_LIGHTS * 3;
  float* Lpdir = Lppos + MSK_MAX_LOCAL_LIGHTS * 3;
  if (tid >= 64 && tid - 64 < rm->nlocal) {
    const int l = tid - 64;
    const v3 xc = pose_apply(Tci, v3_make(rm->ppos[l][0], rm->ppos[l][1], rm->ppos[l][2]));
    const v3 dc = quat_rotate(Tci.q, v3_make(rm->pdir[l][0], rm->pdir[l][1], rm->pdir[l][2]));
    Lppos[l * 3] = xc.x; Lppos[l * 3 + 1] = xc.y; Lppos[l * 3 + 2] = xc.z;
    Lpdir[l * 3] = dc.x; Lpdir[l * 3 + 1] = dc.y; Lpdir[l * 3 + 2] = dc.z;
  }
  for (int s = tid; s < rm->ns; s += MSK_RENDER_THREADS) {
    const RShape* sh = &rm->shapes[s];
    pose L = sh->local;
    v3 scale = v3_make(1.0f, 1.0f, 1.0f);
    if (sh->xs >= 0) { /* follows a per-env box instance: unit-box vertices times the env's half sizes, the env's local position */
      const float* x = E + m->lay.xshape + sh->xs * 8;
      scale = v3_make(x[0], x[1], x[2]);
      L.p = v3_make(x[4], x[5], x[6]);
    }
    pose T = L;
    if (sh->body >= 0) T = pose_mul(load_pose(E, m->lay.bpose, sh->body), L);
    T = pose_mul(Tci, T);
    float* o = Lshape + s * MSK_RSHAPE_WORDS;
    o[0] = T.p.x; o[1] = T.p.y; o[2] = T.p.z; o[3] = T.q.w; o[4] = T.q.x; o[5] = T.q.y; o[6] = T.q.z;
    o[8] = scale.x; o[9] = scale.y; o[10] = scale.z;
  }
  __syncthreads();
#ifdef MSK_PROFILE_PHASES
#define RCUT(k) do { if (cam.dbg_cut == (k)) return; } while (0)
#else
#define RCUT(k)
#endif
  RCUT(1);
  /* ---- triangles: camera-frame corners (each thread transforms its triangles' own corners: the same arithmetic per vertex as a shared
   * vertex pass, without the LDS image and its barrier), near clip, projection, cull, edge and 1/depth planes -> records ---- */
  auto put_record = [&](int slot, const TriSetup& t) {
    if (slot < rcap) *(TriSetup*)(Lrec + (size_t)slot * MSK_SETUP_WORDS) = t;
    else spill[slot - rcap] = t;
  };
  for (int ti = tid; ti < rm->nt; ti += MSK_RENDER_THREADS) {
    const RTri tr = rm->tris[ti];
    const int vid[3] = {tr.v0, tr.v1, tr.v2};
    v3 p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* o = Lshape + rm->vshape[vid[k]] * MSK_RSHAPE_WORDS;
      pose T;
      T.p = v3_make(o[0], o[1], o[2]);
      T.q = quat_make(o[3], o[4], o[5], o[6]);
      const v3 vl = rm->verts[vid[k]];
      p[k] = pose_apply(T, v3_make(vl.x * o[8], vl.y * o[9], vl.z * o[10]));
    }
    const int seg = rm->shapes[tr.shape].seg;
    const unsigned col = cam.color ? shade_triangle(p[0], p[1], p[2], rm->shapes[tr.shape].color, rm->ambient, rm->nlights, Llight, &rm->lcol[0][0],
                                                    rm->nlocal, Lppos, Lpdir, &rm->pcol[0][0], &rm->pcone[0][0]) : 0u;
    /* clip against the near plane x >= near: a triangle becomes 0, 1 or 2 triangles */
    const bool in0 = p[0].x >= cam.near_, in1 = p[1].x >= cam.near_, in2 = p[2].x >= cam.near_;
    const int nin = (int)in0 + (int)in1 + (int)in2;
    v3 q[4];
    int nq = 0;
    if (nin == 3) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; nq = 3; }
    else if (nin > 0) {
      for (int k = 0; k < 3; ++k) { /* Sutherland-Hodgman against one plane, keeps the winding */
        const v3 a = p[k], b = p[(k + 1) % 3];
        const bool ia = a.x >= cam.near_, ib = b.x >= cam.near_;
        if (ia) q[nq++] = a;
        if (ia != ib) q[nq++] = ia ? lerp_near(a, b, cam.near_) : lerp_near(b, a, cam.near_);
      }
    }
    for (int sub = 0; sub + 2 < nq; ++sub) {
      TriSetup t;
      if (!setup_triangle(cam, q[0], q[sub + 1], q[sub + 2], seg, ti * 2 + sub, &t)) continue;
      t.color = col;
      const int slot = atomicAdd(&Lmisc[0], 1);
      if (slot >= rcap + cam.spill_cap) { atomicOr(cam.overflow, 1); continue; }
      const int tx0 = BB_X0(t.bb), tx1 = BB_X1(t.bb), ty0 = BB_Y0(t.bb), ty1 = BB_Y1(t.bb);
      if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > MSK_BIG_TILES) { /* table / ground sized: goes to the env's list of large triangles */
        const int b = atomicAdd(&Lmisc[1], 1);
        if (b < MSK_MAX_BIG) {
          Lbig[b] = slot;
          t.seg |= MSK_SEG_BIG;
          put_record(slot, t);
          continue;
        } /* list full: binned like a small one */
      }
      put_record(slot, t);
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          if (tile_touches_wh(t.A0, t.B0, t.C0, t.A1, t.B1, t.C1, t.A2, t.B2, t.C2, tx, ty)) atomicAdd(&Lcnt[ty * cam.tiles_x + tx], 1);
    }
  }
  __threadfence_block();   /* spilled records are read back by other threads of the workgroup */
  __syncthreads();
  const int ns = min(Lmisc[0], rcap + cam.spill_cap);
  const int nbig = min(Lmisc[1], MSK_MAX_BIG);
  RCUT(2);
  /* ---- list starts: exclusive scan of the tile counts (a thread owns `chunk` consecutive tiles) ---- */
  const int chunk = (ntiles + MSK_RENDER_THREADS - 1) / MSK_RENDER_THREADS;
  int mine = 0;
  for (int j = 0; j < chunk; ++j) { const int t = tid * chunk + j; if (t < ntiles) mine += Lcnt[t]; }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
  if (lane == 63) Lmisc[2 + wave] = incl;
  __syncthreads();
  int run = incl - mine;
  for (int w = 0; w < wave; ++w) run += Lmisc[2 + w];
  const int total = Lmisc[2] + Lmisc[3] + Lmisc[4] + Lmisc[5];
  for (int j = 0; j < chunk; ++j) {
    const int t = tid * chunk + j;
    if (t < ntiles) { const int c = Lcnt[t]; Lcnt[t] = min(run, icap); run += c; }
  }
  if (tid == 0) {
    Lcnt[ntiles] = min(total, icap);
    if (total > icap) atomicOr(cam.overflow, 1);
  }
  __syncthreads();
  RCUT(3);
  /* ---- fill: every tile's list of the records that can cover one of its pixel centres ---- */
  for (int s = tid; s < ns; s += MSK_RENDER_THREADS) {
    const TriSetup* t = (s < rcap) ? (const TriSetup*)(Lrec + (size_t)s * MSK_SETUP_WORDS) : &spill[s - rcap];
    const float A0 = t->A0, A1 = t->A1, B0 = t->B0, B1 = t->B1, C0 = t->C0, C1 = t->C1, A2 = t->A2, B2 = t->B2, C2 = t->C2;
    const int bb = t->bb;
    if (t->seg & MSK_SEG_BIG) continue;   /* lives in the list of large triangles */
    const int tx0 = BB_X0(bb), tx1 = BB_X1(bb), ty0 = BB_Y0(bb), ty1 = BB_Y1(bb);
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        if (!tile_touches_wh(A0, B0, C0, A1, B1, C1, A2, B2, C2, tx, ty)) continue;   /* same test as the count pass */
        const int tile = ty * cam.tiles_x + tx;
        const int pos = Lcnt[tile] + atomicAdd(&Lfill[tile], 1);
        if (pos < Lcnt[tile + 1]) Lidx[pos] = (unsigned short)s;
      }
  }
  /* per tile, the large triangles that can cover one of its pixel centres (bit b = entry b of the list) */
  for (int tile = tid; tile < ntiles; tile += MSK_RENDER_THREADS) {
    const int tx = tile % cam.tiles_x, ty = tile / cam.tiles_x;
    unsigned mk = 0u, cv = 0u;
    for (int b = 0; b < nbig; ++b) {
      const int s = Lbig[b];
      const TriSetup* t = (s < rcap) ? (const TriSetup*)(Lrec + (size_t)s * MSK_SETUP_WORDS) : &spill[s - rcap];
      if (BB_X0(t->bb) > tx || BB_X1(t->bb) < tx || BB_Y0(t->bb) > ty || (int)BB_Y1(t->bb) < ty) continue;
      if (tile_touches_wh(t->A0, t->B0, t->C0, t->A1, t->B1, t->C1, t->A2, t->B2, t->C2, tx, ty)) {
        mk |= 1u << b;
        if (tile_covered_wh(t->A0, t->B0, t->C0, t->A1, t->B1, t->C1, t->A2, t->B2, t->C2, tx, ty)) cv |= 1u << b;
      }
    }
    Lmask[tile] = (unsigned short)mk;
    Lcover[tile] = (unsigned short)cv;
  }
  __syncthreads();
  RCUT(4);
  /* ---- rasterise: a wavefront per tile, lane = pixel.  A tile's records are FETCHED lane = record (one LDS round trip for the whole list:
   * number -> record, 64 bytes per lane) and then handed round as scalars: v_readlane of the twelve coefficients into SGPRs, which the
   * three edge functions and the 1/depth plane of all 64 pixels then use as operands.  No memory access inside the record loop -- reading
   * record after record out of LDS as broadcasts (number, then record: two dependent round trips each) made a tile ~3 k cycles of waiting
   * for ~11 records.  The env's large triangles sit in lanes 0..15 for the whole walk. ---- */
  struct RecRegs { float4 a, b, c, d; };
  /* record number s (my lane's) -> registers.  SPILL = false: the env's records all fit the LDS image, and the walk below holds no
   * global LOAD at all -- with one in it (even on a path never taken) every tile waited at s_waitcnt vmcnt(0) for the previous tile's
   * STORES to be acknowledged, ~2 us x 64 tiles per wavefront: the whole picture took as long as with the two-launch form. */
  auto fetch_record = [&](int s, auto spill_tag) {
    RecRegs r;
    if (!decltype(spill_tag)::value || s < rcap) { const float4* t4 = (const float4*)(Lrec + (size_t)s * MSK_SETUP_WORDS); r.a = t4[0]; r.b = t4[1]; r.c = t4[2]; r.d = t4[3]; }
    else { const float4* t4 = (const float4*)&spill[s - rcap]; r.a = t4[0]; r.b = t4[1]; r.c = t4[2]; r.d = t4[3]; }
    return r;
  };
  const float wmin = 1.0f / cam.far_;
  auto walk_tiles = [&](auto spill_tag) {
    RecRegs big;
    big.a = big.b = big.c = big.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lane < nbig) big = fetch_record(Lbig[lane], spill_tag);
  #define MSK_LANE_F(v, j) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (j)))
  #define MSK_LANE_I(v, j) __builtin_amdgcn_readlane(__float_as_int(v), (j))
    /* lane j's record against my pixel: the same operations as oracle/orc_render.c's inner loop (record words: A0 A1 B0 B1 | C0 C1 A2 B2 |
     * C2 Aw Bw Cw | seg prim bb color) */
  #define MSK_RASTER_LANE(R, j)                                                                                                    \
    do {                                                                                                                            \
      const float e0 = fmaf(MSK_LANE_F((R).a.x, j), x, fmaf(MSK_LANE_F((R).a.z, j), y, MSK_LANE_F((R).b.x, j)));                    \
      const float e1 = fmaf(MSK_LANE_F((R).a.y, j), x, fmaf(MSK_LANE_F((R).a.w, j), y, MSK_LANE_F((R).b.y, j)));                    \
      const float e2 = fmaf(MSK_LANE_F((R).b.z, j), x, fmaf(MSK_LANE_F((R).b.w, j), y, MSK_LANE_F((R).c.x, j)));                    \
      const float w = fmaf(MSK_LANE_F((R).c.y, j), x, fmaf(MSK_LANE_F((R).c.z, j), y, MSK_LANE_F((R).c.w, j)));                     \
      const int prim = MSK_LANE_I((R).d.y, j), seg_j = MSK_LANE_I((R).d.x, j), col_j = MSK_LANE_I((R).d.w, j);   /* (broadcasts stay outside the   \
                                                                                                  * divergent branch: every lane takes part in them) */ \
      if (fminf(fminf(e0, e1), e2) >= 0.0f && w >= wmin && (w > best_w || (w == best_w && prim < best_prim))) {                      \
        best_w = w; best_prim = prim; best_seg = seg_j; best_col = (unsigned)col_j;                                                  \
      }                                                                                                                             \
    } while (0)
    /* the first chunk (<= 64 records) of a tile's list, lane = record; the next tile's is requested before this tile is rasterised */
    auto fetch_chunk = [&](int k0, int k1) {
      RecRegs r;
      r.a = r.b = r.c = r.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (k0 + lane < k1) r = fetch_record((int)Lidx[k0 + lane], spill_tag);
      return r;
    };
    /* lane j's record covers the whole tile (Lcover): only the 1/depth plane is left to evaluate */
#define MSK_RASTER_LANE_W(R, j)                                                                                                  \
  do {                                                                                                                            \
    const float w = fmaf(MSK_LANE_F((R).c.y, j), x, fmaf(MSK_LANE_F((R).c.z, j), y, MSK_LANE_F((R).c.w, j)));                     \
    const int prim = MSK_LANE_I((R).d.y, j), seg_j = MSK_LANE_I((R).d.x, j), col_j = MSK_LANE_I((R).d.w, j);                      \
    if (w >= wmin && (w > best_w || (w == best_w && prim < best_prim))) {                                                          \
      best_w = w; best_prim = prim; best_seg = seg_j; best_col = (unsigned)col_j;                                                  \
    }                                                                                                                             \
  } while (0)
    /* the wavefront walks tile columns wave, wave + 4, ... of every tile row: what depends on the row only (the pixel's y, its camera-space
     * y per unit depth) is computed once per row */
    const int wstep = MSK_RENDER_THREADS / 64;
    const int tiles_x = cam.tiles_x, tiles_y = cam.tiles_y;
    const int first = wave < tiles_x ? wave : -1;       /* my first tile of row 0 (-1: a picture narrower than four tiles leaves me idle) */
    RecRegs nxt = fetch_chunk(first >= 0 ? Lcnt[first] : 0, first >= 0 ? Lcnt[first + 1] : 0);
    for (int ty = 0; ty < tiles_y && first >= 0; ++ty) {
      const int py = ty * MSK_TH + (lane / MSK_TW);
      const float y = (float)py + 0.5f;
      const float gyc = -(y - cam.cy) / cam.fy;
      const size_t rowpix = ((size_t)e * cam.H + py) * cam.W + (lane & (MSK_TW - 1));
    for (int tx = wave; tx < tiles_x; tx += wstep) {
      const int tile = ty * tiles_x + tx;
      const int px = tx * MSK_TW + (lane & (MSK_TW - 1));
      const float x = (float)px + 0.5f;
      float best_w = 0.0f;
      int best_seg = 0, best_prim = 0x7FFFFFFF;
      unsigned best_col = 0u;
      const int l0 = __builtin_amdgcn_readfirstlane(Lcnt[tile]), l1 = __builtin_amdgcn_readfirstlane(Lcnt[tile + 1]);
      const unsigned bigmask = (unsigned)__builtin_amdgcn_readfirstlane((int)Lmask[tile]);
      const unsigned covmask = (unsigned)__builtin_amdgcn_readfirstlane((int)Lcover[tile]);
      RecRegs cur = nxt;
      { /* the next tile of my walk: same row, or the first of the next row */
        const int ntile = (tx + wstep < tiles_x) ? tile + wstep : ((ty + 1 < tiles_y) ? (ty + 1) * tiles_x + wave : -1);
        if (ntile >= 0) nxt = fetch_chunk(Lcnt[ntile], Lcnt[ntile + 1]);
      }
#ifdef MSK_PROFILE_PHASES
      if (cam.dbg_cut != 6)       /* 6: no record loops, stores only */
#endif
      {
        for (unsigned mk = bigmask & covmask; mk != 0u; mk &= mk - 1u) {
          const int j = __builtin_ctz(mk);
          MSK_RASTER_LANE_W(big, j);
        }
        for (unsigned mk = bigmask & ~covmask; mk != 0u; mk &= mk - 1u) {
          const int j = __builtin_ctz(mk);
          MSK_RASTER_LANE(big, j);
        }
        for (int k0 = l0; k0 < l1; k0 += 64) {
          if (k0 > l0) cur = fetch_chunk(k0, l1);      /* a list longer than a wavefront: the further chunks are fetched in place */
          const int n = min(64, l1 - k0);
          for (int j = 0; j < n; ++j) MSK_RASTER_LANE(cur, j);
        }
      }
      /* camera-space OpenGL position in millimetres (x right, y up, z backwards), int16 saturated */
      short4 o = make_short4(0, 0, 0, 0);
      if (best_w > 0.0f) {
        const float d = 1.0f / best_w;
        const float gx = (x - cam.cx) / cam.fx * d, gy = gyc * d, gz = -d;
        o.x = (short)fminf(fmaxf(rintf(gx * 1000.0f), -32768.0f), 32767.0f);
        o.y = (short)fminf(fmaxf(rintf(gy * 1000.0f), -32768.0f), 32767.0f);
        o.z = (short)fminf(fmaxf(rintf(gz * 1000.0f), -32768.0f), 32767.0f);
        o.w = (short)(best_seg & 0xFFFF);
      }
      const size_t pix = rowpix + (size_t)tx * MSK_TW;
#ifdef MSK_PROFILE_PHASES
      if (cam.dbg_cut == 5 && best_prim != -7) continue;      /* 5: the walk without its stores */
      if (cam.dbg_cut == 7) { ((short4*)cam.out)[pix] = o; continue; }   /* 7: only the 8-byte texture store */
#endif
      if (cam.want_tex) ((short4*)cam.out)[pix] = o;
      if (cam.color) cam.color[pix] = best_col;
      cam.depth[pix] = (short)(-(int)o.z);   /* int16 negation wraps like the host-side `-position[..., 2]` */
      cam.seg[pix] = o.w;
    }
    }
  };
  if (ns > rcap) walk_tiles(std::true_type{});
  else walk_tiles(std::false_type{});
}

/* ---- k_render_splat: small triangles are splatted lane = record row, the rest walks the tiles as in k_render_env ------------------------
 *
 * Measured on k_render_env (round 4): the walk is VALU-issue bound, ~45 instructions per (record, 16 x 4 tile) pair whatever the record
 * covers, and 80 % of the benchmarked scenes' records have a pixel box of <= 32 pixels (tools/oracle_render_stats.py).  Here a record whose
 * box is <= 16 x 16 pixels never enters a tile list.  The picture is produced tile row by tile row (4 pixel rows): the records that reach
 * row r (a list per tile row, built like the tile lists) are taken four lanes to a record -- lane k of the quad owns pixel row k of the
 * tile row and walks the record's own columns with the record's coefficients in ITS registers: no broadcast -- and a covered pixel centre
 * does one ds_max_u64 of (1/depth bits, 16383 - primitive id, record number) on the row's key buffer: the largest key is the nearest
 * surface, ties to the lower primitive id, exactly oracle/orc_render.c's rule, whatever the order of the atomics.  The tile walk of row r
 * then starts each pixel from its key instead of from "nothing", runs the large-triangle masks and the (now short) tile lists as
 * k_render_env does, and reads segmentation id and colour from the winning record.  Two key buffers: the splats of row r + 1 and the walk
 * of row r share one barrier interval. */
typedef __attribute__((address_space(3))) unsigned long long msk_lds_u64;

__global__ void __launch_bounds__(MSK_RENDER_THREADS) k_render_splat(const DModel* __restrict__ m, DState st, const RModel* __restrict__ rm, RCamera cam) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = cam.tile_cap, rcap = cam.rcap, icap = cam.icap, bcap = cam.bcap, tiles_x = cam.tiles_x, tiles_y = cam.tiles_y;
  const int segs_x = (tiles_x + MSK_SEG_TILES - 1) / MSK_SEG_TILES, nseg = tiles_y * segs_x;
  float* Lrec = lds;                                                /* [rcap][16] screen triangles (first: 16-byte aligned) */
  float* Luv = Lrec + (size_t)rcap * MSK_SETUP_WORDS;               /* [uvcap][8] Au Bu Cu Av Bv Cv, w | h << 16, first texel: textured records */
  unsigned long long* Lkey = (unsigned long long*)(Luv + (size_t)cam.uvcap * 8);   /* [4 wavefronts][MSK_SEG_PX] */
  float* Lshape = (float*)(Lkey + (size_t)(MSK_RENDER_THREADS / 64) * MSK_SEG_PX);   /* [ns][12] */
  float* Llight = Lshape + cam.ns * MSK_RSHAPE_WORDS;
  int* Lcnt = (int*)(Llight + MSK_LIGHT_WORDS);                     /* [ntiles + 1] */
  int* Lfill = Lcnt + ntiles + 1;                                   /* [ntiles] */
  int* Lbig = Lfill + ntiles;                                       /* [MSK_MAX_BIG] */
  int* Lmisc = Lbig + MSK_MAX_BIG;                                  /* [16]: 0 records, 1 large ones, 2..5 wave sums of a scan, 6 textured records, 7 packed work items, 8 the next segment to draw */
  unsigned short* Lmask = (unsigned short*)(Lmisc + 16);
  unsigned short* Lcover = Lmask + ((ntiles + 1) & ~1);
  unsigned short* Lidx = Lcover + ((ntiles + 1) & ~1);              /* [icap] the tiles' lists: medium records only */
  int* Bcnt = (int*)(Lidx + ((icap + 1) & ~1));                     /* [nseg + 1] per segment: its small records, then list starts */
  int* Bfill = Bcnt + nseg + 1;                                     /* [nseg] */
  unsigned short* Bidx = (unsigned short*)(Bfill + nseg);           /* [bcap] */
  const float* E = EREC(st, m, e);
  TriSetup* spill = (TriSetup*)(cam.setups + (size_t)e * cam.spill_cap * MSK_SETUP_WORDS);
  for (int i = tid; i <= ntiles; i += MSK_RENDER_THREADS) { Lcnt[i] = 0; if (i < ntiles) Lfill[i] = 0; }
  for (int i = tid; i <= nseg; i += MSK_RENDER_THREADS) { Bcnt[i] = 0; if (i < nseg) Bfill[i] = 0; }
  for (int i = tid; i < (MSK_RENDER_THREADS / 64) * MSK_SEG_PX; i += MSK_RENDER_THREADS) Lkey[i] = 0ull;
  if (tid < 16) Lmisc[tid] = 0;
  /* camera-from-shape transforms, lights in the camera frame (as k_render_env) */
  pose Tc = cam.local;
  if (cam.mount >= 0) Tc = pose_mul(load_pose(E, m->lay.bpose, cam.mount), cam.local);
  const pose Tci = pose_inv(Tc);
  if (tid < rm->nlights) {
    const v3 dcam = quat_rotate(Tci.q, v3_make(rm->ldir[tid][0], rm->ldir[tid][1], rm->ldir[tid][2]));
    Llight[tid * 3] = dcam.x; Llight[tid * 3 + 1] = dcam.y; Llight[tid * 3 + 2] = dcam.z;
  }
  float* Lppos = Llight + MSK_MAX_LIGHTS * 3;
  float* Lpdir = Lppos + MSK_MAX_LOCAL_LIGHTS * 3;
  if (tid >= 64 && tid - 64 < rm->nlocal) {
    const int l = tid - 64;
    const v3 xc = pose_apply(Tci, v3_make(rm->ppos[l][0], rm->ppos[l][1], rm->ppos[l][2]));
    const v3 dc = quat_rotate(Tci.q, v3_make(rm->pdir[l][0], rm->pdir[l][1], rm->pdir[l][2]));
    Lppos[l * 3] = xc.x; Lppos[l * 3 + 1] = xc.y; Lppos[l * 3 + 2] = xc.z;
    Lpdir[l * 3] = dc.x; Lpdir[l * 3 + 1] = dc.y; Lpdir[l * 3 + 2] = dc.z;
  }
  for (int s = tid; s < rm->ns; s += MSK_RENDER_THREADS) {
    const RShape* sh = &rm->shapes[s];
    pose L = sh->local;
    v3 scale = v3_make(1.0f, 1.0f, 1.0f);
    if (sh->xs >= 0) {
      const float* x = E + m->lay.xshape + sh->xs * 8;
      scale = v3_make(x[0], x[1], x[2]);
      L.p = v3_make(x[4], x[5], x[6]);
    }
    pose T = L;
    if (sh->body >= 0) T = pose_mul(load_pose(E, m->lay.bpose, sh->body), L);
    T = pose_mul(Tci, T);
    float* o = Lshape + s * MSK_RSHAPE_WORDS;
    o[0] = T.p.x; o[1] = T.p.y; o[2] = T.p.z; o[3] = T.q.w; o[4] = T.q.x; o[5] = T.q.y; o[6] = T.q.z;
    o[8] = scale.x; o[9] = scale.y; o[10] = scale.z;
  }
  __syncthreads();
  RCUT(1);
  /* ---- triangles -> records; small ones are counted per tile row, medium ones per tile, large ones go to the env's list ---- */
  auto put_record = [&](int slot, const TriSetup& t) {
    if (slot < rcap) *(TriSetup*)(Lrec + (size_t)slot * MSK_SETUP_WORDS) = t;
    else spill[slot - rcap] = t;
  };
  auto corners = [&](const RTri& tr, v3* p) {
    const int vid[3] = {tr.v0, tr.v1, tr.v2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* o = Lshape + rm->vshape[vid[k]] * MSK_RSHAPE_WORDS;
      pose T;
      T.p = v3_make(o[0], o[1], o[2]);
      T.q = quat_make(o[3], o[4], o[5], o[6]);
      const v3 vl = rm->verts[vid[k]];
      p[k] = pose_apply(T, v3_make(vl.x * o[8], vl.y * o[9], vl.z * o[10]));
    }
  };
  /* First pass: which triangles can get a record at all.  Two of three are culled (back faces, off the picture), and a wavefront that sets up 64 triangles
   * of which 20 survive pays the whole set-up -- shading, clipping, the planes -- for all 64.  Here every triangle gets its corners and the cull's own tests
   * (~ a quarter of the set-up), the survivors' numbers are packed into a work list (it lives where the tile lists will be: those are filled later), and the
   * set-up below runs on full wavefronts.  Which slot a record gets was never defined (an atomic counter): the pictures do not depend on it. */
  unsigned short* Lwork = Lidx;
  const int nt = rm->nt;
  const bool packed = nt <= icap;      /* (icap >= 2 x triangles up to 32767 triangles: always, for templates within the rasteriser's capacity) */
  if (packed) {
    for (int t0 = 0; t0 < nt; t0 += MSK_RENDER_THREADS) { /* uniform trip count: the ballot below needs the whole wavefront */
      const int ti = t0 + tid;
      bool keep = false;
      if (ti < nt) {
        const RTri tr = rm->tris[ti];
        v3 p[3];
        corners(tr, p);
        const int nin = (int)(p[0].x >= cam.near_) + (int)(p[1].x >= cam.near_) + (int)(p[2].x >= cam.near_);
        keep = nin == 3 ? triangle_survives_cull(cam, p[0], p[1], p[2]) != 0 : nin > 0;      /* (a clipped triangle is decided by the set-up itself) */
      }
      const unsigned long long mk = __ballot(keep);
      if (mk != 0ull) {
        int base = 0;
        if (lane == __ffsll((long long)mk) - 1) base = atomicAdd(&Lmisc[7], __popcll(mk));
        base = __builtin_amdgcn_readlane(base, __ffsll((long long)mk) - 1);
        if (keep) Lwork[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)ti;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  const int nwork = packed ? Lmisc[7] : nt;
  for (int wi = tid; wi < nwork; wi += MSK_RENDER_THREADS) {
    const int ti = packed ? (int)Lwork[wi] : wi;
    const RTri tr = rm->tris[ti];
    const int vid[3] = {tr.v0, tr.v1, tr.v2};
    v3 p[3];
    corners(tr, p);
    const int seg = rm->shapes[tr.shape].seg;
    const unsigned col = cam.color ? shade_triangle(p[0], p[1], p[2], rm->shapes[tr.shape].color, rm->ambient, rm->nlights, Llight, &rm->lcol[0][0],
                                                    rm->nlocal, Lppos, Lpdir, &rm->pcol[0][0], &rm->pcone[0][0]) : 0u;
    const bool in0 = p[0].x >= cam.near_, in1 = p[1].x >= cam.near_, in2 = p[2].x >= cam.near_;
    const int nin = (int)in0 + (int)in1 + (int)in2;
    const int tex = (cam.uvcap > 0 && cam.color) ? rm->shapes[tr.shape].tex : -1;
    v3 q[4];
    float quv[4][2];
    int nq = 0;
    if (nin == 3) {
      q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; nq = 3;
      if (tex >= 0)
        for (int k = 0; k < 3; ++k) { quv[k][0] = rm->vuv[vid[k]][0]; quv[k][1] = rm->vuv[vid[k]][1]; }
    } else if (nin > 0) {
      for (int k = 0; k < 3; ++k) {
        const v3 a = p[k], b = p[(k + 1) % 3];
        const bool ia = a.x >= cam.near_, ib = b.x >= cam.near_;
        float ua[2] = {0.0f, 0.0f}, ub[2] = {0.0f, 0.0f};
        if (tex >= 0) { ua[0] = rm->vuv[vid[k]][0]; ua[1] = rm->vuv[vid[k]][1]; ub[0] = rm->vuv[vid[(k + 1) % 3]][0]; ub[1] = rm->vuv[vid[(k + 1) % 3]][1]; }
        if (ia) { quv[nq][0] = ua[0]; quv[nq][1] = ua[1]; q[nq++] = a; }
        if (ia != ib) {
          if (ia) { lerp_near_uv(a, b, cam.near_, ua, ub, quv[nq]); q[nq++] = lerp_near(a, b, cam.near_); }
          else { lerp_near_uv(b, a, cam.near_, ub, ua, quv[nq]); q[nq++] = lerp_near(b, a, cam.near_); }
        }
      }
    }
    for (int sub = 0; sub + 2 < nq; ++sub) {
      TriSetup t;
      int box[4];
      float uvp[6];
      const float uv6[6] = {quv[0][0], quv[0][1], quv[sub + 1][0], quv[sub + 1][1], quv[sub + 2][0], quv[sub + 2][1]};
      if (!setup_triangle(cam, q[0], q[sub + 1], q[sub + 2], seg, ti * 2 + sub, &t, box, tex >= 0 ? uv6 : nullptr, uvp)) continue;
      t.color = col;
      const int slot = atomicAdd(&Lmisc[0], 1);
      if (slot >= rcap + cam.spill_cap) { atomicOr(cam.overflow, 1); continue; }
      if (tex >= 0) { /* the record's texture planes: a slot of the workgroup's table, named in the record's seg word */
        const int us = atomicAdd(&Lmisc[6], 1);
        if (us < cam.uvcap && us < MSK_SEG_UV_MASK) {
          float* o = Luv + (size_t)us * 8;
          o[0] = uvp[0]; o[1] = uvp[1]; o[2] = uvp[2]; o[3] = uvp[3]; o[4] = uvp[4]; o[5] = uvp[5];
          o[6] = __int_as_float(rm->tex[tex].w | (rm->tex[tex].h << 16)); o[7] = __int_as_float(rm->tex[tex].ofs);
          t.seg |= (us + 1) << MSK_SEG_UV_SHIFT;
        } else atomicOr(cam.overflow, 1);      /* drawn in its flat colour */
      }
      if (box[1] - box[0] < MSK_SPLAT_MAX && box[3] - box[2] < MSK_SPLAT_MAX) { /* small: its own pixel box, per tile row */
        t.seg |= MSK_SEG_SMALL;
        t.bb = box[0] | (box[2] << 10) | ((box[1] - box[0]) << 20) | ((box[3] - box[2]) << 24);
        put_record(slot, t);
        for (int ty = box[2] / MSK_TH; ty <= box[3] / MSK_TH; ++ty)
          for (int sx = box[0] / (MSK_SEG_TILES * MSK_TW); sx <= box[1] / (MSK_SEG_TILES * MSK_TW); ++sx) atomicAdd(&Bcnt[ty * segs_x + sx], 1);
        continue;
      }
      const int tx0 = BB_X0(t.bb), tx1 = BB_X1(t.bb), ty0 = BB_Y0(t.bb), ty1 = BB_Y1(t.bb);
      if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > MSK_BIG_TILES) {
        const int b = atomicAdd(&Lmisc[1], 1);
        if (b < MSK_MAX_BIG) {
          Lbig[b] = slot;
          t.seg |= MSK_SEG_BIG;
          put_record(slot, t);
          continue;
        }
      }
      put_record(slot, t);
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          if (tile_touches_wh(t.A0, t.B0, t.C0, t.A1, t.B1, t.C1, t.A2, t.B2, t.C2, tx, ty)) atomicAdd(&Lcnt[ty * tiles_x + tx], 1);
    }
  }
  __threadfence_block();
  __syncthreads();
  const int ns = min(Lmisc[0], rcap + cam.spill_cap);
  const int nbig = min(Lmisc[1], MSK_MAX_BIG);
  RCUT(2);
  /* ---- list starts: exclusive scans of the tile counts and of the tile-row counts ---- */
  auto block_scan = [&](int* cnt, int n, int cap) {
    const int chunk = (n + MSK_RENDER_THREADS - 1) / MSK_RENDER_THREADS;
    int mine = 0;
    for (int j = 0; j < chunk; ++j) { const int t = tid * chunk + j; if (t < n) mine += cnt[t]; }
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
    if (lane == 63) Lmisc[2 + wave] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += Lmisc[2 + w];
    const int total = Lmisc[2] + Lmisc[3] + Lmisc[4] + Lmisc[5];
    for (int j = 0; j < chunk; ++j) {
      const int t = tid * chunk + j;
      if (t < n) { const int c = cnt[t]; cnt[t] = min(run, cap); run += c; }
    }
    if (tid == 0) {
      cnt[n] = min(total, cap);
      if (total > cap) atomicOr(cam.overflow, 1);
    }
    __syncthreads();
  };
  block_scan(Lcnt, ntiles, icap);
  block_scan(Bcnt, nseg, bcap);
  RCUT(3);
  /* ---- fill the lists ---- */
  const float wmin_far = 1.0f / cam.far_;
  for (int s = tid; s < ns; s += MSK_RENDER_THREADS) {
    const TriSetup* t = (s < rcap) ? (const TriSetup*)(Lrec + (size_t)s * MSK_SETUP_WORDS) : &spill[s - rcap];
    const int segf = t->seg, bb = t->bb;
    if (segf & MSK_SEG_BIG) continue;
    if (segf & MSK_SEG_SMALL) {
      const int x0 = bb & 1023, x1 = x0 + ((bb >> 20) & 15), y0 = (bb >> 10) & 1023, y1 = y0 + ((bb >> 24) & 15);
      for (int ty = y0 / MSK_TH; ty <= y1 / MSK_TH; ++ty)
        for (int sx = x0 / (MSK_SEG_TILES * MSK_TW); sx <= x1 / (MSK_SEG_TILES * MSK_TW); ++sx) {
          const int sg = ty * segs_x + sx;
          const int pos = Bcnt[sg] + atomicAdd(&Bfill[sg], 1);
          if (pos < Bcnt[sg + 1]) Bidx[pos] = (unsigned short)s;
        }
      continue;
    }
    const float A0 = t->A0, A1 = t->A1, B0 = t->B0, B1 = t->B1, C0 = t->C0, C1 = t->C1, A2 = t->A2, B2 = t->B2, C2 = t->C2;
    const int tx0 = BB_X0(bb), tx1 = BB_X1(bb), ty0 = BB_Y0(bb), ty1 = BB_Y1(bb);
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        if (!tile_touches_wh(A0, B0, C0, A1, B1, C1, A2, B2, C2, tx, ty)) continue;
        const int tile = ty * tiles_x + tx;
        const int pos = Lcnt[tile] + atomicAdd(&Lfill[tile], 1);
        if (pos < Lcnt[tile + 1]) Lidx[pos] = (unsigned short)s;
      }
  }
  for (int tile = tid; tile < ntiles; tile += MSK_RENDER_THREADS) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    unsigned mk = 0u, cv = 0u;
    /* 1/depth is affine on the screen and fma rounding is monotone, so over the tile's pixel centres a record's 1/depth peaks and bottoms
     * out at corner centres.  A large triangle that covers the whole tile and is in front of the far plane at its farthest corner hides
     * every large triangle whose NEAREST corner value is still farther: those leave the tile's mask (the ground under the table top). */
    const float x0 = (float)(tx * MSK_TW) + 0.5f, x1 = (float)(tx * MSK_TW + MSK_TW - 1) + 0.5f;
    const float y0 = (float)(ty * MSK_TH) + 0.5f, y1 = (float)(ty * MSK_TH + MSK_TH - 1) + 0.5f;
    float shield = 0.0f;
    for (int b = 0; b < nbig; ++b) {
      const int s = Lbig[b];
      const TriSetup* t = (s < rcap) ? (const TriSetup*)(Lrec + (size_t)s * MSK_SETUP_WORDS) : &spill[s - rcap];
      if (BB_X0(t->bb) > tx || BB_X1(t->bb) < tx || BB_Y0(t->bb) > ty || (int)BB_Y1(t->bb) < ty) continue;
      if (tile_touches_wh(t->A0, t->B0, t->C0, t->A1, t->B1, t->C1, t->A2, t->B2, t->C2, tx, ty)) {
        mk |= 1u << b;
        if (tile_covered_wh(t->A0, t->B0, t->C0, t->A1, t->B1, t->C1, t->A2, t->B2, t->C2, tx, ty)) {
          cv |= 1u << b;
          const float w00 = fmaf(t->Aw, x0, fmaf(t->Bw, y0, t->Cw)), w10 = fmaf(t->Aw, x1, fmaf(t->Bw, y0, t->Cw));
          const float w01 = fmaf(t->Aw, x0, fmaf(t->Bw, y1, t->Cw)), w11 = fmaf(t->Aw, x1, fmaf(t->Bw, y1, t->Cw));
          const float wlo = fminf(fminf(w00, w10), fminf(w01, w11));
          if (wlo >= wmin_far) shield = fmaxf(shield, wlo);
        }
      }
    }
    if (shield > 0.0f)
      for (unsigned left = mk; left != 0u; left &= left - 1u) {
        const int b = __builtin_ctz(left);
        const int s = Lbig[b];
        const TriSetup* t = (s < rcap) ? (const TriSetup*)(Lrec + (size_t)s * MSK_SETUP_WORDS) : &spill[s - rcap];
        const float w00 = fmaf(t->Aw, x0, fmaf(t->Bw, y0, t->Cw)), w10 = fmaf(t->Aw, x1, fmaf(t->Bw, y0, t->Cw));
        const float w01 = fmaf(t->Aw, x0, fmaf(t->Bw, y1, t->Cw)), w11 = fmaf(t->Aw, x1, fmaf(t->Bw, y1, t->Cw));
        if (fmaxf(fmaxf(w00, w10), fmaxf(w01, w11)) < shield) { mk &= ~(1u << b); cv &= ~(1u << b); }
      }
    Lmask[tile] = (unsigned short)mk;
    Lcover[tile] = (unsigned short)cv;
  }
  __syncthreads();
  RCUT(4);
  /* ---- segments: a wavefront splats the small records of its segment into its own key buffer, then walks the segment's tiles.  Nothing
   * below is shared between wavefronts but read-only lists and records: no workgroup barrier. ---- */
  struct RecRegs { float4 a, b, c, d; };
  const float wmin = 1.0f / cam.far_;
  auto rows = [&](auto spill_tag) {
    auto rec4 = [&](int s) -> const float4* {
      if (!decltype(spill_tag)::value || s < rcap) return (const float4*)(Lrec + (size_t)s * MSK_SETUP_WORDS);
      return (const float4*)&spill[s - rcap];
    };
    auto fetch_record = [&](int s) { RecRegs r; const float4* t4 = rec4(s); r.a = t4[0]; r.b = t4[1]; r.c = t4[2]; r.d = t4[3]; return r; };
    unsigned long long* keys = Lkey + (size_t)wave * MSK_SEG_PX;      /* [4 rows][64 columns] of my current segment */
    /* small records of segment (ty, sx): four lanes per record, lane k of the quad = pixel row k of the tile row, columns clipped to the segment */
    auto splat_segment = [&](int sg, int ty, int sx) {
      const int b0 = Bcnt[sg], b1 = Bcnt[sg + 1];
      const int py = ty * MSK_TH + (lane & 3);
      const float y = (float)py + 0.5f;
      const int cx0 = sx * (MSK_SEG_TILES * MSK_TW);
      msk_lds_u64* krow = (msk_lds_u64*)(keys + (size_t)(lane & 3) * (MSK_SEG_TILES * MSK_TW)) - cx0;
      for (int qi = b0 + (lane >> 2); qi < b1; qi += 16) {
        const int s = (int)Bidx[qi];
        const float4* t4 = rec4(s);
        const float4 ra = t4[0], rb = t4[1], rc = t4[2], rd = t4[3];
        const int prim = __float_as_int(rd.y), bb = __float_as_int(rd.z);
        const int x0 = max(bb & 1023, cx0), y0 = (bb >> 10) & 1023, x1 = min((bb & 1023) + ((bb >> 20) & 15), cx0 + MSK_SEG_TILES * MSK_TW - 1), y1 = y0 + ((bb >> 24) & 15);
        if (py < y0 || py > y1) continue;
        const float t0 = fmaf(ra.z, y, rb.x), t1 = fmaf(ra.w, y, rb.y), t2 = fmaf(rb.w, y, rc.x), tw = fmaf(rc.z, y, rc.w);
        const unsigned lo = ((unsigned)(16383 - prim) << 16) | (unsigned)s;
        for (int px = x0; px <= x1; ++px) {
          const float x = (float)px + 0.5f;
          const float e0 = fmaf(ra.x, x, t0), e1 = fmaf(ra.y, x, t1), e2 = fmaf(rb.z, x, t2), w = fmaf(rc.y, x, tw);
          if (fminf(fminf(e0, e1), e2) >= 0.0f && w >= wmin)
            __hip_atomic_fetch_max(krow + px, ((unsigned long long)__float_as_uint(w) << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    };
    RecRegs big;
    big.a = big.b = big.c = big.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int big_s = 0;
    if (lane < nbig) { big_s = Lbig[lane]; big = fetch_record(big_s); }
    /* lane j's record against my pixel: the operations of oracle/orc_render.c's inner loop; the winner is remembered by record number */
#define MSK_SPLAT_LANE(R, S, j)                                                                                                     \
    do {                                                                                                                            \
      const float e0 = fmaf(MSK_LANE_F((R).a.x, j), x, fmaf(MSK_LANE_F((R).a.z, j), y, MSK_LANE_F((R).b.x, j)));                    \
      const float e1 = fmaf(MSK_LANE_F((R).a.y, j), x, fmaf(MSK_LANE_F((R).a.w, j), y, MSK_LANE_F((R).b.y, j)));                    \
      const float e2 = fmaf(MSK_LANE_F((R).b.z, j), x, fmaf(MSK_LANE_F((R).b.w, j), y, MSK_LANE_F((R).c.x, j)));                    \
      const float w = fmaf(MSK_LANE_F((R).c.y, j), x, fmaf(MSK_LANE_F((R).c.z, j), y, MSK_LANE_F((R).c.w, j)));                     \
      const int prim = MSK_LANE_I((R).d.y, j), slot_j = __builtin_amdgcn_readlane((S), (j));   /* (outside the divergent branch: every lane takes part) */ \
      if (fminf(fminf(e0, e1), e2) >= 0.0f && w >= wmin && (w > best_w || (w == best_w && prim < best_prim))) {                      \
        best_w = w; best_prim = prim; best_slot = slot_j;                                                                          \
      }                                                                                                                             \
    } while (0)
#define MSK_SPLAT_LANE_W(R, S, j)                                                                                                   \
    do {                                                                                                                            \
      const float w = fmaf(MSK_LANE_F((R).c.y, j), x, fmaf(MSK_LANE_F((R).c.z, j), y, MSK_LANE_F((R).c.w, j)));                     \
      const int prim = MSK_LANE_I((R).d.y, j), slot_j = __builtin_amdgcn_readlane((S), (j));                                       \
      if (w >= wmin && (w > best_w || (w == best_w && prim < best_prim))) {                                                          \
        best_w = w; best_prim = prim; best_slot = slot_j;                                                                          \
      }                                                                                                                             \
    } while (0)
    auto fetch_chunk = [&](int k0, int k1, int* sout) {
      RecRegs r;
      r.a = r.b = r.c = r.d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      int s = 0;
      if (k0 + lane < k1) { s = (int)Lidx[k0 + lane]; r = fetch_record(s); }
      *sout = s;
      return r;
    };
#ifdef MSK_PROFILE_PHASES   /* tools/gpu_render_probe.py: 5 splats only, 6 no splats, 7 no large triangles, 8 no record loops at all, 9 no stores */
    const int cut = cam.dbg_cut;
#define MSK_CUT_IS(k) (cut == (k))
#else
#define MSK_CUT_IS(k) false
#endif
    /* segments are TAKEN, not dealt: wavefront w used to own segments w, w + 4, ... = every other tile row of one half of the picture, so a robot in the left half made
     * two wavefronts do the work while two waited at the end of the workgroup; a counter in LDS hands the next segment to whoever is free (which wavefront draws a segment
     * does not reach the picture: its key buffer is its own and empty again after every segment) */
    for (;;) {
      int sg = 0;
      if (lane == 0) sg = atomicAdd(&Lmisc[8], 1);
      sg = __builtin_amdgcn_readfirstlane(sg);
      if (sg >= nseg) break;
      const int ty = sg / segs_x, sx = sg - ty * segs_x;
      if (!MSK_CUT_IS(6) && !MSK_CUT_IS(8)) splat_segment(sg, ty, sx);
      /* my wavefront's atomics above, my wavefront's reads below: LDS operations of one wavefront complete in order */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (MSK_CUT_IS(5)) continue;
      const int py = ty * MSK_TH + (lane / MSK_TW);
      const float y = (float)py + 0.5f;
      const float gyc = -(y - cam.cy) / cam.fy;
      const size_t rowpix = ((size_t)e * cam.H + py) * cam.W + (lane & (MSK_TW - 1));
      unsigned long long* krow = keys + (size_t)(lane / MSK_TW) * (MSK_SEG_TILES * MSK_TW) + (lane & (MSK_TW - 1));
      const int txe = min(tiles_x, (sx + 1) * MSK_SEG_TILES);
      int nxt_s = 0;
      RecRegs nxt = fetch_chunk(Lcnt[ty * tiles_x + sx * MSK_SEG_TILES], Lcnt[ty * tiles_x + sx * MSK_SEG_TILES + 1], &nxt_s);
      for (int tx = sx * MSK_SEG_TILES; tx < txe; ++tx) {
        const int tile = ty * tiles_x + tx;
        const int px = tx * MSK_TW + (lane & (MSK_TW - 1));
        const float x = (float)px + 0.5f;
        const int kofs = (tx - sx * MSK_SEG_TILES) * MSK_TW;
        const unsigned long long key = krow[kofs];
        krow[kofs] = 0ull;                           /* my next segment starts from an empty buffer */
        float best_w = __uint_as_float((unsigned)(key >> 32));
        int best_prim = 16383 - (int)(((unsigned)key) >> 16), best_slot = (int)(((unsigned)key) & 0xFFFFu);
        const int l0 = __builtin_amdgcn_readfirstlane(Lcnt[tile]), l1 = __builtin_amdgcn_readfirstlane(Lcnt[tile + 1]);
        const unsigned bigmask = (unsigned)__builtin_amdgcn_readfirstlane((int)Lmask[tile]);
        const unsigned covmask = (unsigned)__builtin_amdgcn_readfirstlane((int)Lcover[tile]);
        RecRegs cur = nxt;
        int cur_s = nxt_s;
        if (tx + 1 < txe) nxt = fetch_chunk(Lcnt[tile + 1], Lcnt[tile + 2], &nxt_s);
        if (!MSK_CUT_IS(7) && !MSK_CUT_IS(8)) {
          for (unsigned mk = bigmask & covmask; mk != 0u; mk &= mk - 1u) {
            const int j = __builtin_ctz(mk);
            MSK_SPLAT_LANE_W(big, big_s, j);
          }
          for (unsigned mk = bigmask & ~covmask; mk != 0u; mk &= mk - 1u) {
            const int j = __builtin_ctz(mk);
            MSK_SPLAT_LANE(big, big_s, j);
          }
        }
        if (!MSK_CUT_IS(8))
          for (int k0 = l0; k0 < l1; k0 += 64) {
            if (k0 > l0) cur = fetch_chunk(k0, l1, &cur_s);
            const int n = min(64, l1 - k0);
            for (int j = 0; j < n; ++j) MSK_SPLAT_LANE(cur, cur_s, j);
          }
        short4 o = make_short4(0, 0, 0, 0);
        unsigned best_col = 0u, texel = 0u;
        if (best_w > 0.0f) {
          const float4 rd = rec4(best_slot)[3];
          const float d = 1.0f / best_w;
          const float gz = -d;
          if (cam.want_tex) { /* (wave-uniform: a kernel argument) */
            const float gx = (x - cam.cx) / cam.fx * d, gy = gyc * d;
            o.x = (short)fminf(fmaxf(rintf(gx * 1000.0f), -32768.0f), 32767.0f);
            o.y = (short)fminf(fmaxf(rintf(gy * 1000.0f), -32768.0f), 32767.0f);
          }
          o.z = (short)fminf(fmaxf(rintf(gz * 1000.0f), -32768.0f), 32767.0f);
          o.w = (short)(__float_as_int(rd.x) & 0xFFFF);
          best_col = __float_as_uint(rd.w);
          const int us = (__float_as_int(rd.x) >> MSK_SEG_UV_SHIFT) & MSK_SEG_UV_MASK;
          if (us != 0) { /* the texel under the pixel centre: u = (u / depth) * depth */
            const float4 ua = ((const float4*)(Luv + (size_t)(us - 1) * 8))[0], ub = ((const float4*)(Luv + (size_t)(us - 1) * 8))[1];
            const float uu = fmaf(ua.x, x, fmaf(ua.y, y, ua.z)) * d, vv = fmaf(ua.w, x, fmaf(ub.x, y, ub.y)) * d;
            const int wh = __float_as_int(ub.z);
            const float4 rc = rec4(best_slot)[2];     /* (C2, Aw, Bw, Cw) */
            const float ux = fmaf(-uu, rc.y, ua.x) * d, uy = fmaf(-uu, rc.z, ua.y) * d;
            const float vx = fmaf(-vv, rc.y, ua.w) * d, vy = fmaf(-vv, rc.z, ub.x) * d;
            const float rho = fmaxf(fmaxf(fabsf(ux), fabsf(uy)) * (float)(wh & 0xFFFF), fmaxf(fabsf(vx), fabsf(vy)) * (float)(wh >> 16));
            texel = 1u + (unsigned)(__float_as_int(ub.w) + texel_index(wh & 0xFFFF, wh >> 16, uu, vv, rho));
          }
        }
        const size_t pix = rowpix + (size_t)tx * MSK_TW;
        if (MSK_CUT_IS(9) && best_prim != -7) continue;
        if (cam.want_tex) ((short4*)cam.out)[pix] = o;
        if (cam.color) cam.color[pix] = best_col;
        if (cam.uvt) cam.uvt[pix] = texel;
        cam.depth[pix] = (short)(-(int)o.z);
        cam.seg[pix] = o.w;
      }
      /* my zeroing stores above, my next segment's atomics below */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (ns > rcap) rows(std::true_type{});
  else rows(std::false_type{});
}

/* Color = texel * shade / 255 per channel (rounded) where k_render_splat named a texel: the texture fetch is a pass of its own because a
 * global LOAD inside the tile walk makes every tile wait for the previous tile's stores (see k_render_env's fetch_record). */
__global__ void __launch_bounds__(256) k_render_texture(const RModel* __restrict__ rm, unsigned* __restrict__ color, const unsigned* __restrict__ uvt, size_t npix) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const unsigned t = uvt[i];
  if (t == 0u) return;
  const unsigned texel = rm->texels[t - 1u], shade = color[i];
  unsigned out = shade & 0xFF000000u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const unsigned a = (texel >> (8 * ch)) & 0xFFu, b = (shade >> (8 * ch)) & 0xFFu;
    out |= ((a * b + 127u) / 255u) << (8 * ch);
  }
  color[i] = out;
}

#endif
