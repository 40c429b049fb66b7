/*
 * tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a CPU emulation of the slice of the HIP programming model that
 * maniskill_amd/csrc uses, so that the PRODUCT's kernel sources (msk_physx.hip and its headers, unmodified) can be compiled with g++ and
 * executed in the GPU-less build container: tests/test_hip_emulation.py runs them against the CPU oracle.
 *
 * What it models: a launch runs its workgroups one after the other; the work-items of a workgroup are cooperative fibers of one OS thread
 * (emu_runtime.cpp).  __syncthreads() is a workgroup barrier; every cross-lane operation (readlane, shuffles, ballot, DPP moves) and
 * __builtin_amdgcn_wave_barrier() is a rendezvous of the lanes of a wavefront that reach the SAME call site (MSK_WAVE_REJOIN(): of all live lanes) -- lanes that took another branch
 * are inactive for it, as under an exec mask.  "Shared" memory is thread_local storage (one workgroup at a time lives on the OS thread);
 * device memory is host memory; streams and events are immediate.
 *
 * What it does not model: concurrency between workgroups or wavefronts (no data race, no missing fence can show), the memory hierarchy, timing --
 * and the LOCKSTEP of a wavefront: a SIMT machine runs the two sides of a divergent branch one after the other, here the lanes of both sides
 * run interleaved.  Where lane groups that worked on items of their own continue as one wavefront the sources say MSK_WAVE_REJOIN() (msk_math.h:
 * a scheduling barrier on hardware, "every live lane" here: the hull items' sign-off), and where the groups that reached a region take turns
 * MSK_LANE_GROUP_TURN() (the same on hardware; here "the groups in the region", released once nothing else of the wavefront can move: the hull
 * queue's EPA turns).  With those, all 45 of the reference's download-free tasks step bit-equal to the oracle (FMBAssembly1Easy-v1: deep
 * contacts, many queue items per wavefront, 64 coordinates, over the contact capacity).  It checks arithmetic, indexing, the lane mappings and the lists / scans / masks the kernels build -- against the
 * oracle, bit for bit, in the configurations tests/test_hip_emulation.py runs.
 */
#ifndef MSK_HIPEMU_RUNTIME_H
#define MSK_HIPEMU_RUNTIME_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <type_traits>

#define MSK_HIP_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local
#define __restrict__ __restrict
#define address_space(n)            /* __attribute__((address_space(3))) -> __attribute__(()) */
#define MSK_WAIT_VMCNT0() ((void)0) /* the kernels' s_waitcnt vmcnt(0): a no-op here */
#define MSK_OPAQUE_VGPR(p) ((void)0) /* ... and their register-class hints */

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
struct alignas(8) short4 { short x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline short4 make_short4(short x, short y, short z, short w) { short4 r = {x, y, z, w}; return r; }
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }

/* ---- runtime API (immediate) ---------------------------------------------------------------------------------------------------------- */
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
  for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

/* ---- launches --------------------------------------------------------------------------------------------------------------------------- */
void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& work_item);
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu_launch((grid), (block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })
#define hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ev0, ev1, flags, ...) emu_launch((grid), (block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })

/* ---- synchronisation and cross-lane operations (emu_runtime.cpp) -------------------------------------------------------------------------- */
void emu_block_barrier();
/* rendezvous of the lanes of my wavefront that reach call site `site`: every participant hands in `v` and receives all 64 values and the mask of
 * participants (bit = lane of the wavefront) */
uint64_t emu_wave_gather(uint32_t v, uint32_t out[64], const void* site, int converge = 0);   /* converge: 0 a cross-lane operation, 1 the end of a turn of lane groups, 2 a rejoin of the whole wavefront */
#define EMU_SITE() ([]() __attribute__((noinline)) -> const void* { static const char tag = 0; return &tag; }())
static inline int emu_lane() { return (int)(threadIdx.x & 63u); }

#define __syncthreads() emu_block_barrier()
static inline void __threadfence_block() {}
static inline void __threadfence() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() do { uint32_t emu_o_[64]; (void)emu_wave_gather(0u, emu_o_, EMU_SITE()); } while (0)
#define MSK_WAVE_REJOIN() do { uint32_t emu_o_[64]; (void)emu_wave_gather(0u, emu_o_, EMU_SITE(), 2); } while (0)   /* msk_math.h: every live lane */
#define MSK_LANE_GROUP_TURN() do { uint32_t emu_o_[64]; (void)emu_wave_gather(0u, emu_o_, EMU_SITE(), 1); } while (0)   /* msk_math.h: the lane groups taking turns */
#define __builtin_amdgcn_s_barrier() emu_block_barrier()
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull

static inline uint32_t emu_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline uint32_t emu_bits(int x) { return (uint32_t)x; }
static inline uint32_t emu_bits(unsigned x) { return x; }
static inline uint32_t emu_bits(bool x) { return x ? 1u : 0u; }
template <typename T> static inline T emu_from(uint32_t u) { T r; if constexpr (std::is_same<T, bool>::value) r = u != 0; else memcpy(&r, &u, 4); return r; }

#define __builtin_amdgcn_readlane(v, j) emu_readlane((int)(v), (int)(j), EMU_SITE())
static inline int emu_readlane(int v, int j, const void* site) { uint32_t o[64]; (void)emu_wave_gather((uint32_t)v, o, site); return (int)o[j & 63]; }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((int)(v), EMU_SITE())
static inline int emu_readfirstlane(int v, const void* site) { uint32_t o[64]; const uint64_t m = emu_wave_gather((uint32_t)v, o, site); return (int)o[__builtin_ctzll(m)]; }

template <typename T> static inline T emu_shfl(T v, int src, int width, const void* site) {
  uint32_t o[64]; const uint64_t m = emu_wave_gather(emu_bits(v), o, site);
  const int lane = emu_lane(), base = lane & ~(width - 1), s = base + (src & (width - 1));
  return ((m >> s) & 1) ? emu_from<T>(o[s]) : v;
}
template <typename T> static inline T emu_shfl_up(T v, int d, int width, const void* site) {
  uint32_t o[64]; const uint64_t m = emu_wave_gather(emu_bits(v), o, site);
  const int lane = emu_lane(), base = lane & ~(width - 1), s = lane - d;
  return (s >= base && ((m >> s) & 1)) ? emu_from<T>(o[s]) : v;
}
template <typename T> static inline T emu_shfl_xor(T v, int mask, int width, const void* site) {
  uint32_t o[64]; const uint64_t m = emu_wave_gather(emu_bits(v), o, site);
  const int lane = emu_lane(), s = lane ^ mask;
  return ((s & ~(width - 1)) == (lane & ~(width - 1)) && ((m >> s) & 1)) ? emu_from<T>(o[s]) : v;
}
#define EMU_SHFL_ARGS_(v, a, w, ...) (v), (a), (w)
#define __shfl(...) emu_shfl(EMU_SHFL_ARGS_(__VA_ARGS__, 64, 64), EMU_SITE())
#define __shfl_up(...) emu_shfl_up(EMU_SHFL_ARGS_(__VA_ARGS__, 64, 64), EMU_SITE())
#define __shfl_xor(...) emu_shfl_xor(EMU_SHFL_ARGS_(__VA_ARGS__, 64, 64), EMU_SITE())
#define __ballot(p) emu_ballot((p) ? 1u : 0u, EMU_SITE())
static inline unsigned long long emu_ballot(uint32_t p, const void* site) {
  uint32_t o[64]; const uint64_t m = emu_wave_gather(p, o, site);
  unsigned long long r = 0;
  for (int i = 0; i < 64; ++i) if (((m >> i) & 1) && o[i]) r |= 1ull << i;
  return r;
}
#define __any(p) (__ballot(p) != 0ull)

/* v_mov_b32 dpp: the source lane of `lane` under dpp_ctrl (gfx9 / gfx90a encodings), -1 if the control reads outside the row */
static inline int emu_dpp_source(int lane, int ctrl) {
  const int row = lane & ~15, l = lane & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);       /* quad_perm */
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = l + (ctrl & 15); return s < 16 ? row + s : -1; }   /* row_shl */
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = l - (ctrl & 15); return s >= 0 ? row + s : -1; }   /* row_shr */
  if (ctrl >= 0x121 && ctrl <= 0x12F) return row + ((l - (ctrl & 15)) & 15);                        /* row_ror */
  if (ctrl == 0x140) return row + (15 - l);                                                         /* row_mirror */
  if (ctrl == 0x141) return row + ((l & 8) | (7 - (l & 7)));                                        /* row_half_mirror */
  if (ctrl >= 0x150 && ctrl <= 0x15F) return row + (ctrl & 15);                                     /* row_newbcast (gfx90a+) */
  fprintf(stderr, "hipemu: dpp_ctrl 0x%x is not modelled\n", ctrl);
  abort();
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
  emu_update_dpp((int)(old), (int)(src), (ctrl), (row_mask), (bank_mask), (bound_ctrl), EMU_SITE())
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, const void* site) {
  uint32_t o[64]; const uint64_t m = emu_wave_gather((uint32_t)src, o, site);
  const int lane = emu_lane();
  if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane & 15) >> 2)) & 1)) return old;
  const int s = emu_dpp_source(lane, ctrl);
  if (s < 0 || !((m >> s) & 1)) return bound_ctrl ? 0 : old;
  return (int)o[s];
}
#define __builtin_amdgcn_mov_dpp(src, ctrl, row_mask, bank_mask, bound_ctrl) __builtin_amdgcn_update_dpp(0, src, ctrl, row_mask, bank_mask, bound_ctrl)
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }

/* ---- scalar intrinsics ----------------------------------------------------------------------------------------------------------------- */
static inline float __int_as_float(int x) { float f; memcpy(&f, &x, 4); return f; }
static inline int __float_as_int(float x) { int i; memcpy(&i, &x, 4); return i; }
static inline unsigned __float_as_uint(float x) { unsigned i; memcpy(&i, &x, 4); return i; }
static inline float __uint_as_float(unsigned x) { float f; memcpy(&f, &x, 4); return f; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
using std::max;
using std::min;
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
static inline int max(int a, unsigned b) { return a > (int)b ? a : (int)b; }
static inline size_t min(size_t a, int b) { return a < (size_t)b ? a : (size_t)b; }

/* atomics: one fiber runs at a time, so plain read-modify-writes are atomic */
template <typename T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, unsigned v) { const int o = *p; *p = o + (int)v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { const T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
template <typename T, typename V> static inline T __hip_atomic_fetch_max(T* p, V v, int, int) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }

#endif
