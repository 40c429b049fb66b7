// Development aid: what does ONE step of a serial dependent chain cost on an MI355X compute unit when a SIMD holds one wavefront?
// The substep kernels are bound by such chains; this measures the floor their row update / level step can reach.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/chain_microbench tools/chain_microbench.hip && ./gpurun_out/chain_microbench
// One wavefront per workgroup, 256 workgroups (one per CU) or 2048 (two per SIMD): s_memtime around N iterations of each chain.
// Round 6: the body is unrolled U = 64 times per loop iteration (round 5 timed a non-unrolled loop: every iteration carried s_add / s_cmp and a TAKEN
// s_cbranch, and the tick count was divided by the VALU count alone -- "14 ticks per dependent v_fma" was mostly the loop), and an EMPTY loop of the same
// trip count is timed beside it and subtracted.  Printed: ticks per unrolled body, per dependent VALU instruction, and the same in ns from the launch time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xF, 0xF, false));
}

#define U 64

template <int KIND>
__global__ void __launch_bounds__(64) k_chain(float* out, long long* cyc, int n, float a, float b, float lo, float hi, int never) {
  float x = out[threadIdx.x], lam = 0.25f * threadIdx.x, r = 0.5f;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (KIND == 0) {            // 1 dependent fma
        x = fmaf(x, a, b);
      } else if (KIND == 1 || KIND == 3) {     // the packed solver class's row update: fma, med3, sub, dpp broadcast, fma -- 5 dependent (3: + a not-taken scalar branch per update)
        const float nl = __builtin_amdgcn_fmed3f(fmaf(-x, r, lam), lo, hi);
        const float dl = dpp_mov<0x153>(nl - lam);
        if ((threadIdx.x & 15) == 3) lam = nl;
        x = fmaf(a, dl, x);
        if (KIND == 3 && never == i * U + u) goto done;
      } else if (KIND == 2) {     // the one-env-per-wave classes: the broadcast is v_readlane
        const float nl = __builtin_amdgcn_fmed3f(fmaf(-x, r, lam), lo, hi);
        const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl - lam), 3));
        if (threadIdx.x == 3) lam = nl;
        x = fmaf(a, dl, x);
      } else if (KIND == 4) {     // a division and a square root on the chain (quat_normalize, Cholesky pivots): IEEE-correct sequences
        x = sqrtf(fabsf(x) + 1.0f) / (fabsf(b) + 2.0f);
      } else if (KIND == 5) {     // the same normalisation with the hardware approximations: v_rsq_f32 and one multiply
        x = (fabsf(x) + 1.0f) * __builtin_amdgcn_rsqf(fabsf(x) + 1.0f) * b;
      } else if (KIND == 6) {     // the row update with the clamp and the subtraction off the chain is not possible (they ARE the chain); this is the bare broadcast: fma, dpp, fma
        const float dl = dpp_mov<0x153>(fmaf(-x, r, lam));
        x = fmaf(a, dl, x);
      }
    }
    if (KIND == 7) asm volatile("" : "+v"(x));     // empty body: the loop's own cost (ONE opaque use per loop iteration: an asm per unrolled body is padded with an s_nop each)
  }
done:
  const long long t1 = (long long)__builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = x + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void launch(int grid, float* o, long long* c, int n) {
  hipLaunchKernelGGL(k_chain<KIND>, dim3(grid), dim3(64), 0, 0, o, c, n, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
}

int main() {
  const int WG = 256, N = 400;        // N loop iterations x U unrolled bodies
  const int NK = 8;
  const char* names[NK] = {"dependent v_fma", "fma med3 sub mov_dpp fma (packed row update: 5 dependent)", "fma med3 sub v_readlane fma (one env per wave: 5 dependent)",
                           "the packed row update + one not-taken scalar branch", "sqrtf + division (IEEE sequences)", "v_rsq_f32 normalise (add, rsq, mul, mul)",
                           "fma mov_dpp fma (3 dependent)", "empty body (the loop alone)"};
  const int per[NK] = {1, 5, 5, 5, 1, 4, 3, 1};
  for (int waves = 1; waves <= 2; ++waves) {      // 256 workgroups = one wavefront per CU; 2048 = two per SIMD
    const int grid = waves == 1 ? WG : WG * 8;
    float* o2; long long* c2;
    hipMalloc(&o2, grid * 64 * sizeof(float)); hipMalloc(&c2, grid * sizeof(long long)); hipMemset(o2, 0, grid * 64 * sizeof(float));
    std::vector<long long> h(grid);
    double ticks[NK], nsec[NK];
    for (int kind = NK - 1; kind >= 0; --kind) {      // (the empty loop first: it is subtracted from the others)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0.0f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        switch (kind) {
          case 0: launch<0>(grid, o2, c2, N); break; case 1: launch<1>(grid, o2, c2, N); break; case 2: launch<2>(grid, o2, c2, N); break;
          case 3: launch<3>(grid, o2, c2, N); break; case 4: launch<4>(grid, o2, c2, N); break; case 5: launch<5>(grid, o2, c2, N); break;
          case 6: launch<6>(grid, o2, c2, N); break; default: launch<7>(grid, o2, c2, N); break;
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(h.data(), c2, grid * sizeof(long long), hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
      ticks[kind] = s / grid / N;            // per loop iteration (U bodies)
      nsec[kind] = ms * 1e6 / N;
      const double body = (ticks[kind] - (kind == NK - 1 ? 0.0 : ticks[NK - 1])) / U;
      printf("grid %4d x 64 (%s): %-62s %7.2f counter ticks per body (loop subtracted), %5.2f per dependent instruction; launch %.1f us = %.2f ns per body\n", grid,
             waves == 1 ? "one wavefront per CU" : "two wavefronts per SIMD", names[kind], kind == NK - 1 ? ticks[kind] : body, kind == NK - 1 ? ticks[kind] : body / per[kind],
             ms * 1e3, nsec[kind] / U);
    }
    hipFree(o2); hipFree(c2);
  }
  return 0;
}
