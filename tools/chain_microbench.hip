// Development aid (round 5): what does ONE step of a serial dependent chain cost on an MI355X compute unit when a SIMD holds one wavefront?
// The substep kernels are bound by such chains; this measures the floor their row update / level step can reach.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/chain_microbench tools/chain_microbench.hip && ./gpurun_out/chain_microbench
// One wavefront per workgroup, 256 workgroups (one per CU): s_memtime around N iterations of each chain; prints cycles per dependent instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <int KIND>
__global__ void __launch_bounds__(64) k_chain(float* out, long long* cyc, int n, float a, float b, float lo, float hi, int never) {
  float x = out[threadIdx.x], lam = 0.25f * threadIdx.x, r = 0.5f;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {            // 4 dependent fma
      x = fmaf(x, a, b); x = fmaf(x, a, b); x = fmaf(x, a, b); x = fmaf(x, a, b);
    } else if (KIND == 1 || KIND == 3) {     // the packed solver class's row update: fma, med3, sub, dpp broadcast, fma -- 5 dependent (3: + a not-taken scalar branch)
      const float nl = __builtin_amdgcn_fmed3f(fmaf(-x, r, lam), lo, hi);
      const float dl = dpp_mov<0x153>(nl - lam);
      if ((threadIdx.x & 15) == 3) lam = nl;
      x = fmaf(a, dl, x);
      if (KIND == 3 && never == i) break;
    } else if (KIND == 2) {     // the one-env-per-wave classes: the broadcast is v_readlane
      const float nl = __builtin_amdgcn_fmed3f(fmaf(-x, r, lam), lo, hi);
      const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl - lam), 3));
      if (threadIdx.x == 3) lam = nl;
      x = fmaf(a, dl, x);
    } else if (KIND == 4) {     // a division and a square root on the chain (quat_normalize, Cholesky pivots): IEEE-correct sequences
      x = sqrtf(fabsf(x) + 1.0f) / (fabsf(b) + 2.0f);
    }
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = x + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int WG = 256, N = 20000;
  float* out; long long* cyc;
  hipMalloc(&out, WG * 64 * sizeof(float)); hipMalloc(&cyc, WG * sizeof(long long));
  hipMemset(out, 0, WG * 64 * sizeof(float));
  std::vector<long long> h(WG);
  const char* names[5] = {"4 dependent v_fma", "fma med3 sub mov_dpp fma (packed row update: 5 dependent)", "fma med3 sub v_readlane fma (one env per wave: 5 dependent)",
                          "the packed row update + one not-taken scalar branch", "sqrtf + division (IEEE sequences)"};
  const int per[5] = {4, 5, 5, 5, 1};
  for (int waves = 1; waves <= 2; ++waves) {      // 256 workgroups = one wavefront per CU; 2048 = two per SIMD
    const int grid = waves == 1 ? WG : WG * 8;
    float* o2 = out; long long* c2 = cyc;
    if (grid > WG) { hipMalloc(&o2, grid * 64 * sizeof(float)); hipMalloc(&c2, grid * sizeof(long long)); hipMemset(o2, 0, grid * 64 * sizeof(float)); h.resize(grid); }
    for (int kind = 0; kind < 5; ++kind) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0.0f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(grid), dim3(64), 0, 0, o2, c2, N, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
        if (kind == 1) hipLaunchKernelGGL(k_chain<1>, dim3(grid), dim3(64), 0, 0, o2, c2, N, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
        if (kind == 2) hipLaunchKernelGGL(k_chain<2>, dim3(grid), dim3(64), 0, 0, o2, c2, N, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
        if (kind == 3) hipLaunchKernelGGL(k_chain<3>, dim3(grid), dim3(64), 0, 0, o2, c2, N, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
        if (kind == 4) hipLaunchKernelGGL(k_chain<4>, dim3(grid), dim3(64), 0, 0, o2, c2, N, 0.999f, 1e-3f, -1.0f, 1.0f, -1);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(h.data(), c2, grid * sizeof(long long), hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
      const double per_iter = s / grid / N;
      printf("grid %4d x 64 (%s): %-62s %7.1f counter ticks per iteration, %5.1f per dependent instruction; launch %.1f us = %.2f ns per iteration\n", grid,
             waves == 1 ? "one wavefront per CU" : "two wavefronts per SIMD", names[kind], per_iter, per_iter / per[kind], ms * 1e3, ms * 1e6 / N);
    }
  }
  return 0;
}
