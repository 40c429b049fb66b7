// Development aid, round 6: ONE number for the question "would MFMA shorten the solver's A = J W J^T build?" (BASELINE.json north_star names MFMA for the dense
// pieces; DESIGN 1 had declined by argument).  The build of a one-env-per-wavefront solver class is, per env, Y = W J^T (G x R) and A = J Y (R x R) with
// R = 3 rows x up to 64 blocks and G = 16..64 generalized coordinates; here the dense core of it, A = J Y for R = 64 rows, G = 16 coordinates, fp32:
//   VALU form (what k_csolve does, densely): lane = row r; A[r][c] = sum_k J[r][k] Y[k][c] with J[r][:] in registers and Y read from LDS as broadcasts
//   MFMA form: sixteen 16 x 16 output tiles, four v_mfma_f32_16x16x4_f32 each (K = 16), operands fetched from LDS in the instruction's own layout
// Both write A to LDS in the solver's layout (row-major by lane) so that neither gets its result for free.  One wavefront per workgroup, 256 workgroups;
// s_memtime around N repetitions; printed: cycles per build, the ratio, and the largest |difference| between the two results (MFMA accumulates in another order).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_ab tools/mfma_ab.hip && ./gpurun_out/mfma_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define R 64
#define G 16
typedef float float4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(64) k_build(const float* __restrict__ Jg, const float* __restrict__ Yg, float* __restrict__ out, long long* cyc, int n) {
  __shared__ __attribute__((aligned(16))) float Jl[R * G];        // J[r][k]
  __shared__ __attribute__((aligned(16))) float Yl[G * R];        // Y[k][c]
  __shared__ __attribute__((aligned(16))) float Al[R * (R + 1)];  // A[r][c], odd stride
  const int l = threadIdx.x;
  for (int i = l; i < R * G; i += 64) { Jl[i] = Jg[i]; Yl[i] = Yg[i]; }
  __syncthreads();
  float sink = 0.0f;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
    if (KIND == 0) {
      float Jr[G];
#pragma unroll
      for (int k = 0; k < G; ++k) Jr[k] = Jl[l * G + k];
#pragma unroll 4
      for (int c = 0; c < R; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < G; ++k) acc = fmaf(Jr[k], Yl[k * R + c], acc);
        Al[l * (R + 1) + c] = acc;
      }
    } else {
      const int i16 = l & 15, q = l >> 4;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        float a[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = Jl[(ti * 16 + i16) * G + kk * 4 + q];      // A operand: lane holds A[i = l % 16][k = l / 16]
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          float4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const float b = Yl[(kk * 4 + q) * R + tj * 16 + i16];                           // B operand: lane holds B[k = l / 16][j = l % 16]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b, acc, 0, 0, 0);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) Al[(ti * 16 + 4 * q + v) * (R + 1) + tj * 16 + i16] = acc[v];   // D: lane holds D[i = 4 (l / 16) + v][j = l % 16]
        }
      }
    }
    __syncthreads();
    sink += Al[l * (R + 1) + ((it + l) & 63)];
    asm volatile("" ::: "memory");
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
  for (int c = 0; c < R; ++c) out[((size_t)blockIdx.x * R + l) * R + c] = Al[l * (R + 1) + c];
  if (sink == 123.456f) out[0] = sink;
}

int main() {
  const int grid = 256, n = 200;
  std::vector<float> J(R * G), Y(G * R);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : J) v = rnd();
  for (auto& v : Y) v = rnd();
  float *dJ, *dY, *dO[2];
  long long* dC;
  hipMalloc(&dJ, sizeof(float) * R * G); hipMalloc(&dY, sizeof(float) * R * G); hipMalloc(&dC, sizeof(long long) * grid);
  for (int k = 0; k < 2; ++k) hipMalloc(&dO[k], sizeof(float) * (size_t)grid * R * R);
  hipMemcpy(dJ, J.data(), sizeof(float) * R * G, hipMemcpyHostToDevice);
  hipMemcpy(dY, Y.data(), sizeof(float) * R * G, hipMemcpyHostToDevice);
  double cycles[2];
  std::vector<float> res[2];
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 3; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(k_build<0>, dim3(grid), dim3(64), 0, 0, dJ, dY, dO[0], dC, n);
      else hipLaunchKernelGGL(k_build<1>, dim3(grid), dim3(64), 0, 0, dJ, dY, dO[1], dC, n);
      hipDeviceSynchronize();
    }
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), dC, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : c) sum += (double)v;
    cycles[kind] = sum / grid / n;
    res[kind].resize((size_t)R * R);
    hipMemcpy(res[kind].data(), dO[kind], sizeof(float) * R * R, hipMemcpyDeviceToHost);
  }
  double worst = 0, ref = 0;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < R; ++c) {
      double e = 0;
      for (int k = 0; k < G; ++k) e += (double)J[r * G + k] * (double)Y[k * R + c];
      worst = fmax(worst, fabs((double)res[0][r * R + c] - (double)res[1][r * R + c]));
      ref = fmax(ref, fmax(fabs(res[0][r * R + c] - e), fabs(res[1][r * R + c] - e)));
    }
  printf("A = J Y, J 64 x 16, Y 16 x 64, fp32, one wavefront per CU, counter ticks per build (incl. the store of A to LDS and one barrier):\n");
  printf("  VALU form (lane = row, Y broadcast from LDS)      %8.0f\n", cycles[0]);
  printf("  MFMA form (16 tiles x 4 v_mfma_f32_16x16x4_f32)  %8.0f   = %.2f x\n", cycles[1], cycles[0] / cycles[1]);
  printf("  largest |VALU - MFMA| over the 4096 entries %.3g; largest error of either against float64 %.3g (entries are O(1))\n", worst, ref);
  return 0;
}
