// dev aid: VALU / transcendental issue rates on one SIMD with 1, 2, 4 resident wavefronts (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#define N 4096
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.01f + threadIdx.x * 1e-3f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
      else if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.25f;            // exp + mul
      else if (KIND == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);                     // exp only (dependent chain of 8 independent)
      else if (KIND == 3) { a[i] = __builtin_amdgcn_rcpf(a[i] + 1.0f); }           // add + rcp
      else if (KIND == 5) {     // packed fma: two values per instruction (a[i], a[i^1] pairs handled below)
      }
      else if (KIND == 4) { float e = __builtin_amdgcn_exp2f(a[i]); a[i] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f); }   // tanh
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND>
__global__ void k2(float* out, long long* cyc, float seed) {
  f32x2 a[8];
  f32x4 acc[4];
  bf16x8 A, B;
  for (int i = 0; i < 8; ++i) { a[i][0] = seed + i * 0.01f + threadIdx.x * 1e-3f; a[i][1] = a[i][0] + 0.5f; }
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
    if (KIND == 0) {                      // 8 packed fma
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], f32x2{1.0001f, 1.0002f}, f32x2{0.5f, 0.25f});
    } else if (KIND == 1) {               // 4 MFMA 16x16x32 only
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[i], 0, 0, 0);
    } else if (KIND == 2) {               // 4 MFMA + 8 fma
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[i], 0, 0, 0);
        a[2 * i][0] = __builtin_fmaf(a[2 * i][0], 1.0001f, 0.5f);
        a[2 * i + 1][0] = __builtin_fmaf(a[2 * i + 1][0], 1.0001f, 0.5f);
      }
    } else if (KIND == 3) {               // 4 MFMA + 8 exp
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[i], 0, 0, 0);
        a[2 * i][0] = __builtin_amdgcn_exp2f(a[2 * i][0]);
        a[2 * i + 1][0] = __builtin_amdgcn_exp2f(a[2 * i + 1][0]);
      }
    } else if (KIND == 4) {               // 8 log
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i][0] = __builtin_amdgcn_logf(a[i][0]) + 3.0f;
    } else if (KIND == 5) {               // 8 cvt_pk_bf16
#pragma unroll
      for (int i = 0; i < 8; ++i) { typedef __attribute__((ext_vector_type(2))) __bf16 b2; b2 v = {(__bf16)a[i][0], (__bf16)a[i][1]}; a[i][0] = __builtin_bit_cast(float, v) ; }
    } else if (KIND == 6) {               // 8 f64 add
#pragma unroll
      for (int i = 0; i < 8; ++i) { double d = __builtin_bit_cast(double, a[i]); d += 1.0; a[i] = __builtin_bit_cast(f32x2, d); }
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run2(const char* name) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int waves : {1, 2, 4}) {
    const int threads = waves * 4 * 64;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k2<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f); hipDeviceSynchronize(); }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s waves/SIMD %d: %.1f cycles per loop iteration per wave, %.1f per SIMD per wave-iteration\n", name, waves, (double)c / N, (double)c / N / waves);
  }
}
template <int KIND> void run(const char* name, int ops_per_iter) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int waves : {1, 2, 4, 8}) {                 // waves per SIMD = block waves / 4
    const int threads = waves * 4 * 64;
    if (threads > 1024) {                           // two blocks on the CU: approximate with 2 x 1024? skip
      continue;
    }
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-12s waves/SIMD %d: %.2f cycles per wave-instruction-slot (per SIMD: %.2f cycles per instruction)\n", name, waves,
           (double)c / (N * 8.0 * ops_per_iter), (double)c / (N * 8.0 * ops_per_iter * waves));
  }
}
__global__ void clk(long long* o) { long long a = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
  for (volatile int i = 0; i < 200000; ++i) {} long long b = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime(); o[0] = b - a; o[1] = r2 - r; }
int main() {
  { long long* o; hipMalloc(&o, 16); hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize(); long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per 100 MHz realtime tick: %.2f (=> memtime clock %.0f MHz)\n", (double)h[0] / h[1], 100.0 * h[0] / h[1]); }
  run2<0>("8 pk_fma"); run2<1>("4 mfma"); run2<2>("4 mfma + 8 fma"); run2<3>("4 mfma + 8 exp"); run2<4>("8 log(+add)"); run2<5>("8 cvt_pk"); run2<6>("8 add_f64");
  run<0>("fma", 1); run<2>("exp", 1); run<1>("exp+mul", 2); run<3>("add+rcp", 2); run<4>("tanh(4 ops)", 4);
  return 0;
}
