// dev aid: f16 transcendental / packed f16 rates vs f32 on one SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#define N 4096
typedef _Float16 h1;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
  h1 a[8]; h2 p[8]; float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = (h1)(seed + i * 0.01f + threadIdx.x * 1e-3f); p[i] = h2{a[i], a[i]}; f[i] = (float)a[i]; }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      else if (KIND == 1) asm volatile("v_rcp_f16 %0, %0" : "+v"(a[i]));
      else if (KIND == 2) asm volatile("v_log_f16 %0, %0" : "+v"(a[i]));
      else if (KIND == 3) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(p[i]));
      else if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      else if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      else if (KIND == 6) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      else if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      else if (KIND == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
      else if (KIND == 9) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
      else if (KIND == 10) asm volatile("v_sin_f32 %0, %0" : "+v"(f[i]));
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += (float)a[i] + (float)p[i][0] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int waves : {1, 2, 4}) {
    const int threads = waves * 4 * 64;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f); hipDeviceSynchronize(); }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-12s waves/SIMD %d: %.2f cycles per instruction per SIMD\n", name, waves, (double)c / (N * 8.0 * waves));
  }
}
int main() {
  run<0>("v_exp_f16"); run<1>("v_rcp_f16"); run<2>("v_log_f16"); run<3>("v_pk_fma_f16"); run<4>("v_exp_f32"); run<5>("v_rcp_f32"); run<6>("v_log_f32");
  run<7>("v_fma_f32"); run<8>("v_sqrt_f32"); run<9>("v_rsq_f32"); run<10>("v_sin_f32");
  return 0;
}
