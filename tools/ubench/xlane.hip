// dev aid: cost of the cross-lane forms used for the sum over the four 16-lane rows (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#define N 2048
template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
  unsigned a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 17 + i; b[i] = a[i] ^ 0x55; }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
      else if (KIND == 1) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
      else if (KIND == 2) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      else if (KIND == 3) a[i] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 16) * 4, a[i]) + 1;
      else if (KIND == 4) asm volatile("v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
      else if (KIND == 5) { asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i])); }   // dependent chain
      else if (KIND == 6) { asm volatile("v_add_f64 %0, %0, %0" : "+v"(*(double*)&a[i & 6])); }
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int waves : {1, 2}) {
    const int threads = waves * 4 * 64;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f); hipDeviceSynchronize(); }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %.2f ticks per instruction per SIMD\n", name, waves, (double)c / (N * 8.0 * waves));
  }
}
int main() {
  run<0>("v_permlane16_swap"); run<1>("v_permlane32_swap"); run<2>("v_add_u32_dpp quad_perm"); run<3>("ds_bpermute + add"); run<4>("v_add_u32_dpp row_bcast:15");
  run<5>("permlane16_swap + add (pair)"); run<6>("v_add_f64 (4 chains)");
  return 0;
}
