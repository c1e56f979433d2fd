// dev aid: do MFMA and VALU / transcendental instructions overlap on one SIMD (gfx950)?  Inline asm keeps the order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#define N 2048
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// KIND 0: 4 mfma | 1: 16 exp | 2: 4 x (mfma, 4 exp) interleaved, same wave | 3: 16 fma | 4: 4 x (mfma, 4 fma)
// KIND 8: waves 0..3 run 16 exp, waves 4..7 16 fma | 9: 8 x (exp, fma) | 10: 4 x (exp, 3 fma)
// KIND 5: waves 0..3 (one per SIMD) run 4 mfma, waves 4..7 run 16 exp  | 6: same with fma | 7: 4 x (mfma, 2 exp, 2 fma)
template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
  f32x4 acc[4]; bf16x8 A, B; float f[16];
  typedef __attribute__((ext_vector_type(2))) float f32x2; f32x2 p2[8];
  for (int i = 0; i < 8; ++i) p2[i] = f32x2{seed + i, seed - i};
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  for (int i = 0; i < 16; ++i) f[i] = seed + i * 0.01f;
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
#define MF(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B))
#define EX(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]))
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p2[i]))
#define FM(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]))
    if (KIND == 0 || (KIND >= 5 && KIND <= 6 && wave < 4)) { MF(0); MF(1); MF(2); MF(3); }
    else if (KIND == 1 || (KIND == 5 && wave >= 4)) { EX(0); EX(1); EX(2); EX(3); EX(4); EX(5); EX(6); EX(7); EX(8); EX(9); EX(10); EX(11); EX(12); EX(13); EX(14); EX(15); }
    else if (KIND == 2) { MF(0); EX(0); EX(1); EX(2); EX(3); MF(1); EX(4); EX(5); EX(6); EX(7); MF(2); EX(8); EX(9); EX(10); EX(11); MF(3); EX(12); EX(13); EX(14); EX(15); }
    else if (KIND == 3 || (KIND == 6 && wave >= 4)) { FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15); }
    else if (KIND == 4) { MF(0); FM(0); FM(1); FM(2); FM(3); MF(1); FM(4); FM(5); FM(6); FM(7); MF(2); FM(8); FM(9); FM(10); FM(11); MF(3); FM(12); FM(13); FM(14); FM(15); }
    else if (KIND == 8 && wave < 4) { EX(0); EX(1); EX(2); EX(3); EX(4); EX(5); EX(6); EX(7); EX(8); EX(9); EX(10); EX(11); EX(12); EX(13); EX(14); EX(15); }
    else if (KIND == 8) { FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15); }
    else if (KIND == 9) { EX(0); FM(1); EX(2); FM(3); EX(4); FM(5); EX(6); FM(7); EX(8); FM(9); EX(10); FM(11); EX(12); FM(13); EX(14); FM(15); }
    else if (KIND == 10) { EX(0); FM(1); FM(3); FM(5); EX(2); FM(7); FM(9); FM(11); EX(4); FM(13); FM(15); FM(1); EX(6); FM(3); FM(5); FM(7); }
    else if (KIND == 11) { PK(0); PK(1); PK(2); PK(3); PK(4); PK(5); PK(6); PK(7); PK(0); PK(1); PK(2); PK(3); PK(4); PK(5); PK(6); PK(7); }
    else if (KIND == 12) { EX(0); PK(1); PK(2); PK(3); EX(2); PK(4); PK(5); PK(6); EX(4); PK(7); PK(1); PK(2); EX(6); PK(3); PK(4); PK(5); }
    else if (KIND == 13) { EX(0); PK(1); EX(2); PK(2); EX(4); PK(3); EX(6); PK(4); EX(8); PK(5); EX(10); PK(6); EX(12); PK(7); EX(14); PK(1); }
    else if (KIND == 14) { MF(0); PK(0); PK(1); PK(2); PK(3); MF(1); PK(4); PK(5); PK(6); PK(7); MF(2); PK(0); PK(1); PK(2); PK(3); MF(3); PK(4); PK(5); PK(6); PK(7); }
    else if (KIND == 15) { EX(0); FM(1); PK(1); FM(3); EX(2); PK(2); FM(5); PK(3); EX(4); FM(7); PK(4); FM(9); EX(6); PK(5); FM(11); PK(6); }
    else if (KIND == 16) { for (int g = 0; g < 4; ++g) { MF(g); EX(0); FM(1); FM(3); FM(5); EX(2); FM(7); FM(9); FM(11); EX(4); FM(13); FM(15); FM(1); EX(6); FM(3); FM(5); FM(7); } }
    else if (KIND == 17) { for (int g = 0; g < 4; ++g) { MF(g); EX(0); PK(0); PK(1); EX(2); PK(2); EX(4); PK(3); PK(4); EX(6); PK(5); } }
    else if (KIND == 18) { for (int g = 0; g < 4; ++g) { EX(0); FM(1); FM(3); FM(5); EX(2); FM(7); FM(9); FM(11); EX(4); FM(13); FM(15); FM(1); EX(6); FM(3); FM(5); FM(7); } }
    else if (KIND == 19) { for (int g = 0; g < 4; ++g) { EX(0); PK(0); PK(1); EX(2); PK(2); EX(4); PK(3); PK(4); EX(6); PK(5); } }
    else if (KIND == 20) { for (int g = 0; g < 4; ++g) { MF(g); EX(0); EX(2); PK(0); PK(1); PK(2); EX(4); PK(3); PK(4); EX(6); PK(5); } }
    else if (KIND == 21) { for (int g = 0; g < 4; ++g) { MF(g); EX(0); EX(2); EX(4); PK(0); PK(1); PK(2); PK(3); PK(4); EX(6); PK(5); } }
    else if (KIND == 22) { for (int g = 0; g < 4; ++g) { MF(g); FM(1); FM(3); PK(0); PK(1); PK(2); EX(4); PK(3); PK(4); EX(6); PK(5); EX(0); EX(2); } }
    else if (KIND == 7) { MF(0); EX(0); FM(1); EX(2); FM(3); MF(1); EX(4); FM(5); EX(6); FM(7); MF(2); EX(8); FM(9); EX(10); FM(11); MF(3); EX(12); FM(13); EX(14); FM(15); }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 8; ++i) s += p2[i][0] + p2[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, int waves) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int threads = waves * 4 * 64;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.3f); hipDeviceSynchronize(); }
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD %d: %.1f ticks per loop iteration\n", name, waves, (double)c / N);
}
int main() {
  run<0>("4 mfma", 1); run<0>("4 mfma", 2);
  run<1>("16 exp", 1); run<1>("16 exp", 2);
  run<3>("16 fma", 1); run<3>("16 fma", 2);
  run<2>("4 x (mfma, 4 exp) one wave", 1); run<2>("4 x (mfma, 4 exp) two waves", 2);
  run<4>("4 x (mfma, 4 fma) one wave", 1); run<4>("4 x (mfma, 4 fma) two waves", 2);
  run<7>("4 x (mfma, 2 exp, 2 fma) one wave", 1); run<7>("4 x (mfma, 2 exp, 2 fma) two waves", 2);
  run<5>("wave A: 4 mfma | wave B: 16 exp", 2);
  run<6>("wave A: 4 mfma | wave B: 16 fma", 2);
  run<8>("wave A: 16 exp | wave B: 16 fma", 2);
  run<9>("8 x (exp, fma) one wave", 1); run<9>("8 x (exp, fma) two waves", 2);
  run<10>("4 x (exp, 3 fma) one wave", 1); run<10>("4 x (exp, 3 fma) two waves", 2); run<10>("4 x (exp, 3 fma) four waves", 4);
  run<11>("16 pk_fma", 1); run<11>("16 pk_fma", 2);
  run<12>("4 x (exp, 3 pk_fma) one wave", 1); run<12>("4 x (exp, 3 pk_fma) two waves", 2);
  run<13>("8 x (exp, pk_fma) one wave", 1); run<13>("8 x (exp, pk_fma) two waves", 2);
  run<14>("4 x (mfma, 4 pk_fma) one wave", 1); run<14>("4 x (mfma, 4 pk_fma) two waves", 2);
  run<15>("4 exp, 6 fma, 6 pk_fma mixed, one wave", 1); run<15>("4 exp, 6 fma, 6 pk_fma mixed, two waves", 2);
  run<16>("4 x (mfma, 4 exp, 12 fma) one wave", 1); run<16>("4 x (mfma, 4 exp, 12 fma) two waves", 2);
  run<17>("4 x (mfma, 4 exp, 6 pk_fma) one wave", 1); run<17>("4 x (mfma, 4 exp, 6 pk_fma) two waves", 2);
  run<18>("4 x (4 exp, 12 fma) one wave", 1); run<18>("4 x (4 exp, 12 fma) two waves", 2);
  run<19>("4 x (4 exp, 6 pk_fma) one wave", 1); run<19>("4 x (4 exp, 6 pk_fma) two waves", 2);
  run<20>("4 x (mfma, 2 exp, 3 pk, exp, 2 pk, exp, pk) one wave", 1); run<20>("same, two waves", 2);
  run<21>("4 x (mfma, 3 exp, 5 pk, exp, pk) one wave", 1); run<21>("same, two waves", 2);
  run<22>("4 x (mfma, 2 fma, 3 pk, exp, 2 pk, exp, pk, 2 exp) one wave", 1); run<22>("same, two waves", 2);
  return 0;
}
