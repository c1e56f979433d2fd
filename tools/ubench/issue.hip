// dev aid (round 3): VALU issue cost on one SIMD of gfx950 by instruction mix AND by the number of resident wavefronts
// (1, 2, 4 in one workgroup; 8 = two 1024-thread workgroups per CU, grid = 2 x CUs so that every CU holds two).
// Inline asm keeps the order; 16 independent registers per kind, so no dependent instruction is closer than 16 slots.
// Prints elapsed ticks (s_memtime of workgroup 0) per loop iteration AND per wavefront-iteration (elapsed / waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#define N 1024
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define EX(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]))
#define FM(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]))
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p2[i]))
#define MF(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B))
#define D2(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]))
#define CV(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(f[i]), "v"(f[(i + 1) & 15]))
#define SH(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[i]) : "v"(u[i]))
#define R16(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11); M(12); M(13); M(14); M(15)
#define R8(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7)
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
  f32x4 acc[4]; bf16x8 A, B; float f[16]; f32x2 p2[8]; unsigned u[16];
  for (int i = 0; i < 8; ++i) p2[i] = f32x2{seed + i, seed - i};
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed + i); B[i] = (__bf16)(seed - i); }
  for (int i = 0; i < 16; ++i) { f[i] = seed + i * 0.01f; u[i] = 0x3f803f80u + i; }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; ++it) {
    if (KIND == 0) { R16(FM); R16(FM); R16(FM); R16(FM); }                       // 64 fma
    else if (KIND == 1) { R16(EX); }                                             // 16 exp
    else if (KIND == 2) {                                                        // 16 x (exp, 3 fma) interleaved
#define G2(i) EX(i); FM((i + 5) & 15); FM((i + 9) & 15); FM((i + 13) & 15)
      R16(G2); }
    else if (KIND == 3) { R16(EX); R16(FM); R16(FM); R16(FM); }                   // 16 exp, then 48 fma (blocked)
    else if (KIND == 4) { R8(PK); R8(PK); R8(PK); R8(PK); }                       // 32 pk_fma
    else if (KIND == 5) {                                                        // 16 x (exp, 2 pk) interleaved
#define G5(i) EX(i); PK(i & 7); PK((i + 3) & 7)
      R16(G5); }
    else if (KIND == 6) { R16(EX); R8(PK); R8(PK); R8(PK); R8(PK); }              // 16 exp, then 32 pk (blocked)
    else if (KIND == 7) { R16(D2); R16(D2); }                                     // 32 dot2_f32_bf16
    else if (KIND == 8) { R16(CV); R16(CV); }                                     // 32 cvt_pk_bf16_f32
    else if (KIND == 9) { R16(SH); R16(SH); }                                     // 32 shifts
    else if (KIND == 10) {                                                       // chain-like: 4 x (mfma, 4 exp, 3 pk, 2 fma)
#define G10(g) MF(g); EX(4 * g); EX(4 * g + 1); PK(g); EX(4 * g + 2); PK(g + 4); EX(4 * g + 3); PK((g + 2) & 7); FM((4 * g + 8) & 15); FM((4 * g + 9) & 15)
      G10(0); G10(1); G10(2); G10(3); }
    else if (KIND == 11) {                                                       // same without the mfma
#define G11(g) EX(4 * g); EX(4 * g + 1); PK(g); EX(4 * g + 2); PK(g + 4); EX(4 * g + 3); PK((g + 2) & 7); FM((4 * g + 8) & 15); FM((4 * g + 9) & 15)
      G11(0); G11(1); G11(2); G11(3); }
    else if (KIND == 12) {                                                       // hg-like: 8 x (2 shift, pk, 2 rcp(exp), 2 pk)
#define G12(g) SH(2 * g); SH(2 * g + 1); PK(g); EX(2 * g); EX(2 * g + 1); PK((g + 3) & 7); PK((g + 5) & 7)
      R8(G12); }
    else if (KIND == 13) { R16(EX); R16(SH); R8(PK); R8(PK); R8(PK); }            // the same instructions, blocked by class
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += f[i] + (float)u[i];
  for (int i = 0; i < 8; ++i) s += p2[i][0] + p2[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
float* g_out; long long* g_cyc;
template <int KIND> void run(const char* name) {
  const int ws[4] = {1, 2, 4, 8};
  printf("%-52s", name);
  for (int wi = 0; wi < 4; ++wi) {
    const int w = ws[wi];
    const int threads = (w == 8 ? 4 : w) * 256, grid = w == 8 ? 512 : 1;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, g_out, g_cyc, 0.3f); hipDeviceSynchronize(); }
    long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("  w%d: %7.1f (%6.1f/wave)", w, (double)c / N, (double)c / N / w);
  }
  printf("\n");
}
int main() {
  hipMalloc(&g_out, 512 * 1024 * 4); hipMalloc(&g_cyc, 8);
  printf("ticks per loop iteration, elapsed (and / waves per SIMD)\n");
  run<0>("64 fma");
  run<1>("16 exp");
  run<2>("16 x (exp, 3 fma) interleaved");
  run<3>("16 exp then 48 fma (blocked)");
  run<4>("32 pk_fma");
  run<5>("16 x (exp, 2 pk) interleaved");
  run<6>("16 exp then 32 pk (blocked)");
  run<7>("32 dot2_f32_bf16");
  run<8>("32 cvt_pk_bf16_f32");
  run<9>("32 lshlrev");
  run<10>("4 x (mfma, 4 exp, 3 pk, 2 fma)");
  run<11>("4 x (4 exp, 3 pk, 2 fma)");
  run<12>("8 x (2 shift, pk, 2 exp, 2 pk) interleaved");
  run<13>("16 exp, 16 shift, 24 pk blocked");
  return 0;
}
