// dev aid: checks the DPP / permlane-swap reductions of csrc/common.h on the GPU
#include <cstdio>
#include "../guided-vae-nmf_amd/csrc/common.h"
void vaenmf_set_error(const char*, ...) {}
__global__ void k(float* o16, float* o4, double* od) {
  const int l = threadIdx.x;
  const float v = (float)(1 << (l & 15)) + 0.001f * (l >> 4);
  o16[l] = sum_row16(v);
  const float u = (float)(1 << (l >> 4)) * 100 + (l & 15);
  o4[l] = sum_rows4(u);
  od[l] = sum_rows4_d((double)u * 1.000001);
}
int main() {
  float *a, *b; double* c;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c);
  float ha[64], hb[64]; double hc[64];
  hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    float e16 = 65535.0f + 16 * 0.001f * (l >> 4);
    float e4 = 1500.0f + 4 * (l & 15);
    double ed = (1500.0 + 4 * (l & 15)) * 1.000001;
    if (fabsf(ha[l] - e16) > 0.01f || fabsf(hb[l] - e4) > 1e-3f || fabs(hc[l] - ed) > 1e-6) { ++bad; if (bad < 8) printf("lane %d: %f (%f) %f (%f) %f (%f)\n", l, ha[l], e16, hb[l], e4, hc[l], ed); }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
