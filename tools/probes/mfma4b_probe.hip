// Layout check of v_mfma_f32_16x16x1_4b_f32 (gfx950): 4 independent 16x16 outer products per instruction.
// Model under test: block b = lane / 16; A_b[i = lane % 16], B_b[j = lane % 16]; D_b[i][j] in VGPR 4 b + (i % 4) of lane 16 (i / 4) + j.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma4b_probe.bin tools/probes/mfma4b_probe.hip && tools/probes/mfma4b_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* a, const float* b, float* d) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[64 * 16], *da, *db, *dd;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 4; ++blk)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const float want = ha[16 * blk + i] * hb[16 * blk + j];
        const float got = hd[(16 * (i / 4) + j) * 16 + 4 * blk + (i % 4)];
        if (fabsf(want - got) > 1e-3f * fabsf(want)) { if (bad < 5) printf("mismatch b%d i%d j%d want %g got %g\n", blk, i, j, want, got); ++bad; }
      }
  printf("mfma_f32_16x16x1_4b layout model: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return bad ? 1 : 0;
}
