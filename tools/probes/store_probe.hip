// Memory-pattern probe for the epilogue of the 1x1 convolution kernels (not part of the library):
//   what does the chip give for "every workgroup reads a contiguous 128-row tile and writes 128 row pieces of 256 B at a row stride"?
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_probe.hip -o /tmp/store_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// stand-in for the tile's MFMA work: NB back-to-back v_mfma_f32_32x32x2_f32 (64 cycles each) that depend on the loaded data
template <int NB>
__device__ __forceinline__ float burn(float a, float b) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll 1
  for (int k = 0; k < NB / 2; ++k) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
  }
  return acc0[0] + acc1[3];
}

// mode 5: one tile per workgroup: load tile -> MFMA burn -> store pieces  (the shape of the conv kernel's life)
// mode 6: persistent, `per` tiles per workgroup, software-pipelined: [loads of tile t+1] [burn t] [stores t]
template <int MODE, int NB>
__global__ __launch_bounds__(512, 4) void probe_mfma(const float* __restrict__ x, float* __restrict__ y, int ldx, int ldy, int ntiles_n, int tiles, int per) {
  const int tid = threadIdx.x;
  const int cc = (tid & 15) * 4, row0 = tid >> 4;
  if (MODE == 5) {
    const int t = blockIdx.x;
    const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
    f32x4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt * 128 + row0 + 32 * i) * ldx + cc);
    const float e = burn<NB>(v[0][0], v[1][1]) * 1e-30f;
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f + e;
    return;
  }
  // strided tile assignment (t = b, b + G, ...) keeps neighbouring workgroups on neighbouring tiles at any time
  const int G = gridDim.x;
  int t = blockIdx.x;
  f32x4 v[4], nv[4];
  {
    const int mt = t / ntiles_n;
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt * 128 + row0 + 32 * i) * ldx + cc);
  }
  for (; t < tiles; t += G) {
    const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
    const int t2 = t + G < tiles ? t + G : t;
    const int mt2 = t2 / ntiles_n;
    for (int i = 0; i < 4; ++i) nv[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt2 * 128 + row0 + 32 * i) * ldx + cc);
    const float e = burn<NB>(v[0][0], v[1][1]) * 1e-30f;
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f + e;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the 4 loads of the next tile (older than the 4 stores) have returned
    for (int i = 0; i < 4; ++i) v[i] = nv[i];
  }
}

// mode 0: store only (strided pieces)   1: load tile + store pieces   2: store only, one wave = 1 KB contiguous (row-major full rows)
// mode 3: like 1, persistent over `per` tiles with the next tile's loads issued before this tile's stores
template <int MODE>
__global__ __launch_bounds__(512, 4) void probe(const float* __restrict__ x, float* __restrict__ y, int ldx, int ldy, int ntiles_n, int tiles, int per) {
  const int tid = threadIdx.x;
  const int cc = (tid & 15) * 4, row0 = tid >> 4;
  if (MODE == 3) {
    int t = blockIdx.x * per;
    f32x4 v[4], nv[4];
    {
      const int mt = t / ntiles_n;
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt * 128 + row0 + 32 * i) * ldx + cc);
    }
    for (int k = 0; k < per; ++k, ++t) {
      if (t >= tiles) break;
      const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
      if (k + 1 < per && t + 1 < tiles) {
        const int mt2 = (t + 1) / ntiles_n;
        for (int i = 0; i < 4; ++i) nv[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt2 * 128 + row0 + 32 * i) * ldx + cc);
      }
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f;
      for (int i = 0; i < 4; ++i) v[i] = nv[i];
    }
    return;
  }
  const int t = blockIdx.x;
  const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
  f32x4 v[4];
  for (int i = 0; i < 4; ++i) v[i] = (f32x4){1.f * tid, 2.f, 3.f, 4.f + i};
  if (MODE == 1) {
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(mt * 128 + row0 + 32 * i) * ldx + cc);
  }
  if (MODE == 2) {   // the tile's bytes as one contiguous 32 KB block
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)t * 8192 + (i * 512 + tid) * 4) = v[i];
    return;
  }
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f;
}

// mode 9+: persistent, LDS-DMA loads D tiles ahead in a ring of D + 1 buffers, explicit counted waits: the stores of tile t stay in flight
// through the loads / MFMAs of the following tiles (the vector-memory counter is in issue order)
template <int NB, int D>
__global__ __launch_bounds__(512, 2) void probe_dma(const float* __restrict__ x, float* __restrict__ y, int ldx, int ldy, int ntiles_n, int tiles, unsigned xbytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [D + 1][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = (tid & 15) * 4, row0 = tid >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, xbytes, 0x00020000);
  const int G = gridDim.x;
  int t = blockIdx.x;
  auto dma = [&](int tt, int b) {   // the tile = 32 KB contiguous: wave w moves pieces w*4 .. w*4+3 (1 KB each); past the end: zero-fill pieces
    const int mt = tt / ntiles_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = tt < tiles ? (unsigned)(mt * 128 * 64 * 4) + (unsigned)((wave * 4 + i) * 1024 + lane * 16) : 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(smem + b * 8192 + (wave * 4 + i) * 256), 16, off, 0, 0, 0);
#endif
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) dma(t + d * G, d);
  int buf = 0, nbuf = D;
  for (; t < tiles; t += G) {
    const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
    dma(t + D * G, nbuf);
    // outstanding, oldest first: dma(t) | stores(t - 1) | dma(t+1) ... dma(t+D): everything newer than dma(t) may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + 4 * D) : "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(smem + buf * 8192 + (row0 + 32 * i) * 64 + cc);
    const float e = burn<NB>(v[0][0], v[1][1]) * 1e-30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f + e;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone has read `buf` before a later DMA overwrites it
    buf = buf == D ? 0 : buf + 1;
    nbuf = nbuf == D ? 0 : nbuf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// warp-specialised form: wave 8 only issues the LDS-DMA loads (D tiles ahead) and waits for them with ITS OWN vector-memory counter;
// waves 0-7 read LDS, run the MFMAs and store — their stores never stand in front of a load in any counter, so they drain whenever
// the memory system gets to them.  One s_barrier per tile couples the two roles.
template <int NB, int D>
__global__ __launch_bounds__(640, 1) void probe_ws(const float* __restrict__ x, float* __restrict__ y, int ldx, int ldy, int ntiles_n, int tiles, unsigned xbytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [D + 1][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = (tid & 15) * 4, row0 = tid >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, xbytes, 0x00020000);
  const int G = gridDim.x;
  int t = blockIdx.x;
  auto dma = [&](int tt, int b) {   // the two loader waves move 16 of the tile's 32 pieces (1 KB each) each
    const int mt = tt / ntiles_n;
#pragma unroll 8
    for (int i = (wave - 8) * 16; i < (wave - 8) * 16 + 16; ++i) {
      const unsigned off = tt < tiles ? (unsigned)(mt * 128 * 64 * 4) + (unsigned)(i * 1024 + lane * 16) : 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(smem + b * 8192 + i * 256), 16, off, 0, 0, 0);
#endif
    }
  };
  int buf = 0, nbuf = D;
  if (wave >= 8) {
    for (int d = 0; d < D; ++d) dma(t + d * G, d);
    for (; t < tiles; t += G) {
      dma(t + D * G, nbuf);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * D) : "memory");   // tile t has landed
      __builtin_amdgcn_s_barrier();                                     // -> the compute waves may read it
      __builtin_amdgcn_s_barrier();                                     // <- they have read it: its slot may be refilled
      nbuf = nbuf == D ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (; t < tiles; t += G) {
    const int mt = t / ntiles_n, nt = t - mt * ntiles_n;
    __builtin_amdgcn_s_barrier();
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(smem + buf * 8192 + (row0 + 32 * i) * 64 + cc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float e = burn<NB>(v[0][0], v[1][1]) * 1e-30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + (size_t)(mt * 128 + row0 + 32 * i) * ldy + nt * 64 + cc) = v[i] * 1.0001f + e;
    buf = buf == D ? 0 : buf + 1;
  }
}

int main(int argc, char** argv) {
  const int M = 8 * 128 * 128, reps = 20;
  const int couts[3] = {64, 192, 256};
  hipStream_t st; hipStreamCreate(&st);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
  for (int ci = 0; ci < 3; ++ci) {
    const int Cout = couts[ci], ntn = Cout / 64, tiles = (M / 128) * ntn;
    const size_t xb = (size_t)M * 64 * 4, yb = (size_t)M * Cout * 4;
    const int nbuf = (int)(800e6 / (xb + yb)) + 2;
    std::vector<float*> xs(nbuf), ys(nbuf);
    for (int i = 0; i < nbuf; ++i) { hipMalloc(&xs[i], xb); hipMalloc(&ys[i], yb); hipMemset(xs[i], 0, xb); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 21; ++mode) {
      if (mode == 0 || mode == 2 || mode == 3 || mode == 4 || mode == 6 || mode == 7) continue;
      float best = 1e9;
      for (int rnd = 0; rnd < 4; ++rnd) {
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) {
          const float* x = xs[r % nbuf]; float* y = ys[r % nbuf];
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(tiles), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(tiles), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(tiles), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3((tiles + 3) / 4), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 4);
          if (mode == 4) hipMemcpyAsync(y, x, xb, hipMemcpyDeviceToDevice, st);
          if (mode == 5) hipLaunchKernelGGL((probe_mfma<5, 32>), dim3(tiles), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 6) hipLaunchKernelGGL((probe_mfma<6, 32>), dim3(768), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 7) hipLaunchKernelGGL((probe_mfma<6, 32>), dim3(512), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 8) hipLaunchKernelGGL((probe_mfma<5, 2>), dim3(tiles), dim3(512), 0, st, x, y, 64, Cout, ntn, tiles, 1);
          if (mode == 9) hipLaunchKernelGGL((probe_dma<32, 1>), dim3(512), dim3(512), 2 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 10) hipLaunchKernelGGL((probe_dma<32, 1>), dim3(256), dim3(512), 2 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 11) hipLaunchKernelGGL((probe_dma<32, 2>), dim3(512), dim3(512), 3 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 12) hipLaunchKernelGGL((probe_dma<32, 2>), dim3(256), dim3(512), 3 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 13) hipLaunchKernelGGL((probe_dma<32, 3>), dim3(256), dim3(512), 4 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 14) hipLaunchKernelGGL((probe_dma<2, 2>), dim3(256), dim3(512), 3 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 15) hipLaunchKernelGGL((probe_dma<32, 1>), dim3(768), dim3(512), 2 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 16) hipLaunchKernelGGL((probe_ws<32, 1>), dim3(256), dim3(640), 2 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 17) hipLaunchKernelGGL((probe_ws<32, 2>), dim3(256), dim3(640), 3 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 18) hipLaunchKernelGGL((probe_ws<32, 1>), dim3(512), dim3(640), 2 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 19) hipLaunchKernelGGL((probe_ws<32, 3>), dim3(256), dim3(640), 4 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
          if (mode == 20) hipLaunchKernelGGL((probe_ws<2, 2>), dim3(256), dim3(640), 3 * 32768, st, x, y, 64, Cout, ntn, tiles, (unsigned)xb);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double us = best * 1e3 / reps;
      const double rd = (mode == 1 || mode == 3 || mode >= 5) ? (double)xb * ntn : (mode == 4 ? (double)xb : 0.0);
      const double wr = mode == 4 ? (double)xb : (double)yb;
      printf("Cout %3d mode %d (%s): %7.1f us   write %6.0f GB/s   read(issued) %6.0f GB/s\n", Cout, mode,
             mode == 0 ? "store pieces" : mode == 1 ? "load tile + store pieces" : mode == 2 ? "store contiguous tiles" : mode == 3 ? "persistent x4, next loads before stores" : mode == 4 ? "hipMemcpy x only" : mode == 5 ? "load -> 32 MFMA/wave -> store, one tile per WG" :
             mode == 6 ? "persistent 768 WGs pipelined loads | MFMA | stores" : mode == 7 ? "persistent 512 WGs pipelined" : mode == 8 ? "load -> 2 MFMA -> store" :
             mode == 9 ? "DMA ring D=1, 512 WGs" : mode == 10 ? "DMA ring D=1, 256 WGs" : mode == 11 ? "DMA ring D=2, 512 WGs" : mode == 12 ? "DMA ring D=2, 256 WGs" :
             mode == 13 ? "DMA ring D=3, 256 WGs" : mode == 14 ? "DMA ring D=2, 256 WGs, 2 MFMA" : mode == 15 ? "DMA ring D=1, 768 WGs" :
             mode == 16 ? "loader wave D=1, 256 WGs" : mode == 17 ? "loader wave D=2, 256 WGs" : mode == 18 ? "loader wave D=1, 512 WGs" : mode == 19 ? "loader wave D=3, 256 WGs" : "loader wave D=2, 256 WGs, 2 MFMA",
             us, wr / us / 1e3, rd / us / 1e3);
    }
    for (int i = 0; i < nbuf; ++i) { hipFree(xs[i]); hipFree(ys[i]); }
  }
  return 0;
}
