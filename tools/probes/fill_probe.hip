// Per-CU fill-rate probe: how fast can a CU pull L2-resident tiles (a) into VGPRs with
// global_load_dwordx4 and (b) into LDS with buffer_load ... lds, as a function of waves per CU.
// Build: hipcc -O3 --offload-arch=gfx950 fill_probe.hip -o fill_probe ; run: ./fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// each block walks `iters` tiles of TILE bytes inside its own 256-KiB window (L2 resident after warm-up)
template <int NT>
__global__ __launch_bounds__(NT) void vgpr_kernel(const char* src, int iters, unsigned* sink) {
  const char* base = src + (size_t)(blockIdx.x % 64) * 262144;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    const char* p = base + ((it * 16384) & 262143) + threadIdx.x * 16;
    u32x4 v[16384 / (NT * 16)];
#pragma unroll
    for (int k = 0; k < 16384 / (NT * 16); k++) v[k] = *(const u32x4*)(p + k * NT * 16);
#pragma unroll
    for (int k = 0; k < 16384 / (NT * 16); k++) acc ^= v[k];
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

template <int NT>
__global__ __launch_bounds__(NT) void dma_kernel(const char* src, int iters, unsigned* sink, unsigned bytes) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const char* base = src + (size_t)(blockIdx.x % 64) * 262144;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 262144, 0x00020000);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int NW = NT / 64, PC = 16 / NW;   // 16 KiB tile = 16 pieces
  for (int it = 0; it < iters; it++) {
    const unsigned off = ((it * 16384) & 262143) + wid * 1024 + lane * 16;
    char* dst = lds + (it & 1) * 16384 + wid * 1024;
#pragma unroll
    for (int k = 0; k < PC; k++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LPTR(dst + k * NW * 1024), 16, off + k * NW * 1024, 0, 0, 0);
    if (it >= 1) __builtin_amdgcn_s_waitcnt((PC & 15) | 0x0F70);   // leave one tile in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (((unsigned*)lds)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}

template <typename F>
float time_ms(F launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  char* src; unsigned* sink;
  hipMalloc(&src, 64 * 262144); hipMemset(src, 1, 64 * 262144); hipMalloc(&sink, 64);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)dma_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  hipFuncSetAttribute((const void*)dma_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  for (int blocks : {256, 512, 1024, 2048}) {
    const double gb = (double)blocks * iters * 16384 / 1e9;
    float t1 = time_ms([&] { hipLaunchKernelGGL(vgpr_kernel<256>, dim3(blocks), dim3(256), 0, 0, src, iters, sink); });
    float t2 = time_ms([&] { hipLaunchKernelGGL(dma_kernel<256>, dim3(blocks), dim3(256), 32768, 0, src, iters, sink, 0u); });
    float t3 = time_ms([&] { hipLaunchKernelGGL(vgpr_kernel<512>, dim3(blocks), dim3(512), 0, 0, src, iters, sink); });
    float t4 = time_ms([&] { hipLaunchKernelGGL(dma_kernel<512>, dim3(blocks), dim3(512), 32768, 0, src, iters, sink, 0u); });
    printf("blocks %4d: vgpr256 %6.1f TB/s  dma256 %6.1f TB/s  vgpr512 %6.1f TB/s  dma512 %6.1f TB/s\n", blocks,
           gb / t1, gb / t2, gb / t3, gb / t4);
  }
  return 0;
}
