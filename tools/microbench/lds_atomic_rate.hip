// LDS atomic throughput on gfx950, the way the search kernel uses it: one wavefront per workgroup, 12 wavefronts per CU
// (13 KB of LDS each), ds_add_rtn_u32 on 2048 counter words.  Variants: random addresses (what docID buckets look like),
// lane-linear addresses (no bank conflict), random without return value, random with two lanes-per-bank skew.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  extern __shared__ uint32_t cnt[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) cnt[i] = 0;
  __syncthreads();
  const uint32_t cbase = (uint32_t)(uintptr_t)(lds_u32*)cnt;
  // the 14 addresses of a lane are drawn once (the loop below issues nothing but the atomics, one wait and the max: it
  // measures the LDS, not an address generator)
  uint32_t x = (blockIdx.x * 64 + lane) * 2654435761u + 12345u, acc = 0;
  uint32_t addr[14];
#pragma unroll
  for (int e = 0; e < 14; e++) {
    x = x * 1664525u + 1013904223u;
    if (MODE == 1) addr[e] = (((uint32_t)lane * 4u + (uint32_t)e * 256u) & 8188u) | cbase;      // lane-linear: conflict-free
    else addr[e] = ((x >> 8) & 8188u) | cbase;                                                     // random word
  }
  const int active = MODE == 3 ? 16 : MODE == 4 ? 32 : MODE == 5 ? 8 : 64;
  for (int it = 0; it < iters; it++) {
    uint32_t old[14];
#pragma unroll
    for (int e = 0; e < 14; e++) {
      lds_u32* w = (lds_u32*)(uintptr_t)addr[e];
      old[e] = 0;
      if (MODE == 2) __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (lane < active) old[e] = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int e = 0; e < 14; e++) acc = max(acc, old[e]);
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc + cnt[lane];
}
template <int MODE> void run(const char* name, uint32_t* d) {
  const int iters = 2000, grid = 256 * 12 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 13000, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 13000, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)grid * iters * 14;
  printf("%-28s %.3f ms: %.2f G atomic instr/s = %.1f ps each GPU-wide = %.1f CU-cycles each at 2.4 GHz (256 CUs)\n", name, ms, instr / ms / 1e6,
         ms * 1e9 / instr, ms * 1e-3 * 2.4e9 * 256 / instr);
}
int main() {
  uint32_t* d; hipMalloc(&d, 64);
  run<0>("random, with return", d);
  run<1>("lane-linear, with return", d);
  run<2>("random, no return", d);
  run<4>("random, 32 of 64 lanes (EXEC)", d);
  run<3>("random, 16 of 64 lanes (EXEC)", d);
  run<5>("random, 8 of 64 lanes (EXEC)", d);
  return 0;
}
