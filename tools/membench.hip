// membench.hip — what HBM delivers for this engine's access pattern: every wavefront reads a
// contiguous segment of `seg` bytes at a random 16-byte-aligned offset of a large buffer
// (a posting list), `inflight` 16-byte loads per lane in flight.  Roofline calibration only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t k) {
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 27; k *= 0x94D049BB133111EBull; k ^= k >> 31; return k;
}

template <int INF>
__global__ __launch_bounds__(64) void gather(const uint4* buf, uint64_t n_chunks, uint32_t seg_chunks, uint32_t segs_per_wave,
                                             uint32_t* out) {
  const int lane = threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < segs_per_wave; s++) {
    const uint64_t base = mix((uint64_t)blockIdx.x * segs_per_wave + s) % (n_chunks - seg_chunks);
    for (uint32_t c0 = 0; c0 < seg_chunks; c0 += 64 * INF) {
      uint4 v[INF];
#pragma unroll
      for (int u = 0; u < INF; u++) {
        uint32_t c = c0 + u * 64 + lane;
        c = c < seg_chunks ? c : seg_chunks - 1;
        v[u] = buf[base + c];
      }
#pragma unroll
      for (int u = 0; u < INF; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint64_t bytes = 1ull << 30;  // 1 GiB > 256 MiB infinity cache
  uint4* buf; uint32_t* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%8s %4s %10s %12s\n", "seg_B", "inf", "waves", "GB/s");
  for (uint32_t seg : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 16384u, 65536u}) {
    for (int inf : {1, 4}) {
      const uint32_t seg_chunks = seg / 16;
      const uint32_t waves = 1 << 20;
      const uint32_t spw = 16;
      auto run = [&]() {
        if (inf == 1) hipLaunchKernelGGL(gather<1>, dim3(waves), dim3(64), 0, 0, buf, bytes / 16, seg_chunks, spw, out);
        else hipLaunchKernelGGL(gather<4>, dim3(waves), dim3(64), 0, 0, buf, bytes / 16, seg_chunks, spw, out);
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%8u %4d %10u %12.1f\n", seg, inf, waves, (double)waves * spw * seg / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
