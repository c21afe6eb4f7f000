// membench.hip — what the memory system delivers for this engine's access pattern, and what the PMC counters say about it
// (roofline calibration, DESIGN.md "Roofline accounting"): every wavefront reads contiguous segments of `seg` bytes (a posting
// list, or a sub-row of one) at random 16-byte-aligned offsets of a buffer of `buf_MiB` MiB — 128 MiB sits inside the 256 MiB
// Infinity Cache, 1 GiB and more do not — with `inf` 16-byte loads per lane in flight.
//   membench <buf_MiB> <seg_B> <inf: 1|2|4> [reps]   ->  one line: bytes per launch, GB/s  (seg_B: a power of two, 64 .. 65536)
// Run under `rocprofv3 --pmc FETCH_SIZE ...` the known byte count calibrates the counter for THIS pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t k) {
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 27; k *= 0x94D049BB133111EBull; k ^= k >> 31; return k;
}

// Every load instruction is fully useful: the 64 lanes of a load cover 1024 / seg whole segments (seg <= 1 KiB: lane groups of
// seg / 16 lanes, each on a segment of its own) or the next 1 KiB of one segment (seg > 1 KiB), and the INF loads in flight go to
// DIFFERENT places.  (The first version walked ONE segment with all INF loads: at 1 KiB three of four were clamped duplicates
// and the figure it gave was the instruction rate, not the memory system's.)
template <int INF>
__global__ __launch_bounds__(256) void gather(const uint4* buf, uint64_t n_chunks, uint32_t seg_chunks, uint32_t rounds, uint32_t salt,
                                             uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t per = seg_chunks < 64u ? seg_chunks : 64u;          // lanes on one segment
  const uint32_t grp = lane / per, off = lane % per, groups = 64u / per;
  const uint32_t walks = seg_chunks > 64u ? seg_chunks / 64u : 1u;   // loads that walk one long segment
  uint32_t acc = 0;
  for (uint32_t r = 0; r < rounds; r++) {
    uint64_t base[INF];
#pragma unroll
    for (int u = 0; u < INF; u++) base[u] = mix(((uint64_t)salt << 44) + (((uint64_t)wave * rounds + r) * INF + u) * groups + grp) % (n_chunks - seg_chunks);
    for (uint32_t w = 0; w < walks; w++) {
      uint4 v[INF];
#pragma unroll
      for (int u = 0; u < INF; u++) v[u] = buf[base[u] + w * 64u + off];
#pragma unroll
      for (int u = 0; u < INF; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint64_t buf_mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1024;
  const uint32_t seg = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024;
  const int inf = argc > 3 ? atoi(argv[3]) : 4;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const uint64_t bytes = buf_mib << 20;
  uint4* buf; uint32_t* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // 2^31 bytes per launch whatever the segment size: rounds x INF loads x 1 KiB per wavefront
  const uint32_t seg_chunks = seg / 16, waves = 1u << 17;
  const uint32_t walks = seg_chunks > 64u ? seg_chunks / 64u : 1u;
  const uint32_t rounds = (uint32_t)((1ull << 31) / ((uint64_t)waves * inf * 1024u * walks));
  const double launch_bytes = (double)waves * rounds * inf * 1024.0 * walks;
  auto run = [&](uint32_t salt) {
    if (inf == 1) hipLaunchKernelGGL(gather<1>, dim3(waves / 4), dim3(256), 0, 0, buf, bytes / 16, seg_chunks, rounds, salt, out);
    else if (inf == 2) hipLaunchKernelGGL(gather<2>, dim3(waves / 4), dim3(256), 0, 0, buf, bytes / 16, seg_chunks, rounds, salt, out);
    else hipLaunchKernelGGL(gather<4>, dim3(waves / 4), dim3(256), 0, 0, buf, bytes / 16, seg_chunks, rounds, salt, out);
  };
  run(0); hipDeviceSynchronize();
  double best = 0;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0); run(r + 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gbs = launch_bytes / (ms * 1e-3) / 1e9;
    best = gbs > best ? gbs : best;
  }
  printf("buf_MiB %llu seg_B %u inf %d bytes_per_launch %.0f best_GBps %.1f\n", (unsigned long long)buf_mib, seg, inf, launch_bytes, best);
  return 0;
}
