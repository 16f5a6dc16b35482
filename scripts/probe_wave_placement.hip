// Where do the waves of persistent 256-lane workgroups land?  (GPU box; hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_wp scripts/probe_wave_placement.hip)
// 4 workgroups per CU by LDS (40 KB each), all co-resident (they wait for each other), every wave records HW_ID / XCC_ID.
// Prints, per (simd of wave 0 .. wave 3) pattern and per CU, how the four co-resident workgroups' wave 0 spread over the SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <array>
__global__ __launch_bounds__(256, 4) void probe(unsigned* out, int* counter, int total) {
  extern __shared__ unsigned char smem[];
  smem[threadIdx.x] = 1;
  if (threadIdx.x == 0) atomicAdd(counter, 1);
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
  if (threadIdx.x == 0) {
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total && spins < 20000000) spins++;
  }
  __syncthreads();
}
int main(int argc, char** argv) {
  const int wg = argc > 1 ? atoi(argv[1]) : 1024;
  unsigned* d; int* c;
  hipMalloc(&d, wg * 4 * 2 * sizeof(unsigned)); hipMalloc(&c, 4); hipMemset(c, 0, 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  hipLaunchKernelGGL(probe, dim3(wg), dim3(256), 40 * 1024, 0, d, c, wg);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<unsigned> h(wg * 8);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<std::array<int, 4>, int> pat;             // simd of waves 0..3
  std::map<unsigned, std::vector<std::array<int, 3>>> cu;  // (xcc, se, sh, cu) -> {block, simd of wave 0, wave slot of wave 0}
  for (int b = 0; b < wg; b++) {
    std::array<int, 4> s;
    for (int w = 0; w < 4; w++) s[w] = (h[(b * 4 + w) * 2] >> 4) & 3;
    pat[s]++;
    const unsigned id = h[b * 8], x = h[b * 8 + 1] & 15;
    const unsigned key = (x << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 8) | ((id >> 8) & 15);
    cu[key].push_back({b, (int)((id >> 4) & 3), (int)(id & 15)});
  }
  printf("workgroups %d, distinct CUs %zu\n", wg, cu.size());
  for (auto& kv : pat) printf("simd of waves 0..3 = %d %d %d %d : %d workgroups\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  std::map<std::array<int, 4>, int> spread;  // workgroups' wave 0 per SIMD on a CU
  int shown = 0;
  for (auto& kv : cu) {
    std::array<int, 4> n{0, 0, 0, 0};
    for (auto& e : kv.second) n[e[1]]++;
    spread[n]++;
    if (shown++ < 12) {
      printf("cu %06x:", kv.first);
      for (auto& e : kv.second) printf("  block %4d wave0 simd %d slot %d", e[0], e[1], e[2]);
      printf("\n");
    }
  }
  for (auto& kv : spread) printf("wave-0 count per SIMD on a CU = %d %d %d %d : %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  return 0;
}
