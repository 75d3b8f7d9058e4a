// Diagnostic: which workgroups of a 512/768-block launch share a CU?  Reads HW_REG_XCC_ID / HW_REG_HW_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID = 20
  unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID = 4
  long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  lds[threadIdx.x] = 0;
}
int main() {
  for (int grid : {512, 768}) {
    int lds = grid == 512 ? 73728 : 46080;
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned* d; hipMalloc(&d, grid * 8);
    census<<<grid, 256, lds>>>(d, 200000);
    std::vector<unsigned> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < grid; ++b) {
      unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
      unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("grid %d: %zu distinct CUs\n", grid, cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ < 6) { printf("  xcc %llu key %llx:", kv.first >> 32, kv.first & 0xffffffff); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    hipFree(d);
  }
  return 0;
}
