// Dev tool: where does the dispatcher place the first workgroups of a 2-per-CU grid?
//   (block id -> XCC id, SE/CU id, start time)  Used to choose the phase-stagger predicate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
    __shared__ float pad[17000];   // ~68 KB: two workgroups per CU, like the conv kernel
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = xcc; out[blockIdx.x * 4 + 1] = hwid; out[blockIdx.x * 4 + 2] = (unsigned)t0; }
    pad[threadIdx.x] = threadIdx.x;
    unsigned long long t = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (pad[threadIdx.x] < 0) out[0] = 1;
}
int main() {
    const int n = 1536;
    unsigned* d; hipMalloc(&d, n * 16);
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // key = xcc<<16 | se<<8 | cu
    for (int b = 0; b < n; ++b) {
        unsigned xcc = h[b * 4] & 0xF, hw = h[b * 4 + 1];
        unsigned cuid = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        if (b < 24) printf("block %3d xcc %u se %u sh %u cu %2u t %u\n", b, xcc, se, sh, cuid, h[b * 4 + 2]);
        cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    printf("distinct CUs seen: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ >= 6) break; printf("cu key %06x blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    // how many CUs have their first two blocks both < 256, one <256 and one in [256,512), ...
    int both_low = 0, split = 0, other = 0;
    for (auto& kv : cu) { auto& v = kv.second; if (v.size() < 2) { ++other; continue; } if (v[0] < 256 && v[1] < 256) ++both_low; else if (v[0] < 256 && v[1] >= 256 && v[1] < 512) ++split; else ++other; }
    printf("first two blocks of a CU: both <256: %d, one <256 one in [256,512): %d, other: %d\n", both_low, split, other);
    return 0;
}
