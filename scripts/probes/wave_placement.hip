#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
extern __shared__ double lds[];
__global__ void __launch_bounds__(128) k(unsigned* out, int spin) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the wave alive for a while so that all blocks are co-resident
    long long t0 = clock64();
    double acc = threadIdx.x;
    while (clock64() - t0 < spin) acc = acc * 1.0000001 + 1.0;
    lds[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hwid;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    const int B = 1024;
    unsigned* d; hipMalloc(&d, B * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(B), dim3(128), 26000, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 4);
    hipMemcpy(h.data(), d, B * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    // HW_ID gfx9: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned long long, std::vector<int>> bySimd;
    int same = 0;
    for (int b = 0; b < B; ++b) {
        unsigned m = h[b * 4], hl = h[b * 4 + 2];
        unsigned xm = h[b * 4 + 1] & 0xf, xh = h[b * 4 + 3] & 0xf;
        auto key = [](unsigned hw, unsigned x) { return ((unsigned long long)x << 32) | (hw & 0xff30u) ; };
        bySimd[key(m, xm)].push_back(0);
        bySimd[key(hl, xh)].push_back(1);
        if (((m >> 4) & 3) == ((hl >> 4) & 3)) same++;
        if (b < 12) printf("block %d main: xcc %u se %u cu %u simd %u wave %u | helper: xcc %u se %u cu %u simd %u wave %u\n", b,
                           xm, (m >> 13) & 7, (m >> 8) & 15, (m >> 4) & 3, m & 15, xh, (hl >> 13) & 7, (hl >> 8) & 15, (hl >> 4) & 3, hl & 15);
    }
    int mm = 0, hh = 0, mh = 0, other = 0;
    for (auto& kv : bySimd) {
        auto& v = kv.second;
        if (v.size() == 2) { if (v[0] == 0 && v[1] == 0) mm++; else if (v[0] == 1 && v[1] == 1) hh++; else mh++; }
        else other++;
    }
    printf("SIMDs used %zu: main+main %d, helper+helper %d, main+helper %d, other occupancy %d; blocks with both waves on same simd id %d\n",
           bySimd.size(), mm, hh, mh, other, same);
    return 0;
}
