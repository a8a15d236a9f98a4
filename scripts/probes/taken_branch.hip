#include <hip/hip_runtime.h>
#include <cstdio>
// cost of a taken forward branch for a lone wavefront
template <int MODE>
__global__ void k(double* out, long long* cyc, int n, int flag, double a, double b) {
    double x = threadIdx.x * 1e-3, y = 1.0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        x = __builtin_fma(x, a, b); asm volatile("" : "+v"(x));
        x = __builtin_fma(x, a, b); asm volatile("" : "+v"(x));
        if (MODE == 1) {
            // uniform branch, normally taken (skips the block)
            if (__builtin_expect(flag == i, 0)) {
#pragma unroll
                for (int t = 0; t < 40; ++t) { y = __builtin_fma(y, a, b); asm volatile("" : "+v"(y)); }
            }
        }
        if (MODE == 2) {
            // two such branches
            if (__builtin_expect(flag == i, 0)) {
#pragma unroll
                for (int t = 0; t < 40; ++t) { y = __builtin_fma(y, a, b); asm volatile("" : "+v"(y)); }
            }
            x = __builtin_fma(x, a, b); asm volatile("" : "+v"(x));
            if (__builtin_expect(flag == i + 1, 0)) {
#pragma unroll
                for (int t = 0; t < 40; ++t) { y = __builtin_fma(y, b, a); asm volatile("" : "+v"(y)); }
            }
        }
        x = __builtin_fma(x, a, b); asm volatile("" : "+v"(x));
        x = __builtin_fma(x, a, b); asm volatile("" : "+v"(x));
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* o; long long* c; (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 8);
    long long h; const int n = 4096;
    for (int r = 0; r < 2; ++r) { k<0><<<1, 64>>>(o, c, n, -5, 1.0000001, 1e-9); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
    printf("4 fma + loop branch           : %.1f cycles/iter\n", (double)h / n);
    for (int r = 0; r < 2; ++r) { k<1><<<1, 64>>>(o, c, n, -5, 1.0000001, 1e-9); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
    printf("+ 1 skipped block (40 instrs) : %.1f cycles/iter\n", (double)h / n);
    for (int r = 0; r < 2; ++r) { k<2><<<1, 64>>>(o, c, n, -5, 1.0000001, 1e-9); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
    printf("+ 2 skipped blocks (+1 fma)   : %.1f cycles/iter\n", (double)h / n);
    return 0;
}
