// Does a wave64 FP64 instruction cost fewer issue cycles when most of EXEC is off?  (gfx950: a SIMD executes a
// wavefront's vector instruction 16 lanes per cycle; the question is whether quarter-passes with no active lane are
// skipped.)  Eight independent FMA chains per lane = issue-bound; timed with the 100 MHz wall clock over a long loop.
//   hipcc -O2 -ffp-contract=off --offload-arch=gfx950 exec_skip.hip -o exec_skip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CH 8
#define INNER 32
__global__ void k(double* out, long long* ticks, unsigned long long mask, int outer, double a, double b) {
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    long long t0 = wall_clock64();
    if (on) {
        for (int o = 0; o < outer; ++o) {
#pragma unroll
            for (int i = 0; i < INNER; ++i) {
#pragma unroll
                for (int c = 0; c < CH; ++c) { x[c] = __builtin_fma(x[c], a, b); asm volatile("" : "+v"(x[c])); }
            }
        }
    }
    long long t1 = wall_clock64();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void k32(float* out, long long* ticks, unsigned long long mask, int outer, float a, float b) {
    float x[CH];
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3f + c;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    long long t0 = wall_clock64();
    if (on) {
        for (int o = 0; o < outer; ++o) {
#pragma unroll
            for (int i = 0; i < INNER; ++i) {
#pragma unroll
                for (int c = 0; c < CH; ++c) { x[c] = __builtin_fmaf(x[c], a, b); asm volatile("" : "+v"(x[c])); }
            }
        }
    }
    long long t1 = wall_clock64();
    float s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* ticks; float* outf;
    hipMalloc(&out, 1 << 20); hipMalloc(&outf, 1 << 20); hipMalloc(&ticks, 1 << 16);
    const int outer = 20000;
    struct { const char* name; unsigned long long m; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lane 0", 1ull},
        {"lanes 0,16,32,48", 0x0001000100010001ull}, {"lanes 48-63", 0xffff000000000000ull}};
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("{\"sclk_khz\": %d, \"wall_clock_khz\": %d, \"instructions_per_wave\": %lld, \"cases\": [\n", clk_khz, wall_khz,
           (long long)outer * INNER * CH);
    for (int threads = 64; threads <= 512; threads *= 2) {       // 1, 2 (one per SIMD pair?), 4, 8 wavefronts in ONE block = one CU
        for (auto& c : cases) {
            for (int w = 0; w < 2; ++w) {
                long long h[1];
                if (w == 0) { hipLaunchKernelGGL(k, 1, threads, 0, 0, out, ticks, c.m, 10, 1.0000001, 1e-9); hipDeviceSynchronize();
                              hipLaunchKernelGGL(k, 1, threads, 0, 0, out, ticks, c.m, outer, 1.0000001, 1e-9); }
                else        { hipLaunchKernelGGL(k32, 1, threads, 0, 0, outf, ticks, c.m, 10, 1.0000001f, 1e-9f); hipDeviceSynchronize();
                              hipLaunchKernelGGL(k32, 1, threads, 0, 0, outf, ticks, c.m, outer, 1.0000001f, 1e-9f); }
                hipDeviceSynchronize();
                hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
                double ns = h[0] * 1e6 / wall_khz;
                printf(" {\"type\": \"%s\", \"wavefronts_on_one_cu\": %d, \"exec\": \"%s\", \"ns_per_instruction\": %.4f, "
                       "\"sclk_cycles_per_instruction\": %.3f},\n", w ? "f32" : "f64", threads / 64, c.name,
                       ns / ((double)outer * INNER * CH), ns * clk_khz * 1e-6 / ((double)outer * INNER * CH));
            }
        }
    }
    printf(" {}]}\n");
    return 0;
}
