#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int CH>
__global__ void k_fma(double* out, long long* cyc, double a, double b) {
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { x[c] = __builtin_fma(x[c], a, b); asm volatile("" : "+v"(x[c])); }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_add(double* out, long long* cyc, double b) {
    double x = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = x + b; asm volatile("" : "+v"(x)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mulmix(double* out, long long* cyc, double b) {
    // dependent chain alternating mul and add (like Horner without fma)
    double x = threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) { x = x * b; asm volatile("" : "+v"(x)); x = x + b; asm volatile("" : "+v"(x)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_sqrt(double* out, long long* cyc) {
    double x = threadIdx.x + 2.0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) { x = __builtin_sqrt(x) + 1.0; asm volatile("" : "+v"(x)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = (t1 - t0) * (REP / 64);
}
__global__ void k_mov32(double* out, long long* cyc, int b) {
    int x = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = x + b; asm volatile("" : "+v"(x)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_div(double* out, long long* cyc, double b) {
    double x = threadIdx.x + 1.5;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) { x = b / x; asm volatile("" : "+v"(x)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = (t1 - t0) * (REP / 64);
}
__global__ void k_bperm(double* out, long long* cyc) {
    int v = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v); asm volatile("" : "+v"(v)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dpp(double* out, long long* cyc) {
    int v = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { v = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); asm volatile("" : "+v"(v)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rdlane(double* out, long long* cyc) {
    int v = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { int s = __builtin_amdgcn_readlane(v, 5); v = v + s; asm volatile("" : "+v"(v)); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* cyc) {
    __shared__ int sh[256];
    sh[threadIdx.x] = ((threadIdx.x * 7) & 63) * 4;
    __syncthreads();
    int v = sh[threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { v = *(int*)((char*)sh + v); }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* o; long long* c; (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 8);
    long long h;
#define RUN(name, launch, ops) for (int r = 0; r < 3; ++r) { launch; (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); } printf("%-28s %8.2f cycles per op (%d ops per rep)\n", name, (double)h / REP / ops, ops);
    RUN("fma dependent (1 chain)", (k_fma<1><<<1, 64>>>(o, c, 1.0000001, 1e-9)), 1);
    RUN("fma 2 chains", (k_fma<2><<<1, 64>>>(o, c, 1.0000001, 1e-9)), 2);
    RUN("fma 4 chains", (k_fma<4><<<1, 64>>>(o, c, 1.0000001, 1e-9)), 4);
    RUN("fma 8 chains", (k_fma<8><<<1, 64>>>(o, c, 1.0000001, 1e-9)), 8);
    RUN("add f64 dependent", (k_add<<<1, 64>>>(o, c, 1e-9)), 1);
    RUN("mul/add f64 alternating", (k_mulmix<<<1, 64>>>(o, c, 1.0000001)), 1);
    RUN("sqrt+add f64 dependent", (k_sqrt<<<1, 64>>>(o, c)), 1);
    RUN("add i32 dependent", (k_mov32<<<1, 64>>>(o, c, 3)), 1);
    RUN("div f64 dependent", (k_div<<<1, 64>>>(o, c, 3.0)), 1);
    RUN("ds_bpermute dependent", (k_bperm<<<1, 64>>>(o, c)), 1);
    RUN("dpp row_shr dependent", (k_dpp<<<1, 64>>>(o, c)), 1);
    RUN("readlane+add dependent", (k_rdlane<<<1, 64>>>(o, c)), 1);
    RUN("LDS load dependent", (k_lds<<<1, 64>>>(o, c)), 1);
    return 0;
}
