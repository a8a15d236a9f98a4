// traffic_calibration.hip — known-byte kernels in THIS solver's access widths, run through the same rocprofv3 --pmc
// passes as the solve kernel (scripts/calibrate_traffic.sh), to learn what FETCH_SIZE / WRITE_SIZE mean for them on
// gfx950 and whether Infinity-Cache hits are counted (MI355X_MICROARCH.md: "calibrate on a known
// byte count in your own access pattern").
//
//   cal_read16      16 B per lane, coalesced (the guide's calibrated case: FETCH_SIZE reports half the bytes)
//   cal_read8       8 B per lane, coalesced (lane-table / obstacle-route reads)
//   cal_read8_s160  8 B per lane, 160 B between lanes (a trial cost reading one step size out of the slab)
//   cal_write8      8 B per lane, coalesced (results, first-trial buffer)
//   cal_write8_slab 20 lanes x 8 B contiguous per row (the rollout's slab stores: 160-byte rows)
//
// Each kernel touches `bytes` useful bytes of a working set of `set` bytes, `passes` times over (the second and later
// passes of a 64 MiB set are served by the 256 MiB Infinity Cache, those of a 1 GiB set by HBM).  Prints one JSON line
// per launch with the useful bytes and the bytes of the cache lines touched (128-B lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void cal_read16(const double2* __restrict__ p, size_t n, int passes, double* sink) {
    double acc = 0;
    for (int r = 0; r < passes; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            double2 v = p[i]; acc += v.x + v.y;
        }
    if (acc == 1.2345e-300) *sink = acc;
}
__global__ void cal_read8(const double* __restrict__ p, size_t n, int passes, double* sink) {
    double acc = 0;
    for (int r = 0; r < passes; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e-300) *sink = acc;
}
// element i of a logical array whose consecutive elements lie 160 B apart (20 doubles): reads column `col` of rows
__global__ void cal_read8_s160(const double* __restrict__ p, size_t rows, int passes, double* sink) {
    double acc = 0;
    for (int r = 0; r < passes; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 20 + (r % 20)];
    if (acc == 1.2345e-300) *sink = acc;
}
__global__ void cal_write8(double* __restrict__ p, size_t n, int passes) {
    for (int r = 0; r < passes; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)(i + r);
}
// one wavefront per row group: lanes 0..19 write the 160-byte row, the other lanes idle (as in the rollout)
__global__ void cal_write8_slab(double* __restrict__ p, size_t rows, int passes) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (int r = 0; r < passes; ++r)
        for (size_t row = wave; row < rows; row += nw)
            if (lane < 20) p[row * 20 + lane] = (double)(row + r);
}

int main(int argc, char** argv) {
    const size_t sets[2] = {(size_t)64 << 20, (size_t)1 << 30};
    const int passes = 8;
    void* buf; double* sink;
    CHECK(hipMalloc(&buf, sets[1]));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 0, sets[1]));
    const int grid = 256 * 8, block = 256;
    for (int s = 0; s < 2; ++s) {
        const size_t S = sets[s];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(cal_read16, dim3(grid), dim3(block), 0, 0, (const double2*)buf, S / 16, passes, sink);
            hipLaunchKernelGGL(cal_read8, dim3(grid), dim3(block), 0, 0, (const double*)buf, S / 8, passes, sink);
            hipLaunchKernelGGL(cal_read8_s160, dim3(grid), dim3(block), 0, 0, (const double*)buf, S / 160, passes, sink);
            hipLaunchKernelGGL(cal_write8, dim3(grid), dim3(block), 0, 0, (double*)buf, S / 8, passes);
            hipLaunchKernelGGL(cal_write8_slab, dim3(grid), dim3(block), 0, 0, (double*)buf, S / 160, passes);
            CHECK(hipDeviceSynchronize());
        }
        // the launch order above is the dispatch order in the counter file: 5 kernels x 2 repeats per set
        printf("{\"set_bytes\": %zu, \"passes\": %d, \"useful_bytes\": {\"cal_read16\": %zu, \"cal_read8\": %zu, \"cal_read8_s160\": %zu, "
               "\"cal_write8\": %zu, \"cal_write8_slab\": %zu}, \"line_bytes_touched\": {\"cal_read8_s160\": %zu}}\n",
               S, passes, S * passes, S * passes, (S / 160) * 8 * passes, S * passes, S * passes, (size_t)S * passes);
    }
    CHECK(hipFree(buf));
    return 0;
}
