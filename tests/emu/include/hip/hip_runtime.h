// TEST INFRASTRUCTURE ONLY — a stand-in for <hip/hip_runtime.h> under which the UNMODIFIED kernel sources of
// toy-example-of-ilqr_amd/csrc/ compile as plain C++ for the build container's x86 host, so that tests/test_emulator.py can
// execute their wave64 logic without a GPU (tests/emu/README.md).  Nothing in the product includes, links or loads this:
// the shipped libraries are built by hipcc for gfx950 only and cilqr_create() fails without a GPU.
//
// What it provides: the device-language keywords as no-ops, threadIdx / blockIdx / dynamic LDS of the CURRENT emulated lane
// (every lane of every resident block is a fibre, emu_runtime.cpp), the handful of gfx950 builtins the kernels use — cross-lane
// moves (DPP, ds_bpermute, v_readlane, v_readfirstlane, ballot), wave / block barriers, raw buffer loads and stores with their
// range check, the buffer -> LDS DMA — and the thin slice of the HIP host API csrc/cilqr_amd.hip calls (memory = host memory,
// streams and events = program order, a kernel launch = run the grid to completion).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define CILQR_EMULATED 1
#define __host__
#define __device__
#define __global__
#define __launch_bounds__(...)
#define __forceinline__ inline __attribute__((always_inline))
#ifndef __HIP_MEMORY_SCOPE_SINGLETHREAD
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Idx { unsigned x, y, z; };
const Idx& thread_idx();
const Idx& block_idx();
const Idx& block_dim();
const Idx& grid_dim();
double* lds_base();  // the current block's dynamic LDS (allocated below 4 GB: kernels keep LDS addresses in 32 bits)
int lane_id();

enum XKind { X_READFIRSTLANE, X_READLANE, X_BPERMUTE, X_DPP, X_BALLOT, X_WAVE_BARRIER, X_BLOCK_BARRIER, X_SLEEP };
// one cross-lane operation of the calling lane; returns when the lanes that execute it together have all arrived
// `tag`: the identity of the operation in the SOURCE — the address of a static object of the macro expansion (one per enclosing
// function instantiation).  The return address will not do: the optimiser duplicates a call into both arms of an `if (lane == 0)`
// and the lanes of one v_readfirstlane would arrive from two addresses.
long long xlane(int kind, long long v, int p1, int p2, int p3, int p4, const void* tag);
#define EMU_TAG() ({ static const char emu_tag_ = 0; (const void*)&emu_tag_; })
long long clock_ticks();
// PREEMPTION POINT (adversarial scheduling with CILQR_EMU_PREEMPT=1): a wavefront can lose the SIMD between any two instructions;
// what that can break are the protocols BETWEEN wavefronts, whose steps are atomic operations on global memory.  Before every such
// operation the executing lane may hand the processor back (its wavefront resumes at a later visit), so that other blocks run
// inside windows like the one between rq_push's reservation and the store of its entry.
void preempt_point();
void probe(int id);  // build_emu.py PROBES
#define EMU_PROBE(id) emu::probe(id)
void launch(dim3 grid, dim3 block, size_t shm, std::function<void()> body);

struct BufferRsrc { char* base; unsigned num_records; };
}  // namespace emu

#define threadIdx (emu::thread_idx())
#define blockIdx (emu::block_idx())
#define blockDim (emu::block_dim())
#define gridDim (emu::grid_dim())

#define __builtin_amdgcn_readfirstlane(v) ((int)emu::xlane(emu::X_READFIRSTLANE, (long long)(int)(v), 0, 0, 0, 0, EMU_TAG()))
#define __builtin_amdgcn_readlane(v, src) ((int)emu::xlane(emu::X_READLANE, (long long)(int)(v), (int)(src), 0, 0, 0, EMU_TAG()))
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)emu::xlane(emu::X_BPERMUTE, (long long)(int)(v), (int)(addr), 0, 0, 0, EMU_TAG()))
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl); p1 = ctrl, p2 = row_mask << 4 | bank_mask, p3 = bound_ctrl, p4 = old
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) \
    ((int)emu::xlane(emu::X_DPP, (long long)(int)(src), (int)(ctrl), ((int)(rm) << 4) | (int)(bm), (bc) ? 1 : 0, (int)(old), EMU_TAG()))
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) __builtin_amdgcn_update_dpp(0, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_ballot_w64(p) ((unsigned long long)emu::xlane(emu::X_BALLOT, (p) ? 1 : 0, 0, 0, 0, 0, EMU_TAG()))
#define __ballot(p) __builtin_amdgcn_ballot_w64(p)
#define __builtin_amdgcn_wave_barrier() ((void)emu::xlane(emu::X_WAVE_BARRIER, 0, 0, 0, 0, 0, EMU_TAG()))
#define __syncthreads() ((void)emu::xlane(emu::X_BLOCK_BARRIER, 0, 0, 0, 0, 0, EMU_TAG()))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __hip_atomic_fetch_add(p, v, order, scope) (emu::preempt_point(), __atomic_fetch_add((p), (v), (order)))
#define __hip_atomic_fetch_or(p, v, order, scope) (emu::preempt_point(), __atomic_fetch_or((p), (v), (order)))
#define __hip_atomic_load(p, order, scope) (emu::preempt_point(), __atomic_load_n((p), (order)))
#define __hip_atomic_store(p, v, order, scope) (emu::preempt_point(), __atomic_store_n((p), (v), (order)))
#define __hip_atomic_compare_exchange_strong(p, e, d, so, fo, scope) (emu::preempt_point(), __atomic_compare_exchange_n((p), (e), (d), false, (so), (fo)))
// s_sleep n: the wavefront stays off the scheduler for about n / 4 of its turns (64 n cycles on the device) — a spin wait that polls
// with s_sleep in its loop then burns its bounded poll count as slowly, relative to working wavefronts, as it does on the GPU
#define __builtin_amdgcn_s_sleep(n) ((void)emu::xlane(emu::X_SLEEP, 0, (int)(n), 0, 0, 0, EMU_TAG()))
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_dcache_inv() ((void)0)
#define __builtin_amdgcn_s_memrealtime() ((unsigned long long)emu::clock_ticks())
#define __builtin_amdgcn_s_getreg(x) (0u)
#undef __builtin_readcyclecounter
#define __builtin_readcyclecounter() ((unsigned long long)emu::clock_ticks())
// HIP's wave shuffles (ds_bpermute underneath; an inactive source lane reads as zero), any 4- or 8-byte type
template <class T>
static inline __attribute__((always_inline)) T emu_shfl_from(T v, int src_lane, const void* tag) {
    static_assert(sizeof(T) <= 8, "shuffle of a 4- or 8-byte value");
    long long bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    const long long r = emu::xlane(emu::X_BPERMUTE, bits, src_lane << 2, 0, 0, 0, tag);
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T>
static inline __attribute__((always_inline)) T emu_shfl(T v, int src, int width, const void* tag) {
    const int self = emu::lane_id();
    return emu_shfl_from(v, (self & ~(width - 1)) + (src & (width - 1)), tag);
}
template <class T>
static inline __attribute__((always_inline)) T emu_shfl_up(T v, unsigned delta, int width, const void* tag) {
    const int self = emu::lane_id();
    int idx = self - (int)delta;
    if (idx < (self & ~(width - 1))) idx = self;
    return emu_shfl_from(v, idx, tag);
}
template <class T>
static inline __attribute__((always_inline)) T emu_shfl_down(T v, unsigned delta, int width, const void* tag) {
    const int self = emu::lane_id();
    int idx = self + (int)delta;
    if ((int)((self & (width - 1)) + delta) >= width) idx = self;
    return emu_shfl_from(v, idx, tag);
}
#define EMU_SHFL_PICK(_1, _2, _3, name, ...) name
#define __shfl(...) EMU_SHFL_PICK(__VA_ARGS__, emu_shfl3, emu_shfl2)(emu_shfl, __VA_ARGS__)
#define __shfl_up(...) EMU_SHFL_PICK(__VA_ARGS__, emu_shfl3, emu_shfl2)(emu_shfl_up, __VA_ARGS__)
#define __shfl_down(...) EMU_SHFL_PICK(__VA_ARGS__, emu_shfl3, emu_shfl2)(emu_shfl_down, __VA_ARGS__)
#define emu_shfl2(fn, v, s) fn((v), (s), 64, EMU_TAG())
#define emu_shfl3(fn, v, s, w) fn((v), (s), (w), EMU_TAG())
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
#define EMU_LOCKSTEP() __builtin_amdgcn_wave_barrier() /* build_emu.py LOCKSTEP_POINTS */
#define EMU_ASM_BARRIER() __asm__ volatile("" ::: "memory")

// ---- raw buffers (V#: base, num_records; stride 0): an access whose bytes do not all lie inside the range is dropped / reads zero
typedef emu::BufferRsrc __amdgpu_buffer_rsrc_t;
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short stride, int num_records, int flags) {
    (void)stride; (void)flags;
    return emu::BufferRsrc{static_cast<char*>(p), (unsigned)num_records};
}
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
static inline bool emu_buf_ok(const emu::BufferRsrc& r, unsigned voff, unsigned soff, unsigned bytes) {
    const unsigned long long o = (unsigned long long)voff + soff;  // (raw buffer, stride 0: offset = voffset + soffset [+ imm])
    return o + bytes <= r.num_records;
}
static inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(const emu::BufferRsrc& r, unsigned voff, unsigned soff, int aux) {
    emu_u32x2 v = {0u, 0u};
    if (emu_buf_ok(r, voff, soff, 8)) std::memcpy(&v, r.base + voff + soff, 8);
    return v;
}
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(const emu::BufferRsrc& r, unsigned voff, unsigned soff, int aux) {
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if (emu_buf_ok(r, voff, soff, 16)) std::memcpy(&v, r.base + voff + soff, 16);
    return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u32x2 v, const emu::BufferRsrc& r, unsigned voff, unsigned soff, int aux) {
    if (emu_buf_ok(r, voff, soff, 8)) std::memcpy(r.base + voff + soff, &v, 8);
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, const emu::BufferRsrc& r, unsigned voff, unsigned soff, int aux) {
    if (emu_buf_ok(r, voff, soff, 16)) std::memcpy(r.base + voff + soff, &v, 16);
}
// buffer -> LDS DMA: every lane moves `size` bytes from the buffer to LDS at (wave-uniform base + lane * size)
template <class LdsPtr>
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(const emu::BufferRsrc& r, LdsPtr lds, unsigned size, unsigned voff, unsigned soff,
                                                            int imm, int aux) {
    char* dst = (char*)(size_t)lds + (size_t)imm + (size_t)emu::lane_id() * size;
    if (emu_buf_ok(r, voff, soff + (unsigned)imm, size)) std::memcpy(dst, r.base + voff + soff + imm, size);
    else std::memset(dst, 0, size);
}

// ---- the HIP host API as csrc/cilqr_amd.hip uses it: device memory is host memory, everything completes in program order
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef void* hipDeviceptr_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags) { return hipHostMalloc(reinterpret_cast<void**>(p), n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t st);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
int emu_blocks_per_cu();
template <class K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K kern, int block, size_t shm) {
    (void)kern; (void)block; (void)shm;
    *n = emu_blocks_per_cu();
    return hipSuccess;
}
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (size_t)(shm), [=]() { kern(__VA_ARGS__); })
