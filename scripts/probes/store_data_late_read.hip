// Does a 16-byte buffer store on gfx950 read its DATA registers after a later instruction of the same wavefront has rewritten
// them?  (Round 4: the first tiled build of the grouped rollout pass lost the low dword of lanes 12-15 of a store that the
// compiler had wrapped in a loop over the lanes' buffer descriptors: profiles/r04_experiments/tiled_slab_lost_rows.txt.)
// Each variant issues  buffer_store_dwordx4 v[D:D+3], voff, rsrc, soff offen  in a given instruction shape, lets GAP scalar
// instructions pass, overwrites v[D] with a poison value, and later reads the 16 bytes back: a poisoned dword in memory
// means the store read its data late.  Many wavefronts, three stores per step, no waits between steps: a busy memory pipe.
//   hipcc -O2 --offload-arch=gfx950 store_data_late_read.hip -o store_data_late_read && ./store_data_late_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SLOTS 16
#define ROWB 4096 /* 64 lanes x 64 bytes */

// VARIANT 0: plain store.  1: EXEC narrowed (s_and_saveexec) before, restored behind.  2: the compiler's per-descriptor loop
// (v_readfirstlane of the descriptor, s_and_saveexec, store, s_xor exec, s_cbranch_execnz, exec restored).  3: store, then a
// branch that is not taken.
template <int VARIANT, int GAP>
__device__ inline void store_then_clobber(u32x4& d, unsigned voff, u32x4 rs, unsigned soff, unsigned poison) {
#define GAPS "s_nop 0\n"
    if (VARIANT == 0) {
        asm volatile("buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n"
                     ".rept %5\n s_nop 0\n .endr\n"
                     "v_mov_b32 v20, %4\n" // (the FIRST register of the data tuple)
                     : "+{v[20:23]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(poison), "n"(GAP) : "memory");
    } else if (VARIANT == 1) {
        asm volatile("s_mov_b64 s[40:41], exec\n"
                     "s_and_saveexec_b64 s[42:43], s[40:41]\n"
                     "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n"
                     "s_mov_b64 exec, s[40:41]\n"
                     ".rept %5\n s_nop 0\n .endr\n"
                     "v_mov_b32 v20, %4\n"
                     : "+{v[20:23]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(poison), "n"(GAP) : "memory", "scc", "s40", "s41", "s42", "s43");
    } else if (VARIANT == 2) {
        asm volatile("s_mov_b64 s[40:41], exec\n"
                     "1:\n"
                     "v_readfirstlane_b32 s44, %6\n"
                     "v_cmp_eq_u32 vcc, s44, %6\n"
                     "s_nop 3\n"
                     "s_and_saveexec_b64 s[42:43], vcc\n"
                     "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n"
                     "s_xor_b64 exec, exec, s[42:43]\n"
                     "s_cbranch_execnz 1b\n"
                     "s_mov_b64 exec, s[40:41]\n"
                     ".rept %5\n s_nop 0\n .endr\n"
                     "v_mov_b32 v20, %4\n"
                     : "+{v[20:23]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(poison), "n"(GAP), "v"(soff /* a uniform value in a VGPR */)
                     : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44");
    } else {
        asm volatile("buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n"
                     "s_cmp_eq_u32 %3, 0x7fff\n"
                     "s_cbranch_scc1 2f\n"
                     "2:\n"
                     ".rept %5\n s_nop 0\n .endr\n"
                     "v_mov_b32 v20, %4\n"
                     : "+{v[20:23]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(poison), "n"(GAP) : "memory", "scc");
    }
}

template <int VARIANT, int GAP>
__global__ void __launch_bounds__(64) probe(unsigned* buf, unsigned* bad, unsigned* first_bad, int steps) {
    const unsigned lane = threadIdx.x;
    unsigned* mine = buf + (size_t)blockIdx.x * (3 * SLOTS * ROWB / 4);
    const unsigned long long a = (unsigned long long)(size_t)mine;
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)(a & 0xffffffffull));
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    rs.z = 3 * SLOTS * ROWB;
    rs.w = 0x00020000;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 3 * SLOTS * ROWB, 0x00020000);
    unsigned nbad = 0;
    for (int it = 0; it < steps; ++it) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((it & 3) * 16);
        for (int p = 0; p < 3; ++p) { // three stores per step, like a rollout step
            u32x4 d;
            d.x = 0xA0000000u | ((unsigned)it << 4) | (unsigned)p;
            d.y = lane; d.z = (unsigned)it * 3u + (unsigned)p; d.w = 0x5a5a5a5au;
            const unsigned voff = lane * 64u + (unsigned)((it >> 2) & (SLOTS - 1)) * ROWB + (unsigned)p * (SLOTS * ROWB);
            store_then_clobber<VARIANT, GAP>(d, voff, rs, soff, 0xDEAD0000u | lane);
            asm volatile("" :: "v"(d)); // (keep the tuple alive past the clobber)
        }
        if ((it & 63) == 63) { // the ring is full of this block's last 64 steps: read it back
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int j = it - 63; j <= it; ++j)
                for (int p = 0; p < 3; ++p) {
                    const unsigned voff = lane * 64u + (unsigned)((j >> 2) & (SLOTS - 1)) * ROWB + (unsigned)p * (SLOTS * ROWB) + (unsigned)(j & 3) * 16u;
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 1 /* glc */);
                    const unsigned want = 0xA0000000u | ((unsigned)j << 4) | (unsigned)p;
                    if (r.x != want) {
                        ++nbad;
                        if (atomicAdd(first_bad, 1u) < 8u) atomicExch(first_bad + 1 + lane % 8, r.x), atomicExch(first_bad + 9, lane);
                    }
                }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int VARIANT, int GAP>
static void run(unsigned* buf, unsigned* dbad, const char* name) {
    (void)hipMemset(dbad, 0, 64);
    const int blocks = 4096, steps = 2048;
    hipLaunchKernelGGL((probe<VARIANT, GAP>), dim3(blocks), dim3(64), 0, 0, buf, dbad, dbad + 1, steps);
    (void)hipDeviceSynchronize();
    unsigned h[16];
    (void)hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"shape\": \"%s\", \"scalar_instructions_before_the_overwrite\": %d, \"stores\": %lld, \"stores_that_read_the_overwritten_dword\": %u, "
           "\"a_lane\": %u, \"a_value\": \"0x%08x\"}\n", name, GAP, (long long)blocks * steps * 3 * 64, h[0], h[10], h[2]);
}

int main() {
    unsigned *buf, *dbad;
    const size_t bytes = (size_t)4096 * 3 * SLOTS * ROWB;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&dbad, 64);
    (void)hipMemset(buf, 0, bytes);
    run<0, 0>(buf, dbad, "plain store");
    run<0, 1>(buf, dbad, "plain store");
    run<0, 2>(buf, dbad, "plain store");
    run<0, 5>(buf, dbad, "plain store");
    run<1, 0>(buf, dbad, "EXEC narrowed before, restored behind");
    run<1, 5>(buf, dbad, "EXEC narrowed before, restored behind");
    run<3, 0>(buf, dbad, "store, branch not taken");
    run<3, 5>(buf, dbad, "store, branch not taken");
    run<2, 0>(buf, dbad, "per-descriptor loop (readfirstlane, saveexec, store, xor exec, cbranch_execnz, restore)");
    run<2, 2>(buf, dbad, "per-descriptor loop (readfirstlane, saveexec, store, xor exec, cbranch_execnz, restore)");
    run<2, 5>(buf, dbad, "per-descriptor loop (readfirstlane, saveexec, store, xor exec, cbranch_execnz, restore)");
    run<2, 12>(buf, dbad, "per-descriptor loop (readfirstlane, saveexec, store, xor exec, cbranch_execnz, restore)");
    return 0;
}
