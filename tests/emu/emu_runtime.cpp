// TEST INFRASTRUCTURE ONLY — the wave64 emulator behind tests/emu/include/hip/hip_runtime.h (see tests/emu/README.md).
//
// Execution model.  A launch runs its grid to completion inside hipLaunchKernelGGL.  Every lane of every RESIDENT block is a
// fibre (own stack, cooperative switches, one host thread); as many blocks are resident as the emulated chip holds
// (CILQR_EMU_CUS x CILQR_EMU_BLOCKS_PER_CU), the others start as resident ones finish — persistent kernels, spin waits between
// blocks and hand-over protocols behave as on the device.  A lane runs until its next CROSS-LANE operation (DPP move,
// ds_bpermute, v_readlane, v_readfirstlane, ballot, wave barrier, __syncthreads) and blocks there.  When no lane of a wavefront
// can run, the lanes blocked at the same operation OF THE SOURCE (a static tag per macro expansion and enclosing function
// instantiation — not the return address: the optimiser duplicates calls into both arms of an `if (lane == 0)`) form a group —
// the lanes that execute the instruction together, i.e. its EXEC mask — and the group is resolved at once: sources outside the
// group read as zero / keep `old`, as on the hardware.
// Divergence: groups at different sites are resolved innermost first (deeper stack, then lower code address: loop bodies and
// if-branches before the code after them); a barrier is only resolved when every live lane of the wavefront (block) stands at
// one — a barrier met by part of the live lanes while the others wait elsewhere is counted (`partial_barriers`) and reported,
// it would mean the order above guessed wrong.  Memory: lanes run one at a time, so every access is sequentially consistent;
// what the emulator cannot show are hardware effects (LDS bank conflicts, the lost-store anomaly of round 4, timing).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

#ifdef CILQR_EMU_SANITIZE
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#else
#define ASAN_POISON_MEMORY_REGION(p, n) ((void)0)
#define ASAN_UNPOISON_MEMORY_REGION(p, n) ((void)0)
#endif

namespace emu {

extern "C" void emu_swap(void** save_sp, void* load_sp);
__asm__(
    ".text\n.globl emu_swap\n.type emu_swap,@function\nemu_swap:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_swap, .-emu_swap\n");

enum St { RUN, XWAIT, DONE, PREEMPTED };
struct Block;
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    Block* blk = nullptr;
    int tid = 0, lane = 0, wave = 0;
    St st = RUN;
    Idx tidx{0, 0, 0};
    // the pending cross-lane operation
    int kind = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
    int stall = 0;  // PREEMPTED: visits of its wavefront still to sit out
    long long v = 0, result = 0;
    const void* site = nullptr;
    const void* ret = nullptr;
    size_t depth = 0;  // stack pointer at the operation (lower = deeper)
    void* asan_fake = nullptr; // (sanitizer build: the fibre's fake-stack handle across switches)
    long long nops = 0; // cross-lane operations executed so far: the lane's SEGMENT number (lockstep-hazard detector)
};
struct Block {
    Idx bidx{0, 0, 0};
    char* lds = nullptr;
    std::vector<Fiber*> f;  // [threads]
    int live = 0;
    int n_waves = 0;
};

static const size_t STACK_BYTES = 1u << 20;
static std::vector<char*> g_stack_pool;
static Fiber* g_cur = nullptr;
static void* g_sched_sp = nullptr;
static Idx g_bdim{1, 1, 1}, g_gdim{1, 1, 1};
static std::function<void()>* g_body = nullptr;
static long long g_clock = 0;
static char* g_lds_arena = nullptr;
static const size_t LDS_ARENA = 64u << 20, LDS_PER_BLOCK = 160u << 10;
static Idx g_host_idx{0, 0, 0};

struct Stats {
    long long launches = 0, xops = 0, groups = 0, split_resolutions = 0, partial_barriers = 0, readlane_inactive = 0, preemptions = 0;
    std::map<std::pair<const void*, const void*>, long long> split_sites;
    std::map<const void*, long long> inactive_sites;
} g_stats;

const Idx& thread_idx() { return g_cur ? g_cur->tidx : g_host_idx; }
const Idx& block_idx() { return g_cur ? g_cur->blk->bidx : g_host_idx; }
const Idx& block_dim() { return g_bdim; }
const Idx& grid_dim() { return g_gdim; }
double* lds_base() { return reinterpret_cast<double*>(g_cur->blk->lds); }
int lane_id() { return g_cur->lane; }
long long clock_ticks() { return ++g_clock; }

static unsigned long long g_sched_seed = 0;
static unsigned sched_rand() {  // xorshift64*
    g_sched_seed ^= g_sched_seed >> 12; g_sched_seed ^= g_sched_seed << 25; g_sched_seed ^= g_sched_seed >> 27;
    return (unsigned)((g_sched_seed * 2685821657736338717ULL) >> 33);
}
static int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

#ifdef CILQR_EMU_SANITIZE
static const void* g_sched_stack_bottom = nullptr;
static size_t g_sched_stack_size = 0;
#endif
// fibre -> scheduler (AddressSanitizer is told about every stack switch)
static inline void to_scheduler(Fiber* f, bool last) {
#ifdef CILQR_EMU_SANITIZE
    __sanitizer_start_switch_fiber(last ? nullptr : &f->asan_fake, g_sched_stack_bottom, g_sched_stack_size);
#endif
    emu_swap(&f->sp, g_sched_sp);
#ifdef CILQR_EMU_SANITIZE
    __sanitizer_finish_switch_fiber(f->asan_fake, &g_sched_stack_bottom, &g_sched_stack_size);
#endif
}

static long long g_probes[16];
void probe(int id) { if (g_cur && g_cur->lane == 0 && id >= 0 && id < 16) ++g_probes[id]; }
static int g_preempt = 0;  // CILQR_EMU_PREEMPT (read per launch; only under adversarial scheduling)
void preempt_point() {
    Fiber* f = g_cur;
    if (!f || !g_preempt || !g_sched_seed) return;
    if (sched_rand() % (unsigned)g_preempt) return;   // one point in g_preempt, on average
    ++g_stats.preemptions;
    // two times in three the wavefront is back at its next visit; otherwise it stays away for up to 400 visits — long enough for
    // other wavefronts to complete whole protocol steps inside the window (measured on the claimed-place branch of the grouped
    // kernel's queue, build_emu.py PROBES: reached by 1 launch in 2 400 without the long stalls, by 1 in 30 with them)
    f->stall = (sched_rand() % 3 == 0) ? (int)(sched_rand() % 400) : 0;
    f->st = PREEMPTED;
    to_scheduler(f, false);
}

__attribute__((noinline)) long long xlane(int kind, long long v, int p1, int p2, int p3, int p4, const void* tag) {
    Fiber* f = g_cur;
    if (!f) {
        std::fprintf(stderr, "emu: cross-lane operation outside a kernel\n");
        std::abort();
    }
    f->kind = kind; f->v = v; f->p1 = p1; f->p2 = p2; f->p3 = p3; f->p4 = p4;
    f->site = tag;  // identity for grouping
    f->ret = __builtin_extract_return_addr(__builtin_return_address(0));  // code address: order among groups, diagnostics
    f->depth = reinterpret_cast<size_t>(__builtin_frame_address(0));
    f->st = XWAIT;
    ++f->nops;
    ++g_clock;
    to_scheduler(f, false);
    return f->result;
}

static void fiber_main() {
    Fiber* f = g_cur;
#ifdef CILQR_EMU_SANITIZE
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack_bottom, &g_sched_stack_size);
#endif
    (*g_body)();
    f->st = DONE;
    to_scheduler(f, true);
    std::abort();  // (a finished fibre is never resumed)
}

static char* get_stack() {
    if (!g_stack_pool.empty()) {
        char* s = g_stack_pool.back();
        g_stack_pool.pop_back();
        return s;
    }
    void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::perror("emu: mmap stack"); std::abort(); }
    return static_cast<char*>(p);
}

static void start_fiber(Fiber* f) {
    f->stack = get_stack();
    // initial frame for emu_swap: six callee-saved registers, then the "return address" = fiber_main; at fiber_main's entry
    // the stack pointer must be 8 modulo 16 (as after a call)
    size_t top = (reinterpret_cast<size_t>(f->stack) + STACK_BYTES) & ~size_t(15);
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                                   // (alignment: after `ret` pops fiber_main, rsp = top - 8)
    *--sp = reinterpret_cast<void*>(&fiber_main);
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f->sp = sp;
    f->st = RUN;
}

static void run_fiber(Fiber* f) {
    g_cur = f;
#ifdef CILQR_EMU_SANITIZE
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, f->stack, STACK_BYTES);
#endif
    emu_swap(&g_sched_sp, f->sp);
#ifdef CILQR_EMU_SANITIZE
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    g_cur = nullptr;
}

// ---- resolving a group -----------------------------------------------------------------------------------------------------
static inline bool is_barrier(int k) { return k == X_WAVE_BARRIER || k == X_BLOCK_BARRIER || k == X_SLEEP; }

static Fiber** g_dbg_lanes = nullptr;
static int g_dbg_nl = 0;
static void resolve(std::vector<Fiber*>& grp) {
    // grp: lanes of ONE wavefront blocked at the same site with the same kind
    Fiber* by_lane[64] = {nullptr};
    for (Fiber* f : grp) by_lane[f->lane] = f;
    const int kind = grp[0]->kind;
    ++g_stats.groups;
    {   // lanes that execute an operation together leave it in the same segment (the hazard detector compares segment numbers)
        long long m = 0;
        for (Fiber* f : grp) m = std::max(m, f->nops);
        for (Fiber* f : grp) f->nops = m;
    }
    if (kind == X_BALLOT) {
        unsigned long long m = 0;
        for (Fiber* f : grp) if (f->v) m |= 1ULL << f->lane;
        for (Fiber* f : grp) f->result = (long long)m;
    } else if (kind == X_READFIRSTLANE) {
        Fiber* first = nullptr;
        for (int l = 0; l < 64 && !first; ++l) first = by_lane[l];
        for (Fiber* f : grp) f->result = first->v;
    } else if (kind == X_READLANE) {
        for (Fiber* f : grp) {
            Fiber* s = by_lane[f->p1 & 63];
            if (!s) {
                if (!g_stats.readlane_inactive && env_int("CILQR_EMU_DEBUG", 0) && g_dbg_lanes) {
                    std::fprintf(stderr, "emu: v_readlane of lane %d, which is not in the group (site %p, lane %d asks, group of %zu)\n", f->p1, f->site, f->lane, grp.size());
                    for (int l = 0; l < g_dbg_nl; ++l)
                        std::fprintf(stderr, "  lane %2d st %d kind %d site %p depth %zx nops %lld\n", l, (int)g_dbg_lanes[l]->st, g_dbg_lanes[l]->kind, g_dbg_lanes[l]->site, g_dbg_lanes[l]->depth, g_dbg_lanes[l]->nops);
                }
                ++g_stats.readlane_inactive; ++g_stats.inactive_sites[f->ret];
            }  // (v_readlane reads an inactive lane's register on the hardware: not representable)
            f->result = s ? s->v : 0;
        }
    } else if (kind == X_BPERMUTE) {
        for (Fiber* f : grp) {
            Fiber* s = by_lane[(f->p1 >> 2) & 63];
            f->result = s ? s->v : 0;
        }
    } else if (kind == X_DPP) {
        for (Fiber* f : grp) {
            const int lane = f->lane, ctrl = f->p1, row_mask = (f->p2 >> 4) & 0xf, bank_mask = f->p2 & 0xf, bound = f->p3;
            const int row = lane >> 4, in_row = lane & 15, bank = in_row >> 2;
            int src = -1;
            if (ctrl >= 0 && ctrl <= 0xff) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);            // quad_perm
            else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; if (in_row + n < 16) src = lane + n; }  // row_shl
            else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (in_row >= n) src = lane - n; }      // row_shr
            else if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl - 0x120; src = (lane & ~15) | ((in_row - n) & 15); }  // row_ror
            else if (ctrl == 0x140) src = (lane & ~15) | (15 - in_row);                                        // row_mirror
            else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));                                      // row_half_mirror
            else { std::fprintf(stderr, "emu: dpp_ctrl 0x%x is not implemented\n", ctrl); std::abort(); }
            long long r;
            if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) r = f->p4;        // the lane does not take the move: keeps old
            else if (src < 0 || !by_lane[src]) r = bound ? 0 : f->p4;                      // invalid / inactive source
            else r = by_lane[src]->v;
            f->result = r;
        }
    } else {
        for (Fiber* f : grp) f->result = 0;
    }
    for (Fiber* f : grp) f->st = RUN;
}

// one visit of a wavefront: run what can run; when nothing can, resolve one group.  Returns true if anything happened.
static bool visit_wave(Block& b, int w) {
    Fiber** lanes = &b.f[(size_t)w * 64];
    const int nl = std::min<int>(64, (int)b.f.size() - w * 64);
    bool any = false;
    g_dbg_lanes = lanes; g_dbg_nl = nl;
    for (int l = 0; l < nl; ++l)
        if (lanes[l]->st == PREEMPTED && lanes[l]->stall-- <= 0) lanes[l]->st = RUN;  // (a lane that lost the processor at a preemption point goes on)
    bool preempted = false;
    for (int l = 0; l < nl; ++l) preempted |= lanes[l]->st == PREEMPTED;
    for (int l = 0; l < nl; ++l)
        if (lanes[l]->st == RUN) {
            run_fiber(lanes[l]);
            if (lanes[l]->st == DONE) --b.live;
            preempted |= lanes[l]->st == PREEMPTED;
            any = true;
        }
    if (preempted) return true;  // (the wavefront is in mid-stretch: nothing of it is resolved before that lane has arrived)
    // every lane is now blocked or done
    std::vector<Fiber*> waiting;
    for (int l = 0; l < nl; ++l)
        if (lanes[l]->st == XWAIT) waiting.push_back(lanes[l]);
    if (waiting.empty()) return any;
    g_stats.xops += 0;
    // groups by (site, kind); non-barrier groups first, innermost first
    const void* best_site = nullptr;
    const void* best_ret = nullptr;
    size_t best_depth = 0;
    bool have = false, all_barrier = true;
    int n_sites = 0;
    const void* first_site = waiting[0]->site;
    for (Fiber* f : waiting) {
        if (f->site != first_site) n_sites = 2;
        if (is_barrier(f->kind)) continue;
        all_barrier = false;
        if (!have || f->depth < best_depth || (f->depth == best_depth && f->ret < best_ret)) {
            have = true; best_depth = f->depth; best_site = f->site; best_ret = f->ret;
        }
    }
    if (!all_barrier) {
        std::vector<Fiber*> grp;
        for (Fiber* f : waiting)
            if (f->site == best_site && !is_barrier(f->kind)) grp.push_back(f);
        if (n_sites > 1) {
            ++g_stats.split_resolutions;
            for (Fiber* f : waiting)
                if (f->site != best_site) { ++g_stats.split_sites[{best_ret, f->ret}]; break; }
        }
        g_stats.xops += (long long)grp.size();
        resolve(grp);
        return true;
    }
    // only barriers are waited for.  Wave barrier: every live lane of the wavefront stands at one (whatever the site: lanes of one
    // wavefront that sit at two DIFFERENT wave barriers have parted ways for good — counted)
    bool block_bar = false;
    for (Fiber* f : waiting) block_bar |= f->kind == X_BLOCK_BARRIER;
    if (!block_bar) {
        if (n_sites > 1) ++g_stats.partial_barriers;
        // the lanes at the innermost barrier go on (a wave barrier is no rendezvous on the hardware: it orders, it does not wait)
        std::vector<Fiber*> grp;
        const void* site = nullptr; const void* ret = nullptr; size_t depth = 0; bool h2 = false;
        for (Fiber* f : waiting)
            if (!h2 || f->depth < depth || (f->depth == depth && f->ret < ret)) { h2 = true; depth = f->depth; site = f->site; ret = f->ret; }
        for (Fiber* f : waiting) if (f->site == site) grp.push_back(f);
        if (grp[0]->kind == X_SLEEP) {  // the wavefront sleeps: n / 4 more visits before it goes on
            bool awake = true;
            for (Fiber* f : grp) { f->p1 -= 4; awake = awake && f->p1 <= 0; }
            if (!awake) return true;  // (time passes: not a deadlock)
        }
        g_stats.xops += (long long)grp.size();
        resolve(grp);
        return true;
    }
    return any;  // (__syncthreads: resolved per block, below)
}

static bool visit_block_barrier(Block& b) {
    // every live thread of the block stands at __syncthreads
    int at = 0;
    for (Fiber* f : b.f) {
        if (f->st == DONE) continue;
        if (f->st == XWAIT && f->kind == X_BLOCK_BARRIER) ++at;
        else return false;
    }
    if (at == 0) return false;
    for (Fiber* f : b.f)
        if (f->st == XWAIT) { f->result = 0; f->st = RUN; }
    ++g_stats.groups;
    return true;
}

static std::recursive_mutex g_launch_mutex;  // one launch at a time: host threads (one per shard of ShardedSolver) take turns
void launch(dim3 grid, dim3 block, size_t shm, std::function<void()> body) {
    std::lock_guard<std::recursive_mutex> lock(g_launch_mutex);
    if (g_cur) { std::fprintf(stderr, "emu: nested launch\n"); std::abort(); }
    if (!g_lds_arena) {
        void* p = mmap(nullptr, LDS_ARENA, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_32BIT | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { std::perror("emu: mmap LDS arena below 4 GB"); std::abort(); }
        g_lds_arena = static_cast<char*>(p);
    }
    if (shm > LDS_PER_BLOCK) { std::fprintf(stderr, "emu: %zu bytes of LDS asked for, a CU has %zu\n", shm, LDS_PER_BLOCK); std::abort(); }
    ++g_stats.launches;
    if (env_int("CILQR_EMU_DEBUG", 0))
        std::fprintf(stderr, "emu: launch %lld grid %u block %u dynamic LDS %zu bytes\n", g_stats.launches, grid.x, block.x, shm);
    {
        const int seed = env_int("CILQR_EMU_SCHED_SEED", 0);
        g_sched_seed = seed ? (0x9E3779B97F4A7C15ULL * (unsigned long long)seed + (unsigned long long)g_stats.launches) | 1ULL : 0ULL;
        g_preempt = env_int("CILQR_EMU_PREEMPT", 0);
    }
    const int threads = (int)(block.x * block.y * block.z);
    const long long n_blocks = (long long)grid.x * grid.y * grid.z;
    const int max_res = std::max(1, std::min<int>((int)(LDS_ARENA / LDS_PER_BLOCK), env_int("CILQR_EMU_RESIDENT_BLOCKS", 16)));
    g_bdim = Idx{block.x, block.y, block.z};
    g_gdim = Idx{grid.x, grid.y, grid.z};
    g_body = &body;
    std::vector<Block*> res;
    std::vector<int> free_slots;
    for (int i = max_res - 1; i >= 0; --i) free_slots.push_back(i);
    std::map<Block*, int> slot_of;
    long long next_block = 0;
    long long idle_rounds = 0;
    while (next_block < n_blocks || !res.empty()) {
        while (next_block < n_blocks && !free_slots.empty()) {
            Block* b = new Block();
            const int slot = free_slots.back();
            free_slots.pop_back();
            slot_of[b] = slot;
            b->lds = g_lds_arena + (size_t)slot * LDS_PER_BLOCK;
            ASAN_UNPOISON_MEMORY_REGION(b->lds, LDS_PER_BLOCK);
            if (env_int("CILQR_EMU_POISON_LDS", 1)) std::memset(b->lds, 0xff, shm);  // (uninitialised LDS reads as NaNs)
            ASAN_POISON_MEMORY_REGION(b->lds + shm, LDS_PER_BLOCK - shm);  // (sanitizer build: an access beyond the launch's dynamic LDS is reported)
            const unsigned bi = (unsigned)next_block++;
            b->bidx = Idx{bi % grid.x, (bi / grid.x) % grid.y, bi / (grid.x * grid.y)};
            b->n_waves = (threads + 63) / 64;
            b->live = threads;
            for (int t = 0; t < threads; ++t) {
                Fiber* f = new Fiber();
                f->blk = b; f->tid = t; f->lane = t & 63; f->wave = t >> 6;
                f->tidx = Idx{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
                start_fiber(f);
                b->f.push_back(f);
            }
            res.push_back(b);
        }
        bool any = false;
        if (g_sched_seed) {
            // ADVERSARIAL SCHEDULING (CILQR_EMU_SCHED_SEED): the resident blocks are visited in random order and a block is given a
            // random number of turns — or none: it stalls for a while — so that protocols between blocks (tickets, claims, pushes,
            // bounded waits) meet interleavings a fair round robin never produces.  Results must not depend on it.
            std::vector<Block*> order(res);
            for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[sched_rand() % i]);
            for (Block* b : order) {
                const unsigned r = sched_rand() % 16;
                const int turns = r < 4 ? 0 : (r < 12 ? 1 : (r < 15 ? 4 : 64));
                for (int t = 0; t < turns; ++t) {
                    bool a2 = false;
                    for (int w = 0; w < b->n_waves; ++w) a2 |= visit_wave(*b, w);
                    a2 |= visit_block_barrier(*b);
                    any |= a2;
                    if (!a2) break;
                }
                if (turns == 0 && b->live > 0) any = true;  // (a stalled block is not a deadlock)
            }
        } else
        for (Block* b : res) {
            for (int w = 0; w < b->n_waves; ++w) any |= visit_wave(*b, w);
            any |= visit_block_barrier(*b);
        }
        for (size_t i = 0; i < res.size();) {
            Block* b = res[i];
            if (b->live == 0) {
                for (Fiber* f : b->f) { g_stack_pool.push_back(f->stack); delete f; }
                free_slots.push_back(slot_of[b]);
                slot_of.erase(b);
                delete b;
                res.erase(res.begin() + (long)i);
                any = true;
            } else ++i;
        }
        if (!any) {
            if (++idle_rounds > 4) {
                std::fprintf(stderr, "emu: deadlock — %zu resident blocks, none can make progress\n", res.size());
                for (Block* b : res)
                    for (Fiber* f : b->f)
                        if (f->st == XWAIT && f->lane < 2)
                            std::fprintf(stderr, "  block %u lane %d kind %d at %p\n", b->bidx.x, f->tid, f->kind, f->ret);
                std::abort();
            }
        } else idle_rounds = 0;
    }
    g_body = nullptr;
}

}  // namespace emu

// ---- lockstep-hazard detector (the instrumented build only: tests/emu/build_emu.py --hazards) ----------------------------------
// The emulator runs the lanes of a wavefront ONE AFTER THE OTHER between two cross-lane operations, the hardware runs them in
// lockstep.  The two agree unless lanes of one wavefront exchange data through memory INSIDE such a segment — lane A stores, lane
// B loads, no cross-lane operation or wave barrier in between — which is legal on the device (a wavefront's LDS operations
// execute in order) and is exactly where the emulator needs an explicit lockstep point (build_emu.py LOCKSTEP_POINTS).  The
// kernels' loads and stores are traced (-fsanitize-coverage=trace-loads,trace-stores); a word touched by two lanes of one
// wavefront in the same segment, at least one of them storing, is reported with both code addresses.
namespace emu {
struct Shadow { const Fiber* wf = nullptr; long long wn = -1; const void* ws = nullptr; const Fiber* rf[2] = {nullptr, nullptr}; long long rn[2] = {-1, -1}; const void* rs[2] = {nullptr, nullptr}; };
static std::map<size_t, Shadow>* g_shadow = nullptr;
static std::map<std::pair<const void*, const void*>, long long> g_hazards;
static bool g_haz_on = false;
static inline bool same_wave(const Fiber* a, const Fiber* b) { return a->blk == b->blk && a->wave == b->wave && a != b; }
static void hazard_access(const void* addr, int bytes, bool store, const void* pc) {
    const Fiber* f = g_cur;
    if (!f || !g_haz_on) return;
    // (a lane's own stack is private: skip it)
    const char* a = static_cast<const char*>(addr);
    if (a >= f->stack && a < f->stack + STACK_BYTES) return;
    if (!g_shadow) g_shadow = new std::map<size_t, Shadow>();
    for (int o = 0; o < bytes; o += 4) {
        Shadow& s = (*g_shadow)[(reinterpret_cast<size_t>(a) + o) >> 2];
        if (store) {
            if (s.wf && s.wn == f->nops && same_wave(s.wf, f) ) ++g_hazards[{pc, s.ws}];
            for (int k = 0; k < 2; ++k)
                if (s.rf[k] && s.rn[k] == f->nops && same_wave(s.rf[k], f)) ++g_hazards[{pc, s.rs[k]}];
            s.wf = f; s.wn = f->nops; s.ws = pc;
        } else {
            if (s.wf && s.wn == f->nops && same_wave(s.wf, f)) ++g_hazards[{pc, s.ws}];
            const int k = (s.rf[0] == f || !s.rf[0]) ? 0 : 1;
            s.rf[k] = f; s.rn[k] = f->nops; s.rs[k] = pc;
        }
    }
}
}  // namespace emu
#define EMU_COV(n) \
    extern "C" void __sanitizer_cov_load##n(void* a) { emu::hazard_access(a, n, false, __builtin_return_address(0)); } \
    extern "C" void __sanitizer_cov_store##n(void* a) { emu::hazard_access(a, n, true, __builtin_return_address(0)); }
EMU_COV(1) EMU_COV(2) EMU_COV(4) EMU_COV(8) EMU_COV(16)
// ---- basic-block coverage of the kernel sources (build_emu.py --coverage: trace-pc-guard + pc-table) --------------------------
// Which blocks of the DEVICE code do the emulator tests execute?  (There is no coverage tool for gfx950 code; here the same source
// runs as host code.)  Guards and the table of block addresses are parallel arrays; at exit the table is written with a hit flag
// per block to $CILQR_EMU_COV_DIR/<pid>.cov — scripts/emu_coverage.py symbolises and merges.
namespace emu {
static unsigned* g_guard_beg = nullptr; static unsigned* g_guard_end = nullptr;
static const unsigned long* g_pcs_beg = nullptr; static const unsigned long* g_pcs_end = nullptr;
static std::vector<unsigned char>* g_hit = nullptr;
static void cov_dump() {
    const char* dir = std::getenv("CILQR_EMU_COV_DIR");
    if (!dir || !g_hit || !g_pcs_beg) return;
    char path[512];
    std::snprintf(path, sizeof(path), "%s/%d.cov", dir, (int)getpid());
    FILE* f = std::fopen(path, "w");
    if (!f) return;
    Dl_info info;
    unsigned long base = 0;
    if (dladdr(reinterpret_cast<const void*>(&cov_dump), &info)) base = reinterpret_cast<unsigned long>(info.dli_fbase);
    const size_t n = (size_t)(g_pcs_end - g_pcs_beg) / 2;
    for (size_t i = 0; i < n && i + 1 < g_hit->size(); ++i)
        std::fprintf(f, "%lx %d\n", g_pcs_beg[2 * i] - base, (int)(*g_hit)[i + 1]);
    std::fclose(f);
}
}  // namespace emu
extern "C" void __sanitizer_cov_trace_pc_guard_init(unsigned* start, unsigned* stop) {
    if (start == stop || *start) return;
    emu::g_guard_beg = start; emu::g_guard_end = stop;
    unsigned id = 0;
    for (unsigned* g = start; g < stop; ++g) *g = ++id;
    emu::g_hit = new std::vector<unsigned char>(id + 2, 0);
    std::atexit(emu::cov_dump);
}
extern "C" void __sanitizer_cov_trace_pc_guard(unsigned* guard) {
    if (!*guard) return;
    if (emu::g_hit && *guard < emu::g_hit->size()) (*emu::g_hit)[*guard] = 1;
    *guard = 0;  // (once is enough)
}
extern "C" void __sanitizer_cov_pcs_init(const unsigned long* beg, const unsigned long* end) { emu::g_pcs_beg = beg; emu::g_pcs_end = end; }
extern "C" void cilqr_emu_hazards_enable(int on) { emu::g_haz_on = on != 0; if (emu::g_shadow) emu::g_shadow->clear(); }
extern "C" int cilqr_emu_hazards(const void** a, const void** b, long long* n, int cap) {
    int i = 0;
    for (const auto& kv : emu::g_hazards) { if (i >= cap) break; a[i] = kv.first.first; b[i] = kv.first.second; n[i] = kv.second; ++i; }
    return i;
}

// what the run did (tests read it): launches, cross-lane operations, resolutions with lanes waiting at more than one site, ...
extern "C" void cilqr_emu_stats(long long out[8]) {
    out[0] = emu::g_stats.launches; out[1] = emu::g_stats.xops; out[2] = emu::g_stats.groups;
    out[3] = emu::g_stats.split_resolutions; out[4] = emu::g_stats.partial_barriers; out[5] = emu::g_stats.readlane_inactive;
    out[6] = (long long)emu::g_stats.split_sites.size(); out[7] = emu::g_stats.preemptions;
}
extern "C" void cilqr_emu_probe_counts(long long out[16]) { for (int i = 0; i < 16; ++i) out[i] = emu::g_probes[i]; }
extern "C" int cilqr_emu_inactive_sites(const void** a, long long* n, int cap) {
    int i = 0;
    for (const auto& kv : emu::g_stats.inactive_sites) { if (i >= cap) break; a[i] = kv.first; n[i] = kv.second; ++i; }
    return i;
}
// the distinct (resolved site, other site waited at) pairs of split resolutions, for addr2line
extern "C" int cilqr_emu_split_sites(const void** a, const void** b, long long* n, int cap) {
    int i = 0;
    for (const auto& kv : emu::g_stats.split_sites) {
        if (i >= cap) break;
        a[i] = kv.first.first; b[i] = kv.first.second; n[i] = kv.second; ++i;
    }
    return i;
}

// ---- the HIP host API ------------------------------------------------------------------------------------------------------
struct emu_stream { int id; };
struct emu_event { long long t; };
// What the HOST side asks of the runtime (cilqr_emu_api_counts): on the device every one of these calls costs microseconds — the
// budget of the latency-bound entry points (the single-ego solve of the drop-in class) is counted here, where no clock is.
enum { API_MALLOC, API_FREE, API_MEMCPY_SYNC, API_MEMCPY_ASYNC, API_MEMSET, API_SYNC, API_EVENT_RECORD, API_STREAM_WAIT, API_BYTES_COPIED, API_N };
static long long g_api[API_N];
extern "C" void cilqr_emu_api_counts(long long out[10]) {
    out[0] = emu::g_stats.launches;
    for (int i = 0; i < API_N; ++i) out[1 + i] = g_api[i];
}
int emu_blocks_per_cu() { return std::max(1, emu::env_int("CILQR_EMU_BLOCKS_PER_CU", 8)); }
// CILQR_EMU_DEVICES: how many devices the emulated box shows (cilqr_amd::ShardedSolver, one handle and host thread per device)
hipError_t hipGetDeviceCount(int* n) { *n = std::max(1, emu::env_int("CILQR_EMU_DEVICES", 1)); return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    p->multiProcessorCount = std::max(1, emu::env_int("CILQR_EMU_CUS", 1));
    std::snprintf(p->name, sizeof(p->name), "wave64 emulator (tests/emu)");
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    ++g_api[API_MALLOC];
    // 0xff fill: memory the kernels never wrote reads as NaNs, not as zeros (as the poisoned-scratch stress runs do on the GPU)
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
    std::memset(q, 0xff, n);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { ++g_api[API_FREE]; std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { ++g_api[API_MEMCPY_SYNC]; g_api[API_BYTES_COPIED] += (long long)n; std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { ++g_api[API_MEMCPY_ASYNC]; g_api[API_BYTES_COPIED] += (long long)n; std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { ++g_api[API_MEMSET]; std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { ++g_api[API_MEMSET]; std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) {
    ++g_api[API_MEMSET];
    for (size_t i = 0; i < count; ++i) static_cast<int*>(d)[i] = v;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
hipError_t hipDeviceSynchronize() { ++g_api[API_SYNC]; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { ++g_api[API_SYNC]; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu_stream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new emu_stream{2}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { ++g_api[API_STREAM_WAIT]; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ++g_api[API_EVENT_RECORD]; e->t = emu::clock_ticks(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { ++g_api[API_SYNC]; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t) * 1e-6f; return hipSuccess; }
