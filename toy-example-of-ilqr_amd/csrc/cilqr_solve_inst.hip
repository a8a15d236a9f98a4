// cilqr_solve_inst.hip — explicit instantiations of k_solve.  Compiled once per group (-DCILQR_INST_GROUP=g, g = 0 ..
// CILQR_SOLVE_GROUPS - 1): each compilation carries the builds of its group (see CILQR_SOLVE_VARIANTS).
#include "cilqr_kernels.hpp"

#ifndef CILQR_INST_GROUP
#error "compile with -DCILQR_INST_GROUP=<group>"
#endif

#define CILQR_X_INST(g, DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP) CILQR_X_INST_##g(DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP)
#define CILQR_INST_DEF(DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP) \
    template __global__ void k_solve<DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP> CILQR_SOLVE_SIGNATURE;
#define CILQR_INST_NONE(DBG, NCH, ALM, HELP, PROF, WPS, NTP, NC, LG, SHARE, RES, LOOP)
#define CILQR_X_INST_0 CILQR_INST_NONE
#define CILQR_X_INST_1 CILQR_INST_NONE
#define CILQR_X_INST_2 CILQR_INST_NONE
#define CILQR_X_INST_3 CILQR_INST_NONE
#define CILQR_X_INST_4 CILQR_INST_NONE
#define CILQR_X_INST_5 CILQR_INST_NONE
#define CILQR_X_INST_6 CILQR_INST_NONE
#define CILQR_X_INST_7 CILQR_INST_NONE
#if CILQR_INST_GROUP == 0
#undef CILQR_X_INST_0
#define CILQR_X_INST_0 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 1
#undef CILQR_X_INST_1
#define CILQR_X_INST_1 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 2
#undef CILQR_X_INST_2
#define CILQR_X_INST_2 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 3
#undef CILQR_X_INST_3
#define CILQR_X_INST_3 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 4
#undef CILQR_X_INST_4
#define CILQR_X_INST_4 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 5
#undef CILQR_X_INST_5
#define CILQR_X_INST_5 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 6
#undef CILQR_X_INST_6
#define CILQR_X_INST_6 CILQR_INST_DEF
#elif CILQR_INST_GROUP == 7
#undef CILQR_X_INST_7
#define CILQR_X_INST_7 CILQR_INST_DEF
#else
#error "CILQR_INST_GROUP out of range"
#endif
CILQR_SOLVE_VARIANTS(CILQR_X_INST)

// the grouped builds (k_solve_grp: two trajectories per wavefront): compile-time horizons 50 (BASELINE) and 30 (the reference's
// own YAMLs), any horizon up to 63; each also as the closed planning loop in one launch.  The two kernels of a horizon share
// a compilation: their phases (expansion + sweep, trial costs, rollout pass, initial trajectory) are out-of-line functions of
// the horizon alone, 190 KB of code that would otherwise be in the library twice.
#if CILQR_INST_GROUP == 6
template __global__ void k_solve_grp<50, 2> CILQR_GRP_SIGNATURE;
template __global__ void k_solve_grp<50, 2, true> CILQR_GRP_SIGNATURE;
#elif CILQR_INST_GROUP == 7
template __global__ void k_solve_grp<0, 2> CILQR_GRP_SIGNATURE;
template __global__ void k_solve_grp<0, 2, true> CILQR_GRP_SIGNATURE;
#elif CILQR_INST_GROUP == 5
template __global__ void k_solve_grp<30, 2> CILQR_GRP_SIGNATURE;
template __global__ void k_solve_grp<30, 2, true> CILQR_GRP_SIGNATURE;
// horizons of 64 ... 127 (two rows per lane, the long layout of cilqr_group.hpp): BASELINE's 100, any
#elif CILQR_INST_GROUP == 3
template __global__ void k_solve_grp<100, 2, false, 2> CILQR_GRP_SIGNATURE;
#elif CILQR_INST_GROUP == 4
template __global__ void k_solve_grp<0, 2, false, 2> CILQR_GRP_SIGNATURE;
#endif
#if CILQR_INST_GROUP == 1
// round 6: horizons of 128 ... 255 — the long layout with FOUR rows per lane (cs:19: N is any int upstream); barrier and ALM
template __global__ void k_solve_grp<0, 2, false, 4> CILQR_GRP_SIGNATURE;
// ... and the closed planning loop in one launch on the long layout (two / four rows per lane), barrier mode
template __global__ void k_solve_grp<0, 2, true, 4> CILQR_GRP_SIGNATURE;
template __global__ void k_solve_grp<0, 2, true, 2> CILQR_GRP_SIGNATURE;
#endif
#if CILQR_INST_GROUP == 0
template __global__ void k_solve_grp<0, 2, false, 4, true, true> CILQR_GRP_SIGNATURE;
#endif
#if CILQR_INST_GROUP == 2
// (prepared, unmeasured) augmented Lagrangian in pairs: the long layout with one row per lane / two rows per lane
template __global__ void k_solve_grp<0, 2, false, 1, true, true> CILQR_GRP_SIGNATURE;
template __global__ void k_solve_grp<0, 2, false, 2, true, true> CILQR_GRP_SIGNATURE;
#endif
