"""Host-side layout arithmetic of the grouped kernel (csrc/cilqr_group.hpp), compiled for the HOST with hipcc
(--cuda-host-only: the __host__ __device__ helpers the dispatcher in cilqr_amd.hip calls) — no GPU needed.  Guards the LDS
budgets DESIGN.md section 3 states: a layout that silently grows past 20 KB per wavefront halves nothing visibly, it just drops a
CU from 8 resident blocks to 7."""
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
CSRC = ROOT / "toy-example-of-ilqr_amd" / "csrc"

PROG = r"""
#include "cilqr_group.hpp"
#include <cstdio>
using namespace cilqr;
int main() {
    for (int N = 1; N <= 255; ++N) {
        if (N <= 63)
            std::printf("S %d %zu %d %d %d\n", N, grp_lds_bytes(N, 0, 2), grp_expansion_doubles(N), grp_pg_doubles(N), kd_doubles(N, 1));
        else
            std::printf("L %d %zu %d %d %d %d\n", N, grpl_lds_bytes(N, 0, 2), grpl_shared_doubles(N, 0, 2), grpl_cs_doubles(N),
                        grpl_gring_doubles(2), grp_pg_doubles(N));
        std::printf("X %d %zu %zu %zu %zu %zu %zu\n", N, grp_scratch_doubles(N), grp_rows_offset(N), slab_doubles(N), first_trial_doubles(N),
                    grp_park_doubles(N), park_doubles(N));
    }
    std::printf("K %d %d %d %d %d %d\n", CILQR_GRPL_CHUNK, CILQR_KD, CILQR_GL_RING, CILQR_XCH, CILQR_GRP_ROW, CILQR_GRP_Q_PER_TRAJECTORY);
    return 0;
}
"""


@pytest.fixture(scope="module")
def rows(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not pathlib.Path(hipcc).exists():
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("layout")
    (d / "layout.cpp").write_text(PROG)
    p = subprocess.run([hipcc, "-std=c++17", "--cuda-host-only", "-x", "hip", "-I", str(CSRC), "-I", str(ROOT / "include"),
                        str(d / "layout.cpp"), "-o", str(d / "layout")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = subprocess.run([str(d / "layout")], capture_output=True, text=True, timeout=60).stdout.split("\n")
    return [l.split() for l in out if l]


def fixed_of(L, N):
    return int(L[N][2]) - 8 * int(L[N][3]) + 8 * int(L[N][4])


def test_grouped_layouts_keep_eight_wavefronts_on_a_cu(rows):
    per_block = 163840 // 8
    S = {int(r[1]): r for r in rows if r[0] == "S"}
    L = {int(r[1]): r for r in rows if r[0] == "L"}
    # horizons up to 63: the lane window shares the expansion's area, so W = 0 is the whole block up to that size
    for N in (30, 50):
        fixed, exp_d = int(S[N][2]), int(S[N][3])
        assert fixed <= per_block, (N, fixed)
        # BASELINE's N = 50: 584 window samples (two doubles each) still fit the shared area / the 20 KB
        room = (per_block - (fixed - 8 * exp_d)) // 16
        assert room >= (570 if N == 50 else 700), (N, room)
    # the long layout: nothing of a horizon's length in the shared area — N = 100 with a window of >= 256 samples at 8 per CU,
    # N = 127 at 7 per CU
    fixed100 = int(L[100][2]) - 8 * int(L[100][3]) + 8 * int(L[100][4])
    assert (per_block - fixed100) // 16 >= 256, fixed100
    fixed127 = int(L[127][2]) - 8 * int(L[127][3]) + 8 * int(L[127][4])
    assert (163840 // 7 - fixed127) // 16 >= 128, fixed127
    # round 6: horizons of 128 ... 255 (four rows per lane) — the block with a 128-sample window stays under the 64 KB a launch may
    # ask for without opting in, and at least three blocks (six trajectories) share a CU at the cap
    for N in (128, 200, 255):
        fixedN = int(L[N][2]) - 8 * int(L[N][3]) + 8 * int(L[N][4])
        assert fixedN + 16 * 128 <= 65536, (N, fixedN)
    assert (163840 // 3 - fixed_of(L, 255)) // 16 >= 128
    K = [r for r in rows if r[0] == "K"][0]
    chunk, kd, ring, xch = int(K[1]), int(K[2]), int(K[3]), int(K[4])
    for N, r in L.items():
        shared, cs, gring = int(r[3]), int(r[4]), int(r[5])
        assert shared >= xch + 2 * ring and shared >= gring and shared >= cs, (N, r)  # the three tenants of the shared area
        assert gring == 2 * 2 * chunk * kd and cs % 2 == 0 and cs >= 3 * (N + 1)
        # a chunk of the gains ring = one full LDS-DMA of 64 sixteen-byte elements + a partial one
        assert 64 < 2 * chunk * kd // 2 <= 128


def test_scratch_and_parked_state_layouts(rows):
    K = [r for r in rows if r[0] == "K"][0]
    grow = int(K[5])
    for r in (r for r in rows if r[0] == "X"):
        N, scratch, rows_off, slab, first, gpark, park = (int(v) for v in r[1:])
        assert slab == 20 * first and first == 3 * ((N + 1 + 3) // 4) * 8
        assert rows_off % 32 == 0 and rows_off >= slab + first + 10 * N      # gains [N][10] behind the first-trial buffer
        assert scratch == rows_off + grow * (N + 1)                            # one 256-byte row per step behind them
        assert gpark >= park                                                    # one park buffer serves both kernels (max)
    assert int(K[6]) >= 8  # queue places per trajectory: sixteen slices of 100 iterations at 12 per slice need nine
