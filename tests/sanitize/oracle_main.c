/* AddressSanitizer / UndefinedBehaviorSanitizer job for the CPU oracle (sanitizers row): TEST
 * INFRASTRUCTURE.  Built by tests/test_sanitizers.py from oracle/cilqr_oracle.c with
 * -fsanitize=address,undefined -fno-sanitize-recover=all and run: barrier and ALM solves with traces and decision
 * margins, a warm-started tick sequence, every piecewise entry point, the OpenMP batch driver.  Exit code 0 and the
 * line SANITIZE-ORACLE-OK mean no finding. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cilqr_oracle.h"

#define NL 900
#define NT_ 120
#define NM 2

static void fill_params(orc_params* p, int N, int solve_type, int rp) {
    memset(p, 0, sizeof(*p));
    p->N = N; p->max_iter = 60; p->solve_type = solve_type; p->reference_point = rp; p->use_last_solution = 1;
    p->dt = 0.1; p->w_pos = 1.0; p->w_vel = 1.0; p->w_yaw = 1.0; p->w_acc = 1.0; p->w_stl = 20.0;
    p->obstacle_exp_q1 = 5.5; p->obstacle_exp_q2 = 5.75; p->state_exp_q1 = 3.0; p->state_exp_q2 = 3.5;
    p->alm_rho_init = 1.0; p->alm_gamma = 0.1; p->max_rho = 100.0; p->max_mu = 1000.0;
    p->init_lamb = 0.0; p->lamb_decay = 0.5; p->lamb_amplify = 2.0; p->max_lamb = 1000.0;
    p->convergence_threshold = 0.01; p->accept_step_threshold = 0.5;
    p->wheelbase = 2.9; p->width = 2.0; p->length = 4.8; p->velo_max = 12.0; p->velo_min = 0.0; p->yaw_lim = 1.57;
    p->acc_max = 2.0; p->acc_min = -3.0; p->stl_lim = 0.3; p->d_safe = 0.8;
}

int main(void) {
    static double lx[NL], ly[NL], lyaw[NL], obs[NM * NT_ * 3];
    for (int i = 0; i < NL; ++i) {
        const double s = 0.1 * i;
        lx[i] = -10.0 + s;
        ly[i] = 2.0 * sin(s / 25.0);
        lyaw[i] = atan2(2.0 / 25.0 * cos(s / 25.0), 1.0);
    }
    for (int j = 0; j < NM; ++j)
        for (int k = 0; k < NT_; ++k) {
            double* o = obs + (j * NT_ + k) * 3;
            o[0] = 20.0 + 15.0 * j + 0.4 * k; o[1] = 3.6 * j; o[2] = 0.0;
        }
    orc_scene sc;
    memset(&sc, 0, sizeof(sc));
    sc.lane_x = lx; sc.lane_y = ly; sc.lane_yaw = lyaw; sc.L = NL; sc.M = NM; sc.obs = obs; sc.T = NT_; sc.tick = 0;
    sc.road_borders[0] = 5.4; sc.road_borders[1] = -1.8; sc.ref_velo = 8.0;
    int total_iters = 0;
    for (int st = 0; st < 2; ++st)
        for (int rp = 0; rp < 2; ++rp) {
            const int N = st ? 25 : 40;
            orc_params p;
            fill_params(&p, N, st, rp);
            orc_solver* s = orc_create(&p);
            double* u = malloc(sizeof(double) * 2 * N);
            double* x = malloc(sizeof(double) * 4 * (N + 1));
            orc_trace_rec* tr = malloc(sizeof(orc_trace_rec) * 8);  /* deliberately shorter than the solve */
            orc_margin_rec* mg = malloc(sizeof(orc_margin_rec) * 5);
            double x0[4] = {0.0, 0.4, 7.0, 0.02};
            orc_set_margin_buffer(s, mg, 5);
            for (int tick = 0; tick < 6; ++tick) {  /* closed loop with warm starts */
                orc_result res;
                sc.tick = tick;
                if (orc_solve(s, x0, &sc, u, x, &res, tr, 8) != 0) return 2;
                total_iters += res.iters;
                memcpy(x0, x + 4, sizeof(x0));
            }
            orc_set_margin_buffer(s, NULL, 0);
            sc.tick = 0;
            /* piecewise entry points on the last trajectory */
            double J = orc_total_cost(s, u, x, &sc);
            double* l_x = malloc(sizeof(double) * 4 * (N + 1));
            double* l_u = malloc(sizeof(double) * 2 * N);
            double* l_xx = malloc(sizeof(double) * 16 * (N + 1));
            double* l_uu = malloc(sizeof(double) * 4 * N);
            orc_cost_derivatives(s, u, x, &sc, l_x, l_u, l_xx, l_uu);
            double* d = malloc(sizeof(double) * 2 * N);
            double* K = malloc(sizeof(double) * 8 * N);
            double dV[2];
            orc_backward_pass(s, u, x, 0.5, &sc, d, K, dV);
            double* nu = malloc(sizeof(double) * 2 * N);
            double* nx = malloc(sizeof(double) * 4 * (N + 1));
            orc_forward_pass(&p, u, x, d, K, 0.25, nu, nx);
            double* A = malloc(sizeof(double) * 16 * N);
            double* Bm = malloc(sizeof(double) * 8 * N);
            orc_model_derivatives(x, u, p.dt, p.wheelbase, N, rp, A, Bm);
            double* ref = malloc(sizeof(double) * 3 * (N + 1));
            int32_t* idx = malloc(sizeof(int32_t) * (N + 1));
            orc_ref_exact_points(x, N + 1, &sc, ref, idx);
            if (!(J == J)) return 3;
            free(u); free(x); free(tr); free(mg); free(l_x); free(l_u); free(l_xx); free(l_uu); free(d); free(K); free(nu);
            free(nx); free(A); free(Bm); free(ref); free(idx);
            orc_destroy(s);
        }
    /* the OpenMP batch driver: 2 parameter sets, ids, ticks */
    {
        enum { B = 24 };
        orc_params ps[2];
        fill_params(&ps[0], 30, 0, 0);
        fill_params(&ps[1], 30, 0, 1);
        ps[0].use_last_solution = ps[1].use_last_solution = 0;
        double x0[B * 4];
        int32_t pid[B], tick[B];
        for (int b = 0; b < B; ++b) {
            x0[4 * b] = -2.0 + 0.3 * b; x0[4 * b + 1] = ((b & 1) ? -1 : 1) * (0.1 + 0.03 * b); x0[4 * b + 2] = 6.0 + 0.1 * b;
            x0[4 * b + 3] = 0.01 * (b % 5);
            pid[b] = b % 2; tick[b] = b % 7;
        }
        double* u = malloc(sizeof(double) * B * 2 * 30);
        double* x = malloc(sizeof(double) * B * 4 * 31);
        orc_result* res = malloc(sizeof(orc_result) * B);
        if (orc_solve_batch(ps, 2, &sc, 1, B, x0, NULL, pid, tick, 2, u, x, res) != 0) return 4;
        for (int b = 0; b < B; ++b) total_iters += res[b].iters;
        free(u); free(x); free(res);
    }
    printf("SANITIZE-ORACLE-OK %d iterations\n", total_iters);
    return 0;
}
