/* CPU sanitizer run of the oracles (AddressSanitizer + UBSan): block and streaming WSOLA chain, vocoder oracle.
 * Built and run by tests/test_oracle_sanitize.py; GPU sanitizers are not available on the pool. */
#include "nae_oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(void)
{
    /* atan2 revision 2 at the edges of its domain: huge finite bins, Inf, NaN, zeros (UBSan: no float -> int conversion out of range) */
    {
        const float edge[] = {0.0f, -0.0f, 1e-38f, 1e-35f, 1.0f, -1.0f, 1.2676506e30f, 3e38f, -3e38f, 2e38f, (float)(1.0 / 0.0), -(float)(1.0 / 0.0), (float)(0.0 / 0.0)};
        const unsigned n = sizeof edge / sizeof edge[0];
        long acc = 0;
        for (unsigned a = 0; a < n; a++)
            for (unsigned b = 0; b < n; b++) acc += orc_atan2_q32(edge[a], edge[b]) & 1;
        printf("atan2 edges: %ld\n", acc);
    }
    const int cases[][3] = {{48000, 2, 30000}, {44100, 1, 25000}, {8000, 2, 9000}, {22050, 1, 500}, {48000, 2, 0}};
    const double rp[][2] = {{1.0, 1.189207115}, {1.0, 0.8}, {1.5, 0.6666666}, {0.8, 1.0}, {1.0, 1.0}, {2.0, 0.5}};
    for (unsigned c = 0; c < sizeof cases / sizeof cases[0]; c++)
        for (unsigned r = 0; r < sizeof rp / sizeof rp[0]; r++) {
            const int sr = cases[c][0], ch = cases[c][1];
            const size_t L = (size_t)cases[c][2];
            float* x = malloc((L * ch + 1) * sizeof(float));
            orc_fill_uniform(x, L * ch, 77 + c);
            /* block */
            size_t bound = orc_st_out_bound(L, rp[r][0], rp[r][1]) + 64, n = 0;
            float* y = malloc(bound * ch * sizeof(float));
            if (orc_st_process_f32(x, L, ch, sr, rp[r][0], rp[r][1], y, &n)) return 1;
            /* streaming with interleaved receives */
            orc_st* s;
            if (orc_st_create(sr, ch, rp[r][0], rp[r][1], &s)) return 2;
            float* z = malloc(bound * ch * sizeof(float));
            size_t got = 0;
            for (size_t a = 0; a < L; a += 1152) {
                orc_st_put(s, x + a * ch, L - a < 1152 ? L - a : 1152);
                got += orc_st_receive(s, z + got * ch, 700);
            }
            orc_st_flush(s);
            got += orc_st_receive(s, z + got * ch, bound - got);
            orc_st_destroy(s);
            /* the PV oracle too */
            orc_stretch_plan pl;
            if (orc_stretch_plan_make(rp[r][0], rp[r][1], L, &pl) == 0) {
                float* w = malloc((pl.out_len + 1) * ch * sizeof(float));
                orc_stretch_f32(x, L, ch, rp[r][0], rp[r][1], w);
                free(w);
            }
            printf("sr %d ch %d L %zu rate %.2f pitch %.3f: block %zu stream %zu\n", sr, ch, L, rp[r][0], rp[r][1], n, got);
            free(x); free(y); free(z);
        }
    return 0;
}
