// Test infrastructure (built and run by tests/test_oracle_golden.py::test_a_ball_keeps_its_landing_point_along_a_free_flight):
// the statement the k-frame pair kernel's `known` landing points rest on (pz_physics.hpp, pair_frame_head), checked on EVERY
// ball B of the landing table's domain against the oracle's predictor: if the world step (physics.py:359-431) moves B to
// W(B) without touching the ground, then P(W(B)) == P(B) -- unless B is over the net at y == 192 or has no x velocity
// there (flight_keeps_landing_point).   gcc -O2 -fopenmp tests/flight_rule.c oracle/pz_oracle.c -lm
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int32_t pzo_expected_landing_x(int32_t x, int32_t y, int32_t xv, int32_t yv);
// process_collision_between_ball_and_world_and_set_ball_position (physics.py:359-431), position / velocity part
static int world(int *x, int *y, int *xv, int *yv) {
    int fx = *x + *xv;
    if (fx < 20 || fx > 432) *xv = -*xv;
    if (*y + *yv < 0) *yv = 1;
    if (abs(*x - 216) < 25 && *y > 176) {
        if (*y <= 192) { if (*yv > 0) *yv = -*yv; }
        else { if (*x < 216) *xv = -abs(*xv); else *xv = abs(*xv); }
    }
    if (*y + *yv > 252) { *yv = -*yv; *y = 252; return 1; }
    *y += *yv; *x += *xv; *yv += 1; return 0;
}
int main(int argc, char **argv) {
    int with_rule = argc > 1 ? atoi(argv[1]) : 1;
    long long total = 0, checked = 0, bad = 0, excluded = 0, bad_excl = 0;
    #pragma omp parallel for schedule(dynamic) reduction(+:total,checked,bad,excluded,bad_excl)
    for (int yv = -96; yv <= 96; ++yv)
        for (int xi = 0; xi < 23; ++xi) {
            int xv0 = xi == 0 ? -20 : (xi == 22 ? 20 : xi - 11);
            for (int y = 0; y <= 252; ++y)
                for (int x = 20; x <= 432; ++x) {
                    ++total;
                    int bx = x, by = y, bxv = xv0, byv = yv;
                    if (world(&bx, &by, &bxv, &byv)) continue;  // ground: the round ends
                    if (bx < 20 || bx > 432 || by < 0 || by > 252 || abs(byv) > 96) continue;  // W(B) outside the domain
                    int axv = abs(bxv); if (!(axv <= 10 || axv == 20)) continue;
                    int excl = (abs(x - 216) < 25 && y == 192) || (xv0 == 0 && abs(x - 216) < 25);
                    int p0 = pzo_expected_landing_x(x, y, xv0, yv), p1 = pzo_expected_landing_x(bx, by, bxv, byv);
                    if (excl) { ++excluded; bad_excl += p0 != p1; continue; }
                    ++checked;
                    if (p0 != p1) { if (bad < 5) printf("violation: B=(%d,%d,%d,%d) P=%d  W(B)=(%d,%d,%d,%d) P=%d\n", x, y, xv0, yv, p0, bx, by, bxv, byv, p1); ++bad; }
                }
        }
    printf("states %lld, rule applies to %lld: violations %lld; excluded by the rule %lld (of which differ: %lld)\n", total, checked, bad, excluded, bad_excl);
    return bad != 0;
}
