/* Diagnostic (not product, not a test): how many loop trips do the device's flight predictors
 * need on states sampled from play, per 64-game wavefront and frame?
 *
 *   gcc -O2 -fopenmp -DPZO_FLIGHT_TRACE -Ioracle -o /tmp/flight_trips tests/flight_trips.c -lm && /tmp/flight_trips
 *
 * Plays `waves` x 64 games of config 3 (player 2 = computer, player 1 random) with the oracle, records
 * every predictor call through the PZO_FLIGHT_TRACE hook, and replays the calls through a host model
 * of predict_landing_x<FULL_NET> (pz_physics.hpp).  A wavefront pays the maximum over its lanes, so
 * the figures of merit are the per-wave maxima, not the per-call means.
 */
#include "../oracle/pz_oracle.c"

#include <math.h>
#include <stdio.h>

enum { kWaves = 64, kLanes = 64, kSteps = 3000, kMaxCalls = 8 };

typedef struct { int kind, x, y, xv, yv; } Call;
static Call g_calls[kMaxCalls];
static int g_ncalls;

void pzo_flight_trace(int kind, int x, int y, int xv, int yv)
{
    if (g_ncalls < kMaxCalls) g_calls[g_ncalls++] = (Call){kind, x, y, xv, yv};
}

static int iterative(int full, int x, int y, int xv, int yv, int *iters)
{
    int count = 0;
    for (;;) {
        ++count;
        int fx = x + xv;
        if (fx < 20 || fx > 432) xv = -xv;
        if (y + yv < 0) yv = 1;
        if (abs(x - 216) < 25 && y > 176) {
            if (!full || y < 192) { if (yv > 0) yv = -yv; }
            else xv = (x < 216) ? -abs(xv) : abs(xv);
        }
        y += yv;
        if (y > 252 || count >= 1000) break;
        x += xv;
        yv += 1;
    }
    *iters = count;
    return x;
}

static int fh(int y, int yv, int m) { return y + m * yv + ((m * (m - 1)) >> 1); }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* event classes of the single iteration that closes a trip */
enum { EV_LAND, EV_WALL, EV_CEIL, EV_BOX, EV_PLAIN, EV_N };
static const char *kEvName[EV_N] = {"landing", "wall flip", "ceiling", "in net box", "plain (jump failed/short)"};

/* why a trip closed with a plain iteration: [jumped][inside box][K bound by x limit] */
static long g_plain_why[2][2][2];

static int fast(int full, int x, int y, int xv, int yv, int *trips, long *hist)
{
    int count = 0;
    *trips = 0;
    for (;;) {
        ++*trips;
        const int ascending = !full && yv < 0;
        const int left = !ascending && x <= 191, right = !ascending && x >= 241;
        const int outside = left | right;
        const int ymax = (outside | ascending) ? 252 : 176;
        const int rightward = xv > 0;
        const int axv = abs(xv);
        const int edge = rightward ? (left ? 191 : 432) : (right ? 241 : 20);
        const int room = rightward ? edge - x : x - edge;
        const int toward_box = (rightward ? left : right) ? 1 : 0;
        const float q = (float)room / (float)(axv ? axv : 1);
        const int kx = (axv ? (int)q : 1000) + toward_box;
        const float hb = (float)(2 * yv - 1);
        const int apex = y - ((yv * (yv - 1)) >> 1);
        const int to_ceiling = (yv < 0) & (apex < 0);
        const float c8 = 8.0f * (float)(to_ceiling ? -y : ymax - y);
        const float root = sqrtf(fmaxf(hb * hb + c8, 0.0f));
        int K = (int)(((to_ceiling ? -root : root) - hb) * 0.5f - 0.001f);
        K = imin(imin(K, kx), 1000 - 2 - count);
        K = ascending ? imin(K, 1 - yv) : K;
        const int ye = fh(y, yv, K);
        const int xe = x + K * xv;
        const int xl = xe - xv;
        const int lowest = fh(y, yv, imin(imax(-yv, 1), K));
        const int side_kept = left ? xl <= 191 : (right ? xl >= 241 : 1);
        const int ok = (K >= 2) & ((unsigned)y <= (unsigned)ymax) & (abs(yv) < 4096) &
                       ((unsigned)(xe - 20) <= 412u) & ((unsigned)ye <= (unsigned)ymax) & (lowest >= 0) & side_kept;
        if (ok) { x = xe; y = ye; yv += K; count += K; }

        ++count;
        int ev = EV_PLAIN;
        const int fx = x + xv;
        if (fx < 20 || fx > 432) { xv = -xv; ev = EV_WALL; }
        if (y + yv < 0) { yv = 1; ev = EV_CEIL; }
        if (abs(x - 216) < 25 && y > 176) {
            ev = EV_BOX;
            if (!full || y < 192) { if (yv > 0) yv = -yv; }
            else xv = (x < 216) ? -abs(xv) : abs(xv);
        }
        y += yv;
        if (y > 252 || count >= 1000) { if (hist) hist[EV_LAND]++; break; }
        if (hist) hist[ev]++;
        if (hist && ev == EV_PLAIN) g_plain_why[ok][!outside && !ascending][K == kx]++;
        x += xv;
        yv += 1;
    }
    return x;
}


/* candidate formulation: every trip ends with a real event (wall flip, ceiling clamp, net-box hit,
 * landing); passing over the net above its top is part of the jump */
static long g_fail2[8];
static int fast2(int full, int x, int y, int xv, int yv, int *trips, long *hist)
{
    int count = 0;
    *trips = 0;
    for (;;) {
        ++*trips;
        const int rightward = xv > 0;
        const int axv = abs(xv);
        const float r = 1.0f / (float)(axv ? axv : 1);
        const int room = rightward ? 432 - x : x - 20;
        const int kw = axv ? (int)((float)room * r + 0.001f) : 1000;
        const float hb = (float)(2 * yv - 1);
        const int apex = y - ((yv * (yv - 1)) >> 1);
        const int to_ceiling = (yv < 0) & (apex < 0);
        const float hb2 = hb * hb;
        const int d1 = rightward ? 192 - x : x - 240, d2 = rightward ? 240 - x : x - 192;
        const int in_cols = d1 <= 0 && d2 >= 0;
        const float root = sqrtf(fmaxf(hb2 + 8.0f * (float)(to_ceiling ? -y : 252 - y), 0.0f));
        const int Ky = (int)(((to_ceiling ? -root : root) - hb) * 0.5f - 0.001f);
        int K = imax(imin(imin(kw, Ky), 998 - count), 0);
        /* positions m (0-based, X(m) = x + m xv) inside the box columns 192..240: m1..m2 */
        const int m1 = (axv && d1 > 0) ? (int)((float)(d1 + axv - 1) * r + 0.001f) : 0;
        const int m2 = axv ? (d2 >= 0 ? (int)((float)d2 * r + 0.001f) : -1) : ((d1 <= 0 && d2 >= 0) ? 1000 : -1);
        const int lo = full ? m1 : imax(m1, 1 - yv);
        const int hi = imin(m2, K - 1);
        const int Ylo = fh(y, yv, lo);
        const int box_hit = lo <= hi && (Ylo > 176 || fh(y, yv, hi) > 176);
        const float root176 = sqrtf(fmaxf(hb2 + 8.0f * (float)(176 - y), 0.0f));
        const int K176 = (int)((root176 - hb) * 0.5f - 0.001f);
        K = box_hit ? (Ylo > 176 ? lo : K176 + 1) : K;
        /* exact verification */
        const int ye = fh(y, yv, K), xe = x + K * xv;
        const int lowest = fh(y, yv, imin(imax(-yv, 1), K));
        const int hi2 = imin(m2, K - 1);
        const int box_ok = lo > hi2 || (Ylo <= 176 && fh(y, yv, hi2) <= 176);
        const int xb = x + (m1 - 1) * xv, xa = x + (m2 + 1) * xv;
        const int cols_ok = (m1 == 0 || (rightward ? xb <= 191 : xb >= 241)) &&
                            (m2 >= K - 1 || (axv ? (rightward ? xa >= 241 : xa <= 191) : 1));
        const int ok = (K >= 1) & ((unsigned)y <= 252u) & (abs(yv) < 4096) & ((unsigned)(xe - 20) <= 412u) &
                       ((unsigned)ye <= 252u) & (lowest >= 0) & box_ok & cols_ok;
        if (K >= 1 && !ok) {
            g_fail2[(!((unsigned)(xe - 20) <= 412u)) + 2 * (!((unsigned)ye <= 252u)) ? 1 : (!box_ok ? 2 : (!cols_ok ? 3 : (lowest < 0 ? 4 : 5)))]++;
        }
        if (ok) { x = xe; y = ye; yv += K; count += K; }

        ++count;
        int ev = EV_PLAIN;
        const int fx = x + xv;
        if (fx < 20 || fx > 432) { xv = -xv; ev = EV_WALL; }
        if (y + yv < 0) { yv = 1; ev = EV_CEIL; }
        if (abs(x - 216) < 25 && y > 176) {
            ev = EV_BOX;
            if (!full || y < 192) { if (yv > 0) yv = -yv; }
            else xv = (x < 216) ? -abs(xv) : abs(xv);
        }
        y += yv;
        if (y > 252 || count >= 1000) { if (hist) hist[EV_LAND]++; break; }
        if (hist) hist[ev]++;
        x += xv;
        yv += 1;
    }
    return x;
}

#ifndef FAST
#define FAST fast
#endif
int main(void)
{
    const int n = kWaves * kLanes;
    pzo_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.winning_score = 15;
    cfg.p2_computer = 1;
    cfg.auto_reset = 1;
    int32_t *state = calloc((size_t)PZO_W * n, 4);
    pzo_init(state, n, n, &cfg);
    pzo_reset(state, n, n, &cfg, NULL, NULL, NULL, NULL);
    int32_t o1[PZO_OBS], o2[PZO_OBS], r1, r2;
    uint8_t term;

    double sumA = 0, sumB = 0, sumC = 0, sumA_it = 0, sumC_it = 0, sumB_it = 0;
    double meanA = 0, meanC = 0;
    long nA = 0, nC = 0, wavesteps = 0, wavesC = 0, wavesB = 0;
    long histA[EV_N] = {0}, histC[EV_N] = {0};     /* all trips */
    long histAmax[EV_N] = {0}, histCmax[EV_N] = {0}; /* trips of the wave's longest flight */
    long tripsHistA[64] = {0}, tripsHistC[64] = {0};
    for (int s = 0; s < kSteps; ++s) {
        for (int w = 0; w < kWaves; ++w) {
            int maxA = 0, maxB = 0, maxC = 0, maxAit = 0, maxBit = 0, maxCit = 0;
            long hA[EV_N] = {0}, hC[EV_N] = {0};
            for (int l = 0; l < kLanes; ++l) {
                const int i = w * kLanes + l;
                int32_t a1, a2;
                pzo_random_actions(&a1, &a2, 1, i, 1, (uint64_t)s, 18);
                pzo_config c = cfg;
                c.env_id_base = i;
                g_ncalls = 0;
                pzo_step(state + i, 1, n, &c, &a1, &a2, o1, o2, &r1, &r2, &term, NULL, 1);
                int landing_calls = 0;
                for (int k = 0; k < g_ncalls; ++k) {
                    const Call *cl = &g_calls[k];
                    if (cl->kind == 0) {
                        int it, tr;
                        long h[EV_N] = {0};
                        const int ref = iterative(1, cl->x, cl->y, cl->xv, cl->yv, &it);
                        if (FAST(1, cl->x, cl->y, cl->xv, cl->yv, &tr, h) != ref) { puts("MODEL MISMATCH"); return 1; }
                        if (landing_calls++ < 2) {
                            if (landing_calls == 2) continue; /* :314-315 evaluates twice on the same ball */
                            for (int e = 0; e < EV_N; ++e) histA[e] += h[e];
                            meanA += tr; ++nA; tripsHistA[imin(tr, 63)]++;
                            if (tr > maxA) { maxA = tr; memcpy(hA, h, sizeof h); }
                            maxAit = imax(maxAit, it);
                        } else {
                            maxB = imax(maxB, tr);
                            maxBit = imax(maxBit, it);
                        }
                    } else {
                        for (int c6 = 0; c6 < 6; ++c6) {
                            const int xdir = c6 < 3 ? 1 : 0, ydir = (c6 % 3) - 1;
                            const int xv = cl->x < 216 ? (xdir + 1) * 10 : -(xdir + 1) * 10;
                            const int yv = cl->yv * ydir * 2;
                            int it, tr;
                            long h[EV_N] = {0};
                            const int ref = iterative(0, cl->x, cl->y, xv, yv, &it);
                            if (FAST(0, cl->x, cl->y, xv, yv, &tr, h) != ref) { puts("MODEL MISMATCH"); return 1; }
                            for (int e = 0; e < EV_N; ++e) histC[e] += h[e];
                            meanC += tr; ++nC; tripsHistC[imin(tr, 63)]++;
                            if (tr > maxC) { maxC = tr; memcpy(hC, h, sizeof h); }
                            maxCit = imax(maxCit, it);
                        }
                    }
                }
            }
            if (getenv("FLIGHT_TRIPS_DUMP")) {
                static FILE *dump;
                if (!dump) dump = fopen(getenv("FLIGHT_TRIPS_DUMP"), "w");
                fprintf(dump, "%d %d %d\n", maxA, maxC, maxB);
            }
            ++wavesteps;
            sumA += maxA; sumB += maxB; sumC += maxC;
            sumA_it += maxAit; sumB_it += maxBit; sumC_it += maxCit;
            wavesC += maxC > 0; wavesB += maxB > 0;
            for (int e = 0; e < EV_N; ++e) { histAmax[e] += hA[e]; histCmax[e] += hC[e]; }
        }
    }
    printf("%d waves x %d frames (config 3)\n", kWaves, kSteps);
    printf("per call: landing mean %.2f trips (%ld calls), power-hit candidate mean %.2f trips (%ld flights)\n",
           meanA / nA, nA, meanC / nC, nC);
    printf("per wave and frame, trips of the longest lane (iterations of the reference loop in brackets):\n");
    printf("  landing A          %.2f  (%.1f)\n", sumA / wavesteps, sumA_it / wavesteps);
    printf("  candidates         %.2f  (%.1f)   waves with a decider: %.1f %%\n", sumC / wavesteps, sumC_it / wavesteps,
           100.0 * wavesC / wavesteps);
    printf("  landing B          %.2f  (%.1f)   waves with a collision: %.1f %%\n", sumB / wavesteps, sumB_it / wavesteps,
           100.0 * wavesB / wavesteps);
    printf("what closes a trip (all flights | the wave's longest flight):\n");
    for (int e = 0; e < EV_N; ++e)
        printf("  %-28s A %9ld | %9ld     C %9ld | %9ld\n", kEvName[e], histA[e], histAmax[e], histC[e], histCmax[e]);
    for (int j = 0; j < 2; ++j)
        for (int b = 0; b < 2; ++b)
            for (int kx = 0; kx < 2; ++kx)
                printf("  plain closure: jumped=%d in_box_columns=%d K_bound_by_x=%d : %ld\n", j, b, kx, g_plain_why[j][b][kx]);
    printf("fast2 failed verifications: x/y range %ld, box %ld, cols %ld, lowest %ld, other %ld\n", g_fail2[1], g_fail2[2], g_fail2[3], g_fail2[4], g_fail2[5]);
    printf("trips histogram A:");
    for (int t = 1; t < 24; ++t) printf(" %ld", tripsHistA[t]);
    printf("\ntrips histogram C:");
    for (int t = 1; t < 40; ++t) printf(" %ld", tripsHistC[t]);
    puts("");
    return 0;
}
