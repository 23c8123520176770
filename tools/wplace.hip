// wplace.hip -- does the rate of the trajectory kernels' OUTPUT PATTERN depend on where the tensors were placed, and
// on which workgroup writes which 64-game span? (diagnostic; pure stores, no game logic)
//
//   hipcc -O3 --offload-arch=gfx950 tools/wplace.hip -o tools/bin/wplace && tools/bin/wplace [n] [k] [sets]
//
// The rollout's geometry (tools/wstream.hip `traj`): per frame and wave one contiguous 8 960-byte span into each of two
// [k][n][35] int32 tensors, four dword rows, one byte row.  `sets` separately hipMalloc'ed output sets and one set
// carved out of a single allocation are each written by the same kernel under several workgroup -> span mappings:
//   id      span = blockIdx.x                                  (what the step kernels do: neighbouring spans on neighbouring XCDs)
//   xcd     span = (blockIdx.x % 8) * (waves / 8) + blockIdx.x / 8     (each XCD writes one contiguous eighth of a frame)
//   xcd2    the same in blocks of 2 spans per XCD turn
//   rot     span = (blockIdx.x + frame * 37) % waves            (a wave walks through the frame rows over time: the
//                                                                 spans written at one moment differ from frame to frame)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using Rsrc = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

struct Out {
    char *obs1, *obs2, *rew1, *rew2, *act, *term;
    int64_t n;
    int k;
    uint32_t mask = 7;  // 1: the five small rows, 2: obs1, 4: obs2, 8: both observation tensors striped frame by frame over z[0..2]
    char* z[3] = {nullptr, nullptr, nullptr};
    uint32_t fs;  // frame stride in games (>= n; == n: the contiguous [k][n][...] tensors of the ABI)
};

__device__ __forceinline__ void store_span(char* tensor, uint32_t frame_bytes, uint32_t wave_off, int lane, uint32_t seed)
{
    const Rsrc span = make_rsrc(tensor + wave_off, frame_bytes - wave_off);
#pragma unroll
    for (int pass = 0; pass < 9; ++pass) {
        const int v = pass * 64 + lane;
        const u32x4 w = {seed, seed + pass, seed, seed};
        __builtin_amdgcn_raw_buffer_store_b128(w, span, v < 560 ? (uint32_t)v * 16u : ~0u, 0, 2);
    }
}

template <int MAP>
__global__ __launch_bounds__(64) void traj(Out o)
{
    const int lane = threadIdx.x;
    const uint32_t n32 = o.fs, frame = n32 * 140u, waves = gridDim.x;
    uint32_t span = blockIdx.x;
    if (MAP == 1) span = (blockIdx.x & 7u) * (waves >> 3) + (blockIdx.x >> 3);
    if (MAP == 2) span = ((blockIdx.x >> 1) & 7u) * (waves >> 3) + ((blockIdx.x >> 4) << 1) + (blockIdx.x & 1u);
    for (int s = 0; s < o.k; ++s) {
        uint32_t sp = span;
        if (MAP == 3) sp = (span + (uint32_t)s * 37u) % waves;
        const uint32_t wave_off = sp * 8960u, voff = (sp * 64u + lane) * 4u;
        if (o.mask & 1u) {
            const Rsrc ao = make_rsrc(o.act + (int64_t)s * n32 * 8, n32 * 8u);
            __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, n32 * 4u, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew1 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew2 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)s, make_rsrc(o.term + (int64_t)s * n32, n32), voff >> 2, 0, 0);
        }
        if (o.mask & 8u) {  // frame s of obs1 in zone s % 3, of obs2 in zone (s + 1) % 3 (second half of each zone's block)
            const int64_t slot = s / 3;
            store_span(o.z[s % 3] + slot * frame, frame, wave_off, lane, (uint32_t)s);
            store_span(o.z[(s + 1) % 3] + (slot + (o.k + 2) / 3) * frame, frame, wave_off, lane, (uint32_t)s);
        }
        if (o.mask & 2u) store_span(o.obs1 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
        if (o.mask & 4u) store_span(o.obs2 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
    }
}

__global__ __launch_bounds__(256) void linear(u32x4* dst, size_t vecs, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < vecs; v += stride) {
        const u32x4 w = {seed, seed, seed, seed};
        __builtin_nontemporal_store(w, dst + v);
    }
}

__global__ __launch_bounds__(256) void linear_read(const u32x4* src, size_t vecs, uint32_t* sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < vecs; v += stride) {
        const u32x4 w = __builtin_nontemporal_load(src + v);
        acc += w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) *sink = acc;  // (never: keeps the loads)
}

template <class F>
static double time_us(F&& launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int round = 0; round < 3; ++round) {
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3 / reps);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    const int k = argc > 2 ? atoi(argv[2]) : 32;
    const int sets = argc > 3 ? atoi(argv[3]) : 8;
    const size_t obs_bytes = (size_t)k * n * 140;
    const size_t sizes[6] = {obs_bytes, obs_bytes, (size_t)k * n * 4, (size_t)k * n * 4, (size_t)k * n * 8, (size_t)k * n};
    const unsigned waves = (unsigned)((n + 63) / 64);
    const int reps = getenv("WPLACE_REPS") ? atoi(getenv("WPLACE_REPS")) : 30;
    if (getenv("WPLACE_STRIDES")) {  // frame-stride sweep on `sets` separately allocated output sets
        const uint32_t pads[] = {0, 64, 448, 1984, 4096, 64 * 37, 16320};
        const uint32_t maxpad = 16384, maxfs = (uint32_t)n + maxpad;
        const size_t ob = (size_t)k * maxfs * 140;
        printf("n=%lld k=%d: us per frame by frame stride n + pad (games); one row per separately allocated set\n", (long long)n, k);
        printf("pads:");
        for (uint32_t pad : pads) printf(" %8u", pad);
        printf("\n");
        for (int s = 0; s < sets; ++s) {
            Out o;
            o.n = n, o.k = k;
            CHECK(hipMalloc(&o.obs1, ob));
            CHECK(hipMalloc(&o.obs2, ob));
            CHECK(hipMalloc(&o.rew1, (size_t)k * maxfs * 4));
            CHECK(hipMalloc(&o.rew2, (size_t)k * maxfs * 4));
            CHECK(hipMalloc(&o.act, (size_t)k * maxfs * 8));
            CHECK(hipMalloc(&o.term, (size_t)k * maxfs));
            printf("set %2d", s);
            for (uint32_t pad : pads) {
                if (pad > maxpad) {  // the tensors hold frames of at most n + maxpad games
                    fprintf(stderr, "pad %u beyond the allocation\n", pad);
                    return 1;
                }
                o.fs = (uint32_t)n + pad;
                printf(" %8.3f", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps) / k);
                fflush(stdout);
            }
            printf("\n");
        }
        return 0;
    }
    if (getenv("WPLACE_DELTAS")) {  // one big allocation: obs1 at its (1 GiB aligned) start, obs2 at start + D
        const size_t gib = (size_t)1 << 30, mib = (size_t)1 << 20;
        const size_t arena_bytes = 244 * gib;
        char* arena;
        CHECK(hipMalloc(&arena, arena_bytes));
        char* base = arena + ((gib - ((uintptr_t)arena & (gib - 1))) & (gib - 1));
        std::vector<size_t> deltas = {282 * mib, 32 * gib, 48 * gib, 52 * gib, 56 * gib, 58 * gib, 60 * gib, 61 * gib, 62 * gib, 63 * gib,
                                      63 * gib + 512 * mib, 64 * gib, 65 * gib, 66 * gib, 68 * gib, 72 * gib, 80 * gib, 96 * gib, 112 * gib,
                                      120 * gib, 126 * gib, 128 * gib, 130 * gib, 144 * gib, 160 * gib, 176 * gib, 190 * gib, 192 * gib,
                                      194 * gib, 208 * gib, 224 * gib, 240 * gib};
        printf("n=%lld k=%d: us per launch of %d frames, obs1 + obs2 only, obs2 = obs1 + D inside one allocation of %zu GiB\n", (long long)n, k, k,
               arena_bytes >> 30);
        for (int pass = 0; pass < 1; ++pass)
            for (size_t d : deltas) {
                if ((size_t)(base - arena) + d + obs_bytes > arena_bytes || d < obs_bytes) {
                    fprintf(stderr, "delta %zu does not fit\n", d);
                    return 1;
                }
                Out o;
                o.n = n, o.k = k, o.fs = (uint32_t)n, o.mask = 6u;
                o.obs1 = base, o.obs2 = base + d;
                o.rew1 = o.rew2 = o.act = o.term = nullptr;  // (mask 6: never touched)
                printf("  D = %8.3f GiB (%6zu MiB): %8.2f\n", d / (double)gib, d / mib,
                       time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps));
                fflush(stdout);
            }
        return 0;
    }
    if (getenv("WPLACE_ZONES")) {  // one big allocation: the pair (obs1 at reference R, obs2 at every GiB step) -> which steps pair well with R
        const size_t gib = (size_t)1 << 30;
        const size_t arena_gib = 250;
        char* arena;
        CHECK(hipMalloc(&arena, arena_gib * gib));
        char* base = arena + ((gib - ((uintptr_t)arena & (gib - 1))) & (gib - 1));
        const size_t steps = arena_gib - 2;
        printf("n=%lld k=%d: obs1 + obs2 only, us per launch; obs1 at GiB step R of one %zu GiB allocation (VA %012llx), obs2 at step P\n",
               (long long)n, k, arena_gib, (unsigned long long)(uintptr_t)base);
        auto pair_us = [&](size_t sa, size_t sb) {
            Out o;
            o.n = n, o.k = k, o.fs = (uint32_t)n, o.mask = 6u;
            o.obs1 = base + sa * gib, o.obs2 = base + sb * gib;
            o.rew1 = o.rew2 = o.act = o.term = nullptr;
            return time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, 3);
        };
        // zone of every GiB step relative to step 0 and to the first step that pairs well with it
        std::vector<int> zone(steps, -1);
        const double same = pair_us(0, 1);
        size_t b0 = 0;
        zone[0] = 0;
        for (size_t pstep = 1; pstep < steps; ++pstep)
            if (pair_us(0, pstep) > 0.9 * same) zone[pstep] = 0;
        for (size_t pstep = 2; pstep + 1 < steps && !b0; ++pstep)  // (a step in the middle of a run: a GiB step may straddle two zones)
            if (zone[pstep - 1] && zone[pstep] && zone[pstep + 1]) b0 = pstep;
        for (size_t pstep = 1; pstep < steps && b0; ++pstep)
            if (zone[pstep] < 0) zone[pstep] = (pstep == b0 || pair_us(b0, pstep) > 0.9 * same) ? 1 : 2;
        printf("zones by GiB step (0: that of step 0): ");
        for (size_t pstep = 0; pstep < steps; ++pstep) printf("%d", zone[pstep]);
        printf("\n");
        // a run of >= 3 consecutive GiB steps in each zone (a set needs 0.6 GiB; the striped case 0.4 GiB per zone)
        size_t at[3] = {0, 0, 0};
        bool have[3] = {false, false, false};
        for (size_t pstep = 0; pstep + 2 < steps; ++pstep)
            for (int zz = 0; zz < 3; ++zz)
                if (!have[zz] && zone[pstep] == zz && zone[pstep + 1] == zz && zone[pstep + 2] == zz) at[zz] = pstep, have[zz] = true;
        if (!(have[0] && have[1] && have[2])) {
            printf("fewer than three zones with 3 GiB runs found\n");
            return 0;
        }
        printf("using GiB steps %zu / %zu / %zu for zones 0 / 1 / 2\n", at[0], at[1], at[2]);
        const size_t small_bytes = sizes[2] + sizes[3] + sizes[4] + sizes[5] + 4 * 256;
        auto full = [&](const char* tag, size_t s1, size_t s2, size_t ssmall, size_t off_small, uint32_t mask) {
            Out o;
            o.n = n, o.k = k, o.fs = (uint32_t)n, o.mask = mask;
            o.obs1 = base + s1 * gib, o.obs2 = base + s2 * gib + (s1 == s2 ? obs_bytes : 0);
            char* sm = base + ssmall * gib + off_small;
            o.rew1 = sm, o.rew2 = o.rew1 + sizes[2], o.act = o.rew2 + sizes[3], o.term = o.act + sizes[4];
            for (int zz = 0; zz < 3; ++zz) o.z[zz] = base + at[zz] * gib;
            const double t = time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps);
            printf("  %-64s %8.2f us per launch = %6.3f us per frame\n", tag, t, t / k);
            fflush(stdout);
        };
        if (2 * obs_bytes + small_bytes > 3 * gib) {
            fprintf(stderr, "set larger than the 3 GiB runs\n");
            return 1;
        }
        for (int pass = 0; pass < 2; ++pass) {
            full("all tensors in zone 0", at[0], at[0], at[0], 2 * obs_bytes, 7u);
            full("obs1 zone 0, obs2 zone 1, small rows zone 0", at[0], at[1], at[0], obs_bytes, 7u);
            full("obs1 zone 0, obs2 zone 1, small rows zone 2", at[0], at[1], at[2], 0, 7u);
            full("obs1 + obs2 only: zone 0 / zone 0", at[0], at[0], at[0], 2 * obs_bytes, 6u);
            full("obs1 + obs2 only: zone 0 / zone 1", at[0], at[1], at[0], obs_bytes, 6u);
            full("obs1 + obs2 only: both striped frame by frame over 3 zones", at[0], at[1], at[2], 0, 8u);
            full("the same + small rows in zone 2 (behind the stripes)", at[0], at[1], at[2], 2 * obs_bytes, 9u);
            full("obs1 only, zone 0", at[0], at[0], at[0], 0, 2u);
        }
        return 0;
    }
    if (getenv("WPLACE_SUBSETS")) {  // which of a set's tensors carry the difference between sets?
        printf("n=%lld k=%d: us per launch of %d frames, by the tensors written (one row per separately allocated set)\n", (long long)n, k, k);
        printf("%-6s %9s %9s %9s %9s %9s %9s | obs1 of this set with obs2 of the previous one\n", "set", "all", "obs1+obs2", "obs1", "obs2", "small",
               "small+obs1");
        Out prev{};
        for (int s = 0; s < sets; ++s) {
            Out o;
            o.n = n, o.k = k, o.fs = (uint32_t)n;
            char** p[6] = {&o.obs1, &o.obs2, &o.rew1, &o.rew2, &o.act, &o.term};
            for (int i = 0; i < 6; ++i) CHECK(hipMalloc(p[i], sizes[i]));
            printf("set %2d", s);
            for (uint32_t mask : {7u, 6u, 2u, 4u, 1u, 3u}) {
                o.mask = mask;
                printf(" %9.2f", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            }
            if (s) {
                Out x = o;
                x.mask = 6u, x.obs2 = prev.obs2;
                printf(" | %9.2f", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, x); }, reps));
                x.mask = 7u;
                printf(" (all: %9.2f)", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, x); }, reps));
            }
            printf("\n");
            fflush(stdout);
            prev = o;
        }
        return 0;
    }
    if (argc > 4) {  // map: `chunks` allocations of 1 GiB each, one output set carved into each, all kept: rate by chunk
        const int chunks = atoi(argv[4]);
        std::vector<char*> mem(chunks);
        for (int c = 0; c < chunks; ++c) CHECK(hipMalloc(&mem[c], (size_t)1 << 30));
        printf("n=%lld k=%d: us per frame of the output pattern by 1 GiB allocation (in allocation order)\n", (long long)n, k);
        for (int pass = 0; pass < 2; ++pass)
            for (int c = 0; c < chunks; ++c) {
                Out o;
                o.n = n, o.k = k, o.fs = (uint32_t)n;
                char* at = mem[c];
                char** p[6] = {&o.act, &o.obs1, &o.obs2, &o.rew1, &o.rew2, &o.term};
                const size_t order[6] = {sizes[4], sizes[0], sizes[1], sizes[2], sizes[3], sizes[5]};
                for (int i = 0; i < 6; ++i) {
                    *p[i] = at;
                    at += (order[i] + 255) / 256 * 256;
                }
                if ((size_t)(at - mem[c]) > ((size_t)1 << 30)) {
                    fprintf(stderr, "set does not fit into 1 GiB\n");
                    return 1;
                }
                const double t0 = time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, 8) / k;
                printf("pass %d chunk %3d @%012llx %7.3f\n", pass, c, (unsigned long long)(uintptr_t)mem[c], t0);
            }
        return 0;
    }
    std::vector<Out> outs;
    for (int s = 0; s < sets; ++s) {
        Out o;
        o.n = n, o.k = k, o.fs = (uint32_t)n;
        char** p[6] = {&o.obs1, &o.obs2, &o.rew1, &o.rew2, &o.act, &o.term};
        for (int i = 0; i < 6; ++i) CHECK(hipMalloc(p[i], sizes[i]));
        outs.push_back(o);
    }
    {   // one allocation, the six tensors one behind the other from a 1 GiB boundary
        size_t total = (size_t)1 << 30;
        for (size_t b : sizes) total += (b + 255) / 256 * 256;
        char* arena;
        CHECK(hipMalloc(&arena, total));
        char* at = arena + ((((size_t)1 << 30) - ((uintptr_t)arena & (((size_t)1 << 30) - 1))) & (((size_t)1 << 30) - 1));
        Out o;
        o.n = n, o.k = k, o.fs = (uint32_t)n;
        char** p[6] = {&o.act, &o.obs1, &o.obs2, &o.rew1, &o.rew2, &o.term};
        const size_t order[6] = {sizes[4], sizes[0], sizes[1], sizes[2], sizes[3], sizes[5]};
        for (int i = 0; i < 6; ++i) {
            *p[i] = at;
            at += (order[i] + 255) / 256 * 256;
        }
        outs.push_back(o);
    }
    printf("n=%lld k=%d: us per frame, pure stores of the rollout's output pattern (nt); fill of each observation tensor alone in TB/s\n", (long long)n, k);
    printf("%-10s %9s %9s %9s %9s  %9s %9s\n", "set", "id", "xcd", "id again", "rot", "fill obs1", "fill obs2");
    uint32_t* sink;
    CHECK(hipMalloc(&sink, 4));
    std::vector<size_t> order;
    for (size_t s = 0; s < outs.size(); ++s) order.push_back(s);
    for (size_t s = outs.size(); s-- > 0;) order.push_back(s);  // and back again: is a set's rate a property of its memory?
    for (size_t s : order) {
        const Out o = outs[s];
        const double t0 = time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps) / k;
        const double t1 = time_us([&] { hipLaunchKernelGGL(traj<1>, dim3(waves), dim3(64), 0, 0, o); }, reps) / k;
        const double t2 = time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps) / k;
        const double t3 = time_us([&] { hipLaunchKernelGGL(traj<3>, dim3(waves), dim3(64), 0, 0, o); }, reps) / k;
        const double f1 = obs_bytes / time_us([&] { hipLaunchKernelGGL(linear, dim3(4096), dim3(256), 0, 0, (u32x4*)o.obs1, obs_bytes / 16, 1u); }, reps) / 1e6;
        const double f2 = obs_bytes / time_us([&] { hipLaunchKernelGGL(linear, dim3(4096), dim3(256), 0, 0, (u32x4*)o.obs2, obs_bytes / 16, 1u); }, reps) / 1e6;
        const double r1 = obs_bytes / time_us([&] { hipLaunchKernelGGL(linear_read, dim3(4096), dim3(256), 0, 0, (const u32x4*)o.obs1, obs_bytes / 16, sink); }, reps) / 1e6;
        const double r2 = obs_bytes / time_us([&] { hipLaunchKernelGGL(linear_read, dim3(4096), dim3(256), 0, 0, (const u32x4*)o.obs2, obs_bytes / 16, sink); }, reps) / 1e6;
        printf("%-7s %2zu %9.3f %9.3f %9.3f %9.3f  fill %6.2f %6.2f  read %6.2f %6.2f TB/s  obs1@%012llx obs2@%012llx\n", s + 1 == outs.size() ? "arena" : "separate", s, t0, t1,
               t2, t3, f1, f2, r1, r2, (unsigned long long)(uintptr_t)o.obs1, (unsigned long long)(uintptr_t)o.obs2);
        fflush(stdout);
    }
    return 0;
}
