// launch_floor.hip -- what does ONE dependent launch cost on this box before it has done anything? (diagnostic)
//
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/bin/launch_floor && tools/bin/launch_floor [games]
//
// The bench headline is a chain of dependent launches replayed from a hipGraph: 6.96 us per launch at 65 536 games,
// of which the per-wave stamps (tools/stamps.py) see 5.4 us between a launch's first wave and its last acknowledged
// store.  This program prices the rest and the parts of the chain with kernels that have the headline's GEOMETRY
// (games / 64 workgroups of 128 threads, 17 920 bytes of LDS, same-stream dependency) but none of its game logic,
// each captured K times into a hipGraph and replayed, HIP events around the replays:
//   empty        nothing: the period of a dependent launch chain on this runtime (packet, dispatch, fences)
//   load44       the 44 state columns + 2 action words loaded (one dword per lane and column), one word kept alive
//   load_store   ... and the launch's 24.8 MB written: 20 always-written dword columns + a reward per wave pair + the two
//                observation tensors as 9 x 16 B per lane and wave (`nt`), nothing computed in between
//   store_only   the same stores with no load in front of them
// So: empty = the runtime's floor; load44 - empty = the load latency of a wave; load_store - load44 = what the bytes
// cost once nothing is in front of them; headline - load_store = what the frame's logic adds to the chain.
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

struct Args {
    int32_t* state;        // [44][n]
    const int32_t* act1;   // [n]
    const int32_t* act2;
    int32_t* obs1;         // [n][35]
    int32_t* obs2;
    int32_t* rew1;
    int32_t* rew2;
    int64_t n;
};

// MODE bit 0: loads, bit 1: stores, bit 2: a stand-in for the frame between them (VALU_N dependent full-rate VALU
// instructions on two chains per wave, an LDS exchange and one workgroup barrier in the middle), bit 3: every other
// workgroup issues at s_setprio 3, bit 4: TWO tiles of 64 games per workgroup, one after the other (half as many
// workgroups: one wave per SIMD at 65 536 games; tile 1's loads are issued before tile 0 is computed, tile 0's stores
// drain while tile 1 is computed)
#ifndef VALU_N
#define VALU_N 408
#endif
#ifndef VALU_ILP2
#define VALU_ILP2 0  // 1: the wave's two chains are independent of each other (a lone wave can then fill its own issue slots)
#endif

#ifndef VALU_KIND
#define VALU_KIND 0  // what a "dependent instruction" of the stand-in is: 0 one v_mad_u32_u24; 1 the frame's own idiom -- two
                     // compares into SGPR masks, an s_and_b64, a v_cndmask_b32 on it (3 VALU + 1 SALU per step); 2 the same
                     // select with the predicate kept in a VGPR (sub, ashr, sub, ashr, and, bfi: 6 VALU, no SALU)
#endif
__device__ __forceinline__ void stand_in_step(uint32_t& a, uint32_t b)
{
#if VALU_KIND == 0
    asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#elif VALU_KIND == 1
    unsigned long long m0, m1;  // (SGPR pairs the compiler allocates: no fixed registers in the asm)
    asm volatile(
        "v_cmp_lt_u32_e64 %1, %0, %3\n\t"
        "v_cmp_gt_i32_e64 %2, %0, 17\n\t"
        "s_and_b64 %1, %1, %2\n\t"
        "v_cndmask_b32_e64 %0, %3, %0, %1\n\t"
        "v_add_u32_e32 %0, 3, %0"
        : "+v"(a), "=&s"(m0), "=&s"(m1) : "v"(b) : "scc");
#else
    uint32_t t0, t1;
    asm volatile(
        "v_sub_u32_e32 %1, %0, %3\n\t"
        "v_ashrrev_i32_e32 %1, 31, %1\n\t"
        "v_sub_u32_e32 %2, 17, %0\n\t"
        "v_ashrrev_i32_e32 %2, 31, %2\n\t"
        "v_and_b32_e32 %1, %1, %2\n\t"
        "v_bfi_b32 %0, %1, %0, %3\n\t"
        "v_add_u32_e32 %0, 3, %0"
        : "+v"(a), "=&v"(t0), "=&v"(t1) : "v"(b));
#endif
}
constexpr int kValuPerStep = VALU_KIND == 0 ? 1 : (VALU_KIND == 1 ? 4 : 7);

__device__ __forceinline__ uint32_t frame_stand_in(uint32_t acc, uint32_t other, int32_t* mine, const int32_t* theirs, int lane)
{
    uint32_t a0 = acc, a1 = other;
#if VALU_KIND != 0
#pragma unroll
    for (int k = 0; k < VALU_N / 2 / kValuPerStep / 2; ++k) {
        stand_in_step(a0, a1);
        stand_in_step(a1, a0);
    }
    mine[lane] = (int32_t)a0;
    __syncthreads();
    a1 ^= (uint32_t)theirs[lane];
#pragma unroll
    for (int k = 0; k < VALU_N / 2 / kValuPerStep / 2; ++k) {
        stand_in_step(a0, a1);
        stand_in_step(a1, a0);
    }
    return a0 ^ a1;
#endif
#pragma unroll
    for (int k = 0; k < VALU_N / 4; ++k) {
#if VALU_ILP2
        asm volatile("v_mad_u32_u24 %0, %0, %0, %1" : "+v"(a0) : "v"(other));
        asm volatile("v_mad_u32_u24 %0, %0, %0, %1" : "+v"(a1) : "v"(other));
#else
        asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a0) : "v"(a1));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a1) : "v"(a0));
#endif
    }
    mine[lane] = (int32_t)a0;
    __syncthreads();
    a1 ^= (uint32_t)theirs[lane];
#pragma unroll
    for (int k = 0; k < VALU_N / 4; ++k) {
#if VALU_ILP2
        asm volatile("v_mad_u32_u24 %0, %0, %0, %1" : "+v"(a0) : "v"(other));
        asm volatile("v_mad_u32_u24 %0, %0, %0, %1" : "+v"(a1) : "v"(other));
#else
        asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a0) : "v"(a1));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a1) : "v"(a0));
#endif
    }
    return a0 ^ a1;
}

template <int MODE>
__global__ __launch_bounds__(128) void chain_kernel(Args a)
{
    __shared__ __attribute__((aligned(16))) int32_t lds[2][64 * 35];
    __shared__ int32_t xchg_mem[2][128];
    constexpr int TILES = (MODE & 16) ? 2 : 1;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t n32 = (uint32_t)a.n, pitch = n32 * 4u;
    const Rsrc st = make_rsrc(a.state, n32 * 176u);
    if ((MODE & 8) && (blockIdx.x & 1u)) __builtin_amdgcn_s_setprio(3);
    const int first = role == 0 ? 0 : 13;
    uint32_t w[TILES][33];
    if (MODE & 1) {
        // each wave loads its player's 13 columns, the ball's 12 and the env's 6 like the pair kernel, plus both action words
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint32_t voff = (uint32_t)((blockIdx.x * TILES + t) * 64 + lane) * 4u;
#pragma unroll
            for (int c = 0; c < 13; ++c) w[t][c] = __builtin_amdgcn_raw_buffer_load_b32(st, voff, (uint32_t)(first + c) * pitch, 0);
#pragma unroll
            for (int c = 0; c < 18; ++c) w[t][13 + c] = __builtin_amdgcn_raw_buffer_load_b32(st, voff, (uint32_t)(26 + c) * pitch, 0);
            w[t][31] = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(a.act1, pitch), voff, 0, 0);
            w[t][32] = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(a.act2, pitch), voff, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t tile = blockIdx.x * TILES + t;
        const int64_t i = (int64_t)tile * 64 + lane;
        const uint32_t voff = (uint32_t)i * 4u;
        uint32_t acc = (uint32_t)lane, other = (uint32_t)lane * 3u;
        if (MODE & 1) {
#pragma unroll
            for (int c = 0; c < 33; ++c) {
                if (c & 1) acc ^= w[t][c] + (uint32_t)c; else other += w[t][c];
            }
        }
        if (!(MODE & 4)) acc ^= other;  // (every load stays alive)
        if (MODE & 4) acc = frame_stand_in(acc, other, &xchg_mem[t & 1][role * 64], &xchg_mem[t & 1][(1 - role) * 64], lane);
        if (MODE & 2) {
            // ten always-written dword columns per wave + its reward, then its observation tensor's span (9 x 16 B per lane)
#pragma unroll
            for (int c = 0; c < 6; ++c) __builtin_amdgcn_raw_buffer_store_b32(acc + c, st, voff, (uint32_t)(first + c) * pitch, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_amdgcn_raw_buffer_store_b32(acc + 7 + c, st, voff, (uint32_t)(26 + role * 4 + c) * pitch, 0);
            __builtin_amdgcn_raw_buffer_store_b32(acc, make_rsrc(role == 0 ? a.rew1 : a.rew2, pitch), voff, 0, 0);
            int32_t* rows = lds[role];
#pragma unroll
            for (int k = 0; k < 35; ++k) rows[lane * 35 + k] = (int32_t)(acc + k);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t wave_off = tile * 8960u;
            const Rsrc span = make_rsrc(reinterpret_cast<char*>(role == 0 ? a.obs1 : a.obs2) + wave_off, n32 * 140u - wave_off);
            const u32x4* src4 = reinterpret_cast<const u32x4*>(rows);
#pragma unroll
            for (int pass = 0; pass < 9; ++pass) {
                const int v = pass * 64 + lane;
                if (v < 560) __builtin_amdgcn_raw_buffer_store_b128(src4[v], span, (uint32_t)v * 16u, 0, 2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            // keep the work alive: one lane of the launch may store (never true for the data this program writes)
            if (acc == 0xFFFFFFFFu && i == 0) a.rew1[0] = (int32_t)acc;
        }
    }
}

template <int MODE>
static double time_chain(const Args& a, int launches, int replays, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.n + 63) / 64 / ((MODE & 16) ? 2 : 1))), block(128);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int j = 0; j < launches; ++j) hipLaunchKernelGGL(chain_kernel<MODE>, grid, block, 0, stream, a);
    CHECK(hipStreamEndCapture(stream, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, stream));  // untimed
    CHECK(hipStreamSynchronize(stream));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int round = 0; round < 5; ++round) {
        CHECK(hipEventRecord(e0, stream));
        for (int r = 0; r < replays; ++r) CHECK(hipGraphLaunch(exec, stream));
        CHECK(hipEventRecord(e1, stream));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3 / ((double)launches * replays));
    }
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    Args a{};
    a.n = n;
    CHECK(hipMalloc(&a.state, n * 176));
    CHECK(hipMalloc((void**)&a.act1, n * 4));
    CHECK(hipMalloc((void**)&a.act2, n * 4));
    CHECK(hipMalloc(&a.obs1, n * 140));
    CHECK(hipMalloc(&a.obs2, n * 140));
    CHECK(hipMalloc(&a.rew1, n * 4));
    CHECK(hipMalloc(&a.rew2, n * 4));
    CHECK(hipMemset(a.state, 0, n * 176));
    CHECK(hipMemset((void*)a.act1, 0, n * 4));
    CHECK(hipMemset((void*)a.act2, 0, n * 4));
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    const int launches = 2048, replays = 8;
    const double written_mb = n * (20 * 4 + 2 * 4 + 2 * 140) / 1e6;
    printf("games %lld, %d dependent launches per graph, %d replays per round, median of 5 rounds; us per launch\n", (long long)n,
           launches, replays);
    const double t_empty = time_chain<0>(a, launches, replays, stream);
    printf("  empty        %7.3f\n", t_empty);
    const double t_load = time_chain<1>(a, launches, replays, stream);
    printf("  load44       %7.3f   (+%.3f over empty)\n", t_load, t_load - t_empty);
    const double t_store = time_chain<2>(a, launches, replays, stream);
    printf("  store_only   %7.3f   (+%.3f over empty; %.1f MB written -> %.2f TB/s over that difference)\n", t_store,
           t_store - t_empty, written_mb, written_mb / (t_store - t_empty));
    const double t_both = time_chain<3>(a, launches, replays, stream);
    printf("  load_store   %7.3f   (+%.3f over load44)\n", t_both, t_both - t_load);
    const double t_frame = time_chain<7>(a, launches, replays, stream);
    printf("  + a %d-instruction VALU stand-in for the frame per wave (two chains, one exchange + barrier)\n", VALU_N);
    printf("  frame        %7.3f   (+%.3f over load_store)\n", t_frame, t_frame - t_both);
    const double t_prio = time_chain<15>(a, launches, replays, stream);
    printf("  frame, every other workgroup at s_setprio 3                       %7.3f\n", t_prio);
    const double t_tiles = time_chain<23>(a, launches, replays, stream);
    printf("  frame, TWO tiles per workgroup one after the other (half the grid) %7.3f\n", t_tiles);
    const double t_tiles0 = time_chain<19>(a, launches, replays, stream);
    printf("  load_store, two tiles per workgroup                                %7.3f\n", t_tiles0);
    return 0;
}
