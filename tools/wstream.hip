// wstream.hip -- what does this box's memory system take for the trajectory kernels' output pattern? (diagnostic)
//
//   hipcc -O3 --offload-arch=gfx950 tools/wstream.hip -o tools/bin/wstream && tools/bin/wstream [n] [k]
//
// pz_rollout_random writes, per frame and wave (64 games), one contiguous 8 960-byte span into each of two
// [k][n][35] int32 tensors (16 B per lane, nine passes) plus four dword rows and one byte row; consecutive frames of a
// wave are n * 140 bytes apart.  The kernels below issue exactly those stores with no game logic in front of them:
//   traj      -- the rollout's geometry: one wave per 64 games, k frames in a loop            (1 wave / SIMD at n = 65 536)
//   traj2     -- the same bytes from two waves per 64 games (one tensor each)                (2 waves / SIMD)
//   traj_wide -- one wave per 64 games, but all k frames' stores of a tensor issued back to back per pass (no frame loop
//                dependency; an upper bound for the geometry)
//   linear    -- the same number of bytes written as one linear 16-B-per-lane fill, 8 waves per CU (the box's ceiling)
// each with and without the `nt` cache policy.  Prints us per frame and TB/s.
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
};

template <int AUX>
__device__ __forceinline__ void store_span(char* tensor, uint32_t frame_bytes, uint32_t wave_off, int lane, uint32_t seed)
{
    const Rsrc span = make_rsrc(tensor + wave_off, frame_bytes - wave_off);
#pragma unroll
    for (int pass = 0; pass < 9; ++pass) {
        const int v = pass * 64 + lane;
        const u32x4 w = {seed, seed + pass, seed, seed};
        __builtin_amdgcn_raw_buffer_store_b128(w, span, v < 560 ? (uint32_t)v * 16u : ~0u, 0, AUX);
    }
}

template <int AUX>
__global__ __launch_bounds__(64) void traj(Out o)
{
    const int lane = threadIdx.x;
    const uint32_t n32 = (uint32_t)o.n, frame = n32 * 140u, wave_off = blockIdx.x * 8960u;
    const uint32_t voff = (blockIdx.x * 64u + lane) * 4u;
    for (int s = 0; s < o.k; ++s) {
        const Rsrc ao = make_rsrc(o.act + (int64_t)s * n32 * 8, n32 * 8u);
        __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, n32 * 4u, 0);
        __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew1 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew2 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)s, make_rsrc(o.term + (int64_t)s * n32, n32), voff >> 2, 0, 0);
        store_span<AUX>(o.obs1 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
        store_span<AUX>(o.obs2 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
    }
}

template <int AUX>
__global__ __launch_bounds__(128) void traj2(Out o)
{
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const uint32_t n32 = (uint32_t)o.n, frame = n32 * 140u, wave_off = blockIdx.x * 8960u;
    const uint32_t voff = (blockIdx.x * 64u + lane) * 4u;
    for (int s = 0; s < o.k; ++s) {
        if (role == 0) {
            const Rsrc ao = make_rsrc(o.act + (int64_t)s * n32 * 8, n32 * 8u);
            __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, n32 * 4u, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew1 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
            store_span<AUX>(o.obs1 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew2 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)s, make_rsrc(o.term + (int64_t)s * n32, n32), voff >> 2, 0, 0);
            store_span<AUX>(o.obs2 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
        }
    }
}

// the frames of a wave are independent streams here: frame s handled by workgroup (blockIdx.y = s)
template <int AUX>
__global__ __launch_bounds__(64) void traj_wide(Out o)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    const uint32_t n32 = (uint32_t)o.n, frame = n32 * 140u, wave_off = blockIdx.x * 8960u;
    const uint32_t voff = (blockIdx.x * 64u + lane) * 4u;
    const Rsrc ao = make_rsrc(o.act + (int64_t)s * n32 * 8, n32 * 8u);
    __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(s, ao, voff, n32 * 4u, 0);
    __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew1 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(s, make_rsrc(o.rew2 + (int64_t)s * n32 * 4, n32 * 4u), voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)s, make_rsrc(o.term + (int64_t)s * n32, n32), voff >> 2, 0, 0);
    store_span<AUX>(o.obs1 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
    store_span<AUX>(o.obs2 + (int64_t)s * frame, frame, wave_off, lane, (uint32_t)s);
}

template <int AUX>
__global__ __launch_bounds__(256) void linear(u32x4* dst, size_t vecs, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < vecs; v += stride) {
        const u32x4 w = {seed, seed, seed, seed};
        if (AUX == 2)
            __builtin_nontemporal_store(w, dst + v);
        else
            dst[v] = w;
    }
}

template <class F>
static double time_us(F&& launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch();  // warm up
    CHECK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int round = 0; round < 5; ++round) {
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
    Out o;
    o.n = n;
    o.k = k;
    const size_t obs_bytes = (size_t)k * n * 140;
    CHECK(hipMalloc(&o.obs1, obs_bytes));
    CHECK(hipMalloc(&o.obs2, obs_bytes));
    CHECK(hipMalloc(&o.rew1, (size_t)k * n * 4));
    CHECK(hipMalloc(&o.rew2, (size_t)k * n * 4));
    CHECK(hipMalloc(&o.act, (size_t)k * n * 8));
    CHECK(hipMalloc(&o.term, (size_t)k * n));
    const double frame_bytes = (double)n * 297.0;
    const unsigned waves = (unsigned)((n + 63) / 64);
    const int reps = 40;
    auto report = [&](const char* name, double us_launch) {
        printf("%-22s %8.3f us/frame  %6.2f TB/s  (n=%lld k=%d)\n", name, us_launch / k, frame_bytes * k / us_launch / 1e6,
               (long long)n, k);
        fflush(stdout);
    };
    if (argc > 3) {  // cache-policy sweep of the trajectory pattern (aux bits: 1 = sc0, 2 = nt, 16 = sc1)
        for (int round = 0; round < 2; ++round) {
            report("traj aux=0 plain", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=1 sc0", time_us([&] { hipLaunchKernelGGL(traj<1>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=2 nt", time_us([&] { hipLaunchKernelGGL(traj<2>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=3 nt sc0", time_us([&] { hipLaunchKernelGGL(traj<3>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=16 sc1", time_us([&] { hipLaunchKernelGGL(traj<16>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=17 sc0 sc1", time_us([&] { hipLaunchKernelGGL(traj<17>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=18 nt sc1", time_us([&] { hipLaunchKernelGGL(traj<18>, dim3(waves), dim3(64), 0, 0, o); }, reps));
            report("traj aux=19 nt sc0 sc1", time_us([&] { hipLaunchKernelGGL(traj<19>, dim3(waves), dim3(64), 0, 0, o); }, reps));
        }
        return 0;
    }
    for (int round = 0; round < 2; ++round) {
        report("traj nt", time_us([&] { hipLaunchKernelGGL(traj<2>, dim3(waves), dim3(64), 0, 0, o); }, reps));
        report("traj plain", time_us([&] { hipLaunchKernelGGL(traj<0>, dim3(waves), dim3(64), 0, 0, o); }, reps));
        report("traj2 nt", time_us([&] { hipLaunchKernelGGL(traj2<2>, dim3(waves), dim3(128), 0, 0, o); }, reps));
        report("traj2 plain", time_us([&] { hipLaunchKernelGGL(traj2<0>, dim3(waves), dim3(128), 0, 0, o); }, reps));
        report("traj_wide nt", time_us([&] { hipLaunchKernelGGL(traj_wide<2>, dim3(waves, k), dim3(64), 0, 0, o); }, reps));
        report("traj_wide plain", time_us([&] { hipLaunchKernelGGL(traj_wide<0>, dim3(waves, k), dim3(64), 0, 0, o); }, reps));
        // linear fill of as many bytes as one launch of the above writes into the two observation tensors
        const size_t vecs = obs_bytes / 16;
        const double scale = frame_bytes * k / (2.0 * obs_bytes);  // report on the same "bytes per frame" basis
        report("linear nt (2 tensors)", scale * (time_us([&] { hipLaunchKernelGGL(linear<2>, dim3(2048), dim3(256), 0, 0, (u32x4*)o.obs1, vecs, 1u); }, reps) +
                                        time_us([&] { hipLaunchKernelGGL(linear<2>, dim3(2048), dim3(256), 0, 0, (u32x4*)o.obs2, vecs, 1u); }, reps)));
        report("linear plain", scale * (time_us([&] { hipLaunchKernelGGL(linear<0>, dim3(2048), dim3(256), 0, 0, (u32x4*)o.obs1, vecs, 1u); }, reps) +
                                time_us([&] { hipLaunchKernelGGL(linear<0>, dim3(2048), dim3(256), 0, 0, (u32x4*)o.obs2, vecs, 1u); }, reps)));
    }
    return 0;
}
