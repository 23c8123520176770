// calib_traffic.hip -- known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// for the access widths the step kernels use (diagnostic; tools/profile.sh section `calib`).
//
//   hipcc -O3 --offload-arch=gfx950 tools/calib_traffic.hip -o tools/bin/calib_traffic
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o run -f csv -- tools/bin/calib_traffic
//
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reads half of the bytes for 16-B-per-lane streaming loads and "other
// access widths are uncalibrated: calibrate on a known byte count".  The step kernels read the int32 state with
// one dword per lane (a wave instruction = one 256-byte segment of a column) and write observations with 16 B per
// lane, state columns with 4 B per lane and flags with 1 B per lane.  Every kernel below moves exactly
// `bytes` (printed) per launch in one of those shapes; tools/pmc_summary.py divides the counters by them.
//
// Each shape runs on a 64 MiB buffer (stays in the 256 MiB Infinity Cache between launches: the 65 536-game
// regime) and on a 1 GiB buffer (streams from / to HBM: the 524 288-game regime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// reads: every lane keeps a running xor and one lane per wave writes it (4 B per 64 lanes x ITER loads: noise)
template <int ITER>
__global__ __launch_bounds__(64) void read_b32(const uint32_t* __restrict__ src, uint32_t* __restrict__ sink)
{
    // like the state columns: a wave reads ITER "columns", each a contiguous 256-byte segment, pitch = grid * 256 B
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) acc ^= src[(size_t)c * pitch + i];
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;  // practically never: the loads cannot be dropped
}

template <int ITER>
__global__ __launch_bounds__(64) void read_b128(const u32x4* __restrict__ src, uint32_t* __restrict__ sink)
{
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) acc ^= src[(size_t)c * pitch + i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

template <int ITER>
__global__ __launch_bounds__(64) void write_b32(uint32_t* __restrict__ dst, uint32_t v)
{
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) dst[(size_t)c * pitch + i] = v + c;
}

template <int ITER>
__global__ __launch_bounds__(64) void write_b128(u32x4* __restrict__ dst, uint32_t v)
{
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) dst[(size_t)c * pitch + i] = u32x4{v, v + c, v, v};
}

// one byte per lane (the `terminated` flags): 64 contiguous bytes per wave instruction
template <int ITER>
__global__ __launch_bounds__(64) void write_b8(uint8_t* __restrict__ dst, uint32_t v)
{
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) dst[(size_t)c * pitch + i] = (uint8_t)(v + c);
}

// the changed-only write-back's shape: a dword store executed by every 5th lane only (sparse sectors of a column)
template <int ITER>
__global__ __launch_bounds__(64) void write_b32_sparse(uint32_t* __restrict__ dst, uint32_t v)
{
    const size_t pitch = (size_t)gridDim.x * 64;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x % 5 != 0) return;
#pragma unroll 8
    for (int c = 0; c < ITER; ++c) dst[(size_t)c * pitch + i] = v + c;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const size_t small = (size_t)64 << 20, big = (size_t)1 << 30;
    void* buf;
    uint32_t* sink;
    CHECK(hipMalloc(&buf, big));
    CHECK(hipMalloc(&sink, 1 << 22));
    CHECK(hipMemset(buf, 1, big));
    CHECK(hipDeviceSynchronize());
    constexpr int ITER = 32;
    for (size_t bytes : {small, big}) {
        const char* where = bytes == small ? "64MiB" : "1GiB";
        for (int r = 0; r < reps; ++r) {
            // grid chosen so that grid * 64 lanes * ITER accesses * lane_bytes == bytes
            const unsigned g4 = (unsigned)(bytes / (64 * ITER * 4)), g16 = (unsigned)(bytes / (64 * ITER * 16));
            const unsigned g1 = (unsigned)(bytes / 16 / (64 * ITER * 1));  // the byte shape moves bytes / 16
            hipLaunchKernelGGL(read_b32<ITER>, dim3(g4), dim3(64), 0, 0, (const uint32_t*)buf, sink);
            hipLaunchKernelGGL(read_b128<ITER>, dim3(g16), dim3(64), 0, 0, (const u32x4*)buf, sink);
            hipLaunchKernelGGL(write_b32<ITER>, dim3(g4), dim3(64), 0, 0, (uint32_t*)buf, (uint32_t)r);
            hipLaunchKernelGGL(write_b128<ITER>, dim3(g16), dim3(64), 0, 0, (u32x4*)buf, (uint32_t)r);
            hipLaunchKernelGGL(write_b8<ITER>, dim3(g1), dim3(64), 0, 0, (uint8_t*)buf, (uint32_t)r);
            hipLaunchKernelGGL(write_b32_sparse<ITER>, dim3(g4), dim3(64), 0, 0, (uint32_t*)buf, (uint32_t)r);
            CHECK(hipDeviceSynchronize());
            if (r == 0) {
                printf("calib %s read_b32 grid=%u bytes=%zu\n", where, g4, (size_t)g4 * 64 * ITER * 4);
                printf("calib %s read_b128 grid=%u bytes=%zu\n", where, g16, (size_t)g16 * 64 * ITER * 16);
                printf("calib %s write_b32 grid=%u bytes=%zu\n", where, g4, (size_t)g4 * 64 * ITER * 4);
                printf("calib %s write_b128 grid=%u bytes=%zu\n", where, g16, (size_t)g16 * 64 * ITER * 16);
                printf("calib %s write_b8 grid=%u bytes=%zu\n", where, g1, (size_t)g1 * 64 * ITER);
                printf("calib %s write_b32_sparse grid=%u bytes=%zu (13 of 64 lanes: 13 dwords spread over the "
                       "eight 32-byte sectors of a 256-byte segment)\n", where, g4, (size_t)g4 * 13 * ITER * 4);
            }
        }
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
