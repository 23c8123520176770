// geometry_floor.hip -- does the period of a chain of dependent launches depend on how its 2 048 waves are cut into
// workgroups? (diagnostic; round 6)
//
//   hipcc -O3 --offload-arch=gfx950 tools/geometry_floor.hip -o tools/bin/geometry_floor && tools/bin/geometry_floor
//
// The headline launch is 1 024 workgroups of two waves (64 games, 17 920 bytes of LDS); an EMPTY launch of that geometry
// costs 1.55 us of the chain's 6.97 (tools/launch_floor.hip).  Here the same 131 072 threads as 2 048 x 64, 1 024 x 128,
// 512 x 256, 256 x 512 and 128 x 1 024, LDS in proportion, each empty and with one load + one store per lane, K launches
// captured into a hipGraph and replayed.
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

template <int THREADS, int TOUCH>
__global__ __launch_bounds__(THREADS) void probe(int32_t* buf, int64_t n)
{
    __shared__ int32_t lds[THREADS * 35];  // 17 920 bytes per 128 threads, like the pair kernel's staging rows
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (TOUCH) {
        if (i < n) {
            const int32_t v = buf[i];
            lds[threadIdx.x * 35] = v;
            __syncthreads();
            buf[n + i] = lds[(threadIdx.x ^ 64) * 35] + 1;
        }
    } else if (n < 0) {
        lds[threadIdx.x] = 0;  // (keeps the allocation)
        buf[0] = lds[threadIdx.x ^ 1];
    }
}

template <int THREADS, int TOUCH>
static double time_chain(int32_t* buf, int64_t n, hipStream_t stream, int launches)
{
    const dim3 grid((unsigned)(n / THREADS)), block(THREADS);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < launches; ++k) hipLaunchKernelGGL((probe<THREADS, TOUCH>), grid, block, 0, stream, buf, n);
    CHECK(hipStreamEndCapture(stream, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, stream));
    CHECK(hipStreamSynchronize(stream));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int rep = 0; rep < 9; ++rep) {
        CHECK(hipEventRecord(e0, stream));
        for (int r = 0; r < 8; ++r) CHECK(hipGraphLaunch(exec, stream));
        CHECK(hipEventRecord(e1, stream));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3 / (8.0 * launches));
    }
    std::sort(us.begin(), us.end());
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    return us[us.size() / 2];
}

int main()
{
    const int64_t n = 131072;  // threads = two waves per 64 games of a 65 536-game launch
    int32_t* buf;
    CHECK(hipMalloc(&buf, 2 * n * sizeof(int32_t)));
    CHECK(hipMemset(buf, 0, 2 * n * sizeof(int32_t)));
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    const int K = 2048;
    printf("us per dependent launch, %lld threads, median of 9 x 8 graph replays of %d launches\n", (long long)n, K);
    printf("  workgroup   empty   load+barrier+store per lane\n");
    printf("  %4d x %4d  %6.3f  %6.3f\n", (int)(n / 64), 64, time_chain<64, 0>(buf, n, stream, K), time_chain<64, 1>(buf, n, stream, K));
    printf("  %4d x %4d  %6.3f  %6.3f\n", (int)(n / 128), 128, time_chain<128, 0>(buf, n, stream, K), time_chain<128, 1>(buf, n, stream, K));
    printf("  %4d x %4d  %6.3f  %6.3f\n", (int)(n / 256), 256, time_chain<256, 0>(buf, n, stream, K), time_chain<256, 1>(buf, n, stream, K));
    printf("  %4d x %4d  %6.3f  %6.3f\n", (int)(n / 512), 512, time_chain<512, 0>(buf, n, stream, K), time_chain<512, 1>(buf, n, stream, K));
    printf("  %4d x %4d  %6.3f  %6.3f\n", (int)(n / 1024), 1024, time_chain<1024, 0>(buf, n, stream, K), time_chain<1024, 1>(buf, n, stream, K));
    return 0;
}
