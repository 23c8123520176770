// pz_diagnostic.hpp -- the ONE compile-time switch of the kernel sources.
//
// The product (pika-zoo_amd/build.py) never defines PZ_DIAGNOSTIC_BUILD: build.py refuses the flag, and a library
// compiled with it reports the build id "diagnostic" (pz_build_id), which pikazoo_amd/_native.py refuses to load.
// Everything a tuning session wants to switch at compile time is a bit of that one value, set by tools/ab.py /
// tools/stamps.py when they build their variants into tools/bin/ (`-DPZ_DIAGNOSTIC_BUILD=0x...`,
// tools/ab.py: diag_bits()):
//   bit 0        stamps: per-wave s_memrealtime / s_memtime stamps of one launch (tools/stamps.py)
//   bit 1        no step_pair_kernel: single-frame launches on one wave per 64 games
//   bit 2        no rollout_pair_kernel
//   bit 3        no scout wave
//   bit 4        config 3's early stores WITHOUT the LDS hand-shake (round 4's form: breaks under a held-back partner,
//                profiles/r05_experiments/early_store_edge_partner_held_back.log)
//   bit 5        no early stores at all
//   bits 8-15    hold the computer's wave back N x 127 sleep cycles in front of its first load
//   bits 16-29   keep only a subset of the step kernels' instantiations (dev_keep in pz_kernels.hip): a variant that is
//                timed on one configuration builds in seconds instead of the 100 s of the full library; 0 = all
// Closed experiments are not switches any more: their measured value is a constexpr beside the one-line result
// (pz_kernels.hip, pz_memory.hpp), the logs are under profiles/.
#pragma once
#include <hip/hip_runtime.h>

#ifdef PZ_DIAGNOSTIC_BUILD
namespace pz {
namespace diag {
constexpr unsigned kBits = (PZ_DIAGNOSTIC_BUILD);
constexpr bool kDiagnosticBuild = true;
// per-wave timelines of one launch (100 MHz ticks); read back through pz_debug_read_stamps / _frame_stamps
__device__ unsigned long long g_pz_stamps[8192 * 8];
__device__ unsigned long long g_pz_frame_stamps[8192 * 8];
}  // namespace diag
}  // namespace pz
#define PZ_STAMP_AT(table, slot, k, timer)                                                  \
    do {                                                                                    \
        if (pz::diag::kBits & 1u) {                                                         \
            unsigned long long t_;                                                          \
            asm volatile(timer " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            pz::diag::table[(slot) * 8 + (k)] = t_;                                         \
        }                                                                                   \
    } while (0)
// lane 0 of a single-wave workgroup, slot = workgroup; the frame's sub-phases in a table of their own
#define PZ_STAMP(k)                                                                                          \
    do {                                                                                                     \
        if (blockIdx.x < 8192 && threadIdx.x == 0) PZ_STAMP_AT(g_pz_stamps, blockIdx.x, k, "s_memrealtime"); \
    } while (0)
#define PZ_FRAME_STAMP(k)                                                                                      \
    do {                                                                                                       \
        if (blockIdx.x < 8192 && threadIdx.x == 0) PZ_STAMP_AT(g_pz_frame_stamps, blockIdx.x, k, "s_memtime"); \
    } while (0)
// the pair kernel: lane 0 of BOTH waves, slot 2 * workgroup + role
#define PZ_PAIR_STAMP(role, k)                                                                                     \
    do {                                                                                                           \
        if (blockIdx.x < 4096 && lane == 0) PZ_STAMP_AT(g_pz_stamps, blockIdx.x * 2 + (role), k, "s_memrealtime"); \
    } while (0)
// where the wave runs: HW_ID (wave / SIMD / CU / SE) and XCC_ID in slot 7
#define PZ_PAIR_WHERE(role)                                                                                       \
    do {                                                                                                          \
        if ((pz::diag::kBits & 1u) && blockIdx.x < 4096 && lane == 0) {                                           \
            unsigned int hw_, xcc_;                                                                               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)"          \
                         : "=s"(hw_), "=s"(xcc_));                                                                \
            pz::diag::g_pz_stamps[(blockIdx.x * 2 + (role)) * 8 + 7] = ((unsigned long long)xcc_ << 32) | hw_;    \
        }                                                                                                         \
    } while (0)
#define PZ_DRAIN_VMEM()                                                            \
    do {                                                                           \
        if (pz::diag::kBits & 1u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    } while (0)
#else
namespace pz {
namespace diag {
constexpr unsigned kBits = 0u;
constexpr bool kDiagnosticBuild = false;
}  // namespace diag
}  // namespace pz
#define PZ_STAMP(k)
#define PZ_FRAME_STAMP(k)
#define PZ_PAIR_STAMP(role, k)
#define PZ_PAIR_WHERE(role)
#define PZ_DRAIN_VMEM()
#endif

namespace pz {
namespace diag {
constexpr bool kStamps = (kBits & 1u) != 0;
constexpr bool kNoPairKernel = (kBits & 2u) != 0;
constexpr bool kNoRolloutPair = (kBits & 4u) != 0 || kStamps;
constexpr bool kNoScoutWave = (kBits & 8u) != 0 || kStamps;
constexpr bool kUnorderedEarlyStores = (kBits & 16u) != 0;
constexpr bool kNoEarlyStores = (kBits & 32u) != 0;
constexpr int kDelayPartnerLoads = (int)((kBits >> 8) & 0xFFu);
constexpr unsigned kSubset = (kBits >> 16) & 0x3FFFu;  // 0: every instantiation
}  // namespace diag
}  // namespace pz
