// pz_memory.hpp -- the lowest layer of the gfx950 kernels: launch geometry constants, buffer descriptors and the
// row flush of the observation tensors.  Shared by pz_kernels.hip (the product, libpikazoo_hip.so) and pz_diag.hip
// (libpikazoo_diag.so: diagnostics that replay the product launch's memory pattern without its game).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pikazoo_hip.h"
#include "pz_physics.hpp"

namespace pz {

constexpr int kLanes = kWaveGames;  // lanes (games) per workgroup = one wavefront
constexpr uint32_t kRowBytes = PZ_OBS_DIM * 4;            // 140
constexpr uint32_t kWaveObsBytes = kLanes * kRowBytes;    // 8 960: a wave's rows are contiguous
constexpr int kWaveObsVecs = (int)(kWaveObsBytes / 16);   // 560 16-byte pieces

// cache-policy bits of the single-frame launches' stores (aux operand: 1 = sc0, 2 = nt, 16 = sc1), as measured:
constexpr int kStateAux = 0;  // the state is re-read by the next launch: default policy
constexpr int kObsAux = 2;    // observations are written once and not re-read by the step chain: nt, -2.4 % per launch

using Rsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor over [p, p + bytes): stride 0, 32-bit data format (gfx9 family word 3)
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

// Copy the wave's 64 staged rows (8 960 B) to its span of an [n][35] tensor (`tensor_bytes` = n rows):
// 9 passes of 16 B per lane.  The span gets its own descriptor, which ends at row n, so the rows of
// lanes past the end of the batch are dropped by the range check -- and the stores carry no SGPR offset.
// That matters: for a buffer store of more than 64 bits WITH an SGPR offset the compiler assumes no
// wait state is needed before a VALU write of the store's data registers (LLVM
// GCNHazardRecognizer::createsVALUHazard) and schedules e.g. the next address computation into
// them; on gfx950 a quarter of the wave then stores the new value (seen as address bits in
// observation words).  tests/test_cabi_and_host.py scans the built code object for that pattern.
__device__ __forceinline__ void flush_rows(const int32_t* __restrict__ lds, const void* tensor, uint32_t tensor_bytes,
                                           int lane)
{
    const uint32_t wave_off = blockIdx.x * kWaveObsBytes;
    const Rsrc span = make_rsrc(static_cast<const char*>(tensor) + wave_off, tensor_bytes - wave_off);
    const u32x4* src4 = reinterpret_cast<const u32x4*>(lds);
#pragma unroll
    for (int pass = 0; pass < (kWaveObsVecs + kLanes - 1) / kLanes; ++pass) {
        const int v = pass * kLanes + lane;
        if (v < kWaveObsVecs) __builtin_amdgcn_raw_buffer_store_b128(src4[v], span, (uint32_t)v * 16u, 0, kObsAux);
    }
}

}  // namespace pz
