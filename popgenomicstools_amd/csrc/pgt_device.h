// pgt_device.h — device-side helpers shared by the HIP translation units (wave64 cross-lane sums,
// streaming loads).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pgt_internal.h"

namespace pgt {
namespace dev {

// ------------------------------------------------------------------------------------------
// wave64 cross-lane sums.  Steps 1-4 stay inside a 16-lane DPP row (quad_perm, then the two
// mirror controls, valid because after each step the value is uniform inside the sub-group);
// steps 5-6 cross rows through ds_bpermute.  Every lane returns the same bits.
// ------------------------------------------------------------------------------------------
constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<kDppQuadXor1>(v);
    v += dpp_f64<kDppQuadXor2>(v);
    v += dpp_f64<kDppRowHalfMirror>(v);
    v += dpp_f64<kDppRowMirror>(v);
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += dpp_u32<kDppQuadXor1>(v);
    v += dpp_u32<kDppQuadXor2>(v);
    v += dpp_u32<kDppRowHalfMirror>(v);
    v += dpp_u32<kDppRowMirror>(v);
    v += (uint32_t)__shfl_xor((int)v, 16, kWave);
    v += (uint32_t)__shfl_xor((int)v, 32, kWave);
    return v;
}

// Column loads; NT = non-temporal hint (`global_load_dwordx4 ... nt`): the columns are read exactly
// once per pass, and in interleaved A/B runs the hint was worth +5 % (10^9 sites) to +17 % (10^8).
template <bool NT>
__device__ __forceinline__ double2 load16(const double2 *p) {
    if constexpr (NT) {
        double2 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    } else {
        return *p;
    }
}
__device__ __forceinline__ uint4 load16_nt(const uint4 *p) {
    uint4 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z);
    v.w = __builtin_nontemporal_load(&p->w);
    return v;
}
// "The value has arrived; later loads stay behind": an empty asm that reads the register (the compiler places the
// s_waitcnt for it here) and clobbers memory (no later load is moved above it).  Separates the column bursts of the
// build kernels: one column at a time, a short queue (pgt_kernels.hip: fst_build_kernel).
__device__ __forceinline__ void load_fence(double v) { asm volatile("" ::"v"(v) : "memory"); }
__device__ __forceinline__ void load_fence(int v) { asm volatile("" ::"v"(v) : "memory"); }

__device__ __forceinline__ int4 load16_nt(const int4 *p) {
    int4 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z);
    v.w = __builtin_nontemporal_load(&p->w);
    return v;
}


}  // namespace dev
}  // namespace pgt
