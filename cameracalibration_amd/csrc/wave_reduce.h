// wave_reduce.h — all-lanes reductions of one wavefront (64 lanes) without trips through LDS.
//
// __shfl_xor compiles to ds_bpermute_b32: an LDS round trip (> 100 cycles) per 32 bits and stage, twelve of them
// in a row for a 64-bit butterfly, and the per-item kernels here (one wave per contour, quad, corner, frame) are
// chains of such reductions.  The butterfly below pairs the same lanes in the same order (32, 16, 8, 4, 2, 1), so
// floating-point sums round exactly as with __shfl_xor, but moves the data with v_permlane32_swap /
// v_permlane16_swap (gfx950) and DPP row rotations.  (Stages 8..1 rotate instead of exchanging: by then all lanes
// that are equal modulo 2 * stage hold equal values, and lane + stage is the xor partner modulo 2 * stage.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace camsync {

namespace wave_detail {

struct Pair32 { uint32_t a, b; };

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// the two operands of butterfly stage D as this lane sees them
template <int D>
__device__ __forceinline__ Pair32 stage(uint32_t v) {
    if constexpr (D == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {low half, low half}, {high, high}
        return Pair32{r[0], r[1]};
    } else if constexpr (D == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // {rows 0,0,2,2}, {rows 1,1,3,3}
        return Pair32{r[0], r[1]};
    } else {
        return Pair32{v, dpp_mov<0x120 + D>(v)};  // row_ror:D
    }
}

template <int D, class T, class Op>
__device__ __forceinline__ T step(T v, Op op) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        uint32_t u;
        memcpy(&u, &v, 4);
        const Pair32 p = stage<D>(u);
        T a, b;
        memcpy(&a, &p.a, 4);
        memcpy(&b, &p.b, 4);
        return op(a, b);
    } else {
        uint32_t u[2];
        memcpy(u, &v, 8);
        const Pair32 lo = stage<D>(u[0]), hi = stage<D>(u[1]);
        const uint32_t ua[2] = {lo.a, hi.a}, ub[2] = {lo.b, hi.b};
        T a, b;
        memcpy(&a, ua, 8);
        memcpy(&b, ub, 8);
        return op(a, b);
    }
}

}  // namespace wave_detail

// op must be commutative; every lane must be active; returns the reduction over the 64 lanes in every lane
template <class T, class Op>
__device__ __forceinline__ T wave_allreduce(T v, Op op) {
    v = wave_detail::step<32>(v, op);
    v = wave_detail::step<16>(v, op);
    v = wave_detail::step<8>(v, op);
    v = wave_detail::step<4>(v, op);
    v = wave_detail::step<2>(v, op);
    v = wave_detail::step<1>(v, op);
    return v;
}

}  // namespace camsync
