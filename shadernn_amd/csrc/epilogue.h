// epilogue.h -- fused bias -> batch-norm -> activation epilogue shared by the conv kernels (device code).
// Restates shadertemplate_vk_conv2d.comp:276-340 of the reference for fp32.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/snnhip.h"

namespace snnhip {

// epi = {bias, bnScale, bnMean, bnBeta} for this output channel; acc excludes the bias.
__device__ __forceinline__ float epi_affine(float acc, const float4& epi, int useBN) {
    float v = acc + epi.x;
    if (useBN) v = (epi.y * (v - epi.z)) + epi.w;
    return v;
}

// `first` = post-activation value of pixel 0 of the aligned 4-pixel x group, used only by SILU_QUIRK for the other
// three pixels (the reference overwrites color1 before re-using it: vk_conv2d.comp:336-339).
__device__ __forceinline__ float epi_act(int act, float leaky, float v, float first) {
    switch (act) {
    case SNNHIP_ACT_RELU: return fmaxf(v, 0.0f);
    case SNNHIP_ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
    case SNNHIP_ACT_TANH: return tanhf(v);
    case SNNHIP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case SNNHIP_ACT_LEAKY: return fmaxf(v, v * leaky);
    case SNNHIP_ACT_SILU: return v * 1.0f / (1.0f + expf(-v));
    case SNNHIP_ACT_SILU_QUIRK: return v * 1.0f / (1.0f + expf(-first));
    default: return v;
    }
}

// The cheap activation family {none, relu, relu6, leakyRelu} as ONE branch-free expression
//     y = med3(max(v, v*alpha), lo, hi)
// none: (1,-inf,+inf)  relu: (1,0,+inf)  relu6: (1,0,6)  leaky: (alpha,-inf,+inf) == the shader's max(c, c*alpha).
// Kernels are instantiated twice: SIMPLE (this, 3 VALU ops, no branches in the hot loops -- a run-time switch costs
// ~16 taken branches per 16-pixel group and made the fused ESPCN kernel 2x slower) and generic (epi_act switch).
struct ActCfg {
    int act;
    float leaky;
    float alpha, lo, hi;
};

inline bool act_is_simple(int act) { return act == SNNHIP_ACT_NONE || act == SNNHIP_ACT_RELU || act == SNNHIP_ACT_RELU6 || act == SNNHIP_ACT_LEAKY; }

inline ActCfg make_act_cfg(int act, float leaky) {
    ActCfg a{act, leaky, 1.0f, -__builtin_huge_valf(), __builtin_huge_valf()};
    if (act == SNNHIP_ACT_RELU) a.lo = 0.0f;
    if (act == SNNHIP_ACT_RELU6) {
        a.lo = 0.0f;
        a.hi = 6.0f;
    }
    if (act == SNNHIP_ACT_LEAKY) a.alpha = leaky;
    return a;
}

template <bool SIMPLE>
__device__ __forceinline__ float apply_act(const ActCfg& a, float v, float first) {
    if (SIMPLE) return __builtin_amdgcn_fmed3f(fmaxf(v, v * a.alpha), a.lo, a.hi);
    return epi_act(a.act, a.leaky, v, first);
}

// Activation of a fused Add layer (chain rule E).  It is none / relu / relu6 / leakyRelu in every graph of the zoo: those go through the
// branch-free med3 form (bit-identical to epi_act for them) instead of epi_act's run-time switch.  (Measured: not what made fused adds
// slow -- that was the residual loads interleaved with the stores, see conv2d_mfma_kernel's epilogue -- but it keeps the switch out of the
// unrolled store loops.)
__device__ __forceinline__ bool act_is_simple_dev(int act) {
    return act == SNNHIP_ACT_NONE || act == SNNHIP_ACT_RELU || act == SNNHIP_ACT_RELU6 || act == SNNHIP_ACT_LEAKY;
}
__device__ __forceinline__ float add_act(const ActCfg& a, bool simple, float v) {
    return simple ? __builtin_amdgcn_fmed3f(fmaxf(v, v * a.alpha), a.lo, a.hi) : epi_act(a.act, a.leaky, v, 0.0f);
}

// tanh as 1 - 2/(e^{2x}+1) on the hardware exp/rcp units: branch-free (ocml's tanhf is a multi-range, branchy
// routine that serialises the epilogue).  Absolute error <= 3e-7 over the whole range, saturates to +-1 correctly.
// Five instructions (v_mul, v_exp, v_add, v_rcp, v_fma): __fdividef compiles to the full IEEE division sequence (div_scale / div_fmas /
// div_fixup, ~12 instructions per quotient) on this toolchain, which made the four tanh of the ESPCN tail a quarter of that kernel's
// non-FMA instructions; v_rcp_f32 is accurate to 1 ulp, far inside the 3e-7 bound.
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f); // e^{2x} = 2^{2x log2(e)}
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

// Element access: every kernel is instantiated for T = float and T = _Float16 (SNNHIP_F16 tensors: half storage, fp32 arithmetic,
// round-to-nearest-even on store) and for CV = 4 (C % 4 == 0: one 16- or 8-byte access) or CV = 1.
template <typename T, int CV>
__device__ __forceinline__ void ldv(const T* __restrict__ p, float (&v)[CV]) {
    if (CV == 8 && sizeof(T) == 2) { // eight halfs: one 16-byte access
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k % CV] = static_cast<float>(t[k]);
    } else if (CV == 4) {
        if (sizeof(T) == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x;
            v[1 % CV] = t.y;
            v[2 % CV] = t.z;
            v[3 % CV] = t.w;
        } else {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 t = *reinterpret_cast<const h4*>(p);
            v[0] = static_cast<float>(t[0]);
            v[1 % CV] = static_cast<float>(t[1]);
            v[2 % CV] = static_cast<float>(t[2]);
            v[3 % CV] = static_cast<float>(t[3]);
        }
    } else {
        v[0] = static_cast<float>(p[0]);
    }
}
template <typename T, int CV>
__device__ __forceinline__ void stv(T* __restrict__ p, const float (&v)[CV]) {
    if (CV == 8 && sizeof(T) == 2) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = static_cast<_Float16>(v[k % CV]);
        *reinterpret_cast<h8*>(p) = t;
    } else if (CV == 4) {
        if (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % CV], v[2 % CV], v[3 % CV]);
        } else {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 t;
            t[0] = static_cast<_Float16>(v[0]);
            t[1] = static_cast<_Float16>(v[1 % CV]);
            t[2] = static_cast<_Float16>(v[2 % CV]);
            t[3] = static_cast<_Float16>(v[3 % CV]);
            *reinterpret_cast<h4*>(p) = t;
        }
    } else {
        p[0] = static_cast<T>(v[0]);
    }
}

// LDS-DMA (global_load_lds_dwordx4): every lane copies 16 bytes from its own global address to LDS byte (wave-uniform base) + 16 * lane, no
// staging registers and no ds_write.  Issued as inline assembly on purpose: behind __builtin_amdgcn_global_load_lds the compiler cannot tell
// which LDS bytes the copy writes and puts s_waitcnt vmcnt(0) in front of the NEXT ds_read of the kernel -- the wave then sits out the full
// latency of a prefetch it will not touch before the next barrier (seen in conv2d_wino's chunk loop).  The asm form is invisible to that
// tracking: the caller waits with lds_dma_wait() (or any later wait for a younger load: vmcnt retires in order) before the barrier that
// publishes the data.  m0 is not otherwise used by these kernels (gfx9 ds instructions do not read it).
__device__ __forceinline__ void lds_dma16(const void* gsrc, const void* ldsWaveBase) {
    const unsigned base = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)(ldsWaveBase))));
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(base) : "memory");
}
// The same copy with a wave-uniform base pointer in SGPRs and a 32-bit byte offset per lane: no 64-bit address arithmetic on the VALU (a loader wave's
// VALU instructions queue behind the MFMAs of the compute waves it shares a SIMD with: phase trace of conv1x1_march_kernel, 300 cycles per copy).
__device__ __forceinline__ void lds_dma16_sbase(const void* uniformBase, unsigned laneByteOffset, unsigned ldsByteAddr) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(laneByteOffset), "s"(uniformBase), "s"(ldsByteAddr) : "memory");
}
// 4-byte store to (wave-uniform base in SGPRs) + a 32-bit byte offset per lane
__device__ __forceinline__ void store_dword_sbase(void* uniformBase, unsigned laneByteOffset, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(laneByteOffset), "v"(v), "s"(uniformBase) : "memory");
}
__device__ __forceinline__ unsigned lds_byte_addr(const void* ldsPtr) { // wave-uniform LDS address as a scalar
    return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)(ldsPtr))));
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// coordinate resolution of vk_conv2d.comp:168-218; returns -1 when the fetch yields 0
__device__ __forceinline__ int resolve_coord(int s, int size, int padMode) {
    if (padMode == SNNHIP_PAD_REPLICATE) return min(max(s, 0), size - 1);
    if (padMode == SNNHIP_PAD_REFLECT) {
        s = s < 0 ? -s : s;
        s = s >= size ? 2 * size - 2 - s : s;
    }
    return (s >= 0 && s < size) ? s : -1;
}

// resolve_coord without its switch, for staging code that resolves many coordinates per thread: the mode is uniform, but every taken s_cbranch
// still drains the pipeline (conv2d_rowfold with a fused reflect Pad spent more on these branches than a separate Pad launch costs)
__device__ __forceinline__ int resolve_nobranch(int s, int size, int mode) {
    const int cl = min(max(s, 0), size - 1);
    int rf = s < 0 ? -s : s;
    rf = rf >= size ? 2 * size - 2 - rf : rf;
    const int t = mode == SNNHIP_PAD_REPLICATE ? cl : (mode == SNNHIP_PAD_REFLECT ? rf : s);
    return (t >= 0 && t < size) ? t : -1;
}

} // namespace snnhip
