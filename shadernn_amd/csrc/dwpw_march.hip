// dwpw_march.hip -- DepthwiseConv2D 3x3 (stride 1, "same") -> Conv2D 1x1 as ONE row-marching, HBM-streaming kernel (chain rule G without an expand
// layer: MobileNetV2's first block, 112x112x32 -> 16 at batch 256 -- 617 MB in and out, 5 GFLOP: a layer pair that IS its input / output stream).
//
// Replaces two passes of the reference (shadertemplate_vk_depthwise.comp:64-137, shadertemplate_vk_conv2d_1x1.comp:68-210) and, for this shape,
// round 2's wave-autonomous tile kernel (irb_fused.hip, no-expand mode): that one re-fetched its 6x10 halo tiles 1.6x and kept 12 waves x 8 KB in
// flight per CU -- bandwidth through latency, 229 us = 2.7 TB/s.  Here:
//
//   * a block owns a run of rows of ONE image over the full width and marches down it; every input row is fetched from HBM exactly once (plus one
//     halo row at each end of the run), by a dedicated LOADER wave that does nothing else: LDS-DMA (global_load_lds_dwordx4), three rows in flight,
//     into a ring of six row slots.  The loader's vmcnt only ever counts loads, so its waits are COUNTED (on gfx9 loads and stores share vmcnt and
//     retire out of order with respect to each other: a wave that also stores can only wait with vmcnt(0), i.e. drain its prefetches);
//   * the COMPUTE waves (seven beside one loader) own one 16-pixel group of the row each (112 pixels = 7 groups): lane = (pixel, channel-quad group g); the depthwise taps
//     are 9 ds_read_b128 per channel quad from the three resident rows (pixels 144 bytes apart: 8 data slots + 1 zero slot from the DMA, an odd
//     number of 16-byte slots), 4 FMAs each with the weights in registers; bias / BN / activation; the lane's float4s ARE the B operands of the
//     pointwise layer's v_mfma_f32_16x16x4_f32 (K order = the lanes' channel order, weights pre-permuted on the host); epilogue; one 16-byte store per
//     lane = 1 KB contiguous per wave;
//   * one s_barrier per output row: the loader has waited for row r + 1, the compute waves have left row r - 1; then the loader reuses the slot of
//     row r - 2 for row r + 4.  Rows above / below the image and the pad pixels are DMA'd from a block of zeros: no border code in the taps.
#include "epilogue.h"
#include "snnhip_internal.h"

#include <cstring>

namespace snnhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;  // waves per block (512 threads, one block per CU): NLW loader waves + (8 - NLW) compute waves

struct DwPwParams {
    int N, H, W, C, Co;
    int nseg, rowsPerSeg; // a block = rows [seg * rowsPerSeg, min(H, +rowsPerSeg)) of image blockIdx.x / nseg
    int groups;           // 16-pixel groups per row
    int slotFloats;       // floats per ring slot
    ActCfg acD, acP;
};

// s_waitcnt vmcnt(n) for a compile-time n (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; the other counters left at their maximum)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// What a compute lane (pixel px16 of its 16-pixel group, lane group g) keeps in registers for the depthwise -> pointwise pair.  The lane owns channel
// quads q = g + 4 qi (channels 16 qi + 4 g + i): depthwise weights per (quad, tap), the depthwise epilogue (scale, shift), the pointwise A operands of
// its K steps (host-permuted to the lanes' channel order) and the pointwise epilogue of its four output channels per block.
template <int CQ, int NCB>
struct DwPwRegs {
    f32x4 wdw[CQ][9], scD[CQ], shD[CQ], scP[NCB], shP[NCB];
    float a[NCB][4 * CQ];
    __device__ __forceinline__ void load(int lane, const float4* __restrict__ wd4, const float* __restrict__ wA, const float4* __restrict__ epiD, const float4* __restrict__ epiP) {
        const int g = lane >> 4;
#pragma unroll
        for (int qi = 0; qi < CQ; ++qi) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 w = wd4[(g * CQ + qi) * 9 + t];
                wdw[qi][t] = f32x4{w.x, w.y, w.z, w.w};
            }
            const float4 s4 = epiD[(g * CQ + qi) * 2], h4 = epiD[(g * CQ + qi) * 2 + 1];
            scD[qi] = f32x4{s4.x, s4.y, s4.z, s4.w};
            shD[qi] = f32x4{h4.x, h4.y, h4.z, h4.w};
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int s = 0; s < 4 * CQ; ++s) a[cb][s] = wA[(cb * 4 * CQ + s) * 64 + lane];
            const float4 s4 = epiP[(cb * 4 + g) * 2], h4 = epiP[(cb * 4 + g) * 2 + 1];
            scP[cb] = f32x4{s4.x, s4.y, s4.z, s4.w};
            shP[cb] = f32x4{h4.x, h4.y, h4.z, h4.w};
        }
    }
};

// One 16-pixel group of one output row: depthwise taps from the three resident rows (rowsL[dy], pixel 0 = the left pad column, QP 16-byte slots per
// pixel) -> epilogue -> pointwise MFMAs -> epilogue -> one 16-byte store per lane and output block.  pc = this lane's output column; yrow = the output row.
template <int CQ, int NCB>
__device__ __forceinline__ void dwpw_group(const DwPwRegs<CQ, NCB>& R, const float* const (&rowsL)[3], int pc, int g, int W, int Co, const ActCfg& acD, const ActCfg& acP,
                                           float* __restrict__ yrow) {
    constexpr int QP = 4 * CQ + 1;
    // all 9 CQ operand reads first (independent: one LDS latency for the lot), then the FMAs
    f32x4 v[3][3][CQ];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* base = rowsL[dy] + (pc * QP + g) * 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int qi = 0; qi < CQ; ++qi) v[dy][dx][qi] = *reinterpret_cast<const f32x4*>(base + (dx * QP + 4 * qi) * 4);
    }
    f32x4 acc[CQ];
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) acc[qi] = R.wdw[qi][0] * v[0][0][qi];
#pragma unroll
    for (int t = 1; t < 9; ++t)
#pragma unroll
        for (int qi = 0; qi < CQ; ++qi) acc[qi] += R.wdw[qi][t] * v[t / 3][t % 3][qi];
    // two accumulation chains per output block (a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, an independent one every 32)
    f32x4 d[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) d[cb][0] = d[cb][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) {
        f32x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = apply_act<true>(acD, fmaf(acc[qi][i], R.scD[qi][i], R.shD[qi][i]), 0.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) d[cb][i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.a[cb][qi * 4 + i], h[i], d[cb][i & 1], 0, 0, 0);
    }
    if (pc < W) {
        float* yo = yrow + static_cast<size_t>(pc) * Co + 4 * g;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            if (16 * cb + 4 * g < Co) {
                const f32x4 dd = d[cb][0] + d[cb][1];
                float4 o;
                o.x = apply_act<true>(acP, fmaf(dd[0], R.scP[cb][0], R.shP[cb][0]), 0.0f);
                o.y = apply_act<true>(acP, fmaf(dd[1], R.scP[cb][1], R.shP[cb][1]), 0.0f);
                o.z = apply_act<true>(acP, fmaf(dd[2], R.scP[cb][2], R.shP[cb][2]), 0.0f);
                o.w = apply_act<true>(acP, fmaf(dd[3], R.scP[cb][3], R.shP[cb][3]), 0.0f);
                *reinterpret_cast<float4*>(yo + 16 * cb) = o;
            }
    }
}

// CQ = channel quads per lane (C = 16 CQ), NCB = 16-channel blocks of the pointwise output, CH = 1 KB DMA pieces per row, NLW = loader waves (each copies
// CH / NLW pieces of every row), PF = rows in flight behind the row pair the compute waves need next (PF * CH / NLW <= 63: vmcnt is a 6-bit counter);
// the ring holds PF + 3 rows
template <int CQ, int NCB, int CH, int NLW, int PF>
__global__ __launch_bounds__(64 * kWaves, 2) void dwpw_march_kernel(DwPwParams p, const float* __restrict__ x, const float4* __restrict__ wd4, const float* __restrict__ wA,
                                                                       const float4* __restrict__ epiD, const float4* __restrict__ epiP, const float* __restrict__ zeros,
                                                                       float* __restrict__ y) {
    constexpr int kNCW = kWaves - NLW, kRing = PF + 3, CW = CH / NLW;
    static_assert(CH % NLW == 0 && PF * CW <= 63 && PF >= 2, "loader geometry");
    constexpr int QP = 4 * CQ + 1; // 16-byte slots per pixel in LDS: C / 4 of data + 1 of zeros (odd: consecutive pixels start in different bank groups)
    extern __shared__ __attribute__((aligned(16))) float ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / p.nseg, seg = blockIdx.x - n * p.nseg;
    const int r0 = seg * p.rowsPerSeg, r1 = min(p.H, r0 + p.rowsPerSeg);
    const bool loader = wave >= kNCW;
    const int lw = wave - kNCW; // loader wave index: pieces lw * CW .. lw * CW + CW - 1 of every row
    // ring slot of image row rho (rho >= r0 - 1): (rho - (r0 - 1)) mod kRing
    auto slot_of = [&](int rho) { return ring + ((rho - (r0 - 1)) % kRing) * p.slotFloats; };

    // ---- loader state: source offset (floats, from the row's first pixel) of every 16-byte element this lane copies; -1 = zeros
    int gofs[CW];
    if (loader) {
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int e = 64 * (lw * CW + k) + lane, px = e / QP, ql = e - px * QP;
            gofs[k] = (px < p.W && ql < QP - 1) ? px * p.C + ql * 4 : -1;
        }
    }
    auto issue_row = [&](int rho) { // (wave-uniform)
        const bool inside = rho >= 0 && rho < p.H && rho <= r1; // (rows past the run's lower halo row are never read: zeros, no HBM traffic)
        const float* xrow = x + (static_cast<size_t>(n) * p.H + (inside ? rho : 0)) * p.W * p.C;
        float* dst = slot_of(rho) + QP * 4 + lw * CW * 256; // pixel 0 of the slot is the left pad column
#pragma unroll
        for (int k = 0; k < CW; ++k) lds_dma16((inside && gofs[k] >= 0) ? xrow + gofs[k] : zeros, dst + k * 256);
    };

    // ---- compute-wave state
    const int px16 = lane & 15, g = lane >> 4;
    DwPwRegs<CQ, NCB> R;
    if (!loader) {
        R.load(lane, wd4, wA, epiD, epiP);
        // the left pad pixel of every slot (the DMA never writes it) is zero for the life of the block
        for (int i = tid; i < kRing * QP * 4; i += 64 * kNCW) ring[(i / (QP * 4)) * p.slotFloats + i % (QP * 4)] = 0.0f;
        wait_vmcnt<0>(); // the weight loads: from here on this wave's vmcnt only sees stores
    } else {
        issue_row(r0 - 1);
        issue_row(r0);
        issue_row(r0 + 1);
    }

    // Two loops, one per role, with the SAME number of barriers (one per output row): in a single loop the register allocation is the sum of the
    // loader's and the compute waves' state (it spilled the loader's source pointers to scratch, each reload behind an s_waitcnt vmcnt(0) that drained
    // the prefetches); apart it is the maximum.  s_barrier counts arrivals, not call sites.
    if (loader) {
        for (int r = r0; r < r1; ++r) {
            if (r == r0) wait_vmcnt<0>();          // rows r0 - 1, r0, r0 + 1
            else wait_vmcnt<(PF - 1) * CW>();      // row r + 1 has landed; rows r + 2 .. r + PF may be in flight
            __syncthreads();                       // row r + 1 is published; every compute wave has left row r - 1
            if (r == r0) {
#pragma unroll
                for (int k = 2; k <= PF; ++k) issue_row(r0 + k);
            }
            issue_row(r + PF + 1); // the slot of row r - 2
        }
        wait_vmcnt<0>(); // no copy may still be on its way into LDS when the block ends
        return;
    }
    for (int r = r0; r < r1; ++r) {
        __syncthreads();
        const float* const rowsL[3] = {slot_of(r - 1), slot_of(r), slot_of(r + 1)};
        for (int grp = wave; grp < p.groups; grp += kNCW)
            dwpw_group<CQ, NCB>(R, rowsL, grp * 16 + px16, g, p.W, p.Co, p.acD, p.acP, y + (static_cast<size_t>(n) * p.H + r) * p.W * p.Co);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// The same march with the network's STEM in front: Conv2D 3x3 stride 2 (RGB -> 32 channels, "same") -> DepthwiseConv2D 3x3 -> Conv2D 1x1, the head of
// MobileNetV2, as ONE launch.  The stem's output (411 MB at batch 256: written by one kernel, read back by the next) never reaches memory: the
// loader wave streams IMAGE rows (224 x 3 floats, verbatim, 3 x 1 KB) into a ring of image rows, the compute waves turn three of them into one stem
// row -- im2col on the fly: K = 27 (+1) = 7 steps of v_mfma_f32_16x16x4_f32 per 16 pixels and output block, the B operand gathered with one ds_read_b32
// per step -- and write it (bias / BN / activation applied) into the ring of stem rows in exactly the layout the depthwise taps read.  Marching
// means no vertical halo: every stem row is computed once per row run (round 3's attempt inside the tile kernel computed the stem on 6x10 halos of
// 4x8 tiles, 1.9x its work, and was issue-bound at 403 us for the pair of launches it replaced).  Iteration r computes stem row r + 2 and output row r.
struct StemDwPwParams {
    DwPwParams d;       // the depthwise -> pointwise pair on the stem's output grid (d.H x d.W x 32)
    int IH, IW;         // image extent (IH = 2 d.H, IW = 2 d.W)
    int imgSlotFloats;  // floats per image-row slot (4 of lead: the left pad pixel sits in its last 12 bytes)
    int stemOfs;        // float offset of the stem-row ring behind the image-row ring
    ActCfg acS;
};

template <int NCB, int PI>
__global__ __launch_bounds__(64 * kWaves, 2) void stem_dwpw_march_kernel(StemDwPwParams sp, const float* __restrict__ x, const float* __restrict__ wS, const float4* __restrict__ epiS,
                                                                        const float4* __restrict__ wd4, const float* __restrict__ wA, const float4* __restrict__ epiD,
                                                                        const float4* __restrict__ epiP, const float* __restrict__ zeros, float* __restrict__ y) {
    constexpr int CQ = 2, QP = 4 * CQ + 1, kNCW = kWaves - 1;
    constexpr int PFI = (63 / (2 * PI)) < 8 ? (63 / (2 * PI)) : 8; // iterations of image rows in flight (2 rows = 2 PI pieces each)
    constexpr int IR = 2 * PFI + 4;                                // image-row slots
    constexpr int SR = 4;                                          // stem-row slots: rows r - 1, r, r + 1 are read while r + 2 is written
    const DwPwParams& p = sp.d;
    extern __shared__ __attribute__((aligned(16))) float ring[];
    float* const stemRing = ring + sp.stemOfs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / p.nseg, seg = blockIdx.x - n * p.nseg;
    const int r0 = seg * p.rowsPerSeg, r1 = min(p.H, r0 + p.rowsPerSeg);
    const int rs = r0 - 3; // first iteration: computes stem row r0 - 1
    auto img_slot = [&](int rho) { return ring + ((rho + 4 * IR) % IR) * sp.imgSlotFloats + 4; };   // (rho >= -7)
    auto stem_slot = [&](int srow) { return stemRing + ((srow + 4 * SR) % SR) * p.slotFloats; };

    if (wave == kNCW) {
        // ---- loader: image row rho -> its slot, PI pieces of 1 KB (the row, then zeros: the right pad pixel).  Unit u = the two rows 2 r + 4, 2 r + 5 of iteration r.
        int gofs[PI];
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int e = 64 * k + lane;
            gofs[k] = e * 4 < sp.IW * 3 ? e * 4 : -1;
        }
        auto issue_img = [&](int rho) {
            const bool inside = rho >= 0 && rho < sp.IH && rho <= 2 * r1 + 1; // (stem rows past r1 are never computed)
            const float* xrow = x + (static_cast<size_t>(n) * sp.IH + (inside ? rho : 0)) * sp.IW * 3;
            float* dst = img_slot(rho);
#pragma unroll
            for (int k = 0; k < PI; ++k) lds_dma16((inside && gofs[k] >= 0) ? xrow + gofs[k] : zeros, dst + k * 256);
        };
        issue_img(2 * rs + 3); // the top row of the first stem row
#pragma unroll
        for (int u = 0; u < PFI; ++u) {
            issue_img(2 * (rs + u) + 4);
            issue_img(2 * (rs + u) + 5);
        }
        for (int r = rs; r < r1; ++r) {
            wait_vmcnt<(PFI - 1) * 2 * PI>(); // rows <= 2 r + 5 have landed
            __syncthreads();
            issue_img(2 * (r + PFI) + 4);
            issue_img(2 * (r + PFI) + 5);
        }
        wait_vmcnt<0>();
        return;
    }

    // ---- compute waves
    const int px16 = lane & 15, g = lane >> 4;
    DwPwRegs<CQ, NCB> R;
    R.load(lane, wd4, wA, epiD, epiP);
    // stem: A operands (weights of K step s at lane (m = oc, k = g): k = 4 s + g = 3 tap + channel; k = 27 is zero), epilogue of the lane's four channels per
    // output block, and where K step s of this lane reads: tap row dy (0..2) and byte offset of (tap column dx, channel c) from the pixel pair's first byte
    float aS[2][7];
    f32x4 scS[2], shS[2];
    int tapDy[7], tapOfs[7];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) aS[t2][s7] = wS[(t2 * 7 + s7) * 64 + lane];
        const float4 s4 = epiS[(t2 * 4 + g) * 2], h4 = epiS[(t2 * 4 + g) * 2 + 1];
        scS[t2] = f32x4{s4.x, s4.y, s4.z, s4.w};
        shS[t2] = f32x4{h4.x, h4.y, h4.z, h4.w};
    }
#pragma unroll
    for (int s7 = 0; s7 < 7; ++s7) {
        const int k = min(4 * s7 + g, 26), tap = (k * 11) >> 5, c = k - 3 * tap; // (k * 11) >> 5 == k / 3 for k < 32
        tapDy[s7] = tap / 3;
        tapOfs[s7] = ((tap % 3 - 1) * 3 + c) * 4; // bytes: tap column dx - 1 relative to image pixel 2 pc
    }
    // pad columns of the stem-row slots (pixel 0 and pixel W + 1) and the lead of the image-row slots (pixel -1): zero for the life of the block
    for (int i = tid; i < SR * 2 * QP * 4; i += 64 * kNCW) {
        const int sl = i / (2 * QP * 4), e = i % (2 * QP * 4);
        stemRing[sl * p.slotFloats + (e < QP * 4 ? e : (p.W + 1) * QP * 4 + e - QP * 4)] = 0.0f;
    }
    for (int i = tid; i < IR * 4; i += 64 * kNCW) ring[(i >> 2) * sp.imgSlotFloats + (i & 3)] = 0.0f;
    wait_vmcnt<0>();

    // one 16-pixel group of stem row srow: gathers, MFMAs, epilogue; the caller stores d0 / d1 (channel quads g and 4 + g of the lane's pixel)
    auto stem_group = [&](const char* const (&rowB)[3], int pc, f32x4& d0, f32x4& d1) {
        float bv[7];
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            const char* rb = tapDy[s7] == 0 ? rowB[0] : (tapDy[s7] == 1 ? rowB[1] : rowB[2]);
            bv[s7] = *reinterpret_cast<const float*>(rb + pc * 24 + tapOfs[s7]);
        }
        d0 = d1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[0][s7], bv[s7], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[1][s7], bv[s7], d1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d0[i] = apply_act<true>(sp.acS, fmaf(d0[i], scS[0][i], shS[0][i]), 0.0f);
            d1[i] = apply_act<true>(sp.acS, fmaf(d1[i], scS[1][i], shS[1][i]), 0.0f);
        }
    };
    for (int r = rs; r < r1; ++r) {
        __syncthreads(); // image rows <= 2 r + 5 are published; stem row r + 1 (written in the previous iteration) is complete
        const int srow = r + 2;
        float* const dstRow = stem_slot(srow);
        const char* const rowB[3] = {reinterpret_cast<const char*>(img_slot(2 * srow - 1)), reinterpret_cast<const char*>(img_slot(2 * srow)),
                                     reinterpret_cast<const char*>(img_slot(2 * srow + 1))};
        const float* const rowsL[3] = {stem_slot(r - 1), stem_slot(r), stem_slot(r + 1)};
        float* const yrow = y + (static_cast<size_t>(n) * p.H + r) * p.W * p.Co;
        if (srow < p.H && srow <= r1 && r >= r0) {
            // steady state: stem row r + 2 and output row r of the same group in ONE straight-line body -- the two are independent (the stem row is
            // written to a slot nobody reads before the next barrier), so the scheduler overlaps the stem's gathers and MFMA chain with the taps' reads
            // and FMAs instead of running them back to back
            for (int grp = wave; grp < p.groups; grp += kNCW) {
                const int pc = grp * 16 + px16;
                f32x4 d0, d1;
                stem_group(rowB, pc, d0, d1);
                dwpw_group<CQ, NCB>(R, rowsL, pc, g, p.W, p.Co, p.acD, p.acP, yrow);
                if (pc < p.W) { // lane (pixel, j = g) holds channels 16 t + 4 j + i = quad 4 t + j of the pixel
                    *reinterpret_cast<f32x4*>(dstRow + ((pc + 1) * QP + g) * 4) = d0;
                    *reinterpret_cast<f32x4*>(dstRow + ((pc + 1) * QP + 4 + g) * 4) = d1;
                }
            }
            continue;
        }
        if (srow <= r1) { // ---- the run's first and last rows: stem row srow -> its slot (zeros outside the image: the depthwise layer's zero padding)
            const bool inside = srow >= 0 && srow < p.H;
            for (int grp = wave; grp < p.groups; grp += kNCW) {
                const int pc = grp * 16 + px16;
                f32x4 d0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, d1 = d0;
                if (inside) stem_group(rowB, pc, d0, d1);
                if (pc < p.W) {
                    *reinterpret_cast<f32x4*>(dstRow + ((pc + 1) * QP + g) * 4) = d0;
                    *reinterpret_cast<f32x4*>(dstRow + ((pc + 1) * QP + 4 + g) * 4) = d1;
                }
            }
        }
        if (r >= r0) {
            for (int grp = wave; grp < p.groups; grp += kNCW) dwpw_group<CQ, NCB>(R, rowsL, grp * 16 + px16, g, p.W, p.Co, p.acD, p.acP, yrow);
        }
    }
}

typedef void (*DwPwFn)(DwPwParams, const float*, const float4*, const float*, const float4*, const float4*, const float*, float*);

struct DwPwPlan : snnhip_plan {
    DwPwParams p;
    float *d_wd = nullptr, *d_wA = nullptr, *d_eD = nullptr, *d_eP = nullptr, *d_zero = nullptr;
    DwPwFn kernel = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "dwpw_march: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C && x->dtype == SNNHIP_F32, "dwpw_march: input dims %dx%dx%dx%d != plan %dx%dx%dx%d fp32",
                       x->n, x->h, x->w, x->c, p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.H && out->w == p.W && out->c == p.Co && out->dtype == SNNHIP_F32, "dwpw_march: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.H, p.W, p.Co);
        SNNHIP_LAUNCH(kernel, grid, dim3(64 * kWaves), ldsBytes, ctx->stream, p, x->data, reinterpret_cast<const float4*>(d_wd), d_wA, reinterpret_cast<const float4*>(d_eD),
                      reinterpret_cast<const float4*>(d_eP), d_zero, out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

// loader geometry per piece count: one loader wave with 3 rows in flight where that is 48 KB (CH = 16: MobileNetV2's 112-pixel rows), otherwise the deepest that fits vmcnt
template <int CQ, int NCB>
DwPwFn pick_dwpw(int ch, int nlw, int* ring, int* ncw) {
    DwPwFn fn = nullptr;
    int pf = 0;
    if (ch <= 8) { fn = dwpw_march_kernel<CQ, NCB, 8, 1, 6>; pf = 6; nlw = 1; }
    else if (ch <= 16 && nlw == 2) { fn = dwpw_march_kernel<CQ, NCB, 16, 2, 5>; pf = 5; }
    else if (ch <= 16) { fn = dwpw_march_kernel<CQ, NCB, 16, 1, 3>; pf = 3; nlw = 1; }
    else if (ch <= 20) { fn = dwpw_march_kernel<CQ, NCB, 20, 2, 4>; pf = 4; nlw = 2; }
    *ring = pf + 3;
    *ncw = kWaves - nlw;
    return fn;
}

// (scale, shift) of act(acc * scale + shift): bias and batch norm folded (vk_conv2d.comp:277-288)
void fold_row(const std::vector<float>& epi4, int c, int useBN, float* sc, float* sh) {
    const float bias = epi4[c * 4 + 0], s = epi4[c * 4 + 1], mean = epi4[c * 4 + 2], beta = epi4[c * 4 + 3];
    *sc = useBN ? s : 1.0f;
    *sh = useBN ? s * (bias - mean) + beta : bias;
}

// the lane-ordered parameter blobs of a depthwise -> pointwise pair (DwPwRegs::load): depthwise weights [g][qi][tap] x float4, its epilogue rows [g][qi][2] x
// float4, the pointwise A operand of K step s = 4 qi + i at lane (m = oc, k = g) = W[16 cb + m][16 qi + 4 g + i], the pointwise epilogue [cb][j][2] x float4
void pack_dwpw(const ConvPlanBase* cd, const ConvPlanBase* cp, int C, int Co, std::vector<float>& wd, std::vector<float>& eD, std::vector<float>& wA, std::vector<float>& eP) {
    const int CQ = C / 16, NCB = up_div(Co, 16);
    wd.assign(static_cast<size_t>(4) * CQ * 9 * 4, 0.0f);
    eD.assign(static_cast<size_t>(4) * CQ * 2 * 4, 0.0f);
    wA.assign(static_cast<size_t>(NCB) * 4 * CQ * 64, 0.0f);
    eP.assign(static_cast<size_t>(NCB) * 4 * 2 * 4, 0.0f);
    for (int g = 0; g < 4; ++g)
        for (int qi = 0; qi < CQ; ++qi)
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * qi + 4 * g + i;
                for (int t = 0; t < 9; ++t) wd[((static_cast<size_t>(g) * CQ + qi) * 9 + t) * 4 + i] = cd->w_oihw[static_cast<size_t>(c) * 9 + t];
                fold_row(cd->epi4, c, cd->g.useBN, &eD[((static_cast<size_t>(g) * CQ + qi) * 2 + 0) * 4 + i], &eD[((static_cast<size_t>(g) * CQ + qi) * 2 + 1) * 4 + i]);
            }
    for (int cb = 0; cb < NCB; ++cb) {
        for (int s = 0; s < 4 * CQ; ++s)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, k = l >> 4, oc = 16 * cb + m, c = 16 * (s / 4) + 4 * k + (s % 4);
                if (oc < Co) wA[(static_cast<size_t>(cb) * 4 * CQ + s) * 64 + l] = cp->w_oihw[static_cast<size_t>(oc) * C + c];
            }
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                const int oc = 16 * cb + 4 * j + i;
                if (oc < Co) fold_row(cp->epi4, oc, cp->g.useBN, &eP[((static_cast<size_t>(cb) * 4 + j) * 2 + 0) * 4 + i], &eP[((static_cast<size_t>(cb) * 4 + j) * 2 + 1) * 4 + i]);
            }
    }
}

} // namespace

// dwPlan: DepthwiseConv2D 3x3 stride 1 "same" (fp32, C in {16, 32}); pwPlan: Conv2D 1x1 on its output (Co <= 32, Co % 4 == 0); simple activations.
// Borrowed plans: only their host-side descriptions are read.  SNNHIP_E_UNSUPPORTED for everything else (the caller falls back to irb_fused).
int make_dwpw_march_plan(snnhip_ctx* ctx, snnhip_plan* dwPlan, snnhip_plan* pwPlan, snnhip_plan** out) {
    if (const char* e = snnhip::option("SNNHIP_DWPW_MARCH"); e && atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    auto* cd = dynamic_cast<ConvPlanBase*>(dwPlan);
    auto* cp = dynamic_cast<ConvPlanBase*>(pwPlan);
    if (!cd || !cp || !cd->depthwise || cp->depthwise) return SNNHIP_E_UNSUPPORTED;
    const ConvGeom &gd = cd->g, &gp = cp->g;
    if (gd.dtype != SNNHIP_F32 || gp.dtype != SNNHIP_F32 || gd.kh != 3 || gd.kw != 3 || gd.sh != 1 || gd.sw != 1 || gd.padx != 1 || gd.pady != 1 || gd.preMode != 0 ||
        (gd.padMode != SNNHIP_PAD_CONSTANT && gd.padMode != SNNHIP_PAD_NONE) || gd.OH != gd.H || gd.OW != gd.W)
        return SNNHIP_E_UNSUPPORTED;
    if (gp.kh != 1 || gp.kw != 1 || gp.sh != 1 || gp.sw != 1 || gp.preMode != 0 || gp.addAct >= 0 || gp.normShift || gd.normShift || gp.IC != gd.OC || gp.N != gd.N ||
        gp.H != gd.OH || gp.W != gd.OW)
        return SNNHIP_E_UNSUPPORTED;
    if (!act_is_simple(gd.act) || !act_is_simple(gp.act)) return SNNHIP_E_UNSUPPORTED;
    const int C = gd.IC, Co = gp.OC, W = gd.W, H = gd.H, N = gd.N;
    if ((C != 16 && C != 32) || Co % 4 != 0 || Co > 32) return SNNHIP_E_UNSUPPORTED; // (64 channels: 144 weight + 144 operand registers per lane -- spills)
    const int CQ = C / 16, QP = C / 4 + 1, NCB = up_div(Co, 16);
    const int chunks = up_div((W + 1) * QP, 64); // the row's pixels and the right pad column
    // where it pays: maps wide enough to keep the compute waves busy and big enough to be a stream (MobileNetV2's 112x112; SNNHIP_DWPW_MARCH=1 forces it)
    const char* force = snnhip::option("SNNHIP_DWPW_MARCH");
    const bool forced = force && atoi(force) == 1;
    if (!forced && (W < 64 || static_cast<double>(N) * H * W * C * 4.0 < 64.0 * 1048576.0)) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(N) * H * W * std::max(C, Co) >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    DwPwFn fn = nullptr;
    const int CH = chunks <= 8 ? 8 : chunks <= 16 ? 16 : 20;
    const char* lwOpt = snnhip::option("SNNHIP_DWPW_LOADERS"); // 1 | 2 loader waves where both forms exist (16 pieces per row); experiments
    const int wantLoaders = lwOpt ? atoi(lwOpt) : 1;
    int ringRows = 0, computeWaves = 0;
    if (CQ == 1 && NCB == 1) fn = pick_dwpw<1, 1>(chunks, wantLoaders, &ringRows, &computeWaves);
    if (CQ == 1 && NCB == 2) fn = pick_dwpw<1, 2>(chunks, wantLoaders, &ringRows, &computeWaves);
    if (CQ == 2 && NCB == 1) fn = pick_dwpw<2, 1>(chunks, wantLoaders, &ringRows, &computeWaves);
    if (CQ == 2 && NCB == 2) fn = pick_dwpw<2, 2>(chunks, wantLoaders, &ringRows, &computeWaves);
    if (!fn) return SNNHIP_E_UNSUPPORTED;

    auto* plan = new DwPwPlan();
    plan->ctx = ctx;
    DwPwParams& p = plan->p;
    p = DwPwParams{};
    p.N = N; p.H = H; p.W = W; p.C = C; p.Co = Co;
    p.groups = up_div(W, 16);
    // ring slot: left pad pixel + CH pieces of 64 slots; the taps of the last group's unused lanes read up to pixel 16 groups + 1
    const int slotSlots = std::max(QP + CH * 64, (16 * p.groups + 2) * QP) + 4;
    p.slotFloats = round_up(slotSlots * 4, 64);
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    p.nseg = std::max(1, std::min(H / 8, up_div(cus, N))); // one block per CU (the ring is ~100 KB): at least as many blocks as CUs, runs of >= 8 rows
    p.rowsPerSeg = up_div(H, p.nseg);
    p.nseg = up_div(H, p.rowsPerSeg);
    p.acD = make_act_cfg(gd.act, gd.leaky);
    p.acP = make_act_cfg(gp.act, gp.leaky);
    plan->kernel = fn;
    plan->ldsBytes = static_cast<size_t>(ringRows) * p.slotFloats * sizeof(float);
    plan->grid = dim3(static_cast<unsigned>(N * p.nseg));
    if (plan->ldsBytes > 160 * 1024) {
        delete plan;
        return SNNHIP_E_UNSUPPORTED;
    }
    if (plan->ldsBytes > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes)) != hipSuccess) {
        set_error("dwpw_march: hipFuncSetAttribute(%zu) failed", plan->ldsBytes);
        delete plan;
        return SNNHIP_E_HIP;
    }
    std::vector<float> wd, eD, wA, eP;
    pack_dwpw(cd, cp, C, Co, wd, eD, wA, eP); // the lane-ordered parameter blobs (DwPwRegs::load)
    std::vector<float> zero(64, 0.0f);
    int rc = plan->upload(wd.data(), wd.size(), &plan->d_wd);
    if (rc == SNNHIP_OK) rc = plan->upload(wA.data(), wA.size(), &plan->d_wA);
    if (rc == SNNHIP_OK) rc = plan->upload(eD.data(), eD.size(), &plan->d_eD);
    if (rc == SNNHIP_OK) rc = plan->upload(eP.data(), eP.size(), &plan->d_eP);
    if (rc == SNNHIP_OK) rc = plan->upload(zero.data(), zero.size(), &plan->d_zero);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    memcpy(plan->inDims, dwPlan->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, pwPlan->outDims, sizeof(plan->outDims));
    plan->dtype = SNNHIP_F32;
    plan->flops = cd->flops + cp->flops;
    plan->bytes = cd->bytes + cp->bytes; // unfused accounting of the two layers it replaces
    plan->kernelBytes = 4.0 * (static_cast<double>(N) * H * W * (C + Co) + 9.0 * C + static_cast<double>(C) * Co);
    char buf[320];
    snprintf(buf, sizeof(buf), "dwpw_march_f32 [depthwise3x3 %d s1 + conv1x1 %d->%d] row-marching: %d loader wave(s) (LDS-DMA, %d rows in flight, %d x 1 KB per row) + %d compute waves "
             "(16 px each, mfma_f32_16x16x4), ring=%d rows, %d row run(s) per image lds=%zuB hbm_bytes=%.6g kernel=dwpw_march_kernel<%d,%d,%d>",
             C, C, Co, kWaves - computeWaves, ringRows - 3, CH, computeWaves, ringRows, p.nseg, plan->ldsBytes, plan->kernelBytes, CQ, NCB, CH);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}


namespace {
typedef void (*StemDwPwFn)(StemDwPwParams, const float*, const float*, const float4*, const float4*, const float*, const float4*, const float4*, const float*, float*);

struct StemDwPwPlan : snnhip_plan {
    StemDwPwParams sp;
    float *d_wS = nullptr, *d_eS = nullptr, *d_wd = nullptr, *d_wA = nullptr, *d_eD = nullptr, *d_eP = nullptr, *d_zero = nullptr;
    StemDwPwFn kernel = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "stem_dwpw_march: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        const DwPwParams& p = sp.d;
        SNNHIP_REQUIRE(x->n == p.N && x->h == sp.IH && x->w == sp.IW && x->c == 3 && x->dtype == SNNHIP_F32, "stem_dwpw_march: input dims %dx%dx%dx%d != plan %dx%dx%dx3 fp32", x->n,
                       x->h, x->w, x->c, p.N, sp.IH, sp.IW);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.H && out->w == p.W && out->c == p.Co && out->dtype == SNNHIP_F32, "stem_dwpw_march: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.H, p.W, p.Co);
        SNNHIP_LAUNCH(kernel, grid, dim3(64 * kWaves), ldsBytes, ctx->stream, sp, x->data, d_wS, reinterpret_cast<const float4*>(d_eS), reinterpret_cast<const float4*>(d_wd), d_wA,
                      reinterpret_cast<const float4*>(d_eD), reinterpret_cast<const float4*>(d_eP), d_zero, out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

// stemPlan: Conv2D 3x3 stride 2 "same" of a 3-channel image -> 32 channels; dwPlan / pwPlan as make_dwpw_march_plan (C = 32).  Borrowed plans.
int make_stem_dwpw_march_plan(snnhip_ctx* ctx, snnhip_plan* stemPlan, snnhip_plan* dwPlan, snnhip_plan* pwPlan, snnhip_plan** out) {
    if (const char* e = snnhip::option("SNNHIP_DWPW_MARCH"); e && atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    if (const char* e = snnhip::option("SNNHIP_STEM_MARCH"); e && atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    auto* cs = dynamic_cast<ConvPlanBase*>(stemPlan);
    auto* cd = dynamic_cast<ConvPlanBase*>(dwPlan);
    auto* cp = dynamic_cast<ConvPlanBase*>(pwPlan);
    if (!cs || !cd || !cp || cs->depthwise || !cd->depthwise || cp->depthwise) return SNNHIP_E_UNSUPPORTED;
    const ConvGeom &gs = cs->g, &gd = cd->g, &gp = cp->g;
    if (gs.dtype != SNNHIP_F32 || gs.kh != 3 || gs.kw != 3 || gs.sh != 2 || gs.sw != 2 || gs.IC != 3 || gs.OC != 32 || gs.padx != 1 || gs.pady != 1 || gs.preMode != 0 ||
        gs.addAct >= 0 || gs.normShift || (gs.padMode != SNNHIP_PAD_CONSTANT && gs.padMode != SNNHIP_PAD_NONE) || !act_is_simple(gs.act))
        return SNNHIP_E_UNSUPPORTED;
    if (gs.H != 2 * gs.OH || gs.W != 2 * gs.OW || gs.W % 4 != 0) return SNNHIP_E_UNSUPPORTED; // (even extents; whole 16-byte elements per image row)
    if (gd.dtype != SNNHIP_F32 || gp.dtype != SNNHIP_F32 || gd.kh != 3 || gd.kw != 3 || gd.sh != 1 || gd.sw != 1 || gd.padx != 1 || gd.pady != 1 || gd.preMode != 0 ||
        (gd.padMode != SNNHIP_PAD_CONSTANT && gd.padMode != SNNHIP_PAD_NONE) || gd.OH != gd.H || gd.OW != gd.W || gd.IC != 32 || gd.N != gs.N || gd.H != gs.OH || gd.W != gs.OW)
        return SNNHIP_E_UNSUPPORTED;
    if (gp.kh != 1 || gp.kw != 1 || gp.sh != 1 || gp.sw != 1 || gp.preMode != 0 || gp.addAct >= 0 || gp.normShift || gd.normShift || gp.IC != 32 || gp.N != gd.N || gp.H != gd.OH ||
        gp.W != gd.OW)
        return SNNHIP_E_UNSUPPORTED;
    if (!act_is_simple(gd.act) || !act_is_simple(gp.act)) return SNNHIP_E_UNSUPPORTED;
    const int C = 32, Co = gp.OC, W = gd.W, H = gd.H, N = gd.N, IW = gs.W, IH = gs.H;
    if (Co % 4 != 0 || Co > 16) return SNNHIP_E_UNSUPPORTED;
    const char* force = snnhip::option("SNNHIP_DWPW_MARCH");
    const bool forced = force && atoi(force) == 1;
    if (!forced && (W < 64 || static_cast<double>(N) * H * W * C * 4.0 < 64.0 * 1048576.0)) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(N) * IH * IW * 3 >= 2147483647.0 || static_cast<double>(N) * H * W * Co >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const int NCB = up_div(Co, 16), QP = 9;
    const int pieces = up_div(IW * 12, 1024);
    StemDwPwFn fn = nullptr;
    int PI = 0;
    if (NCB != 1) return SNNHIP_E_UNSUPPORTED; // (two output blocks: the lane's state no longer fits 256 registers)
    if (pieces <= 2) { PI = 2; fn = stem_dwpw_march_kernel<1, 2>; }
    else if (pieces == 3) { PI = 3; fn = stem_dwpw_march_kernel<1, 3>; }
    else if (pieces == 4) { PI = 4; fn = stem_dwpw_march_kernel<1, 4>; }
    if (!fn) return SNNHIP_E_UNSUPPORTED;
    const int PFI = std::min(8, 63 / (2 * PI)), IR = 2 * PFI + 4, SR = 4;

    auto* plan = new StemDwPwPlan();
    plan->ctx = ctx;
    StemDwPwParams& sp = plan->sp;
    sp = StemDwPwParams{};
    DwPwParams& p = sp.d;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Co = Co;
    p.groups = up_div(W, 16);
    p.slotFloats = round_up(((16 * p.groups + 2) * QP + 4) * 4, 64);
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    p.nseg = std::max(1, std::min(H / 8, up_div(cus, N)));
    p.rowsPerSeg = up_div(H, p.nseg);
    p.nseg = up_div(H, p.rowsPerSeg);
    p.acD = make_act_cfg(gd.act, gd.leaky);
    p.acP = make_act_cfg(gp.act, gp.leaky);
    sp.IH = IH; sp.IW = IW;
    sp.imgSlotFloats = round_up(4 + std::max(PI * 256, (32 * p.groups + 2) * 3), 64); // lead + the DMA pieces (and what the unused lanes of a ragged last group read)
    sp.stemOfs = IR * sp.imgSlotFloats;
    sp.acS = make_act_cfg(gs.act, gs.leaky);
    plan->kernel = fn;
    plan->ldsBytes = (static_cast<size_t>(sp.stemOfs) + static_cast<size_t>(SR) * p.slotFloats) * sizeof(float);
    plan->grid = dim3(static_cast<unsigned>(N * p.nseg));
    if (plan->ldsBytes > 160 * 1024) {
        delete plan;
        return SNNHIP_E_UNSUPPORTED;
    }
    if (plan->ldsBytes > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes)) != hipSuccess) {
        set_error("stem_dwpw_march: hipFuncSetAttribute(%zu) failed", plan->ldsBytes);
        delete plan;
        return SNNHIP_E_HIP;
    }
    // stem A operand of K step s7 and output block t2 at lane (m = oc, k = l / 16): W[16 t2 + m][channel c][tap], k = 4 s7 + l / 16 = 3 tap + c (k = 27: zero)
    std::vector<float> wS(static_cast<size_t>(2) * 7 * 64, 0.0f), eS(static_cast<size_t>(2) * 4 * 2 * 4, 0.0f), wd, eD, wA, eP;
    for (int t2 = 0; t2 < 2; ++t2) {
        for (int s7 = 0; s7 < 7; ++s7)
            for (int l = 0; l < 64; ++l) {
                const int k = 4 * s7 + (l >> 4), oc = 16 * t2 + (l & 15);
                if (k < 27) wS[(static_cast<size_t>(t2) * 7 + s7) * 64 + l] = cs->w_oihw[(static_cast<size_t>(oc) * 3 + k % 3) * 9 + k / 3];
            }
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i)
                fold_row(cs->epi4, 16 * t2 + 4 * j + i, gs.useBN, &eS[((static_cast<size_t>(t2) * 4 + j) * 2 + 0) * 4 + i], &eS[((static_cast<size_t>(t2) * 4 + j) * 2 + 1) * 4 + i]);
    }
    pack_dwpw(cd, cp, C, Co, wd, eD, wA, eP);
    std::vector<float> zero(64, 0.0f);
    int rc = plan->upload(wS.data(), wS.size(), &plan->d_wS);
    if (rc == SNNHIP_OK) rc = plan->upload(eS.data(), eS.size(), &plan->d_eS);
    if (rc == SNNHIP_OK) rc = plan->upload(wd.data(), wd.size(), &plan->d_wd);
    if (rc == SNNHIP_OK) rc = plan->upload(wA.data(), wA.size(), &plan->d_wA);
    if (rc == SNNHIP_OK) rc = plan->upload(eD.data(), eD.size(), &plan->d_eD);
    if (rc == SNNHIP_OK) rc = plan->upload(eP.data(), eP.size(), &plan->d_eP);
    if (rc == SNNHIP_OK) rc = plan->upload(zero.data(), zero.size(), &plan->d_zero);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    memcpy(plan->inDims, stemPlan->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, pwPlan->outDims, sizeof(plan->outDims));
    plan->dtype = SNNHIP_F32;
    plan->flops = cs->flops + cd->flops + cp->flops;
    plan->bytes = cs->bytes + cd->bytes + cp->bytes; // unfused accounting of the three layers it replaces
    plan->kernelBytes = 4.0 * (static_cast<double>(N) * IH * IW * 3 + static_cast<double>(N) * H * W * Co + 27.0 * 32 + 9.0 * C + static_cast<double>(C) * Co);
    char buf[400];
    snprintf(buf, sizeof(buf), "stem_dwpw_march_f32 [stem conv3x3 s2 3->32 + depthwise3x3 32 s1 + conv1x1 32->%d] row-marching: 1 loader wave (image rows by LDS-DMA, %d x 1 KB per row, "
             "%d iterations in flight) + %d compute waves (stem: 7 K steps of mfma_f32_16x16x4 per 16 px and output block, written to the stem-row ring), rings=%d image + %d stem rows, "
             "%d row run(s) per image lds=%zuB hbm_bytes=%.6g kernel=stem_dwpw_march_kernel<%d,%d>",
             Co, PI, PFI, kWaves - 1, IR, SR, p.nseg, plan->ldsBytes, plan->kernelBytes, NCB, PI);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
