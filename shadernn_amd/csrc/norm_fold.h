// norm_fold.h -- chain rule F, device side of "the convolution folds its own tile statistics": every block of a convolution kernel that
// leaves {mean, M2} records of its output tile (conv2d_wide_f16.hip) ends with tile_stats_finish(); the LAST block of an image (and output-channel
// block) to get there merges that image's records in a fixed order (deterministic, whichever block happens to be last) and writes the
// InstanceNorm's per-(image, channel) multiplier and shift -- the norm behind the convolution needs no statistics sweep and no fold launch.
// Counterpart in the reference: vk_instancenorm.comp:53-160 computes mean and variance with two loops over the plane per invocation.
#pragma once
#include <hip/hip_runtime.h>

namespace snnhip {

struct NormFoldArgs {
    unsigned* counter; // [N][gridDim.y] blocks of the image that have written their record; null = the kernel does not fold
    const float* gamma;
    const float* beta;
    float* shift; // [N][OC]  beta - mean * mul
    float* mul;   // [N][OC]  gamma / sqrt(var + eps)
    float eps;
};

// The records cross from one block to another INSIDE a launch, and the chip's eight XCDs have an L2 each.  __threadfence() (an agent-scope
// release / acquire) is a write-back of the XCD's L2 on gfx942 / gfx950 -- every block flushing the output tile it has just stored doubled the
// convolution's time (measured).  Agent-scope ATOMIC stores and loads (sc1: written through to / read from the coherence point) carry the few
// floats that must travel instead; the writer waits for its stores' acknowledgements (vmcnt) before the block is counted.
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// parallel-variance update: (na, ma, M2a) += (nb, mb, M2b); nb = 0 changes nothing
__device__ __forceinline__ void stat_merge3(float& na, float& ma, float& M2, float nb, float mb, float qb) {
    const float n2 = na + nb, d = mb - ma, r = nb * __builtin_amdgcn_rcpf(fmaxf(n2, 1.0f));
    ma = fmaf(d, r, ma);
    M2 += qb + d * d * na * r;
    na = n2;
}

__device__ __forceinline__ float2 ld_agent2(const float* p) { // 8 bytes (two adjacent channels) per load
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float(static_cast<unsigned>(u)), __uint_as_float(static_cast<unsigned>(u >> 32)));
}

constexpr int kFoldScratchFloats = 3 * 512 + 1; // LDS the fold needs

// Called by all 256 threads of a block after the threads that wrote the block's record (part[(n*tiles + tile)*2*OC + ...], with st_agent) have done so.
// scratch: kFoldScratchFloats floats of LDS nobody else uses any more.  BN = output channels per block (even, 512 % BN == 0), ocb = the block's
// first channel (even: the 8-byte loads).  The fold is the tail of the launch (the last image's last block runs it when everything else is
// done), so it is built for latency: thread = (channel pair, 1 of 512 / BN parts), 16 records (64 floats) in flight per thread -- a 720p map of
// 286 tiles x 128 channels takes 5 round trips to the coherence point instead of the 143 of a one-record-at-a-time walk.
// The fold proper: called by all 256 threads of the block that saw an image's last record counted (after ONE agent-scope acquire, below).
// counterBlocks / counterIndex stand for gridDim.y / blockIdx.y of the tile kernels (a persistent kernel has neither).
template <int BN>
__device__ __forceinline__ void tile_stats_fold(const NormFoldArgs& f, const float* part, float* scratch, int n, int tilesX, int tilesY, int TH, int TW, int OH, int OW, int OC,
                                                int ocb);

template <int BN>
__device__ __forceinline__ void tile_stats_finish(const NormFoldArgs& f, const float* part, float* scratch, int n, int tilesX, int tilesY, int TH, int TW, int OH, int OW,
                                                  int OC, int ocb) {
    const int tid = threadIdx.x, tiles = tilesX * tilesY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's record stores (if it wrote any) are acknowledged before the block is counted
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = f.counter + n * gridDim.y + blockIdx.y;
        const unsigned prev = atomicAdd(cnt, 1u);
        const bool last = prev + 1u == static_cast<unsigned>(tiles);
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // nobody else touches it in this launch: ready for the next one (a replayed hipGraph)
        scratch[3 * 512] = last ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (scratch[3 * 512] == 0.0f) return;
    tile_stats_fold<BN>(f, part, scratch, n, tilesX, tilesY, TH, TW, OH, OW, OC, ocb);
}

template <int BN>
__device__ __forceinline__ void tile_stats_fold(const NormFoldArgs& f, const float* part, float* scratch, int n, int tilesX, int tilesY, int TH, int TW, int OH, int OW, int OC,
                                                int ocb) {
    const int tid = threadIdx.x, tiles = tilesX * tilesY;
    // The hand-off is relaxed on purpose (a release in EVERY block would be the L2 write-back described at the top).  What makes the records
    // visible is that they are written through (sc1 stores, acknowledged before the block is counted) and read with sc1 loads; formally that still
    // leaves the last block without an acquire.  It gets one here -- ONE agent-scope acquire per image (an invalidate, no write-back): nothing this
    // block has cached from an earlier launch of the same plan (the records live in the same buffer launch after launch) can satisfy the loads below.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int HP = BN / 2, PARTS = 256 / HP, B = 16;
    const int cp = tid % HP, q = tid / HP;
    const float* pn = part + static_cast<size_t>(n) * tiles * 2 * OC + ocb + 2 * cp;
    float na = 0.0f, ma0 = 0.0f, ma1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
    for (int t0 = q; t0 < tiles; t0 += B * PARTS) { // thread (cp, q) takes tiles q, q + PARTS, ... in order
        float nb[B];
        float2 mb[B], qb[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int t = t0 + j * PARTS;
            nb[j] = 0.0f;
            mb[j] = qb[j] = make_float2(0.0f, 0.0f);
            if (t < tiles) {
                const int ty = t / tilesX, tx = t - ty * tilesX;
                nb[j] = static_cast<float>(max(0, min(TH, OH - ty * TH)) * max(0, min(TW, OW - tx * TW))); // (a record row / column entirely outside the image counts nothing)
                mb[j] = ld_agent2(pn + static_cast<size_t>(t) * 2 * OC);
                qb[j] = ld_agent2(pn + static_cast<size_t>(t) * 2 * OC + OC);
            }
        }
#pragma unroll
        for (int j = 0; j < B; ++j) { // the two channels share the counts
            const float n2 = na + nb[j], r = nb[j] * __builtin_amdgcn_rcpf(fmaxf(n2, 1.0f)), w = na * r;
            const float d0 = mb[j].x - ma0, d1 = mb[j].y - ma1;
            ma0 = fmaf(d0, r, ma0);
            ma1 = fmaf(d1, r, ma1);
            q0 += qb[j].x + d0 * d0 * w;
            q1 += qb[j].y + d1 * d1 * w;
            na = n2;
        }
    }
    scratch[q * BN + 2 * cp] = na;
    scratch[q * BN + 2 * cp + 1] = na;
    scratch[512 + q * BN + 2 * cp] = ma0;
    scratch[512 + q * BN + 2 * cp + 1] = ma1;
    scratch[1024 + q * BN + 2 * cp] = q0;
    scratch[1024 + q * BN + 2 * cp + 1] = q1;
    __syncthreads();
    if (tid < BN) {
        float nn = 0.0f, mm = 0.0f, M2 = 0.0f;
#pragma unroll
        for (int j = 0; j < PARTS; ++j) stat_merge3(nn, mm, M2, scratch[j * BN + tid], scratch[512 + j * BN + tid], scratch[1024 + j * BN + tid]);
        const float var = fmaxf(M2 / nn, 0.0f);
        const float mu = f.gamma[ocb + tid] / sqrtf(var + f.eps);
        f.mul[n * OC + ocb + tid] = mu;
        f.shift[n * OC + ocb + tid] = f.beta[ocb + tid] - mm * mu; // y = x * mul + shift (instancenorm_fold_kernel's form: every consumer evaluates this fma)
    }
}

} // namespace snnhip
