// conv2d_wino.hip -- fp32 3x3 stride-1 convolution as Winograd F(2x2,3x3) on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32), for the
// GEMM-shaped 3x3 layers of the reference's Conv2D operator: the ResNet-18 body (BASELINE configs[2]), U-Net, YOLO.
//
// Same operator contract as conv2d_mfma.hip / conv2d_generic.hip (shadertemplate_vk_conv2d.comp:148-347: zero padding by clipped
// taps, bias -> BN -> activation epilogue, optional fused residual Add of chain rule E); only the evaluation differs:
//     Y = At [ sum_ic (G g Gt) .* (Bt d B) ] A          per 2x2 output tile, 16 "positions" per tile
// which needs 16 multiplies per 2x2 tile, channel pair instead of 36 -- the matrix pipe does 2.25x fewer flops than the direct implicit
// GEMM.  fp32 MFMA and the fp32 VALU share the FMA lanes on this chip (tools/ubench_issue.hip), so the transforms are kept to the
// minimum: activations are only added / subtracted (Bt, At have entries 0, +-1), G g Gt is computed on the host in double.
//
// Work decomposition (default: one block = 256 threads = 4 waves, two blocks per CU; SNNHIP_WINO_OPB=2: 512 threads / 64 channels, one per CU):
//   block tile = 64 Winograd tiles (TB images x TTH x TTW tiles = TB x 2TTH x 2TTW output pixels) x 32 output channels
//   wave tg: tile group = 16 consecutive tiles (the N of the MFMA) x 2 blocks of 16 channels (the M)
//   K loop over 8-channel chunks, double-buffered in LDS, one barrier per chunk:
//     U slab   [16 positions][2 channel pairs][lane = (k, m)] float4 = the MFMA A operands of both K steps and both channel blocks of a wave,
//              pre-packed on the host in exactly this order (a chunk's slab is 32 KB of contiguous global memory; a lane's read is
//              base + 16*lane: conflict-free ds_read_b128)
//     input    4 channel-pair planes [image][row][col] float2 of the (2TTH+2) x (2TTW+2) halo tile; plane pitch = 0 mod 64 floats and row
//              pitch = TTW mod 16 pixels make the ds_read_b128 of two adjacent patch pixels conflict-free across the wave's 16 tiles x 4
//              pairs (lane groups {0-3,12-15,20-27} ...: MI355X_MICROARCH.md, LDS)
//     per chunk and wave: 8 ds_read_b128 (its 4x4 patch: lane = (tile n, channel pair k)), Bt d B on the VALU (64 adds), then
//     2 K steps x 16 positions x 2 channel blocks = 64 MFMAs whose B operand is the transform result straight from registers (the
//     transformed tile never goes through LDS) and whose A operands are one ds_read_b128 per position
//   accumulators: 16 positions x 2 blocks x 4 = 128 registers; At M A on the VALU once per tile, epilogue, 16-byte channel-contiguous stores
//   split-K over blockIdx.z (raw partial sums to a workspace + the reduce/epilogue pass of conv2d_mfma.hip) when a layer has too few
//   block tiles to fill 256 CUs (ResNet's 14x14 / 7x7 stages).
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

#ifndef SNNHIP_WINO_ABL
#define SNNHIP_WINO_ABL 0 // ablation builds only (tools/ablate_wino.sh; results are wrong by construction): 1 no barrier in the chunk loop,
#endif                    // 2 no LDS stores, 4 no global loads, 8 no input transform, 16 no U reads, 32 no patch reads, 64 no U DMA, 128 no activation loads

namespace snnhip {

// conv2d_mfma.hip: the split-K reduce / epilogue pass
int launch_splitk_reduce(snnhip_ctx* ctx, int OC, int splitK, int useBN, const ActCfg& ac, const float* ws, const float4* e4, snnhip_tensor* out,
                         const snnhip_tensor* res, const ActCfg& ac2);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoParams {
    int N, H, W, IC, OC, OH, OW, padx, pady;
    int TBs, TTHs, TTWs;   // log2: images, tile rows, tile columns of a block tile (TB * TTH * TTW = 64 tiles)
    int tilesX, tilesY;    // block tiles along x / y
    int inH, inW;          // staged halo tile per image: 2*TTH + 2, 2*TTW + 2 pixels
    int rowPitch;          // pixels per staged row
    int planeStride;       // floats between the channel-pair planes
    int total;             // float4 elements staged per chunk (TB * inH * inW * 2)
    int bufFloats;         // floats per LDS buffer (U slab + 4 planes)
    int nChunks, splitK, chunksPerSplit;
    int OCblocks;          // ceil(OC / (32 * OPB))
    int useBN;
    const float* res;      // fused residual Add (chain rule E), or nullptr
    ActCfg ac2;
};

// OPB = output-channel pairs (2 x 16 channels) per block: 2 -> 512 threads, 64 channels, one block per CU; 1 -> 256 threads, 32 channels,
// two independent blocks per CU (their barriers and LDS-read bursts drift apart, the pair of waves on a SIMD is no longer in lock step)
// KS = 2 (round 6; OPB = 1 only): the block's channel chunks are split over TWO groups of four waves -- each group stages its own chunks into its own
// pair of LDS buffers and runs the K loop below on them, barrier for barrier --, then the groups exchange half of their sums through LDS (group g keeps
// output-channel block g: it hands the partner the other block's sixteen positions and adds the partner's to its own -- p0 + p1 in both groups, the order
// of the reduce pass), and every wave runs the epilogue of ONE channel block.  What a 14x14 stage needed split-K over blockIdx.z + a reduce launch for
// (partial sums through HBM, 6 us per launch) stays inside the block.
template <bool SIMPLE, int OPB, int KS = 1>
__global__ __launch_bounds__(256 * OPB * KS, KS > 1 ? 1 : 3 - OPB) void conv2d_wino_kernel(WinoParams p, ActCfg ac, const float* __restrict__ x, const float4* __restrict__ ug,
                                                                        const float4* __restrict__ epi, float* __restrict__ y, float* __restrict__ ws) {
    static_assert(KS == 1 || (KS == 2 && OPB == 1), "in-block K split: two groups of four waves");
    constexpr int NT = 256 * OPB;            // threads of a group
    constexpr int kUFloats = 4096 * OPB;     // U slab of one chunk: 16 positions x OPB pairs x 64 lanes x float4 (16 / 32 KB)
    constexpr int UQ = 1024 * OPB;           // ... in float4
    constexpr int R = 4 / OPB;               // staged activation quads per thread (halo tile of up to 1024 quads)
    extern __shared__ __attribute__((aligned(16))) float smemAll[];
    const int lane = threadIdx.x & 63;
    const int ks = KS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0; // K group of this wave
    const int tid = KS > 1 ? (threadIdx.x & 255) : threadIdx.x, wave = tid >> 6;     // thread / wave within the group
    float* const smem = smemAll + ks * 2 * p.bufFloats;                              // the group's two buffers
    const int tg = wave & 3, op = wave >> 2;
    const int n16 = lane & 15, k = lane >> 4;

    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, tb = mt / (p.tilesX * p.tilesY);
    const int TTW = 1 << p.TTWs, TTH = 1 << p.TTHs;
    const int ox0 = tx * 2 * TTW, oy0 = ty * 2 * TTH, b0 = tb << p.TBs;
    const int ix0 = ox0 - p.padx, iy0 = oy0 - p.pady;

    // ---- staging descriptors: element e = tid + NT r -> (pixel of the halo tile, channel quad q = e & 1)
    const int q = tid & 1;
    int gofs[4], lofs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gofs[r] = lofs[r] = -1;
        if (r >= R) continue;
        const int e = tid + NT * r;
        gofs[r] = -1;
        lofs[r] = -1;
        if (e < p.total) {
            const int pix = e >> 1;
            const int c = pix % p.inW, t2 = pix / p.inW;
            const int rr = t2 % p.inH, b = t2 / p.inH;
            const int sy = iy0 + rr, sx = ix0 + c, n = b0 + b;
            lofs[r] = kUFloats + 2 * q * p.planeStride + ((b * p.inH + rr) * p.rowPitch + c) * 2;
            if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W && n < p.N) gofs[r] = ((n * p.H + sy) * p.W + sx) * p.IC + q * 4;
        }
    }
    int chunk0 = blockIdx.z * p.chunksPerSplit, chunk1 = min(p.nChunks, chunk0 + p.chunksPerSplit);
    if (KS > 1) { // (the planner only splits even chunk counts: both groups run the same number of iterations, their barriers pair up)
        const int half = (chunk1 - chunk0) >> 1;
        chunk0 += ks * half;
        chunk1 = chunk0 + half;
    }
    // U slab of (oc block, chunk): UQ float4, thread t copies float4 t + NT j, j < 4
    const float4* uptr = ug + (static_cast<size_t>(blockIdx.y) * p.nChunks + chunk0) * UQ + tid;

    // Staging.  U slab: LDS-DMA (global_load_lds_dwordx4: a wave copies 64 consecutive float4 = 1 KiB per instruction straight into the slab,
    // no staging registers and -- what the ablation builds showed to matter -- no VGPR -> LDS store traffic: the four ds_write_b128 per thread
    // and chunk it replaces cost the kernel 20 %).  Activations: R quads per thread through registers (zero fill outside the image, split into
    // the channel-pair planes); plain scalars: arrays captured by reference went to scratch.
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sxa = zero4, sxb = zero4, sxc = zero4, sxd = zero4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
#define WINO_STAGE_LOAD(chunk_, buf_)                                                                                             \
    do {                                                                                                                          \
        const float4* u_ = uptr + static_cast<size_t>((chunk_) - chunk0) * UQ;                                                    \
        float* ub_ = (buf_) + 4 * (wave * 64);                                                                                    \
        if (!(SNNHIP_WINO_ABL & 64)) {                                                                                            \
            lds_dma16(u_, ub_);                                                                                                   \
            lds_dma16(u_ + NT, ub_ + 4 * NT);                                                                                     \
            lds_dma16(u_ + 2 * NT, ub_ + 8 * NT);                                                                                 \
            lds_dma16(u_ + 3 * NT, ub_ + 12 * NT);                                                                                \
        }                                                                                                                         \
        sxa = sxb = sxc = sxd = zero4;                                                                                            \
        if (SNNHIP_WINO_ABL & 128) break;                                                                                         \
        if (gofs[0] >= 0) sxa = *reinterpret_cast<const float4*>(x + gofs[0] + (chunk_) * 8);                                     \
        if (gofs[1] >= 0) sxb = *reinterpret_cast<const float4*>(x + gofs[1] + (chunk_) * 8);                                     \
        if (R > 2 && gofs[2] >= 0) sxc = *reinterpret_cast<const float4*>(x + gofs[2] + (chunk_) * 8);                            \
        if (R > 2 && gofs[3] >= 0) sxd = *reinterpret_cast<const float4*>(x + gofs[3] + (chunk_) * 8);                            \
    } while (0)
#define WINO_STORE_X(r_, v_)                                                                                     \
    if (lofs[r_] >= 0) {                                                                                         \
        *reinterpret_cast<float2*>(b_ + lofs[r_]) = make_float2(v_.x, v_.y);                                     \
        *reinterpret_cast<float2*>(b_ + lofs[r_] + p.planeStride) = make_float2(v_.z, v_.w);                     \
    }
#define WINO_STAGE_STORE(buf_)                                                                                   \
    do {                                                                                                         \
        float* b_ = (buf_);                                                                                      \
        WINO_STORE_X(0, sxa)                                                                                     \
        WINO_STORE_X(1, sxb)                                                                                     \
        if (R > 2) {                                                                                             \
            WINO_STORE_X(2, sxc)                                                                                 \
            WINO_STORE_X(3, sxd)                                                                                 \
        }                                                                                                        \
    } while (0)

    // ---- this lane's tile: t = 16 tg + n16 -> (image b, tile row, tile column) of the block tile
    const int t = 16 * tg + n16;
    const int ttx = t & (TTW - 1), tty = (t >> p.TTWs) & (TTH - 1), tbi = t >> (p.TTWs + p.TTHs);
    const int patch = kUFloats + k * p.planeStride + ((tbi * p.inH + 2 * tty) * p.rowPitch + 2 * ttx) * 2; // + (dy*rowPitch + dx)*2
    const int uoff = op * 256 + lane * 4;                                                                  // + pos * 256 * OPB: float4 {U[2op].s0, .s1, U[2op+1].s0, .s1}

    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o) acc[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};

    WINO_STAGE_LOAD(chunk0, smem);
    WINO_STAGE_STORE(smem);
    lds_dma_wait();
    __syncthreads();

    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        const float* buf = smem + ((chunk - chunk0) & 1) * p.bufFloats;
        const bool more = chunk + 1 < chunk1;
        float* nbuf = smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats; // the buffer nobody reads during this chunk
        if (more && !(SNNHIP_WINO_ABL & 4)) WINO_STAGE_LOAD(chunk + 1, nbuf);

        // Issue order is pinned with sched_barrier: the patch and the first eight U operands are requested up front, the rest while the
        // first MFMAs run, and the next chunk goes to the OTHER LDS buffer in the middle of the MFMA stream (its global loads were issued a
        // thousand cycles earlier), so that the end of the chunk is only the barrier.  Left to itself the scheduler sinks every ds_read next
        // to its use (fewest registers) and the wave eats one LDS latency per position: 31 % matrix-pipe utilisation (PMC), 48 % after.
        float4 t4[4][2], u[16];
#define WINO_LDU(i) u[i] = (SNNHIP_WINO_ABL & 16) ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(buf + uoff + (i) * (256 * OPB))
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) // one ds_read_b128 = two adjacent pixels of the pair plane
                t4[dy][h2] = (SNNHIP_WINO_ABL & 32) ? make_float4(1.f + dy, 2.f, 3.f + h2, 4.f) : *reinterpret_cast<const float4*>(buf + patch + (dy * p.rowPitch + 2 * h2) * 2);
        WINO_LDU(0); WINO_LDU(1); WINO_LDU(2); WINO_LDU(3); WINO_LDU(4); WINO_LDU(5); WINO_LDU(6); WINO_LDU(7);
        __builtin_amdgcn_sched_barrier(0);
        // the 4x4 patch of (tile, channel pair), both channels of the pair
        float2 d[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                d[dy][2 * h2] = make_float2(t4[dy][h2].x, t4[dy][h2].y);
                d[dy][2 * h2 + 1] = make_float2(t4[dy][h2].z, t4[dy][h2].w);
            }
        // V = Bt d B, Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float2 v[4][4];
#pragma unroll
        for (int j = 0; j < 4 && !(SNNHIP_WINO_ABL & 8); ++j) {
            const float2 a0 = d[0][j], a1 = d[1][j], a2 = d[2][j], a3 = d[3][j];
            d[0][j] = make_float2(a0.x - a2.x, a0.y - a2.y);
            d[1][j] = make_float2(a1.x + a2.x, a1.y + a2.y);
            d[2][j] = make_float2(a2.x - a1.x, a2.y - a1.y);
            d[3][j] = make_float2(a1.x - a3.x, a1.y - a3.y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 a0 = d[i][0], a1 = d[i][1], a2 = d[i][2], a3 = d[i][3];
            if (SNNHIP_WINO_ABL & 8) {
                v[i][0] = a0; v[i][1] = a1; v[i][2] = a2; v[i][3] = a3;
                continue;
            }
            v[i][0] = make_float2(a0.x - a2.x, a0.y - a2.y);
            v[i][1] = make_float2(a1.x + a2.x, a1.y + a2.y);
            v[i][2] = make_float2(a2.x - a1.x, a2.y - a1.y);
            v[i][3] = make_float2(a1.x - a3.x, a1.y - a3.y);
        }
        // 16 positions x 2 channel blocks x 2 K steps: M[pos][oc][tile] += U[pos][oc][ic] V[pos][ic][tile]
#define WINO_STEP(pos)                                                                                       \
    do {                                                                                                     \
        const float2 vv_ = v[(pos) >> 2][(pos) & 3];                                                         \
        acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pos].x, vv_.x, acc[pos][0], 0, 0, 0);           \
        acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pos].z, vv_.x, acc[pos][1], 0, 0, 0);           \
        acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pos].y, vv_.y, acc[pos][0], 0, 0, 0);           \
        acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pos].w, vv_.y, acc[pos][1], 0, 0, 0);           \
    } while (0)
        WINO_STEP(0); WINO_STEP(1); WINO_STEP(2); WINO_STEP(3);
        __builtin_amdgcn_sched_barrier(0);
        WINO_LDU(8); WINO_LDU(9); WINO_LDU(10); WINO_LDU(11);
        __builtin_amdgcn_sched_barrier(0);
        WINO_STEP(4); WINO_STEP(5); WINO_STEP(6); WINO_STEP(7);
        __builtin_amdgcn_sched_barrier(0);
        WINO_LDU(12); WINO_LDU(13); WINO_LDU(14); WINO_LDU(15);
        if (more && !(SNNHIP_WINO_ABL & 2)) WINO_STAGE_STORE(nbuf);
        __builtin_amdgcn_sched_barrier(0);
        WINO_STEP(8); WINO_STEP(9); WINO_STEP(10); WINO_STEP(11); WINO_STEP(12); WINO_STEP(13); WINO_STEP(14); WINO_STEP(15);
#undef WINO_STEP
#undef WINO_LDU
        lds_dma_wait(); // (free here: the activation loads issued after the DMA were waited for by the stage store above)
        if (!(SNNHIP_WINO_ABL & 1)) __syncthreads();
    }

    // ---- KS = 2: the two K groups exchange one channel block's sums (region of wave w: [pos][lane] float4 = 16 KB, 128 KB in all; the staging buffers are
    // free: every wave is past the loop's last barrier)
    if (KS > 1) {
        f32x4* const xb = reinterpret_cast<f32x4*>(smemAll);
        f32x4* const mine = xb + ((ks * 4 + wave) * 16) * 64 + lane;
        const f32x4* const theirs = xb + (((1 - ks) * 4 + wave) * 16) * 64 + lane;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[i * 64] = acc[i][1];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[i * 64] = acc[i][0];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i][0] += theirs[i * 64]; // p0 + p1
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i][1] = theirs[i * 64] + acc[i][1]; // p0 + p1 as well
        }
    }

    // ---- Y = At M A, At = [1 1 1 0; 0 1 -1 -1]; lane holds output channels 4k..4k+3 of each 16-channel block for its tile
    const int oy = oy0 + 2 * tty, ox = ox0 + 2 * ttx, n = b0 + tbi;
    const bool addSimple = act_is_simple_dev(p.ac2.act);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (KS > 1 && o != ks) continue; // (wave-uniform: group g finishes channel block g)
        const int oc = blockIdx.y * (32 * OPB) + (2 * op + o) * 16 + 4 * k;
        if (oc >= p.OC) continue;
        // the epilogue rows of the lane's four channels and the residual of its 2 x 2 output pixels are requested FIRST, the inverse transform runs in
        // the shadow of those loads (requested where they are used -- a table load, then per pixel a residual load, each waited out -- they were eight
        // exposed L2 round trips per wave: conv1x1_stream's phase trace, DESIGN.md 5.1-8, found the same pattern)
        float4 e4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) e4[c] = epi[oc + c];
        size_t idx4[2][2];
        bool ok4[2][2];
        float4 rv4[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                ok4[r][cx] = !(n >= p.N || oy + r >= p.OH || ox + cx >= p.OW);
                idx4[r][cx] = ((static_cast<size_t>(n) * p.OH + oy + r) * p.OW + ox + cx) * p.OC + oc;
                rv4[r][cx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.res && p.splitK == 1 && ok4[r][cx]) rv4[r][cx] = *reinterpret_cast<const float4*>(p.res + idx4[r][cx]);
            }
        f32x4 yv[2][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = acc[0 * 4 + j][o][c] + acc[1 * 4 + j][o][c] + acc[2 * 4 + j][o][c];
                t1[j] = acc[1 * 4 + j][o][c] - acc[2 * 4 + j][o][c] - acc[3 * 4 + j][o][c];
            }
            yv[0][0][c] = t0[0] + t0[1] + t0[2];
            yv[0][1][c] = t0[1] - t0[2] - t0[3];
            yv[1][0][c] = t1[0] + t1[1] + t1[2];
            yv[1][1][c] = t1[1] - t1[2] - t1[3];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                if (!ok4[r][cx]) continue;
                const size_t idx = idx4[r][cx];
                float o4[4];
                if (p.splitK > 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o4[c] = yv[r][cx][c];
                    *reinterpret_cast<float4*>(ws + static_cast<size_t>(blockIdx.z) * (static_cast<size_t>(p.N) * p.OH * p.OW * p.OC) + idx) =
                        make_float4(o4[0], o4[1], o4[2], o4[3]);
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float vv = epi_affine(yv[r][cx][c], e4[c], p.useBN);
                    o4[c] = apply_act<SIMPLE>(ac, vv, 0.0f);
                }
                if (p.res) {
                    const float4 rv = rv4[r][cx];
                    o4[0] = add_act(p.ac2, addSimple, o4[0] + rv.x);
                    o4[1] = add_act(p.ac2, addSimple, o4[1] + rv.y);
                    o4[2] = add_act(p.ac2, addSimple, o4[2] + rv.z);
                    o4[3] = add_act(p.ac2, addSimple, o4[3] + rv.w);
                }
                *reinterpret_cast<float4*>(y + idx) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
    }
}

struct WinoConvPlan : ConvPlanBase {
    WinoParams p;
    ActCfg ac;
    float* d_u = nullptr;
    float* d_epi = nullptr;
    float* d_ws = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    bool fusedAdd = false;
    bool simple = true;
    int opb = 1;
    int ks = 1; // K groups per block (2: eight waves, the channel chunks split inside the block)

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == (fusedAdd ? 2 : 1), "conv2d: expects %d input(s), got %d", fusedAdd ? 2 : 1, nIn);
        const snnhip_tensor* x = in[0];
        WinoParams q = p; // per-launch copy: the residual pointer travels in the kernel argument block
        q.res = nullptr;
        if (fusedAdd) {
            const snnhip_tensor* r = in[1];
            SNNHIP_REQUIRE(r->n == p.N && r->h == p.OH && r->w == p.OW && r->c == p.OC && r->dtype == dtype,
                           "conv2d+add: residual %dx%dx%dx%d (dtype %d) does not match the output %dx%dx%dx%d", r->n, r->h, r->w, r->c, r->dtype, p.N, p.OH,
                           p.OW, p.OC);
            q.res = p.splitK > 1 ? nullptr : r->data; // split-K: the reduce pass adds it
        }
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w,
                       x->c, p.N, p.H, p.W, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        const float4* u4 = reinterpret_cast<const float4*>(d_u);
        const float4* e4 = reinterpret_cast<const float4*>(d_epi);
        if (ks == 2) {
            if (simple) SNNHIP_LAUNCH((conv2d_wino_kernel<true, 1, 2>), grid, dim3(512), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
            else SNNHIP_LAUNCH((conv2d_wino_kernel<false, 1, 2>), grid, dim3(512), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
        } else if (opb == 2) {
            if (simple) SNNHIP_LAUNCH((conv2d_wino_kernel<true, 2>), grid, dim3(512), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
            else SNNHIP_LAUNCH((conv2d_wino_kernel<false, 2>), grid, dim3(512), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
        } else {
            if (simple) SNNHIP_LAUNCH((conv2d_wino_kernel<true, 1>), grid, dim3(256), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
            else SNNHIP_LAUNCH((conv2d_wino_kernel<false, 1>), grid, dim3(256), ldsBytes, ctx->stream, q, ac, x->data, u4, e4, out->data, d_ws);
        }
        SNNHIP_CHECK_HIP(hipGetLastError());
        if (p.splitK > 1) return launch_splitk_reduce(ctx, p.OC, p.splitK, p.useBN, ac, d_ws, e4, out, fusedAdd ? in[1] : nullptr, p.ac2);
        return SNNHIP_OK;
    }
};

} // namespace

int make_conv2d_wino_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    // eligibility: fp32, 3x3, stride 1, zero padding (or none), no fused Pad in front, whole 8-channel chunks and 16-channel output blocks
    if (g.dtype != SNNHIP_F32 || g.kh != 3 || g.kw != 3 || g.sh != 1 || g.sw != 1 || g.preMode != 0) return SNNHIP_E_UNSUPPORTED;
    if (g.padMode != SNNHIP_PAD_CONSTANT && g.padMode != SNNHIP_PAD_NONE) return SNNHIP_E_UNSUPPORTED;
    if (g.IC % 8 != 0 || g.OC % 16 != 0 || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED;
    const double inCount = static_cast<double>(g.N) * g.H * g.W * g.IC, outCount = static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    if (inCount >= 2147483647.0 || outCount >= 2147483647.0) return SNNHIP_E_UNSUPPORTED; // 32-bit element offsets in the staging path

    // block tile: TB x TTH x TTW = 64 Winograd tiles, minimising padded tiles; a wave's 16 consecutive tiles must not straddle images
    // (TTH * TTW >= 16 and TTH >= 16 / TTW when TTW < 16: the bank analysis of the patch reads assumes whole tile rows of one image)
    const int tilesW = up_div(g.OW, 2), tilesH = up_div(g.OH, 2);
    // OPB = 1 (default): 256-thread blocks of 32 output channels, two per CU; SNNHIP_WINO_OPB=2: 512-thread blocks of 64 channels, one per CU
    int opb = 1;
    if (const char* e = snnhip::option("SNNHIP_WINO_OPB")) opb = atoi(e) == 2 ? 2 : 1;
    const int ocPerBlock = 32 * opb, kUFloats = 4096 * opb;
    int best[3] = {0, 2, 4};
    double bestCost = 1e300;
    for (int ws = 2; ws <= 5; ++ws)
        for (int hs = 0; hs + ws <= 6; ++hs) {
            const int TTW = 1 << ws, TTH = 1 << hs, TB = 64 / (TTW * TTH);
            if (TTW < 16 && TTH < 16 / TTW) continue;
            if (TB * (2 * TTH + 2) * (2 * TTW + 2) * 2 > 1024) continue; // two staged float4 per thread
            const double cost = static_cast<double>(up_div(g.N, TB)) * up_div(tilesH, TTH) * up_div(tilesW, TTW) * (1.0 + 0.02 * (2.0 / TTH + 2.0 / TTW));
            if (cost < bestCost) {
                bestCost = cost;
                best[0] = 6 - ws - hs;
                best[1] = hs;
                best[2] = ws;
            }
        }
    WinoParams p = {};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW; p.padx = g.padx; p.pady = g.pady;
    p.TBs = best[0]; p.TTHs = best[1]; p.TTWs = best[2];
    const int TB = 1 << p.TBs, TTH = 1 << p.TTHs, TTW = 1 << p.TTWs;
    p.tilesX = up_div(tilesW, TTW);
    p.tilesY = up_div(tilesH, TTH);
    p.inH = 2 * TTH + 2;
    p.inW = 2 * TTW + 2;
    p.rowPitch = p.inW;
    // a wave's 16 tiles span 16 / TTW tile rows: their bank offsets (tile-row stride = 4 * rowPitch floats) must be distinct multiples of 4 * TTW
    // mod 64 -> rowPitch = 8 mod 16 for TTW = 8, an odd multiple of 4 for TTW = 4
    if (TTW == 8)
        while (p.rowPitch % 16 != 8) ++p.rowPitch;
    if (TTW == 4)
        while (p.rowPitch % 8 != 4) ++p.rowPitch;
    p.planeStride = round_up(TB * p.inH * p.rowPitch * 2, 64); // 0 mod 64 floats: with ds_read_b128's lane groups the k and k+1 planes interleave bank-exactly
    p.total = TB * p.inH * p.inW * 2;
    p.bufFloats = kUFloats + 4 * p.planeStride;
    p.nChunks = g.IC / 8;
    p.OCblocks = up_div(g.OC, ocPerBlock);
    p.useBN = g.useBN;
    p.ac2 = make_act_cfg(g.addAct >= 0 ? g.addAct : 0, g.addLeaky);
    const size_t lds = static_cast<size_t>(2) * p.bufFloats * sizeof(float);
    if (lds > 160 * 1024) return SNNHIP_E_UNSUPPORTED;

    // split-K: split the channel chunks while the layer has fewer block tiles than resident block slots
    const int cus = (ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256) * (3 - opb);
    const int blocks = p.tilesX * p.tilesY * up_div(g.N, TB) * p.OCblocks;
    int splitK = 1;
    if (const char* e = snnhip::option("SNNHIP_CONV_SPLITK")) splitK = std::max(1, atoi(e));
    else
        while (blocks * splitK * 2 <= cus && splitK * 2 <= p.nChunks / 2) splitK *= 2; // doubling must still fit one round of blocks
    splitK = std::min(splitK, p.nChunks);
    // round 6: the first factor of two of the split stays INSIDE the block (two K groups of four waves, kernel template KS = 2): 256 -> 256 @14x14 needs no
    // reduce launch at all, 512 -> 512 @7x7 reduces two partial tensors instead of four.  SNNHIP_WINO_KGROUPS=1 keeps the split over blockIdx.z only.
    int ks = 1;
    {
        const char* e = snnhip::option("SNNHIP_WINO_KGROUPS");
        const bool wanted = e ? atoi(e) == 2 : (splitK == 2 && !snnhip::option("SNNHIP_CONV_SPLITK")); // (where the reduce launch disappears; a deeper split keeps it)
        if (wanted && opb == 1 && splitK % 2 == 0 && p.nChunks % splitK == 0) {
            ks = 2;
            splitK /= 2;
        } else if (e && atoi(e) == 2 && opb == 1 && splitK == 1 && p.nChunks % 2 == 0) {
            ks = 2; // forced on a layer that would not split at all (tests)
        }
    }
    p.chunksPerSplit = up_div(p.nChunks, splitK);
    p.splitK = up_div(p.nChunks, p.chunksPerSplit);
    if (ks == 2 && (p.nChunks % p.splitK != 0 || p.chunksPerSplit % 2 != 0)) return SNNHIP_E_UNSUPPORTED; // (cannot happen with the conditions above)
    size_t ldsAll = ks == 2 ? std::max<size_t>(2 * lds, 128 * 1024) : lds; // two groups' buffers; the exchange of the sums needs 8 x 16 KB
    if (const char* padOpt = snnhip::option("SNNHIP_WINO_LDS_PAD")) ldsAll = std::min<size_t>(160 * 1024, ldsAll + static_cast<size_t>(atoi(padOpt))); // developer switch: residency experiments
    if (ldsAll > 160 * 1024) return SNNHIP_E_UNSUPPORTED;

    auto* plan = new WinoConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * 9);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->simple = act_is_simple(g.act);
    plan->ldsBytes = ldsAll;
    plan->ks = ks;
    plan->fusedAdd = g.addAct >= 0;
    if (plan->fusedAdd) plan->numInputs = 2;
    plan->opb = opb;
    plan->grid = dim3(p.tilesX * p.tilesY * up_div(g.N, TB), p.OCblocks, p.splitK);
    plan->dtype = SNNHIP_F32;
    for (int s = 0; s < 2; ++s) {
        const void* fn = ks == 2 ? (s ? reinterpret_cast<const void*>(conv2d_wino_kernel<true, 1, 2>) : reinterpret_cast<const void*>(conv2d_wino_kernel<false, 1, 2>))
                         : opb == 2 ? (s ? reinterpret_cast<const void*>(conv2d_wino_kernel<true, 2>) : reinterpret_cast<const void*>(conv2d_wino_kernel<false, 2>))
                                  : (s ? reinterpret_cast<const void*>(conv2d_wino_kernel<true, 1>) : reinterpret_cast<const void*>(conv2d_wino_kernel<false, 1>));
        if (ldsAll > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsAll)) != hipSuccess) {
            set_error("conv2d_wino: hipFuncSetAttribute(%zu) failed", ldsAll);
            delete plan;
            return SNNHIP_E_HIP;
        }
    }
    if (p.splitK > 1) {
        void* wsp = nullptr;
        const size_t wsBytes = static_cast<size_t>(p.splitK) * g.N * g.OH * g.OW * g.OC * sizeof(float);
        if (snnhip::dev_malloc(&wsp, wsBytes) != hipSuccess) {
            set_error("conv2d_wino: split-K workspace of %zu bytes", wsBytes);
            delete plan;
            return SNNHIP_E_HIP;
        }
        plan->deviceAllocs.push_back(wsp);
        plan->d_ws = static_cast<float*>(wsp);
    }

    // U = G g Gt in double, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], packed as the kernel's LDS image:
    //   Ug[oc block of 64][chunk][pos = 4i + j][op][lane = 16k + m] float4 {ocb = 2 op: ic = 8 chunk + 2k, + 1; ocb = 2 op + 1: the same two}, oc = 64 block + 16 ocb + m
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U(static_cast<size_t>(p.OCblocks) * p.nChunks * kUFloats, 0.0f);
    for (int oc = 0; oc < g.OC; ++oc)
        for (int ic = 0; ic < g.IC; ++ic) {
            const float* w = w_oihw + (static_cast<size_t>(oc) * g.IC + ic) * 9;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * w[0 * 3 + b] + G[i][1] * w[1 * 3 + b] + G[i][2] * w[2 * 3 + b];
            const int blk = oc / ocPerBlock, ocb = (oc % ocPerBlock) / 16, m = oc % 16, chunk = ic / 8, kk = (ic % 8) / 2, s = ic % 2;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                    const size_t at = ((static_cast<size_t>(blk) * p.nChunks + chunk) * 16 * opb + (i * 4 + j) * opb + (ocb >> 1)) * 256 + (kk * 16 + m) * 4 + (ocb & 1) * 2 + s;
                    U[at] = static_cast<float>(u);
                }
        }
    std::vector<float> epiP(static_cast<size_t>(p.OCblocks) * ocPerBlock * 4, 0.0f);
    std::memcpy(epiP.data(), epi4.data(), sizeof(float) * 4 * static_cast<size_t>(g.OC));
    int rc = plan->upload(U.data(), U.size(), &plan->d_u);
    if (rc == SNNHIP_OK) rc = plan->upload(epiP.data(), epiP.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * 9 * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N; // algorithmic (direct-convolution) count, SURVEY 8d
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * 9);
    const double mfmaFlops = 2.0 * 16 * 64 * 8 * ocPerBlock * static_cast<double>(plan->grid.x) * p.OCblocks * p.nChunks; // executed on the matrix pipe (padded tiles included)
    char buf[320];
    snprintf(buf, sizeof(buf), "conv2d_mfma_wino_f32_16x16x4 F(2x2,3x3) k=3x3 s=1 ic=%d oc=%d tile=%dx%dx%dpx x %doc chunk=8 lds=%zuB kgroups=%d splitK=%d mfma_flops=%.6g",
             g.IC, g.OC, TB, 2 * TTH, 2 * TTW, ocPerBlock, ldsAll, ks, p.splitK, mfmaFlops);
    plan->desc = buf;
    if (plan->fusedAdd) {
        plan->desc += " +add";
        plan->bytes += 4.0 * static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    }
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
