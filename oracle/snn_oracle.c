/*
 * snn_oracle.c -- CPU restatement of the ShaderNN operator hot path.  TEST INFRASTRUCTURE ONLY (see snn_oracle.h).
 *
 * Every function cites the reference file:line it restates.  Build with -ffp-contract=off so that the two
 * walks (NHWC / texel) are plain, reproducible fp32 multiply-then-add chains.
 */
#include "snn_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ROUND_UP(x, y) (((x) + (y) - 1) / (y) * (y))
#define UP_DIV(x, y) (((x) + (y) - 1) / (y))

/* ------------------------------------------------------------------------------------------------ */
/* host rules                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* core/src/ic2/conv2d.cpp:39-74 (identical copy in separableconvolution.cpp:27-62) */
void snn_oracle_padding_offsets(const char* padding, int kernel, int offsets[4]) {
    int alldigit = 1;
    for (const char* p = padding; *p; ++p) {
        if (!isdigit((unsigned char) *p)) alldigit = 0;
    }
    /* std::all_of over an empty string is true and std::stoul("") throws in the reference; callers never do that */
    if (alldigit && padding[0]) {
        int v = (int) strtoul(padding, NULL, 10);
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = v;
        return;
    }
    if (!strcmp(padding, "valid") || !strcmp(padding, "none")) {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = 0;
        return;
    }
    if (kernel > 1) {
        int p = kernel / 2;
        if (p < 1) p = 1;
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = p;
        if (kernel % 2 == 0) {
            offsets[0] -= 1;
            offsets[2] -= 1;
        }
    } else {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = 0;
    }
}

/* core/src/ic2/conv2d.cpp:102-113 + genericlayer.cpp:64-90 (float arithmetic, truncating conversion) */
int snn_oracle_out_dim(int in, int kernel, int stride, int padT, int padB) {
    float scale = 1 / (float) stride;
    float translation;
    if (kernel % 2 != 0) {
        translation = 1 + ((float) (unsigned) (padT + padB) - (float) kernel) / (float) stride;
    } else {
        translation = 1 + ((float) (unsigned) (padT + padB - 1) - (float) kernel) / (float) stride;
    }
    /* genericlayer.cpp:75-80: accumulated = max(0, scale*dim); translate = max(0, translation); width = sum */
    float s = scale * (float) in;
    if (s < 0.0f) s = 0.0f;
    float t = translation < 0.0f ? 0.0f : translation;
    return (int) (unsigned) (s + t);
}

/* core/src/ic2/conv2d.cpp:76-100 */
void snn_oracle_pack_conv_weights(const float* oihw, int IC, int OC, int kw, int kh, float* out) {
    const int unit = 4;
    int aligned = ROUND_UP(OC, unit) * kw * kh * ROUND_UP(IC, unit);
    int planeSize = ROUND_UP(OC, unit) * ROUND_UP(IC, unit);
    memset(out, 0, (size_t) aligned * sizeof(float));
    for (int b = 0; b < OC; ++b) {
        int b_4 = b / unit, mx = b % unit;
        for (int d = 0; d < IC; ++d) {
            for (int y = 0; y < kh; ++y) {
                for (int x = 0; x < kw; ++x) {
                    int base = (y * kw + x) * planeSize;
                    int inSize = ROUND_UP(IC, unit) * unit;
                    out[base + inSize * b_4 + d * unit + mx] = oihw[((size_t) (b * IC + d) * kh + y) * kw + x];
                }
            }
        }
    }
}

/* core/src/ic2/separableconvolution.cpp:88-111 */
void snn_oracle_pack_depthwise_weights(const float* w_chw, int C, int kw, int kh, float* out) {
    const int unit = 4;
    int aligned = ROUND_UP(C, unit) * kw * kh;
    int planeSize = ROUND_UP(C, unit) * kw;
    memset(out, 0, (size_t) aligned * sizeof(float));
    for (int b = 0; b < C; ++b) {
        int b_4 = b / unit, mx = b % unit;
        for (int y = 0; y < kh; ++y) {
            for (int x = 0; x < kw; ++x) {
                int base = y * planeSize;
                int inSize = ROUND_UP(C, unit);
                out[base + inSize * x + b_4 * unit + mx] = w_chw[((size_t) b * kh + y) * kw + x];
            }
        }
    }
}

/* demo/common/shaderUnitTest.cpp:109-128 */
void snn_oracle_hwc_to_c4hw4(const float* hwc, int H, int W, int C, float* c4) {
    int planes = UP_DIV(C, 4);
    memset(c4, 0, (size_t) planes * H * W * 4 * sizeof(float));
    for (int p = 0; p < planes; ++p) {
        float* dst = c4 + (size_t) p * H * W * 4;
        for (int i = 0; i < H * W; ++i) {
            for (int k = 0; k < 4; ++k) {
                if (p * 4 + k < C) dst[i * 4 + k] = hwc[(size_t) i * C + p * 4 + k];
            }
        }
    }
}

void snn_oracle_c4hw4_to_hwc(const float* c4, int H, int W, int C, float* hwc) {
    for (int i = 0; i < H * W; ++i) {
        for (int c = 0; c < C; ++c) {
            hwc[(size_t) i * C + c] = c4[((size_t) (c / 4) * H * W + i) * 4 + (c % 4)];
        }
    }
}

/* core/src/utils.cpp:127-174 */
float snn_oracle_to_medium_precision(float in) {
    union { unsigned int u; float f; } a, b;
    a.f = in;
    unsigned sign = (a.u & 0x80000000u) >> 31;
    unsigned exponent = (a.u & 0x7F800000u) >> 23;
    unsigned mantissa = a.u & 0x7FFFFFu;
    int newexp = (int) exponent + (-127 + 15);
    unsigned newMantissa;
    if (newexp >= 31) {
        newexp = 31;
        newMantissa = 0;
    } else if (newexp <= 0) {
        newexp = 0;
        newMantissa = 0;
    } else {
        newMantissa = mantissa >> 13;
    }
    if (newexp == 0) {
        b.u = sign << 31; /* newMantissa is always 0 here in the reference, so the denormal branch is dead */
    } else if (newexp == 31) {
        b.u = (sign << 31) | (0xFFu << 23) | (newMantissa << 13);
    } else {
        b.u = (sign << 31) | ((unsigned) (newexp + (-15 + 127)) << 23) | (newMantissa << 13);
    }
    return b.f;
}

/* ------------------------------------------------------------------------------------------------ */
/* epilogue: bias -> BN -> activation, as in shadertemplate_vk_conv2d.comp:276-340                  */
/* ------------------------------------------------------------------------------------------------ */

static inline float bn_apply(float v, float beta, float gamma, float mean, float var) {
    float sqrtVar = sqrtf(var + 0.001f);          /* :282 */
    if (sqrtVar < 0.0001f) sqrtVar = 0.0001f;     /* :283 */
    return ((gamma / sqrtVar) * (v - mean)) + beta; /* :284 */
}

/* `first` = POST-activation value of pixel 0 of the aligned 4-pixel x group (only used by SILU_QUIRK for pixels
 * 1..3: the shader overwrites color1 with silu(color1) before it is reused, vk_conv2d.comp:336-339).
 * Callers pass SNN_ACT_SILU for pixel 0 of a group. */
static inline float act_apply(int act, float leaky, float v, float first) {
    switch (act) {
    case SNN_ACT_RELU: return v > 0.0f ? v : 0.0f;                       /* :291-296 max(color, 0) */
    case SNN_ACT_RELU6: return v < 0.0f ? 0.0f : (v > 6.0f ? 6.0f : v);  /* :299-304 clamp */
    case SNN_ACT_TANH: return tanhf(v);                                  /* :307-312 */
    case SNN_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));               /* :315-320 */
    case SNN_ACT_LEAKY: {                                                /* :323-332 max(c, c*alpha) */
        float a = v * leaky;
        return v > a ? v : a;
    }
    case SNN_ACT_SILU: return v * 1.0f / (1.0f + expf(-v));              /* :335-336 (color1 form) */
    case SNN_ACT_SILU_QUIRK: return v * 1.0f / (1.0f + expf(-first));    /* :337-339 */
    default: return v;
    }
}

/* coordinate resolution of shadertemplate_vk_conv2d.comp:168-185 (y) / :196-218 (x). returns -1 => fetch is 0 */
static inline int resolve_coord(int s, int size, int padMode) {
    switch (padMode) {
    case SNN_PAD_CONSTANT: return (s >= 0 && s < size) ? s : -1; /* redirected to uInputSize => OOB texel = 0 */
    case SNN_PAD_REPLICATE: return s < 0 ? 0 : (s > size - 1 ? size - 1 : s);
    case SNN_PAD_REFLECT:
        s = s < 0 ? -s : s;
        s = s >= size ? 2 * size - 2 - s : s;
        return (s >= 0 && s < size) ? s : -1;
    default: return (s >= 0 && s < size) ? s : -1; /* no adjustment: out-of-range texelFetch reads 0 */
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* conv2d, NHWC walk                                                                                */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
    const snn_oracle_conv_desc* d;
    const float *x, *w_hwio, *bias, *beta, *gamma, *mean, *var;
    float* y;
    int row0, row1; /* rows of N*OH */
} conv_job;

/* Quirk Q3 (conv2dVulkan.cpp:183-184, vk_conv2d.comp:154-155): uPadx <- offsets[0] (T), uPady <- offsets[2] (L).
 * k==1 uses the 1x1 shader: no padding at all (conv2d_1x1.comp:74-75). */
static void conv_rows(const conv_job* j) {
    const snn_oracle_conv_desc* d = j->d;
    const int is1x1 = (d->kh == 1 && d->kw == 1);
    const int padx = is1x1 ? 0 : d->padT, pady = is1x1 ? 0 : d->padL;
    const int padMode = is1x1 ? SNN_PAD_NONE : d->padMode;
    const int OC = d->OC, IC = d->IC;
    float* acc = (float*) malloc(sizeof(float) * (size_t) OC * 4);
    for (int r = j->row0; r < j->row1; ++r) {
        int n = r / d->OH, oy = r % d->OH;
        const float* xn = j->x + (size_t) n * d->H * d->W * IC;
        for (int ox0 = 0; ox0 < d->OW; ox0 += 4) { /* 4-pixel groups like the shader's per-thread tile */
            int npx = d->OW - ox0 < 4 ? d->OW - ox0 : 4;
            for (int p = 0; p < npx; ++p) {
                int ox = ox0 + p;
                float* a = acc + p * OC;
                for (int o = 0; o < OC; ++o) a[o] = (d->useBias && j->bias) ? j->bias[o] : 0.0f;
                for (int fy = 0; fy < d->kh; ++fy) {
                    int sy = resolve_coord(oy * d->sh - pady + fy, d->H, padMode);
                    if (sy < 0) continue;
                    for (int fx = 0; fx < d->kw; ++fx) {
                        int sx = resolve_coord(ox * d->sw - padx + fx, d->W, padMode);
                        if (sx < 0) continue;
                        const float* xp = xn + ((size_t) sy * d->W + sx) * IC;
                        const float* wp = j->w_hwio + (size_t) (fy * d->kw + fx) * IC * OC;
                        for (int i = 0; i < IC; ++i) {
                            float xv = xp[i];
                            const float* wr = wp + (size_t) i * OC;
                            for (int o = 0; o < OC; ++o) a[o] += wr[o] * xv;
                        }
                    }
                }
                if (d->useBN) {
                    for (int o = 0; o < OC; ++o) a[o] = bn_apply(a[o], j->beta[o], j->gamma[o], j->mean[o], j->var[o]);
                }
            }
            float* y0 = j->y + (((size_t) n * d->OH + oy) * d->OW + ox0) * OC;
            for (int p = 0; p < npx; ++p) {
                float* yo = y0 + (size_t) p * OC;
                int a = (d->act == SNN_ACT_SILU_QUIRK && p == 0) ? SNN_ACT_SILU : d->act;
                for (int o = 0; o < OC; ++o) yo[o] = act_apply(a, d->leaky, acc[p * OC + o], y0[o]);
            }
        }
    }
    free(acc);
}

static void* conv_thread(void* arg) {
    conv_rows((const conv_job*) arg);
    return NULL;
}

void snn_oracle_conv2d_nhwc_mt(const snn_oracle_conv_desc* d, const float* x, const float* w_oihw, const float* bias,
                               const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var, float* y,
                               int threads) {
    /* re-order weights OIHW -> [kh][kw][IC][OC] so that the innermost loop runs over contiguous OC */
    size_t wn = (size_t) d->kh * d->kw * d->IC * d->OC;
    float* w_hwio = (float*) malloc(wn * sizeof(float));
    for (int o = 0; o < d->OC; ++o)
        for (int i = 0; i < d->IC; ++i)
            for (int fy = 0; fy < d->kh; ++fy)
                for (int fx = 0; fx < d->kw; ++fx)
                    w_hwio[((size_t) (fy * d->kw + fx) * d->IC + i) * d->OC + o] = w_oihw[(((size_t) o * d->IC + i) * d->kh + fy) * d->kw + fx];
    int rows = d->N * d->OH;
    if (threads < 1) threads = 1;
    if (threads > rows) threads = rows;
    conv_job* jobs = (conv_job*) calloc((size_t) threads, sizeof(conv_job));
    pthread_t* tids = (pthread_t*) calloc((size_t) threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) {
        conv_job j = {d, x, w_hwio, bias, bn_beta, bn_gamma, bn_mean, bn_var, y, (int) ((long) rows * t / threads),
                      (int) ((long) rows * (t + 1) / threads)};
        jobs[t] = j;
    }
    if (threads == 1) {
        conv_rows(&jobs[0]);
    } else {
        for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, conv_thread, &jobs[t]);
        for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
    }
    free(jobs);
    free(tids);
    free(w_hwio);
}

void snn_oracle_conv2d_nhwc(const snn_oracle_conv_desc* d, const float* x, const float* w_oihw, const float* bias,
                            const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var, float* y) {
    snn_oracle_conv2d_nhwc_mt(d, x, w_oihw, bias, bn_beta, bn_gamma, bn_mean, bn_var, y, 1);
}

/* ------------------------------------------------------------------------------------------------ */
/* conv2d, texel walk == the GLSL main() bodies                                                     */
/* ------------------------------------------------------------------------------------------------ */

static inline void fetch_texel(const float* x_c4, int H, int W, int D, int sx, int sy, int fz, float v[4]) {
    /* texelFetch on an out-of-range coordinate returns 0 (the shaders rely on it: vk_conv2d.comp:170) */
    if (sx < 0 || sx >= W || sy < 0 || sy >= H || fz < 0 || fz >= D) {
        v[0] = v[1] = v[2] = v[3] = 0.0f;
        return;
    }
    const float* p = x_c4 + (((size_t) fz * H + sy) * W + sx) * 4;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
}

static inline int shader_coord(int s, int size, int padMode) {
    /* vk_conv2d.comp:168-185: returns the coordinate the shader would fetch (may be out of range => 0) */
    if (padMode == SNN_PAD_CONSTANT) return (s >= 0 && s < size) ? s : size;
    if (padMode == SNN_PAD_REPLICATE) return s < 0 ? 0 : (s > size - 1 ? size - 1 : s);
    if (padMode == SNN_PAD_REFLECT) {
        s = s < 0 ? -s : s;
        s = s >= size ? 2 * size - 2 - s : s;
        return s;
    }
    return s;
}

void snn_oracle_conv2d_texel(const snn_oracle_conv_desc* d, const float* x_c4, const float* w_packed, const float* bias4,
                             const float* bn_beta4, const float* bn_gamma4, const float* bn_mean4, const float* bn_var4,
                             float* y_c4) {
    const int ic_4 = UP_DIV(d->IC, 4), oc_4 = UP_DIV(d->OC, 4);
    const int is1x1 = (d->kh == 1 && d->kw == 1);
    /* weight image dims [ic_4*4, oc_4, k*k] RGBA (conv2dVulkan.cpp:99): texel (x=i, y=oz, z=tap) = 4 out-channels */
    const int wRow = ic_4 * 4 * 4, wPlane = wRow * oc_4;
    for (int oz = 0; oz < oc_4; ++oz) {
        for (int oy = 0; oy < d->OH; ++oy) {
            for (int gx = 0; gx * 4 < d->OW; ++gx) {
                float color[4][4];
                for (int p = 0; p < 4; ++p)
                    for (int c = 0; c < 4; ++c) color[p][c] = bias4[oz * 4 + c];
                int posx = gx * 4;
                if (is1x1) { /* shadertemplate_vk_conv2d_1x1.comp:72-112 */
                    int sy = oy * d->sh;
                    int sx[4];
                    float m[4];
                    for (int p = 0; p < 4; ++p) {
                        sx[p] = posx * d->sw + p * d->sw;
                        m[p] = (sx[p] >= 0 && sx[p] < d->W) ? 1.0f : 0.0f;
                    }
                    for (int fz = 0; fz < ic_4; ++fz) {
                        const float* k = w_packed + oz * wRow + fz * 16;
                        for (int p = 0; p < 4; ++p) {
                            float v[4];
                            fetch_texel(x_c4, d->H, d->W, ic_4, sx[p], sy, fz, v);
                            for (int c = 0; c < 4; ++c) {
                                float t = k[0 * 4 + c] * v[0] + k[1 * 4 + c] * v[1] + k[2 * 4 + c] * v[2] + k[3 * 4 + c] * v[3];
                                color[p][c] += t * m[p];
                            }
                        }
                    }
                } else { /* shadertemplate_vk_conv2d.comp:152-274 */
                    int s0x = posx * d->sw - d->padT; /* uPadx = offsets[0] */
                    int s0y = oy * d->sh - d->padL;   /* uPady = offsets[2] */
                    for (int fy = 0; fy < d->kh; ++fy) {
                        int sy = shader_coord(fy + s0y, d->H, d->padMode);
                        for (int fx = 0; fx < d->kw; ++fx) {
                            int tap = fx + fy * d->kw;
                            int sx[4];
                            for (int p = 0; p < 4; ++p) sx[p] = shader_coord(fx + s0x + p * d->sw, d->W, d->padMode);
                            for (int fz = 0; fz < ic_4; ++fz) {
                                const float* k = w_packed + (size_t) tap * wPlane + oz * wRow + fz * 16;
                                for (int p = 0; p < 4; ++p) {
                                    float v[4];
                                    fetch_texel(x_c4, d->H, d->W, ic_4, sx[p], sy, fz, v);
                                    for (int c = 0; c < 4; ++c) {
                                        color[p][c] += k[0 * 4 + c] * v[0] + k[1 * 4 + c] * v[1] + k[2 * 4 + c] * v[2] + k[3 * 4 + c] * v[3];
                                    }
                                }
                            }
                        }
                    }
                }
                if (d->useBN) {
                    for (int p = 0; p < 4; ++p)
                        for (int c = 0; c < 4; ++c) {
                            int o = oz * 4 + c;
                            color[p][c] = bn_apply(color[p][c], bn_beta4[o], bn_gamma4[o], bn_mean4[o], bn_var4[o]);
                        }
                }
                float out0[4];
                for (int p = 0; p < 4; ++p) {
                    int a = (d->act == SNN_ACT_SILU_QUIRK && p == 0) ? SNN_ACT_SILU : d->act;
                    float o4[4];
                    for (int c = 0; c < 4; ++c) o4[c] = act_apply(a, d->leaky, color[p][c], out0[c]);
                    if (p == 0) memcpy(out0, o4, sizeof(o4));
                    if (posx + p >= d->OW) continue; /* imageStore outside the image is dropped */
                    float* yo = y_c4 + (((size_t) oz * d->OH + oy) * d->OW + posx + p) * 4;
                    memcpy(yo, o4, sizeof(o4));
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* depthwise                                                                                         */
/* ------------------------------------------------------------------------------------------------ */

/* shadertemplate_vk_depthwise.comp:64-137 restated over NHWC.  Taps outside the image are skipped
 * (sfxy/efxy clip, :77-78) == zero padding.  uPadx <- offsets[0], uPady <- offsets[2]
 * (separableconvolutionVulkan.cpp:112-113). */
void snn_oracle_depthwise_nhwc(const snn_oracle_conv_desc* d, const float* x, const float* w_chw, const float* bias,
                               const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var, float* y) {
    const int C = d->OC;
    for (int n = 0; n < d->N; ++n) {
        const float* xn = x + (size_t) n * d->H * d->W * C;
        for (int oy = 0; oy < d->OH; ++oy) {
            for (int ox = 0; ox < d->OW; ++ox) {
                int s0x = ox * d->sw - d->padT, s0y = oy * d->sh - d->padL;
                float* yo = y + (((size_t) n * d->OH + oy) * d->OW + ox) * C;
                for (int c = 0; c < C; ++c) {
                    float acc = bias ? bias[c] : 0.0f; /* bias buffer is always read (:79) */
                    for (int fy = 0; fy < d->kh; ++fy) {
                        int sy = s0y + fy;
                        if (sy < 0 || sy >= d->H) continue;
                        for (int fx = 0; fx < d->kw; ++fx) {
                            int sx = s0x + fx;
                            if (sx < 0 || sx >= d->W) continue;
                            acc += w_chw[((size_t) c * d->kh + fy) * d->kw + fx] * xn[((size_t) sy * d->W + sx) * C + c];
                        }
                    }
                    if (d->useBN) acc = bn_apply(acc, bn_beta[c], bn_gamma[c], bn_mean[c], bn_var[c]);
                    yo[c] = act_apply(d->act == SNN_ACT_SILU_QUIRK ? SNN_ACT_SILU : d->act, d->leaky, acc, acc);
                }
            }
        }
    }
}

void snn_oracle_depthwise_texel(const snn_oracle_conv_desc* d, const float* x_c4, const float* w_packed, const float* bias4,
                                const float* bn_beta4, const float* bn_gamma4, const float* bn_mean4, const float* bn_var4,
                                float* y_c4) {
    const int c_4 = UP_DIV(d->OC, 4);
    for (int z = 0; z < c_4; ++z) {
        for (int oy = 0; oy < d->OH; ++oy) {
            for (int ox = 0; ox < d->OW; ++ox) {
                int s0x = ox * d->sw - d->padT, s0y = oy * d->sh - d->padL;
                /* sfxy = max(0, UP_DIV(-s0, dilate)); efxy = min(kernel, UP_DIV(input - s0, dilate)), dilate = 1 */
                int sfx = -s0x > 0 ? -s0x : 0, sfy = -s0y > 0 ? -s0y : 0;
                int efx = d->W - s0x < d->kw ? d->W - s0x : d->kw, efy = d->H - s0y < d->kh ? d->H - s0y : d->kh;
                float color[4];
                for (int c = 0; c < 4; ++c) color[c] = bias4[z * 4 + c];
                for (int fy = sfy; fy < efy; ++fy) {
                    int sy = fy + s0y;
                    for (int fx = sfx; fx < efx; ++fx) {
                        int sx = fx + s0x;
                        /* Filter.data[pos.z + fx*uInputSizez + fy*uInputSizez*uKernelSizey] (:86) */
                        const float* k = w_packed + ((size_t) z + (size_t) fx * c_4 + (size_t) fy * c_4 * d->kh) * 4;
                        float v[4];
                        fetch_texel(x_c4, d->H, d->W, c_4, sx, sy, z, v);
                        for (int c = 0; c < 4; ++c) color[c] += k[c] * v[c];
                    }
                }
                float* yo = y_c4 + (((size_t) z * d->OH + oy) * d->OW + ox) * 4;
                for (int c = 0; c < 4; ++c) {
                    int o = z * 4 + c;
                    float v = color[c];
                    if (d->useBN) v = bn_apply(v, bn_beta4[o], bn_gamma4[o], bn_mean4[o], bn_var4[o]);
                    yo[c] = act_apply(d->act == SNN_ACT_SILU_QUIRK ? SNN_ACT_SILU : d->act, d->leaky, v, v);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* dense (reference CPU operator path, Eigen)                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* cpulayer.h:38-42: the map only knows these keys; operator[] on anything else yields RELU (enum value 0) */
int snn_oracle_dense_act_from_string(const char* s) {
    if (!strcmp(s, "relu")) return SNN_DENSE_ACT_RELU;
    if (!strcmp(s, "leakyRelu")) return SNN_DENSE_ACT_LEAKY;
    if (!strcmp(s, "sigmoid")) return SNN_DENSE_ACT_SIGMOID;
    if (!strcmp(s, "softmax")) return SNN_DENSE_ACT_SOFTMAX;
    if (!strcmp(s, "tanh")) return SNN_DENSE_ACT_TANH;
    if (!strcmp(s, "SiLU")) return SNN_DENSE_ACT_SILU;
    if (!strcmp(s, "identity") || !strcmp(s, "")) return SNN_DENSE_ACT_IDENTITY;
    return SNN_DENSE_ACT_RELU;
}

/* cpulayer.h:136-171 (transform) + :199-261 (activation).
 * Eigen evaluates W(Out x In, row major) * x; the summation order inside Eigen's GEMV kernel is not the
 * plain left-to-right chain, so parity with oracle/_ref is checked at 1e-5 relative, not bit-exact. */
void snn_oracle_dense(const float* x, int batch, int In, int Out, const float* w_flat, const float* bias, int act, float leaky,
                      float* y) {
    for (int b = 0; b < batch; ++b) {
        const float* xb = x + (size_t) b * In;
        float* yb = y + (size_t) b * Out;
        for (int o = 0; o < Out; ++o) {
            float acc = 0.0f;
            const float* wr = w_flat + (size_t) o * In;
            for (int i = 0; i < In; ++i) acc += wr[i] * xb[i];
            yb[o] = acc + (bias ? bias[o] : 0.0f);
        }
        switch (act) {
        case SNN_DENSE_ACT_RELU:
            for (int o = 0; o < Out; ++o) yb[o] = yb[o] > 0 ? yb[o] : 0.0f * yb[o]; /* leakyRelu(val, 0.0), :187 */
            break;
        case SNN_DENSE_ACT_LEAKY:
            for (int o = 0; o < Out; ++o) yb[o] = yb[o] > 0 ? yb[o] : leaky * yb[o];
            break;
        case SNN_DENSE_ACT_SIGMOID:
            for (int o = 0; o < Out; ++o) yb[o] = 1.0f / (1.0f + expf(-yb[o]));
            break;
        case SNN_DENSE_ACT_SOFTMAX: { /* :173-189 */
            float mx = -3.402823466e+38f;
            for (int o = 0; o < Out; ++o) mx = yb[o] > mx ? yb[o] : mx;
            float div = 0.0f;
            for (int o = 0; o < Out; ++o) {
                yb[o] = expf(yb[o] - mx);
                div += yb[o];
            }
            for (int o = 0; o < Out; ++o) yb[o] = yb[o] / div;
            break;
        }
        case SNN_DENSE_ACT_TANH: /* :195 */
            for (int o = 0; o < Out; ++o) yb[o] = (expf(2 * yb[o]) - 1) / (expf(2 * yb[o]) + 1);
            break;
        case SNN_DENSE_ACT_SILU: /* by-value loop: no effect (:245-252) */
        default: break;
        }
    }
}

/* cpulayer.h:94-113 */
void snn_oracle_flatten_c4hw4(const float* x_c4, int H, int W, int C, float* out_hwc) {
    int depth = UP_DIV(C, 4);
    size_t k = 0;
    for (int row = 0; row < H; ++row)
        for (int col = 0; col < W; ++col)
            for (int plane = 0; plane < depth; ++plane)
                for (int ch = 0; ch < 4; ++ch) {
                    if (plane * 4 + ch < C) out_hwc[k++] = x_c4[(((size_t) plane * H + row) * W + col) * 4 + ch];
                }
}

/* ------------------------------------------------------------------------------------------------ */
/* subpixel                                                                                          */
/* ------------------------------------------------------------------------------------------------ */

void snn_oracle_subpixel_nhwc(const float* x, int N, int H, int W, int C, int factor, int mode, float* y) {
    const int OH = H * factor, OW = W * factor;
    const int depth = UP_DIV(C, 4);
    for (int n = 0; n < N; ++n) {
        for (int oy = 0; oy < OH; ++oy) {
            for (int ox = 0; ox < OW; ++ox) {
                int x1 = ox / factor, y1 = oy / factor;              /* floor(pos/f), vk_subpixel.comp:49-55 */
                if (x1 > W - 1) x1 = W - 1;
                if (y1 > H - 1) y1 = H - 1;
                int z1 = (ox % factor) + (oy % factor) * factor;     /* :57 */
                int ch;
                if (mode == SNN_SUBPIXEL_VK_QUIRK) {
                    int z11 = z1 < depth - 1 ? z1 : depth - 1;       /* :58 clamp to depth SLICE */
                    ch = z11 * 4;                                     /* :60-62 takes .x of that texel */
                } else {
                    ch = z1;                                          /* shadertemplate_fs_subpixel.glsl:41-64 */
                }
                float v = (ch < C) ? x[(((size_t) n * H + y1) * W + x1) * C + ch] : 0.0f;
                y[((size_t) n * OH + oy) * OW + ox] = tanhf(v);       /* :64-66 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* deterministic generator: 24,55 lagged Fibonacci with run-ahead (TAoCP 3.2.2(7)), the public-domain  */
/* algorithm demo/common/prng.h states; RandomFloat as demo/common/testutil.cpp:41-46.               */
/* ------------------------------------------------------------------------------------------------ */

/* ===================================================================================================== */
/* element-wise / pooling / shape operators                                         */
/* ===================================================================================================== */

/* shadertemplate_vk_add.comp:47-88 (the same ladder is in vk_activation.comp:52-86 and vk_batchnorm.comp:70-103) */
void snn_oracle_add_act(const float* a, const float* b, long count, int act, float leaky, float* y) {
    for (long i = 0; i < count; ++i) {
        float v = a[i];
        if (b) v = v + b[i];
        y[i] = act_apply(act, leaky, v, 0.0f);
    }
}

void snn_oracle_add_ragged(const float* a, int H0, int W0, const float* b, int H1, int W1, int N, int C, int act, float leaky, float* y) {
    int H = H0 > H1 ? H0 : H1, W = W0 > W1 ? W0 : W1;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy)
            for (int ox = 0; ox < W; ++ox)
                for (int c = 0; c < C; ++c) {
                    float v = 0.0f;
                    if (oy < H0 && ox < W0) { /* all(lessThan(pos, inSize)), inSize = first input */
                        v = a[(((long) n * H0 + oy) * W0 + ox) * C + c];
                        if (oy < H1 && ox < W1) v = v + b[(((long) n * H1 + oy) * W1 + ox) * C + c];
                        v = act_apply(act, leaky, v, 0.0f);
                    }
                    y[(((long) n * H + oy) * W + ox) * C + c] = v;
                }
}

/* shadertemplate_vk_batchnorm.comp:61-68 */
void snn_oracle_batchnorm(const float* x, long pixels, int C, const float* beta, const float* gamma, const float* mean, const float* var, int act,
                          float leaky, float* y) {
    for (long p = 0; p < pixels; ++p)
        for (int c = 0; c < C; ++c) {
            float sqrtVar = sqrtf(var[c] + 0.001f);
            sqrtVar = sqrtVar > 0.0001f ? sqrtVar : 0.0001f;
            float v = ((gamma[c] / sqrtVar) * (x[p * C + c] - mean[c])) + beta[c];
            y[p * C + c] = act_apply(act, leaky, v, 0.0f);
        }
}

/* maxpool2d.cpp:26-36 / avgpool2d.cpp:20-29 + genericlayer.cpp:64-90 (translation accumulated with max(0, .), truncating uint32 store) */
int snn_oracle_pool_out_dim(int in, int kernel, int stride, int same) {
    float scale = 1.0f / (float) stride;
    float translation = same ? 1.0f - 1.0f / (float) stride : 1.0f - ((float) kernel / (float) stride);
    float s = scale * (float) in;
    if (s < 0.0f) s = 0.0f;
    float t = translation < 0.0f ? 0.0f : translation;
    return (int) (uint32_t) (s + t);
}

/* shadertemplate_vk_maxpool2d.comp:52-73 / shadertemplate_vk_avgpool2d.comp:52-68 */
void snn_oracle_pool2d(const float* x, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int padT, int padL, int OH, int OW, int type,
                       float* y) {
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                int sx = ox * sw - padL, sy = oy * sh - padT;     /* spos = pos.xy*stride - pad */
                int sfx = 0 > -sx ? 0 : -sx, sfy = 0 > -sy ? 0 : -sy; /* sfxy = max(0, -spos) */
                int efx = kw < W - sx ? kw : W - sx, efy = kh < H - sy ? kh : H - sy; /* efxy = min(kernelSize, inputSize - spos) */
                for (int c = 0; c < C; ++c) {
                    float color = type == 0 ? -100000.0f : 0.0f, num = 0.0f;
                    for (int fy = sfy; fy < efy; ++fy)
                        for (int fx = sfx; fx < efx; ++fx) {
                            float v = x[(((long) n * H + sy + fy) * W + sx + fx) * C + c];
                            if (type == 0) color = color > v ? color : v;
                            else color += v;
                            num += 1.0f;
                        }
                    y[(((long) n * OH + oy) * OW + ox) * C + c] = type == 0 ? color : color / num;
                }
            }
}

/* shadertemplate_vk_pad.comp:50-70 */
void snn_oracle_pad(const float* x, int N, int H, int W, int C, int padT, int padB, int padL, int padR, int mode, float* y) {
    int OH = H + padT + padB, OW = W + padL + padR;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                int sx = ox - padT, sy = oy - padL; /* s0 = pos.xy - uPad, uPad = {offsets[0], offsets[2]} = {T, L} */
                if (mode == 0) {
                    sx = (sx >= 0 && sx < W) ? sx : W;
                    sy = (sy >= 0 && sy < H) ? sy : H;
                } else if (mode == 1) {
                    sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
                    sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
                } else {
                    sx = sx < 0 ? -sx : sx;
                    sx = sx >= W ? 2 * W - 2 - sx : sx;
                    sy = sy < 0 ? -sy : sy;
                    sy = sy >= H ? 2 * H - 2 - sy : sy;
                }
                int inside = sx >= 0 && sx < W && sy >= 0 && sy < H; /* texelFetch outside the texture: 0 */
                for (int c = 0; c < C; ++c) y[(((long) n * OH + oy) * OW + ox) * C + c] = inside ? x[(((long) n * H + sy) * W + sx) * C + c] : 0.0f;
            }
}

static float up_fetch(const float* x, int H, int W, int C, int px, int py, int c) {
    return (px >= 0 && px < W && py >= 0 && py < H) ? x[((long) py * W + px) * C + c] : 0.0f;
}

/* shadertemplate_vk_upsampling2d_nearest.comp:50-65 / shadertemplate_vk_upsampling2d_bilinear.comp:49-74 (means 0, norms 1) */
void snn_oracle_upsample(const float* x, int N, int H, int W, int C, float scale, int mode, float* y) {
    int OH = (int) (uint32_t) (scale * (float) H), OW = (int) (uint32_t) (scale * (float) W);
    float inv = 1.0f / scale;
    for (int n = 0; n < N; ++n) {
        const float* xn = x + (long) n * H * W * C;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                for (int c = 0; c < C; ++c) {
                    float out;
                    if (mode == 0) {
                        int x1 = (int) floorf((float) ox * inv), y1 = (int) floorf((float) oy * inv);
                        x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
                        y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
                        out = up_fetch(xn, H, W, C, x1, y1, c);
                    } else {
                        float off = 0.5f - 0.5f * inv;
                        float srcX = (float) ox * inv - off;
                        srcX = srcX < 0.0f ? 0.0f : (srcX > (float) (W - 1) ? (float) (W - 1) : srcX);
                        int x11 = (int) floorf(srcX), x12 = x11 + 1;
                        float srcY = (float) oy * inv - off;
                        srcY = srcY < 0.0f ? 0.0f : (srcY > (float) (H - 1) ? (float) (H - 1) : srcY);
                        int y11 = (int) floorf(srcY), y12 = y11 + 1;
                        float r4 = up_fetch(xn, H, W, C, x11, y12, c), r3 = up_fetch(xn, H, W, C, x12, y12, c);
                        float r1 = up_fetch(xn, H, W, C, x11, y11, c), r2 = up_fetch(xn, H, W, C, x12, y11, c);
                        out = r1 * (((float) x12 - srcX) * ((float) y12 - srcY)) + r2 * ((srcX - (float) x11) * ((float) y12 - srcY)) +
                              r3 * ((srcX - (float) x11) * (srcY - (float) y11)) + r4 * (((float) x12 - srcX) * (srcY - (float) y11));
                    }
                    y[(((long) n * OH + oy) * OW + ox) * C + c] = out;
                }
    }
}

/* ---- SURVEY.md section 8(f) rank 4 ---- */

void snn_oracle_concat(const float* x0, const float* x1, long pixels, int C0, int C1, int OC, float* y) {
    int P0 = (C0 + 3) / 4, P1 = (C1 + 3) / 4;
    for (long px = 0; px < pixels; ++px)
        for (int c = 0; c < OC; ++c) {
            int p = c / 4, l = c % 4;
            float v = 0.0f;
            if (p < P0) { /* vk_concat.comp:45-47 */
                int ch = 4 * p + l;
                if (ch < C0) v = x0[px * C0 + ch];
            } else if (p - P0 < P1) { /* :48-49 */
                int ch = 4 * (p - P0) + l;
                if (ch < C1) v = x1[px * C1 + ch];
            }
            y[px * OC + c] = v;
        }
}

void snn_oracle_unary(const float* x, long count, int op, float value, float* y) {
    for (long i = 0; i < count; ++i) {
        float c = x[i];
        switch (op) {
        case 1: c = value; break;
        case 2: c = 0.0f - c; break;
        case 3: c = 1.0f / c; break;
        case 4: c = c * c; break;
        case 5: c = expf(c); break;
        case 6: c = fabsf(c); break;
        default: break;
        }
        y[i] = c;
    }
}

void snn_oracle_calculate(const float* x, long pixels, int C, int OC, float* y) {
    for (long px = 0; px < pixels; ++px) {
        float illumination = x[px * C + 8]; /* texelFetch(inputTextures, ivec3(uv, 2), 0).r */
        for (int c = 0; c < OC; ++c) y[px * OC + c] = (c % 4 < 3) ? x[px * C + c % 4] / illumination : 0.0f;
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void snn_oracle_resize(const float* x, int N, int H, int W, int C, int OH, int OW, const float* means4, const float* norms4, int linear, float* y) {
    for (int n = 0; n < N; ++n) {
        const float* img = x + (long) n * H * W * C;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                float u = ((float) ox + 0.5f) / (float) OW * (float) W; /* texCoords * textureSize */
                float v = ((float) oy + 0.5f) / (float) OH * (float) H;
                for (int c = 0; c < C; ++c) {
                    float val;
                    if (linear) {
                        float fu = u - 0.5f, fv = v - 0.5f;
                        float x0f = floorf(fu), y0f = floorf(fv);
                        float ax = fu - x0f, ay = fv - y0f;
                        int x0 = clampi((int) x0f, 0, W - 1), x1 = clampi((int) x0f + 1, 0, W - 1);
                        int y0 = clampi((int) y0f, 0, H - 1), y1 = clampi((int) y0f + 1, 0, H - 1);
                        float top = img[((long) y0 * W + x0) * C + c] * (1.0f - ax) + img[((long) y0 * W + x1) * C + c] * ax;
                        float bot = img[((long) y1 * W + x0) * C + c] * (1.0f - ax) + img[((long) y1 * W + x1) * C + c] * ax;
                        val = top * (1.0f - ay) + bot * ay;
                    } else {
                        val = img[((long) clampi((int) floorf(v), 0, H - 1) * W + clampi((int) floorf(u), 0, W - 1)) * C + c];
                    }
                    y[(((long) n * OH + oy) * OW + ox) * C + c] = (val - means4[c % 4]) * norms4[c % 4];
                }
            }
    }
}

void snn_oracle_image_u8(const unsigned char* x, long pixels, int sc, const float* m, const float* s, float* y) {
    for (long i = 0; i < pixels; ++i) {
        const unsigned char* src = x + i * sc;
        float* dst = y + i * 4;
        if (sc == 4) {
            for (int c = 0; c < 4; ++c) dst[c] = (float) (((float) src[c] - m[c]) * s[c]);
        } else if (sc == 3) {
            for (int c = 0; c < 3; ++c) dst[c] = (float) (((float) src[c] - m[c]) * s[c]);
            dst[3] = 1.0f;
        } else {
            dst[0] = (float) (((float) src[0] - m[0]) * s[0]);
            dst[1] = dst[2] = dst[3] = (float) ((0.0f - m[0]) * s[0]);
        }
    }
}

long snn_oracle_argmax(const float* x, long count) {
    long best = 0;
    for (long i = 1; i < count; ++i)
        if (x[best] < x[i]) best = i; /* std::max_element keeps the first of equal elements */
    return best;
}

void snn_oracle_deconv2d(const float* x, int N, int H, int W, int IC, int OC, int k, int s, int p, int OH, int OW, const float* w, const float* bias,
                         const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var, int act, float leaky, float* y) {
    int base = k - 1 - p;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                for (int o = 0; o < OC; ++o) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < H; ++iy) {
                        int ky = base - oy + s * iy;
                        if (ky < 0 || ky >= k) continue;
                        for (int ix = 0; ix < W; ++ix) {
                            int kx = base - ox + s * ix;
                            if (kx < 0 || kx >= k) continue;
                            const float* px = x + (((long) n * H + iy) * W + ix) * IC;
                            for (int i = 0; i < IC; ++i) acc += px[i] * w[(((long) o * IC + i) * k + ky) * k + kx];
                        }
                    }
                    float v = acc + (bias ? bias[o] : 0.0f);
                    if (bn_beta) v = ((bn_gamma[o] / sqrtf(bn_var[o] + 0.001f)) * (v - bn_mean[o])) + bn_beta[o];
                    y[(((long) n * OH + oy) * OW + ox) * OC + o] = act_apply(act, leaky, v, 0.0f);
                }
}

static void deconv_fetch(const float* x, int H, int W, int IC, int px, int py, int layer, float t[4]) {
    for (int j = 0; j < 4; ++j) {
        int c = 4 * layer + j;
        t[j] = (px >= 0 && px < W && py >= 0 && py < H && c < IC) ? x[((long) py * W + px) * IC + c] : 0.0f;
    }
}

void snn_oracle_deconv4x4s2_shader(const float* x, int H, int W, int IC, int OC, const float* w_oihw, const float* bias, float* y) {
    int layers = (IC + 3) / 4, OH = 2 * H, OW = 2 * W;
    /* deconv2dGL.cpp:28-80: weightMatrix[o][16*layer + tap] = vec4 over the layer's 4 input channels */
    float* wm = (float*) calloc((size_t) OC * layers * 16 * 4, sizeof(float));
    for (int o = 0; o < OC; ++o)
        for (int i = 0; i < IC; ++i)
            for (int t = 0; t < 16; ++t) wm[(((long) o * layers + i / 4) * 16 + t) * 4 + i % 4] = w_oihw[((long) o * IC + i) * 16 + t];
    for (int gy = 0; gy < OH; ++gy)
        for (int gx = 0; gx < OW; ++gx) {
            int bx = (gx + 1) / 2, by = (gy + 1) / 2;                 /* :151 */
            int bw = (gx % 2) + 4 * (gy % 2);                         /* :156 */
            int widx[4] = {0 + bw, 2 + bw, 8 + bw, 10 + bw};
            for (int o = 0; o < OC; ++o) {
                float s = bias ? bias[o] : 0.0f;                      /* :143 */
                for (int layer = 0; layer < layers; ++layer) {        /* :172-187 */
                    float t0[4], t1[4], t2[4], t3[4];
                    deconv_fetch(x, H, W, IC, bx - 1, by - 1, layer, t0); /* texCoord_0 = baseCoord + (-0.5,-0.5): texel baseCoord - 1 */
                    deconv_fetch(x, H, W, IC, bx, by - 1, layer, t1);
                    deconv_fetch(x, H, W, IC, bx - 1, by, layer, t2);
                    deconv_fetch(x, H, W, IC, bx, by, layer, t3);
                    const float* wo = wm + ((long) o * layers + layer) * 16 * 4;
                    float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                    for (int j = 0; j < 4; ++j) {
                        d0 += t0[j] * wo[widx[0] * 4 + j];
                        d1 += t1[j] * wo[widx[1] * 4 + j];
                        d2 += t2[j] * wo[widx[2] * 4 + j];
                        d3 += t3[j] * wo[widx[3] * 4 + j];
                    }
                    s += d0 + d1 + d2 + d3;
                }
                y[((long) gy * OW + gx) * OC + o] = s;
            }
        }
    free(wm);
}

/* shadertemplate_vk_instancenorm.comp:70-160; the sums are accumulated in double here (the shader's 256-thread tree in fp32
 * has no defined order to copy) and rounded to float where the shader stores them */
void snn_oracle_instancenorm(const float* x, int N, int H, int W, int C, const float* beta, const float* gamma, float eps, int act, float leaky,
                             float* y) {
    long HW = (long) H * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* xn = x + (long) n * HW * C + c;
            double sum = 0.0;
            for (long p = 0; p < HW; ++p) sum += xn[p * C];
            float mean = (float) (sum / (double) HW);
            double sq = 0.0;
            for (long p = 0; p < HW; ++p) {
                float dv = xn[p * C] - mean;
                sq += (double) (dv * dv);
            }
            float var = (float) (sq / (double) HW);
            float sigma = sqrtf(var + eps);
            float multiplier = gamma[c] / sigma;
            for (long p = 0; p < HW; ++p) {
                float color = (xn[p * C] - mean) * multiplier + beta[c];
                y[((long) n * HW + p) * C + c] = act_apply(act, leaky, color, 0.0f);
            }
        }
}

static struct {
    uint64_t s[64];
    unsigned i, c;
} g_rng;

static uint64_t rng_next(void) {
    unsigned n, r, idx = 0;
    if (!g_rng.c) { /* exhausted: run forward 55*10-55+1 numbers */
        n = 55 * 10 - 55 + 1;
        g_rng.c = 55 - 1;
    } else {
        n = 1;
        g_rng.c--;
    }
    for (r = 0; r < n; ++r) {
        idx = g_rng.i;
        g_rng.s[idx & 63] = g_rng.s[(idx + 64 - 24) & 63] + g_rng.s[(idx + 64 - 55) & 63];
        g_rng.i = (g_rng.i + 1) & 0xFFFF; /* uint_fast16_t is 64-bit on glibc x86-64; only the low 6 bits matter */
    }
    return g_rng.s[idx & 63];
}

void snn_oracle_srand(uint64_t seed) {
    g_rng.c = 55;
    g_rng.i = 0;
    g_rng.s[0] = seed;
    for (unsigned i = 1; i < 64; ++i) g_rng.s[i] = (uint64_t) i * 2147483647ull + seed;
    for (unsigned i = 0; i < 10000; ++i) rng_next();
}

float snn_oracle_random_float(float a, float b) {
    float random = ((float) rng_next()) / (float) UINT64_MAX;
    float diff = b - a;
    float rd = random * diff;
    return a + rd;
}
