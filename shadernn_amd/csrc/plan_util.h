// plan_util.h -- helpers shared by the dtype-agnostic (fp32 / fp16) element-wise, pooling and shape plans
#pragma once
#include "snnhip_internal.h"

namespace snnhip {

inline unsigned grid_for(const snnhip_ctx* ctx, size_t items) {
    size_t blocks = (items + 255) / 256;
    const size_t cap = static_cast<size_t>(ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256) * 16;
    if (blocks > cap) blocks = cap;
    return static_cast<unsigned>(blocks ? blocks : 1);
}

inline bool dims_match(const snnhip_tensor* t, int n, int h, int w, int c) { return t->n == n && t->h == h && t->w == w && t->c == c; }

// These plans take their element type from the tensors they are run on (fp32 or fp16, all tensors of one call alike).
#define SNNHIP_SAME_DTYPE(what)                                                                                                   \
    do {                                                                                                                          \
        for (int _i = 0; _i < nIn; ++_i)                                                                                          \
            SNNHIP_REQUIRE(in[_i]->dtype == out->dtype, "%s: input %d has dtype %d, output %d", what, _i, in[_i]->dtype, out->dtype); \
    } while (0)
#define SNNHIP_WITH_T(DT, ...)        \
    do {                              \
        if ((DT) == SNNHIP_F16) {     \
            typedef _Float16 T;       \
            __VA_ARGS__               \
        } else {                      \
            typedef float T;          \
            __VA_ARGS__               \
        }                             \
    } while (0)
template <typename T>
const T* cptr(const snnhip_tensor* t) { return reinterpret_cast<const T*>(t->data); }
template <typename T>
T* mptr(snnhip_tensor* t) { return reinterpret_cast<T*>(t->data); }

} // namespace snnhip
