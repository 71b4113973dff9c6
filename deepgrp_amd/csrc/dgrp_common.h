// Shared helpers of libdeepgrp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/deepgrp_hip.h"

#define DGRP_EXPORT extern "C" __attribute__((visibility("default")))
// classes: the fused kernels lay the logits out as a 16-wide tile; up to DGRP_MAXC classes run on the plain-fp32 kernels (the labels are int8)
#define DGRP_MAXC 64

void dgrp_set_error(const char *fmt, ...);

#define DGRP_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            dgrp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                       \
            return DGRP_EHIP;                                                               \
        }                                                                                   \
    } while (0)

#define DGRP_LAUNCH_CHECK()                                                                 \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            dgrp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),       \
                           __FILE__, __LINE__);                                             \
            return DGRP_EHIP;                                                               \
        }                                                                                   \
    } while (0)

#define DGRP_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            dgrp_set_error(__VA_ARGS__);     \
            return DGRP_EINVAL;              \
        }                                    \
    } while (0)

static inline int64_t dgrp_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Placement of window w in the per-base array (deepgrp/prediction.py:104-105):
// `index = i * batch.shape[0] * step` uses the size of the CURRENT batch.
struct dgrp_placement {
    int64_t nfullB;   // nfull * B : first window of the short last batch
    int64_t shift;    // nfull * r - nfull * B : added to w for windows of the short batch
};
static inline dgrp_placement dgrp_make_placement(int64_t nwin_total, int64_t B)
{
    dgrp_placement p;
    int64_t nfull = nwin_total / B, r = nwin_total % B;
    p.nfullB = nfull * B;
    p.shift = nfull * r - nfull * B;
    return p;
}
__host__ __device__ static inline int64_t dgrp_place_row(dgrp_placement p, int64_t w, int64_t s)
{
    return (w < p.nfullB ? w : w + p.shift) * s;
}
