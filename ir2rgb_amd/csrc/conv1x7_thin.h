// conv1x7_thin.h -- interface of conv1x7_thin.hip towards the convolution dispatcher (conv_mfma.hip)
#pragma once
#include "common.h"

struct T7Geom {
    int N, H, W, Cout;            // Cout <= 32 row responses
    int ldx, ci_off, ldy, co_off; // pixel strides (elements) / first channel of X (half) and Y (fp32)
    int nsx;                      // segments per image row
    long nseg;                    // N * H * nsx
    unsigned x_bytes, w_bytes;
};
bool conv1x7_thin_plan(const ir2rgb_conv_desc *d, T7Geom *g);
int conv1x7_thin_launch(const T7Geom &g, int dtype, int cin, const void *x, const void *wp, void *y, hipStream_t s);
