// conv7x1_col.h -- interface of conv7x1_col.hip towards the convolution dispatcher (conv_mfma.hip)
#pragma once
#include "common.h"

struct C7Geom {
    int N, H, W, Cout;            // 64 -> Cout (64 | 128) channels, 7 x 1 taps, reflection padding 3 in y
    int ldx, ci_off, ldy, co_off; // pixel strides (elements) / first channel of X and Y (half)
    int nty, ntx;                 // 8 x 32 pixel tiles per image
    unsigned x_bytes, y_bytes, w_bytes;
};
bool conv7x1_col_plan(const ir2rgb_conv_desc *d, C7Geom *g);
int conv7x1_col_tiles(const C7Geom &g);
int conv7x1_col_launch(const C7Geom &g, int dtype, const void *x, const void *wp, const float *bias, void *y, float *stats,
                       hipStream_t s);
