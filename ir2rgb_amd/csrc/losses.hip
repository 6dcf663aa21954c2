// losses.hip -- the scalar losses of the vid2vid loop as ONE launch per group (gfx950, HBM-bound).
//
// The reference evaluates every discriminator-feature L1 term and every least-squares GAN term as its
// own chain of torch elementwise + reduction kernels (models/networks.py GANLoss, discriminator.py
// compute_loss_D: criterionFeat / criterionGAN) -- ~10 tensor passes and ~10 launches per term
// forward + backward.  Here a group of up to 32 terms is one kernel: every block owns a contiguous
// range of one term, sums it in fp32 and writes one partial; a second tiny kernel adds the partials
// in block order (deterministic) into up to 4 output scalars.  The backward is one launch as well.
//
//   kind 0 (L1, half operands)   : out[slot] += weight * mean |a - b|        ga = sign(a-b) * g * weight / n
//   kind 1 (MSE, fp32 operand)   : out[slot] += weight * mean (a - target)^2  ga = 2 (a-target) * g * weight / n
//   kind 2 (masked L1, fp32 NCHW): out[slot] += weight * mean |a*m - b*m|,  m = mask[n][0][h][w]
//                                   ga = sign(a*m - b*m) * m * g * weight / n   (b may be NULL = zeros)
#include "common.h"

#define LOSS_BLOCKS 512

struct LossArgs {
    ir2rgb_loss_item it[IR2RGB_LOSS_MAX_ITEMS];
    int blk0[IR2RGB_LOSS_MAX_ITEMS + 1];
    int count;
};

static __device__ __forceinline__ float lh2f(uint16_t h, int dt) {
    if (dt == IR2RGB_BF16) return __uint_as_float(((uint32_t)h) << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
static __device__ __forceinline__ uint16_t lf2h(float f, int dt) {
    if (dt == IR2RGB_BF16) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

__device__ __forceinline__ int loss_find_item(const LossArgs &A, int blk) {
    int i = 0;
    while (i + 1 < A.count && blk >= A.blk0[i + 1]) ++i;
    return i;
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
loss_multi_fwd_kernel(const LossArgs A, int dt, float *__restrict__ partial) {
    __shared__ float red[4];
    const int i = loss_find_item(A, blockIdx.x);
    const ir2rgb_loss_item it = A.it[i];
    const int nb = A.blk0[i + 1] - A.blk0[i], lb = blockIdx.x - A.blk0[i];
    float s = 0.f;
    if (it.kind == 0) {
        const long n8 = it.n >> 3;
        const long per = (n8 + nb - 1) / nb, q0 = lb * per, q1 = min(n8, q0 + per);
        const uint4 *a = (const uint4 *)it.a, *b = (const uint4 *)it.b;
        // four operand pairs in flight per lane (one pair per trip keeps the whole launch near 2 TB/s: 131 k lanes
        // x 32 B per ~2 us round trip); loads from clamped indices, masked afterwards; summed in index order
        for (long q = q0 + threadIdx.x; q < q1; q += 1024) {
            uint4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long qq = min(q + 256 * u, q1 - 1);
                va[u] = a[qq];
                vb[u] = b[qq];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t += fabsf(lh2f((uint16_t)(wa[j] & 0xffff), dt) - lh2f((uint16_t)(wb[j] & 0xffff), dt));
                    t += fabsf(lh2f((uint16_t)(wa[j] >> 16), dt) - lh2f((uint16_t)(wb[j] >> 16), dt));
                }
                s += q + 256 * u < q1 ? t : 0.f;
            }
        }
    } else {
        const long per = (it.n + nb - 1) / nb, q0 = lb * per, q1 = min(it.n, q0 + per);
        const float *a = (const float *)it.a, *b = (const float *)it.b, *m = (const float *)it.mask;
        for (long q = q0 + threadIdx.x; q < q1; q += 256) {
            if (it.kind == 1) {
                const float d = a[q] - it.target;
                s += d * d;
            } else {
                const long img = q / it.chw, hw = (q - img * it.chw) % it.hw;
                const float mv = m[img * it.hw + hw];
                s += fabsf(a[q] * mv - (b ? b[q] * mv : 0.f));
            }
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s * (it.weight / (float)it.n);
}

// out[slot] = sum of the partials of that slot's blocks, in block order (thread t takes blocks
// t, t+256, ... then a fixed LDS tree).
__global__ void __launch_bounds__(256)
loss_multi_finish_kernel(const LossArgs A, const float *__restrict__ partial, int nblocks, float *__restrict__ out,
                         int nslots) {
    __shared__ float red[4][256];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const int slot = A.it[loss_find_item(A, b)].slot;
        const float v = partial[b];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += slot == k ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x < nslots) out[threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(256)
loss_multi_bwd_kernel(const LossArgs A, int dt, const float *__restrict__ gout) {
    const int i = loss_find_item(A, blockIdx.x);
    const ir2rgb_loss_item it = A.it[i];
    if (!it.ga) return;
    const int nb = A.blk0[i + 1] - A.blk0[i], lb = blockIdx.x - A.blk0[i];
    const float g = gout[it.slot] * (it.weight / (float)it.n);
    if (it.kind == 0) {
        const long n8 = it.n >> 3;
        const long per = (n8 + nb - 1) / nb, q0 = lb * per, q1 = min(n8, q0 + per);
        const uint4 *a = (const uint4 *)it.a, *b = (const uint4 *)it.b;
        uint4 *ga = (uint4 *)it.ga;
        const uint32_t gp = lf2h(g, dt), gn = lf2h(-g, dt);
        for (long q = q0 + threadIdx.x; q < q1; q += 1024) {      // four operand pairs in flight per lane (see the forward)
            uint4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long qq = min(q + 256 * u, q1 - 1);
                va[u] = a[qq];
                vb[u] = b[qq];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d0 = lh2f((uint16_t)(wa[j] & 0xffff), dt) - lh2f((uint16_t)(wb[j] & 0xffff), dt);
                    const float d1 = lh2f((uint16_t)(wa[j] >> 16), dt) - lh2f((uint16_t)(wb[j] >> 16), dt);
                    const uint32_t lo = d0 > 0.f ? gp : (d0 < 0.f ? gn : 0u), hi = d1 > 0.f ? gp : (d1 < 0.f ? gn : 0u);
                    o[j] = lo | (hi << 16);
                }
                if (q + 256 * u < q1) ga[q + 256 * u] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        const long per = (it.n + nb - 1) / nb, q0 = lb * per, q1 = min(it.n, q0 + per);
        const float *a = (const float *)it.a, *b = (const float *)it.b, *m = (const float *)it.mask;
        float *ga = (float *)it.ga;
        for (long q = q0 + threadIdx.x; q < q1; q += 256) {
            if (it.kind == 1) {
                ga[q] = 2.f * (a[q] - it.target) * g;
            } else {
                const long img = q / it.chw, hw = (q - img * it.chw) % it.hw;
                const float mv = m[img * it.hw + hw];
                const float d = a[q] * mv - (b ? b[q] * mv : 0.f);
                ga[q] = (d > 0.f ? g : (d < 0.f ? -g : 0.f)) * mv;
            }
        }
    }
}

static int loss_pack(const ir2rgb_loss_item *items, int count, int dtype, LossArgs &A, int &nslots) {
    if (!items || count < 1 || count > IR2RGB_LOSS_MAX_ITEMS) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    double total = 0;
    nslots = 0;
    for (int i = 0; i < count; ++i) {
        const ir2rgb_loss_item &it = items[i];
        if (!it.a || it.n < 1 || it.slot < 0 || it.slot > 3 || it.kind < 0 || it.kind > 2) return IR2RGB_EINVAL;
        if (it.kind == 0 && (!it.b || (it.n & 7))) return IR2RGB_EINVAL;
        if (it.kind == 0 && (((uintptr_t)it.a | (uintptr_t)it.b | (uintptr_t)it.ga) & 15)) return IR2RGB_EALIGN;
        if (it.kind == 2 && (!it.mask || it.hw < 1 || it.chw < it.hw || it.chw % it.hw || it.n % it.chw)) return IR2RGB_EINVAL;
        total += (double)it.n;
        nslots = it.slot + 1 > nslots ? it.slot + 1 : nslots;
        A.it[i] = it;
    }
    A.count = count;
    int used = 0;
    for (int i = 0; i < count; ++i) {
        // blocks in proportion to the element count; at least 1, at most one block per 2048 elements
        long nb = (long)((double)(LOSS_BLOCKS - count) * (double)items[i].n / total) + 1;
        const long cap = (items[i].n + 2047) / 2048;
        nb = nb > cap ? cap : nb;
        A.blk0[i] = used;
        used += (int)nb;
    }
    A.blk0[count] = used;
    return IR2RGB_OK;
}

extern "C" int ir2rgb_loss_partial_elems(void) { return LOSS_BLOCKS; }

extern "C" int ir2rgb_loss_multi_fwd(const ir2rgb_loss_item *items, int count, int dtype, float *partial, float *out,
                                     void *stream) {
    LossArgs A;
    int nslots;
    int rc = loss_pack(items, count, dtype, A, nslots);
    if (rc != IR2RGB_OK) return rc;
    if (!partial || !out) return IR2RGB_EINVAL;
    hipStream_t s = as_stream(stream);
    const int nblocks = A.blk0[count];
    loss_multi_fwd_kernel<<<nblocks, 256, 0, s>>>(A, dtype, partial);
    loss_multi_finish_kernel<<<1, 256, 0, s>>>(A, partial, nblocks, out, nslots);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_loss_multi_bwd(const ir2rgb_loss_item *items, int count, int dtype, const float *gout,
                                     void *stream) {
    LossArgs A;
    int nslots;
    int rc = loss_pack(items, count, dtype, A, nslots);
    if (rc != IR2RGB_OK) return rc;
    if (!gout) return IR2RGB_EINVAL;
    loss_multi_bwd_kernel<<<A.blk0[count], 256, 0, as_stream(stream)>>>(A, dtype, gout);
    return ir2rgb_launch_status();
}
