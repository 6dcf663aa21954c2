// adam.hip -- the optimizer step of the loop body as one launch per optimizer (gfx950, HBM-bound:
// 28 bytes per parameter: read p, g, m, v, write p, m, v).
//
// torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False) as the reference runs it
// (models/base_model.py / generator.py optimizer_G, discriminator.py optimizer_D; train_vid2vid.py:93-105):
//     m = beta1*m + (1-beta1)*g;  v = beta2*v + (1-beta2)*g*g
//     p = p - (lr / (1-beta1^t)) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)
// over a device table of tensors {p, g, m, v, n}; `blocks` maps a workgroup to (tensor, chunk).
#include "common.h"

#define ADAM_CHUNK 8192   // elements per workgroup: 256 lanes x 8 float4

struct AdamTensor {
    float *p;
    const float *g;
    float *m;
    float *v;
    long n;
};

struct AdamCoef {
    float step_size, beta1, beta2, omb1, omb2, bc2_sqrt, eps;   // omb = 1 - beta, rounded from double like torch's
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamCoef &k) {
    const float step_size = k.step_size, bc2_sqrt = k.bc2_sqrt, eps = k.eps;
    m = k.beta1 * m + k.omb1 * g;
    v = k.beta2 * v + k.omb2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256)
adam_kernel(const AdamTensor *__restrict__ table, const int2 *__restrict__ blocks, const AdamCoef k) {
    const int2 tb = blocks[blockIdx.x];
    const AdamTensor t = table[tb.x];
    const long e0 = (long)tb.y * ADAM_CHUNK;
    const long e1 = min(t.n, e0 + ADAM_CHUNK);
    const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
    if (vec && e1 - e0 == ADAM_CHUNK) {
        // a whole chunk (all but the last one of a tensor): no per-load bounds, global (not flat) accesses, and the
        // streams that nobody reads again before the next step (the gradient in, the two moments out) bypass the caches
        typedef float vf4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) vf4 gf4;
        gf4 *p4 = (gf4 *)(uintptr_t)t.p, *m4 = (gf4 *)(uintptr_t)t.m, *v4 = (gf4 *)(uintptr_t)t.v;
        const gf4 *g4 = (const gf4 *)(uintptr_t)t.g;
        const long q0 = (e0 >> 2) + threadIdx.x;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            vf4 P[4], G[4], M[4], V[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = q0 + (half * 4 + u) * 256;
                P[u] = p4[q];
                G[u] = __builtin_nontemporal_load(&g4[q]);
                M[u] = m4[q];
                V[u] = v4[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = q0 + (half * 4 + u) * 256;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float pp = P[u][c], mm = M[u][c], vv = V[u][c];
                    adam_one(pp, G[u][c], mm, vv, k);
                    P[u][c] = pp; M[u][c] = mm; V[u][c] = vv;
                }
                p4[q] = P[u];
                __builtin_nontemporal_store(M[u], &m4[q]);
                __builtin_nontemporal_store(V[u], &v4[q]);
            }
        }
    } else if (vec) {
        const long q1 = e1 >> 2;   // whole float4s below e1 (e0 is a multiple of 4)
        float4 *p4 = (float4 *)t.p, *m4 = (float4 *)t.m, *v4 = (float4 *)t.v;
        const float4 *g4 = (const float4 *)t.g;
        // all loads of the chunk first (8 float4 per array per lane would be 128 VGPRs: two halves of 4)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 P[4], G[4], M[4], V[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = (e0 >> 2) + (half * 4 + u) * 256 + threadIdx.x;
                if (q < q1) { P[u] = p4[q]; G[u] = g4[q]; M[u] = m4[q]; V[u] = v4[q]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = (e0 >> 2) + (half * 4 + u) * 256 + threadIdx.x;
                if (q < q1) {
                    adam_one(P[u].x, G[u].x, M[u].x, V[u].x, k);
                    adam_one(P[u].y, G[u].y, M[u].y, V[u].y, k);
                    adam_one(P[u].z, G[u].z, M[u].z, V[u].z, k);
                    adam_one(P[u].w, G[u].w, M[u].w, V[u].w, k);
                    p4[q] = P[u]; m4[q] = M[u]; v4[q] = V[u];
                }
            }
        }
        // tail of the tensor (n % 4 elements) belongs to the last chunk
        for (long e = (q1 << 2) + threadIdx.x; e < e1; e += 256) {
            float p = t.p[e], m = t.m[e], v = t.v[e];
            adam_one(p, t.g[e], m, v, k);
            t.p[e] = p; t.m[e] = m; t.v[e] = v;
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += 256) {
            float p = t.p[e], m = t.m[e], v = t.v[e];
            adam_one(p, t.g[e], m, v, k);
            t.p[e] = p; t.m[e] = m; t.v[e] = v;
        }
    }
}

extern "C" int ir2rgb_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int ir2rgb_adam_step(const void *table, const void *blocks, int nblocks, float lr, float beta1, float beta2,
                                float eps, int step, void *stream) {
    if (!table || !blocks || nblocks < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
        return IR2RGB_EINVAL;
    if (nblocks == 0) return IR2RGB_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamCoef k;
    k.step_size = (float)((double)lr / bc1);
    k.beta1 = beta1;
    k.beta2 = beta2;
    k.omb1 = (float)(1.0 - (double)beta1);
    k.omb2 = (float)(1.0 - (double)beta2);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.eps = eps;
    adam_kernel<<<nblocks, 256, 0, as_stream(stream)>>>((const AdamTensor *)table, (const int2 *)blocks, k);
    return ir2rgb_launch_status();
}
