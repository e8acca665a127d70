// Bandwidth-bound pieces of the InternViT encoder around its GEMMs and its attention, for gfx950:
// LayerNorm, bias + GELU, bias + LayerScale + residual, patch extraction for the 14x14 stride-14
// convolution (done as a GEMM), and the token assembly (class token, convolution bias, position
// embedding, [b, s, h] -> [s, b, h]).  bf16 in / out, fp32 arithmetic, 16-byte accesses.
#include "internal.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---------------------------------------------------------------------------
// LayerNorm over the last dimension: y = (x - mean) / sqrt(var + eps) * gamma + beta.
// One wave per row; a lane holds CH chunks of 8 columns (h = 512 * CH), kept in registers between the
// statistics and the normalisation, so x is read once.
template <int CH>
__global__ __launch_bounds__(256) void lv_layernorm_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y,
                                                            const __bf16* __restrict__ gamma, const __bf16* __restrict__ beta,
                                                            int64_t rows, int h, int64_t x_rs, int64_t y_rs, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const __bf16* xp = x + row * x_rs;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
            bf16x8 t = *reinterpret_cast<const bf16x8*>(xp + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[c][i] = (float)t[i]; s += v[c][i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)h;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)h + eps);
    __bf16* yp = y + row * y_rs;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(gamma + col);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(beta + col);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (__bf16)((v[c][i] - mean) * rstd * (float)g[i] + (float)b[i]);
            *reinterpret_cast<bf16x8*>(yp + col) = o;
        }
    }
}

int lv_launch_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int h, int64_t x_rs,
                        int64_t y_rs, float eps, hipStream_t stream) {
    if (rows == 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int ch = (h + 511) / 512;
#define LV_LN(CH) hipLaunchKernelGGL((lv_layernorm_kernel<CH>), grid, block, 0, stream, (const __bf16*)x, (__bf16*)y, \
                                     (const __bf16*)gamma, (const __bf16*)beta, rows, h, x_rs, y_rs, eps)
    if (ch <= 1) LV_LN(1); else if (ch <= 2) LV_LN(2); else if (ch <= 4) LV_LN(4); else if (ch <= 8) LV_LN(8); else LV_LN(16);
#undef LV_LN
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------
// LayerNorm backward.  Per row (one wave per row, rows taken grid-stride so that a wave keeps running
// column sums in registers):  xhat = (x - mean) * rstd,  g = dy * gamma,
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),   dgamma += dy * xhat,   dbeta += dy.
// The statistics are recomputed from x (read anyway).  Each wave ends by writing its column sums to
// part[wave_id][2][h] (fp32); lv_colsum_kernel adds the partials in a fixed order, so the parameter
// gradients are bitwise reproducible.
template <int CH>
__global__ __launch_bounds__(256) void lv_layernorm_bwd_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                                const __bf16* __restrict__ gamma, __bf16* __restrict__ dx,
                                                                float* __restrict__ part, int64_t rows, int h, int64_t x_rs,
                                                                int64_t dy_rs, int64_t dx_rs, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float gm[CH][8], sg[CH][8], sb[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        bf16x8 g;
        if (col < h) g = *reinterpret_cast<const bf16x8*>(gamma + col);
#pragma unroll
        for (int i = 0; i < 8; ++i) { gm[c][i] = col < h ? (float)g[i] : 0.f; sg[c][i] = 0.f; sb[c][i] = 0.f; }
    }
    const float inv_h = 1.f / (float)h;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const __bf16* xp = x + row * x_rs;
        const __bf16* dp = dy + row * dy_rs;
        float v[CH][8], d[CH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 8;
            if (col < h) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(xp + col);
                const bf16x8 u = *reinterpret_cast<const bf16x8*>(dp + col);
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[c][i] = (float)t[i]; d[c][i] = (float)u[i]; s += v[c][i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[c][i] = 0.f; d[c][i] = 0.f; }
            }
        }
        const float mean = wave_sum(s) * inv_h;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 8;
            if (col < h) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[c][i] -= mean; q += v[c][i] * v[c][i]; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_h + eps);
        float a = 0.f, b = 0.f;   // sum(g), sum(g * xhat)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[c][i] *= rstd;                       // xhat (0 in the padding columns)
                sg[c][i] += d[c][i] * v[c][i];
                sb[c][i] += d[c][i];
                d[c][i] *= gm[c][i];                   // g
                a += d[c][i];
                b += d[c][i] * v[c][i];
            }
        a = wave_sum(a) * inv_h;
        b = wave_sum(b) * inv_h;
        __bf16* op = dx + row * dx_rs;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 8;
            if (col < h) {
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (__bf16)(rstd * (d[c][i] - a - v[c][i] * b));
                *reinterpret_cast<bf16x8*>(op + col) = o;
            }
        }
    }
    float* pg = part + wave * 2 * (int64_t)h;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                *reinterpret_cast<f32x4*>(pg + col + i) = f32x4{sg[c][i], sg[c][i + 1], sg[c][i + 2], sg[c][i + 3]};
                *reinterpret_cast<f32x4*>(pg + h + col + i) = f32x4{sb[c][i], sb[c][i + 1], sb[c][i + 2], sb[c][i + 3]};
            }
        }
    }
}

// out[j] = sum over p of part[p][j], j < width; one lane per column, partials added in index order.
__global__ __launch_bounds__(256) void lv_colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int width) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= width) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * width + j];
    out[j] = s;
}

int lv_layernorm_bwd_parts(int64_t rows) {   // number of partial column-sum rows the backward writes
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    return (int)blocks * 4;
}

int lv_launch_layernorm_bwd(const void* x, const void* dy, const void* gamma, void* dx, float* dgamma_dbeta, float* part,
                            int64_t rows, int h, int64_t x_rs, int64_t dy_rs, int64_t dx_rs, float eps, hipStream_t stream) {
    const int nparts = lv_layernorm_bwd_parts(rows);
    dim3 grid((unsigned)(nparts / 4)), block(256);
    const int ch = (h + 511) / 512;
#define LV_LNB(CH) hipLaunchKernelGGL((lv_layernorm_bwd_kernel<CH>), grid, block, 0, stream, (const __bf16*)x, (const __bf16*)dy, \
                                      (const __bf16*)gamma, (__bf16*)dx, part, rows, h, x_rs, dy_rs, dx_rs, eps)
    if (ch <= 1) LV_LNB(1); else if (ch <= 2) LV_LNB(2); else if (ch <= 4) LV_LNB(4); else LV_LNB(8);
#undef LV_LNB
    hipLaunchKernelGGL(lv_colsum_kernel, dim3((2 * h + 255) / 256), dim3(256), 0, stream, (const float*)part, dgamma_dbeta, nparts, 2 * h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------
// y = gelu(x + bias), the exact (erf) form torch.nn.functional.gelu defaults to.
__global__ __launch_bounds__(256) void lv_bias_gelu_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ bias,
                                                            __bf16* __restrict__ y, int64_t rows, int h) {
    const int cpr = h / 8;
    const int64_t total = rows * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % cpr) * 8;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + idx * 8);
        bf16x8 b;
        if (bias) b = *reinterpret_cast<const bf16x8*>(bias + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = (float)t[i] + (bias ? (float)b[i] : 0.f);
            o[i] = (__bf16)(0.5f * v * (1.f + erff(v * 0.70710678118654752f)));
        }
        *reinterpret_cast<bf16x8*>(y + idx * 8) = o;
    }
}

// dx = dy * gelu'(x + bias),  gelu'(v) = Phi(v) + v * phi(v)
__global__ __launch_bounds__(256) void lv_bias_gelu_bwd_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ bias,
                                                                const __bf16* __restrict__ dy, __bf16* __restrict__ dx, int64_t rows,
                                                                int h) {
    const int cpr = h / 8;
    const int64_t total = rows * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % cpr) * 8;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + idx * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + idx * 8);
        bf16x8 b;
        if (bias) b = *reinterpret_cast<const bf16x8*>(bias + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = (float)t[i] + (bias ? (float)b[i] : 0.f);
            const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
            o[i] = (__bf16)((float)g[i] * (cdf + v * pdf));
        }
        *reinterpret_cast<bf16x8*>(dx + idx * 8) = o;
    }
}

// out = residual + (x + bias) * ls   (LayerScale; bias may be null)
__global__ __launch_bounds__(256) void lv_bias_scale_residual_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ bias,
                                                                      const __bf16* __restrict__ ls, const __bf16* __restrict__ res,
                                                                      __bf16* __restrict__ out, int64_t rows, int h) {
    const int cpr = h / 8;
    const int64_t total = rows * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % cpr) * 8;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + idx * 8);
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(res + idx * 8);
        const bf16x8 s = *reinterpret_cast<const bf16x8*>(ls + col);
        bf16x8 b;
        if (bias) b = *reinterpret_cast<const bf16x8*>(bias + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = (float)t[i] + (bias ? (float)b[i] : 0.f);
            o[i] = (__bf16)((float)r[i] + v * (float)s[i]);
        }
        *reinterpret_cast<bf16x8*>(out + idx * 8) = o;
    }
}

// x1 = residual + (x + bias) * ls  AND  y = LayerNorm(x1) in one pass (intern_vit_model.py:59-66 and :74-77 followed by
// the next sub-block's norm): one wave per row keeps the row in registers, so x1 is written but never read back.
// x1 is rounded to bf16 before the statistics, exactly what the separate LayerNorm pass would read: results are
// bit-identical to lv_bias_scale_residual followed by lv_layernorm.
template <int CH>
__global__ __launch_bounds__(256) void lv_bsr_layernorm_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ bias,
                                                                const __bf16* __restrict__ ls, const __bf16* __restrict__ res,
                                                                __bf16* __restrict__ x1, __bf16* __restrict__ y,
                                                                const __bf16* __restrict__ gamma, const __bf16* __restrict__ beta,
                                                                int64_t rows, int h, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const __bf16* xp = x + row * h;
    const __bf16* rp = res + row * h;
    __bf16* op = x1 + row * h;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(xp + col);
            const bf16x8 r = *reinterpret_cast<const bf16x8*>(rp + col);
            const bf16x8 sc = *reinterpret_cast<const bf16x8*>(ls + col);
            bf16x8 b;
            if (bias) b = *reinterpret_cast<const bf16x8*>(bias + col);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float u = (float)t[i] + (bias ? (float)b[i] : 0.f);
                o[i] = (__bf16)((float)r[i] + u * (float)sc[i]);
                v[c][i] = (float)o[i];
                s += v[c][i];
            }
            *reinterpret_cast<bf16x8*>(op + col) = o;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)h;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)h + eps);
    __bf16* yp = y + row * h;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < h) {
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(gamma + col);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(beta + col);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (__bf16)((v[c][i] - mean) * rstd * (float)g[i] + (float)b[i]);
            *reinterpret_cast<bf16x8*>(yp + col) = o;
        }
    }
}

int lv_launch_bsr_layernorm(const void* x, const void* bias, const void* ls, const void* res, void* x1, void* y, const void* gamma,
                            const void* beta, int64_t rows, int h, float eps, hipStream_t stream) {
    if (rows == 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int ch = (h + 511) / 512;
#define LV_BL(CH) hipLaunchKernelGGL((lv_bsr_layernorm_kernel<CH>), grid, block, 0, stream, (const __bf16*)x, (const __bf16*)bias, \
                                     (const __bf16*)ls, (const __bf16*)res, (__bf16*)x1, (__bf16*)y, (const __bf16*)gamma,          \
                                     (const __bf16*)beta, rows, h, eps)
    if (ch <= 1) LV_BL(1); else if (ch <= 2) LV_BL(2); else if (ch <= 4) LV_BL(4); else if (ch <= 8) LV_BL(8); else LV_BL(16);
#undef LV_BL
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static unsigned stride_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (unsigned)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

int lv_launch_bias_gelu(const void* x, const void* bias, void* y, int64_t rows, int h, hipStream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lv_bias_gelu_kernel, dim3(stride_grid(rows * (h / 8))), dim3(256), 0, stream, (const __bf16*)x,
                       (const __bf16*)bias, (__bf16*)y, rows, h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lv_launch_bias_gelu_bwd(const void* x, const void* bias, const void* dy, void* dx, int64_t rows, int h, hipStream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lv_bias_gelu_bwd_kernel, dim3(stride_grid(rows * (h / 8))), dim3(256), 0, stream, (const __bf16*)x,
                       (const __bf16*)bias, (const __bf16*)dy, (__bf16*)dx, rows, h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lv_launch_bias_scale_residual(const void* x, const void* bias, const void* ls, const void* res, void* out, int64_t rows, int h,
                                  hipStream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lv_bias_scale_residual_kernel, dim3(stride_grid(rows * (h / 8))), dim3(256), 0, stream, (const __bf16*)x,
                       (const __bf16*)bias, (const __bf16*)ls, (const __bf16*)res, (__bf16*)out, rows, h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------
// Patch extraction: img [n, c, H, W] bf16 -> patches [n * gh * gw, kpad] with column
// (ch * p + ky) * p + kx  (the order of a Conv2d weight [out, c, p, p] flattened), zero-padded to kpad.
__global__ __launch_bounds__(256) void lv_patchify_kernel(const __bf16* __restrict__ img, __bf16* __restrict__ out, int n, int c,
                                                           int H, int W, int p, int kpad) {
    const int gh = H / p, gw = W / p;
    const int64_t total = (int64_t)n * gh * gw * kpad;
    const int k = c * p * p;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % kpad);
        const int64_t tok = idx / kpad;
        __bf16 v = (__bf16)0.f;
        if (col < k) {
            const int kx = col % p, ky = (col / p) % p, ch = col / (p * p);
            const int px = (int)(tok % gw), py = (int)((tok / gw) % gh);
            const int64_t im = tok / ((int64_t)gw * gh);
            v = img[((im * c + ch) * H + (py * p + ky)) * (int64_t)W + px * p + kx];
        }
        out[idx] = v;
    }
}

int lv_launch_patchify(const void* img, void* out, int n, int c, int H, int W, int p, int kpad, hipStream_t stream) {
    const int64_t total = (int64_t)n * (H / p) * (W / p) * kpad;
    if (total == 0) return 0;
    hipLaunchKernelGGL(lv_patchify_kernel, dim3(stride_grid(total)), dim3(256), 0, stream, (const __bf16*)img, (__bf16*)out, n, c, H, W,
                       p, kpad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Token assembly: tokens [s, n, h] (sequence-major) from the patch GEMM output proj [n, np, h]:
//   s < ncls : class_token[s] + pos[s]         s >= ncls : proj[n][s - ncls] + conv_bias + pos[s]
__global__ __launch_bounds__(256) void lv_embed_tokens_kernel(const __bf16* __restrict__ proj, const __bf16* __restrict__ conv_bias,
                                                               const __bf16* __restrict__ cls, const __bf16* __restrict__ pos,
                                                               __bf16* __restrict__ out, int n, int np, int ncls, int h) {
    const int cpr = h / 8;
    const int s_len = np + ncls;
    const int64_t total = (int64_t)s_len * n * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % cpr) * 8;
        const int64_t sn = idx / cpr;
        const int im = (int)(sn % n);
        const int s = (int)(sn / n);
        const bf16x8 pe = *reinterpret_cast<const bf16x8*>(pos + (int64_t)s * h + col);
        bf16x8 o;
        if (s < ncls) {
            const bf16x8 ct = *reinterpret_cast<const bf16x8*>(cls + (int64_t)s * h + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (__bf16)((float)ct[i] + (float)pe[i]);
        } else {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(proj + ((int64_t)im * np + (s - ncls)) * h + col);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(conv_bias + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (__bf16)((float)t[i] + (float)b[i] + (float)pe[i]);
        }
        *reinterpret_cast<bf16x8*>(out + idx * 8) = o;
    }
}

int lv_launch_embed_tokens(const void* proj, const void* conv_bias, const void* cls, const void* pos, void* out, int n, int np,
                           int ncls, int h, hipStream_t stream) {
    const int64_t total = (int64_t)(np + ncls) * n * (h / 8);
    if (total == 0) return 0;
    hipLaunchKernelGGL(lv_embed_tokens_kernel, dim3(stride_grid(total)), dim3(256), 0, stream, (const __bf16*)proj,
                       (const __bf16*)conv_bias, (const __bf16*)cls, (const __bf16*)pos, (__bf16*)out, n, np, ncls, h);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
