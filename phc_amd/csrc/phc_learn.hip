// phc_learn.hip -- learner-side kernels (P1 observation normaliser).
//
// The PPO update calls the running normaliser four times per optimizer step (obs + three AMP batches); as torch ops that is
// ~40 launches per call (normalise, batch mean / var, the fp64 moment update on D-element vectors): 85 ms of a 250 ms update
// at 4096 envs (scripts/profile_step.py).  Here it is one bandwidth-bound pass + one tiny finishing block.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>
#include "../../include/phc_amd.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

#define RN_COLS 256   // columns per block == threads per block (thread <-> column: loads coalesce across the block)
#define RN_ROWS 32    // rows per block (4096 x 1960 -> 1024 blocks: four per CU; 128 rows left one wavefront per SIMD, latency-bound)
#define RN_UNROLL 8   // independent loads in flight per thread

// One pass over x [rows, cols]: y = clamp((x - mean) / sqrt(var + eps), -c, c)  (running_mean_std.py:95-96, fp32 like the reference:
// the fp64 statistics are rounded to fp32 first) and, when `partial` is given, per-block column sums of x and x^2 in fp64.
template <bool BF16>
__global__ __launch_bounds__(RN_COLS) void k_running_norm(const float* __restrict__ x, const int64_t* __restrict__ idx, int64_t rows, int cols,
                                                          const double* __restrict__ mean, const double* __restrict__ var, float eps, float clampv,
                                                          void* __restrict__ out, int out_stride, double* __restrict__ partial) {
    const int c = blockIdx.x * RN_COLS + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * RN_ROWS;
    const int64_t r1 = r0 + RN_ROWS < rows ? r0 + RN_ROWS : rows;
    const float m = (float)mean[c];
    const float s = sqrtf((float)var[c] + eps);
    double sum = 0.0, sq = 0.0;
    auto one = [&](int64_t r, float v) {
        if (partial) { sum += (double)v; sq += (double)v * (double)v; }
        if (out) {
            const float t = (v - m) / s;
            float y = fminf(fmaxf(t, -clampv), clampv);
            if (t != t) y = t;   // torch.clamp propagates NaN; fminf / fmaxf do not
            if (BF16) reinterpret_cast<__hip_bfloat16*>(out)[r * out_stride + c] = __float2bfloat16(y);
            else reinterpret_cast<float*>(out)[r * out_stride + c] = y;
        }
    };
    int64_t r = r0;
    for (; r + RN_UNROLL <= r1; r += RN_UNROLL) {
        float v[RN_UNROLL];
#pragma unroll
        for (int k = 0; k < RN_UNROLL; ++k) v[k] = x[(idx ? idx[r + k] : r + k) * cols + c];   // minibatch row r = dataset row idx[r]
#pragma unroll
        for (int k = 0; k < RN_UNROLL; ++k) one(r + k, v[k]);
    }
    for (; r < r1; ++r) one(r, x[(idx ? idx[r] : r) * cols + c]);
    if (partial) {
        partial[((int64_t)blockIdx.y * 2 + 0) * cols + c] = sum;
        partial[((int64_t)blockIdx.y * 2 + 1) * cols + c] = sq;
    }
}

// Batch moments from the partial sums and the parallel-variance update of the running statistics
// (running_mean_std.py:56-67,100-104).  Block = RNF_COLS columns x RNF_SLICES slices of the chunk list.  Every block reads the old count; the
// block that finishes LAST (a ticket counter behind the partial sums, left at zero again) writes the new one.
// (round 6: 16 columns x 64 slices instead of 64 x 16 -- the 934-column observation batch of an optimizer step has 512 chunks: 15 blocks walked them in eight
// dependent rounds of loads, 29 us at the head of the policy pass; 59 blocks need one round.)
#define RNF_COLS 16
#define RNF_SLICES 64
__global__ __launch_bounds__(1024) void k_running_norm_finish(const double* __restrict__ partial, int nchunks, int64_t rows, int cols,
                                                              double* __restrict__ run_mean, double* __restrict__ run_var,
                                                              double* __restrict__ run_count, unsigned int* __restrict__ ticket) {
    __shared__ double ls[RNF_SLICES][RNF_COLS], lq[RNF_SLICES][RNF_COLS];
    const int cx = threadIdx.x & (RNF_COLS - 1), sy = threadIdx.x / RNF_COLS;
    const int c = blockIdx.x * RNF_COLS + cx;
    double s = 0.0, q = 0.0;
    if (c < cols) {
        // eight independent loads in flight per thread
        int k = sy;
        double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
        for (; k + 3 * RNF_SLICES < nchunks; k += 4 * RNF_SLICES) {
            const double a0 = partial[((int64_t)k * 2 + 0) * cols + c], b0 = partial[((int64_t)k * 2 + 1) * cols + c];
            const double a1 = partial[((int64_t)(k + RNF_SLICES) * 2 + 0) * cols + c], b1 = partial[((int64_t)(k + RNF_SLICES) * 2 + 1) * cols + c];
            const double a2 = partial[((int64_t)(k + 2 * RNF_SLICES) * 2 + 0) * cols + c], b2 = partial[((int64_t)(k + 2 * RNF_SLICES) * 2 + 1) * cols + c];
            const double a3 = partial[((int64_t)(k + 3 * RNF_SLICES) * 2 + 0) * cols + c], b3 = partial[((int64_t)(k + 3 * RNF_SLICES) * 2 + 1) * cols + c];
            s += a0; q += b0; s1 += a1; q1 += b1; s2 += a2; q2 += b2; s3 += a3; q3 += b3;
        }
        for (; k < nchunks; k += RNF_SLICES) {
            s += partial[((int64_t)k * 2 + 0) * cols + c];
            q += partial[((int64_t)k * 2 + 1) * cols + c];
        }
        s = (s + s1) + (s2 + s3); q = (q + q1) + (q2 + q3);
    }
    ls[sy][cx] = s; lq[sy][cx] = q;
    const double count = *run_count;
    const double n = (double)rows;
    const double tot = count + n;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) { *run_count = tot; *ticket = 0u; }   // all blocks have read the old count by now
    }
    // tree over the slices: 64 -> 4 per column, then the column's first lane adds the four
    for (int h = RNF_SLICES / 2; h >= 4; h >>= 1) {
        if (sy < h) { ls[sy][cx] += ls[sy + h][cx]; lq[sy][cx] += lq[sy + h][cx]; }
        __syncthreads();
    }
    if (sy != 0 || c >= cols) return;
    s = (ls[0][cx] + ls[1][cx]) + (ls[2][cx] + ls[3][cx]);
    q = (lq[0][cx] + lq[1][cx]) + (lq[2][cx] + lq[3][cx]);
    // input.mean(0), input.var(0) are fp32 tensors in the reference: round the batch moments to fp32 before the fp64 update
    const double bm = (double)(float)(s / n);
    const double bv = (double)(float)((q - s * s / n) / (n - 1.0));
    const double mean = run_mean[c], var = run_var[c];
    const double delta = bm - mean;
    run_mean[c] = mean + delta * n / tot;
    run_var[c] = (var * count + bv * n + delta * delta * count * n / tot) / tot;
}

// ------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix [rows, cols] -> fp32 [cols]: the bias gradient of a linear layer (torch's generic reduce takes
// 25 us for 16384 x 1024 and 95 us for 16384 x 69).  Two deterministic stages: 64 columns x 4 row lanes per block over a
// 256-row chunk, then one sum over the chunks.
// ------------------------------------------------------------------------------------------
#define CS_ROWS 256
__global__ __launch_bounds__(256) void k_colsum_bf16(const __hip_bfloat16* __restrict__ x, int64_t rows, int cols, float* __restrict__ partial) {
    __shared__ float l[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float a = 0.f;
    if (c < cols) {
        int64_t r = r0 + ry;
        for (; r + 28 < r1; r += 32) {   // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __bfloat162float(x[(r + 4 * k) * cols + c]);
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; r < r1; r += 4) a += __bfloat162float(x[r * cols + c]);
    }
    l[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = (l[0][cx] + l[1][cx]) + (l[2][cx] + l[3][cx]);
}
// The same with the ReLU mask of a saved layer output applied on the way: gm = (y > 0) ? gy : 0 is written back, its column sums go to `partial`.
__global__ __launch_bounds__(256) void k_colsum_relu_bf16(const __hip_bfloat16* __restrict__ gy, const __hip_bfloat16* __restrict__ y, int64_t rows, int cols,
                                                          __hip_bfloat16* __restrict__ gm, float* __restrict__ partial) {
    __shared__ float l[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float a = 0.f;
    if (c < cols) {
        int64_t r = r0 + ry;
        for (; r + 28 < r1; r += 32) {   // 16 independent loads in flight
            float g[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { g[k] = __bfloat162float(gy[(r + 4 * k) * cols + c]); o[k] = __bfloat162float(y[(r + 4 * k) * cols + c]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float m = o[k] > 0.f ? g[k] : 0.f;
                gm[(r + 4 * k) * cols + c] = __float2bfloat16(m);
                a += m;
            }
        }
        for (; r < r1; r += 4) {
            const float m = __bfloat162float(y[r * cols + c]) > 0.f ? __bfloat162float(gy[r * cols + c]) : 0.f;
            gm[r * cols + c] = __float2bfloat16(m);
            a += m;
        }
    }
    l[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = (l[0][cx] + l[1][cx]) + (l[2][cx] + l[3][cx]);
}
// block = 64 columns x 16 slices of the chunk list (up to 1024 chunks: 64 loads per thread, four in flight)
__device__ __forceinline__ void colsum_finish_block(const float* __restrict__ partial, int nchunks, int cols, float* __restrict__ out, int accumulate, float (*l)[64]) {
    const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int k = sy;
        for (; k + 48 < nchunks; k += 64) {
            a0 += partial[(int64_t)k * cols + c]; a1 += partial[(int64_t)(k + 16) * cols + c];
            a2 += partial[(int64_t)(k + 32) * cols + c]; a3 += partial[(int64_t)(k + 48) * cols + c];
        }
        for (; k < nchunks; k += 16) a0 += partial[(int64_t)k * cols + c];
    }
    l[sy][cx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sy == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += l[i][cx];
        out[c] = accumulate ? out[c] + t : t;
    }
}
__global__ __launch_bounds__(1024) void k_colsum_finish(const float* __restrict__ partial, int nchunks, int cols, float* __restrict__ out) {
    __shared__ float l[16][64];
    colsum_finish_block(partial, nchunks, cols, out, 0, l);
}
// Round 5: the second stage of SEVERAL column sums in one launch (grid.y = job).  The bias gradients of a backward pass are read by nobody before
// clip + Adam, so the layers only run their first stage and the pass ends with one of these instead of a ~5 us dependent launch per layer.
struct ColsumJobs { phc_colsum_job_t job[PHC_COLSUM_MAX_JOBS]; };
__global__ __launch_bounds__(1024) void k_colsum_finish_batch(ColsumJobs jobs) {
    __shared__ float l[16][64];
    const phc_colsum_job_t j = jobs.job[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.cols) return;
    colsum_finish_block(j.partial, j.nchunks, j.cols, j.out, j.accumulate, l);
}

// ------------------------------------------------------------------------------------------
// Linear layer with ONE output (the value head, 512 -> 1): as library GEMMs its forward, input gradient and weight gradient are
// three 1-row / 1-column problems of 40-50 us each; they are a dot product per row, a scaled copy and a weighted column sum.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linear1_fwd(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                     const __hip_bfloat16* __restrict__ b, int64_t rows, int cols, __hip_bfloat16* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float a = 0.f;
    for (int k = lane; k < cols; k += 64) a += __bfloat162float(x[r * cols + k]) * __bfloat162float(w[k]);
    a = wave_sum(a);
    if (lane == 0) y[r] = __float2bfloat16(a + __bfloat162float(b[0]));
}
#define L1_ROWS 16   // rows per block: 16 384 rows -> 1024 blocks (64 rows per block ran 256 latency-bound blocks: 69 us)
// gx[r, k] = gy[r] w[k] (optional) and per-chunk partial sums of gy[r] x[r, k] (columns 0..cols-1) and gy[r] (column `cols`)
__global__ __launch_bounds__(256) void k_linear1_bwd(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                     const __hip_bfloat16* __restrict__ gy, int64_t rows, int cols,
                                                     __hip_bfloat16* __restrict__ gx, float* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * L1_ROWS;
    const int nr = (int)(r0 + L1_ROWS < rows ? L1_ROWS : rows - r0);
    float g[L1_ROWS];
#pragma unroll
    for (int i = 0; i < L1_ROWS; ++i) g[i] = i < nr ? __bfloat162float(gy[r0 + i]) : 0.f;
    for (int k = threadIdx.x; k < cols; k += 256) {
        const float wk = __bfloat162float(w[k]);
        float xv[L1_ROWS];
#pragma unroll
        for (int i = 0; i < L1_ROWS; ++i) xv[i] = i < nr ? __bfloat162float(x[(r0 + i) * cols + k]) : 0.f;   // all loads in flight
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < L1_ROWS; ++i) {
            a += g[i] * xv[i];
            if (gx && i < nr) gx[(r0 + i) * cols + k] = __float2bfloat16(g[i] * wk);
        }
        partial[(int64_t)blockIdx.x * (cols + 1) + k] = a;
    }
    if (threadIdx.x == 0) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < L1_ROWS; ++i) s2 += g[i];
        partial[(int64_t)blockIdx.x * (cols + 1) + cols] = s2;
    }
}

// ------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on the flat fp32 parameter (FlatGradBucket): torch.nn.utils.clip_grad_norm_ (coefficient
// min(1, max_norm / (|g| + 1e-6))) followed by torch.optim.Adam's update (L2 weight decay, bias corrections, eps outside the
// square root of the corrected second moment) -- two launches over 4 arrays instead of norm + scale + fused multi-tensor Adam.
// ------------------------------------------------------------------------------------------
#define AD_BLOCK 256
#define AD_PER_THREAD 8
__global__ __launch_bounds__(AD_BLOCK) void k_sumsq(const float* __restrict__ g, int64_t n, double* __restrict__ partial, int64_t* __restrict__ step_dev) {
    __shared__ double l[AD_BLOCK / 64];
    if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) *step_dev += 1;   // device-resident step count (graph replays): read by k_adam
    const int64_t base = ((int64_t)blockIdx.x * AD_BLOCK + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * AD_BLOCK * 4;
    float a = 0.f;
    for (int64_t i = base; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int64_t k = i; k < n; ++k) a += g[k] * g[k];
        }
    }
    double d = (double)a;
    for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
    if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < AD_BLOCK / 64; ++k) t += l[k];
        partial[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(AD_BLOCK) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float bias1,
                                                   float bias2_sqrt, float max_norm, const double* __restrict__ partial, int npartial,
                                                   float* __restrict__ norm_out, __hip_bfloat16* __restrict__ shadow,
                                                   const int64_t* __restrict__ step_dev, int vec_ok) {
    // bias corrections from the device step count (the host-computed ones are baked into a captured graph) and the clip coefficient: one lane per block works them out
    // (two double-precision pow() per THREAD and a scalar element loop made this launch 39.7 us for 163 MB, 4.1 TB/s -- round 6)
    __shared__ double l[AD_BLOCK / 64];
    __shared__ float sc[3];
    double t = 0.0;
    if (max_norm > 0.f) {   // every block re-reduces the npartial (512) L2-resident partial sums: cheaper than a third launch
        for (int k = threadIdx.x; k < npartial; k += AD_BLOCK) t += partial[k];
        for (int m2 = 32; m2 >= 1; m2 >>= 1) t += __shfl_xor(t, m2, 64);
        if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (step_dev) {
            const double ts = (double)*step_dev;
            bias1 = (float)(1.0 - pow((double)beta1, ts));
            bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, ts));
        }
        float coef0 = 1.f;
        if (max_norm > 0.f) {
            t = 0.0;
            for (int k = 0; k < AD_BLOCK / 64; ++k) t += l[k];
            const float total = (float)sqrt(t);
            coef0 = fminf(max_norm / (total + 1e-6f), 1.0f);
            if (norm_out && blockIdx.x == 0) *norm_out = total;
        }
        sc[0] = bias1; sc[1] = bias2_sqrt; sc[2] = coef0;
    }
    __syncthreads();
    bias1 = sc[0]; bias2_sqrt = sc[1];
    const float coef = sc[2];
    const float step_size = lr / bias1;
    const int64_t i0 = ((int64_t)blockIdx.x * AD_BLOCK + threadIdx.x) * 4;
    if (i0 >= n) return;
    auto one = [&](float gi, float pi, float mo, float vo, float& go, float& po, float& mn, float& vn) {
        gi = gi * coef;
        go = gi;                                      // clip_grad_norm_ scales the gradient in place
        if (weight_decay != 0.f) gi += weight_decay * pi;
        mn = beta1 * mo + (1.f - beta1) * gi;         // exp_avg.lerp_(grad, 1 - beta1)
        vn = beta2 * vo + (1.f - beta2) * gi * gi;
        po = pi - step_size * (mn / (sqrtf(vn) / bias2_sqrt + eps));
    };
    if (vec_ok && i0 + 4 <= n) {      // (the flat parameter / gradient / moment buffers are 16-byte aligned: one 16-byte load and store per array and lane)
        const float4 g4 = *reinterpret_cast<const float4*>(g + i0), p4 = *reinterpret_cast<const float4*>(p + i0);
        const float4 m4 = *reinterpret_cast<const float4*>(m + i0), v4 = *reinterpret_cast<const float4*>(v + i0);
        float4 go, po, mn, vn;
        one(g4.x, p4.x, m4.x, v4.x, go.x, po.x, mn.x, vn.x);
        one(g4.y, p4.y, m4.y, v4.y, go.y, po.y, mn.y, vn.y);
        one(g4.z, p4.z, m4.z, v4.z, go.z, po.z, mn.z, vn.z);
        one(g4.w, p4.w, m4.w, v4.w, go.w, po.w, mn.w, vn.w);
        *reinterpret_cast<float4*>(g + i0) = go; *reinterpret_cast<float4*>(m + i0) = mn; *reinterpret_cast<float4*>(v + i0) = vn; *reinterpret_cast<float4*>(p + i0) = po;
        if (shadow) {       // the copy the bf16 GEMMs of the next step read
            const __hip_bfloat16 b0 = __float2bfloat16(po.x), b1 = __float2bfloat16(po.y), b2 = __float2bfloat16(po.z), b3 = __float2bfloat16(po.w);
            uint2 q;
            q.x = (uint32_t)*reinterpret_cast<const uint16_t*>(&b0) | ((uint32_t)*reinterpret_cast<const uint16_t*>(&b1) << 16);
            q.y = (uint32_t)*reinterpret_cast<const uint16_t*>(&b2) | ((uint32_t)*reinterpret_cast<const uint16_t*>(&b3) << 16);
            *reinterpret_cast<uint2*>(shadow + i0) = q;
        }
        return;
    }
    for (int64_t i = i0; i < n; ++i) {
        float go, po, mn, vn;
        one(g[i], p[i], m[i], v[i], go, po, mn, vn);
        g[i] = go; m[i] = mn; v[i] = vn; p[i] = po;
        if (shadow) shadow[i] = __float2bfloat16(po);
    }
}

// ------------------------------------------------------------------------------------------
// P8: the actor / critic part of the PPO loss and its gradient in one pass (amp_agent.py:598-640 `calc_gradients`, rl_games
// `neglogp`, common_agent.py:512-520 `bound_loss`, torch_ext.policy_kl):
//   neglogp = 0.5 sum_d ((a - mu) / sigma)^2 + 0.5 log(2 pi) D + sum_d logstd,  ratio = exp(old_neglogp - neglogp)
//   a_loss  = max(-adv ratio, -adv clamp(ratio, 1 - e, 1 + e));  c_loss = (ret - v)^2  (or the clipped variant);
//   b_loss  = sum_d clamp_min(mu - 1, 0)^2 + clamp_max(mu + 1, 0)^2;  entropy = sum_d (0.5 + 0.5 log(2 pi) + logstd)
//   loss    = mean(a_loss) + critic_coef mean(c_loss) - entropy_coef mean(entropy) + bounds_loss_coef mean(b_loss)
// and d loss / d mu [B, D], d loss / d value [B] -- ~100 torch launches forward + backward otherwise.  One wavefront per row,
// lanes over the action dimension; batch sums in fp64 through per-block partials (deterministic).
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld_f(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_f<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f<__hip_bfloat16>(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ void st_f(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_f<__hip_bfloat16>(__hip_bfloat16* p, int64_t i, float v) { p[i] = __float2bfloat16(v); }


#define PPO_NSUM 5   // a_loss, c_loss, b_loss, kl, clipped (|ratio - 1| > e_clip: common_agent.py:570-571)
// Round 6: one HALF-wavefront (32 lanes) per row instead of a whole one -- 69 action dimensions are three passes of 32 lanes or two of 64 with 59 idle lanes in
// the second; two rows per wavefront halve the number of dependent row iterations (gather index -> row loads -> three reductions -> exp -> second pass) a
// wavefront walks through (27 -> 16 us at 16384 x 69).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <typename T>
__global__ __launch_bounds__(256) void k_ppo_loss(const T* __restrict__ mu, const T* __restrict__ value, const float* __restrict__ logstd,
                                                  const float* __restrict__ actions, const float* __restrict__ old_neglogp,
                                                  const float* __restrict__ adv, const float* __restrict__ ret, const float* __restrict__ old_value,
                                                  const float* __restrict__ old_mu, const float* __restrict__ old_sigma,
                                                  const int64_t* __restrict__ idx, int64_t B, int D, phc_ppo_params_t prm, T* __restrict__ grad_mu, T* __restrict__ grad_value, double* __restrict__ partial) {
    __shared__ double lsum[8][PPO_NSUM];
    const int lane = threadIdx.x & 31, w = threadIdx.x >> 5;   // lane of the half-wavefront, half-wavefront of the block
    const float invB = 1.0f / (float)B;
    float sum_logstd = 0.f;
    for (int d = lane; d < D; d += 32) sum_logstd += logstd[d];
    sum_logstd = half_sum(sum_logstd);
    const float nlp_const = 0.5f * 1.8378770664093453f * (float)D + sum_logstd;   // log(2 pi)
    double acc[PPO_NSUM] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * 8 + w; r < B; r += (int64_t)gridDim.x * 8) {
        float s_nlp = 0.f, s_b = 0.f, s_kl = 0.f;
        const int64_t q = idx ? idx[r] : r;   // row of the rollout tensors (actions, old_*, adv, ret); mu / value are minibatch-ordered
        for (int d = lane; d < D; d += 32) {
            const float m = ld_f(mu, r * D + d), a = actions[q * D + d], sg = expf(logstd[d]);
            const float z = (a - m) / sg;
            s_nlp += z * z;
            const float hi = fmaxf(m - 1.0f, 0.f), lo = fminf(m + 1.0f, 0.f);
            s_b += lo * lo + hi * hi;
            const float m1 = old_mu[q * D + d], s1 = old_sigma[q * D + d];
            s_kl += logf(s1 / sg + 1e-5f) + (sg * sg + (m1 - m) * (m1 - m)) / (2.0f * (s1 * s1 + 1e-5f)) - 0.5f;
        }
        s_nlp = half_sum(s_nlp); s_b = half_sum(s_b); s_kl = half_sum(s_kl);
        const float neglogp = 0.5f * s_nlp + nlp_const;
        const float ratio = expf(old_neglogp[q] - neglogp);
        const float A = adv[q];
        const float lo_r = 1.0f - prm.e_clip, hi_r = 1.0f + prm.e_clip;
        const float t1 = -A * ratio, t2 = -A * fminf(fmaxf(ratio, lo_r), hi_r);
        const float a_loss = fmaxf(t1, t2);
        // torch.max backward: the larger operand takes the gradient (ties: half each); clamp passes it inside [lo, hi]
        const float w1 = t1 > t2 ? 1.0f : (t1 == t2 ? 0.5f : 0.f);
        const float inside = (ratio >= lo_r && ratio <= hi_r) ? 1.0f : 0.f;
        const float c_mu = -A * (w1 + (1.0f - w1) * inside) * ratio * invB;
        const float v = ld_f(value, r), R = ret[q];
        float c_loss, dC;
        if (prm.clip_value) {
            const float vp = old_value[q];
            const float dv = v - vp, vpc = vp + fminf(fmaxf(dv, -prm.e_clip), prm.e_clip);
            const float l1 = (v - R) * (v - R), l2 = (vpc - R) * (vpc - R);
            c_loss = fmaxf(l1, l2);
            const float u1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.f);
            const float in2 = (dv >= -prm.e_clip && dv <= prm.e_clip) ? 1.0f : 0.f;
            dC = u1 * 2.0f * (v - R) + (1.0f - u1) * 2.0f * (vpc - R) * in2;
        } else {
            c_loss = (R - v) * (R - v);
            dC = 2.0f * (v - R);
        }
        if (lane == 0) {
            st_f(grad_value, r, prm.critic_coef * dC * invB);
            acc[0] += (double)a_loss; acc[1] += (double)c_loss; acc[2] += (double)s_b; acc[3] += (double)s_kl;
            acc[4] += fabsf(ratio - 1.0f) > prm.e_clip ? 1.0 : 0.0;
        }
        const float cb = prm.bounds_loss_coef * invB;
        for (int d = lane; d < D; d += 32) {
            const float m = ld_f(mu, r * D + d), a = actions[q * D + d], sg = expf(logstd[d]);
            const float g = c_mu * (a - m) / (sg * sg) + cb * (2.0f * fmaxf(m - 1.0f, 0.f) + 2.0f * fminf(m + 1.0f, 0.f));
            st_f(grad_mu, r * D + d, g);
        }
    }
    if (lane == 0)
        for (int k = 0; k < PPO_NSUM; ++k) lsum[w][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < PPO_NSUM) {
        const int k = threadIdx.x;
        partial[(int64_t)blockIdx.x * PPO_NSUM + k] = ((lsum[0][k] + lsum[1][k]) + (lsum[2][k] + lsum[3][k])) + ((lsum[4][k] + lsum[5][k]) + (lsum[6][k] + lsum[7][k]));
    }
}

// stats[0..6] = loss, mean a_loss, mean c_loss, mean b_loss, entropy, mean kl, clip fraction.  320 threads: wavefront k reduces sum k;
// wavefront 0 also the entropy.
__global__ __launch_bounds__(64 * PPO_NSUM) void k_ppo_loss_finish(const double* __restrict__ partial, int nblocks, int64_t B, int D,
                                                         const float* __restrict__ logstd, phc_ppo_params_t prm, float* __restrict__ stats) {
    __shared__ double l[PPO_NSUM];
    __shared__ float lent;
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double t = 0.0;
    for (int b = lane; b < nblocks; b += 64) t += partial[(int64_t)b * PPO_NSUM + k];
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    if (lane == 0) l[k] = t / (double)B;
    if (k == 0) {
        float e = 0.f;
        for (int d = lane; d < D; d += 64) e += 0.5f + 0.5f * 1.8378770664093453f + logstd[d];
        e = wave_sum(e);
        if (lane == 0) lent = e;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float ent = lent;
    stats[1] = (float)l[0]; stats[2] = (float)l[1]; stats[3] = (float)l[2]; stats[4] = ent; stats[5] = (float)l[3]; stats[6] = (float)l[4];
    stats[0] = (float)l[0] + prm.critic_coef * (float)l[1] - prm.entropy_coef * ent + prm.bounds_loss_coef * (float)l[2];
}

// ------------------------------------------------------------------------------------------
// Rollout: sample the action and everything the experience buffer stores about it in one pass (amp_agent.py:309-341 `play_steps` /
// rl_games ModelA2CContinuousLogStd in eval mode): action = mu + sigma * noise, neglogp(action), sigma, and the critic's value
// un-normalised (RunningMeanStd.forward(unnorm=True), running_mean_std.py:87-90) -- ~25 torch launches per rollout step otherwise.
// One wavefront per env.  `mask` (optional): value *= 1 - mask[r]  (next_values of terminated envs, amp_agent.py:352-354).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_policy_sample(const T* __restrict__ mu, const float* __restrict__ logstd, const float* __restrict__ noise,
                                                       const T* __restrict__ value, const double* __restrict__ vmean, const double* __restrict__ vvar,
                                                       float eps, const float* __restrict__ mask, int64_t N, int D, float* __restrict__ actions,
                                                       float* __restrict__ mus, float* __restrict__ sigmas, float* __restrict__ neglogp,
                                                       float* __restrict__ values) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    if (mu) {
        float s = 0.f, sl = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float m = ld_f(mu, r * D + d), ls = logstd[d], sg = expf(ls);
            const float a = m + sg * noise[r * D + d];
            const float z = (a - m) / sg;
            s += z * z; sl += ls;
            actions[r * D + d] = a; mus[r * D + d] = m; sigmas[r * D + d] = sg;
        }
        s = wave_sum(s); sl = wave_sum(sl);
        if (lane == 0) neglogp[r] = 0.5f * s + 0.5f * 1.8378770664093453f * (float)D + sl;
    }
    if (lane == 0 && value) {
        float v = ld_f(value, r);
        if (vmean) v = sqrtf((float)vvar[0] + eps) * fminf(fmaxf(v, -5.0f), 5.0f) + (float)vmean[0];
        if (mask) v *= 1.0f - mask[r];
        values[r] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Discriminator loss pieces (amp_agent.py:732-808 `_disc_loss`), each one or two launches instead of a dozen tiny torch kernels:
//   k_disc_bce:   0.5 (BCEWithLogits(agent rows, 0) + BCEWithLogits(demo rows, 1)), both accuracies, and d loss / d logit
//   k_sumsq_multi (+ finish): sum_i coef_i |w_i|^2 over up to 4 tensors (logit regulariser + weight decay), or with one bf16 tensor
//                 the gradient penalty coef * mean_rows(sum_cols g^2)
// ------------------------------------------------------------------------------------------
// One 1024-lane block per 1024 logits (round 6: ONE block for all 12 288 was 34 k wavefront-instructions of exp / log1p on a single CU, 17.8 us); the block that finishes
// last (ticket counter) adds the per-block sums in block order and writes the scalars.  The partial sums live in a static device buffer: launches of this kernel on
// DIFFERENT streams of one process must not overlap (the learner issues one per optimizer step, on the discriminator's stream).
#define BCE_MAX_BLOCKS 64
__device__ float g_bce_partial[BCE_MAX_BLOCKS * 6];
__device__ unsigned int g_bce_ticket = 0;
template <typename T>
__global__ __launch_bounds__(1024) void k_disc_bce(const T* __restrict__ logits, int n_agent, int n_demo, float scale, T* __restrict__ grad,
                                                   float* __restrict__ stats) {
    __shared__ float l[6][16];
    __shared__ int last;
    float la = 0.f, ld = 0.f, ca = 0.f, cd = 0.f, xa = 0.f, xd = 0.f;
    const int n = n_agent + n_demo;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
        const float x = ld_f(logits, i);
        const float sp = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));   // softplus(x) = BCEWithLogits(x, 0); BCEWithLogits(x, 1) = softplus(x) - x
        const float sg = 1.0f / (1.0f + expf(-x));
        if (i < n_agent) { la += sp; ca += x < 0.f ? 1.f : 0.f; xa += x; st_f(grad, i, scale * 0.5f * sg / (float)n_agent); }
        else { ld += sp - x; cd += x > 0.f ? 1.f : 0.f; xd += x; st_f(grad, i, scale * 0.5f * (sg - 1.0f) / (float)n_demo); }
    }
    la = wave_sum(la); ld = wave_sum(ld); ca = wave_sum(ca); cd = wave_sum(cd); xa = wave_sum(xa); xd = wave_sum(xd);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { l[0][w] = la; l[1][w] = ld; l[2][w] = ca; l[3][w] = cd; l[4][w] = xa; l[5][w] = xd; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += l[threadIdx.x][k];
        g_bce_partial[blockIdx.x * 6 + threadIdx.x] = t;
        __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&g_bce_ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 6) {
        float t = 0.f;
        for (int b = 0; b < (int)gridDim.x; ++b) t += __builtin_nontemporal_load(&g_bce_partial[b * 6 + threadIdx.x]);
        l[threadIdx.x][0] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g_bce_ticket = 0u;
        stats[0] = scale * 0.5f * (l[0][0] / (float)n_agent + l[1][0] / (float)n_demo);
        stats[1] = l[2][0] / (float)n_agent;
        stats[2] = l[3][0] / (float)n_demo;
        stats[3] = l[4][0] / (float)n_agent;      // mean logits: the reference's `disc/agent_logit`, `disc/demo_logit` scalars (amp_agent.py:911-912)
        stats[4] = l[5][0] / (float)n_demo;
    }
}

// ------------------------------------------------------------------------------------------
// Sum of the split-K slabs of a weight gradient: part [slabs, n] bf16 (the batched GEMM's output) -> out fp32 [n], written or ADDED
// (out += sum: a later contribution to a gradient that already holds one -- autograd's AccumulateGrad launch disappears).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_slabs_bf16(const __hip_bfloat16* __restrict__ part, int slabs, int64_t n, float* __restrict__ out, int accumulate) {
    const int64_t i8 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i8 >= n) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i8 + 8 <= n && (n & 7) == 0) {
        int s = 0;
        for (; s + 8 <= slabs; s += 8) {     // (SPLIT_K = 8: all eight slab loads in flight -- one at a time made this a chain of dependent L2 / HBM round trips)
            uint4 q[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) q[t] = *reinterpret_cast<const uint4*>(part + (int64_t)(s + t) * n + i8);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t w[4] = {q[t].x, q[t].y, q[t].z, q[t].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { a[2 * k] += __uint_as_float(w[k] << 16); a[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
            }
        }
        for (; s < slabs; ++s) {
            const uint4 q = *reinterpret_cast<const uint4*>(part + (int64_t)s * n + i8);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[2 * k] += __uint_as_float(w[k] << 16); a[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
        }
        float4* o = reinterpret_cast<float4*>(out + i8);
        float4 lo = make_float4(a[0], a[1], a[2], a[3]), hi = make_float4(a[4], a[5], a[6], a[7]);
        if (accumulate) { const float4 p = o[0], r = o[1]; lo.x += p.x; lo.y += p.y; lo.z += p.z; lo.w += p.w; hi.x += r.x; hi.y += r.y; hi.z += r.z; hi.w += r.w; }
        o[0] = lo; o[1] = hi;
        return;
    }
    for (int k = 0; k < 8 && i8 + k < n; ++k) {
        float v = 0.f;
        for (int s = 0; s < slabs; ++s) v += __bfloat162float(part[(int64_t)s * n + i8 + k]);
        out[i8 + k] = accumulate ? out[i8 + k] + v : v;
    }
}

// ------------------------------------------------------------------------------------------
// Per-step bookkeeping of the rollout (amp_agent.py:321-341 of the reference's play_steps): row n of the experience buffer gets the
// step's (scaled) rewards and done flags; the episode statistics advance; the per-term reward means accumulate.  One single-block launch
// instead of a dozen elementwise torch kernels on 4096-element vectors (~5 us each in a captured graph).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rollout_bookkeeping(const float* __restrict__ rewards, float reward_scale, const int64_t* __restrict__ dones,
                                                              const int64_t* __restrict__ terminate, const float* __restrict__ reward_raw, int nraw,
                                                              int64_t n, float* __restrict__ exp_rewards, uint8_t* __restrict__ exp_dones,
                                                              float* __restrict__ terminated_flags, float* __restrict__ terminated_mask,
                                                              float* __restrict__ reward_raw_acc, float* __restrict__ current_rewards,
                                                              float* __restrict__ current_lengths) {
    __shared__ double l[16][8];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float r = rewards[i];
        const float nd = dones[i] != 0 ? 0.f : 1.f;
        const float t = terminate[i] != 0 ? 1.f : 0.f;
        exp_rewards[i] = r * reward_scale;
        exp_dones[i] = dones[i] != 0 ? 1 : 0;
        terminated_flags[i] += t;
        terminated_mask[i] = t;
        current_rewards[i] = (current_rewards[i] + r) * nd;
        current_lengths[i] = (current_lengths[i] + 1.0f) * nd;
        for (int k = 0; k < nraw; ++k) acc[k] += (double)reward_raw[i * nraw + k];
    }
    for (int k = 0; k < nraw; ++k) {
        double v = acc[k];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < nraw) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += l[w][threadIdx.x];
        reward_raw_acc[threadIdx.x] += (float)(s / (double)n);
    }
}

struct SumsqArgs { const void* ptr[4]; int64_t n[4]; float coef[4]; int count; int is_bf16; };
#define SSM_BLOCKS 1024
// partial[t][block] = this block's share of |tensor t|^2 (unweighted).  16-byte loads (8 bf16 / 4 fp32) over the aligned bulk, scalar tail;
// 1024 blocks (round 2: 256 blocks of scalar loads read the 16 MB penalty gradient at 0.47 TB/s)
__global__ __launch_bounds__(256) void k_sumsq_multi(SumsqArgs a, double* __restrict__ partial) {
    __shared__ double l[4][4];
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)SSM_BLOCKS * 256;
    for (int t = 0; t < a.count; ++t) {
        float s = 0.f;
        const int64_t n = a.n[t];
        const bool aligned = (reinterpret_cast<uintptr_t>(a.ptr[t]) & 15) == 0;
        int64_t done = 0;
        if (aligned && a.is_bf16) {
            const int64_t nv = n / 8;
            const uint4* p = reinterpret_cast<const uint4*>(a.ptr[t]);
            for (int64_t i = tid; i < nv; i += nthreads) {
                const uint4 q = p[i];
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
                    s += lo * lo; s += hi * hi;
                }
            }
            done = nv * 8;
        } else if (aligned) {
            const int64_t nv = n / 4;
            const float4* p = reinterpret_cast<const float4*>(a.ptr[t]);
            for (int64_t i = tid; i < nv; i += nthreads) {
                const float4 q = p[i];
                s += q.x * q.x; s += q.y * q.y; s += q.z * q.z; s += q.w * q.w;
            }
            done = nv * 4;
        }
        for (int64_t i = done + tid; i < n; i += nthreads) {
            const float v = a.is_bf16 ? __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(a.ptr[t])[i]) : reinterpret_cast<const float*>(a.ptr[t])[i];
            s += v * v;
        }
        double acc = (double)s;
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if ((threadIdx.x & 63) == 0) l[t][threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x < a.count) partial[threadIdx.x * SSM_BLOCKS + blockIdx.x] = (l[threadIdx.x][0] + l[threadIdx.x][1]) + (l[threadIdx.x][2] + l[threadIdx.x][3]);
}
// out[0] = sum_t coef[t] |tensor t|^2, out[1 + t] = |tensor t|^2
__global__ __launch_bounds__(256) void k_sumsq_multi_finish(SumsqArgs a, const double* __restrict__ partial, float* __restrict__ out) {
    __shared__ double l[4];
    double total = 0.0;
    for (int t = 0; t < a.count; ++t) {
        double v = 0.0;
        for (int k = threadIdx.x; k < SSM_BLOCKS; k += 256) v += partial[t * SSM_BLOCKS + k];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = v;
        __syncthreads();
        const double s = (l[0] + l[1]) + (l[2] + l[3]);
        if (threadIdx.x == 0) out[1 + t] = (float)s;
        total += (double)a.coef[t] * s;
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}

extern "C" {

// partial sums + 8 bytes holding the finish kernel's ticket counter, which must be ZERO before the first call (it is left at zero)
int64_t phc_running_norm_workspace(int64_t rows, int32_t cols) {
    return ((rows + RN_ROWS - 1) / RN_ROWS) * 2 * (int64_t)cols * (int64_t)sizeof(double) + 8;
}

int32_t phc_running_norm(const float* x, const int64_t* row_index, int64_t rows, int32_t cols, const double* norm_mean, const double* norm_var, float epsilon,
                         float clamp, void* out, int32_t out_bf16, int32_t out_stride, double* run_mean, double* run_var, double* run_count,
                         double* workspace, void* stream) {
    if (!x || rows < 0 || cols < 1 || !norm_mean || !norm_var) return PHC_EINVAL;
    if (out_stride == 0) out_stride = cols;
    if (out_stride < cols) return PHC_EINVAL;
    const bool update = run_mean != nullptr;
    if (update && (!run_var || !run_count || !workspace)) return PHC_EINVAL;
    if (!update && !out) return PHC_EINVAL;
    if (rows == 0) return 0;
    const int64_t nchunks = (rows + RN_ROWS - 1) / RN_ROWS;
    if (nchunks > 65535) return PHC_EUNSUPPORTED;
    const dim3 grid((cols + RN_COLS - 1) / RN_COLS, (unsigned)nchunks);
    hipStream_t st = (hipStream_t)stream;
    if (out_bf16)
        hipLaunchKernelGGL(k_running_norm<true>, grid, dim3(RN_COLS), 0, st, x, row_index, rows, cols, norm_mean, norm_var, epsilon, clamp, out, out_stride, update ? workspace : nullptr);
    else
        hipLaunchKernelGGL(k_running_norm<false>, grid, dim3(RN_COLS), 0, st, x, row_index, rows, cols, norm_mean, norm_var, epsilon, clamp, out, out_stride, update ? workspace : nullptr);
    if (update)
        hipLaunchKernelGGL(k_running_norm_finish, dim3((cols + RNF_COLS - 1) / RNF_COLS), dim3(RNF_COLS * RNF_SLICES), 0, st, workspace, (int)nchunks, rows, cols, run_mean, run_var, run_count,
                           reinterpret_cast<unsigned int*>(workspace + nchunks * 2 * (int64_t)cols));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int64_t phc_colsum_workspace(int64_t rows, int32_t cols) { return ((rows + CS_ROWS - 1) / CS_ROWS) * (int64_t)cols * (int64_t)sizeof(float); }

int32_t phc_colsum_bf16(const void* x, int64_t rows, int32_t cols, float* out, float* workspace, void* stream) {
    if (!x || !workspace || rows < 1 || cols < 1) return PHC_EINVAL;
    const int64_t nchunks = (rows + CS_ROWS - 1) / CS_ROWS;
    if (nchunks > 65535) return PHC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_bf16, dim3((cols + 63) / 64, (unsigned)nchunks), dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(x), rows, cols, workspace);
    if (out) hipLaunchKernelGGL(k_colsum_finish, dim3((cols + 63) / 64), dim3(1024), 0, st, workspace, (int)nchunks, cols, out);   // (NULL: phc_colsum_finish_batch later)
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_colsum_relu_bf16(const void* gy, const void* y, int64_t rows, int32_t cols, void* gm, float* out, float* workspace, void* stream) {
    if (!gy || !y || !gm || !workspace || rows < 1 || cols < 1) return PHC_EINVAL;
    const int64_t nchunks = (rows + CS_ROWS - 1) / CS_ROWS;
    if (nchunks > 65535) return PHC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_relu_bf16, dim3((cols + 63) / 64, (unsigned)nchunks), dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(gy),
                       reinterpret_cast<const __hip_bfloat16*>(y), rows, cols, reinterpret_cast<__hip_bfloat16*>(gm), workspace);
    if (out) hipLaunchKernelGGL(k_colsum_finish, dim3((cols + 63) / 64), dim3(1024), 0, st, workspace, (int)nchunks, cols, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_colsum_chunks(int64_t rows) { return (int32_t)((rows + CS_ROWS - 1) / CS_ROWS); }
int32_t phc_linear1_chunks(int64_t rows) { return (int32_t)((rows + L1_ROWS - 1) / L1_ROWS); }

int32_t phc_colsum_finish_batch(int32_t count, const phc_colsum_job_t* jobs, void* stream) {
    if (count < 0 || (count > 0 && !jobs)) return PHC_EINVAL;
    for (int32_t i = 0; i < count; ++i)
        if (!jobs[i].partial || !jobs[i].out || jobs[i].nchunks < 1 || jobs[i].cols < 1) return PHC_EINVAL;
    for (int32_t i0 = 0; i0 < count; i0 += PHC_COLSUM_MAX_JOBS) {
        ColsumJobs a;
        const int n = count - i0 < PHC_COLSUM_MAX_JOBS ? count - i0 : PHC_COLSUM_MAX_JOBS;
        int maxcols = 1;
        for (int i = 0; i < n; ++i) { a.job[i] = jobs[i0 + i]; if (a.job[i].cols > maxcols) maxcols = a.job[i].cols; }
        for (int i = n; i < PHC_COLSUM_MAX_JOBS; ++i) a.job[i] = a.job[0];
        hipLaunchKernelGGL(k_colsum_finish_batch, dim3((maxcols + 63) / 64, (unsigned)n), dim3(1024), 0, (hipStream_t)stream, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_sum_slabs_bf16(const void* part, int32_t slabs, int64_t n, float* out, int32_t accumulate, void* stream) {
    if (!part || !out || slabs < 1 || n < 1) return PHC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(part) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return PHC_EINVAL;
    const int64_t blocks = (n + 2047) / 2048;
    hipLaunchKernelGGL(k_sum_slabs_bf16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __hip_bfloat16*>(part), slabs, n, out,
                       accumulate);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

// ------------------------------------------------------------------------------------------
// Split-precision operand of a linear layer (`actor_precision=split_bf16`): x fp32 [rows, cols] (rows `ld_in` apart), optionally gated by another fp32 tensor
// (the ReLU mask of a backward pass: x where gate > 0, else 0), is cut into a bf16 head h = bf16(x) and a bf16 tail l = bf16(x - h) and stored as the three
// chunks one long-reduction GEMM reads: out[row * row_stride + c * chunk_stride + col], c = 0..2 holding (h, h, l) (order 0) or (h, l, h) (order 1); columns
// cols..cols_pad-1 and rows rows..rows_pad-1 are written as zeros (the GEMM's reduction length is a multiple of 32 elements) -- except column `cols` of the valid
// rows with extra_mode 1 (the constant 1) or 2 (extra[row]): an activation operand with the ones column against a weight operand whose column `cols` is the bias
// makes the bias part of the product, and the ones column's row of the weight-gradient product IS the bias gradient.  4 columns per lane.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split3_bf16(const float* __restrict__ x, int64_t ld_in, const float* __restrict__ gate, int64_t ld_gate, int64_t rows, int cols,
                                                     int64_t rows_pad, int cols_pad, const float* __restrict__ extra, int extra_mode, __hip_bfloat16* __restrict__ out,
                                                     int64_t row_stride, int64_t chunk_stride, int order) {
    const int q = cols_pad >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / q;
    if (row >= rows_pad) return;
    const int c0 = (int)(t - row * q) << 2;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        const float* xr = x + row * ld_in;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < cols) v[k] = xr[c0 + k];
        if (gate) {
            const float* gr = gate + row * ld_gate;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < cols && !(gr[c0 + k] > 0.f)) v[k] = 0.f;
        }
        if (extra_mode && cols >= c0 && cols < c0 + 4) v[cols - c0] = extra_mode == 1 ? 1.f : extra[row];
    }
    uint16_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const __hip_bfloat16 hb = __float2bfloat16(v[k]);
        const __hip_bfloat16 lb = __float2bfloat16(v[k] - __bfloat162float(hb));
        h[k] = *reinterpret_cast<const uint16_t*>(&hb);
        l[k] = *reinterpret_cast<const uint16_t*>(&lb);
    }
    const uint2 hq = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    const uint2 lq = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    __hip_bfloat16* o = out + row * row_stride + c0;
    *reinterpret_cast<uint2*>(o) = hq;
    *reinterpret_cast<uint2*>(o + chunk_stride) = order ? lq : hq;
    *reinterpret_cast<uint2*>(o + 2 * chunk_stride) = order ? hq : lq;
}

int32_t phc_split3_bf16(const float* x, int64_t ld_in, const float* gate, int64_t ld_gate, int64_t rows, int32_t cols, int64_t rows_pad, int32_t cols_pad,
                        const float* extra, int32_t extra_mode, void* out, int64_t row_stride, int64_t chunk_stride, int32_t order, void* stream) {
    if (!x || !out || rows < 1 || cols < 1 || rows_pad < rows || cols_pad < cols || (cols_pad & 3) || (row_stride & 3) || (chunk_stride & 3)) return PHC_EINVAL;
    if (extra_mode < 0 || extra_mode > 2 || (extra_mode && cols_pad <= cols) || (extra_mode == 2 && !extra)) return PHC_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 7) return PHC_EINVAL;
    const int64_t threads = rows_pad * (cols_pad >> 2);
    hipLaunchKernelGGL(k_split3_bf16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld_in, gate, ld_gate, rows, cols, rows_pad, cols_pad,
                       extra, extra_mode, reinterpret_cast<__hip_bfloat16*>(out), row_stride, chunk_stride, order);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_rollout_bookkeeping(const float* rewards, float reward_scale, const int64_t* dones, const int64_t* terminate, const float* reward_raw,
                                int32_t num_reward_terms, int64_t num_envs, float* exp_rewards, uint8_t* exp_dones, float* terminated_flags,
                                float* terminated_mask, float* reward_raw_acc, float* current_rewards, float* current_lengths, void* stream) {
    if (!rewards || !dones || !terminate || !reward_raw || !exp_rewards || !exp_dones || !terminated_flags || !terminated_mask || !reward_raw_acc ||
        !current_rewards || !current_lengths || num_envs < 1 || num_reward_terms < 1 || num_reward_terms > 8)
        return PHC_EINVAL;
    hipLaunchKernelGGL(k_rollout_bookkeeping, dim3(1), dim3(1024), 0, (hipStream_t)stream, rewards, reward_scale, dones, terminate, reward_raw, num_reward_terms,
                       num_envs, exp_rewards, exp_dones, terminated_flags, terminated_mask, reward_raw_acc, current_rewards, current_lengths);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int64_t phc_linear1_workspace(int64_t rows, int32_t cols) { return ((rows + L1_ROWS - 1) / L1_ROWS) * (int64_t)(cols + 1) * (int64_t)sizeof(float); }

int32_t phc_linear1_forward(const void* x, const void* w, const void* b, int64_t rows, int32_t cols, void* y, void* stream) {
    if (!x || !w || !b || !y || rows < 1 || cols < 1) return PHC_EINVAL;
    hipLaunchKernelGGL(k_linear1_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16*)x, (const __hip_bfloat16*)w,
                       (const __hip_bfloat16*)b, rows, cols, (__hip_bfloat16*)y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_linear1_backward(const void* x, const void* w, const void* gy, int64_t rows, int32_t cols, void* gx, float* gw_gb, float* workspace,
                             void* stream) {
    if (!x || !w || !gy || !workspace || rows < 1 || cols < 1) return PHC_EINVAL;
    const int64_t nchunks = (rows + L1_ROWS - 1) / L1_ROWS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_linear1_bwd, dim3((unsigned)nchunks), dim3(256), 0, st, (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, (const __hip_bfloat16*)gy,
                       rows, cols, (__hip_bfloat16*)gx, workspace);
    if (gw_gb) hipLaunchKernelGGL(k_colsum_finish, dim3((cols + 1 + 63) / 64), dim3(1024), 0, st, workspace, (int)nchunks, cols + 1, gw_gb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_policy_sample(const void* mu, const void* value, int32_t is_bf16, const float* logstd, const float* noise, const double* value_mean,
                          const double* value_var, float epsilon, const float* mask, int64_t num_envs, int32_t num_actions, float* actions, float* mus,
                          float* sigmas, float* neglogp, float* values, void* stream) {
    if (num_envs < 1 || (!mu && !value)) return PHC_EINVAL;
    if (mu && (!logstd || !noise || !actions || !mus || !sigmas || !neglogp || num_actions < 1)) return PHC_EINVAL;
    if (value && (!values || ((value_mean == nullptr) != (value_var == nullptr)))) return PHC_EINVAL;
    const dim3 grid((unsigned)((num_envs + 3) / 4));
    if (is_bf16)
        hipLaunchKernelGGL(k_policy_sample<__hip_bfloat16>, grid, dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16*)mu, logstd, noise,
                           (const __hip_bfloat16*)value, value_mean, value_var, epsilon, mask, num_envs, num_actions, actions, mus, sigmas, neglogp, values);
    else
        hipLaunchKernelGGL(k_policy_sample<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)mu, logstd, noise, (const float*)value, value_mean,
                           value_var, epsilon, mask, num_envs, num_actions, actions, mus, sigmas, neglogp, values);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int32_t phc_disc_bce(const void* logits, int32_t is_bf16, int32_t n_agent, int32_t n_demo, float scale, void* grad, float* stats, void* stream) {
    if (!logits || !grad || !stats || n_agent < 1 || n_demo < 1) return PHC_EINVAL;
    const int64_t nb = ((int64_t)n_agent + n_demo + 1023) / 1024;
    const dim3 grid((unsigned)(nb < BCE_MAX_BLOCKS ? nb : BCE_MAX_BLOCKS));
    if (is_bf16)
        hipLaunchKernelGGL(k_disc_bce<__hip_bfloat16>, grid, dim3(1024), 0, (hipStream_t)stream, (const __hip_bfloat16*)logits, n_agent, n_demo, scale,
                           (__hip_bfloat16*)grad, stats);
    else
        hipLaunchKernelGGL(k_disc_bce<float>, grid, dim3(1024), 0, (hipStream_t)stream, (const float*)logits, n_agent, n_demo, scale, (float*)grad, stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int64_t phc_sumsq_workspace(void) { return 4 * SSM_BLOCKS * (int64_t)sizeof(double); }

int32_t phc_weighted_sumsq(int32_t count, const void* const* tensors, const int64_t* sizes, const float* coefs, int32_t is_bf16, float* out,
                           double* workspace, void* stream) {
    if (count < 1 || count > 4 || !tensors || !sizes || !coefs || !out || !workspace) return PHC_EINVAL;
    SumsqArgs a;
    a.count = count; a.is_bf16 = is_bf16;
    for (int t = 0; t < 4; ++t) { a.ptr[t] = t < count ? tensors[t] : nullptr; a.n[t] = t < count ? sizes[t] : 0; a.coef[t] = t < count ? coefs[t] : 0.f; }
    for (int t = 0; t < count; ++t) if (!a.ptr[t] || a.n[t] < 0) return PHC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sumsq_multi, dim3(SSM_BLOCKS), dim3(256), 0, st, a, workspace);
    hipLaunchKernelGGL(k_sumsq_multi_finish, dim3(1), dim3(256), 0, st, a, workspace, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

#define AD_NORM_BLOCKS 512
int64_t phc_adam_workspace(void) { return AD_NORM_BLOCKS * (int64_t)sizeof(double); }

int32_t phc_adam_clip_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, float max_norm, double* workspace, float* grad_norm_out, void* param_bf16,
                           int64_t* step_device, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || (step < 1 && !step_device) || !workspace) return PHC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (max_norm > 0.f || step_device) hipLaunchKernelGGL(k_sumsq, dim3(AD_NORM_BLOCKS), dim3(AD_BLOCK), 0, st, grad, n, workspace, step_device);
    if (step < 1) step = 1;
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t blocks = (n + AD_BLOCK * 4 - 1) / (AD_BLOCK * 4);
    const int vec_ok = !((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
                       && !(reinterpret_cast<uintptr_t>(param_bf16) & 7);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(AD_BLOCK), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                       (float)b1, (float)sqrt(b2), max_norm, workspace, AD_NORM_BLOCKS, grad_norm_out, (__hip_bfloat16*)param_bf16, step_device, vec_ok);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

#define PPO_BLOCKS 1024
int64_t phc_ppo_loss_workspace(void) { return PPO_BLOCKS * PPO_NSUM * (int64_t)sizeof(double); }

int32_t phc_ppo_loss(const void* mu, const void* value, int32_t is_bf16, const float* logstd, const float* actions, const float* old_neglogp,
                     const float* advantages, const float* returns, const float* old_values, const float* old_mu, const float* old_sigma,
                     const int64_t* row_index, int64_t batch, int32_t num_actions, const phc_ppo_params_t* prm, void* grad_mu, void* grad_value, float* stats,
                     double* workspace, void* stream) {
    if (!mu || !value || !logstd || !actions || !old_neglogp || !advantages || !returns || !old_mu || !old_sigma || !prm || !grad_mu || !grad_value ||
        !stats || !workspace || batch < 1 || num_actions < 1)
        return PHC_EINVAL;
    if (prm->clip_value && !old_values) return PHC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (int)((batch + 7) / 8 < PPO_BLOCKS ? (batch + 7) / 8 : PPO_BLOCKS);   // eight rows (half-wavefronts) per block and pass
    if (is_bf16)
        hipLaunchKernelGGL(k_ppo_loss<__hip_bfloat16>, dim3(nblocks), dim3(256), 0, st, (const __hip_bfloat16*)mu, (const __hip_bfloat16*)value, logstd, actions,
                           old_neglogp, advantages, returns, old_values, old_mu, old_sigma, row_index, batch, num_actions, *prm, (__hip_bfloat16*)grad_mu,
                           (__hip_bfloat16*)grad_value, workspace);
    else
        hipLaunchKernelGGL(k_ppo_loss<float>, dim3(nblocks), dim3(256), 0, st, (const float*)mu, (const float*)value, logstd, actions, old_neglogp,
                           advantages, returns, old_values, old_mu, old_sigma, row_index, batch, num_actions, *prm, (float*)grad_mu, (float*)grad_value, workspace);
    hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64 * PPO_NSUM), 0, st, workspace, nblocks, batch, num_actions, logstd, *prm, stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

}  // extern "C"
