// BatchNorm statistics, fused affine+activation forward/backward, GateLayer2d epilogue, small elementwise glue.
// All HBM-bound: NHWC rows of C channels are read as float4 per lane (C % 4 == 0) with consecutive lanes on
// consecutive channels/rows, per-channel reductions go through registers -> LDS -> per-chunk partials and a
// fixed-order finalize (no float atomics: bitwise reproducible).
#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

constexpr int kMaxChunks = 1024;  // ~4 workgroups per CU on the big layers; the finalize kernels are wave-parallel over chunks

struct RowMap {
  int cols;   // float4 (or scalar) columns per row handled by distinct threads
  int rpp;    // rows per pass of a 256-thread block
};

static inline RowMap row_map(int C, int vec) {
  RowMap r;
  r.cols = C / vec;
  r.rpp = 256 / r.cols;
  return r;
}

static inline int chunk_count(int64_t M, int rpp) {
  int64_t want = (M + (int64_t)rpp * 4 - 1) / ((int64_t)rpp * 4);
  if (want < 1) want = 1;
  if (want > kMaxChunks) want = kMaxChunks;
  return (int)want;
}

template <int V>
struct Vec;
struct alignas(16) F4 {
  float v[4];
};
template <>
struct Vec<4> {
  typedef F4 T;
  static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const F4*>(p); }
  static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<F4*>(p) = v; }
  static __device__ __forceinline__ void store_wt(float* p, T v) { store_wt4(p, f32x4{v.v[0], v.v[1], v.v[2], v.v[3]}); }
  static __device__ __forceinline__ T load_dt(const float* base, size_t off, bool bf) {
    const f32x4 v = load4_dt(base, off, bf);
    return T{{v[0], v[1], v[2], v[3]}};
  }
  static __device__ __forceinline__ void store_dt(float* base, size_t off, T v, bool bf) {
    store4_dt(base, off, f32x4{v.v[0], v.v[1], v.v[2], v.v[3]}, bf);
  }
};
template <>
struct Vec<1> {
  typedef float T;
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
  static __device__ __forceinline__ void store_wt(float* p, T v) { *p = v; }
  static __device__ __forceinline__ T load_dt(const float* base, size_t off, bool) { return base[off]; }
  static __device__ __forceinline__ void store_dt(float* base, size_t off, T v, bool) { base[off] = v; }
};
template <int V>
__device__ __forceinline__ float& at(typename Vec<V>::T& v, int j);
template <>
__device__ __forceinline__ float& at<4>(F4& v, int j) { return v.v[j]; }
template <>
__device__ __forceinline__ float& at<1>(float& v, int) { return v; }

// ---------------------------------------------------------------------------------------------------------
// bn_stats: per chunk (sum(x-pivot), sum((x-pivot)^2)) per channel with ONE pivot for all chunks (pivot[c] = x[0, c]: any
// value inside the data range conditions the variance; a shared one turns the finalize into plain sums — no per-chunk
// Chan-combine divisions in double). ws layout: [chunks][2][C]
// ---------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t M, int C, int cols,
                                                          int rpp, int64_t rows_per_chunk, float* __restrict__ ws) {
  __shared__ float red[2][256 * 4];
  const int t = threadIdx.x;
  const int col = t % cols, rg = t / cols;
  const bool active = rg < rpp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = min(M, r0 + rows_per_chunk);
  typename Vec<V>::T piv, s1, s2;
  for (int j = 0; j < V; ++j) at<V>(piv, j) = at<V>(s1, j) = at<V>(s2, j) = 0.f;
  piv = Vec<V>::load(x + col * V);
  if (active) {
    for (int64_t r = r0 + rg; r < r1; r += rpp) {
      typename Vec<V>::T v = Vec<V>::load(x + r * C + col * V);
      for (int j = 0; j < V; ++j) {
        float dlt = at<V>(v, j) - at<V>(piv, j);
        at<V>(s1, j) += dlt;
        at<V>(s2, j) += dlt * dlt;
      }
    }
  }
  for (int j = 0; j < V; ++j) {
    red[0][t * 4 + j] = active ? at<V>(s1, j) : 0.f;
    red[1][t * 4 + j] = active ? at<V>(s2, j) : 0.f;
  }
  __syncthreads();
  if (t < cols) {
    float* out = ws + (size_t)blockIdx.x * 2 * C;
    for (int j = 0; j < V; ++j) {
      float a = 0.f, b = 0.f;
      for (int g = 0; g < rpp; ++g) {
        a += red[0][(g * cols + t) * 4 + j];
        b += red[1][(g * cols + t) * 4 + j];
      }
      const int c = t * V + j;
      out[c] = a;
      out[C + c] = b;
    }
  }
}

__device__ __forceinline__ double shfl_xor_d(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return __hiloint2double(hi, lo);
}

// one 256-thread block per channel: thread k sums chunks k, k+256, ... in double, then a fixed xor-shuffle tree inside each
// wave and a fixed-order sum of the 4 wave results through LDS
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ ws, int chunks, int C, int64_t M,
                                                          const float* x, const float* gamma, const float* beta,
                                                          float eps, float momentum, float* running_mean, float* running_var,
                                                          float* scale, float* shift, float* mean_out, float* rstd_out) {
  __shared__ double part[4][2];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the pivot row may BE running_mean (ops.ResBlockFn passes it as both): not __restrict__, and read before any store below
  const float pivot = x[c];
  // everything thread 0 needs at the end is requested now, beside the partial rows, not after the barrier (one round trip less)
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float rm0 = running_mean ? running_mean[c] : 0.f, rv0 = running_mean ? running_var[c] : 0.f;
  // per-thread and in-wave sums in fp32 (at most chunks/256 + 6 additions per value, fixed order), doubles from there on
  float af = 0.f, bf = 0.f;
#pragma unroll 4
  for (int k = tid; k < chunks; k += 256) {
    af += ws[(size_t)k * 2 * C + c];
    bf += ws[(size_t)k * 2 * C + C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    af += __shfl_xor(af, o, 64);
    bf += __shfl_xor(bf, o, 64);
  }
  if (lane == 0) {
    part[wv][0] = (double)af;
    part[wv][1] = (double)bf;
  }
  __syncthreads();
  if (tid != 0) return;
  double a = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
  double b = (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]);
  const double inv_m = 1.0 / (double)M, dm = a * inv_m;  // mean - pivot
  double m2 = b - a * dm;                                 // sum (x - mean)^2
  if (m2 < 0.0) m2 = 0.0;
  const double mean = (double)pivot + dm, var = m2 * inv_m;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = g * rstd;
  scale[c] = sc;
  shift[c] = be - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    const double unbiased = M > 1 ? m2 / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * rm0 + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * rstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

// ---------------------------------------------------------------------------------------------------------
// y = act(x*scale + shift) * row_scale[n, c]
// ---------------------------------------------------------------------------------------------------------
// Row-major mapping shared by the streaming kernels below: thread -> (row group rg, column col) once; rows advance by
// a fixed stride, so the loop body has no integer division (only n = row / rows_per_n when a per-sample scale is used).
template <int V>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int M, int C, int cols, int rpp,
                                                          const float* scale, const float* shift, int act,
                                                          const float* row_scale, int rows_per_n,
                                                          float* __restrict__ y) {
  const int t = threadIdx.x, col = t % cols, rg = t / cols;
  if (rg >= rpp) return;
  const int c = col * V;
  float sc[V], sh[V];
  for (int j = 0; j < V; ++j) {
    sc[j] = scale ? scale[c + j] : 1.f;
    sh[j] = scale ? shift[c + j] : 0.f;
  }
  for (int row = blockIdx.x * rpp + rg; row < M; row += gridDim.x * rpp) {
    const size_t off = (size_t)row * C + c;
    typename Vec<V>::T v = Vec<V>::load(x + off);
    const float* rs = row_scale ? row_scale + (size_t)(row / rows_per_n) * C + c : nullptr;
    for (int j = 0; j < V; ++j) {
      float u = act_fwd(at<V>(v, j) * sc[j] + sh[j], act);
      if (rs) u *= rs[j];
      at<V>(v, j) = u;
    }
    Vec<V>::store(y + off, v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward of h = act(x*scale+shift): reduce pass -> ws [chunks][2][C] of (sum g, sum g*xhat)
// ---------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void affine_bwd_partial_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                                  int64_t M, int C, int cols, int rpp,
                                                                  int64_t rows_per_chunk, const float* scale,
                                                                  const float* shift, int act, const float* mean,
                                                                  const float* rstd, float* __restrict__ ws) {
  __shared__ float red[2][256 * 4];
  const int t = threadIdx.x;
  const int col = t % cols, rg = t / cols;
  const bool active = rg < rpp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = min(M, r0 + rows_per_chunk);
  float sg[V], sgx[V], sc[V], sh[V], mu[V], rs[V];
  for (int j = 0; j < V; ++j) {
    sg[j] = sgx[j] = 0.f;
    const int c = col * V + j;
    sc[j] = scale[c];
    sh[j] = shift[c];
    mu[j] = mean[c];
    rs[j] = rstd[c];
  }
  if (active) {
    for (int64_t r = r0 + rg; r < r1; r += rpp) {
      typename Vec<V>::T xv = Vec<V>::load(x + r * C + col * V);
      typename Vec<V>::T gv = Vec<V>::load(dh + r * C + col * V);
      for (int j = 0; j < V; ++j) {
        const float xx = at<V>(xv, j);
        const float g = at<V>(gv, j) * act_grad(xx * sc[j] + sh[j], act);
        sg[j] += g;
        sgx[j] += g * (xx - mu[j]) * rs[j];
      }
    }
  }
  for (int j = 0; j < V; ++j) {
    red[0][t * 4 + j] = active ? sg[j] : 0.f;
    red[1][t * 4 + j] = active ? sgx[j] : 0.f;
  }
  __syncthreads();
  if (t < cols) {
    float* out = ws + (size_t)blockIdx.x * 2 * C;
    for (int j = 0; j < V; ++j) {
      float a = 0.f, b = 0.f;
      for (int g = 0; g < rpp; ++g) {
        a += red[0][(g * cols + t) * 4 + j];
        b += red[1][(g * cols + t) * 4 + j];
      }
      out[t * V + j] = a;
      out[C + t * V + j] = b;
    }
  }
}

// coef layout (tail of ws): [2][C] = (mean g, mean g*xhat)
__global__ __launch_bounds__(256) void affine_bwd_finalize_kernel(const float* __restrict__ ws, int chunks, int C, int64_t M,
                                                                  float* dgamma, float* dbeta, float* coef) {
  __shared__ double part[4][2];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float db0 = dbeta ? dbeta[c] : 0.f, dg0 = dgamma ? dgamma[c] : 0.f;  // requested beside the partial rows, not after the barrier
  float af = 0.f, bf = 0.f;
#pragma unroll 4
  for (int k = tid; k < chunks; k += 256) {
    af += ws[(size_t)k * 2 * C + c];
    bf += ws[(size_t)k * 2 * C + C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    af += __shfl_xor(af, o, 64);
    bf += __shfl_xor(bf, o, 64);
  }
  if (lane == 0) {
    part[wv][0] = (double)af;
    part[wv][1] = (double)bf;
  }
  __syncthreads();
  if (tid != 0) return;
  const double a = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
  const double b = (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]);
  if (dbeta) dbeta[c] = db0 + (float)a;
  if (dgamma) dgamma[c] = dg0 + (float)b;
  coef[c] = (float)(a / (double)M);
  coef[C + c] = (float)(b / (double)M);
}

// U: rows per thread and round trip (independent loads in flight)
template <int V, int U = 4>
__global__ __launch_bounds__(256) void affine_bwd_apply_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                                int M, int C, int cols, int rpp, const float* scale,
                                                                const float* shift, int act, const float* mean,
                                                                const float* rstd, const float* coef,
                                                                const float* drop, int rows_per_n,
                                                                const float* __restrict__ add, float* __restrict__ dx, int dtypes) {
  const int t = threadIdx.x, col = t % cols, rg = t / cols;
  if (rg >= rpp) return;
  const int c = col * V;
  // dtypes (V == 4 only): bit 0 dh, bit 1 x, bit 2 dx stored as bf16 (lvae_affine_act_bwd_parts_f32); `add` is always fp32
  const bool dh_bf = V >= 4 && (dtypes & 1), x_bf = V >= 4 && (dtypes & 2), dx_bf = V >= 4 && (dtypes & 4);
  float sc[V], sh[V], mu[V], rs[V], c1[V], c2[V];
  for (int j = 0; j < V; ++j) {
    sc[j] = scale ? scale[c + j] : 1.f;
    sh[j] = scale ? shift[c + j] : 0.f;
    mu[j] = coef ? mean[c + j] : 0.f;
    rs[j] = coef ? rstd[c + j] : 0.f;
    c1[j] = coef ? coef[c + j] : 0.f;
    c2[j] = coef ? coef[C + c + j] : 0.f;
  }
  // four rows per round trip: the loads of a pass are independent, so a thread pays the memory latency once per four rows (a rolled
  // row loop paid it per row: 15.3 us for the 50 MB of a 256x16x16x64 layer)
  const int stride = gridDim.x * rpp;
  int row = blockIdx.x * rpp + rg;
  for (; row + (U - 1) * stride < M; row += U * stride) {
    typename Vec<V>::T xv[U], gv[U], av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(row + u * stride) * C + c;
      xv[u] = Vec<V>::load_dt(x, off, x_bf);
      gv[u] = Vec<V>::load_dt(dh, off, dh_bf);
      if (add) av[u] = Vec<V>::load(add + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rw = row + u * stride;
      const float* dr = drop ? drop + (size_t)(rw / rows_per_n) * C + c : nullptr;
      for (int j = 0; j < V; ++j) {
        const float xx = at<V>(xv[u], j);
        float g = at<V>(gv[u], j) * act_grad(xx * sc[j] + sh[j], act);
        g = (g - c1[j] - (xx - mu[j]) * rs[j] * c2[j]) * sc[j];
        if (dr) g *= dr[j];
        if (add) g += at<V>(av[u], j);
        at<V>(gv[u], j) = g;
      }
      Vec<V>::store_dt(dx, (size_t)rw * C + c, gv[u], dx_bf);
    }
  }
  for (; row < M; row += stride) {
    const size_t off = (size_t)row * C + c;
    typename Vec<V>::T xv = Vec<V>::load_dt(x, off, x_bf);
    typename Vec<V>::T gv = Vec<V>::load_dt(dh, off, dh_bf);
    typename Vec<V>::T av;
    if (add) av = Vec<V>::load(add + off);
    const float* dr = drop ? drop + (size_t)(row / rows_per_n) * C + c : nullptr;
    for (int j = 0; j < V; ++j) {
      const float xx = at<V>(xv, j);
      float g = at<V>(gv, j) * act_grad(xx * sc[j] + sh[j], act);
      g = (g - c1[j] - (xx - mu[j]) * rs[j] * c2[j]) * sc[j];
      if (dr) g *= dr[j];
      if (add) g += at<V>(av, j);
      at<V>(gv, j) = g;
    }
    Vec<V>::store_dt(dx, off, gv, dx_bf);
  }
}

// Low-resolution levels: few partial rows (<= 128) and few apply workgroups, so every workgroup sums the partials itself (a few
// KB out of L2) instead of waiting for a separate finalize launch; workgroup 0 also accumulates dgamma / dbeta. C % 4 == 0,
// C <= 256; the summation order is the same in every workgroup (bitwise identical coefficients).
__global__ __launch_bounds__(256) void affine_bwd_apply_parts_kernel(const float* __restrict__ parts, int rows, const float* __restrict__ dh,
                                                                     const float* __restrict__ x, int M, int C, int cols, int rpp,
                                                                     const float* scale, const float* shift, int act,
                                                                     const float* mean, const float* rstd, float* dgamma,
                                                                     float* dbeta, const float* drop, int rows_per_n,
                                                                     const float* __restrict__ add, float* __restrict__ dx, int dtypes) {
  const bool dh_bf = dtypes & 1, x_bf = dtypes & 2, dx_bf = dtypes & 4;  // storage of dh, x, dx (bf16 when set); `add` is fp32
  __shared__ float red[2][256 * 4];
  __shared__ float cf[2][256];
  const int t = threadIdx.x, col = t % cols, rg = t / cols;
  const int c = col * 4;
  // the first four rows of this thread are requested BEFORE the partial rows are summed: the two memory round trips overlap instead
  // of following each other (these launches are latency chains: 8.2 us whether the level has 1,024 or 16,384 pixels)
  constexpr int RP = 4;
  const int stride = gridDim.x * rpp, row0 = blockIdx.x * rpp + rg;
  F4 xv[RP], gv[RP], av[RP];
  if (rg < rpp) {
#pragma unroll
    for (int u = 0; u < RP; ++u) {
      const int row = row0 + u * stride;
      const size_t off = (size_t)(row < M ? row : 0) * C + c;
      xv[u] = Vec<4>::load_dt(x, off, x_bf);
      gv[u] = Vec<4>::load_dt(dh, off, dh_bf);
      if (add) av[u] = Vec<4>::load(add + off);
    }
  }
  // ... and so are the per-channel coefficients and workgroup 0's gradient accumulators
  float sc[4], sh[4], mu[4], rs[4];
  for (int j = 0; j < 4; ++j) {
    sc[j] = scale[c + j];
    sh[j] = shift[c + j];
    mu[j] = mean[c + j];
    rs[j] = rstd[c + j];
  }
  const bool acc_here = blockIdx.x == 0 && t < C;
  const float db0 = acc_here && dbeta ? dbeta[t] : 0.f, dg0 = acc_here && dgamma ? dgamma[t] : 0.f;
  {
    // slice rg of the partial rows for channel group col: rows rg, rg + nsl, ... (nsl = 256 / cols slices)
    const int nsl = 256 / cols;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (rg < nsl)
      for (int r = rg; r < rows; r += 4 * nsl) {  // 8 independent 16-byte loads per round trip, summed in row order
        f32x4 pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * nsl;
          const size_t o = (size_t)(rr < rows ? rr : 0) * 2 * C + c;
          pa[u] = *reinterpret_cast<const f32x4*>(parts + o);
          pb[u] = *reinterpret_cast<const f32x4*>(parts + o + C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (r + u * nsl < rows) {
            a += pa[u];
            b += pb[u];
          }
      }
    *reinterpret_cast<f32x4*>(&red[0][t * 4]) = a;
    *reinterpret_cast<f32x4*>(&red[1][t * 4]) = b;
    __syncthreads();
    if (t < C) {
      const int cg = t >> 2, j = t & 3;
      double sa = 0.0, sb = 0.0;
      for (int sl = 0; sl < nsl; ++sl) {
        sa += (double)red[0][(sl * cols + cg) * 4 + j];
        sb += (double)red[1][(sl * cols + cg) * 4 + j];
      }
      cf[0][t] = (float)(sa / (double)M);
      cf[1][t] = (float)(sb / (double)M);
      if (blockIdx.x == 0) {
        if (dbeta) dbeta[t] = db0 + (float)sa;
        if (dgamma) dgamma[t] = dg0 + (float)sb;
      }
    }
    __syncthreads();
  }
  if (rg >= rpp) return;
  float c1[4], c2[4];
  for (int j = 0; j < 4; ++j) {
    c1[j] = cf[0][c + j];
    c2[j] = cf[1][c + j];
  }
  auto apply = [&](int row, const F4& xq, F4 gq, const F4& aq) {
    const float* dr = drop ? drop + (size_t)(row / rows_per_n) * C + c : nullptr;
    for (int j = 0; j < 4; ++j) {
      const float xx = xq.v[j];
      float g = gq.v[j] * act_grad(xx * sc[j] + sh[j], act);
      g = (g - c1[j] - (xx - mu[j]) * rs[j] * c2[j]) * sc[j];
      if (dr) g *= dr[j];
      if (add) g += aq.v[j];
      gq.v[j] = g;
    }
    Vec<4>::store_dt(dx, (size_t)row * C + c, gq, dx_bf);
  };
#pragma unroll
  for (int u = 0; u < RP; ++u) {
    const int row = row0 + u * stride;
    if (row < M) apply(row, xv[u], gv[u], av[u]);
  }
  for (int row = row0 + RP * stride; row < M; row += stride) {
    const size_t off = (size_t)row * C + c;
    const F4 xq = Vec<4>::load_dt(x, off, x_bf), gq = Vec<4>::load_dt(dh, off, dh_bf);
    F4 aq = xq;
    if (add) aq = Vec<4>::load(add + off);
    apply(row, xq, gq, aq);
  }
}

// ---------------------------------------------------------------------------------------------------------
// gate
// ---------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ ab, const float* __restrict__ res,
                                                        int M, int C, int cols, int rpp, int act, float* __restrict__ out) {
  const int t = threadIdx.x, col = t % cols, rg = t / cols;
  if (rg >= rpp) return;
  const int c = col * V;
  for (int row = blockIdx.x * rpp + rg; row < M; row += gridDim.x * rpp) {
    const float* pa = ab + (size_t)row * 2 * C + c;
    typename Vec<V>::T a = Vec<V>::load(pa);
    typename Vec<V>::T b = Vec<V>::load(pa + C);
    typename Vec<V>::T r;
    if (res) r = Vec<V>::load(res + (size_t)row * C + c);
    for (int j = 0; j < V; ++j) {
      float o = act_fwd(at<V>(a, j), act) * sigmoidf_(at<V>(b, j));
      if (res) o += at<V>(r, j);
      at<V>(a, j) = o;
    }
    Vec<V>::store(out + (size_t)row * C + c, a);
  }
}

template <int V>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ ab,
                                                        int M, int C, int cols, int rpp, int act, float* __restrict__ dab) {
  const int t = threadIdx.x, col = t % cols, rg = t / cols;
  if (rg >= rpp) return;
  const int c = col * V;
  for (int row = blockIdx.x * rpp + rg; row < M; row += gridDim.x * rpp) {
    const float* pa = ab + (size_t)row * 2 * C + c;
    typename Vec<V>::T a = Vec<V>::load(pa);
    typename Vec<V>::T b = Vec<V>::load(pa + C);
    typename Vec<V>::T g = Vec<V>::load(dout + (size_t)row * C + c);
    for (int j = 0; j < V; ++j) {
      const float s = sigmoidf_(at<V>(b, j));
      const float aa = at<V>(a, j), gg = at<V>(g, j);
      at<V>(a, j) = gg * s * act_grad(aa, act);
      at<V>(b, j) = gg * act_fwd(aa, act) * s * (1.f - s);
    }
    float* pd = dab + (size_t)row * 2 * C + c;
    Vec<V>::store(pd, a);
    Vec<V>::store(pd + C, b);
  }
}

// ---------------------------------------------------------------------------------------------------------
// small glue
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_from_out_kernel(const float* dy, const float* y, int64_t n, int act,
                                                                float* dx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dx[i] = dy[i] * act_grad_from_out(y[i], act);
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

// out = (a + b) + c: gradient of a tensor with three consumers, one pass instead of two
__global__ __launch_bounds__(256) void add3_kernel(const float* a, const float* b, const float* c, int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (a[i] + b[i]) + c[i];
}

// out[0] = sum_l mean_n x[l][n]  (logp of models/lvae.py:301: the sum over layers of the batch-mean log p(z_l)); one workgroup,
// fixed summation order
__global__ __launch_bounds__(256) void sum_of_row_means_kernel(const float* __restrict__ x, int L, int N, float* out) {
  __shared__ float red[4];
  float tot = 0.f;
  for (int l = 0; l < L; ++l) {
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) s += x[(size_t)l * N + n];
    s = block_sum_256(s, red);
    tot += s / (float)N;
  }
  if (threadIdx.x == 0) out[0] = tot;
}

__global__ __launch_bounds__(256) void scale_rows_add_kernel(const float* a, const float* row_scale, int64_t rows_per_n,
                                                              int C, const float* b, int64_t total, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / C;
    const int c = (int)(i - row * C);
    float v = a[i];
    if (row_scale) v *= row_scale[(row / rows_per_n) * C + c];
    if (b) v += b[i];
    out[i] = v;
  }
}

// 64 columns per workgroup, the four waves take interleaved rows (fixed order) and meet in LDS: a (256, 256) input used to
// be ONE workgroup walking 256 dependent row loads (56 us)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int64_t R, int64_t P, float* out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (p < P) {
#pragma unroll 8
    for (int64_t r = wv; r < R; r += 4) s += x[r * P + p];
  }
  red[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && p < P) {
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[p] = accumulate ? out[p] + s : s;
  }
}

static bool vec_ok(int C, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return C % 4 == 0 && C / 4 <= 256 && al(a) && al(b) && al(c) && al(d);
}

}  // namespace lvae

using namespace lvae;

extern "C" size_t lvae_bn_stats_workspace(int64_t M, int32_t C) {
  (void)M;
  return (size_t)(kMaxChunks * 4 + 2) * (size_t)C * sizeof(float);
}

extern "C" int lvae_bn_stats_f32(const float* x, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream) {
  LVAE_REQUIRE(x && scale && shift && mean && rstd && workspace, LVAE_EINVAL, "lvae_bn_stats_f32: null pointer");
  LVAE_REQUIRE(M > 0 && C > 0 && C <= 256 * 4, LVAE_EINVAL, "lvae_bn_stats_f32: bad M=%lld C=%d", (long long)M, C);
  LVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), LVAE_EINVAL, "lvae_bn_stats_f32: running pair");
  LVAE_REQUIRE(workspace_bytes >= lvae_bn_stats_workspace(M, C), LVAE_EWORKSPACE, "lvae_bn_stats_f32: workspace");
  hipStream_t s = (hipStream_t)stream;
  float* ws = static_cast<float*>(workspace);
  const bool v4 = vec_ok(C, x);
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_bn_stats_f32: C=%d needs C%%4==0 above 256 channels", C);
  RowMap rm = row_map(C, v4 ? 4 : 1);
  const int chunks = chunk_count(M, rm.rpp);
  int64_t rpc = (M + chunks - 1) / chunks;
  rpc = (rpc + rm.rpp - 1) / rm.rpp * rm.rpp;
  const int used = (int)((M + rpc - 1) / rpc);
  if (v4)
    hipLaunchKernelGGL(bn_partial_kernel<4>, dim3(used), dim3(256), 0, s, x, M, C, rm.cols, rm.rpp, rpc, ws);
  else
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(used), dim3(256), 0, s, x, M, C, rm.cols, rm.rpp, rpc, ws);
  LVAE_LAUNCH_CHECK("bn_partial");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, ws, used, C, M, x, gamma, beta, eps, momentum,
                     running_mean, running_var, scale, shift, mean, rstd);
  LVAE_LAUNCH_CHECK("bn_finalize");
  return 0;
}

extern "C" int lvae_bn_finalize_parts_f32(const float* parts, int32_t rows, int64_t M, int32_t C, const float* pivot,
                                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                          float* running_var, float* scale, float* shift, float* mean, float* rstd, void* stream) {
  LVAE_REQUIRE(parts && pivot && scale && shift && mean && rstd && rows > 0 && M > 0 && C > 0, LVAE_EINVAL,
               "lvae_bn_finalize_parts_f32: bad arguments");
  LVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), LVAE_EINVAL, "lvae_bn_finalize_parts_f32: running pair");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, parts, rows, C, M, pivot, gamma, beta, eps,
                     momentum, running_mean, running_var, scale, shift, mean, rstd);
  LVAE_LAUNCH_CHECK("bn_finalize_parts");
  return 0;
}

extern "C" int lvae_bn_eval_coeffs_f32(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, float* scale, float* shift, void* stream) {
  LVAE_REQUIRE(C > 0 && running_mean && running_var && scale && shift, LVAE_EINVAL, "lvae_bn_eval_coeffs_f32: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  LVAE_LAUNCH_CHECK("bn_eval_coeffs");
  return 0;
}

extern "C" int lvae_affine_act_f32(const float* x, int64_t M, int32_t C, const float* scale, const float* shift,
                                   int32_t act, const float* row_scale, int64_t rows_per_n, float* y, void* stream) {
  LVAE_REQUIRE(x && y && M > 0 && C > 0, LVAE_EINVAL, "lvae_affine_act_f32: bad args");
  LVAE_REQUIRE((scale == nullptr) == (shift == nullptr), LVAE_EINVAL, "lvae_affine_act_f32: scale/shift pair");
  LVAE_REQUIRE(!row_scale || rows_per_n > 0, LVAE_EINVAL, "lvae_affine_act_f32: rows_per_n");
  hipStream_t s = (hipStream_t)stream;
  LVAE_REQUIRE(M < ((int64_t)1 << 31), LVAE_EINVAL, "lvae_affine_act_f32: too many rows");
  const bool v4 = vec_ok(C, x, y);
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_affine_act_f32: C=%d unsupported", C);
  const RowMap rm = row_map(C, v4 ? 4 : 1);
  const int grid = grid_for(M, rm.rpp * 4);
  if (v4)
    hipLaunchKernelGGL(affine_act_kernel<4>, dim3(grid), dim3(256), 0, s, x, (int)M, C, rm.cols, rm.rpp, scale, shift, act,
                       row_scale, (int)rows_per_n, y);
  else
    hipLaunchKernelGGL(affine_act_kernel<1>, dim3(grid), dim3(256), 0, s, x, (int)M, C, rm.cols, rm.rpp, scale, shift, act,
                       row_scale, (int)rows_per_n, y);
  LVAE_LAUNCH_CHECK("affine_act");
  return 0;
}

extern "C" int lvae_affine_act_bwd_f32(const float* dh, const float* x, int64_t M, int32_t C, const float* scale,
                                       const float* shift, int32_t act, int32_t bn_train, const float* mean,
                                       const float* rstd, float* dgamma, float* dbeta, const float* drop,
                                       int64_t rows_per_n, const float* add, float* dx, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  LVAE_REQUIRE(dh && x && dx && M > 0 && C > 0, LVAE_EINVAL, "lvae_affine_act_bwd_f32: bad args");
  LVAE_REQUIRE((scale == nullptr) == (shift == nullptr), LVAE_EINVAL, "lvae_affine_act_bwd_f32: scale/shift pair");
  LVAE_REQUIRE(!drop || rows_per_n > 0, LVAE_EINVAL, "lvae_affine_act_bwd_f32: rows_per_n");
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = vec_ok(C, x, dh, dx, add);
  float* coef = nullptr;
  if (bn_train) {
    LVAE_REQUIRE(scale && mean && rstd && workspace, LVAE_EINVAL, "lvae_affine_act_bwd_f32: bn_train needs stats");
    LVAE_REQUIRE(workspace_bytes >= lvae_bn_stats_workspace(M, C), LVAE_EWORKSPACE, "lvae_affine_act_bwd_f32: workspace");
    LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_affine_act_bwd_f32: C=%d unsupported", C);
    float* ws = static_cast<float*>(workspace);
    RowMap rm = row_map(C, v4 ? 4 : 1);
    const int chunks = chunk_count(M, rm.rpp);
    int64_t rpc = (M + chunks - 1) / chunks;
    rpc = (rpc + rm.rpp - 1) / rm.rpp * rm.rpp;
    const int used = (int)((M + rpc - 1) / rpc);
    coef = ws + (size_t)kMaxChunks * 4 * C;
    if (v4)
      hipLaunchKernelGGL(affine_bwd_partial_kernel<4>, dim3(used), dim3(256), 0, s, dh, x, M, C, rm.cols, rm.rpp, rpc,
                         scale, shift, act, mean, rstd, ws);
    else
      hipLaunchKernelGGL(affine_bwd_partial_kernel<1>, dim3(used), dim3(256), 0, s, dh, x, M, C, rm.cols, rm.rpp, rpc,
                         scale, shift, act, mean, rstd, ws);
    LVAE_LAUNCH_CHECK("affine_bwd_partial");
    hipLaunchKernelGGL(affine_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, ws, used, C, M, dgamma, dbeta, coef);
    LVAE_LAUNCH_CHECK("affine_bwd_finalize");
  }
  LVAE_REQUIRE(M < ((int64_t)1 << 31), LVAE_EINVAL, "lvae_affine_act_bwd_f32: too many rows");
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_affine_act_bwd_f32: C=%d unsupported", C);
  {
    const RowMap rm2 = row_map(C, v4 ? 4 : 1);
    const int grid = grid_for(M, rm2.rpp * 4);
    if (v4)
      hipLaunchKernelGGL(affine_bwd_apply_kernel<4>, dim3(grid), dim3(256), 0, s, dh, x, (int)M, C, rm2.cols, rm2.rpp, scale,
                         shift, act, mean, rstd, coef, drop, (int)rows_per_n, add, dx, 0);
    else
      hipLaunchKernelGGL(affine_bwd_apply_kernel<1>, dim3(grid), dim3(256), 0, s, dh, x, (int)M, C, rm2.cols, rm2.rpp, scale,
                         shift, act, mean, rstd, coef, drop, (int)rows_per_n, add, dx, 0);
  }
  LVAE_LAUNCH_CHECK("affine_bwd_apply");
  return 0;
}

extern "C" int lvae_affine_act_bwd_parts_f32(const float* parts, int32_t rows, const float* dh, const float* x, int64_t M, int32_t C,
                                             const float* scale, const float* shift, int32_t act, const float* mean,
                                             const float* rstd, float* dgamma, float* dbeta, const float* drop, int64_t rows_per_n,
                                             const float* add, float* dx, void* workspace, size_t workspace_bytes, int32_t dtypes,
                                             void* stream) {
  LVAE_REQUIRE(parts && rows > 0 && dh && x && dx && M > 0 && C > 0 && scale && shift && mean && rstd && workspace, LVAE_EINVAL,
               "lvae_affine_act_bwd_parts_f32: bad args");
  LVAE_REQUIRE((dtypes & ~7) == 0 && (dtypes == 0 || (C % 4 == 0 && vec_ok(C, x, dh, dx, add))), LVAE_EINVAL,
               "lvae_affine_act_bwd_parts_f32: bf16 storage needs C %% 4 == 0 and 16-byte aligned buffers");
  LVAE_REQUIRE(workspace_bytes >= (size_t)2 * C * sizeof(float), LVAE_EWORKSPACE, "lvae_affine_act_bwd_parts_f32: workspace");
  LVAE_REQUIRE(!drop || rows_per_n > 0, LVAE_EINVAL, "lvae_affine_act_bwd_parts_f32: rows_per_n");
  LVAE_REQUIRE(M < ((int64_t)1 << 31), LVAE_EINVAL, "lvae_affine_act_bwd_parts_f32: too many rows");
  hipStream_t s = (hipStream_t)stream;
  // measured on the CIFAR-15 step (one box): 128 -> 36.96 ms, 256 -> 37.28, 1024 -> 37.54: beyond 128 rows the redundant per-workgroup sums cost
  // more than the finalize launch they replace
  static const int parts_max_rows = (int)tune("LVAE_APPLY_PARTS_MAX_ROWS", 128);
  // ... except where few workgroups do the summing: the 8x8 level under LVAE_PREC_BF16 has 256 partial rows (64-pixel tiles) and 256 apply workgroups
  const bool few_wgs = rows <= 2 * parts_max_rows && M <= 16384;
  if ((rows <= parts_max_rows || few_wgs) && vec_ok(C, x, dh, dx, add) && vec_ok(C, parts, drop) && 256 % (C / 4) == 0) {
    const RowMap rm = row_map(C, 4);
    int grid = grid_for(M, rm.rpp * 4);
    // round 4: 256 partial rows summed by a CAPPED grid (the redundant sums shrink with the grid) — 33.51 ms base, 33.84 uncapped (1024
    // workgroups), 33.94 at 256, 33.40 at 512: no cap wins clearly, the finalize launch stays for > 128 rows
    static const int parts_grid_cap = (int)tune("LVAE_APPLY_PARTS_GRID", 0);  // 0: no cap (tuning builds only)
    if (parts_grid_cap > 0 && grid > parts_grid_cap) grid = parts_grid_cap;
    hipLaunchKernelGGL(affine_bwd_apply_parts_kernel, dim3(grid), dim3(256), 0, s, parts, rows, dh, x, (int)M, C, rm.cols, rm.rpp, scale,
                       shift, act, mean, rstd, dgamma, dbeta, drop, (int)rows_per_n, add, dx, (int)dtypes);
    LVAE_LAUNCH_CHECK("affine_bwd_apply_parts");
    return 0;
  }
  float* coef = static_cast<float*>(workspace);
  hipLaunchKernelGGL(affine_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, parts, rows, C, M, dgamma, dbeta, coef);
  LVAE_LAUNCH_CHECK("affine_bwd_finalize");
  const bool v4 = vec_ok(C, x, dh, dx, add);
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_affine_act_bwd_parts_f32: C=%d unsupported", C);
  const RowMap rm2 = row_map(C, v4 ? 4 : 1);
  const int grid = grid_for(M, rm2.rpp * 4);
  // (an 8-channel-per-thread form with 16-byte bf16 accesses was built and measured SLOWER than these 8-byte accesses: 19.7 vs 14.5 us at
  // 256x16x16, 57.9 vs 44.4 us at 32x32; the fp32 form takes 15.6 / 48.6 us)
  if (v4)
    hipLaunchKernelGGL(affine_bwd_apply_kernel<4>, dim3(grid), dim3(256), 0, s, dh, x, (int)M, C, rm2.cols, rm2.rpp, scale, shift,
                       act, mean, rstd, coef, drop, (int)rows_per_n, add, dx, (int)dtypes);
  else
    hipLaunchKernelGGL(affine_bwd_apply_kernel<1>, dim3(grid), dim3(256), 0, s, dh, x, (int)M, C, rm2.cols, rm2.rpp, scale, shift,
                       act, mean, rstd, coef, drop, (int)rows_per_n, add, dx, 0);
  LVAE_LAUNCH_CHECK("affine_bwd_apply");
  return 0;
}

extern "C" int lvae_gate_fwd_f32(const float* ab, const float* res, int64_t M, int32_t C, int32_t act, float* out,
                                 void* stream) {
  LVAE_REQUIRE(ab && out && M > 0 && C > 0, LVAE_EINVAL, "lvae_gate_fwd_f32: bad args");
  hipStream_t s = (hipStream_t)stream;
  LVAE_REQUIRE(M < ((int64_t)1 << 31), LVAE_EINVAL, "lvae_gate_fwd_f32: too many rows");
  const bool v4 = vec_ok(C, ab, res, out);
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_gate_fwd_f32: C=%d unsupported", C);
  const RowMap rm = row_map(C, v4 ? 4 : 1);
  const int grid = grid_for(M, rm.rpp * 4);
  if (v4) hipLaunchKernelGGL(gate_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, ab, res, (int)M, C, rm.cols, rm.rpp, act, out);
  else hipLaunchKernelGGL(gate_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, ab, res, (int)M, C, rm.cols, rm.rpp, act, out);
  LVAE_LAUNCH_CHECK("gate_fwd");
  return 0;
}

extern "C" int lvae_gate_bwd_f32(const float* dout, const float* ab, int64_t M, int32_t C, int32_t act, float* dab,
                                 void* stream) {
  LVAE_REQUIRE(dout && ab && dab && M > 0 && C > 0, LVAE_EINVAL, "lvae_gate_bwd_f32: bad args");
  hipStream_t s = (hipStream_t)stream;
  LVAE_REQUIRE(M < ((int64_t)1 << 31), LVAE_EINVAL, "lvae_gate_bwd_f32: too many rows");
  const bool v4 = vec_ok(C, ab, dout, dab);
  LVAE_REQUIRE(v4 || C <= 256, LVAE_EINVAL, "lvae_gate_bwd_f32: C=%d unsupported", C);
  const RowMap rm = row_map(C, v4 ? 4 : 1);
  const int grid = grid_for(M, rm.rpp * 4);
  if (v4) hipLaunchKernelGGL(gate_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, dout, ab, (int)M, C, rm.cols, rm.rpp, act, dab);
  else hipLaunchKernelGGL(gate_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, dout, ab, (int)M, C, rm.cols, rm.rpp, act, dab);
  LVAE_LAUNCH_CHECK("gate_bwd");
  return 0;
}

extern "C" int lvae_act_bwd_from_out_f32(const float* dy, const float* y, int64_t n, int32_t act, float* dx, void* stream) {
  LVAE_REQUIRE(dy && y && dx && n > 0, LVAE_EINVAL, "lvae_act_bwd_from_out_f32: bad args");
  hipLaunchKernelGGL(act_bwd_from_out_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n, act, dx);
  LVAE_LAUNCH_CHECK("act_bwd_from_out");
  return 0;
}

extern "C" int lvae_add_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
  LVAE_REQUIRE(a && b && out && n > 0, LVAE_EINVAL, "lvae_add_f32: bad args");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
  LVAE_LAUNCH_CHECK("add");
  return 0;
}

extern "C" int lvae_add3_f32(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream) {
  LVAE_REQUIRE(a && b && c && out && n > 0, LVAE_EINVAL, "lvae_add3_f32: bad args");
  hipLaunchKernelGGL(add3_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, n, out);
  LVAE_LAUNCH_CHECK("add3");
  return 0;
}

extern "C" int lvae_sum_of_row_means_f32(const float* x, int32_t L, int32_t N, float* out, void* stream) {
  LVAE_REQUIRE(x && out && L > 0 && N > 0, LVAE_EINVAL, "lvae_sum_of_row_means_f32: bad args");
  hipLaunchKernelGGL(sum_of_row_means_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, L, N, out);
  LVAE_LAUNCH_CHECK("sum_of_row_means");
  return 0;
}

extern "C" int lvae_scale_rows_add_f32(const float* a, const float* row_scale, int64_t rows_per_n, int32_t C,
                                       const float* b, int64_t M, float* out, void* stream) {
  LVAE_REQUIRE(a && out && M > 0 && C > 0 && (!row_scale || rows_per_n > 0), LVAE_EINVAL, "lvae_scale_rows_add_f32: bad args");
  const int64_t total = M * C;
  hipLaunchKernelGGL(scale_rows_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, row_scale,
                     rows_per_n, C, b, total, out);
  LVAE_LAUNCH_CHECK("scale_rows_add");
  return 0;
}

extern "C" int lvae_colsum_f32(const float* x, int64_t R, int64_t P, float* out, int32_t accumulate, void* stream) {
  LVAE_REQUIRE(x && out && R > 0 && P > 0, LVAE_EINVAL, "lvae_colsum_f32: bad args");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, R, P, out,
                     accumulate);
  LVAE_LAUNCH_CHECK("colsum");
  return 0;
}
