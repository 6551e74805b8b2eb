// Geometry (bilinear x2, centred pad/crop with layout change), KL bookkeeping + free bits, ELBO/loss assembly,
// flat-arena Adamax and L2 norm, Philox noise, and the library-level API (version, last error).
#include <stdarg.h>

#include "lvae_common.h"

namespace lvae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------------------
// bilinear x2, align_corners=False:  src = (dst + 0.5)/2 - 0.5 clamped at 0
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int o, int limit, int& i0, int& i1, float& lam) {
  float s = 0.5f * (o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = min(i0 + 1, limit - 1);
  lam = s - (float)i0;
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C4,
                                                              float* __restrict__ y) {
  const int64_t total = (int64_t)N * 2 * H * 2 * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int ow = (int)(r % (2 * W));
    r /= 2 * W;
    const int oh = (int)(r % (2 * H));
    const int n = (int)(r / (2 * H));
    int h0, h1, w0, w1;
    float lh, lw;
    up2_src(oh, H, h0, h1, lh);
    up2_src(ow, W, w0, w1, lw);
    const f32x4* xp = reinterpret_cast<const f32x4*>(x) + (size_t)n * H * W * C4;
    const f32x4 v00 = xp[((size_t)h0 * W + w0) * C4 + c], v01 = xp[((size_t)h0 * W + w1) * C4 + c];
    const f32x4 v10 = xp[((size_t)h1 * W + w0) * C4 + c], v11 = xp[((size_t)h1 * W + w1) * C4 + c];
    reinterpret_cast<f32x4*>(y)[i] = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
  }
}

// gather form of the adjoint: input pixel (ih, iw) collects from the <= 3x3 outputs that read it
__device__ __forceinline__ int up2_adjoint(int i, int limit, int* o, float* w) {
  int cnt = 0;
  for (int cand = 2 * i - 2; cand <= 2 * i + 2; ++cand) {
    if (cand < 0 || cand >= 2 * limit) continue;
    int i0, i1;
    float lam;
    up2_src(cand, limit, i0, i1, lam);
    float ww = 0.f;
    if (i0 == i) ww += 1.f - lam;
    if (i1 == i) ww += lam;
    if (ww != 0.f) {
      o[cnt] = cand;
      w[cnt] = ww;
      ++cnt;
    }
  }
  return cnt;
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C4,
                                                              float* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int iw = (int)(r % W);
    r /= W;
    const int ih = (int)(r % H);
    const int n = (int)(r / H);
    int oh[5], ow[5];
    float wh[5], ww[5];
    const int nh = up2_adjoint(ih, H, oh, wh), nw = up2_adjoint(iw, W, ow, ww);
    const f32x4* yp = reinterpret_cast<const f32x4*>(dy) + (size_t)n * 4 * H * W * C4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) acc += (wh[a] * ww[b]) * yp[((size_t)oh[a] * 2 * W + ow[b]) * C4 + c];
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// centred zero pad / centre crop with optional NCHW <-> NHWC change
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_crop_kernel(const float* __restrict__ x, int N, int C, int H, int W,
                                                        int src_nchw, float* __restrict__ y, int OH, int OW, int dst_nchw) {
  const int offh = OH >= H ? (OH - H) / 2 : -((H - OH) / 2);
  const int offw = OW >= W ? (OW - W) / 2 : -((W - OW) / 2);
  const int64_t total = (int64_t)N * C * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int n, c, oh, ow;
    int64_t r = i;
    if (dst_nchw) {
      ow = (int)(r % OW); r /= OW;
      oh = (int)(r % OH); r /= OH;
      c = (int)(r % C);
      n = (int)(r / C);
    } else {
      c = (int)(r % C); r /= C;
      ow = (int)(r % OW); r /= OW;
      oh = (int)(r % OH);
      n = (int)(r / OH);
    }
    const int ih = oh - offh, iw = ow - offw;
    float v = 0.f;
    if (ih >= 0 && ih < H && iw >= 0 && iw < W)
      v = src_nchw ? x[(((size_t)n * C + c) * H + ih) * W + iw] : x[(((size_t)n * H + ih) * W + iw) * C + c];
    y[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// KL bookkeeping + free bits (models/lvae.py:192-198; boilr free_bits_kl restated)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kl_book_fwd_kernel(const float* __restrict__ kl, int L, int N, float free_bits,
                                                           float* kl_sep, float* kl_avg, float* scalars) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  float kl_loss = 0.f;
  for (int l = 0; l < L; ++l) {
    float s = 0.f, sc = 0.f;
    for (int n = t; n < N; n += 256) {
      const float v = kl[(size_t)l * N + n];
      s += v;
      sc += free_bits < 1e-6f ? v : fmaxf(v, free_bits);
    }
    s = block_sum_256(s, red);
    sc = block_sum_256(sc, red);
    if (t == 0) kl_avg[l] = s / (float)N;
    kl_loss += sc / (float)N;
  }
  float tot = 0.f;
  for (int n = t; n < N; n += 256) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += kl[(size_t)l * N + n];
    kl_sep[n] = s;
    tot += s;
  }
  tot = block_sum_256(tot, red);
  if (t == 0) {
    scalars[0] = kl_loss;
    scalars[1] = tot / (float)N;
  }
}

__global__ __launch_bounds__(256) void kl_book_bwd_kernel(const float* __restrict__ kl, int L, int N, float free_bits,
                                                           const float* g_sep, const float* g_avg, const float* g_scalars,
                                                           float* __restrict__ dkl) {
  const float g_loss = g_scalars ? g_scalars[0] : 0.f, g_kl = g_scalars ? g_scalars[1] : 0.f;
  const float invn = 1.f / (float)N;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L * N; i += gridDim.x * 256) {
    const int l = i / N, n = i - l * N;
    float g = g_kl * invn;
    if (g_sep) g += g_sep[n];
    if (g_avg) g += g_avg[l] * invn;
    if (free_bits < 1e-6f || kl[i] >= free_bits) g += g_loss * invn;
    dkl[i] = g;
  }
}

// experiment/experiment_manager.py:329-344
__global__ __launch_bounds__(256) void elbo_loss_fwd_kernel(const float* __restrict__ ll, const float* __restrict__ kl_sep,
                                                             const float* kl_loss, float beta, int N, float* elbo_sep,
                                                             float* scalars) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  float sl = 0.f, se = 0.f;
  for (int n = t; n < N; n += 256) {
    const float e = -(-ll[n] + kl_sep[n]);
    elbo_sep[n] = e;
    se += e;
    sl += -ll[n];
  }
  sl = block_sum_256(sl, red);
  se = block_sum_256(se, red);
  if (t == 0) {
    const float recons = sl / (float)N;
    scalars[0] = recons + kl_loss[0] * beta;
    scalars[1] = se / (float)N;
    scalars[2] = recons;
  }
}

__global__ __launch_bounds__(256) void elbo_loss_bwd_kernel(const float* g_loss, float beta, int N, float* d_ll,
                                                             float* d_kl_loss) {
  const float g = g_loss[0];
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) d_ll[n] = -g / (float)N;
  if (blockIdx.x == 0 && threadIdx.x == 0) d_kl_loss[0] = g * beta;
}

// importance-weighted bound: out[n] = logsumexp_s elbo[s][n] - log S   (evaluate.py / boilr test_procedure, restated)
__global__ __launch_bounds__(256) void iw_logmeanexp_kernel(const float* __restrict__ elbo, int S, int N, float* out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float mx = -INFINITY;
  for (int s = 0; s < S; ++s) mx = fmaxf(mx, elbo[(size_t)s * N + n]);
  double acc = 0.0;
  for (int s = 0; s < S; ++s) acc += (double)expf(elbo[(size_t)s * N + n] - mx);
  out[n] = mx + (float)log(acc) - logf((float)S);
}

// The same bound accumulated ONLINE, one sample at a time (state [3][N]: running max, sum of exp(elbo - max), plain sum), so that a
// captured top-down + likelihood graph can be replayed S times without an S x N buffer or a per-replay output pointer:
//   mode 0: init state;  mode 1: fold in one sample's elbo [N];  mode 2: iw[n] = max + log(sumexp) - log S, mean[n] = sum / S
__global__ __launch_bounds__(256) void iw_online_kernel(const float* __restrict__ elbo, float* __restrict__ state, int N, int mode, int S,
                                                        float* iw, float* mean) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  if (mode == 0) {
    state[n] = -INFINITY;
    state[N + n] = 0.f;
    state[2 * N + n] = 0.f;
  } else if (mode == 1) {
    const float e = elbo[n], m0 = state[n], m1 = fmaxf(m0, e);
    state[N + n] = state[N + n] * expf(m0 - m1) + expf(e - m1);   // exp(-inf) = 0 on the first sample
    state[n] = m1;
    state[2 * N + n] += e;
  } else {
    iw[n] = state[n] + logf(state[N + n]) - logf((float)S);
    mean[n] = state[2 * N + n] / (float)S;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Adamax over the flat arena (torch.optim.Adamax semantics) and L2 norm
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamax_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ u,
                                                      const float* __restrict__ mask, int64_t n4, float lr, float b1,
                                                      float b2, float eps, float wd, const float* gscale,
                                                      const uint64_t* step_count) {
  const float step = (float)(step_count[0] + 1);
  const float clr = lr / (1.f - powf(b1, step));
  const float gs = gscale ? gscale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], uv = reinterpret_cast<f32x4*>(u)[i];
    f32x4 mk = {1.f, 1.f, 1.f, 1.f};
    if (mask) mk = reinterpret_cast<const f32x4*>(mask)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (mk[j] == 0.f) continue;
      float gg = gv[j] * gs;
      if (wd != 0.f) gg += wd * pv[j];
      mv[j] = mv[j] + (gg - mv[j]) * (1.f - b1);  // lerp, as torch
      uv[j] = fmaxf(uv[j] * b2, fabsf(gg) + eps);
      pv[j] -= clr * mv[j] / uv[j];
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(u)[i] = uv;
  }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int cnt, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = sqrtf(acc);
}

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// counter words: (element index lo, hi, call-site id, step) -> draws of different steps / call sites never overlap
__device__ __forceinline__ void philox4(uint64_t ctr, uint32_t stream_id, uint32_t step, uint64_t seed, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream_id, c3 = step;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t r) { return ((r >> 8) + 0.5f) * (1.f / 16777216.f); }  // (0,1)

__global__ __launch_bounds__(256) void rng_fill_kernel(float* __restrict__ out, int64_t n, int kind, float lo, float hi,
                                                        uint64_t seed, const uint64_t* offset, uint64_t stream_id) {
  const uint64_t off = offset ? offset[0] : 0;
  const int64_t n4 = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint32_t r[4];
    philox4((uint64_t)i, (uint32_t)stream_id, (uint32_t)off, seed ^ (off >> 32 << 32) ^ (stream_id >> 32), r);
    float v[4];
    if (kind == 0) {
      const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
      const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u1, &s0, &c0);
      sincosf(6.283185307179586f * u3, &s1, &c1);
      v[0] = ra * c0; v[1] = ra * s0; v[2] = rb * c1; v[3] = rb * s1;
    } else if (kind == 1) {
      for (int j = 0; j < 4; ++j) v[j] = lo + (hi - lo) * u01(r[j]);
    } else {
      for (int j = 0; j < 4; ++j) v[j] = u01(r[j]) < lo ? hi : 0.f;
    }
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = v[j];
  }
}

__global__ void counter_advance_kernel(uint64_t* c, uint64_t by) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += by;
}

}  // namespace lvae

using namespace lvae;

extern "C" int lvae_abi_version(void) { return LVAE_ABI_VERSION; }
extern "C" const char* lvae_last_error(void) { return g_err; }

extern "C" int lvae_upsample2x_fwd_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream) {
  LVAE_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, LVAE_EINVAL, "lvae_upsample2x_fwd_f32: bad args");
  LVAE_REQUIRE(C % 4 == 0, LVAE_EALIGN, "lvae_upsample2x_fwd_f32: C=%d must be a multiple of 4", C);
  const int64_t total = (int64_t)N * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C / 4, y);
  LVAE_LAUNCH_CHECK("upsample2x_fwd");
  return 0;
}

extern "C" int lvae_upsample2x_bwd_f32(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx, void* stream) {
  LVAE_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, LVAE_EINVAL, "lvae_upsample2x_bwd_f32: bad args");
  LVAE_REQUIRE(C % 4 == 0, LVAE_EALIGN, "lvae_upsample2x_bwd_f32: C=%d must be a multiple of 4", C);
  const int64_t total = (int64_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, N, H, W, C / 4, dx);
  LVAE_LAUNCH_CHECK("upsample2x_bwd");
  return 0;
}

extern "C" int lvae_pad_crop_f32(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_nchw, float* y,
                                 int32_t OH, int32_t OW, int32_t dst_nchw, void* stream) {
  LVAE_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, LVAE_EINVAL, "lvae_pad_crop_f32: bad args");
  LVAE_REQUIRE((OH >= H) == (OW >= W) || OH == H || OW == W, LVAE_EINVAL, "lvae_pad_crop_f32: mixed pad/crop");
  const int64_t total = (int64_t)N * C * OH * OW;
  hipLaunchKernelGGL(pad_crop_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, src_nchw,
                     y, OH, OW, dst_nchw);
  LVAE_LAUNCH_CHECK("pad_crop");
  return 0;
}

extern "C" int lvae_kl_bookkeeping_fwd_f32(const float* kl, int32_t L, int32_t N, float free_bits, float* kl_sep,
                                           float* kl_avg_layerwise, float* scalars, void* stream) {
  LVAE_REQUIRE(kl && kl_sep && kl_avg_layerwise && scalars && L > 0 && N > 0, LVAE_EINVAL, "lvae_kl_bookkeeping_fwd_f32: bad args");
  hipLaunchKernelGGL(kl_book_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kl, L, N, free_bits, kl_sep,
                     kl_avg_layerwise, scalars);
  LVAE_LAUNCH_CHECK("kl_bookkeeping_fwd");
  return 0;
}

extern "C" int lvae_kl_bookkeeping_bwd_f32(const float* kl, int32_t L, int32_t N, float free_bits, const float* g_sep,
                                           const float* g_avg, const float* g_scalars, float* dkl, void* stream) {
  LVAE_REQUIRE(kl && dkl && L > 0 && N > 0, LVAE_EINVAL, "lvae_kl_bookkeeping_bwd_f32: bad args");
  hipLaunchKernelGGL(kl_book_bwd_kernel, dim3(grid_for((int64_t)L * N, 256)), dim3(256), 0, (hipStream_t)stream, kl, L, N,
                     free_bits, g_sep, g_avg, g_scalars, dkl);
  LVAE_LAUNCH_CHECK("kl_bookkeeping_bwd");
  return 0;
}

extern "C" int lvae_elbo_loss_fwd_f32(const float* ll, const float* kl_sep, const float* kl_loss, float beta, int32_t N,
                                      float* elbo_sep, float* scalars, void* stream) {
  LVAE_REQUIRE(ll && kl_sep && kl_loss && elbo_sep && scalars && N > 0, LVAE_EINVAL, "lvae_elbo_loss_fwd_f32: bad args");
  hipLaunchKernelGGL(elbo_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ll, kl_sep, kl_loss, beta, N,
                     elbo_sep, scalars);
  LVAE_LAUNCH_CHECK("elbo_loss_fwd");
  return 0;
}

extern "C" int lvae_elbo_loss_bwd_f32(const float* g_loss, float beta, int32_t N, float* d_ll, float* d_kl_loss, void* stream) {
  LVAE_REQUIRE(g_loss && d_ll && d_kl_loss && N > 0, LVAE_EINVAL, "lvae_elbo_loss_bwd_f32: bad args");
  hipLaunchKernelGGL(elbo_loss_bwd_kernel, dim3(grid_for(N, 256)), dim3(256), 0, (hipStream_t)stream, g_loss, beta, N, d_ll,
                     d_kl_loss);
  LVAE_LAUNCH_CHECK("elbo_loss_bwd");
  return 0;
}

extern "C" int lvae_adamax_step_f32(float* p, const float* g, float* exp_avg, float* exp_inf, const float* mask, int64_t n,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, const float* gscale,
                                    const uint64_t* step_count, void* stream) {
  LVAE_REQUIRE(p && g && exp_avg && exp_inf && step_count && n > 0, LVAE_EINVAL, "lvae_adamax_step_f32: bad args");
  LVAE_REQUIRE(n % 4 == 0, LVAE_EALIGN, "lvae_adamax_step_f32: arena length %lld must be a multiple of 4", (long long)n);
  hipLaunchKernelGGL(adamax_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_inf,
                     mask, n / 4, lr, beta1, beta2, eps, weight_decay, gscale, step_count);
  LVAE_LAUNCH_CHECK("adamax_step");
  return 0;
}

extern "C" size_t lvae_sumsq_workspace(int64_t n) {
  (void)n;
  return 2048 * sizeof(float);
}

extern "C" int lvae_l2norm_f32(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  LVAE_REQUIRE(x && out && workspace && n > 0, LVAE_EINVAL, "lvae_l2norm_f32: bad args");
  LVAE_REQUIRE(workspace_bytes >= lvae_sumsq_workspace(n), LVAE_EWORKSPACE, "lvae_l2norm_f32: workspace");
  const int g = grid_for(n, 256 * 8);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(g), dim3(256), 0, s, x, n, static_cast<float*>(workspace));
  LVAE_LAUNCH_CHECK("sumsq_partial");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), g, out);
  LVAE_LAUNCH_CHECK("sumsq_final");
  return 0;
}

extern "C" int lvae_rng_fill_f32(float* out, int64_t n, int32_t kind, float lo, float hi, uint64_t seed,
                                 const uint64_t* offset, uint64_t stream_id, void* stream) {
  LVAE_REQUIRE(out && n > 0 && kind >= 0 && kind <= 2, LVAE_EINVAL, "lvae_rng_fill_f32: bad args");
  hipLaunchKernelGGL(rng_fill_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n, kind, lo,
                     hi, seed, offset, stream_id);
  LVAE_LAUNCH_CHECK("rng_fill");
  return 0;
}

// out[0:n] = value: the gradient arena's zero_grad (16-byte stores; n * 4 bytes need not be a multiple of 16)
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ out, int64_t n, float value) {
  const int64_t n4 = n >> 2;
  const f32x4 v = {value, value, value, value};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) reinterpret_cast<f32x4*>(out)[i] = v;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = value;
}

extern "C" int lvae_fill_f32(float* out, int64_t n, float value, void* stream) {
  LVAE_REQUIRE(out != nullptr && n > 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, LVAE_EINVAL, "lvae_fill_f32: null / misaligned buffer");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n >> 2, 256 * 4)), dim3(256), 0, (hipStream_t)stream, out, n, value);
  LVAE_LAUNCH_CHECK("fill");
  return 0;
}

extern "C" int lvae_counter_advance(uint64_t* counter, uint64_t by, void* stream) {
  LVAE_REQUIRE(counter != nullptr, LVAE_EINVAL, "lvae_counter_advance: null counter");
  hipLaunchKernelGGL(counter_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, by);
  LVAE_LAUNCH_CHECK("counter_advance");
  return 0;
}

extern "C" int lvae_iw_online_f32(const float* elbo, float* state, int32_t N, int32_t mode, int32_t S, float* iw, float* mean, void* stream) {
  LVAE_REQUIRE(state && N > 0 && mode >= 0 && mode <= 2, LVAE_EINVAL, "lvae_iw_online_f32: bad args");
  LVAE_REQUIRE(mode != 1 || elbo, LVAE_EINVAL, "lvae_iw_online_f32: elbo missing");
  LVAE_REQUIRE(mode != 2 || (iw && mean && S > 0), LVAE_EINVAL, "lvae_iw_online_f32: outputs missing");
  hipLaunchKernelGGL(iw_online_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, elbo, state, N, mode, S, iw, mean);
  LVAE_LAUNCH_CHECK("iw_online");
  return 0;
}

extern "C" int lvae_iw_logmeanexp_f32(const float* elbo, int32_t S, int32_t N, float* out, void* stream) {
  LVAE_REQUIRE(elbo && out && S > 0 && N > 0, LVAE_EINVAL, "lvae_iw_logmeanexp_f32: bad args");
  hipLaunchKernelGGL(iw_logmeanexp_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, elbo, S, N, out);
  LVAE_LAUNCH_CHECK("iw_logmeanexp");
  return 0;
}
