// Likelihood heads, elementwise part (lib/likelihoods.py): Bernoulli and the 10-component discretized mixture of
// logistics (log-likelihood with its analytic parameter gradient, and the Gumbel-max / logistic sampler).
// The DMoL kernels stage the 100 parameters of each pixel through LDS with coalesced 16-B global accesses and
// then work one pixel per lane on a padded row (stride 101 floats -> conflict-free column walks).
#include "lvae_common.h"

namespace lvae {

// ---------------------------------------------------------------------------------------------------------
// Bernoulli (lib/likelihoods.py:60-78, 385-388)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bernoulli_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ x,
                                                             const float* __restrict__ u, int64_t P, float* mean,
                                                             float* mode, float* sample, float* ll, float* dll) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const size_t base = (size_t)n * P;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < P; i += 256) {
    const float m = sigmoidf_(logits[base + i]);
    if (mean) mean[base + i] = m;
    if (mode) mode[base + i] = rintf(m);  // torch.round: half to even
    if (sample) sample[base + i] = u[base + i] < m ? 1.f : 0.f;
    if (x) {
      const float xv = x[base + i];
      // F.binary_cross_entropy clamps each log term at -100
      const float l1 = fmaxf(logf(m), -100.f), l0 = fmaxf(logf(1.f - m), -100.f);
      acc += xv * l1 + (1.f - xv) * l0;
      if (dll) {
        // autograd of BCE then sigmoid: -(m - x) / max(m(1-m), 1e-12) * m(1-m)
        const float v = m * (1.f - m);
        dll[base + i] = -(m - xv) / fmaxf(v, 1e-12f) * v;
      }
    }
  }
  if (x) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ll[n] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Discretized mixture of logistics (lib/likelihoods.py:291-382) with NMIX components (the reference's default and only use: 10;
// lib/likelihoods.py:183-202 takes any count, so the kernels are instantiated for 1-6, 8, 10, 12, 16 and 20)
// parameter layout per pixel: [0,NMIX) logits; colour c block at NMIX+3*NMIX*c: [+0,NMIX) means, [+NMIX,2NMIX) log-scales, [+2NMIX,3NMIX) coeffs
// ---------------------------------------------------------------------------------------------------------

struct ColourTerm {
  float lp, dmean, dls;
};

__device__ __forceinline__ ColourTerm dmol_colour(float x, float mean, float ls_raw) {
  ColourTerm r;
  const float ls = fmaxf(ls_raw, -7.f);
  const float inv_s = expf(-ls);
  const float cen = x - mean;
  const float plus = inv_s * (cen + 1.f / 255.f), mn = inv_s * (cen - 1.f / 255.f), mid = inv_s * cen;
  float dplus = 0.f, dmin = 0.f, dmid = 0.f, dls_direct = 0.f;
  if (x < -0.999f) {
    r.lp = plus - softplusf_(plus);
    dplus = 1.f - sigmoidf_(plus);
  } else if (x > 0.999f) {
    r.lp = -softplusf_(mn);
    dmin = -sigmoidf_(mn);
  } else {
    const float sp = sigmoidf_(plus), sm = sigmoidf_(mn);
    const float cdf_delta = sp - sm;
    if (cdf_delta > 1e-5f) {
      r.lp = logf(fmaxf(cdf_delta, 1e-12f));
      dplus = sp * (1.f - sp) / cdf_delta;
      dmin = -sm * (1.f - sm) / cdf_delta;
    } else {
      r.lp = mid - ls - 2.f * softplusf_(mid) - 4.8481163645876185f;  // log(127.5)
      dmid = 1.f - 2.f * sigmoidf_(mid);
      dls_direct = -1.f;
    }
  }
  // d(plus)/d(mean) = -inv_s ; d(plus)/d(ls) = -plus (same for min, mid)
  r.dmean = -inv_s * (dplus + dmin + dmid);
  r.dls = (ls_raw >= -7.f) ? (-(plus * dplus + mn * dmin + mid * dmid) + dls_direct) : 0.f;
  return r;
}

template <int NMIX, int DM_PIX>  // DM_PIX pixels per workgroup: 128 while their parameter rows fit the 64 KB of static LDS, else 64
__global__ __launch_bounds__(DM_PIX) void dmol_ll_kernel(const float* __restrict__ l, const float* __restrict__ x,
                                                          int64_t npix, float* __restrict__ ll_pix,
                                                          float* __restrict__ dl) {
  constexpr int NP = 10 * NMIX, NPS = NP + 1;  // parameters per pixel, padded LDS row
  __shared__ float buf[DM_PIX * NPS];
  const int t = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * DM_PIX;
  const int cnt = (int)min((int64_t)DM_PIX, npix - p0);
  // coalesced stage-in: cnt*100 contiguous floats
  const float* src = l + p0 * NP;
  for (int i = t; i < cnt * NP; i += DM_PIX) buf[(i / NP) * NPS + (i % NP)] = src[i];
  __syncthreads();
  if (t < cnt) {
    float* row = buf + t * NPS;
    const int64_t pix = p0 + t;
    const float x0 = 2.f * x[pix * 3 + 0] - 1.f, x1 = 2.f * x[pix * 3 + 1] - 1.f, x2 = 2.f * x[pix * 3 + 2] - 1.f;
    float lpk[NMIX];
    float lmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < NMIX; ++k) lmax = fmaxf(lmax, row[k]);
    float lse = 0.f;
#pragma unroll
    for (int k = 0; k < NMIX; ++k) lse += expf(row[k] - lmax);
    lse = lmax + logf(lse);
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < NMIX; ++k) {
      const float c0 = tanhf(row[NMIX + 2 * NMIX + k]);
      const float c1 = tanhf(row[NMIX + 3 * NMIX + 2 * NMIX + k]);
      const float c2 = tanhf(row[NMIX + 6 * NMIX + 2 * NMIX + k]);
      const float m0 = row[NMIX + k];
      const float m1 = row[NMIX + 3 * NMIX + k] + c0 * x0;
      const float m2 = row[NMIX + 6 * NMIX + k] + c1 * x0 + c2 * x1;
      const ColourTerm t0 = dmol_colour(x0, m0, row[NMIX + NMIX + k]);
      const ColourTerm t1 = dmol_colour(x1, m1, row[NMIX + 3 * NMIX + NMIX + k]);
      const ColourTerm t2 = dmol_colour(x2, m2, row[NMIX + 6 * NMIX + NMIX + k]);
      lpk[k] = t0.lp + t1.lp + t2.lp + (row[k] - lse);
      best = fmaxf(best, lpk[k]);
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NMIX; ++k) se += expf(lpk[k] - best);
    const float llp = best + logf(se);
    ll_pix[pix] = llp;
    if (dl) {
      // second sweep: responsibilities w_k and the chain rule; overwrites the row in place
#pragma unroll
      for (int k = 0; k < NMIX; ++k) {
        const float w = expf(lpk[k] - llp);
        const float logit = row[k];
        const float c0 = tanhf(row[NMIX + 2 * NMIX + k]);
        const float c1 = tanhf(row[NMIX + 3 * NMIX + 2 * NMIX + k]);
        const float c2 = tanhf(row[NMIX + 6 * NMIX + 2 * NMIX + k]);
        const float m0 = row[NMIX + k];
        const float m1 = row[NMIX + 3 * NMIX + k] + c0 * x0;
        const float m2 = row[NMIX + 6 * NMIX + k] + c1 * x0 + c2 * x1;
        const ColourTerm t0 = dmol_colour(x0, m0, row[NMIX + NMIX + k]);
        const ColourTerm t1 = dmol_colour(x1, m1, row[NMIX + 3 * NMIX + NMIX + k]);
        const ColourTerm t2 = dmol_colour(x2, m2, row[NMIX + 6 * NMIX + NMIX + k]);
        row[k] = w - expf(logit - lse);
        row[NMIX + k] = w * t0.dmean;
        row[NMIX + NMIX + k] = w * t0.dls;
        row[NMIX + 2 * NMIX + k] = w * t1.dmean * x0 * (1.f - c0 * c0);
        row[NMIX + 3 * NMIX + k] = w * t1.dmean;
        row[NMIX + 3 * NMIX + NMIX + k] = w * t1.dls;
        row[NMIX + 3 * NMIX + 2 * NMIX + k] = w * t2.dmean * x0 * (1.f - c1 * c1);
        row[NMIX + 6 * NMIX + k] = w * t2.dmean;
        row[NMIX + 6 * NMIX + NMIX + k] = w * t2.dls;
        row[NMIX + 6 * NMIX + 2 * NMIX + k] = w * t2.dmean * x1 * (1.f - c2 * c2);
      }
    }
  }
  if (dl) {
    __syncthreads();
    float* dst = dl + p0 * NP;
    for (int i = t; i < cnt * NP; i += DM_PIX) dst[i] = buf[(i / NP) * NPS + (i % NP)];
  }
}

// ll[n] = sum over the HW pixels of sample n (fixed order)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ v, int64_t P, float* out) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < P; i += 256) acc += v[(size_t)n * P + i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[n] = acc;
}

// lib/stochastic.py:141-206 and the rescale/clamp of lib/likelihoods.py:221-225
template <int NMIX>
__global__ __launch_bounds__(256) void dmol_sample_kernel(const float* __restrict__ l, const float* __restrict__ u_mix,
                                                           const float* __restrict__ u_log, int64_t npix,
                                                           float* __restrict__ sample) {
  constexpr int NP = 10 * NMIX;
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    const float* row = l + pix * NP;
    int sel = 0;
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < NMIX; ++k) {
      const float g = row[k] - logf(-logf(u_mix[pix * NMIX + k]));
      if (g > best) {
        best = g;
        sel = k;
      }
    }
    float xs[3], co[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mean = row[NMIX + 3 * NMIX * c + sel];
      const float ls = fmaxf(row[NMIX + 3 * NMIX * c + NMIX + sel], -7.f);
      co[c] = tanhf(row[NMIX + 3 * NMIX * c + 2 * NMIX + sel]);
      const float u = u_log[pix * 3 + c];
      xs[c] = mean + expf(ls) * (logf(u) - logf(1.f - u));
    }
    const float x0 = fminf(fmaxf(xs[0], -1.f), 1.f);
    const float x1 = fminf(fmaxf(xs[1] + co[0] * x0, -1.f), 1.f);
    const float x2 = fminf(fmaxf(xs[2] + co[1] * x0 + co[2] * x1, -1.f), 1.f);
    sample[pix * 3 + 0] = fminf(fmaxf((x0 + 1.f) * 0.5f, 0.f), 1.f);
    sample[pix * 3 + 1] = fminf(fmaxf((x1 + 1.f) * 0.5f, 0.f), 1.f);
    sample[pix * 3 + 2] = fminf(fmaxf((x2 + 1.f) * 0.5f, 0.f), 1.f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Gaussian head (lib/likelihoods.py:81-114, log_normal :391-411). params [N][P][2C] = (mean | logvar) per pixel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_fwd_kernel(const float* __restrict__ params, const float* __restrict__ x,
                                                            const float* __restrict__ eps, int64_t P, int C, float* sample,
                                                            float* ll, float* dll) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const int64_t per = P * C;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) {
    const int64_t pix = i / C;
    const int c = (int)(i - pix * C);
    const size_t pb = ((size_t)n * P + pix) * 2 * C;
    const float mean = params[pb + c], lv = params[pb + C + c];
    if (sample) sample[(size_t)n * per + i] = mean + expf(0.5f * lv) * eps[(size_t)n * per + i];
    if (x) {
      const float d = x[(size_t)n * per + i] - mean, iv = expf(-lv);
      acc += -0.5f * (d * d * iv + lv + 1.8378770664093453f);  // log(2*pi)
      if (dll) {
        dll[pb + c] = d * iv;
        dll[pb + C + c] = -0.5f * (1.f - d * d * iv);
      }
    }
  }
  if (x) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ll[n] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Discretized logistic head (lib/likelihoods.py:117-180, log_discretized_logistic :233-288), 256 bins.
// raw [N][P][2C] = (mean_raw | log_scale_raw); outputs mean = mean_raw + 0.5, logscale = max(ls_raw - 1, -7).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void discr_logistic_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ x,
                                                                  const float* __restrict__ u, int64_t P, int C,
                                                                  float* mean_out, float* ls_out, float* sample, float* ll,
                                                                  float* dll) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const int64_t per = P * C;
  const float nb = 256.f;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) {
    const int64_t pix = i / C;
    const int c = (int)(i - pix * C);
    const size_t pb = ((size_t)n * P + pix) * 2 * C, oi = (size_t)n * per + i;
    const float mean = raw[pb + c] + 0.5f;
    const float ls_raw = raw[pb + C + c] - 1.f;
    const float ls = fmaxf(ls_raw, -7.f);
    const float scale = expf(ls);
    if (mean_out) mean_out[oi] = mean;
    if (ls_out) ls_out[oi] = ls;
    if (sample) {
      const float uu = u[oi];
      sample[oi] = fminf(fmaxf(mean + scale * (logf(uu) - logf(1.f - uu)), 0.f), 1.f);
    }
    if (x) {
      const float xs = x[oi] * (255.f / 256.f) + 1.f / 512.f;
      const float xq = floorf(xs * nb) / nb;
      const bool has_plus = xq < (nb - 1.f) / nb, has_minus = xq >= 1.f / nb;
      const float a = (xq + 1.f / nb - mean) / scale, b = (xq - mean) / scale;
      const float sa = sigmoidf_(a), sb = sigmoidf_(b);
      const float cp = has_plus ? sa : 1.f, cm = has_minus ? sb : 0.f;
      const float prob = cp - cm + 1e-7f;
      acc += logf(prob);
      if (dll) {
        const float da = has_plus ? sa * (1.f - sa) : 0.f, db = has_minus ? sb * (1.f - sb) : 0.f;
        dll[pb + c] = -(da - db) / scale / prob;
        dll[pb + C + c] = (ls_raw >= -7.f) ? -(a * da - b * db) / prob : 0.f;
      }
    }
  }
  if (x) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ll[n] = acc;
  }
}

__global__ __launch_bounds__(256) void scale_per_sample_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                                int64_t P, int64_t total, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    out[i] = a[i] * g[i / P];
}

}  // namespace lvae

using namespace lvae;

extern "C" int lvae_bernoulli_fwd_f32(const float* logits, const float* x, const float* u, int32_t N, int64_t P,
                                      float* mean, float* mode, float* sample, float* ll, float* dll_dlogits,
                                      void* stream) {
  LVAE_REQUIRE(logits && N > 0 && P > 0, LVAE_EINVAL, "lvae_bernoulli_fwd_f32: bad args");
  LVAE_REQUIRE(!sample || u, LVAE_EINVAL, "lvae_bernoulli_fwd_f32: sample needs uniforms");
  LVAE_REQUIRE(!x || ll, LVAE_EINVAL, "lvae_bernoulli_fwd_f32: x given but ll is null");
  hipLaunchKernelGGL(bernoulli_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, x, u, P, mean, mode, sample,
                     ll, dll_dlogits);
  LVAE_LAUNCH_CHECK("bernoulli_fwd");
  return 0;
}

extern "C" size_t lvae_dmol_workspace(int32_t N, int32_t HW) { return (size_t)N * HW * sizeof(float); }

// component counts the DMoL kernels are instantiated for
#define LVAE_DMOL_COUNTS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(8) X(10) X(12) X(16) X(20)
static bool dmol_supported(int nmix) {
  switch (nmix) {
#define X(n) case n:
    LVAE_DMOL_COUNTS(X)
#undef X
    return true;
    default: return false;
  }
}

extern "C" int lvae_dmol_ll_fwd_f32(const float* l, const float* x, int32_t N, int32_t HW, int32_t nmix, float* ll,
                                    float* dll_dl, void* workspace, size_t workspace_bytes, void* stream) {
  LVAE_REQUIRE(l && x && ll && workspace && N > 0 && HW > 0, LVAE_EINVAL, "lvae_dmol_ll_fwd_f32: bad args");
  LVAE_REQUIRE(dmol_supported(nmix), LVAE_EINVAL, "lvae_dmol_ll_fwd_f32: %d mixture components: instantiated for 1-6, 8, 10, 12, 16, 20", nmix);
  LVAE_REQUIRE(workspace_bytes >= lvae_dmol_workspace(N, HW), LVAE_EWORKSPACE, "lvae_dmol_ll_fwd_f32: workspace");
  const int64_t npix = (int64_t)N * HW;
  hipStream_t s = (hipStream_t)stream;
  float* ll_pix = static_cast<float*>(workspace);
  switch (nmix) {
#define X(n)                                                                                                                         \
  case n: {                                                                                                                          \
    constexpr int PIX = (128 * (10 * n + 1) * 4 <= 64 * 1024) ? 128 : 64;                                                            \
    hipLaunchKernelGGL((dmol_ll_kernel<n, PIX>), dim3((unsigned)((npix + PIX - 1) / PIX)), dim3(PIX), 0, s, l, x, npix, ll_pix, dll_dl); \
  } break;
    LVAE_DMOL_COUNTS(X)
#undef X
  }
  LVAE_LAUNCH_CHECK("dmol_ll");
  hipLaunchKernelGGL(rowsum_kernel, dim3(N), dim3(256), 0, s, ll_pix, (int64_t)HW, ll);
  LVAE_LAUNCH_CHECK("dmol_rowsum");
  return 0;
}

extern "C" int lvae_dmol_sample_f32(const float* l, const float* u_mix, const float* u_log, int32_t N, int32_t HW,
                                    int32_t nmix, float* sample, void* stream) {
  LVAE_REQUIRE(l && u_mix && u_log && sample && N > 0 && HW > 0, LVAE_EINVAL, "lvae_dmol_sample_f32: bad args");
  LVAE_REQUIRE(dmol_supported(nmix), LVAE_EINVAL, "lvae_dmol_sample_f32: %d mixture components: instantiated for 1-6, 8, 10, 12, 16, 20", nmix);
  const int64_t npix = (int64_t)N * HW;
  switch (nmix) {
#define X(n) \
  case n: hipLaunchKernelGGL(dmol_sample_kernel<n>, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, l, u_mix, u_log, npix, sample); break;
    LVAE_DMOL_COUNTS(X)
#undef X
  }
  LVAE_LAUNCH_CHECK("dmol_sample");
  return 0;
}

extern "C" int lvae_scale_per_sample_f32(const float* a, const float* g, int32_t N, int64_t P, float* out, void* stream) {
  LVAE_REQUIRE(a && g && out && N > 0 && P > 0, LVAE_EINVAL, "lvae_scale_per_sample_f32: bad args");
  const int64_t total = (int64_t)N * P;
  hipLaunchKernelGGL(scale_per_sample_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, g, P,
                     total, out);
  LVAE_LAUNCH_CHECK("scale_per_sample");
  return 0;
}

extern "C" int lvae_gaussian_fwd_f32(const float* params, const float* x, const float* eps, int32_t N, int64_t P, int32_t C,
                                     float* sample, float* ll, float* dll_dparams, void* stream) {
  LVAE_REQUIRE(params && N > 0 && P > 0 && C > 0, LVAE_EINVAL, "lvae_gaussian_fwd_f32: bad args");
  LVAE_REQUIRE(!sample || eps, LVAE_EINVAL, "lvae_gaussian_fwd_f32: sample needs eps");
  LVAE_REQUIRE(!x || ll, LVAE_EINVAL, "lvae_gaussian_fwd_f32: x given but ll is null");
  hipLaunchKernelGGL(gaussian_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, params, x, eps, P, C, sample, ll,
                     dll_dparams);
  LVAE_LAUNCH_CHECK("gaussian_fwd");
  return 0;
}

extern "C" int lvae_discr_logistic_fwd_f32(const float* raw, const float* x, const float* u, int32_t N, int64_t P, int32_t C,
                                           float* mean, float* logscale, float* sample, float* ll, float* dll_draw,
                                           void* stream) {
  LVAE_REQUIRE(raw && N > 0 && P > 0 && C > 0, LVAE_EINVAL, "lvae_discr_logistic_fwd_f32: bad args");
  LVAE_REQUIRE(!sample || u, LVAE_EINVAL, "lvae_discr_logistic_fwd_f32: sample needs uniforms");
  LVAE_REQUIRE(!x || ll, LVAE_EINVAL, "lvae_discr_logistic_fwd_f32: x given but ll is null");
  hipLaunchKernelGGL(discr_logistic_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, raw, x, u, P, C, mean, logscale,
                     sample, ll, dll_draw);
  LVAE_LAUNCH_CHECK("discr_logistic_fwd");
  return 0;
}
