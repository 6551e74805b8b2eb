// NormalStochasticBlock2d elementwise core (lib/stochastic.py:45-99, 209-226): reparameterised sample, log p(z),
// log q(z), Monte-Carlo and analytical KL with their per-sample and per-pixel reductions, and the hand-written
// backward. One workgroup per sample; the Z channels of a pixel are contiguous (NHWC) so the per-pixel KL sum is
// a shuffle reduction over Z lanes. HBM-bound: reads p, q, eps once and writes z once.
#include "lvae_common.h"

namespace lvae {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

struct StochArgs {
  const float* p;
  const float* q;
  const float* eps;
  int p_bcast, N, HW, Z, mode, analytical;
};

__device__ __forceinline__ float normal_logprob(float z, float mu, float lv) {
  // torch.distributions.Normal(mu, exp(lv/2)).log_prob(z): var = std^2, log std = log(std)
  const float sd = expf(0.5f * lv);
  const float d = z - mu;
  return -(d * d) / (2.f * sd * sd) - logf(sd) - kLogSqrt2Pi;
}

__device__ __forceinline__ float normal_kl(float qmu, float qlv, float pmu, float plv) {
  const float qs = expf(0.5f * qlv), ps = expf(0.5f * plv);
  const float r = qs / ps, vr = r * r;
  const float t = (qmu - pmu) / ps;
  return 0.5f * (vr + t * t - 1.f - logf(vr));
}

__global__ __launch_bounds__(256) void stoch_fwd_kernel(StochArgs a, float* __restrict__ z_out, float* logprob_p,
                                                         float* logprob_q, float* kl_samplewise, float* kl_spatial) {
  __shared__ float red[4];
  const int n = blockIdx.x, t = threadIdx.x;
  const int Z = a.Z, per = a.HW * Z;
  const float* pn = a.p + (a.p_bcast ? 0 : (size_t)n * a.HW * 2 * Z);
  const float* qn = a.q ? a.q + (size_t)n * a.HW * 2 * Z : nullptr;
  const bool zpow2 = (Z & (Z - 1)) == 0 && Z <= 64;
  float s_lp = 0.f, s_lq = 0.f, s_kl = 0.f;
  // iterate in whole 256-element passes so that every lane of a wave takes part in the shuffles
  const int passes = (per + 255) / 256;
  for (int it = 0; it < passes; ++it) {
    const int e = it * 256 + t;
    const bool ok = e < per;
    float kan = 0.f;
    int pix = 0;
    if (ok) {
      pix = e / Z;
      const int c = e - pix * Z;
      const float pmu = pn[(size_t)pix * 2 * Z + c], plv = pn[(size_t)pix * 2 * Z + Z + c];
      float smu = pmu, slv = plv, qmu = 0.f, qlv = 0.f;
      if (qn) {
        qmu = qn[(size_t)pix * 2 * Z + c];
        qlv = qn[(size_t)pix * 2 * Z + Z + c];
        smu = qmu;
        slv = qlv;
      }
      float z;
      const size_t zi = (size_t)n * per + e;
      if (a.mode == 0) z = smu + expf(0.5f * slv) * a.eps[zi];
      else if (a.mode == 1) z = smu;
      else z = a.eps[zi];  // forced latent is passed through the eps pointer
      z_out[zi] = z;
      const float lp = normal_logprob(z, pmu, plv);
      s_lp += lp;
      if (qn) {
        const float lq = normal_logprob(z, qmu, qlv);
        s_lq += lq;
        kan = normal_kl(qmu, qlv, pmu, plv);
        s_kl += a.analytical ? kan : (lq - lp);
      }
    }
    if (qn && kl_spatial) {
      if (zpow2) {
        float v = kan;
        for (int o = Z >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (ok && (e % Z) == 0) kl_spatial[(size_t)n * a.HW + pix] = v;
      }
    }
  }
  if (qn && kl_spatial && !zpow2) {
    for (int pix = t; pix < a.HW; pix += 256) {
      float v = 0.f;
      for (int c = 0; c < Z; ++c)
        v += normal_kl(qn[(size_t)pix * 2 * Z + c], qn[(size_t)pix * 2 * Z + Z + c], pn[(size_t)pix * 2 * Z + c],
                       pn[(size_t)pix * 2 * Z + Z + c]);
      kl_spatial[(size_t)n * a.HW + pix] = v;
    }
  }
  s_lp = block_sum_256(s_lp, red);
  if (t == 0) logprob_p[n] = s_lp;
  if (qn) {
    s_lq = block_sum_256(s_lq, red);
    s_kl = block_sum_256(s_kl, red);
    if (t == 0) {
      logprob_q[n] = s_lq;
      kl_samplewise[n] = s_kl;
    }
  }
}

// The same block for Z a multiple of 4 with Z / 4 a power of two <= 16 (every model of BASELINE.json: Z = 32) and 16-byte aligned tensors:
// a thread takes 4 consecutive channels of a pixel (16-byte loads / stores), two 1,024-element passes are in flight per iteration, the
// per-pixel KL is a shuffle reduction over the Z / 4 lanes of a pixel. The scalar form above issues one dependent round trip per 256
// elements from ONE workgroup per sample: 32 round trips at 16x16 x 32 (measured 19 us per launch averaged over the 15 levels of the
// CIFAR model, ~45 us at the 16x16 levels, for 50 MB = 10 us of HBM time). Per-element arithmetic is the scalar kernel's, unchanged.
__global__ __launch_bounds__(256) void stoch_fwd_v4_kernel(StochArgs a, float* __restrict__ z_out, float* logprob_p,
                                                            float* logprob_q, float* kl_samplewise, float* kl_spatial) {
  __shared__ float red[4];
  const int n = blockIdx.x, t = threadIdx.x;
  const int Z = a.Z, Z4 = Z >> 2, per4 = a.HW * Z4;   // 4-channel groups of this sample
  const float* pn = a.p + (a.p_bcast ? 0 : (size_t)n * a.HW * 2 * Z);
  const float* qn = a.q ? a.q + (size_t)n * a.HW * 2 * Z : nullptr;
  const float* en = a.eps ? a.eps + (size_t)n * a.HW * Z : nullptr;
  float* zn = z_out + (size_t)n * a.HW * Z;
  float s_lp = 0.f, s_lq = 0.f, s_kl = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 2;
  for (int it = 0; it * 256 * U < per4; ++it) {
    f32x4 pmu[U], plv[U], qmu[U], qlv[U], ev[U];
    int pix[U], c[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = (it * U + u) * 256 + t;
      ok[u] = e < per4;
      const int ee = ok[u] ? e : 0;
      pix[u] = ee / Z4;
      c[u] = (ee - pix[u] * Z4) * 4;
      const size_t b = (size_t)pix[u] * 2 * Z + c[u];
      pmu[u] = *reinterpret_cast<const f32x4*>(pn + b);
      plv[u] = *reinterpret_cast<const f32x4*>(pn + b + Z);
      qmu[u] = qn ? *reinterpret_cast<const f32x4*>(qn + b) : zero4;
      qlv[u] = qn ? *reinterpret_cast<const f32x4*>(qn + b + Z) : zero4;
      ev[u] = a.mode != 1 ? *reinterpret_cast<const f32x4*>(en + (size_t)pix[u] * Z + c[u]) : zero4;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 zv;
      float kan = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float smu = qn ? qmu[u][j] : pmu[u][j], slv = qn ? qlv[u][j] : plv[u][j];
        float z;
        if (a.mode == 0) z = smu + expf(0.5f * slv) * ev[u][j];
        else if (a.mode == 1) z = smu;
        else z = ev[u][j];  // forced latent is passed through the eps pointer
        zv[j] = z;
        if (ok[u]) {
          const float lp = normal_logprob(z, pmu[u][j], plv[u][j]);
          s_lp += lp;
          if (qn) {
            const float lq = normal_logprob(z, qmu[u][j], qlv[u][j]);
            s_lq += lq;
            const float k = normal_kl(qmu[u][j], qlv[u][j], pmu[u][j], plv[u][j]);
            kan += k;
            s_kl += a.analytical ? k : (lq - lp);
          }
        }
      }
      if (ok[u]) *reinterpret_cast<f32x4*>(zn + (size_t)pix[u] * Z + c[u]) = zv;
      if (qn && kl_spatial) {   // all lanes take part in the shuffles; the Z / 4 lanes of a pixel are consecutive
        float v = kan;
        for (int o = Z4 >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (ok[u] && c[u] == 0) kl_spatial[(size_t)n * a.HW + pix[u]] = v;
      }
    }
  }
  s_lp = block_sum_256(s_lp, red);
  if (t == 0) logprob_p[n] = s_lp;
  if (qn) {
    s_lq = block_sum_256(s_lq, red);
    s_kl = block_sum_256(s_kl, red);
    if (t == 0) {
      logprob_q[n] = s_lq;
      kl_samplewise[n] = s_kl;
    }
  }
}

struct StochBwdArgs {
  const float *p, *q, *eps, *z, *dz, *g_lp, *g_lq, *g_kl, *g_ks;
  int p_bcast, N, HW, Z, mode, analytical;
};

__global__ __launch_bounds__(256) void stoch_bwd_kernel(StochBwdArgs a, float* __restrict__ dp, float* __restrict__ dq) {
  const int Z = a.Z;
  const int64_t per = (int64_t)a.HW * Z, total = per * a.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / per);
    const int e = (int)(i - (int64_t)n * per);
    const int pix = e / Z, c = e - pix * Z;
    const size_t pb = ((a.p_bcast ? 0 : (size_t)n * a.HW) + pix) * 2 * Z, ob = ((size_t)n * a.HW + pix) * 2 * Z;
    const float pmu = a.p[pb + c], plv = a.p[pb + Z + c];
    const float z = a.z[i];
    const float dzu = a.dz ? a.dz[i] : 0.f;
    const float glp = a.g_lp ? a.g_lp[n] : 0.f;
    const float ivp = expf(-plv);  // 1/sigma_p^2
    const float dpz = z - pmu;
    if (a.q) {
      const float qmu = a.q[ob + c], qlv = a.q[ob + Z + c];
      const float gkl = a.g_kl ? a.g_kl[n] : 0.f;
      const float G_lp = glp - (a.analytical ? 0.f : gkl);
      const float G_lq = (a.g_lq ? a.g_lq[n] : 0.f) + (a.analytical ? 0.f : gkl);
      const float G_an = (a.g_ks ? a.g_ks[(size_t)n * a.HW + pix] : 0.f) + (a.analytical ? gkl : 0.f);
      const float ivq = expf(-qlv);
      const float dqz = z - qmu;
      const float Gz = dzu - G_lp * dpz * ivp - G_lq * dqz * ivq;
      const float dmu = qmu - pmu;
      const float vr = expf(qlv - plv);
      float dpmu = G_lp * dpz * ivp - G_an * dmu * ivp;
      float dplv = G_lp * 0.5f * (dpz * dpz * ivp - 1.f) + G_an * 0.5f * (1.f - vr - dmu * dmu * ivp);
      float dqmu = G_lq * dqz * ivq + G_an * dmu * ivp;
      float dqlv = G_lq * 0.5f * (dqz * dqz * ivq - 1.f) + G_an * 0.5f * (vr - 1.f);
      if (a.mode <= 1) dqmu += Gz;
      if (a.mode == 0) dqlv += Gz * 0.5f * expf(0.5f * qlv) * a.eps[i];
      dp[ob + c] = dpmu;
      dp[ob + Z + c] = dplv;
      dq[ob + c] = dqmu;
      dq[ob + Z + c] = dqlv;
    } else {
      const float Gz = dzu - glp * dpz * ivp;
      float dpmu = glp * dpz * ivp, dplv = glp * 0.5f * (dpz * dpz * ivp - 1.f);
      if (a.mode <= 1) dpmu += Gz;
      if (a.mode == 0) dplv += Gz * 0.5f * expf(0.5f * plv) * a.eps[i];
      dp[ob + c] = dpmu;
      dp[ob + Z + c] = dplv;
    }
  }
}


// Elementwise KL (lib/stochastic.py:88-91 `kl_elementwise`, :209-226 kl_normal_mc): out[n,pix,c] = log q(z) - log p(z) (Monte
// Carlo) or KL(q || p) (analytical). Not on the training path (TopDownLayer consumes the per-sample sums of stoch_fwd_kernel);
// kept as its own pass so the training step does not write a tensor nobody reads.
struct KlElemArgs {
  const float *p, *q, *z, *g;
  int p_bcast, q_bcast, HW, Z, analytical;
  int64_t total;
};

__global__ __launch_bounds__(256) void kl_elem_fwd_kernel(KlElemArgs a, float* __restrict__ out) {
  const int Z = a.Z;
  const int64_t per = (int64_t)a.HW * Z;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per;
    const int e = (int)(i - n * per);
    const int pix = e / Z, c = e - pix * Z;
    const size_t pb = ((a.p_bcast ? 0 : (size_t)n * a.HW) + pix) * 2 * Z, qb = ((a.q_bcast ? 0 : (size_t)n * a.HW) + pix) * 2 * Z;
    const float pmu = a.p[pb + c], plv = a.p[pb + Z + c], qmu = a.q[qb + c], qlv = a.q[qb + Z + c];
    out[i] = a.analytical ? normal_kl(qmu, qlv, pmu, plv) : normal_logprob(a.z[i], qmu, qlv) - normal_logprob(a.z[i], pmu, plv);
  }
}

// g = d/d(out); writes dp, dq (full batch shape; the caller reduces a broadcast operand) and dz (Monte Carlo only, else zeros)
__global__ __launch_bounds__(256) void kl_elem_bwd_kernel(KlElemArgs a, float* __restrict__ dp, float* __restrict__ dq,
                                                           float* __restrict__ dz) {
  const int Z = a.Z;
  const int64_t per = (int64_t)a.HW * Z;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per;
    const int e = (int)(i - n * per);
    const int pix = e / Z, c = e - pix * Z;
    const size_t pb = ((a.p_bcast ? 0 : (size_t)n * a.HW) + pix) * 2 * Z, qb = ((a.q_bcast ? 0 : (size_t)n * a.HW) + pix) * 2 * Z;
    const size_t ob = ((size_t)n * a.HW + pix) * 2 * Z;
    const float pmu = a.p[pb + c], plv = a.p[pb + Z + c], qmu = a.q[qb + c], qlv = a.q[qb + Z + c];
    const float g = a.g[i];
    const float ivp = expf(-plv);
    float dpmu, dplv, dqmu, dqlv, dzz = 0.f;
    if (a.analytical) {
      const float dmu = qmu - pmu, vr = expf(qlv - plv);
      dpmu = -g * dmu * ivp;
      dplv = g * 0.5f * (1.f - vr - dmu * dmu * ivp);
      dqmu = g * dmu * ivp;
      dqlv = g * 0.5f * (vr - 1.f);
    } else {
      const float z = a.z[i], ivq = expf(-qlv), dpz = z - pmu, dqz = z - qmu;
      dpmu = -g * dpz * ivp;
      dplv = -g * 0.5f * (dpz * dpz * ivp - 1.f);
      dqmu = g * dqz * ivq;
      dqlv = g * 0.5f * (dqz * dqz * ivq - 1.f);
      dzz = g * (dpz * ivp - dqz * ivq);
    }
    dp[ob + c] = dpmu;
    dp[ob + Z + c] = dplv;
    dq[ob + c] = dqmu;
    dq[ob + Z + c] = dqlv;
    if (dz) dz[i] = dzz;
  }
}

}  // namespace lvae

using namespace lvae;

extern "C" int lvae_normal_stochastic_fwd_f32(const float* p, int32_t p_bcast, const float* q, const float* eps,
                                              int32_t N, int32_t HW, int32_t Z, int32_t mode, int32_t analytical_kl,
                                              float* z, float* logprob_p, float* logprob_q, float* kl_samplewise,
                                              float* kl_spatial, void* stream) {
  LVAE_REQUIRE(p && z && logprob_p && N > 0 && HW > 0 && Z > 0, LVAE_EINVAL, "lvae_normal_stochastic_fwd_f32: bad args");
  LVAE_REQUIRE(mode >= 0 && mode <= 2, LVAE_EINVAL, "lvae_normal_stochastic_fwd_f32: mode %d", mode);
  LVAE_REQUIRE(mode == 1 || eps, LVAE_EINVAL, "lvae_normal_stochastic_fwd_f32: eps / forced latent missing");
  LVAE_REQUIRE(!q || (logprob_q && kl_samplewise), LVAE_EINVAL, "lvae_normal_stochastic_fwd_f32: q outputs missing");
  StochArgs a{p, q, eps, p_bcast, N, HW, Z, mode, analytical_kl};
  const auto al = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
  const int z4 = Z >> 2;
  if (Z % 4 == 0 && z4 <= 16 && (z4 & (z4 - 1)) == 0 && al(p) && al(q) && al(eps) && al(z))
    hipLaunchKernelGGL(stoch_fwd_v4_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a, z, logprob_p, logprob_q, kl_samplewise, kl_spatial);
  else
    hipLaunchKernelGGL(stoch_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a, z, logprob_p, logprob_q,
                       kl_samplewise, kl_spatial);
  LVAE_LAUNCH_CHECK("normal_stochastic_fwd");
  return 0;
}

extern "C" int lvae_normal_stochastic_bwd_f32(const float* p, int32_t p_bcast, const float* q, const float* eps,
                                              const float* z, const float* dz, const float* g_lp, const float* g_lq,
                                              const float* g_kl, const float* g_ks, int32_t N, int32_t HW, int32_t Z,
                                              int32_t mode, int32_t analytical_kl, float* dp, float* dq, void* stream) {
  LVAE_REQUIRE(p && z && dp && N > 0 && HW > 0 && Z > 0, LVAE_EINVAL, "lvae_normal_stochastic_bwd_f32: bad args");
  LVAE_REQUIRE((q == nullptr) == (dq == nullptr), LVAE_EINVAL, "lvae_normal_stochastic_bwd_f32: q/dq mismatch");
  LVAE_REQUIRE(mode != 0 || eps, LVAE_EINVAL, "lvae_normal_stochastic_bwd_f32: eps missing");
  StochBwdArgs a{p, q, eps, z, dz, g_lp, g_lq, g_kl, g_ks, p_bcast, N, HW, Z, mode, analytical_kl};
  const int64_t total = (int64_t)N * HW * Z;
  hipLaunchKernelGGL(stoch_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, dp, dq);
  LVAE_LAUNCH_CHECK("normal_stochastic_bwd");
  return 0;
}

extern "C" int lvae_kl_elementwise_fwd_f32(const float* p, int32_t p_bcast, const float* q, int32_t q_bcast, const float* z,
                                           int32_t N, int32_t HW, int32_t Z, int32_t analytical_kl, float* out, void* stream) {
  LVAE_REQUIRE(p && q && out && N > 0 && HW > 0 && Z > 0, LVAE_EINVAL, "lvae_kl_elementwise_fwd_f32: bad args");
  LVAE_REQUIRE(analytical_kl || z, LVAE_EINVAL, "lvae_kl_elementwise_fwd_f32: z missing");
  KlElemArgs a{p, q, z, nullptr, p_bcast, q_bcast, HW, Z, analytical_kl, (int64_t)N * HW * Z};
  hipLaunchKernelGGL(kl_elem_fwd_kernel, dim3(grid_for(a.total, 256)), dim3(256), 0, (hipStream_t)stream, a, out);
  LVAE_LAUNCH_CHECK("kl_elementwise_fwd");
  return 0;
}

extern "C" int lvae_kl_elementwise_bwd_f32(const float* p, int32_t p_bcast, const float* q, int32_t q_bcast, const float* z,
                                           const float* g, int32_t N, int32_t HW, int32_t Z, int32_t analytical_kl, float* dp,
                                           float* dq, float* dz, void* stream) {
  LVAE_REQUIRE(p && q && g && dp && dq && N > 0 && HW > 0 && Z > 0, LVAE_EINVAL, "lvae_kl_elementwise_bwd_f32: bad args");
  LVAE_REQUIRE(analytical_kl || z, LVAE_EINVAL, "lvae_kl_elementwise_bwd_f32: z missing");
  KlElemArgs a{p, q, z, g, p_bcast, q_bcast, HW, Z, analytical_kl, (int64_t)N * HW * Z};
  hipLaunchKernelGGL(kl_elem_bwd_kernel, dim3(grid_for(a.total, 256)), dim3(256), 0, (hipStream_t)stream, a, dp, dq, dz);
  LVAE_LAUNCH_CHECK("kl_elementwise_bwd");
  return 0;
}
