// WMSE-MS-SSIM reconstruction loss, forward + gradient (the loss the reference's live `elbo` ships:
// prob_unet.py:229-267 -> wmse_ms_ssim_loss, prob_unet_utils.py:270-305).
//   L = lam * WMSE + (1 - lam) * (1 - MS-SSIM),   WMSE = mean( min(alpha e^{beta y}, 1) (x - y)^2 )
// MS-SSIM is the third-party `pytorch-msssim==1.0.0` (uv.lock pin; NOT present in the image -> parity unpinned), restated
// from its published algorithm: 5 scales, 7-tap Gaussian window (sigma 1.5) applied separably without padding, per-scale
//   cs = (2 s12 + C2) / (s1 + s2 + C2),  ssim = (2 m1 m2 + C1) / (m1^2 + m2^2 + C1) * cs,   C1 = (0.01 R)^2, C2 = (0.03 R)^2,
// spatial means per (image, channel), relu, 2x2 average pooling (padding = size % 2) between scales,
//   ms[n,c] = prod_l v_l ^ w_l,  w = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333),  v_l = cs mean (l < 4) / ssim mean (l = 4).
// Everything is fp32 and HBM/LDS-bound stencil work (no MFMA): one 32x32 tile per block, the five filtered moments
// (x, y, xx, yy, xy) are produced in LDS by a horizontal then a vertical 7-tap pass.  The backward recomputes the moments on a
// 6-pixel halo instead of storing ten maps per scale, forms the three adjoint maps (d/dm1, d/dE[xx], d/dE[xy]) and applies the
// transposed (full) filter, adds the 2x2-pool adjoint of the next coarser scale, and writes d/dx once per scale.
#include "pu_common.h"
#include "pu_kernels.h"
#include "../../include/probunet.h"

namespace pu {

namespace {
constexpr int T = 32, R = 6, TI = T + R;          // output tile, window - 1, forward input tile (38)
constexpr int TB = T + 2 * R;                      // backward input tile (44)
constexpr int NLV = 5;
struct Win { float g[7]; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// value range of the target: R = max(max - min, 1e-5)   (prob_unet_utils.py:288-289).  Two stages: per-block (min, max)
// partials, then one block folds them (deterministic, no float atomics).
constexpr int RANGE_BLOCKS = 256;
__global__ __launch_bounds__(256) void ms_range_partial_kernel(const float* __restrict__ y, long n, float* __restrict__ part) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = y[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}
__global__ __launch_bounds__(256) void ms_range_final_kernel(const float* __restrict__ part, int nblk, float fixed, const float* __restrict__ fixed_dev,
                                                             float* __restrict__ dr) {
  __shared__ float smn[4], smx[4];
  if (fixed_dev) { if (threadIdx.x == 0) dr[0] = fmaxf(fixed_dev[0], 1e-5f); return; }      // range decided by the caller on the device
  if (fixed > 0.f) { if (threadIdx.x == 0) dr[0] = fixed; return; }
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 256) { mn = fminf(mn, part[i * 2]); mx = fmaxf(mx, part[i * 2 + 1]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3])); mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    dr[0] = fmaxf(mx - mn, 1e-5f);
  }
}

// 2x2 average pooling, stride 2, zero padding (ph, pw) counted in the divisor (F.avg_pool2d defaults)
__global__ void ms_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, long nimg, int H, int W, int OH, int OW, int ph, int pw) {
  const long total = nimg * OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW); const long t = i / OW; const int oy = (int)(t % OH); const long im = t / OH;
    const float* s = src + im * H * W;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int yy = 2 * oy - ph + (d >> 1), xx = 2 * ox - pw + (d & 1);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += s[(long)yy * W + xx];
    }
    dst[i] = 0.25f * acc;
  }
}

// per-pixel SSIM terms from the five filtered moments
struct Terms { float cs, l, den, lden; };
__device__ __forceinline__ Terms ssim_terms(float a, float b, float p, float q, float r, float C1, float C2) {
  Terms t;
  const float aa = a * a, bb = b * b, ab = a * b;
  const float s1 = p - aa, s2 = q - bb, s12 = r - ab;
  t.den = s1 + s2 + C2;
  t.cs = (2.f * s12 + C2) / t.den;
  t.lden = aa + bb + C1;
  t.l = (2.f * ab + C1) / t.lden;
  return t;
}

// ---- forward of one scale: per-block partial sum of the cs map (ssim map at the last scale) over the valid outputs
template <bool LAST>
__global__ __launch_bounds__(256) void ms_level_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, int M, int C,
                                                           Win win, const float* __restrict__ dr, float* __restrict__ part) {
  __shared__ float sx[TI][TI + 1], sy[TI][TI + 1];
  __shared__ float h5[5][TI][T + 1];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int nc = blockIdx.z, n = nc / C, c = nc - n * C;
  const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T;
  const int OH = H - R, OW = W - R;
  const float* xp = X + (long)nc * H * W;
  const float* yp = Y + ((long)(n / M) * C + c) * H * W;
  for (int i = tid; i < TI * TI; i += 256) {
    const int r = i / TI, cc = i - r * TI, gy = oy0 + r, gx = ox0 + cc;
    const bool ok = gy < H && gx < W;
    sx[r][cc] = ok ? xp[(long)gy * W + gx] : 0.f;
    sy[r][cc] = ok ? yp[(long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TI * T; i += 256) {
    const int r = i / T, cc = i - r * T;
    float a = 0.f, b = 0.f, p = 0.f, q = 0.f, s = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float x = sx[r][cc + k], y = sy[r][cc + k], g = win.g[k];
      a += g * x; b += g * y; p += g * (x * x); q += g * (y * y); s += g * (x * y);
    }
    h5[0][r][cc] = a; h5[1][r][cc] = b; h5[2][r][cc] = p; h5[3][r][cc] = q; h5[4][r][cc] = s;
  }
  __syncthreads();
  const float Rv = dr[0], C1 = (0.01f * Rv) * (0.01f * Rv), C2 = (0.03f * Rv) * (0.03f * Rv);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < (T * T) / 256; ++j) {
    const int i = tid + j * 256, r = i / T, cc = i - r * T;
    if (oy0 + r < OH && ox0 + cc < OW) {
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float g = win.g[k];
#pragma unroll
        for (int qn = 0; qn < 5; ++qn) m[qn] += g * h5[qn][r + k][cc];
      }
      const Terms t = ssim_terms(m[0], m[1], m[2], m[3], m[4], C1, C2);
      sum += LAST ? t.l * t.cs : t.cs;
    }
  }
  const float tot = block_sum_256(sum, red);
  if (tid == 0) part[(long)nc * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

struct LevelDims { int H[NLV], W[NLV], ntiles[NLV]; long part_off[NLV]; };

// ---- combine: per (image, channel) ms value, the per-scale upstream coefficients, and the scalars.  One block.
//   scalars[PU_S_RECON] += (1 - lam) * (1 - mean_{n,c} ms);  scalars[PU_S_MSSSIM] = 1 - mean over the LAST member's (b, c)
__global__ __launch_bounds__(256) void ms_combine_kernel(const float* __restrict__ part, LevelDims d, int NC, int M, int C, float lam, float gscale,
                                                         float* __restrict__ gco, float* __restrict__ scalars) {
  __shared__ float red[4];
  const float w[NLV] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
  float s_all = 0.f, s_last = 0.f;
  for (int nc = threadIdx.x; nc < NC; nc += 256) {
    float v[NLV], ms = 1.f;
#pragma unroll
    for (int l = 0; l < NLV; ++l) {
      const float* p = part + d.part_off[l] + (long)nc * d.ntiles[l];
      float s = 0.f;
      for (int t = 0; t < d.ntiles[l]; ++t) s += p[t];
      const float npix = (float)(d.H[l] - R) * (float)(d.W[l] - R);
      v[l] = fmaxf(s / npix, 0.f);
      ms *= powf(v[l], w[l]);
    }
    if (gco) {
      const float up = -gscale * (1.f - lam) / (float)NC;              // d(beta0 * S * recon) / d ms[n,c]
#pragma unroll
      for (int l = 0; l < NLV; ++l) {
        const float npix = (float)(d.H[l] - R) * (float)(d.W[l] - R);
        gco[(long)nc * NLV + l] = v[l] > 0.f ? up * w[l] * ms / v[l] / npix : 0.f;
      }
    }
    s_all += ms;
    if ((nc / C) % M == M - 1) s_last += ms;
  }
  const float ta = block_sum_256(s_all, red);
  __syncthreads();
  const float tl = block_sum_256(s_last, red);
  if (threadIdx.x == 0) {
    scalars[PU_S_RECON] += (1.f - lam) * (1.f - ta / (float)NC);
    scalars[PU_S_MSSSIM] = 1.f - tl / ((float)NC / (float)M);
  }
}

// ---- weighted MSE (all members; the logged scalar is the last member's) + its gradient (WRITES dpred)
__global__ __launch_bounds__(256) void ms_wmse_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ dpred, int B, int M,
                                                      long CHW, float alpha, float beta, float lam, float gscale, float* __restrict__ scalars) {
  __shared__ float red[4];
  const long total = (long)B * M * CHW;
  float s_all = 0.f, s_last = 0.f;
  const float gk = gscale * lam * 2.f / (float)total;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / CHW, e = i - n * CHW;
    const int b = (int)(n / M), m = (int)(n - (long)b * M);
    const float y = Y[(long)b * CHW + e], df = X[i] - y;
    const float w = fminf(alpha * expf(beta * y), 1.f);
    const float v = w * df * df;
    s_all += v; if (m == M - 1) s_last += v;
    if (dpred) dpred[i] = gk * w * df;
  }
  const float ta = block_sum_256(s_all, red);
  __syncthreads();
  const float tl = block_sum_256(s_last, red);
  if (threadIdx.x == 0) {
    atomicAdd(scalars + PU_S_RECON, lam * ta / (float)total);
    atomicAdd(scalars + PU_S_WMSE, tl / (float)((long)B * CHW));
  }
}

// ---- backward of one scale.  dX (+)= F^T(A) + 2 x F^T(P) + y F^T(Rm) + pool-adjoint(dXn)
template <bool LAST>
__global__ __launch_bounds__(256) void ms_level_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, int M, int C, Win win,
                                                           const float* __restrict__ dr, const float* __restrict__ gco, int level,
                                                           const float* __restrict__ dXn, int H2, int W2, int ph, int pw,
                                                           float* __restrict__ dX, int accumulate) {
  extern __shared__ float smem[];
  float (*sx)[TB + 1] = reinterpret_cast<float (*)[TB + 1]>(smem);                               // [TB][TB+1]
  float (*sy)[TB + 1] = reinterpret_cast<float (*)[TB + 1]>(smem + TB * (TB + 1));
  float* h5 = smem + 2 * TB * (TB + 1);                                                          // [5][TB][TI]   (later [3][TI][T])
  float* m3 = h5 + 5 * TB * TI;                                                                  // [3][TI][TI+1]
  const int tid = threadIdx.x;
  const int nc = blockIdx.z, n = nc / C, c = nc - n * C;
  const int ix0 = blockIdx.x * T, iy0 = blockIdx.y * T;
  const int OH = H - R, OW = W - R;
  const float* xp = X + (long)nc * H * W;
  const float* yp = Y + ((long)(n / M) * C + c) * H * W;
  for (int i = tid; i < TB * TB; i += 256) {
    const int r = i / TB, cc = i - r * TB, gy = iy0 - R + r, gx = ix0 - R + cc;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[r][cc] = ok ? xp[(long)gy * W + gx] : 0.f;
    sy[r][cc] = ok ? yp[(long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TB * TI; i += 256) {                      // horizontal moments: rows TB, output columns TI
    const int r = i / TI, oc = i - r * TI;
    float a = 0.f, b = 0.f, p = 0.f, q = 0.f, s = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float x = sx[r][oc + k], y = sy[r][oc + k], g = win.g[k];
      a += g * x; b += g * y; p += g * (x * x); q += g * (y * y); s += g * (x * y);
    }
    h5[(0 * TB + r) * TI + oc] = a; h5[(1 * TB + r) * TI + oc] = b; h5[(2 * TB + r) * TI + oc] = p;
    h5[(3 * TB + r) * TI + oc] = q; h5[(4 * TB + r) * TI + oc] = s;
  }
  __syncthreads();
  const float Rv = dr[0], C1 = (0.01f * Rv) * (0.01f * Rv), C2 = (0.03f * Rv) * (0.03f * Rv);
  const float gl = gco[(long)nc * NLV + level];
  for (int i = tid; i < TI * TI; i += 256) {                      // adjoint maps on the TI x TI outputs around the tile
    const int orow = i / TI, oc = i - orow * TI;
    const int oy = iy0 - R + orow, ox = ix0 - R + oc;
    float A = 0.f, P = 0.f, Rm = 0.f;
    if (oy >= 0 && oy < OH && ox >= 0 && ox < OW && gl != 0.f) {
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float g = win.g[k];
#pragma unroll
        for (int qn = 0; qn < 5; ++qn) m[qn] += g * h5[(qn * TB + orow + k) * TI + oc];
      }
      const float a = m[0], b = m[1];
      const Terms t = ssim_terms(a, b, m[2], m[3], m[4], C1, C2);
      // cs = (2 s12 + C2) / den:  d/ds1 = -cs/den, d/ds12 = 2/den;  s1 = p - a^2, s12 = r - a b
      const float dcs_dp = -t.cs / t.den, dcs_dr = 2.f / t.den;
      const float dcs_da = (2.f * a * t.cs - 2.f * b) / t.den;
      if (LAST) {
        const float dl_da = (2.f * b - 2.f * a * t.l) / t.lden;
        A = gl * (t.cs * dl_da + t.l * dcs_da); P = gl * t.l * dcs_dp; Rm = gl * t.l * dcs_dr;
      } else { A = gl * dcs_da; P = gl * dcs_dp; Rm = gl * dcs_dr; }
    }
    m3[(0 * TI + orow) * (TI + 1) + oc] = A; m3[(1 * TI + orow) * (TI + 1) + oc] = P; m3[(2 * TI + orow) * (TI + 1) + oc] = Rm;
  }
  __syncthreads();
  float* h3 = h5;                                                 // [3][TI][T]: transposed filter along x
  for (int i = tid; i < TI * T; i += 256) {
    const int orow = i / T, j = i - orow * T;
    float a = 0.f, p = 0.f, r = 0.f;
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const float g = win.g[u]; const int oc = j - u + R;
      a += g * m3[(0 * TI + orow) * (TI + 1) + oc]; p += g * m3[(1 * TI + orow) * (TI + 1) + oc]; r += g * m3[(2 * TI + orow) * (TI + 1) + oc];
    }
    h3[(0 * TI + orow) * T + j] = a; h3[(1 * TI + orow) * T + j] = p; h3[(2 * TI + orow) * T + j] = r;
  }
  __syncthreads();
  float* dxp = dX + (long)nc * H * W;
#pragma unroll
  for (int jn = 0; jn < (T * T) / 256; ++jn) {
    const int i = tid + jn * 256, ii = i / T, j = i - ii * T;
    const int gy = iy0 + ii, gx = ix0 + j;
    if (gy < H && gx < W) {
      float a = 0.f, p = 0.f, r = 0.f;
#pragma unroll
      for (int v = 0; v < 7; ++v) {
        const float g = win.g[v]; const int orow = ii - v + R;
        a += g * h3[(0 * TI + orow) * T + j]; p += g * h3[(1 * TI + orow) * T + j]; r += g * h3[(2 * TI + orow) * T + j];
      }
      float o = a + 2.f * sx[ii + R][j + R] * p + sy[ii + R][j + R] * r;
      if (dXn) o += 0.25f * dXn[((long)nc * H2 + ((gy + ph) >> 1)) * W2 + ((gx + pw) >> 1)];
      const long at = (long)gy * W + gx;
      dxp[at] = accumulate ? dxp[at] + o : o;
    }
  }
}

constexpr size_t BWD_LDS = (size_t)(2 * TB * (TB + 1) + 5 * TB * TI + 3 * TI * (TI + 1)) * sizeof(float);

}  // namespace

size_t msssim_ws_floats(int B, int M, int C, int H, int W) {
  size_t n = 16 + 2 * RANGE_BLOCKS;               // dr + range partials
  int h = H, w = W;
  const size_t NC = (size_t)B * M * C, BC = (size_t)B * C;
  for (int l = 0; l < NLV; ++l) {
    n += NC * (size_t)cdiv(h - R, T) * cdiv(w - R, T);                       // partial sums
    if (l > 0) n += 2 * NC * (size_t)h * w + BC * (size_t)h * w;             // pooled x, its gradient, pooled y
    h = (h + 2 * (h % 2) - 2) / 2 + 1; w = (w + 2 * (w % 2) - 2) / 2 + 1;
  }
  n += NC * NLV;                                  // per-scale coefficients
  return n + 64;
}

hipError_t launch_wmse_msssim(const MsssimArgs& a, hipStream_t s) {
  if (a.H <= 96 || a.W <= 96) return hipErrorInvalidValue;     // ms_ssim asserts min(H, W) > (7 - 1) * 2^4
  if (a.ws_floats < msssim_ws_floats(a.B, a.M, a.C, a.H, a.W)) return hipErrorInvalidValue;
  Win win; float sum = 0.f;
  for (int k = 0; k < 7; ++k) { const float cdd = (float)(k - 3); win.g[k] = expf(-(cdd * cdd) / (2.f * 1.5f * 1.5f)); sum += win.g[k]; }
  for (int k = 0; k < 7; ++k) win.g[k] /= sum;
  const int NC = a.B * a.M * a.C, BC = a.B * a.C;
  // carve the workspace
  float* p = a.ws;
  float* dr = p; p += 16; float* rpart = p; p += 2 * RANGE_BLOCKS;
  LevelDims d; const float* Xl[NLV]; const float* Yl[NLV]; float* dXl[NLV]; float* partl[NLV];
  float* part0 = p; long poff = 0;
  int h = a.H, w = a.W;
  for (int l = 0; l < NLV; ++l) {
    d.H[l] = h; d.W[l] = w; d.ntiles[l] = cdiv(h - R, T) * cdiv(w - R, T); d.part_off[l] = poff;
    partl[l] = part0 + poff; poff += (long)NC * d.ntiles[l];
    h = (h + 2 * (h % 2) - 2) / 2 + 1; w = (w + 2 * (w % 2) - 2) / 2 + 1;
  }
  p += poff;
  Xl[0] = a.pred; Yl[0] = a.target; dXl[0] = a.dpred;
  for (int l = 1; l < NLV; ++l) {
    const size_t hw = (size_t)d.H[l] * d.W[l];
    Xl[l] = p; p += (size_t)NC * hw; dXl[l] = p; p += (size_t)NC * hw; Yl[l] = p; p += (size_t)BC * hw;
  }
  float* gco = p; p += (size_t)NC * NLV;

  if (a.data_range <= 0.f && !a.data_range_dev)
    hipLaunchKernelGGL(ms_range_partial_kernel, dim3(RANGE_BLOCKS), dim3(256), 0, s, a.target, (long)BC * a.H * a.W, rpart);
  hipLaunchKernelGGL(ms_range_final_kernel, dim3(1), dim3(256), 0, s, rpart, RANGE_BLOCKS, a.data_range, a.data_range_dev, dr);
  {
    const long total = (long)NC * a.H * a.W;
    const unsigned g = (unsigned)(total / 256 / 8 > 2048 ? 2048 : (total / 256 / 8 < 1 ? 1 : total / 256 / 8));
    hipLaunchKernelGGL(ms_wmse_kernel, dim3(g), dim3(256), 0, s, a.pred, a.target, a.dpred, a.B, a.M, (long)a.C * a.H * a.W, a.alpha_w, a.beta_w,
                       a.lam_w, a.gscale, a.scalars);
  }
  for (int l = 0; l < NLV; ++l) {
    if (l > 0) {
      const int ph = d.H[l - 1] % 2, pw = d.W[l - 1] % 2;
      const long tx = (long)NC * d.H[l] * d.W[l], ty = (long)BC * d.H[l] * d.W[l];
      hipLaunchKernelGGL(ms_pool_kernel, dim3((unsigned)cdiv(tx, 256)), dim3(256), 0, s, Xl[l - 1], const_cast<float*>(Xl[l]), (long)NC, d.H[l - 1], d.W[l - 1], d.H[l], d.W[l], ph, pw);
      hipLaunchKernelGGL(ms_pool_kernel, dim3((unsigned)cdiv(ty, 256)), dim3(256), 0, s, Yl[l - 1], const_cast<float*>(Yl[l]), (long)BC, d.H[l - 1], d.W[l - 1], d.H[l], d.W[l], ph, pw);
    }
    const dim3 grid(cdiv(d.W[l] - R, T), cdiv(d.H[l] - R, T), NC);
    if (l == NLV - 1) hipLaunchKernelGGL((ms_level_fwd_kernel<true>), grid, dim3(256), 0, s, Xl[l], Yl[l], d.H[l], d.W[l], a.M, a.C, win, dr, partl[l]);
    else hipLaunchKernelGGL((ms_level_fwd_kernel<false>), grid, dim3(256), 0, s, Xl[l], Yl[l], d.H[l], d.W[l], a.M, a.C, win, dr, partl[l]);
  }
  hipLaunchKernelGGL(ms_combine_kernel, dim3(1), dim3(256), 0, s, part0, d, NC, a.M, a.C, a.lam_w, a.gscale, a.dpred ? gco : nullptr, a.scalars);
  if (a.dpred) {
    static AttrOnce attr_once;
    if (!attr_once.cur()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ms_level_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(ms_level_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS);
      if (e != hipSuccess) return e;
      attr_once.cur() = true;
    }
    for (int l = NLV - 1; l >= 0; --l) {
      const dim3 grid(cdiv(d.W[l], T), cdiv(d.H[l], T), NC);
      const float* dXn = l + 1 < NLV ? dXl[l + 1] : nullptr;
      const int H2 = l + 1 < NLV ? d.H[l + 1] : 0, W2 = l + 1 < NLV ? d.W[l + 1] : 0, ph = d.H[l] % 2, pw = d.W[l] % 2;
      const int acc = l == 0 ? 1 : 0;                         // scale 0 adds onto the WMSE gradient already in dpred
      if (l == NLV - 1) hipLaunchKernelGGL((ms_level_bwd_kernel<true>), grid, dim3(256), BWD_LDS, s, Xl[l], Yl[l], d.H[l], d.W[l], a.M, a.C, win, dr, gco, l, dXn, H2, W2, ph, pw, dXl[l], acc);
      else hipLaunchKernelGGL((ms_level_bwd_kernel<false>), grid, dim3(256), BWD_LDS, s, Xl[l], Yl[l], d.H[l], d.W[l], a.M, a.C, win, dr, gco, l, dXn, H2, W2, ph, pw, dXl[l], acc);
    }
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ ClimEx-shaped data transforms on the device (row f4)
// lrinterp_to_residuals item transform (climex_utils.py:197-225):  lr = AvgPool2d(k)(hr);  lrinterp = nearest-upsample(lr, k);
//   inputs = (lrinterp - mean) / (std + eps);  targets = (hr - mean) / (std + eps) - inputs
__global__ void lrinterp_residuals_kernel(const float* __restrict__ hr, int B, int C, int H, int W, int k, const float* __restrict__ mean,
                                          const float* __restrict__ stdv, float eps, float* __restrict__ inputs, float* __restrict__ targets,
                                          float* __restrict__ lrinterp, float* __restrict__ lr) {
  const int h = H / k, w = W / k;
  const long cells = (long)B * C * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    const int cx = (int)(i % w); long t = i / w; const int cy = (int)(t % h); t /= h;            // t = b * C + c
    const int c = (int)(t % C);
    const float* src = hr + (t * H + (long)cy * k) * W + (long)cx * k;
    float acc = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) acc += src[(long)dy * W + dx];
    const float m = acc / (float)(k * k);
    if (lr) lr[i] = m;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const long at = (t * H + (long)cy * k + dy) * W + (long)cx * k + dx;
        const long st = ((long)c * H + (long)cy * k + dy) * W + (long)cx * k + dx;
        const float sd = stdv[st] + eps, mu = mean[st];
        const float li = (m - mu) / sd, hs = (hr[at] - mu) / sd;
        if (lrinterp) lrinterp[at] = m;
        inputs[at] = li; targets[at] = hs - li;
      }
  }
}
hipError_t launch_lrinterp_residuals(const float* hr, int B, int C, int H, int W, int k, const float* mean, const float* stdv, float eps,
                                     float* inputs, float* targets, float* lrinterp, float* lr, hipStream_t s) {
  const long cells = (long)B * C * (H / k) * (W / k);
  hipLaunchKernelGGL(lrinterp_residuals_kernel, dim3((unsigned)(cdiv(cells, 256) > 4096 ? 4096 : cdiv(cells, 256))), dim3(256), 0, s, hr, B, C, H, W, k,
                     mean, stdv, eps, inputs, targets, lrinterp, lr);
  return hipGetLastError();
}
// compute_stats (climex_utils.py:255-264): per low-resolution cell, mean and UNBIASED std over the N time steps of the
// k x k block means; expanded to the high-resolution grid by repeat_interleave.  One thread per (c, cell); Welford in fp32
// over block means computed in fp32 (torch's AvgPool2d + mean/std are fp32 as well).
__global__ void lr_stats_kernel(const float* __restrict__ hr, int N, int C, int H, int W, int k, float* __restrict__ mean_lr, float* __restrict__ std_lr,
                                float* __restrict__ mean_hr, float* __restrict__ std_hr) {
  const int h = H / k, w = W / k;
  const long cells = (long)C * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    const int cx = (int)(i % w); long t = i / w; const int cy = (int)(t % h); const int c = (int)(t / h);
    double mu = 0.0, m2 = 0.0;
    for (int n = 0; n < N; ++n) {
      const float* src = hr + (((long)n * C + c) * H + (long)cy * k) * W + (long)cx * k;
      float acc = 0.f;
      for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) acc += src[(long)dy * W + dx];
      const double v = (double)(acc / (float)(k * k));
      const double dl = v - mu; mu += dl / (double)(n + 1); m2 += dl * (v - mu);
    }
    const float mf = (float)mu, sf = N > 1 ? (float)sqrt(m2 / (double)(N - 1)) : NAN;
    if (mean_lr) mean_lr[i] = mf;
    if (std_lr) std_lr[i] = sf;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const long st = ((long)c * H + (long)cy * k + dy) * W + (long)cx * k + dx;
        if (mean_hr) mean_hr[st] = mf;
        if (std_hr) std_hr[st] = sf;
      }
  }
}
hipError_t launch_lr_stats(const float* hr, int N, int C, int H, int W, int k, float* mean_lr, float* std_lr, float* mean_hr, float* std_hr, hipStream_t s) {
  const long cells = (long)C * (H / k) * (W / k);
  hipLaunchKernelGGL(lr_stats_kernel, dim3((unsigned)cdiv(cells, 64)), dim3(64), 0, s, hr, N, C, H, W, k, mean_lr, std_lr, mean_hr, std_hr);
  return hipGetLastError();
}

// softplus / softplus_inv of climex_utils.py:36-46, in place like the reference:
//   softplus    : v > thr ? v : log(exp(v) + 1) - c          (inverse of the load-time pre-transform)
//   softplus_inv: v > thr ? v : log(exp(v + c) - 1)          (applied to precipitation and tasmax - tasmin when transfo=True, :141-143)
__global__ void softplus_transform_kernel(float* __restrict__ d, long n, int inverse, float thr, float c) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = d[i];
    if (!(v > thr)) d[i] = inverse ? logf(expf(v + c) - 1.f) : logf(expf(v) + 1.f) - c;
  }
}
hipError_t launch_softplus_transform(float* d, long n, int inverse, float thr, float c, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(softplus_transform_kernel, dim3((unsigned)(cdiv(n, 256) > 8192 ? 8192 : cdiv(n, 256))), dim3(256), 0, s, d, n, inverse, thr, c);
  return hipGetLastError();
}
// invstand_residual / residual_to_hr (climex_utils.py:270-285) on tensors that already exist:
//   out[b,s,c,p] = (base ? base[b,c,p] : 0) + x[b,s,c,p] * (std[c,p] + eps) + (mean ? mean[c,p] : 0)
__global__ void destandardize_kernel(const float* __restrict__ x, const float* __restrict__ base, const float* __restrict__ stdv,
                                     const float* __restrict__ mean, float eps, int B, int n, long CHW, float* __restrict__ out) {
  const long total = (long)B * n * CHW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i % CHW; const long b = i / (CHW * n);
    float v = x[i] * (stdv[q] + eps);
    if (mean) v += mean[q];
    if (base) v = base[b * CHW + q] + v;
    out[i] = v;
  }
}
hipError_t launch_destandardize(const float* x, const float* base, const float* stdv, const float* mean, float eps, int B, int n, long CHW,
                                float* out, hipStream_t s) {
  const long total = (long)B * n * CHW;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(destandardize_kernel, dim3((unsigned)(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256))), dim3(256), 0, s, x, base, stdv, mean,
                     eps, B, n, CHW, out);
  return hipGetLastError();
}

}  // namespace pu
