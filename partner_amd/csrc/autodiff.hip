// Differentiable primitives for the TRAINING forms of the attention blocks: the
// inference kernels of attention.hip / swin_attn.hip fuse score, softmax and P.V per window and keep nothing; training needs the
// intermediate maps, so there the blocks are composed from a handful of generic kernels, each with its backward:
//   pn_contract_f32          strided tensor contraction  C[g, m, n] (+)= alpha * sum_k A[g, m, k] * B[g, n, k]   (3 batch levels, 2-level
//                            m / n / k indices): every q.k^T, P.V and their four gradients, with the head / window / "raw view"
//                            permutations of set_transformer.py:7-19, 331-334, 417-425 expressed as strides (no permuted copies)
//   pn_softmax_f32 / _bwd    softmax over a strided middle axis
//   pn_layernorm_bwd_f32     backward of pn_layernorm_f32 (dx, dgamma, dbeta; fixed-order partial sums)
//   pn_gelu_f32 / _bwd       exact-erf GELU
//   pn_pair_diff_f32         rel[g, m, n, :] = a[g, m, :] - b[g, n, :] (Cartesian offsets fed to the relative-position MLPs)
//   pn_dropout_f32 / pn_mul_f32  Dropout / DropPath masks (counter-based draws) and the elementwise product of their backward
//   pn_scatter_rows_f32      backward of the key-point row gather (set_transformer.py:144-147)
//   pn_roll_w_f32            torch.roll along the azimuth axis of a (B, H, W, C) token map (odd SetBlocks)
//   pn_l2_normalize_f32/_bwd x / max(||x||, eps) over the last axis (cosine attention of the Swin stage)
// Reference arithmetic: det3d/models/utils/set_transformer.py:118-166, 216-259, 307-354, 392-440 under torch autograd.
// These are correctness-first kernels (one thread per output element): the training iteration of the Waymo config is dominated by
// the convolutions and linear layers, which run on the MFMA kernels.
#include "pn_common.h"
#include <algorithm>

namespace {

struct Contract {
  const float* A; const float* B; float* C;
  long long sA[7], sB[7], sC[7];   // A: g0 g1 g2 m0 m1 k0 k1;  B: g0 g1 g2 n0 n1 k0 k1;  C: g0 g1 g2 m0 m1 n0 n1
  int d[9];                        // G0 G1 G2 M0 M1 N0 N1 K0 K1
  float alpha; int accumulate;
};

__global__ __launch_bounds__(256) void contract_kernel(Contract c) {
  const long long G = (long long)c.d[0] * c.d[1] * c.d[2], M = (long long)c.d[3] * c.d[4], N = (long long)c.d[5] * c.d[6];
  const long long total = G * M * N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long r = i;
    const int n1 = (int)(r % c.d[6]); r /= c.d[6];
    const int n0 = (int)(r % c.d[5]); r /= c.d[5];
    const int m1 = (int)(r % c.d[4]); r /= c.d[4];
    const int m0 = (int)(r % c.d[3]); r /= c.d[3];
    const int g2 = (int)(r % c.d[2]); r /= c.d[2];
    const int g1 = (int)(r % c.d[1]); r /= c.d[1];
    const int g0 = (int)r;
    const float* a = c.A + g0 * c.sA[0] + g1 * c.sA[1] + g2 * c.sA[2] + m0 * c.sA[3] + m1 * c.sA[4];
    const float* b = c.B + g0 * c.sB[0] + g1 * c.sB[1] + g2 * c.sB[2] + n0 * c.sB[3] + n1 * c.sB[4];
    float acc = 0.f;
    for (int k0 = 0; k0 < c.d[7]; ++k0)
      for (int k1 = 0; k1 < c.d[8]; ++k1) acc = fmaf(a[k0 * c.sA[5] + k1 * c.sA[6]], b[k0 * c.sB[5] + k1 * c.sB[6]], acc);
    float* o = c.C + g0 * c.sC[0] + g1 * c.sC[1] + g2 * c.sC[2] + m0 * c.sC[3] + m1 * c.sC[4] + n0 * c.sC[5] + n1 * c.sC[6];
    *o = c.accumulate ? *o + c.alpha * acc : c.alpha * acc;
  }
}

// LDS-tiled form for the larger contractions (the 49 x 49 x 64 windows of the Swin stage are 3/4 of the naive kernel's time): one block
// = one group g and a 32 x 32 tile of (m, n); the K axis goes through LDS in chunks of 32 (gathered through the strides, the axis with
// the smaller stride fastest across the lanes), every thread owns a 2 x 2 patch of outputs.
constexpr int kCT = 32;
__global__ __launch_bounds__(256) void contract_tiled_kernel(Contract c, int a_k_fast, int b_k_fast) {
  __shared__ float sa[kCT][kCT + 1];   // [k][m]
  __shared__ float sb[kCT][kCT + 1];   // [k][n]
  const int M = c.d[3] * c.d[4], N = c.d[5] * c.d[6], K = c.d[7] * c.d[8];
  const int mt = (M + kCT - 1) / kCT, nt = (N + kCT - 1) / kCT;
  long long b = blockIdx.x;
  const int tn = (int)(b % nt); b /= nt;
  const int tm = (int)(b % mt); b /= mt;
  const int g2 = (int)(b % c.d[2]); b /= c.d[2];
  const int g1 = (int)(b % c.d[1]);
  const int g0 = (int)(b / c.d[1]);
  const float* A = c.A + g0 * c.sA[0] + g1 * c.sA[1] + g2 * c.sA[2];
  const float* B = c.B + g0 * c.sB[0] + g1 * c.sB[1] + g2 * c.sB[2];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // outputs (m = 2 ty + {0,1}, n = 2 tx + {0,1}) of the tile
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += kCT) {
    for (int e = threadIdx.x; e < kCT * kCT; e += 256) {
      const int r = a_k_fast ? e / kCT : e % kCT, kk = a_k_fast ? e % kCT : e / kCT;
      const int m = tm * kCT + r, k = k0 + kk;
      float v = 0.f;
      if (m < M && k < K) v = A[(m / c.d[4]) * c.sA[3] + (m % c.d[4]) * c.sA[4] + (k / c.d[8]) * c.sA[5] + (k % c.d[8]) * c.sA[6]];
      sa[kk][r] = v;
    }
    for (int e = threadIdx.x; e < kCT * kCT; e += 256) {
      const int r = b_k_fast ? e / kCT : e % kCT, kk = b_k_fast ? e % kCT : e / kCT;
      const int n = tn * kCT + r, k = k0 + kk;
      float v = 0.f;
      if (n < N && k < K) v = B[(n / c.d[6]) * c.sB[3] + (n % c.d[6]) * c.sB[4] + (k / c.d[8]) * c.sB[5] + (k % c.d[8]) * c.sB[6]];
      sb[kk][r] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < kCT; ++kk) {
      const float a0 = sa[kk][2 * ty], a1 = sa[kk][2 * ty + 1], b0 = sb[kk][2 * tx], b1 = sb[kk][2 * tx + 1];
      acc[0][0] = fmaf(a0, b0, acc[0][0]);
      acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]);
      acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  float* C = c.C + g0 * c.sC[0] + g1 * c.sC[1] + g2 * c.sC[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = tm * kCT + 2 * ty + i, n = tn * kCT + 2 * tx + j;
      if (m < M && n < N) {
        float* o = C + (m / c.d[4]) * c.sC[3] + (m % c.d[4]) * c.sC[4] + (n / c.d[6]) * c.sC[5] + (n % c.d[6]) * c.sC[6];
        *o = c.accumulate ? *o + c.alpha * acc[i][j] : c.alpha * acc[i][j];
      }
    }
}

// softmax over the middle axis of (outer, n, inner) contiguous
__global__ void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, long long outer, int n, int inner) {
  const long long total = outer * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long o = i / inner, in = i - o * inner;
    const float* p = x + o * n * inner + in;
    float m = -3.4e38f;
    for (int k = 0; k < n; ++k) m = fmaxf(m, p[(long long)k * inner]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(p[(long long)k * inner] - m);
    float* q = y + o * n * inner + in;
    const float inv = 1.f / s;
    for (int k = 0; k < n; ++k) q[(long long)k * inner] = expf(p[(long long)k * inner] - m) * inv;
  }
}

__global__ void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long long outer, int n,
                                   int inner) {
  const long long total = outer * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long o = i / inner, in = i - o * inner;
    const long long base = o * n * inner + in;
    float dot = 0.f;
    for (int k = 0; k < n; ++k) dot = fmaf(y[base + (long long)k * inner], dy[base + (long long)k * inner], dot);
    for (int k = 0; k < n; ++k) dx[base + (long long)k * inner] = y[base + (long long)k * inner] * (dy[base + (long long)k * inner] - dot);
  }
}

// LayerNorm backward: one wave per row for dx; per-block partial (dgamma, dbeta) folded by a second kernel
constexpr int kLnRowsPerBlock = 64;
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                            float eps, long long rows, int c, float* __restrict__ dx, double* __restrict__ part) {
  extern __shared__ double acc[];   // [2][c] per block
  for (int i = threadIdx.x; i < 2 * c; i += 256) acc[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * kLnRowsPerBlock;
  // rows are taken by the block's four waves in a fixed order; the per-channel sums are accumulated wave after wave (sequential
  // passes separated by barriers) so that the association order does not depend on scheduling
  for (int pass = 0; pass < kLnRowsPerBlock / 4; ++pass) {
    const long long row = r0 + pass * 4 + wv;
    float xs[16], ds[16];   // c <= 1024: up to 16 values per lane
    const int per = (c + 63) / 64;
    float mean = 0.f, var = 0.f, rstd = 0.f;
    if (row < rows) {
      float s = 0.f;
      for (int k = 0; k < per; ++k) {
        const int ch = lane + 64 * k;
        xs[k] = ch < c ? x[row * c + ch] : 0.f;
        ds[k] = ch < c ? dy[row * c + ch] : 0.f;
        s += xs[k];
      }
      mean = pn::wave_sum(s) / c;
      float v = 0.f;
      for (int k = 0; k < per; ++k) {
        const int ch = lane + 64 * k;
        const float d = ch < c ? xs[k] - mean : 0.f;
        v += d * d;
      }
      var = pn::wave_sum(v) / c;
      rstd = 1.f / sqrtf(var + eps);
      float s1 = 0.f, s2 = 0.f;   // sum(g), sum(g * xhat), g = dy * gamma
      for (int k = 0; k < per; ++k) {
        const int ch = lane + 64 * k;
        if (ch < c) {
          const float g = ds[k] * gamma[ch], xh = (xs[k] - mean) * rstd;
          s1 += g;
          s2 += g * xh;
        }
      }
      s1 = pn::wave_sum(s1);
      s2 = pn::wave_sum(s2);
      for (int k = 0; k < per; ++k) {
        const int ch = lane + 64 * k;
        if (ch < c) {
          const float g = ds[k] * gamma[ch], xh = (xs[k] - mean) * rstd;
          dx[row * c + ch] = rstd * (g - s1 / c - xh * s2 / c);
        }
      }
    }
    for (int w = 0; w < 4; ++w) {   // fixed order: wave 0, 1, 2, 3
      if (wv == w && row < rows) {
        for (int k = 0; k < per; ++k) {
          const int ch = lane + 64 * k;
          if (ch < c) {
            acc[ch] += (double)ds[k] * (double)((xs[k] - mean) * rstd);
            acc[c + ch] += (double)ds[k];
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < 2 * c; i += 256) part[(size_t)blockIdx.x * 2 * c + i] = acc[i];
}

// one block per output value (dgamma[c] or dbeta[c]): lane-strided partial sums over the row blocks, then a fixed-shape tree in LDS
__global__ __launch_bounds__(256) void layernorm_bwd_fold_kernel(const double* __restrict__ part, int nblocks, int c, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int accumulate) {
  __shared__ double red[256];
  const int i = blockIdx.x;   // 0 .. 2c-1
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) t += part[(size_t)b * 2 * c + i];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* dst = i < c ? dgamma + i : dbeta + (i - c);
    *dst = accumulate ? *dst + (float)red[0] : (float)red[0];
  }
}

__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}

// rel[g0, g1, m0, m1, n0, n1][0..1] = a[g.., m0, m1][0..1] - b[g.., n0, n1][0..1] (strides in floats), columns 2..cols-1 zero (pad to the
// MFMA loader's 4 input channels); output contiguous
struct PairDiff {
  const float* a; const float* b; float* rel;
  long long sa[4], sb[4];   // g0 g1 m0 m1 / g0 g1 n0 n1
  int d[6];                 // G0 G1 M0 M1 N0 N1
  int cols;
};
__global__ void pair_diff_kernel(PairDiff p) {
  const long long total = (long long)p.d[0] * p.d[1] * p.d[2] * p.d[3] * p.d[4] * p.d[5];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int n1 = (int)(r % p.d[5]); r /= p.d[5];
    const int n0 = (int)(r % p.d[4]); r /= p.d[4];
    const int m1 = (int)(r % p.d[3]); r /= p.d[3];
    const int m0 = (int)(r % p.d[2]); r /= p.d[2];
    const int g1 = (int)(r % p.d[1]);
    const int g0 = (int)(r / p.d[1]);
    const float* pa = p.a + g0 * p.sa[0] + g1 * p.sa[1] + m0 * p.sa[2] + m1 * p.sa[3];
    const float* pb = p.b + g0 * p.sb[0] + g1 * p.sb[1] + n0 * p.sb[2] + n1 * p.sb[3];
    float* o = p.rel + i * p.cols;
    o[0] = pa[0] - pb[0];
    o[1] = pa[1] - pb[1];
    for (int k = 2; k < p.cols; ++k) o[k] = 0.f;
  }
}

// y = x * m, m in {0, 1 / (1 - p)} drawn per element (row_len == 1) or per row of row_len elements (DropPath, one draw per sample);
// counter-based: the draw of element i depends on (seed, i / row_len) only
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void dropout_kernel(const float* __restrict__ x, size_t n, size_t row_len, float p, uint64_t seed, float* __restrict__ y,
                               float* __restrict__ mask) {
  const float keep_scale = 1.f / (1.f - p);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix64(mix64(seed) ^ (uint64_t)(i / row_len));
    const float u = (float)(h >> 40) * (1.f / 16777216.f);
    const float m = u < p ? 0.f : keep_scale;
    mask[i] = m;
    y[i] = x[i] * m;
  }
}
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] * b[i];
}

// dst[b, idx[b, k, w], w, :] += src[b, k, w, :]   (the (b, row, w) targets of one column are distinct: plain read-modify-write)
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int B, int K, int H, int W, int c,
                                    float* __restrict__ dst) {
  const long long total = (long long)B * K * W * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long long r = i / c;
    const int w = (int)(r % W); r /= W;
    const int k = (int)(r % K);
    const int b = (int)(r / K);
    const int row = idx[((size_t)b * K + k) * W + w];
    dst[(((size_t)b * H + row) * W + w) * c + ch] += src[i];
  }
}

// y[b, h, w, :] = x[b, h, (w - shift) mod W, :]  == torch.roll(x, shift, dims=2)
__global__ void roll_w_kernel(const float* __restrict__ x, int B, int H, int W, int c, int shift, float* __restrict__ y) {
  const long long total = (long long)B * H * W * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long long r = i / c;
    const int w = (int)(r % W);
    const long long bh = r / W;
    const int ws = ((w - shift) % W + W) % W;
    y[i] = x[(bh * W + ws) * c + ch];
  }
}

// rows of length c: y = x / max(||x||, eps)  (F.normalize); one wave per row
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, long long rows, int c, float eps, float* __restrict__ y,
                                                     float* __restrict__ inv_norm) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int k = lane; k < c; k += 64) s += x[row * c + k] * x[row * c + k];
  s = pn::wave_sum(s);
  const float inv = 1.f / fmaxf(sqrtf(s), eps);
  for (int k = lane; k < c; k += 64) y[row * c + k] = x[row * c + k] * inv;
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ inv_norm,
                                                         long long rows, int c, float* __restrict__ dx) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float d = 0.f;
  for (int k = lane; k < c; k += 64) d += y[row * c + k] * dy[row * c + k];
  d = pn::wave_sum(d);
  const float inv = inv_norm[row];
  for (int k = lane; k < c; k += 64) dx[row * c + k] = inv * (dy[row * c + k] - y[row * c + k] * d);   // norm above eps (clamped rows: dx = dy / eps, not handled)
}


// window-attention plumbing of the shifted-window stage (sw2votev4_util.py:140-176): zero-pad a (B, H, W, C) map to (B, Hp, Wp, C) and
// roll it by (-shift, -shift); crop_roll is the adjoint (roll by (+shift, +shift), crop to H x W) and also the forward's way back
__global__ void pad_roll_kernel(const float* __restrict__ x, int B, int H, int W, int Hp, int Wp, int c, int shift, float* __restrict__ y) {
  const long long total = (long long)B * Hp * Wp * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long long r = i / c;
    const int j = (int)(r % Wp); r /= Wp;
    const int ii = (int)(r % Hp);
    const int b = (int)(r / Hp);
    const int h = (ii + shift) % Hp, w = (j + shift) % Wp;
    y[i] = (h < H && w < W) ? x[(((size_t)b * H + h) * W + w) * c + ch] : 0.f;
  }
}
__global__ void crop_roll_kernel(const float* __restrict__ y, int B, int H, int W, int Hp, int Wp, int c, int shift, float* __restrict__ x) {
  const long long total = (long long)B * H * W * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long long r = i / c;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    const int ii = ((h - shift) % Hp + Hp) % Hp, j = ((w - shift) % Wp + Wp) % Wp;
    x[i] = y[(((size_t)b * Hp + ii) * Wp + j) * c + ch];
  }
}
// y[r, c] = x[r, c] * s[c]
__global__ void scale_channels_kernel(const float* __restrict__ x, const float* __restrict__ s, size_t n, int c, float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * s[i % c];
}
// y = 1 / max(x, lo);  dx = x > lo ? -dy / x^2 : 0   (the clamped temperature of the cosine attention, sw2votev4_util.py:84)
__global__ void recip_clamp_kernel(const float* __restrict__ x, float lo, int n, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 1.f / fmaxf(x[i], lo);
}
__global__ void recip_clamp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float lo, int n, float* __restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = x[i] > lo ? -dy[i] / (x[i] * x[i]) : 0.f;
}

unsigned grid_for(long long total, int threads = 256) { return (unsigned)std::min<long long>(65535, (total + threads - 1) / threads); }

}  // namespace

extern "C" {

int pn_contract_f32(const float* a, const int64_t* stride_a, const float* b, const int64_t* stride_b, float* c, const int64_t* stride_c,
                    const int32_t* dims, float alpha, int accumulate, pn_stream_t stream) {
  PN_REQUIRE(a && b && c && stride_a && stride_b && stride_c && dims, "contract: null pointer");
  Contract k;
  k.A = a; k.B = b; k.C = c; k.alpha = alpha; k.accumulate = accumulate;
  long long total = 1;
  for (int i = 0; i < 9; ++i) {
    PN_REQUIRE(dims[i] >= 1, "contract: dimensions must be >= 1");
    k.d[i] = dims[i];
    if (i < 7) total *= dims[i];
  }
  for (int i = 0; i < 7; ++i) {
    k.sA[i] = stride_a[i]; k.sB[i] = stride_b[i]; k.sC[i] = stride_c[i];
  }
  const long long M = (long long)dims[3] * dims[4], N = (long long)dims[5] * dims[6], K = (long long)dims[7] * dims[8];
  const long long G = (long long)dims[0] * dims[1] * dims[2];
  const long long blocks = G * ((M + kCT - 1) / kCT) * ((N + kCT - 1) / kCT);
  if (M * N >= 512 && K >= 8 && blocks < (1ll << 31)) {
    // lanes run along the axis with the smaller innermost stride (the k index is the same accumulation order either way)
    const int a_k_fast = llabs(stride_a[dims[8] > 1 ? 6 : 5]) <= llabs(stride_a[dims[4] > 1 ? 4 : 3]);
    const int b_k_fast = llabs(stride_b[dims[8] > 1 ? 6 : 5]) <= llabs(stride_b[dims[6] > 1 ? 4 : 3]);
    hipLaunchKernelGGL(contract_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, pn::S(stream), k, a_k_fast, b_k_fast);
    return pn::check_launch("contract_tiled_kernel");
  }
  hipLaunchKernelGGL(contract_kernel, dim3(grid_for(total)), dim3(256), 0, pn::S(stream), k);
  return pn::check_launch("contract_kernel");
}

int pn_softmax_f32(const float* x, float* y, long long outer, int n, int inner, pn_stream_t stream) {
  PN_REQUIRE(x && y && outer >= 1 && n >= 1 && inner >= 1, "softmax: bad arguments");
  hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(outer * inner)), dim3(256), 0, pn::S(stream), x, y, outer, n, inner);
  return pn::check_launch("softmax_kernel");
}

int pn_softmax_bwd_f32(const float* y, const float* dy, float* dx, long long outer, int n, int inner, pn_stream_t stream) {
  PN_REQUIRE(y && dy && dx && outer >= 1 && n >= 1 && inner >= 1, "softmax_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(outer * inner)), dim3(256), 0, pn::S(stream), y, dy, dx, outer, n, inner);
  return pn::check_launch("softmax_bwd_kernel");
}

size_t pn_layernorm_bwd_workspace_bytes(long long rows, int c) {
  return (size_t)((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock) * 2 * c * sizeof(double);
}

int pn_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float eps, long long rows, int c, float* dx, float* dgamma,
                         float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && workspace && rows >= 1 && c >= 1 && c <= 1024, "layernorm_bwd: bad arguments (c <= 1024)");
  if (workspace_bytes < pn_layernorm_bwd_workspace_bytes(rows, c)) return pn::fail(PN_ERR_WORKSPACE, "layernorm_bwd: workspace too small");
  const int nb = (int)((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), (size_t)2 * c * sizeof(double), pn::S(stream), x, dy, gamma, eps, rows, c, dx,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(layernorm_bwd_fold_kernel, dim3(2 * c), dim3(256), 0, pn::S(stream), static_cast<const double*>(workspace), nb, c, dgamma, dbeta,
                     accumulate);
  return pn::check_launch("layernorm_bwd");
}

int pn_gelu_f32(const float* x, float* y, size_t n, pn_stream_t stream) {
  PN_REQUIRE(x && y, "gelu: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(gelu_kernel, dim3(grid_for((long long)n)), dim3(256), 0, pn::S(stream), x, y, n);
  return pn::check_launch("gelu_kernel");
}

int pn_gelu_bwd_f32(const float* x, const float* dy, float* dx, size_t n, pn_stream_t stream) {
  PN_REQUIRE(x && dy && dx, "gelu_bwd: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for((long long)n)), dim3(256), 0, pn::S(stream), x, dy, dx, n);
  return pn::check_launch("gelu_bwd_kernel");
}

int pn_pair_diff_f32(const float* a, const int64_t* stride_a, const float* b, const int64_t* stride_b, const int32_t* dims, int cols, float* rel,
                     pn_stream_t stream) {
  PN_REQUIRE(a && b && rel && stride_a && stride_b && dims && cols >= 2, "pair_diff: bad arguments");
  PairDiff p;
  p.a = a; p.b = b; p.rel = rel; p.cols = cols;
  long long total = 1;
  for (int i = 0; i < 6; ++i) {
    PN_REQUIRE(dims[i] >= 1, "pair_diff: dimensions must be >= 1");
    p.d[i] = dims[i];
    total *= dims[i];
  }
  for (int i = 0; i < 4; ++i) {
    p.sa[i] = stride_a[i];
    p.sb[i] = stride_b[i];
  }
  hipLaunchKernelGGL(pair_diff_kernel, dim3(grid_for(total)), dim3(256), 0, pn::S(stream), p);
  return pn::check_launch("pair_diff_kernel");
}

int pn_dropout_f32(const float* x, size_t n, size_t row_len, float p, uint64_t seed, float* y, float* mask, pn_stream_t stream) {
  PN_REQUIRE(x && y && mask && row_len >= 1 && p >= 0.f && p < 1.f, "dropout: bad arguments (0 <= p < 1)");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((long long)n)), dim3(256), 0, pn::S(stream), x, n, row_len, p, seed, y, mask);
  return pn::check_launch("dropout_kernel");
}

int pn_mul_f32(const float* a, const float* b, float* y, size_t n, pn_stream_t stream) {
  PN_REQUIRE(a && b && y, "mul: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for((long long)n)), dim3(256), 0, pn::S(stream), a, b, y, n);
  return pn::check_launch("mul_kernel");
}

int pn_scatter_rows_f32(const float* src, const int32_t* index, int batch, int k, int h, int w, int c, float* dst, pn_stream_t stream) {
  PN_REQUIRE(src && index && dst && batch >= 1 && k >= 1 && h >= 1 && w >= 1 && c >= 1, "scatter_rows: bad arguments");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long long)batch * k * w * c)), dim3(256), 0, pn::S(stream), src, index, batch, k, h, w, c, dst);
  return pn::check_launch("scatter_rows_kernel");
}

int pn_roll_w_f32(const float* x, int batch, int h, int w, int c, int shift, float* y, pn_stream_t stream) {
  PN_REQUIRE(x && y && x != y && batch >= 1 && h >= 1 && w >= 1 && c >= 1, "roll_w: bad arguments");
  hipLaunchKernelGGL(roll_w_kernel, dim3(grid_for((long long)batch * h * w * c)), dim3(256), 0, pn::S(stream), x, batch, h, w, c, shift, y);
  return pn::check_launch("roll_w_kernel");
}

int pn_l2_normalize_f32(const float* x, long long rows, int c, float eps, float* y, float* inv_norm, pn_stream_t stream) {
  PN_REQUIRE(x && y && rows >= 1 && c >= 1, "l2_normalize: bad arguments");
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, pn::S(stream), x, rows, c, eps, y, inv_norm);
  return pn::check_launch("l2norm_kernel");
}

int pn_l2_normalize_bwd_f32(const float* y, const float* dy, const float* inv_norm, long long rows, int c, float* dx, pn_stream_t stream) {
  PN_REQUIRE(y && dy && inv_norm && dx && rows >= 1 && c >= 1, "l2_normalize_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, pn::S(stream), y, dy, inv_norm, rows, c, dx);
  return pn::check_launch("l2norm_bwd_kernel");
}

int pn_pad_roll_f32(const float* x, int batch, int h, int w, int hp, int wp, int c, int shift, float* y, pn_stream_t stream) {
  PN_REQUIRE(x && y && batch >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w && c >= 1 && shift >= 0, "pad_roll: bad arguments");
  hipLaunchKernelGGL(pad_roll_kernel, dim3(grid_for((long long)batch * hp * wp * c)), dim3(256), 0, pn::S(stream), x, batch, h, w, hp, wp, c, shift, y);
  return pn::check_launch("pad_roll_kernel");
}

int pn_crop_roll_f32(const float* y, int batch, int h, int w, int hp, int wp, int c, int shift, float* x, pn_stream_t stream) {
  PN_REQUIRE(x && y && batch >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w && c >= 1 && shift >= 0, "crop_roll: bad arguments");
  hipLaunchKernelGGL(crop_roll_kernel, dim3(grid_for((long long)batch * h * w * c)), dim3(256), 0, pn::S(stream), y, batch, h, w, hp, wp, c, shift, x);
  return pn::check_launch("crop_roll_kernel");
}

int pn_scale_channels_f32(const float* x, const float* scale, size_t n, int c, float* y, pn_stream_t stream) {
  PN_REQUIRE(x && scale && y && c >= 1, "scale_channels: bad arguments");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for((long long)n)), dim3(256), 0, pn::S(stream), x, scale, n, c, y);
  return pn::check_launch("scale_channels_kernel");
}

int pn_recip_clamp_f32(const float* x, float lo, int n, float* y, pn_stream_t stream) {
  PN_REQUIRE(x && y && n >= 1, "recip_clamp: bad arguments");
  hipLaunchKernelGGL(recip_clamp_kernel, dim3(pn::cdiv(n, 256)), dim3(256), 0, pn::S(stream), x, lo, n, y);
  return pn::check_launch("recip_clamp_kernel");
}

int pn_recip_clamp_bwd_f32(const float* x, const float* dy, float lo, int n, float* dx, pn_stream_t stream) {
  PN_REQUIRE(x && dy && dx && n >= 1, "recip_clamp_bwd: bad arguments");
  hipLaunchKernelGGL(recip_clamp_bwd_kernel, dim3(pn::cdiv(n, 256)), dim3(256), 0, pn::S(stream), x, dy, lo, n, dx);
  return pn::check_launch("recip_clamp_bwd_kernel");
}

}  // extern "C"
