// assign_mfma.hip — exact nearest-centroid assignment with an f32-MFMA filter.
//
// Reference semantics: assign_points_brute_force / find_nearest_centroid (src/kmeans.rs:355-373,
// 462-470): label = arg-min over ALL centroids of compute_distance_simd, strict '<' (lowest index wins).
//
// N x k x D is GEMM shaped, so the bulk runs on the matrix cores as  m(i,c) = ||c||^2 - 2 x_i.c
// (v_mfma_f32_32x32x2_f32, exact f32 products, 157 TF = the f32 vector peak but 2 flops per
// (i,c,d) instead of the 3 VALU ops of the exact-order chain, and the VALU stays free for the
// arg-min epilogue).  m differs from the reference's lane-ordered sum only by rounding, and that
// difference is BOUNDED:
//
//   |m(i,c) + ||x_i||^2 - Dist(i,c)| <= E_i   = (D+2) u' (||x_i||^2 + 2 max_c ||c||^2)      (fma chain of D+1 terms + norm rounding)
//   |d_ref(i,c) - Dist(i,c)|         <= G_i   = (D/8+8) u' 2 (||x_i||^2 + max_c ||c||^2)   (reference's own rounding)
//   u' = 2^-24 * 1.01
//
// A row whose best and second-best m are further apart than 2(E_i + G_i) has a provably unique
// reference arg-min and keeps the MFMA label; every other row ("ambiguous": near ties, exact ties,
// duplicate centroids, NaNs) is re-evaluated by the exact-order scan kernel over all centroids.
// Labels are therefore bit-identical to assign_points_brute_force for every input.
//
// Tiling (per workgroup = 4 waves, 256 points; per wave 64 points x 64 centroids per step):
//   B operand = X^T : each wave keeps its 64 points' D <= 128 dims in REGISTERS for the whole sweep
//                     (2 x 16 float4 = 128 VGPRs, pre-scaled by -2), so X is read from HBM once;
//   A operand = C   : 64-centroid tiles streamed through LDS (double buffered, 132-float row
//                     stride => conflict-free ds_read_b128), shared by the 4 waves;
//   D             : D[centroid][point] - the point sits on the lane, the 16 accumulator registers are
//                     16 centroids, so the running (best, second best, arg) update is lane-local:
//                     v_cmp + v_cndmask + v_med3 + v_min per candidate, no cross-lane traffic until
//                     the two lane halves are merged once at the very end;
//   accumulators are initialised with ||c||^2 (no extra K step).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "assign_mfma.hpp"
#include "device_math.hpp"
#include "mfma_bf16.hpp"
#include "wave_sort.hpp"

namespace vi {
namespace {


constexpr int kTileC = 64;          // centroids per LDS tile
constexpr int kRowStride = 132;     // floats per LDS row (528 B: conflict-free b128 reads)
constexpr int kTileFloats = kTileC * kRowStride + kTileC;  // rows + norms

__global__ void centroid_norm_kernel(const float *C, uint32_t k, uint32_t d, float *cn) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  double s = 0.0;
  for (uint32_t j = 0; j < d; ++j) { const double v = C[(size_t)c * d + j]; s += v * v; }
  cn[c] = (float)s;
}

struct MfmaArgs {
  const float *X;
  uint32_t n, dim;
  const float *C, *cn;
  uint32_t k;
  float margin_scale_x, margin_const;  // margin_i = margin_scale_x * ||x_i||^2 + margin_const
  uint32_t *label, *amb_list, *namb;
  const float4 *img;  // bf16 kernels: hi/lo images of the centroid tiles (mfma_bf16.hpp)
};

// C tile staging, split in two halves (issue-early / write-late): the global loads of tile t+1 are
// issued before the MFMAs of tile t and land in LDS after them, so their latency is hidden.
// dim % 4 == 0 is required (mfma_assign_supported); columns >= dim are zero, rows >= k get a +inf
// norm so they never win.
template <int NG>
struct TileRegs {
  static constexpr int kNvec = 2 * NG;                 // float4 per row
  static constexpr int kPerThread = kTileC * kNvec / 256;
  float4 v[kPerThread];
  float norm;
};

template <int NG>
__device__ __forceinline__ void tile_load(TileRegs<NG> &t, const float *C, const float *cn, uint32_t k, uint32_t dim,
                                          uint32_t tile) {
  const uint32_t row0 = tile * kTileC;
#pragma unroll
  for (int i = 0; i < TileRegs<NG>::kPerThread; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int r = idx / TileRegs<NG>::kNvec, c4 = idx % TileRegs<NG>::kNvec;
    const uint32_t row = row0 + r;
    t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < k && (uint32_t)(c4 * 4) < dim)
      t.v[i] = *reinterpret_cast<const float4 *>(C + (size_t)row * dim + c4 * 4);
  }
  t.norm = INFINITY;
  if (threadIdx.x < kTileC && row0 + threadIdx.x < k) t.norm = cn[row0 + threadIdx.x];
}

template <int NG>
__device__ __forceinline__ void tile_write(const TileRegs<NG> &t, float *buf) {
#pragma unroll
  for (int i = 0; i < TileRegs<NG>::kPerThread; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int r = idx / TileRegs<NG>::kNvec, c4 = idx % TileRegs<NG>::kNvec;
    *reinterpret_cast<float4 *>(buf + r * kRowStride + c4 * 4) = t.v[i];
  }
  if (threadIdx.x < kTileC) buf[kTileC * kRowStride + threadIdx.x] = t.norm;
}

template <int NG, int NP>  // dims padded to 8*NG (NG <= 16); NP x 32 points per wave
__global__ void __launch_bounds__(256, 2) mfma_assign_kernel(MfmaArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t pbase = blockIdx.x * (128 * NP) + wave * (32 * NP);

  // ---- this wave's 64 points -> B fragments in registers, scaled by -2 (exact) ----
  float4 xf[NP][NG];
  float xnv[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    xnv[p] = 0.0f;
    const uint32_t pt = pbase + p * 32 + j;
    const bool live = pt < a.n;
    const float *row = a.X + (size_t)(live ? pt : 0) * a.dim;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t e = 8 * g + 4 * h;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && e < a.dim) v = *reinterpret_cast<const float4 *>(row + e);
      xnv[p] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      xf[p][g] = make_float4(-2.f * v.x, -2.f * v.y, -2.f * v.z, -2.f * v.w);
    }
  }
  float b1[NP], b2[NP];
  uint32_t code[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) { b1[p] = INFINITY; b2[p] = INFINITY; code[p] = 0u; }

  const uint32_t ntiles = (a.k + kTileC - 1) / kTileC;
  TileRegs<NG> stage;
  tile_load<NG>(stage, a.C, a.cn, a.k, a.dim, 0);
  tile_write<NG>(stage, lds);
  __syncthreads();
  for (uint32_t ct = 0; ct < ntiles; ++ct) {
    float *cur = lds + (ct & 1) * kTileFloats;
    const bool more = ct + 1 < ntiles;
    if (more) tile_load<NG>(stage, a.C, a.cn, a.k, a.dim, ct + 1);  // in flight during this tile's MFMAs

    // accumulators start at ||c||^2 of their centroid row: rows 8*q + 4*h + (0..3) for regs 4q..4q+3
    f32x16 acc[NP][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 nn = *reinterpret_cast<const float4 *>(cur + kTileC * kRowStride + rt * 32 + 8 * q + 4 * h);
        acc[0][rt][4 * q + 0] = nn.x; acc[0][rt][4 * q + 1] = nn.y;
        acc[0][rt][4 * q + 2] = nn.z; acc[0][rt][4 * q + 3] = nn.w;
      }
#pragma unroll
    for (int p = 1; p < NP; ++p) { acc[p][0] = acc[0][0]; acc[p][1] = acc[0][1]; }

#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // A fragments: centroid row j (and 32 + j), dims 8g + 4h .. +3 — the same k permutation as xf
      const float4 a0 = *reinterpret_cast<const float4 *>(cur + j * kRowStride + 8 * g + 4 * h);
      const float4 a1 = *reinterpret_cast<const float4 *>(cur + (32 + j) * kRowStride + 8 * g + 4 * h);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float bv = t == 0 ? xf[p][g].x : t == 1 ? xf[p][g].y : t == 2 ? xf[p][g].z : xf[p][g].w;
          acc[p][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv, acc[p][0], 0, 0, 0);
          acc[p][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv, acc[p][1], 0, 0, 0);
        }
      }
    }

    // lane-local running best / second best / code (code = tile*32 + rt*16 + reg)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[p][rt][r];
          const uint32_t cd = ct * 32 + rt * 16 + r;
          code[p] = v < b1[p] ? cd : code[p];
          b2[p] = __builtin_amdgcn_fmed3f(b1[p], b2[p], v);
          b1[p] = fminf(b1[p], v);
        }
    if (more) tile_write<NG>(stage, lds + ((ct + 1) & 1) * kTileFloats);  // that buffer was last read before the previous barrier
    __syncthreads();  // everyone is done with `cur`, and the next tile is staged
  }

  // ---- merge the two lane halves (same point, disjoint centroid rows) ----
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t cd = code[p];
    const uint32_t r = cd & 15, rt = (cd >> 4) & 1, ct = cd >> 5;
    uint32_t cen = ct * kTileC + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const float ob1 = __shfl_xor(b1[p], 32), ob2 = __shfl_xor(b2[p], 32);
    const uint32_t ocen = (uint32_t)__shfl_xor((int)cen, 32);
    const float x2 = xnv[p];
    const float xnt = x2 + __shfl_xor(x2, 32);
    const float nb1 = fminf(b1[p], ob1);
    const float nb2 = fminf(fmaxf(b1[p], ob1), fminf(b2[p], ob2));
    if (ob1 < b1[p] || (ob1 == b1[p] && ocen < cen)) cen = ocen;
    const uint32_t pt = pbase + p * 32 + j;
    if (h == 0 && pt < a.n) {
      const float margin = a.margin_scale_x * xnt + a.margin_const;
      const bool sure = (nb2 - nb1) > margin;  // false for NaN / inf-inf
      a.label[pt] = cen < a.k ? cen : 0u;
      if (!sure) a.amb_list[atomicAdd(a.namb, 1u)] = pt;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bf16 x 3 variant: the same sweep on the bf16 matrix pipe (16x the f32 rate, 3 products per multiply)
// ------------------------------------------------------------------------------------------
// centroid rows (row-major k x d) -> per 64-centroid tile the A-operand image of mfma_bf16.hpp
__global__ void centroid_image_kernel(const float *C, uint32_t k, uint32_t d, uint32_t nc, uint4 *img) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (tile, chunk, half, vector)
  const uint64_t ntiles = (k + 63) / 64;
  if (t >= ntiles * nc * 128) return;
  const uint32_t v = (uint32_t)(t & 63), h = (uint32_t)((t >> 6) & 1);
  const uint64_t bc = t >> 7;
  const uint32_t c = (uint32_t)(bc % nc);
  const uint64_t b = bc / nc;
  const uint64_t row = b * 64 + v;
  const uint32_t e = 16 * c + 8 * h;
  float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
  if (row < k && e < d) x0 = *reinterpret_cast<const float4 *>(C + row * d + e);
  if (row < k && e + 4 < d) x1 = *reinterpret_cast<const float4 *>(C + row * d + e + 4);
  uint4 hi, lo;
  split8(x0, x1, 1.0f, hi, lo);
  uint4 *dst = img + ((b * nc + c) * 4) * 64;
  dst[(0 * 2 + h) * 64 + v] = hi;
  dst[(1 * 2 + h) * 64 + v] = lo;
}

__global__ void centroid_norm_pad_kernel(const float *cn, uint32_t k, uint32_t kpad, float *out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < kpad) out[c] = c < k ? cn[c] : INFINITY;  // pad rows never win
}

template <int NG>  // dims padded to 8*NG (NG even <= 16); 32 points per wave, 128 per workgroup
__global__ void __launch_bounds__(256, 2) mfma_assign_bf16_kernel(MfmaArgs a) {
  extern __shared__ float lds[];
  constexpr int kImgFloats = 2 * NG * 256 + 64;  // image + 64 norms
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t pt = blockIdx.x * 128 + wave * 32 + j;
  const bool live = pt < a.n;
  const float *row = a.X + (size_t)(live ? pt : 0) * a.dim;
  // B fragments: this lane's point, dims 16c+8h.., scaled by -2 (exact), split hi/lo: xf[2c] = hi, xf[2c+1] = lo
  float4 xf[NG];
  float xnv = 0.0f;
#pragma unroll
  for (int c = 0; c < NG / 2; ++c) {
    const uint32_t e = 16 * c + 8 * h;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (live && e < a.dim) v0 = *reinterpret_cast<const float4 *>(row + e);
    if (live && e + 4 < a.dim) v1 = *reinterpret_cast<const float4 *>(row + e + 4);
    xnv += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    uint4 hi, lo;
    split8(v0, v1, -2.0f, hi, lo);
    xf[2 * c] = __builtin_bit_cast(float4, hi);
    xf[2 * c + 1] = __builtin_bit_cast(float4, lo);
  }
  float b1 = INFINITY, b2 = INFINITY;
  uint32_t code = 0u;

  const uint32_t ntiles = (a.k + kTileC - 1) / kTileC;
  const size_t img_stride = (size_t)2 * NG * 64;  // float4 per tile image
  tile_dma_image<NG>(lds, a.img, a.cn, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (uint32_t ct = 0; ct < ntiles; ++ct) {
    const float *cur = lds + (ct & 1) * kImgFloats;
    // next tile lands in the other buffer during this tile's MFMAs (everyone left it at the last barrier)
    // (inline-asm copies: with the builtin hipcc waits vmcnt(0) before the first ds_read below — it cannot tell the two
    // buffers apart — and the load would not overlap the MFMAs)
    if (ct + 1 < ntiles)
      tile_dma_image_asm<NG>(lds + ((ct + 1) & 1) * kImgFloats, a.img + (ct + 1) * img_stride,
                             a.cn + (size_t)(ct + 1) * kTileC, wave, lane);
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // rows 8q + 4h + (0..3) live in regs 4q .. 4q+3
      const float4 n0 = *reinterpret_cast<const float4 *>(cur + 2 * NG * 256 + 8 * q + 4 * h);
      const float4 n1 = *reinterpret_cast<const float4 *>(cur + 2 * NG * 256 + 32 + 8 * q + 4 * h);
      acc0[4 * q + 0] = n0.x; acc0[4 * q + 1] = n0.y; acc0[4 * q + 2] = n0.z; acc0[4 * q + 3] = n0.w;
      acc1[4 * q + 0] = n1.x; acc1[4 * q + 1] = n1.y; acc1[4 * q + 2] = n1.z; acc1[4 * q + 3] = n1.w;
    }
    auto frag = [&](int c, int p, int t) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(cur + (((c * 2 + p) * 2 + h) * 64 + 32 * t + j) * 4));
    };
    bf16x8 h0 = frag(0, 0, 0), h1 = frag(0, 0, 1);
#pragma unroll
    for (int c = 0; c < NG / 2; ++c) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, xf[2 * c]), bl = __builtin_bit_cast(bf16x8, xf[2 * c + 1]);
      const bf16x8 l0 = frag(c, 1, 0), l1 = frag(c, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bh, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bl, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bl, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < NG / 2) { h0 = frag(c + 1, 0, 0); h1 = frag(c + 1, 0, 1); }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, bh, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane-local running best / second best / code (code = tile*32 + rt*16 + reg)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc0[r];
      code = v < b1 ? ct * 32 + r : code;
      b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
      b1 = fminf(b1, v);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc1[r];
      code = v < b1 ? ct * 32 + 16 + r : code;
      b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
      b1 = fminf(b1, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next tile visible; this one free to be overwritten
  }
  // ---- merge the two lane halves (same point, disjoint centroid rows) ----
  const uint32_t r = code & 15, rt = (code >> 4) & 1, ctb = code >> 5;
  uint32_t cen = ctb * kTileC + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
  const float ob1 = __shfl_xor(b1, 32), ob2 = __shfl_xor(b2, 32);
  const uint32_t ocen = (uint32_t)__shfl_xor((int)cen, 32);
  const float xnt = xnv + __shfl_xor(xnv, 32);
  const float nb1 = fminf(b1, ob1);
  const float nb2 = fminf(fmaxf(b1, ob1), fminf(b2, ob2));
  if (ob1 < b1 || (ob1 == b1 && ocen < cen)) cen = ocen;
  if (h == 0 && live) {
    const float margin = a.margin_scale_x * xnt + a.margin_const;
    const bool sure = (nb2 - nb1) > margin;  // false for NaN / inf-inf
    a.label[pt] = cen < a.k ? cen : 0u;
    if (!sure) a.amb_list[atomicAdd(a.namb, 1u)] = pt;
  }
}

// ------------------------------------------------------------------------------------------
// hi-only candidate sweep: ONE bf16 MFMA per product instead of three
// ------------------------------------------------------------------------------------------
// m0(i,c) = ||c||^2 - 2 hi(x_i).hi(c) differs from ||c||^2 - 2 x_i.c by at most E0_i = 2^-7 (1 + 2^-9) (||x_i||^2 +
// max||c||^2) + the accumulation error — bf16 keeps 8 significant bits (|x - hi(x)| <= 2^-8 |x|), so
// 2 |x.c - hi(x).hi(c)| <= 2^-6 (1 + 2^-9) |x||c|.  That is far too coarse to NAME the nearest centroid (the best and
// second best of a point are ~2 % apart: 3.6 of 160 at C3, the margin is 5), but it is enough to name the few that can
// be it: with c* the reference's arg-min,
//     m0(i,c*) <= m0(i,c') + 2 (E0_i + G_i)      for EVERY c'                               (**)
// (G_i: the reference's own rounding, file header), in particular for the centroid that holds the running minimum
// when c* comes by.  So a sweep that lists, per point, every centroid whose m0 is within the margin of the running
// minimum lists c* (and every centroid tied with it), and the exact lane-order distance (src/kmeans.rs:377-419) of
// the listed few decides — ~6 candidates per point at C3 once the running minimum has seen 1 024 centroids (the first
// `warm` tiles are swept without listing and once more at the end), 2.5 of which are within the margin of the final
// minimum.  Points that list more than kCandCap per lane half (or nothing: NaN rows) go to the tiers above.
//
// Tiling: workgroup = 4 waves x 64 points; a wave keeps its 2 x 32 points' hi planes in registers (B operand, 64 VGPRs at
// D = 128) and multiplies each LDS fragment of the centroid tile with both — 32 MFMAs and 16 ds_read_b128 per 64
// centroids.  The epilogue of a 32 x 32 accumulator tile is its minimum (12 instructions) and one compare against the
// lane's threshold; only when some lane has a candidate (27 % of the tiles) are the 16 values looked at one by one.
constexpr int kCandCap = 16;     // candidates a (point, lane half) may list
constexpr uint32_t kCandWarmTiles = 16;

struct CandArgs {
  const float *X;
  uint32_t n, dim;
  const float4 *img;  // hi image of the centroid tiles: [tile][chunk][half of 8 dims][64] x 16 B
  const float *cn;    // norms padded to whole tiles (+inf)
  uint32_t k, ntiles, warm;
  float margin_scale_x, margin_const;
  uint32_t *cand, *cand_cnt;  // [(point * 2 + lane half) * kCandCap + slot], [point * 2 + lane half]
  float *cand_thr;            // [point]: the final threshold (minimum + margin) — entries listed under an earlier, looser one are dropped
  uint32_t pack16;            // k <= 65536: an entry is centroid | (its m0 rounded DOWN to bf16) << 16, so that the exact pass can drop them
  uint32_t xmode;             // ablation knob (VI_CAND_XMODE, wrong results): 1 no candidate listing, 2 no epilogue at all, 4 no tile copy / barrier
};

__global__ void centroid_hi_image_kernel(const float *C, uint32_t k, uint32_t d, uint32_t nc, uint4 *img) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (tile, chunk, half, vector)
  const uint64_t ntiles = (k + 63) / 64;
  if (t >= ntiles * nc * 128) return;
  const uint32_t v = (uint32_t)(t & 63), h = (uint32_t)((t >> 6) & 1);
  const uint64_t bc = t >> 7;
  const uint32_t c = (uint32_t)(bc % nc);
  const uint64_t b = bc / nc;
  const uint64_t row = b * 64 + v;
  const uint32_t e = 16 * c + 8 * h;
  float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
  if (row < k && e < d) x0 = *reinterpret_cast<const float4 *>(C + row * d + e);
  if (row < k && e + 4 < d) x1 = *reinterpret_cast<const float4 *>(C + row * d + e + 4);
  uint4 hi, lo;
  split8(x0, x1, 1.0f, hi, lo);
  img[((b * nc + c) * 2 + h) * 64 + v] = hi;
}

template <int NC>  // dims padded to 16 * NC (NC <= 8)
__global__ void __launch_bounds__(256, 2) mfma_assign_cand_kernel(CandArgs a) {
  extern __shared__ float lds[];
  constexpr int kImgFloats = 2 * NC * 256 + 64;  // the tile's hi image + 64 norms
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  // B fragments: two points per lane (tile p: point 32p + j of the wave), dims 16c + 8h .., scaled by -2 (exact), hi plane
  float4 xb[2][NC];
  float margin[2];
  uint32_t slot0[2];
  bool live[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint32_t pt = blockIdx.x * 256 + wave * 64 + 32 * p + j;
    live[p] = pt < a.n;
    const float *row = a.X + (size_t)(live[p] ? pt : 0) * a.dim;
    float xnv = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const uint32_t e = 16 * c + 8 * h;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (live[p] && e < a.dim) v0 = *reinterpret_cast<const float4 *>(row + e);
      if (live[p] && e + 4 < a.dim) v1 = *reinterpret_cast<const float4 *>(row + e + 4);
      xnv += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
      uint4 hi, lo;
      split8(v0, v1, -2.0f, hi, lo);
      xb[p][c] = __builtin_bit_cast(float4, hi);
    }
    const float xnt = xnv + __shfl_xor(xnv, 32);
    margin[p] = a.margin_scale_x * xnt + a.margin_const;
    slot0[p] = (pt * 2u + (uint32_t)h) * (uint32_t)kCandCap;
  }
  float b1[2] = {INFINITY, INFINITY}, thr[2] = {INFINITY, INFINITY};
  uint32_t cnt[2] = {0u, 0u};
  uint32_t *lists = reinterpret_cast<uint32_t *>(lds + 2 * kImgFloats);  // [point tile][slot][thread]: 2 x kCandCap x 256 words

  const size_t img_stride = (size_t)2 * NC * 64;  // float4 per tile image
  const uint32_t steps = a.ntiles + a.warm;       // tiles 0 .. ntiles-1 (the first `warm` without listing), then 0 .. warm-1 again
  auto tile_of = [&](uint32_t s) { return s < a.ntiles ? s : s - a.ntiles; };
  auto dma_tile = [&](float *buf, uint32_t tile) {  // asm copies: see mfma_assign_bf16_kernel
    // 2 NC pieces of 1 KB round-robin over the four waves (2 NC need not be a multiple of 4: NC is odd for D = 16, 48, ...)
#pragma unroll
    for (int i = 0; i < (2 * NC + 3) / 4; ++i) {
      const int piece = wave + 4 * i;
      if (piece < 2 * NC) glds16_asm(a.img + tile * img_stride + piece * 64 + lane, buf + piece * 256);
    }
    if (wave == 0) glds4_asm(a.cn + (size_t)tile * kTileC + lane, buf + 2 * NC * 256);
  };
  // one step: tile tile_of(s) sits in buffer s & 1; the next tile's image is requested first
  auto step = [&](uint32_t s) {
    const float *cur = lds + (s & 1) * kImgFloats;
    const uint32_t ct = tile_of(s);
    const bool record = s >= a.warm;
    if (s + 1 < steps && !(a.xmode & 4u)) dma_tile(lds + ((s + 1) & 1) * kImgFloats, tile_of(s + 1));
    f32x16 acc[2][2];  // [centroid row tile][point tile]
    auto frag = [&](int c, int t) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(cur + ((c * 2 + h) * 64 + 32 * t + j) * 4));
    };
    // fragments are requested TWO chunks ahead (three register sets): one chunk's four MFMAs (128 cycles) do not cover
    // an LDS read when eight waves share the port — requested one chunk ahead, every chunk waited (half the MFMA rate)
    bf16x8 fa[3][2];
    fa[0][0] = frag(0, 0); fa[0][1] = frag(0, 1);
    if (NC > 1) { fa[1][0] = frag(1, 0); fa[1][1] = frag(1, 1); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // rows 8q + 4h + (0..3) live in regs 4q .. 4q+3
      const float4 n0 = *reinterpret_cast<const float4 *>(cur + 2 * NC * 256 + 8 * q + 4 * h);
      const float4 n1 = *reinterpret_cast<const float4 *>(cur + 2 * NC * 256 + 32 + 8 * q + 4 * h);
      acc[0][0][4 * q + 0] = n0.x; acc[0][0][4 * q + 1] = n0.y; acc[0][0][4 * q + 2] = n0.z; acc[0][0][4 * q + 3] = n0.w;
      acc[1][0][4 * q + 0] = n1.x; acc[1][0][4 * q + 1] = n1.y; acc[1][0][4 * q + 2] = n1.z; acc[1][0][4 * q + 3] = n1.w;
    }
    acc[0][1] = acc[0][0];
    acc[1][1] = acc[1][0];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bf16x8 p0 = __builtin_bit_cast(bf16x8, xb[0][c]), p1 = __builtin_bit_cast(bf16x8, xb[1][c]);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NC) { fa[(c + 2) % 3][0] = frag(c + 2, 0); fa[(c + 2) % 3][1] = frag(c + 2, 1); }
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c % 3][0], p0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c % 3][1], p0, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c % 3][0], p1, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c % 3][1], p1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.xmode & 2u) {
      if (acc[0][0][0] + acc[1][0][1] + acc[0][1][2] + acc[1][1][3] == 1.2345f) b1[0] = 0.0f;
    } else
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float smin = INFINITY;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float tmin = tile_min(acc[rt][p]);
        const bool flag = record && live[p] && !(tmin > thr[p]) && !(a.xmode & 1u);  // (!(x > thr): a NaN is listed, never skipped)
        if (__ballot(flag)) {  // some lane has a candidate among its 16 values (a third of the tiles): which ones?
          auto append = [&](uint32_t r, float v) {
            uint32_t entry = ct * kTileC + rt * 32 + (r & 3u) + 8u * (r >> 2) + 4u * h;
            if (a.pack16) {  // + m0 rounded toward -inf to 16 bits (never above the value: the later filter only keeps too much)
              const uint32_t vb = __float_as_uint(v);
              entry |= ((vb & 0x80000000u) ? vb + 0xFFFFu : vb) & 0xFFFF0000u;
            }
            if (cnt[p] < (uint32_t)kCandCap) lists[(p * kCandCap + cnt[p]) * 256 + threadIdx.x] = entry;
            ++cnt[p];
          };
          uint32_t hits = 0u;
#pragma unroll
          for (int r = 0; r < 16; ++r) hits |= !(acc[rt][p][r] > thr[p]) ? (1u << r) : 0u;
          hits = flag ? hits : 0u;
          const bool one = hits != 0u && (hits & (hits - 1u)) == 0u;
          if (one) append((uint32_t)__builtin_ctz(hits), tmin);  // the rule: the tile's minimum alone
          const bool more = hits != 0u && !one;
          if (__ballot(more)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (more && ((hits >> r) & 1u)) append((uint32_t)r, acc[rt][p][r]);
          }
        }
        smin = fminf(smin, tmin);
      }
      // the running minimum of the POINT: both lane halves (disjoint centroid rows) share it — (**) holds for every c'
      smin = fminf(smin, __uint_as_float(exchange_u32<Ex::X32>(__float_as_uint(smin), lane)));
      b1[p] = fminf(b1[p], smin);
      thr[p] = b1[p] + margin[p];
    }
    if (!(a.xmode & 4u)) {  // (ablation 4: the same tile again and again, no copy, no barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // next tile visible; this one free to be overwritten
    }
  };
  dma_tile(lds, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (uint32_t s = 0; s < steps; ++s) step(s);
  // the lists leave LDS once (a global store inside the sweep would make the step's vmcnt(0) wait for its round trip)
#pragma unroll
  for (int p = 0; p < 2; ++p)
    if (live[p]) {
      const uint32_t m = min(cnt[p], (uint32_t)kCandCap);
      for (uint32_t i = 0; i < m; ++i) a.cand[slot0[p] + i] = lists[(p * kCandCap + i) * 256 + threadIdx.x];
      a.cand_cnt[slot0[p] / (uint32_t)kCandCap] = cnt[p];
      if (h == 0) a.cand_thr[slot0[p] / (2u * (uint32_t)kCandCap)] = thr[p];
    }
}

// the listed candidates of a point in the reference's own arithmetic: arg-min of compute_distance_simd under
// (distance, centroid index) — find_nearest_centroid's strict '<' over ascending indices (src/kmeans.rs:355-373).
// Points whose lists overflowed, are empty, or hold no finite distance go to the ambiguous list (exact over all centroids).
__global__ void __launch_bounds__(256) cand_exact_kernel(const float *X, uint32_t n, uint32_t d, const float *C, uint32_t k,
                                                         const uint32_t *cand, const uint32_t *cand_cnt, const float *cand_thr,
                                                         uint32_t pack16, uint32_t *label, uint32_t *amb_list, uint32_t *namb) {
  // 8 lanes per point: lane l owns the reference's l-th lane accumulator (dims l, 8 + l, ...: kmeans.rs:387-396), lanes
  // 0..3 its 4-lane accumulators (:399-408), lane 0 the scalar tail (:411-416) and the final reduction (:418).  The
  // point's own values stay in registers (d <= 128: 16 per lane); entries listed under a threshold looser than the final
  // one are dropped first (6 listed, 2.7 kept at C3), the survivors queue up in LDS.
  __shared__ uint32_t s_keep[32][2 * kCandCap];
  const uint32_t grp = threadIdx.x >> 3, pt = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, l = threadIdx.x & 7u;
  const int lane = threadIdx.x & 63, g0 = lane & ~7;
  const bool in = pt < n;
  const uint32_t n0 = in ? cand_cnt[2 * pt] : 0u, n1 = in ? cand_cnt[2 * pt + 1] : 0u;
  const bool ok = in && n0 <= (uint32_t)kCandCap && n1 <= (uint32_t)kCandCap;
  const float thr = in ? cand_thr[pt] : 0.0f;
  const float *x = X + (size_t)(in ? pt : 0u) * d;
  const uint32_t d8 = d & ~7u, has4 = (d & 4u) ? 1u : 0u;
  float xr[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) xr[t] = (uint32_t)(8 * t) < d8 ? x[8 * t + l] : 0.0f;
  // survivors of the point's two lists -> s_keep[grp]
  uint32_t kept = 0;
#pragma unroll
  for (uint32_t r = 0; r < 2 * kCandCap; r += 8) {
    const uint32_t i = r + l;                       // entry i of the concatenated lists
    const bool have = ok && i < n0 + n1;
    uint32_t e = have ? cand[(size_t)(2 * pt + (i < n0 ? 0u : 1u)) * kCandCap + (i < n0 ? i : i - n0)] : 0u;
    bool keep = have;
    if (pack16) {
      keep = keep && !(__uint_as_float(e & 0xFFFF0000u) > thr);  // (its rounded-down m0 against the final threshold; NaN stays)
      e &= 0xFFFFu;
    }
    keep = keep && e < k;  // (pad rows of the last tile carry +inf norms and are never listed; belt and braces)
    const uint32_t bits = (uint32_t)(__ballot(keep) >> g0) & 0xFFu;
    if (keep) s_keep[grp][kept + (uint32_t)__popc(bits & ((1u << l) - 1u))] = e;
    kept += (uint32_t)__popc(bits);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float best = INFINITY;
  uint32_t bc = 0xFFFFFFFFu;
  uint32_t tmax = kept;  // (the wave's eight points go round together)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, o));
  for (uint32_t i = 0; i < tmax; ++i) {
    const bool valid = i < kept;
    const uint32_t c = valid ? s_keep[grp][i] : 0u;
    const float *cr = C + (size_t)c * d;
    float a8 = 0.0f, a4 = 0.0f, tail = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if ((uint32_t)(8 * t) < d8) sq_add(a8, xr[t], cr[8 * t + l]);
    if (has4 && l < 4u) sq_add(a4, x[d8 + l], cr[d8 + l]);
    if (l == 0u)
      for (uint32_t j = d8 + 4u * has4; j < d; ++j) sq_add(tail, x[j], cr[j]);
    float v8[8], v4[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) v8[t] = __shfl(a8, g0 + t);
#pragma unroll
    for (int t = 0; t < 4; ++t) v4[t] = __shfl(a4, g0 + t);
    const float lo = VI_REDUCE4(v8[0], v8[1], v8[2], v8[3]);  // include/vi_reduce_order.h, as l2sq_lanes_dev
    const float hi = VI_REDUCE4(v8[4], v8[5], v8[6], v8[7]);
    const float r4 = VI_REDUCE4(v4[0], v4[1], v4[2], v4[3]);
    const float dist = ((lo + hi) + r4) + __shfl(tail, g0);
    if (valid && (dist < best || (dist == best && c < bc))) { best = dist; bc = c; }
  }
  if (in && l == 0u) {
    if (bc == 0xFFFFFFFFu) {
      label[pt] = 0u;
      amb_list[atomicAdd(namb, 1u)] = pt;
    } else {
      label[pt] = bc;
    }
  }
}

template <int NC>
vi_status launch_cand(const CandArgs &a, hipStream_t st) {
  const size_t smem = 2 * (2 * NC * 256 + 64) * sizeof(float) + 2 * kCandCap * 256 * sizeof(uint32_t);
  VI_HIP(hipFuncSetAttribute((const void *)mfma_assign_cand_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((mfma_assign_cand_kernel<NC>), dim3((a.n + 255) / 256), dim3(256), smem, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

template <int NG>
vi_status launch_mfma_bf16(const MfmaArgs &a, hipStream_t st) {
  const size_t smem = 2 * (2 * NG * 256 + 64) * sizeof(float);
  VI_HIP(hipFuncSetAttribute((const void *)mfma_assign_bf16_kernel<NG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)smem));
  hipLaunchKernelGGL((mfma_assign_bf16_kernel<NG>), dim3((a.n + 127) / 128), dim3(256), smem, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

__global__ void gather_amb_rows_kernel(const float *X, const uint32_t *rows, uint32_t nrows, uint32_t d, float *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread (d % 4 == 0)
  const uint32_t q4 = d / 4;
  if (t >= (uint64_t)nrows * q4) return;
  const uint32_t i = (uint32_t)(t / q4), c = (uint32_t)(t % q4);
  reinterpret_cast<float4 *>(out)[t] = reinterpret_cast<const float4 *>(X + (size_t)rows[i] * d)[c];
}

__global__ void export_rows_kernel(const uint32_t *rows, uint32_t n, uint32_t base, uint32_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rows[i] + base;
}

__global__ void scatter_amb_labels_kernel(const uint32_t *rows, uint32_t nrows, const uint32_t *lab_c, uint32_t *labels) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrows) labels[rows[i]] = lab_c[i];
}

template <int NG, int NP>
vi_status launch_mfma(const MfmaArgs &a, hipStream_t st) {
  const size_t smem = 2 * kTileFloats * sizeof(float);
  VI_HIP(hipFuncSetAttribute((const void *)mfma_assign_kernel<NG, NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)smem));
  constexpr uint32_t ppb = 128 * NP;  // points per workgroup
  hipLaunchKernelGGL((mfma_assign_kernel<NG, NP>), dim3((a.n + ppb - 1) / ppb), dim3(256), smem, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

vi_status launch_mfma_f32(const MfmaArgs &a, int ng, hipStream_t st) {
  if (ng <= 4) return launch_mfma<4, 2>(a, st);
  if (ng <= 8) return launch_mfma<8, 1>(a, st);
  if (ng <= 12) return launch_mfma<12, 1>(a, st);
  return launch_mfma<16, 1>(a, st);  // 128 dims: 2 x 32 points would not fit 256 VGPRs
}

}  // namespace

bool mfma_assign_supported(uint64_t n, uint64_t k, uint32_t d) {
  return d >= 4 && d <= 128 && (d % 4) == 0 && k >= 128 && n >= 1 && k < (1ull << 26);
}

vi_status mfma_assign_device(const float *Xd, uint64_t n, const float *Cd, uint64_t k, uint32_t d,
                             uint32_t *labels_dev, MfmaAssignWs &ws, hipStream_t st, ExactRowsFn exact, void *exact_ctx,
                             MfmaAssignStats *stats) {
  VI_TRY(ws.cn.reserve(k));
  VI_TRY(ws.namb.reserve(1));
  hipLaunchKernelGGL(centroid_norm_kernel, dim3((uint32_t)((k + 255) / 256)), dim3(256), 0, st, Cd, (uint32_t)k, d,
                     ws.cn.p);
  VI_HIP(hipGetLastError());
  std::vector<float> h_cn(k);
  VI_HIP(hipMemcpyAsync(h_cn.data(), ws.cn.p, k * 4, hipMemcpyDeviceToHost, st));
  VI_HIP(hipStreamSynchronize(st));
  double cmax = 0.0;
  for (uint64_t c = 0; c < k; ++c) cmax = std::max(cmax, (double)h_cn[c]);
  // margin_i = 2 (E_i + G_i), see the file header; computed in double, rounded up.  bf16 x 3: E grows to
  // (3D+2) 2u' (accumulating 3D exact bf16 products; 2u' also covers a truncating accumulator) + 2^-15
  // (the dropped lo.lo product and the two split residuals: |x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-17 |x|, so
  // 2 (|xl.cl| + |xr.c| + |x.cr|) <= 2^-14 |x||c| <= 2^-15 (|x|^2 + |c|^2); round 2 budgeted 3 * 2^-18, too little)
  static const bool bf16 = [] { const char *e = getenv("VI_ASSIGN_BF16"); return !(e && *e == '0'); }();
  const double u = 1.01 * std::ldexp(1.0, -24);
  const double e = bf16 ? (3.0 * d + 2.0) * 2.0 * u + 1.01 * std::ldexp(1.0, -15) : (d + 2.0) * u;
  const double e32 = (d + 2.0) * u;
  const double g = (d / 8.0 + 8.0) * u * 2.0;
  MfmaArgs a{};
  a.X = Xd; a.dim = d; a.C = Cd; a.cn = ws.cn.p; a.k = (uint32_t)k;
  a.margin_scale_x = (float)(2.0 * (e + g) * 1.0001);
  a.margin_const = (float)(2.0 * (2.0 * e + g) * cmax * 1.0001);
  a.namb = ws.namb.p;
  const int ng = (int)((d + 7) / 8);
  const int ngb = 2 * (int)((d + 15) / 16);  // bf16 kernels: dims padded to 16
  if (bf16) {
    const uint64_t ntiles = (k + 63) / 64, nc = ngb / 2;
    VI_TRY(ws.img.reserve(ntiles * ngb * 2 * 64 * 4));  // uint32 words: 2*NG float4-sized pieces of 64 per tile
    VI_TRY(ws.cnpad.reserve(ntiles * 64));
    const uint64_t nt = ntiles * nc * 128;
    hipLaunchKernelGGL(centroid_image_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, st, Cd, (uint32_t)k, d,
                       (uint32_t)nc, (uint4 *)ws.img.p);
    hipLaunchKernelGGL(centroid_norm_pad_kernel, dim3((uint32_t)((ntiles * 64 + 255) / 256)), dim3(256), 0, st, ws.cn.p,
                       (uint32_t)k, (uint32_t)(ntiles * 64), ws.cnpad.p);
    VI_HIP(hipGetLastError());
    a.img = (const float4 *)ws.img.p;
    a.cn = ws.cnpad.p;
  }
  // tier 0: the hi-only candidate sweep (one MFMA per product) + exact evaluation of the listed candidates; what it
  // cannot decide (list overflow, NaN rows) continues through the tiers below.  VI_ASSIGN_CAND=0: start at bf16 x 3.
  const char *cand_env = getenv("VI_ASSIGN_CAND");
  const bool cand_on = !(cand_env && *cand_env == '0');
  const bool use_cand = bf16 && cand_on && k >= 64ull * 8 * kCandWarmTiles;
  CandArgs ca{};
  if (use_cand) {
    const uint64_t ntiles = (k + 63) / 64, nc = ngb / 2;
    // 2 |x.c - hi(x).hi(c)| <= 2 (|x - hi(x)||c| + |hi(x)||c - hi(c)|) <= 2^-6 (1 + 2^-9) |x||c| <= e_tr (|x|^2 + |c|^2)
    const double e_tr = std::ldexp(1.0, -7) * (1.0 + std::ldexp(1.0, -9)) * 1.01;
    const double e_acc = (d + 2.0) * 2.0 * u;  // f32 accumulation of D exact bf16 products + the norm, per (|x|^2 + 2 max|c|^2)
    VI_TRY(ws.img_hi.reserve(ntiles * nc * 2 * 64 * 4));
    const uint64_t nt = ntiles * nc * 128;
    hipLaunchKernelGGL(centroid_hi_image_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, st, Cd, (uint32_t)k, d,
                       (uint32_t)nc, (uint4 *)ws.img_hi.p);
    VI_HIP(hipGetLastError());
    ca.dim = d; ca.img = (const float4 *)ws.img_hi.p; ca.cn = ws.cnpad.p; ca.k = (uint32_t)k; ca.ntiles = (uint32_t)ntiles;
    ca.warm = kCandWarmTiles;
    { const char *xm = getenv("VI_CAND_XMODE"); ca.xmode = xm ? (uint32_t)atoi(xm) : 0u; }
    // margin_i = 2 (E0_i + G_i), E0_i = e_tr (|x_i|^2 + max|c|^2) + e_acc (|x_i|^2 + 2 max|c|^2): see the kernel's header
    ca.margin_scale_x = (float)(2.0 * (e_tr + e_acc + g) * 1.0001);
    ca.margin_const = (float)(2.0 * (e_tr + 2.0 * e_acc + g) * cmax * 1.0001);
  }
  uint64_t total_amb = 0, total_tier1 = 0, total_tier2 = 0;
  float ms_filter = 0.0f;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (stats) { VI_HIP(hipEventCreate(&ev0)); VI_HIP(hipEventCreate(&ev1)); }
  const uint64_t chunk = use_cand ? 1ull << 22 : 1ull << 24;  // points per launch (bounds the ambiguous list and the candidate lists)
  VI_TRY(ws.amb_list.reserve(std::min(chunk, n)));
  if (use_cand) {
    VI_TRY(ws.cand.reserve(std::min(chunk, n) * 2 * kCandCap));
    VI_TRY(ws.cand_cnt.reserve(std::min(chunk, n) * 2));
    VI_TRY(ws.cand_thr.reserve(std::min(chunk, n)));
  }
  for (uint64_t p0 = 0; p0 < n; p0 += chunk) {
    const uint64_t m = std::min(chunk, n - p0);
    VI_HIP(hipMemsetAsync(ws.namb.p, 0, sizeof(uint32_t), st));
    a.X = Xd + p0 * d; a.n = (uint32_t)m; a.label = labels_dev + p0; a.amb_list = ws.amb_list.p;
    if (stats) VI_HIP(hipEventRecord(ev0, st));
    if (use_cand) {
      ca.X = a.X; ca.n = a.n; ca.cand = ws.cand.p; ca.cand_cnt = ws.cand_cnt.p; ca.cand_thr = ws.cand_thr.p;
      ca.pack16 = k <= 65536 ? 1u : 0u;
      switch (ngb / 2) {
        case 1: VI_TRY(launch_cand<1>(ca, st)); break;
        case 2: VI_TRY(launch_cand<2>(ca, st)); break;
        case 3: VI_TRY(launch_cand<3>(ca, st)); break;
        case 4: VI_TRY(launch_cand<4>(ca, st)); break;
        case 5: VI_TRY(launch_cand<5>(ca, st)); break;
        case 6: VI_TRY(launch_cand<6>(ca, st)); break;
        case 7: VI_TRY(launch_cand<7>(ca, st)); break;
        default: VI_TRY(launch_cand<8>(ca, st)); break;
      }
      if (stats) VI_HIP(hipEventRecord(ev1, st));  // (the sweep alone; the candidates' exact distances count in ms_total)
      hipLaunchKernelGGL(cand_exact_kernel, dim3((uint32_t)((m * 8 + 255) / 256)), dim3(256), 0, st, a.X, a.n, d, Cd, (uint32_t)k,
                         ws.cand.p, ws.cand_cnt.p, ws.cand_thr.p, ca.pack16, a.label, ws.amb_list.p, ws.namb.p);
      VI_HIP(hipGetLastError());
    } else if (bf16) {
      switch (ngb) {
        case 2: VI_TRY(launch_mfma_bf16<2>(a, st)); break;
        case 4: VI_TRY(launch_mfma_bf16<4>(a, st)); break;
        case 6: VI_TRY(launch_mfma_bf16<6>(a, st)); break;
        case 8: VI_TRY(launch_mfma_bf16<8>(a, st)); break;
        case 10: VI_TRY(launch_mfma_bf16<10>(a, st)); break;
        case 12: VI_TRY(launch_mfma_bf16<12>(a, st)); break;
        case 14: VI_TRY(launch_mfma_bf16<14>(a, st)); break;
        default: VI_TRY(launch_mfma_bf16<16>(a, st)); break;
      }
    } else {
      VI_TRY(launch_mfma_f32(a, ng, st));
    }
    if (stats && !use_cand) VI_HIP(hipEventRecord(ev1, st));
    uint32_t namb = 0;
    VI_HIP(hipMemcpyAsync(&namb, ws.namb.p, 4, hipMemcpyDeviceToHost, st));
    VI_HIP(hipStreamSynchronize(st));
    if (stats) { float ms = 0; (void)hipEventElapsedTime(&ms, ev0, ev1); ms_filter += ms; }
    if (stats && stats->export_rows && namb && stats->exported < stats->export_cap) {
      const uint32_t m2 = (uint32_t)std::min<uint64_t>(namb, stats->export_cap - stats->exported);
      hipLaunchKernelGGL(export_rows_kernel, dim3((m2 + 255) / 256), dim3(256), 0, st, ws.amb_list.p, m2, (uint32_t)p0,
                         stats->export_rows + stats->exported);
      VI_HIP(hipGetLastError());
      stats->exported += m2;
    }
    total_tier1 += namb;
    if (namb && bf16 && namb >= 2048) {
      // second tier: the bf16 margin is ~5x the f32 one, so most of its ambiguous rows are decided by the f32
      // MFMA kernel on the gathered rows; only what that leaves goes through the exact-order scan
      VI_TRY(ws.xc.reserve((uint64_t)namb * d));
      VI_TRY(ws.lab_c.reserve(namb));
      VI_TRY(ws.amb_list2.reserve(namb));
      VI_TRY(ws.namb2.reserve(1));
      const uint64_t nt = (uint64_t)namb * (d / 4);
      hipLaunchKernelGGL(gather_amb_rows_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, st, Xd + p0 * d,
                         ws.amb_list.p, namb, d, ws.xc.p);
      VI_HIP(hipGetLastError());
      VI_HIP(hipMemsetAsync(ws.namb2.p, 0, sizeof(uint32_t), st));
      MfmaArgs b = a;
      b.X = ws.xc.p; b.n = namb; b.label = ws.lab_c.p; b.amb_list = ws.amb_list2.p; b.namb = ws.namb2.p; b.cn = ws.cn.p;
      b.margin_scale_x = (float)(2.0 * (e32 + g) * 1.0001);
      b.margin_const = (float)(2.0 * (2.0 * e32 + g) * cmax * 1.0001);
      VI_TRY(launch_mfma_f32(b, ng, st));
      uint32_t namb2 = 0;
      VI_HIP(hipMemcpyAsync(&namb2, ws.namb2.p, 4, hipMemcpyDeviceToHost, st));
      VI_HIP(hipStreamSynchronize(st));
      if (namb2) VI_TRY(exact(exact_ctx, ws.xc.p, ws.amb_list2.p, namb2, ws.lab_c.p));
      hipLaunchKernelGGL(scatter_amb_labels_kernel, dim3((namb + 255) / 256), dim3(256), 0, st, ws.amb_list.p, namb,
                         ws.lab_c.p, labels_dev + p0);
      VI_HIP(hipGetLastError());
      total_amb += namb2;
      total_tier2 += namb;
    } else if (namb) {
      // exact-order re-evaluation of the ambiguous rows over ALL centroids
      VI_TRY(exact(exact_ctx, Xd + p0 * d, ws.amb_list.p, namb, labels_dev + p0));
      total_amb += namb;
    }
  }
  if (stats) {
    stats->ambiguous_rows = total_amb;
    stats->tier2_rows = total_tier2;
    stats->tier1_rows = total_tier1;
    stats->ms_filter = ms_filter;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
  }
  return VI_OK;
}

}  // namespace vi
