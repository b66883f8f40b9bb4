// filter_search.hip — list scan and coarse quantizer on the matrix cores: MFMA ranking (bf16 x 3 split
// arithmetic by default, f32 MFMA with VI_FILTER_BF16=0) + exact-order re-evaluation of the few vectors
// that can be results.
//
// The exact-order VALU scan (search_kernels.hip) spends 3 vector ops per (query, vector, dim) and is
// bound by f32 VALU issue.  When many queries of a batch probe the same list, (queries x vectors x
// dims) is GEMM shaped: the matrix cores can RANK the candidates, and only the handful that can
// reach the top-k need the reference's exact arithmetic (src/utils.rs:28-30).
//
//   1. rank      per work item (list segment of <= segb blocks, group of <= 128 queries):
//                m(q,v) = ||v||^2 - 2 q.v on the matrix cores (A = 64 vectors staged in LDS, B = the group's
//                queries in registers, accumulator initialised with ||v||^2): v_mfma_f32_32x32x16_bf16 on
//                operands split hi + lo (hi.hi + hi.lo + lo.hi, mfma_bf16.hpp), or v_mfma_f32_32x32x2_f32.
//                A lane owns one query and 32 of the 64 rows of every block: the 16 accumulator registers of each
//                of the two 32-row tiles, a SUB-BLOCK of 16 rows.  All that is kept of a sub-block is its minimum
//                (8 v_min3_f32): two blocks' four minima form a 16-byte PAIR RECORD, and the four smallest minima
//                of the segment, T0 <= T1 <= T2 <= T3 (a v_med3_f32 network, no positions), a GROUP RECORD — no
//                thresholds, no atomics, no candidate lists, no index bits stolen from the values, nothing that can
//                overflow.  (Round 1 kept the four smallest rows of every 32 with their indices in the low mantissa
//                bits: 226 vector instructions and 16 B per block and lane against 28 and 8 B now.)
//   2. select    one wave per query reads its group records (4 values per 1024 scanned vectors).
//                With m_K the K-th smallest recorded value, every vector of the true top-K has
//                    m <= thr = m_K + 2E + 3 gamma (m_K + ||q||^2 + E)            (*)
//                    gamma = (D+2) u'                      rounding of the reference's sequential sum
//                    E     = e (||q||^2 + 2 max||v||^2), e = rank arithmetic (select_common)
//                because the K recorded minima below m_K belong to K different vectors, which already bound the
//                K-th reference distance.  A vector with m <= thr sits in a sub-block whose minimum is <= thr, and a
//                sub-block with minimum <= thr sits in a group with T0 <= thr: the pair records of exactly those
//                groups are read, and every sub-block whose minimum is <= thr is re-evaluated as a whole — 16
//                reference distances (exact sequential f32, src/utils.rs:28-30), four sub-blocks per wave
//                instruction.  The top-K of those under the reference's stable order (distance, shard visiting
//                order, position) is the answer, bit for bit (tests/test_search_gpu.py).  Groups whose T3 is below
//                the first bound hide neighbours behind their four listed minima; their pair records tighten the
//                bound before anything is re-evaluated.
//
// The coarse quantizer (ivf_index.rs:205-220) is the same computation with the centroid table as one
// list probed by every query and K = n_probe.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_index.hpp"
#include "device_math.hpp"
#include "mfma_bf16.hpp"
#include "rank_stream.hpp"
#include "scan.hpp"
#include "wave_select.hpp"
#include "wave_sort.hpp"

namespace vi {

// provided by search_kernels.hip
vi_status stage_coarse(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st);
vi_status adopt_probes(const DeviceIndex &ix, uint64_t nq, uint32_t P, const uint32_t *probes_in, const uint32_t *order_in,
                       bool histogram, hipStream_t st);
vi_status launch_grouping(const DeviceIndex &ix, const uint32_t *probes, uint64_t nq, uint32_t P, int qg, uint32_t segb0,
                          uint64_t hstats[14], hipStream_t st, bool histogram_done, const uint32_t *qtot = nullptr,
                          uint32_t *qoff = nullptr, const uint32_t *pair_rank = nullptr);
bool grouping_fuses_query_offsets(const DeviceIndex &ix);

namespace {

constexpr int kWave = 64;
constexpr uint32_t kMaxFilterDim = 1536;  // the MFMA engine's dimension limit (rank_wide_kernel above 128)
constexpr int kGroupQ = 128;       // queries per work item: 4 waves x one MFMA column tile of 32
constexpr uint32_t kPosBits = 26;  // candidate key = (probe rank << 26) | position in list
constexpr uint32_t kPosMask = (1u << kPosBits) - 1u;
constexpr double kApproxRatio = 0.04;  // rank_approx_mode: margin unit / list spread up to which the hi planes alone rank
constexpr float kBig = 3.0e38f;            // norm of pad slots inside the kernel (finite: low bits are reused)

// mu (or null): the centre the ranking images are taken about (mean_kernel) — the norm of fl(v - mu) then
__global__ void slot_norms_kernel(const float4 *blocks, uint32_t dq, uint64_t nslots, float *xnorm, uint32_t *xmax_bits,
                                  const float4 *mu = nullptr) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  double acc = 0.0;  // (pad slots hold zeros: they are marked by pad_norms_kernel from the list layout afterwards)
  const float4 *p = blocks + (s / kWave) * dq * kWave + (s % kWave);
  for (uint32_t qd = 0; qd < dq; ++qd) {
    float4 v = p[(size_t)qd * kWave];
    if (mu) { const float4 m = mu[qd]; v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  float out = (float)acc;
  if (out < kBig) atomicMax(xmax_bits, __float_as_uint(out));
  xnorm[s] = fminf(out, kBig);
}

// pad slots (positions len .. 64*ceil(len/64) of every list) never rank: the mask comes from the layout, not from the
// stored ids — the reference accepts ANY u64 as external_id (api.rs:57-62), 2^64-1 included
__global__ void pad_norms_kernel(const uint32_t *first_block, const uint32_t *list_len, uint32_t nlists, float *xnorm) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t l = t >> 6, j = t & 63u;
  if (l >= nlists) return;
  const uint32_t len = list_len[l], p = len + j;
  if (p < ((len + 63u) & ~63u)) xnorm[(size_t)first_block[l] * kWave + p] = kBig;  // (finite: the kernel reuses the low mantissa bits)
}

// component sums of all stored vectors (pad slots hold zeros): WG (quad, g) walks blocks g, g + G, ..., a lane per vector;
// one atomic per component and work-group.  sums: dq * 4 doubles.
__global__ void mean_kernel(const float4 *blocks, uint32_t dq, uint64_t nblocks, double *sums) {
  const uint32_t qd = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (uint64_t b = (uint64_t)blockIdx.y * 4 + wave; b < nblocks; b += (uint64_t)gridDim.y * 4) {
    const float4 v = blocks[(b * dq + qd) * 64 + lane];
    a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); a3 += __shfl_xor(a3, o);
  }
  if (lane == 0) {
    atomicAdd(sums + 4 * qd + 0, a0); atomicAdd(sums + 4 * qd + 1, a1);
    atomicAdd(sums + 4 * qd + 2, a2); atomicAdd(sums + 4 * qd + 3, a3);
  }
}
__global__ void sum_f32_kernel(const float *x, uint64_t n, double *out) {  // finite entries only (pad slots hold kBig)
  double a = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v < 1.0e37f) a += v;
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if ((threadIdx.x & 63u) == 0u) atomicAdd(out, a);
}
__global__ void max_finite_kernel(const float *x, uint64_t n, uint32_t *bits) {  // non-negative entries; pad slots (kBig) skipped
  float m = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v < 1.0e37f) m = fmaxf(m, v);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63u) == 0u) atomicMax(bits, __float_as_uint(m));
}
// max over the stored vectors of |x - hi(x)|^2, x = v - mu and hi = its bf16 image: what ranking from the hi planes alone
// leaves out of q.v is at most |q| times the root of this (pad slots — norm kBig in `norms` — do not count)
__global__ void trunc_residual_kernel(const float4 *blocks, uint32_t dq, uint64_t nslots, const float *norms, const float4 *mu,
                                      uint32_t *max_bits) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float out = 0.0f;
  if (s < nslots && norms[s] < 1.0e37f) {
    double acc = 0.0;
    const float4 *p = blocks + (s / kWave) * dq * kWave + (s % kWave);
    for (uint32_t qd = 0; qd < dq; ++qd) {
      float4 v = p[(size_t)qd * kWave];
      if (mu) { const float4 m = mu[qd]; v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
      const float r0 = v.x - __uint_as_float(bf16_rn(v.x) << 16), r1 = v.y - __uint_as_float(bf16_rn(v.y) << 16);
      const float r2 = v.z - __uint_as_float(bf16_rn(v.z) << 16), r3 = v.w - __uint_as_float(bf16_rn(v.w) << 16);
      acc += (double)r0 * r0 + (double)r1 * r1 + (double)r2 * r2 + (double)r3 * r3;
    }
    out = (float)(acc * 1.000001);
  }
  for (int o = 32; o > 0; o >>= 1) out = fmaxf(out, __shfl_xor(out, o));
  if ((threadIdx.x & 63u) == 0u && out > 0.0f) atomicMax(max_bits, __float_as_uint(out));
}
__global__ void mean_finish_kernel(const double *sums, uint32_t dim, uint32_t dim_pad, double n, float *mu) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < dim_pad) mu[e] = e < dim ? (float)(sums[e] / n) : 0.0f;
}

// sampled spread of the lists: sums of ||v - c(list)||^2 and ||v||^2 over the first blocks of every list (one wave per list)
__global__ void list_spread_kernel(const float4 *blocks, uint32_t dq, const float4 *cent_rows, uint32_t dim, const uint32_t *first_block,
                                   const uint32_t *list_len, uint32_t nlists, uint32_t max_blocks, double *out) {
  const uint32_t l = blockIdx.x, lane = threadIdx.x;
  if (l >= nlists) return;
  const uint32_t len = list_len[l], nb = min((len + 63u) / 64u, max_blocks);
  double s_spread = 0.0, s_norm = 0.0, cnt = 0.0;
  for (uint32_t b = 0; b < nb; ++b) {
    if (b * 64u + lane >= len) continue;
    const float4 *p = blocks + ((size_t)(first_block[l] + b) * dq) * 64 + lane;
    float sp = 0.0f, nn = 0.0f;
    for (uint32_t qd = 0; qd < dim / 4; ++qd) {
      const float4 v = p[(size_t)qd * 64], c = cent_rows[(size_t)l * (dim / 4) + qd];
      sp += (v.x - c.x) * (v.x - c.x) + (v.y - c.y) * (v.y - c.y) + (v.z - c.z) * (v.z - c.z) + (v.w - c.w) * (v.w - c.w);
      nn += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s_spread += sp; s_norm += nn; cnt += 1.0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s_spread += __shfl_xor(s_spread, o); s_norm += __shfl_xor(s_norm, o); cnt += __shfl_xor(cnt, o);
  }
  if (lane == 0 && cnt > 0.0) { atomicAdd(out, s_spread); atomicAdd(out + 1, s_norm); atomicAdd(out + 2, cnt); }
}

// ------------------------------------------------------------------------------------------
// bf16 x 3 ranking: every stored value x is split as hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (|x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-17 |x|); q.v ~ hi.hi + hi.lo + lo.hi on the bf16 matrix pipe (16x the f32 rate)
// ------------------------------------------------------------------------------------------
// f32 blocks [quad][64] float4 -> bf16 blocks [chunk of 16 dims][plane hi/lo][half of 8 dims][64] x 16 B: the
// image a 32x32x16 MFMA wants (lane (j,h) reads the 8 consecutive dims 16c+8h.. of vector j as one ds_read_b128),
// same bytes per block as the f32 form
// Column of vector v (0..63 of its block) in the bf16 image.  A lane of lane half h ends an MFMA holding rows
// (e&3) + 8(e>>2) + 4h (e = 0..15) of each 32-row tile: a SUB-BLOCK, the unit the select re-evaluates exactly.  The image
// places vectors so that sub-block (tile t, half h) is the 16 CONSECUTIVE vectors 32t + 16h .. + 15 of the block: their
// f32 quads are 256 contiguous bytes, two whole cache lines, where the identity placement touched half of four.
__host__ __device__ inline uint32_t image_column(uint32_t v) {
  const uint32_t t = v >> 5, h = (v >> 4) & 1u, e = v & 15u;
  return 32u * t + (e & 3u) + 8u * (e >> 2) + 4u * h;
}

__global__ void split_bf16_kernel(const float4 *blocks, uint32_t dq, uint64_t nblocks, uint4 *out, const float4 *mu = nullptr) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (block, chunk, half, vector)
  const uint32_t nc = dq / 4;
  if (t >= nblocks * nc * 2 * 64) return;
  const uint32_t v = (uint32_t)(t & 63), h = (uint32_t)((t >> 6) & 1);
  const uint64_t bc = t >> 7;
  const uint32_t c = (uint32_t)(bc % nc);
  const uint64_t b = bc / nc;
  const float4 *src = blocks + (b * dq + 4 * c + 2 * h) * 64 + v;
  uint4 hi, lo;
  float4 v0 = src[0], v1 = src[64];
  if (mu) {  // the image of fl(v - mu)
    const float4 m0 = mu[4 * c + 2 * h], m1 = mu[4 * c + 2 * h + 1];
    v0.x -= m0.x; v0.y -= m0.y; v0.z -= m0.z; v0.w -= m0.w;
    v1.x -= m1.x; v1.y -= m1.y; v1.z -= m1.z; v1.w -= m1.w;
  }
  split8(v0, v1, 1.0f, hi, lo);
  uint4 *dst = out + ((b * nc + c) * 4) * 64;
  const uint32_t col = image_column(v);
  dst[(0 * 2 + h) * 64 + col] = hi;
  dst[(1 * 2 + h) * 64 + col] = lo;
}

// the squared norms in image order (the accumulator of MFMA row i starts at the norm of the vector in column i)
__global__ void image_norms_kernel(const float *xnorm, uint64_t nslots, float *out) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nslots) out[(s & ~63ull) + image_column((uint32_t)(s & 63u))] = xnorm[s];
}

// bf16-exact stored values: the hi plane IS the value.  A second copy of it in the blocks' own vector order — piece
// (chunk c, half h) of vector v at (block * 2 nc + 2c + h) * 64 + v — lets the select re-evaluate a 16-vector sub-block
// from 256 contiguous bytes per 8 dimensions: half the cache lines of the f32 quads (the MFMA image's column order
// interleaves the two sub-blocks of a tile within every line).
__global__ void hi_natural_kernel(const uint4 *img, uint32_t nc, uint64_t nblocks, uint4 *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (block, chunk, half, vector)
  if (t >= nblocks * nc * 2 * 64) return;
  const uint32_t v = (uint32_t)(t & 63), h = (uint32_t)((t >> 6) & 1);
  const uint64_t bc = t >> 7;
  const uint32_t c = (uint32_t)(bc % nc);
  const uint64_t b = bc / nc;
  out[((b * nc + c) * 2 + h) * 64 + v] = img[(((b * nc + c) * 4) + h) * 64 + image_column(v)];
}

// The coarse table once more, row-major (centroid c = dim consecutive floats): the coarse select re-evaluates ONE
// centroid per candidate sub-block, and a lone row of a lane-interleaved block is 16 bytes in each of dim / 4 cache
// lines; from this copy it is dim / 32 whole lines.  (k' x dim floats; the lists stay lane-interleaved only.)
__global__ void rows_from_blocks_kernel(const float4 *blocks, uint32_t dq, uint32_t nrows, uint32_t nquad, float4 *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)nrows * nquad) return;
  const uint32_t row = (uint32_t)(t / nquad), qd = (uint32_t)(t % nquad);
  out[t] = blocks[((size_t)(row / 64) * dq + qd) * 64 + (row % 64)];
}

// 8-bit descriptors (every stored value an integer in 0..255, as SIFT's): one byte per dimension, 16 dimensions of vector
// v at (block * ceil(dq / 4) + p) * 64 + v — a 16-vector sub-block is re-evaluated from 256 contiguous bytes per 16
// dimensions, half of the bf16 copy again.  `not_u8` is raised if some value does not fit.
__global__ void u8_check_kernel(const float4 *blocks, uint64_t nquads, uint32_t *not_u8) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (t < nquads) {
    const float4 v = blocks[t];
    auto ok = [](float x) { return x >= 0.0f && x <= 255.0f && x == floorf(x); };
    bad = !(ok(v.x) && ok(v.y) && ok(v.z) && ok(v.w));
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(not_u8, 1u);
}
__global__ void u8_natural_kernel(const float4 *blocks, uint32_t dq, uint64_t nblocks, uint4 *out) {
  const uint32_t np = (dq + 3) / 4;  // pieces of 16 dimensions (dq is a multiple of 4)
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (block, piece, vector)
  if (t >= nblocks * np * 64) return;
  const uint32_t v = (uint32_t)(t & 63);
  const uint64_t bp = t >> 6;
  const uint32_t p = (uint32_t)(bp % np);
  const uint64_t b = bp / np;
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 x = blocks[(b * dq + 4 * p + i) * 64 + v];
    w[i] = (uint32_t)x.x | ((uint32_t)x.y << 8) | ((uint32_t)x.z << 16) | ((uint32_t)x.w << 24);
  }
  out[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

// any nonzero lo half in an image? (pieces of 64 uint4: plane = (piece >> 1) & 1)
__global__ void lo_plane_any_kernel(const uint4 *img, uint64_t npieces, uint32_t *any) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npieces * 64) return;
  const uint64_t piece = t >> 6;
  if (((piece >> 1) & 1) == 0) return;
  const uint4 v = img[t];
  if (((v.x | v.y | v.z | v.w) & 0x7FFF7FFFu) != 0u) atomicOr(any, 1u);
}

// ------------------------------------------------------------------------------------------
// record bookkeeping: where the records of (query, probe) start
// ------------------------------------------------------------------------------------------
// rel[q*P+r] = group records of the query's probes before rank r ; qtot[q] = group records of the query
__global__ void pair_groups_kernel(const uint32_t *probes, const uint32_t *list_len, uint32_t nq, uint32_t P,
                                   uint32_t segb0, uint32_t *rel, uint32_t *qtot) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t run = 0;
  for (uint32_t r = 0; r < P; ++r) {
    const uint32_t l = probes[(size_t)q * P + r];
    rel[(size_t)q * P + r] = run;
    if (l != kNoPos) {
      uint32_t sb;
      run += 2u * list_segments(list_len[l], segb0, &sb);
    }
  }
  qtot[q] = run;
}

// qoff = exclusive scan of qtot over the queries (one workgroup), qoff[nq] = total
__global__ void __launch_bounds__(1024) query_offsets_kernel(const uint32_t *qtot, uint32_t nq, uint32_t *qoff) {
  __shared__ uint32_t s[16];
  const uint32_t t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const uint32_t per = (nq + 1023) / 1024;
  const uint32_t beg = min(nq, t * per), end = min(nq, beg + per);
  uint32_t sum = 0;
  for (uint32_t i = beg; i < end; ++i) sum += qtot[i];
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= o) inc += x;
  }
  if (lane == 63) s[wave] = inc;
  __syncthreads();
  uint32_t w = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    if (i < wave) w += s[i];
    tot += s[i];
  }
  uint32_t run = w + inc - sum;
  for (uint32_t i = beg; i < end; ++i) { qoff[i] = run; run += qtot[i]; }
  if (t == 0) qoff[nq] = tot;
}

// list of every work item: keeps a 12-step dependent binary search out of each rank workgroup's prologue
__global__ void item_list_kernel(const uint32_t *item_start, uint32_t nlists, uint32_t nitems, uint32_t *item_list) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= nitems) return;
  uint32_t lo = 0, hi = nlists;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (item_start[mid] <= item) lo = mid; else hi = mid;
  }
  item_list[item] = lo;
}

// everything a list-rank work item needs to know about itself, 32 bytes it reads with two wave-uniform loads instead
// of a chain of five dependent ones (list -> offsets -> length -> ...) at the head of every workgroup:
// {first pair, queries, first block of the list, b0, b1, segment, first record tile, -}
// Workgroup -> item: the hardware deals workgroups to the 8 XCDs round-robin (workgroup w runs on XCD w % 8), and each
// XCD has its own L2.  The query groups of one list segment stream the SAME blocks, so they are numbered next to each
// other (segment-major) and dealt in runs of `run` items to one XCD: workgroup w = (cycle, r, x) -> item
// cycle * 8 run + x * run + r.  They start together, the followers hit the L2 lines the first one brought in, and a
// hit is faster than a miss, which keeps them together.  (run <= 1: workgroup w takes item w.)
__global__ void item_desc_kernel(const uint32_t *item_start, const uint32_t *seg_start, const uint32_t *list_len,
                                 const uint32_t *first_block, const uint32_t *tile_start, uint32_t nlists, uint32_t nitems,
                                 uint32_t segb0, uint32_t gq, uint32_t run, uint4 *items) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nitems) return;
  uint32_t item = w;
  if (run > 1u) {
    const uint32_t span = 8u * run, cycle = w / span;
    if ((cycle + 1u) * span <= nitems) {  // (the last, partial cycle keeps its order)
      const uint32_t in = w - cycle * span;
      item = cycle * span + (in & 7u) * run + (in >> 3);
    }
  }
  uint32_t lo = 0, hi = nlists;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (item_start[mid] <= item) lo = mid; else hi = mid;
  }
  const uint32_t l = lo;
  const uint32_t s0 = seg_start[l], cnt = seg_start[l + 1] - s0, len = list_len[l];
  uint32_t segb;
  const uint32_t nseg = list_segments(len, segb0, &segb);
  const uint32_t local = item - item_start[l];
  const uint32_t nchunk = (cnt + gq - 1u) / gq;  // query groups of the list: its items are nchunk * nseg
  const uint32_t seg = local / nchunk, chunk = local - seg * nchunk;
  const uint32_t j0 = chunk * gq;
  const uint32_t nblk = (len + 63u) / 64u, b0 = seg * segb, b1 = min(nblk, b0 + segb);
  items[2 * (size_t)w] = make_uint4(s0 + j0, min(gq, cnt - j0), first_block[l], b0);
  items[2 * (size_t)w + 1] = make_uint4(b1, seg, tile_start[l] + (chunk * nseg + seg) * seg_records(segb), 0u);
}

// Per (work item, column of its query group): the query, and where its group record goes — so that the rank
// workgroup finds everything about an item at addresses it can compute from the item's index alone (no chain
// descriptor -> pairs -> query offsets at the head of every item) — and the place word of the pair's two group records
// (probe rank | segment << 6 | lane half << 13; every (pair, segment) sits in exactly one item).  One workgroup per
// item; workgroup 0 also resets the rank kernel's work counter and the "a query has a lo plane" flag of the next batch.
__global__ void item_cols_kernel(const uint4 *items, const uint32_t *pairs, const uint32_t *qoff, const uint32_t *rel, uint32_t P,
                                 uint32_t gq, uint32_t *qcol, uint32_t *grec, uint4 *sdesc, uint32_t *gmeta, uint64_t *stats) {
  const uint32_t w = blockIdx.x;
  const uint4 d0 = items[2 * (size_t)w], d1 = items[2 * (size_t)w + 1];
  if (threadIdx.x == 0) {
    sdesc[w] = make_uint4(d0.y, d0.z + d0.w, 2u * (d1.x - d0.w), d1.z);  // queries, first block, tiles, first record tile
    if (w == 0) stats[13] = 0;
  }
  if (w == 0 && threadIdx.x < 8) stats[16 + 16 * threadIdx.x] = 0;  // the rank kernel's work counters (one per XCD queue, 128 bytes apart)
  for (uint32_t col = threadIdx.x; col < gq; col += blockDim.x) {
    uint32_t q = ~0u, g = ~0u;
    if (col < d0.y) {
      const uint32_t slot = pairs[d0.x + col];
      q = div_probes(slot, P);
      g = qoff[q] + rel[slot] + 2u * d1.y;
      const uint32_t place = (slot - q * P) | (d1.y << 6);
      *reinterpret_cast<uint2 *>(gmeta + g) = make_uint2(place, place | (1u << 13));  // (g is even: 8-byte aligned)
    }
    qcol[(size_t)w * gq + col] = q;
    grec[(size_t)w * gq + col] = g;
  }
}

__global__ void iota_kernel(uint32_t *p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// ------------------------------------------------------------------------------------------
// rank kernel
// ------------------------------------------------------------------------------------------
struct FilterArgs {
  const float4 *blocks;  // f32 blocks, or the bf16 hi/lo image of the same size (BF16 kernels)
  const float *xnorm;
  uint32_t dq, dim;
  const float *Q;
  const uint32_t *first_block, *list_len, *item_start, *seg_start, *pairs;
  uint32_t nlists, P, segb0;
  const uint32_t *qoff, *rel;  // group-record offsets (lists) ...
  uint32_t rec_stride;         // ... or a fixed number of records per slot when qoff is null (coarse table)
  const uint32_t *tile_start;  // pair records: first record tile (2 * GQ records) of each list
  const uint32_t *item_list;   // list of each work item (null: binary search over item_start)
  const uint4 *items;          // list phase: the work items' descriptors (item_desc_kernel), or null: derived here
  float4 *gval;                // group records: the four smallest sub-block minima of a (pair, segment, lane half)
  uint32_t *gmeta;             // ... and where the record belongs: probe rank | segment << 6 | lane half << 13
  float4 *brec;                // pair records: the sub-block minima of two blocks
  const uint4 *qimg;  // -2 q of the batch split hi / lo once per search (split_queries_kernel), or null: split here
  uint32_t direct;  // coarse table only: records = the minima of the 8-row sub-blocks of every block, no group records
  uint32_t xmode;  // experiment knob (VI_FILTER_XMODE): 1 = do not restage tiles, 2 = no ranking epilogue
};

// Workgroup = 4 waves = up to 128 queries (4 column tiles of 32) probing ONE list segment.  Every
// 64-vector block of the segment is copied once per workgroup into LDS by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write) and consumed by all four waves.  The LDS image is the block verbatim —
// [quad][vector] float4, so a lane's A fragment (quad 2g+h of vector j) is a conflict-free ds_read_b128 —
// followed by the block's 64 squared norms.  With NBUF = 2 the next block lands in the other buffer while
// this one is multiplied (one barrier per block); with NBUF = 1 the load is exposed and hidden by the other
// workgroups of the CU (3 per CU instead of 2).
// registers r0 .. r0+7 of an accumulator tile (an 8-row sub-block): the smallest value with its row in the 3 low
// mantissa bits (|packed - m| < 2^-20 |m|) and the second smallest — 3 instructions per element
__device__ __forceinline__ float2 tile_min8_idx(const f32x16 &a, int r0) {
  float b1 = INFINITY, b2 = INFINITY;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float p = __uint_as_float((__float_as_uint(a[r0 + e]) & ~7u) | (uint32_t)e);
    b2 = __builtin_amdgcn_fmed3f(b1, b2, p);
    b1 = min3_raw(b1, p, p);
  }
  return make_float2(b1, b2);
}
// one block image -> LDS: all of it, or (RANK 2) only its hi planes: pieces (chunk c, plane 0, half h) = 4c + h
template <int NG, int RANK, int NBUF, int WAVES>
__device__ __forceinline__ void tile_dma_rank(float *tile, const float4 *src, const float *xn, int wave, int lane) {
  if constexpr (RANK != 2) {
    if constexpr (NBUF >= 2) tile_dma_image_asm<NG, WAVES>(tile, src, xn, wave, lane);  // the loop places its own waits
    else tile_dma_image<NG, WAVES>(tile, src, xn, wave, lane);
  } else {
#pragma unroll
    for (int i0 = 0; i0 < NG; i0 += WAVES) {
      const int i = i0 + wave;
      if (i < NG) {
        const int piece = 4 * (i >> 1) + (i & 1);  // hi piece i = 2c + h lands compactly at i
        glds16_asm(src + piece * 64 + lane, tile + i * 256);
      }
    }
    if (wave == 0) glds4_asm(xn + lane, tile + NG * 256);  // (asm: the double-buffered loop places its own waits)
  }
}

__global__ void split_queries_kernel(const float *Q, uint32_t nq, uint32_t dim, uint32_t nc, uint4 *out, unsigned long long *any_lo,
                                     uint32_t *zero, uint32_t zero_words, const float *mu);

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(uint32_t n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // (never asked for more than 7; a stricter wait is always safe)
  }
}

// NG = dq/2 exactly: a block holds 2*NG quads (dims padded to 16); dim % 4 == 0.  TABLE only names the instance
// that ranks the centroid table (coarse step), so that profiles tell it from the list scan.
// RANK 0: f32 MFMA (eight per 16 dims).  RANK 1: three bf16 MFMAs per 16 dims (hi.hi + hi.lo + lo.hi).
// RANK 2: the stored values are bf16-exact (every lo plane is zero: 8-bit descriptors such as SIFT) — only the hi
// planes are staged (half the bytes) and lo.hi is dropped; hi.lo is dropped too for a wave whose 32 queries are
// bf16-exact (wave-uniform test), which leaves ONE MFMA per 16 dims with the result still exact to f32 rounding.
// GQ = queries per work item: 128 (4 waves) when lists are probed by many queries of the batch, 32 (one wave per
// workgroup, up to 9 workgroups per CU) when a list is probed by a handful — large balanced indexes, where a
// 128-query group would leave three of its four waves idle behind the same tile stream.
template <int NG, int NBUF, bool TABLE, int RANK, int GQ>
__global__ void __launch_bounds__(GQ * 2, GQ == 32 ? (RANK == 2 ? 2 : 1) : ((NBUF == 1 || RANK == 2) ? 3 : 2))
    filter_kernel(FilterArgs a) {  // (second bound = waves per SIMD the register allocation must allow)
  constexpr int WAVES = GQ / 32;
  constexpr int kImage = (RANK == 2 ? NG : 2 * NG) * 256;  // floats of the staged image (RANK 2: hi planes only)
  constexpr int kTileFloats = kImage + 64;                 // + the block's 64 norms
  __shared__ __attribute__((aligned(16))) float s_tiles[NBUF][kTileFloats];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t item = blockIdx.x;  // grid == number of items
  uint32_t pair0, nqi, fb, b0, b1, seg, rec0, chunk = 0, nblk = 0;
  if (!TABLE && a.items) {
    const uint4 d0 = a.items[2 * (size_t)item], d1 = a.items[2 * (size_t)item + 1];
    pair0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0.x); nqi = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0.y);
    fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0.z); b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0.w);
    b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d1.x); seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)d1.y);
    rec0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d1.z);
  } else {
    uint32_t lo = 0;
    if (a.item_list) {
      lo = a.item_list[item];
    } else {
      uint32_t hi = a.nlists;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.item_start[mid] <= item) lo = mid; else hi = mid;
      }
    }
    const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    const uint32_t s0 = a.seg_start[l], cnt = a.seg_start[l + 1] - s0;
    const uint32_t len = a.list_len[l];
    uint32_t segb;
    const uint32_t nseg = list_segments(len, a.segb0, &segb);
    const uint32_t local = item - a.item_start[l];
    chunk = local / nseg;
    seg = local - chunk * nseg;
    const uint32_t j0 = chunk * GQ;
    pair0 = s0 + j0;
    nqi = min((uint32_t)GQ, cnt - j0);
    fb = a.first_block[l];
    nblk = (len + kWave - 1) / kWave;
    b0 = seg * segb;
    b1 = min(nblk, b0 + segb);
    // tile_start counts the record tiles of the lists before this one: chunks x segments x seg_records
    rec0 = a.tile_start[l] + (chunk * nseg + seg) * seg_records(segb);
  }

  // the first tile is requested before the queries are fetched: both latencies run together
  tile_dma_rank<NG, RANK, NBUF, WAVES>(s_tiles[0], a.blocks + ((size_t)(fb + b0) * a.dq) * kWave, a.xnorm + (size_t)(fb + b0) * kWave, wave, lane);

  // ---- this lane's query: both lane halves hold query j of the wave's tile ----
  // the tile a wave owns rotates with the item so that partially filled groups do not always idle the same SIMD
  const uint32_t wtile = ((uint32_t)wave + item) % (uint32_t)WAVES;
  const uint32_t jq_grp = 32u * wtile + (uint32_t)j;
  const bool qlive = jq_grp < nqi;
  const bool wave_live = 32u * wtile < nqi;  // wave-uniform
  const uint32_t slot = qlive ? a.pairs[pair0 + jq_grp] : 0u;
  const uint32_t qid = slot / a.P;
  const float *qrow = a.Q + (size_t)qid * a.dim;
  float4 qf[NG];  // f32: -2q, dims 8g+4h.. ; BF16: qf[2c] = hi, qf[2c+1] = lo halves (bit patterns) of -2q, dims 16c+8h..
  constexpr bool BF16 = RANK != 0;
  bool q_lo_zero = false;  // RANK 2: every query of this wave splits with lo = 0
  if (!BF16) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t e = 8 * g + 4 * h;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qlive && e < a.dim) v = *reinterpret_cast<const float4 *>(qrow + e);
      qf[g] = make_float4(-2.f * v.x, -2.f * v.y, -2.f * v.z, -2.f * v.w);
    }
  } else if (a.qimg) {
    // the batch's queries were split once (every query sits in n_probe work items): 16-byte pieces, no arithmetic here
    const uint4 *qi = a.qimg + (size_t)qid * (NG / 2) * 4 + h;  // [plane][chunk][half]
#pragma unroll
    for (int c = 0; c < NG / 2; ++c) {
      uint4 hi = make_uint4(0u, 0u, 0u, 0u), lo = hi;
      if (qlive) { hi = qi[c * 2]; lo = qi[NG + c * 2]; }
      qf[2 * c] = __builtin_bit_cast(float4, hi);
      qf[2 * c + 1] = __builtin_bit_cast(float4, lo);
    }
  } else {
#pragma unroll
    for (int c = 0; c < NG / 2; ++c) {
      const uint32_t e = 16 * c + 8 * h;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (qlive && e < a.dim) v0 = *reinterpret_cast<const float4 *>(qrow + e);
      if (qlive && e + 4 < a.dim) v1 = *reinterpret_cast<const float4 *>(qrow + e + 4);
      uint4 hi, lo;
      split8(v0, v1, -2.0f, hi, lo);
      qf[2 * c] = __builtin_bit_cast(float4, hi);
      qf[2 * c + 1] = __builtin_bit_cast(float4, lo);
    }
  }
  if constexpr (RANK == 2) {
    uint32_t any = 0;
#pragma unroll
    for (int c = 0; c < NG / 2; ++c) {
      const uint4 lo = __builtin_bit_cast(uint4, qf[2 * c + 1]);
      any |= lo.x | lo.y | lo.z | lo.w;
    }
    q_lo_zero = __ballot((any & 0x7FFF7FFFu) != 0u) == 0ull;  // (-0 halves are zero too)
  }

  float T0 = INFINITY, T1 = INFINITY, T2 = INFINITY, T3 = INFINITY;  // four smallest sub-block minima of the segment
  float4 w = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);    // the pair record being filled
  // pair records are item-major — [record tile = (query group, segment, pair of blocks)][lane half][query of the
  // group] — so that a wave's 32 queries store 512 contiguous bytes (record counts are checked < 2^32 on the host);
  const uint32_t bi = rec0 * (2u * GQ) + (uint32_t)GQ * (uint32_t)h + jq_grp;

  // LDS-DMA instructions this wave issues per tile (RANK 2: its share of the NG hi pieces; wave 0 also the norms)
  const uint32_t dma_ops = (RANK == 2 ? ((uint32_t)wave < (uint32_t)NG ? ((uint32_t)NG - (uint32_t)wave + WAVES - 1) / WAVES : 0u)
                                      : (uint32_t)(2 * NG / WAVES)) + (wave == 0 ? 1u : 0u);
  if (NBUF == 3 && b0 + 1 < b1 && !(a.xmode & 1u)) {  // ring of three: two tiles ahead
    tile_dma_rank<NG, RANK, NBUF, WAVES>(s_tiles[1], a.blocks + ((size_t)(fb + b0 + 1) * a.dq) * kWave, a.xnorm + (size_t)(fb + b0 + 1) * kWave, wave, lane);
    wait_vmcnt(dma_ops);  // the first tile has landed, the second may still be on its way
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed ...
  }
  __syncthreads();                     // ... and so have everyone else's
  uint32_t ring = 0;                   // buffer of the current block (NBUF == 3)
  for (uint32_t blk = b0; blk < b1; ++blk) {
    const bool more = (blk + 1 < b1) && !(a.xmode & 1u);
    uint32_t nstores = 0;  // record stores this lane issued in this iteration
    const float *s_tile = s_tiles[NBUF == 3 ? ring : (NBUF == 2 ? ((blk - b0) & 1u) : 0)];
    // next block: lands in the other buffer during this block's MFMAs (every wave left that buffer at the
    // barrier that ended the previous iteration); with three buffers the block after the next is requested here, so
    // that two tiles per workgroup are in flight (an experiment, VI_FILTER_NBUF=3: it measured equal)
    uint32_t issued = 0;  // LDS-DMA instructions of this iteration (younger than the tile the block's end waits for)
    if (NBUF == 2 && more)
      tile_dma_rank<NG, RANK, NBUF, WAVES>(s_tiles[((blk - b0) & 1u) ^ 1u], a.blocks + ((size_t)(fb + blk + 1) * a.dq) * kWave,
                   a.xnorm + (size_t)(fb + blk + 1) * kWave, wave, lane);
    if (NBUF == 3 && blk + 2 < b1 && !(a.xmode & 1u)) {
      tile_dma_rank<NG, RANK, NBUF, WAVES>(s_tiles[ring >= 1 ? ring - 1 : 2], a.blocks + ((size_t)(fb + blk + 2) * a.dq) * kWave,
                   a.xnorm + (size_t)(fb + blk + 2) * kWave, wave, lane);
      issued = dma_ops;
    }
    if (wave_live) {
      // both row tiles (vectors 0..31 and 32..63) advance together: two independent accumulator chains
      f32x16 acc0, acc1;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {  // rows 8*q4 + 4*h + (0..3) live in regs 4*q4 .. 4*q4+3
        const float4 n0 = *reinterpret_cast<const float4 *>(s_tile + kImage + 8 * q4 + 4 * h);
        const float4 n1 = *reinterpret_cast<const float4 *>(s_tile + kImage + 32 + 8 * q4 + 4 * h);
        acc0[4 * q4 + 0] = n0.x; acc0[4 * q4 + 1] = n0.y; acc0[4 * q4 + 2] = n0.z; acc0[4 * q4 + 3] = n0.w;
        acc1[4 * q4 + 0] = n1.x; acc1[4 * q4 + 1] = n1.y; acc1[4 * q4 + 2] = n1.z; acc1[4 * q4 + 3] = n1.w;
      }
      if constexpr (!BF16) {
        // A fragments are read one K group ahead of the MFMAs that consume them (the compiler would otherwise
        // issue each pair of ds_read_b128 right before its 8 MFMAs and stall on the LDS latency every time)
        float4 a0 = *reinterpret_cast<const float4 *>(s_tile + h * 256 + 4 * j);
        float4 a1 = *reinterpret_cast<const float4 *>(s_tile + h * 256 + 4 * (32 + j));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float4 n0 = a0, n1 = a1;
          if (g + 1 < NG) {
            n0 = *reinterpret_cast<const float4 *>(s_tile + (2 * (g + 1) + h) * 256 + 4 * j);
            n1 = *reinterpret_cast<const float4 *>(s_tile + (2 * (g + 1) + h) * 256 + 4 * (32 + j));
          }
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, qf[g].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, qf[g].x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, qf[g].y, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, qf[g].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, qf[g].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, qf[g].z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, qf[g].w, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, qf[g].w, acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          a0 = n0;
          a1 = n1;
        }
      } else {
        // image: [chunk][plane][half][vector] x 16 B; fragment (plane p, tile t) of chunk c for lane (j,h) =
        // float4 index ((c*2 + p)*2 + h)*64 + 32t + j
        auto frag = [&](int c, int p, int t) {  // RANK 2 stages the hi pieces compactly: piece 2c + h
          const int piece = RANK == 2 ? 2 * c + h : (c * 2 + p) * 2 + h;
          return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(s_tile + (piece * 64 + 32 * t + j) * 4));
        };
        if constexpr (RANK == 1) {
          // hi fragments of chunk c+1 are requested while the lo MFMAs of chunk c run, lo fragments of chunk c
          // while its hi MFMAs run: 16 fragment registers, every read has MFMAs to hide behind
          bf16x8 h0 = frag(0, 0, 0), h1 = frag(0, 0, 1);
#pragma unroll
          for (int c = 0; c < NG / 2; ++c) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, qf[2 * c]), bl = __builtin_bit_cast(bf16x8, qf[2 * c + 1]);
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
        } else {
          // stored values are bf16-exact: hi planes only
          bf16x8 h0 = frag(0, 0, 0), h1 = frag(0, 0, 1);
#pragma unroll
          for (int c = 0; c < NG / 2; ++c) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, qf[2 * c]), bl = __builtin_bit_cast(bf16x8, qf[2 * c + 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bh, acc1, 0, 0, 0);
            if (!q_lo_zero) {  // wave-uniform
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bl, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bl, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < NG / 2) { h0 = frag(c + 1, 0, 0); h1 = frag(c + 1, 0, 1); }  // requested behind the MFMAs
          }
        }
      }
      if (TABLE && a.direct) {
        // the coarse table (<= 256 blocks): per block and lane the four 8-row sub-blocks' (minimum with its row, second
        // minimum) go out as they are — two 16-byte records; the select reads all of a query's records at once
        // (coarse_select_direct_kernel) and re-evaluates ONE row per candidate sub-block unless the second minimum
        // can matter too
        nstores = qlive ? 2u : 0u;
        if (qlive) {
          const float2 s00 = tile_min8_idx(acc0, 0), s01 = tile_min8_idx(acc0, 8), s10 = tile_min8_idx(acc1, 0), s11 = tile_min8_idx(acc1, 8);
          if (a.direct == 2u) {
            // query-major: the select reads a query's records as 1 KB runs (these stores pay for it: 32 queries x 32 bytes each)
            float4 *dst = a.brec + ((size_t)chunk * GQ + jq_grp) * (4u * nblk) + 4u * blk + (uint32_t)h;
            dst[0] = make_float4(s00.x, s00.y, s01.x, s01.y);
            dst[2] = make_float4(s10.x, s10.y, s11.x, s11.y);
          } else {
            float4 *dst = a.brec + ((size_t)chunk * nblk + blk) * (4u * GQ) + (uint32_t)GQ * (uint32_t)h + jq_grp;
            dst[0] = make_float4(s00.x, s00.y, s01.x, s01.y);
            dst[2u * GQ] = make_float4(s10.x, s10.y, s11.x, s11.y);
          }
        }
      } else if (!(a.xmode & 2u)) {
        // all that is kept of the two 16-row sub-blocks: their minima
        const float m0 = tile_min(acc0), m1 = tile_min(acc1);
        VI_TOP4(m0) VI_TOP4(m1)
        if (((blk - b0) & 1u) == 0u) { w.x = m0; w.y = m1; w.z = INFINITY; w.w = INFINITY; }
        else { w.z = m0; w.w = m1; }
        if (((blk - b0) & 1u) != 0u || blk + 1 == b1) {  // the pair is complete (wave-uniform)
          nstores = (qlive && !(a.xmode & 8u)) ? 1u : 0u;
          if (nstores) a.brec[(size_t)bi + (2u * GQ) * ((blk - b0) >> 1)] = w;
        }
      }
    }
    if (NBUF == 1) {
      __syncthreads();  // every wave is done reading the tile
      if (more)
        tile_dma_rank<NG, RANK, NBUF, WAVES>(s_tiles[0], a.blocks + ((size_t)(fb + blk + 1) * a.dq) * kWave,
                     a.xnorm + (size_t)(fb + blk + 1) * kWave, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // next tile visible
    } else {
      // The next tile's LDS-DMA was issued before this block's MFMAs; the only younger vector-memory operation
      // are this wave's record stores of this block, if it made any (vmcnt counts loads, stores and LDS-DMA together, in issue order).
      // Waiting for all but that store keeps the store's latency off the critical path; __syncthreads() would
      // insert vmcnt(0), hence the raw barrier (LDS reads of this tile are complete: lgkmcnt(0)).
      // (a store instruction is issued iff some lane of the wave stores: the ballots are the wave-uniform form of that)
      const uint32_t st_ops = __ballot(nstores == 2u) != 0ull ? 2u : (__ballot(nstores == 1u) != 0ull ? 1u : 0u);
      if (NBUF == 3) {
        wait_vmcnt(issued + st_ops);  // everything but this iteration's DMA and stores: the NEXT block's tile has landed
        ring = ring == 2 ? 0 : ring + 1;
      } else if (st_ops == 2u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (st_ops == 1u) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // next tile visible; this tile free to be overwritten
    }
  }
  if (qlive && !(TABLE && a.direct)) {
    const size_t gi = (a.qoff ? (size_t)a.qoff[qid] + a.rel[slot] : (size_t)slot * a.rec_stride) + 2u * seg + (uint32_t)h;
    a.gval[gi] = make_float4(T0, T1, T2, T3);
    a.gmeta[gi] = (slot - qid * a.P) | (seg << 6) | ((uint32_t)h << 13);  // where the record belongs
  }
}

// ------------------------------------------------------------------------------------------
// rank kernel for wide vectors (128 < D <= 1536)
// ------------------------------------------------------------------------------------------
// Above 128 dimensions the queries of a group no longer fit in registers for a whole work item, so the ranking becomes
// a GEMM proper: per workgroup a C tile of 4 blocks (256 vectors) x 128 queries stays in the accumulators (128 VGPRs
// per wave) while BOTH operands stream through LDS in K steps of 32 dimensions — 32 KB of the blocks' bf16 hi/lo image
// and 16 KB of the queries' (gathered by LDS-DMA with per-lane row addresses from a query-major image of the batch,
// split once per search), double buffered: 97 KB of LDS, one workgroup per CU, 24 MFMAs per wave and 16-dim chunk.
// What a C tile leaves behind is what filter_kernel leaves: sub-block minima in pair records and the segment's four
// smallest in a group record, so the select is the same kernel.
constexpr int kWideBlocks = 4;
constexpr int kWideChunks = 2;
constexpr int kWideBufFloats = (kWideBlocks + 2) * kWideChunks * 4 * 256;  // A: 4 blocks, B: 2 query blocks; x chunks x 4 pieces x 1 KB
constexpr int kWideLdsFloats = 2 * kWideBufFloats + kWideBlocks * 64;      // two buffers + the norms of the C tile's blocks (97 KB)

struct WideArgs {
  const uint4 *img;      // bf16 hi/lo image of the lists: per block nc chunks x [plane][half] x 64 columns x 16 B
  const float *xnorm;    // squared norms in image-column order
  const uint4 *qimg;     // query-major image of the batch: [query][nc][plane][half] x 16 B of -2 q split hi / lo
  uint32_t nc;           // 16-dim chunks per vector
  const uint32_t *first_block, *list_len, *item_start, *seg_start, *pairs, *item_list;
  uint32_t P, segb0;
  const uint32_t *qoff, *rel, *tile_start;
  float4 *gval;
  uint32_t *gmeta;
  float4 *brec;
};

// -2 q split hi / lo, query-major, hi plane first: piece (plane p, chunk c, half h) of query q at q * 4 nc + p * 2 nc + 2 c + h
// (a batch of bf16-exact queries is ranked from its hi planes alone: they are whole cache lines of their own)
// *any_lo is raised when some query has a non-zero lo plane (a batch of bf16-exact queries is ranked without them)
// (`zero`: a buffer the next kernels count into — the coarse step's per-list histogram — cleared here instead of by a memset launch)
// mu (or null): the centre of the ranking images (DeviceIndex::centre) — the image is then that of -2 fl(q - mu)
__global__ void split_queries_kernel(const float *Q, uint32_t nq, uint32_t dim, uint32_t nc, uint4 *out, unsigned long long *any_lo,
                                     uint32_t *zero, uint32_t zero_words, const float *mu) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (query, chunk, half)
  for (uint64_t i = t; i < zero_words; i += (uint64_t)gridDim.x * blockDim.x) zero[i] = 0u;
  if (t >= (uint64_t)nq * nc * 2) return;
  const uint32_t h = (uint32_t)(t & 1u);
  const uint64_t qc = t >> 1;
  const uint32_t c = (uint32_t)(qc % nc);
  const uint64_t q = qc / nc;
  const uint32_t e = 16 * c + 8 * h;
  const float *row = Q + q * dim;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (e < dim) v0 = *reinterpret_cast<const float4 *>(row + e);        // dim % 4 == 0
  if (e + 4 < dim) v1 = *reinterpret_cast<const float4 *>(row + e + 4);
  if (mu) {
    if (e < dim) { const float4 m = *reinterpret_cast<const float4 *>(mu + e); v0.x -= m.x; v0.y -= m.y; v0.z -= m.z; v0.w -= m.w; }
    if (e + 4 < dim) { const float4 m = *reinterpret_cast<const float4 *>(mu + e + 4); v1.x -= m.x; v1.y -= m.y; v1.z -= m.z; v1.w -= m.w; }
  }
  uint4 hi, lo;
  split8(v0, v1, -2.0f, hi, lo);
  out[q * 4 * nc + 2 * c + h] = hi;
  out[q * 4 * nc + 2 * nc + 2 * c + h] = lo;
  if (__ballot(((lo.x | lo.y | lo.z | lo.w) & 0x7FFF7FFFu) != 0u) != 0ull && (threadIdx.x & 63u) == 0u)  // (-0 halves are zero too)
    atomicOr(any_lo, 1ull);
}

__global__ void __launch_bounds__(256, 1) rank_wide_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_wide[];
  float *s_norm = s_wide + 2 * kWideBufFloats;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t item = blockIdx.x;
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.item_list[item]);
  const uint32_t s0 = a.seg_start[l], cnt = a.seg_start[l + 1] - s0;
  const uint32_t len = a.list_len[l];
  uint32_t segb;
  const uint32_t nseg = list_segments(len, a.segb0, &segb);
  const uint32_t local = item - a.item_start[l];
  const uint32_t chunk = local / nseg, seg = local - chunk * nseg;
  const uint32_t j0 = chunk * kGroupQ;
  const uint32_t nqi = min((uint32_t)kGroupQ, cnt - j0);
  const uint32_t fb = a.first_block[l];
  const uint32_t nblk = (len + kWave - 1) / kWave;
  const uint32_t b0 = seg * segb, b1 = min(nblk, b0 + segb);
  const uint32_t nc = a.nc, nslab = (nc + kWideChunks - 1) / kWideChunks;

  // this lane's query (both lane halves hold query j of the wave's tile), and the two queries it gathers for the B tiles
  const uint32_t wtile = ((uint32_t)wave + item) & 3u;
  const uint32_t jq_grp = 32u * wtile + (uint32_t)j;
  const bool qlive = jq_grp < nqi;
  const bool wave_live = 32u * wtile < nqi;
  const uint32_t slot = qlive ? a.pairs[s0 + j0 + jq_grp] : 0u;
  const uint32_t qid = slot / a.P;
  uint32_t gq[2];  // query of column `lane` of query block 0 / 1
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const uint32_t col = 64u * qb + (uint32_t)lane;
    gq[qb] = col < nqi ? a.pairs[s0 + j0 + col] / a.P : 0u;
  }

  float T0 = INFINITY, T1 = INFINITY, T2 = INFINITY, T3 = INFINITY;
  const uint32_t bi = (a.tile_start[l] + (chunk * nseg + seg) * seg_records(segb)) * (2u * kGroupQ) + (uint32_t)kGroupQ * (uint32_t)h + jq_grp;

  // one K step (slab s of the C tile at blk0) into buffer `buf`: 32 A pieces + 16 B pieces of 1 KB, 12 per wave
  auto stage = [&](uint32_t blk0, uint32_t s, float *buf) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int pi = wave + 4 * i;
      if (pi < 32) {
        const uint32_t blk = (uint32_t)pi >> 3, c = ((uint32_t)pi >> 2) & 1u, piece = (uint32_t)pi & 3u;
        const uint32_t cc = s * kWideChunks + c;
        if (blk0 + blk < b1 && cc < nc)
          glds16_asm(a.img + (((size_t)(fb + blk0 + blk) * nc + cc) * 4 + piece) * kWave + lane, buf + pi * 256);
      } else {
        const uint32_t r = (uint32_t)pi - 32u, qb = r >> 3, c = (r >> 2) & 1u, piece = r & 3u;
        const uint32_t cc = s * kWideChunks + c;
        if (cc < nc) glds16_asm(a.qimg + (size_t)gq[qb] * nc * 4 + (piece >> 1) * 2 * nc + cc * 2 + (piece & 1u), buf + pi * 256);
      }
    }
  };

  for (uint32_t blk0 = b0; blk0 < b1; blk0 += kWideBlocks) {
    const uint32_t nb = min((uint32_t)kWideBlocks, b1 - blk0);  // blocks of this C tile
    stage(blk0, 0, s_wide);
    if ((uint32_t)wave < nb) glds4_asm(a.xnorm + (size_t)(fb + blk0 + wave) * kWave + lane, s_norm + wave * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc[kWideBlocks][2];
#pragma unroll
    for (int b = 0; b < kWideBlocks; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {  // rows 8*q4 + 4*h + (0..3) of tile t live in regs 4*q4 .. 4*q4+3
          const float4 n = *reinterpret_cast<const float4 *>(s_norm + b * 64 + 32 * t + 8 * q4 + 4 * h);
          acc[b][t][4 * q4 + 0] = n.x; acc[b][t][4 * q4 + 1] = n.y; acc[b][t][4 * q4 + 2] = n.z; acc[b][t][4 * q4 + 3] = n.w;
        }
    for (uint32_t s = 0; s < nslab; ++s) {
      const float *buf = s_wide + (s & 1u) * kWideBufFloats;
      if (s + 1 < nslab) stage(blk0, s + 1, s_wide + ((s + 1u) & 1u) * kWideBufFloats);
      if (wave_live) {
        const float *bq = buf + (32 + 8 * (int)(wtile >> 1)) * 256;  // this wave's query block
        const int qcol = 32 * (int)(wtile & 1u) + j;
#pragma unroll
        for (int c = 0; c < kWideChunks; ++c) {
          if (s * kWideChunks + c < nc) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(bq + ((c * 4 + 0 + h) * 64 + qcol) * 4));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(bq + ((c * 4 + 2 + h) * 64 + qcol) * 4));
#pragma unroll
            for (int b = 0; b < kWideBlocks; ++b) {
              if ((uint32_t)b < nb) {
                const float *ab = buf + (b * 8 + c * 4) * 256;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(ab + ((0 + h) * 64 + 32 * t + j) * 4));
                  const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(ab + ((2 + h) * 64 + 32 * t + j) * 4));
                  acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b][t], 0, 0, 0);
                  acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b][t], 0, 0, 0);
                  acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b][t], 0, 0, 0);
                }
              }
            }
          }
        }
      }
      // the next step's tiles have landed (nothing but the LDS-DMA is in flight), every wave is done with this buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (wave_live) {
      float m[kWideBlocks][2];
#pragma unroll
      for (int b = 0; b < kWideBlocks; ++b) {
        m[b][0] = INFINITY; m[b][1] = INFINITY;
        if ((uint32_t)b < nb) {
          m[b][0] = tile_min(acc[b][0]);
          m[b][1] = tile_min(acc[b][1]);
          VI_TOP4(m[b][0]) VI_TOP4(m[b][1])
        }
      }
      if (qlive) {
        const uint32_t p0 = (blk0 - b0) >> 1;
        a.brec[(size_t)bi + (2u * kGroupQ) * p0] = make_float4(m[0][0], m[0][1], m[1][0], m[1][1]);
        if (nb > 2) a.brec[(size_t)bi + (2u * kGroupQ) * (p0 + 1u)] = make_float4(m[2][0], m[2][1], m[3][0], m[3][1]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record stores, before the next C tile counts its DMA
    __syncthreads();                                    // s_norm and both buffers are free again
  }
  if (qlive) {
    const size_t gi = (size_t)a.qoff[qid] + a.rel[slot] + 2u * seg + (uint32_t)h;
    a.gval[gi] = make_float4(T0, T1, T2, T3);
    a.gmeta[gi] = (slot - qid * a.P) | (seg << 6) | ((uint32_t)h << 13);
  }
}

// ------------------------------------------------------------------------------------------
// select
// ------------------------------------------------------------------------------------------
struct SelectCommon {
  const float *Q;
  uint32_t dim, dq;
  const float4 *blocks;
  const float4 *gval;
  const uint32_t *gmeta;
  const float4 *brec;
  float gamma, e_scale, xmax2;
  uint32_t gq;  // queries per rank work item (a record tile holds 2 * gq pair records)
  unsigned long long *dbg;  // [6] exact re-evaluations, [7] groups whose pair records were read, [8..] see select_body
  uint32_t image_order;     // the rank kernel multiplied the permuted bf16 image (subblock_vector)
  const uint4 *hi_nat;      // bf16-exact lists: natural-order hi plane for the exact re-evaluation (hi_natural_kernel), else null
  const uint4 *u8_nat;      // 8-bit descriptors: one byte per dimension (u8_natural_kernel), else null
  uint32_t wave_order;      // pair records in the streaming kernel's wave order (scan.hpp: seg_records), else pair order
  uint32_t xmode;           // ablation knob (VI_SELECT_XMODE, wrong results): 1 no exact evaluation, 2 no stage 2, 4 no stage 1b
  uint32_t dbg_mask;        // counters of the queries with (q & dbg_mask) == 0 only (VI_FILTER_STATS=4: every 64th — ten thousand
                            // waves adding to the same few addresses are most of the kernel's time, which the stage clocks then measure)
  const float *mu;          // centre of the ranking images (rank values are those of q - mu against v - mu), or null
  uint32_t trunc;           // real-valued lists ranked from their hi planes: 1 queries hi + lo, 2 queries' hi plane only; 0 otherwise
  float rho_max, vmax;      // ... max |v - hi(v)| and max |v| over the lists (rounded up)
};

// the query's probes, one per lane r < P
struct ProbeRegs {
  uint32_t rel, ng;   // first group record (relative to the query's) / number of group records of the probe
  uint32_t boff;      // pair record of (segment 0, pair 0, lane half 0) of the probe; + 2*gq per pair
                      // (pair p of segment s = s * seg_records(segb) + p), + gq for half 1
  uint32_t len, fb;   // list length and first block
  uint32_t segb;      // blocks per segment
  uint32_t g;         // candidate-order rank (shard visiting order)
};

// Loads with the address space spelled out.  The exact-evaluation pieces below are real functions (noinline), so their
// pointer arguments are generic and every access through them compiles to flat_load: the query row in LDS then goes through
// the vector-memory address pipe — the unit the gathers of stored vectors saturate — and every wait covers both counters.
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
// a - b on two floats in one instruction (v_pk_add_f32 with the second operand negated: every component rounds as
// v_sub_f32 does; the compiler turns a vector subtraction back into two scalar ones)
__device__ __forceinline__ vf2 pk_sub_f32(vf2 a, vf2 b) {
  vf2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// acc += (q - x)^2 over four consecutive dimensions in the reference's order: differences and squares two at a time
// (packed instructions: each component rounds exactly as the scalar instruction does), the sum one term after the other
__device__ __forceinline__ void sq_add4(float &acc, const float4 &q, const float4 &x) {
  const vf2 qa = {q.x, q.y}, qb = {q.z, q.w}, xa = {x.x, x.y}, xb = {x.z, x.w};
  const vf2 ta = pk_sub_f32(qa, xa), tb = pk_sub_f32(qb, xb);
  const vf2 sa = ta * ta, sb = tb * tb;
  acc = acc + sa.x; acc = acc + sa.y; acc = acc + sb.x; acc = acc + sb.y;
}
#define VI_AS_LDS __attribute__((address_space(3)))
#define VI_AS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 lds_f4(const float *p) {
  const vf4 v = *(const VI_AS_LDS vf4 *)(const VI_AS_LDS float *)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 lds_u4(const uint32_t *p) {
  const vu4 v = *(const VI_AS_LDS vu4 *)(const VI_AS_LDS uint32_t *)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 glb_f4(const float4 *p) {
  const vf4 v = *(const VI_AS_GLOBAL vf4 *)(const VI_AS_GLOBAL float *)reinterpret_cast<const float *>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 glb_u4(const uint4 *p) {
  const vu4 v = *(const VI_AS_GLOBAL vu4 *)(const VI_AS_GLOBAL uint32_t *)reinterpret_cast<const uint32_t *>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// exact distance of one (query row, stored vector) pair, one lane per pair (src/utils.rs:28-30).  The query row sits
// in LDS (every lane reads the same address: a broadcast, no vector-memory slot), so all eight loads in flight per
// lane are the stored vector's
__device__ __forceinline__ float exact_pair(const float *qrow, const float4 *xv, uint32_t dim) {
  float acc = 0.0f;
  const uint32_t nquad = dim >> 2;
  uint32_t qd = 0;
  for (; qd + 8 <= nquad; qd += 8) {
    float4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = glb_f4(xv + (size_t)(qd + i) * kWave);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 qq = lds_f4(qrow + 4 * (qd + i));
      sq_add4(acc, qq, x[i]);
    }
  }
  for (; qd < nquad; ++qd) {
    const float4 qq = lds_f4(qrow + 4 * qd);
    const float4 xx = glb_f4(xv + (size_t)qd * kWave);
    sq_add4(acc, qq, xx);
  }
  return acc;
}

// the same sum for a vector stored as dim consecutive floats (coarse table, rows_from_blocks_kernel)
__device__ __forceinline__ float exact_pair_row(const float *qrow, const float4 *xr, uint32_t dim) {
  float acc = 0.0f;
  const uint32_t nquad = dim >> 2;
  uint32_t qd = 0;
  for (; qd + 8 <= nquad; qd += 8) {
    float4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = glb_f4(xr + qd + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 qq = lds_f4(qrow + 4 * (qd + i));
      sq_add4(acc, qq, x[i]);
    }
  }
  for (; qd < nquad; ++qd) {
    const float4 qq = lds_f4(qrow + 4 * qd);
    const float4 xx = glb_f4(xr + qd);
    sq_add4(acc, qq, xx);
  }
  return acc;
}

// the same sum from the natural-order bf16 hi plane of bf16-exact vectors (hi_natural_kernel): x = bf16 << 16 exactly,
// so every term and the sequential order are those of exact_pair; 16 bytes carry 8 dimensions
__device__ __forceinline__ float exact_pair_bf16(const float *qrow, const uint4 *xh, uint32_t dim) {
  float acc = 0.0f;
  const uint32_t npiece = (dim + 7u) >> 3;  // (dim % 4 == 0: the last piece may hold 4 dimensions)
  auto piece = [&](const uint4 &x, uint32_t p) {
    const float4 q0 = lds_f4(qrow + 8 * p);
    sq_add4(acc, q0, make_float4(__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xFFFF0000u), __uint_as_float(x.y << 16),
                                 __uint_as_float(x.y & 0xFFFF0000u)));
    if (8 * p + 4 < dim) {
      const float4 q1 = lds_f4(qrow + 8 * p + 4);
      sq_add4(acc, q1, make_float4(__uint_as_float(x.z << 16), __uint_as_float(x.z & 0xFFFF0000u), __uint_as_float(x.w << 16),
                                   __uint_as_float(x.w & 0xFFFF0000u)));
    }
  };
  uint32_t p = 0;
  for (; p + 8 <= npiece; p += 8) {
    uint4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = glb_u4(xh + (size_t)(p + i) * kWave);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(x[i], p + i);
  }
  for (; p + 4 <= npiece; p += 4) {  // (a half round: D = 96 has 6 / 12 pieces)
    uint4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = glb_u4(xh + (size_t)(p + i) * kWave);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece(x[i], p + i);
  }
  for (; p < npiece; ++p) piece(glb_u4(xh + (size_t)p * kWave), p);
  return acc;
}

// ... and from one byte per dimension (u8_natural_kernel): x = (float)byte exactly
__device__ __forceinline__ float exact_pair_u8(const float *qrow, const uint4 *xb, uint32_t dim) {
  float acc = 0.0f;
  const uint32_t npiece = (dim + 15u) >> 4;  // (dim % 4 == 0: the last piece may hold 4, 8 or 12 dimensions)
  auto word = [&](uint32_t w, uint32_t e) {   // 4 dimensions starting at e
    const float4 q = lds_f4(qrow + e);
    sq_add4(acc, q, make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24)));
  };
  auto piece = [&](const uint4 &x, uint32_t p) {
    const uint32_t e = 16 * p;
    word(x.x, e);
    if (e + 4 < dim) word(x.y, e + 4);
    if (e + 8 < dim) word(x.z, e + 8);
    if (e + 12 < dim) word(x.w, e + 12);
  };
  uint32_t p = 0;
  for (; p + 8 <= npiece; p += 8) {
    uint4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = glb_u4(xb + (size_t)(p + i) * kWave);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(x[i], p + i);
  }
  for (; p + 4 <= npiece; p += 4) {  // (a half round: D = 96 has 6 / 12 pieces)
    uint4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = glb_u4(xb + (size_t)(p + i) * kWave);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece(x[i], p + i);
  }
  for (; p < npiece; ++p) piece(glb_u4(xb + (size_t)p * kWave), p);
  return acc;
}

// 8-bit descriptors AND an integer-valued query in 0..255 (SIFT queries are): every term (q - x)^2 of the reference's
// sum (src/utils.rs:28-30) is an integer <= 255^2 and every partial sum an integer <= D 255^2 < 2^24 (D <= 256), so the
// sequential f32 sum never rounds: its value IS the integer sum, whatever the order.  It is formed with byte dot
// products: |q|^2 + |x|^2 - 2 q.x, four dimensions per v_dot4_u32_u8 — 90 instructions per distance instead of 512.
// qb: the query as bytes (LDS, D / 4 words, zero padded to whole 16-byte pieces); qn = |q|^2.
__device__ __forceinline__ float exact_pair_u8_int(const uint32_t *qb, uint32_t qn, const uint4 *xb, uint32_t dim) {
  const uint32_t npiece = (dim + 15u) >> 4;
  uint32_t dot = 0u, xx = 0u;
  auto piece = [&](const uint4 &x, uint32_t p) {
    const uint4 q = lds_u4(qb + 4 * p);
    dot = __builtin_amdgcn_udot4(q.x, x.x, dot, false); xx = __builtin_amdgcn_udot4(x.x, x.x, xx, false);
    dot = __builtin_amdgcn_udot4(q.y, x.y, dot, false); xx = __builtin_amdgcn_udot4(x.y, x.y, xx, false);
    dot = __builtin_amdgcn_udot4(q.z, x.z, dot, false); xx = __builtin_amdgcn_udot4(x.z, x.z, xx, false);
    dot = __builtin_amdgcn_udot4(q.w, x.w, dot, false); xx = __builtin_amdgcn_udot4(x.w, x.w, xx, false);
  };
  uint32_t p = 0;
  for (; p + 8 <= npiece; p += 8) {
    uint4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = glb_u4(xb + (size_t)(p + i) * kWave);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(x[i], p + i);
  }
  for (; p < npiece; ++p) piece(glb_u4(xb + (size_t)p * kWave), p);
  return (float)(qn + xx - 2u * dot);  // (an integer below 2^24: exact)
}

// The select kernels are latency-sensitive code executed once per query; inlining the two heavy pieces at
// every call site made them ~150 KB each and instruction-fetch bound.  They are real functions with their state
// passed and returned in registers.
template <class Top>
__device__ __attribute__((noinline)) Top offer_bulk_fn(Top s, float dist, uint32_t pos, int K) {
  s.offer_bulk(dist, pos, K);
  return s;
}

// exact distance of (qrow, xv) on live lanes, then offer (distance, key) to `sel`
template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_fn(Top sel, const float *qrow, const float4 *xv, uint32_t dim, bool live,
                                                        uint32_t key, int K) {
  float d = INFINITY;
  if (live) d = exact_pair(qrow, xv, dim);
  sel.offer_bulk(d, live ? key : kNoPos, K);
  return sel;
}

template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_u8_fn(Top sel, const float *qrow, const uint4 *xb, uint32_t dim, bool live, uint32_t key,
                                                           int K) {
  float d = INFINITY;
  if (live) d = exact_pair_u8(qrow, xb, dim);
  sel.offer_bulk(d, live ? key : kNoPos, K);
  return sel;
}

template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_u8_int_fn(Top sel, const uint32_t *qb, uint32_t qn, const uint4 *xb, uint32_t dim, bool live,
                                                               uint32_t key, int K) {
  float d = INFINITY;
  if (live) d = exact_pair_u8_int(qb, qn, xb, dim);
  sel.offer_bulk(d, live ? key : kNoPos, K);
  return sel;
}

template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_row_fn(Top sel, const float *qrow, const float4 *xr, uint32_t dim, bool live,
                                                            uint32_t key, int K) {
  float d = INFINITY;
  if (live) d = exact_pair_row(qrow, xr, dim);
  sel.offer_bulk(d, live ? key : kNoPos, K);
  return sel;
}

// Up to 64 rows of a row-major table (one per lane, `cnt` of them live) against the query, with the ROWS FETCHED BY THE
// WHOLE WAVE: a lane reading its own row asks the texture unit for 64 different cache lines per load instruction (one
// 16-byte piece of each) — 2 048 line requests for 64 rows of 128 floats, and the address pipe, not the arithmetic, set
// the pace of the coarse select.  Here four lanes fetch 64 consecutive bytes of a row (16 rows per instruction, 512
// requests in all) straight into LDS (LDS-DMA: no registers in between, two chunks of 16 dimensions in flight), and
// every lane then runs the reference's chain (utils.rs:28-30) over its own row as before.  A DMA instruction writes
// lane l's 16 bytes at LDS offset 16 l, so a row's four pieces sit 64 bytes apart from the next row's — a lane per row
// reading piece p would hit the same banks eight times over; lane l therefore fetches piece (l & 3) ^ ((l >> 3) & 3)
// and row r reads its piece p from slot 4 r + (p ^ ((r >> 1) & 3)): conflict free.
// dim % 16 == 0; stage: kStageFloats floats of LDS per wave (a ring of two chunks); the waits are counted by hand
// (the copies are inline asm, invisible to the compiler's own counting: mfma_bf16.hpp).
constexpr uint32_t kStageFloats = 2048;
template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_rows_staged_fn(Top sel, const float *qrow, const float4 *rows, uint32_t nrows, uint32_t dim,
                                                                    uint32_t cnt, uint32_t pos, int K, float *stage) {
  const uint32_t lane = threadIdx.x & 63u, r0 = lane >> 2, piece = (lane & 3u) ^ ((lane >> 3) & 3u);
  const uint32_t nquad = dim >> 2, nch = nquad >> 2;
  const VI_AS_LDS vf4 *xq = (const VI_AS_LDS vf4 *)(const VI_AS_LDS float *)qrow;
  const uint32_t sbase = (uint32_t)(size_t)(VI_AS_LDS float *)stage;
  const float4 *src[4];  // piece `piece` of the rows r0 + 16 j (rows beyond cnt: the last live one again)
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j)
    src[j] = rows + (size_t)min((uint32_t)__shfl((int)pos, (int)min(r0 + 16u * j, cnt - 1u)), nrows - 1u) * nquad + piece;
  auto issue = [&](uint32_t c) {  // chunk c -> ring slot c & 1: four copies of 1 KiB
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) glds16_at(src[j] + 4u * c, sbase + (c & 1u) * 4096u + j * 1024u);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (whatever the caller left in flight: the counts below are this function's)
  issue(0u);
  if (nch > 1u) issue(1u);
  const uint32_t swz = (lane >> 1) & 3u;
  float acc = 0.0f;
#pragma unroll 1
  for (uint32_t c = 0; c < nch; ++c) {
    if (c + 1u < nch) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const VI_AS_LDS vf4 *rd = (const VI_AS_LDS vf4 *)((const VI_AS_LDS float *)stage + (c & 1u) * 1024u + lane * 16u);
    const vf4 x0 = rd[0u ^ swz], x1 = rd[1u ^ swz], x2 = rd[2u ^ swz], x3 = rd[3u ^ swz];
    const vf4 q0 = xq[4u * c], q1 = xq[4u * c + 1u], q2 = xq[4u * c + 2u], q3 = xq[4u * c + 3u];
    // (differences and squares four at a time — packed f32 instructions round every component as the scalar ones do;
    //  the sum stays the reference's sequential chain)
#define VI_PK_QUAD(QQ, XX)                                                         \
  {                                                                                \
    const vf2 ta = pk_sub_f32(QQ.xy, XX.xy), tb = pk_sub_f32(QQ.zw, XX.zw);        \
    const vf2 sa = ta * ta, sb = tb * tb;                                          \
    acc = acc + sa.x; acc = acc + sa.y; acc = acc + sb.x; acc = acc + sb.y;        \
  }
    VI_PK_QUAD(q0, x0) VI_PK_QUAD(q1, x1) VI_PK_QUAD(q2, x2) VI_PK_QUAD(q3, x3)
#undef VI_PK_QUAD
    if (c + 2u < nch) {  // the slot's values are in registers (the chain above consumed them): refill it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(c + 2u);
    }
  }
  const bool live = lane < cnt && pos < nrows;
  sel.offer_bulk(live ? acc : INFINITY, live ? pos : kNoPos, K);
  return sel;
}

template <class Top>
__device__ __attribute__((noinline)) Top exact_batch_bf16_fn(Top sel, const float *qrow, const uint4 *xh, uint32_t dim, bool live,
                                                             uint32_t key, int K) {
  float d = INFINITY;
  if (live) d = exact_pair_bf16(qrow, xh, dim);
  sel.offer_bulk(d, live ? key : kNoPos, K);
  return sel;
}

// vector (within its 64-vector block) of row e (0..15) of sub-block (tile t, lane half hh): the bf16 images are built
// so that it is 32t + 16hh + e (image_column); the f32 MFMA (VI_FILTER_BF16=0) multiplies the f32 blocks as they
// are, where the 16 registers of a lane of half hh hold rows (e&3) + 8(e>>2) + 4hh of the tile
__device__ __forceinline__ uint32_t subblock_vector(uint32_t e, uint32_t t, uint32_t hh, bool image_order) {
  return image_order ? 32u * t + 16u * hh + e : 32u * t + (e & 3u) + 8u * (e >> 2) + 4u * hh;
}

constexpr uint32_t kNarrowDim = 128;     // up to here the queries of a work item stay in registers (filter_kernel)
constexpr uint32_t kPickCap = 256;     // sub-blocks waiting for their 16 exact distances (per wave)
constexpr uint32_t kSubBits = 21;      // request key = (probe rank << 22) | (sub-block of the list << 1) | lane half
constexpr uint32_t kCacheG = 256;      // group records (values + probe/segment/half) kept in LDS per wave

// One wave: top-K of query q under (exact distance, (g << 26) | position) from its G group records at gbase.
// Leaves the result in `sel` (entry e of lane i = result 64e + i, key kNoPos when there are fewer than K).
// Top = FastTopK (K <= 64) or FastTop128 (K <= 128: the Faiss-style harness asks for 100 neighbours), wave_sort.hpp.
template <class Top>
__device__ __forceinline__ void select_body(const SelectCommon &c, uint32_t q, size_t gbase, uint32_t G, uint32_t P,
                                            const ProbeRegs &pr, uint32_t K, int lane, uint32_t *pick, float4 *tcache,
                                            uint32_t *lcache, float *qlds, Top &sel, uint32_t *qbytes = nullptr) {
  const uint64_t below = (1ull << lane) - 1ull;
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float qn = 0.0f, qres = 0.0f;
  bool q_bytes = c.u8_nat != nullptr && qbytes != nullptr && c.dim <= 256u;  // -> the query is integer-valued in 0..255
  for (uint32_t e = lane; e < c.dim; e += kWave) {  // the query row: into LDS for the exact evaluations, and its norm
    const float v = c.Q[(size_t)q * c.dim + e];
    qlds[e] = v;
    const float vc = c.mu ? v - c.mu[e] : v;  // the margins live where the rank values do
    qn += vc * vc;
    const float im = -2.0f * vc, ir = im - __uint_as_float(bf16_rn(im) << 16);  // what the hi plane of the query image leaves out
    qres += ir * ir;
    q_bytes = q_bytes && v >= 0.0f && v <= 255.0f && v == floorf(v);
  }
  q_bytes = __ballot(!q_bytes) == 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { qn += __shfl_xor(qn, o); qres += __shfl_xor(qres, o); }
  uint32_t qn_int = 0u;
  if (q_bytes) {  // the query as bytes, zero padded to whole 16-byte pieces (exact_pair_u8_int), and |q|^2 as an integer
    const uint32_t nw = ((c.dim + 15u) >> 4) * 4u;
    for (uint32_t w = lane; w < nw; w += kWave) {
      uint32_t word = 0u;
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t e = 4u * w + b;
        const uint32_t v = e < c.dim ? (uint32_t)c.Q[(size_t)q * c.dim + e] : 0u;
        word |= v << (8u * b);
        qn_int += v * v;
      }
      qbytes[w] = word;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn_int += (uint32_t)__shfl_xor((int)qn_int, o);
  }
  float E = c.e_scale * (qn * (1.0f + c.gamma) + 2.0f * c.xmax2);
  // hi planes of real-valued lists: the image (-2q) . v is ranked as (-2q) . hi(v) [trunc 1] or hi(-2q) . hi(v) [trunc 2];
  // |(-2q) . (v - hi v)| <= 2 |q| rho_max, and |(-2q - hi(-2q)) . v| <= |query residual| max|v|  (|hi(-2q)| <= 2 |q| (1 + 2^-8))
  if (c.trunc) E += 1.02f * (2.0f * sqrtf(qn) * c.rho_max * (1.0f + 0.00391f) + (c.trunc == 2u ? sqrtf(qres) * c.vmax : 0.0f));
  // a query so large that the rank arithmetic may have overflowed (inf - inf = NaN, and NaN fails every guard
  // below): trust no rank value, re-evaluate everything the query probes
  const bool distrust = !(qn < 1.0e30f);
  auto threshold_of = [&](float mk) {  // (*) ; anything non-finite or huge means "no bound"
    if (distrust || !(mk < 1.0e37f)) return INFINITY;
    const float scale = fmaxf(mk + qn, 0.0f) + E;
    return mk + (2.0f * E + 3.0f * c.gamma * scale) * 1.001f + 1e-30f;
  };
  uint32_t npick = 0, n_exact = 0, n_scanned = 0, n_full = 0, n_sub = 0;
  Top s1;
  float thr = INFINITY;
  sel.init();
  const float *qrow = qlds;  // (visible to the wave after the lds_sync of stage 0)
  auto exact_offer = [&](bool live, uint32_t r, uint32_t pos) {  // one (probe rank, position) per lane
    const uint32_t fb = (uint32_t)__shfl((int)pr.fb, (int)r);
    const uint32_t g = (uint32_t)__shfl((int)pr.g, (int)r);
    const uint32_t len = (uint32_t)__shfl((int)pr.len, (int)r);
    live = live && pos < len && !(c.xmode & 1u);
    n_exact += (uint32_t)__popcll(__ballot(live));
    if (c.u8_nat && q_bytes)
      sel = exact_batch_u8_int_fn(sel, qbytes, qn_int, c.u8_nat + ((size_t)(fb + (live ? pos : 0u) / kWave) * (c.dq / 4)) * kWave + (pos % kWave),
                                  c.dim, live, (g << kPosBits) | pos, (int)K);
    else if (c.u8_nat)
      sel = exact_batch_u8_fn(sel, qrow, c.u8_nat + ((size_t)(fb + (live ? pos : 0u) / kWave) * (c.dq / 4)) * kWave + (pos % kWave), c.dim,
                              live, (g << kPosBits) | pos, (int)K);
    else if (c.hi_nat)
      sel = exact_batch_bf16_fn(sel, qrow, c.hi_nat + ((size_t)(fb + (live ? pos : 0u) / kWave) * (c.dq / 2)) * kWave + (pos % kWave),
                                c.dim, live, (g << kPosBits) | pos, (int)K);
    else
      sel = exact_batch_fn(sel, qrow, c.blocks + ((size_t)(fb + (live ? pos : 0u) / kWave) * c.dq) * kWave + (pos % kWave),
                           c.dim, live, (g << kPosBits) | pos, (int)K);
  };
  // sub-blocks waiting in `pick`: four per round, 16 lanes (= the 16 rows of the sub-block) each
  auto drain_pick = [&]() {
    while (npick > 0) {
      const uint32_t cnt = npick >= 4u ? 4u : npick;
      npick -= cnt;
      const uint32_t rq = (uint32_t)lane >> 4;
      const bool live = rq < cnt;
      const uint32_t ck = live ? pick[npick + rq] : 0u;
      const uint32_t r = ck >> (kSubBits + 1), sub = (ck >> 1) & ((1u << kSubBits) - 1u), hh = ck & 1u;
      exact_offer(live, r, (sub >> 1) * kWave + subblock_vector((uint32_t)lane & 15u, sub & 1u, hh, c.image_order != 0u));
    }
  };
  auto push_sub = [&](bool want, uint32_t r, uint32_t sub, uint32_t hh) {  // every lane calls
    const uint64_t m = __ballot(want);
    if (!m) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    n_sub += cnt;
    if (npick + cnt > kPickCap) drain_pick();
    if (want) pick[npick + (uint32_t)__popcll(m & below)] = (r << (kSubBits + 1)) | (sub << 1) | hh;
    npick += cnt;
    lds_sync();
  };
  // The pair records of the groups flagged `want`, four groups per round (16 lanes x one pair record = the 64
  // sub-block minima of a 32-block segment half).  mode 0: the minima go to the running top-K of rank values (s1);
  // mode 1: every sub-block whose minimum is at or below thr is queued for exact evaluation.
  auto scan_groups = [&](bool want, uint32_t r, uint32_t seg, uint32_t hh, int mode) {
    uint64_t m = __ballot(want);
    const uint32_t slot = (uint32_t)lane >> 4, pi = (uint32_t)lane & 15u;
    while (m) {
      int src = 0;
      uint32_t taken = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i)
        if (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1ull;
          if (slot == i) src = b;
          ++taken;
        }
      const bool mine = slot < taken;
      n_scanned += taken;
      const uint32_t rr = (uint32_t)__shfl((int)r, src), sg = (uint32_t)__shfl((int)seg, src);
      const uint32_t h2 = (uint32_t)__shfl((int)hh, src);
      const uint32_t segb = (uint32_t)__shfl((int)pr.segb, (int)rr), ln = (uint32_t)__shfl((int)pr.len, (int)rr);
      const uint32_t boff = (uint32_t)__shfl((int)pr.boff, (int)rr);
      const uint32_t nblk = (ln + kWave - 1) / kWave;
      const uint32_t bs = sg * segb, be = min(nblk, bs + segb);
      // records of the segment: pair order (record p = blocks 2p, 2p + 1; component j = sub-block 4p + j), or the streaming
      // kernel's wave order (record p, component j = 32-vector tile 16 (p >> 2) + 4 j + (p & 3))
      const uint32_t ntile = be > bs ? 2u * (be - bs) : 0u;
      const uint32_t npairs = !mine ? 0u : (c.wave_order ? 4u * ((ntile + 15u) / 16u) : (ntile + 3u) / 4u);
      uint32_t mx = npairs;  // wave maximum: segments of very long lists hold more than 16 records
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
      for (uint32_t p0 = 0; p0 < mx; p0 += 16u) {
        const uint32_t p = p0 + pi;
        const bool live = p < npairs;
        float4 B = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        if (live) B = c.brec[(size_t)boff + 2u * c.gq * (sg * seg_records(segb) + p) + c.gq * h2];
        const float bv[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t tl = c.wave_order ? 16u * (p >> 2) + 4u * j + (p & 3u) : 4u * p + j;  // tile of the segment
          const bool ok = live && tl < ntile;  // (a record's unused components, and records no wave wrote, are not read as values)
          if (mode == 0) s1 = offer_bulk_fn(s1, ok ? bv[j] : INFINITY, ok ? j : kNoPos, (int)K);
          else push_sub(ok && !(bv[j] > thr), rr, 2u * bs + tl, h2);  // (!(v > thr): a NaN minimum is expanded, never skipped)
        }
      }
    }
  };

  // (diagnostic, VI_FILTER_STATS: s_memtime ticks per stage, summed over the queries into dbg[150..155])
  unsigned long long tk = c.dbg ? __builtin_amdgcn_s_memtime() : 0ull, tks[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) {
    if (c.dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tks[i] += now - tk;
      tk = now;
    }
  };
  // ---- stage 0: the first 256 group records go to LDS in one round of loads (the passes below would
  //      otherwise each pay the global-memory latency per 64 groups) ----
  {
    float4 t4[kCacheG / kWave];
#pragma unroll
    for (uint32_t ch = 0; ch < kCacheG / kWave; ++ch) {
      const uint32_t gidx = ch * kWave + lane;
      t4[ch] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
      if (gidx < G) t4[ch] = c.gval[gbase + gidx];
    }
    uint32_t l4[kCacheG / kWave];
#pragma unroll
    for (uint32_t ch = 0; ch < kCacheG / kWave; ++ch) {
      const uint32_t gidx = ch * kWave + lane;
      l4[ch] = 0u;
      if (gidx < G) l4[ch] = c.gmeta[gbase + gidx];  // probe rank | segment << 6 | lane half << 13
    }
#pragma unroll
    for (uint32_t ch = 0; ch < kCacheG / kWave; ++ch) {
      const uint32_t gidx = ch * kWave + lane;
      if (ch * kWave < G) {
        tcache[gidx] = t4[ch];
        lcache[gidx] = l4[ch];
      }
    }
    lds_sync();
  }
  auto group_values = [&](uint32_t gidx, bool live) {
    float4 T = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    if (live) T = gidx < kCacheG ? tcache[gidx] : c.gval[gbase + gidx];
    return T;
  };
  auto group_place = [&](uint32_t gidx, bool live, uint32_t &r, uint32_t &seg, uint32_t &hh) {
    uint32_t L = 0u;
    if (live) L = gidx < kCacheG ? lcache[gidx] : c.gmeta[gbase + gidx];
    r = L & 63u; seg = (L >> 6) & 127u; hh = L >> 13;
  };
  lap(0);
  // ---- stage 1a: threshold (*) from the K-th smallest value of the group records (key = 4*group + slot);
  //      the groups' smallest values first: they shut the door on most of the others ----
  bool any_full = false;
  {
    s1.init();
    // The K-th smallest of the 4 G recorded values, with its keys.  Offering all of them costs a 64-lane sort per 64 values
    // (16 sorts at G = 256).  Instead: every lane's smallest group minimum belongs to a different sub-block, so the K-th
    // smallest of the 64 lane minima (ONE sort) bounds the K-th smallest of all; the values at or below that bound —
    // K to 2 K of them as a rule — are compacted through LDS and offered in one or two rounds.
    float U = INFINITY;
    if (K <= 64u) {
      float lm = INFINITY;
      for (uint32_t gb = 0; gb < G; gb += kWave) {
        const uint32_t gidx = gb + lane;
        lm = fminf(lm, group_values(gidx, gidx < G).x);
      }
      uint64_t kk = pack_key(lm, (uint32_t)lane);
      wave_sort_u64(kk, lane);
      U = sortable_f32((uint32_t)(readlane_u64(kk, (int)K - 1) >> 32));  // (+inf or NaN: no bound, everything is offered)
    }
    uint32_t ncand = 0;
    bool overflow = !(U < INFINITY);
    if (!overflow) {
      for (uint32_t gb = 0; gb < G && !overflow; gb += kWave) {
        const uint32_t gidx = gb + lane;
        const bool live = gidx < G;
        const float4 T = group_values(gidx, live);
        const float tv[4] = {T.x, T.y, T.z, T.w};
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const bool pass = live && !(tv[j] > U);
          const uint64_t m = __ballot(pass);
          if (!m) continue;
          const uint32_t cnt = (uint32_t)__popcll(m);
          if (ncand + cnt > kPickCap / 2u) { overflow = true; break; }
          if (pass) {
            const uint32_t at = ncand + (uint32_t)__popcll(m & below);
            pick[2u * at] = __float_as_uint(tv[j]);
            pick[2u * at + 1u] = 4u * gidx + j;
          }
          ncand += cnt;
        }
      }
      lds_sync();
    }
    if (!overflow) {
      for (uint32_t c0 = 0; c0 < ncand; c0 += kWave) {
        const bool live = c0 + lane < ncand;
        const float v = live ? __uint_as_float(pick[2u * (c0 + lane)]) : INFINITY;
        s1 = offer_bulk_fn(s1, v, live ? pick[2u * (c0 + lane) + 1u] : kNoPos, (int)K);
      }
      lds_sync();  // (pick is reused by the stages below)
    } else {
      s1.init();
      for (uint32_t gb = 0; gb < G; gb += kWave) {
        const uint32_t gidx = gb + lane;
        const bool live = gidx < G;
        s1 = offer_bulk_fn(s1, group_values(gidx, live).x, live ? 4u * gidx : kNoPos, (int)K);
      }
      for (uint32_t gb = 0; gb < G; gb += kWave) {
        const uint32_t gidx = gb + lane;
        const bool live = gidx < G;
        const float4 T = group_values(gidx, live);
        s1 = offer_bulk_fn(s1, T.y, live ? 4u * gidx + 1u : kNoPos, (int)K);
        s1 = offer_bulk_fn(s1, T.z, live ? 4u * gidx + 2u : kNoPos, (int)K);
        s1 = offer_bulk_fn(s1, T.w, live ? 4u * gidx + 3u : kNoPos, (int)K);
      }
    }
    thr = threshold_of(s1.kth((int)K));
    for (uint32_t gb = 0; gb < G; gb += kWave) {  // is any group's 4th value at or below it?
      const uint32_t gidx = gb + lane;
      const float4 T = group_values(gidx, gidx < G);
      any_full = any_full || __ballot(gidx < G && T.w <= thr) != 0ull;  // (distrust: thr = inf, stage 1b is moot)
    }
  }
  lap(1);
  // ---- stage 1b: neighbours concentrated in few groups hide behind the 4 listed minima and leave the bound
  //      loose; the pair records of those groups list every sub-block minimum.  Their values REPLACE the
  //      group's own (which are among them, so they must not be counted twice): drop the group's entries from
  //      the running top-K, then offer its pair records ----
  if (any_full && !distrust && !(c.xmode & 4u)) {
    {
      bool keep[2] = {false, false};
#pragma unroll
      for (int e = 0; e < Top::kEntries; ++e) {
        const uint32_t key = s1.ent_p(e);
        const bool mine = key != kNoPos && (uint32_t)(64 * e + lane) < K;  // entries beyond the K-th are not needed
        keep[e] = mine && !(group_values(mine ? (key >> 2) : 0u, true).w <= thr);
      }
      s1.rebuild(keep[0], keep[1], (int)K);  // the survivors close ranks
    }
    for (uint32_t gb = 0; gb < G; gb += kWave) {
      const uint32_t gidx = gb + lane;
      const bool live = gidx < G;
      uint32_t r, seg, hh;
      group_place(gidx, live, r, seg, hh);
      scan_groups(live && group_values(gidx, live).w <= thr, r, seg, hh, 0);
    }
    thr = fminf(thr, threshold_of(s1.kth((int)K)));
  }
  lap(2);
  // ---- stage 2: exact re-evaluation of every sub-block whose minimum is at or below thr; such a sub-block sits in a
  //      group whose smallest minimum is at or below thr ----
  for (uint32_t gb = 0; gb < G; gb += kWave) {
    const uint32_t gidx = gb + lane;
    const bool live = gidx < G;
    uint32_t r, seg, hh;
    group_place(gidx, live, r, seg, hh);
    const float4 T = group_values(gidx, live);
    n_full += (uint32_t)__popcll(__ballot(live && T.w <= thr));
    scan_groups(live && !(T.x > thr) && !(c.xmode & 2u), r, seg, hh, 1);
  }
  lap(3);
  drain_pick();
  lap(4);
  if (c.dbg && lane == 0 && (q & c.dbg_mask) == 0u) {
#pragma unroll
    for (int i = 0; i < 5; ++i) atomicAdd(&c.dbg[150 + i], tks[i]);
    atomicAdd(&c.dbg[6], (unsigned long long)n_exact);
    atomicAdd(&c.dbg[7], (unsigned long long)n_scanned);
    atomicAdd(&c.dbg[8], (unsigned long long)(any_full ? 1u : 0u));
    atomicAdd(&c.dbg[9], (unsigned long long)n_full);
    atomicAdd(&c.dbg[10], (unsigned long long)n_sub);
  }
}

struct SelectArgs {
  SelectCommon c;
  uint32_t nq, P, k, segb0;
  const uint32_t *qoff, *qtot, *rel, *pair_pos, *tile_start;
  const uint32_t *probes, *gorder, *first_block, *list_len;
  const uint64_t *ext_ids;
  float *D;
  int64_t *I;
  uint64_t *tie, *slots;
  uint32_t *counts;
};

// one wave per query: top-k over its probed lists in the reference's stable order (ivf_index.rs:264-274)
template <class Top>
__global__ void __launch_bounds__(256, 4) select_kernel(SelectArgs a) {
  __shared__ uint32_t s_pick[4][kPickCap], s_lcache[4][kCacheG];
  __shared__ float4 s_tcache[4][kCacheG];
  __shared__ __attribute__((aligned(16))) uint32_t s_qbytes[4][64];  // the 4 queries as bytes (8-bit lists, D <= 256)
  extern __shared__ __attribute__((aligned(16))) float s_qrows[];  // the 4 query rows of the workgroup: 4 x dim floats
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;
  ProbeRegs pr{0u, 0u, 0u, 0u, 0u, 1u, kNoPos};
  uint32_t mylist = kNoPos;
  if ((uint32_t)lane < a.P) {
    const size_t s = (size_t)q * a.P + lane;
    mylist = a.probes[s];
    pr.g = a.gorder[s];
    pr.rel = a.rel[s];
    if (mylist != kNoPos) {
      pr.len = a.list_len[mylist];
      pr.fb = a.first_block[mylist];
      const uint32_t pp = a.pair_pos[s];  // where the pair sits among the pairs of its list
      const uint32_t nseg = list_segments(pr.len, a.segb0, &pr.segb);
      pr.ng = 2u * nseg;
      pr.boff = (a.tile_start[mylist] + (pp / a.c.gq) * nseg * seg_records(pr.segb)) * (2u * a.c.gq) + (pp % a.c.gq);
    }
  }
  Top sel;
  select_body<Top>(a.c, q, a.qoff[q], a.qtot[q], a.P, pr, a.k, lane, s_pick[wave], s_tcache[wave],
                   s_lcache[wave], s_qrows + (size_t)wave * a.c.dim, sel, s_qbytes[wave]);
  // entry e of lane i holds result 64e + i: map the candidate-order rank g back to the probe rank r
  uint32_t found = 0;
  // probe rank of candidate-order rank g: lane r pushes r to lane g(r) (the ranks of a query's probes are a permutation of
  // 0 .. found-1; lanes without a probe push to themselves, at or above found) — one crossbar push instead of a readlane
  // and a compare per probe and result entry
  const uint32_t inv = (uint32_t)__builtin_amdgcn_ds_permute((int)(4u * (mylist != kNoPos ? pr.g : (uint32_t)lane)), lane);
#pragma unroll
  for (int e = 0; e < Top::kEntries; ++e) {
    const uint32_t key = sel.ent_p(e), idx = 64u * (uint32_t)e + (uint32_t)lane;
    const uint32_t g = key >> kPosBits, pos = key & kPosMask;
    const uint32_t r = (uint32_t)__shfl((int)inv, (int)(g & 63u));
    const bool have = idx < a.k && key != kNoPos;
    found += (uint32_t)__popcll(__ballot(have));
    const uint32_t fbk = (uint32_t)__shfl((int)pr.fb, (int)r);
    if (idx < a.k) {
      const size_t o = (size_t)q * a.k + idx;
      if (have) {
        const uint64_t gslot = (uint64_t)fbk * kWave + pos;
        a.D[o] = sel.ent_d(e);
        a.I[o] = (int64_t)a.ext_ids[gslot];
        if (a.tie) a.tie[o] = ((uint64_t)g << 32) | pos;
        if (a.slots) a.slots[o] = gslot;
      } else {
        a.D[o] = INFINITY;
        a.I[o] = -1;
        if (a.tie) a.tie[o] = ~0ull;
        if (a.slots) a.slots[o] = ~0ull;
      }
    }
  }
  if (a.counts && lane == 0) a.counts[q] = found;
}

struct CoarseSelectArgs {
  SelectCommon c;  // blocks = centroid table
  uint32_t nq, P, nlists, segb, recs;  // recs = group records per query
  const uint32_t *list_shard, *list_len;
  uint32_t *probes, *gorder, *cnt;
  uint32_t query_major;  // direct records laid out [query][record] (filter_kernel: a.direct == 2)
  const float4 *cent_rows;  // the table row-major (rows_from_blocks_kernel) for single-row re-evaluation, or null
  // record counts of the list phase (what pair_groups_kernel computes otherwise)
  uint32_t list_segb0;
  uint32_t *rel, *qtot;
  uint32_t staged;  // single rows fetched by the whole wave through LDS (exact_batch_rows_staged_fn); VI_COARSE_STAGED=0: a row per lane
  uint32_t *pair_rank;  // where the pair stands among the pairs of its (list, sub-bin) — the value its histogram
                        // increment returns — so that the grouping's scatter needs no atomics of its own; or null
};

// one wave per query: the P nearest centroids in (distance, centroid index) order (the reference's stable
// sort, ivf_index.rs:205-220), then shard visiting order + histogram as in coarse_merge_kernel
__global__ void __launch_bounds__(256, 4) coarse_select_kernel(CoarseSelectArgs a) {
  __shared__ uint32_t s_pick[4][kPickCap], s_lcache[4][kCacheG];
  __shared__ float4 s_tcache[4][kCacheG];
  __shared__ __attribute__((aligned(16))) float s_q[4][kNarrowDim];  // (the coarse step runs here only for D <= 128)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;
  ProbeRegs pr{0u, 0u, 0u, 0u, 0u, 1u, 0u};
  if (lane == 0) {
    pr.ng = a.recs; pr.len = a.nlists; pr.segb = a.segb;
    pr.boff = (q / a.c.gq) * (a.recs / 2u) * seg_records(a.segb) * (2u * a.c.gq) + (q % a.c.gq);  // pairs = iota: the query's own position
  }
  FastTopK sel;
  select_body<FastTopK>(a.c, q, (size_t)q * a.recs, a.recs, 1u, pr, a.P, lane, s_pick[wave], s_tcache[wave],
              s_lcache[wave], s_q[wave], sel);
  const uint32_t found = (uint32_t)__popcll(__ballot((uint32_t)lane < a.P && sel.ent_p(0) != kNoPos));
  const uint32_t mylist = (uint32_t)lane < found ? sel.ent_p(0) : kNoPos;
  const uint32_t g = probe_candidate_order(lane, found, mylist, a.list_shard);
  if ((uint32_t)lane < a.P) {
    a.probes[(size_t)q * a.P + lane] = mylist;
    a.gorder[(size_t)q * a.P + lane] = g;
    if (mylist != kNoPos && a.list_len[mylist] > 0) {
      const uint32_t before = atomicAdd(&a.cnt[subbin_index(mylist, q & (kSubBins - 1), a.nlists)], 1u);
      if (a.pair_rank) a.pair_rank[(size_t)q * a.P + lane] = before;
    }
  }
  // group records of the list phase: 2 per (probe, segment); query_offsets_kernel turns the per-query totals
  // into offsets
  uint32_t ng = 0;
  if (mylist != kNoPos) {
    uint32_t sb;
    ng = 2u * list_segments(a.list_len[mylist], a.list_segb0, &sb);
  }
  const uint32_t ig = wave_incl_scan_u32(ng);
  if ((uint32_t)lane < a.P) a.rel[(size_t)q * a.P + lane] = ig - ng;
  if (lane == 63) a.qtot[q] = ig;
}

// The coarse table up to 256 blocks (16 384 centroids): the rank kernel leaves two records per (block, lane half) —
// (minimum with its row, second minimum) of its four 8-row sub-blocks — and this kernel reads ALL of a query's records at
// once (<= 16 per lane): the P-th smallest minimum bounds the P-th distance (the minima belong to different centroids);
// a sub-block whose minimum is at or below the threshold (*) contributes that ONE row to the exact re-evaluation, or
// all 8 when its second minimum is at or below the threshold too.  No group records, no refinement rounds: at P = 32
// about 40 exact distances per query decide the probe list.
constexpr uint32_t kDirectBlocks = 256;
__global__ void __launch_bounds__(256) coarse_select_direct_kernel(CoarseSelectArgs a) {
  __shared__ uint32_t s_pick[4][kPickCap];
  __shared__ __attribute__((aligned(16))) float s_q[4][kNarrowDim];  // (the coarse step runs here only for D <= 128)
  __shared__ __attribute__((aligned(16))) float s_stage[4][kStageFloats];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;
  const SelectCommon &c = a.c;
  uint32_t *pick = s_pick[wave];
  float *qlds = s_q[wave];
  const uint64_t below = (1ull << lane) - 1ull;
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // (diagnostic, VI_FILTER_STATS=2: s_memtime ticks per stage, summed over the queries into dbg[150..155])
  unsigned long long tk = c.dbg ? __builtin_amdgcn_s_memtime() : 0ull, tks[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) {
    if (c.dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tks[i] += now - tk;
      tk = now;
    }
  };
  float qn = 0.0f;
  for (uint32_t e = lane; e < c.dim; e += kWave) {
    const float v = c.Q[(size_t)q * c.dim + e];
    qlds[e] = v;
    const float vc = c.mu ? v - c.mu[e] : v;
    qn += vc * vc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  lds_sync();
  lap(0);
  const float E = c.e_scale * (qn * (1.0f + c.gamma) + 2.0f * c.xmax2);
  const bool distrust = !(qn < 1.0e30f);  // see select_body
  const uint32_t K = a.P, nblk = (a.nlists + kWave - 1) / kWave, nrec = 4u * nblk;  // records: (block, tile, lane half)
  const size_t base = (size_t)(q / c.gq) * nblk * (4u * c.gq) + (q % c.gq);
  constexpr uint32_t kPer = kDirectBlocks * 4 / kWave;  // records per lane
  // (min of sub-block 0 with its row, its second min, the same of sub-block 1)
  auto record = [&](uint32_t i) {
    const uint32_t rec = i * kWave + lane;  // block rec >> 2, tile (rec >> 1) & 1, lane half rec & 1
    float4 r = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    if (rec < nrec)
      r = a.query_major ? c.brec[(size_t)q * nrec + rec]
                        : c.brec[base + (size_t)(rec >> 2) * (4u * c.gq) + (size_t)((rec >> 1) & 1u) * (2u * c.gq) + c.gq * (rec & 1u)];
    return r;
  };
  // bound of the K-th distance: every lane's smallest minimum belongs to a different centroid, so K centroids are at
  // or below the K-th smallest of the 64 lane minima — one 64-lane sort instead of a running top-K over all the minima
  // (K <= 64; the bound sits a few ranks above the exact K-th minimum, which costs a few more single-row evaluations)
  float lm = INFINITY;
  float rb1[kPer][2], rb2[kPer][2];  // (kept for the flags below: 16 registers; a second read from L2 was a round trip per query)
#pragma unroll
  for (uint32_t i = 0; i < kPer; ++i) {
    rb1[i][0] = rb1[i][1] = rb2[i][0] = rb2[i][1] = INFINITY;
    if (i * kWave < nrec) {
      const float4 r = record(i);
      lm = min3_raw(lm, r.x, r.z);
      rb1[i][0] = r.x; rb2[i][0] = r.y; rb1[i][1] = r.z; rb2[i][1] = r.w;
    }
  }
  FastTopK s1;
  s1.init();
  s1 = offer_bulk_fn(s1, lm, (uint32_t)lane, (int)K);
  float thr = INFINITY;
  {
    const float mk = s1.kth((int)K);
    if (!distrust && mk < 1.0e37f) {
      const float scale = fmaxf(mk + qn, 0.0f) + E;
      thr = mk + (2.0f * E + 3.0f * c.gamma * scale) * 1.001f + 1e-30f;
    }
  }
  lap(1);
  FastTopK sel;
  sel.init();
  uint32_t npick = 0;
  auto exact_rows = [&](bool live, uint32_t pos) {
    live = live && pos < a.nlists && !(c.xmode & 1u);
    sel = exact_batch_fn(sel, qlds, c.blocks + ((size_t)((live ? pos : 0u) / kWave) * c.dq) * kWave + (pos % kWave), c.dim, live, pos,
                         (int)K);
  };
  // sub-block s (0/1) of record rec: its first row is register 8s of tile t of lane half h
  auto sub_row = [&](uint32_t rec, uint32_t s, uint32_t e) {
    return (rec >> 2) * kWave + subblock_vector(8u * s + e, (rec >> 1) & 1u, rec & 1u, c.image_order != 0u);
  };
  uint32_t n_single = 0, n_whole = 0;  // (VI_FILTER_STATS=2: rows evaluated alone / whole 8-row sub-blocks)
  auto drain_singles = [&]() {
    while (npick > 0) {
      const uint32_t cnt = npick >= (uint32_t)kWave ? (uint32_t)kWave : npick;
      npick -= cnt;
      n_single += cnt;
      bool live = (uint32_t)lane < cnt;
      const uint32_t pos = live ? pick[npick + lane] : 0u;
      if (a.cent_rows && a.staged) {  // one centroid per lane, the rows fetched by the whole wave
        if (!(c.xmode & 1u)) sel = exact_batch_rows_staged_fn(sel, qlds, a.cent_rows, a.nlists, c.dim, cnt, pos, (int)K, s_stage[wave]);
      } else if (a.cent_rows) {  // one centroid per lane, each its own whole cache lines
        live = live && pos < a.nlists && !(c.xmode & 1u);
        sel = exact_batch_row_fn(sel, qlds, a.cent_rows + (size_t)(live ? pos : 0u) * (c.dim / 4), c.dim, live, pos, (int)K);
      } else {
        exact_rows(live, pos);
      }
    }
  };
  // Which rows go to the exact evaluation: of a sub-block whose minimum is at or below thr the row of that minimum, all
  // eight when its second minimum is too.  A lane counts the rows of its eight sub-blocks, one scan over the wave gives
  // every lane its place in the list, one LDS round writes them (a ballot, a count and an LDS round per sub-block column
  // and kind — sixteen of each — were a fifth of the kernel).  More rows than the list holds (a distrusted query flags
  // everything): the ballot loop below, which drains the list as it fills.
  bool listed = false;
  if (!(c.xmode & 16u)) {
    uint32_t cls = 0, rowbits = 0, mine = 0;  // 2 bits per sub-block: 0 none, 1 the row of the minimum, 2 all eight; 3 bits: that row
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i)
      if (i * kWave < nrec) {
        const uint32_t rec = i * kWave + lane;
#pragma unroll
        for (uint32_t s2 = 0; s2 < 2; ++s2) {
          const bool cand = rec < nrec && !(rb1[i][s2] > thr);
          const bool all8 = cand && (!(rb2[i][s2] > thr) || distrust);
          const uint32_t k = all8 ? 2u : (cand ? 1u : 0u);
          cls |= k << (2u * (2u * i + s2));
          rowbits |= (__float_as_uint(rb1[i][s2]) & 7u) << (3u * (2u * i + s2));
          mine += all8 ? 8u : (cand ? 1u : 0u);
        }
      }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint32_t total = readlane_u(incl, 63);
    if (total <= kPickCap) {
      uint32_t at = incl - mine, left = cls;
      // (a lane flags 0.7 of its 8 sub-blocks on average: as many rounds as the busiest lane has flags — three or four —
      //  each lane taking its next flagged sub-block, instead of eight rounds of mostly idle lanes)
      while (__ballot(left != 0u)) {
        if (left != 0u) {
          const uint32_t sb = (uint32_t)__builtin_ctz(left) >> 1;
          const uint32_t k = (left >> (2u * sb)) & 3u, rec = (sb >> 1) * kWave + (uint32_t)lane;
          left &= ~(3u << (2u * sb));
          if (k == 1u) {
            pick[at] = sub_row(rec, sb & 1u, (rowbits >> (3u * sb)) & 7u);
            at += 1u;
          } else {
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) pick[at + e] = sub_row(rec, sb & 1u, e);
            at += 8u;
            n_whole += 1u;  // (per lane here; summed below when the counters are on)
          }
        }
      }
      npick = total;
      listed = true;
      lds_sync();
      if (c.dbg) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_whole += (uint32_t)__shfl_xor((int)n_whole, o);
      }
    }
  }
#pragma unroll
  for (uint32_t i = 0; i < kPer; ++i)
    if (i * kWave < nrec && !(c.xmode & 16u) && !listed) {
      const uint32_t rec = i * kWave + lane;
      const float4 Ri = record(i);
      const float b1[2] = {Ri.x, Ri.z}, b2[2] = {Ri.y, Ri.w};
#pragma unroll
      for (uint32_t s2 = 0; s2 < 2; ++s2) {
        const bool cand = rec < nrec && !(b1[s2] > thr);
        const bool all8 = cand && (!(b2[s2] > thr) || distrust);
        const bool one = cand && !all8;
        uint64_t m = __ballot(one);
        if (m) {
          const uint32_t cnt = (uint32_t)__popcll(m);
          if (npick + cnt > kPickCap) drain_singles();
          if (one) pick[npick + (uint32_t)__popcll(m & below)] = sub_row(rec, s2, __float_as_uint(b1[s2]) & 7u);
          npick += cnt;
          lds_sync();
        }
        m = __ballot(all8);
        if (m) {  // all 8 rows of the sub-block: into the same list (they used to wait for rounds of their own — a second exact
          // round and a second merge per query for 1.6 sub-blocks on average, a third of the kernel's instructions)
          // (32 lanes at a time: their 256 rows fill the list exactly — a distrusted query flags every sub-block)
#pragma unroll
          for (uint32_t half = 0; half < 2u; ++half) {
            const uint64_t mh = m & (half ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull);
            if (!mh) continue;
            const uint32_t cnt = 8u * (uint32_t)__popcll(mh);
            if (npick + cnt > kPickCap) drain_singles();
            if (all8 && ((uint32_t)lane >> 5) == half) {
              const uint32_t at = npick + 8u * (uint32_t)__popcll(mh & below);
#pragma unroll
              for (uint32_t e = 0; e < 8u; ++e) pick[at + e] = sub_row(rec, s2, e);
            }
            npick += cnt;
            n_whole += cnt >> 3;
            lds_sync();
          }
        }
      }
    }
  lap(2);
  drain_singles();
  lap(3);
  // ---- the same tail as coarse_select_kernel: probes, candidate order, histogram, record offsets of the list phase ----
  const uint32_t found = (uint32_t)__popcll(__ballot((uint32_t)lane < a.P && sel.ent_p(0) != kNoPos));
  const uint32_t mylist = (uint32_t)lane < found ? sel.ent_p(0) : kNoPos;
  const uint32_t g = (c.xmode & 32u) ? (uint32_t)lane : probe_candidate_order(lane, found, mylist, a.list_shard);
  if ((uint32_t)lane < a.P) {
    a.probes[(size_t)q * a.P + lane] = mylist;
    a.gorder[(size_t)q * a.P + lane] = g;
    if (mylist != kNoPos && a.list_len[mylist] > 0) {
      const uint32_t before = atomicAdd(&a.cnt[subbin_index(mylist, q & (kSubBins - 1), a.nlists)], 1u);
      if (a.pair_rank) a.pair_rank[(size_t)q * a.P + lane] = before;
    }
  }
  uint32_t ng = 0;
  if (mylist != kNoPos) {
    uint32_t sb;
    ng = 2u * list_segments(a.list_len[mylist], a.list_segb0, &sb);
  }
  const uint32_t ig = wave_incl_scan_u32(ng);
  if ((uint32_t)lane < a.P) a.rel[(size_t)q * a.P + lane] = ig - ng;
  if (lane == 63) a.qtot[q] = ig;
  lap(4);
  if (c.dbg && lane == 0 && (q & 63u) == 0u) {  // (every 64th query: 70 000 same-address atomics would be most of the kernel)
    for (int i = 0; i < 5; ++i) atomicAdd(&c.dbg[150 + i], tks[i]);
    atomicAdd(c.dbg + 6, (unsigned long long)n_single);
    atomicAdd(c.dbg + 7, (unsigned long long)n_whole);
  }
}

// a ring of three tile buffers (two tiles in flight per workgroup) for the hi-planes-only list ranking: measured equal to
// two buffers on the bench workload (0.303 ms both: the launch is not waiting on the fabric), so off unless VI_FILTER_NBUF=3
inline bool nbuf3_ok() {
  const char *e = getenv("VI_FILTER_NBUF");
  return e && *e == '3';
}

template <int NG>
vi_status launch_filter_t(const FilterArgs &a, uint32_t nitems, int rank_mode, uint32_t gq, hipStream_t st) {
  if (nitems == 0) return VI_OK;
  const bool table = a.qoff == nullptr;
  const dim3 grid(nitems);
  if (gq == 32) {  // one wave per work item (lists only): single buffer, many workgroups per CU
    const dim3 block(64);
    if (rank_mode == 2) hipLaunchKernelGGL((filter_kernel<NG, 1, false, 2, 32>), grid, block, 0, st, a);
    else if (rank_mode == 1) hipLaunchKernelGGL((filter_kernel<NG, 1, false, 1, 32>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((filter_kernel<NG, 1, false, 0, 32>), grid, block, 0, st, a);
  } else {
    const dim3 block(256);
    if (rank_mode == 2) {  // half-size tiles: two buffers fit where one full image did, the next tile loads during the MFMAs
      if (table) hipLaunchKernelGGL((filter_kernel<NG, 2, true, 2, 128>), grid, block, 0, st, a);
      else if (nbuf3_ok()) hipLaunchKernelGGL((filter_kernel<NG, 3, false, 2, 128>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((filter_kernel<NG, 2, false, 2, 128>), grid, block, 0, st, a);
    } else if (rank_mode == 1) {  // full images: one buffer, three workgroups per CU (two buffers cost the third: measured slower)
      if (table) hipLaunchKernelGGL((filter_kernel<NG, 1, true, 1, 128>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((filter_kernel<NG, 1, false, 1, 128>), grid, block, 0, st, a);
    } else {
      if (table) hipLaunchKernelGGL((filter_kernel<NG, 1, true, 0, 128>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((filter_kernel<NG, 1, false, 0, 128>), grid, block, 0, st, a);
    }
  }
  VI_HIP(hipGetLastError());
  return VI_OK;
}

vi_status launch_filter(const FilterArgs &a, uint32_t dq, uint32_t nitems, int rank_mode, uint32_t gq, hipStream_t st) {
  switch (dq / 2) {  // dq is a multiple of 4
    case 2: return launch_filter_t<2>(a, nitems, rank_mode, gq, st);
    case 4: return launch_filter_t<4>(a, nitems, rank_mode, gq, st);
    case 6: return launch_filter_t<6>(a, nitems, rank_mode, gq, st);
    case 8: return launch_filter_t<8>(a, nitems, rank_mode, gq, st);
    case 10: return launch_filter_t<10>(a, nitems, rank_mode, gq, st);
    case 12: return launch_filter_t<12>(a, nitems, rank_mode, gq, st);
    case 14: return launch_filter_t<14>(a, nitems, rank_mode, gq, st);
    case 16: return launch_filter_t<16>(a, nitems, rank_mode, gq, st);
    default: return fail(VI_ERR_OTHER, "unsupported dimension for the MFMA filter");
  }
}

// rank arithmetic: bf16 x 3 unless VI_FILTER_BF16=0 (f32 MFMA)
bool rank_bf16() {
  const char *e = getenv("VI_FILTER_BF16");
  return !(e && *e == '0');
}

// RANK 2 (hi planes only when the stored values are bf16-exact) unless VI_FILTER_HI_ONLY=0
bool hi_only_ok() {
  const char *e = getenv("VI_FILTER_HI_ONLY");
  return !(e && *e == '0');
}

// Real-valued lists (not bf16-exact) ranked from their hi planes alone instead of hi + lo (bf16 x 3): a third (queries'
// hi + lo planes: mode 1) or a sixth (queries' hi plane only: mode 2) of the matrix work, paid for with a wider margin
// (select_body: 2 |q| max|v - hi(v)|, plus |query residual| max|v| in mode 2 — the residual norms are measured, not
// bounded by 2^-8 |v|: rounding to nearest leaves about a third of that), which the select turns into more sub-blocks
// re-evaluated exactly; no result depends on a rank value (file header).  Chosen per index from its sampled spread;
// VI_RANK_APPROX=0 / 1 / 2 forces bf16 x 3 / queries hi + lo / queries' hi plane only.
int rank_approx_mode(const DeviceIndex &ix) {
  if (const char *e = getenv("VI_RANK_APPROX")) {
    const int v = atoi(e);
    return v < 0 || v > 2 ? 1 : v;
  }
  // a typical query is as long as a typical stored vector, and its image's residual about twice that of a stored vector's
  const double qlen = std::sqrt((double)(ix.centered ? ix.mean_norm2_c : ix.mean_norm2));
  const double vmax = std::sqrt((double)(ix.centered ? ix.xmax2_c : ix.xmax2)), rho = std::sqrt((double)ix.rho2_max);
  const double unit1 = 2.0 * qlen * rho, unit2 = unit1 + 2.0 * rho * vmax;
  if (getenv("VI_DEBUG_APPROX"))
    fprintf(stderr, "[vi] approx: centred %d qlen %.4g vmax %.4g rho %.4g spread %.4g unit1/spread %.4g unit2/spread %.4g\n", (int)ix.centered,
            qlen, vmax, rho, (double)ix.mean_spread, unit1 / (double)ix.mean_spread, unit2 / (double)ix.mean_spread);
  // The margin grows by `unit`; what it admits grows with unit / (distance of a vector to its neighbours), for which the
  // spread of the lists stands in.
  if (!(ix.mean_spread > 0.0f)) return 0;
  if (unit2 <= kApproxRatio * ix.mean_spread) return 2;
  if (unit1 <= kApproxRatio * ix.mean_spread) return 1;
  return 0;
}

SelectCommon select_common(const DeviceIndex &ix, const float *Qd, const float4 *blocks, float xmax2, uint32_t gq, bool wave_order = false,
                           int trunc = 0) {
  const double u = 1.01 * std::ldexp(1.0, -24);
  SelectCommon c{};
  c.Q = Qd; c.dim = ix.dim; c.dq = ix.dq; c.blocks = blocks;
  c.gval = (const float4 *)ix.cur().ws.gval.p; c.gmeta = (const uint32_t *)ix.cur().ws.gpos.p;
  c.brec = (const float4 *)ix.cur().ws.brec.p;
  c.gamma = (float)((ix.dim + 2.0) * u);
  // |ranked value - (||v||^2 - 2 q.v)| <= e_scale (||q||^2 + 2 max||v||^2):
  //   f32 MFMA : (D+2) u'  accumulation of D products + the norm
  //   bf16 x 3 : 2^-15 for the dropped lo.lo product and the two split residuals (bf16 keeps 8 significant bits:
  //              |x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-17 |x|; 2 (|ql.vl| + |qr.v| + |q.vr|) <= 2 (2^-16 + 2 * 2^-17)
  //              |q||v| <= 2^-15 (|q|^2 + |v|^2) — round 2 budgeted 3 * 2^-18 here, 2.7 times too little), and
  //              (3D+2) * 2u' for the f32 accumulation of 3D exact bf16 products (2u': also covers an accumulator that truncates)
  const double acc = rank_bf16() ? (3.0 * ix.dim + 2.0) * 2.0 * u + 1.01 * std::ldexp(1.0, -15) : (ix.dim + 2.0) * u;
  //   (real-valued lists ranked from their bf16 hi planes alone: SelectCommon::trunc, added per query in select_body)
  // centred images (DeviceIndex::centered): v - mu and q - mu are rounded before they are split — the ranked pair sits
  // within 2^-24 (|q'| + |v'|) of the true one, its distance within 4 * 2^-24 (|q'|^2 + |v'|^2) of the true distance
  const bool centred = ix.centered && rank_bf16();
  c.e_scale = (float)(acc + (centred ? 6.0 * u : 0.0));
  c.trunc = (uint32_t)trunc;
  c.rho_max = (float)(std::sqrt((double)ix.rho2_max) * 1.0001);
  c.vmax = (float)(std::sqrt((double)xmax2) * 1.0001);
  c.xmax2 = xmax2;
  c.mu = centred ? ix.centre.p : nullptr;
  c.gq = gq;
  c.image_order = rank_bf16() ? 1u : 0u;
  c.wave_order = wave_order ? 1u : 0u;
  c.hi_nat = nullptr;
  c.u8_nat = nullptr;
  // per-wave counters go to two addresses: 2 same-address atomics per query cost more than the whole select, so
  // they are a diagnostic (VI_FILTER_STATS=1), not part of the normal path
  c.dbg = getenv("VI_FILTER_STATS") ? (unsigned long long *)ix.cur().ws.stats.p : nullptr;
  { const char *e = getenv("VI_FILTER_STATS"); c.dbg_mask = e && *e == '4' ? 63u : 0u; }
  { const char *xm = getenv("VI_SELECT_XMODE"); c.xmode = xm ? (uint32_t)atoi(xm) : 0u; }
  return c;
}

uint32_t env_xmode() {
  const char *xm = getenv("VI_FILTER_XMODE");
  return xm ? (uint32_t)atoi(xm) : 0u;
}

// items of one tile stream dealt to the same XCD in a row (item_desc_kernel); VI_ITEM_RUN=1: plain round-robin
uint32_t item_run() {
  const char *e = getenv("VI_ITEM_RUN");
  const int v = e ? atoi(e) : 8;
  return (uint32_t)std::min(std::max(v, 1), 256);
}

}  // namespace

// norms of the stored vectors (MFMA accumulator init) — called once after the blocks are built
vi_status compute_slot_norms(DeviceIndex *ix) {
  const uint64_t nslots = ix->lists.nblocks * kWave;
  VI_TRY(ix->xnorm.reserve(std::max<uint64_t>(1, nslots)));
  DevBuf<uint32_t> mx;
  VI_TRY(mx.reserve(1));
  VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
  if (nslots) {
    hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream,
                       (const float4 *)ix->lists.blocks.p, ix->dq, nslots, ix->xnorm.p, mx.p);
    if (ix->nlists)
      hipLaunchKernelGGL(pad_norms_kernel, dim3((uint32_t)((ix->nlists * 64 + 255) / 256)), dim3(256), 0, ix->stream,
                         ix->list_first_block.p, ix->list_len.p, (uint32_t)ix->nlists, ix->xnorm.p);
    VI_HIP(hipGetLastError());
  }
  uint32_t bits = 0;
  VI_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
  VI_HIP(hipStreamSynchronize(ix->stream));
  float f;
  std::memcpy(&f, &bits, 4);
  ix->xmax2 = f;
  // the coarse table: pad slots (>= nlists) must never rank
  const uint64_t cslots = ix->centroids.nblocks * kWave;
  VI_TRY(ix->cent_xnorm.reserve(std::max<uint64_t>(1, cslots)));
  VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
  if (cslots) {
    hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((cslots + 255) / 256)), dim3(256), 0, ix->stream,
                       (const float4 *)ix->centroids.blocks.p, ix->dq, cslots, ix->cent_xnorm.p, mx.p);
    VI_HIP(hipGetLastError());
    const uint64_t npad = cslots - ix->nlists;
    if (npad) {
      std::vector<float> inf(npad, kBig);
      VI_HIP(hipMemcpyAsync(ix->cent_xnorm.p + ix->nlists, inf.data(), npad * 4, hipMemcpyHostToDevice, ix->stream));
    }
  }
  VI_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
  VI_HIP(hipStreamSynchronize(ix->stream));
  std::memcpy(&f, &bits, 4);
  ix->cent_xmax2 = f;
  VI_TRY(ix->xnorm_img.reserve(std::max<uint64_t>(1, nslots)));
  VI_TRY(ix->cent_xnorm_img.reserve(std::max<uint64_t>(1, cslots)));
  if (nslots)
    hipLaunchKernelGGL(image_norms_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream, ix->xnorm.p,
                       nslots, ix->xnorm_img.p);
  if (cslots)
    hipLaunchKernelGGL(image_norms_kernel, dim3((uint32_t)((cslots + 255) / 256)), dim3(256), 0, ix->stream,
                       ix->cent_xnorm.p, cslots, ix->cent_xnorm_img.p);
  VI_HIP(hipGetLastError());
  // bf16 hi/lo images of the lists and of the centroid table (same size as the f32 blocks)
  if ((ix->dim & 3) == 0 && ix->dim <= kMaxFilterDim) {
    const uint64_t per_block = (uint64_t)ix->dq * kWave * 4;  // uint32 words per block
    VI_TRY(ix->lists_bf16.reserve(std::max<uint64_t>(1, ix->lists.nblocks * per_block)));
    VI_TRY(ix->cent_bf16.reserve(std::max<uint64_t>(1, ix->centroids.nblocks * per_block)));
    const uint64_t nt_l = ix->lists.nblocks * (ix->dq / 4) * 128, nt_c = ix->centroids.nblocks * (ix->dq / 4) * 128;
    if (nt_l)
      hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nt_l + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->lists.blocks.p, ix->dq, ix->lists.nblocks, (uint4 *)ix->lists_bf16.p);
    if (nt_c)
      hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nt_c + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->centroids.blocks.p, ix->dq, ix->centroids.nblocks, (uint4 *)ix->cent_bf16.p);
    VI_HIP(hipGetLastError());
    // bf16-exact stored values (8-bit descriptors): the lo planes are all zero and need not be streamed
    uint32_t h_any[2] = {0u, 0u};
    VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
    const uint64_t np_l = ix->lists.nblocks * ix->dq, np_c = ix->centroids.nblocks * ix->dq;
    if (np_l)
      hipLaunchKernelGGL(lo_plane_any_kernel, dim3((uint32_t)((np_l * 64 + 255) / 256)), dim3(256), 0, ix->stream,
                         (const uint4 *)ix->lists_bf16.p, np_l, mx.p);
    VI_HIP(hipMemcpyAsync(&h_any[0], mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
    VI_HIP(hipStreamSynchronize(ix->stream));
    VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
    if (np_c)
      hipLaunchKernelGGL(lo_plane_any_kernel, dim3((uint32_t)((np_c * 64 + 255) / 256)), dim3(256), 0, ix->stream,
                         (const uint4 *)ix->cent_bf16.p, np_c, mx.p);
    VI_HIP(hipMemcpyAsync(&h_any[1], mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
    VI_HIP(hipStreamSynchronize(ix->stream));
    ix->lists_lo_zero = np_l > 0 && h_any[0] == 0;
    ix->cent_lo_zero = np_c > 0 && h_any[1] == 0;
    ix->centered = false;
    const char *ce = getenv("VI_CENTER");
    if (!ix->lists_lo_zero && ix->dim <= kNarrowDim && ix->nlists && ix->nvec_resident && !(ce && *ce == '0')) {
      // real-valued lists: images about the mean of the stored vectors when that at least halves the norms the margins
      // scale with (VI_CENTER=1: always, =0: never)
      DevBuf<double> sums;
      DevBuf<float> cn, ccn;
      DevBuf<double> nsum;
      VI_TRY(sums.reserve((uint64_t)ix->dq * 4));
      VI_TRY(ix->centre.reserve((uint64_t)ix->dq * 4));
      VI_TRY(cn.reserve(nslots));
      VI_TRY(ccn.reserve(cslots));
      VI_HIP(hipMemsetAsync(sums.p, 0, (uint64_t)ix->dq * 4 * sizeof(double), ix->stream));
      hipLaunchKernelGGL(mean_kernel, dim3(ix->dq, 64), dim3(256), 0, ix->stream, (const float4 *)ix->lists.blocks.p, ix->dq,
                         ix->lists.nblocks, sums.p);
      hipLaunchKernelGGL(mean_finish_kernel, dim3((ix->dq * 4 + 255) / 256), dim3(256), 0, ix->stream, sums.p, ix->dim, ix->dq * 4,
                         (double)ix->nvec_resident, ix->centre.p);
      uint32_t mxb[2] = {0u, 0u};
      DevBuf<uint32_t> mx2;
      VI_TRY(mx2.reserve(2));
      VI_HIP(hipMemsetAsync(mx2.p, 0, 8, ix->stream));
      hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->lists.blocks.p, ix->dq, nslots, cn.p, mx2.p, (const float4 *)ix->centre.p);
      hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((cslots + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->centroids.blocks.p, ix->dq, (uint64_t)ix->nlists, ccn.p, mx2.p + 1, (const float4 *)ix->centre.p);
      hipLaunchKernelGGL(pad_norms_kernel, dim3((uint32_t)((ix->nlists * 64 + 255) / 256)), dim3(256), 0, ix->stream,
                         ix->list_first_block.p, ix->list_len.p, (uint32_t)ix->nlists, cn.p);
      VI_HIP(hipMemsetAsync(mx2.p, 0, 8, ix->stream));  // (the pad slots' zero vectors entered the kernel's own maximum)
      hipLaunchKernelGGL(max_finite_kernel, dim3(256), dim3(256), 0, ix->stream, cn.p, nslots, mx2.p);
      hipLaunchKernelGGL(max_finite_kernel, dim3(64), dim3(256), 0, ix->stream, ccn.p, (uint64_t)ix->nlists, mx2.p + 1);
      VI_HIP(hipGetLastError());
      VI_HIP(hipMemcpyAsync(mxb, mx2.p, 8, hipMemcpyDeviceToHost, ix->stream));
      std::vector<double> hs((size_t)ix->dq * 4);
      VI_HIP(hipMemcpyAsync(hs.data(), sums.p, hs.size() * sizeof(double), hipMemcpyDeviceToHost, ix->stream));
      VI_HIP(hipStreamSynchronize(ix->stream));
      float xc, cc;
      std::memcpy(&xc, &mxb[0], 4);
      std::memcpy(&cc, &mxb[1], 4);
      double mu2 = 0.0;  // |mu|^2; the mean of |v - mu|^2 is the mean of |v|^2 less this
      for (uint32_t e = 0; e < ix->dim; ++e) { const double m = hs[e] / (double)ix->nvec_resident; mu2 += m * m; }
      double mean_raw = 0.0;
      {  // the mean squared norm of ALL stored vectors (list_spread_kernel below only samples)
        DevBuf<double> acc;
        VI_TRY(acc.reserve(1));
        VI_HIP(hipMemsetAsync(acc.p, 0, sizeof(double), ix->stream));
        hipLaunchKernelGGL(sum_f32_kernel, dim3(256), dim3(256), 0, ix->stream, ix->xnorm.p, nslots, acc.p);
        VI_HIP(hipGetLastError());
        VI_HIP(hipMemcpyAsync(&mean_raw, acc.p, sizeof(double), hipMemcpyDeviceToHost, ix->stream));
        VI_HIP(hipStreamSynchronize(ix->stream));
        mean_raw /= (double)ix->nvec_resident;
      }
      const double mean_c = std::max(0.0, mean_raw - mu2);
      const bool gain = mean_c + 2.0 * (double)xc < 0.5 * (mean_raw + 2.0 * (double)ix->xmax2);
      if ((ce && *ce == '1') || gain) {
        ix->centered = true;
        ix->mean_norm2_c = (float)mean_c;
        ix->xmax2_c = xc;
        ix->cent_xmax2_c = cc;
        const uint64_t npad = cslots - ix->nlists;
        if (npad) {
          std::vector<float> inf(npad, kBig);
          VI_HIP(hipMemcpyAsync(ccn.p + ix->nlists, inf.data(), npad * 4, hipMemcpyHostToDevice, ix->stream));
          VI_HIP(hipStreamSynchronize(ix->stream));  // (inf lives on this frame)
        }
        hipLaunchKernelGGL(image_norms_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream, cn.p, nslots,
                           ix->xnorm_img.p);
        hipLaunchKernelGGL(image_norms_kernel, dim3((uint32_t)((cslots + 255) / 256)), dim3(256), 0, ix->stream, ccn.p, cslots,
                           ix->cent_xnorm_img.p);
        hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nt_l + 255) / 256)), dim3(256), 0, ix->stream,
                           (const float4 *)ix->lists.blocks.p, ix->dq, ix->lists.nblocks, (uint4 *)ix->lists_bf16.p, (const float4 *)ix->centre.p);
        hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nt_c + 255) / 256)), dim3(256), 0, ix->stream,
                           (const float4 *)ix->centroids.blocks.p, ix->dq, ix->centroids.nblocks, (uint4 *)ix->cent_bf16.p, (const float4 *)ix->centre.p);
        VI_HIP(hipGetLastError());
        VI_HIP(hipStreamSynchronize(ix->stream));
        ix->cent_lo_zero = false;
      }
    }
    ix->rho2_max = 0.0f;
    if (!ix->lists_lo_zero && ix->dim <= kNarrowDim && nslots) {  // real-valued lists: what their hi planes leave out (rank_approx_mode)
      VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
      hipLaunchKernelGGL(trunc_residual_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->lists.blocks.p, ix->dq, nslots, ix->xnorm.p, ix->centered ? (const float4 *)ix->centre.p : nullptr, mx.p);
      VI_HIP(hipGetLastError());
      VI_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
      VI_HIP(hipStreamSynchronize(ix->stream));
      std::memcpy(&ix->rho2_max, &bits, 4);
    }
    if (ix->nlists && ix->dim <= kNarrowDim) {  // single-row exact re-evaluation of the coarse select
      const uint32_t nquad = ix->dim / 4;
      VI_TRY(ix->cent_rows.reserve((uint64_t)ix->nlists * ix->dim));
      const uint64_t nt = (uint64_t)ix->nlists * nquad;
      hipLaunchKernelGGL(rows_from_blocks_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->centroids.blocks.p, ix->dq, (uint32_t)ix->nlists, nquad, (float4 *)ix->cent_rows.p);
      VI_HIP(hipGetLastError());
    }
    if (ix->nlists && ix->dim <= kNarrowDim && ix->lists.nblocks && !ix->lists_lo_zero) {
      // real-valued lists: how far the vectors sit from their centroids, against how large they are (rank_approx_mode)
      DevBuf<double> sums;
      VI_TRY(sums.reserve(3));
      VI_HIP(hipMemsetAsync(sums.p, 0, 3 * sizeof(double), ix->stream));
      hipLaunchKernelGGL(list_spread_kernel, dim3((uint32_t)ix->nlists), dim3(64), 0, ix->stream, (const float4 *)ix->lists.blocks.p, ix->dq,
                         (const float4 *)ix->cent_rows.p, ix->dim, ix->list_first_block.p, ix->list_len.p, (uint32_t)ix->nlists, 8u, sums.p);
      VI_HIP(hipGetLastError());
      double h[3] = {0, 0, 0};
      VI_HIP(hipMemcpyAsync(h, sums.p, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
      VI_HIP(hipStreamSynchronize(ix->stream));
      if (h[2] > 0) { ix->mean_spread = (float)(h[0] / h[2]); ix->mean_norm2 = (float)(h[1] / h[2]); }
    }
    if (ix->lists_lo_zero && ix->dim <= kNarrowDim && ix->lists.nblocks) {  // 8-bit descriptors?  (bf16-exact is necessary)
      VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
      const uint64_t nquads = ix->lists.nblocks * ix->dq * 64;
      hipLaunchKernelGGL(u8_check_kernel, dim3((uint32_t)((nquads + 255) / 256)), dim3(256), 0, ix->stream,
                         (const float4 *)ix->lists.blocks.p, nquads, mx.p);
      uint32_t not_u8 = 1;
      VI_HIP(hipMemcpyAsync(&not_u8, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
      VI_HIP(hipStreamSynchronize(ix->stream));
      if (!not_u8) {
        const uint64_t nt8 = ix->lists.nblocks * (ix->dq / 4) * 64;
        VI_TRY(ix->lists_u8_nat.reserve(nt8 * 4));
        hipLaunchKernelGGL(u8_natural_kernel, dim3((uint32_t)((nt8 + 255) / 256)), dim3(256), 0, ix->stream,
                           (const float4 *)ix->lists.blocks.p, ix->dq, ix->lists.nblocks, (uint4 *)ix->lists_u8_nat.p);
        VI_HIP(hipGetLastError());
      }
    }
    if (ix->lists_lo_zero && ix->dim <= kNarrowDim && !ix->lists_u8_nat.p) {  // exact re-evaluation from bf16 (select_kernel), VI_EXACT_BF16=0: from f32
      VI_TRY(ix->lists_hi_nat.reserve(ix->lists.nblocks * per_block / 2));
      hipLaunchKernelGGL(hi_natural_kernel, dim3((uint32_t)((nt_l + 255) / 256)), dim3(256), 0, ix->stream,
                         (const uint4 *)ix->lists_bf16.p, ix->dq / 4, ix->lists.nblocks, (uint4 *)ix->lists_hi_nat.p);
      VI_HIP(hipGetLastError());
    }
  }
  const uint32_t one_first[1] = {0u}, one_len[1] = {(uint32_t)ix->nlists};
  VI_TRY(ix->c_first.reserve(1));
  VI_TRY(ix->c_len.reserve(1));
  VI_HIP(hipMemcpy(ix->c_first.p, one_first, 4, hipMemcpyHostToDevice));
  VI_HIP(hipMemcpy(ix->c_len.p, one_len, 4, hipMemcpyHostToDevice));
  return VI_OK;
}

// coarse quantizer on the matrix cores: the centroid table is one "list" probed by every query.
// Leaves probes / gorder and the per-list histogram (ws.cnt) behind, like stage_coarse.
vi_status stage_coarse_filter(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, uint32_t list_segb0,
                              hipStream_t st, bool histogram_cleared = false) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  VI_TRY(ws.cnt.reserve(2 * subbin_words(nlists)));
  if (!histogram_cleared) VI_HIP(hipMemsetAsync(ws.cnt.p, 0, subbin_words(nlists) * sizeof(uint32_t), st));
  VI_TRY(ws.probes.reserve(nq * P));
  VI_TRY(ws.gorder.reserve(nq * P));
  // one list, every query probes it: groups of 128 queries x segments of segb blocks
  const uint32_t segb0 = 4;
  uint32_t segb;
  const uint32_t nseg = list_segments((uint32_t)nlists, segb0, &segb);
  const uint32_t recs = 2u * nseg;
  const uint32_t ngroups = (uint32_t)((nq + kGroupQ - 1) / kGroupQ);
  const uint32_t h_seg[2] = {0u, (uint32_t)nq}, h_item[2] = {0u, ngroups * nseg};
  VI_TRY(ws.c_seg.reserve(2));
  VI_TRY(ws.c_item.reserve(2));
  VI_TRY(ws.c_pairs.reserve(nq));
  VI_TRY(ws.gval.reserve(nq * recs * 4));
  VI_TRY(ws.gpos.reserve(nq * recs));
  const char *de = getenv("VI_COARSE_DIRECT");
  const bool direct = ix.centroids.nblocks <= kDirectBlocks && !(de && *de == '0');
  VI_TRY(ws.brec.reserve((uint64_t)ngroups * (direct ? 2 * ix.centroids.nblocks : (uint64_t)nseg * seg_records(segb)) * 256 * 4));
  VI_TRY(ws.stats.reserve(160));
  if (ws.c_nq != nq) {  // the table's one-list grouping depends on the batch size only
    VI_HIP(hipMemcpyAsync(ws.c_seg.p, h_seg, 8, hipMemcpyHostToDevice, st));
    VI_HIP(hipMemcpyAsync(ws.c_item.p, h_item, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(iota_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.c_pairs.p, (uint32_t)nq);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(st));  // h_seg / h_item live on this stack frame
    ws.c_nq = nq;
  }
  bool qmajor = false;
  {
    FilterArgs a{};
    a.blocks = rank_bf16() ? (const float4 *)ix.cent_bf16.p : (const float4 *)ix.centroids.blocks.p;
    a.xnorm = rank_bf16() ? ix.cent_xnorm_img.p : ix.cent_xnorm.p; a.dq = dq; a.dim = dim; a.Q = Qd;
    a.first_block = ix.c_first.p; a.list_len = ix.c_len.p; a.item_start = ws.c_item.p; a.seg_start = ws.c_seg.p;
    a.pairs = ws.c_pairs.p; a.nlists = 1; a.P = 1; a.segb0 = segb0;
    a.qoff = nullptr; a.rel = nullptr; a.rec_stride = recs;
    a.tile_start = ix.c_first.p;  // one list: its tiles start at 0 (c_first holds a single 0)
    a.gval = (float4 *)ws.gval.p; a.gmeta = ws.gpos.p; a.brec = (float4 *)ws.brec.p;
    { const char *e = getenv("VI_COARSE_QMAJOR"); qmajor = direct && !(e && *e == '0'); }
    a.direct = direct ? (qmajor ? 2u : 1u) : 0u;
    a.qimg = rank_bf16() ? (const uint4 *)ws.qimg.p : nullptr;
    VI_TRY(launch_filter(a, dq, ngroups * nseg, rank_bf16() ? (ix.cent_lo_zero && hi_only_ok() ? 2 : 1) : 0, kGroupQ, st));
  }
  {
    CoarseSelectArgs a{select_common(ix, Qd, (const float4 *)ix.centroids.blocks.p, rank_bf16() && ix.centered ? ix.cent_xmax2_c : ix.cent_xmax2, kGroupQ), (uint32_t)nq, P,
                       (uint32_t)nlists, segb, recs, ix.list_shard.p, ix.list_len.p, ws.probes.p, ws.gorder.p,
                       ws.cnt.p, qmajor ? 1u : 0u, nullptr, list_segb0, ws.pair_rel.p, ws.qtot.p};
    { const char *e = getenv("VI_COARSE_ROWS"); if (ix.cent_rows.p && !(e && *e == '0')) a.cent_rows = (const float4 *)ix.cent_rows.p; }
    { const char *e = getenv("VI_COARSE_STAGED"); a.staged = (e && *e == '0') || (ix.dim & 15u) ? 0u : 1u; }
    {
      const char *e = getenv("VI_SCATTER_RANKED");
      ws.pair_rank_valid = !(e && *e == '0');  // (both coarse selects keep what their histogram increment returns)
      if (ws.pair_rank_valid) { VI_TRY(ws.pair_rank.reserve(nq * P)); a.pair_rank = ws.pair_rank.p; }
    }
    { const char *e = getenv("VI_FILTER_STATS"); if (!(e && *e == '2')) a.c.dbg = nullptr; }  // '2': count the coarse step
    { const char *e = getenv("VI_SELECT_XMODE_COARSE"); a.c.xmode = e ? (uint32_t)atoi(e) : 0u; }
    if (direct) a.c.e_scale += (float)(1.01 * std::ldexp(1.0, -20));  // the row index rides in 3 mantissa bits of the minima
    if (direct) hipLaunchKernelGGL(coarse_select_direct_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(coarse_select_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, a);
    VI_HIP(hipGetLastError());
  }
  return VI_OK;
}

bool filter_path_applicable(const DeviceIndex &ix, uint64_t nq, uint64_t k, uint32_t P) {
  const char *force = getenv("VI_FILTER");
  if (force && *force == '0') return false;
  if (ix.order != VI_ORDER_SCALAR || ix.dim > kMaxFilterDim || (ix.dim & 3) || ix.dim < 4) return false;
  if (ix.dim > kNarrowDim && !rank_bf16()) return false;  // the wide kernel ranks with bf16 x 3 only
  if (k > 2 * kMaxSelect || P > kMaxSelect || P < 1) return false;  // k <= 128 (WaveTop128), n_probe <= 64
  if (ix.lists.nblocks * 64ull >= (1ull << kPosBits)) return false;  // record position < 2^26, block < 2^20
  if (!(ix.xmax2 < 1.0e30f) || !(ix.cent_xmax2 < 1.0e30f)) return false;  // norms must stay far below kBig
  (void)nq;
  // measured on the bench index from nq = 1 (0.19 ms vs 0.83 ms) to nq = 10 000 (0.93 ms vs 6 ms): the MFMA engine
  // also wins on tiny batches, because it cuts long lists into segments that run in parallel
  return true;
}

// the list-phase segment size (VI_FILTER_SEGB)
static uint32_t list_segb0() {
  const char *sb = getenv("VI_FILTER_SEGB");
  return sb ? (uint32_t)std::max(1, atoi(sb)) : 32u;  // <= 2048 vectors per work item
}

// the batch's queries as MFMA operands: -2 q split into bf16 hi / lo once (a query sits in n_probe work items)
static vi_status build_query_image(const DeviceIndex &ix, const float *Qd, uint64_t nq, hipStream_t st, uint32_t *zero = nullptr,
                                   uint64_t zero_words = 0) {
  if (!rank_bf16()) return VI_OK;
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t nc = ix.dq / 4;
  VI_TRY(ws.qimg.reserve((uint64_t)nq * nc * 4 * 4));  // uint32 words: 4 pieces of 16 B per (query, chunk)
  if (!ws.stats_zeroed) {  // [13]: some query has a lo plane (read back with the grouping's counts); [14]: the rank kernel's
    VI_TRY(ws.stats.reserve(160));  // work counter — both reset by item_cols_kernel after their use
    VI_HIP(hipMemsetAsync(ws.stats.p, 0, 160 * sizeof(uint64_t), st));
    ws.stats_zeroed = true;
  }
  const uint64_t nt = (uint64_t)nq * nc * 2;
  hipLaunchKernelGGL(split_queries_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, st, Qd, (uint32_t)nq, ix.dim, nc,
                     (uint4 *)ws.qimg.p, (unsigned long long *)(ws.stats.p + 13), zero, (uint32_t)zero_words,
                     ix.centered ? (const float *)ix.centre.p : nullptr);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

// coarse step alone on the matrix cores (probe export for other ranks): fills ws.probes / ws.gorder
vi_status coarse_only_filter(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st) {
  SearchWorkspace &ws = ix.cur().ws;
  VI_TRY(build_query_image(ix, Qd, nq, st));
  VI_TRY(ws.pair_rel.reserve(nq * P));
  VI_TRY(ws.qtot.reserve(nq));
  return stage_coarse_filter(ix, Qd, nq, P, list_segb0(), st);
}

vi_status search_filter_pipeline(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P, uint32_t K,
                                 float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts, hipStream_t st,
                                 int timing_level, const uint32_t *probes_in, const uint32_t *order_in) {
  SearchWorkspace &ws = ix.cur().ws;
  vi_search_stats &stt = ix.cur().stats;
  // 1: an event at every phase boundary; 2: around the rank kernel only (every record is a barrier packet the next
  // kernel's dispatch waits behind: five of them cost 0.01 ms of a 0.5 ms step)
  const bool timing = timing_level == 1, rank_timing = timing_level != 0;
  const uint32_t dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  (void)K;
  const uint32_t segb0 = list_segb0();
  VI_TRY(ws.pair_rel.reserve(nq * P));
  VI_TRY(ws.qtot.reserve(nq));
  VI_TRY(ws.qoff.reserve(nq + 1));
  VI_TRY(ws.stats.reserve(160));
  if (getenv("VI_FILTER_STATS")) {
    VI_HIP(hipMemsetAsync(ws.stats.p + 6, 0, 6 * sizeof(uint64_t), st));
    VI_HIP(hipMemsetAsync(ws.stats.p + 150, 0, 8 * sizeof(uint64_t), st));
  }
  ws.pair_rank_valid = false;  // (set by the direct coarse select of THIS search)
  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[0], st));
  // the batch's queries as MFMA operands: -2 q split into bf16 hi / lo once (a query sits in n_probe work items)
  const char *cf = getenv("VI_COARSE_FILTER");
  const bool coarse_mfma = !probes_in && !(cf && *cf == '0') && nq >= 256 && nlists >= 1024 && ix.dim <= kNarrowDim && rank_bf16();
  if (coarse_mfma) {  // (its per-list histogram is cleared by the kernel that splits the queries)
    VI_TRY(ws.cnt.reserve(2 * subbin_words(nlists)));
    VI_TRY(build_query_image(ix, Qd, nq, st, ws.cnt.p, subbin_words(nlists)));
  } else {
    VI_TRY(build_query_image(ix, Qd, nq, st));
  }
  // ---- 1. coarse quantizer: probes, shard visiting order, per-list histogram, record offsets ----
  {
    if (coarse_mfma) {
      VI_TRY(stage_coarse_filter(ix, Qd, nq, P, segb0, st, true));
    } else if (!probes_in && !(cf && *cf == '0') && nq >= 256 && nlists >= 1024 && ix.dim <= kNarrowDim) {
      VI_TRY(stage_coarse_filter(ix, Qd, nq, P, segb0, st));
    } else {
      if (probes_in) VI_TRY(adopt_probes(ix, nq, P, probes_in, order_in, true, st));
      else VI_TRY(stage_coarse(ix, Qd, nq, P, st));
      hipLaunchKernelGGL(pair_groups_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.probes.p,
                         ix.list_len.p, (uint32_t)nq, P, segb0, ws.pair_rel.p, ws.qtot.p);
    }
    // (the grouping's scan kernel scans the record offsets too, as a second workgroup)
    if (!grouping_fuses_query_offsets(ix)) hipLaunchKernelGGL(query_offsets_kernel, dim3(1), dim3(1024), 0, st, ws.qtot.p, (uint32_t)nq, ws.qoff.p);
    VI_HIP(hipGetLastError());
  }
  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[1], st));
  // ---- 2. group all (query, probe) pairs by list ----
  uint64_t hstats[14];
  // queries per rank work item: 128 when lists are shared by many queries of the batch, 32 when a list is probed by a
  // handful (large balanced indexes): a 128-query group would keep three of its four waves idle
  // The choice needs the batch's histogram, which only the grouping produces: the first batch of a shape (nq, P) goes by
  // the mean (queries per list), every later one by what the previous batch of that shape measured — the fill a
  // 128-query grouping has (pairs per tile slot; the grouping counts its tiles whichever size runs).  The mean alone is
  // wrong on skewed indexes: the reference's k-means on unclustered data leaves a few enormous lists that every query
  // probes (C5-shaped run: 4.9 queries per list on average, yet 128-query groups are three quarters full).
  const char *gqe = getenv("VI_FILTER_GQ");
  uint32_t gq = (double)nq * P / (double)std::max<uint64_t>(1, nlists) >= 24.0 ? 128u : 32u;
  for (const auto &h : ws.gq_hint)
    if (h.nq == nq && h.P == P) gq = h.gq;
  if (gqe) gq = atoi(gqe) == 32 ? 32u : 128u;
  if (ix.dim > kNarrowDim) gq = 128u;  // the wide kernel's C tile holds 128 queries
  // D <= 128, stored values bf16-exact (hi planes only): the streaming kernel (rank_stream.hip).  A work item holds up to
  // 128 queries and costs MFMAs for its live 32-query tiles only, so there is no group size to choose (VI_STREAM_GQ=256:
  // groups of 256 when the queries are bf16-exact too — measured equal).  VI_RANK_STREAM=0: block-synchronous kernel.
  // (bf16 x 3 keeps the block-synchronous kernel: two tiles of hi + lo planes do not fit in the streaming kernel's registers)
  // (measured at D = 32 / 64 / 96 / 128: its helper kernels and item skeleton pay off from 7 chunks of 16 dimensions on;
  // VI_RANK_STREAM=1 forces it for any D <= 128)
  const char *se = getenv("VI_RANK_STREAM");
  // real-valued lists: hi planes only + a wider margin (rank_approx_mode) — the streaming kernel serves them too
  const int approx = (rank_bf16() && !ix.lists_lo_zero && hi_only_ok() && ix.dim <= kNarrowDim) ? rank_approx_mode(ix) : 0;
  const bool hi_lists = (ix.lists_lo_zero && hi_only_ok()) || approx != 0;
  const bool stream = ix.dim <= kNarrowDim && rank_bf16() && hi_lists && !(se && *se == '0') &&
                      (dq / 4 >= 7 || (se && *se == '1') || (approx != 0 && dq / 4 >= 5));
  if (stream) {
    const char *e = getenv("VI_STREAM_GQ");
    gq = e && atoi(e) == 256 && ws.queries_hi_only ? 256u : 128u;
  }
  const bool fuse_q = grouping_fuses_query_offsets(ix);
  VI_TRY(launch_grouping(ix, ws.probes.p, nq, P, (int)gq, segb0, hstats, st, true, fuse_q ? ws.qtot.p : nullptr, fuse_q ? ws.qoff.p : nullptr,
                         ws.pair_rank_valid ? ws.pair_rank.p : nullptr));
  {
    const double fill128 = hstats[12] ? (double)hstats[0] / ((double)hstats[12] * 128.0 * 64.0) : 0.0;
    const uint32_t next = fill128 >= 0.3 ? 128u : 32u;
    bool seen = false;
    for (auto &h : ws.gq_hint)
      if (h.nq == nq && h.P == P) { h.gq = next; seen = true; }
    if (!seen) {
      if (ws.gq_hint.size() >= 64) ws.gq_hint.clear();
      ws.gq_hint.push_back({nq, P, next});
    }
  }
  ws.queries_hi_only = hstats[13] == 0;
  stt.scanned_vectors = hstats[0];
  stt.scan_items = hstats[1];
  stt.filter_tile_blocks = hstats[3];
  const uint64_t nrec = hstats[4], nbrec = hstats[5] * 2 * gq;  // pair records: 2 x gq per (query group, segment, 2 blocks)
  if (nrec >= (1ull << 31) || nbrec >= (1ull << 32)) return fail(VI_ERR_INVALID_INPUT, "batch too large: split nq");
  VI_TRY(ws.gval.reserve(std::max<uint64_t>(1, nrec) * 4));
  VI_TRY(ws.gpos.reserve(std::max<uint64_t>(1, nrec)));
  VI_TRY(ws.brec.reserve(std::max<uint64_t>(1, nbrec) * 4));
  // (the phase clock of the rank kernel starts right in front of it: the work-item helper kernels count as grouping)
  bool rank_clock_started = false;
  auto start_rank_clock = [&]() -> vi_status {
    if (rank_timing && !rank_clock_started) VI_HIP(hipEventRecord(ix.cur().ev[2], st));
    rank_clock_started = true;
    return VI_OK;
  };
  // ---- 3. rank on the matrix cores ----
  if (ix.dim > kNarrowDim) {
    const uint32_t nc = dq / 4;
    const uint32_t nitems = (uint32_t)hstats[1];
    VI_TRY(ws.item_list.reserve(std::max<uint32_t>(1, nitems)));
    if (nitems) {
      hipLaunchKernelGGL(item_list_kernel, dim3((nitems + 255) / 256), dim3(256), 0, st, ws.item_start.p, (uint32_t)nlists, nitems,
                         ws.item_list.p);
      WideArgs a{(const uint4 *)ix.lists_bf16.p, ix.xnorm_img.p, (const uint4 *)ws.qimg.p, nc, ix.list_first_block.p, ix.list_len.p,
                 ws.item_start.p, ws.seg_start.p, ws.pairs.p, ws.item_list.p, P, segb0, ws.qoff.p, ws.pair_rel.p, ws.tile_start.p,
                 (float4 *)ws.gval.p, ws.gpos.p, (float4 *)ws.brec.p};
      // (set on the device that launches, every time: a process may hold indexes on several GPUs)
      if (hipFuncSetAttribute((const void *)rank_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kWideLdsFloats * (int)sizeof(float)) != hipSuccess)
        return fail(VI_ERR_DEVICE, "cannot reserve %d bytes of LDS for the wide rank kernel", kWideLdsFloats * 4);
      VI_TRY(start_rank_clock());
      hipLaunchKernelGGL(rank_wide_kernel, dim3(nitems), dim3(256), kWideLdsFloats * sizeof(float), st, a);
    }
    VI_HIP(hipGetLastError());
    stt.rank_mode = 2;
    stt.group_queries = gq;
  } else {
    const uint32_t nitems = (uint32_t)hstats[1];
    VI_TRY(ws.items.reserve(std::max<uint32_t>(1, nitems) * 8ull));
    if (nitems) {
      hipLaunchKernelGGL(item_desc_kernel, dim3((nitems + 255) / 256), dim3(256), 0, st, ws.item_start.p, ws.seg_start.p,
                         ix.list_len.p, ix.list_first_block.p, ws.tile_start.p, (uint32_t)nlists, nitems, segb0, gq,
                         item_run(), (uint4 *)ws.items.p);
      VI_HIP(hipGetLastError());
    }
    const int rank_mode = rank_bf16() ? (hi_lists ? 2 : 1) : 0;
    stt.rank_mode = approx ? 4u : (uint64_t)rank_mode + 1;
    if (rank_bf16() && ix.centered && (stt.rank_mode == 2 || stt.rank_mode == 4)) stt.rank_mode = stt.rank_mode == 2 ? 5 : 6;  // the same about the mean
    stt.group_queries = gq;
    if (stream) {
      // queries in LDS, vectors through registers, no barrier in the block loop, persistent workgroups (rank_stream.hip)
      VI_TRY(ws.item_qcol.reserve(std::max<uint64_t>(1, (uint64_t)nitems * gq)));
      VI_TRY(ws.item_grec.reserve(std::max<uint64_t>(1, (uint64_t)nitems * gq)));
      VI_TRY(ws.item_sdesc.reserve(std::max<uint64_t>(1, (uint64_t)nitems * 4)));
      if (nitems) {
        hipLaunchKernelGGL(item_cols_kernel, dim3(nitems), dim3(128), 0, st, (const uint4 *)ws.items.p, ws.pairs.p, ws.qoff.p,
                           ws.pair_rel.p, P, gq, ws.item_qcol.p, ws.item_grec.p, (uint4 *)ws.item_sdesc.p, ws.gpos.p, ws.stats.p);
        VI_HIP(hipGetLastError());
      }
      RankStreamArgs a{(const uint4 *)ix.lists_bf16.p, ix.xnorm_img.p, (const uint4 *)ws.qimg.p, (const uint4 *)ws.item_sdesc.p, nitems,
                       ws.item_qcol.p, ws.item_grec.p, (uint32_t *)(ws.stats.p + 16), (float4 *)ws.gval.p,
                       (float4 *)ws.brec.p, nullptr, env_xmode()};
      const bool qlo = (hstats[13] != 0 || !hi_only_ok()) && approx != 2;
      const bool prof = getenv("VI_STREAM_PROF") != nullptr;
      if (prof) {
        VI_TRY(ws.prof.reserve(32 + 4 * 1024));
        VI_HIP(hipMemsetAsync(ws.prof.p, 0, (32 + 4 * 1024) * sizeof(uint64_t), st));
        VI_HIP(hipMemsetAsync(ws.prof.p + 16, 0xFF, sizeof(uint64_t), st));
        a.prof = (unsigned long long *)ws.prof.p;
      }
      VI_TRY(start_rank_clock());
      VI_TRY(launch_rank_stream(a, dq / 4, nitems, rank_mode, qlo || rank_mode == 1, gq, st));
      if (prof) {
        uint64_t h[24];
        VI_HIP(hipMemcpyAsync(h, ws.prof.p, sizeof(h), hipMemcpyDeviceToHost, st));
        VI_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "rank_stream wave-0 ticks (100 MHz) summed over workgroups: multiply %llu (of which waiting for tiles %llu) "
                "end-of-item wait %llu gather %llu merge %llu | items %llu steps %llu | loop total %llu\n",
                (unsigned long long)h[0], (unsigned long long)h[6], (unsigned long long)h[1], (unsigned long long)h[2],
                (unsigned long long)h[3], (unsigned long long)h[4], (unsigned long long)h[5], (unsigned long long)h[7]);
        if (const char *dump = getenv("VI_STREAM_PROF_DUMP")) {
          std::vector<uint64_t> w(4 * 1024);
          VI_HIP(hipMemcpy(w.data(), ws.prof.p + 32, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
          if (FILE *f = fopen(dump, "w")) {
            uint64_t base = ~0ull;
            for (int i = 0; i < 1024; ++i) if (w[4 * i + 1]) base = std::min(base, w[4 * i]);
            for (int i = 0; i < 1024; ++i)  // workgroup, start, end (10 ns ticks after the first start), items
              if (w[4 * i + 1]) fprintf(f, "%d %llu %llu %llu\n", i, (unsigned long long)(w[4 * i] - base), (unsigned long long)(w[4 * i + 1] - base),
                                        (unsigned long long)w[4 * i + 2]);
            fclose(f);
          }
        }
        fprintf(stderr, "   loops entered over %llu ticks, last exit %llu ticks after the first entry\n", (unsigned long long)(h[17] - h[16]),
                (unsigned long long)(h[18] - h[16]));
        fprintf(stderr, "   longest workgroup loop %llu ticks, workgroups with items %llu, most items in one %llu\n", (unsigned long long)h[13],
                (unsigned long long)h[14], (unsigned long long)h[15]);
        fprintf(stderr, "   after B1: T+idx %llu, DMA issue %llu, vmcnt(0) %llu, B2 %llu | after B2: merge+stores %llu, idx read %llu, load_item %llu\n",
                (unsigned long long)h[8], (unsigned long long)h[9], (unsigned long long)h[10], (unsigned long long)h[2],
                (unsigned long long)h[11], (unsigned long long)h[12], (unsigned long long)h[3]);
      }
    } else {
      FilterArgs a{};
      a.blocks = rank_bf16() ? (const float4 *)ix.lists_bf16.p : (const float4 *)ix.lists.blocks.p;
      a.xnorm = rank_bf16() ? ix.xnorm_img.p : ix.xnorm.p; a.dq = dq; a.dim = ix.dim; a.Q = Qd;
      a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
      a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = P; a.segb0 = segb0;
      a.qoff = ws.qoff.p; a.rel = ws.pair_rel.p; a.rec_stride = 0;
      a.tile_start = ws.tile_start.p;
      a.item_list = nullptr;
      a.items = (const uint4 *)ws.items.p;
      a.gval = (float4 *)ws.gval.p; a.gmeta = ws.gpos.p; a.brec = (float4 *)ws.brec.p;
      a.xmode = env_xmode();
      a.qimg = rank_bf16() ? (const uint4 *)ws.qimg.p : nullptr;
      VI_TRY(start_rank_clock());
      VI_TRY(launch_filter(a, dq, nitems, rank_mode, gq, st));
    }
  }
  VI_TRY(start_rank_clock());  // (nothing to rank)
  if (rank_timing) VI_HIP(hipEventRecord(ix.cur().ev[3], st));
  // ---- 4. select ----
  {
    SelectArgs a{select_common(ix, Qd, (const float4 *)ix.lists.blocks.p, rank_bf16() && ix.centered ? ix.xmax2_c : ix.xmax2, gq, stream, approx), (uint32_t)nq, P, (uint32_t)k, segb0,
                 ws.qoff.p, ws.qtot.p, ws.pair_rel.p, ws.pair_pos.p, ws.tile_start.p, ws.probes.p, ws.gorder.p, ix.list_first_block.p, ix.list_len.p,
                 ix.ext_ids.p, Dd, Id, Td, slots, counts};
    { const char *e = getenv("VI_FILTER_STATS"); if (e && *e == '2') a.c.dbg = nullptr; }
    { const char *e = getenv("VI_EXACT_BF16"); if (ix.lists_hi_nat.p && !(e && *e == '0')) a.c.hi_nat = (const uint4 *)ix.lists_hi_nat.p; }
    { const char *e = getenv("VI_EXACT_U8"); if (ix.lists_u8_nat.p && !(e && *e == '0')) a.c.u8_nat = (const uint4 *)ix.lists_u8_nat.p; }
    const size_t qsm = 4ull * ix.dim * sizeof(float);
    if (k <= 64) hipLaunchKernelGGL(select_kernel<FastTopK>, dim3((uint32_t)((nq + 3) / 4)), dim3(256), qsm, st, a);
    else hipLaunchKernelGGL(select_kernel<FastTop128>, dim3((uint32_t)((nq + 3) / 4)), dim3(256), qsm, st, a);
    VI_HIP(hipGetLastError());
  }
  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[4], st));
  if (timing && getenv("VI_FILTER_STATS")) {
    uint64_t dbg[12], tks[8];
    VI_HIP(hipMemcpyAsync(dbg, ws.stats.p, sizeof(dbg), hipMemcpyDeviceToHost, st));
    VI_HIP(hipMemcpyAsync(tks, ws.stats.p + 150, sizeof(tks), hipMemcpyDeviceToHost, st));
    VI_HIP(hipStreamSynchronize(st));
    if (const char *e = getenv("VI_FILTER_STATS"); e && *e == '2')
      fprintf(stderr, "coarse select ticks (every 64th query): query row %llu, records + bound %llu, flags (+ rounds a full list forces) %llu, exact rounds %llu, tail %llu; rows %llu "
              "of which in whole sub-blocks %llu\n", (unsigned long long)tks[0], (unsigned long long)tks[1], (unsigned long long)tks[2],
              (unsigned long long)tks[3], (unsigned long long)tks[4], (unsigned long long)dbg[6], 8ull * (unsigned long long)dbg[7]);
    if (const char *e = getenv("VI_FILTER_STATS"); e && (*e == '3' || *e == '4'))
      fprintf(stderr, "select ticks: records -> LDS %llu, threshold %llu, refinement %llu, scan of pair records (+ exact rounds it triggers) %llu, "
              "last exact rounds %llu\n", (unsigned long long)tks[0], (unsigned long long)tks[1], (unsigned long long)tks[2],
              (unsigned long long)tks[3], (unsigned long long)tks[4]);
    stt.filter_rechecked = dbg[6]; stt.filter_accepted = dbg[7];
    if (const char *e = getenv("VI_FILTER_STATS"); e && (*e == '3' || *e == '4'))
      fprintf(stderr, "select stats: exact %llu groups_scanned %llu queries_with_full_group %llu full_groups %llu sub_blocks %llu\n",
              (unsigned long long)dbg[6], (unsigned long long)dbg[7], (unsigned long long)dbg[8], (unsigned long long)dbg[9],
              (unsigned long long)dbg[10]);
  }
  return VI_OK;
}

}  // namespace vi
