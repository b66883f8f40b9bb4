// filter_search.hip — list scan on the matrix cores: f32-MFMA filter + exact-order re-check.
//
// The exact-order VALU scan (search_kernels.hip) spends 3 vector ops per (query, vector, dim) and is
// bound by f32 VALU issue.  When many queries of a batch probe the same list, (queries x vectors x
// dims) is GEMM shaped, so the bulk of the candidates can be REJECTED on the matrix cores and only
// the few survivors need the reference's exact arithmetic:
//
//   1. bound     tau_q = k-th smallest EXACT distance among the first 512 vectors of the query's
//                nearest list (existing scan kernel, max_blocks = 8).  Any k exact candidates
//                bound the final k-th distance from above, so every true result has d_ref <= tau_q.
//   2. filter    per (list, tile of 32 queries): m(q,v) = ||v||^2 - 2 q.v with
//                v_mfma_f32_32x32x2_f32 (A = 32 vectors straight from the lane-interleaved blocks,
//                B = the tile's queries held in registers, accumulator initialised with ||v||^2).
//                (q,v) survives iff m <= thr_q, where
//                    thr_q = tau_q (1 + 2 gamma) + E_q - ||q||^2 (lower bound)
//                    gamma = (D+2) u'            rounding of the reference's sequential sum
//                    E_q   = (D+2) u' (||q||^2 + 2 max ||v||^2)   rounding of the fma chain + norms
//                so d_ref(q,v) <= tau_q  ==>  m(q,v) <= thr_q  (no true result is ever rejected).
//   3. re-check  survivors are compacted across the tile (one lane per (q,v) pair, 64 pairs per
//                pass), their distance is recomputed in the reference's exact order and those with
//                d_ref <= tau_q are appended to the query's candidate list.
//   4. select    one wave per query: wave-resident top-k over its candidates with the reference's
//                stable order (dist, shard-visit order, position) -> D, I, tie.
//
// Queries whose bound is infinite (nearest list shorter than k) or whose candidate list overflows
// go through the exact VALU pipeline instead; results are bit-identical on both paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_index.hpp"
#include "device_math.hpp"
#include "scan.hpp"
#include "wave_select.hpp"

namespace vi {

// provided by search_kernels.hip
vi_status stage_coarse(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st);
vi_status search_valu_pipeline(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P, uint32_t K,
                               float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts, hipStream_t st,
                               bool timing);
vi_status launch_grouping(const DeviceIndex &ix, const uint32_t *probes, uint64_t nq, uint32_t P, int qg, uint32_t segb0,
                          uint64_t hstats[3], hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;
constexpr int kGroupQ = 128;            // queries per work item: 4 waves x one MFMA column tile of 32
constexpr uint32_t kSampleBlocks = 8;   // blocks of the nearest list sampled for the bound
constexpr uint32_t kCap = 16384;        // candidate slots per query
constexpr uint32_t kSampleRanks = 2;    // nearest lists sampled for the bound (<= 4)
constexpr uint32_t kPosBits = 26;       // candidate key = (probe rank << 26) | position in list

__global__ void slot_norms_kernel(const float4 *blocks, uint32_t dq, uint64_t nslots, const uint64_t *ext_ids,
                                  float *xnorm, uint32_t *xmax_bits) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  float out = INFINITY;  // pad slots never pass the filter
  if (!ext_ids || ext_ids[s] != ~0ull) {
    double acc = 0.0;
    const float4 *p = blocks + (s / kWave) * dq * kWave + (s % kWave);
    for (uint32_t qd = 0; qd < dq; ++qd) {
      const float4 v = p[(size_t)qd * kWave];
      acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    out = (float)acc;
    if (out < INFINITY) atomicMax(xmax_bits, __float_as_uint(out));
  }
  xnorm[s] = out;
}

__global__ void take_first_ranks_kernel(const uint32_t *probes, uint32_t nq, uint32_t P, uint32_t R0,
                                        uint32_t *probes0) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nq * R0) probes0[t] = probes[(size_t)(t / R0) * P + (t % R0)];
}

// tau_q = k-th smallest exact distance over the union of the query's R0 sampled runs (each sorted, K entries)
__global__ void tau_kernel(const float *run_dist, const uint32_t *run_pos, uint32_t nq, uint32_t R0, uint32_t K,
                           uint32_t k, float *tau) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t head[4] = {0, 0, 0, 0};
  float last = INFINITY;
  for (uint32_t i = 0; i < k; ++i) {
    float best = INFINITY;
    int br = -1;
    for (uint32_t r = 0; r < R0; ++r) {
      const size_t o = ((size_t)q * R0 + r) * K + head[r];
      if (head[r] < K && run_pos[o] != kNoPos && (br < 0 || run_dist[o] < best)) { best = run_dist[o]; br = (int)r; }
    }
    if (br < 0) { last = INFINITY; break; }
    head[br]++;
    last = best;
  }
  tau[q] = last;
}

struct FilterArgs {
  const float4 *blocks;
  const float *xnorm;
  uint32_t dq, dim;
  const float *Q;
  const uint32_t *first_block, *list_len, *item_start, *seg_start, *pairs;
  uint32_t nlists, P, segb0;
  const float *tau;
  float gamma2, e_scale, xmax2;
  uint32_t cap;
  uint32_t *cand_cnt;
  float *cand_dist;
  uint32_t *cand_key;
  unsigned long long *dbg;  // [4]=pairs re-checked, [5]=pairs accepted
  uint32_t xmode;           // experiment knob (VI_FILTER_XMODE): 1 = do not restage tiles, 2 = no compaction, 4 = no barriers
};

// Workgroup = 4 waves = up to 128 queries (4 column tiles of 32) probing ONE list (segment).  Every
// 64-vector block of the list is staged once per workgroup into LDS as 64 rows of dq*4 floats (row stride
// 132 floats => conflict-free ds_read_b128 / ds_write_b128) together with the 64 squared norms, and is
// consumed by all four waves; the next block's global loads are issued before the MFMAs of the current
// one and written to LDS after them (issue-early / write-late), as in assign_mfma.hip.
constexpr int kRowStride = 132;
constexpr int kTileFloats = 64 * kRowStride + 64;

// block staging: 64 vectors x 2*NG quads over 256 threads => NG/2 float4 each (+ one norm for threads < 64)
template <int NG>
struct StageRegs {
  static constexpr int kPerThread = (64 * 2 * NG) / 256;  // NG even => integral
  float4 v[kPerThread];
  float norm;
};

template <int NG>
__device__ __forceinline__ void stage_load(StageRegs<NG> &s, const float4 *src, const float *xn, bool live) {
#pragma unroll
  for (int i = 0; i < StageRegs<NG>::kPerThread; ++i) {
    s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) s.v[i] = src[threadIdx.x + 256 * i];  // idx = quad*64 + vector: fully coalesced
  }
  s.norm = INFINITY;
  if (live && threadIdx.x < 64) s.norm = xn[threadIdx.x];
}

template <int NG>
__device__ __forceinline__ void stage_write(const StageRegs<NG> &s, float *tile) {
#pragma unroll
  for (int i = 0; i < StageRegs<NG>::kPerThread; ++i) {
    const int idx = threadIdx.x + 256 * i;
    *reinterpret_cast<float4 *>(tile + (idx & 63) * kRowStride + (idx >> 6) * 4) = s.v[i];
  }
  if (threadIdx.x < 64) tile[64 * kRowStride + threadIdx.x] = s.norm;
}

template <int NG>  // NG = dq/2 exactly: a block holds 2*NG quads (dims padded to 16); dim % 4 == 0
__global__ void __launch_bounds__(256, 2) filter_kernel(FilterArgs a) {
  __shared__ float s_tile[kTileFloats];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const uint32_t item = blockIdx.x;  // grid == number of items
  uint32_t lo = 0, hi = a.nlists;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.item_start[mid] <= item) lo = mid; else hi = mid;
  }
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
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

  // ---- this lane's query: wave w owns queries 32w .. 32w+31 of the group; both lane halves hold query j ----
  const uint32_t jq_grp = 32u * wave + (uint32_t)j;
  const bool qlive = jq_grp < nqi;
  const bool wave_live = 32u * wave < nqi;  // wave-uniform
  const uint32_t slot = qlive ? a.pairs[s0 + j0 + jq_grp] : 0u;
  const uint32_t qid = slot / a.P;
  const float *qrow = a.Q + (size_t)qid * a.dim;
  float4 qf[NG];
  float qn = 0.0f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const uint32_t e = 8 * g + 4 * h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qlive && e < a.dim) v = *reinterpret_cast<const float4 *>(qrow + e);
    qn += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    qf[g] = make_float4(-2.f * v.x, -2.f * v.y, -2.f * v.z, -2.f * v.w);
  }
  qn += __shfl_xor(qn, 32);
  const float tau = qlive ? a.tau[qid] : -INFINITY;
  // thr on m = ||v||^2 - 2 q.v ; an infinite bound means "handled by the exact pipeline": reject all
  float thr = -INFINITY;
  if (qlive && tau < INFINITY) {
    const float qn_hi = qn * (1.0f + a.gamma2), qn_lo = qn * (1.0f - a.gamma2);
    thr = (tau * (1.0f + a.gamma2) + a.e_scale * (qn_hi + 2.0f * a.xmax2)) * 1.0001f - qn_lo;
  }

  // survivors are appended straight to their query's candidate list as (m, key): m = ||v||^2 - 2 q.v is the
  // MFMA value; the select kernel decides which of them need the exact arithmetic
  const uint32_t rkey = (slot - qid * a.P) << kPosBits;
  auto emit = [&](float m, uint32_t pos) {
    const uint32_t idx = atomicAdd(&a.cand_cnt[qid], 1u);
    if (idx < a.cap) {
      a.cand_dist[(size_t)qid * a.cap + idx] = m;
      a.cand_key[(size_t)qid * a.cap + idx] = rkey | pos;
    }
  };

  StageRegs<NG> stage;
  stage_load<NG>(stage, a.blocks + ((size_t)(fb + b0) * a.dq) * kWave, a.xnorm + (size_t)(fb + b0) * kWave, b0 < b1);
  stage_write<NG>(stage, s_tile);
  __syncthreads();
  for (uint32_t blk = b0; blk < b1; ++blk) {
    const bool more = (blk + 1 < b1) && !(a.xmode & 1u);
    // next block: in flight during this block's MFMAs
    stage_load<NG>(stage, a.blocks + ((size_t)(fb + blk + 1) * a.dq) * kWave, a.xnorm + (size_t)(fb + blk + 1) * kWave, more);
    if (wave_live) {
      // both row tiles (vectors 0..31 and 32..63) advance together: two independent accumulator chains
      f32x16 acc0, acc1;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {  // rows 8*q4 + 4*h + (0..3) live in regs 4*q4 .. 4*q4+3
        const float4 n0 = *reinterpret_cast<const float4 *>(s_tile + 64 * kRowStride + 8 * q4 + 4 * h);
        const float4 n1 = *reinterpret_cast<const float4 *>(s_tile + 64 * kRowStride + 32 + 8 * q4 + 4 * h);
        acc0[4 * q4 + 0] = n0.x; acc0[4 * q4 + 1] = n0.y; acc0[4 * q4 + 2] = n0.z; acc0[4 * q4 + 3] = n0.w;
        acc1[4 * q4 + 0] = n1.x; acc1[4 * q4 + 1] = n1.y; acc1[4 * q4 + 2] = n1.z; acc1[4 * q4 + 3] = n1.w;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float4 a0 = *reinterpret_cast<const float4 *>(s_tile + j * kRowStride + 8 * g + 4 * h);
        const float4 a1 = *reinterpret_cast<const float4 *>(s_tile + (32 + j) * kRowStride + 8 * g + 4 * h);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, qf[g].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, qf[g].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, qf[g].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, qf[g].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, qf[g].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, qf[g].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, qf[g].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, qf[g].w, acc1, 0, 0, 0);
      }
      if (!(a.xmode & 2u)) {
        const uint32_t pbase = blk * kWave + 4u * (uint32_t)h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // reg r <-> vector row (r&3) + 8*(r>>2) + 4*h of its 32-row tile
          if (acc0[r] <= thr) emit(acc0[r], pbase + (r & 3) + 8 * (r >> 2));
          if (acc1[r] <= thr) emit(acc1[r], pbase + 32u + (r & 3) + 8 * (r >> 2));
        }
      }
    }
    if (!(a.xmode & 4u)) __syncthreads();  // every wave is done reading the tile
    if (more) stage_write<NG>(stage, s_tile);
    if (!(a.xmode & 4u)) __syncthreads();  // next tile visible
  }
}


// ------------------------------------------------------------------------------------------
// select: which survivors need the reference's exact arithmetic, and the final top-K
// ------------------------------------------------------------------------------------------
// A query's candidate list holds (m, key) with m = ||v||^2 - 2 q.v from the MFMA.  Let m_K be the K-th
// smallest m.  The K candidates with the smallest m have d_ref <= (m_K + ||q||^2 + E)(1 + gamma), so the K-th
// smallest d_ref is at most that, and every candidate of the true top-K satisfies
//     m <= m_K + 2E + 3 gamma (m_K + ||q||^2 + E)                                   (*)
// Only candidates passing (*) (K plus a handful) are re-evaluated in exact order; the top-K of those exact
// distances under the reference's stable order is the answer.
struct SelectCommon {
  const float *Q;
  uint32_t dim, dq, cap;
  const uint32_t *cand_cnt, *cand_key;
  const float *cand_dist;
  float gamma, e_scale, xmax2;
};

// exact distance of one (query row, stored vector) pair, one lane per pair (src/utils.rs:28-30)
__device__ __forceinline__ float exact_pair(const float *qrow, const float4 *xv, uint32_t dim) {
  const float4 *xq = reinterpret_cast<const float4 *>(qrow);
  float acc = 0.0f;
  const uint32_t nquad = dim >> 2;
  uint32_t qd = 0;
  for (; qd + 4 <= nquad; qd += 4) {  // 8 independent 16-byte loads in flight per lane
    const float4 q0 = xq[qd], q1 = xq[qd + 1], q2 = xq[qd + 2], q3 = xq[qd + 3];
    const float4 x0 = xv[(size_t)qd * kWave], x1 = xv[(size_t)(qd + 1) * kWave];
    const float4 x2 = xv[(size_t)(qd + 2) * kWave], x3 = xv[(size_t)(qd + 3) * kWave];
    sq_add(acc, q0.x, x0.x); sq_add(acc, q0.y, x0.y); sq_add(acc, q0.z, x0.z); sq_add(acc, q0.w, x0.w);
    sq_add(acc, q1.x, x1.x); sq_add(acc, q1.y, x1.y); sq_add(acc, q1.z, x1.z); sq_add(acc, q1.w, x1.w);
    sq_add(acc, q2.x, x2.x); sq_add(acc, q2.y, x2.y); sq_add(acc, q2.z, x2.z); sq_add(acc, q2.w, x2.w);
    sq_add(acc, q3.x, x3.x); sq_add(acc, q3.y, x3.y); sq_add(acc, q3.z, x3.z); sq_add(acc, q3.w, x3.w);
  }
  for (; qd < nquad; ++qd) {
    const float4 qq = xq[qd];
    const float4 xx = xv[(size_t)qd * kWave];
    sq_add(acc, qq.x, xx.x); sq_add(acc, qq.y, xx.y); sq_add(acc, qq.z, xx.z); sq_add(acc, qq.w, xx.w);
  }
  return acc;
}

// stage 1 of the select: threshold (*) on m for query q with n candidates (wave-uniform result)
__device__ __forceinline__ float select_threshold(const SelectCommon &c, uint32_t q, uint32_t n, uint32_t K, int lane) {
  WaveTopK s1;
  s1.init();
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t i = base + lane;
    const bool live = i < n;
    s1.offer(live ? c.cand_dist[(size_t)q * c.cap + i] : INFINITY, live ? i : kNoPos, (int)K);
  }
  if (n < K) return INFINITY;  // fewer candidates than wanted: all of them are results
  const float mk = readlane_f(s1.d, (int)K - 1);
  float qn = 0.0f;
  for (uint32_t e = lane; e < c.dim; e += kWave) { const float v = c.Q[(size_t)q * c.dim + e]; qn += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  const float E = c.e_scale * (qn * (1.0f + c.gamma) + 2.0f * c.xmax2);
  const float scale = fmaxf(mk + qn, 0.0f) + E;
  return mk + (2.0f * E + 3.0f * c.gamma * scale) * 1.001f + 1e-30f;
}

struct SelectArgs {
  SelectCommon c;
  uint32_t nq, P, k;
  const float4 *blocks;
  const uint32_t *probes, *gorder, *first_block;
  const float *tau;
  const uint64_t *ext_ids;
  float *D;
  int64_t *I;
  uint64_t *tie, *slots;
  uint32_t *counts;
  uint8_t *fallback;
};

// one wave per query: top-k of its candidates in the reference's stable order
__global__ void __launch_bounds__(256) select_kernel(SelectArgs a) {
  __shared__ uint32_t s_pick[4][128];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;
  const SelectCommon &c = a.c;
  const uint32_t n = c.cand_cnt[q];
  const bool noinf = a.tau[q] < INFINITY;
  const bool fb = !noinf || n > c.cap;
  if (lane == 0) a.fallback[q] = !noinf ? 1 : (n > c.cap ? 2 : 0);
  if (fb) return;
  const uint32_t g_of_r = (uint32_t)lane < a.P ? a.gorder[(size_t)q * a.P + lane] : kNoPos;
  const uint32_t list_of_r = (uint32_t)lane < a.P ? a.probes[(size_t)q * a.P + lane] : 0u;
  const uint32_t fb_of_r = (uint32_t)lane < a.P && list_of_r != kNoPos ? a.first_block[list_of_r] : 0u;
  const float thr2 = select_threshold(c, q, n, a.k, lane);
  WaveTopK sel;
  sel.init();
  const int K = (int)a.k;
  const float *qrow = c.Q + (size_t)q * c.dim;
  uint32_t *pick = s_pick[wave];
  uint32_t npick = 0;
  auto flush = [&](uint32_t off, uint32_t cnt) {  // exact distances of pick[off..off+cnt), one lane each
    const bool live = (uint32_t)lane < cnt;
    const uint32_t ck = live ? pick[off + lane] : 0u;
    const uint32_t r = ck >> kPosBits, pos = ck & ((1u << kPosBits) - 1u);
    const uint32_t g = (uint32_t)__shfl((int)g_of_r, (int)r);
    const uint32_t fbk = (uint32_t)__shfl((int)fb_of_r, (int)r);
    float d = INFINITY;
    if (live) d = exact_pair(qrow, a.blocks + ((size_t)(fbk + pos / kWave) * c.dq) * kWave + (pos % kWave), c.dim);
    sel.offer(d, live ? ((g << kPosBits) | pos) : kNoPos, K);
  };
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t i = base + lane;
    const bool pass = i < n && c.cand_dist[(size_t)q * c.cap + i] <= thr2;
    const uint64_t m = __ballot(pass);
    const uint32_t cntp = (uint32_t)__popcll(m);
    if (npick + cntp > 128) {
      __builtin_amdgcn_wave_barrier();
      while (npick >= (uint32_t)kWave) { npick -= kWave; flush(npick, kWave); }
      if (npick) { flush(0, npick); npick = 0; }
      __builtin_amdgcn_wave_barrier();
    }
    if (pass) pick[npick + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c.cand_key[(size_t)q * c.cap + i];
    npick += cntp;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  while (npick >= (uint32_t)kWave) { npick -= kWave; flush(npick, kWave); }
  if (npick) flush(0, npick);
  // lane i holds result i: map the candidate-order rank g back to the probe rank r
  const uint32_t g = sel.p >> kPosBits, pos = sel.p & ((1u << kPosBits) - 1u);
  uint32_t r = 0;
  for (uint32_t rr = 0; rr < a.P; ++rr) {
    const uint32_t gv = readlane_u(g_of_r, (int)rr);
    if (gv == g) r = rr;
  }
  const uint32_t found = n < a.k ? n : a.k;
  if ((uint32_t)lane < a.k) {
    const size_t o = (size_t)q * a.k + lane;
    if ((uint32_t)lane < found && sel.p != kNoPos) {
      const uint32_t list = a.probes[(size_t)q * a.P + r];
      const uint64_t gslot = (uint64_t)a.first_block[list] * kWave + pos;
      a.D[o] = sel.d;
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
  if (a.counts && lane == 0) a.counts[q] = found;
}

__global__ void iota_kernel(uint32_t *p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

__global__ void coarse_tau_kernel(const float *run_dist, const uint32_t *run_pos, uint32_t nq, uint32_t P, float *tau) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const size_t o = (size_t)q * P + (P - 1);
  tau[q] = run_pos[o] == kNoPos ? INFINITY : run_dist[o];
}

struct CoarseSelectArgs {
  SelectCommon c;
  uint32_t nq, P;
  const float4 *blocks;  // centroid table
  const uint32_t *list_shard, *list_len;
  uint32_t *probes, *gorder, *cnt, *overflow;
};

// one wave per query: the P nearest centroids among the filter's survivors, in (distance, centroid index)
// order (the reference's stable sort, ivf_index.rs:205-220), then shard visiting order + histogram as in
// coarse_merge_kernel
__global__ void __launch_bounds__(256) coarse_select_kernel(CoarseSelectArgs a) {
  __shared__ uint32_t s_pick[4][128];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;
  const SelectCommon &c = a.c;
  const uint32_t n = c.cand_cnt[q];
  if (n > c.cap || n < a.P) {  // cannot happen with a finite bound unless the list overflowed
    if (lane == 0) atomicAdd(a.overflow, 1u);
    return;
  }
  const float thr2 = select_threshold(c, q, n, a.P, lane);
  WaveTopK sel;
  sel.init();
  const int K = (int)a.P;
  const float *qrow = c.Q + (size_t)q * c.dim;
  uint32_t *pick = s_pick[wave];
  uint32_t npick = 0;
  auto flush = [&](uint32_t off, uint32_t cnt) {
    const bool live = (uint32_t)lane < cnt;
    const uint32_t pos = live ? (pick[off + lane] & ((1u << kPosBits) - 1u)) : 0u;
    float d = INFINITY;
    if (live) d = exact_pair(qrow, a.blocks + ((size_t)(pos / kWave) * c.dq) * kWave + (pos % kWave), c.dim);
    sel.offer(d, live ? pos : kNoPos, K);
  };
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t i = base + lane;
    const bool pass = i < n && c.cand_dist[(size_t)q * c.cap + i] <= thr2;
    const uint64_t m = __ballot(pass);
    const uint32_t cntp = (uint32_t)__popcll(m);
    if (npick + cntp > 128) {
      __builtin_amdgcn_wave_barrier();
      while (npick >= (uint32_t)kWave) { npick -= kWave; flush(npick, kWave); }
      if (npick) { flush(0, npick); npick = 0; }
      __builtin_amdgcn_wave_barrier();
    }
    if (pass) pick[npick + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c.cand_key[(size_t)q * c.cap + i];
    npick += cntp;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  while (npick >= (uint32_t)kWave) { npick -= kWave; flush(npick, kWave); }
  if (npick) flush(0, npick);
  const uint32_t found = a.P;
  const uint32_t mylist = (uint32_t)lane < found ? sel.p : kNoPos;
  const uint32_t g = probe_candidate_order(lane, found, mylist, a.list_shard);
  if ((uint32_t)lane < a.P) {
    a.probes[(size_t)q * a.P + lane] = mylist;
    a.gorder[(size_t)q * a.P + lane] = g;
    if (a.list_len[mylist] > 0) atomicAdd(&a.cnt[mylist * kSubBins + (q & (kSubBins - 1))], 1u);
  }
}

__global__ void gather_queries_kernel(const float *Q, const uint32_t *ids, uint32_t n, uint32_t dim, float *out) {
  const uint32_t r = blockIdx.x;
  if (r >= n) return;
  for (uint32_t e = threadIdx.x; e < dim; e += blockDim.x) out[(size_t)r * dim + e] = Q[(size_t)ids[r] * dim + e];
}

__global__ void scatter_results_kernel(const uint32_t *ids, uint32_t n, uint32_t k, const float *Ds, const int64_t *Is,
                                       const uint64_t *Ts, const uint64_t *Ss, const uint32_t *Cs, float *D,
                                       int64_t *I, uint64_t *T, uint64_t *S, uint32_t *Cn) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * k) return;
  const uint32_t i = t / k, c = t % k;
  const size_t o = (size_t)ids[i] * k + c;
  D[o] = Ds[t];
  I[o] = Is[t];
  if (T) T[o] = Ts[t];
  if (S) S[o] = Ss[t];
  if (Cn && c == 0) Cn[ids[i]] = Cs[i];
}

template <int NG>
vi_status launch_filter_t(const FilterArgs &a, uint32_t nitems, hipStream_t st) {
  if (nitems == 0) return VI_OK;
  hipLaunchKernelGGL((filter_kernel<NG>), dim3(nitems), dim3(256), 0, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

}  // namespace

// norms of the stored vectors (filter accumulator init) — called once after the blocks are built
vi_status compute_slot_norms(DeviceIndex *ix) {
  const uint64_t nslots = ix->lists.nblocks * kWave;
  VI_TRY(ix->xnorm.reserve(std::max<uint64_t>(1, nslots)));
  DevBuf<uint32_t> mx;
  VI_TRY(mx.reserve(1));
  VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
  if (nslots) {
    hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, ix->stream,
                       (const float4 *)ix->lists.blocks.p, ix->dq, nslots, ix->ext_ids.p, ix->xnorm.p, mx.p);
    VI_HIP(hipGetLastError());
  }
  uint32_t bits = 0;
  VI_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
  VI_HIP(hipStreamSynchronize(ix->stream));
  float f;
  std::memcpy(&f, &bits, 4);
  ix->xmax2 = f;
  // the coarse table: pad slots (>= nlists) must never pass the filter
  const uint64_t cslots = ix->centroids.nblocks * kWave;
  VI_TRY(ix->cent_xnorm.reserve(std::max<uint64_t>(1, cslots)));
  VI_HIP(hipMemsetAsync(mx.p, 0, 4, ix->stream));
  if (cslots) {
    hipLaunchKernelGGL(slot_norms_kernel, dim3((uint32_t)((cslots + 255) / 256)), dim3(256), 0, ix->stream,
                       (const float4 *)ix->centroids.blocks.p, ix->dq, cslots, (const uint64_t *)nullptr,
                       ix->cent_xnorm.p, mx.p);
    VI_HIP(hipGetLastError());
    const uint64_t npad = cslots - ix->nlists;
    if (npad) {
      std::vector<float> inf(npad, INFINITY);
      VI_HIP(hipMemcpyAsync(ix->cent_xnorm.p + ix->nlists, inf.data(), npad * 4, hipMemcpyHostToDevice, ix->stream));
    }
  }
  VI_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
  VI_HIP(hipStreamSynchronize(ix->stream));
  std::memcpy(&f, &bits, 4);
  ix->cent_xmax2 = f;
  const uint32_t one_first[1] = {0u}, one_len[1] = {(uint32_t)ix->nlists};
  VI_TRY(ix->c_first.reserve(1));
  VI_TRY(ix->c_len.reserve(1));
  VI_HIP(hipMemcpy(ix->c_first.p, one_first, 4, hipMemcpyHostToDevice));
  VI_HIP(hipMemcpy(ix->c_len.p, one_len, 4, hipMemcpyHostToDevice));
  return VI_OK;
}

static vi_status launch_filter(const FilterArgs &a, uint32_t dq, uint32_t nitems, hipStream_t st) {
  switch (dq / 2) {  // dq is a multiple of 4
    case 2: return launch_filter_t<2>(a, nitems, st);
    case 4: return launch_filter_t<4>(a, nitems, st);
    case 6: return launch_filter_t<6>(a, nitems, st);
    case 8: return launch_filter_t<8>(a, nitems, st);
    case 10: return launch_filter_t<10>(a, nitems, st);
    case 12: return launch_filter_t<12>(a, nitems, st);
    case 14: return launch_filter_t<14>(a, nitems, st);
    case 16: return launch_filter_t<16>(a, nitems, st);
    default: return fail(VI_ERR_OTHER, "unsupported dimension for the MFMA filter");
  }
}

static void filter_margins(FilterArgs &a, uint32_t dim, float xmax2) {
  const double u = 1.01 * std::ldexp(1.0, -24);
  a.gamma2 = (float)(2.0 * (dim + 2.0) * u);
  a.e_scale = (float)((dim + 2.0) * u);
  a.xmax2 = xmax2;
}

// coarse quantizer on the matrix cores: the centroid table is one "list" probed by every query
vi_status stage_coarse_filter(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st,
                              bool *done) {
  SearchWorkspace &ws = ix.ws;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  *done = false;
  VI_TRY(ws.cnt.reserve(2 * nlists * kSubBins));
  VI_HIP(hipMemsetAsync(ws.cnt.p, 0, nlists * kSubBins * sizeof(uint32_t), st));
  VI_TRY(ws.probes.reserve(nq * P));
  VI_TRY(ws.gorder.reserve(nq * P));
  VI_TRY(ws.tau.reserve(nq));
  // a. bound: exact top-P over the first 512 centroids of the table
  const uint32_t nblk_c = (uint32_t)ix.centroids.nblocks;
  const uint32_t sblk = std::min<uint32_t>(nblk_c, kSampleBlocks);
  VI_TRY(ws.crun_dist.reserve(nq * P));
  VI_TRY(ws.crun_pos.reserve(nq * P));
  {
    const int qg = pick_qg(dq, (double)nq, ix.order);
    ScanArgs a{};
    a.blocks = (const float4 *)ix.centroids.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd; a.nq = (uint32_t)nq;
    a.K = P; a.run_dist = ws.crun_dist.p; a.run_pos = ws.crun_pos.p;
    a.nvec = (uint32_t)std::min<uint64_t>(nlists, (uint64_t)sblk * kWave); a.S = 1; a.bps = sblk;
    VI_TRY(launch_scan(a, qg, ix.order, true, (uint32_t)((nq + qg - 1) / qg), st));
    hipLaunchKernelGGL(coarse_tau_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.crun_dist.p,
                       ws.crun_pos.p, (uint32_t)nq, P, ws.tau.p);
    VI_HIP(hipGetLastError());
  }
  // b. one list, every query probes it: groups of 128 queries x segments of 4 blocks
  const uint32_t segb0 = 4;
  uint32_t segb;
  const uint32_t nseg = list_segments((uint32_t)nlists, segb0, &segb);
  const uint32_t ngroups = (uint32_t)((nq + kGroupQ - 1) / kGroupQ);
  const uint32_t h_seg[2] = {0u, (uint32_t)nq}, h_item[2] = {0u, ngroups * nseg};
  VI_TRY(ws.c_seg.reserve(2));
  VI_TRY(ws.c_item.reserve(2));
  VI_TRY(ws.c_pairs.reserve(nq));
  VI_HIP(hipMemcpyAsync(ws.c_seg.p, h_seg, 8, hipMemcpyHostToDevice, st));
  VI_HIP(hipMemcpyAsync(ws.c_item.p, h_item, 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(iota_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.c_pairs.p, (uint32_t)nq);
  // c. filter + exact re-check
  VI_TRY(ws.cand_cnt.reserve(nq + 1));
  VI_TRY(ws.cand_dist.reserve(nq * kCap));
  VI_TRY(ws.cand_key.reserve(nq * kCap));
  VI_TRY(ws.stats.reserve(8));
  VI_HIP(hipMemsetAsync(ws.cand_cnt.p, 0, (nq + 1) * sizeof(uint32_t), st));
  {
    FilterArgs a{};
    a.blocks = (const float4 *)ix.centroids.blocks.p; a.xnorm = ix.cent_xnorm.p; a.dq = dq; a.dim = dim; a.Q = Qd;
    a.first_block = ix.c_first.p; a.list_len = ix.c_len.p; a.item_start = ws.c_item.p; a.seg_start = ws.c_seg.p;
    a.pairs = ws.c_pairs.p; a.nlists = 1; a.P = 1; a.segb0 = segb0; a.tau = ws.tau.p;
    filter_margins(a, dim, ix.cent_xmax2);
    a.dbg = (unsigned long long *)ws.stats.p;
    a.cap = kCap; a.cand_cnt = ws.cand_cnt.p; a.cand_dist = ws.cand_dist.p; a.cand_key = ws.cand_key.p;
    VI_TRY(launch_filter(a, dq, ngroups * nseg, st));
  }
  // d. select the P probes, shard order, histogram
  {
    FilterArgs m{};
    filter_margins(m, dim, ix.cent_xmax2);
    SelectCommon c{Qd, dim, dq, kCap, ws.cand_cnt.p, ws.cand_key.p, ws.cand_dist.p, m.gamma2 * 0.5f, m.e_scale, m.xmax2};
    CoarseSelectArgs a{c, (uint32_t)nq, P, (const float4 *)ix.centroids.blocks.p, ix.list_shard.p, ix.list_len.p,
                       ws.probes.p, ws.gorder.p, ws.cnt.p, ws.cand_cnt.p + nq};
    hipLaunchKernelGGL(coarse_select_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, a);
    VI_HIP(hipGetLastError());
  }
  uint32_t overflow = 0;
  VI_HIP(hipMemcpyAsync(&overflow, ws.cand_cnt.p + nq, 4, hipMemcpyDeviceToHost, st));
  VI_HIP(hipStreamSynchronize(st));
  *done = overflow == 0;  // otherwise the caller runs the exact VALU coarse step
  return VI_OK;
}

bool filter_path_applicable(const DeviceIndex &ix, uint64_t nq, uint64_t k, uint32_t P) {
  const char *force = getenv("VI_FILTER");
  if (force && *force == '0') return false;
  if (ix.order != VI_ORDER_SCALAR || ix.dim > 128 || (ix.dim & 3) || ix.dim < 4) return false;
  if (k > kMaxSelect || P > kMaxSelect || P < 1) return false;
  if (ix.lists.nblocks * 64ull >= (1ull << kPosBits)) return false;  // candidate key holds position < 2^26
  if (force && *force == '1') return true;
  // worth it when query tiles fill up: on average >= 8 queries per probed list
  return (double)nq * P / (double)std::max<uint64_t>(1, ix.nlists) >= 8.0;
}

vi_status search_filter_pipeline(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P, uint32_t K,
                                 float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts, hipStream_t st,
                                 bool timing) {
  SearchWorkspace &ws = ix.ws;
  vi_search_stats &stt = ix.stats;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  if (timing) VI_HIP(hipEventRecord(ix.ev[0], st));
  {
    bool done = false;
    const char *cf = getenv("VI_COARSE_FILTER");
    if (!(cf && *cf == '0') && nq >= 256 && nlists >= 1024) VI_TRY(stage_coarse_filter(ix, Qd, nq, P, st, &done));
    if (!done) VI_TRY(stage_coarse(ix, Qd, nq, P, st));
  }
  if (timing) VI_HIP(hipEventRecord(ix.ev[1], st));

  // ---- 1. bound: exact top-k over the first 512 vectors of each of the query's nearest lists ----
  const uint32_t R0 = std::min<uint32_t>(kSampleRanks, P);
  VI_TRY(ws.probes0.reserve(nq * R0));
  VI_TRY(ws.tau.reserve(nq));
  VI_TRY(ws.run_dist.reserve(nq * R0 * K));
  VI_TRY(ws.run_pos.reserve(nq * R0 * K));
  VI_HIP(hipMemsetAsync(ws.run_pos.p, 0xFF, nq * R0 * K * sizeof(uint32_t), st));
  hipLaunchKernelGGL(take_first_ranks_kernel, dim3((uint32_t)((nq * R0 + 255) / 256)), dim3(256), 0, st, ws.probes.p,
                     (uint32_t)nq, P, R0, ws.probes0.p);
  VI_HIP(hipGetLastError());
  uint64_t hstats[3];
  {
    const int qg = pick_qg(dq, (double)nq * R0 / (double)std::max<uint64_t>(1, nlists), ix.order);
    const uint32_t segb0 = 1u << 20;  // never segment here: only the first blocks are read
    VI_TRY(launch_grouping(ix, ws.probes0.p, nq, R0, qg, segb0, hstats, st));
    ScanArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd; a.nq = (uint32_t)nq;
    a.K = K; a.run_dist = ws.run_dist.p; a.run_pos = ws.run_pos.p;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = R0;
    a.segb0 = segb0; a.segrun_start = ws.segrun_start.p; a.max_blocks = kSampleBlocks;
    VI_TRY(launch_scan(a, qg, ix.order, false, (uint32_t)hstats[1], st));
    hipLaunchKernelGGL(tau_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.run_dist.p, ws.run_pos.p,
                       (uint32_t)nq, R0, K, (uint32_t)std::min<uint64_t>(k, K), ws.tau.p);
    VI_HIP(hipGetLastError());
  }
  // ---- 2. group all (query, probe) pairs by list in tiles of 32 queries ----
  const char *sb = getenv("VI_FILTER_SEGB");
  const uint32_t segb0 = sb ? (uint32_t)std::max(1, atoi(sb)) : 16u;  // <= 1024 vectors per work item
  VI_TRY(launch_grouping(ix, ws.probes.p, nq, P, kGroupQ, segb0, hstats, st));
  stt.scanned_vectors = hstats[0];
  stt.scan_items = hstats[1];
  if (timing) VI_HIP(hipEventRecord(ix.ev[2], st));
  // ---- 3. MFMA filter + exact re-check -> per-query candidate lists ----
  VI_TRY(ws.cand_cnt.reserve(nq));
  VI_TRY(ws.cand_dist.reserve(nq * kCap));
  VI_TRY(ws.cand_key.reserve(nq * kCap));
  VI_TRY(ws.fallback.reserve(nq));
  VI_HIP(hipMemsetAsync(ws.cand_cnt.p, 0, nq * sizeof(uint32_t), st));
  {
    FilterArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.xnorm = ix.xnorm.p; a.dq = dq; a.dim = dim; a.Q = Qd;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = P; a.segb0 = segb0;
    a.tau = ws.tau.p;
    filter_margins(a, dim, ix.xmax2);
    a.dbg = (unsigned long long *)ws.stats.p;
    { const char *xm = getenv("VI_FILTER_XMODE"); a.xmode = xm ? (uint32_t)atoi(xm) : 0u; }
    a.cap = kCap; a.cand_cnt = ws.cand_cnt.p; a.cand_dist = ws.cand_dist.p; a.cand_key = ws.cand_key.p;
    VI_TRY(launch_filter(a, dq, (uint32_t)hstats[1], st));
  }
  if (timing) VI_HIP(hipEventRecord(ix.ev[3], st));
  // ---- 4. select ----
  {
    FilterArgs m{};
    filter_margins(m, dim, ix.xmax2);
    SelectCommon c{Qd, dim, dq, kCap, ws.cand_cnt.p, ws.cand_key.p, ws.cand_dist.p, m.gamma2 * 0.5f, m.e_scale, m.xmax2};
    SelectArgs a{c, (uint32_t)nq, P, (uint32_t)k, (const float4 *)ix.lists.blocks.p, ws.probes.p, ws.gorder.p,
                 ix.list_first_block.p, ws.tau.p, ix.ext_ids.p, Dd, Id, Td, slots, counts, ws.fallback.p};
    hipLaunchKernelGGL(select_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, a);
    VI_HIP(hipGetLastError());
  }
  if (timing) VI_HIP(hipEventRecord(ix.ev[4], st));
  // ---- 5. queries without a finite bound / with an overflowing candidate list: exact pipeline ----
  std::vector<uint8_t> hfb(nq);
  VI_HIP(hipMemcpyAsync(hfb.data(), ws.fallback.p, nq, hipMemcpyDeviceToHost, st));
  VI_HIP(hipStreamSynchronize(st));
  std::vector<uint32_t> ids;
  uint64_t n_inf = 0;
  for (uint64_t q = 0; q < nq; ++q)
    if (hfb[q]) { ids.push_back((uint32_t)q); n_inf += hfb[q] == 1; }
  if (getenv("VI_FILTER_VERBOSE"))
    fprintf(stderr, "[vi] filter fallback: %zu queries (%llu without a finite bound, %llu candidate overflow)\n",
            ids.size(), (unsigned long long)n_inf, (unsigned long long)(ids.size() - n_inf));
  stt.fallback_queries = ids.size();
  {
    uint64_t dbg[8];
    VI_HIP(hipMemcpy(dbg, ws.stats.p, sizeof(dbg), hipMemcpyDeviceToHost));
    stt.filter_tile_blocks = dbg[3]; stt.filter_rechecked = dbg[4]; stt.filter_accepted = dbg[5];
  }
  if (!ids.empty()) {
    const uint64_t m = ids.size();
    DevBuf<uint32_t> d_ids, cs;
    DevBuf<float> qs, ds;
    DevBuf<int64_t> is;
    DevBuf<uint64_t> ts, ss;
    VI_TRY(d_ids.reserve(m)); VI_TRY(qs.reserve(m * dim)); VI_TRY(ds.reserve(m * k)); VI_TRY(is.reserve(m * k));
    VI_TRY(ts.reserve(m * k)); VI_TRY(ss.reserve(m * k)); VI_TRY(cs.reserve(m));
    VI_HIP(hipMemcpyAsync(d_ids.p, ids.data(), m * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_queries_kernel, dim3((uint32_t)m), dim3(64), 0, st, Qd, d_ids.p, (uint32_t)m, dim, qs.p);
    VI_HIP(hipGetLastError());
    const uint64_t keep_scanned = stt.scanned_vectors, keep_items = stt.scan_items;
    VI_TRY(search_valu_pipeline(ix, qs.p, m, k, P, K, ds.p, is.p, ts.p, ss.p, cs.p, st, false));
    stt.scanned_vectors = keep_scanned; stt.scan_items = keep_items;
    hipLaunchKernelGGL(scatter_results_kernel, dim3((uint32_t)((m * k + 255) / 256)), dim3(256), 0, st, d_ids.p,
                       (uint32_t)m, (uint32_t)k, ds.p, is.p, ts.p, ss.p, cs.p, Dd, Id, Td, slots, counts);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(st));
  }
  return VI_OK;
}

}  // namespace vi
