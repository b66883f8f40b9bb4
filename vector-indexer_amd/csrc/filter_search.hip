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
constexpr int kTileQ = 32;              // queries per work item (one MFMA column tile)
constexpr uint32_t kSampleBlocks = 8;   // blocks of the nearest list sampled for the bound
constexpr uint32_t kCap = 4096;         // candidate slots per query
constexpr uint32_t kQueue = 2048 + 64;  // pair queue per wave (one block can emit 32 x 64 pairs)
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

__global__ void take_rank0_kernel(const uint32_t *probes, uint32_t nq, uint32_t P, uint32_t *probes0) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) probes0[q] = probes[(size_t)q * P];
}

__global__ void tau_kernel(const float *run_dist, const uint32_t *run_pos, uint32_t nq, uint32_t K, uint32_t k,
                           float *tau) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const size_t o = (size_t)q * K + (k - 1);
  tau[q] = run_pos[o] == kNoPos ? INFINITY : run_dist[o];
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
};

template <int NG>  // NG = dq/2 exactly: the block layout holds 2*NG quads (dims padded to 16); dim % 4 == 0
__global__ void __launch_bounds__(256, 2) filter_kernel(FilterArgs a) {
  __shared__ uint32_t s_queue[4][kQueue];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  uint32_t *queue = s_queue[wave];
  const uint32_t item = blockIdx.x * 4 + wave;
  if (item >= a.item_start[a.nlists]) return;
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
  const uint32_t j0 = chunk * kTileQ;
  const uint32_t nqi = min((uint32_t)kTileQ, cnt - j0);
  const uint32_t fb = a.first_block[l];
  const uint32_t nblk = (len + kWave - 1) / kWave;
  const uint32_t b0 = seg * segb, b1 = min(nblk, b0 + segb);

  // ---- this lane's query (both lane halves hold the same query j) ----
  const bool qlive = (uint32_t)j < nqi;
  const uint32_t slot = qlive ? a.pairs[s0 + j0 + j] : 0u;
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

  uint32_t qcount = 0;  // wave-uniform number of queued pairs

  // exact re-check of queue[off .. off+npairs) (npairs <= 64), one lane per (query, vector) pair
  auto drain = [&](uint32_t off, uint32_t npairs) {
    const bool active = (uint32_t)lane < npairs;
    const uint32_t pr = active ? queue[off + lane] : 0u;
    const uint32_t jq = pr & 31u, v = (pr >> 5) & 63u, blk = pr >> 11;
    const uint32_t pslot = (uint32_t)__shfl((int)slot, (int)jq);
    const float ptau = __shfl(tau, (int)jq);
    const uint32_t pq = pslot / a.P;
    const float4 *xq = reinterpret_cast<const float4 *>(a.Q + (size_t)pq * a.dim);
    const float4 *xv = a.blocks + ((size_t)(fb + blk) * a.dq) * kWave + v;
    float acc = 0.0f;
    const uint32_t nquad = a.dim >> 2;
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
    const unsigned long long nacc = __popcll(__ballot(active && acc <= ptau));
    if (lane == 0) { atomicAdd(&a.dbg[4], (unsigned long long)npairs); atomicAdd(&a.dbg[5], nacc); }
    if (active && acc <= ptau) {
      const uint32_t idx = atomicAdd(&a.cand_cnt[pq], 1u);
      if (idx < a.cap) {
        const uint32_t r = pslot - pq * a.P;
        a.cand_dist[(size_t)pq * a.cap + idx] = acc;
        a.cand_key[(size_t)pq * a.cap + idx] = (r << kPosBits) | (blk * kWave + v);
      }
    }
  };

  // Per block: two row tiles of 32 vectors, each tile = two halves of NG/2 k-groups.  The A fragments
  // (float4 per lane and k-group, straight from the lane-interleaved blocks) of the NEXT half are loaded
  // into the other of two register buffers while the current half's 2*NG MFMAs (>= 1024 cycles) run; a
  // sched_barrier pins those loads in front of the MFMAs (hipcc otherwise sinks them next to their use
  // and the L2 latency shows).
  constexpr int NH = NG / 2;  // NG is even
  auto tile_base = [&](uint32_t blk, uint32_t t) {
    return a.blocks + ((size_t)(fb + blk) * a.dq) * kWave + 32 * t + j;
  };
  auto load_half = [&](const float4 *vb, int half, float4 (&dst)[NH]) {
#pragma unroll
    for (int g = 0; g < NH; ++g) dst[g] = vb[(size_t)(2 * (half * NH + g) + h) * kWave];
  };
  auto load_norms = [&](uint32_t blk, uint32_t t, float4 (&nn)[4]) {
    const float *xn = a.xnorm + (size_t)(fb + blk) * kWave + 32 * t + 4 * h;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) nn[q4] = *reinterpret_cast<const float4 *>(xn + 8 * q4);
  };
  auto init_acc = [&](f32x16 &acc, const float4 (&nn)[4]) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {  // rows 8*q4 + 4*h + (0..3) live in regs 4*q4 .. 4*q4+3
      acc[4 * q4 + 0] = nn[q4].x; acc[4 * q4 + 1] = nn[q4].y; acc[4 * q4 + 2] = nn[q4].z; acc[4 * q4 + 3] = nn[q4].w;
    }
  };
  auto mfma_half = [&](f32x16 &acc, const float4 (&buf)[NH], int half) {
#pragma unroll
    for (int g = 0; g < NH; ++g) {
      const float4 qv = qf[half * NH + g];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[g].x, qv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[g].y, qv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[g].z, qv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[g].w, qv.w, acc, 0, 0, 0);
    }
  };
  float4 buf0[NH], buf1[NH], nn[4];
  if (b0 < b1) { load_norms(b0, 0, nn); load_half(tile_base(b0, 0), 0, buf0); }
  for (uint32_t blk = b0; blk < b1; ++blk) {
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 *vb = tile_base(blk, t);
      f32x16 acc;
      init_acc(acc, nn);
      load_half(vb, 1, buf1);                       // second half of this tile
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(acc, buf0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const bool more = (t == 0) || (blk + 1 < b1);
      if (more) {                                   // first half + norms of the next tile
        const uint32_t nb = t == 0 ? blk : blk + 1, nt = t == 0 ? 1u : 0u;
        load_norms(nb, nt, nn);
        load_half(tile_base(nb, nt), 0, buf0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(acc, buf1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) bits |= (acc[r] <= thr ? 1u : 0u) << (16 * t + r);
    }
    if (__ballot(bits != 0u) == 0ull) continue;  // nothing survived in this block (the common case)
    // ---- compact the surviving (query j, vector) pairs of this block into the wave's queue ----
    const uint32_t mycnt = __popc(bits);
    uint32_t incl = mycnt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t w = qcount + incl - mycnt;
    uint32_t bb = bits;
    while (bb) {
      const uint32_t bpos = (uint32_t)__builtin_ctz(bb);
      bb &= bb - 1;
      const uint32_t t = bpos >> 4, r = bpos & 15u;
      const uint32_t vec = 32u * t + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)h;
      queue[w++] = (blk << 11) | (vec << 5) | (uint32_t)j;
    }
    qcount += total;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    while (qcount >= (uint32_t)kWave) {  // re-check 64 pairs at a time, from the tail of the queue
      qcount -= kWave;
      drain(qcount, kWave);
    }
    __builtin_amdgcn_wave_barrier();  // queue reads above complete before the next block appends
  }
  if (qcount) drain(0, qcount);
}

struct SelectArgs {
  uint32_t nq, P, k, cap;
  const uint32_t *cand_cnt, *cand_key, *probes, *gorder, *first_block;
  const float *cand_dist, *tau;
  const uint64_t *ext_ids;
  float *D;
  int64_t *I;
  uint64_t *tie, *slots;
  uint32_t *counts;
  uint8_t *fallback;
};

// one wave per query: top-k of its candidates in the reference's stable order
__global__ void __launch_bounds__(256) select_kernel(SelectArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= a.nq) return;
  const uint32_t n = a.cand_cnt[q];
  const bool fb = !(a.tau[q] < INFINITY) || n > a.cap;
  if (lane == 0) a.fallback[q] = fb ? 1 : 0;
  if (fb) return;
  const uint32_t g_of_r = (uint32_t)lane < a.P ? a.gorder[(size_t)q * a.P + lane] : kNoPos;
  WaveTopK sel;
  sel.init();
  const int K = (int)a.k;
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t i = base + lane;
    const bool live = i < n;
    const float d = live ? a.cand_dist[(size_t)q * a.cap + i] : INFINITY;
    const uint32_t ck = live ? a.cand_key[(size_t)q * a.cap + i] : 0u;
    // cross-lane read outside of any divergent branch: the source lane (probe rank) must be active
    const uint32_t g = (uint32_t)__shfl((int)g_of_r, (int)(ck >> kPosBits));
    const uint32_t key = live ? ((g << kPosBits) | (ck & ((1u << kPosBits) - 1u))) : kNoPos;
    sel.offer(d, key, K);
  }
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
  hipLaunchKernelGGL((filter_kernel<NG>), dim3((nitems + 3) / 4), dim3(256), 0, st, a);
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
  VI_TRY(stage_coarse(ix, Qd, nq, P, st));
  if (timing) VI_HIP(hipEventRecord(ix.ev[1], st));

  // ---- 1. bound: exact top-k over the first 512 vectors of each query's nearest list ----
  VI_TRY(ws.probes0.reserve(nq));
  VI_TRY(ws.tau.reserve(nq));
  VI_TRY(ws.run_dist.reserve(nq * K));
  VI_TRY(ws.run_pos.reserve(nq * K));
  VI_HIP(hipMemsetAsync(ws.run_pos.p, 0xFF, nq * K * sizeof(uint32_t), st));
  hipLaunchKernelGGL(take_rank0_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.probes.p, (uint32_t)nq,
                     P, ws.probes0.p);
  VI_HIP(hipGetLastError());
  uint64_t hstats[3];
  {
    const int qg = pick_qg(dq, (double)nq / (double)std::max<uint64_t>(1, nlists), ix.order);
    const uint32_t segb0 = 1u << 20;  // never segment here: only the first blocks are read
    VI_TRY(launch_grouping(ix, ws.probes0.p, nq, 1, qg, segb0, hstats, st));
    ScanArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd; a.nq = (uint32_t)nq;
    a.K = K; a.run_dist = ws.run_dist.p; a.run_pos = ws.run_pos.p;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = 1;
    a.segb0 = segb0; a.segrun_start = ws.segrun_start.p; a.max_blocks = kSampleBlocks;
    VI_TRY(launch_scan(a, qg, ix.order, false, (uint32_t)hstats[1], st));
    hipLaunchKernelGGL(tau_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, st, ws.run_dist.p, ws.run_pos.p,
                       (uint32_t)nq, K, (uint32_t)std::min<uint64_t>(k, K), ws.tau.p);
    VI_HIP(hipGetLastError());
  }
  // ---- 2. group all (query, probe) pairs by list in tiles of 32 queries ----
  const uint32_t segb0 = 64;  // <= 4096 vectors per work item
  VI_TRY(launch_grouping(ix, ws.probes.p, nq, P, kTileQ, segb0, hstats, st));
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
    const double u = 1.01 * std::ldexp(1.0, -24);
    FilterArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.xnorm = ix.xnorm.p; a.dq = dq; a.dim = dim; a.Q = Qd;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = P; a.segb0 = segb0;
    a.tau = ws.tau.p;
    a.gamma2 = (float)(2.0 * (dim + 2.0) * u);
    a.e_scale = (float)((dim + 2.0) * u);
    a.xmax2 = ix.xmax2;
    a.dbg = (unsigned long long *)ws.stats.p;
    a.cap = kCap; a.cand_cnt = ws.cand_cnt.p; a.cand_dist = ws.cand_dist.p; a.cand_key = ws.cand_key.p;
    const uint32_t nitems = (uint32_t)hstats[1];
    switch (dq / 2) {  // dq is a multiple of 4
      case 2: VI_TRY(launch_filter_t<2>(a, nitems, st)); break;
      case 4: VI_TRY(launch_filter_t<4>(a, nitems, st)); break;
      case 6: VI_TRY(launch_filter_t<6>(a, nitems, st)); break;
      case 8: VI_TRY(launch_filter_t<8>(a, nitems, st)); break;
      case 10: VI_TRY(launch_filter_t<10>(a, nitems, st)); break;
      case 12: VI_TRY(launch_filter_t<12>(a, nitems, st)); break;
      case 14: VI_TRY(launch_filter_t<14>(a, nitems, st)); break;
      case 16: VI_TRY(launch_filter_t<16>(a, nitems, st)); break;
      default: return fail(VI_ERR_OTHER, "unsupported dimension for the MFMA filter");
    }
  }
  if (timing) VI_HIP(hipEventRecord(ix.ev[3], st));
  // ---- 4. select ----
  {
    SelectArgs a{(uint32_t)nq, P, (uint32_t)k, kCap, ws.cand_cnt.p, ws.cand_key.p, ws.probes.p, ws.gorder.p,
                 ix.list_first_block.p, ws.cand_dist.p, ws.tau.p, ix.ext_ids.p, Dd, Id, Td, slots, counts,
                 ws.fallback.p};
    hipLaunchKernelGGL(select_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, a);
    VI_HIP(hipGetLastError());
  }
  if (timing) VI_HIP(hipEventRecord(ix.ev[4], st));
  // ---- 5. queries without a finite bound / with an overflowing candidate list: exact pipeline ----
  std::vector<uint8_t> hfb(nq);
  VI_HIP(hipMemcpyAsync(hfb.data(), ws.fallback.p, nq, hipMemcpyDeviceToHost, st));
  VI_HIP(hipStreamSynchronize(st));
  std::vector<uint32_t> ids;
  for (uint64_t q = 0; q < nq; ++q)
    if (hfb[q]) ids.push_back((uint32_t)q);
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
