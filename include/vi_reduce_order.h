/*
 * vi_reduce_order.h — the ONE place that fixes the order in which the four lanes of a `wide::f32x4` are summed by
 * `reduce_add` (compute_distance_simd, src/kmeans.rs:418; `wide` 0.7.33 is not vendored under the reference, so the
 * order on the default SSE2 build is PARITY UNPINNED).  Both the product (csrc/device_math.hpp, csrc/search_kernels.hip)
 * and the oracle (oracle/vi_oracle.c) include this header, so that the choice can be switched in one line once a
 * real `cargo` build pins it; GPU == oracle holds for every choice.
 *
 *   0  ((l0 + l1) + l2) + l3      sequential                      <- assumed
 *   1  (l0 + l1) + (l2 + l3)      adjacent pairs (two haddps)
 *   2  (l0 + l2) + (l1 + l3)      strided pairs (movehl + shuffle)
 *
 * An f32x8 on that build is two f32x4: reduce_add(f32x8) is taken as reduce4(low half) + reduce4(high half).
 */
#ifndef VI_REDUCE_ORDER_H
#define VI_REDUCE_ORDER_H

#ifndef VI_REDUCE4_ORDER
#define VI_REDUCE4_ORDER 0
#endif

#if VI_REDUCE4_ORDER == 0
#define VI_REDUCE4(l0, l1, l2, l3) ((((l0) + (l1)) + (l2)) + (l3))
#elif VI_REDUCE4_ORDER == 1
#define VI_REDUCE4(l0, l1, l2, l3) (((l0) + (l1)) + ((l2) + (l3)))
#elif VI_REDUCE4_ORDER == 2
#define VI_REDUCE4(l0, l1, l2, l3) (((l0) + (l2)) + ((l1) + (l3)))
#else
#error "VI_REDUCE4_ORDER must be 0, 1 or 2"
#endif

#endif /* VI_REDUCE_ORDER_H */
