// pgt_af_kernels.hip — allele-frequency front end: population allele frequencies
// -> Reynolds / Weir-Cockerham variance components -> the fstWindow reduction, for ALL pairs of up
// to 8 populations in one pass over the frequency columns.
//
// Spec: WCFst() of the reference's betaAFOutlier.R:400-418 — per site and pair (1,2)
//     npool = n1+n2;  fpool = n1/npool*f1 + n2/npool*f2;  alpha_k = 2*f_k*(1-f_k)
//     b = (n1*alpha1 + n2*alpha2)/(npool-1)
//     a = (4*n1*(f1-fpool)^2 + 4*n2*(f2-fpool)^2 - b)/(4*n1*n2/npool)
// and fstWindow's window statistic Σa / Σ(a+b) (fstWindow.cpp:76-85 on the columns a, a+b that
// betaAFOutlier.R:415-417 returns).
//
// Since f1-fpool = (n2/npool)(f1-f2) and f2-fpool = -(n1/npool)(f1-f2), the numerator's first two
// terms are exactly (4*n1*n2/npool)(f1-f2)^2, hence a = (f1-f2)^2 - b*npool/(4*n1*n2): window sums
// of every pair follow from  A_k = Σ 2 f_k (1-f_k)  (one per population) and  D_ij = Σ (f_i-f_j)^2
// (one per pair):   Σb = (n_i A_i + n_j A_j)/(npool-1),   Σa = D_ij - Σb * npool/(4 n_i n_j).
// So the tree carries V = NP + NP(NP-1)/2 scalar sums per node (36 for 8 populations) and the
// build streams 8 B/site/population (64 B/site for 8 populations) instead of the 16 B/site/pair
// (448 B/site for 28 pairs) of precomputed component columns.
//
// Cross-lane cost: V sums per leaf node would be V six-step butterflies; instead a
// REDUCE-SCATTER halves the live values at every exchange step (18+9+5+3+2+1 = 38 exchanges for
// V = 36), leaving each total in exactly one lane, which stores it and keeps the level-2 running sum.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "pgt_device.h"
#include "pgt_internal.h"

namespace pgt {
namespace {

using namespace dev;

// Tree shape of the AF front end: 512-site leaf nodes (round 6; 256 until then), 16 of them per level-2 node (8192 sites, as
// in every other f64 tree), radix 64 above.  What decides the leaf size is the BYTES OF LEVEL-1 NODES THE BUILD WRITES: node
// stores cost ~6.5 x their share of the kernel's bytes (profiles/r06/af8_issue_stall.md: with the stores removed the
// 8-population build reads at 86 % of the HBM peak; the arithmetic and the reduce-scatter cost nothing).  128-site leaves
// (until late in round 2): 3.5 % of the bytes read at 8 populations; 256: 1.8 %, +4.7 / +5.1 / +7.9 points at 8 / 4 / 2
// populations (profiles/r02/af_leaf256.txt); 512: 0.9 %, +4 points more at 8 populations (76.2 -> 80.5 % with one wave per
// SIMD) — in round 2 this step had shown nothing, other things bound the kernel then; 1024: 80.5 % at two waves per SIMD, but
// the query's ragged ends (up to 2 x 1023 sites per window) cost more than the build gains (step 1.164 against 1.105 ms).
// The query pays for 512 too (ragged ends up to 2 x 511 sites) and therefore requests a side's sites together (af_query_kernel).
constexpr int kAfPieces = kAfLeafPieces;        // 128-site pieces (one 16-byte load per lane and column) per level-1 node
constexpr int kAfLeaf = kAfPieces * kLeafF64;    // sites per level-1 node
constexpr int kAfRadix1 = kRadix / kAfPieces;    // level-1 nodes per level-2 node

template <int NP>
struct Shape {
    static constexpr int kPairs = NP * (NP - 1) / 2;
    static constexpr int kVals = NP + kPairs;  // A_0..A_{NP-1}, then D_ij in lexicographic (i<j) order
};

struct AfCols {
    const double *f[kAfMaxPops];
    double nsamp[kAfMaxPops];
};

// ---- reduce-scatter across the wave ----------------------------------------------------------------
// Step order: the steps with the MOST exchanges (18 and 9 of the 38 at 8 populations) pair lanes across the wave halves and
// across 16-lane rows, where gfx950 has an instruction made for exactly this exchange: v_permlane32_swap / v_permlane16_swap
// swap the upper lanes of one register with the lower lanes of another, so that "keep one half of my values, receive the
// other half of my partner's" is two swaps (low and high dword) and ONE addition — no select, no LDS crossbar.  Until round
// 6 these two steps came last (xor 16 by ds_swizzle, xor 32 by ds_bpermute) and the 27 busiest exchanges cost 4 v_cndmask +
// 2 DPP moves + 1 add each (profiles/r06/af8_issue_stall.md: 29 % of the wave cycles were instruction-issue waits, the
// kernel ran 2 waves per SIMD at 228 VGPRs).  The remaining steps (5 + 3 + 2 + 1 exchanges) stay on DPP / ds_swizzle.
constexpr int kRsMask[6] = {32, 16, 1, 2, 8, 4};
template <int STEP>
__device__ __forceinline__ double xchg(double v) {
    static_assert(STEP >= 2, "steps 0 and 1 are swaps (rs_swap)");
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (STEP == 2) {         // xor 1: quad_perm [1,0,3,2]
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    } else if constexpr (STEP == 3) {  // xor 2: quad_perm [2,3,0,1]
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    } else if constexpr (STEP == 4) {  // xor 8: row_ror:8 inside the 16-lane row
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, false);
    } else {                           // xor 4: ds_swizzle bit mode (and 0x1f, or 0, xor 4)
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F);
        hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F);
    }
    return __hiloint2double(hi, lo);
}

// lower lanes (mask bit clear) keep `a` and receive the partner's `a`; upper lanes keep `b` and receive the partner's `b`:
// after the swaps register A holds {own a | partner's b} and B {partner's a | own b}, so A + B is the exchange's result
// in every lane (an addition is commutative bit for bit: own + received = received + own).
template <int STEP>
__device__ __forceinline__ double rs_swap(double a, double b) {
    const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    if constexpr (STEP == 0) {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    }
}

// One reduce-scatter step: C live values -> (C+1)/2.  A lane whose mask bit is set keeps the upper
// half and sends the lower half, its partner does the opposite; an odd C is padded with 0.
template <int C, int STEP>
__device__ __forceinline__ void rs_steps(double *v, int lane) {
    if constexpr (STEP < 6) {
        constexpr int H = (C + 1) / 2;
        if constexpr (STEP < 2) {
#pragma unroll
            for (int k = 0; k < H; ++k) v[k] = rs_swap<STEP>(v[k], (k + H < C) ? v[k + H] : 0.0);
        } else {
            const bool up = (lane & kRsMask[STEP]) != 0;
#pragma unroll
            for (int k = 0; k < H; ++k) {
                const double lo_v = v[k];
                const double hi_v = (k + H < C) ? v[k + H] : 0.0;
                const double keep = up ? hi_v : lo_v;
                const double send = up ? lo_v : hi_v;
                v[k] = keep + xchg<STEP>(send);
            }
        }
        rs_steps<H, STEP + 1>(v, lane);
    }
}

// Which of the V values ends up in this lane (-1: a padding slot)
template <int V>
__device__ __forceinline__ int rs_my_index(int lane) {
    int base = 0, real = V, c = V;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int H = (c + 1) / 2;
        if (lane & kRsMask[s]) { base += H; real = real > H ? real - H : 0; }
        else real = real < H ? real : H;
        c = H;
    }
    return real >= 1 ? base : -1;
}

// ---- per-site contributions ----------------------------------------------------------------------
// FIRST: the leaf's first site STARTS the sums — x*y is bit for bit fma(x, y, +0.0) (one rounding either way; a product that
// rounds to -0.0 would differ, but both factors' signs make these products >= +0.0: d*d, and 2f(1-f) for 0 <= f <= 1; a
// frequency outside [0, 1] is outside WCFst's domain) — and saves the 36 register clears per leaf.
template <int NP, bool FIRST = false>
__device__ __forceinline__ void af_accumulate(double *vals, const double *f) {
#pragma unroll
    for (int i = 0; i < NP; ++i)  // alpha_i, betaAFOutlier.R:408-409
        vals[i] = FIRST ? (2.0 * f[i]) * (1.0 - f[i]) + 0.0 : fma(2.0 * f[i], 1.0 - f[i], vals[i]);
    int p = NP;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = i + 1; j < NP; ++j) {
            const double d = f[i] - f[j];
            vals[p] = FIRST ? d * d : fma(d, d, vals[p]);  // explicit FMA: one op, one rounding (the TU is built with -ffp-contract=off)
            ++p;
        }
}

template <int V>
__device__ __forceinline__ double *af_node(const AfTree &tv, int level_slot, int v, uint64_t i) {
    return reinterpret_cast<double *>(tv.base + tv.off[level_slot]) + i * V + v;  // node-major: V doubles per node
}

// ---- BUILD: one wave per level-2 tile (64 pieces of 128 sites = kAfRadix1 leaf nodes of kAfLeaf sites) ---------------
// BURST > 0: one column at a time (pgt_kernels.hip: fst_build_kernel) — per group of BURST pieces each column's BURST
// kibibytes are requested and awaited in turn.  Taken for TWO populations only (BURST = 4: 77.7 -> 85.3 % of the HBM peak
// at 10^9 sites, 74.0 -> 75.6 % at 10^8); with 4 and 8 populations the piece-by-piece form below wins (8 populations: 78.2
// against 76.5 % at 10^9, 73.1 against 69.1 % at 10^8; profiles/r03/af_column_bursts_ab.txt; again in round 6 at one wave per
// SIMD: 58.7 ... 61.9 % against 80.5, profiles/r06/af8_variants_ab_6_bursts_and_grid.md).
template <int NP, int BURST>
__device__ __forceinline__ void af_build_body(const AfCols &cols, uint64_t n, uint64_t n_l2, const AfTree &tv) {
    constexpr int V = Shape<NP>::kVals;

    const int lane = threadIdx.x & (kWave - 1);
    // the wave's index, SAID to be wave-uniform (readfirstlane): tile base and rotation are then scalar registers and the only
    // per-lane part of an address is the lane's 16 bytes — with the index derived from threadIdx the compiler computed 64-bit
    // per-lane addresses for every load (91 v_lshl_add_u64 per leaf; 16 now: it still keeps `column + lane bytes` in vector
    // registers rather than use the scalar-base form of the load)
    const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int my = rs_my_index<V>(lane);
    const uint32_t lane_bytes = (uint32_t)lane * 16u;  // the lane's 16 bytes of a 1-KiB piece: the only per-lane part of an address
    constexpr uint64_t kTile2 = (uint64_t)kLeafF64 * kRadix;
    // Level-1 nodes are staged in LDS and leave once per level-2 tile as ONE contiguous block (node-major
    // tree: kAfRadix1 nodes x V doubles = 4.5 KiB at 8 populations and 512-site leaves).  History with 128-site leaves, 8 populations, % of
    // the HBM peak: each leaf total stored straight from the lane that held it (36 scattered 8-byte stores per
    // leaf) 59.6; value-major tree with one 512-byte row per value and tile (36 rows in 36 different arrays)
    // 57-62; node-major blocks 68-75; a timing-only build without level-1 stores 76.9.  As in fst_build_kernel
    // what costs is node writes interleaved with the read stream — the cost goes with their BYTES, not with how
    // they are issued: requesting the next level-2 tile's first leaf BEFORE the block is stored (69.6 vs 71.2 %,
    // profiles/r02/af_pipe.txt), two level-2 tiles staged per flush and / or three leaves prefetched (67.9-69.9
    // vs 69.7 %, af_stage2.txt) changed nothing; halving the bytes (256-site leaves in round 2, 512 in round 6) did.
    extern __shared__ __attribute__((aligned(16))) double af_stage[];
    double *stage = af_stage + (size_t)(threadIdx.x >> 6) * V * kAfRadix1;

    for (uint64_t t = wave0; t < n_l2; t += n_waves) {
        const uint64_t base = t * kTile2;
        const bool full = base + kTile2 <= n;
        // WAVES START THEIR TILES AT DIFFERENT PIECES (round 5; pgt_kernels.hip: tile_rotation): the waves run in lockstep, and
        // without this all of them are at the same offset of their tiles at any moment.  Even, a multiple of the burst; the
        // partial last tile is walked from its start (its guarded loads stop at n).
        static_assert(kAfPieces == 2 || kAfPieces == 4 || kAfPieces == 8, "the walks below take a leaf's pieces two at a time, in bursts of 4");
        const int rot = full ? (int)(((wave0 * 0x9E3779B1ull) >> 13) & (uint64_t)(kRadix - (kAfPieces > 4 ? kAfPieces : 4))) : 0;
        double l2acc = 0.0;
        double vals[V];
        auto load_full = [&](double2 *dst, int j) {  // piece j of a FULL tile: one 16-byte nt load per lane and column
#pragma unroll
            for (int k = 0; k < NP; ++k)
                dst[k] = load16<true>(reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(cols.f[k] + base + (uint64_t)j * kLeafF64) + lane_bytes));
        };
        auto reduce_piece = [&](auto first, const double2 *pc) {  // the lane's two sites of one piece into the leaf's running sums
            double fx[NP], fy[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) { fx[k] = pc[k].x; fy[k] = pc[k].y; }
            af_accumulate<NP, decltype(first)::value>(vals, fx);  // a leaf's first site starts the sums
            af_accumulate<NP>(vals, fy);
        };
        if constexpr (BURST > 0) {
            if (full) {
#pragma unroll 1
                for (int i0 = 0; i0 < kRadix; i0 += BURST) {
                    const int j0 = (i0 + rot) & (kRadix - 1);  // the walk starts at a piece of the wave's own (see `rot`)
                    double2 d[NP][BURST];
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
#pragma unroll
                        for (int u = 0; u < BURST; ++u)
                            d[k][u] = load16<true>(reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(cols.f[k] + base + (uint64_t)(j0 + u) * kLeafF64) + lane_bytes));
                        if (k + 1 < NP) load_fence(d[k][BURST - 1].y);
                    }
#pragma unroll
                    for (int u = 0; u < BURST; ++u) {
                        const int j = j0 + u;
                        double fx[NP], fy[NP];
#pragma unroll
                        for (int k = 0; k < NP; ++k) { fx[k] = d[k][u].x; fy[k] = d[k][u].y; }
                        const int ph = (i0 + u) & (kAfPieces - 1);  // the piece's place in its leaf (i0 and j0 agree modulo the leaf: see `rot`)
                        if (ph == 0) af_accumulate<NP, true>(vals, fx);  // a leaf's first site starts the sums
                        else af_accumulate<NP>(vals, fx);
                        af_accumulate<NP>(vals, fy);
                        if (ph == kAfPieces - 1) {
                            rs_steps<V, 0>(vals, lane);
                            if (my >= 0) stage[(j / kAfPieces) * V + my] = vals[0];
                        }
                    }
                }
            }
        }
        if (BURST == 0 && full) {
            // Piece by piece, software-pipelined over TWO register sets that swap roles without a copy (round 6: the `cur = nxt`
            // of the rolled form cost 16 v_mov_b64 per piece once the partial-tile branch shared its loop): while leaf q's first
            // piece (set A) is reduced its second (set B) is in flight, while B is reduced the next leaf's first piece is.
            double2 pa[NP], pb[NP];
            load_full(pa, rot);
#pragma unroll 1
            for (int i = 0; i < kRadix; i += kAfPieces) {  // one leaf per turn: kAfPieces pieces, two at a time
                const int j = (i + rot) & (kRadix - 1);  // rot is a multiple of the leaf's pieces: they stay together
                auto pair = [&](auto first, int h) {  // pieces j+h (in pa, requested a pair ago) and j+h+1 of the leaf
                    load_full(pb, j + h + 1);
                    reduce_piece(first, pa);  // a leaf's first site starts the sums
                    if (i + h + 2 < kRadix) load_full(pa, (j + h + 2) & (kRadix - 1));
                    reduce_piece(std::false_type{}, pb);
                };
                pair(std::true_type{}, 0);
                if constexpr (kAfPieces == 4) pair(std::false_type{}, 2);
                if constexpr (kAfPieces > 4) {
#pragma unroll 1
                    for (int h = 2; h < kAfPieces; h += 2) pair(std::false_type{}, h);  // (unrolled, a whole leaf's loads are hoisted: 408 registers at 8 pieces)
                }
                rs_steps<V, 0>(vals, lane);  // the leaf is complete: one reduce-scatter per leaf
                if (my >= 0) stage[(j / kAfPieces) * V + my] = vals[0];  // node j/kAfPieces of the wave's LDS stage: V consecutive doubles (conflict-free)
            }
        }
        if (!full) {  // the last, partial level-2 tile (one wave of the grid, once): guarded loads, no prefetch; f = 0 contributes nothing
#pragma unroll 1
            for (int q = 0; q < kAfRadix1; ++q) {
#pragma unroll 1
                for (int h = 0; h < kAfPieces; ++h) {
                    const uint64_t i0 = base + (uint64_t)(q * kAfPieces + h) * kLeafF64 + 2 * lane;
                    double2 pc[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        pc[k].x = i0 < n ? cols.f[k][i0] : 0.0;
                        pc[k].y = i0 + 1 < n ? cols.f[k][i0 + 1] : 0.0;
                    }
                    if (h == 0) reduce_piece(std::true_type{}, pc);
                    else reduce_piece(std::false_type{}, pc);
                }
                rs_steps<V, 0>(vals, lane);
                if (my >= 0) stage[q * V + my] = vals[0];
            }
        }
        // the level-2 node = the tile's leaf nodes added in LEAF order, whatever order they were produced in (they were added as
        // produced until round 5: now the walk starts somewhere else in every wave, and the node must not depend on the wave)
        if (my >= 0) {
#pragma unroll 8
            for (int q = 0; q < kAfRadix1; ++q) l2acc += stage[q * V + my];
            *af_node<V>(tv, 1, my, t) = l2acc;  // V consecutive doubles, one 8*V-byte run
        }
        // the tile's 64 level-1 nodes = ONE contiguous block of 512*V bytes, written as 1-KiB wave stores
        // (the stage belongs to this wave alone, LDS operations of a wave complete in order: no barrier)
        {
            constexpr int kNodes = kAfRadix1;
            double2 *dst = reinterpret_cast<double2 *>(af_node<V>(tv, 0, 0, t * kNodes));
            const double2 *src = reinterpret_cast<const double2 *>(stage);
            constexpr int kVec = V * kNodes / 2;  // double2 elements of the block
#pragma unroll 4
            for (int e = lane; e < kVec; e += kWave) {
                const double2 w = src[e];
                __builtin_nontemporal_store(w.x, &dst[e].x);
                __builtin_nontemporal_store(w.y, &dst[e].y);
            }
        }
    }
}

// Two occupancies of the one body.  The kernel is bound by its node stores and by how many column streams the chip has open
// at once (profiles/r06/af8_issue_stall.md), not by latency: with 4 or more populations ONE wave per SIMD (4 per CU; the
// allocator then takes ~280 registers and keeps a whole leaf's loads in flight) beats two — 8 populations, 10^8 sites:
// 76.2 -> 78.7 % of the HBM peak with 256-site leaves, -> 80.9 % with the 512-site leaves of round 6; two populations (one
// column at a time, 68 registers) stay at two waves per SIMD.
template <int NP, int BURST = 0>
__global__ __launch_bounds__(256, 2) void af_build_kernel(AfCols cols, uint64_t n, uint64_t n_l2, AfTree tv) {
    af_build_body<NP, BURST>(cols, n, n_l2, tv);
}
template <int NP, int BURST = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_build_kernel_w1(AfCols cols, uint64_t n, uint64_t n_l2, AfTree tv) {
    af_build_body<NP, BURST>(cols, n, n_l2, tv);
}
// from how many populations on the one-wave-per-SIMD build is taken (PGT_AF_ONE_WAVE_FROM: a measuring knob; 9 = never)
inline int af_one_wave_from() {
    static const int v = [] {
        const char *e = std::getenv("PGT_AF_ONE_WAVE_FROM");
        return e ? std::atoi(e) : 4;
    }();
    return v;
}

// ---- upper levels: parent = Σ of 64 children, per value (blockIdx.y) -----------------------------
__global__ __launch_bounds__(256) void af_up_kernel(AfTree tv, int child_slot, uint64_t n_child, uint64_t n_parent) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int v = blockIdx.y, V = tv.n_vals;
    const double *child = reinterpret_cast<const double *>(tv.base + tv.off[child_slot]);
    double *parent = reinterpret_cast<double *>(tv.base + tv.off[child_slot + 1]);
    for (uint64_t p = wave0; p < n_parent; p += n_waves) {
        const uint64_t i = p * kRadix + lane;
        double x = i < n_child ? child[i * V + v] : 0.0;
        x = wave_sum(x);
        if (lane == 0) parent[p * V + v] = x;
    }
}

// ---- QUERY: one wave per window, all pairs at once ---------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256, 2) void af_query_kernel(AfCols cols, const uint32_t *__restrict__ pos, AfTree tv,
                                                       const pgt_win *__restrict__ win, uint64_t n_win,
                                                       pgt_fst_row *__restrict__ out, uint64_t n_sites) {
    constexpr int V = Shape<NP>::kVals;
    constexpr int P = Shape<NP>::kPairs;
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;

    for (uint64_t w = wave0; w < n_win; w += n_waves) {
        const pgt_win wd = win[w];
        const uint64_t hi = wd.hi < n_sites ? wd.hi : n_sites;
        const uint64_t lo = wd.lo < hi ? wd.lo : hi;
        double acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.0;
        // A window's answer is a chain of dependent memory round trips (its table entry, ragged sites left and right, ragged
        // nodes left and right of every level, the top level, its coordinates), ~1.5 us each under load: until round 6 there
        // were up to 13 of them — every 64 ragged sites were a trip of their own.  Now a side's ragged sites are REQUESTED TOGETHER,
        // before the first addition (256 sites in flight at once), and the two ragged sides of level 1 (fewer
        // than 16 nodes each) go to the two halves of the wave in one trip: 8 are left (two per ragged side of up to 511 sites).  A window's sum depends on the window
        // alone (lane j always adds the same items in the same order), not on which other windows are asked for or on how many
        // GPUs share the table.
        // ... by 16-byte loads: a lane takes the PAIR of sites 2j, 2j+1 of a 128-site stride (the columns are 16-byte aligned, so an
        // even site index is a 16-byte boundary), two strides (256 sites) in flight at once; the pair's halves outside [from, to)
        // are skipped, the column's very last site (odd n) is read alone, by lane 0.  Measured at 8 populations, 10^8 sites,
        // 512-site leaves (profiles/r06/round6_f_ab_af_query.md, whole step): 8-byte loads with 8 strides of 64 sites in flight
        // 1.113 ms, this form 1.106, four strides of 128 in flight (256 registers and spills) 1.131.
        constexpr int kStrides = 2;
        auto sum_sites = [&](uint64_t from, uint64_t to) {
            for (uint64_t at = from & ~(uint64_t)1; at < to; at += (uint64_t)kStrides * kLeafF64) {
                double2 f[kStrides][NP];
#pragma unroll
                for (int t = 0; t < kStrides; ++t) {
                    const uint64_t i = at + (uint64_t)t * kLeafF64 + 2 * (uint64_t)lane;
                    const bool in = i < to && i + 1 < n_sites;  // the pair lies inside the column
#pragma unroll
                    for (int k = 0; k < NP; ++k) f[t][k] = in ? *reinterpret_cast<const double2 *>(cols.f[k] + i) : double2{0.0, 0.0};
                }
#pragma unroll
                for (int t = 0; t < kStrides; ++t) {
                    const uint64_t i = at + (uint64_t)t * kLeafF64 + 2 * (uint64_t)lane;
                    double fx[NP], fy[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) { fx[k] = f[t][k].x; fy[k] = f[t][k].y; }
                    if (i >= from && i < to && i + 1 < n_sites) af_accumulate<NP>(acc, fx);
                    if (i + 1 >= from && i + 1 < to) af_accumulate<NP>(acc, fy);
                }
            }
            if ((n_sites & 1) && to == n_sites && from < n_sites && lane == 0) {  // the column's last site when it has no partner
                double fl[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) fl[k] = cols.f[k][n_sites - 1];
                af_accumulate<NP>(acc, fl);
            }
        };
        auto sum_nodes = [&](int level, uint64_t from, uint64_t to) {
            for (uint64_t i = from + lane; i < to; i += kWave) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] += *af_node<V>(tv, level - 1, v, i);
            }
        };
        uint64_t clo = lo, chi = hi;
        for (int k = 0;; ++k) {
            const bool top = k == tv.n_levels;
            const uint64_t r = k == 0 ? (uint64_t)kAfLeaf : (k == 1 ? (uint64_t)kAfRadix1 : (uint64_t)kRadix);
            const uint64_t ulo = (clo + r - 1) / r, uhi = chi / r;
            if (top || ulo >= uhi) {
                if (k == 0) sum_sites(clo, chi); else sum_nodes(k, clo, chi);
                break;
            }
            if (k == 0) { sum_sites(clo, ulo * r); sum_sites(uhi * r, chi); }
            else {
                // both ragged sides of a node level in ONE trip when each holds at most 32 nodes (always on level 1: fewer than
                // kAfRadix1 = 16): lanes 0-31 take the left side's nodes, lanes 32-63 the right side's
                const uint64_t nl = ulo * r - clo, nr = chi - uhi * r;
                if (nl <= 32 && nr <= 32) {
                    const uint64_t q = (uint64_t)(lane & 31);
                    if (lane < 32 ? q < nl : q < nr) {
                        const uint64_t i = lane < 32 ? clo + q : uhi * r + q;
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[v] += *af_node<V>(tv, k - 1, v, i);
                    }
                } else {
                    sum_nodes(k, clo, ulo * r);
                    sum_nodes(k, uhi * r, chi);
                }
            }
            clo = ulo;
            chi = uhi;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = wave_sum(acc[v]);  // every lane holds every total
        // lanes 0..P-1 each finish one pair (i<j, lexicographic)
        int pi = 0, pj = 1, p = 0;
        double Ai = 0.0, Aj = 0.0, D = 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = i + 1; j < NP; ++j) {
                if (p == lane) { pi = i; pj = j; Ai = acc[i]; Aj = acc[j]; D = acc[NP + p]; }
                ++p;
            }
        if (lane < P) {
            const double ni = cols.nsamp[pi], nj = cols.nsamp[pj], npool = ni + nj;
            const double sb = (ni * Ai + nj * Aj) / (npool - 1.0);      // Σb, betaAFOutlier.R:410
            const double sa = D - sb * (npool / (4.0 * ni * nj));       // Σa, betaAFOutlier.R:411 (see header)
            pgt_fst_row r;
            uint32_t start = wd.start, end = wd.end;
            if (!(wd.flags & PGT_WIN_COORDS)) {
                start = hi > lo ? pos[lo] : 0u;
                end = hi > lo ? pos[hi - 1] : 0u;
            }
            r.start = start;
            r.end = end;
            r.mid = (uint32_t)(start + end) / 2u;      // fstWindow.cpp:73
            r.n = (uint32_t)(hi - lo);
            r.asum = sa + 0.0;
            r.bsum = (sa + sb) + 0.0;                    // Σ(a+b): the tool's second column
            r.fst = r.bsum != 0.0 ? r.asum / r.bsum : 0.0;  // fstWindow.cpp:85
            out[(uint64_t)lane * n_win + w] = r;
        }
    }
}

inline int hip_fail(hipError_t e, const char *what, std::string *err) {
    if (e == hipSuccess) return PGT_OK;
    if (err) *err = std::string(what) + ": " + hipGetErrorString(e);
    return PGT_EDEVICE;
}

template <int NP>
int launch_af_np(const AfCols &cols, const uint32_t *pos, uint64_t n, const pgt_win *win, uint64_t n_win,
                 pgt_fst_row *out, const AfTree &tv, const TreeLayout &tl, hipStream_t s, void *ev_b0, void *ev_b1,
                 void *ev_q1, std::string *err) {
    auto rec = [&](void *ev) {
        return ev ? hip_fail(hipEventRecord(static_cast<hipEvent_t>(ev), s), "hipEventRecord", err) : PGT_OK;
    };
    if (int rc = rec(ev_b0)) return rc;
    if (n > 0) {
        // Balanced grid-stride over the level-2 tiles: r = ceil(tiles / (4 cap)) rounds, ceil(tiles / r) waves,
        // so every wave does r tiles.  Measured at 10^8 sites, 8 populations (profiles/r02/af_caps.txt): the
        // unbalanced 2048-workgroup grid of round 1 (12207 tiles over 8192 waves: half the waves do 2 tiles,
        // half 1) 68-69 % of the HBM peak, balanced 74.9 %; caps of 512 / 1024 workgroups 70.8 / 74.0 %.
        // Round 6: the grid is larger than what is resident at once (one wave per SIMD: 256 workgroups; two: 512), so the
        // workgroups run in GENERATIONS that end together — and a last generation that is nearly empty idles the chip for a
        // whole generation (3e8 sites, 8 populations: 1832 workgroups = 7.15 generations of 256, 76 % of the peak where 10^8
        // sites reach 81 %).  Among the next few round counts the one whose last generation is fullest is taken.
        const uint64_t cap = NP == 2 ? 512 : 2048;
        const bool w1 = NP >= af_one_wave_from();
        const uint64_t resident = w1 ? 256 : 512, max_waves = cap * 4;
        const uint64_t r0 = (tl.count[1] + max_waves - 1) / max_waves;
        uint64_t rounds = r0;
        double best = -1.0;
        for (uint64_t r = r0; r < r0 + 4; ++r) {
            const uint64_t wg = ((tl.count[1] + r - 1) / r + 3) / 4;
            const uint64_t gens = (wg + resident - 1) / resident;
            const double fill = (double)wg / (double)(gens * resident);  // 1 = every generation full
            if (fill > best + 0.02) { best = fill; rounds = r; }           // (a longer round only for a clearly fuller tail)
        }
        const uint64_t waves = (tl.count[1] + rounds - 1) / rounds;
        uint64_t blocks = (waves + 3) / 4;
        constexpr size_t kStage = (size_t)4 * Shape<NP>::kVals * kAfRadix1 * sizeof(double);  // 4 waves x kAfRadix1 nodes x V doubles
        if (w1)
            hipLaunchKernelGGL((af_build_kernel_w1<NP, (NP == 2 ? 4 : 0)>), dim3((unsigned)blocks), dim3(256), kStage, s, cols, n, tl.count[1], tv);
        else
            hipLaunchKernelGGL((af_build_kernel<NP, (NP == 2 ? 4 : 0)>), dim3((unsigned)blocks), dim3(256), kStage, s, cols, n, tl.count[1], tv);
        if (int rc = hip_fail(hipGetLastError(), "af_build_kernel", err)) return rc;
        for (int k = 2; k < tv.n_levels; ++k) {
            uint64_t b = (tl.count[k] + 3) / 4;
            if (b > 65536) b = 65536;
            hipLaunchKernelGGL(af_up_kernel, dim3((unsigned)b, Shape<NP>::kVals), dim3(256), 0, s, tv, k - 1,
                               tl.count[k - 1], tl.count[k]);
            if (int rc = hip_fail(hipGetLastError(), "af_up_kernel", err)) return rc;
        }
    }
    if (int rc = rec(ev_b1)) return rc;
    if (n_win > 0) {
        uint64_t b = (n_win + 3) / 4;
        if (b > 65536) b = 65536;
        hipLaunchKernelGGL((af_query_kernel<NP>), dim3((unsigned)b), dim3(256), 0, s, cols, pos, tv, win, n_win, out, n);
        if (int rc = hip_fail(hipGetLastError(), "af_query_kernel", err)) return rc;
    }
    return rec(ev_q1);
}

}  // namespace

namespace {
template <int NP>
void af_allow_lds() {
    const int bytes = (int)((size_t)4 * Shape<NP>::kVals * kAfRadix1 * sizeof(double));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(af_build_kernel_w1<NP, (NP == 2 ? 4 : 0)>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(af_build_kernel<NP, (NP == 2 ? 4 : 0)>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace

int init_af_kernels(std::string *err) {
    af_allow_lds<2>(); af_allow_lds<3>(); af_allow_lds<4>(); af_allow_lds<5>(); af_allow_lds<6>(); af_allow_lds<7>(); af_allow_lds<8>();
    return hip_fail(hipGetLastError(), "hipFuncSetAttribute", err);
}

int launch_fst_af(const uint32_t *pos, const double *const *freq, const double *nsamp, uint32_t n_pops, uint64_t n,
                  const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree, void *stream, void *ev_build0,
                  void *ev_build1, void *ev_query1, std::string *err, const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout tl = tree_layout(PGT_STAT_FST, n);
    const int n_vals = (int)(n_pops + n_pops * (n_pops - 1) / 2);
    const AfTree tv = af_tree_view(tl, n_vals, tree, useful_levels(tl, PGT_STAT_FST, hints.max_window));
    AfCols cols{};
    for (uint32_t k = 0; k < n_pops; ++k) { cols.f[k] = freq[k]; cols.nsamp[k] = nsamp[k]; }
    switch (n_pops) {
        case 2: return launch_af_np<2>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 3: return launch_af_np<3>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 4: return launch_af_np<4>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 5: return launch_af_np<5>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 6: return launch_af_np<6>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 7: return launch_af_np<7>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        case 8: return launch_af_np<8>(cols, pos, n, win, n_win, out, tv, tl, s, ev_build0, ev_build1, ev_query1, err);
        default:
            if (err) *err = "pgt_fst_af_reduce: 2 <= n_pops <= 8";
            return PGT_EARG;
    }
}

}  // namespace pgt
