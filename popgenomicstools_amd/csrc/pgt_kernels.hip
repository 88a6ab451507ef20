// pgt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the window-scan engine.
//
// What is replaced: the per-window re-summation of the reference's calcWindow
//   fstWindow.cpp:76-85 (Σa, Σb, ratio)   hetWindow.cpp:73-84 (counts, ratio)
//   dxyWindow.cpp:174-186 (Σd, neffective, nskip)   dxyWindow.cpp:381-385 (per-site dxy, totals)
// and the O(W) buffer shift after every window (fstWindow.cpp:92-99).
//
// How: every window of every tool/mode is one contiguous range [lo,hi) of the global site index
// (SURVEY.md §4), so the device keeps a radix-64 RANGE TREE over the site axis:
//   level 0 = the site columns themselves (SoA in HBM),
//   level 1 = one node per `leaf` sites  (128 for f64 columns, 1024 for the int8 column:
//             exactly one 16-byte load per lane of a 64-lane wave),
//   level k = one node per 64 level-(k-1) nodes (one lane per child).
// BUILD is the only pass that streams the columns: one wave owns one level-2 tile (64 leaf
// tiles), issues 16-byte non-temporal loads only, reduces each leaf tile with a wave butterfly and
// parks the leaf total in lane j; the finished tile's 64 level-1 nodes go to a per-wave LDS stage
// and reach HBM later as coalesced 1-KiB rows, all waves of the chip at about the same times
// (deferred stores: node writes interleaved with the read stream cost 10x their byte share).
// No barrier, no atomics: bitwise deterministic.
// QUERY gives one wave per window: at each level the ragged left/right remainders (< radix
// nodes each) are read lane-parallel and loop-free, the aligned interior moves up a level.
// Memory-bound: 2 f64 adds per 16 B, no contraction to feed MFMA (none is used).
#include <hip/hip_runtime.h>

#include "pgt_device.h"
#include "pgt_internal.h"

namespace pgt {
namespace {

using namespace dev;

// ------------------------------------------------------------------------------------------
// Tree nodes
// ------------------------------------------------------------------------------------------
struct alignas(16) NodeFst { double x, y; };
struct alignas(8) NodeHet { uint32_t nonmiss, nhet; };
struct alignas(16) NodeDxy { double s; uint32_t neff, nskip; };

// ihsWindow / xpehhWindow: running extreme (key, first site index attaining it) and the count beyond
// the cutoff.  The combine is commutative and associative (ties go to the smaller site index, i.e. the
// first occurrence, as the strict `>` of ihsWindow.cpp:199 gives), so any reduction tree is exact.
struct alignas(16) NodeExt { double key; uint32_t idx, count; };

template <class N> __device__ __forceinline__ N node_identity() { return N{}; }
template <> __device__ __forceinline__ NodeExt node_identity<NodeExt>() {
    return NodeExt{-__builtin_huge_val(), 0xFFFFFFFFu, 0u};
}

__device__ __forceinline__ void node_add(NodeFst &a, const NodeFst &b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void node_add(NodeHet &a, const NodeHet &b) { a.nonmiss += b.nonmiss; a.nhet += b.nhet; }
__device__ __forceinline__ void node_add(NodeDxy &a, const NodeDxy &b) { a.s += b.s; a.neff += b.neff; a.nskip += b.nskip; }
__device__ __forceinline__ NodeFst node_wave_sum(NodeFst v) { return {wave_sum(v.x), wave_sum(v.y)}; }
__device__ __forceinline__ NodeHet node_wave_sum(NodeHet v) { return {wave_sum(v.nonmiss), wave_sum(v.nhet)}; }
__device__ __forceinline__ NodeDxy node_wave_sum(NodeDxy v) { return {wave_sum(v.s), wave_sum(v.neff), wave_sum(v.nskip)}; }
__device__ __forceinline__ void node_add(NodeExt &a, const NodeExt &b) {
    a.count += b.count;
    if (b.key > a.key || (b.key == a.key && b.idx < a.idx)) { a.key = b.key; a.idx = b.idx; }
}
__device__ __forceinline__ NodeExt node_wave_sum(NodeExt v) {
    NodeExt o;
    o = {dpp_f64<kDppQuadXor1>(v.key), dpp_u32<kDppQuadXor1>(v.idx), dpp_u32<kDppQuadXor1>(v.count)}; node_add(v, o);
    o = {dpp_f64<kDppQuadXor2>(v.key), dpp_u32<kDppQuadXor2>(v.idx), dpp_u32<kDppQuadXor2>(v.count)}; node_add(v, o);
    o = {dpp_f64<kDppRowHalfMirror>(v.key), dpp_u32<kDppRowHalfMirror>(v.idx), dpp_u32<kDppRowHalfMirror>(v.count)}; node_add(v, o);
    o = {dpp_f64<kDppRowMirror>(v.key), dpp_u32<kDppRowMirror>(v.idx), dpp_u32<kDppRowMirror>(v.count)}; node_add(v, o);
    o = {__shfl_xor(v.key, 16, kWave), (uint32_t)__shfl_xor((int)v.idx, 16, kWave), (uint32_t)__shfl_xor((int)v.count, 16, kWave)}; node_add(v, o);
    o = {__shfl_xor(v.key, 32, kWave), (uint32_t)__shfl_xor((int)v.idx, 32, kWave), (uint32_t)__shfl_xor((int)v.count, 32, kWave)}; node_add(v, o);
    return v;
}

constexpr int kMaxPairs = 32;
struct PairCols {
    const double *a[kMaxPairs];
    const double *b[kMaxPairs];
};

struct TreeView {
    char *base;            // workspace of pair 0
    size_t pair_stride;    // bytes between the trees of consecutive pairs
    size_t off[kMaxLevels];
    int n_levels;
    uint32_t n_partials;   // dxy: build waves that left a partial sum at `partials` (0: none, e.g. no sites)
    size_t partials;       // dxy: byte offset of those sums (TreeLayout::partials)
};

// ------------------------------------------------------------------------------------------
// BUILD, fst: level-1 {Σa,Σb} per 128 sites, level-2 per 8192 sites.        16 B/site read.
// grid.x: waves stride over level-2 tiles; grid.y: population pair.
//
// Node stores are DEFERRED.  Measured on the first version of this kernel, which stored a tile's 64
// level-1 nodes as soon as the tile was done: removing that store made the kernel 7-10 % faster
// although it is 0.8 % of the bytes (profiles/r01/ablate_stores.txt).  It is not a wave stall (a
// dedicated store wave, and software pipelining across tiles, changed nothing); the explanation that
// fits is on the memory side: L2 is write-through, so each node row reaches the DRAM channels as
// isolated writes among ~300 reads, each paying a bus turnaround.  So a wave parks its finished
// tiles' node rows in LDS (STAGE tiles = STAGE KiB per wave) and writes them out only when the stage
// is full or its work is done; all waves progress in near lockstep, hence the whole chip flushes at
// about the same times and the memory controllers see dense write bursts: +3.4 ... +8.6 % across boxes
// and sizes, results bit for bit the same.  The stage is private to the wave: no barrier.
// ------------------------------------------------------------------------------------------
// ONE COLUMN AT A TIME, FEW LOADS IN FLIGHT (round 3).  Until round 3 a wave requested `a` and `b` of 4 (or 8) leaf tiles
// together: 8 (16) loads in flight from two streams interleaved kibibyte by kibibyte.  The extreme-score build — ONE
// column, 128 KiB contiguous per tile — reached 84-86 % of the HBM peak where this kernel reached 78-83 %, and the
// difference turned out to be exactly that: a wave now reads the tile's 64 KiB of `a` (four loads in flight, consumed
// batch by batch), then its 64 KiB of `b`.  Same leaf sums, same bits.  Interleaved A/B in one process on five boxes
// (profiles/r03/phased_columns_ab_*.md; % of the HBM peak, old -> new): 10^9 sites 78.4 -> 83.8, 79.7 -> 80.9, 82.9 ->
// 85.4, 80.6 -> 81.4; 1.25e8 sites (the per-GPU shard of the 8-GPU run) 79.7 -> 82.5, 76.8 -> 83.8, 76.1 -> 83.4, 77.2 ->
// 82.5; 10^8 sites 75.5 -> 80.4, 75.6 -> 77.7, 78.1 -> 81.6, 77.6 -> 81.7; 5e7 sites 74.3 -> 79.4, 75.0 -> 78.5.  What matters
// is (i) at least two consecutive batches from the same stream — phases of 16 leaves do as well as 64, phases of one
// 8-load batch are back at the old rate — and (ii) a SHORT queue: 4 loads in flight beat 8 beat 16 (83.8 / 82.0 / 80.4 at
// 10^9 sites on one box), 2 and 1 are latency-bound (65 / 40 %); more waves per CU with a shallower stage lose (77 %).
// IN BURSTS (round 5).  What the "short queue" really is: the four kibibytes of a batch requested back to back — one 4-KiB-aligned
// contiguous burst per wave — and then nothing until they are consumed.  Refilling every load slot the moment its value is used
// (four 1-KiB loads in flight at ALL times, a steady trickle instead of bursts) costs 20 points at every size; bursts that do
// not start on a 4-KiB boundary (a rotation in steps of one or two leaves) cost one (profiles/r05/ab_series.md rolling4, fine1).
// THE TILE SCHEDULE STAYS STATIC (round 4).  Handing the tiles out by an atomic ticket counter after a static first round
// (self-resetting counter, one per pair and stream; rows bit-identical) was measured on the fst, dxy, fused and extreme-score
// builds: -12 ... -23 % at 1e8-1.25e8 sites, -1 ... -3 % at 1e9, with the ticket drawn at the tile's end or half a tile ahead:
// a fixed ~50 us per launch, because the first round ends in lockstep and ~2000 waves reach the one counter together — and
// again with their failing draws at the end (one word takes ~88 draws per microsecond).  Other geometries (4 / 16 / 32 waves
// per CU, 8 / 16 loads in flight) and a deeper queue in a wave's last tile (TAIL_* below) are within +-2 % or win at one size
// and lose at the next.  profiles/HISTORY.md A1-A2, profiles/r04/README.md.
constexpr int kFstStage = 16;                         // tiles staged per wave: 16 KiB of LDS
constexpr unsigned kFstBuildBlocks = 512;             // 64 KiB of LDS per workgroup -> 2 per CU, 8 waves per CU
static_assert(kFstBuildBlocks * 4 == kMaxBuildWaves, "the dxy tree reserves one partial sum per build wave");
constexpr size_t kFstStageBytes = (size_t)4 * kFstStage * 1024;

// The per-wave LDS stage shared by the fst, dxy and extreme-score builds (16-byte nodes: 1 KiB per row).
template <class Node>
__device__ __forceinline__ void store_node_nt(Node *dst, const Node &v) {
    static_assert(sizeof(Node) == 16, "16-byte nodes");
    double w[2];
    __builtin_memcpy(w, &v, 16);
    double *q = reinterpret_cast<double *>(dst);
    __builtin_nontemporal_store(w[0], q);
    __builtin_nontemporal_store(w[1], q + 1);
}
template <class Node, int STAGE, bool NT_STORE>
struct NodeStage {
    Node *rows;  // LDS: [STAGE][64] of this wave
    Node *l1, *l2;
    uint64_t n_waves, first_t = 0;  // row k holds tile first_t + k * n_waves
    Node sum = node_identity<Node>();  // of the level-2 nodes written so far, in tile order (wave-uniform; only the dxy builds read it)
    int lane, held = 0;
    __device__ __forceinline__ NodeStage(char *lds, int wib, int lane_, Node *l1_, Node *l2_, uint64_t n_waves_)
        : rows(reinterpret_cast<Node *>(lds) + (size_t)wib * STAGE * kWave), l1(l1_), l2(l2_), n_waves(n_waves_), lane(lane_) {}
    __device__ __forceinline__ void flush() {
        for (int k = 0; k < held; ++k) {
            const uint64_t t = first_t + (uint64_t)k * n_waves;
            const Node v = rows[k * kWave + lane];
            if constexpr (NT_STORE) store_node_nt(&l1[t * kRadix + lane], v);
            else l1[t * kRadix + lane] = v;  // one 1-KiB coalesced wave store
            const Node top = node_wave_sum(v);
            if (lane == 0) l2[t] = top;
            node_add(sum, top);
        }
        held = 0;
    }
    __device__ __forceinline__ void put(uint64_t t, const Node &keep) {  // lane j holds the node of leaf tile j
        if (held == 0) first_t = t;
        rows[held * kWave + lane] = keep;  // the wave's own LDS rows: no barrier needed
        if (++held == STAGE) flush();
    }
};
// WAVES START THEIR TILES AT DIFFERENT LEAVES (round 5).  All waves of a launch run in near lockstep (that is what makes the
// deferred node stores land together), so without this every wave is at the SAME offset of its 64-KiB tile at any moment:
// the chip's ~2000 outstanding 4-KiB requests all have the same address bits 12-15, and whatever part of the HBM channel /
// bank selection those bits feed is hit by all of them at once.  Each wave therefore walks its tile from a leaf of its own
// — a hash of the wave index, in steps of the batch size — and wraps around; a leaf's sum does not depend on when it is read
// and lane j still keeps leaf j's, so every node is bit for bit what it was.  Interleaved A/B of the two builds in one
// process (tools/lib_ab.py; % of the HBM peak; profiles/r05/ab_series.md, lib_ab_rotation_all_*.md): fst build 79.5 -> 82.9 and
// 78.8 -> 82.0 at 10^8 sites, 80.3 -> 81.4 and 81.1 -> 82.3 at 10^9; dxy 77.5 -> 82.2 at 10^8, 83.6 -> 85.4 at 10^9; fused 79.2 -> 82.1, 82.6 -> 83.7; the AF front end
// +3 ... +5 points at 10^8 (pgt_af_kernels.hip); the extreme-score build unchanged by itself, but its geometry could then be
// chosen (ext_build_launch).  A plain (4 x wave) & 63 gives the same at 10^8 sites and a point less at 10^9; a start that
// changes from tile to tile (hash of the tile index) or differs between the two columns of a tile gains less; starts at any
// leaf (steps of 1 or 2 leaves, the batch wrapping inside) lose a point at 10^8 sites and gain half a point at 10^9; other
// hashes tie; delaying the waves' starts against each other (s_sleep, 0.3 ... 3 us), odd waves reading `b` before `a`, and the
// fused build's genotype bursts at a batch of the wave's own gain nothing.  With the rotation in place the launch geometry was
// measured again (build_ab_fst_geometry_with_rotation*.md): 8 waves per CU x 4 loads stays ahead of 8 / 16 loads (-3 ... -8 %)
// and of 12 / 16 / 24 / 32 waves per CU with 2 or 4 loads (-1 ... -8 %) at every size.
template <int U>
__device__ __forceinline__ int tile_rotation(uint64_t wave) {
    return (int)(((wave * 0x9E3779B1ull) >> 13) & (uint64_t)(kRadix - U));  // a multiple of U below 64 (U a power of two)
}
// 64 KiB of one column of a level-2 tile -> the 64 leaf sums, lane j keeps leaf j's; U loads in flight per lane; the walk
// starts at leaf `rot` (a multiple of U) and wraps
template <int U>
__device__ __forceinline__ void fst_column_sums(const double2 *__restrict__ p, int lane, double &keep, int rot = 0) {
#pragma unroll 1
    for (int j0 = 0; j0 < kRadix; j0 += U) {
        const int j = (j0 + rot) & (kRadix - 1);
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load16<true>(p + (j + u) * kWave + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double s = wave_sum(v[u].x + v[u].y);
            if (lane == j + u) keep = s;
        }
    }
}
// UNROLL: loads in flight per lane (one column at a time).  TAIL_UNROLL / TAIL_SCOPE (tuning): the queue depth of a wave's LAST
// tile (scope 2: both columns, 1: only `b`, 0: off) — during the last round ever fewer waves are left to keep the HBM busy.
template <int STAGE = kFstStage, int UNROLL = 4, bool NT_STORE = true, int TAIL_UNROLL = UNROLL, int TAIL_SCOPE = 0>
__global__ __launch_bounds__(256) void fst_build_kernel(PairCols cols, uint64_t n, uint64_t n_l2, TreeView tv) {
    extern __shared__ __attribute__((aligned(16))) char lds_stage[];
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const double *__restrict__ a = cols.a[blockIdx.y];
    const double *__restrict__ b = cols.b[blockIdx.y];
    char *tree = tv.base + (size_t)blockIdx.y * tv.pair_stride;
    NodeFst *__restrict__ l1 = reinterpret_cast<NodeFst *>(tree + tv.off[0]);
    NodeFst *__restrict__ l2 = reinterpret_cast<NodeFst *>(tree + tv.off[1]);
    constexpr uint64_t kTile2 = (uint64_t)kLeafF64 * kRadix;
    NodeStage<NodeFst, STAGE, NT_STORE> stage(lds_stage, wib, lane, l1, l2, n_waves);

    for (uint64_t t = wave0; t < n_l2; t += n_waves) {
        const uint64_t base = t * kTile2;
        double keep_a = 0.0, keep_b = 0.0;
        if (base + kTile2 <= n) {
            const double2 *__restrict__ pa = reinterpret_cast<const double2 *>(a + base);
            const double2 *__restrict__ pb = reinterpret_cast<const double2 *>(b + base);
            const bool last = TAIL_SCOPE != 0 && t + n_waves >= n_l2;  // wave-uniform: this wave's last tile
            const int rot = tile_rotation<UNROLL>(wave0);
            if (TAIL_SCOPE == 2 && last) fst_column_sums<TAIL_UNROLL>(pa, lane, keep_a);
            else fst_column_sums<UNROLL>(pa, lane, keep_a, rot);  // the tile's 64 KiB of `a` ...
            if (TAIL_SCOPE != 0 && last) fst_column_sums<TAIL_UNROLL>(pb, lane, keep_b);
            else fst_column_sums<UNROLL>(pb, lane, keep_b, rot);  // ... then its 64 KiB of `b`
        } else {
            for (int j = 0; j < kRadix; ++j) {
                const uint64_t i0 = base + (uint64_t)j * kLeafF64 + 2 * lane;
                if (base + (uint64_t)j * kLeafF64 >= n) break;  // wave-uniform
                const double a0 = i0 < n ? a[i0] : 0.0, a1 = i0 + 1 < n ? a[i0 + 1] : 0.0;
                const double b0 = i0 < n ? b[i0] : 0.0, b1 = i0 + 1 < n ? b[i0 + 1] : 0.0;
                const double sa = wave_sum(a0 + a1);
                const double sb = wave_sum(b0 + b1);
                if (lane == j) { keep_a = sa; keep_b = sb; }
            }
        }
        stage.put(t, NodeFst{keep_a, keep_b});
    }
    stage.flush();
}

// ------------------------------------------------------------------------------------------
// BUILD, het: level-1 {nonmissing, nhet} per 1024 sites (16 int8 genotypes per lane).  1 B/site.
// nonmissing = g >= 0, nhet = g == 1 (hetWindow.cpp:78-80), counted bytewise on packed words.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void het_count_word(uint32_t w, uint32_t &nonmiss, uint32_t &nhet) {
    nonmiss += 4u - (uint32_t)__popc(w & 0x80808080u);
    const uint32_t x = w ^ 0x01010101u;  // bytes equal to 1 become 0
    const uint32_t y = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 per zero byte
    nhet += (uint32_t)__popc(y);
}

// One work item = 8 leaf tiles (8192 sites, 8 KiB): 8 x 16-byte loads in flight per lane, eight
// level-1 nodes stored by lanes 0-7.  Level 2 is derived from level 1 by tree_up_kernel: with
// 65536-site level-2 tiles a wave-per-level-2-tile build has too few work items to fill the chip
// (1526 at 10^8 sites; measured 0.8 TB/s), this form has 8x as many.  (Round 2, tried and dropped: one
// WORKGROUP per level-2 tile, 16 leaf tiles per wave, level 2 summed through LDS so that the tree_up launch
// disappears: 35 us instead of 30 at 10^8 sites, 0.162 instead of 0.158 ms at 10^9 — fewer, fatter items.)
constexpr int kHetChunk = 8;
constexpr uint64_t kHetSmallItems = 18000;  // 8192-site work items (1.5e8 sites) up to which the 128-VGPR build is used (round 6: re-measured
                                             // under the resident-sized grids, profiles/r06/het_kernel_crossover_ab.md: 1.25e8 sites 32.9 against 34.8 us, 2e8 41.5 against 37.9)
__device__ __forceinline__ void het_build_body(const int8_t *__restrict__ g, uint64_t n, uint64_t n_items,
                                               const TreeView &tv) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    NodeHet *__restrict__ l1 = reinterpret_cast<NodeHet *>(tv.base + tv.off[0]);
    constexpr uint64_t kItem = (uint64_t)kLeafI8 * kHetChunk;  // 8192 sites

    for (uint64_t t = wave0; t < n_items; t += n_waves) {
        const uint64_t base = t * kItem;
        uint32_t keep_nm = 0, keep_nh = 0;
        if (base + kItem <= n) {
            const uint4 *__restrict__ pg = reinterpret_cast<const uint4 *>(g + base);
            uint4 w[kHetChunk];
#pragma unroll
            for (int u = 0; u < kHetChunk; ++u) w[u] = load16_nt(pg + u * kWave + lane);
#pragma unroll
            for (int u = 0; u < kHetChunk; ++u) {
                uint32_t nm = 0, nh = 0;
                het_count_word(w[u].x, nm, nh);
                het_count_word(w[u].y, nm, nh);
                het_count_word(w[u].z, nm, nh);
                het_count_word(w[u].w, nm, nh);
                nm = wave_sum(nm);
                nh = wave_sum(nh);
                if (lane == u) { keep_nm = nm; keep_nh = nh; }
            }
        } else {  // last, partial item: bytewise, sites beyond n count as nothing
            for (int j = 0; j < kHetChunk; ++j) {
                const uint64_t tile0 = base + (uint64_t)j * kLeafI8;
                uint32_t nm = 0, nh = 0;
                for (int q = 0; q < 16; ++q) {
                    const uint64_t i = tile0 + (uint64_t)lane * 16 + q;
                    if (i < n) {
                        const int v = g[i];
                        nm += v >= 0;
                        nh += v == 1;
                    }
                }
                nm = wave_sum(nm);
                nh = wave_sum(nh);
                if (lane == j) { keep_nm = nm; keep_nh = nh; }
            }
        }
        if (lane < kHetChunk) l1[t * kHetChunk + lane] = NodeHet{keep_nm, keep_nh};
    }
}

__global__ __launch_bounds__(256) void het_build_kernel(const int8_t *g, uint64_t n, uint64_t n_items, TreeView tv) {
    het_build_body(g, n, n_items, tv);
}
// the same with at least 4 waves per SIMD (at most 128 VGPRs; left alone the compiler spends 254 on this body)
__global__ __launch_bounds__(256, 4) void het_build_kernel_w4(const int8_t *g, uint64_t n, uint64_t n_items, TreeView tv) {
    het_build_body(g, n, n_items, tv);
}

// ------------------------------------------------------------------------------------------
// BUILD, dxy: per-site value exactly as dxyWindow.cpp:381 (no FMA contraction: the products
// and sums are rounded one by one, as the host's SSE2 code does), then level-1 {Σd over d>=0,
// neffective, nskip} per 128 sites.                                            24 B/site read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dxy_site(double p1, double p2, int n1, int n2, int minind) {
    const double d = __dadd_rn(__dmul_rn(p1, __dsub_rn(1.0, p2)), __dmul_rn(p2, __dsub_rn(1.0, p1)));
    return (n1 >= minind && n2 >= minind) ? d : -9.0;
}
__device__ __forceinline__ void dxy_acc(NodeDxy &acc, double v) {  // dxyWindow.cpp:180-185
    if (v >= 0.0) { acc.s += v; acc.neff += 1; }
    else if (v == -9.0) acc.nskip += 1;
}

// THE COUNT COLUMNS BY 16-BYTE LOADS (round 5).  n1 / n2 are 4 B/site: with the f64 columns' layout (lane l owns sites 2l,
// 2l+1 of a 128-site leaf tile) they were 8-byte loads (`global_load_dwordx2 nt`, a third of the kernel's bytes).  Now one
// `global_load_dwordx4 nt` per lane covers 4 sites, a wave instruction 256 = the PAIR of leaf tiles j, j+1 (j even): lane L
// holds sites 4L .. 4L+3 of the pair.  The only thing the counts decide is the predicate of dxyWindow.cpp:381 (`nind >=
// minind` for both populations), so it is evaluated in THAT layout and transposed as four 64-bit ballots (wave-uniform,
// scalar registers): bit L of mask c = site 4L + c of the pair.  The lane that owns sites 2l + q (q = 0, 1) of leaf tile h
// (0, 1) of the pair — pair site 128h + 2l + q, i.e. count lane 32h + (l >> 1), component 2(l & 1) + q — picks its two bits
// from the masks.  Integer-exact, the f64 arithmetic is untouched: rows bit for bit those of the 8-byte form.
// MEASURED (interleaved A/B of the two builds in one process, profiles/r05/lib_ab_v1_1e8.md, ab_series.md `v1_pairs`): a TIE —
// dxy 80.9 -> 79.8 and 80.1 -> 79.9 % at 10^8 sites, 80.9 -> 81.0 at 10^9; fused 80.3 -> 80.8 at 10^8.  The guide's 0.54-0.70x for
// 8-byte nt accesses does not bind here: the kernel waits on HBM, not on the load unit.  Kept because every byte of every
// build kernel is now read by a 16-byte load, the one width rocprofv3's FETCH_SIZE is calibrated for (profiles/r05).
struct PairPred { unsigned long long m0, m1, m2, m3; };  // four scalars (an array of them lands in scratch)
__device__ __forceinline__ PairPred count_pair_pred(const int4 &k1, const int4 &k2, int minind) {
    return PairPred{__ballot(k1.x >= minind && k2.x >= minind), __ballot(k1.y >= minind && k2.y >= minind),
                    __ballot(k1.z >= minind && k2.z >= minind), __ballot(k1.w >= minind && k2.w >= minind)};
}
template <int H, int Q>  // leaf tile H of the pair, site 2 * lane + Q of it
__device__ __forceinline__ bool pair_pred_bit(const PairPred &p, int lane) {
    // shift both candidate masks, then choose (choosing the 64-bit mask per lane first makes the compiler index a scratch copy)
    const unsigned long long me = Q == 0 ? p.m0 : p.m1, mo = Q == 0 ? p.m2 : p.m3;  // even / odd lanes' mask
    const uint32_t e = (H ? (uint32_t)(me >> 32) : (uint32_t)me) >> (lane >> 1);
    const uint32_t o = (H ? (uint32_t)(mo >> 32) : (uint32_t)mo) >> (lane >> 1);
    return (((lane & 1) ? o : e) & 1u) != 0;
}
__device__ __forceinline__ double dxy_site_pred(double p1, double p2, bool counted) {  // dxy_site with the predicate given
    const double d = __dadd_rn(__dmul_rn(p1, __dsub_rn(1.0, p2)), __dmul_rn(p2, __dsub_rn(1.0, p1)));
    return counted ? d : -9.0;
}
// leaf tile H of a pair: the two sites of this lane -> the leaf's node in every lane
template <int H>
__device__ __forceinline__ NodeDxy dxy_leaf_node(const double2 &x1, const double2 &x2, const PairPred &pp, int lane) {
    NodeDxy acc{0.0, 0u, 0u};
    dxy_acc(acc, dxy_site_pred(x1.x, x2.x, pair_pred_bit<H, 0>(pp, lane)));
    dxy_acc(acc, dxy_site_pred(x1.y, x2.y, pair_pred_bit<H, 1>(pp, lane)));
    return node_wave_sum(acc);
}


// ONE COLUMN AT A TIME here too (round 3, see fst_build_kernel): per batch of U = 4 leaf tiles the wave requests p1's
// four kibibytes and waits, p2's and waits, then the two count columns together (round 3: eight 8-byte loads; round 5: four
// 16-byte loads, see count_pair_pred).  The per-site
// value needs all four columns, so the bursts are separated by load fences instead of by consumption.  Interleaved
// A/B, two boxes (profiles/r03/dxy_ab_*.txt; % of the HBM peak on 24 B/site, the round-2 form -> this one): 10^9 sites
// 80.0 -> 81.4, 81.7 -> 83.7; 1.25e8 sites 74.5 -> 78.5, 77.1 -> 80.1; 10^8 sites 74.9 -> 79.4, 72.5 -> 75.6.  Waiting after
// every column (n1 and n2 apart) loses 3-5 points, batches of 8 leaf tiles are no better, batches of 2 are latency-bound.
// Round 5, five more schedules of the same loads, all bit-identical, none ahead at both sizes (profiles/r05/ab_series.md v2-v6,
// dxy_schedule_variants_rejected.patch): the count columns as 4-load bursts per 8 leaf tiles (ballots kept in scalar
// registers; -0.9 / -2.8 points at 10^8 / 10^9 sites, with n1 awaited before n2 +1.8 / -2.4); the next batch's p1 burst requested
// before this batch is computed (+0.5 / -1.6); super-batches of 8 or 16 leaf tiles in which EVERY stream is visited for two or
// four consecutive 4-load bursts — counts, then p1 held in registers, then p2 consumed burst by burst as the fst build consumes
// its column (-2.2 / -0.6 and -3.4 / +0.3).
template <int U = 4>  // leaf tiles per batch (even: the count columns are read per PAIR of leaf tiles)
__device__ __forceinline__ void dxy_build_body(const double *__restrict__ p1, const double *__restrict__ p2,
                                               const int32_t *__restrict__ n1, const int32_t *__restrict__ n2,
                                               uint64_t n, int minind, uint64_t n_l2, const TreeView &tv, char *lds_stage) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    NodeDxy *__restrict__ l1 = reinterpret_cast<NodeDxy *>(tv.base + tv.off[0]);
    NodeDxy *__restrict__ l2 = reinterpret_cast<NodeDxy *>(tv.base + tv.off[1]);
    constexpr uint64_t kTile2 = (uint64_t)kLeafF64 * kRadix;
    NodeStage<NodeDxy, kFstStage, true> stage(lds_stage, threadIdx.x >> 6, lane, l1, l2, n_waves);  // deferred node stores

    for (uint64_t t = wave0; t < n_l2; t += n_waves) {
        const uint64_t base = t * kTile2;
        NodeDxy keep{0.0, 0u, 0u};
        if (base + kTile2 <= n) {
            const double2 *__restrict__ q1 = reinterpret_cast<const double2 *>(p1 + base);
            const double2 *__restrict__ q2 = reinterpret_cast<const double2 *>(p2 + base);
            const int4 *__restrict__ m1 = reinterpret_cast<const int4 *>(n1 + base);  // base is a multiple of 8192: 16-byte aligned
            const int4 *__restrict__ m2 = reinterpret_cast<const int4 *>(n2 + base);
            const int rot = tile_rotation<U>(wave0);  // see fst_column_sums: the walk over the tile starts at a leaf of the wave's own
#pragma unroll 1
            for (int j0 = 0; j0 < kRadix; j0 += U) {
                const int j = (j0 + rot) & (kRadix - 1);
                double2 x1[U], x2[U];
                int4 k1[U / 2], k2[U / 2];  // one 16-byte load per lane and PAIR of leaf tiles
#pragma unroll
                for (int u = 0; u < U; ++u) x1[u] = load16<true>(q1 + (j + u) * kWave + lane);
                load_fence(x1[U - 1].y);
#pragma unroll
                for (int u = 0; u < U; ++u) x2[u] = load16<true>(q2 + (j + u) * kWave + lane);
                load_fence(x2[U - 1].y);
#pragma unroll
                for (int u = 0; u < U / 2; ++u) k1[u] = load16_nt(m1 + (j / 2 + u) * kWave + lane);
#pragma unroll
                for (int u = 0; u < U / 2; ++u) k2[u] = load16_nt(m2 + (j / 2 + u) * kWave + lane);
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    const PairPred pp = count_pair_pred(k1[u / 2], k2[u / 2], minind);
                    const NodeDxy a0 = dxy_leaf_node<0>(x1[u], x2[u], pp, lane);
                    const NodeDxy a1 = dxy_leaf_node<1>(x1[u + 1], x2[u + 1], pp, lane);
                    if (lane == j + u) keep = a0;
                    if (lane == j + u + 1) keep = a1;
                }
            }
        } else {
            for (int j = 0; j < kRadix; ++j) {
                const uint64_t tile0 = base + (uint64_t)j * kLeafF64;
                if (tile0 >= n) break;  // wave-uniform
                NodeDxy acc{0.0, 0u, 0u};
                for (int q = 0; q < 2; ++q) {
                    const uint64_t i = tile0 + 2 * lane + q;
                    if (i < n) dxy_acc(acc, dxy_site(p1[i], p2[i], n1[i], n2[i], minind));
                }
                acc = node_wave_sum(acc);
                if (lane == j) keep = acc;
            }
        }
        stage.put(t, keep);
    }
    stage.flush();
    // THE GENOME-WIDE LINE WITHOUT UPPER LEVELS (round 5).  dxyWindow's last line (dxyWindow.cpp:382-385,429-433) is the sum over
    // ALL sites; it used to be the query [0, n), for which every tree level above 2 had to be built: two or three
    // tree_up launches of ~5 us each behind every dxy / fused build whose windows (W = 50000 sites) need none of them.
    // Every wave now leaves the sum of the level-2 nodes it wrote (tile order), and the total is the sum of these <= 2048
    // partials in wave order — fixed by the static grid, hence a function of the input alone.
    if (lane == 0) reinterpret_cast<NodeDxy *>(tv.base + tv.partials)[wave0] = stage.sum;
}

__global__ __launch_bounds__(256) void dxy_build_kernel(const double *p1, const double *p2, const int32_t *n1,
                                                        const int32_t *n2, uint64_t n, int minind, uint64_t n_l2,
                                                        TreeView tv) {
    extern __shared__ __attribute__((aligned(16))) char lds_stage[];
    dxy_build_body<>(p1, p2, n1, n2, n, minind, n_l2, tv, lds_stage);
}

// BASELINE config 3: dxyWindow + hetWindow (two genotype columns) over one position column and one
// window table — ONE stream of 26 B/site (p1,p2 f64 + n1,n2 i32 + g1,g2 i8).  The wave that owns level-2
// tile t of the dxy tree (8192 sites) also owns the SAME 8192 sites of both genotype columns: that is
// exactly one het work item (8 leaf tiles of 1024 sites, 8 KiB per column).  A genotype column's 8 KiB are
// requested in ONE burst of eight 16-byte loads per lane — column 0 before the first batch of dxy loads of the
// tile, column 1 before the batch in the middle — and awaited like every other column burst (one column at a
// time, see dxy_build_body); the byte counts go through the same packed-word popcounts as het_build_body and
// ONE wave reduction per leaf (nonmissing and nhet packed into one register: both are at most 1024).
// Measured in one process, interleaved (% of the HBM peak on 26 B/site at 10^8 / 10^9 sites).  First
// (profiles/r03/fused_ab.txt, all columns of a batch requested together): 8 dxy loads per batch + genotype bursts
// 79.6 / 80.8, 8 + one genotype load per batch 78.6 / 80.6, 16 dxy loads per batch 77.4 / 79.0.  Then
// (profiles/r03/dxy_ab_second.txt, fused_ab_awaited_box*.txt; three boxes): that best form 76.2 / 80.5, 77.3 / 76.9,
// 76.0 / 77.1 against THIS one — batches of 4 leaf tiles, every column burst awaited, the genotype burst on its own —
// 76.8 / 81.8, 76.8 / 78.2, 75.9 / 78.4: +1.3 points at 10^9 sites on every box, a tie at 10^8 and 1.25e8; the genotype
// burst travelling with p1's 75.3 / 81.1, batches of 8 leaf tiles 77.3 / 78.8, batches of 2 awaited 62 / 65.  The 2 x 8 level-1 het nodes of a tile are parked in the wave's
// LDS stage beside the tile's dxy row and leave with it (deferred stores, see NodeStage); level 2 of the het
// trees comes from tree_up_kernel as in the separate build.  Node values and tree layout are those of the
// separate kernels bit for bit (integer counts; the dxy arithmetic is the same code).
// (Round 2 had a CONCATENATION here: blockIdx.y = 0 ran the dxy body, 1 and 2 the het body — the het
// workgroups were scheduled behind the resident dxy ones and ran alone at the tail, every workgroup reserved
// 64 KiB of LDS whether it used it or not: 1.5 % SLOWER than the three separate launches.)
struct DxyHetBuildArgs {
    const double *p1, *p2;
    const int32_t *n1, *n2;
    const int8_t *g[2];
    uint64_t n;
    int minind;
    uint64_t n_l2_dxy;
    TreeView tv_dxy, tv_het;  // tv_het: the two genotype trees, pair_stride apart
};
constexpr size_t kDxyHetStageBytes = kFstStageBytes + (size_t)4 * kFstStage * 2 * kHetChunk * sizeof(NodeHet);  // + 2 KiB per wave

__device__ __forceinline__ uint32_t het_count_packed(const uint4 &w) {  // nonmissing | nhet << 16 of 16 genotypes
    uint32_t nm = 0, nh = 0;
    het_count_word(w.x, nm, nh);
    het_count_word(w.y, nm, nh);
    het_count_word(w.z, nm, nh);
    het_count_word(w.w, nm, nh);
    return nm | (nh << 16);
}

template <int U = 4>  // dxy leaf tiles per batch
__global__ __launch_bounds__(256) void dxy_het_build_kernel(DxyHetBuildArgs f) {
    static_assert(kRadix % (2 * U) == 0 && U % 2 == 0, "a batch starts in the middle of the tile; counts are read per pair of leaf tiles");
    static_assert((uint64_t)kLeafF64 * kRadix == (uint64_t)kLeafI8 * kHetChunk, "a dxy level-2 tile is one het work item");
    extern __shared__ __attribute__((aligned(16))) char lds_stage[];
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n = f.n;
    NodeDxy *__restrict__ l1 = reinterpret_cast<NodeDxy *>(f.tv_dxy.base + f.tv_dxy.off[0]);
    NodeDxy *__restrict__ l2 = reinterpret_cast<NodeDxy *>(f.tv_dxy.base + f.tv_dxy.off[1]);
    constexpr uint64_t kTile2 = (uint64_t)kLeafF64 * kRadix;  // 8192 sites
    NodeStage<NodeDxy, kFstStage, true> stage(lds_stage, wib, lane, l1, l2, n_waves);
    // het side of the stage: [tile of the stage][column][leaf] behind the four waves' dxy rows
    NodeHet *hstage = reinterpret_cast<NodeHet *>(lds_stage + kFstStageBytes) + (size_t)wib * kFstStage * 2 * kHetChunk;
    int hheld = 0;
    uint64_t hfirst = 0;
    auto hflush = [&]() {
        // 16 lanes: column = lane / 8, leaf = lane % 8 -> one 64-byte run per column and tile
        for (int k = 0; k < hheld; ++k) {
            const uint64_t t = hfirst + (uint64_t)k * n_waves;
            if (lane < 2 * kHetChunk) {
                NodeHet *dst = reinterpret_cast<NodeHet *>(f.tv_het.base + (size_t)(lane >> 3) * f.tv_het.pair_stride + f.tv_het.off[0]);
                dst[t * kHetChunk + (lane & 7)] = hstage[k * 2 * kHetChunk + lane];
            }
        }
        hheld = 0;
    };

    for (uint64_t t = wave0; t < f.n_l2_dxy; t += n_waves) {
        const uint64_t base = t * kTile2;
        NodeDxy keep{0.0, 0u, 0u};
        uint32_t hkeep = 0;  // lane c * 8 + u: packed counts of leaf u of genotype column c
        if (base + kTile2 <= n) {
            const double2 *__restrict__ q1 = reinterpret_cast<const double2 *>(f.p1 + base);
            const double2 *__restrict__ q2 = reinterpret_cast<const double2 *>(f.p2 + base);
            const int4 *__restrict__ m1 = reinterpret_cast<const int4 *>(f.n1 + base);  // 16-byte loads: see count_pair_pred
            const int4 *__restrict__ m2 = reinterpret_cast<const int4 *>(f.n2 + base);
            const uint4 *__restrict__ h0 = reinterpret_cast<const uint4 *>(f.g[0] + base);
            const uint4 *__restrict__ h1 = reinterpret_cast<const uint4 *>(f.g[1] + base);
            const int rot = tile_rotation<U>(wave0);  // see fst_column_sums (the genotype bursts keep their places in the walk)
#pragma unroll 1
            for (int j0 = 0; j0 < kRadix; j0 += U) {
                const int j = (j0 + rot) & (kRadix - 1);
                double2 x1[U], x2[U];
                int4 k1[U / 2], k2[U / 2];
                uint4 gb[kHetChunk];
                const bool burst = j0 == 0 || j0 == kRadix / 2;  // wave-uniform: genotype column 0 / 1
                if (burst) {  // the genotype column's 8 KiB: a burst of its own, awaited
#pragma unroll
                    for (int u = 0; u < kHetChunk; ++u) gb[u] = load16_nt((j0 == 0 ? h0 : h1) + u * kWave + lane);
                    load_fence((int)gb[kHetChunk - 1].w);
                }
                // the dxy columns one at a time, as in dxy_build_body
#pragma unroll
                for (int u = 0; u < U; ++u) x1[u] = load16<true>(q1 + (j + u) * kWave + lane);
                load_fence(x1[U - 1].y);
#pragma unroll
                for (int u = 0; u < U; ++u) x2[u] = load16<true>(q2 + (j + u) * kWave + lane);
                load_fence(x2[U - 1].y);
#pragma unroll
                for (int u = 0; u < U / 2; ++u) k1[u] = load16_nt(m1 + (j / 2 + u) * kWave + lane);
#pragma unroll
                for (int u = 0; u < U / 2; ++u) k2[u] = load16_nt(m2 + (j / 2 + u) * kWave + lane);
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    const PairPred pp = count_pair_pred(k1[u / 2], k2[u / 2], f.minind);
                    const NodeDxy a0 = dxy_leaf_node<0>(x1[u], x2[u], pp, lane);
                    const NodeDxy a1 = dxy_leaf_node<1>(x1[u + 1], x2[u + 1], pp, lane);
                    if (lane == j + u) keep = a0;
                    if (lane == j + u + 1) keep = a1;
                }
                if (burst) {
#pragma unroll
                    for (int u = 0; u < kHetChunk; ++u) {
                        const uint32_t c = wave_sum(het_count_packed(gb[u]));  // both fields <= 1024: no carry between them
                        if (lane == (j0 == 0 ? 0 : kHetChunk) + u) hkeep = c;
                    }
                }
            }
        } else {  // the last, partial tile: site by site, sites beyond n count as nothing
            for (int j = 0; j < kRadix; ++j) {
                const uint64_t tile0 = base + (uint64_t)j * kLeafF64;
                if (tile0 >= n) break;  // wave-uniform
                NodeDxy acc{0.0, 0u, 0u};
                for (int q = 0; q < 2; ++q) {
                    const uint64_t i = tile0 + 2 * lane + q;
                    if (i < n) dxy_acc(acc, dxy_site(f.p1[i], f.p2[i], f.n1[i], f.n2[i], f.minind));
                }
                acc = node_wave_sum(acc);
                if (lane == j) keep = acc;
            }
            for (int hb = 0; hb < 2 * kHetChunk; ++hb) {
                const int8_t *__restrict__ g = f.g[hb >> 3];
                const uint64_t tile0 = base + (uint64_t)(hb & 7) * kLeafI8;
                uint32_t nm = 0, nh = 0;
                for (int q = 0; q < 16; ++q) {
                    const uint64_t i = tile0 + (uint64_t)lane * 16 + q;
                    if (i < n) {
                        const int v = g[i];
                        nm += v >= 0;
                        nh += v == 1;
                    }
                }
                const uint32_t c = wave_sum(nm | (nh << 16));
                if (lane == hb) hkeep = c;
            }
        }
        if (hheld == 0) hfirst = t;
        if (lane < 2 * kHetChunk) hstage[hheld * 2 * kHetChunk + lane] = NodeHet{hkeep & 0xFFFFu, hkeep >> 16};
        ++hheld;
        const bool full = stage.held + 1 == kFstStage;
        stage.put(t, keep);  // flushes the dxy rows when the stage is full
        if (full) hflush();
    }
    stage.flush();
    hflush();
    if (lane == 0) reinterpret_cast<NodeDxy *>(f.tv_dxy.base + f.tv_dxy.partials)[wave0] = stage.sum;  // see dxy_build_body
}

// ------------------------------------------------------------------------------------------
// BUILD, extreme score (ihsWindow / xpehhWindow): level-1 {key max, first index, count beyond the
// cutoff} per 256 sites, level-2 per 16384 sites.                                 8 B/site read.
// ------------------------------------------------------------------------------------------
struct ExtBuildArgs { const double *s; int mode; double thr; };

// ------------------------------------------------------------------------------------------
// Upper levels (only exist when level 2 has more than 64 nodes): parent = Σ of 64 children.
// ------------------------------------------------------------------------------------------
template <class Node>
__global__ __launch_bounds__(256) void tree_up_kernel(TreeView tv, int child_level /*0-based slot*/,
                                                      uint64_t n_child, uint64_t n_parent) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    char *tree = tv.base + (size_t)blockIdx.y * tv.pair_stride;
    const Node *__restrict__ child = reinterpret_cast<const Node *>(tree + tv.off[child_level]);
    Node *__restrict__ parent = reinterpret_cast<Node *>(tree + tv.off[child_level + 1]);
    for (uint64_t p = wave0; p < n_parent; p += n_waves) {
        const uint64_t i = p * kRadix + lane;
        Node v = node_identity<Node>();
        if (i < n_child) v = child[i];
        v = node_wave_sum(v);
        if (lane == 0) parent[p] = v;
    }
}

// ------------------------------------------------------------------------------------------
// QUERY: one wave per window.
// ------------------------------------------------------------------------------------------
struct FstTraits {
    using Node = NodeFst;
    using Row = pgt_fst_row;
    static constexpr int kLeaf = kLeafF64;
    struct Args { PairCols cols; };
    struct Cols { const double *a, *b; };
    static __device__ __forceinline__ Cols cols(const Args &g, int pair) { return {g.cols.a[pair], g.cols.b[pair]}; }
    static __device__ __forceinline__ Node leaf(const Cols &c, uint64_t i) { return {c.a[i], c.b[i]}; }
    // sites i0 (even) and i0 + 1 by one 16-byte load per column (the columns are 16-byte aligned)
    static __device__ __forceinline__ void leaf_pair(const Cols &c, uint64_t i0, Node &v0, Node &v1) {
        const double2 a2 = *reinterpret_cast<const double2 *>(c.a + i0), b2 = *reinterpret_cast<const double2 *>(c.b + i0);
        v0 = {a2.x, b2.x};
        v1 = {a2.y, b2.y};
    }
    static __device__ __forceinline__ void sum_sites(Node &acc, const Cols &c, uint64_t from, uint64_t to, int lane,
                                                     uint64_t) {
        for (uint64_t i = from + lane; i < to; i += kWave) node_add(acc, leaf(c, i));
    }
    static __device__ __forceinline__ void finish(Row *out, const Node &t, uint32_t start, uint32_t end,
                                                  uint64_t lo, uint64_t hi, const Cols &, const uint32_t *) {
        Row r;
        r.start = start;
        r.end = end;
        r.mid = (uint32_t)(start + end) / 2u;  // unsigned 32-bit wrap, fstWindow.cpp:73
        r.n = (uint32_t)(hi - lo);
        // the reference starts its sums at +0.0 (fstWindow.cpp:76-77): a window of only -0.0
        // must give +0.0, which adding +0.0 restores without touching any other value
        r.asum = t.x + 0.0;
        r.bsum = t.y + 0.0;
        r.fst = r.bsum != 0.0 ? r.asum / r.bsum : 0.0;  // fstWindow.cpp:85
        *out = r;
    }
    static __device__ __forceinline__ void store_total(pgt_dxy_total *, const Node &) {}
};

struct HetTraits {
    using Node = NodeHet;
    using Row = pgt_het_row;
    static constexpr int kLeaf = kLeafI8;
    struct Args { const int8_t *g; };
    using Cols = const int8_t *;
    static __device__ __forceinline__ Cols cols(const Args &g, int) { return g.g; }
    static __device__ __forceinline__ Node leaf(const Cols &c, uint64_t i) {
        const int v = c[i];
        return {(uint32_t)(v >= 0), (uint32_t)(v == 1)};
    }
    // Ragged site ranges (< 1024 bytes each side): aligned 16-byte loads with the bytes outside
    // [from,to) forced to "missing" (0x80), instead of one byte per lane and iteration.
    static __device__ __forceinline__ uint32_t byte_mask(int64_t l, int64_t h) {  // bytes [l,h) of a word
        l = l < 0 ? 0 : (l > 4 ? 4 : l);
        h = h < 0 ? 0 : (h > 4 ? 4 : h);
        if (h <= l) return 0u;
        const uint32_t ones = (h - l) == 4 ? 0xFFFFFFFFu : ((1u << (8 * (int)(h - l))) - 1u);
        return ones << (8 * (int)l);
    }
    static __device__ __forceinline__ void sum_sites(Node &acc, const Cols &c, uint64_t from, uint64_t to, int lane,
                                                     uint64_t n_sites) {
        if (from >= to) return;
        for (uint64_t base = (from & ~15ull) + 16ull * lane; base < to; base += 16ull * kWave) {
            if (base + 16 <= n_sites) {  // the column is 16-byte aligned (checked by the API)
                const uint4 w = *reinterpret_cast<const uint4 *>(c + base);
                const int64_t l = (int64_t)from - (int64_t)base, h = (int64_t)to - (int64_t)base;
                const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t m = byte_mask(l - 4 * j, h - 4 * j);
                    het_count_word((wd[j] & m) | (~m & 0x80808080u), acc.nonmiss, acc.nhet);
                }
            } else {  // last, partial 16 bytes of the column
                for (uint64_t i = base > from ? base : from; i < to && i < base + 16; ++i) node_add(acc, leaf(c, i));
            }
        }
    }
    // The two ragged ends of a window, [l0,l1) and [r0,r1), each shorter than a leaf: one 16-byte load per lane and side, BOTH
    // requested before either is counted (the counts are integers: the order of the additions is immaterial).  A side that
    // needs a second block per lane (it starts late in its first 16 bytes) or whose last block would reach beyond the
    // column takes sum_sites; the decision is wave-uniform.
    static __device__ __forceinline__ void sum_two_ranges(Node &acc, const Cols &c, uint64_t l0, uint64_t l1, uint64_t r0,
                                                          uint64_t r1, int lane, uint64_t n_sites) {
        const uint64_t lbase = l0 & ~15ull, rbase = r0 & ~15ull;
        const bool simple = ((l1 + 15) & ~15ull) <= n_sites && ((r1 + 15) & ~15ull) <= n_sites &&
                            l1 - lbase <= 16ull * kWave && r1 - rbase <= 16ull * kWave;
        if (!simple) {
            sum_sites(acc, c, l0, l1, lane, n_sites);
            sum_sites(acc, c, r0, r1, lane, n_sites);
            return;
        }
        const uint64_t lb = lbase + 16ull * lane, rb = rbase + 16ull * lane;
        const bool inl = l0 < l1 && lb < l1, inr = r0 < r1 && rb < r1;
        const uint4 missing{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
        const uint4 wl = inl ? *reinterpret_cast<const uint4 *>(c + lb) : missing;
        const uint4 wr = inr ? *reinterpret_cast<const uint4 *>(c + rb) : missing;
        auto count = [&](const uint4 &w, bool in, uint64_t from, uint64_t to, uint64_t base) {
            const int64_t l = in ? (int64_t)from - (int64_t)base : 0, h = in ? (int64_t)to - (int64_t)base : 0;
            const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t m = byte_mask(l - 4 * j, h - 4 * j);
                het_count_word((wd[j] & m) | (~m & 0x80808080u), acc.nonmiss, acc.nhet);
            }
        };
        count(wl, inl, l0, l1, lb);
        count(wr, inr, r0, r1, rb);
    }
    static __device__ __forceinline__ void finish(Row *out, const Node &t, uint32_t start, uint32_t end,
                                                  uint64_t, uint64_t, const Cols &, const uint32_t *) {
        Row r;
        r.start = start;
        r.end = end;
        r.mid = (uint32_t)(start + end) / 2u;  // hetWindow.cpp:70
        r.nonmissing = t.nonmiss;
        r.nhet = t.nhet;
        r.pad_ = 0;
        r.h = t.nonmiss != 0 ? (double)t.nhet / (double)t.nonmiss : 0.0;  // hetWindow.cpp:84
        *out = r;
    }
    static __device__ __forceinline__ void store_total(pgt_dxy_total *, const Node &) {}
};

struct DxyTraits {
    using Node = NodeDxy;
    using Row = pgt_dxy_row;
    static constexpr int kLeaf = kLeafF64;
    struct Args { const double *p1, *p2; const int32_t *n1, *n2; int minind; };
    using Cols = Args;
    static __device__ __forceinline__ Cols cols(const Args &g, int) { return g; }
    static __device__ __forceinline__ Node leaf(const Cols &c, uint64_t i) {
        Node v{0.0, 0u, 0u};
        dxy_acc(v, dxy_site(c.p1[i], c.p2[i], c.n1[i], c.n2[i], c.minind));
        return v;
    }
    static __device__ __forceinline__ void leaf_pair(const Cols &c, uint64_t i0, Node &v0, Node &v1) {  // i0 even
        const double2 x1 = *reinterpret_cast<const double2 *>(c.p1 + i0), x2 = *reinterpret_cast<const double2 *>(c.p2 + i0);
        const int2 k1 = *reinterpret_cast<const int2 *>(c.n1 + i0), k2 = *reinterpret_cast<const int2 *>(c.n2 + i0);
        v0 = Node{0.0, 0u, 0u};
        v1 = Node{0.0, 0u, 0u};
        dxy_acc(v0, dxy_site(x1.x, x2.x, k1.x, k2.x, c.minind));
        dxy_acc(v1, dxy_site(x1.y, x2.y, k1.y, k2.y, c.minind));
    }
    static __device__ __forceinline__ void sum_sites(Node &acc, const Cols &c, uint64_t from, uint64_t to, int lane,
                                                     uint64_t) {
        for (uint64_t i = from + lane; i < to; i += kWave) node_add(acc, leaf(c, i));
    }
    static __device__ __forceinline__ void finish(Row *out, const Node &t, uint32_t start, uint32_t end,
                                                  uint64_t, uint64_t, const Cols &, const uint32_t *) {
        Row r;
        r.start = start;
        r.end = end;
        r.neff = t.neff;
        r.nskip = t.nskip;
        r.sum = t.s + 0.0;
        *out = r;
    }
    // genome-wide line, dxyWindow.cpp:382-385,429-433 (equal to Σ over d != -9 for freq in [0,1])
    static __device__ __forceinline__ void store_total(pgt_dxy_total *tot, const Node &t) {
        tot->sum = t.s + 0.0;
        tot->neff = t.neff;
        tot->nskip = t.nskip;
    }
};

struct ExtTraits {
    using Node = NodeExt;
    using Row = pgt_ext_row;
    static constexpr int kLeaf = kLeafExt;
    struct Args { const double *s; int mode; double thr; };
    using Cols = Args;
    static __device__ __forceinline__ Cols cols(const Args &g, int) { return g; }
    static __device__ __forceinline__ double key_of(double s, int mode) {
        return mode == PGT_EXT_IHS ? fabs(s) : (mode == PGT_EXT_XP_MAX ? s : -s);
    }
    static __device__ __forceinline__ Node leaf(const Cols &c, uint64_t i) {
        const double k = key_of(c.s[i], c.mode);
        return {k, (uint32_t)i, (uint32_t)(k > c.thr)};  // ihsWindow.cpp:203, xpehhWindow.cpp:212,215
    }
    static __device__ __forceinline__ void sum_sites(Node &acc, const Cols &c, uint64_t from, uint64_t to, int lane,
                                                     uint64_t) {
        for (uint64_t i = from + lane; i < to; i += kWave) node_add(acc, leaf(c, i));
    }
    static __device__ __forceinline__ void finish(Row *out, const Node &t, uint32_t start, uint32_t end, uint64_t lo,
                                                  uint64_t hi, const Cols &c, const uint32_t *pos) {
        Row r;
        r.start = start;
        r.end = end;
        r.nsites = (uint32_t)(hi - lo);
        r.nbig = t.count;
        r.pad_ = 0;
        const bool any = hi > lo && t.idx != 0xFFFFFFFFu;
        r.position = any ? pos[t.idx] : 0u;   // maxihs[2], ihsWindow.cpp:98
        r.value = any ? c.s[t.idx] : 0.0;     // maxihs[1]: the signed score itself
        *out = r;
    }
    static __device__ __forceinline__ void store_total(pgt_dxy_total *, const Node &) {}
};

// One 256-site leaf tile -> its NodeExt, wave-uniform.  The wave owns 4 keys per lane, site offsets
// h*128 + 2*lane + {0,1} (h = 0,1).  Instead of carrying {key, idx, count} through a three-field
// butterfly (six steps of 4 cross-lane moves + a two-level compare), only the KEY is max-reduced across
// lanes; the count is four ballots + s_bcnt1 and the first site attaining the maximum comes from four
// equality ballots + s_ff1 — scalar instructions on wave-uniform masks, no cross-lane traffic.
// Semantics are those of node_add (ties -> smallest site index; a NaN key never wins; a tile without
// any comparable key stays the identity {-inf, none}); the result is bit-identical to the butterfly's.
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<kDppQuadXor1>(v));
    v = fmax(v, dpp_f64<kDppQuadXor2>(v));
    v = fmax(v, dpp_f64<kDppRowHalfMirror>(v));
    v = fmax(v, dpp_f64<kDppRowMirror>(v));
    v = fmax(v, __shfl_xor(v, 16, kWave));
    v = fmax(v, __shfl_xor(v, 32, kWave));
    return v;
}
__device__ __forceinline__ NodeExt ext_leaf_tile(const double (&k)[4], bool (&valid)[4], double thr, uint64_t tile0, int lane) {
    const double ninf = -__builtin_huge_val();
    uint32_t count = 0;
    double m = ninf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        count += (uint32_t)__popcll(__ballot(valid[q] && k[q] > thr));
        m = fmax(m, valid[q] ? k[q] : ninf);  // fmax drops a NaN operand (v_max_f64)
    }
    const double wmax = wave_max(m);
    unsigned long long eq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) eq[q] = __ballot(valid[q] && k[q] == wmax);
    uint32_t idx = 0xFFFFFFFFu;
    double key = ninf;
    const unsigned long long m0 = eq[0] | eq[1], m1 = eq[2] | eq[3];
    if (m0 != 0) {  // wave-uniform: the first half of the tile holds the first occurrence
        const int L = __ffsll((long long)m0) - 1;
        idx = (uint32_t)(tile0 + 2 * (uint64_t)L + (((eq[0] >> L) & 1ull) ? 0 : 1));
        key = wmax;
    } else if (m1 != 0) {
        const int L = __ffsll((long long)m1) - 1;
        idx = (uint32_t)(tile0 + 2 * kWave + 2 * (uint64_t)L + (((eq[2] >> L) & 1ull) ? 0 : 1));
        key = wmax;
    }
    (void)lane;
    return NodeExt{key, idx, count};
}

constexpr int kExtStage = 16;  // tiles staged per wave (16 KiB of LDS): 64 KiB per workgroup -> 2 per CU, as the fst build
constexpr int kExtStageSmall = 8;  // short inputs: 8 KiB per wave, 32 KiB per workgroup -> 4 per CU = 16 waves (ext_build_launch)
constexpr uint64_t kExtSmallTiles = 20000;  // level-2 tiles of 16384 sites (3.3e8 sites) up to which an input counts as short
template <int STAGE = kExtStage, int UNROLL = 4, bool DEFER = true>
__global__ __launch_bounds__(256) void ext_build_kernel(ExtBuildArgs g, uint64_t n, uint64_t n_l2, TreeView tv) {
    extern __shared__ __attribute__((aligned(16))) char lds_stage[];
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    NodeExt *__restrict__ l1 = reinterpret_cast<NodeExt *>(tv.base + tv.off[0]);
    NodeExt *__restrict__ l2 = reinterpret_cast<NodeExt *>(tv.base + tv.off[1]);
    constexpr uint64_t kTile2 = (uint64_t)kLeafExt * kRadix;  // 16384 sites
    NodeStage<NodeExt, STAGE, true> stage(lds_stage, threadIdx.x >> 6, lane, l1, l2, n_waves);
    for (uint64_t t = wave0; t < n_l2; t += n_waves) {
        const uint64_t base = t * kTile2;
        NodeExt keep = node_identity<NodeExt>();
        if (base + kTile2 <= n) {
            const double2 *__restrict__ ps = reinterpret_cast<const double2 *>(g.s + base);
            const int rot = tile_rotation<UNROLL>(wave0);  // see fst_column_sums
#pragma unroll 1
            for (int j0 = 0; j0 < kRadix; j0 += UNROLL) {
                const int j = (j0 + rot) & (kRadix - 1);
                double2 v[UNROLL][2];  // leaf tile = 256 sites = two 1-KiB wave loads; 2*UNROLL loads in flight per lane
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[u][h] = load16<true>(ps + ((j + u) * 2 + h) * kWave + lane);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const double k[4] = {ExtTraits::key_of(v[u][0].x, g.mode), ExtTraits::key_of(v[u][0].y, g.mode),
                                         ExtTraits::key_of(v[u][1].x, g.mode), ExtTraits::key_of(v[u][1].y, g.mode)};
                    bool valid[4] = {true, true, true, true};
                    const NodeExt a = ext_leaf_tile(k, valid, g.thr, base + (uint64_t)(j + u) * kLeafExt, lane);
                    if (lane == j + u) keep = a;
                }
            }
        } else {  // last, partial level-2 tile
            for (int j = 0; j < kRadix; ++j) {
                const uint64_t tile0 = base + (uint64_t)j * kLeafExt;
                if (tile0 >= n) break;  // wave-uniform
                double k[4];
                bool valid[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t i = tile0 + (uint64_t)(q >> 1) * 2 * kWave + 2 * lane + (q & 1);
                    valid[q] = i < n;
                    k[q] = valid[q] ? ExtTraits::key_of(g.s[i], g.mode) : 0.0;
                }
                const NodeExt a = ext_leaf_tile(k, valid, g.thr, tile0, lane);
                if (lane == j) keep = a;
            }
        }
        if constexpr (DEFER) {
            stage.put(t, keep);
        } else {
            l1[t * kRadix + lane] = keep;
            const NodeExt tot = node_wave_sum(keep);
            if (lane == 0) l2[t] = tot;
        }
    }
    if constexpr (DEFER) stage.flush();
}

template <class Tr>
__device__ __forceinline__ void sum_level(typename Tr::Node &acc, const typename Tr::Cols &c,
                                          const char *tree, const TreeView &tv, int level,
                                          uint64_t from, uint64_t to, int lane, uint64_t n_sites) {
    if (level == 0) {
        Tr::sum_sites(acc, c, from, to, lane, n_sites);
    } else {
        const typename Tr::Node *nodes = reinterpret_cast<const typename Tr::Node *>(tree + tv.off[level - 1]);
        for (uint64_t i = from + lane; i < to; i += kWave) node_add(acc, nodes[i]);
    }
}

// The ragged left [l0,l1) and right [r0,r1) remainders of one level, each shorter than that level's
// radix.  Loop-free: every lane issues its (at most 4 site or 2 node) loads unconditionally from a
// clamped, always valid index and masks the value afterwards, so all loads of the level are in
// flight together instead of one dependent load->add round trip per loop iteration (the query is
// latency-bound: 0.094 -> see profiles/ for the effect).  Accumulation order is unchanged.
template <class Tr>
__device__ __forceinline__ void ragged_pair(typename Tr::Node &acc, const typename Tr::Cols &c, const char *tree,
                                            const TreeView &tv, int level, uint64_t l0, uint64_t l1, uint64_t r0,
                                            uint64_t r1, int lane, uint64_t n_sites) {
    using Node = typename Tr::Node;
    if (l0 >= l1 && r0 >= r1) return;  // wave-uniform
    const Node none = node_identity<Node>();
    if (level == 0) {
        if constexpr (Tr::kLeaf <= 4 * kWave) {
            constexpr int kSlots = Tr::kLeaf / kWave;  // per side: 2 for 128-site leaves, 4 for 256
            const uint64_t last = n_sites - 1;  // some valid site: a non-empty range implies n_sites > 0
            Node v[2 * kSlots];
#pragma unroll
            for (int u = 0; u < 2 * kSlots; ++u) {
                const uint64_t i = (u < kSlots ? l0 : r0) + lane + (uint64_t)(u % kSlots) * kWave;
                v[u] = Tr::leaf(c, i < (u < kSlots ? l1 : r1) ? i : last);
            }
#pragma unroll
            for (int u = 0; u < 2 * kSlots; ++u) {
                const uint64_t i = (u < kSlots ? l0 : r0) + lane + (uint64_t)(u % kSlots) * kWave;
                node_add(acc, i < (u < kSlots ? l1 : r1) ? v[u] : none);
            }
        } else {  // int8 genotypes: up to 1023 sites per side, 16 per lane and load (own vector path)
            Tr::sum_sites(acc, c, l0, l1, lane, n_sites);
            Tr::sum_sites(acc, c, r0, r1, lane, n_sites);
        }
    } else {
        const Node *nodes = reinterpret_cast<const Node *>(tree + tv.off[level - 1]);
        const uint64_t il = l0 + lane, ir = r0 + lane;
        const Node vl = nodes[il < l1 ? il : 0], vr = nodes[ir < r1 ? ir : 0];  // node 0 always exists
        node_add(acc, il < l1 ? vl : none);
        node_add(acc, ir < r1 ? vr : none);
    }
}

// Wave-wide reduction of the site range [lo,hi): every lane returns a partial (node_wave_sum of it is
// the range's node).  The accumulation order depends on the range only, never on the window table.
// (Round 2, tried and dropped: planning the whole descent first in scalar registers and issuing the loads
// of ALL levels before the first addition, with the table read by scalar loads one window ahead — rows
// bit-identical, but 0.095 instead of 0.085 ms for the 10^5 windows of the headline run and 0.76 instead of
// 0.66 ms at S = 100: the kernel moves ~3.3 KB per window and runs at the equivalent of 6 TB/s, it is not
// waiting on round trips, and the plan costs 27 VGPRs of occupancy.  profiles/r02/measure_query_pipelined.md)
template <class Tr>
__device__ __forceinline__ typename Tr::Node range_partial(const typename Tr::Cols &c, const char *tree, const TreeView &tv,
                                                           uint64_t lo, uint64_t hi, int lane, uint64_t n_sites) {
    typename Tr::Node acc = node_identity<typename Tr::Node>();
    uint64_t clo = lo, chi = hi;  // current range, in nodes of level k (level 0 = sites)
    for (int k = 0;; ++k) {
        if (k == tv.n_levels) {  // top level: whatever is left
            sum_level<Tr>(acc, c, tree, tv, k, clo, chi, lane, n_sites);
            break;
        }
        const uint64_t r = k == 0 ? (uint64_t)Tr::kLeaf : (uint64_t)kRadix;
        const uint64_t ulo = (clo + r - 1) / r, uhi = chi / r;
        if (ulo >= uhi) {  // no whole parent inside: finish at this level
            sum_level<Tr>(acc, c, tree, tv, k, clo, chi, lane, n_sites);
            break;
        }
        ragged_pair<Tr>(acc, c, tree, tv, k, clo, ulo * r, uhi * r, chi, lane, n_sites);  // < r nodes per side
        clo = ulo;
        chi = uhi;
    }
    return acc;
}

// The sum over ALL sites from the partial sums the build waves left (dxy: the genome-wide line); every lane returns a
// partial, node_wave_sum of it is the total.  Fixed order: lane l adds partials l, l + 64, ... in turn.
template <class Tr>
__device__ __forceinline__ typename Tr::Node total_partial(const char *tree, const TreeView &tv, int lane) {
    using Node = typename Tr::Node;
    const Node *__restrict__ p = reinterpret_cast<const Node *>(tree + tv.partials);
    Node acc = node_identity<Node>();
    constexpr int kAhead = 8;  // loads in flight per lane: 2048 partials are 32 per lane
    for (uint32_t i0 = 0; i0 < tv.n_partials; i0 += kAhead * kWave) {
        Node v[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kWave + (uint32_t)lane;
            v[u] = p[i < tv.n_partials ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kWave + (uint32_t)lane;
            node_add(acc, i < tv.n_partials ? v[u] : node_identity<Node>());
        }
    }
    return acc;
}

// THE COMMON DESCENT WITH ALL ITS LOADS UP FRONT (round 5).  range_partial asks for one level's ragged nodes, adds them, and
// only then works out the next level: three dependent round trips per window, and the per-window query is a chain of round
// trips — at 10^4 windows (10^8 sites, the 8-GPU shard) there is about one window per resident wave and the kernel's time IS
// that chain.  For the shape nearly every window of a table has — the tree built to exactly two levels (pgt_set_max_window
// below a level-3 node), the window spanning at least one whole level-2 node and at most 64 of them — every address follows
// from (lo, hi) alone, so the ragged sites, the ragged level-1 nodes and the level-2 run are all requested before the first
// addition.  The additions are range_partial's, in its order (sites left / right, level-1 left / right, level 2; a masked
// slot adds the identity, which changes no bit: the accumulator starts at +0.0 and can never be -0.0): rows unchanged.
// Any other shape returns false and takes range_partial.  (Round 2 had planned the WHOLE descent for any depth in scalar
// registers: 27 more VGPRs, slower at 10^5 windows, profiles/r02/measure_query_pipelined.md; this form has no plan to keep.)
template <class Tr>
__device__ __forceinline__ bool range_partial_two_levels(typename Tr::Node &out, const typename Tr::Cols &c, const char *tree,
                                                         const TreeView &tv, uint64_t lo, uint64_t hi, int lane, uint64_t n_sites) {
    using Node = typename Tr::Node;
    if constexpr (Tr::kLeaf == kLeafI8) {
        // the int8 tree built to ONE level (windows shorter than a 65536-site level-2 node): the level-1 run [a1, b1) is requested
        // first, then the two ragged ends together (HetTraits::sum_two_ranges); integer counts, any order
        constexpr uint64_t kLeaf = (uint64_t)Tr::kLeaf;
        if (tv.n_levels != 1) return false;
        const uint64_t a1 = (lo + kLeaf - 1) / kLeaf, b1 = hi / kLeaf;
        if (a1 >= b1 || b1 - a1 > (uint64_t)kWave) return false;
        const Node *__restrict__ n1 = reinterpret_cast<const Node *>(tree + tv.off[0]);
        const uint64_t it = a1 + lane;
        const Node vt = n1[it < b1 ? it : 0];
        Node acc = node_identity<Node>();
        Tr::sum_two_ranges(acc, c, lo, a1 * kLeaf, b1 * kLeaf, hi, lane, n_sites);
        node_add(acc, it < b1 ? vt : node_identity<Node>());
        out = acc;
        return true;
    } else if constexpr (Tr::kLeaf != kLeafF64) {
        return false;  // the extreme-score tree (256-site leaves: four slots a side) keeps range_partial: the front-loaded form needs 66 VGPRs there
    } else {
        constexpr uint64_t kLeaf = (uint64_t)Tr::kLeaf;
        constexpr int kSlots = Tr::kLeaf / kWave;
        if (tv.n_levels != 2) return false;
        const uint64_t a1 = (lo + kLeaf - 1) / kLeaf, b1 = hi / kLeaf;  // whole level-1 nodes [a1, b1)
        if (a1 >= b1) return false;
        const uint64_t a2 = (a1 + kRadix - 1) / kRadix, b2 = b1 / kRadix;  // whole level-2 nodes [a2, b2)
        if (a2 >= b2 || b2 - a2 > (uint64_t)kWave) return false;
        const Node none = node_identity<Node>();
        const Node *__restrict__ n1 = reinterpret_cast<const Node *>(tree + tv.off[0]);
        const Node *__restrict__ n2 = reinterpret_cast<const Node *>(tree + tv.off[1]);
        // requests: sites [lo, a1 * leaf) and [b1 * leaf, hi); level-1 nodes [a1, a2 * 64) and [b2 * 64, b1); level-2 nodes [a2, b2)
        const uint64_t l0 = lo, l1 = a1 * kLeaf, r0 = b1 * kLeaf, r1 = hi, last = n_sites - 1;
        Node v[2 * kSlots];
#pragma unroll
        for (int u = 0; u < 2 * kSlots; ++u) {
            const uint64_t i = (u < kSlots ? l0 : r0) + lane + (uint64_t)(u % kSlots) * kWave;
            v[u] = Tr::leaf(c, i < (u < kSlots ? l1 : r1) ? i : last);
        }
        const uint64_t il = a1 + lane, ir = b2 * kRadix + lane, it = a2 + lane;
        const Node vl = n1[il < a2 * kRadix ? il : 0], vr = n1[ir < b1 ? ir : 0], vt = n2[it < b2 ? it : 0];  // node 0 always exists
        Node acc = none;
#pragma unroll
        for (int u = 0; u < 2 * kSlots; ++u) {
            const uint64_t i = (u < kSlots ? l0 : r0) + lane + (uint64_t)(u % kSlots) * kWave;
            node_add(acc, i < (u < kSlots ? l1 : r1) ? v[u] : none);
        }
        node_add(acc, il < a2 * kRadix ? vl : none);
        node_add(acc, ir < b1 ? vr : none);
        node_add(acc, it < b2 ? vt : none);
        out = acc;
        return true;
    }
}

template <class Tr>
__device__ __forceinline__ void query_body(const typename Tr::Args &args, const uint32_t *__restrict__ pos,
                                           const TreeView &tv, const pgt_win *__restrict__ win, uint64_t n_win,
                                           typename Tr::Row *__restrict__ out, uint64_t n_sites,
                                           pgt_dxy_total *tot, int pair) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const typename Tr::Cols c = Tr::cols(args, pair);
    const char *tree = tv.base + (size_t)pair * tv.pair_stride;
    const uint64_t n_items = n_win + (tot ? 1 : 0);  // the extra item is the genome-wide total

    for (uint64_t w = wave0; w < n_items; w += n_waves) {
        const bool is_total = w == n_win;
        pgt_win wd;
        if (is_total) {
            wd.lo = 0; wd.hi = n_sites; wd.flags = PGT_WIN_COORDS; wd.start = wd.end = 0; wd.label_run = 0;
        } else {
            wd = win[w];
        }
        if (is_total) {  // wave-uniform
            const typename Tr::Node acc = node_wave_sum(total_partial<Tr>(tree, tv, lane));
            if (lane == 0) Tr::store_total(tot, acc);
            continue;
        }
        // clamp to the columns so that a corrupt table can never fault the GPU
        const uint64_t hi = wd.hi < n_sites ? wd.hi : n_sites;
        const uint64_t lo = wd.lo < hi ? wd.lo : hi;
        // the window's coordinates are requested BEFORE the descent (they depend on the table row alone): their round trip
        // overlaps the tree's instead of following it (round 5; the kernel is a chain of dependent round trips per window)
        uint32_t start = wd.start, end = wd.end;
        if (!(wd.flags & PGT_WIN_COORDS)) {  // fstWindow.cpp:71-72
            start = hi > lo ? pos[lo] : 0u;
            end = hi > lo ? pos[hi - 1] : 0u;
        }
        typename Tr::Node part;
        if (!range_partial_two_levels<Tr>(part, c, tree, tv, lo, hi, lane, n_sites))  // wave-uniform
            part = range_partial<Tr>(c, tree, tv, lo, hi, lane, n_sites);
        const typename Tr::Node acc = node_wave_sum(part);
        if (lane == 0) Tr::finish(out + (uint64_t)pair * n_win + w, acc, start, end, lo, hi, c, pos);
    }
}

// ------------------------------------------------------------------------------------------
// QUERY, sliding form (pgt_set_window_step <= 32): one wave answers `group` CONSECUTIVE windows, one per
// lane.  A window [lo,hi) is split at the 128-site grid:
//     [lo, A)   the rest of lo's 128-site tile      -> a SUFFIX scan of that tile, shared by the group
//     [A, B)    whole tiles                         -> ONE wave-wide range query per distinct (A,B) of the
//                                                     group (at step 1: 2-3 per 64 windows)
//     [B, hi)   the start of hi's tile              -> a PREFIX scan of that tile, shared by the group
// so a group costs 4 tiles of sites (the two tiles under its starts, the two under its ends) plus a few
// tree queries instead of one tree query (~3.3 KB of ragged reads) per window: the regime of
// `-winsize W -stepsize 1`, where the reference re-sums W sites per window (fstWindow.cpp:80-83) and
// shifts W-S (fstWindow.cpp:95-99).  Each of the three pieces is computed in an order that depends on
// the window alone (whole-tile scans in a fixed lane order; the interior by range_partial), never on
// which other windows share the wave: rows are bitwise independent of the grouping, hence of the
// number of GPUs a table is sharded over.  Windows that start or end outside the group's two tiles take
// the same three pieces from scans of their OWN tiles, one by one, after the group's common work; windows
// inside one tile (nothing to split) and windows of at most 32 sites are summed by their lane alone.
// ------------------------------------------------------------------------------------------
constexpr int kSlideTile = 128;
constexpr int kSlideDirect = 32;  // windows of at most this many sites are summed by their lane alone, site by site
// Exclusive scans over the lanes in a fixed order; UP = sum of the lanes ABOVE, else of the lanes BELOW.
// Inside each 16-lane row: four Hillis-Steele steps with DPP row shifts (lanes without a source add the
// identity); then every lane adds the totals of the rows before (after) its own, read from lanes 15/31/47
// (0/16/32/48 for UP) into scalar registers and added nearest row first; a one-lane wave shift turns inclusive
// into exclusive.  All of it on the VALU: the first version's 28 ds_bpermute per scan went through LDS, which
// the scan tables of this kernel need for themselves.  (Nodes of the sliding query have an all-zero identity.)
template <class Node, int CTRL>
__device__ __forceinline__ Node node_dpp(const Node &v) {  // lanes the DPP control gives no source keep 0 = the identity
    constexpr int kWords = sizeof(Node) / 4;
    int w[kWords];
    __builtin_memcpy(w, &v, sizeof(Node));
#pragma unroll
    for (int k = 0; k < kWords; ++k) w[k] = __builtin_amdgcn_update_dpp(0, w[k], CTRL, 0xF, 0xF, false);
    Node o;
    __builtin_memcpy(&o, w, sizeof(Node));
    return o;
}
template <class Node>
__device__ __forceinline__ Node node_readlane(const Node &v, int src_lane) {  // wave-uniform result
    constexpr int kWords = sizeof(Node) / 4;
    int w[kWords];
    __builtin_memcpy(w, &v, sizeof(Node));
#pragma unroll
    for (int k = 0; k < kWords; ++k) w[k] = __builtin_amdgcn_readlane(w[k], src_lane);
    Node o;
    __builtin_memcpy(&o, w, sizeof(Node));
    return o;
}
template <class Node, bool UP>
__device__ __forceinline__ Node lane_scan_exclusive(Node v, int lane) {
    const Node none = node_identity<Node>();
    // inclusive scan inside the row: row_shl:n = 0x100 + n reads lane + n, row_shr:n = 0x110 + n reads lane - n
    node_add(v, UP ? node_dpp<Node, 0x101>(v) : node_dpp<Node, 0x111>(v));
    node_add(v, UP ? node_dpp<Node, 0x102>(v) : node_dpp<Node, 0x112>(v));
    node_add(v, UP ? node_dpp<Node, 0x104>(v) : node_dpp<Node, 0x114>(v));
    node_add(v, UP ? node_dpp<Node, 0x108>(v) : node_dpp<Node, 0x118>(v));
    // totals of the four rows (the lane where the row's inclusive scan ends)
    const Node r0 = node_readlane(v, UP ? 0 : 15), r1 = node_readlane(v, UP ? 16 : 31), r2 = node_readlane(v, UP ? 32 : 47),
               r3 = node_readlane(v, UP ? 48 : 63);
    const int row = lane >> 4;
    if (UP) {  // rows above mine, nearest first
        node_add(v, row < 1 ? r1 : none);
        node_add(v, row < 2 ? r2 : none);
        node_add(v, row < 3 ? r3 : none);
    } else {   // rows below mine, nearest first
        node_add(v, row > 2 ? r2 : none);
        node_add(v, row > 1 ? r1 : none);
        node_add(v, row > 0 ? r0 : none);
    }
    // inclusive -> exclusive: the neighbour's inclusive value (wave_shl:1 = 0x130 reads lane + 1, wave_shr:1 = 0x138 lane - 1)
    return UP ? node_dpp<Node, 0x130>(v) : node_dpp<Node, 0x138>(v);
}

template <class Tr>
__device__ __forceinline__ void query_slide_body(const typename Tr::Args &args, const uint32_t *__restrict__ pos,
                                                 const TreeView &tv, const pgt_win *__restrict__ win, uint64_t n_win,
                                                 typename Tr::Row *__restrict__ out, uint64_t n_sites, pgt_dxy_total *tot,
                                                 int pair, uint32_t group, char *lds) {
    using Node = typename Tr::Node;
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const typename Tr::Cols c = Tr::cols(args, pair);
    const char *tree = tv.base + (size_t)pair * tv.pair_stride;
    const uint64_t n_groups = (n_win + group - 1) / group;
    const uint64_t n_tasks = n_groups + (tot ? 1 : 0);
    // this wave's scan tables: [side: 0 suffix under the starts, 1 prefix under the ends][tile 0,1][site]
    Node *span = reinterpret_cast<Node *>(lds) + (size_t)wib * 4 * kSlideTile;
    const Node none = node_identity<Node>();
    // Scan of one whole 128-site tile, lane l owning its sites 2l and 2l+1, always in the same order:
    // side 0: dst[x] = sum of the tile's sites x .. 127;  side 1: dst[x] = sum of its sites 0 .. x.
    auto scan_tile = [&](int side, uint64_t tile, Node *dst) {
        const uint64_t i0 = tile * kSlideTile + 2 * (uint64_t)lane;
        const Node v0 = i0 < n_sites ? Tr::leaf(c, i0) : none;
        const Node v1 = i0 + 1 < n_sites ? Tr::leaf(c, i0 + 1) : none;
        Node pairsum = v0;
        node_add(pairsum, v1);
        if (side == 0) {
            Node s1 = lane_scan_exclusive<Node, true>(pairsum, lane);  // lanes above
            node_add(s1, v1);
            Node s0 = s1;
            node_add(s0, v0);
            dst[2 * lane + 1] = s1;
            dst[2 * lane] = s0;
        } else {
            Node p0 = lane_scan_exclusive<Node, false>(pairsum, lane);  // lanes below
            node_add(p0, v0);
            Node p1 = p0;
            node_add(p1, v1);
            dst[2 * lane] = p0;
            dst[2 * lane + 1] = p1;
        }
    };

    for (uint64_t task = wave0; task < n_tasks; task += n_waves) {
        if (task == n_groups) {  // the genome-wide total (dxy)
            const Node acc = node_wave_sum(total_partial<Tr>(tree, tv, lane));
            if (lane == 0) Tr::store_total(tot, acc);
            continue;
        }
        const uint64_t w = task * group + (uint64_t)lane;
        const bool active = (uint32_t)lane < group && w < n_win;
        pgt_win wd;
        wd.lo = wd.hi = 0; wd.flags = PGT_WIN_COORDS; wd.start = wd.end = 0; wd.label_run = 0;
        if (active) wd = win[w];
        const uint64_t hi = wd.hi < n_sites ? wd.hi : n_sites;
        const uint64_t lo = wd.lo < hi ? wd.lo : hi;
        const uint64_t tl = lo / kSlideTile, th = hi / kSlideTile;
        const uint32_t offl = (uint32_t)(lo % kSlideTile), offh = (uint32_t)(hi % kSlideTile);
        // Windows of a few sites (`-winsize 1 -stepsize 1`, the per-site mode dxyWindow.cpp:47 documents) and windows
        // that lie inside one 128-site tile (no boundary to split them at): the lane adds its own window's sites from
        // left to right — the reference's own order (fstWindow.cpp:76-83) — instead of taking part in up to 64
        // wave-wide queries, one per such window.
        const uint64_t A = offl ? (tl + 1) * kSlideTile : lo, B = th * kSlideTile;  // first / last tile boundary inside [lo, hi]
        const bool tiny = active && hi > lo && (hi - lo <= (uint64_t)kSlideDirect || A > B);  // A > B: inside one tile (< 128 sites)
        if (tiny) {
            Node acc = none;
            for (uint64_t i = lo; i < hi; ++i) node_add(acc, Tr::leaf(c, i));
            uint32_t start = wd.start, end = wd.end;
            if (!(wd.flags & PGT_WIN_COORDS)) {  // fstWindow.cpp:71-72
                start = pos[lo];
                end = pos[hi - 1];
            }
            Tr::finish(out + (uint64_t)pair * n_win + w, acc, start, end, lo, hi, c, pos);
        }
        if (__ballot(active && !tiny) == 0) continue;  // wave-uniform: nothing left that needs the scans
        // the group's first start tile / end tile (wave-uniform)
        uint64_t tl0 = active ? tl : ~0ull, th0 = active ? th : ~0ull;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint64_t a = __shfl_xor(tl0, d, kWave), b = __shfl_xor(th0, d, kWave);
            tl0 = a < tl0 ? a : tl0;
            th0 = b < th0 ? b : th0;
        }
        // scans of the two tiles under the starts (suffix) and the two under the ends (prefix)
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                scan_tile(side, (side ? th0 : tl0) + (uint64_t)k, span + (side * 2 + k) * kSlideTile);
        // the wave's own LDS rows: LDS operations of one wave complete in order, no barrier needed
        const bool fast = active && !tiny && hi > lo && tl - tl0 <= 1 && th - th0 <= 1 && A <= B;
        Node left = none, right = none;
        if (fast && offl) left = span[(0 * 2 + (int)(tl - tl0)) * kSlideTile + offl];
        if (fast && offh) right = span[(1 * 2 + (int)(th - th0)) * kSlideTile + offh - 1];
        uint32_t start = wd.start, end = wd.end;
        if (active && !(wd.flags & PGT_WIN_COORDS)) {  // fstWindow.cpp:71-72
            start = hi > lo ? pos[lo] : 0u;
            end = hi > lo ? pos[hi - 1] : 0u;
        }
        typename Tr::Row *row = out + (uint64_t)pair * n_win + w;
        bool pending = fast;
        for (unsigned long long m = __ballot(pending); m != 0; m = __ballot(pending)) {
            const int L = __ffsll((long long)m) - 1;
            const uint64_t Au = __shfl(A, L, kWave), Bu = __shfl(B, L, kWave);
            const Node mid = node_wave_sum(range_partial<Tr>(c, tree, tv, Au, Bu, lane, n_sites));
            if (pending && A == Au && B == Bu) {
                Node total = left;
                node_add(total, mid);
                node_add(total, right);
                Tr::finish(row, total, start, end, lo, hi, c, pos);
                pending = false;
            }
        }
        // windows outside the group's tiles: the same three pieces from scans of their own tiles; windows
        // shorter than the split allows (A > B) or empty: the plain range query.  Either way the order
        // of the sums depends on the window alone.
        bool slow = active && !fast && !tiny;
        for (unsigned long long m = __ballot(slow); m != 0; m = __ballot(slow)) {
            const int L = __ffsll((long long)m) - 1;
            const uint64_t lo_u = __shfl(lo, L, kWave), hi_u = __shfl(hi, L, kWave);
            const uint64_t Au = __shfl(A, L, kWave), Bu = __shfl(B, L, kWave);
            Node acc;
            if (hi_u > lo_u && Au <= Bu) {  // wave-uniform
                const uint32_t ol = (uint32_t)(lo_u % kSlideTile), oh = (uint32_t)(hi_u % kSlideTile);
                if (ol) scan_tile(0, lo_u / kSlideTile, span);
                if (oh) scan_tile(1, hi_u / kSlideTile, span + 2 * kSlideTile);
                acc = ol ? span[ol] : none;
                node_add(acc, node_wave_sum(range_partial<Tr>(c, tree, tv, Au, Bu, lane, n_sites)));
                node_add(acc, oh ? span[2 * kSlideTile + oh - 1] : none);
            } else {
                acc = node_wave_sum(range_partial<Tr>(c, tree, tv, lo_u, hi_u, lane, n_sites));
            }
            if (lane == L) {
                Tr::finish(row, acc, start, end, lo, hi, c, pos);
                slow = false;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// QUERY, group form (pgt_set_window_step in (kSlideMaxStep, kGroupMaxStep], windows of at least two level-2
// tiles): one wave answers 64 CONSECUTIVE windows, one per lane — the regime of `-winsize 50000 -stepsize 100`,
// where the per-window query pays four dependent round trips and ~3.3 KB of ragged reads for every window
// (fstWindow.cpp:80-83,95-99 re-sums W sites and shifts W-S there).  A window [lo,hi) is cut at the leaf grid
// (a1 = first leaf boundary >= lo, b1 = last <= hi) and at the level-2 grid (A, B likewise):
//     [lo,a1) + [b1,hi)   ragged SITES, < one leaf each        -> wave-wide, one window at a time, 4 in flight
//     [a1,A)              whole leaves inside lo's level-2 tile -> ONE suffix scan of that tile's 64 level-1
//                                                                 nodes, shared by every window starting in it
//     [A,B)               whole level-2 tiles                   -> ONE range query per distinct (A,B) of the group
//     [B,b1)              whole leaves inside hi's level-2 tile -> ONE prefix scan, shared likewise
// Consecutive windows start S sites apart, so the 64 starts of a group lie in 64 S / 8192 + 1 level-2 tiles
// (2 at S = 100): the level-1 and level-2 reads of the per-window query (two ragged 1-KiB node loads and a
// level-2 load per window, each a dependent round trip) shrink to a few loads per GROUP.  The pieces are
// added in the fixed order edges, left leaves, interior, right leaves, and each piece is computed in an order
// that depends on the window alone (fixed lane order of the scans; range_partial for the interior): rows are
// bitwise independent of which windows share a wave, hence of the number of GPUs.  Windows the cut does not
// fit (shorter than the level-2 grid allows: A > B) take the plain range query, one by one.
// (Round 3, tried and dropped: the five pieces kept in separate registers, with the coordinates and the level-1 nodes of
// the first start / end tile requested before the edge scans so that a group costs two round trips instead of six in a
// row — 81-89 VGPRs, 5 instead of 7 waves per SIMD, and no faster: S = 1 2.96 vs 2.85 ms per 10^8 windows, S = 100
// 0.496 vs 0.456 ms per 10^6; profiles/r03/measure_query_1e8_prefetch_variant.md.  The kernel is not latency-bound.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src_lane) {  // wave-uniform src_lane
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
constexpr size_t kGroupLdsBytes = (size_t)4 * 2 * kLeafF64 * 16;  // per wave: two 128-entry scan tables (also used, 64 entries each, for the node scans)

template <class Tr>
__device__ __forceinline__ void query_group_body(const typename Tr::Args &args, const uint32_t *__restrict__ pos,
                                                 const TreeView &tv, const pgt_win *__restrict__ win, uint64_t n_win,
                                                 typename Tr::Row *__restrict__ out, uint64_t n_sites, pgt_dxy_total *tot,
                                                 int pair, uint32_t group, bool edge_scans, char *lds) {
    using Node = typename Tr::Node;
    constexpr uint64_t kLeaf = (uint64_t)Tr::kLeaf, kTile2 = kLeaf * kRadix;
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const typename Tr::Cols c = Tr::cols(args, pair);
    const char *tree = tv.base + (size_t)pair * tv.pair_stride;
    const Node *__restrict__ l1 = reinterpret_cast<const Node *>(tree + tv.off[0]);
    const uint64_t n_l1 = (n_sites + kLeaf - 1) / kLeaf;  // level-1 nodes the build wrote (the padding beyond is never read)
    Node *table = reinterpret_cast<Node *>(lds) + (size_t)wib * 2 * kLeafF64;  // [2][128]
    const Node none = node_identity<Node>();
    const uint64_t n_groups = (n_win + group - 1) / group;
    const uint64_t n_tasks = n_groups + (tot ? 1 : 0);

    for (uint64_t task = wave0; task < n_tasks; task += n_waves) {
        if (task == n_groups) {  // the genome-wide total (dxy)
            const Node acc = node_wave_sum(total_partial<Tr>(tree, tv, lane));
            if (lane == 0) Tr::store_total(tot, acc);
            continue;
        }
        const uint64_t w = task * group + (uint64_t)lane;
        const bool active = (uint32_t)lane < group && w < n_win;
        pgt_win wd;
        wd.lo = wd.hi = 0; wd.flags = PGT_WIN_COORDS; wd.start = wd.end = 0; wd.label_run = 0;
        if (active) wd = win[w];
        const uint64_t hi = wd.hi < n_sites ? wd.hi : n_sites;
        const uint64_t lo = wd.lo < hi ? wd.lo : hi;
        const uint64_t a1 = (lo + kLeaf - 1) / kLeaf * kLeaf, b1 = hi / kLeaf * kLeaf;
        const uint64_t A = (lo + kTile2 - 1) / kTile2 * kTile2, B = hi / kTile2 * kTile2;
        const bool fast = active && hi > lo && A <= B;  // then lo <= a1 <= A <= B <= b1 <= hi
        Node sum = none;

        // (1) the ragged sites at both ends.  edge_scans (steps up to 64 sites, f64 trees; wave-uniform, fixed by the hint):
        // [lo,a1) is a suffix of lo's 128-site leaf tile, [b1,hi) a prefix of hi's: ONE scan per DISTINCT tile (16-byte loads,
        // fixed lane order), shared by every window of the group that starts (ends) in it — at S = 32 the 64 starts lie in
        // 17 tiles, at S = 1 in one or two.  A left and a right tile are loaded together per iteration.  From 64 sites on
        // nearly every window has tiles of its own and the scan costs more than it shares (S = 100, 10^6 windows: 0.535 ms
        // against 0.477): the edges are then summed wave-wide per window, the loads of four windows in flight together.
        bool direct_edges = true;
        if constexpr (Tr::kLeaf == kLeafF64) direct_edges = !edge_scans;
        if constexpr (Tr::kLeaf == kLeafF64) if (edge_scans) {
            bool pl = fast && lo < a1, pr = fast && b1 < hi;
            const uint64_t tl = lo / kLeaf, th = hi / kLeaf;
            for (;;) {
                const unsigned long long ml = __ballot(pl), mr = __ballot(pr);
                if ((ml | mr) == 0) break;
                const uint64_t TL = ml ? readlane_u64(tl, __ffsll((long long)ml) - 1) : 0;
                const uint64_t TR = mr ? readlane_u64(th, __ffsll((long long)mr) - 1) : 0;
                Node l0 = none, l1v = none, r0 = none, r1v = none;
                const uint64_t il = TL * kLeaf + 2 * (uint64_t)lane, ir = TR * kLeaf + 2 * (uint64_t)lane;
                if (ml) {  // wave-uniform
                    if (il + 1 < n_sites) Tr::leaf_pair(c, il, l0, l1v);
                    else if (il < n_sites) l0 = Tr::leaf(c, il);
                }
                if (mr) {
                    if (ir + 1 < n_sites) Tr::leaf_pair(c, ir, r0, r1v);
                    else if (ir < n_sites) r0 = Tr::leaf(c, ir);
                }
                if (ml) {  // suffix: table[x] = sites x .. 127 of the tile
                    Node ps = l0;
                    node_add(ps, l1v);
                    Node s1 = lane_scan_exclusive<Node, true>(ps, lane);
                    node_add(s1, l1v);
                    Node s0 = s1;
                    node_add(s0, l0);
                    table[2 * lane + 1] = s1;
                    table[2 * lane] = s0;
                    if (pl && tl == TL) {
                        sum = table[lo - TL * kLeaf];  // the first piece: the edges are added left, then right
                        pl = false;
                    }
                }
                if (mr) {  // prefix: table[128 + x] = sites 0 .. x
                    Node ps = r0;
                    node_add(ps, r1v);
                    Node p0 = lane_scan_exclusive<Node, false>(ps, lane);
                    node_add(p0, r0);
                    Node p1 = p0;
                    node_add(p1, r1v);
                    table[kLeafF64 + 2 * lane] = p0;
                    table[kLeafF64 + 2 * lane + 1] = p1;
                }
                // a window's right edge is added after its left edge: only once the left one is in (or there is none)
                if (mr && pr && th == TR && !pl) {
                    node_add(sum, table[kLeafF64 + (hi - TR * kLeaf) - 1]);
                    pr = false;
                }
            }
        }
        if (direct_edges) {
            for (unsigned long long m = __ballot(fast); m != 0;) {
                Node acc[4];
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    src[u] = m != 0 ? __ffsll((long long)m) - 1 : -1;
                    if (m != 0) m &= m - 1;
                    acc[u] = none;
                    if (src[u] >= 0)  // wave-uniform
                        ragged_pair<Tr>(acc[u], c, tree, tv, 0, readlane_u64(lo, src[u]), readlane_u64(a1, src[u]),
                                        readlane_u64(b1, src[u]), readlane_u64(hi, src[u]), lane, n_sites);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (src[u] >= 0) {
                        const Node e = node_wave_sum(acc[u]);
                        if (lane == src[u]) sum = e;
                    }
            }
        }
        // (2) whole leaves between a1 and A: a suffix scan of the 64 level-1 nodes of lo's level-2 tile
        {
            bool pend = fast && a1 < A;
            const uint64_t myT = lo / kTile2;
            for (unsigned long long m = __ballot(pend); m != 0; m = __ballot(pend)) {
                const uint64_t T = readlane_u64(myT, __ffsll((long long)m) - 1);
                const uint64_t i = T * kRadix + (uint64_t)lane;
                const Node v = i < n_l1 ? l1[i] : none;
                Node sfx = lane_scan_exclusive<Node, true>(v, lane);  // the lanes above, in a fixed order
                node_add(sfx, v);
                table[lane] = sfx;  // the wave's own LDS rows: operations of one wave complete in order
                if (pend && myT == T) {
                    node_add(sum, table[(a1 - T * kTile2) / kLeaf]);
                    pend = false;
                }
            }
        }
        // (3) whole level-2 tiles: one range query per distinct (A,B)
        {
            bool pend = fast && A < B;
            for (unsigned long long m = __ballot(pend); m != 0; m = __ballot(pend)) {
                const int L = __ffsll((long long)m) - 1;
                const uint64_t Au = readlane_u64(A, L), Bu = readlane_u64(B, L);
                const Node mid = node_wave_sum(range_partial<Tr>(c, tree, tv, Au, Bu, lane, n_sites));
                if (pend && A == Au && B == Bu) {
                    node_add(sum, mid);
                    pend = false;
                }
            }
        }
        // (4) whole leaves between B and b1: a prefix scan of the level-1 nodes of hi's level-2 tile
        {
            bool pend = fast && B < b1;
            const uint64_t myT = hi / kTile2;  // B == myT * kTile2
            for (unsigned long long m = __ballot(pend); m != 0; m = __ballot(pend)) {
                const uint64_t T = readlane_u64(myT, __ffsll((long long)m) - 1);
                const uint64_t i = T * kRadix + (uint64_t)lane;
                const Node v = i < n_l1 ? l1[i] : none;
                table[lane] = lane_scan_exclusive<Node, false>(v, lane);  // the lanes below: nodes 0 .. lane-1
                if (pend && myT == T) {
                    node_add(sum, table[(b1 - B) / kLeaf]);  // 1 .. 63
                    pend = false;
                }
            }
        }
        uint32_t start = wd.start, end = wd.end;
        if (active && !(wd.flags & PGT_WIN_COORDS)) {  // fstWindow.cpp:71-72
            start = hi > lo ? pos[lo] : 0u;
            end = hi > lo ? pos[hi - 1] : 0u;
        }
        typename Tr::Row *row = out + (uint64_t)pair * n_win + w;
        if (fast) Tr::finish(row, sum, start, end, lo, hi, c, pos);
        // windows the cut does not fit (or empty ones): the plain range query, whose order depends on the window alone too
        bool slow = active && !fast;
        for (unsigned long long m = __ballot(slow); m != 0; m = __ballot(slow)) {
            const int L = __ffsll((long long)m) - 1;
            const Node acc = node_wave_sum(range_partial<Tr>(c, tree, tv, readlane_u64(lo, L), readlane_u64(hi, L), lane, n_sites));
            if (lane == L) {
                Tr::finish(row, acc, start, end, lo, hi, c, pos);
                slow = false;
            }
        }
    }
}

template <class Tr>
__global__ __launch_bounds__(256) void query_kernel(typename Tr::Args args, const uint32_t *pos, TreeView tv,
                                                    const pgt_win *win, uint64_t n_win, typename Tr::Row *out,
                                                    uint64_t n_sites, pgt_dxy_total *tot) {
    query_body<Tr>(args, pos, tv, win, n_win, out, n_sites, tot, (int)blockIdx.y);
}

struct DxyHetQueryArgs {
    DxyTraits::Args dxy;
    const int8_t *g[2];
    TreeView tv_dxy, tv_het[2];
    pgt_dxy_row *dxy_out;
    pgt_dxy_total *tot;
    pgt_het_row *het_out[2];
};
__global__ __launch_bounds__(256) void dxy_het_query_kernel(DxyHetQueryArgs f, const uint32_t *pos, const pgt_win *win,
                                                            uint64_t n_win, uint64_t n_sites) {
    if (blockIdx.y == 0) {
        query_body<DxyTraits>(f.dxy, pos, f.tv_dxy, win, n_win, f.dxy_out, n_sites, f.tot, 0);
    } else {
        const int k = blockIdx.y - 1;
        query_body<HetTraits>(HetTraits::Args{f.g[k]}, pos, f.tv_het[k], win, n_win, f.het_out[k], n_sites, nullptr, 0);
    }
}

template <class Tr>
__global__ __launch_bounds__(256) void query_slide_kernel(typename Tr::Args args, const uint32_t *pos, TreeView tv,
                                                          const pgt_win *win, uint64_t n_win, typename Tr::Row *out,
                                                          uint64_t n_sites, pgt_dxy_total *tot, uint32_t group) {
    extern __shared__ __attribute__((aligned(16))) char lds_span[];
    query_slide_body<Tr>(args, pos, tv, win, n_win, out, n_sites, tot, (int)blockIdx.y, group, lds_span);
}
__global__ __launch_bounds__(256) void dxy_het_query_slide_kernel(DxyHetQueryArgs f, const uint32_t *pos, const pgt_win *win,
                                                                  uint64_t n_win, uint64_t n_sites, uint32_t group) {
    extern __shared__ __attribute__((aligned(16))) char lds_span[];
    if (blockIdx.y == 0) {
        query_slide_body<DxyTraits>(f.dxy, pos, f.tv_dxy, win, n_win, f.dxy_out, n_sites, f.tot, 0, group, lds_span);
    } else {
        const int k = blockIdx.y - 1;
        query_slide_body<HetTraits>(HetTraits::Args{f.g[k]}, pos, f.tv_het[k], win, n_win, f.het_out[k], n_sites, nullptr, 0,
                                    group, lds_span);
    }
}
constexpr size_t kSlideLdsBytes = (size_t)4 * 4 * kSlideTile * 16;  // 4 waves x 4 tiles x 128 nodes of <= 16 B

template <class Tr>
__global__ __launch_bounds__(256) void query_group_kernel(typename Tr::Args args, const uint32_t *pos, TreeView tv,
                                                          const pgt_win *win, uint64_t n_win, typename Tr::Row *out,
                                                          uint64_t n_sites, pgt_dxy_total *tot, uint32_t group, int edge_scans) {
    extern __shared__ __attribute__((aligned(16))) char lds_group[];
    query_group_body<Tr>(args, pos, tv, win, n_win, out, n_sites, tot, (int)blockIdx.y, group, edge_scans != 0, lds_group);
}
// config 3 with the dxy table on the group query; the genotype tables (int8 tree: 65536-site level-2 tiles) follow their
// own plan: groups too when the windows are long enough, else the sliding query (steps up to 32) or one wave per window
enum { kHetPerWindow = 0, kHetGroups = 1, kHetSlide = 2 };
__global__ __launch_bounds__(256) void dxy_het_query_group_kernel(DxyHetQueryArgs f, const uint32_t *pos, const pgt_win *win,
                                                                  uint64_t n_win, uint64_t n_sites, int het_mode, uint32_t group,
                                                                  int edge_scans, uint32_t het_slide_group) {
    extern __shared__ __attribute__((aligned(16))) char lds_group[];
    if (blockIdx.y == 0) {
        query_group_body<DxyTraits>(f.dxy, pos, f.tv_dxy, win, n_win, f.dxy_out, n_sites, f.tot, 0, group, edge_scans != 0, lds_group);
    } else {
        const int k = blockIdx.y - 1;
        const HetTraits::Args a{f.g[k]};
        if (het_mode == kHetGroups)
            query_group_body<HetTraits>(a, pos, f.tv_het[k], win, n_win, f.het_out[k], n_sites, nullptr, 0, group, false, lds_group);
        else if (het_mode == kHetSlide)
            query_slide_body<HetTraits>(a, pos, f.tv_het[k], win, n_win, f.het_out[k], n_sites, nullptr, 0, het_slide_group, lds_group);
        else
            query_body<HetTraits>(a, pos, f.tv_het[k], win, n_win, f.het_out[k], n_sites, nullptr, 0);
    }
}

inline int hip_fail(hipError_t e, const char *what, std::string *err) {
    if (e == hipSuccess) return PGT_OK;
    if (err) *err = std::string(what) + ": " + hipGetErrorString(e);
    return PGT_EDEVICE;
}

inline unsigned build_grid(uint64_t n_items, unsigned cap = 2048) {
    // 4 waves per 256-thread workgroup, one work item (level-2 tile) per wave-iteration, at most `cap`
    // workgroups (what is resident at once); the rest is grid-strided.  The grid is BALANCED: with
    // r = ceil(items / (4 cap)) rounds, only ceil(items / r) waves are launched, so every wave does r
    // items (the last few r - 1) instead of some waves doing one item more than the others while the
    // rest of the chip idles — 7.45 rounds cost 8 either way, but 15259 items over 1908 waves finish
    // together (1.25e8 sites, the per-GPU shard of the 8-GPU run: +7 %).
    if (cap == 0) cap = 1u << 28;
    if (n_items == 0) return 1;
    const uint64_t max_waves = (uint64_t)cap * 4;
    const uint64_t rounds = (n_items + max_waves - 1) / max_waves;
    const uint64_t waves = (n_items + rounds - 1) / rounds;
    uint64_t blocks = (waves + 3) / 4;
    if (blocks > (1u << 30)) blocks = 1u << 30;
    return (unsigned)blocks;
}

inline unsigned query_grid(uint64_t n_items) {
    uint64_t blocks = (n_items + 3) / 4;
    if (blocks > (1u << 16)) blocks = 1u << 16;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

inline uint64_t het_items(uint64_t n) { return (n + (uint64_t)kLeafI8 * kHetChunk - 1) / ((uint64_t)kLeafI8 * kHetChunk); }

// levels = how many tree levels are built and may be used by queries (useful_levels()).
TreeView make_view(const TreeLayout &tl, void *tree, size_t pair_stride, int levels) {
    TreeView tv{};
    tv.base = static_cast<char *>(tree);
    tv.pair_stride = pair_stride;
    tv.n_levels = levels;
    tv.n_partials = 0;
    tv.partials = tl.partials;
    for (int k = 0; k < tl.n_levels; ++k) tv.off[k] = tl.offset[k];
    return tv;
}

// Levels above what the build kernel wrote itself: slot k-1 -> slot k for k = first..n_levels-1.
// first = 2 when the build wrote levels 1 and 2 (fst, dxy), 1 when it wrote level 1 only (het);
// n_level1 = level-1 nodes actually written (the padding beyond them is never read).
template <class Node>
int launch_upper(const TreeLayout &tl, const TreeView &tv, unsigned n_pairs, hipStream_t s, std::string *err,
                 int first = 2, uint64_t n_level1 = 0) {
    for (int k = first; k < tv.n_levels; ++k) {
        const uint64_t n_child = k == 1 ? n_level1 : tl.count[k - 1], n_parent = tl.count[k];
        dim3 grid(query_grid(n_parent), n_pairs);
        hipLaunchKernelGGL(tree_up_kernel<Node>, grid, dim3(256), 0, s, tv, k - 1, n_child, n_parent);
        if (int rc = hip_fail(hipGetLastError(), "tree_up_kernel", err)) return rc;
    }
    return PGT_OK;
}

inline int record(void *ev, hipStream_t s, std::string *err) {
    if (!ev) return PGT_OK;
    return hip_fail(hipEventRecord(static_cast<hipEvent_t>(ev), s), "hipEventRecord", err);
}

// ------------------------------------------------------------------------------------------
// The site-window table written on the device from the per-run plan (pgt_windows.cpp: plan_entry_windows /
// for_each_window are the host form of exactly this): thread i finds its run by bisection over the runs'
// first window indices and writes window i.  40 B of plan per run in, 32 B per window out.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void windows_from_plan_kernel(const RunPlan *__restrict__ plan, uint64_t n_runs, uint64_t n_win,
                                                                uint32_t W, uint32_t S, pgt_win *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_win; i += stride) {
        uint64_t lo = 0, hi = n_runs;  // the last run with out0 <= i: a run that emits nothing shares its out0 with its
        while (hi - lo > 1) {          // successor, so the last one is the run that holds window i
            const uint64_t mid = (lo + hi) / 2;
            if (plan[mid].out0 <= i) lo = mid; else hi = mid;
        }
        uint64_t r = lo;
        while (r > 0 && i - plan[r].out0 >= plan[r].K + plan[r].tail) --r;  // (cannot happen for a consistent plan)
        const RunPlan p = plan[r];
        const uint64_t k = i - p.out0;
        pgt_win w;
        if (k < p.K) {
            w.hi = p.hi0 + k * S;
            w.lo = w.hi - W;
        } else {
            w.lo = p.tail_lo;
            w.hi = p.tail_hi;
        }
        w.label_run = (uint32_t)r;
        w.flags = 0;
        w.start = w.end = 0;
        out[i] = w;
    }
}

// pgt_rowbuf_fill: the pattern a rank stores through a freshly mapped peer buffer before the mapping is trusted
__global__ __launch_bounds__(256) void fill_pattern_kernel(uint64_t *__restrict__ words, uint64_t n_words, uint64_t seed) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        uint64_t z = seed + i + 0x9E3779B97F4A7C15ull;  // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        words[i] = z ^ (z >> 31);
    }
}

}  // namespace

int launch_fill_pattern(uint64_t *words, uint64_t n_words, uint64_t seed, void *stream, std::string *err) {
    if (n_words == 0) return PGT_OK;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), words, n_words, seed);
    return hip_fail(hipGetLastError(), "fill_pattern_kernel", err);
}

int launch_windows_from_plan(const RunPlan *d_plan, uint64_t n_runs, uint64_t n_win, uint32_t W, uint32_t S, pgt_win *d_out,
                             void *stream, std::string *err) {
    if (n_win == 0) return PGT_OK;
    uint64_t blocks = (n_win + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(windows_from_plan_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_plan, n_runs,
                       n_win, W, S, d_out);
    return hip_fail(hipGetLastError(), "windows_from_plan_kernel", err);
}

// Called once from pgt_open: declares the dynamic-LDS needs of the staged build kernels, so that no
// attribute call can fall inside a caller's stream capture.
int init_kernels(std::string *err) {  // per pgt_open, i.e. per device: function attributes are per device
    const void *staged[] = {reinterpret_cast<const void *>(fst_build_kernel<>),
                            reinterpret_cast<const void *>(dxy_build_kernel),
                            reinterpret_cast<const void *>(ext_build_kernel<>), reinterpret_cast<const void *>(ext_build_kernel<kExtStageSmall, 2, true>)};
    for (const void *k : staged)
        if (int rc = hip_fail(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFstStageBytes),
                              "hipFuncSetAttribute", err))
            return rc;
    if (int rc = hip_fail(hipFuncSetAttribute(reinterpret_cast<const void *>(dxy_het_build_kernel<>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDxyHetStageBytes),
                          "hipFuncSetAttribute", err))
        return rc;
    return PGT_OK;
}

// The streaming build launch of the fst tree (levels 1 and 2 of `np` pairs): one instantiation at every size (until
// round 3 short inputs took a form with 16 loads in flight to shorten the latency-bound last tile; with one column at a
// time the short queue wins at every size, see fst_build_kernel).
namespace {
void fst_build_launch(hipStream_t s, const PairCols &cols, uint32_t np, uint64_t n, const TreeLayout &tl, const TreeView &tv) {
    const dim3 grid(build_grid(tl.count[1], kFstBuildBlocks), np);
    hipLaunchKernelGGL((fst_build_kernel<>), grid, dim3(256), kFstStageBytes, s, cols, n, tl.count[1], tv);
}

// launch_fst with the build launch as a parameter: the product passes fst_build_launch; tools/pgt_kernels_tuning.hip
// (a separate translation unit that textually includes this file, never part of the product build) passes
// its experiments.  No preprocessor switch lives in this file.
template <class BuildFn>
int launch_fst_with(BuildFn build, const uint32_t *pos, const double *const *a, const double *const *b, uint32_t n_pairs,
                    uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree, void *stream,
                    void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout tl = tree_layout(PGT_STAT_FST, n);
    const int levels = useful_levels(tl, PGT_STAT_FST, hints.max_window);
    for (uint32_t p0 = 0; p0 < n_pairs; p0 += kMaxPairs) {
        const uint32_t np = n_pairs - p0 < (uint32_t)kMaxPairs ? n_pairs - p0 : (uint32_t)kMaxPairs;
        PairCols cols{};
        for (uint32_t p = 0; p < np; ++p) { cols.a[p] = a[p0 + p]; cols.b[p] = b[p0 + p]; }
        const TreeView tv = make_view(tl, static_cast<char *>(tree) + (size_t)p0 * tl.bytes, tl.bytes, levels);
        if (p0 == 0) if (int rc = record(ev_build0, s, err)) return rc;
        if (n > 0) {
            build(s, cols, np, n, tl, tv);
            if (int rc = hip_fail(hipGetLastError(), "fst_build_kernel", err)) return rc;
            if (int rc = launch_upper<NodeFst>(tl, tv, np, s, err)) return rc;
        }
        if (p0 + np >= n_pairs) if (int rc = record(ev_build1, s, err)) return rc;
        if (n_win > 0) {
            FstTraits::Args args{cols};
            if (group_query(hints, kLeafF64)) {
                const uint32_t g = group_size(n_win);
                hipLaunchKernelGGL(query_group_kernel<FstTraits>, dim3(query_grid((n_win + g - 1) / g), np), dim3(256),
                                   kGroupLdsBytes, s, args, pos, tv, win, n_win, out + (uint64_t)p0 * n_win, n,
                                   (pgt_dxy_total *)nullptr, g, group_edge_scans(hints));
            } else if (const uint32_t group = slide_group(hints.window_step); group > 1)
                hipLaunchKernelGGL(query_slide_kernel<FstTraits>, dim3(query_grid((n_win + group - 1) / group), np), dim3(256),
                                   kSlideLdsBytes, s, args, pos, tv, win, n_win, out + (uint64_t)p0 * n_win, n,
                                   (pgt_dxy_total *)nullptr, group);
            else
                hipLaunchKernelGGL(query_kernel<FstTraits>, dim3(query_grid(n_win), np), dim3(256), 0, s, args, pos,
                                   tv, win, n_win, out + (uint64_t)p0 * n_win, n, (pgt_dxy_total *)nullptr);
            if (int rc = hip_fail(hipGetLastError(), "query_kernel<fst>", err)) return rc;
        }
    }
    return record(ev_query1, s, err);
}
}  // namespace

int launch_fst(const uint32_t *pos, const double *const *a, const double *const *b, uint32_t n_pairs,
               uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree, void *stream,
               void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints) {
    return launch_fst_with(fst_build_launch, pos, a, b, n_pairs, n, win, n_win, out, tree, stream, ev_build0, ev_build1, ev_query1,
                           err, hints);
}

int launch_het(const uint32_t *pos, const int8_t *g, uint64_t n, const pgt_win *win, uint64_t n_win,
               pgt_het_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1, void *ev_query1,
               std::string *err, const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout tl = tree_layout(PGT_STAT_HET, n);
    const TreeView tv = make_view(tl, tree, tl.bytes, useful_levels(tl, PGT_STAT_HET, hints.max_window));
    if (int rc = record(ev_build0, s, err)) return rc;
    if (n > 0) {
        const uint64_t n_items = het_items(n);
        // short inputs live on many short work items and want the waves (128-VGPR build, 16 waves per CU): 12.7 against 13.7 us
        // at 3e7 sites, 32.9 against 34.8 at 1.25e8; long ones stream better with the 254-VGPR build: 149 against 172 us at 10^9
        // (interleaved A/B in one process, profiles/r06/het_kernel_crossover_ab.md)
        // The grid is what is RESIDENT at once (round 6): 1024 workgroups of the 128-VGPR build (16 waves per CU), 512 of the
        // 254-VGPR one (8) — until then both took up to 2048, i.e. 1.5 ... 4 generations of workgroups, and a last generation that
        // is half empty idles half the chip: 10^8 sites 25.0 -> 22.7 us, 3e8 57.6 -> 50.8 (65 -> 74 % of the HBM peak), 10^9
        // 158.2 -> 152.5 (79 -> 82 %), interleaved A/B (profiles/r06/het_grid_resident_ab.md).
        if (n_items <= kHetSmallItems)
            hipLaunchKernelGGL(het_build_kernel_w4, dim3(build_grid(n_items, 1024)), dim3(256), 0, s, g, n, n_items, tv);
        else
            hipLaunchKernelGGL(het_build_kernel, dim3(build_grid(n_items, 512)), dim3(256), 0, s, g, n, n_items, tv);
        if (int rc = hip_fail(hipGetLastError(), "het_build_kernel", err)) return rc;
        if (int rc = launch_upper<NodeHet>(tl, tv, 1, s, err, 1, n_items * kHetChunk)) return rc;
    }
    if (int rc = record(ev_build1, s, err)) return rc;
    if (n_win > 0) {
        HetTraits::Args args{g};
        if (group_query(hints, kLeafI8)) {
            const uint32_t g = group_size(n_win);
            hipLaunchKernelGGL(query_group_kernel<HetTraits>, dim3(query_grid((n_win + g - 1) / g)), dim3(256),
                               kGroupLdsBytes, s, args, pos, tv, win, n_win, out, n, (pgt_dxy_total *)nullptr, g, 0);
        } else if (const uint32_t group = slide_group(hints.window_step); group > 1)
            hipLaunchKernelGGL(query_slide_kernel<HetTraits>, dim3(query_grid((n_win + group - 1) / group)), dim3(256),
                               kSlideLdsBytes, s, args, pos, tv, win, n_win, out, n, (pgt_dxy_total *)nullptr, group);
        else
            hipLaunchKernelGGL(query_kernel<HetTraits>, dim3(query_grid(n_win)), dim3(256), 0, s, args, pos, tv, win,
                               n_win, out, n, (pgt_dxy_total *)nullptr);
        if (int rc = hip_fail(hipGetLastError(), "query_kernel<het>", err)) return rc;
    }
    return record(ev_query1, s, err);
}

int launch_dxy(const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1, const int32_t *n2,
               uint64_t n, int minind, const pgt_win *win, uint64_t n_win, pgt_dxy_row *out, pgt_dxy_total *tot,
               void *tree, void *stream, void *ev_build0, void *ev_build1, void *ev_query1, std::string *err,
               const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout tl = tree_layout(PGT_STAT_DXY, n);
    // the genome-wide total is the sum of the build waves' partial sums (dxy_build_body): it needs no level of its own
    TreeView tv = make_view(tl, tree, tl.bytes, useful_levels(tl, PGT_STAT_DXY, hints.max_window));
    if (int rc = record(ev_build0, s, err)) return rc;
    if (n > 0) {
        const dim3 grid(build_grid(tl.count[1], kFstBuildBlocks));
        tv.n_partials = grid.x * 4;  // every wave of the grid writes one (waves without a tile: the identity)
        hipLaunchKernelGGL(dxy_build_kernel, grid, dim3(256), kFstStageBytes, s, p1, p2, n1, n2, n, minind, tl.count[1], tv);
        if (int rc = hip_fail(hipGetLastError(), "dxy_build_kernel", err)) return rc;
        if (int rc = launch_upper<NodeDxy>(tl, tv, 1, s, err)) return rc;
    }
    if (int rc = record(ev_build1, s, err)) return rc;
    if (n_win > 0 || tot) {
        DxyTraits::Args args{p1, p2, n1, n2, minind};
        if (group_query(hints, kLeafF64)) {
            const uint32_t g = group_size(n_win);
            hipLaunchKernelGGL(query_group_kernel<DxyTraits>, dim3(query_grid((n_win + g - 1) / g + 1)), dim3(256),
                               kGroupLdsBytes, s, args, pos, tv, win, n_win, out, n, tot, g, group_edge_scans(hints));
        } else if (const uint32_t group = slide_group(hints.window_step); group > 1)
            hipLaunchKernelGGL(query_slide_kernel<DxyTraits>, dim3(query_grid((n_win + group - 1) / group + 1)), dim3(256),
                               kSlideLdsBytes, s, args, pos, tv, win, n_win, out, n, tot, group);
        else
            hipLaunchKernelGGL(query_kernel<DxyTraits>, dim3(query_grid(n_win + 1)), dim3(256), 0, s, args, pos, tv,
                               win, n_win, out, n, tot);
        if (int rc = hip_fail(hipGetLastError(), "query_kernel<dxy>", err)) return rc;
    }
    return record(ev_query1, s, err);
}

// grid = exactly the workgroups that are resident together (LDS-limited), so that all waves run in near
// lockstep and their staged node rows are flushed at about the same times (see NodeStage)
namespace {
template <int STAGE, int UNROLL, bool DEFER>
void launch_ext_variant(hipStream_t s, const ExtBuildArgs &g, uint64_t n, uint64_t n_l2, const TreeView &tv, unsigned cap) {
    const size_t lds = DEFER ? (size_t)4 * STAGE * 1024 : 0;
    hipLaunchKernelGGL((ext_build_kernel<STAGE, UNROLL, DEFER>), dim3(build_grid(n_l2, cap)), dim3(256), lds, s, g, n, n_l2, tv);
}
void ext_build_launch(hipStream_t s, const ExtBuildArgs &g, uint64_t n, uint64_t n_l2, const TreeView &tv) {
    // Short inputs (up to 20000 tiles of 16384 sites = 3.3e8 sites): 16 waves per CU with 4 loads in flight each (two leaf tiles
    // per batch, an 8-KiB stage per wave, 1024 workgroups) — a tile is 128 KiB here, so 8 waves per CU leave a wave only 3
    // tiles at 10^8 sites and the launch is over before the stream is steady.  Round 5, with the tiles walked from a start of
    // the wave's own (tile_rotation), interleaved against the round-4 choice (8 waves x 16 loads): 5e7 sites +9.6 %, 10^8
    // +13.8 % (67.4 -> 76.8 % of the HBM peak), 1.25e8 -1.0 %, 2.5e8 +8.1 %; from 5e8 sites on 8 waves x 8 loads stay ahead
    // (10^9: 84.7 % against 81.8).  profiles/r05/build_ab_ext_geometry_with_rotation.md; round 4 had measured -12 % at 1.25e8
    // sites for the same geometry without the rotation.
    if (n_l2 <= kExtSmallTiles)
        launch_ext_variant<kExtStageSmall, 2, true>(s, g, n, n_l2, tv, 2 * kFstBuildBlocks);
    else
        launch_ext_variant<kExtStage, 4, true>(s, g, n, n_l2, tv, kFstBuildBlocks);
}

template <class BuildFn>  // the build launch is a parameter for the same reason as in launch_fst_with
int launch_ext_with(BuildFn build, const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff, const pgt_win *win,
                    uint64_t n_win, pgt_ext_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1,
                    void *ev_query1, std::string *err, const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout tl = tree_layout(PGT_STAT_EXT, n);
    const TreeView tv = make_view(tl, tree, tl.bytes, useful_levels(tl, PGT_STAT_EXT, hints.max_window));
    const double thr = mode == PGT_EXT_XP_MIN ? -cutoff : cutoff;  // s < cutoff  <=>  -s > -cutoff
    if (int rc = record(ev_build0, s, err)) return rc;
    if (n > 0) {
        build(s, ExtBuildArgs{score, mode, thr}, n, tl.count[1], tv);
        if (int rc = hip_fail(hipGetLastError(), "ext_build_kernel", err)) return rc;
        if (int rc = launch_upper<NodeExt>(tl, tv, 1, s, err)) return rc;
    }
    if (int rc = record(ev_build1, s, err)) return rc;
    if (n_win > 0) {
        ExtTraits::Args args{score, mode, thr};
        hipLaunchKernelGGL(query_kernel<ExtTraits>, dim3(query_grid(n_win)), dim3(256), 0, s, args, pos, tv, win, n_win,
                           out, n, (pgt_dxy_total *)nullptr);
        if (int rc = hip_fail(hipGetLastError(), "query_kernel<ext>", err)) return rc;
    }
    return record(ev_query1, s, err);
}
}  // namespace

int launch_ext(const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff, const pgt_win *win,
               uint64_t n_win, pgt_ext_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1,
               void *ev_query1, std::string *err, const Hints &hints) {
    return launch_ext_with(ext_build_launch, pos, score, n, mode, cutoff, win, n_win, out, tree, stream, ev_build0, ev_build1,
                           ev_query1, err, hints);
}

int launch_dxy_het(const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1, const int32_t *n2,
                   const int8_t *g1, const int8_t *g2, uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
                   pgt_dxy_row *dxy_out, pgt_dxy_total *tot, pgt_het_row *het_out1, pgt_het_row *het_out2, void *tree,
                   void *stream, void *ev_build0, void *ev_build1, void *ev_query1, std::string *err,
                   const Hints &hints) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TreeLayout td = tree_layout(PGT_STAT_DXY, n), th = tree_layout(PGT_STAT_HET, n);
    char *base = static_cast<char *>(tree);
    const int lh = useful_levels(th, PGT_STAT_HET, hints.max_window);
    TreeView tvd = make_view(td, base, td.bytes, useful_levels(td, PGT_STAT_DXY, hints.max_window));
    const TreeView tvh = make_view(th, base + td.bytes, th.bytes, lh);  // the two genotype trees, th.bytes apart
    TreeView tvh1 = tvh;
    tvh1.base += th.bytes;
    if (int rc = record(ev_build0, s, err)) return rc;
    if (n > 0) {
        const uint64_t n_items = het_items(n);  // == td.count[1]: a dxy level-2 tile is one het work item
        const dim3 grid(build_grid(td.count[1], kFstBuildBlocks));
        tvd.n_partials = grid.x * 4;  // the genome-wide line: see launch_dxy
        DxyHetBuildArgs f{p1, p2, n1, n2, {g1, g2}, n, minind, td.count[1], tvd, tvh};
        hipLaunchKernelGGL(dxy_het_build_kernel<>, grid, dim3(256), kDxyHetStageBytes, s, f);
        if (int rc = hip_fail(hipGetLastError(), "dxy_het_build_kernel", err)) return rc;
        if (int rc = launch_upper<NodeDxy>(td, tvd, 1, s, err)) return rc;
        if (int rc = launch_upper<NodeHet>(th, tvh, 2, s, err, 1, n_items * kHetChunk)) return rc;  // both genotype trees per launch
    }
    if (int rc = record(ev_build1, s, err)) return rc;
    if (n_win > 0 || tot) {
        DxyHetQueryArgs q{DxyTraits::Args{p1, p2, n1, n2, minind}, {g1, g2}, tvd, {tvh, tvh1}, dxy_out, tot,
                          {het_out1, het_out2}};
        if (group_query(hints, kLeafF64)) {
            const uint32_t g = group_size(n_win), hs = slide_group(hints.window_step);
            const int het_mode = group_query(hints, kLeafI8) ? kHetGroups : (hs > 1 ? kHetSlide : kHetPerWindow);
            const uint64_t items = het_mode == kHetPerWindow ? n_win + 1 : (het_mode == kHetSlide ? (n_win + hs - 1) / hs + 1 : (n_win + g - 1) / g + 1);
            hipLaunchKernelGGL(dxy_het_query_group_kernel, dim3(query_grid(items), 3), dim3(256),
                               kGroupLdsBytes > kSlideLdsBytes ? kGroupLdsBytes : kSlideLdsBytes, s, q, pos, win, n_win, n, het_mode, g,
                               group_edge_scans(hints), hs);
        } else if (const uint32_t group = slide_group(hints.window_step); group > 1)
            hipLaunchKernelGGL(dxy_het_query_slide_kernel, dim3(query_grid((n_win + group - 1) / group + 1), 3), dim3(256),
                               kSlideLdsBytes, s, q, pos, win, n_win, n, group);
        else
            hipLaunchKernelGGL(dxy_het_query_kernel, dim3(query_grid(n_win + 1), 3), dim3(256), 0, s, q, pos, win, n_win, n);
        if (int rc = hip_fail(hipGetLastError(), "dxy_het_query_kernel", err)) return rc;
    }
    return record(ev_query1, s, err);
}

}  // namespace pgt
