// pgt_internal.h — shared between the HIP kernels, the C-ABI and the host window builders.
// Not part of the public interface (that is include/pgtwin.h).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "pgtwin.h"

namespace pgt {

constexpr int kWave = 64;        // gfx950 wavefront
constexpr int kRadix = 64;       // children per tree node above the leaf level = one lane each
constexpr int kLeafF64 = 128;    // sites per level-1 node for f64 columns: one 16-B load per lane
constexpr int kLeafI8 = 1024;    // sites per level-1 node for the int8 genotype column: 16 B per lane
constexpr int kLeafExt = 256;    // sites per level-1 node for the single-column extreme-score scan: 2 x 16 B per lane
constexpr int kMaxLevels = 8;

// Node sizes in bytes (all levels of one tree use the same node type).
constexpr size_t kNodeFst = 16;  // {double Σa, double Σb}
constexpr size_t kNodeHet = 8;   // {u32 nonmissing, u32 nhet}
constexpr size_t kNodeDxy = 16;  // {double Σd, u32 neff, u32 nskip}

struct TreeLayout {
    int n_levels = 0;                   // levels 1..n_levels exist
    uint64_t count[kMaxLevels] = {};    // nodes stored per level (level 1 padded to 64 * count[1])
    size_t offset[kMaxLevels] = {};     // byte offset of each level in the workspace
    size_t partials = 0;                // dxy only: byte offset of the build waves' partial sums (the genome-wide line)
    size_t bytes = 0;
};
// The build kernels of the f64 trees run on at most this many waves (512 workgroups of 4: what is resident at once, see
// build_grid in pgt_kernels.hip); the dxy build leaves one partial sum per wave behind the tree.
constexpr uint32_t kMaxBuildWaves = 2048;

inline int leaf_sites(int stat) { return stat == PGT_STAT_HET ? kLeafI8 : (stat == PGT_STAT_EXT ? kLeafExt : kLeafF64); }
inline size_t node_bytes(int stat) { return stat == PGT_STAT_HET ? kNodeHet : 16; }

// Level 1 and 2 come out of the streaming build kernel; higher levels are added while a level
// still has more than 64 nodes (so the top level is always reducible by one wave-wide load).
inline TreeLayout tree_layout_for(uint64_t leaf, size_t nb, uint64_t n_sites) {
    TreeLayout t;
    uint64_t n_l2 = (n_sites + leaf * kRadix - 1) / (leaf * kRadix);
    if (n_l2 == 0) n_l2 = 1;
    uint64_t c = n_l2 * kRadix;
    size_t off = 0;
    int k = 0;
    auto push = [&](uint64_t cnt) {
        t.count[k] = cnt;
        t.offset[k] = off;
        off += ((cnt * nb + 255) / 256) * 256;
        ++k;
    };
    push(c);       // level 1
    push(n_l2);    // level 2
    c = n_l2;
    while (c > (uint64_t)kRadix && k < kMaxLevels) {
        c = (c + kRadix - 1) / kRadix;
        push(c);
    }
    t.n_levels = k;
    t.bytes = off;
    return t;
}
inline TreeLayout tree_layout(int stat, uint64_t n_sites) {
    TreeLayout t = tree_layout_for((uint64_t)leaf_sites(stat), node_bytes(stat), n_sites);
    if (stat == PGT_STAT_DXY) {  // + one node per build wave: the genome-wide line is their sum (no upper levels needed for it)
        t.partials = t.bytes;
        t.bytes += (size_t)kMaxBuildWaves * kNodeDxy;
    }
    return t;
}

// Levels worth building when no window is longer than max_window sites (0 = unknown: all).
// Level k (k >= 2) has nodes of leaf*64^(k-1) sites; a window can only contain such a node if it
// is at least that long.  The levels the build kernel writes itself always exist: `built` = 2 for the f64
// trees (levels 1 and 2 leave the streaming kernel together), 1 for the int8 tree, whose level 2 (65536 sites)
// is a tree_up launch of its own — skipped for windows shorter than that (round 5: W = 50000).
inline int useful_levels_for(const TreeLayout &t, uint64_t leaf, uint64_t max_window, int built = 2) {
    if (max_window == 0) return t.n_levels;
    int k = built;
    uint64_t node = leaf * kRadix;  // level-2 node
    for (int j = 2; j <= built; ++j) node *= kRadix;  // the first level above the built ones
    while (k < t.n_levels && node <= max_window) {
        ++k;
        node *= kRadix;
    }
    return k;
}
inline int useful_levels(const TreeLayout &t, int stat, uint64_t max_window) {
    return useful_levels_for(t, (uint64_t)leaf_sites(stat), max_window, stat == PGT_STAT_HET ? 1 : 2);
}

// ---- allele-frequency front end (pgt_af_kernels.hip): nodes of V scalar sums, node-major ------
// A node is V consecutive doubles (V = NP + NP(NP-1)/2), the nodes of a level are contiguous: the 16
// level-1 nodes (512 sites each) a build wave finishes together are ONE contiguous block of 128*V bytes (4.5 KiB
// at 8 populations), and a query lane reads a node as one contiguous run.  Levels 2 and up are those of the
// f64 layout (8192 sites, x64 per level).
constexpr int kAfMaxPops = 8;
constexpr int kAfLeafPieces = 4;  // 128-site pieces per level-1 node of the AF tree (pgt_af_kernels.hip: kAfPieces)
struct AfTree {
    char *base;
    size_t off[kMaxLevels];  // byte offset of level slot k
    int n_levels;
    int n_vals;              // V: doubles per node
};
// the AF tree's level 1 holds 64 / kAfLeafPieces nodes per level-2 node (pgt_af_kernels.hip), a fraction of the f64 layout's count
inline size_t af_level_bytes(const TreeLayout &t, int k, int n_vals) {
    const uint64_t nodes = k == 0 ? t.count[0] / kAfLeafPieces : t.count[k];
    return ((nodes * (size_t)n_vals * 8 + 255) / 256) * 256;
}
inline size_t af_tree_bytes(const TreeLayout &t, int n_vals) {
    size_t b = 0;
    for (int k = 0; k < t.n_levels; ++k) b += af_level_bytes(t, k, n_vals);
    return b;
}
inline AfTree af_tree_view(const TreeLayout &t, int n_vals, void *tree, int levels) {
    AfTree v{};
    v.base = static_cast<char *>(tree);
    v.n_levels = levels;
    v.n_vals = n_vals;
    size_t off = 0;
    for (int k = 0; k < t.n_levels; ++k) {
        v.off[k] = off;
        off += af_level_bytes(t, k, n_vals);
    }
    return v;
}

// Speed-only hints of a context (pgt_set_max_window, pgt_set_window_step); 0 = unknown.
struct Hints {
    uint64_t max_window = 0;   // longest window in sites: tree levels with larger nodes are not built
    uint64_t window_step = 0;  // typical distance between consecutive window starts: selects the query strategy
    uint64_t typical_window = 0;  // typical window length where the library saw the table itself (host tables); 0: max_window stands for it
};
// ---- query strategies, chosen from the hints alone (the same on every rank when the hints are) ----------------
// step == 0 (unknown) or large: one wave per window.  0 < step <= kGroupMaxStep and windows of at least two level-2
// tiles (max-window hint): the GROUP query (64 consecutive windows per wave share tile scans, level-1 node scans and the
// interior) — it beats the sliding query at every step it applies to (10^8 sites, W = 50000: S = 1 2.85 against 4.10 ms,
// S = 8 0.65 against 1.80, S = 32 0.60 against 1.55; profiles/r03/measure_query_1e8_strategies.md).  Shorter windows
// with step <= kSlideMaxStep: the sliding query (per-site scans of 128-site tiles; it needs only 256-site windows).
constexpr uint64_t kSlideMaxStep = 32;
constexpr uint64_t kGroupMaxStep = 1024;  // above, a group's starts span too many level-2 tiles and there are too few groups to fill the chip (10^8 sites: a tie at 2048)
// Windows per wave of the sliding query for a given step (0 or 1 = not the sliding query).
inline uint32_t slide_group(uint64_t step) {
    if (step == 0 || step > kSlideMaxStep) return 1;
    const uint64_t g = 128 / step + 1;
    return (uint32_t)(g > 64 ? 64 : g);
}
// leaf = sites per level-1 node of the statistic's tree (its level-2 tiles are 64 leaves)
inline bool group_query(const Hints &h, uint64_t leaf) {
    const uint64_t typical = h.typical_window ? h.typical_window : h.max_window;
    return h.window_step > 0 && h.window_step <= kGroupMaxStep && typical >= 2 * leaf * kRadix;
}
// the group query's ragged ends: shared 128-site tile scans up to this step, per window above (see query_group_body)
inline int group_edge_scans(const Hints &h) { return h.window_step <= 64 ? 1 : 0; }
// Windows per wave of the group query: 64 when that still gives the chip ~8000 waves, else fewer (rows do not depend on it)
inline uint32_t group_size(uint64_t n_win) {
    return n_win >= 64ull * 8192 ? 64u : (n_win >= 32ull * 8192 ? 32u : 16u);
}

// ---- launchers implemented in pgt_kernels.hip (stream = hipStream_t as void*) ----------
int launch_fst(const uint32_t *pos, const double *const *a, const double *const *b, uint32_t n_pairs,
               uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree,
               void *stream, void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints);
int launch_het(const uint32_t *pos, const int8_t *g, uint64_t n, const pgt_win *win, uint64_t n_win,
               pgt_het_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1,
               void *ev_query1, std::string *err, const Hints &hints);
int launch_dxy(const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
               const int32_t *n2, uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
               pgt_dxy_row *out, pgt_dxy_total *tot, void *tree, void *stream, void *ev_build0,
               void *ev_build1, void *ev_query1, std::string *err, const Hints &hints);

int launch_dxy_het(const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
                   const int32_t *n2, const int8_t *g1, const int8_t *g2, uint64_t n, int minind,
                   const pgt_win *win, uint64_t n_win, pgt_dxy_row *dxy_out, pgt_dxy_total *tot,
                   pgt_het_row *het_out1, pgt_het_row *het_out2, void *tree, void *stream,
                   void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints);

int launch_ext(const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff, const pgt_win *win,
               uint64_t n_win, pgt_ext_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1,
               void *ev_query1, std::string *err, const Hints &hints);
int launch_fst_af(const uint32_t *pos, const double *const *freq, const double *nsamp, uint32_t n_pops,
                  uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree, void *stream,
                  void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints);

// pgt_windows.cpp: the site-window rules per chromosome run in closed form (see there); plain data, also read by
// the kernel that writes a window table on the device (pgt_kernels.hip: launch_windows_from_plan)
struct RunPlan {
    uint64_t out0 = 0;  // index of the run's first window in the table
    uint64_t K = 0;     // full windows [hi0 + k*S - W, hi0 + k*S), k < K
    uint64_t hi0 = 0;
    uint64_t tail_lo = 0, tail_hi = 0;  // one more window [tail_lo, tail_hi) at the end of the run ...
    uint32_t tail = 0, pad_ = 0;        // ... if tail != 0
};
// -> PGT_OK and the plan of (run_len, W, S) in site mode + the number of windows; argument errors as pgt_build_windows_sites
int plan_site_windows(const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S, std::vector<RunPlan> &plan, uint64_t *count);
int launch_windows_from_plan(const RunPlan *d_plan, uint64_t n_runs, uint64_t n_win, uint32_t W, uint32_t S, pgt_win *d_out,
                             void *stream, std::string *err);

// pgt_kernels.hip: words[i] = splitmix64(seed + i) by plain global stores (pgt_rowbuf_fill: mapping self-test)
int launch_fill_pattern(uint64_t *words, uint64_t n_words, uint64_t seed, void *stream, std::string *err);

// pgt_ingest.hip: text -> device columns + chromosome runs (synchronous, default stream of `device`)
int ingest_text(int device, const char *text, size_t len, const uint8_t *tokens, int n_tokens, uint64_t front, pgt_ingest **out, std::string *err);
size_t ingest_column_bytes(const pgt_ingest *ing, int token);  // rows * element size of the token's column (0: no column)

int init_kernels(std::string *err);     // pgt_kernels.hip: one-time kernel attributes (called by pgt_open)
int init_af_kernels(std::string *err);  // pgt_af_kernels.hip

// thread-local message for the ctx-less entry points
void set_global_error(const std::string &msg);
const std::string &global_error();

}  // namespace pgt
