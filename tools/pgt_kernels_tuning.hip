// pgt_kernels_tuning.hip — the TUNING build of the kernel translation unit (tools/build_ab.py, wave_timeline.py, size_sweep.py).
// NOT part of the product: popgenomicstools_amd/build.py compiles this file INSTEAD of csrc/pgt_kernels.hip only when
// PGT_EXTRA_HIPCC_FLAGS contains -DPGT_TUNING_BUILD.  It includes the product kernels textually, renames the two launchers that
// have tunable builds and re-defines them with the build launch chosen per call from environment variables (PGT_TUNE_FST,
// PGT_TUNE_BUILD_STAMPS, PGT_EXT_VARIANT[_NOW]); without those variables the product launch runs.  The product file itself
// carries no preprocessor switch and no getenv.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define launch_fst launch_fst_product
#define launch_ext launch_ext_product
#include "../popgenomicstools_amd/csrc/pgt_kernels.hip"
#undef launch_fst
#undef launch_ext

namespace pgt {
namespace {
#include "pgt_build_experiments.inc"

void ext_build_tuned(hipStream_t s, const ExtBuildArgs &g, uint64_t n, uint64_t n_l2, const TreeView &tv) {
    static const int variant = getenv("PGT_EXT_VARIANT") ? atoi(getenv("PGT_EXT_VARIANT")) : 0;
    const char *e = getenv("PGT_EXT_VARIANT_NOW");  // re-read per call for interleaved A/B in one process
    switch (e ? atoi(e) : variant) {                 // <stage tiles, leaf tiles per batch (x2 loads), deferred>(workgroup cap)
        case 1: return launch_ext_variant<16, 2, true>(s, g, n, n_l2, tv, 512);   // 8 waves per CU x 4 loads in flight
        case 2: return launch_ext_variant<16, 4, true>(s, g, n, n_l2, tv, 512);   // 8 x 8
        case 3: return launch_ext_variant<16, 8, true>(s, g, n, n_l2, tv, 512);   // 8 x 16
        case 4: return launch_ext_variant<16, 4, true>(s, g, n, n_l2, tv, 256);   // 4 x 8
        case 5: return launch_ext_variant<16, 8, true>(s, g, n, n_l2, tv, 256);   // 4 x 16
        case 6: return launch_ext_variant<8, 4, true>(s, g, n, n_l2, tv, 1024);   // 16 x 8
        case 7: return launch_ext_variant<8, 8, true>(s, g, n, n_l2, tv, 1024);   // 16 x 16
        case 8: return launch_ext_variant<8, 2, true>(s, g, n, n_l2, tv, 1024);   // 16 x 4
        case 9: return launch_ext_variant<4, 4, true>(s, g, n, n_l2, tv, 2048);   // 32 x 8
        case 10: return launch_ext_variant<4, 2, true>(s, g, n, n_l2, tv, 2048);  // 32 x 4
        default: return ext_build_launch(s, g, n, n_l2, tv);
    }
}
}  // namespace

int launch_fst(const uint32_t *pos, const double *const *a, const double *const *b, uint32_t n_pairs,
               uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, void *tree, void *stream,
               void *ev_build0, void *ev_build1, void *ev_query1, std::string *err, const Hints &hints) {
    auto build = [](hipStream_t s, const PairCols &cols, uint32_t np, uint64_t n_, const TreeLayout &tl, const TreeView &tv) {
        if (!launch_fst_experiment(s, cols, np, n_, tl, tv)) fst_build_launch(s, cols, np, n_, tl, tv);
    };
    return launch_fst_with(build, pos, a, b, n_pairs, n, win, n_win, out, tree, stream, ev_build0, ev_build1, ev_query1, err, hints);
}

int launch_ext(const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff, const pgt_win *win,
               uint64_t n_win, pgt_ext_row *out, void *tree, void *stream, void *ev_build0, void *ev_build1,
               void *ev_query1, std::string *err, const Hints &hints) {
    return launch_ext_with(ext_build_tuned, pos, score, n, mode, cutoff, win, n_win, out, tree, stream, ev_build0, ev_build1,
                           ev_query1, err, hints);
}

}  // namespace pgt
