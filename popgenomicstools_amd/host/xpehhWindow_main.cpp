// xpehhWindow (MI355X host) — most extreme normalised XP-EHH per fixed bp window (minimum for a
// negative cutoff, maximum otherwise), its position, and the fraction of sites beyond the cutoff.
// Same command line, messages, TSV and exit codes as the reference tool (xpehhWindow.cpp:16-84
// usage/arguments, :104-113 rows); windows and scores reduced through include/pgtwin.h.
//
//   xpehhWindow <selscan normalized XP-EHH *.norm file> <cutoff> [-winsize INT] [-chrlen FILE]
#include "extreme_common.h"

using namespace pgthost;

static void usage(unsigned W) {
    std::printf("\nUsage:\nxpehhWindow <selscan normalized XP-EHH *.norm file> <cutoff> [options]\n"
                "\nInput file must have locus ID in format chr*_position\n"
                "cutoff (FLOAT): Calculate proportion of sites with EXP-EHH less (if negative) or greater (if positive) than cutoff\n"
                "\nOptions:\n-winsize INT Window size (bp) [%u]\n"
                "-chrlen FILE TSV-file with columns (1) chr (2) chromosome length (bp), and each row is a different chromosome\n"
                "\nOutput:\n(1) chromosome\n(2) window start\n(3) window stop\n"
                "(4) minimum (negative cutoff) or maximum (postive cutoff) XP-EHH score\n(5) extreme XP-EHH position\n"
                "(6) proportion XP-EHH scores > or < cutoff\n(7) Number SNPs in window\n\n", W);
}

int main(int argc, char **argv) {
    uint32_t W = 100000;
    const char *chrlen = nullptr;
    if (argc < 3) {  // xpehhWindow.cpp:42-46: message, usage, exit status 1
        std::fprintf(stderr, "Must supply XP-EHH file and cutoff value\n");
        usage(W);
        return 1;
    }
    const double cutoff = std::atof(argv[2]);
    if (cutoff == 0) std::fprintf(stderr, "WARNING: cutoff value of zero will calculate proportion of non-negative XP-EHH scores\n");
    for (int i = 3; i < argc; i += 2) {  // xpehhWindow.cpp:59-80
        const char *opt = argv[i], *val = i + 1 < argc ? argv[i + 1] : "";
        if (!std::strcmp(opt, "-winsize")) {
            const int w = std::atoi(val);
            if (w <= 0) die("Window size must be a positive integer");
            W = (uint32_t)w;
        } else if (!std::strcmp(opt, "-chrlen")) {
            chrlen = val;
        } else {
            die(std::string("Unknown argument ") + opt);
        }
    }
    return run_extreme(argv[1], /*skip_header=*/true, /*score_field=*/6, W, cutoff < 0 ? PGT_EXT_XP_MIN : PGT_EXT_XP_MAX,
                       cutoff, chrlen, "Unable to open XP-EHH inpt file ");
}
