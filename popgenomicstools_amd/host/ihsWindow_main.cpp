// ihsWindow (MI355X host) — largest |iHS| per fixed bp window, its position, and the fraction of
// sites with |iHS| above a cutoff.  Same command line, messages, TSV and exit codes as the reference
// tool (ihsWindow.cpp:16-79 usage/arguments, :101-110 rows); windows and scores reduced through
// include/pgtwin.h.
//
//   ihsWindow <selscan normalized iHS *.norm file> [-winsize INT] [-cutoff FLOAT] [-chrlen FILE]
#include "extreme_common.h"

using namespace pgthost;

static void usage(unsigned W, double cutoff) {
    std::printf("\nUsage:\nihsWindow [selscan normalized iHS *.norm file] [options]\n"
                "\nAssumes iHS locus ID in format chr*_position\n\nOptions:\n"
                "-winsize INT Window size (bp) [%u]\n"
                "-cutoff FLOAT Determine fraction of sites with |iHS| > cutoff [%g]\n"
                "-chrlen FILE TSV-file with columns (1) chr (2) chromosome length (bp), and each row is a different chromosome\n"
                "\nOutput:\n(1) chromosome\n(2) window start\n(3) window stop\n(4) most extreme iHS score\n"
                "(5) extreme iHS position\n(6) proportion |iHS| > cutoff\n(7) Number SNPs in window\n\n", W, cutoff);
}

int main(int argc, char **argv) {
    uint32_t W = 100000;  // ihsWindow.cpp:225-226
    double cutoff = 2;
    const char *chrlen = nullptr;
    if (argc < 2) {  // ihsWindow.cpp:37-41: message, usage, exit status 1
        std::fprintf(stderr, "Must supply iHS input file\n");
        usage(W, cutoff);
        return 1;
    }
    for (int i = 2; i < argc; i += 2) {  // ihsWindow.cpp:49-76
        const char *opt = argv[i], *val = i + 1 < argc ? argv[i + 1] : "";
        if (!std::strcmp(opt, "-winsize")) {
            const int w = std::atoi(val);
            if (w <= 0) die("Window size must be a positive integer");
            W = (uint32_t)w;
        } else if (!std::strcmp(opt, "-cutoff")) {
            cutoff = std::atof(val);
            if (cutoff < 0) die("|iHS| cutoff must be >= zero");
        } else if (!std::strcmp(opt, "-chrlen")) {
            chrlen = val;
        } else {
            die(std::string("Unknown argument ") + opt);
        }
    }
    return run_extreme(argv[1], /*skip_header=*/false, /*score_field=*/4, W, PGT_EXT_IHS, cutoff, chrlen,
                       "Unable to open iHS file ");
}
