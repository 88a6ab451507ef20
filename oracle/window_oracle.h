/*
 * window_oracle.h — CPU restatement of PopGenomicsTools' window-scan hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or
 * executed from the product (popgenomicstools_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
 * the checker.
 *
 * What it restates (file:line are relative to the reference tree):
 *   fstWindow.cpp:69-107  calcWindow      fstWindow.cpp:109-155 calcFst
 *   hetWindow.cpp:66-105  calcWindow      hetWindow.cpp:107-153 calcHeterozygosity
 *   dxyWindow.cpp:172-209 calcWindow      dxyWindow.cpp:253-436 maf2dxy
 *
 * How it is pinned:
 *   fst / het : byte-for-byte against oracle/_ref/{fstWindow,hetWindow} (the unmodified
 *               reference sources compiled by oracle/Makefile) on seeded random inputs and
 *               on the known-answer cases of SURVEY.md §4; fixtures in tests/golden/.
 *   dxy       : PARITY UNPINNED.  dxyWindow.cpp cannot be built in this image (it needs the
 *               Boost.Iostreams headers, dxyWindow.cpp:17-19; stand-ins are not allowed) and the
 *               reference holds no vectors.  tests/golden/dxy_kat.json (four cases recorded in
 *               SURVEY.md §4) pins nothing by the rules of this build.  oracle/Makefile builds
 *               _ref/dxyWindow wherever the real Boost exists and tests/golden/make_golden.py
 *               then writes ref_dxy.json, which the tests consume: one command pins it there.
 *
 * The algorithm here is deliberately the reference's streaming one — a W-entry buffer that
 * is re-summed sequentially for every window and shifted left by S — and deliberately NOT
 * the product's (run-length window table + radix-64 range tree), so that the two check
 * each other.
 */
#ifndef PGT_WINDOW_ORACLE_H
#define PGT_WINDOW_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t label;   /* chromosome RUN index whose name labels the row */
    uint32_t start;   /* first coordinate in the buffer  (fstWindow.cpp:71)  */
    uint32_t end;     /* last coordinate in the buffer   (fstWindow.cpp:72)  */
    uint32_t mid;     /* (start+end)/2 in unsigned 32-bit (fstWindow.cpp:73); 0 for dxy */
    uint32_t n;       /* fst: nsites; het: nonmissing; dxy: neffective */
    uint32_t nskip;   /* dxy: sites skipped for minind; else 0 */
    uint32_t printed; /* dxy: 0 when -skip_missing suppressed the row; else 1 */
    uint32_t pad_;
    uint64_t lo, hi;  /* global data-site index range that was reduced: [lo,hi) */
    double value;     /* fst: Σa/Σb or 0; het: nhet/nonmissing or 0; dxy: Σd */
    double num, den;  /* fst: Σa, Σb; het: nhet, nonmissing; dxy: Σd, 0 */
} orc_row;

typedef struct {
    double   sum;     /* dxyWindow.cpp:383 */
    uint32_t neff;    /* dxyWindow.cpp:384 */
    uint32_t nskip;   /* dxyWindow.cpp:385 */
} orc_dxy_total;

enum { ORC_OK = 0, ORC_EARG = 1, ORC_ECAP = 2, ORC_EIO = 3, ORC_EDOMAIN = 4 };

/* chr[i] is any id that changes exactly where the chromosome name changes between adjacent
 * lines (the reference compares adjacent names only, fstWindow.cpp:132).  Rows carry the RUN
 * index (0-based count of name changes) as label.  Returns ORC_ECAP if cap is too small
 * (n_out then holds the required count). */
int orc_fst_scan(const uint32_t *chr, const uint32_t *pos, const double *a, const double *b,
                 size_t n, uint32_t W, uint32_t S, orc_row *out, size_t cap, size_t *n_out);

int orc_het_scan(const uint32_t *chr, const uint32_t *pos, const int32_t *g,
                 size_t n, uint32_t W, uint32_t S, orc_row *out, size_t cap, size_t *n_out);

/* Both populations already synchronised (orc_dxy_text restates the two-file synchronisation of
 * dxyWindow.cpp:315-331 itself and passes the matched sites on).  run_chr_len[r] = -sizefile length of the chromosome
 * of run r (ignored when fixedsite != 0).  W == 0 → global only (requires fixedsite). */
int orc_dxy_scan(const uint32_t *chr, const uint32_t *pos, const double *p1, const double *p2,
                 const int32_t *n1, const int32_t *n2, size_t n, uint32_t W, uint32_t S,
                 int minind, int fixedsite, int skip_missing, const uint32_t *run_chr_len,
                 size_t n_runs, orc_row *out, size_t cap, size_t *n_out, orc_dxy_total *tot);

/* Allele-frequency front end.  WCFst() of betaAFOutlier.R:400-418, restated
 * literally: per site, from the allele frequencies f1, f2 of two populations with diploid sample
 * sizes n1, n2 (scalars), the Reynolds/Weir-Cockerham components (a, a+b) — the two columns
 * fstWindow consumes.  PARITY UNPINNED: the reference is an R script, Rscript is absent from this
 * image, and the reference holds no test vector for it; the formula at those lines is the spec. */
void orc_wcfst_site(double f1, double f2, double n1, double n2, double *a, double *a_plus_b);
/* a[i], ab[i] for every site (betaAFOutlier.R:415-417) */
void orc_wcfst_columns(const double *f1, const double *f2, size_t n, double n1, double n2,
                       double *a, double *ab);

/* ---- ihsWindow / xpehhWindow: non-overlapping bp windows, extreme score -------
 * Restates ihsWindow.cpp:123-221 and xpehhWindow.cpp:126-232 (one loop, two scoring rules).
 * Pinned byte-for-byte against oracle/_ref/{ihsWindow,xpehhWindow} (tests/golden/ref_extreme.json). */
enum { ORC_EXT_IHS = 0,      /* key |s|, count |s| > cutoff            (ihsWindow.cpp:193-205)   */
       ORC_EXT_XP_MAX = 1,   /* key s,   count s > cutoff (cutoff >= 0) (xpehhWindow.cpp:213-216) */
       ORC_EXT_XP_MIN = 2 }; /* key -s,  count s < cutoff (cutoff < 0)  (xpehhWindow.cpp:210-212) */
typedef struct {
    uint32_t label;   /* chromosome run */
    uint32_t start, end;
    uint32_t nsites, nbig;
    uint32_t position; /* of the extreme score; 0 when nsites == 0 */
    uint64_t lo, hi;   /* site range of the window */
    double value;      /* the extreme score itself (signed); 0 when nsites == 0 */
} orc_ext_row;
/* run_chr_len[r] = -chrlen length of run r's chromosome, 0 = not given (ihsWindow.cpp:156,172-174). */
int orc_extreme_scan(const uint32_t *chr, const uint32_t *pos, const double *score, size_t n, uint32_t W,
                     int mode, double cutoff, const uint32_t *run_chr_len, size_t n_runs,
                     orc_ext_row *out, size_t cap, size_t *n_out);
/* Text front ends with the tools' argv meaning (chrlen_path may be NULL/""), TSV to out_path. */
int orc_ihs_text_path(const char *in, uint32_t W, double cutoff, const char *chrlen_path, const char *out_path);
int orc_xpehh_text_path(const char *in, double cutoff, uint32_t W, const char *chrlen_path, const char *out_path);

/* Text front ends: same argv meaning and TSV as the reference tools, written to `out`
 * (and `err` for the dxy genome-wide line, dxyWindow.cpp:429-433).  Used for byte parity with
 * oracle/_ref and as the "port" CPU baseline when oracle/_ref is absent. */
int orc_fst_text(const char *path, uint32_t W, uint32_t S, FILE *out);
int orc_het_text(const char *path, uint32_t W, uint32_t S, FILE *out);
int orc_dxy_text(const char *maf1, const char *maf2, const char *sizefile, uint32_t W, uint32_t S,
                 int minind, int fixedsite, int skip_missing, FILE *out, FILE *err);

/* Path-based wrappers for ctypes callers. */
int orc_fst_text_path(const char *in, uint32_t W, uint32_t S, const char *out_path);
int orc_het_text_path(const char *in, uint32_t W, uint32_t S, const char *out_path);
int orc_dxy_text_path(const char *maf1, const char *maf2, const char *sizefile, uint32_t W, uint32_t S,
                      int minind, int fixedsite, int skip_missing, const char *out_path,
                      const char *err_path);

/* Fast text writers for synthetic inputs (bench cpu_baseline leg, golden generation). */
int orc_write_maf_text(const char *path, const uint32_t *chr, const uint32_t *pos, const double *freq,
                       const int32_t *nind, size_t n);
int orc_write_fst_text(const char *path, const uint32_t *chr, const uint32_t *pos,
                       const double *a, const double *b, size_t n);
int orc_write_het_text(const char *path, const uint32_t *chr, const uint32_t *pos,
                       const int32_t *g, size_t n);

#ifdef __cplusplus
}
#endif
#endif
