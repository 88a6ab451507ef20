/*
 * pgtwin.h — C-ABI of libpgtwin: the MI355X (gfx950) window-scan engine that replaces the
 * per-window reduction of PopGenomicsTools' fstWindow / hetWindow / dxyWindow.
 *
 * The reference has no FFI or plugin interface; the seam this library replaces is the
 * function `calcWindow` and the bookkeeping around its call sites (file:line in the reference):
 *     fstWindow.cpp:69-107   called from fstWindow.cpp:134,137,151
 *     hetWindow.cpp:66-105   called from hetWindow.cpp:132,135,149
 *     dxyWindow.cpp:172-209  called from dxyWindow.cpp:346,354,358,366,377,416,425
 * The reference calls it once per window on a W-entry buffer it owns and re-sums; this library
 * is called once per input: the host describes every window as a range [lo,hi) of the global
 * site index (pgt_build_windows_*), and the device reduces all of them from structure-of-arrays
 * columns resident in HBM (pgt_*_reduce*).
 *
 * Conventions
 *   - plain C types only; every function returns PGT_OK (0) or a PGT_E* code; the message is
 *     retrievable with pgt_last_error(ctx) (ctx may be NULL for the ctx-less functions);
 *     the reference's tools exit 255 on error (`return -1`, fstWindow.cpp:42,48,55) and the
 *     retained C++ hosts do the same after printing that message;
 *   - the caller owns every buffer; the library retains none of the caller's memory past return (a context keeps
 *     its own scratch between calls: the pinned staging ring and the window / row / tree workspace of the host-buffer
 *     entry points, 64 MiB + about 1 % of the largest input seen; pgt_close frees them);
 *   - one pgt_ctx per GPU and per thread (one process per GPU); a ctx is not thread-safe;
 *   - there is NO CPU fallback: without a usable gfx950 device pgt_open fails.
 */
#ifndef PGTWIN_H
#define PGTWIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 5): the i32 count columns of the dxy entry points must be 16-byte aligned (8 sufficed); pgt_tree_bytes(PGT_STAT_DXY)
 * includes the build waves' partial sums; pgt_extreme_reduce_cols, PGT_TOK_CHR_PREFIX and 12 tokens per line (added under
 * version 4 in round 4) are part of it.  A binding written for one version never calls a library of another. */
/* 6 (round 6): pgt_prepare_host_io added; the host-buffer entry points stage their uploads through a per-context pinned
 * ring and keep a per-context workspace (no argument list changed). */
#define PGT_ABI_VERSION 6

enum {
    PGT_OK = 0,
    PGT_EARG = 1,     /* bad argument (NULL, S>W, misaligned device pointer, ...) */
    PGT_ECAP = 2,     /* output capacity too small; *n_out holds the required count */
    PGT_EDEVICE = 3,  /* HIP error / no gfx950 device */
    PGT_EDOMAIN = 4,  /* input outside the reference's defined domain */
    PGT_ENOMEM = 5
};

typedef struct pgt_ctx pgt_ctx;

/* One window = one call of the reference's calcWindow.
 * flags bit0 (PGT_WIN_COORDS): start/end are given here (dxy bp mode: bp-slot coordinates,
 * dxyWindow.cpp:347,367,389); otherwise the reduce fills them from pos[lo], pos[hi-1]
 * (fstWindow.cpp:71-72). */
#define PGT_WIN_COORDS 1u
typedef struct {
    uint64_t lo, hi;    /* global site index range [lo,hi) reduced by this window; lo==hi allowed */
    uint32_t label_run; /* chromosome run whose name labels the row */
    uint32_t flags;
    uint32_t start, end;
} pgt_win;

typedef struct { /* fstWindow.cpp:88 row: chr start end mid fst nsites */
    uint32_t start, end, mid, n;
    double fst;        /* bsum != 0 ? asum/bsum : 0   (fstWindow.cpp:85) */
    double asum, bsum; /* the two window sums, full precision */
} pgt_fst_row;

typedef struct { /* hetWindow.cpp:87 row: chr start end mid h nonmissing */
    uint32_t start, end, mid, nonmissing;
    uint32_t nhet, pad_;
    double h; /* nonmissing ? nhet/nonmissing : 0   (hetWindow.cpp:84) */
} pgt_het_row;

typedef struct { /* dxyWindow.cpp:190 row: chr start end dxy neffective nskip */
    uint32_t start, end, neff, nskip;
    double sum;
} pgt_dxy_row;

typedef struct { /* dxyWindow.cpp:429-433 genome-wide line */
    double sum;
    uint64_t neff, nskip;
} pgt_dxy_total;

/* ---- context ------------------------------------------------------------------------- */
/* device: HIP ordinal, or -1 for the current device.  Fails (NULL, message via
 * pgt_last_error(NULL)) when no HIP device is usable. */
pgt_ctx *pgt_open(int device);
void pgt_close(pgt_ctx *ctx);
const char *pgt_last_error(const pgt_ctx *ctx);
int pgt_abi_version(void);
/* Optional: allocate the pinned staging ring of the host-buffer entry points (pgt_*_reduce with host columns) now
 * (~15 ms of hipHostMalloc + the runtime's one-time 30 … 60 ms set-up of the first copies of a process) instead of inside the
 * first such call — the retained hosts call it on the thread that opens the device, beside the text parse.
 * expected_column_bytes: what the caller expects to upload per call (0 = unknown); below 32 MiB uploads do not take the ring
 * and it is not allocated (only the set-up is paid).  Replaces nothing in the reference (its calcWindow reads the caller's
 * buffer in place, fstWindow.cpp:76-83); it exists because the columns have to cross PCIe here. */
int pgt_prepare_host_io(pgt_ctx *ctx, uint64_t expected_column_bytes);

/* ---- window tables (host, O(#windows + #runs), no device needed) ------------------------ */
/* Site-count windows of fstWindow / hetWindow / dxyWindow -fixedsite 1: the emission rules of
 * fstWindow.cpp:132-138,150-152 applied to the chromosome run lengths (run_len[r] = number of
 * consecutive sites carrying the same chromosome name).  Requires 1 <= S <= W (Q9).
 * out may be NULL (count only).  Returns PGT_ECAP when cap < *n_out. */
int pgt_build_windows_sites(const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S,
                            pgt_win *out, size_t cap, size_t *n_out);

/* Base-pair windows of dxyWindow (-fixedsite 0): the slot machine of dxyWindow.cpp:334-378,
 * 407-426 over bp slots 1..chr_len of every run; pos (host pointer, n sites, strictly
 * increasing inside a run) locates the data sites of each window.  chr_len[r] is the -sizefile
 * length of run r's chromosome.  Rows carry PGT_WIN_COORDS. */
int pgt_build_windows_bp(const uint32_t *pos, const uint64_t *run_len, const uint32_t *chr_len,
                         size_t n_runs, uint32_t W, uint32_t S, pgt_win *out, size_t cap,
                         size_t *n_out);

/* ---- reductions, host buffers (copy in, reduce on the GPU, copy out; synchronous) ------- */
int pgt_fst_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b, uint64_t n,
                   const pgt_win *win, uint64_t n_win, pgt_fst_row *out);
int pgt_het_reduce(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n,
                   const pgt_win *win, uint64_t n_win, pgt_het_row *out);
/* tot may be NULL.  n_win may be 0 (global only: dxyWindow -winsize 0 -fixedsite 1). */
int pgt_dxy_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2,
                   const int32_t *n1, const int32_t *n2, uint64_t n, int minind,
                   const pgt_win *win, uint64_t n_win, pgt_dxy_row *out, pgt_dxy_total *tot);

/* ---- reductions, device-resident columns (asynchronous on `stream`) --------------------- */
/* Every pointer is a DEVICE pointer on ctx's device; the f64, i32 (n1, n2) and i8 columns must be 16-byte
 * aligned (they are read by 16-byte loads; the i32 columns since round 5 — 8 bytes sufficed before).
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  `tree` is caller
 * workspace of at least pgt_tree_bytes(stat, n) bytes, 256-byte aligned: it receives the
 * radix-64 range tree and may be reused by later calls.
 * ABI 4: every row output carries its CAPACITY in bytes (`out_bytes`, right after the pointer, as
 * `tree_bytes` follows `tree`): a call whose rows would not fit returns PGT_EARG before anything is
 * launched.  In the multi-GPU peer mode `out` points into ANOTHER GPU's row buffer (pgt_rowbuf_open),
 * where an overrun would be a silent cross-device write. */
enum { PGT_STAT_FST = 0, PGT_STAT_HET = 1, PGT_STAT_DXY = 2, PGT_STAT_EXT = 3 };
size_t pgt_tree_bytes(int stat, uint64_t n_sites);

int pgt_fst_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b,
                       uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out,
                       size_t out_bytes, void *tree, size_t tree_bytes, void *stream);
int pgt_het_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n,
                       const pgt_win *win, uint64_t n_win, pgt_het_row *out, size_t out_bytes,
                       void *tree, size_t tree_bytes, void *stream);
/* tot: device pointer to one pgt_dxy_total (the genome-wide line, dxyWindow.cpp:382-385,429-433), or NULL.  Its counts are
 * exact; its sum is added from one partial sum per build wave, in wave order — a function of n alone (the build grid is static),
 * within 1e-15 relative of any other order of the same additions. */
int pgt_dxy_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2,
                       const int32_t *n1, const int32_t *n2, uint64_t n, int minind,
                       const pgt_win *win, uint64_t n_win, pgt_dxy_row *out, size_t out_bytes,
                       pgt_dxy_total *tot, void *tree, size_t tree_bytes, void *stream);

/* dxyWindow + hetWindow of two genotype columns over ONE position column and ONE window table
 * (BASELINE config 3): one build launch streams all six columns (26 B/site), one query launch
 * answers the three tables.  tree must hold pgt_tree_bytes(PGT_STAT_DXY, n) +
 * 2 * pgt_tree_bytes(PGT_STAT_HET, n) bytes.  Rows equal those of the separate calls bit for bit.
 * het_out_bytes is the capacity of EACH of the two genotype tables. */
int pgt_dxy_het_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2,
                           const int32_t *n1, const int32_t *n2, const int8_t *g1, const int8_t *g2,
                           uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
                           pgt_dxy_row *dxy_out, size_t dxy_out_bytes, pgt_dxy_total *tot,
                           pgt_het_row *het_out1, pgt_het_row *het_out2, size_t het_out_bytes,
                           void *tree, size_t tree_bytes, void *stream);

/* Batched population pairs sharing one position column and one window table (BASELINE
 * config 5): a[p], b[p] are HOST arrays of n_pairs DEVICE column pointers; out holds
 * n_pairs * n_win rows, pair-major; tree holds n_pairs trees (n_pairs * pgt_tree_bytes). */
int pgt_fst_reduce_pairs_dev(pgt_ctx *ctx, const uint32_t *pos, const double *const *a,
                             const double *const *b, uint32_t n_pairs, uint64_t n,
                             const pgt_win *win, uint64_t n_win, pgt_fst_row *out, size_t out_bytes,
                             void *tree, size_t tree_bytes, void *stream);

/* ---- allele-frequency front end ------------------------------------------ */
/* Population allele frequencies -> Reynolds / Weir-Cockerham variance components exactly as WCFst()
 * of the reference's betaAFOutlier.R:400-418 (per site a and a+b: the two columns fstWindow reads),
 * then fstWindow's window statistic Σa / Σ(a+b) (fstWindow.cpp:76-85), for ALL pairs i<j of n_pops
 * populations in one pass over the frequency columns (8 B/site/population instead of 16 B/site/pair).
 *   freq   host array of n_pops DEVICE pointers: f64 allele frequency per site, 16-byte aligned
 *   nsamp  host array of n_pops diploid sample sizes (the R arguments n1, n2)
 *   out    n_pairs * n_win rows, pair-major, pairs ordered (0,1),(0,2),..,(0,n_pops-1),(1,2),..;
 *          asum = Σa, bsum = Σ(a+b), n = sites in the window
 *   tree   pgt_af_tree_bytes(n_pops, n) bytes.    2 <= n_pops <= 8.
 * Parity note: the reference side is an R function with no runnable oracle in this image; the rows
 * are checked against a literal C restatement of those lines (oracle/, parity unpinned). */
size_t pgt_af_tree_bytes(uint32_t n_pops, uint64_t n_sites);
int pgt_fst_af_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *const *freq, const double *nsamp,
                          uint32_t n_pops, uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out,
                          size_t out_bytes, void *tree, size_t tree_bytes, void *stream);

/* ---- ihsWindow / xpehhWindow: extreme score in non-overlapping bp windows ---- */
/* Replaces the window bookkeeping and per-window scan of ihsWindow.cpp:123-221 and
 * xpehhWindow.cpp:126-232: the most extreme score of the window (first occurrence on ties), its
 * position, and the number of scores beyond the cutoff. */
enum {
    PGT_EXT_IHS = 0,    /* extreme = largest |s|, count |s| > cutoff           (ihsWindow.cpp:193-205)   */
    PGT_EXT_XP_MAX = 1, /* extreme = largest s,   count s > cutoff (cutoff>=0) (xpehhWindow.cpp:213-216) */
    PGT_EXT_XP_MIN = 2  /* extreme = smallest s,  count s < cutoff (cutoff<0)  (xpehhWindow.cpp:210-212) */
};
typedef struct { /* ihsWindow.cpp:101-110 row: chr start end score position proportion nsites */
    uint32_t start, end, nsites, nbig;
    uint32_t position, pad_; /* position of the extreme score; 0 when nsites == 0 (the tool prints NA) */
    double value;            /* the extreme score (signed) */
} pgt_ext_row;
/* The tools' window rules applied site by site on the host (they are history dependent: a site at
 * pos >= window end opens the next window, ihsWindow.cpp:176-190): fixed bp windows [1,W],[W+1,2W],..
 * per chromosome, clamped to chr_len[r] where given (0 = not given; chr_len may be NULL), empty
 * windows included, trailing windows up to the chromosome length.  O(n).  Rows carry
 * PGT_WIN_COORDS.  PGT_EDOMAIN for n == 0 or a position beyond a given chromosome length (the
 * reference prints a nameless window / loops forever there). */
int pgt_build_windows_extreme(const uint32_t *pos, const uint64_t *run_len, const uint32_t *chr_len,
                              size_t n_runs, uint32_t W, pgt_win *out, size_t cap, size_t *n_out);
int pgt_extreme_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *score, uint64_t n, int mode,
                       double cutoff, const pgt_win *win, uint64_t n_win, pgt_ext_row *out);
/* device-resident form; tree: pgt_tree_bytes(PGT_STAT_EXT, n) bytes */
int pgt_extreme_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *score, uint64_t n, int mode,
                           double cutoff, const pgt_win *win, uint64_t n_win, pgt_ext_row *out,
                           size_t out_bytes, void *tree, size_t tree_bytes, void *stream);

/* ---- performance hint ------------------------------------------------------------------- */
/* Longest window (in sites) the following *_dev calls will be asked for; 0 (the default) = unknown.
 * Tree levels whose nodes are larger than this are not built (a query never touches them).  The
 * hint only affects speed: a longer window is still answered correctly from the levels that
 * exist.  The host-buffer / *_cols entry points derive a hint from the table only while it is unknown (0): a caller
 * that reduces SLICES of one table on several GPUs sets both hints from the whole table (W, S), so that every
 * slice — and the single-GPU run — takes the same query strategy: rows are bitwise independent of the number of
 * GPUs only under identical hints (the strategies add in different orders).
 * The value also stands for the TYPICAL window length when the query strategy is chosen (pgt_set_window_step): pass the
 * tools' W, not a loose upper bound — the group strategy is taken for windows of at least two level-2 tiles, and with
 * windows much shorter than the hint it degenerates to one range query after the other (correct, slow). */
int pgt_set_max_window(pgt_ctx *ctx, uint64_t max_window_sites);
/* Typical distance, in sites, between the starts of consecutive windows of the following *_dev calls
 * (the tools' step size S); 0 (the default) = unknown: one wave per window.  The step selects the query
 * strategy for the regime `-winsize W -stepsize S` with S << W, where the reference re-sums W sites and
 * shifts W-S per window (fstWindow.cpp:80-83,95-99):
 *   GROUP    1 <= S <= 1024 and a longest-window hint of at least two level-2 tiles (16384 sites for the f64
 *            trees, 131072 for the genotype tree): one wave answers up to 64 CONSECUTIVE windows, one per lane;
 *            they share the scans of the 128-site tiles under their ends (S <= 64), the scans of the level-1
 *            nodes beside those and ONE tree query per distinct interior;
 *   SLIDING  S <= 32 with shorter (or unknown) windows: per-site suffix / prefix scans of the 128-site tiles a
 *            group of 128/S + 1 windows starts and ends in, and one tree query for the interior they share.
 * Speed only: any table is answered correctly whatever the hint (windows a strategy does not fit fall back to
 * the plain range query); the strategies add in different orders, so the last bits of a float may differ between
 * them — under one pair of hints, rows are bitwise independent of which windows share a wave and of the number of
 * GPUs.  The host-buffer entry points derive the hint from the table while it is unset. */
int pgt_set_window_step(pgt_ctx *ctx, uint64_t step_sites);
/* Tables whose windows vary in length (dxyWindow's base-pair windows): the typical (median) window length, which then
 * decides between the strategies instead of the longest.  Call it AFTER pgt_set_max_window (which resets it to "the same").
 * pgt_table_hints gives the three values exactly as the host-buffer entry points derive them from a table while the hints
 * are unset: a caller that reduces SLICES of one table on several GPUs sets them on every context, so that every slice
 * is reduced as the whole table would be (rows bitwise independent of the slicing).  The step comes back as UINT64_MAX
 * where the table has no typical positive step (0 would mean "unknown" and let every slice estimate its own). */
int pgt_set_typical_window(pgt_ctx *ctx, uint64_t typical_sites);
int pgt_table_hints(const pgt_win *win, uint64_t n_win, uint64_t *max_window, uint64_t *typical_window, uint64_t *window_step);

/* ---- per-kernel timing (HIP events on the launch stream; for bench.py's roofline) ------- */
/* When enabled, the *_dev entry points bracket the tree-build kernel and the window-query
 * kernel with HIP events; pgt_last_kernel_ms synchronises on them and returns both. */
int pgt_set_profiling(pgt_ctx *ctx, int enabled);
int pgt_last_kernel_ms(pgt_ctx *ctx, float *build_ms, float *query_ms);

/* ---- multi-GPU sharding plan (host) ----------------------------------------------------- */
/* Split a window table into n_ranks contiguous blocks balanced by reduced sites.  Rank r owns
 * windows [win_begin, win_end) and needs the site columns [site_lo, site_hi); site_lo is
 * rounded down to a multiple of max(65536, the largest power of two <= the longest window), so
 * that every tree node a query can touch (nodes are powers of two of sites for every statistic)
 * has the same boundaries as in the single-GPU tree: results are bitwise independent of n_ranks. */
typedef struct {
    uint64_t win_begin, win_end;
    uint64_t site_lo, site_hi;
} pgt_shard;
int pgt_plan_shards(const pgt_win *win, uint64_t n_win, uint32_t n_ranks, pgt_shard *out);

/* ---- multi-GPU row exchange without a per-step collective ----------------- */
/* The reference prints each window's row as soon as calcWindow has reduced it (fstWindow.cpp:88); in
 * the sharded form rank 0 prints, so every rank's rows must reach rank 0.  Instead of a gather per
 * scan, rank 0 creates ONE row buffer for the whole window table and exports it; every other rank
 * (one process per GPU) opens it and passes `base + its first window * row size` as the `out`
 * pointer of its *_dev call: the query kernel's row stores then travel over xGMI straight into rank
 * 0's HBM.  After every rank has synchronised its stream (plus one barrier of the caller's process
 * group) rank 0 holds the assembled table.  Rows of different ranks should start on 256-byte
 * boundaries of the buffer (pad between blocks) so that no cache line is written by two GPUs.
 *   create  hipMalloc on ctx's device + hipIpcGetMemHandle; the handle is 64 plain bytes that the
 *           caller ships to the other processes (e.g. torch.distributed.broadcast_object_list)
 *   open    hipIpcOpenMemHandle on ctx's device (peer mapping); fails in the creating process
 *   close   owner != 0: hipFree;  owner == 0: hipIpcCloseMemHandle
 *   read    rank 0: device -> host copy of the assembled rows, ordered after `stream` */
#define PGT_IPC_HANDLE_BYTES 64
typedef struct { unsigned char opaque[PGT_IPC_HANDLE_BYTES]; } pgt_ipc_handle;
/* PGT_OK when ctx's device can address memory of HIP device `peer_device` (peer access is enabled as a side
 * effect); every rank checks this against the creating rank's device before it opens the buffer and falls
 * back to a gather otherwise.  peer_device == ctx's own device is always OK. */
int pgt_peer_access(pgt_ctx *ctx, int peer_device);
int pgt_rowbuf_create(pgt_ctx *ctx, size_t bytes, void **dev_ptr, pgt_ipc_handle *handle);
int pgt_rowbuf_open(pgt_ctx *ctx, const pgt_ipc_handle *handle, void **dev_ptr);
int pgt_rowbuf_close(pgt_ctx *ctx, void *dev_ptr, int owner);
int pgt_rowbuf_read(pgt_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes, void *stream);
/* Self-test of a mapping before it is trusted: a kernel on ctx's device stores the 8-byte words
 * splitmix64(seed + i), i = 0 .. bytes/8 - 1, to dev_ptr with the same plain global stores the query kernels
 * use for their rows (asynchronous on `stream`; bytes a multiple of 8, dev_ptr 8-byte aligned).  Every rank
 * fills its slice of the shared buffer, the owner reads it back (pgt_rowbuf_read) and compares. */
int pgt_rowbuf_fill(pgt_ctx *ctx, void *dev_ptr, size_t bytes, uint64_t seed, void *stream);

/* ---- plain device buffers (for callers without a HIP binding of their own: the C++ hosts' multi-GPU path) --------
 * One process may hold one pgt_ctx per GPU (one thread each).  pgt_dev_copy moves bytes between buffers of two
 * contexts — the same device, or two devices (peer copy where the link allows it, staged by the runtime otherwise);
 * it is synchronous and may be called from the thread of either context. */
int pgt_dev_alloc(pgt_ctx *ctx, size_t bytes, void **dev_ptr);
/* free and total memory of ctx's device in bytes (the hosts decide with it whether a table must be reduced in passes) */
int pgt_dev_memory(pgt_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int pgt_dev_free(pgt_ctx *ctx, void *dev_ptr);
/* host memory -> a buffer of ctx's device (synchronous) */
int pgt_dev_upload(pgt_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int pgt_dev_copy(pgt_ctx *dst_ctx, void *dst, pgt_ctx *src_ctx, const void *src, size_t bytes);

/* ---- device-side text ingest ------------------------------------------------ */
/* Replaces the per-line text parse of the reference's streaming loops (fstWindow.cpp:123-146 `chr pos a b`,
 * hetWindow.cpp:121-144 `chr pos genotype`, dxyWindow.cpp:141-153,399-403 `chr pos major minor ref freq nInd`):
 * the raw text is copied to the GPU once and parsed there into the structure-of-arrays columns the
 * *_reduce_dev entry points take, together with the chromosome runs pgt_build_windows_* take.
 *   text, len   the lines to parse, in HOST memory (a header line, if any, already skipped by the caller)
 *   tokens      what the whitespace-separated tokens of a line are, in order; tokens[0] must be PGT_TOK_CHR or PGT_TOK_CHR_PREFIX
 *               (2 <= n_tokens <= 12); tokens beyond n_tokens are ignored, as the tools ignore extra columns
 * Semantics of the tools' own loops are kept: a blank line ends the data (fstWindow.cpp:125); a last line
 * without newline is accepted; \r counts as blank.  Numbers the kernel cannot convert exactly in one f64
 * operation (more than 15 significant digits, |power of ten| > 22, inf, nan, odd signs) are converted on the
 * host with the correctly rounded library routine, so every value has the bits `ss >> double` gives.
 * pgt_ingest_bad_line: 0-based index of the first line (before the end of data) that cannot be parsed, or -1;
 * rows then holds the number of good lines before it, and the caller reports the error as the tools would.
 * PGT_EDOMAIN: a line longer than 64 KiB, or more than 2^20 chromosome runs or irregular lines — such an
 * input is not one of the tools' tables; parse it on the host.
 * Column k (pgt_ingest_column) belongs to token k: u32 for PGT_TOK_U32, f64 for _F64 / _FREQ, i8 for _I8,
 * i32 for _I32, NULL for _CHR / _SKIP; DEVICE pointers, rows elements, owned by the ingest object.
 * pgt_ingest_runs: run lengths, and for every run the byte offset and length of its name inside `text`. */
enum {
    PGT_TOK_CHR = 0,  /* chromosome name: delimits the runs, not stored */
    PGT_TOK_SKIP = 1, /* ignored (MAF major / minor / ref) */
    PGT_TOK_U32 = 2,  /* position */
    PGT_TOK_F64 = 3,  /* any double */
    PGT_TOK_I8 = 4,   /* integer, clamped to int8 (hetWindow genotype: only >= 0 and == 1 are ever tested) */
    PGT_TOK_I32 = 5,  /* integer, clamped to int32 (MAF nInd) */
    PGT_TOK_FREQ = 6, /* double that must lie in [0,1] (MAF allele frequency), else the line is an error */
    PGT_TOK_CHR_PREFIX = 7 /* as PGT_TOK_CHR (allowed as tokens[0] only), the chromosome being the token UP TO ITS FIRST '_':
                            * selscan locus ids `chr_position` (extractChr, ihsWindow.cpp:80-92; xpehhWindow.cpp:82-94); a token
                            * without '_' is the name as a whole (ABI 5) */
};
typedef struct pgt_ingest pgt_ingest;
/* reductions over DEVICE columns (pgt_ingest_column) with the window table and the rows in HOST memory:
 * the host-buffer entry points minus the column upload; synchronous */
int pgt_fst_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_a, const double *d_b, uint64_t n,
                        const pgt_win *win, uint64_t n_win, pgt_fst_row *out, size_t out_bytes);
int pgt_het_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const int8_t *d_g, uint64_t n,
                        const pgt_win *win, uint64_t n_win, pgt_het_row *out, size_t out_bytes);
int pgt_dxy_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_p1, const double *d_p2,
                        const int32_t *d_n1, const int32_t *d_n2, uint64_t n, int minind,
                        const pgt_win *win, uint64_t n_win, pgt_dxy_row *out, size_t out_bytes, pgt_dxy_total *tot);
/* ihsWindow.cpp:123-221 / xpehhWindow.cpp:126-232 over device columns (the device-parsed *.norm table) */
int pgt_extreme_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_score, uint64_t n, int mode, double cutoff,
                            const pgt_win *win, uint64_t n_win, pgt_ext_row *out, size_t out_bytes);
/* device column of token `token` -> host (bytes <= rows * element size, else PGT_EARG) */
int pgt_ingest_download(pgt_ctx *ctx, const pgt_ingest *ing, int token, void *host_dst, size_t bytes);
int pgt_ingest_text(pgt_ctx *ctx, const char *text, size_t len, const uint8_t *tokens, int n_tokens, pgt_ingest **out);
/* The same, with room for `rows_in_front` more rows BEFORE the parsed ones in every column: a caller that parses the head
 * of a text itself (the shipped hosts do, beside HIP start-up) uploads its rows there (pgt_dev_upload to
 * pgt_ingest_column_base) and has one contiguous column without a copy.  pgt_ingest_column still points at the first
 * parsed row, pgt_ingest_rows counts the parsed rows only. */
int pgt_ingest_text_behind(pgt_ctx *ctx, const char *text, size_t len, const uint8_t *tokens, int n_tokens,
                           uint64_t rows_in_front, pgt_ingest **out);
void *pgt_ingest_column_base(const pgt_ingest *ing, int token);
uint64_t pgt_ingest_rows(const pgt_ingest *ing);
/* 1 when the data ended at a blank line of `text`, be it its last line (the tools stop reading there, fstWindow.cpp:125):
 * a caller that hands consecutive pieces of one file to several GPUs must drop the pieces behind such a one. */
int pgt_ingest_blank_before_end(const pgt_ingest *ing);
int64_t pgt_ingest_bad_line(const pgt_ingest *ing);
void *pgt_ingest_column(const pgt_ingest *ing, int token);
size_t pgt_ingest_runs(const pgt_ingest *ing, const uint64_t **run_len, const uint64_t **name_off, const uint32_t **name_len);
void pgt_ingest_free(pgt_ingest *ing);

/* ---- window tables built ON the device (the `-stepsize 1` regime) -------------------
 * With one window per site the table of pgt_build_windows_sites is as large as the columns (32 B per window)
 * and the host spends its time filling and uploading it.  pgt_wintab_sites builds the SAME table — same rules
 * (fstWindow.cpp:132-138,150-152 with calcWindow's carry :92-103), same bytes — in GPU memory from the
 * chromosome run lengths: the host plans every run in O(#runs), a kernel writes the windows.
 * pgt_wintab_first: n_runs + 1 values, the index of every run's first window (the last = the number of windows):
 * row i carries the name of the run r with first[r] <= i < first[r+1] (pgt_win.label_run of the host table).
 * pgt_wintab_device: the table itself, a DEVICE pointer usable as `win` of the *_dev calls.
 * *_reduce_tab: the host-buffer entry points (cols_on_device = 0: pos / a / b ... are host arrays, uploaded here)
 * or the *_reduce_cols ones (cols_on_device = 1) over such a table; rows come back to HOST memory (out_bytes =
 * capacity of `out`, at least pgt_wintab_size rows); synchronous. */
typedef struct pgt_wintab pgt_wintab;
int pgt_wintab_sites(pgt_ctx *ctx, const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S, pgt_wintab **out);
uint64_t pgt_wintab_size(const pgt_wintab *tab);
const uint64_t *pgt_wintab_first(const pgt_wintab *tab);
const pgt_win *pgt_wintab_device(const pgt_wintab *tab);
void pgt_wintab_free(pgt_wintab *tab);
int pgt_fst_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b, uint64_t n, int cols_on_device,
                       const pgt_wintab *tab, pgt_fst_row *out, size_t out_bytes);
int pgt_het_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n, int cols_on_device,
                       const pgt_wintab *tab, pgt_het_row *out, size_t out_bytes);
int pgt_dxy_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
                       const int32_t *n2, uint64_t n, int minind, int cols_on_device, const pgt_wintab *tab,
                       pgt_dxy_row *out, size_t out_bytes, pgt_dxy_total *tot);

#ifdef __cplusplus
}
#endif
#endif /* PGTWIN_H */
