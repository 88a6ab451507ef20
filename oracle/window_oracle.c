/*
 * window_oracle.c — CPU restatement of the reference's streaming window scan.
 * TEST INFRASTRUCTURE ONLY (see window_oracle.h for scope, citations and how it is pinned).
 *
 * Shape of the algorithm (the reference's, not the product's):
 *   - one W-entry buffer, filled site by site;
 *   - a window is emitted (i) when a site of a new chromosome arrives and the buffer is
 *     non-empty, (ii) when the buffer is full and another site arrives, (iii) at end of input
 *     if W-S < fill <= W;
 *   - every emission re-sums the whole buffer sequentially in double precision, then keeps
 *     the last W-S entries if (and only if) the buffer was full, else empties it.
 */
#define _POSIX_C_SOURCE 200809L
#include "window_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * The W-entry buffer.  Columns are kept side by side; which ones are live depends on the tool.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t W, S, fill;
    uint32_t *coord; /* printed coordinate of the entry (site position or bp slot) */
    int64_t *site;   /* global data-site index, -1 for a dxy placeholder slot */
    double *x;       /* fst: a      dxy: per-site value (>=0, -9 skipped, -7 placeholder) */
    double *y;       /* fst: b */
    int32_t *g;      /* het: genotype */
} wbuf;

typedef struct {
    orc_row *rows;
    size_t cap, count;
} sink;

static int wbuf_init(wbuf *wb, uint32_t W, uint32_t S) {
    memset(wb, 0, sizeof *wb);
    wb->W = W;
    wb->S = S;
    size_t m = W ? W : 1;
    wb->coord = calloc(m, sizeof *wb->coord);
    wb->site = calloc(m, sizeof *wb->site);
    wb->x = calloc(m, sizeof *wb->x);
    wb->y = calloc(m, sizeof *wb->y);
    wb->g = calloc(m, sizeof *wb->g);
    return (wb->coord && wb->site && wb->x && wb->y && wb->g) ? 0 : -1;
}

static void wbuf_free(wbuf *wb) {
    free(wb->coord);
    free(wb->site);
    free(wb->x);
    free(wb->y);
    free(wb->g);
}

/* fstWindow.cpp:91-105 / hetWindow.cpp:90-103 / dxyWindow.cpp:194-207: a full buffer keeps its
 * last W-S entries ("same chromosome"), anything else is emptied ("new chromosome"). */
static void wbuf_advance(wbuf *wb) {
    if (wb->fill == wb->W) {
        uint32_t keep = wb->W - wb->S;
        memmove(wb->coord, wb->coord + wb->S, keep * sizeof *wb->coord);
        memmove(wb->site, wb->site + wb->S, keep * sizeof *wb->site);
        memmove(wb->x, wb->x + wb->S, keep * sizeof *wb->x);
        memmove(wb->y, wb->y + wb->S, keep * sizeof *wb->y);
        memmove(wb->g, wb->g + wb->S, keep * sizeof *wb->g);
        wb->fill = keep;
    } else {
        wb->fill = 0;
    }
}

static void site_range(const wbuf *wb, uint64_t *lo, uint64_t *hi) {
    int64_t first = -1, last = -1;
    for (uint32_t i = 0; i < wb->fill; ++i)
        if (wb->site[i] >= 0) {
            if (first < 0) first = wb->site[i];
            last = wb->site[i];
        }
    *lo = first < 0 ? 0 : (uint64_t)first;
    *hi = first < 0 ? 0 : (uint64_t)last + 1;
}

static orc_row *sink_next(sink *sk) {
    static orc_row scratch;
    orc_row *r = sk->count < sk->cap ? &sk->rows[sk->count] : &scratch;
    sk->count++;
    memset(r, 0, sizeof *r);
    r->printed = 1;
    return r;
}

/* fstWindow.cpp:69-107 */
static void fst_emit(wbuf *wb, uint32_t label, sink *sk) {
    orc_row *r = sink_next(sk);
    uint32_t n = wb->fill;
    r->label = label;
    r->start = wb->coord[0];
    r->end = wb->coord[n - 1];
    r->mid = (uint32_t)(r->start + r->end) / 2u; /* unsigned wrap kept on purpose (:73) */
    double asum = 0.0, bsum = 0.0;
    for (uint32_t i = 0; i < n; ++i) { /* :80-83 sequential, re-done per window */
        asum += wb->x[i];
        bsum += wb->y[i];
    }
    r->num = asum;
    r->den = bsum;
    r->value = bsum != 0.0 ? asum / bsum : 0.0; /* :85 */
    r->n = n;
    site_range(wb, &r->lo, &r->hi);
    wbuf_advance(wb);
}

/* hetWindow.cpp:66-105 */
static void het_emit(wbuf *wb, uint32_t label, sink *sk) {
    orc_row *r = sink_next(sk);
    uint32_t n = wb->fill;
    r->label = label;
    r->start = wb->coord[0];
    r->end = wb->coord[n - 1];
    r->mid = (uint32_t)(r->start + r->end) / 2u;
    uint32_t nonmissing = 0, nhet = 0;
    for (uint32_t i = 0; i < n; ++i) /* :77-82 */
        if (wb->g[i] >= 0) {
            ++nonmissing;
            if (wb->g[i] == 1) ++nhet;
        }
    r->num = nhet;
    r->den = nonmissing;
    r->value = nonmissing != 0 ? (double)nhet / nonmissing : 0.0; /* :84 */
    r->n = nonmissing;                                             /* column 6 is nonmissing (:87) */
    site_range(wb, &r->lo, &r->hi);
    wbuf_advance(wb);
}

/* dxyWindow.cpp:172-209 */
static void dxy_emit(wbuf *wb, uint32_t label, int skip_missing, sink *sk) {
    orc_row *r = sink_next(sk);
    uint32_t n = wb->fill;
    double sum = 0.0;
    uint32_t neff = 0, nskip = 0;
    for (uint32_t i = 0; i < n; ++i) { /* :179-186 */
        if (wb->x[i] >= 0) {
            sum += wb->x[i];
            ++neff;
        } else if (wb->x[i] == -9) {
            ++nskip;
        }
    }
    r->label = label;
    r->start = wb->coord[0];
    r->end = wb->coord[n - 1];
    r->mid = 0;
    r->value = r->num = sum;
    r->n = neff;
    r->nskip = nskip;
    r->printed = (neff > 0 || !skip_missing) ? 1u : 0u; /* :189 */
    site_range(wb, &r->lo, &r->hi);
    wbuf_advance(wb);
}

static int check_ws(uint32_t W, uint32_t S) { return (W >= 1 && S >= 1 && S <= W) ? 0 : -1; }

int orc_fst_scan(const uint32_t *chr, const uint32_t *pos, const double *a, const double *b, size_t n,
                 uint32_t W, uint32_t S, orc_row *out, size_t cap, size_t *n_out) {
    if (check_ws(W, S) || !n_out) return ORC_EARG;
    wbuf wb;
    if (wbuf_init(&wb, W, S)) return ORC_EIO;
    sink sk = {out, cap, 0};
    uint32_t run = 0;
    for (size_t i = 0; i < n; ++i) { /* fstWindow.cpp:125-147 */
        int newchr = i > 0 && chr[i] != chr[i - 1];
        if (newchr && wb.fill > 0) fst_emit(&wb, run, &sk);            /* :132-134, label = old chr */
        else if (wb.fill == W) fst_emit(&wb, run + (uint32_t)newchr, &sk); /* :135-138, label = chr  */
        if (newchr) ++run;
        uint32_t k = wb.fill++; /* :141-143 */
        wb.coord[k] = pos[i];
        wb.site[k] = (int64_t)i;
        wb.x[k] = a[i];
        wb.y[k] = b[i];
    }
    if (wb.fill > W - S && wb.fill <= W) fst_emit(&wb, run, &sk); /* :150-152 */
    wbuf_free(&wb);
    *n_out = sk.count;
    return sk.count > cap ? ORC_ECAP : ORC_OK;
}

int orc_het_scan(const uint32_t *chr, const uint32_t *pos, const int32_t *g, size_t n, uint32_t W,
                 uint32_t S, orc_row *out, size_t cap, size_t *n_out) {
    if (check_ws(W, S) || !n_out) return ORC_EARG;
    wbuf wb;
    if (wbuf_init(&wb, W, S)) return ORC_EIO;
    sink sk = {out, cap, 0};
    uint32_t run = 0;
    for (size_t i = 0; i < n; ++i) { /* hetWindow.cpp:123-145 */
        int newchr = i > 0 && chr[i] != chr[i - 1];
        if (newchr && wb.fill > 0) het_emit(&wb, run, &sk);
        else if (wb.fill == W) het_emit(&wb, run + (uint32_t)newchr, &sk);
        if (newchr) ++run;
        uint32_t k = wb.fill++;
        wb.coord[k] = pos[i];
        wb.site[k] = (int64_t)i;
        wb.g[k] = g[i];
    }
    if (wb.fill > W - S && wb.fill <= W) het_emit(&wb, run, &sk); /* :148-150 */
    wbuf_free(&wb);
    *n_out = sk.count;
    return sk.count > cap ? ORC_ECAP : ORC_OK;
}

static void dxy_push(wbuf *wb, uint32_t coord, double v, int64_t site) {
    uint32_t k = wb->fill++;
    wb->coord[k] = coord;
    wb->x[k] = v;
    wb->site[k] = site;
}

int orc_dxy_scan(const uint32_t *chr, const uint32_t *pos, const double *p1, const double *p2,
                 const int32_t *n1, const int32_t *n2, size_t n, uint32_t W, uint32_t S, int minind,
                 int fixedsite, int skip_missing, const uint32_t *run_chr_len, size_t n_runs,
                 orc_row *out, size_t cap, size_t *n_out, orc_dxy_total *tot) {
    if (!n_out || !tot) return ORC_EARG;
    if (W > 0 && check_ws(W, S)) return ORC_EARG; /* dxyWindow.cpp:128-131 + Q9 */
    if (W == 0 && !fixedsite) return ORC_EDOMAIN; /* Q10: the reference indexes an empty vector */
    if (!fixedsite && !run_chr_len) return ORC_EARG;
    /* no site at all: the reference reads a first site of both files unconditionally (:285-292) — undefined on an empty
     * file.  With n_runs >= 1 the caller says "both files had data lines but the synchronisation of :315-331 matched none
     * of them" (orc_dxy_text): the main loop then breaks at its first pass and only the code behind it runs — the padding of
     * the first line's chromosome (:407-421), the last flush (:424) and the all-zero genome-wide line (:429-433). */
    if (n == 0 && n_runs == 0) return ORC_EDOMAIN;
    wbuf wb;
    if (wbuf_init(&wb, W, S)) return ORC_EIO;
    sink sk = {out, cap, 0};
    uint32_t run = 0, positer = 1;
    double gsum = 0.0;
    uint32_t gneff = 0, gskip = 0;

    for (size_t i = 0; i < n; ++i) { /* dxyWindow.cpp:313-404 */
        int newchr = i > 0 && chr[i] != chr[i - 1];
        if (W > 0 && newchr) { /* :334-361 */
            if (!fixedsite) {
                if (run >= n_runs) { wbuf_free(&wb); return ORC_EARG; }
                uint32_t lastpos = run_chr_len[run];
                while (positer <= lastpos) { /* :345-352 pad the old chromosome to its length */
                    if (wb.fill == W) dxy_emit(&wb, run, skip_missing, &sk);
                    dxy_push(&wb, positer, -7.0, -1);
                    ++positer;
                }
                if (wb.fill > W - S) dxy_emit(&wb, run, skip_missing, &sk); /* :353-355 */
            } else if (wb.fill > 0) {
                dxy_emit(&wb, run, skip_missing, &sk); /* :358 */
            }
            positer = 1; /* :360 */
        }
        if (newchr) ++run;
        if (W > 0 && !fixedsite) { /* :363-373 placeholders up to the data site */
            while (positer < pos[i]) {
                if (wb.fill == W) dxy_emit(&wb, run, skip_missing, &sk);
                dxy_push(&wb, positer, -7.0, -1);
                ++positer;
            }
        }
        if (W > 0 && wb.fill == W) dxy_emit(&wb, run, skip_missing, &sk); /* :376-378 */

        double d = (n1[i] >= minind && n2[i] >= minind)
                       ? p1[i] * (1.0 - p2[i]) + p2[i] * (1.0 - p1[i])
                       : -9.0; /* :381 */
        if (d != -9.0) {       /* :382-385 */
            gsum += d;
            ++gneff;
        } else {
            ++gskip;
        }
        if (W > 0) { /* :388-394 */
            dxy_push(&wb, pos[i], d, (int64_t)i);
            ++positer;
        }
    }
    if (W > 0 && !fixedsite) { /* :407-423 */
        if (run >= n_runs) { wbuf_free(&wb); return ORC_EARG; }
        uint32_t lastpos = run_chr_len[run];
        while (positer <= lastpos) {
            if (wb.fill == W) dxy_emit(&wb, run, skip_missing, &sk);
            dxy_push(&wb, positer, -7.0, -1);
            ++positer;
        }
    }
    if (W > 0 && wb.fill > W - S && wb.fill <= W) dxy_emit(&wb, run, skip_missing, &sk); /* :424-426 */
    tot->sum = gsum;
    tot->neff = gneff;
    tot->nskip = gskip;
    wbuf_free(&wb);
    *n_out = sk.count;
    return sk.count > cap ? ORC_ECAP : ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * betaAFOutlier.R:400-418, WCFst(): operation order as R evaluates it (left to right, ^2 == x*x)
 * ---------------------------------------------------------------------------------------- */
void orc_wcfst_site(double f1, double f2, double n1, double n2, double *a_out, double *ab_out) {
    const double npool = n1 + n2;                                      /* :406 */
    const double fpool = n1 / npool * f1 + n2 / npool * f2;            /* :407 */
    const double alpha1 = 2 * f1 * (1 - f1);                           /* :408 */
    const double alpha2 = 2 * f2 * (1 - f2);                           /* :409 */
    const double b = (n1 * alpha1 + n2 * alpha2) / (npool - 1);        /* :410 */
    const double d1 = f1 - fpool, d2 = f2 - fpool;
    const double a = (4 * n1 * (d1 * d1) + 4 * n2 * (d2 * d2) - b) / (4 * n1 * n2 / npool); /* :411 */
    *a_out = a;
    *ab_out = b + a;                                                   /* :416 varcomp[,2]+varcomp[,1] */
}

void orc_wcfst_columns(const double *f1, const double *f2, size_t n, double n1, double n2, double *a,
                       double *ab) {
    for (size_t i = 0; i < n; ++i) orc_wcfst_site(f1[i], f2[i], n1, n2, &a[i], &ab[i]);
}

/* ------------------------------------------------------------------------------------------
 * Text front ends
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t n, cap;
    uint32_t *chr, *pos;
    double *x, *y;
    int32_t *g, *k;
    size_t n_runs, runs_cap;
    char **run_name;
} table;

static void table_free(table *t) {
    free(t->chr); free(t->pos); free(t->x); free(t->y); free(t->g); free(t->k);
    for (size_t i = 0; i < t->n_runs; ++i) free(t->run_name[i]);
    free(t->run_name);
    memset(t, 0, sizeof *t);
}

static int table_grow(table *t) {
    if (t->n < t->cap) return 0;
    size_t c = t->cap ? t->cap * 2 : 4096;
    uint32_t *chr = realloc(t->chr, c * sizeof *chr); if (!chr) return -1; t->chr = chr;
    uint32_t *pos = realloc(t->pos, c * sizeof *pos); if (!pos) return -1; t->pos = pos;
    double *x = realloc(t->x, c * sizeof *x); if (!x) return -1; t->x = x;
    double *y = realloc(t->y, c * sizeof *y); if (!y) return -1; t->y = y;
    int32_t *g = realloc(t->g, c * sizeof *g); if (!g) return -1; t->g = g;
    int32_t *k = realloc(t->k, c * sizeof *k); if (!k) return -1; t->k = k;
    t->cap = c;
    return 0;
}

static int table_run(table *t, const char *name, size_t len) {
    if (t->n_runs && strlen(t->run_name[t->n_runs - 1]) == len &&
        memcmp(t->run_name[t->n_runs - 1], name, len) == 0)
        return 0;
    if (t->n_runs == t->runs_cap) {
        size_t c = t->runs_cap ? t->runs_cap * 2 : 64;
        char **r = realloc(t->run_name, c * sizeof *r);
        if (!r) return -1;
        t->run_name = r;
        t->runs_cap = c;
    }
    char *s = malloc(len + 1);
    if (!s) return -1;
    memcpy(s, name, len);
    s[len] = 0;
    t->run_name[t->n_runs++] = s;
    return 0;
}

static const char *skip_ws(const char *p) {
    while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
    return p;
}
static const char *skip_tok(const char *p) {
    while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
    return p;
}

/* kind: 0 = "chr pos a b" (fst), 1 = "chr pos g" (het), 2 = ANGSD mafs with header
 * "chr pos major minor ref freq nind" (dxyWindow.cpp:141-153).  Reading stops at the first
 * empty line like the reference loop condition (fstWindow.cpp:125), or at EOF. */
static int table_read(const char *path, int kind, table *t) {
    memset(t, 0, sizeof *t);
    FILE *f = fopen(path, "r");
    if (!f) return ORC_EIO;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t len;
    int rc = ORC_OK;
    if (kind == 2 && getline(&line, &lcap, f) < 0) { /* header (dxyWindow.cpp:284) */
        rc = ORC_EDOMAIN;
        goto done;
    }
    while ((len = getline(&line, &lcap, f)) >= 0) {
        const char *p = skip_ws(line);
        if (*p == '\n' || *p == 0) break;
        const char *e = skip_tok(p);
        if (table_grow(t) || table_run(t, p, (size_t)(e - p))) { rc = ORC_EIO; goto done; }
        size_t i = t->n;
        t->chr[i] = (uint32_t)(t->n_runs - 1);
        char *q;
        t->pos[i] = (uint32_t)strtoul(e, &q, 10);
        if (kind == 0) {
            t->x[i] = strtod(q, &q);
            t->y[i] = strtod(q, &q);
        } else if (kind == 1) {
            t->g[i] = (int32_t)strtol(q, &q, 10);
        } else {
            const char *c = q;
            for (int skip = 0; skip < 3; ++skip) c = skip_tok(skip_ws(c)); /* major minor ref */
            t->x[i] = strtod(c, &q);
            t->k[i] = (int32_t)strtol(q, &q, 10);
        }
        t->n++;
    }
done:
    free(line);
    fclose(f);
    if (rc != ORC_OK) table_free(t);
    return rc;
}

static int rows_alloc(size_t n, uint32_t S, size_t extra, orc_row **rows, size_t *cap) {
    *cap = n / (S ? S : 1) + extra + 8;
    *rows = malloc(*cap * sizeof **rows);
    return *rows ? 0 : -1;
}

int orc_fst_text(const char *path, uint32_t W, uint32_t S, FILE *out) {
    if (check_ws(W, S)) return ORC_EARG;
    table t;
    int rc = table_read(path, 0, &t);
    if (rc) return rc;
    orc_row *rows;
    size_t cap, nr = 0;
    if (rows_alloc(t.n, S, t.n_runs, &rows, &cap)) { table_free(&t); return ORC_EIO; }
    rc = orc_fst_scan(t.chr, t.pos, t.x, t.y, t.n, W, S, rows, cap, &nr);
    if (rc == ORC_OK)
        for (size_t i = 0; i < nr; ++i) /* fstWindow.cpp:88, default ostream precision == %g */
            fprintf(out, "%s\t%u\t%u\t%u\t%g\t%u\n", t.run_name[rows[i].label], rows[i].start,
                    rows[i].end, rows[i].mid, rows[i].value, rows[i].n);
    free(rows);
    table_free(&t);
    return rc;
}

int orc_het_text(const char *path, uint32_t W, uint32_t S, FILE *out) {
    if (check_ws(W, S)) return ORC_EARG;
    table t;
    int rc = table_read(path, 1, &t);
    if (rc) return rc;
    orc_row *rows;
    size_t cap, nr = 0;
    if (rows_alloc(t.n, S, t.n_runs, &rows, &cap)) { table_free(&t); return ORC_EIO; }
    rc = orc_het_scan(t.chr, t.pos, t.g, t.n, W, S, rows, cap, &nr);
    if (rc == ORC_OK)
        for (size_t i = 0; i < nr; ++i) /* hetWindow.cpp:87 */
            fprintf(out, "%s\t%u\t%u\t%u\t%g\t%u\n", t.run_name[rows[i].label], rows[i].start,
                    rows[i].end, rows[i].mid, rows[i].value, rows[i].n);
    free(rows);
    table_free(&t);
    return rc;
}

int orc_dxy_text(const char *maf1, const char *maf2, const char *sizefile, uint32_t W, uint32_t S,
                 int minind, int fixedsite, int skip_missing, FILE *out, FILE *err) {
    table t1, t2;
    int rc = table_read(maf1, 2, &t1);
    if (rc) return rc;
    rc = table_read(maf2, 2, &t2);
    if (rc) { table_free(&t1); return rc; }
    uint32_t *run_len = NULL;
    orc_row *rows = NULL;
    table m; /* the synchronised sites: chr / pos / name from pop1 (dxyWindow.cpp:333,389), x = p1, y = p2, k / g = nInd 1 / 2 */
    memset(&m, 0, sizeof m);
    /* The two-file synchronisation of dxyWindow.cpp:315-331, restated literally on line indices i (pop1) and
     * j (pop2): equal (position, chromosome NAME) -> a matched site; otherwise one file is advanced — pop1
     * until its POSITION equals pop2's (names are not compared there, :319), or pop2 while its position is
     * smaller (:326) — and the run ENDS (break out of the main loop, :323,:330) when that fails.  A failed
     * getline leaves the last parsed site in place (:320,:327).  Identical and nested site sets give the
     * intersection; anything else gives whatever these rules give, truncated output included. */
    if (t1.n && t2.n) {
        if (strcmp(t1.run_name[t1.chr[0]], t2.run_name[t2.chr[0]])) { rc = ORC_EDOMAIN; goto done; } /* :294-298 exit 255 */
        size_t i = 0, j = 0;
        const char *cur = t1.run_name[t1.chr[0]]; /* `chr`, :293 */
        for (;;) {
            const char *c1 = t1.run_name[t1.chr[i]], *c2 = t2.run_name[t2.chr[j]];
            const int same = !strcmp(c1, c2);
            if (t1.pos[i] != t2.pos[j] || !same) {
                if ((same && t1.pos[i] < t2.pos[j]) || (!same && strcmp(c2, cur))) {
                    while (t1.pos[i] != t2.pos[j]) {
                        if (i + 1 >= t1.n) break;
                        ++i;
                    }
                    if (t1.pos[i] != t2.pos[j]) break;
                } else {
                    while (t2.pos[j] < t1.pos[i]) {
                        if (j + 1 >= t2.n) break;
                        ++j;
                    }
                    if (t1.pos[i] != t2.pos[j]) break;
                }
            }
            cur = t1.run_name[t1.chr[i]]; /* :333 */
            if (table_grow(&m) || table_run(&m, cur, strlen(cur))) { rc = ORC_EIO; goto done; }
            m.chr[m.n] = (uint32_t)(m.n_runs - 1);
            m.pos[m.n] = t1.pos[i];
            m.x[m.n] = t1.x[i];
            m.y[m.n] = t2.x[j];
            m.k[m.n] = t1.k[i];
            m.g[m.n] = t2.k[j];
            m.n++;
            if (i + 1 >= t1.n) break; /* :399 */
            ++i;
            if (j + 1 >= t2.n) break; /* :402 */
            ++j;
        }
        /* nothing matched (the first pass of the loop broke at :323 / :330): `chr` still names the first line's chromosome
         * (:293) and the code behind the loop pads it — one run without sites (found by tests/dxy_stream_model.py, round 4:
         * this restatement used to refuse such a pair) */
        if (m.n == 0 && table_run(&m, t1.run_name[t1.chr[0]], strlen(t1.run_name[t1.chr[0]]))) { rc = ORC_EIO; goto done; }
    }

    uint64_t slots = 0;
    if (!fixedsite) { /* dxyWindow.cpp:155-170 + :338-343 */
        if (!sizefile) { rc = ORC_EARG; goto done; }
        run_len = calloc(m.n_runs ? m.n_runs : 1, sizeof *run_len);
        FILE *sf = fopen(sizefile, "r");
        if (!sf || !run_len) { if (sf) fclose(sf); rc = ORC_EIO; goto done; }
        char name[4096];
        unsigned len;
        while (fscanf(sf, "%4095s %u", name, &len) == 2)
            for (size_t r = 0; r < m.n_runs; ++r)
                if (!run_len[r] && !strcmp(name, m.run_name[r])) run_len[r] = len; /* map::insert keeps the first */
        fclose(sf);
        for (size_t r = 0; r < m.n_runs; ++r) {
            if (!run_len[r]) { rc = ORC_EDOMAIN; goto done; }
            slots += run_len[r];
        }
    }
    size_t cap, nr = 0;
    if (rows_alloc((size_t)(fixedsite ? m.n : slots + m.n), S, 2 * m.n_runs, &rows, &cap)) { rc = ORC_EIO; goto done; }
    orc_dxy_total tot;
    rc = orc_dxy_scan(m.chr, m.pos, m.x, m.y, m.k, m.g, m.n, W, S, minind, fixedsite,
                      skip_missing, run_len, m.n_runs, rows, cap, &nr, &tot);
    if (rc == ORC_OK) {
        for (size_t i = 0; i < nr; ++i)
            if (rows[i].printed) /* dxyWindow.cpp:190 */
                fprintf(out, "%s\t%u\t%u\t%g\t%u\t%u\n", m.run_name[rows[i].label], rows[i].start,
                        rows[i].end, rows[i].value, rows[i].n, rows[i].nskip);
        fprintf(W == 0 ? out : err, "%g\t%u\t%u\n", tot.sum, tot.neff, tot.nskip); /* :429-433 */
    }
done:
    free(rows);
    free(run_len);
    table_free(&m);
    table_free(&t1);
    table_free(&t2);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Fast writers for synthetic text inputs (values are multiples of 1e-6, as ANGSD text is)
 * ---------------------------------------------------------------------------------------- */
static char *put_u32(char *p, uint32_t v) {
    char tmp[10];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

static char *put_fixed6(char *p, double v) {
    double s = v * 1e6;
    long long k = llround(s);
    if (fabs(s - (double)k) > 1e-3 || fabs(v) > 4e3) return p + sprintf(p, "%.17g", v);
    if (signbit(v)) { *p++ = '-'; k = -k; } /* also "-0.000000", as printf("%.6f") prints it */
    p = put_u32(p, (uint32_t)(k / 1000000));
    uint32_t frac = (uint32_t)(k % 1000000);
    *p++ = '.';
    for (int d = 100000; d; d /= 10) { *p++ = (char)('0' + frac / d); frac %= d; }
    return p;
}

int orc_write_fst_text(const char *path, const uint32_t *chr, const uint32_t *pos, const double *a,
                       const double *b, size_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return ORC_EIO;
    static char buf[1 << 16];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    char line[160];
    for (size_t i = 0; i < n; ++i) {
        char *p = line;
        memcpy(p, "chr", 3); p += 3;
        p = put_u32(p, chr[i] + 1); *p++ = '\t';
        p = put_u32(p, pos[i]); *p++ = '\t';
        p = put_fixed6(p, a[i]); *p++ = '\t';
        p = put_fixed6(p, b[i]); *p++ = '\n';
        fwrite(line, 1, (size_t)(p - line), f);
    }
    return fclose(f) ? ORC_EIO : ORC_OK;
}

int orc_write_het_text(const char *path, const uint32_t *chr, const uint32_t *pos, const int32_t *g,
                       size_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return ORC_EIO;
    static char buf[1 << 16];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    char line[96];
    for (size_t i = 0; i < n; ++i) {
        char *p = line;
        memcpy(p, "chr", 3); p += 3;
        p = put_u32(p, chr[i] + 1); *p++ = '\t';
        p = put_u32(p, pos[i]); *p++ = '\t';
        if (g[i] < 0) { *p++ = '-'; p = put_u32(p, (uint32_t)(-g[i])); }
        else p = put_u32(p, (uint32_t)g[i]);
        *p++ = '\n';
        fwrite(line, 1, (size_t)(p - line), f);
    }
    return fclose(f) ? ORC_EIO : ORC_OK;
}

/* ANGSD .mafs text: header + `chr pos major minor ref freq nind` (the seven columns Mafsite reads,
 * dxyWindow.cpp:24-32; only chr, pos, freq, nind matter). */
int orc_write_maf_text(const char *path, const uint32_t *chr, const uint32_t *pos, const double *freq,
                       const int32_t *nind, size_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return ORC_EIO;
    static char buf[1 << 16];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    fputs("chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd\n", f);
    char line[160];
    for (size_t i = 0; i < n; ++i) {
        char *p = line;
        memcpy(p, "chr", 3); p += 3;
        p = put_u32(p, chr[i] + 1); *p++ = '\t';
        p = put_u32(p, pos[i]);
        memcpy(p, "\tA\tC\tA\t", 7); p += 7;
        p = put_fixed6(p, freq[i]); *p++ = '\t';
        p = put_u32(p, (uint32_t)nind[i]); *p++ = '\n';
        fwrite(line, 1, (size_t)(p - line), f);
    }
    return fclose(f) ? ORC_EIO : ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Path-based wrappers (ctypes cannot hand over a FILE*)
 * ---------------------------------------------------------------------------------------- */
int orc_fst_text_path(const char *in, uint32_t W, uint32_t S, const char *out_path) {
    FILE *o = fopen(out_path, "w");
    if (!o) return ORC_EIO;
    int rc = orc_fst_text(in, W, S, o);
    return fclose(o) ? ORC_EIO : rc;
}

int orc_het_text_path(const char *in, uint32_t W, uint32_t S, const char *out_path) {
    FILE *o = fopen(out_path, "w");
    if (!o) return ORC_EIO;
    int rc = orc_het_text(in, W, S, o);
    return fclose(o) ? ORC_EIO : rc;
}

int orc_dxy_text_path(const char *maf1, const char *maf2, const char *sizefile, uint32_t W, uint32_t S,
                      int minind, int fixedsite, int skip_missing, const char *out_path,
                      const char *err_path) {
    FILE *o = fopen(out_path, "w");
    FILE *e = fopen(err_path, "w");
    if (!o || !e) { if (o) fclose(o); if (e) fclose(e); return ORC_EIO; }
    int rc = orc_dxy_text(maf1, maf2, (sizefile && *sizefile) ? sizefile : NULL, W, S, minind,
                          fixedsite, skip_missing, o, e);
    int c1 = fclose(o), c2 = fclose(e);
    return (c1 || c2) ? ORC_EIO : rc;
}

/* ------------------------------------------------------------------------------------------
 * ihsWindow.cpp:123-221 / xpehhWindow.cpp:126-232
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_ext_row *rows;
    size_t cap, count;
} ext_sink;

static void ext_print(ext_sink *sk, uint32_t label, uint32_t ws, uint32_t we, uint32_t nbig, uint32_t nsites,
                      double value, uint32_t position, uint64_t lo, uint64_t hi) {
    if (sk->count < sk->cap) {
        orc_ext_row *r = &sk->rows[sk->count];
        memset(r, 0, sizeof *r);
        r->label = label; r->start = ws; r->end = we; r->nsites = nsites; r->nbig = nsites ? nbig : 0;
        r->value = nsites ? value : 0.0; r->position = nsites ? position : 0;
        r->lo = lo; r->hi = hi;
    }
    sk->count++;
}

int orc_extreme_scan(const uint32_t *chr, const uint32_t *pos, const double *score, size_t n, uint32_t W, int mode,
                     double cutoff, const uint32_t *run_chr_len, size_t n_runs, orc_ext_row *out, size_t cap,
                     size_t *n_out) {
    if (!n_out || W < 1 || mode < ORC_EXT_IHS || mode > ORC_EXT_XP_MIN) return ORC_EARG;
    if (n == 0) return ORC_EDOMAIN; /* the reference prints one window with an empty name */
    ext_sink sk = {out, cap, 0};
    uint32_t run = 0, ws = 1, we = ws + (W - 1), nsites = 0, nbig = 0, best_pos = 0;
    double best_key = 0, best_val = 0;
    uint32_t chrlen = (run_chr_len && n_runs) ? run_chr_len[0] : 0; /* ihsWindow.cpp:154-157: first window NOT clamped */
    uint64_t lo = 0;
    for (size_t i = 0; i < n; ++i) {
        const int newchr = i > 0 && chr[i] != chr[i - 1];
        {   /* a position beyond its chromosome's given length makes the reference loop forever (:184) */
            const uint32_t own_len = newchr ? ((run_chr_len && run + 1 < n_runs) ? run_chr_len[run + 1] : 0) : chrlen;
            if (own_len && pos[i] > own_len) return ORC_EDOMAIN;
        }
        if (newchr) { /* :160-175 */
            ext_print(&sk, run, ws, we, nbig, nsites, best_val, best_pos, lo, i);
            while (we < chrlen) {
                ws = we + 1; we = ws + (W - 1);
                if (chrlen && we > chrlen) we = chrlen;
                ext_print(&sk, run, ws, we, 0, 0, 0, 0, i, i);
            }
            ++run;
            nsites = 0; lo = i;
            chrlen = (run_chr_len && run < n_runs) ? run_chr_len[run] : 0;
            ws = 1; we = ws + (W - 1);
            if (chrlen && we > chrlen) we = chrlen;
        } else if (pos[i] >= we) { /* :176-190 */
            ext_print(&sk, run, ws, we, nbig, nsites, best_val, best_pos, lo, i);
            ws = we + 1; we = ws + (W - 1);
            if (chrlen && we > chrlen) we = chrlen;
            nsites = 0; lo = i;
            while (pos[i] > we) {
                ext_print(&sk, run, ws, we, 0, 0, 0, 0, i, i);
                ws = we + 1; we = ws + (W - 1);
                if (chrlen && we > chrlen) we = chrlen;
            }
        }
        const double s = score[i];
        const double key = mode == ORC_EXT_IHS ? fabs(s) : (mode == ORC_EXT_XP_MAX ? s : -s);
        const double thr = mode == ORC_EXT_XP_MIN ? -cutoff : cutoff;
        if (nsites == 0) { nbig = 0; best_key = key; best_val = s; best_pos = pos[i]; } /* :194-197 */
        if (key > best_key) { best_key = key; best_val = s; best_pos = pos[i]; }         /* :199-201 */
        if (key > thr) ++nbig;                                                             /* :203-205 */
        ++nsites;
    }
    ext_print(&sk, run, ws, we, nbig, nsites, best_val, best_pos, lo, n); /* :212 */
    while (we < chrlen) { /* :213-218 */
        ws = we + 1; we = ws + (W - 1);
        if (we > chrlen) we = chrlen;
        ext_print(&sk, run, ws, we, 0, 0, 0, 0, n, n);
    }
    *n_out = sk.count;
    return sk.count > cap ? ORC_ECAP : ORC_OK;
}

/* selscan *.norm text: "<chr>_<...> pos f0 f1 ..." ; score = field `score_field` after pos
 * (ihs: 4 -> sitevec[4], xpehh: 6 -> sitevec[6]); the chromosome is the locus id up to the first '_'
 * (ihsWindow.cpp:82-93). */
static int ext_text(const char *in, int skip_header, int score_field, uint32_t W, int mode, double cutoff,
                    const char *chrlen_path, const char *out_path) {
    FILE *f = fopen(in, "r");
    if (!f) return ORC_EIO;
    table t;
    memset(&t, 0, sizeof t);
    char *line = NULL;
    size_t lcap = 0;
    int rc = ORC_OK;
    if (skip_header && getline(&line, &lcap, f) < 0) { fclose(f); free(line); return ORC_EDOMAIN; }
    while (getline(&line, &lcap, f) >= 0) {
        const char *p = skip_ws(line);
        if (*p == '\n' || *p == 0) continue;
        const char *e = skip_tok(p);
        const char *us = memchr(p, '_', (size_t)(e - p));
        if (table_grow(&t) || table_run(&t, p, (size_t)((us ? us : e) - p))) { rc = ORC_EIO; break; }
        char *q;
        t.chr[t.n] = (uint32_t)(t.n_runs - 1);
        t.pos[t.n] = (uint32_t)strtoul(e, &q, 10);
        double v = 0;
        for (int k = 0; k <= score_field; ++k) v = strtod(q, &q);
        t.x[t.n] = v;
        t.n++;
    }
    free(line);
    fclose(f);
    uint32_t *len = NULL;
    orc_ext_row *rows = NULL;
    if (rc == ORC_OK && chrlen_path && *chrlen_path) {
        len = calloc(t.n_runs ? t.n_runs : 1, sizeof *len);
        FILE *cf = fopen(chrlen_path, "r");
        if (!cf || !len) { if (cf) fclose(cf); rc = ORC_EIO; }
        else {
            char name[4096];
            unsigned L;
            /* std::map::insert keeps the first entry of a name; a run's length is looked up by name */
            while (fscanf(cf, "%4095s %u", name, &L) == 2)
                for (size_t r = 0; r < t.n_runs; ++r)
                    if (!strcmp(name, t.run_name[r]) && !len[r]) len[r] = L;
            fclose(cf);
        }
    }
    if (rc == ORC_OK) {
        size_t cap = 16, nr = 0;
        for (;;) {
            free(rows);
            rows = malloc(cap * sizeof *rows);
            rc = orc_extreme_scan(t.chr, t.pos, t.x, t.n, W, mode, cutoff, len, len ? t.n_runs : 0, rows, cap, &nr);
            if (rc != ORC_ECAP) break;
            cap = nr;
        }
        if (rc == ORC_OK) {
            FILE *o = fopen(out_path, "w");
            if (!o) rc = ORC_EIO;
            else {
                for (size_t i = 0; i < nr; ++i) {
                    fprintf(o, "%s\t%u\t%u\t", t.run_name[rows[i].label], rows[i].start, rows[i].end);
                    if (rows[i].nsites) /* ihsWindow.cpp:104-107 */
                        fprintf(o, "%g\t%u\t%g\t%u\n", rows[i].value, rows[i].position,
                                (double)(int)rows[i].nbig / rows[i].nsites, rows[i].nsites);
                    else
                        fprintf(o, "NA\tNA\tNA\t0\n");
                }
                if (fclose(o)) rc = ORC_EIO;
            }
        }
    }
    free(rows);
    free(len);
    table_free(&t);
    return rc;
}

int orc_ihs_text_path(const char *in, uint32_t W, double cutoff, const char *chrlen_path, const char *out_path) {
    return ext_text(in, 0, 4, W, ORC_EXT_IHS, cutoff, chrlen_path, out_path);
}

int orc_xpehh_text_path(const char *in, double cutoff, uint32_t W, const char *chrlen_path, const char *out_path) {
    return ext_text(in, 1, 6, W, cutoff < 0 ? ORC_EXT_XP_MIN : ORC_EXT_XP_MAX, cutoff, chrlen_path, out_path);
}
