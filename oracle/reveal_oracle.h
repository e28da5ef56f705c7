/*
 * reveal_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the reveallib hot path of jasperlinthorst/reveal
 * (SA -> SAi -> LCP -> SO construction, pairwise / multi MUM scans, D-label,
 * split, bubble_sort and the aligner work loop).  Every function cites the
 * reference file:line it follows.  Nothing in the shipped product
 * (reveal_amd/) may include, link, dlopen or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity is PINNED: oracle/pin_oracle.py checks every function below against
 * oracle/_ref/libreveal_ref*.so, which is the reference's own C
 * (reveallib/reveal.c, reveallib/interface.c, the four divsufsort .c files) compiled
 * unmodified from /root/reference by oracle/Makefile, and against the
 * known-answer vectors of SURVEY.md 8(c); tests/golden/ holds vectors
 * produced by that reference build (oracle/gen_golden.py).
 *
 * Build variants (reveallib/reveal.h:7-13):
 *   default      saidx_t=int32_t  lcp_t=int32_t   -> liboracle.so
 *   -DRO_SA64    saidx_t=int64_t  lcp_t=uint32_t  -> liboracle64.so
 */
#ifndef REVEAL_ORACLE_H
#define REVEAL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef RO_SA64
typedef int64_t  ro_saidx_t;
typedef uint32_t ro_lcp_t;
#else
typedef int32_t  ro_saidx_t;
typedef int32_t  ro_lcp_t;
#endif

/* ---- construct pieces (reveallib/interface.c:160-291) ------------------ */

/* Suffix array of T[0..n) by unsigned byte order, shorter suffix first.
 * The reference calls divsufsort (interface.c:215-222); a suffix array is
 * unique, so any correct construction is bit-identical.  If
 * ro_set_divsufsort() was given the reference's own divsufsort entry point
 * it is used, otherwise a prefix-doubling sorter (own code). */
int  ro_suffix_array(const uint8_t *T, ro_saidx_t *SA, ro_saidx_t n);
void ro_set_divsufsort(int (*fn)(const uint8_t *, ro_saidx_t *, ro_saidx_t));
/* own sorter only, regardless of ro_set_divsufsort (for cross-checks) */
int  ro_suffix_array_own(const uint8_t *T, ro_saidx_t *SA, ro_saidx_t n);
int  ro_sufcheck(const uint8_t *T, const ro_saidx_t *SA, ro_saidx_t n);

void ro_inverse(const ro_saidx_t *SA, ro_saidx_t *SAi, ro_saidx_t n);           /* interface.c:236-238 */
void ro_compute_lcp(const char *T, const ro_saidx_t *SA, const ro_saidx_t *SAi,
                    ro_lcp_t *LCP, ro_saidx_t n);                                /* interface.c:97-114 */
void ro_build_so(uint16_t *SO, const ro_saidx_t *nsep, int nsamples, ro_saidx_t n); /* interface.c:116-134 */
void ro_revcomp(char *T, ro_saidx_t n);                                          /* interface.c:136-158 */

/* ---- scans ------------------------------------------------------------- */

/* View of a (sub)index, fields as RevealIndex (reveallib/reveal.h:17-40). */
typedef struct {
    char             *T;         /* shared text (lower-cased where matched) */
    const ro_saidx_t *SA;
    const ro_lcp_t   *LCP;
    const uint16_t   *SO;        /* NULL unless main nsamples > 2 */
    const ro_saidx_t *nsep;
    ro_saidx_t        n;         /* ranks in this (sub)index */
    ro_saidx_t        nT;        /* length of main text */
    int               main_nsamples;
    int               rc;
} ro_view;

/* getmums (reveal.c:55-116) / getmums_rem (reveal.c:119-180): same predicate;
 * they differ in the rc remap (nT vs n).  rem!=0 selects getmums_rem.
 * Writes up to cap triples, returns the total number found. */
int64_t ro_getmums(const ro_view *v, int minl, int rem,
                   ro_lcp_t *out_l, ro_saidx_t *out_a, ro_saidx_t *out_b, int64_t cap);

/* getmultimums (reveal.c:436-580, ismultimum :227-259) and getmultimems
 * (reveal.c:292-434, ismultimem :261-290) in CSR form: match k has length
 * out_l[k], count field out_n[k] (members for mums, distinct samples for
 * mems) and members out_so/out_pos[out_off[k] .. out_off[k+1]) in SA order.
 * Returns number of matches (or -needed if a capacity was too small;
 * *members_needed always receives the total member count). */
int64_t ro_getmultimums(const ro_view *v, int minl, int minn, int mems,
                        ro_lcp_t *out_l, int32_t *out_n, int64_t *out_off, int64_t cap_matches,
                        uint16_t *out_so, ro_saidx_t *out_pos, int64_t cap_members,
                        int64_t *members_needed);

/* ---- split / bubble (reveal.c:582-727) --------------------------------- */

/* D-label by scatter through SAi as the aligner does (reveal.c:1005-1117):
 * lead=1, trail=2, rest=4, then the matched [sp, sp+l) ranges =3.
 * Intervals are half-open (begin,end) pairs.  D must hold idx n bytes, zeroed
 * by the callee.  Returns counts through *nl,*nt,*np (positions labelled,
 * exactly the reference's leadingn/trailingn/parn). */
void ro_label(uint8_t *D, ro_saidx_t n, const ro_saidx_t *SAi,
              const ro_saidx_t *lead, int nlead, const ro_saidx_t *trail, int ntrail,
              const ro_saidx_t *rest, int nrest,
              const ro_saidx_t *sp, int nsp, ro_lcp_t l,
              ro_saidx_t *nl, ro_saidx_t *nt, ro_saidx_t *np);

/* split (reveal.c:582-664).  Child arrays must have room for the counts
 * returned by ro_label.  Rewrites the shared SAi.  Returns ranks written to
 * each child through il/it/ip. */
void ro_split(const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, const uint8_t *D,
              ro_saidx_t *SAi,
              ro_saidx_t *lSA, ro_lcp_t *lLCP, ro_saidx_t *tSA, ro_lcp_t *tLCP,
              ro_saidx_t *pSA, ro_lcp_t *pLCP,
              ro_saidx_t *il, ro_saidx_t *it, ro_saidx_t *ip);

/* bubble_sort (reveal.c:666-727) over the matched intervals' begins, in the
 * order given. */
void ro_bubble_sort(ro_saidx_t *SA, ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                    const ro_saidx_t *match_begin, int nmatch);

/* ---- host-driven single steps (reveal.c:1386-1748) ---------------------- */

/* splitindex (reveal.c:1515-1748) = label from interval lists + split +
 * bubble_sort(leading, matching).  Interval lists are (begin,end) pairs. */
void ro_splitindex(char *T, const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                   const uint16_t *SO, const ro_saidx_t *nsep, int main_nsamples,
                   const ro_saidx_t *lead, int nlead, const ro_saidx_t *trail, int ntrail,
                   const ro_saidx_t *match, int nmatch, const ro_saidx_t *rest, int nrest,
                   ro_saidx_t *lSA, ro_lcp_t *lLCP, ro_saidx_t *tSA, ro_lcp_t *tLCP, ro_saidx_t *pSA, ro_lcp_t *pLCP,
                   ro_saidx_t *counts, int *nsamples);

/* extract (reveal.c:1386-1505); returns the new n (-1: rank 0 matched, where
 * the reference overruns its buffers).  oSA[0] = SA[0] (never written by the
 * reference). */
ro_saidx_t ro_extract(char *T, const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                      const ro_saidx_t *nsep, ro_saidx_t nT, int rc,
                      ro_saidx_t *intervals, int niv, ro_saidx_t *oSA, ro_lcp_t *oLCP);

/* ---- aligner work loop (reveal.c:731-1338, interface.c:293-415) -------- */

typedef struct { ro_saidx_t begin, end; } ro_intv;

typedef struct ro_main ro_main;

typedef struct {
    ro_main    *main;
    ro_saidx_t *SA;
    ro_lcp_t   *LCP;
    ro_saidx_t  n;
    int         depth;
    int         nsamples;
    ro_intv    *nodes;      /* the sub-index' intervals ("nodes", reveal.h:36) */
    int         nnodes;
} ro_index;

struct ro_main {
    char       *T;          /* nT+1 bytes, NUL terminated, mutated (lower-casing) */
    ro_saidx_t *SAi;        /* shared inverse (rewritten by split/bubble) */
    uint16_t   *SO;         /* NULL unless nsamples > 2 */
    ro_saidx_t *nsep;
    ro_saidx_t  nT;
    int         nsamples;
};

/* A match handed to / returned by the callbacks: (l, n, ((sample,pos)...))
 * as built at reveal.c:166-170 / :497. */
typedef struct {
    ro_lcp_t    l;
    int         n;
    uint16_t   *so;     /* n sample ids   */
    ro_saidx_t *pos;    /* n text positions */
} ro_mum;

typedef struct {
    int64_t     count;
    ro_lcp_t   *l;
    int32_t    *n;
    int64_t    *off;    /* count+1 */
    uint16_t   *so;
    ro_saidx_t *pos;
} ro_mumlist;

/* What graphalign returns (reveal.c:987): interval lists.  The callee
 * allocates with malloc; the loop frees. matching is iterated in the order
 * given (reveal.c:673-674). */
typedef struct {
    ro_intv *lead;  int nlead;
    ro_intv *trail; int ntrail;
    ro_intv *match; int nmatch;
    ro_intv *rest;  int nrest;
} ro_splitspec;

/* mumpicker(mums, idx): return 0 = "()" (stop this branch, reveal.c:870),
 * 1 = *chosen filled (index into the list through *pick). */
typedef int (*ro_picker_fn)(void *user, const ro_index *idx, const ro_mumlist *mums, int64_t *pick);
/* graphalign(idx, mum): return 0 = None (reveal.c:962), 1 = *spec filled. */
typedef int (*ro_graphalign_fn)(void *user, const ro_index *idx, const ro_mum *mum, ro_splitspec *spec);

/* One record per popped sub-index, in pop (LIFO) order -- the callback trace. */
typedef struct {
    ro_saidx_t key;        /* smallest interval begin of the sub-index */
    ro_saidx_t n;
    int32_t    depth;
    int32_t    nsamples;
    int32_t    nnodes;
    int64_t    nmums;      /* size of the list handed to the picker */
    int32_t    picked;     /* 1 if a split happened */
    ro_lcp_t   l;          /* chosen match */
    int32_t    mn;
    ro_saidx_t sp_min;     /* smallest member position of the chosen match */
    uint64_t   h_sa;       /* ro_hash_step over the sub-index SA */
    uint64_t   h_lcp;      /* ro_hash_step over the sub-index LCP */
    uint64_t   h_mums;     /* ro_hash_step over the flat scan result l,n,(so,pos)... */
} ro_trace;

typedef struct {
    int64_t   nsteps;      /* sub-indices popped */
    int64_t   nsplits;     /* matches anchored (nmums, interface.c:334) */
    int64_t   anchored_bp; /* sum of l over anchors */
    int       maxdepth;
    double    t_scan, t_pick, t_split, t_bubble;
} ro_align_stats;

/* Runs the LIFO loop of aligner() with threads==0 (interface.c:387-399).
 * Takes ownership of SA/LCP (freed like reveal.c:1279-1284).  trace may be
 * NULL; otherwise up to trace_cap records are written.  anchors (optional):
 * for every split, l and the member positions appended to anchor_l/anchor_off/
 * anchor_pos (CSR; caps given).  Returns 0, or -1 on error. */
int ro_align(ro_main *m, ro_saidx_t *SA, ro_lcp_t *LCP, ro_saidx_t n,
             const ro_intv *nodes, int nnodes,
             ro_picker_fn picker, ro_graphalign_fn galign, void *user,
             int minl, int minn,
             ro_trace *trace, int64_t trace_cap,
             ro_align_stats *stats);

/* The bench callbacks: longest full match, ties -> smallest
 * minimum coordinate; linear interval model for graphalign.  `user` must be
 * an ro_bench_ctx*; anchors are appended to it. */
typedef struct {
    int64_t     nanchors, cap_anchors;
    ro_lcp_t   *l;
    int32_t    *n;
    int64_t    *off;        /* cap_anchors+1 */
    int64_t     npos, cap_pos;
    ro_saidx_t *pos;
} ro_bench_ctx;
int ro_bench_picker(void *user, const ro_index *idx, const ro_mumlist *mums, int64_t *pick);
int ro_bench_graphalign(void *user, const ro_index *idx, const ro_mum *mum, ro_splitspec *spec);

/* order-sensitive sequence hash: h = sum_i mix(v_i + (i+1)*GOLDEN) mod 2^64 */
uint64_t ro_hash_step(uint64_t acc, uint64_t i, int64_t v);

#ifdef __cplusplus
}
#endif
#endif
