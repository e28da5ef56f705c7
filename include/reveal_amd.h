/*
 * reveal_amd.h -- C ABI of the MI355X-native reveallib hot path.
 *
 * Drop-in boundary for jasperlinthorst/reveal's `reveallib` / `reveallib64`
 * C extension (reveallib/interface.c, reveallib/reveal.c).  The reference
 * binds this path as a CPython type (`index`, interface.c:474-487, 731-785);
 * this header is what that type's methods would call if the extension were
 * rebuilt on top of the GPU library (see INTEGRATION.md for the binding).
 * Plain C: opaque handle, pointers and sizes, no torch / HIP types.
 *
 * Two shared objects export the same symbols, mirroring the reference's two
 * modules (reveallib/reveal.h:7-13, setup.py:19-32):
 *     libreveal_amd.so    saidx_t = int32_t, lcp_t = int32_t   (reveallib)
 *     libreveal_amd64.so  saidx_t = int64_t, lcp_t = uint32_t  (reveallib64)
 * Text positions, counts and interval bounds cross the ABI as int64_t in both;
 * only rv_get_array() hands out arrays in the native element width.
 *
 * Every function returns 0 on success and a negative value on error unless
 * stated otherwise; rv_last_error() describes the last failure of the calling
 * thread (the reference raises `reveallib.error` / TypeError there).
 * All work runs on the handle's own HIP stream; calls are synchronous unless
 * stated otherwise.  There is NO CPU fallback: without a usable gfx950 device
 * rv_new() fails.
 */
#ifndef REVEAL_AMD_H
#define REVEAL_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rv_index rv_index;

/* ---- library ------------------------------------------------------------ */
const char *rv_last_error(void);
int  rv_abi_version(void);              /* bumped on incompatible changes: 2 = rv_picker_info writes six values */
int  rv_sa_bits(void);                  /* 32 or 64: which module this library is */
int  rv_device_count(void);             /* visible HIP devices (0 = none) */

/* ---- index lifetime: reveal_init / reveal_dealloc (interface.c:489-521, 787-839) */
rv_index *rv_new(int device);           /* NULL on failure */
void rv_free(rv_index *h);

/* ---- text assembly (host) ------------------------------------------------ */
/* addsample (interface.c:18-49): starts a new sample; records nsep for the
 * previous one. */
int rv_add_sample(rv_index *h);
/* addsequence (interface.c:51-95): appends seq + '$'; returns the half-open
 * interval [*begin,*end) of the sequence (excluding the '$').  The 32-bit
 * library fails like interface.c:61-68 when the text would exceed INT_MAX.
 * rv_add_sample / rv_add_sequence after a construct() make the index
 * "not yet constructed" again (the arrays in HBM describe the old text). */
int rv_add_sequence(rv_index *h, const char *seq, int64_t len, int64_t *begin, int64_t *end);
/* Not in the reference (its callers make a new index object per input, interface.c:489-521): rv_reset forgets the text and the
 * samples and keeps every allocation of the handle -- ~86 B of device memory per position, the page-locked host text, streams --
 * for the next input's rv_add_sample / rv_add_sequence; rv_reserve_text says how many bytes of text are to come (sequences + one
 * separator each), so the host buffer is allocated once.  A text of 8 MB and more lives in page-locked host memory: rv_upload /
 * rv_construct then move it into HBM as one DMA copy (500 MB: 10 ms, no host thread involved), which a caller with several inputs
 * overlaps with the alignment before it by giving each input in flight its own handle and host thread (bench.py --config stream). */
int rv_reset(rv_index *h);
int rv_reserve_text(rv_index *h, int64_t bytes);
int64_t rv_n(const rv_index *h);        /* reveal_getn (interface.c:681-689): ranks in the main index */
int rv_nsamples(const rv_index *h);     /* interface.c:691-695 */
int rv_nnodes(const rv_index *h);       /* number of sequence intervals added so far */
int rv_add_sequences(rv_index *h, const char *text, int64_t total, const int64_t *lens, int64_t count);      /* count sequences at once: each followed by '$' in text */
int rv_node_list(const rv_index *h, int64_t *begin_end);      /* those intervals, 2 * rv_nnodes numbers (sequences added inside the library: rv_graph_read_gfa) */

/* ---- construct (interface.c:160-291) ------------------------------------- */
/* rc!=0 reverse-complements T[nsep[0]..n) first (interface.c:168-175).
 * safile / lcpfile (may be NULL or ""): raw native-endian saidx_t[n] / lcp_t[n]
 * read instead of computed (interface.c:224-232, 255-263).  cache!=0 writes
 * .reveal.t/.reveal.sa/.reveal.lcp to the CWD (interface.c:182-189, 274-285).
 * On return T, SA and LCP live in HBM (the inverse SAi is made when something asks for it: the RV_SAI getter, rv_clone,
 * rv_sx_main, rv_align_begin).  Both libraries carry ranks, child sizes and scan records in 32 bits: an index of
 * n >= 2^32 - 2 positions is refused here, whatever the width of saidx_t and wherever SA comes from (built, or read from
 * safile, which is range- and permutation-checked on the device before anything scatters through it).  The 64-bit library
 * is exercised above 2^31 positions -- where the reference needs reveallib64, reveal.h:7-13 -- at n = 2.2 x 10^9
 * (tests/test_gpu_above_2_31.py: construct, both recursion paths, and this refusal).
 * The byte budget behind that cap (measured, tools/mem_probe.py on one MI355X, 309 GB visible): after construct() + align a handle of the
 * 64-bit library holds 119 B per position (262 GB at n = 2.2 x 10^9; the 32-bit library 85 B: 42.6 GB at n = 5 x 10^8) -- text and working copy
 * 2, SA 8, LCP 4, BWT 1, the SA build's scratch, kept between calls, ~95 (two buffers of 8-byte keys and of 8-byte suffixes sized for n, group heads /
 * seeds / ranks / the round-0 list at 4 to 8 bytes each, the packed text, the diagonal bits), the recursion's lists the rest.  309 GB / 119 B =
 * 2.6 x 10^9 positions: HBM, not the 32-bit ranks, is what stops this library first; two human genomes (6.2 x 10^9) need both wider ranks and a
 * build whose scratch is 40 B per position -- not built.  The reference's 64-bit module has no such ceiling besides host memory
 * (reveal.h:7-13, interface.c:61-68). */
int rv_construct(rv_index *h, int rc, const char *safile, const char *lcpfile, int cache);
/* Copies the assembled text to HBM now (construct does it on demand).  Lets a
 * caller keep the host->device copy out of a timed construct(); repeated
 * construct() calls on an unchanged text start from the HBM-resident copy. */
int rv_upload(rv_index *h);
/* the same copy once more, whether or not the text changed: what the text's way into HBM costs once the buffers exist (bench.py's
 * upload_ms; the first rv_upload of a handle also pays for its device allocations) */
int rv_upload_again(rv_index *h);

/* getters (interface.c:538-729).  which: */
enum { RV_T = 0, RV_SA = 1, RV_SAI = 2, RV_LCP = 3, RV_SO = 4, RV_NSEP = 5, RV_NODES = 6 };
/* Copies the array to `out` (capacity in elements): T = n chars (current
 * text incl. lower-case marks), SA/SAI = saidx_t[n], LCP = lcp_t[n], SO =
 * uint16_t[n] (only when nsamples > 2), NSEP = int64_t[nsamples-1], NODES =
 * int64_t[2*nnodes].  Returns the element count or <0 (e.g. SA/LCP of the main
 * index after align(), which the reference frees, reveal.c:1279-1284). */
int64_t rv_get_array(rv_index *h, int which, void *out, int64_t cap);

/* ---- scans on the main index ---------------------------------------------- */
/* getmums(minl) (reveal.c:55-116): pairwise MUMs between sample 0 and the
 * rest, increasing SA rank.  rv_getmums runs the scan and returns the number
 * of matches; rv_fetch_mums copies them (l, a, b; b already rc-remapped). */
int64_t rv_getmums(rv_index *h, int minl);
int rv_fetch_mums(rv_index *h, uint32_t *l, int64_t *a, int64_t *b, int64_t cap);
/* getmultimums / getmultimems (reveal.c:436-580 / 292-434) in CSR form, in the
 * reference's emission order; mems!=0 selects getmultimems.  Returns the match
 * count, *members the total member count; rv_fetch_multi copies
 * l[k], n[k], off[k..k+1], so[], pos[]. */
int64_t rv_getmultimums(rv_index *h, int minlength, int minn, int mems, int64_t *members);
int rv_fetch_multi(rv_index *h, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos);

/* ---- the recursion: align() / aligner() (interface.c:293-415, reveal.c:731-1338)
 *
 * The reference pops one sub-index at a time (LIFO) and calls back into
 * Python twice per step.  Children of a split cover disjoint text and never
 * read each other's state, so this library processes the recursion level by
 * level: all sub-indices of a level ("the frontier") sit back to back in HBM
 * and are scanned / split by single launches.  The caller (the Python `index`
 * type, or rv_align_builtin) supplies the two callbacks' decisions per
 * sub-index between rv_frontier_scan and rv_frontier_commit. */
typedef struct {
    int64_t n;            /* ranks (reveal.h:25) */
    int32_t depth;        /* reveal.h:28 */
    int32_t nsamples;     /* reveal.h:29 */
    int32_t nnodes;       /* intervals of this sub-index (reveal.h:36) */
    int32_t parent;       /* frontier slot of the parent in the previous level, -1 for the main index */
    int32_t kind;         /* 0 main, 1 leading, 2 trailing, 3 parallel child */
    int32_t reserved;
    int64_t nmums;        /* matches found by the last rv_frontier_scan */
    int64_t nmembers;     /* total members of those matches */
} rv_sub;

int rv_align_begin(rv_index *h, int minl, int minn);      /* frontier = {main index} */
int rv_frontier_size(rv_index *h);
/* getmums_rem / getmultimums on every sub-index of the frontier (reveal.c:802-822) */
int rv_frontier_scan(rv_index *h);
/* Sub-index s of the current frontier (made by the last rv_frontier_commit) was seeded by its parent's mumpicker -- its Python
 * object carries a non-empty `skipmums` list (reveal.c:1157, 1180) -- so, like the reference (reveal.c:802, 830-837), it is
 * not scanned: rv_frontier_scan reports no matches for it, and skips its launch when every sub-index of the level is seeded. */
int rv_sub_skip_scan(rv_index *h, int s);
int rv_sub_info(rv_index *h, int s, rv_sub *out);
int rv_sub_nodes(rv_index *h, int s, int64_t *begin_end /* 2*nnodes */);
/* matches of sub-index s as (l, n, ((sample,pos)...)) in CSR form */
int rv_sub_mums(rv_index *h, int s, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos);
/* SA / LCP of sub-index s (which = RV_SA, RV_LCP, RV_SAI: SAi restricted to
 * the sub-index' text positions is not contiguous, so RV_SAI returns the whole
 * shared array like the reference's getter) */
int64_t rv_sub_array(rv_index *h, int s, int which, void *out, int64_t cap);
/* Decision for sub-index s: the match chosen by mumpicker (l, member
 * positions sp[nsp]) and graphalign's interval lists (reveal.c:987), each as
 * nX (begin,end) pairs.  Sub-indices without a decision end here, exactly
 * like mumpicker returning () (reveal.c:870-884). */
int rv_sub_split(rv_index *h, int s, uint32_t l, int nsp, const int64_t *sp,
                 const int64_t *lead, int nlead, const int64_t *trail, int ntrail,
                 const int64_t *match, int nmatch, const int64_t *rest, int nrest);
/* D-label + split + lower-casing + bubble_sort (reveal.c:1005-1252) for every
 * registered decision; the children become the new frontier.  children (may
 * be NULL) receives 3 ints per old sub-index: new frontier slots of its
 * leading, trailing and parallel child, or -1. */
int rv_frontier_commit(rv_index *h, int32_t *children);
int rv_align_end(rv_index *h);

/* Whole recursion with the built-in deterministic callbacks used by bench.py
 * and the parity tests (longest full match, ties -> smallest minimum
 * coordinate; linear interval model), no Python in the loop. */
typedef struct {
    int64_t steps;            /* sub-indices visited */
    int64_t splits;           /* anchors */
    int64_t anchored_bp;      /* sum of l */
    int32_t levels;
    int32_t maxdepth;
    int64_t scanned_ranks;    /* ranks streamed by the scan kernel over all levels */
    double  t_scan, t_host, t_split, t_bubble;   /* seconds, host clock */
} rv_align_stats;
int rv_align_builtin(rv_index *h, int minl, int minn, rv_align_stats *st);

/* ---- one alignment over several devices: frontier hand-off -------------------
 * No counterpart in the reference's API: it is the work-queue reading of its own
 * recursion (reveal.c:21-25 stack, :731-1338 aligner threads popping independent
 * sub-indices; children of a split cover disjoint text and disjoint ranges of the
 * shared SAi, reveal.c:597-630, and are pushed after their parent's lower-casing,
 * :1230-1234 before :1296).  SURVEY.md 8(e), second granularity.
 *   owner :  rv_align_builtin_until(stop_subs) -> frontier of >= stop_subs sub-indices
 *            rv_frontier_counts / rv_frontier_export -> metadata
 *            rv_frontier_pack(subset) -> SA / LCP / BWT segments of a subset (9 B/rank,
 *            13 B in the 64-bit library), into device memory (peer / RCCL send) or host memory
 *            rv_frontier_import(own subset) ; rv_align_builtin_resume
 *   worker:  rv_new, rv_add_sample / rv_add_sequence (same text, no construct)
 *            rv_frontier_import(received subset) ; rv_align_builtin_resume
 * The union of all handles' anchors (rv_fetch_anchors) is the anchor set of the
 * undivided run; the lower-cased text is the pristine text with every anchor's
 * members lower-cased. */
/* returns the frontier size (0: the run finished before reaching stop_subs and has been collected), < 0 on error */
int rv_align_builtin_until(rv_index *h, int minl, int minn, int stop_subs, rv_align_stats *st);
int rv_align_builtin_resume(rv_index *h, rv_align_stats *st);
/* a run stopped by rv_align_builtin_until goes on (at least one more level) until its frontier holds >= stop_subs sub-indices;
 * returns the new frontier size, 0 when the run finished on the way (collected, as rv_align_builtin_until does) */
int rv_align_builtin_continue(rv_index *h, int stop_subs, rv_align_stats *st);
/* out[0..3] = sub-indices, ranks, intervals of the frontier, level */
int rv_frontier_counts(rv_index *h, int64_t *out);
/* meta: 6 per sub-index (offset, n, depth, nsamples, kind, parent); node_first: nsubs+1; nodes: (begin,end) pairs */
int rv_frontier_export(rv_index *h, int64_t *meta, int64_t *node_first, int64_t *nodes);
/* segments of the listed sub-indices back to back -> caller memory; returns the ranks written */
int64_t rv_frontier_pack(rv_index *h, const int32_t *subs, int k, void *sa, void *lcp, void *bwt, int on_device);
int rv_frontier_import(rv_index *h, int minl, int minn, uint32_t maxlcp, int level, int nsubs, const int64_t *meta,
                       const int64_t *node_first, const int64_t *nodes, int64_t m,
                       const void *sa, const void *lcp, const void *bwt, int on_device);
/* rv_set_picker(h, 1, ..) runs: the seed lists of the listed sub-indices of the frontier (what the reference's children carry as skipmums,
 * reveal.c:1157, 1180; schemes.py:321-332) travel with them -- export before the owner's frontier is replaced, import right after
 * rv_frontier_import of the same sub-indices in the same order.  Words per sub-index: count, then per seed l, n, score, members,
 * (sample, position) x members.  rv_frontier_seeds_export returns the number of words and writes them when cap is large enough. */
int64_t rv_frontier_seeds_export(rv_index *h, const int32_t *subs, int k, int64_t *out, int64_t cap);
int rv_frontier_seeds_import(rv_index *h, int nsubs, const int64_t *words, int64_t nwords);
/* largest LCP value of the constructed index (= window of bubble_sort, reveal.c:666-727; workers need the owner's) */
uint32_t rv_maxlcp(const rv_index *h);
/* what the anchor cascade (rv_cascade.hip) did in the last rv_align_builtin: out[0] = 1 it decided the run / 0 the level
 * pipeline ran (out[1..] then describe the abandoned attempt), out[1] levels, out[2] top-level matches, out[3] repeat
 * witnesses, out[4] sub-indices visited, out[5] sub-indices left undecided (rebuilt from their text and handed to the leaf
 * kernel), out[6] ranks rebuilt, out[7] large undecided sub-indices decided from their repeat witnesses (the second attempt) */
int rv_cascade_info(const rv_index *h, int64_t *out);
const char *rv_cascade_why(const rv_index *h);      /* why the cascade left the last built-in run to the level pipeline ("": it did the run, or was not tried) */
/* anchors chosen by the last rv_align_builtin: l[k], members off[k..k+1] -> pos[] (sorted) */
int64_t rv_anchor_count(rv_index *h, int64_t *members);
int rv_fetch_anchors(rv_index *h, uint32_t *l, int64_t *off, int64_t *pos);
/* Optional: arrays of the caller's that the NEXT rv_align_builtin runs deliver their anchors into directly (capacities in elements; the
 * layout rv_fetch_anchors fills).  The library page-locks them (hipHostRegister) until they are replaced, cleared (all NULL / 0) or the
 * handle is freed -- they must stay allocated that long.  Each array must begin on a page boundary and own the pages it lies on up to
 * the end of the last one (mmap, posix_memalign to whole pages: locking and unlocking act on whole pages, and a page shared with other
 * heap objects loses its GPU mapping under them) -- arrays that do not begin on a page boundary are not locked.  A run whose result
 * does not fit, or arrays that are not / cannot be locked, use the library's staging buffer as before; rv_fetch_anchors called with these same three pointers then only completes what the run has not
 * written itself.  (2 x 250 Mbp: 2 x 10^6 anchors, 65 MB -- copied out of the staging buffer behind the run they were 1.3-1.6 ms of a
 * 25 ms step with the GPU idle.)
 * Lifetime: the arrays are the library's from this call until they are replaced, cleared, or the handle is freed; CLEAR THEM BEFORE
 * FREEING THEM (a freed page that is still locked faults under a later copy).  Clearing or replacing them between a run and its
 * rv_fetch_anchors is allowed: what the run delivered moves to the library's staging buffer first and rv_fetch_anchors hands it out as
 * usual.  Every run (rv_align_builtin, _until, _continue, _resume, and a worker's rv_frontier_import + _resume) writes into whatever
 * is set when it ends: a caller that still reads an earlier result from these arrays must clear or replace them before the next run. */
int rv_set_result_buffers(rv_index *h, uint32_t *l, int64_t l_cap, int64_t *off, int64_t off_cap, int64_t *pos, int64_t pos_cap);
/* per-sub-index trace of the last rv_align_builtin when tracing was enabled
 * (tests only: costs a D2H copy of every sub-index) */
typedef struct {
    int64_t  key, n;
    int32_t  depth, nsamples, nnodes, picked;
    int64_t  nmums;
    uint32_t l; int32_t mn;
    int64_t  sp_min;
    uint64_t h_sa, h_lcp, h_mums;
} rv_trace;
int rv_set_trace(rv_index *h, int on);
/* Anchor pre-selection for the callback protocol (not in the reference's C, it restates what its Python picker
 * does first with every list: schemes.py:227 keep the matches with n == idx.nsamples, schemes.py:240 + 245-247 + 287-289 of
 * those the `maxmums` longest, of equal lengths the later emitted).  With maxmums > 0, rv_sub_info / rv_sub_mums between
 * rv_align_begin and rv_align_end report only those, in emission order; a sub-index without a match in every sample keeps
 * its whole list (schemes.py:229-232 segments over all of them).  0 = off.  A picker that starts with the same filter and
 * cap (graphmumpicker with --maxmums, no --trim) returns what it returns on the full list.  The n == nsamples filter runs in the
 * scan kernel (only what passes it is copied to the host; a sub-index left without a match is scanned again without the filter).  The cap:
 * with two samples the `maxmums` longest of every sub-index are chosen on the device once a level holds RV_PRESEL_DEV_MIN records (option,
 * default 65536: a stable sort of (sub-index, length) keys, the last `maxmums` of every run kept, the kept records in their old order --
 * the others never cross into host memory); smaller levels and more than two samples: on the host side of this ABI. */
int rv_set_preselect(rv_index *h, int64_t maxmums);
/* Switches of one handle: test hooks, diagnostics and A/B paths (reveal_amd/csrc/rv_common.h RV_OPTION_LIST has the names and
 * defaults -- the historical RV_* spellings, e.g. "RV_NO_CASCADE").  The library never reads the process environment: a handle
 * starts with the defaults, rv_set_option changes one switch of that handle (a copy() inherits them); the Python layer applies
 * RV_* environment variables when it makes a handle.  No counterpart in the reference (its only switch is -DREVEALDEBUG). */
int rv_set_option(rv_index *h, const char *name, int64_t value);
int rv_get_option(rv_index *h, const char *name, int64_t *value);
/* process-wide diagnostics (not a handle's switch): print the source line of every kernel launch and wait for it; returns the previous setting */
int rv_set_launch_trace(int on);
int rv_option_count(void);
const char *rv_option_name(int k);
int64_t rv_trace_count(rv_index *h);
int rv_fetch_trace(rv_index *h, rv_trace *out, int64_t cap);

/* ---- host-driven single steps: copy / splitindex / extract ----------------------
 * (interface.c:432-470 copy, reveal.c:1515-1748 splitindex, reveal.c:1386-1505 extract; public methods of the
 * reference's index type with no live caller in its package -- the Python-driven form of the recursion that
 * rem.py:580-609 sketches: scan a (sub)index, split it, go on with the children it returns.)
 * A detached (sub)index owns SA / LCP (and BWT) in HBM and borrows text, shared inverse and separators of its main
 * handle, like the reference's child objects (reveal.c:1679-1735); it must be freed before the handle. */
typedef struct rv_subindex rv_subindex;
rv_index *rv_clone(rv_index *h);                      /* copy() of a main index: an independent handle (own text and arrays) */
rv_subindex *rv_sx_main(rv_index *h);                 /* the constructed main index as a detached (depth 0) index; NULL on failure */
rv_subindex *rv_sx_copy(rv_subindex *x);
void rv_sx_free(rv_subindex *x);
int rv_sx_info(const rv_subindex *x, rv_sub *out);    /* n, depth, nsamples, nnodes */
int rv_sx_nodes(const rv_subindex *x, int64_t *begin_end);
int64_t rv_sx_array(rv_subindex *x, int which, void *out, int64_t cap);      /* RV_SA, RV_LCP */
/* getmums / getmultimums over this index: number of matches (members through *members); rv_sx_fetch hands them out in
 * the CSR form of rv_sub_mums */
int64_t rv_sx_scan(rv_subindex *x, int minl, int minn, int64_t *members);
int rv_sx_fetch(rv_subindex *x, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos);
/* splitindex: interval lists as (begin,end) pairs; out[0..2] = leading, trailing, parallel child or NULL.  Lower-cases
 * the matching intervals, bubble_sorts the leading child over them in the order given.  x itself keeps its arrays. */
int rv_sx_split(rv_subindex *x, const int64_t *lead, int nlead, const int64_t *trail, int ntrail,
                const int64_t *match, int nmatch, const int64_t *rest, int nrest, rv_subindex **out);
/* extract, in place.  `intervals` is rewritten where construct(rc=1) makes the reference remap it (reveal.c:1411-1427). */
int rv_sx_extract(rv_subindex *x, int64_t *intervals, int niv);

/* ---- host side of the anchor picker -----------------------------------
 * chain() of the reference's Python picker (reveal/schemes.py:20-105 with utils.gapcost, utils.py:162-183): the best-scoring
 * collinear chain through m pre-selected matches over k paths, between the sentinels `left` and `right`.  Match i has length
 * len[i], spans nmem[i] samples and starts at crd[i*k + j] on path j (path 0 = the reference's `ref` dimension); score of a
 * match = wscore * len * nmem*(nmem-1)/2, penalty = wpen * gapcost(end of predecessor, start of match); model 0 = sumofpairs,
 * 1 = star-avg, 2 = star-med.  Plain host code (the DP is O(m^2 k^2) on at most --maxmums matches).  out_idx / out_score
 * (capacity m) receive the chain from left to right: input indices and the running scores.  Returns the chain length, < 0 on
 * error.  Decision-for-decision identical to the reference's function, ties included (tests/golden/chain_vectors.json). */
int64_t rv_chain(int64_t m, int k, const uint32_t *len, const int32_t *nmem, const int64_t *crd, const int64_t *left,
                 const int64_t *right, int64_t wscore, int64_t wpen, int model, int64_t *out_idx, int64_t *out_score);

/* One sub-index' decision by the reference's default picker -- `graphmumpicker`, reveal/schemes.py:197-361, the branch that is not
 * "precomputed" -- for FASTA inputs with one sequence per sample: keep the matches present in every sample of the sub-index (the best sample
 * subset when there is none: `segment`), trim overlaps (--trim, the default), order, cap at --maxmums, chain (rv_chain), split on the largest
 * match of the chain, seed the children with the rest of the chain above --seedsize, and with minlength == 0 the p-value cut.  Plain host
 * code, no device involved.  Input: the sub-index' scan result as rv_sub_mums hands it out (m matches, members off[i]..off[i+1] in emission
 * order: sample so[], text position pos[]), nsub = samples of the sub-index, and per sample of the index (nsamples): where its sequence
 * begins and the sub-index' interval of it (iv_begin < 0: none).  Returns 1 with the choice and the seeds in *out (seed w: seed_l / seed_n /
 * members seed_off[w]..seed_off[w+1] / seed_score / seed_right 0 = for the leading, 1 = for the trailing child), 0 for the reference's `()`
 * (stop this branch), -2 where the reference's own code raises (KeyError / IndexError), -1 on bad arguments.  The caller owns every array. */
typedef struct { int64_t wscore, wpen, maxmums, seedsize; int gcmodel; int trim; double pcutoff; } rv_picker_args;
typedef struct {
    int picked; uint32_t pick_l; int32_t pick_n; int pick_members; uint16_t *pick_so; int64_t *pick_pos; int64_t member_cap;
    int64_t nleft, nright, nseed_members, seed_cap, seed_member_cap;
    uint32_t *seed_l; int32_t *seed_n; int64_t *seed_off; uint16_t *seed_so; int64_t *seed_pos; int64_t *seed_score; uint8_t *seed_right;
} rv_picker_out;
/* The picker of rv_align_builtin and its relatives: kind 0 = the benchmark picker (the longest match present in every sample; the
 * default), kind 1 = the reference's default picker with these options (rv_pick_chain per sub-index, seeds handed to the children as the reference's
 * skipmums: reveal.c:802, 830-837, 1157, 1180) under the linear interval model -- `reveal rem a.fa b.fa` with its defaults, no Python callback per
 * sub-index.  Kind 1 takes inputs with one sequence per sample; the scans hand their whole lists to the host (no device-side pick, no leaf
 * kernel, no anchor cascade); rv_fetch_anchors hands out an anchor's members in the picker's order (the scan's emission order), not sorted.  rv_picker_info: out[0] kind, out[1] picker calls of the last run, out[2] of them seeded, out[3] nanoseconds inside rv_pick_chain,
 * out[4] nanoseconds putting the lists together, out[5] nanoseconds inside graphalign (kind 2, rv_set_graph_picker); six values. */
int rv_set_picker(rv_index *h, int kind, const rv_picker_args *args);
int rv_picker_info(const rv_index *h, int64_t *out);
int rv_pick_chain(const rv_picker_args *args, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so,
                  const int64_t *pos, int nsamples, const int64_t *seq_begin, const int64_t *iv_begin, const int64_t *iv_end, int minlength,
                  rv_picker_out *out);

/* ---- the alignment graph of a finished run (host code; reveal/rem.py:14-200, 318-345) ---------------------------------------------
 * With the picker inside the library the recursion hands back its anchors in the order it chose them, and the graph `reveal rem` builds -- every
 * member's node broken around the match (breaknode, rem.py:14-131), the pieces merged into the first (mergenodes, rem.py:133-200) -- depends on
 * that order alone.  rv_graph_replay does this surgery for inputs with one sequence per sample (nseq sequences at [begin[s], end[s]) of the index
 * text, path id = s; anchor a: length an_l[a], members an_pos[an_off[a] .. an_off[a+1]) in the picker's order) starting from the FASTA reader's
 * graph (start sentinel, sequence, end sentinel per sequence: utils.py:304-375).  What comes back keeps the reference structure's ORDER -- nodes in
 * creation order (the GFA writer's numbering), a node's links in the order they were last made (the writer's L lines; which sibling survives
 * prune_nodes) --, so the GFA written from it is the callbacks' byte for byte.  rv_graph_sizes: out[0] nodes, out[1] offset entries, out[2] edges,
 * out[3] path entries.  rv_graph_export: per node (b, e, aligned) with aligned = -1 for a sentinel (b = sample, e = 0 start / 1 end); CSR arrays
 * (ptr arrays one longer than their count) of the nodes' (path id, offset) entries, of their links forwards and backwards as (neighbour's number,
 * edge number), and of the edges' path ids.  rv_graph_error: NULL, or why the replay stopped (an anchor outside every node). */
typedef struct rv_graph rv_graph;
/* ---- graphs as INPUTS (the levels 1 and 2 of `reveal align --order=sequential`: `reveal rem` of GFA files, reveal/utils.py:377-677) ----------------------------
 * rv_graph_import: the graph the readers made (reveal_amd/alngraph.py read_gfa / read_fasta), in its dictionary order: nodes (b, e, aligned; sentinels: aligned = -1,
 * sent = 1 start / 2 end), their (path id, offset) entries as CSR, the links forwards node by node (edge_u / edge_v, path ids as CSR), every node's links backwards as
 * edge numbers in dictionary order, per path: a '*' name? / its length; the start sentinels in the readers' order; literal_segments = alngraph.check_segment_shortcut said no.
 * Forward-strand links only.
 * rv_graph_pick = schemes.graphmumpicker's not-precomputed branch (schemes.py:197-361) for one sub-index of such an alignment (arguments as rv_pick_chain; left / right =
 * the sub-index' left / right graph node as (begin, end), begin < 0: None); rv_graph_align = rem.graphalign (rem.py:318-382): the nodes under the match are broken and
 * merged, the graph walked around the merged node (segmentgraph, rem.py:228-316); counts[0..3] = leading / trailing / matching / rest intervals (rv_graph_align_fetch
 * hands them out back to back as (begin, end) pairs), out6 = merged node, new left node, new right node.  rv_set_graph_picker(h, g, args) makes rv_align_builtin call the
 * two per sub-index (picker kind 2): `reveal rem a.gfa b.gfa` with no Python call per sub-index. */
rv_graph *rv_graph_import(int64_t nnodes, const int64_t *node_b, const int64_t *node_e, const int8_t *node_aligned, const int8_t *node_sent,
                          const int64_t *off_ptr, const int32_t *off_sid, const int64_t *off_val,
                          int64_t nedges, const int32_t *edge_u, const int32_t *edge_v, const int64_t *edge_ptr, const int32_t *edge_paths,
                          const int64_t *pred_ptr, const int32_t *pred_edge, int npaths, const uint8_t *star, const int64_t *id2end,
                          int nstart, const int32_t *start_nodes, int literal_segments);
int rv_graph_pick(rv_graph *g, const rv_picker_args *args, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so, const int64_t *pos,
                  const int64_t *left, const int64_t *right, int minlength, rv_picker_out *out);
int rv_graph_align(rv_graph *g, const int64_t *nodes, int64_t nn, const int64_t *left, const int64_t *right, uint32_t l, const int64_t *pos, int npos,
                   int64_t *counts, int64_t *out6);
int rv_graph_align_fetch(rv_graph *g, int64_t *out);
/* Picker kind 2 of the built-in recursion (rv_align_builtin*): both callbacks inside the library, on the caller's graph of the inputs (rv_graph_import).  Per
 * sub-index, in the reference's order: rv_graph_pick on its match list (or the middle of the list its parent seeded), rv_graph_align for the choice; the children
 * get their left / right nodes as reveal.c:884-950 hands them on.  The graph is the alignment graph when the run returns (rv_graph_finish, then _prune / _gfa /
 * _export); it stays the caller's.  g = NULL turns the picker off.  Not combined with construct(rc=1), tracing or the frontier hand-off. */
int rv_set_graph_picker(rv_index *h, rv_graph *g, const rv_picker_args *args);
/* The readers of `reveal rem` behind the ABI (reveal/utils.py:304-375 read_fasta, :377-677 read_gfa with rem's defaults; reveal_amd/alngraph.py in Python): the
 * graph of the inputs made where the run uses it, the segments' text appended to the index on the way.
 * rv_graph_add_linear: one sequence the caller added to the index as [b, e) -- a new path (-> its id), its start sentinel, the node, its end sentinel.
 * rv_graph_read_gfa: the text of a GFA1 file; every S line becomes a sequence of h's current sample (h == NULL: intervals counted from *text_n on -- tests
 * without a device); -> number of paths added (*names: their names, one per line, valid until the next call), -1 error, -2 the file holds links on the reverse
 * strand (not supported behind the ABI; g and h are then half-filled: start over with the Python readers).
 * rv_graph_seal: once after the last input.  rv_graph_paths -> number of paths, their lengths; rv_graph_node_kinds: per node in rv_graph_export's order 0 sequence node, 1 start, 2 end sentinel. */
rv_graph *rv_graph_new(void);
int rv_graph_add_linear(rv_graph *g, int64_t b, int64_t e, int star);
int64_t rv_graph_read_gfa(rv_graph *g, rv_index *h, int64_t *text_n, const char *data, int64_t len, const char **names);
/* The same reader in two halves, so that the files of a job are parsed side by side: rv_gfa_parse (any thread; touches no graph and no index) -> an object that holds the
 * file's graph and text; rv_graph_adopt (in the order of the inputs) appends both to h and g and returns what rv_graph_read_gfa returns; rv_gfa_parsed_free. */
typedef struct GfaParsed rv_gfa_parsed;
rv_gfa_parsed *rv_gfa_parse(const char *data, int64_t len);
int64_t rv_graph_adopt(rv_graph *g, rv_index *h, int64_t *text_n, rv_gfa_parsed *parsed, const char **names);
void rv_gfa_parsed_free(rv_gfa_parsed *parsed);
int rv_graph_seal(rv_graph *g);      /* after the last rv_graph_read_gfa / rv_graph_add_linear, before the graph is used: renumbers it, decides rv_graph_literal */
int rv_graph_paths(const rv_graph *g, int64_t *id2end);
int rv_graph_node_kinds(const rv_graph *g, int8_t *out);
int rv_graph_literal(const rv_graph *g);      /* 1: some node of the inputs does not go on in both directions -- segmentgraph walks back from its end points as the reference does */
int rv_graph_finish(rv_graph *g);      /* renumber live nodes and links (after rv_graph_align, before rv_graph_sizes / rv_graph_export) */
/* The same surgery as a follower of the run that chooses the anchors: rv_graph_replay_begin = the graph of the sequences alone; rv_set_replay_graph(h, g) makes the
 * next rv_align_builtin (picker kind 1) apply every level's anchors to g on a host thread while the GPU works on the next level; when the run returns g is what
 * rv_graph_replay would have made of its anchors.  g stays the caller's; NULL: off. */
rv_graph *rv_graph_replay_begin(int nseq, const int64_t *begin, const int64_t *end);
int rv_set_replay_graph(rv_index *h, rv_graph *g);
rv_graph *rv_graph_replay(int nseq, const int64_t *begin, const int64_t *end, int64_t na, const uint32_t *an_l, const int64_t *an_off, const int64_t *an_pos);
const char *rv_graph_error(const rv_graph *g);
int rv_graph_sizes(const rv_graph *g, int64_t *out);
int rv_graph_export(const rv_graph *g, int64_t *node_b, int64_t *node_e, int8_t *node_aligned, int64_t *off_ptr, int32_t *off_sid, int64_t *off_val,
                    int64_t *succ_ptr, int32_t *succ_to, int32_t *succ_edge, int64_t *pred_ptr, int32_t *pred_from, int32_t *pred_edge, int64_t *edge_ptr, int32_t *edge_paths);
/* rv_graph_prune: `prune_nodes` (rem.py:384-447; what `reveal rem` does before writing when more than two paths took part) with T = the index text after
 * the run.  rv_graph_gfa: the GFA1 text `reveal rem` writes (utils.py:710-839: H, S + L per sequence node in creation order, P per path; names[s] = the name of
 * path s, cmdline for the header) -- *out points at it until the graph is freed, the return value is its length.  Both identical to the Python graph layer's
 * result (tests/test_cpu_graph_native.py). */
int rv_graph_prune(rv_graph *g, const char *T);
int64_t rv_graph_gfa(rv_graph *g, const char *T, int npaths, const char *const *names, const char *cmdline, const char **out);
void rv_graph_free(rv_graph *g);

/* ---- measurement ------------------------------------------------------------ */
/* HIP-event timing of the kernels on the handle's stream.  kernel ids: */
enum { RV_K_SCAN_PAIR = 0, RV_K_SCAN_MULTI = 1, RV_K_SA_SORT = 2, RV_K_LCP = 3, RV_K_SPLIT = 4, RV_K_LABEL = 5,
       RV_K_BUBBLE = 6,
       /* parts of RV_K_SA_SORT, timed on their own (bytes = what the kernel has to move: 2 x (8 B key + value) per pair and
        * scatter pass, 8 B per key and histogram pass; the text round's bytes are its list traffic) */
       RV_K_RADIX_SCATTER = 7, RV_K_RADIX_HIST = 8, RV_K_TEXT_ROUND = 9,
       RV_K_CASCADE = 10,      /* the anchor cascade behind its scan: witnesses, match sort, levels, rebuild + leaf launch */
       RV_K_DIAG_TABLE = 11,   /* part of RV_K_SA_SORT: piecewise diagonals from seeds (two samples that left their fixed diagonal: indels) */
       RV_K_INIT_KEYS = 12,    /* part of RV_K_SA_SORT: the first keys (bytes: the text once, a key + suffix + digit byte per kept suffix) */
       RV_K_PUBLISH = 13,      /* part of RV_K_SA_SORT: heads of the sorted list and the finished ranks' SA / LCP / BWT (bytes: key + suffix read per entry, key + suffix
                                  written in rank order, 9 B per position written) */
       RV_K_COUNT = 14 };
/* on: 0 = off, 1 = every class, otherwise bit k+1 selects class k (an event pair costs the stream a few
 * microseconds, so a timed run times only what it reports) */
int rv_prof_enable(rv_index *h, int on);
int rv_prof_reset(rv_index *h);
/* launches, total milliseconds and algorithmic bytes of kernel class k since the last reset */
int rv_prof_get(rv_index *h, int k, int64_t *launches, double *ms, double *bytes);
/* ---- a batch of independent alignments -----------------------------------------------------------------------------------------------------------
 * The reference's job-level parallelism is a shell script of independent `reveal rem` commands (reveal/align.py:27-54: the 20 / 4 / 1 jobs of
 * `--order=sequential --chunksize=5`).  On one GPU small jobs are bound by the chains of tiny launches in their anchor cascades, which streams of their
 * own hide from each other only in part; rv_batch_run runs the handles' construct + rv_align_builtin on a host thread each and lets the jobs with more
 * than two samples run the level loops of their cascades as ONE set of launches (every job's lists behind each other, a root per job).
 * Every handle's result is what rv_align_builtin would have given it alone (rv_fetch_anchors etc. as usual).  status / stats: one entry per handle, in
 * the order they were added (either may be NULL). */
typedef struct rv_batch rv_batch;
rv_batch *rv_batch_new(void);
int rv_batch_add(rv_batch *b, rv_index *h);
int rv_batch_run(rv_batch *b, int minl, int minn, int construct, rv_align_stats *stats, int *status);
int rv_batch_info(const rv_batch *b, int64_t *out);      /* out[0] = joint level loops run so far, out[1] = jobs they served */
void rv_batch_free(rv_batch *b);
/* ---- device memory for the frontier hand-off between processes ("child SA/LCP shipped once to the owner GPU, peer copy over
 * xGMI") -- what reveal_amd/shard.py's own transport uses instead of a tensor library: the owner packs the segments it hands out into buffers of its
 * device (rv_frontier_pack, on_device = 1) and exports them once (hipIpcGetMemHandle: 64 bytes that travel over any byte channel); a worker process
 * opens them on ITS device (hipIpcOpenMemHandle) and copies its batches out, device to device (hipMemcpy over the peer link).  No collective, no
 * library besides HIP.  rv_dev_alloc returns NULL, the others -1, with rv_last_error set. */
void *rv_dev_alloc(int device, int64_t bytes);
int rv_dev_free(int device, void *p);
int rv_ipc_export(int device, const void *p, uint8_t handle[64]);     /* p: what rv_dev_alloc returned */
void *rv_ipc_open(int device, const uint8_t handle[64]);
int rv_ipc_close(int device, void *p);
int rv_dev_copy(int device, void *dst, const void *src, int64_t bytes);      /* any two device (or host) addresses; returns when the copy is done */
/* The practical HBM ceiling of this device ("also report measured copy-kernel bandwidth on the node"): a
 * streaming read kernel and a copy kernel over `bytes` of freshly allocated memory, `iters` timed launches each, HIP events.
 * GB/s; the copy figure counts bytes read + bytes written. */
int rv_measure_bandwidth(int device, int64_t bytes, int iters, double *read_gbs, double *copy_gbs);
/* SA-build statistics of the last rv_construct */
int rv_sa_stats(rv_index *h, int *sigma, int *bits, int *k0, int *rounds, int64_t *sorted_elems, int *radix_passes);
int rv_sa_diag_table(rv_index *h);      /* 1: two samples on piecewise diagonals from seeds in the last construct() (rv_construct.hip k_diag_bits_tab) */
/* what finished the suffixes the first key and the text round left tied (agreement beyond 4 KB: near-identical inputs, repeats): out[0] = tied
 * pairs of partners ordered from the diagonal's marks, out[1] = ranks whose LCP / BWT came from the text after the doubling rounds */
int rv_sa_tail(rv_index *h, int64_t *out);

/* ---- self-test hooks for the device primitives (tests/ only) ------------------ */
int rv_test_exclusive_sum_u32(const uint32_t *in, uint32_t *out, int64_t n);
int rv_test_inclusive_max_u32(const uint32_t *in, uint32_t *out, int64_t n);
int rv_test_radix_sort(uint64_t *keys, uint32_t *vals, int64_t n, int bit_lo, int bit_hi);
/* the same sort timed on device-made keys (HIP events on its stream): dist 0 uniform, 1 first keys of a DNA text (base-5 digits 1..4),
 * 2 constant; flags: 1 = 10-bit digits, 2 = XCD-aware tile order, 4 = 16-bit wave counters, 8 = histograms from the keys (no digit bytes);
 * *bad = pairs out of order afterwards */
int rv_test_radix_time(int64_t n, int bits, int dist, int flags, int iters, double *ms, int64_t *bad);

#ifdef __cplusplus
}
#endif
#endif
