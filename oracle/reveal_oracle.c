/*
 * reveal_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, see reveal_oracle.h).
 *
 * Plain-C restatement of jasperlinthorst/reveal's reveallib hot path.  Each
 * function cites the reference lines it follows.  Pinned against the
 * reference's own C built unmodified into oracle/_ref (oracle/pin_oracle.py).
 */
#define _GNU_SOURCE
#include "reveal_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <assert.h>
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* order-sensitive 64-bit hash of an integer sequence, vectorisable in numpy:
 * h = sum_i mix(v_i + (i+1)*GOLDEN) mod 2^64, mix = splitmix64 finaliser. */
uint64_t ro_hash_step(uint64_t acc, uint64_t i, int64_t v) {
    uint64_t x = (uint64_t)v + (i + 1) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return acc + x;
}

/* ------------------------------------------------------------------------ */
/* Suffix array                                                             */
/* ------------------------------------------------------------------------ */

static int (*g_divsufsort)(const uint8_t *, ro_saidx_t *, ro_saidx_t) = NULL;
void ro_set_divsufsort(int (*fn)(const uint8_t *, ro_saidx_t *, ro_saidx_t)) { g_divsufsort = fn; }

/* Own sorter: prefix doubling (Manber-Myers / Larsson-Sadakane style) with
 * group discarding.  Replaces divsufsort (interface.c:215-222): a suffix
 * array is unique, so the result is bit-identical. */
typedef struct { const ro_saidx_t *isa; ro_saidx_t h, n; } pd_ctx;
static inline int64_t pd_key(const pd_ctx *c, ro_saidx_t x) {
    return ((int64_t)x + c->h < (int64_t)c->n) ? (int64_t)c->isa[x + c->h] : -1;
}
static int pd_cmp(const void *a, const void *b, void *ctx) {
    const pd_ctx *c = (const pd_ctx *)ctx;
    int64_t ka = pd_key(c, *(const ro_saidx_t *)a), kb = pd_key(c, *(const ro_saidx_t *)b);
    return (ka > kb) - (ka < kb);
}

int ro_suffix_array_own(const uint8_t *T, ro_saidx_t *SA, ro_saidx_t n) {
    if (n <= 0) return n == 0 ? 0 : -1;
    ro_saidx_t *isa = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)n);
    ro_saidx_t *nw  = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)n);
    const int NB = 256 * 257;
    int64_t *cnt = (int64_t *)calloc((size_t)NB + 1, sizeof(int64_t));
    if (!isa || !nw || !cnt) { free(isa); free(nw); free(cnt); return -2; }
#define K2(i) ((int)T[i] * 257 + (((i) + 1 < n) ? (int)T[(i) + 1] + 1 : 0))
    for (ro_saidx_t i = 0; i < n; i++) cnt[K2(i) + 1]++;
    for (int b = 0; b < NB; b++) cnt[b + 1] += cnt[b];
    /* group start per bucket is cnt[b] before filling */
    int64_t *start = (int64_t *)malloc(sizeof(int64_t) * (size_t)NB);
    memcpy(start, cnt, sizeof(int64_t) * (size_t)NB);
    for (ro_saidx_t i = 0; i < n; i++) { int k = K2(i); isa[i] = (ro_saidx_t)start[k]; SA[cnt[k]++] = i; }
    free(start); free(cnt);
#undef K2
    pd_ctx c; c.isa = isa; c.n = n;
    for (ro_saidx_t h = 2;; h *= 2) {
        c.h = h;
        int any = 0;
        ro_saidx_t s = 0;
        while (s < n) {
            ro_saidx_t e = s + 1;
            while (e < n && isa[SA[e]] == s) e++;
            if (e - s > 1) {
                any = 1;
                qsort_r(SA + s, (size_t)(e - s), sizeof(ro_saidx_t), pd_cmp, &c);
                ro_saidx_t g = s;
                nw[SA[s]] = s;
                for (ro_saidx_t j = s + 1; j < e; j++) {
                    if (pd_key(&c, SA[j]) != pd_key(&c, SA[j - 1])) g = j;
                    nw[SA[j]] = g;
                }
            } else {
                nw[SA[s]] = s;
            }
            s = e;
        }
        /* publish new ranks only after every group used the old ones */
        memcpy(isa, nw, sizeof(ro_saidx_t) * (size_t)n);
        if (!any) break;
        if ((int64_t)h * 2 > (int64_t)n * 2) break;
    }
    free(isa); free(nw);
    return 0;
}

int ro_suffix_array(const uint8_t *T, ro_saidx_t *SA, ro_saidx_t n) {
    if (g_divsufsort) return g_divsufsort(T, SA, n);
    return ro_suffix_array_own(T, SA, n);
}

/* sufcheck (divsufsort/utils.c:161) equivalent: permutation + sortedness. */
int ro_sufcheck(const uint8_t *T, const ro_saidx_t *SA, ro_saidx_t n) {
    uint8_t *seen = (uint8_t *)calloc((size_t)n + 1, 1);
    for (ro_saidx_t i = 0; i < n; i++) {
        if (SA[i] < 0 || SA[i] >= n || seen[SA[i]]) { free(seen); return -1; }
        seen[SA[i]] = 1;
    }
    free(seen);
    for (ro_saidx_t i = 1; i < n; i++) {
        ro_saidx_t a = SA[i - 1], b = SA[i];
        ro_saidx_t la = n - a, lb = n - b, m = la < lb ? la : lb;
        int r = memcmp(T + a, T + b, (size_t)m);
        if (r > 0 || (r == 0 && la > lb)) return -2;
    }
    return 0;
}

/* interface.c:236-238 */
void ro_inverse(const ro_saidx_t *SA, ro_saidx_t *SAi, ro_saidx_t n) {
    for (ro_saidx_t i = 0; i < n; i++) SAi[SA[i]] = i;
}

/* interface.c:97-114 (Kasai with stop characters).  T must be NUL-terminated
 * at T[n] like the reference's text (interface.c:84). */
void ro_compute_lcp(const char *T, const ro_saidx_t *SA, const ro_saidx_t *SAi, ro_lcp_t *LCP, ro_saidx_t n) {
    ro_lcp_t h = 0;
    ro_saidx_t i, j, k;
    for (i = 0; i < n; i++) {
        k = SAi[i];
        if (k == 0) {
            LCP[k] = 0;
        } else {
            j = SA[k - 1];
            while ((i - (ro_saidx_t)h < n) && (j + (ro_saidx_t)h < n) && (T[i + h] == T[j + h]) &&
                   T[i + h] != '$' && T[i + h] != 'N') ++h;
            LCP[k] = h;
        }
        if (h > 0) --h;
    }
}

/* interface.c:116-134 */
void ro_build_so(uint16_t *SO, const ro_saidx_t *nsep, int nsamples, ro_saidx_t n) {
    ro_saidx_t j;
    for (int i = 0; i < nsamples; i++) {
        if (i == 0)                 { for (j = 0; j <= nsep[i]; j++) SO[j] = (uint16_t)i; }
        else if (i == nsamples - 1) { for (j = nsep[i - 1] + 1; j < n; j++) SO[j] = (uint16_t)i; }
        else                        { for (j = nsep[i - 1] + 1; j <= nsep[i]; j++) SO[j] = (uint16_t)i; }
    }
}

/* interface.c:136-158: IUPAC-aware complement table, '$' and everything
 * outside the letters map to themselves. */
static char comp_of(char ch) {
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";   /* complement of 'A'..'Z' */
    unsigned char c = (unsigned char)ch;
    if (c >= 'A' && c <= 'Z') return up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (char)(up[c - 'a'] + 32);
    if (c == 96) return 64;            /* the table's one oddity: '`' -> '@' (interface.c:143) */
    return ch;
}
void ro_revcomp(char *T, ro_saidx_t n) {
    for (ro_saidx_t i = 0; i < n >> 1; ++i) {
        char c0 = comp_of(T[i]), c1 = comp_of(T[n - 1 - i]);
        T[i] = c1; T[n - 1 - i] = c0;
    }
    if (n & 1) T[n >> 1] = comp_of(T[n >> 1]);
}

/* ------------------------------------------------------------------------ */
/* Scans                                                                    */
/* ------------------------------------------------------------------------ */

static inline int is_lower(char c) { return c >= 'a' && c <= 'z'; }

/* reveal.c:55-116 (rem==0) and reveal.c:119-180 (rem!=0) */
int64_t ro_getmums(const ro_view *v, int minl, int rem,
                   ro_lcp_t *out_l, ro_saidx_t *out_a, ro_saidx_t *out_b, int64_t cap) {
    int64_t cnt = 0;
    ro_saidx_t i, aStart, bStart;
    ro_lcp_t lb, la;
    for (i = 1; i < v->n; i++) {
        if (v->LCP[i] < (ro_lcp_t)minl) continue;
        if ((v->SA[i] > v->nsep[0]) == (v->SA[i - 1] > v->nsep[0])) continue;     /* repeat */
        if (v->SA[i] < v->SA[i - 1]) { aStart = v->SA[i]; bStart = v->SA[i - 1]; }
        else                         { aStart = v->SA[i - 1]; bStart = v->SA[i]; }
        if (aStart > 0 && bStart > 0) {
            char ca = v->T[aStart - 1];
            if (!((ca != v->T[bStart - 1]) || ca == 'N' || ca == '$' || is_lower(ca))) continue;  /* not maximal */
        }
        lb = v->LCP[i - 1];
        la = (i == v->n - 1) ? 0 : v->LCP[i + 1];
        if (lb >= v->LCP[i] || la >= v->LCP[i]) continue;                          /* not unique */
        if (v->rc == 1) bStart = v->nsep[0] + ((rem ? v->n : v->nT) - bStart - (ro_saidx_t)v->LCP[i]);
        if (cnt < cap) { out_l[cnt] = v->LCP[i]; out_a[cnt] = aStart; out_b[cnt] = bStart; }
        cnt++;
    }
    return cnt;
}

/* sample of a text position: SO when present, else the two-sample rule */
static inline int sample_of(const ro_view *v, ro_saidx_t pos) {
    if (v->SO) return v->SO[pos];
    return pos > v->nsep[0] ? 1 : 0;
}

/* reveal.c:227-259 */
static int ismultimum(const ro_view *v, ro_lcp_t l, ro_saidx_t lb, ro_saidx_t ub, int *flag_so) {
    if (l > 0) {
        ro_saidx_t j;
        memset(flag_so, 0, (size_t)v->main_nsamples * sizeof(int));
        if (v->main_nsamples == 2) {
            if ((v->SA[ub] > v->nsep[0]) == (v->SA[lb] > v->nsep[0])) return 0;
        } else {
            for (j = lb; j < ub + 1; j++) {
                if (flag_so[v->SO[v->SA[j]]] == 0) flag_so[v->SO[v->SA[j]]] = 1;
                else return 0;
            }
        }
        for (j = lb; j < ub; j++) {
            if (v->SA[j] == 0) return 1;
            if (v->SA[j + 1] == 0) return 1;
            char c = v->T[v->SA[j] - 1];
            if (c != v->T[v->SA[j + 1] - 1] || c == 'N' || c == '$' || is_lower(c)) return 1;
        }
    }
    return 0;
}

/* reveal.c:261-290 */
static int ismultimem(const ro_view *v, ro_lcp_t l, ro_saidx_t lb, ro_saidx_t ub, int *flag_so) {
    if (l > 0) {
        ro_saidx_t j;
        memset(flag_so, 0, (size_t)v->main_nsamples * sizeof(int));
        if (v->main_nsamples == 2) {
            flag_so[(v->SA[ub] > v->nsep[0]) == (v->SA[lb] > v->nsep[0])]++;
        } else {
            for (j = lb; j < ub + 1; j++) flag_so[v->SO[v->SA[j]]]++;
        }
        for (j = lb; j < ub; j++) {
            if (v->SA[j] == 0) return 1;
            if (v->SA[j + 1] == 0) return 1;
            char c = v->T[v->SA[j] - 1];
            if (c != v->T[v->SA[j + 1] - 1] || c == 'N' || c == '$' || is_lower(c)) return 1;
        }
    }
    return 0;
}

typedef struct {
    const ro_view *v; int minl, minn, mems; int *flag_so;
    ro_lcp_t *out_l; int32_t *out_n; int64_t *out_off; int64_t cap_matches;
    uint16_t *out_so; ro_saidx_t *out_pos; int64_t cap_members;
    int64_t nmatch, nmemb;
} mm_ctx;

/* body of the `while` at reveal.c:468-507 (mums) / :323-363 (mems), also the
 * final flush reveal.c:538-574 / :391-428 */
/* returns 1 when the reference `continue`s out of the while body
 * (reveal.c:340-342, mems only), which skips its `lb = i_lb` (reveal.c:362). */
static int mm_close(mm_ctx *c, ro_lcp_t i_lcp, ro_saidx_t i_lb, ro_saidx_t i_ub) {
    const ro_view *v = c->v;
    int64_t n = (int64_t)(i_ub - i_lb) + 1;
    int count_field;
    if (!(i_lcp >= (ro_lcp_t)c->minl)) return 0;
    if (c->mems) {
        if (!(n >= c->minn)) return 0;
        if (ismultimem(v, i_lcp, i_lb, i_ub, c->flag_so) != 1) return 0;
        int cc = 0;
        for (int ci = 0; ci < v->main_nsamples; ci++) if (c->flag_so[ci] > 0) cc++;
        if (cc < c->minn) return 1;
        count_field = cc;
    } else {
        if (!(n <= v->main_nsamples && n >= c->minn)) return 0;
        if (ismultimum(v, i_lcp, i_lb, i_ub, c->flag_so) != 1) return 0;
        count_field = (int)n;
    }
    if (c->nmatch < c->cap_matches) {
        c->out_l[c->nmatch] = i_lcp;
        c->out_n[c->nmatch] = count_field;
        c->out_off[c->nmatch] = c->nmemb;
    }
    for (int64_t x = 0; x < n; x++) {
        if (c->nmemb < c->cap_members && c->nmatch < c->cap_matches) {
            c->out_so[c->nmemb]  = (uint16_t)sample_of(v, v->SA[i_lb + x]);
            c->out_pos[c->nmemb] = v->SA[i_lb + x];
        }
        c->nmemb++;
    }
    c->nmatch++;
    return 0;
}

/* reveal.c:436-580 (mums) and reveal.c:292-434 (mems): LCP-interval stack */
int64_t ro_getmultimums(const ro_view *v, int minl, int minn, int mems,
                        ro_lcp_t *out_l, int32_t *out_n, int64_t *out_off, int64_t cap_matches,
                        uint16_t *out_so, ro_saidx_t *out_pos, int64_t cap_members,
                        int64_t *members_needed) {
    int maxdepth = 1000;
    int *flag_so = (int *)calloc((size_t)(v->main_nsamples > 2 ? v->main_nsamples : 2), sizeof(int));
    ro_lcp_t   *stack_lcp = (ro_lcp_t *)malloc((size_t)maxdepth * sizeof *stack_lcp);
    ro_saidx_t *stack_lb  = (ro_saidx_t *)malloc((size_t)maxdepth * sizeof *stack_lb);
    ro_saidx_t *stack_ub  = (ro_saidx_t *)malloc((size_t)maxdepth * sizeof *stack_ub);
    mm_ctx c = { v, minl, minn, mems, flag_so, out_l, out_n, out_off, cap_matches,
                 out_so, out_pos, cap_members, 0, 0 };
    int depth = 0;
    ro_saidx_t i, lb, i_lb, i_ub;
    ro_lcp_t i_lcp;
    stack_lcp[0] = 0; stack_lb[0] = 0; stack_ub[0] = 0;
    for (i = 1; i < v->n; i++) {
        lb = i - 1;
        while (v->LCP[i] < stack_lcp[depth]) {
            stack_ub[depth] = i - 1;
            i_lcp = stack_lcp[depth]; i_lb = stack_lb[depth]; i_ub = stack_ub[depth];
            depth--;
            if (mm_close(&c, i_lcp, i_lb, i_ub)) continue;   /* reference quirk: lb not updated */
            lb = i_lb;
        }
        if (v->LCP[i] > stack_lcp[depth]) {
            depth++;
            if (depth >= maxdepth) {
                maxdepth += 1000;
                stack_lcp = (ro_lcp_t *)realloc(stack_lcp, (size_t)maxdepth * sizeof *stack_lcp);
                stack_lb  = (ro_saidx_t *)realloc(stack_lb, (size_t)maxdepth * sizeof *stack_lb);
                stack_ub  = (ro_saidx_t *)realloc(stack_ub, (size_t)maxdepth * sizeof *stack_ub);
            }
            stack_lcp[depth] = v->LCP[i];
            stack_lb[depth] = lb;
            stack_ub[depth] = 0;
        }
    }
    while (depth >= 0) {
        stack_ub[depth] = v->n - 1;
        i_lcp = stack_lcp[depth]; i_lb = stack_lb[depth]; i_ub = stack_ub[depth];
        depth--;
        mm_close(&c, i_lcp, i_lb, i_ub);
    }
    free(stack_lcp); free(stack_lb); free(stack_ub); free(flag_so);
    if (c.nmatch <= cap_matches && out_off) out_off[c.nmatch < cap_matches ? c.nmatch : cap_matches] = c.nmemb;
    if (members_needed) *members_needed = c.nmemb;
    if (c.nmatch > cap_matches || c.nmemb > cap_members) return -c.nmatch - 1;
    return c.nmatch;
}

/* ------------------------------------------------------------------------ */
/* D-label, split, bubble_sort                                              */
/* ------------------------------------------------------------------------ */

/* reveal.c:1005-1117 */
void ro_label(uint8_t *D, ro_saidx_t n, const ro_saidx_t *SAi,
              const ro_saidx_t *lead, int nlead, const ro_saidx_t *trail, int ntrail,
              const ro_saidx_t *rest, int nrest,
              const ro_saidx_t *sp, int nsp, ro_lcp_t l,
              ro_saidx_t *nl, ro_saidx_t *nt, ro_saidx_t *np) {
    ro_saidx_t j, leadingn = 0, trailingn = 0, parn = 0;
    memset(D, 0, (size_t)n);
    for (int k = 0; k < nlead; k++)  for (j = lead[2 * k];  j < lead[2 * k + 1];  j++) { D[SAi[j]] = 1; leadingn++; }
    for (int k = 0; k < ntrail; k++) for (j = trail[2 * k]; j < trail[2 * k + 1]; j++) { D[SAi[j]] = 2; trailingn++; }
    for (int k = 0; k < nrest; k++)  for (j = rest[2 * k];  j < rest[2 * k + 1];  j++) { D[SAi[j]] = 4; parn++; }
    for (int k = 0; k < nsp; k++)    for (j = sp[k]; j < sp[k] + (ro_saidx_t)l; j++) D[SAi[j]] = 3;
    if (nl) *nl = leadingn;
    if (nt) *nt = trailingn;
    if (np) *np = parn;
}

/* reveal.c:582-664 */
void ro_split(const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, const uint8_t *D,
              ro_saidx_t *SAi,
              ro_saidx_t *lSA, ro_lcp_t *lLCP, ro_saidx_t *tSA, ro_lcp_t *tLCP,
              ro_saidx_t *pSA, ro_lcp_t *pLCP,
              ro_saidx_t *out_il, ro_saidx_t *out_it, ro_saidx_t *out_ip) {
    ro_saidx_t i = 0, ip = 0, il = 0, it = 0, lastp = 0, lastl = 0, lastt = 0;
    ro_lcp_t minlcpp = 0, minlcpl = 0, minlcpt = 0;
    for (i = 0; i < n; i++) {
        if (D[i] == 1) {
            lSA[il] = SA[i];
            lLCP[il] = (il == 0) ? 0 : minlcpl;
            SAi[SA[i]] = il;
            il++; lastl = i;
        } else if (D[i] == 2) {
            tSA[it] = SA[i];
            tLCP[it] = (it == 0) ? 0 : minlcpt;
            SAi[SA[i]] = it;
            it++; lastt = i;
        } else {
            if (D[i] == 3) {
                /* matched suffix: dropped */
            } else {
                if (D[i] != 4) continue;        /* note: skips the min updates, as the reference does */
                pSA[ip] = SA[i];
                pLCP[ip] = (ip == 0) ? 0 : minlcpp;
                SAi[SA[i]] = ip;
                ip++; lastp = i;
            }
        }
        if (i == n - 1) break;
        if (i == lastt) minlcpt = LCP[i + 1]; else if (LCP[i + 1] < minlcpt) minlcpt = LCP[i + 1];
        if (i == lastl) minlcpl = LCP[i + 1]; else if (LCP[i + 1] < minlcpl) minlcpl = LCP[i + 1];
        if (i == lastp) minlcpp = LCP[i + 1]; else if (LCP[i + 1] < minlcpp) minlcpp = LCP[i + 1];
    }
    if (out_il) *out_il = il;
    if (out_it) *out_it = it;
    if (out_ip) *out_ip = ip;
}

/* reveal.c:666-727.  The reference writes LCP[x+1] / LCP[i+1] one past the
 * array when n==1 (harmless heap slack there); guarded here. */
void ro_bubble_sort(ro_saidx_t *SA, ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                    const ro_saidx_t *match_begin, int nmatch) {
    ro_lcp_t tmpLCP;
    ro_saidx_t i, x, tmpSA, begin;
    for (int m = 0; m < nmatch; m++) {
        begin = match_begin[m];
        for (i = 0; i < n; i++) {
            if ((SA[i] < begin) && (((int64_t)SA[i] + (int64_t)LCP[i]) > (int64_t)begin)) {
                x = i; tmpSA = SA[i]; tmpLCP = LCP[i];
                while (((int64_t)LCP[x] >= (int64_t)(begin - tmpSA)) && (x > 0)) {
                    SAi[SA[x - 1]] = x;
                    SA[x] = SA[x - 1];
                    LCP[x] = LCP[x - 1];
                    x--;
                }
                SAi[tmpSA] = x;
                SA[x] = tmpSA;
                if (x + 1 < n) LCP[x + 1] = (ro_lcp_t)(begin - tmpSA);
                if (i < n - 1) {
                    if (tmpLCP < LCP[i + 1]) LCP[i + 1] = tmpLCP;
                }
            } else {
                if (i < n - 1) {
                    if ((SA[i] < begin) && (((int64_t)SA[i] + (int64_t)LCP[i + 1]) > (int64_t)begin)) {
                        if (LCP[i + 1] > LCP[i]) LCP[i + 1] = (ro_lcp_t)(begin - SA[i]);
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* host-driven single steps: splitindex / extract (reveal.c:1386-1748)       */
/* ------------------------------------------------------------------------ */

/* sample count of an interval list, reveal.c:1552-1567 (the same rule as the aligner's, reveal.c:1028-1042) */
static int count_samples_raw(const ro_saidx_t *iv, int niv, const uint16_t *SO, const ro_saidx_t *nsep, int main_nsamples) {
    int ns = 0;
    int *flag = (int *)calloc((size_t)(main_nsamples > 2 ? main_nsamples : 2), sizeof(int));
    for (int k = 0; k < niv; k++) {
        ro_saidx_t begin = iv[2 * k];
        if (main_nsamples > 2) {
            if (flag[SO[begin]] == 0) { flag[SO[begin]] = 1; ns++; }
        } else {
            if (begin < nsep[0] && flag[0] == 0) { flag[0] = 1; ns++; }
            if (begin > nsep[0] && flag[1] == 0) { flag[1] = 1; ns++; }
        }
    }
    free(flag);
    return ns;
}

/* splitindex (reveal.c:1515-1748): D-label from the four interval lists in the
 * reference's order (leading 1, trailing 2, matching 3 + lower-casing, rest 4),
 * split, bubble_sort of the leading child over the matching intervals in the
 * order given.  Child arrays are caller-allocated (sum of the interval lengths
 * each); counts[3] / nsamples[3] receive leadingn/trailingn/parn and the
 * children's sample counts. */
void ro_splitindex(char *T, const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                   const uint16_t *SO, const ro_saidx_t *nsep, int main_nsamples,
                   const ro_saidx_t *lead, int nlead, const ro_saidx_t *trail, int ntrail,
                   const ro_saidx_t *match, int nmatch, const ro_saidx_t *rest, int nrest,
                   ro_saidx_t *lSA, ro_lcp_t *lLCP, ro_saidx_t *tSA, ro_lcp_t *tLCP, ro_saidx_t *pSA, ro_lcp_t *pLCP,
                   ro_saidx_t *counts, int *nsamples) {
    uint8_t *D = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    ro_saidx_t j, leadingn = 0, trailingn = 0, parn = 0;
    for (int k = 0; k < nlead; k++)  for (j = lead[2 * k];  j < lead[2 * k + 1];  j++) { D[SAi[j]] = 1; leadingn++; }     /* :1548-1551 */
    for (int k = 0; k < ntrail; k++) for (j = trail[2 * k]; j < trail[2 * k + 1]; j++) { D[SAi[j]] = 2; trailingn++; }   /* :1583-1586 */
    for (int k = 0; k < nmatch; k++) for (j = match[2 * k]; j < match[2 * k + 1]; j++) {                                 /* :1618-1621 */
        D[SAi[j]] = 3;
        if (T[j] >= 'A' && T[j] <= 'Z') T[j] = (char)(T[j] + 32);
    }
    for (int k = 0; k < nrest; k++)  for (j = rest[2 * k];  j < rest[2 * k + 1];  j++) { D[SAi[j]] = 4; parn++; }        /* :1637-1640 */
    counts[0] = leadingn; counts[1] = trailingn; counts[2] = parn;
    nsamples[0] = count_samples_raw(lead, nlead, SO, nsep, main_nsamples);
    nsamples[1] = count_samples_raw(trail, ntrail, SO, nsep, main_nsamples);
    nsamples[2] = count_samples_raw(rest, nrest, SO, nsep, main_nsamples);
    ro_saidx_t il = 0, it = 0, ip = 0;
    ro_split(SA, LCP, n, D, SAi, lSA, lLCP, tSA, tLCP, pSA, pLCP, &il, &it, &ip);                                        /* :1738 */
    if (leadingn > 0) {                                                                                                   /* :1740-1742 */
        ro_saidx_t *mb = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)(nmatch > 0 ? nmatch : 1));
        for (int k = 0; k < nmatch; k++) mb[k] = match[2 * k];
        ro_bubble_sort(lSA, lLCP, il, SAi, mb, nmatch);
        free(mb);
    }
    free(D);
}

/* extract (reveal.c:1386-1505): drop the suffixes of the given intervals from
 * the index in place (here: into oSA / oLCP, n - matching entries), lower-case
 * them, bubble_sort over the intervals.  With rc==1 query-side intervals are
 * remapped first (:1411-1427; the remapped pairs are written back to
 * `intervals`, as the reference replaces the list items).
 * The reference never writes _SA[0] (its loop starts at i=1, j=1; :1454-1460):
 * the entry is uninitialised heap there.  Here it is SA[0], the evident intent
 * (rank 0 stays rank 0); pinning compares ranks 1.. and SAi.  When rank 0
 * itself is matched the reference writes one entry past both buffers: -1 here.
 * Returns the new n. */
ro_saidx_t ro_extract(char *T, const ro_saidx_t *SA, const ro_lcp_t *LCP, ro_saidx_t n, ro_saidx_t *SAi,
                      const ro_saidx_t *nsep, ro_saidx_t nT, int rc,
                      ro_saidx_t *intervals, int niv, ro_saidx_t *oSA, ro_lcp_t *oLCP) {
    uint8_t *D = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    ro_saidx_t i, j, matching = 0;
    for (int k = 0; k < niv; k++) {
        ro_saidx_t begin = intervals[2 * k], end = intervals[2 * k + 1];
        if (rc == 1 && begin > nsep[0]) {
            ro_saidx_t b2 = nsep[0] + (nT - begin - (end - begin));
            ro_saidx_t e2 = nsep[0] + (nT - begin);
            begin = b2; end = e2;
            intervals[2 * k] = begin; intervals[2 * k + 1] = end;
        }
        for (j = begin; j < end; j++) {
            D[SAi[j]] = 3;
            if (T[j] >= 'A' && T[j] <= 'Z') T[j] = (char)(T[j] + 32);
            matching++;
        }
    }
    if (n > 0 && D[0] == 3) { free(D); return -1; }
    ro_lcp_t minlcp = 0;
    j = 1;
    oLCP[0] = 0;
    oSA[0] = SA[0];
    for (i = 1; i < n; i++) {
        if (D[i] != 3) {
            oSA[j] = SA[i];
            SAi[oSA[j]] = j;
            if (D[i - 1] == 3) oLCP[j] = (minlcp < LCP[i]) ? minlcp : LCP[i];
            else oLCP[j] = LCP[i];
            j++;
        } else {
            if (D[i - 1] != 3) minlcp = LCP[i];
            else if (LCP[i] < minlcp) minlcp = LCP[i];
        }
    }
    free(D);
    ro_saidx_t nn = n - matching;
    ro_saidx_t *mb = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)(niv > 0 ? niv : 1));
    for (int k = 0; k < niv; k++) mb[k] = intervals[2 * k];
    ro_bubble_sort(oSA, oLCP, nn, SAi, mb, niv);                                                                          /* :1496 */
    free(mb);
    return nn;
}

/* ------------------------------------------------------------------------ */
/* aligner loop (reveal.c:731-1338; queue reveal.c:18-53)                   */
/* ------------------------------------------------------------------------ */

static void free_index(ro_index *ix) {
    if (!ix) return;
    free(ix->SA); free(ix->LCP); free(ix->nodes); free(ix);
}

static void free_mumlist(ro_mumlist *ml) {
    free(ml->l); free(ml->n); free(ml->off); free(ml->so); free(ml->pos);
    memset(ml, 0, sizeof *ml);
}

/* scan dispatch of reveal.c:809-822 */
static int scan_index(const ro_index *ix, int minl, int minn, ro_mumlist *ml) {
    ro_main *m = ix->main;
    ro_view v; v.T = m->T; v.SA = ix->SA; v.LCP = ix->LCP; v.SO = m->SO; v.nsep = m->nsep;
    v.n = ix->n; v.nT = m->nT; v.main_nsamples = m->nsamples; v.rc = 0;
    memset(ml, 0, sizeof *ml);
    if (m->nsamples > 2) {
        int64_t capm = 1024, capp = 4096, need = 0;
        for (;;) {
            ml->l = (ro_lcp_t *)realloc(ml->l, sizeof(ro_lcp_t) * (size_t)capm);
            ml->n = (int32_t *)realloc(ml->n, sizeof(int32_t) * (size_t)capm);
            ml->off = (int64_t *)realloc(ml->off, sizeof(int64_t) * (size_t)(capm + 1));
            ml->so = (uint16_t *)realloc(ml->so, sizeof(uint16_t) * (size_t)capp);
            ml->pos = (ro_saidx_t *)realloc(ml->pos, sizeof(ro_saidx_t) * (size_t)capp);
            int64_t r = ro_getmultimums(&v, minl, minn, 0, ml->l, ml->n, ml->off, capm, ml->so, ml->pos, capp, &need);
            if (r >= 0) { ml->count = r; break; }
            capm = -r - 1 + 16; capp = need + 16;
        }
    } else {
        int64_t cap = 1024;
        ro_saidx_t *a = NULL, *b = NULL;
        int64_t r;
        for (;;) {
            ml->l = (ro_lcp_t *)realloc(ml->l, sizeof(ro_lcp_t) * (size_t)cap);
            a = (ro_saidx_t *)realloc(a, sizeof(ro_saidx_t) * (size_t)cap);
            b = (ro_saidx_t *)realloc(b, sizeof(ro_saidx_t) * (size_t)cap);
            r = ro_getmums(&v, minl, 1, ml->l, a, b, cap);
            if (r <= cap) break;
            cap = r;
        }
        ml->count = r;
        ml->n = (int32_t *)malloc(sizeof(int32_t) * (size_t)(r + 1));
        ml->off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r + 1));
        ml->so = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(2 * r + 2));
        ml->pos = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)(2 * r + 2));
        for (int64_t k = 0; k < r; k++) {       /* (l, 2, ((0,a),(1,b)))  reveal.c:166-170 */
            ml->n[k] = 2; ml->off[k] = 2 * k;
            ml->so[2 * k] = 0; ml->pos[2 * k] = a[k];
            ml->so[2 * k + 1] = 1; ml->pos[2 * k + 1] = b[k];
        }
        ml->off[r] = 2 * r;
        free(a); free(b);
    }
    return 0;
}

/* child sample count, reveal.c:1028-1042 */
static int count_samples(const ro_main *m, const ro_intv *iv, int niv) {
    int ns = 0;
    int *flag = (int *)calloc((size_t)(m->nsamples > 2 ? m->nsamples : 2), sizeof(int));
    for (int k = 0; k < niv; k++) {
        ro_saidx_t begin = iv[k].begin;
        if (m->nsamples > 2) {
            if (flag[m->SO[begin]] == 0) { flag[m->SO[begin]] = 1; ns++; }
        } else {
            if (begin < m->nsep[0] && flag[0] == 0) { flag[0] = 1; ns++; }
            if (begin > m->nsep[0] && flag[1] == 0) { flag[1] = 1; ns++; }
        }
    }
    free(flag);
    return ns;
}

static ro_index *new_child(ro_main *m, ro_saidx_t n, int depth, int nsamples, ro_intv *nodes, int nnodes) {
    ro_index *c = (ro_index *)calloc(1, sizeof *c);
    c->main = m; c->n = n; c->depth = depth; c->nsamples = nsamples;
    c->SA = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)(n > 0 ? n : 1));
    c->LCP = (ro_lcp_t *)malloc(sizeof(ro_lcp_t) * (size_t)(n > 0 ? n : 1));
    c->nodes = nodes; c->nnodes = nnodes;
    return c;
}

int ro_align(ro_main *m, ro_saidx_t *SA, ro_lcp_t *LCP, ro_saidx_t n,
             const ro_intv *nodes, int nnodes,
             ro_picker_fn picker, ro_graphalign_fn galign, void *user,
             int minl, int minn,
             ro_trace *trace, int64_t trace_cap,
             ro_align_stats *stats) {
    int64_t qcap = 10000, qsize = 0;      /* QUEUE_BUF reveal.h:3 */
    ro_index **queue = (ro_index **)malloc(sizeof(ro_index *) * (size_t)qcap);
    ro_align_stats st; memset(&st, 0, sizeof st);
    int err = 0;

    ro_index *root = (ro_index *)calloc(1, sizeof *root);
    root->main = m; root->SA = SA; root->LCP = LCP; root->n = n; root->depth = 0;
    root->nsamples = m->nsamples;
    root->nodes = (ro_intv *)malloc(sizeof(ro_intv) * (size_t)(nnodes > 0 ? nnodes : 1));
    memcpy(root->nodes, nodes, sizeof(ro_intv) * (size_t)nnodes);
    root->nnodes = nnodes;
    queue[qsize++] = root;

    while (qsize > 0) {
        ro_index *idx = queue[--qsize];                      /* LIFO, reveal.c:21-25 */
        ro_trace *tr = (trace && st.nsteps < trace_cap) ? &trace[st.nsteps] : NULL;
        st.nsteps++;
        if (idx->depth > st.maxdepth) st.maxdepth = idx->depth;

        ro_mumlist ml;
        double t0 = now_s();
        scan_index(idx, minl, minn, &ml);                    /* reveal.c:802-822 */
        st.t_scan += now_s() - t0;

        if (tr) {
            memset(tr, 0, sizeof *tr);
            ro_saidx_t key = idx->nnodes ? idx->nodes[0].begin : -1;
            for (int k = 1; k < idx->nnodes; k++) if (idx->nodes[k].begin < key) key = idx->nodes[k].begin;
            tr->key = key; tr->n = idx->n; tr->depth = idx->depth; tr->nsamples = idx->nsamples;
            tr->nnodes = idx->nnodes; tr->nmums = ml.count;
            uint64_t h1 = 0, h2 = 0, h3 = 0, c3 = 0;
            for (ro_saidx_t i = 0; i < idx->n; i++) { h1 = ro_hash_step(h1, (uint64_t)i, idx->SA[i]); h2 = ro_hash_step(h2, (uint64_t)i, (int64_t)idx->LCP[i]); }
            for (int64_t k = 0; k < ml.count; k++) {      /* flat sequence l, n, (so, pos)... */
                h3 = ro_hash_step(h3, c3++, (int64_t)ml.l[k]); h3 = ro_hash_step(h3, c3++, ml.n[k]);
                for (int64_t q = ml.off[k]; q < ml.off[k + 1]; q++) { h3 = ro_hash_step(h3, c3++, ml.so[q]); h3 = ro_hash_step(h3, c3++, ml.pos[q]); }
            }
            tr->h_sa = h1; tr->h_lcp = h2; tr->h_mums = h3;
        }

        t0 = now_s();
        int64_t pick = -1;
        int pr = picker(user, idx, &ml, &pick);              /* reveal.c:851 */
        if (pr < 0) { err = 1; free_mumlist(&ml); free_index(idx); break; }
        if (pr == 0) {                                       /* "()" reveal.c:870-884 */
            st.t_pick += now_s() - t0;
            free_mumlist(&ml); free_index(idx);
            continue;
        }
        ro_mum mum;
        mum.l = ml.l[pick]; mum.n = ml.n[pick];
        mum.so = ml.so + ml.off[pick]; mum.pos = ml.pos + ml.off[pick];
        ro_splitspec sp; memset(&sp, 0, sizeof sp);
        int gr = galign(user, idx, &mum, &sp);               /* reveal.c:939 */
        st.t_pick += now_s() - t0;
        if (gr < 0) { err = 1; free_mumlist(&ml); free_index(idx); break; }
        if (gr == 0) { free_mumlist(&ml); free_index(idx); continue; }   /* None, reveal.c:962-974 */

        if (tr) {
            tr->picked = 1; tr->l = mum.l; tr->mn = mum.n;
            ro_saidx_t mn = mum.pos[0];
            for (int k = 1; k < mum.n; k++) if (mum.pos[k] < mn) mn = mum.pos[k];
            tr->sp_min = mn;
        }

        t0 = now_s();
        /* D-label: reveal.c:1005-1117 */
        uint8_t *D = (uint8_t *)malloc((size_t)(idx->n > 0 ? idx->n : 1));
        ro_saidx_t leadingn, trailingn, parn;
        ro_label(D, idx->n, m->SAi, (const ro_saidx_t *)sp.lead, sp.nlead, (const ro_saidx_t *)sp.trail, sp.ntrail,
                 (const ro_saidx_t *)sp.rest, sp.nrest, mum.pos, mum.n, mum.l, &leadingn, &trailingn, &parn);
        int newdepth = idx->depth + 1;
        ro_index *i_lead = NULL, *i_trail = NULL, *i_par = NULL;      /* reveal.c:1136-1207 */
        if (leadingn > 0) { i_lead = new_child(m, leadingn, newdepth, count_samples(m, sp.lead, sp.nlead), sp.lead, sp.nlead); sp.lead = NULL; }
        if (trailingn > 0) { i_trail = new_child(m, trailingn, newdepth, count_samples(m, sp.trail, sp.ntrail), sp.trail, sp.ntrail); sp.trail = NULL; }
        if (parn > 0) { i_par = new_child(m, parn, newdepth, count_samples(m, sp.rest, sp.nrest), sp.rest, sp.nrest); sp.rest = NULL; }
        ro_saidx_t il, it, ip;
        ro_split(idx->SA, idx->LCP, idx->n, D, m->SAi,           /* reveal.c:1217 */
                 i_lead ? i_lead->SA : NULL, i_lead ? i_lead->LCP : NULL,
                 i_trail ? i_trail->SA : NULL, i_trail ? i_trail->LCP : NULL,
                 i_par ? i_par->SA : NULL, i_par ? i_par->LCP : NULL, &il, &it, &ip);
        /* the reference sizes the children by labelled positions; a matched
         * range overlapping an interval would leave them short.  Not a
         * supported input: */
        assert(il == leadingn && it == trailingn && ip == parn);
        for (int j = 0; j < mum.n; j++)                          /* reveal.c:1230-1234 */
            for (ro_saidx_t i = mum.pos[j]; i < mum.pos[j] + (ro_saidx_t)mum.l; i++)
                if (m->T[i] >= 'A' && m->T[i] <= 'Z') m->T[i] = (char)(m->T[i] + 32);
        st.t_split += now_s() - t0;
        t0 = now_s();
        if (leadingn > 0) {                                      /* reveal.c:1250-1252 */
            ro_saidx_t *mb = (ro_saidx_t *)malloc(sizeof(ro_saidx_t) * (size_t)(sp.nmatch > 0 ? sp.nmatch : 1));
            for (int k = 0; k < sp.nmatch; k++) mb[k] = sp.match[k].begin;
            ro_bubble_sort(i_lead->SA, i_lead->LCP, i_lead->n, m->SAi, mb, sp.nmatch);
            free(mb);
        }
        st.t_bubble += now_s() - t0;
        free(D);
        st.nsplits++; st.anchored_bp += (int64_t)mum.l;
        free(sp.lead); free(sp.trail); free(sp.rest); free(sp.match);
        free_mumlist(&ml);
        free_index(idx);                                         /* incl. main SA/LCP, reveal.c:1279-1290 */

        if (qsize + 3 > qcap) { qcap += 10000; queue = (ro_index **)realloc(queue, sizeof(ro_index *) * (size_t)qcap); }
        if (parn > 0) queue[qsize++] = i_par;                    /* reveal.c:1296-1324 */
        if (leadingn > 0) queue[qsize++] = i_lead;
        if (trailingn > 0) queue[qsize++] = i_trail;
    }
    while (qsize > 0) free_index(queue[--qsize]);
    free(queue);
    if (stats) *stats = st;
    return err ? -1 : 0;
}

/* ------------------------------------------------------------------------ */
/* bench callbacks                                         */
/* ------------------------------------------------------------------------ */

int ro_bench_picker(void *user, const ro_index *idx, const ro_mumlist *mums, int64_t *pick) {
    (void)user;
    int64_t best = -1; ro_lcp_t bl = 0; ro_saidx_t bmin = 0;
    for (int64_t k = 0; k < mums->count; k++) {
        if (mums->n[k] != idx->nsamples) continue;          /* schemes.py:227 */
        ro_saidx_t mn = mums->pos[mums->off[k]];
        for (int64_t q = mums->off[k] + 1; q < mums->off[k + 1]; q++) if (mums->pos[q] < mn) mn = mums->pos[q];
        if (best < 0 || mums->l[k] > bl || (mums->l[k] == bl && mn < bmin)) { best = k; bl = mums->l[k]; bmin = mn; }
    }
    if (best < 0) return 0;
    *pick = best;
    return 1;
}

static int cmp_intv(const void *a, const void *b) {
    const ro_intv *x = (const ro_intv *)a, *y = (const ro_intv *)b;
    return (x->begin > y->begin) - (x->begin < y->begin);
}

int ro_bench_graphalign(void *user, const ro_index *idx, const ro_mum *mum, ro_splitspec *spec) {
    ro_bench_ctx *ctx = (ro_bench_ctx *)user;
    int nn = idx->nnodes, nm = mum->n;
    uint8_t *touched = (uint8_t *)calloc((size_t)(nn > 0 ? nn : 1), 1);
    spec->lead = (ro_intv *)malloc(sizeof(ro_intv) * (size_t)(nm > 0 ? nm : 1));
    spec->trail = (ro_intv *)malloc(sizeof(ro_intv) * (size_t)(nm > 0 ? nm : 1));
    spec->match = (ro_intv *)malloc(sizeof(ro_intv) * (size_t)(nm > 0 ? nm : 1));
    spec->rest = (ro_intv *)malloc(sizeof(ro_intv) * (size_t)(nn > 0 ? nn : 1));
    spec->nlead = spec->ntrail = spec->nmatch = spec->nrest = 0;
    for (int k = 0; k < nm; k++) {
        ro_saidx_t sp = mum->pos[k];
        int hit = -1;
        for (int q = 0; q < nn; q++) if (idx->nodes[q].begin <= sp && sp < idx->nodes[q].end) { hit = q; break; }
        if (hit < 0 || sp + (ro_saidx_t)mum->l > idx->nodes[hit].end) { free(touched); return -1; }
        touched[hit] = 1;
        if (sp > idx->nodes[hit].begin) { spec->lead[spec->nlead].begin = idx->nodes[hit].begin; spec->lead[spec->nlead].end = sp; spec->nlead++; }
        if (sp + (ro_saidx_t)mum->l < idx->nodes[hit].end) { spec->trail[spec->ntrail].begin = sp + (ro_saidx_t)mum->l; spec->trail[spec->ntrail].end = idx->nodes[hit].end; spec->ntrail++; }
        spec->match[spec->nmatch].begin = sp; spec->match[spec->nmatch].end = sp + (ro_saidx_t)mum->l; spec->nmatch++;
    }
    for (int q = 0; q < nn; q++) if (!touched[q]) spec->rest[spec->nrest++] = idx->nodes[q];
    free(touched);
    qsort(spec->lead, (size_t)spec->nlead, sizeof(ro_intv), cmp_intv);
    qsort(spec->trail, (size_t)spec->ntrail, sizeof(ro_intv), cmp_intv);
    qsort(spec->match, (size_t)spec->nmatch, sizeof(ro_intv), cmp_intv);
    qsort(spec->rest, (size_t)spec->nrest, sizeof(ro_intv), cmp_intv);
    if (ctx && ctx->nanchors < ctx->cap_anchors && ctx->npos + nm <= ctx->cap_pos) {
        ctx->l[ctx->nanchors] = mum->l; ctx->n[ctx->nanchors] = nm; ctx->off[ctx->nanchors] = ctx->npos;
        for (int k = 0; k < nm; k++) ctx->pos[ctx->npos + k] = spec->match[k].begin;
        ctx->npos += nm; ctx->nanchors++; ctx->off[ctx->nanchors] = ctx->npos;
    } else if (ctx) {
        ctx->nanchors++;   /* count even when the buffers are full */
    }
    return 1;
}
