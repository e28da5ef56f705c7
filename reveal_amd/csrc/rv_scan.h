// rv_scan.h -- scan kernels' host interface (see rv_scan.hip)
#pragma once
#include "rv_common.h"

#define RV_PAIR_TILE 1024

// one pairwise MUM: a < b text positions, l = LCP[rank], rank inside the
// scanned (concatenated) array
struct RvPairRec {
    sa_t a, b;
    u32  l;
    u32  rank;
};

// Streams SA/LCP[0..m) once.  Survivors of tile t (1024 ranks) are written in
// rank order to out[tiletab[t].x .. +tiletab[t].y); tiles land in atomic
// order, the host concatenates them in tile order.  *counter must be zeroed.
int rv_scan_pair_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, sa_t nsep0, int minl,
                        RvPairRec *out, u32 out_cap, u32 *counter, uint2 *tiletab);

#define RV_MULTI_TILE 256
struct RvMultiRec { u32 l, n, ub, pad; };
// Multi-MUM scan (getmultimums, reveal.c:436-580).  Records and members of
// tile t (256 ranks) land at rec[tab.x .. +tab.y) / so,pos[tab.z .. +tab.w) in
// the reference's emission order; counters[0..1] must be zeroed.
int rv_scan_multi_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, const sa_t *nsep, int nsamples,
                         int minl, int minn, RvMultiRec *rec, uint16_t *so, sa_t *pos, u32 rec_cap, u32 mem_cap, u32 *counters, uint4 *tiletab);
