// rv_graphrem.h -- what the recursion (rv_align.hip, picker kind 2) calls of rv_graphrem.hip: the picker and graphalign of `reveal rem` for graph inputs
#pragma once
#include "../../include/reveal_amd.h"
#include <cstdint>
#include <vector>

struct RvGraphIv { int64_t b, e; };      // a graph node by its text interval; b < 0: None
struct RvGraphAlignOut {
    std::vector<RvGraphIv> lead, trail, match, rest;      // sorted by begin
    RvGraphIv merged{-1, -1}, newleft{-1, -1}, newright{-1, -1};
};
// rem.graphalign (rem.py:318-382): nodes = the sub-index' intervals, left / right = its left / right graph node, the match as (l, members in the picker's order)
int rv_graph_do_align(rv_graph *g, const RvGraphIv *nodes, size_t nn, RvGraphIv left, RvGraphIv right, uint32_t l, const int64_t *pos, int npos, RvGraphAlignOut &out);
// schemes.graphmumpicker, not-precomputed branch (schemes.py:197-361); returns as rv_pick_chain
int rv_graph_do_pick(rv_graph *g, const rv_picker_args *A, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so, const int64_t *pos,
                     RvGraphIv left, RvGraphIv right, int minlength, rv_picker_out *O);
