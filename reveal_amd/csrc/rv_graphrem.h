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

// The anchors' surgery (rv_graph_replay) as a follower of a running recursion: the level loop hands over every level's anchors as it has chosen them, a host thread
// of its own breaks and merges the nodes while the GPU scans the next level (rv_graph.hip).  start: g = rv_graph_replay_begin's graph; push: `count` anchors, member k of
// anchor a at pos[off[a] - off[0] + k]; finish: waits for the thread, renumbers the graph; -1 (rv_last_error) when the surgery failed.
struct RvReplayFeed;
RvReplayFeed *rv_replay_feed_start(rv_graph *g);
void rv_replay_feed_push(RvReplayFeed *f, const uint32_t *l, const int64_t *off, const int64_t *pos, size_t count);
int rv_replay_feed_finish(RvReplayFeed *f);
