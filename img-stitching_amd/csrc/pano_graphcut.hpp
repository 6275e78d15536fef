// pano_graphcut.hpp - the max-flow of cv::detail::GraphCutSeamFinder (host side of pano_build_masks_graphcut).
//
// The reference's seam finder (ocvstitcher.hpp:1033-1035, :1244) cuts the overlap of every image pair with
// GCGraph<float>::maxFlow (imgproc/src/gcgraph.hpp): Boykov-Kolmogorov search trees grown from both terminals, one
// augmentation per found path, orphan adoption by time stamp and distance.  The algorithm is sequential by construction
// (the labelling of vertices that end up in neither tree depends on the order of growth), so it runs on the host, as it
// does in the reference; the pixel work around it - terminal and edge weights from the warped images and masks, and the
// mask update from the labels - are GPU kernels (pano_init.hip).  All weights are integers below 2^24 carried in f32
// (squared colour distances + 1 + penalties), so the arithmetic is exact and the order of operations does not matter.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

namespace pano {

class GridMaxFlow {
  public:
    // a W x H 4-connected grid: term[k] = source weight - sink weight of vertex k, wh[k] = capacity of k <-> k + 1,
    // wv[k] = capacity of k <-> k + W (both directions).  Vertices and edges are added in setGraphWeightsColor's order:
    // all terminal weights, then per vertex its right edge and its down edge
    GridMaxFlow(int W, int H, const float* term, const float* wh, const float* wv) : vtx_((size_t)W * H), edge_(2) {
        edge_.reserve(2 + 2 * ((size_t)(H - 1) * W + (size_t)(W - 1) * H));
        for (size_t k = 0; k < vtx_.size(); k++) vtx_[k].weight = term[k];
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int v = y * W + x;
                if (x < W - 1) link(v, v + 1, wh[v]);
                if (y < H - 1) link(v, v + W, wv[v]);
            }
    }

    // GCGraph::inSourceSegment
    bool inSource(int v) const { return vtx_[v].t == 0; }

    void run() {
        constexpr int kTerminal = -1, kOrphan = -2;
        const int n = (int)vtx_.size();
        head_ = tail_ = kNil;
        std::vector<int> orphans;
        int now = 0;
        for (int i = 0; i < n; i++) {
            Vtx& v = vtx_[i];
            v.ts = 0;
            if (v.weight != 0) {
                activate(i);
                v.dist = 1;
                v.parent = kTerminal;
                v.t = v.weight < 0;
            } else {
                v.parent = 0;
            }
        }
        for (;;) {
            int e0 = -1, ei = 0;
            // grow the two trees until an edge joins them
            while (head_ != kNil) {
                const int vi = head_;
                Vtx& v = vtx_[vi];
                if (v.parent) {
                    const int vt = v.t;
                    for (ei = v.first; ei != 0; ei = edge_[ei].next) {
                        if (edge_[ei ^ vt].cap == 0) continue;
                        const int ui = edge_[ei].dst;
                        Vtx& u = vtx_[ui];
                        if (!u.parent) {
                            u.t = (uint8_t)vt;
                            u.parent = ei ^ 1;
                            u.ts = v.ts;
                            u.dist = v.dist + 1;
                            if (!u.next) activate(ui);
                            continue;
                        }
                        if (u.t != vt) {
                            e0 = ei ^ vt;
                            break;
                        }
                        if (u.dist > v.dist + 1 && u.ts <= v.ts) {
                            u.parent = ei ^ 1;
                            u.ts = v.ts;
                            u.dist = v.dist + 1;
                        }
                    }
                    if (e0 > 0) break;
                }
                head_ = v.next == kNil ? kNil : v.next - 1;
                if (head_ == kNil) tail_ = kNil;
                v.next = 0;
            }
            if (e0 <= 0) break;
            // bottleneck along the path (k = 1: source tree, k = 0: sink tree)
            float push = edge_[e0].cap;
            for (int k = 1; k >= 0; k--) {
                int vi = edge_[e0 ^ k].dst;
                for (;; vi = edge_[ei].dst) {
                    if ((ei = vtx_[vi].parent) < 0) break;
                    push = std::min(push, edge_[ei ^ k].cap);
                }
                push = std::min(push, std::fabs(vtx_[vi].weight));
            }
            edge_[e0].cap -= push;
            edge_[e0 ^ 1].cap += push;
            for (int k = 1; k >= 0; k--) {
                int vi = edge_[e0 ^ k].dst;
                for (;; vi = edge_[ei].dst) {
                    if ((ei = vtx_[vi].parent) < 0) break;
                    edge_[ei ^ (k ^ 1)].cap += push;
                    if ((edge_[ei ^ k].cap -= push) == 0) {
                        orphans.push_back(vi);
                        vtx_[vi].parent = kOrphan;
                    }
                }
                vtx_[vi].weight = vtx_[vi].weight + push * (float)(1 - k * 2);
                if (vtx_[vi].weight == 0) {
                    orphans.push_back(vi);
                    vtx_[vi].parent = kOrphan;
                }
            }
            // give every orphan a new parent or set it free
            now++;
            while (!orphans.empty()) {
                const int oi = orphans.back();
                orphans.pop_back();
                Vtx& o = vtx_[oi];
                int best = INT_MAX, ej = 0;
                e0 = 0;
                const int vt = o.t;
                for (ei = o.first; ei != 0; ei = edge_[ei].next) {
                    if (edge_[ei ^ (vt ^ 1)].cap == 0) continue;
                    Vtx* u = &vtx_[edge_[ei].dst];
                    if (u->t != vt || u->parent == 0) continue;
                    int d = 0;
                    for (;;) {  // distance of u to its root
                        if (u->ts == now) {
                            d += u->dist;
                            break;
                        }
                        ej = u->parent;
                        d++;
                        if (ej < 0) {
                            if (ej == kOrphan) d = INT_MAX - 1;
                            else {
                                u->ts = now;
                                u->dist = 1;
                            }
                            break;
                        }
                        u = &vtx_[edge_[ej].dst];
                    }
                    if (++d < INT_MAX) {
                        if (d < best) {
                            best = d;
                            e0 = ei;
                        }
                        for (u = &vtx_[edge_[ei].dst]; u->ts != now; u = &vtx_[edge_[u->parent].dst]) {
                            u->ts = now;
                            u->dist = --d;
                        }
                    }
                }
                if ((o.parent = e0) > 0) {
                    o.ts = now;
                    o.dist = best;
                    continue;
                }
                o.ts = 0;
                for (ei = o.first; ei != 0; ei = edge_[ei].next) {
                    const int ui = edge_[ei].dst;
                    Vtx& u = vtx_[ui];
                    ej = u.parent;
                    if (u.t != vt || !ej) continue;
                    if (edge_[ei ^ (vt ^ 1)].cap != 0 && !u.next) activate(ui);
                    if (ej > 0 && edge_[ej].dst == oi) {
                        orphans.push_back(ui);
                        u.parent = kOrphan;
                    }
                }
            }
        }
    }

  private:
    static constexpr int kNil = -1;
    struct Vtx {
        int next = 0;    // active list: 0 = not listed, kNil = last, else successor + 1
        int parent = 0;  // edge to the parent, or kTerminal / kOrphan / 0 (free)
        int first = 0;   // first outgoing edge
        int ts = 0, dist = 0;
        float weight = 0.f;  // residual terminal capacity: > 0 towards the source, < 0 towards the sink
        uint8_t t = 0;       // tree: 0 source, 1 sink
    };
    struct Edge {
        int dst = 0, next = 0;
        float cap = 0.f;
    };
    void link(int i, int j, float w) {
        Edge a;
        a.dst = j; a.next = vtx_[i].first; a.cap = w;
        vtx_[i].first = (int)edge_.size();
        edge_.push_back(a);
        Edge b;
        b.dst = i; b.next = vtx_[j].first; b.cap = w;
        vtx_[j].first = (int)edge_.size();
        edge_.push_back(b);
    }
    void activate(int i) {
        vtx_[i].next = kNil;
        if (tail_ == kNil) head_ = i;
        else vtx_[tail_].next = i + 1;
        tail_ = i;
    }
    std::vector<Vtx> vtx_;
    std::vector<Edge> edge_;  // edge e and e ^ 1 are a pair; 0 and 1 are unused so that 0 can mean "none"
    int head_ = kNil, tail_ = kNil;
};

}  // namespace pano
