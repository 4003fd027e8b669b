#include "bvh_builder.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace orion_amd {

namespace {

// Arithmetic notes.  Boxes are combined with the SSE min/max lane rule the
// reference uses ((a<b)?a:b and (a>b)?a:b, orion/math.hpp:357-365), centroids
// are 0.5*(lo+hi) and the sort key is (centroid-lo)/(hi-lo) on the split axis
// (orion/AABB.hpp:63-72).  The median split is std::nth_element, whose result
// the reference inherits from libstdc++ (orion/avx/sbvh.cpp:142-149).
struct Bounds {
    float lo[3], hi[3];
    Bounds() {
        for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const Bounds& o) {
        for (int a = 0; a < 3; a++) {
            lo[a] = (lo[a] < o.lo[a]) ? lo[a] : o.lo[a];
            hi[a] = (hi[a] > o.hi[a]) ? hi[a] : o.hi[a];
        }
    }
    int widest_axis() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        return dy > dz ? 1 : 2;
    }
    float key(const Bounds& tri, int axis) const {
        float c = 0.5f * (tri.lo[axis] + tri.hi[axis]);
        return (c - lo[axis]) / (hi[axis] - lo[axis]);
    }
    // Surface area with the fused-multiply-add placement of the reference's own
    // object code (built with -mfma): dx*dy rounded, the other two products fused.
    float area() const {
        if (lo[0] > hi[0] && lo[1] > hi[1] && lo[2] > hi[2]) return 0;
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2 * std::fmaf(dy, dz, std::fmaf(dx, dz, dx * dy));
    }
};

struct Item { Bounds box; uint32_t id; };

struct Builder {
    std::vector<Item> items;
    std::vector<BvhNode>* nodes;
    SplitStrategy strategy;
    static constexpr uint32_t kLeafTriangles = 4;
    static constexpr uint32_t kBuckets = 12;

    uint32_t build(uint32_t begin, uint32_t end) {
        const uint32_t self = (uint32_t)nodes->size();
        nodes->push_back(BvhNode());
        Bounds all;
        for (uint32_t i = begin; i < end; i++) all.grow(items[i].box);
        {
            BvhNode& n = (*nodes)[self];
            for (int a = 0; a < 3; a++) { n.lo[a] = all.lo[a]; n.hi[a] = all.hi[a]; }
            n.begin = begin; n.end = end; n.is_leaf = 1; n.child[0] = n.child[1] = 0;
        }
        const uint32_t count = end - begin;
        if (count <= kLeafTriangles) return self;

        const int axis = all.widest_axis();
        Item* first = items.data() + begin;
        Item* last = items.data() + end;
        uint32_t mid;
        if (strategy == SplitStrategy::MEDIAN) {
            mid = (begin + end) / 2;
            std::nth_element(first, items.data() + mid, last,
                             [&](const Item& a, const Item& b) { return all.key(a.box, axis) < all.key(b.box, axis); });
        } else if (strategy == SplitStrategy::MIDDLE) {
            Item* p = std::partition(first, last, [&](const Item& t) { return all.key(t.box, axis) <= 0.5f; });
            if (p == first || p == last) return self;
            mid = (uint32_t)(p - items.data());
        } else {
            int counts[kBuckets] = {0};
            Bounds boxes[kBuckets];
            auto bucket_of = [&](const Item& t) {
                unsigned b = (unsigned)(kBuckets * all.key(t.box, axis));
                return b == kBuckets ? b - 1 : b;
            };
            for (uint32_t i = begin; i < end; i++) {
                unsigned b = bucket_of(items[i]);
                counts[b]++;
                boxes[b].grow(items[i].box);
            }
            float best = 0; unsigned best_bucket = 0;
            Bounds left; unsigned nleft = 0;
            for (unsigned i = 0; i + 1 < kBuckets; i++) {
                left.grow(boxes[i]); nleft += counts[i];
                Bounds right; unsigned nright = 0;
                for (unsigned j = i + 1; j < kBuckets; j++) { right.grow(boxes[j]); nright += counts[j]; }
                float cost = 1.0f + std::fmaf((float)nright, right.area(), (float)nleft * left.area()) / all.area();
                if (i == 0 || cost < best) { best = cost; best_bucket = i; }
            }
            if (best >= count * 1.0f) return self;
            Item* p = std::partition(first, last, [&](const Item& t) { return bucket_of(t) <= best_bucket; });
            mid = (uint32_t)(p - items.data());
        }
        (*nodes)[self].is_leaf = 0;
        uint32_t l = build(begin, mid);
        uint32_t r = build(mid, end);
        (*nodes)[self].child[0] = l;
        (*nodes)[self].child[1] = r;
        return self;
    }
};

} // namespace

void build_bvh(Mesh& mesh, SplitStrategy strategy) {
    const uint32_t ntri = (uint32_t)(mesh.indices.size() / 3);
    Builder b;
    b.items.resize(ntri);
    for (uint32_t t = 0; t < ntri; t++) {
        Bounds tri;
        // min(v0, min(v1, v2)) / max(v0, max(v1, v2)) in the reference's operand order (orion/avx/sbvh.hpp:28-29)
        const float* p0 = mesh.vertices[mesh.indices[3 * t]].position;
        const float* p1 = mesh.vertices[mesh.indices[3 * t + 1]].position;
        const float* p2 = mesh.vertices[mesh.indices[3 * t + 2]].position;
        for (int a = 0; a < 3; a++) {
            float mn = (p1[a] < p2[a]) ? p1[a] : p2[a];
            float mx = (p1[a] > p2[a]) ? p1[a] : p2[a];
            tri.lo[a] = (p0[a] < mn) ? p0[a] : mn;
            tri.hi[a] = (p0[a] > mx) ? p0[a] : mx;
        }
        b.items[t].box = tri;
        b.items[t].id = t;
    }
    mesh.nodes.clear();
    b.nodes = &mesh.nodes;
    b.strategy = strategy;
    b.build(0, ntri);
    mesh.tri_order.resize(ntri);
    for (uint32_t i = 0; i < ntri; i++) mesh.tri_order[i] = b.items[i].id;
}

} // namespace orion_amd
