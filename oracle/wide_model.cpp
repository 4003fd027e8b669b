// Test infrastructure only: a CPU model of the product's ordered, distance-culled walk over the
// 4-wide tree (orion_amd/csrc/wide_bvh.h builds it, orion_amd/csrc/wide_walk.h walks it on the GPU).
// The tests run it beside the oracle's restatement of the reference's walk
// (orion/avx/sbvh.cpp:36-83, orion/model.hpp:52-62) on millions of rays: the two must agree bit for
// bit.  It also measures what the culling rests on: how far the reference's own computed hit point
// o + t*d of an accepted triangle can lie outside that triangle's box.
//
// Arithmetic as in orion_oracle.cpp: -ffp-contract=off, explicit fmaf where the reference's object
// code fuses (cross), dot = (p0 + p1) + (p2 + 0.0f).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <omp.h>

#include "../orion_amd/csrc/wide_bvh.h"

namespace {

using orion_dev::Tri;
using orion_dev::WideNode;

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float dotp(V3 a, V3 b) { float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z; return (p0 + p1) + (p2 + 0.0f); }
inline V3 cross(V3 a, V3 b) { return v3(fmaf(-a.z, b.y, a.y * b.z), fmaf(-a.x, b.z, a.z * b.x), fmaf(-a.y, b.x, a.x * b.y)); }
inline float sse_min(float a, float b) { return (a < b) ? a : b; }
inline float sse_max(float a, float b) { return (a > b) ? a : b; }
inline float tpl_min(float a, float b) { return (b < a) ? b : a; }
inline float tpl_max(float a, float b) { return (a < b) ? b : a; }

inline void slab(const float lo[3], const float hi[3], V3 o, V3 inv, float& tmin, float& tmax) {
    const float t1x = (lo[0] - o.x) * inv.x, t1y = (lo[1] - o.y) * inv.y, t1z = (lo[2] - o.z) * inv.z;
    const float t2x = (hi[0] - o.x) * inv.x, t2y = (hi[1] - o.y) * inv.y, t2z = (hi[2] - o.z) * inv.z;
    tmin = tpl_max(sse_min(t1x, t2x), tpl_max(sse_min(t1y, t2y), sse_min(t1z, t2z)));
    tmax = tpl_min(sse_max(t1x, t2x), tpl_min(sse_max(t1y, t2y), sse_max(t1z, t2z)));
}

struct Model {
    std::vector<Tri> tris;
    std::vector<uint32_t> tri_mesh, mesh_base;
    orion_wide::WideTree tree;
    std::vector<std::vector<uint32_t>> below;   // 4 per wide node: triangle slots below (mode 3, the ideal certificate; built on first use)
    // The guard's table (orion_amd/csrc/wide_guard.h), one row per direction cell, built when a ray first asks for the cell:
    // per node four bits, "slot k may hold a triangle that is ill conditioned for some direction of the cell".
    uint32_t grid = 0;
    std::mutex mu;
    std::unordered_map<uint32_t, std::vector<uint8_t>> rows;
};

const uint8_t* cell_row(Model& M, uint32_t cell) {
    std::lock_guard<std::mutex> lock(M.mu);
    auto it = M.rows.find(cell);
    if (it != M.rows.end()) return it->second.data();
    float c[3], sr, cr;
    orion_dev::guard_cell_geometry(M.grid, cell, c, sr, cr);
    std::vector<uint8_t> row(M.tree.nodes.size(), 0);
    for (size_t i = M.tree.nodes.size(); i-- > 0;) {      // children stand behind their parents
        const WideNode& n = M.tree.nodes[i];
        const uint32_t refs[4] = {n.ref0, n.ref1, n.ref2, n.ref3};
        uint8_t bits = 0;
        for (int k = 0; k < 4; k++) {
            const uint32_t ref = refs[k];
            if (ref == orion_dev::kWideEmpty) continue;
            bool a = false;
            if (ref & 0x80000000u) {
                const uint32_t first = ref & 0x1FFFFFFu, count = (ref >> 25) & 63u;
                for (uint32_t s = first; s < first + count && !a; s++) a = orion_dev::guard_band_meets_cell(M.tree.bands[s], c, sr, cr);
            } else if (i >= M.tree.n_virtual || ref < M.tree.n_virtual) a = row[ref] != 0;   // (virtual nodes: forced slots only -- never a mesh's own inner node)
            if (a) bits |= (uint8_t)(1u << k);
        }
        row[i] = bits;
    }
    return M.rows.emplace(cell, std::move(row)).first->second.data();
}

struct Stats {
    uint64_t nodes = 0, own_fail = 0, leaves = 0, tris = 0, stale = 0, candidates = 0, danger_visits = 0, danger_slots = 0, slots_seen = 0;
    uint32_t max_stack = 0;
    double max_excess = 0.0;   // in units of the scene's pad
    uint64_t untame = 0;
    uint64_t depth_hist[32] = {};   // rays by the deepest their stack got
    uint64_t steps_hist[64] = {};   // rays by node + leaf visits / 4
};

struct Best { float t, u, v; uint32_t slot; };

// Triangle::intersect (orion/geometry.hpp:80-136) up to the comparison with the best hit; true: a candidate.
inline bool tri_candidate(const Tri& T, V3 o, V3 d, float& t, float& u, float& v) {
    const float eps = 0.000001f;
    const V3 v0 = v3(T.v0[0], T.v0[1], T.v0[2]), e1 = v3(T.e1[0], T.e1[1], T.e1[2]), e2 = v3(T.e2[0], T.e2[1], T.e2[2]);
    const V3 pvec = cross(d, e2);
    const float det = dotp(e1, pvec);
    if (det > -eps && det < eps) return false;
    const float inv_det = 1.0f / det;
    const V3 tvec = sub(o, v0);
    u = dotp(tvec, pvec) * inv_det;
    if (u < 0 || u > 1) return false;
    const V3 qvec = cross(tvec, e1);
    v = dotp(d, qvec) * inv_det;
    if (v < 0 || u + v > 1) return false;
    t = dotp(e2, qvec) * inv_det;
    return !(t < 0);
}

inline bool tame(const Model& M, V3 o, V3 d) {
    const float ob = M.tree.org_bound;
    auto okd = [](float x) { return std::isnormal(x) && std::fabs(x) < 0x1p+40f && std::fabs(x) > 0x1p-40f; };
    return std::fabs(o.x) <= ob && std::fabs(o.y) <= ob && std::fabs(o.z) <= ob && okd(d.x) && okd(d.y) && okd(d.z);
}

// cull = false: every slot the reference's rule allows is visited (the decoded boxes are ignored), which is
// the reference's candidate set; used to measure the excess.  Returns false for a ray the walk does not take.
// The guard of the product's walk (orion_amd/csrc/wide_walk.h: wide_guard_of): the cell of the ray's direction, or none
// for a ray too short for any triangle of the scene to pass the reference's determinant test ill conditioned.
inline uint32_t guard_cell(const Model& M, V3 d) {
    const float l2 = fmaf(d.x, d.x, fmaf(d.y, d.y, d.z * d.z));
    if (l2 * 1.00001f <= M.tree.guard_dlen * M.tree.guard_dlen) return orion_dev::kGuardNoCell;
    return orion_dev::guard_cell_of(d.x, d.y, d.z, M.grid);
}

bool walk(Model& M, V3 o, V3 d, int cull, float kappa, Best& best, Stats& st) {
    if (!tame(M, o, d)) { st.untame++; return false; }
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float pad = M.tree.pad;
    best.t = std::numeric_limits<float>::infinity(); best.u = best.v = 0.0f; best.slot = 0xFFFFFFFFu;
    struct Entry { uint32_t ref; float tmin; bool danger; float cull_t; };
    std::vector<Entry> stack;
    const uint32_t gcell = cull == 2 ? guard_cell(M, d) : orion_dev::kGuardNoCell;
    const uint8_t* grow = gcell != orion_dev::kGuardNoCell ? cell_row(M, gcell) : nullptr;
    const double dlen = std::sqrt((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z);
    uint32_t cur = 0;           // node index, or 0x80000000 | count << 25 | first triangle slot
    uint32_t my_depth = 0, my_steps = 0;
    const uint32_t kDone = 0x7FFFFFFFu;
    while (cur != kDone) {
        if (!(cur & 0x80000000u)) {
            const WideNode& n = M.tree.nodes[cur];
            st.nodes++; my_steps++;
            float a0, b0;
            slab(n.lo, n.hi, o, inv, a0, b0);
            // (a virtual node -- no box of the reference's -- is entered whenever one of its slots is dangerous)
            const bool vrule = cull == 2 && grow && cur < M.tree.n_virtual;   // (the product's rule for virtual nodes: wide_walk.h)
            const bool in = ((b0 > a0) && (b0 > 0.0f)) || (vrule && grow[cur] != 0);
            if (!in) st.own_fail++;
            Entry hit[4]; int nh = 0;
            if (in) {
                const uint32_t refs[4] = {n.ref0, n.ref1, n.ref2, n.ref3};
                float reach = 0.0f;
                const float oo[3] = {o.x, o.y, o.z};
                for (int a = 0; a < 3; a++) reach = std::fmax(reach, std::fmax(std::fabs(n.lo[a] - oo[a]), std::fabs(n.hi[a] - oo[a])));
                // grow every decoded slot box by the pad and by kappa times (the origin's distance to the farthest
                // corner of the node's box + two pads)
                const float r = pad + kappa * (reach + 2.0f * pad);
                for (int k = 0; k < 4; k++) {
                    if (refs[k] == orion_dev::kWideEmpty) continue;
                    float lo[3], hi[3];
                    for (int a = 0; a < 3; a++) {
                        lo[a] = (float)orion_wide::grid_decode((n.qlo[a] >> (8 * k)) & 0xFFu, n.lo[a], n.hi[a]) - r;
                        hi[a] = (float)orion_wide::grid_decode((n.qhi[a] >> (8 * k)) & 0xFFu, n.lo[a], n.hi[a]) + r;
                    }
                    float a, b;
                    slab(lo, hi, o, inv, a, b);
                    const bool met = (b >= a) && (b >= 0.0f);
                    const bool within = met && (a <= best.t);
                    bool danger = false;
                    st.slots_seen++;
                    if (cull == 2 && grow) danger = ((grow[cur] >> k) & 1u) != 0u;
                    if (cull == 3) {   // the ideal certificate: is any triangle below ill conditioned for this ray and large enough to be accepted?
                        for (uint32_t s : M.below[4 * cur + k]) {
                            const Tri& T = M.tris[s];
                            const double e1[3] = {T.e1[0], T.e1[1], T.e1[2]}, e2[3] = {T.e2[0], T.e2[1], T.e2[2]};
                            const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                            const double m = std::sqrt((e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) * (e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]));
                            if (dlen * m <= (double)orion_dev::kGuardHarmless) continue;
                            static const double rho = std::getenv("WIDE_MODEL_CELL_DEG") ? std::atof(std::getenv("WIDE_MODEL_CELL_DEG")) * 3.14159265358979 / 180.0 : 0.0;   // a direction cell's radius
                            const double lc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                            const double beta = std::asin(std::fmin(1.0, (double)orion_dev::kGuardKappa * m / std::fmax(lc, 1e-300)));
                            if (std::fabs(c[0] * d.x + c[1] * d.y + c[2] * d.z) < std::sin(std::fmin(1.5707963, beta + rho)) * dlen * lc) { danger = true; break; }
                        }
                    }
                    // a dangerous inner slot the ray's line misses fails the reference's own box test; a dangerous leaf is visited anyway
                    if (vrule) danger = danger || met;      // bit: forced whatever the box; otherwise visited iff met, never dropped by distance
                    else if (danger && !(refs[k] & 0x80000000u) && !met) danger = false;
                    const float cull_t = danger ? -1.0f : a;
                    if (danger && !within) st.danger_slots++;
                    if (cull && !within && !danger) continue;
                    hit[nh].ref = refs[k]; hit[nh].tmin = (within ? a : 3.0e38f); hit[nh].danger = danger; hit[nh].cull_t = cull_t; nh++;
                }
            }
            // the nearest slot next; the others wait on the stack in slot order
            int nearest = 0;
            for (int i = 1; i < nh; i++) if (hit[i].tmin < hit[nearest].tmin) nearest = i;
            if (nh) std::swap(hit[0], hit[nearest]);
            for (int i = nh - 1; i >= 1; i--) stack.push_back(hit[i]);
            st.max_stack = std::max<uint32_t>(st.max_stack, (uint32_t)stack.size());
            my_depth = std::max<uint32_t>(my_depth, (uint32_t)stack.size());
            if (nh) { if (hit[0].tmin > best.t) st.danger_visits++; cur = hit[0].ref; continue; }
        } else {
            const uint32_t first = cur & 0x1FFFFFFu, count = (cur >> 25) & 63u;
            st.leaves++; my_steps++;
            for (uint32_t s = first; s < first + count; s++) {
                st.tris++;
                float t, u, v;
                if (!tri_candidate(M.tris[s], o, d, t, u, v)) continue;
                st.candidates++;
                if (!cull) {   // how far outside its triangle's box is the computed hit point? (in units of pad + kappa * reach of that box)
                    const Tri& T = M.tris[s];
                    const float P[3] = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
                    for (int a = 0; a < 3; a++) {
                        const float p1 = T.v0[a] + T.e1[a], p2 = T.v0[a] + T.e2[a];
                        const float lo = std::fmin(T.v0[a], std::fmin(p1, p2)), hi = std::fmax(T.v0[a], std::fmax(p1, p2));
                        double reach = 0.0;
                        for (int c = 0; c < 3; c++) {
                            const float q1 = T.v0[c] + T.e1[c], q2 = T.v0[c] + T.e2[c];
                            const float l2 = std::fmin(T.v0[c], std::fmin(q1, q2)), h2 = std::fmax(T.v0[c], std::fmax(q1, q2));
                            const float oc = c == 0 ? o.x : (c == 1 ? o.y : o.z);
                            reach = std::fmax(reach, std::fmax(std::fabs(l2 - oc), std::fabs(h2 - oc)));
                        }
                        const double ex = std::fmax(0.0, std::fmax((double)lo - P[a], (double)P[a] - hi)) / (pad + kappa * reach);
                        if (ex > st.max_excess) st.max_excess = ex;
                    }
                }
                if (t < best.t || (t == best.t && (best.slot == 0xFFFFFFFFu || s > best.slot))) { best.t = t; best.u = u; best.v = v; best.slot = s; }
            }
        }
        // pop
        cur = kDone;
        while (!stack.empty()) {
            const Entry e = stack.back(); stack.pop_back();
            if (cull && e.cull_t > best.t) { st.stale++; continue; }
            if (e.tmin > best.t) st.danger_visits++;
            cur = e.ref; break;
        }
    }
    st.depth_hist[std::min<uint32_t>(my_depth, 31)]++;
    st.steps_hist[std::min<uint32_t>(my_steps / 4, 63)]++;
    return true;
}

// triangles below a reference
void collect(const Model& M, uint32_t ref, std::vector<uint32_t>& out) {
    if (ref == orion_dev::kWideEmpty) return;
    if (ref & 0x80000000u) { const uint32_t first = ref & 0x1FFFFFFu, count = (ref >> 25) & 63u; for (uint32_t s = first; s < first + count; s++) out.push_back(s); return; }
    const WideNode& n = M.tree.nodes[ref];
    collect(M, n.ref0, out); collect(M, n.ref1, out); collect(M, n.ref2, out); collect(M, n.ref3, out);
}

void build_below(Model& M) {
    M.below.assign(4 * M.tree.nodes.size(), {});
    for (size_t i = 0; i < M.tree.nodes.size(); i++) {
        const WideNode& n = M.tree.nodes[i];
        const uint32_t refs[4] = {n.ref0, n.ref1, n.ref2, n.ref3};
        for (int k = 0; k < 4; k++) collect(M, refs[k], M.below[4 * i + k]);
    }
}

} // namespace

extern "C" {

// meshes: OrionMeshDesc array as for orion_hip_scene_create.  Returns null (and the reason in why[256]) if the
// scene is not eligible for the wide walk.
void* wide_model_create(const OrionMeshDesc* meshes, uint32_t n_meshes, char* why) {
    Model* M = new Model;
    std::vector<uint32_t> slot_base;
    uint32_t tri_base = 0;
    for (uint32_t mi = 0; mi < n_meshes; mi++) {
        const OrionMeshDesc& m = meshes[mi];
        const uint32_t ntri = m.n_indices / 3;
        slot_base.push_back((uint32_t)M->tris.size());
        M->mesh_base.push_back(tri_base);
        for (uint32_t i = 0; i < ntri; i++) {
            const uint32_t o = m.tri_order[i];
            const float* p0 = m.vertices + 8ull * m.indices[3 * o];
            const float* p1 = m.vertices + 8ull * m.indices[3 * o + 1];
            const float* p2 = m.vertices + 8ull * m.indices[3 * o + 2];
            Tri t{};
            for (int a = 0; a < 3; a++) { t.v0[a] = p0[a]; t.e1[a] = p1[a] - p0[a]; t.e2[a] = p2[a] - p0[a]; }
            t.gtri = tri_base + o;
            M->tris.push_back(t);
            M->tri_mesh.push_back(mi);
        }
        tri_base += ntri;
    }
    M->tree = orion_wide::build_wide_tree(meshes, n_meshes, slot_base, M->tris);
    M->grid = orion_dev::guard_grid_for(M->tree.nodes.size());
    if (const char* e = std::getenv("ORION_HIP_GUARD_GRID")) { const int g = std::atoi(e); if (g == 32 || g == 64 || g == 128 || g == 256) M->grid = (uint32_t)g; }
    if (M->tree.nodes.empty()) {
        if (why) { std::strncpy(why, M->tree.why_not.c_str(), 255); why[255] = 0; }
        delete M;
        return nullptr;
    }
    return M;
}

void wide_model_destroy(void* h) { delete static_cast<Model*>(h); }

// info[0] = wide nodes, [1] = stack_need, [2] = virtual nodes (they come first); finfo[0] = pad, [1] = org_bound
void wide_model_info(void* h, uint32_t* info, float* finfo) {
    const Model* M = static_cast<Model*>(h);
    info[0] = (uint32_t)M->tree.nodes.size(); info[1] = M->tree.stack_need; info[2] = M->tree.n_virtual;
    finfo[0] = M->tree.pad; finfo[1] = M->tree.org_bound;
}

// The tree's nodes as they are uploaded (n_nodes x 16 words), and per node and slot the number of triangles below.
void wide_model_nodes(void* h, uint32_t* words, uint32_t* below4) {
    Model* M = static_cast<Model*>(h);
    std::memcpy(words, M->tree.nodes.data(), M->tree.nodes.size() * sizeof(WideNode));
    if (below4) {
        if (M->below.empty()) build_below(*M);
        for (size_t i = 0; i < M->below.size(); i++) below4[i] = (uint32_t)M->below[i].size();
    }
}

// ids: n x 2 (mesh, triangle; 0xFFFFFFFF = miss, 0xFFFFFFFE = ray not taken by the walk), tuv: n x 3.
// stats (doubles): nodes, own-box failures, leaves, triangle tests, stale pops, candidates, deepest stack,
// largest excess (in pads), rays not taken.
void wide_model_closest_hit(void* h, const float* org, const float* dir, uint32_t n, int cull, float kappa, uint32_t* ids, float* tuv,
                            double* stats, int threads) {
    Model* M = static_cast<Model*>(h);
    if (cull == 3 && M->below.empty()) build_below(*M);
    Stats total;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        Stats st;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const V3 o = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
            Best b;
            if (!walk(*M, o, d, cull, kappa, b, st)) { ids[2 * i] = ids[2 * i + 1] = 0xFFFFFFFEu; tuv[3 * i] = tuv[3 * i + 1] = tuv[3 * i + 2] = 0.0f; continue; }
            if (b.slot == 0xFFFFFFFFu) { ids[2 * i] = ids[2 * i + 1] = 0xFFFFFFFFu; }
            else { const uint32_t g = M->tris[b.slot].gtri, mi = M->tri_mesh[b.slot]; ids[2 * i] = mi; ids[2 * i + 1] = g - M->mesh_base[mi]; }
            tuv[3 * i] = b.t; tuv[3 * i + 1] = b.u; tuv[3 * i + 2] = b.v;
        }
#pragma omp critical
        {
            total.nodes += st.nodes; total.own_fail += st.own_fail; total.leaves += st.leaves; total.tris += st.tris;
            total.stale += st.stale; total.candidates += st.candidates; total.untame += st.untame; total.danger_visits += st.danger_visits; total.danger_slots += st.danger_slots; total.slots_seen += st.slots_seen;
            total.max_stack = std::max(total.max_stack, st.max_stack);
            total.max_excess = std::max(total.max_excess, st.max_excess);
            for (int k = 0; k < 32; k++) total.depth_hist[k] += st.depth_hist[k];
            for (int k = 0; k < 64; k++) total.steps_hist[k] += st.steps_hist[k];
        }
    }
    if (stats) {
        stats[0] = (double)total.nodes; stats[1] = (double)total.own_fail; stats[2] = (double)total.leaves; stats[3] = (double)total.tris;
        stats[4] = (double)total.stale; stats[5] = (double)total.candidates; stats[6] = total.max_stack; stats[7] = total.max_excess;
        stats[8] = (double)total.untame;
        for (int k = 0; k < 32; k++) stats[9 + k] = (double)total.depth_hist[k];
        for (int k = 0; k < 64; k++) stats[41 + k] = (double)total.steps_hist[k];
        stats[105] = (double)total.danger_visits; stats[106] = (double)total.danger_slots; stats[107] = (double)total.slots_seen;
    }
}

} // extern "C"
