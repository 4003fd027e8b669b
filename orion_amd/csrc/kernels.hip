// The C ABI declared in include/orion_hip.h: scene upload (flattening the host-built trees into the
// device layout), the two ray-query entry points and the render loop that drives the kernels of
// pathtrace.h.  Built for gfx950 (MI355X) only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/orion_hip.h"
#include "pathtrace.h"
#include "wide_bvh.h"

// ===========================================================================
// Host side
// ===========================================================================
namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(2, std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // grow with headroom so that repeated calls with slowly growing sizes do not reallocate
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(2, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct EventPair { hipEvent_t a, b; int kernel; uint64_t items; };

// What one batch of paths in flight owns: a stream, the queues, the bounce counters and their pinned
// mirror.  Consecutive batches alternate between two lanes so that the head of one batch can run
// beside the tail of the one before it.
struct Lane {
    hipStream_t stream = nullptr;
    DevBuf q[6], hits, rad, counter;
    DevBuf pre_hits, pre_work;               // large scenes: k_extend_wide's per-ray hits and its grab counter
    DevBuf jitter;                           // the batch's per-sample jitters (k_jitter)
    DevBuf hit_mask;                         // a bit per path: did its camera ray hit anything (k_extend's first launch, k_resolve)
    // Per-bounce counters travel to the host through a ring: the device slot of bounce d is
    // d % kRing; the next bounce's extend launch stores it into the pinned mirror
    // {tag << 32 | vertices, tag << 32 | rays, tag << 32 | rays of the largest queue partition, unused}.
    unsigned long long* h_count = nullptr;   // pinned and coherent, kRing x 4 words
    hipEvent_t resolved = nullptr;           // after the lane's newest k_resolve
};

// Bounce-counter ring: the host launches bounce d while it reads the counters of bounce d - kLag,
// so the stream always holds queued work and no launch waits on a host round trip.
constexpr uint32_t kRing = 8, kLag = 2, kClearAhead = 3;
static_assert(kLag + kClearAhead + 1 < kRing, "a slot must be read before it is cleared for reuse");

// Scratch of orion_hip_closest_hit_noalloc: the ray-query kernels' work words (grab counter, number of
// listed chunks) and the list of 64-ray chunks left to the wide walk -- what orion_hip_closest_hit keeps in
// a buffer it allocates.  Every GPU the library runs on holds its own copy of this array until the process
// ends; g_noalloc_scratch[device] is that copy's address, looked up once per device at scene creation.
constexpr uint32_t kNoAllocMaxRays = 1u << 24;
constexpr uint32_t kNoAllocWords = kWorkWords + kNoAllocMaxRays / 64;     // the work words, then the chunk list (1 MiB)
constexpr int kMaxDevices = 64;
uint32_t* g_noalloc_scratch[kMaxDevices] = {};
std::mutex g_noalloc_mutex;
std::atomic<int> g_noalloc_busy[kMaxDevices] = {};   // 1 while a call uses the device's copy: a second call is refused, not raced

} // namespace

static __device__ __align__(256) uint32_t scratch[kNoAllocWords];

namespace {

// With `device` current: the address of this device's copy of `scratch`.
int resolve_noalloc_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return 0;   // (such a device simply has no no-allocation entry point)
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_scratch[device]) return 0;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    g_noalloc_scratch[device] = static_cast<uint32_t*>(ptr);
    return 0;
}

} // namespace

struct OrionScene {
    int device = 0;
    hipStream_t stream = nullptr;
    SceneView view{};
    std::vector<void*> owned;         // device allocations of the flattened scene
    std::vector<uint32_t> mesh_tri_base;
    uint32_t n_nodes = 0, n_tris = 0;
    bool solid = false;   // solid diffuse and emissive colours everywhere, every light's vertices share one normal (Mesh::kd_solid ...)
    // The hand-scheduled wave-synchronous walk steps through the node and triangle arrays with 32-bit byte
    // offsets (intersect.h): scenes whose arrays reach 4 GiB never take it.
    bool wave_ok = true;
    // options
    int traversal = 0;                // 0 auto, 1 lane, 2 wave, 3 one ray per wave at a time, 4 wide tree
    std::string wide_why_not;         // why the scene has no wide tree ("" if it has one)
    uint32_t tail_paths = 1u << 14;   // hand a batch's last paths to k_tail once this few are left
    uint64_t batch_paths = 256u << 20;    // paths in flight per batch (128 B of queues each; less where the device's free memory asks for it)
    uint32_t queue_parts_log2 = kPartsLog2;   // partitions of a render batch's queues (2^n; FrameParams::part_log2)
    uint32_t overlap_below = 128u << 20;   // start the next batch on the other lane once this few rays are left (0: never)
    uint32_t fail_queue_allocs = 0;       // test hook (ORION_HIP_TEST_HOOKS=1 only): the next n attempts to take a lane's queues fail
    uint32_t fail_at_bounce = 0;          // test hook (ORION_HIP_TEST_HOOKS=1 only): the n-th bounce of the next render call fails
    uint64_t queue_budget = 0;            // bytes the render queues may take (0: three quarters of the device's free memory)
    bool draw_rays = true;                // large scenes: bounce rays are handed to lanes as they finish (k_extend_wide)
    bool wave_ahead = false;              // large scenes: the primary rays' wave-synchronous walk fetches one record ahead (slower since a wave holds one pixel's samples)
    // staging + queues
    DevBuf s_org, s_dir, s_out, s_accum;
    DevBuf s_work;                    // grab counter + chunk list of the large-scene ray queries
    DevBuf s_lights;                  // the point lights of the render call in progress
    uint32_t resident_blocks = 1280;  // blocks of the wide-walk query kernel the device holds at once
    Lane lane[2];                     // lane[0].stream == stream
    hipEvent_t ev_upload = nullptr;   // after the upload of a host accumulator
    uint32_t tag = 0;                 // last tag handed out; a tag is used once per scene
    // profiling
    bool profiling = false;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> pool;
    OrionKernelStats stats[ORION_KERNEL_COUNT]{};

    int mode() const {
        if (traversal == 1) return MODE_LANE;
        if (traversal == 2 && wave_ok) return MODE_WAVE;
        if (traversal == 3) return MODE_EACH;
        if (traversal == 4) return MODE_WIDE;
        // The wave-synchronous walk costs the union of the lanes' paths (subtrees no lane enters
        // are jumped over), so it pays while that union stays near one ray's own path.  Measured
        // on tessellated spheres (tools/ab/mode_threshold.py): for rays towards surface points it
        // beats the per-lane walk up to about 350 nodes, for cosine bounce rays up to about 400,
        // for rays in pixel order at every size (those take it anyway, see orion_hip_render).
        // Larger scenes: the ordered, distance-culled walk of the 4-wide tree where the scene has one.
        if (n_nodes <= 384 && wave_ok) return MODE_WAVE;
        return view.n_wide ? MODE_WIDE : MODE_LANE;
    }
    bool per_lane() const { const int m = mode(); return m == MODE_LANE || m == MODE_WIDE; }
    // Ray queries come in any order: on a large scene left to choose, each wave decides for itself.
    int query_mode() const {
        if (traversal != 0 || !per_lane() || !wave_ok) return mode();
        return mode() == MODE_WIDE ? (int)MODE_ADAPT_WIDE : (int)MODE_ADAPT;
    }
};

namespace {

// Launch on one of the scene's streams.  With profiling on, the launch carries a start and a stop
// event bound to the kernel's own dispatch (no extra packets in the stream), so the recorded time is
// the kernel's execution time and the stream runs as it does unprofiled.
template <typename... KArgs, typename... Args>
void launch_on(OrionScene* s, hipStream_t stream, int kernel_id, uint64_t items, void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
    if (!s->profiling) {
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, static_cast<KArgs>(args)...);
        return;
    }
    auto take = [&]() {
        hipEvent_t e;
        if (!s->pool.empty()) { e = s->pool.back(); s->pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    EventPair ev{take(), take(), kernel_id, items};
    hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev.a, ev.b, 0, static_cast<KArgs>(args)...);
    s->pending.push_back(ev);
}

template <typename... KArgs, typename... Args>
void launch(OrionScene* s, int kernel_id, uint64_t items, void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
    launch_on(s, s->stream, kernel_id, items, kernel, grid, block, std::forward<Args>(args)...);
}

void drain_profile(OrionScene* s) {
    for (EventPair& p : s->pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            s->stats[p.kernel].launches++;
            s->stats[p.kernel].items += p.items;
            s->stats[p.kernel].ms += ms;
        }
        s->pool.push_back(p.a); s->pool.push_back(p.b);
    }
    s->pending.clear();
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice;
}

template <typename T>
int upload(OrionScene* s, const std::vector<T>& v, const T** out) {
    void* d = nullptr;
    size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(&d, bytes ? bytes : 16));
    s->owned.push_back(d);
    if (bytes) HIP_TRY(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return 0;
}

inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

// Depth-first flattening of one mesh's tree into the threaded layout.
struct Flattener {
    const OrionMeshDesc& m;
    uint32_t slot_base;
    std::vector<Node>& nodes;
    std::vector<uint8_t> seen;
    std::string err;
    std::vector<uint32_t> where = {};   // caller's node index -> index in `nodes`

    // Emits the subtree of `ni`; returns false on malformed input.
    bool emit(uint32_t ni) {
        // explicit stack: (node, state) to keep deep trees off the C stack
        struct Item { uint32_t node; uint32_t out_index; int state; };
        std::vector<Item> st;
        st.push_back({ni, 0, 0});
        while (!st.empty()) {
            Item it = st.back();
            st.pop_back();
            if (it.state == 1) {   // subtree done: the skip link is wherever the array ends now
                nodes[it.out_index].a = (uint32_t)nodes.size();
                continue;
            }
            if (it.node >= m.n_nodes) { err = "BVH child index out of range"; return false; }
            if (seen[it.node]) { err = "BVH node reachable twice (not a tree)"; return false; }
            seen[it.node] = 1;
            if (where.empty()) where.assign(m.n_nodes, kInvalid);
            where[it.node] = (uint32_t)nodes.size();
            const OrionBvhNode& src = m.nodes[it.node];
            Node n;
            std::memcpy(n.lo, src.lo, 12); std::memcpy(n.hi, src.hi, 12);
            const uint32_t ntri = m.n_indices / 3;
            if (src.is_leaf) {
                if (src.begin > src.end || src.end > ntri) { err = "BVH leaf range outside the triangle array"; return false; }
                n.a = slot_base + src.begin;
                n.b = kLeafFlag | (src.end - src.begin);
                nodes.push_back(n);
            } else {
                n.a = 0; n.b = 0;
                const uint32_t self = (uint32_t)nodes.size();
                nodes.push_back(n);
                st.push_back({0, self, 1});
                st.push_back({src.child[1], 0, 0});
                st.push_back({src.child[0], 0, 0});
            }
        }
        return true;
    }
};

// Waits until the device has stored the three counters of a bounce under `tag` into their pinned slot.
// The wait is a poll of host memory; it gives up when the stream has drained (or failed) without
// the tag turning up.
int await_counters(hipStream_t stream, const unsigned long long* slot, uint32_t tag, uint32_t out[3]) {
    const volatile unsigned long long* v = slot;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; spin++) {
        const unsigned long long a = v[0], b = v[1], c = v[2];
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag && (uint32_t)(c >> 32) == tag) { out[0] = (uint32_t)a; out[1] = (uint32_t)b; out[2] = (uint32_t)c; return 0; }
        for (int k = 0; k < 32; k++) __builtin_ia32_pause();   // poll gently: the device's store has to win the line
        if ((spin & 0x3FF) == 0x3FF) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipErrorNotReady) {   // still running; give up only on a device that has stopped answering
                if (std::chrono::steady_clock::now() - t0 > std::chrono::minutes(10)) return fail(2, "bounce loop: no progress on the device for 10 minutes");
                continue;
            }
            if (q != hipSuccess) return fail(2, std::string("bounce loop: ") + hipGetErrorString(q));
            const unsigned long long a2 = v[0], b2 = v[1], c2 = v[2];   // the stream is idle: every store has landed
            if ((uint32_t)(a2 >> 32) == tag && (uint32_t)(b2 >> 32) == tag && (uint32_t)(c2 >> 32) == tag) { out[0] = (uint32_t)a2; out[1] = (uint32_t)b2; out[2] = (uint32_t)c2; return 0; }
            return fail(2, "bounce loop: counters were not published");
        }
    }
}

// Calls f(std::integral_constant<int, MODE>) for the scene's traversal mode.
template <typename F>
void with_mode(int mode, F&& f) {
    switch (mode) {
        case MODE_WAVE: f(std::integral_constant<int, MODE_WAVE>()); break;
        case MODE_EACH: f(std::integral_constant<int, MODE_EACH>()); break;
        case MODE_WIDE: f(std::integral_constant<int, MODE_WIDE>()); break;
        default: f(std::integral_constant<int, MODE_LANE>()); break;
    }
}

// ... and for the ray-query kernels, which also come in the per-wave adaptive form.
template <typename F>
void with_query_mode(int mode, F&& f) {
    if (mode == MODE_ADAPT) f(std::integral_constant<int, MODE_ADAPT>());
    else if (mode == MODE_ADAPT_WIDE) f(std::integral_constant<int, MODE_ADAPT_WIDE>());
    else with_mode(mode, f);
}

// How many rays a wave of the ray-drawing kernels takes at a time, and how many blocks to launch, for n rays:
// about four grabs per wave, 64 to kQueryGrab rays each; no more blocks than the chip holds at once and
// none without a grab for each of its waves.
void draw_plan(const OrionScene* s, uint32_t n, uint32_t* grab, uint32_t* grid) {
    const uint64_t waves = (uint64_t)s->resident_blocks * (kBlock / 64);
    uint32_t chunks = (uint32_t)(n / (waves * 4 * 64));
    chunks = chunks < 1 ? 1 : (chunks > kQueryGrab / 64 ? kQueryGrab / 64 : chunks);
    *grab = chunks * 64;
    const uint32_t want = (uint32_t)(((uint64_t)n + *grab * (kBlock / 64) - 1) / (*grab * (kBlock / 64)));
    *grid = want < s->resident_blocks ? (want ? want : 1u) : s->resident_blocks;
}

// Ray queries on a large scene: the waves of a launch that just fills the chip draw rays from a global
// counter (k_query_wide).  Left to choose, a first pass sends the 64-ray chunks that are coherent through the
// wave-synchronous walk and lists the others.
template <bool ANY>
int launch_query_wide(OrionScene* s, bool adapt, const float* d_org, const float* d_dir, uint32_t n, OrionHit* d_hits, uint8_t* d_occ,
                      uint32_t* work, bool plain) {
    const int kid = ANY ? ORION_KERNEL_ANY_HIT : ORION_KERNEL_CLOSEST_HIT;
    const uint32_t n_chunks = (n + 63u) / 64u;
    uint32_t* chunks = work + kWorkWords;   // the work words, then the list
    HIP_TRY(hipMemsetAsync(work, 0, kWorkWords * 4, s->stream));
    uint32_t grab, grid;
    draw_plan(s, n, &grab, &grid);
    const dim3 cgrid((n_chunks + kCoherentChunks - 1) / kCoherentChunks);
    if (plain) {   // no profiling events: nothing here may allocate
        if (adapt) {
            hipLaunchKernelGGL(k_query_coherent<ANY>, cgrid, dim3(kBlock), 0, s->stream, s->view, d_org, d_dir, n, d_hits, d_occ, work, chunks);
            hipLaunchKernelGGL((k_query_wide<ANY, true>), dim3(grid), dim3(kBlock), 0, s->stream, s->view, d_org, d_dir, n, d_hits, d_occ, grab, work, (const uint32_t*)chunks);
        } else {
            hipLaunchKernelGGL((k_query_wide<ANY, false>), dim3(grid), dim3(kBlock), 0, s->stream, s->view, d_org, d_dir, n, d_hits, d_occ, grab, work, (const uint32_t*)chunks);
        }
    } else {
        if (adapt) launch(s, kid, 0, k_query_coherent<ANY>, cgrid, dim3(kBlock), s->view, d_org, d_dir, n, d_hits, d_occ, work, chunks);
        if (adapt) launch(s, kid, n, k_query_wide<ANY, true>, dim3(grid), dim3(kBlock), s->view, d_org, d_dir, n, d_hits, d_occ, grab, work, (const uint32_t*)chunks);
        else launch(s, kid, n, k_query_wide<ANY, false>, dim3(grid), dim3(kBlock), s->view, d_org, d_dir, n, d_hits, d_occ, grab, work, (const uint32_t*)chunks);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The scene's own work buffer for the wide ray queries (allocated on first use, grown as needed).
int query_work(OrionScene* s, bool adapt, uint32_t n, uint32_t** work) {
    if (int rc = s->s_work.reserve(kWorkWords * 4 + (adapt ? (size_t)((n + 63u) / 64u) * 4 : 0))) return rc;
    *work = static_cast<uint32_t*>(s->s_work.p);
    return 0;
}

int launch_closest(OrionScene* s, const float* d_org, const float* d_dir, uint32_t n, OrionHit* d_out) {
    const int qm = s->query_mode();
    if (qm == MODE_WIDE || qm == MODE_ADAPT_WIDE) {
        uint32_t* work = nullptr;
        if (int rc = query_work(s, qm == MODE_ADAPT_WIDE, n, &work)) return rc;
        return launch_query_wide<false>(s, qm == MODE_ADAPT_WIDE, d_org, d_dir, n, d_out, nullptr, work, false);
    }
    with_query_mode(qm, [&](auto M) {
        launch(s, ORION_KERNEL_CLOSEST_HIT, n, k_closest_api<decltype(M)::value>, dim3(grid_for(n)), dim3(kBlock), s->view, d_org, d_dir, n, d_out);
    });
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_anyhit(OrionScene* s, const float* d_org, const float* d_dir, uint32_t n, uint8_t* d_out) {
    const int qm = s->query_mode();
    if (qm == MODE_WIDE || qm == MODE_ADAPT_WIDE) {
        uint32_t* work = nullptr;
        if (int rc = query_work(s, qm == MODE_ADAPT_WIDE, n, &work)) return rc;
        return launch_query_wide<true>(s, qm == MODE_ADAPT_WIDE, d_org, d_dir, n, nullptr, d_out, work, false);
    }
    with_query_mode(qm, [&](auto M) {
        launch(s, ORION_KERNEL_ANY_HIT, n, k_anyhit_api<decltype(M)::value>, dim3(grid_for(n)), dim3(kBlock), s->view, d_org, d_dir, n, d_out);
    });
    HIP_TRY(hipGetLastError());
    return 0;
}

// Stage n*3 floats of ray data on the device if the caller's array is host memory.
int stage_in(OrionScene* s, DevBuf& buf, const float* any, uint32_t n, const float** dev) {
    if (is_device_ptr(any)) { *dev = any; return 0; }
    if (int rc = buf.reserve((size_t)n * 12)) return rc;
    HIP_TRY(hipMemcpyAsync(buf.p, any, (size_t)n * 12, hipMemcpyHostToDevice, s->stream));
    *dev = static_cast<const float*>(buf.p);
    return 0;
}

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// host-side vector helpers for the camera basis (same formulas as the kernels)
struct H3 { float x, y, z; };
inline float hdot(H3 a, H3 b) { float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z; return (p0 + p1) + (p2 + 0.0f); }
inline H3 hcross(H3 a, H3 b) {
    return {std::fmaf(-a.z, b.y, a.y * b.z), std::fmaf(-a.x, b.z, a.z * b.x), std::fmaf(-a.y, b.x, a.x * b.y)};
}
inline H3 hscale(H3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline H3 hnormalize(H3 a) { return hscale(a, 1.0f / std::sqrt(hdot(a, a))); }

} // namespace

extern "C" {

const char* orion_hip_last_error(void) { return g_error.c_str(); }

int orion_hip_scene_create(const OrionMeshDesc* meshes, uint32_t n_meshes,
                           const OrionMaterialDesc* materials, uint32_t n_materials,
                           const OrionTextureDesc* textures, uint32_t n_textures,
                           int device, OrionScene** out) {
    if (!out) return fail(1, "scene_create: out is null");
    *out = nullptr;
    if (n_meshes && !meshes) return fail(1, "scene_create: meshes is null");
    if (n_materials && !materials) return fail(1, "scene_create: materials is null");
    if (n_textures && !textures) return fail(1, "scene_create: textures is null");

    // ---- validate and flatten on the host -------------------------------------
    std::vector<Node> nodes;
    std::vector<Tri> tris;
    std::vector<TriAttr> attrs;
    std::vector<uint32_t> slot_of;
    std::vector<Mesh> dmeshes(n_meshes);
    std::vector<uint32_t> emissive;

    for (uint32_t t = 0; t < n_textures; t++) {
        const OrionTextureDesc& td = textures[t];
        if (td.width == 0 || td.height == 0 || !td.texels) return fail(1, "scene_create: texture " + std::to_string(t) + " is empty");
        if (td.format > ORION_TEX_GRAY_U8) return fail(1, "scene_create: texture " + std::to_string(t) + " has an unknown format");
        if ((uint64_t)td.width * td.height > 0x7FFFFFFFull / 4) return fail(1, "scene_create: texture " + std::to_string(t) + " is too large");
    }
    for (uint32_t k = 0; k < n_materials; k++) {
        const OrionMaterialDesc& md = materials[k];
        const uint32_t ids[4] = {md.tex_ambient, md.tex_diffuse, md.tex_specular, md.tex_emissive};
        for (uint32_t id : ids)
            if (id >= n_textures) return fail(1, "scene_create: material " + std::to_string(k) + " names texture " + std::to_string(id) + " of " + std::to_string(n_textures));
    }

    uint32_t tri_base = 0;
    std::vector<uint32_t> slot_bases;
    std::vector<std::vector<uint32_t>> node_where;   // per mesh: caller's node index -> index in `nodes`
    for (uint32_t mi = 0; mi < n_meshes; mi++) {
        const OrionMeshDesc& m = meshes[mi];
        const std::string where = "scene_create: mesh " + std::to_string(mi) + ": ";
        if (m.n_indices % 3) return fail(1, where + "index count is not a multiple of 3");
        if (m.material >= n_materials) return fail(1, where + "material index out of range");
        const uint32_t ntri = m.n_indices / 3;
        if ((ntri && (!m.vertices || !m.indices || !m.tri_order)) || (m.n_nodes && !m.nodes)) return fail(1, where + "null array");
        if ((uint64_t)tri_base + ntri >= kLeafFlag) return fail(1, where + "too many triangles");
        for (uint32_t i = 0; i < m.n_indices; i++)
            if (m.indices[i] >= m.n_vertices) return fail(1, where + "vertex index out of range");
        std::vector<uint8_t> used(ntri, 0);
        for (uint32_t i = 0; i < ntri; i++) {
            if (m.tri_order[i] >= ntri || used[m.tri_order[i]]) return fail(1, where + "tri_order is not a permutation");
            used[m.tri_order[i]] = 1;
        }

        const uint32_t slot_base = (uint32_t)tris.size();
        slot_bases.push_back(slot_base);
        node_where.emplace_back();
        if (m.n_nodes) {
            Flattener fl{m, slot_base, nodes, std::vector<uint8_t>(m.n_nodes, 0), std::string()};
            if (!fl.emit(0)) return fail(1, where + fl.err);
            node_where.back() = std::move(fl.where);
        }
        // triangle slots in tri_order (= visit order inside the mesh)
        const OrionMaterialDesc& md = materials[m.material];
        float area = 0.0f;
        slot_of.resize(tri_base + ntri);
        for (uint32_t i = 0; i < ntri; i++) {
            const uint32_t o = m.tri_order[i];
            const float* p0 = m.vertices + 8ull * m.indices[3 * o];
            const float* p1 = m.vertices + 8ull * m.indices[3 * o + 1];
            const float* p2 = m.vertices + 8ull * m.indices[3 * o + 2];
            Tri t{};
            for (int a = 0; a < 3; a++) { t.v0[a] = p0[a]; t.e1[a] = p1[a] - p0[a]; t.e2[a] = p2[a] - p0[a]; }
            t.gtri = tri_base + o;
            slot_of[tri_base + o] = (uint32_t)tris.size();
            tris.push_back(t);
        }
        attrs.resize(tri_base + ntri);
        for (uint32_t o = 0; o < ntri; o++) {   // original order: attributes and the area sum
            const float* p0 = m.vertices + 8ull * m.indices[3 * o];
            const float* p1 = m.vertices + 8ull * m.indices[3 * o + 1];
            const float* p2 = m.vertices + 8ull * m.indices[3 * o + 2];
            TriAttr& a = attrs[tri_base + o];
            for (int c = 0; c < 3; c++) { a.n0[c] = p0[3 + c]; a.n1[c] = p1[3 + c]; a.n2[c] = p2[3 + c]; }
            a.uv0x = p0[6]; a.uv0y = p0[7]; a.uv1x = p1[6]; a.uv1y = p1[7]; a.uv2x = p2[6]; a.uv2y = p2[7];
            a.mesh = mi;
            H3 e1 = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2 = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
            area += std::fabs(hdot(e1, e2)) * 0.5f;   // orion/geometry.hpp:155-157, orion/mesh.hpp:197-208
        }
        for (uint32_t i = 0; i < ntri; i++) {   // per-triangle light-sampling constants, now that the area is known
            Tri& t = tris[slot_base + i];
            const H3 e1 = {t.e1[0], t.e1[1], t.e1[2]}, e2 = {t.e2[0], t.e2[1], t.e2[2]};
            const float tri_area = std::fabs(hdot(e1, e2)) * 0.5f;
            t.light_bias = (tri_area / area) * (float)ntri;
            t.light_intensity = t.light_bias * area;
        }
        Mesh& dm = dmeshes[mi];
        dm.tri_base = tri_base; dm.n_tri = ntri;
        dm.tex[0] = md.tex_ambient; dm.tex[1] = md.tex_diffuse; dm.tex[2] = md.tex_specular; dm.tex[3] = md.tex_emissive;
        dm.shininess = md.shininess; dm.area = area;
        const bool is_emissive = md.emissive_color[0] != 0.0f || md.emissive_color[1] != 0.0f || md.emissive_color[2] != 0.0f;
        if (is_emissive && ntri) emissive.push_back(mi);
        dm.flat = 0; dm.flat_n[0] = dm.flat_n[1] = dm.flat_n[2] = 0.0f;
        dm.one_normal = 0; dm.raw_n[0] = dm.raw_n[1] = dm.raw_n[2] = 0.0f;
        dm.light_tris = 0;
        if (is_emissive && ntri) {   // one normal for the whole light, finite intensities: see Mesh::flat
            const float* n0 = attrs[tri_base].n0;
            bool one = true, finite = true;
            for (uint32_t o = 0; o < ntri; o++) {
                const TriAttr& a = attrs[tri_base + o];
                one = one && !std::memcmp(a.n0, n0, 12) && !std::memcmp(a.n1, n0, 12) && !std::memcmp(a.n2, n0, 12);
                finite = finite && std::isfinite(tris[slot_base + o].light_intensity);
            }
            if (one) { dm.one_normal = 1; std::memcpy(dm.raw_n, n0, 12); }
            const double len = std::sqrt((double)n0[0] * n0[0] + (double)n0[1] * n0[1] + (double)n0[2] * n0[2]);
            if (one && finite && std::isfinite(len) && len > 1e-18) {
                for (int c = 0; c < 3; c++) dm.flat_n[c] = (float)(n0[c] / len);
                dm.flat = 1;
            }
        }
        tri_base += ntri;
    }

    // One-ray-per-wave table: inner nodes and triangles in visit order, one per lane, each with the
    // set of inner nodes above it (the nodes whose subtree [index, skip) contains it).
    std::vector<WaveItem> items;
    {
        size_t n_inner = 0;
        for (const Node& nd : nodes) if (!(nd.b & kLeafFlag)) n_inner++;
        if (!nodes.empty() && n_inner + tris.size() <= 64) {
            std::vector<uint32_t> item_of_node(nodes.size(), kInvalid);
            for (uint32_t ni = 0; ni < nodes.size(); ni++) {
                const Node& nd = nodes[ni];
                uint64_t anc = 0;
                for (uint32_t a = 0; a < ni; a++)
                    if (!(nodes[a].b & kLeafFlag) && nodes[a].a > ni) anc |= 1ull << item_of_node[a];
                WaveItem it{};
                it.anc_lo = (uint32_t)anc; it.anc_hi = (uint32_t)(anc >> 32);
                if (!(nd.b & kLeafFlag)) {
                    item_of_node[ni] = (uint32_t)items.size();
                    std::memcpy(it.a, nd.lo, 12); std::memcpy(it.b, nd.hi, 12);
                    it.kind = 1;
                    items.push_back(it);
                } else {
                    for (uint32_t k = 0; k < (nd.b & ~kLeafFlag); k++) {
                        const Tri& t = tris[nd.a + k];
                        std::memcpy(it.a, t.v0, 12); std::memcpy(it.b, t.e1, 12); std::memcpy(it.c, t.e2, 12);
                        it.gtri = t.gtri; it.kind = 2;
                        items.push_back(it);
                    }
                }
            }
            items.resize(64, WaveItem{});   // every lane reads an entry
        }
    }

    // The 4-wide tree of the ordered, distance-culled walk: for scenes too large for the wave-synchronous
    // walk, where the trees allow it (wide_bvh.h) and the walk's stack is deep enough for most paths.
    orion_wide::WideTree wide;
    if (nodes.size() > 64) wide = orion_wide::build_wide_tree(meshes, n_meshes, slot_bases, tris);
    else wide.why_not = "the scene is small";

    std::vector<Texture> dtex(n_textures);
    std::vector<float> texf;
    std::vector<uint8_t> texb;
    bool texels_finite = true;   // (bytes always are)
    for (uint32_t t = 0; t < n_textures; t++) {
        const OrionTextureDesc& td = textures[t];
        const size_t px = (size_t)td.width * td.height;
        dtex[t].width = td.width; dtex[t].height = td.height; dtex[t].format = td.format;
        if (td.format == ORION_TEX_RGB_F32) {
            dtex[t].offset = (uint32_t)texf.size();
            const float* src = static_cast<const float*>(td.texels);
            texf.insert(texf.end(), src, src + 3 * px);
            for (size_t i = 0; i < 3 * px && texels_finite; i++) texels_finite = std::isfinite(src[i]);
        } else {
            const size_t bpp = td.format == ORION_TEX_RGB_U8 ? 3 : (td.format == ORION_TEX_RGBA_U8 ? 4 : 1);
            dtex[t].offset = (uint32_t)texb.size();
            const uint8_t* src = static_cast<const uint8_t*>(td.texels);
            texb.insert(texb.end(), src, src + bpp * px);
        }
    }

    if (!texels_finite)
        for (Mesh& dm : dmeshes) dm.flat = 0;
    for (Mesh& dm : dmeshes) {   // solid colours, as Texture::color would return them
        auto solid = [&](uint32_t t, float* rgb) -> uint32_t {
            const OrionTextureDesc& td = textures[t];
            if (td.width != 1 || td.height != 1) return 0;
            if (td.format == ORION_TEX_RGB_F32) { std::memcpy(rgb, td.texels, 12); return 1; }
            const uint8_t* q = static_cast<const uint8_t*>(td.texels);
            if (td.format == ORION_TEX_GRAY_U8) { rgb[0] = rgb[1] = rgb[2] = (float)q[0] / 255.0f; return 1; }
            for (int c = 0; c < 3; c++) rgb[c] = (float)q[c] / 255.0f;
            return 1;
        };
        dm.kd_solid = n_textures ? solid(dm.tex[1], dm.kd) : 0;
        dm.ke_solid = n_textures ? solid(dm.tex[3], dm.ke) : 0;
    }
    // The lights' triangles once more, in their original order, behind the array's spare record.
    std::vector<Tri> tris_up = tris;
    tris_up.push_back(Tri{});
    for (uint32_t mi : emissive) {
        dmeshes[mi].light_tris = (uint32_t)tris_up.size();
        for (uint32_t o = 0; o < dmeshes[mi].n_tri; o++) tris_up.push_back(tris[slot_of[dmeshes[mi].tri_base + o]]);
    }
    if (tris_up.size() >= 0xFFFFFFFFull) return fail(1, "scene_create: too many triangles");

    // ---- upload ----------------------------------------------------------------
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return fail(2, "no HIP device"); }
    DeviceGuard guard(device);
    if (!guard.ok) return fail(2, "cannot select HIP device " + std::to_string(device));
    if (int rc = resolve_noalloc_scratch(device)) return rc;
    OrionScene* s = new OrionScene;
    s->device = device;
    auto bail = [&](int rc) { orion_hip_scene_destroy(s); return rc; };
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { s->stream = nullptr; return bail(fail(2, "hipStreamCreate failed")); }
    s->lane[0].stream = s->stream;
    if (hipStreamCreateWithFlags(&s->lane[1].stream, hipStreamNonBlocking) != hipSuccess) { s->lane[1].stream = nullptr; return bail(fail(2, "hipStreamCreate failed")); }
    for (Lane& L : s->lane) {
        if (hipHostMalloc((void**)&L.h_count, kRing * 32, hipHostMallocCoherent) != hipSuccess) { L.h_count = nullptr; return bail(fail(2, "hipHostMalloc failed")); }
        std::memset(L.h_count, 0, kRing * 32);
        if (hipEventCreateWithFlags(&L.resolved, hipEventDisableTiming) != hipSuccess) { L.resolved = nullptr; return bail(fail(2, "hipEventCreate failed")); }
    }
    if (hipEventCreateWithFlags(&s->ev_upload, hipEventDisableTiming) != hipSuccess) { s->ev_upload = nullptr; return bail(fail(2, "hipEventCreate failed")); }
    SceneView& v = s->view;
    int rc = 0;
    v.n_nodes = (uint32_t)nodes.size(); v.n_tris = (uint32_t)tris.size();
    // One spare record behind each array: the walk that asks for the next node ahead of time (traverse_wave_ahead)
    // reads one past the node it is at.
    // The spare node record holds the box around all the meshes' roots (traverse_wave tests it first on large scenes).
    // A mesh whose root is a leaf has no box in the reference's walk (orion/avx/sbvh.cpp:50-59: its triangles are tested
    // whatever the ray): with such a mesh in the scene the record holds all of space, which every ray the gate sees meets.
    {
        Node all{};
        for (int a = 0; a < 3; a++) { all.lo[a] = INFINITY; all.hi[a] = -INFINITY; }
        bool leaf_root = false;
        for (const std::vector<uint32_t>& w : node_where)
            if (!w.empty() && w[0] != kInvalid) {
                leaf_root = leaf_root || (nodes[w[0]].b & kLeafFlag) != 0u;
                for (int a = 0; a < 3; a++) { all.lo[a] = std::fmin(all.lo[a], nodes[w[0]].lo[a]); all.hi[a] = std::fmax(all.hi[a], nodes[w[0]].hi[a]); }
            }
        if (leaf_root) for (int a = 0; a < 3; a++) { all.lo[a] = -INFINITY; all.hi[a] = INFINITY; }
        nodes.push_back(all);
    }
    if ((rc = upload(s, nodes, &v.nodes))) return bail(rc);
    if ((rc = upload(s, tris_up, &v.tris))) return bail(rc);
    nodes.pop_back();
    if ((rc = upload(s, attrs, &v.attrs))) return bail(rc);
    if ((rc = upload(s, slot_of, &v.slot_of))) return bail(rc);
    if ((rc = upload(s, dmeshes, &v.meshes))) return bail(rc);
    if ((rc = upload(s, dtex, &v.textures))) return bail(rc);
    if ((rc = upload(s, texf, &v.texels_f32))) return bail(rc);
    if ((rc = upload(s, texb, &v.texels_u8))) return bail(rc);
    if ((rc = upload(s, emissive, &v.emissive))) return bail(rc);
    v.n_nodes = (uint32_t)nodes.size(); v.n_tris = (uint32_t)tris.size();
    v.n_meshes = n_meshes; v.n_emissive = (uint32_t)emissive.size();
    if (!items.empty()) { if ((rc = upload(s, items, &v.items))) return bail(rc); v.n_items = 64; }
    if (!wide.nodes.empty()) {
        // the binary node each wide node stands for, as an index into the threaded array (the walk hands a subtree
        // its stack cannot hold to the unculled walk)
        std::vector<uint32_t> wide_bin(wide.nodes.size(), kInvalid);
        for (size_t i = 0; i < wide.nodes.size(); i++)
            if (wide.src_mesh[i] != kInvalid) wide_bin[i] = node_where[wide.src_mesh[i]][wide.src_node[i]];
        if ((rc = upload(s, wide_bin, &v.wide_bin))) return bail(rc);
        if ((rc = upload(s, wide.nodes, &v.wide))) return bail(rc);
        v.n_wide = (uint32_t)wide.nodes.size(); v.wide_n_virtual = wide.n_virtual; v.wide_pad = wide.pad; v.wide_grow0 = wide.pad * (1.0f + 2.0f * kWideSlack); v.wide_org_bound = wide.org_bound;
        // The guard's table (wide_guard.h): which slots may hold an ill-conditioned triangle, per node and cell of a grid over the
        // ray directions; built on the device, bottom-up (children stand behind their parents in the node array).
        {
            uint32_t G = guard_grid_for(wide.nodes.size());
            if (const char* e = std::getenv("ORION_HIP_GUARD_GRID")) { const int g = std::atoi(e); if (g == 32 || g == 64 || g == 128 || g == 256) G = (uint32_t)g; }   // (A/B runs)
            const size_t bytes = guard_table_bytes(wide.nodes.size(), G);
            const GuardBand* d_bands = nullptr;
            if ((rc = upload(s, wide.bands, &d_bands))) return bail(rc);
            void* table = nullptr;
            if (hipMalloc(&table, bytes) != hipSuccess) { (void)hipGetLastError(); return bail(fail(2, "scene_create: out of device memory for the wide walk's guard table")); }
            s->owned.push_back(table);
            bool ok = hipMemsetAsync(table, 0, bytes, s->stream) == hipSuccess;
            if (ok) hipLaunchKernelGGL(k_guard_cells, dim3((G * G + kBlock - 1) / kBlock), dim3(kBlock), 0, s->stream, v.wide, v.n_wide, v.wide_n_virtual, d_bands, G, static_cast<uint8_t*>(table));
            if (!ok || hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) return bail(fail(2, "scene_create: building the wide walk's guard table failed"));
            v.wide_cells = static_cast<const uint8_t*>(table); v.wide_cells_grid = G;
        }
        v.wide_guard = 1u; v.wide_guard_dlen = wide.guard_dlen;
    }
    s->wide_why_not = wide.why_not;
    {   // |det| = |e1 . (d x e2)| <= 6 D E^2 for direction components up to D and edge components up to E;
        // D < 2^122 / E^2 keeps it (and every intermediate, rounding included) below 2^126 with room to spare
        double E = 0.0;
        bool finite = true;
        for (const Tri& t : tris)
            for (int k = 0; k < 3; k++) {
                finite = finite && std::isfinite(t.e1[k]) && std::isfinite(t.e2[k]);
                E = std::fmax(E, std::fmax(std::fabs((double)t.e1[k]), std::fabs((double)t.e2[k])));
            }
        double bound = (finite && E > 0.0) ? std::ldexp(1.0, 122) / (E * E) : (finite ? std::ldexp(1.0, 126) : 0.0);
        if (bound > std::ldexp(1.0, 126)) bound = std::ldexp(1.0, 126);
        v.dir_bound = (float)(bound * (1.0 - 1e-6));
    }
    s->n_nodes = v.n_nodes; s->n_tris = v.n_tris;
    s->solid = n_meshes != 0;
    for (const Mesh& dm : dmeshes) s->solid = s->solid && dm.kd_solid && dm.ke_solid;
    for (uint32_t mi : emissive) s->solid = s->solid && dmeshes[mi].one_normal;
    s->wave_ok = (uint64_t)tris_up.size() * sizeof(Tri) < (1ull << 32) && (uint64_t)(nodes.size() + 1) * sizeof(Node) < (1ull << 32);   // (+ the spare records)
    {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_query_wide<false, false>, kBlock, 0) == hipSuccess && per_cu > 0)
            s->resident_blocks = (uint32_t)cus * (uint32_t)per_cu;
    }
    for (const Mesh& dm : dmeshes) s->mesh_tri_base.push_back(dm.tri_base);
    *out = s;
    return 0;
}

int orion_hip_scene_destroy(OrionScene* s) {
    if (!s) return 0;
    DeviceGuard guard(s->device);
    for (Lane& L : s->lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
    drain_profile(s);
    for (hipEvent_t e : s->pool) (void)hipEventDestroy(e);
    for (void* p : s->owned) (void)hipFree(p);
    s->s_org.release(); s->s_dir.release(); s->s_out.release(); s->s_accum.release(); s->s_work.release(); s->s_lights.release();
    for (Lane& L : s->lane) {
        for (DevBuf& b : L.q) b.release();
        L.hits.release(); L.rad.release(); L.counter.release(); L.pre_hits.release(); L.pre_work.release(); L.jitter.release(); L.hit_mask.release();
        if (L.h_count) (void)hipHostFree(L.h_count);
        if (L.resolved) (void)hipEventDestroy(L.resolved);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (s->ev_upload) (void)hipEventDestroy(s->ev_upload);
    delete s;
    return 0;
}

int orion_hip_scene_set_option(OrionScene* s, const char* name, int64_t value) {
    if (!s || !name) return fail(1, "set_option: null argument");
    if (!std::strcmp(name, "traversal")) {
        if (value < 0 || value > 4) return fail(1, "set_option: traversal must be 0, 1, 2, 3 or 4");
        if (value == 2 && !s->wave_ok) return fail(1, "set_option: the scene's arrays reach 4 GiB, beyond the wave-synchronous walk's 32-bit offsets");
        if (value == 4 && s->view.n_wide == 0) return fail(1, "set_option: the scene has no wide tree (" + s->wide_why_not + ")");
        if (value == 3 && s->view.n_items == 0) return fail(1, "set_option: the scene is too large for one ray per wave (more than 64 inner nodes + triangles)");
        s->traversal = (int)value;
        return 0;
    }
    if (!std::strcmp(name, "wide_guard")) {
        if (value < 0 || value > 1) return fail(1, "set_option: wide_guard must be 0 or 1");
        s->view.wide_guard = (uint32_t)value;
        return 0;
    }
    if (!std::strcmp(name, "tail_paths")) {
        if (value < 0 || value > (int64_t)1 << 24) return fail(1, "set_option: tail_paths out of range");
        s->tail_paths = (uint32_t)value;
        return 0;
    }
    if (!std::strcmp(name, "overlap_below")) {
        if (value < 0 || value > (int64_t)1 << 30) return fail(1, "set_option: overlap_below out of range");
        s->overlap_below = (uint32_t)value;
        return 0;
    }
    if (!std::strcmp(name, "queue_parts")) {
        if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return fail(1, "set_option: queue_parts must be 1, 2, 4, 8 or 16");
        s->queue_parts_log2 = value == 1 ? 0u : value == 2 ? 1u : value == 4 ? 2u : value == 8 ? 3u : 4u;
        return 0;
    }
    if (!std::strcmp(name, "draw_rays")) {
        if (value < 0 || value > 1) return fail(1, "set_option: draw_rays must be 0 or 1");
        s->draw_rays = value != 0;
        return 0;
    }
    if (!std::strcmp(name, "wave_ahead")) {
        if (value < 0 || value > 1) return fail(1, "set_option: wave_ahead must be 0 or 1");
        s->wave_ahead = value != 0;
        return 0;
    }
    if (!std::strcmp(name, "queue_budget_mib")) {
        if (value < 0 || value > (int64_t)1 << 30) return fail(1, "set_option: queue_budget_mib out of range");
        s->queue_budget = (uint64_t)value << 20;
        return 0;
    }
    // Test hook, not part of the documented options: only a process started with ORION_HIP_TEST_HOOKS=1 can arm it, for the
    // next render call only.  The n-th bounce that call launches fails with code 3, whatever is in flight.
    if (!std::strcmp(name, "fail_queue_allocs")) {   // (the other test hook: the next n attempts to take a lane's queues fail as if out of memory)
        const char* hooks = std::getenv("ORION_HIP_TEST_HOOKS");
        if (!hooks || std::strcmp(hooks, "1") != 0) return fail(1, "set_option: unknown option 'fail_queue_allocs'");
        if (value < 0 || value > 64) return fail(1, "set_option: fail_queue_allocs out of range");
        s->fail_queue_allocs = (uint32_t)value;
        return 0;
    }
    if (!std::strcmp(name, "fail_at_bounce")) {
        const char* hooks = std::getenv("ORION_HIP_TEST_HOOKS");
        if (!hooks || std::strcmp(hooks, "1") != 0) return fail(1, "set_option: unknown option 'fail_at_bounce'");
        if (value < 0 || value > (int64_t)1 << 30) return fail(1, "set_option: fail_at_bounce out of range");
        s->fail_at_bounce = (uint32_t)value;
        return 0;
    }
    if (!std::strcmp(name, "batch_paths")) {
        if (value < 1 || value > (int64_t)1 << 30) return fail(1, "set_option: batch_paths out of range");
        s->batch_paths = (uint64_t)value;
        return 0;
    }
    return fail(1, std::string("set_option: unknown option '") + name + "'");
}

int orion_hip_closest_hit(OrionScene* s, const float* org_any, const float* dir_any, uint32_t n, OrionHit* out_any) {
    if (!s) return fail(1, "closest_hit: scene is null");
    if (n == 0) return 0;
    if (!org_any || !dir_any || !out_any) return fail(1, "closest_hit: null array");
    DeviceGuard guard(s->device);
    const float *d_org, *d_dir;
    if (int rc = stage_in(s, s->s_org, org_any, n, &d_org)) return rc;
    if (int rc = stage_in(s, s->s_dir, dir_any, n, &d_dir)) return rc;
    const bool out_dev = is_device_ptr(out_any);
    OrionHit* d_out = out_any;
    if (!out_dev) {
        if (int rc = s->s_out.reserve((size_t)n * sizeof(OrionHit))) return rc;
        d_out = static_cast<OrionHit*>(s->s_out.p);
    }
    if (int rc = launch_closest(s, d_org, d_dir, n, d_out)) return rc;
    if (!out_dev) HIP_TRY(hipMemcpyAsync(out_any, d_out, (size_t)n * sizeof(OrionHit), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int orion_hip_any_hit(OrionScene* s, const float* org_any, const float* dir_any, uint32_t n, uint8_t* occluded_any) {
    if (!s) return fail(1, "any_hit: scene is null");
    if (n == 0) return 0;
    if (!org_any || !dir_any || !occluded_any) return fail(1, "any_hit: null array");
    DeviceGuard guard(s->device);
    const float *d_org, *d_dir;
    if (int rc = stage_in(s, s->s_org, org_any, n, &d_org)) return rc;
    if (int rc = stage_in(s, s->s_dir, dir_any, n, &d_dir)) return rc;
    const bool out_dev = is_device_ptr(occluded_any);
    uint8_t* d_out = occluded_any;
    if (!out_dev) {
        if (int rc = s->s_out.reserve(n)) return rc;
        d_out = static_cast<uint8_t*>(s->s_out.p);
    }
    if (int rc = launch_anyhit(s, d_org, d_dir, n, d_out)) return rc;
    if (!out_dev) HIP_TRY(hipMemcpyAsync(occluded_any, d_out, n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int orion_hip_closest_hit_noalloc(OrionScene* s, const float* d_org, const float* d_dir, uint32_t n, OrionHit* d_out) {
    if (!s) return fail(1, "closest_hit_noalloc: scene is null");
    if (n == 0) return 0;
    if (!d_org || !d_dir || !d_out) return fail(1, "closest_hit_noalloc: null array");
    if (n > kNoAllocMaxRays) return fail(1, "closest_hit_noalloc: at most 16777216 rays per call");
    DeviceGuard guard(s->device);
    if (!is_device_ptr(d_org) || !is_device_ptr(d_dir) || !is_device_ptr(d_out))
        return fail(1, "closest_hit_noalloc: the arrays must be device memory (staging host arrays would need an allocation)");
    const int qm = s->query_mode();
    if (qm == MODE_WIDE || qm == MODE_ADAPT_WIDE) {
        uint32_t* work = (s->device >= 0 && s->device < kMaxDevices) ? g_noalloc_scratch[s->device] : nullptr;
        if (!work) return fail(2, "closest_hit_noalloc: no scratch on this device");
        // One copy of the scratch per GPU: a call that finds it in use (another scene, another thread) is refused.
        if (g_noalloc_busy[s->device].exchange(1, std::memory_order_acquire) != 0)
            return fail(4, "closest_hit_noalloc: this GPU's scratch is in use by another call (calls on one GPU must not overlap)");
        int rc = launch_query_wide<false>(s, qm == MODE_ADAPT_WIDE, d_org, d_dir, n, d_out, nullptr, work, true);
        const hipError_t e = rc ? hipSuccess : hipStreamSynchronize(s->stream);   // the scratch is free once the kernels are done
        g_noalloc_busy[s->device].store(0, std::memory_order_release);
        if (rc) return rc;
        if (e != hipSuccess) return fail(2, std::string("closest_hit_noalloc: ") + hipGetErrorString(e));
        return 0;
    } else {   // the other walks keep nothing beyond their inputs and outputs
        with_query_mode(qm, [&](auto M) {
            hipLaunchKernelGGL(k_closest_api<decltype(M)::value>, dim3(grid_for(n)), dim3(kBlock), 0, s->stream, s->view, d_org, d_dir, n, d_out);
        });
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int orion_hip_render(OrionScene* s, const OrionCamera* cam, const OrionRenderParams* p, float* accum_any) {
    if (!s || !cam || !p || !accum_any) return fail(1, "render: null argument");
    if (p->width == 0 || p->height == 0) return fail(1, "render: empty image");
    if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(1, "render: image too large");
    if (p->spp_end < p->spp_begin) return fail(1, "render: spp_end < spp_begin");
    if (p->n_lights && !p->lights) return fail(1, "render: lights is null");
    if (p->n_lights == 0 && p->light_samples == 0 && s->view.n_emissive) return fail(1, "render: light_samples must be >= 1");
    uint32_t x0 = p->tile_x0, y0 = p->tile_y0, x1 = p->tile_x1, y1 = p->tile_y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = p->width; y1 = p->height; }
    if (x1 > p->width || y1 > p->height || x0 >= x1 || y0 >= y1) return fail(1, "render: tile rectangle outside the image");
    if (p->spp_end == p->spp_begin) return 0;
    DeviceGuard guard(s->device);

    FrameParams F{};
    std::memcpy(F.eye, cam->eye, 12); std::memcpy(F.front, cam->front, 12);
    std::memcpy(F.right, cam->right, 12); std::memcpy(F.up, cam->up, 12);
    F.width = p->width; F.height = p->height;
    F.tile_x0 = x0; F.tile_y0 = y0; F.tile_w = x1 - x0; F.tile_h = y1 - y0;
    F.light_samples = p->light_samples; F.max_depth = p->max_depth;
    F.inv_light_samples = 1.0f / (float)p->light_samples;
    F.seed_lo = (uint32_t)p->seed; F.seed_hi = (uint32_t)(p->seed >> 32);
    F.n_lights = p->n_lights;
    static_assert(sizeof(Light) == sizeof(OrionLight), "light layouts must agree");
    if (p->n_lights) {   // (a blocking copy of a few bytes: both lanes may read it from their first launch on)
        if (int rc = s->s_lights.reserve((size_t)p->n_lights * sizeof(Light))) return rc;
        HIP_TRY(hipMemcpy(s->s_lights.p, p->lights, (size_t)p->n_lights * sizeof(Light), hipMemcpyHostToDevice));
    }
    F.lights = static_cast<const Light*>(s->s_lights.p);
    const bool point = p->n_lights > 0;

    const uint64_t P = (uint64_t)F.tile_w * F.tile_h;
    uint64_t per_batch = s->batch_paths / P;
    if (per_batch > (uint64_t)(p->spp_end - p->spp_begin)) per_batch = p->spp_end - p->spp_begin;
    if (per_batch > (1u << 20)) per_batch = 1u << 20;   // (the jitter table and the camera-ray generator's short divisions)
    if (per_batch < 1) per_batch = 1;
    {
        // The queues of a batch take 144 bytes per path and lane.  Larger batches are faster (fewer thin bounce tails per
        // frame: a 2048-spp Cornell frame +4 % at 256 Mi paths against 64 Mi), but only while they fit: samples per
        // batch are halved until both lanes' queues fit into 3/4 of what the device has free beyond what this scene holds.
        size_t free_b = 0, total_b = 0, held = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            for (const Lane& L : s->lane) {
                for (const DevBuf& b : L.q) held += b.cap;
                held += L.hits.cap + L.rad.cap + L.pre_hits.cap;
            }
            const double budget = s->queue_budget ? (double)s->queue_budget : 0.75 * (double)(free_b + held);
            auto need = [&](uint64_t spb) {   // both lanes' queues once the call has more than one batch
                const double lanes = (s->overlap_below != 0 && (uint64_t)(p->spp_end - p->spp_begin) > spb) ? 2.0 : 1.0;
                return (double)(P * spb) * 144.0 * 1.125 * lanes;
            };
            while (per_batch > 1 && need(per_batch) > budget) per_batch = (per_batch + 1) / 2;
        }
    }
    if (P * per_batch + kParts * kTileWide >= 0xFFFFFFFFull) return fail(1, "render: tile has too many pixels for one batch");
    // The queues are taken now, before anything is launched, and the estimate above is only as good as the moment it was
    // made (another process, another scene, the caller's own tensors may have taken memory since): an allocation that
    // fails halves the samples per batch and tries again with what is left, down to one sample per batch.
    const bool wide_queues = s->mode() == MODE_WIDE && s->draw_rays;
    auto reserve_lane = [&](Lane& L, uint64_t paths) -> int {
        if (s->fail_queue_allocs) { s->fail_queue_allocs--; return fail(2, "hipMalloc: injected failure (option fail_queue_allocs)"); }
        const size_t queue_bytes = (size_t)(paths + kParts * kTileWide) * 16;   // (partitions end on tile boundaries)
        for (DevBuf& b : L.q) if (int rc = b.reserve(queue_bytes)) return rc;
        if (int rc = L.hits.reserve(queue_bytes)) return rc;
        if (int rc = L.rad.reserve(paths * 16)) return rc;
        if (int rc = L.counter.reserve(kRing * kSlotWords * 8)) return rc;
        if (wide_queues) {
            if (int rc = L.pre_hits.reserve(paths * 16)) return rc;
            if (int rc = L.pre_work.reserve(kWorkWords * 4)) return rc;
        }
        return 0;
    };
    for (;;) {
        const bool two = (uint64_t)(p->spp_end - p->spp_begin) > per_batch && s->overlap_below != 0;
        int rc = reserve_lane(s->lane[0], P * per_batch);
        if (rc == 0 && two) rc = reserve_lane(s->lane[1], P * per_batch);
        if (rc == 0) break;
        if (per_batch == 1) { s->fail_queue_allocs = 0; return rc; }
        (void)hipGetLastError();
        for (Lane& L : s->lane) { for (DevBuf& b : L.q) b.release(); L.hits.release(); L.rad.release(); L.pre_hits.release(); }
        per_batch = (per_batch + 1) / 2;
    }
    const uint64_t max_paths = P * per_batch;

    const size_t img_bytes = (size_t)p->width * p->height * 16;
    const bool accum_dev = is_device_ptr(accum_any);
    float4* d_accum = reinterpret_cast<float4*>(accum_any);
    if (!accum_dev) {
        if (int rc = s->s_accum.reserve(img_bytes)) return rc;
        d_accum = static_cast<float4*>(s->s_accum.p);
        HIP_TRY(hipMemcpyAsync(d_accum, accum_any, img_bytes, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->ev_upload, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->lane[1].stream, s->ev_upload, 0));
    }
    const int mode = s->mode();
    const bool per_lane = s->per_lane();
    // Bounce rays of a large scene are handed to lanes as they finish (k_extend_wide) instead of 4 per thread.
    const bool draw_rays = mode == MODE_WIDE && s->draw_rays;
    const bool tail_ok = s->view.n_items != 0 && !per_lane && s->tail_paths != 0;
    const uint64_t rays_per_vertex = point ? p->n_lights : (uint64_t)p->light_samples * s->view.n_emissive;
    const uint32_t n_bins = !per_lane ? (s->n_tris < kMaxBins ? (s->n_tris ? s->n_tris : 1u) : kMaxBins) : 1u;
    const uint64_t n_batches = (p->spp_end - p->spp_begin + per_batch - 1) / per_batch;
    const bool two_lanes = n_batches > 1 && s->overlap_below != 0;

    // One batch of paths in flight: its lane and the host's view of its bounce loop.
    // qa: rays waiting for their closest hit; qb + hits: the vertices that hit, ready to shade.
    // Counter slot of bounce d (two 64-bit words at d % kRing): [0] vertices k_extend kept,
    // [1] rays k_shade emitted for bounce d + 1.  The kernels read their counts on the device;
    // the host only needs an upper bound to size the grids and the zero that ends the loop, and
    // takes both from the bounce kLag launches back.
    struct Batch {
        Lane* L = nullptr;
        FrameParams F;
        Queues qa, qb;
        HitRec* hits = nullptr;
        float4* rad = nullptr;
        unsigned long long* count = nullptr;
        uint32_t bound = 0;           // >= the number of rays entering the next bounce to launch
        uint32_t pbound = 0;          // >= the rays of every queue partition entering that bounce (FrameParams::part_log2)
        uint32_t consumed = 0;        // bounces whose counters the host has read
        uint32_t launched = 0;
        uint64_t rays_in = 0;         // rays that entered bounce `consumed`
        uint32_t tags[kRing] = {};
    };
    auto begin = [&](Batch& B, uint64_t index) -> int {
        Lane& L = s->lane[two_lanes ? (index & 1) : 0];
        if (int rc = reserve_lane(L, max_paths)) return rc;      // (taken above: nothing to do)
        if (draw_rays) HIP_TRY(hipMemsetAsync(L.pre_work.p, 0, kWorkWords * 4, L.stream));
        B = Batch{};
        B.L = &L;
        B.F = F;
        const uint64_t sb = p->spp_begin + index * per_batch;
        const uint32_t ns = (uint32_t)((p->spp_end - sb < per_batch) ? (p->spp_end - sb) : per_batch);
        B.F.spp_begin = (uint32_t)sb; B.F.n_samples = ns;
        B.qa = Queues{static_cast<float4*>(L.q[0].p), static_cast<float4*>(L.q[1].p), static_cast<float4*>(L.q[2].p)};
        B.qb = Queues{static_cast<float4*>(L.q[3].p), static_cast<float4*>(L.q[4].p), static_cast<float4*>(L.q[5].p)};
        B.hits = static_cast<HitRec*>(L.hits.p);
        B.rad = static_cast<float4*>(L.rad.p);
        B.count = static_cast<unsigned long long*>(L.counter.p);
        B.bound = (uint32_t)(P * ns);
        B.rays_in = B.bound;
        {   // Queue partitions (device_types.h).  The first bounce's block b leaves its hits in partition b mod parts, so
            // a partition holds at most ceil(blocks / parts) tiles.  The ray-drawing kernels of large scenes index their
            // queue as one array and keep one partition.
            const uint32_t tile0 = B.bound >= kNarrowBelow ? kTileWide : kBlock;   // (bounce() picks the tile the same way)
            const uint32_t blocks0 = (B.bound + tile0 - 1) / tile0;
            // ... and so do frames without bounces (max_depth 0), whose only appends are the first launch's, one per tile of
            // 1024 paths: partitions buy them nothing (a 4K frame of a 200-triangle scene ran 2.8 % slower with sixteen).
            B.F.part_log2 = (draw_rays || p->max_depth == 0) ? 0u : s->queue_parts_log2;
            B.F.part_stride = ((blocks0 + (1u << B.F.part_log2) - 1) >> B.F.part_log2) * tile0;
            B.pbound = B.F.part_stride;
        }
        HIP_TRY(hipMemsetAsync(B.count, 0, kRing * kSlotWords * 8, L.stream));
        if (int rc = L.hit_mask.reserve((max_paths + 63) / 64 * 8)) return rc;
        if (int rc = L.jitter.reserve((size_t)ns * sizeof(float2))) return rc;
        hipLaunchKernelGGL(k_jitter, dim3(grid_for(ns)), dim3(kBlock), 0, L.stream, B.F, static_cast<float2*>(L.jitter.p));
        HIP_TRY(hipGetLastError());
        return 0;
    };
    // Launches bounce `depth` of a batch: closest hits, then shading.
    auto bounce = [&](Batch& B, uint32_t depth) -> int {
        if (s->fail_at_bounce && --s->fail_at_bounce == 0) return fail(3, "render: injected failure (option fail_at_bounce)");
        const uint32_t slot = depth % kRing, prev_slot = (depth + kRing - 1) % kRing;
        unsigned long long* cnt = B.count + kSlotWords * slot;
        const unsigned long long* prev = depth ? B.count + kSlotWords * prev_slot : nullptr;
        unsigned long long* clear = B.count + kSlotWords * ((depth + kClearAhead) % kRing);
        const uint32_t parts = 1u << B.F.part_log2;
        B.tags[slot] = ++s->tag;     // bounce `depth` is published under this tag by the next extend launch
        const bool wide = B.bound >= kNarrowBelow;
        const uint32_t tile = wide ? kTileWide : kBlock;
        auto extend = [&](auto kernel) {
            // (the first bounce generates its rays from the batch's path ids; later ones read their partition's rays)
            const uint32_t grid = depth == 0 ? (B.bound + tile - 1) / tile : parts * ((B.pbound + tile - 1) / tile);
            launch_on(s, B.L->stream, ORION_KERNEL_CLOSEST_HIT, 0, kernel, dim3(grid), dim3(kBlock), s->view, B.F, B.rad, B.qa,
                      prev, depth == 0 ? B.bound : B.pbound, B.qb, B.hits, cnt, clear, B.L->h_count + 4 * prev_slot, B.tags[prev_slot], n_bins,
                      static_cast<const float4*>(B.L->pre_hits.p), static_cast<uint32_t*>(B.L->pre_work.p), static_cast<const float2*>(B.L->jitter.p),
                      static_cast<unsigned long long*>(B.L->hit_mask.p));
        };
        if (draw_rays && depth > 0) {
            uint32_t grab, grid;
            draw_plan(s, B.bound, &grab, &grid);
            launch_on(s, B.L->stream, ORION_KERNEL_CLOSEST_HIT, 0, k_extend_wide, dim3(grid), dim3(kBlock), s->view, B.qa, prev, B.bound,
                      static_cast<float4*>(B.L->pre_hits.p), grab, static_cast<uint32_t*>(B.L->pre_work.p));
            if (wide) extend(k_extend<MODE_PRE, kRaysWide, false>); else extend(k_extend<MODE_PRE, 1, false>);
        } else {
            // The samples of one pixel keep a wave together whatever the size of the tree, and the
            // wave-synchronous walk costs the union of its lanes' paths: the primary rays take it in
            // scenes of any size (and so do the shadow rays a point-light frame casts from their hit
            // points, k_shade_wide_point0; the rays of later bounces do not).
            const bool coherent = depth == 0 && per_lane && s->traversal == 0 && s->wave_ok;
            if (coherent && s->wave_ahead) {
                // ... in its form that fetches the next record while it tests the present one: in a tree of thousands of
                // nodes the walk waits for its scalar loads (they come from the L2), not for the vector ALU.
                if (wide) extend(k_extend<MODE_WAVE_AHEAD, kRaysWide, true>); else extend(k_extend<MODE_WAVE_AHEAD, 1, true>);
            } else if (coherent) {   // ... with the scene's box ahead of the roots (most camera rays of such a frame miss the model)
                if (wide) extend(k_extend<MODE_WAVE_OUTSIDE, kRaysWide, true>); else extend(k_extend<MODE_WAVE_OUTSIDE, 1, true>);
            } else with_mode(mode, [&](auto M) {
                constexpr int MODE = decltype(M)::value;
                if (depth == 0) {   // primary rays are generated inside the launch
                    if (wide) extend(k_extend<MODE, kRaysWide, true>); else extend(k_extend<MODE, 1, true>);
                } else {
                    if (wide) extend(k_extend<MODE, kRaysWide, false>); else extend(k_extend<MODE, 1, false>);
                }
            });
        }
        with_mode(mode, [&](auto M) {
            constexpr int MODE = decltype(M)::value;
            // (a batch's first bounce leaves path ids instead of whole records: the kernels' FIRST forms)
            auto shade = depth == 0 ? (point ? k_shade<MODE, true, true> : (MODE == MODE_WAVE ? (s->solid ? k_shade_wave_brdf<true, true> : k_shade_wave_brdf<true, false>) : k_shade<MODE, false, true>))
                                    : (point ? k_shade<MODE, true, false> : (MODE == MODE_WAVE ? (s->solid ? k_shade_wave_brdf<false, true> : k_shade_wave_brdf<false, false>) : k_shade<MODE, false, false>));
            // (the first vertices of a large scene left to choose: their shadow rays walk together, see the kernel)
            if (MODE == MODE_WIDE && point && depth == 0 && s->traversal == 0 && s->wave_ok) shade = k_shade_wide_point0;
            launch_on(s, B.L->stream, ORION_KERNEL_SHADE, 0, shade, dim3(parts * grid_for(B.pbound)), dim3(kBlock), s->view, B.F, depth, B.qb, cnt, B.hits, B.qa, cnt + 1, B.rad,
                      static_cast<const float2*>(B.L->jitter.p));
        });
        HIP_TRY(hipGetLastError());
        B.launched = depth + 1;
        return 0;
    };
    auto consume = [&](Batch& B) -> int { // read the counters of bounce `consumed`
        const uint32_t slot = B.consumed % kRing;
        uint32_t h[3];
        if (int rc = await_counters(B.L->stream, B.L->h_count + 4 * slot, B.tags[slot], h)) return rc;
        if (s->profiling) {
            s->stats[ORION_KERNEL_CLOSEST_HIT].items += B.rays_in;
            s->stats[ORION_KERNEL_SHADE].items += h[0];
            s->stats[ORION_KERNEL_SHADOW].launches++;   // light-sample / shadow rays traced inside the shade launch
            s->stats[ORION_KERNEL_SHADOW].items += (uint64_t)h[0] * rays_per_vertex;
        }
        B.rays_in = h[1];
        B.pbound = h[2];
        B.consumed++;
        return 0;
    };

    // The batches.  A failure anywhere in here (an allocation for the next batch, a launch) leaves launches of one
    // or both lanes in flight: the caller below waits for them before it reports, so that the handle can be used again.
    auto run = [&]() -> int {
        Batch cur, next;
        bool next_begun = false;
        Lane* last_resolve = nullptr;
        for (uint64_t index = 0; index < n_batches; index++) {
            if (next_begun) { cur = next; next_begun = false; }
            else if (int rc = begin(cur, index)) return rc;
            // Once this batch is down to its thin tail (launches that leave most of the chip idle), the
            // next batch's first kLag bounces -- which need no counts from the host -- go to the other lane.
            auto start_next = [&]() -> int {
                if (next_begun || !two_lanes || index + 1 >= n_batches) return 0;
                if (int rc = begin(next, index + 1)) return rc;
                for (uint32_t d = 0; d < kLag && d <= p->max_depth; d++) if (int rc = bounce(next, d)) return rc;
                next_begun = true;
                return 0;
            };
            for (uint32_t depth = cur.launched; depth <= p->max_depth; depth++) {
                if (int rc = bounce(cur, depth)) return rc;
                if (depth >= kLag) {
                    if (int rc = consume(cur)) return rc;
                    if (cur.rays_in == 0) break;          // bounce `consumed` has no rays; the ones launched past it find none either
                    cur.bound = (uint32_t)cur.rays_in;
                    if (cur.bound < s->overlap_below) if (int rc = start_next()) return rc;
                    // Few paths left: one launch takes each of them to its end (one wave per path).  The
                    // rays it starts from are the ones the last shade launch emitted, at most `bound`.
                    if (tail_ok && cur.bound <= s->tail_paths && cur.launched <= p->max_depth) {
                        launch_on(s, cur.L->stream, ORION_KERNEL_TAIL, cur.bound, point ? k_tail<true> : k_tail<false>, dim3((cur.bound + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock),
                                  s->view, cur.F, cur.launched, cur.qa, cur.count + kSlotWords * ((cur.launched - 1) % kRing), cur.rad);
                        HIP_TRY(hipGetLastError());
                        break;
                    }
                }
            }
            if (s->profiling) {   // the statistics want every bounce; the last one has no extend launch after it
                const uint32_t last = (cur.launched - 1) % kRing;
                hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, cur.L->stream, cur.count + kSlotWords * last, cur.L->h_count + 4 * last, cur.tags[last], cur.F.part_log2);
                while (cur.consumed < cur.launched) if (int rc = consume(cur)) return rc;
            }
            // The accumulator takes the batches' sums in ascending sample order, whichever lane they ran on.
            if (last_resolve && last_resolve != cur.L) HIP_TRY(hipStreamWaitEvent(cur.L->stream, last_resolve->resolved, 0));
            launch_on(s, cur.L->stream, ORION_KERNEL_RESOLVE, P, k_resolve, dim3(grid_for(P)), dim3(kBlock), cur.F, cur.rad, d_accum, static_cast<const uint32_t*>(cur.L->hit_mask.p));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(cur.L->resolved, cur.L->stream));
            last_resolve = cur.L;
        }
        if (!accum_dev) HIP_TRY(hipMemcpyAsync(accum_any, d_accum, img_bytes, hipMemcpyDeviceToHost, last_resolve->stream));
        for (Lane& L : s->lane) HIP_TRY(hipStreamSynchronize(L.stream));
        return 0;
    };
    const int rc = run();
    s->fail_at_bounce = 0; s->fail_queue_allocs = 0;   // (the test hooks never outlive the call they were armed for)
    if (rc != 0) {
        const std::string why = g_error;
        for (Lane& L : s->lane) (void)hipStreamSynchronize(L.stream);
        (void)hipGetLastError();
        if (s->profiling) drain_profile(s);
        g_error = why;
    }
    return rc;
}

int orion_hip_camera_vectors(const float eye[3], const float look_at[3], const float up[3],
                             float y_view, float aspect_ratio, OrionCamera* out) {
    if (!eye || !look_at || !up || !out) return fail(1, "camera_vectors: null argument");
    H3 vUp = {up[0], up[1], up[2]};
    H3 vFront = {look_at[0] - eye[0], look_at[1] - eye[1], look_at[2] - eye[2]};
    // Gram-Schmidt: up -= (front.up / front.front) * front  (orion/math.hpp:313-317)
    const float k = hdot(vFront, vUp) / hdot(vFront, vFront);
    const H3 proj = hscale(vFront, k);
    vUp = {vUp.x - proj.x, vUp.y - proj.y, vUp.z - proj.z};
    vUp = hnormalize(vUp);
    vFront = hnormalize(vFront);
    H3 vRight = hcross(vFront, vUp);
    vUp = hscale(hscale(vUp, y_view), 0.5f);
    vRight = hscale(hscale(hscale(vRight, y_view), aspect_ratio), 0.5f);
    std::memcpy(out->eye, eye, 12);
    out->front[0] = vFront.x; out->front[1] = vFront.y; out->front[2] = vFront.z;
    out->right[0] = vRight.x; out->right[1] = vRight.y; out->right[2] = vRight.z;
    out->up[0] = vUp.x; out->up[1] = vUp.y; out->up[2] = vUp.z;
    return 0;
}

#if ORION_WIDE_STATS
// Builds with -DORION_WIDE_STATS=1 only: the lane-occupancy counters of the ray-drawing loop (pathtrace.h).
int orion_hip_debug_wide_stats(unsigned long long* out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wide_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stats), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
int orion_hip_debug_wide_times(unsigned long long* out8192x4) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out8192x4, HIP_SYMBOL(g_wide_times), 8192 * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

int orion_hip_profile_enable(OrionScene* s, int enable) {
    if (!s) return fail(1, "profile_enable: scene is null");
    DeviceGuard guard(s->device);
    drain_profile(s);
    std::memset(s->stats, 0, sizeof s->stats);
    s->profiling = enable != 0;
    return 0;
}

int orion_hip_profile_read(OrionScene* s, OrionKernelStats* stats) {
    if (!s || !stats) return fail(1, "profile_read: null argument");
    DeviceGuard guard(s->device);
    drain_profile(s);
    std::memcpy(stats, s->stats, sizeof s->stats);
    return 0;
}

} // extern "C"
