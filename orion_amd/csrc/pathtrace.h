// Device side of the path tracer: the ray-query kernels and the kernels of one render batch
// (a wavefront path tracer over ray queues in HBM):
//   per bounce:
//     k_extend    closest hit of every queued ray; the first bounce also generates
//                 the primary rays                                    orion/raytracer.cpp:72-80, orion/model.hpp:52-62
//     k_shade     surface + direct light (light-sample rays traced
//                 in place) + roulette + next ray, survivors
//                 compacted into the next queue with a wave ballot    orion/raytracer.cpp:116-206
//   k_tail        the last few thousand paths, one path per wave, to their end
//   k_resolve     per-pixel sum of the batch's samples, in sample
//                 order, into the caller's accumulator                orion/raytracer.cpp:84-86
// Included by kernels.hip only (everything lives in its anonymous namespace).
#pragma once

#include "shading.h"
#include "wide_walk.h"

using namespace orion_dev;

// ===========================================================================
// Kernels
// ===========================================================================
namespace {

constexpr int kBlock = 256;
static_assert(kBlock == kWideBlock, "the wide walk's stack is laid out for this block size");

// MODE_LANE: every lane walks on its own.  MODE_WAVE: the wave walks the node array once for its 64
// rays.  MODE_EACH: the wave takes its rays one at a time, all lanes on one ray (a way to run the
// one-ray-per-wave walk on ordinary batches; its use is testing).  MODE_SAME: the caller guarantees
// that all lanes hold the same ray and the same `active` (the tail kernel).
// MODE_ADAPT (ray queries on large scenes): each wave looks at its rays and takes the wave-synchronous
// walk if they leave one point in nearly one direction (camera rays, rays towards a point light:
// that walk costs the union of the lanes' paths, which such rays keep small), else the per-lane walk.
// MODE_WIDE: every lane walks the 4-wide tree, nearest first, with distance culling (wide_walk.h);
// MODE_ADAPT_WIDE: MODE_ADAPT with that walk for the waves that are not coherent.
// MODE_PRE (k_extend only): the closest hits were found by k_extend_wide before the launch; k_extend just takes them.
#ifndef ORION_EXPERIMENT_NO_RAD
#define ORION_EXPERIMENT_NO_RAD 0
#endif
// MODE_WAVE_AHEAD: MODE_WAVE through the walk that fetches one record ahead (traverse_wave_ahead; trees of thousands of nodes).
// MODE_WAVE_OUTSIDE: MODE_WAVE for rays that may come from outside a large scene: the scene's box is tested first (traverse_wave_outside).
enum { MODE_LANE = 0, MODE_WAVE = 1, MODE_EACH = 2, MODE_SAME = 3, MODE_ADAPT = 4, MODE_WIDE = 5, MODE_ADAPT_WIDE = 6, MODE_PRE = 7, MODE_WAVE_AHEAD = 8, MODE_WAVE_OUTSIDE = 9 };

// True for every lane of a wave whose active rays share their origin exactly and lie within about
// 25 degrees of the first active ray's direction.
__device__ __forceinline__ bool wave_is_coherent(V3 o, V3 d, bool active) {
    const uint64_t act = __ballot(active);
    if (act == 0ull) return false;
    const int first = __builtin_ctzll(act);
    const V3 o0 = mk(__shfl(o.x, first), __shfl(o.y, first), __shfl(o.z, first));
    const V3 d0 = mk(__shfl(d.x, first), __shfl(d.y, first), __shfl(d.z, first));
    const float c = dot_raw(d, d0), n = dot_raw(d, d) * dot_raw(d0, d0);
    const bool near = o.x == o0.x && o.y == o0.y && o.z == o0.z && c > 0.0f && c * c >= 0.82f * n;
    return __all(!active || near);
}

template <int MODE, bool ANY>
__device__ __forceinline__ void traverse(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) {
    if (MODE == MODE_WAVE) {
        traverse_wave(S, o, d, h, active);   // (answers any-hit queries too, see there)
    } else if (MODE == MODE_WAVE_AHEAD) {
        traverse_wave_ahead(S, o, d, h, active);
    } else if (MODE == MODE_WAVE_OUTSIDE) {
        traverse_wave_outside(S, o, d, h, active);
    } else if (MODE == MODE_ADAPT) {
        if (wave_is_coherent(o, d, active)) traverse_wave_outside(S, o, d, h, active);
        else if (active) traverse_lane<ANY>(S, o, d, h);
    } else if (MODE == MODE_ADAPT_WIDE) {
        if (wave_is_coherent(o, d, active)) traverse_wave_outside(S, o, d, h, active);
        else if (active) traverse_wide<ANY>(S, o, d, h);
    } else if (MODE == MODE_WIDE) {
        if (active) traverse_wide<ANY>(S, o, d, h);
    } else if (MODE == MODE_EACH) {
        traverse_one_each<ANY>(S, o, d, h, active);
    } else if (MODE == MODE_SAME) {
        if (active) traverse_one<ANY>(S, o, d, h);
    } else {
        if (active) traverse_lane<ANY>(S, o, d, h);
    }
}

__device__ __forceinline__ HitRec miss_rec() { HitRec h; h.t = INFINITY; h.u = 0.0f; h.v = 0.0f; h.gtri = kInvalid; return h; }

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Block-level queue append.  Every thread of the block must call it.  Survivors are
// counted per wave with a ballot, the four wave counts are combined through LDS and
// ONE 64-bit atomic per block reserves the slots (low word: queue length, high word:
// a statistics counter, here the number of shaded vertices).  One atomic per wave on a
// single address would serialise at the memory side (~12 ns each) and bound the kernel.
__device__ __forceinline__ uint32_t queue_slot(bool push, bool stat, unsigned long long* counter) {
    __shared__ uint32_t s_cnt[kBlock / 64];
    __shared__ uint32_t s_stat[kBlock / 64];
    __shared__ uint32_t s_base;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t m = __ballot(push);
    const uint64_t ms = __ballot(stat);
    if (lane_id() == 0) { s_cnt[wave] = (uint32_t)__popcll(m); s_stat[wave] = (uint32_t)__popcll(ms); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0, stat_total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) { total += s_cnt[w]; stat_total += s_stat[w]; }
        unsigned long long old = 0;
        if (total | stat_total) old = atomicAdd(counter, (unsigned long long)total | ((unsigned long long)stat_total << 32));
        s_base = (uint32_t)old;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wave; w++) base += s_cnt[w];
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    return base + rank;
}

// --- ray-query entry points --------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_closest_api(SceneView S, const float* __restrict__ org, const float* __restrict__ dir,
                                                        uint32_t n, OrionHit* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < n;
    V3 o = splat(0.0f), d = splat(1.0f);
    if (active) { o = ld3(org + 3ull * i); d = ld3(dir + 3ull * i); }
    HitRec h = miss_rec();
    traverse<MODE, false>(S, o, d, h, active);
    if (active) {
        OrionHit r;
        r.mesh = kInvalid; r.tri = kInvalid; r.t = h.t; r.u = h.u; r.v = h.v;
        if (h.gtri != kInvalid) {
            r.mesh = S.attrs[h.gtri].mesh;
            r.tri = h.gtri - S.meshes[r.mesh].tri_base;
        }
        out[i] = r;
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_anyhit_api(SceneView S, const float* __restrict__ org, const float* __restrict__ dir,
                                                       uint32_t n, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < n;
    V3 o = splat(0.0f), d = splat(1.0f);
    if (active) { o = ld3(org + 3ull * i); d = ld3(dir + 3ull * i); }
    HitRec h = miss_rec();
    traverse<MODE, true>(S, o, d, h, active);
    if (active) out[i] = h.gtri != kInvalid ? 1 : 0;
}

// The ray queries of large scenes (MODE_WIDE, MODE_ADAPT_WIDE).  Incoherent rays take wildly different
// numbers of steps through the tree (half of them a handful, one in a hundred over a hundred), and a wave
// that holds 64 rays from start to end runs as long as its longest one with most lanes idle.  Here the
// waves of a launch that just fills the chip draw rays until none are left: a wave takes `grab`
// consecutive rays at a time (64 to kQueryGrab, by the size of the batch) from a global counter (one atomic per grab) and hands them to its lanes as
// they finish, kRefillMin lanes at a time.
//   work[1]: number of listed chunks (LISTED only); work[kWorkCounters + 16 p]: the grab counter of partition p
//   (all zero at launch).  The grabs are split into kWorkParts equal ranges with a counter each, 64 bytes
//   apart: a wave starts on the range of its block and moves on to the next when that one is used up.  (One
//   counter for all serialises at the memory side, ~12 ns per grab: batches of short rays were bound by it.)
// LISTED (MODE_ADAPT_WIDE): k_query_coherent has been over the batch first; 64-ray chunks whose rays leave
// one point in nearly one direction (camera rays) went through the wave-synchronous walk there, and the
// others are listed in `chunks` -- a grab is then one listed chunk.
#ifndef ORION_REFILL_MIN
#define ORION_REFILL_MIN 8
#endif
#ifndef ORION_LEAF_QUORUM
#define ORION_LEAF_QUORUM 24
#endif
#ifndef ORION_QUERY_GRAB
#define ORION_QUERY_GRAB 256
#endif
constexpr uint32_t kRefillMin = ORION_REFILL_MIN;       // lanes out of work before a wave fetches
constexpr uint32_t kLeafQuorumWide = ORION_LEAF_QUORUM; // lanes waiting at a leaf before the wave turns to the triangles
constexpr uint32_t kQueryGrab = ORION_QUERY_GRAB;       // most rays a wave takes from the global counter at a time
constexpr uint32_t kWorkParts = 16, kWorkCounters = 64, kWorkWords = kWorkCounters + 16 * kWorkParts;   // layout of the work words
#ifndef ORION_COHERENT_CHUNKS
#define ORION_COHERENT_CHUNKS 16
#endif
constexpr uint32_t kCoherentChunks = ORION_COHERENT_CHUNKS;                // 64-ray chunks per block of k_query_coherent

#ifndef ORION_DRAW_WAVES
#define ORION_DRAW_WAVES 6      // (with the guard's words the compiler's own choice is 87 registers and 5 waves: 6 waves with 2 spilled registers run 2 to 8 % faster)
#endif
#if ORION_DRAW_WAVES
#define ORION_DRAW_ATTR __attribute__((amdgpu_waves_per_eu(ORION_DRAW_WAVES, ORION_DRAW_WAVES)))
#else
#define ORION_DRAW_ATTR
#endif

template <bool ANY>
__device__ __forceinline__ void query_store(const SceneView& S, uint32_t i, const HitRec& h, OrionHit* __restrict__ out, uint8_t* __restrict__ occ) {
    if (ANY) { occ[i] = h.gtri != kInvalid ? 1 : 0; return; }
    OrionHit r;
    r.mesh = kInvalid; r.tri = kInvalid; r.t = h.t; r.u = h.u; r.v = h.v;
    if (h.gtri != kInvalid) {
        r.mesh = S.attrs[h.gtri].mesh;
        r.tri = h.gtri - S.meshes[r.mesh].tri_base;
    }
    out[i] = r;
}

template <bool ANY>
__global__ __launch_bounds__(kBlock) void k_query_coherent(SceneView S, const float* __restrict__ org, const float* __restrict__ dir, uint32_t n,
                                                           OrionHit* __restrict__ out, uint8_t* __restrict__ occ,
                                                           uint32_t* __restrict__ work, uint32_t* __restrict__ chunks) {
    __shared__ uint32_t s_left[kCoherentChunks];
    __shared__ uint32_t s_nleft, s_base;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_nleft = 0u;
    __syncthreads();
    const uint32_t n_chunks = (n + 63u) / 64u;
    for (uint32_t k = wave; k < kCoherentChunks; k += kBlock / 64) {
        const uint32_t chunk = blockIdx.x * kCoherentChunks + k;
        if (chunk >= n_chunks) break;
        const uint32_t i = chunk * 64u + lane;
        const bool active = i < n;
        V3 o = splat(0.0f), d = splat(1.0f);
        if (active) { o = ld3(org + 3ull * i); d = ld3(dir + 3ull * i); }
        if (wave_is_coherent(o, d, active)) {
            HitRec h = miss_rec();
            traverse_wave_outside(S, o, d, h, active);
            if (active) query_store<ANY>(S, i, h, out, occ);
        } else if (lane == 0) {
            s_left[atomicAdd(&s_nleft, 1u)] = chunk;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_nleft) s_base = atomicAdd(&work[1], s_nleft);
    __syncthreads();
    if (threadIdx.x < s_nleft) chunks[s_base + threadIdx.x] = s_left[threadIdx.x];
}

// --- sharing a ray between lanes ------------------------------------------------
// Once a wave's pool of rays is used up its lanes finish one by one, and the wave lives as long as its
// longest ray: hundreds of node steps taken one after the other by one lane while the rest sit idle.  From
// then on a free lane takes the top entry of a busy lane's stack together with a copy of the ray and of its
// best hit so far, walks that subtree on a stack of its own (it may hand parts on in turn), and gives its best
// hit back when it is done.  The owner keeps the better of the two by the walk's result rule ("smaller t, or
// equal t and the later triangle slot"), which does not depend on the order of visits, so the ray's result is
// what one lane alone would have found; a taker only lacks the owner's later improvements of `best`, which
// costs it visits the owner would have culled, never a result.
#ifndef ORION_WIDE_SHARE
#define ORION_WIDE_SHARE 1
#endif
constexpr uint32_t kNoOwner = 64u;
#ifndef ORION_SHARE_MIN
#define ORION_SHARE_MIN 16
#endif
constexpr uint32_t kShareMin = ORION_SHARE_MIN;      // free lanes it takes to make a round of handing out worthwhile
static_assert(kWideStack >= 13, "a handed-over part travels through the taker's (empty) stack column: 13 words");
__shared__ uint8_t g_wide_taker[kWideBlock];

// Free lanes take parts from lanes that have something stacked.  Returns whether free lanes remain or were
// served (the caller then gives the busy lanes a chance to stack something after every node step).
__device__ __forceinline__ bool wide_share(WideState& w, V3& o, V3& d, V3& inv, HitRec& h, bool& quick_det, bool slow, bool idle,
                                           uint32_t& owner, uint32_t& lent) {
    const uint32_t lane = lane_id(), tid = threadIdx.x, wave0 = tid & ~63u;
    const bool give = w.cur != kWideEmpty && !slow && w.sp != 0u;
    const uint64_t mt = __ballot(idle), mg = __ballot(give);
    if ((uint32_t)__popcll(mt) < kShareMin) return false;
    if (mg == 0ull) return true;
    const uint32_t n = min((uint32_t)__popcll(mt), (uint32_t)__popcll(mg));
    const uint32_t rank_t = __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
    const uint32_t rank_g = __builtin_amdgcn_mbcnt_hi((uint32_t)(mg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mg, 0u));
    const bool takes = idle && rank_t < n, gives = give && rank_g < n;
    if (takes) g_wide_taker[wave0 + rank_t] = (uint8_t)lane;
    wave_lds_fence();
    // What a lane hands out is the BOTTOM entry of its stack: the farthest and, as a rule, the largest of the subtrees
    // it has yet to visit (the top entry is what it would visit next itself, usually a few nodes; handing those
    // out leaves the long rays as long as they were).  The entries above move down one row.
    uint32_t ref = 0u, tb = 0u;
    if (gives) { ref = g_wide_ref[0][tid]; tb = g_wide_t[0][tid]; }
    for (uint32_t k = 1; __any(gives && k < w.sp); k++)
        if (gives && k < w.sp) { g_wide_ref[k - 1][tid] = g_wide_ref[k][tid]; g_wide_t[k - 1][tid] = g_wide_t[k][tid]; }
    if (gives) {
        const uint32_t col = wave0 + g_wide_taker[wave0 + rank_g];   // the taker's stack column: empty, it carries the part
        w.sp--;
        g_wide_ref[0][col] = __float_as_uint(o.x); g_wide_ref[1][col] = __float_as_uint(o.y); g_wide_ref[2][col] = __float_as_uint(o.z);
        g_wide_ref[3][col] = __float_as_uint(d.x); g_wide_ref[4][col] = __float_as_uint(d.y); g_wide_ref[5][col] = __float_as_uint(d.z);
        g_wide_ref[6][col] = __float_as_uint(h.t); g_wide_ref[7][col] = __float_as_uint(h.u); g_wide_ref[8][col] = __float_as_uint(h.v);
        g_wide_ref[9][col] = h.gtri; g_wide_ref[10][col] = w.best_slot; g_wide_ref[11][col] = ref;
        g_wide_ref[12][col] = tb | (quick_det ? 1u << 16 : 0u) | lane << 17;
        lent++;
    }
    wave_lds_fence();
    if (takes) {
        o = mk(__uint_as_float(g_wide_ref[0][tid]), __uint_as_float(g_wide_ref[1][tid]), __uint_as_float(g_wide_ref[2][tid]));
        d = mk(__uint_as_float(g_wide_ref[3][tid]), __uint_as_float(g_wide_ref[4][tid]), __uint_as_float(g_wide_ref[5][tid]));
        h.t = __uint_as_float(g_wide_ref[6][tid]); h.u = __uint_as_float(g_wide_ref[7][tid]); h.v = __uint_as_float(g_wide_ref[8][tid]);
        h.gtri = g_wide_ref[9][tid]; w.best_slot = g_wide_ref[10][tid];
        const uint32_t ref = g_wide_ref[11][tid], x = g_wide_ref[12][tid];
        inv = ray_inverse_quick(d);   // (a ray the wide walk has taken: see below)
        quick_det = (x >> 16 & 1u) != 0u;
        owner = x >> 17;
        w.sp = 0u;
        w.cur = __uint_as_float((x & 0xFFFFu) << 16) > h.t ? kWideEmpty : ref;   // (culled by now: given back at once)
    }
    wave_lds_fence();   // the columns are stacks again
    return true;
}

// Lanes that are done with a part give their best hit to the part's owner.
__device__ __forceinline__ void wide_deliver(WideState& w, HitRec& h, bool slow, uint32_t& owner, uint32_t& lent) {
    const uint32_t lane = lane_id();
    for (;;) {
        uint64_t m = __ballot(owner != kNoOwner && w.cur == kWideEmpty && !slow && lent == 0u);
        if (m == 0ull) break;
        while (m) {
            const uint32_t from = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t to = __builtin_amdgcn_readlane(owner, from);
            const float t = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h.t), from));
            const float u = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h.u), from));
            const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h.v), from));
            const uint32_t gtri = __builtin_amdgcn_readlane(h.gtri, from), slot = __builtin_amdgcn_readlane(w.best_slot, from);
            if (lane == to) {
                if (gtri != kInvalid && (t < h.t || (t == h.t && slot > w.best_slot))) { h.t = t; h.u = u; h.v = v; h.gtri = gtri; w.best_slot = slot; }
                lent--;
            }
            if (lane == from) owner = kNoOwner;
        }
    }
}

// Lane-occupancy counters of the loop below, for builds with -DORION_WIDE_STATS=1 only (tools/ab/wide_stats.py):
// [0] outer turns  [1] node steps  [2] lanes in them  [3] leaf steps  [4] lanes in them  [5] triangle turns (longest leaf of
// each step)  [6] lane-triangles  [7] refills  [8] rays handed out  [9] rounds of sharing  [10] unculled walks  [11] waves
#ifndef ORION_WIDE_STATS
#define ORION_WIDE_STATS 0
#endif
#if ORION_WIDE_STATS
__device__ unsigned long long g_wide_stats[16];
__device__ unsigned long long g_wide_times[8192][4];   // per wave: start, pool empty, end (s_memrealtime, 100 MHz), rays walked
#if ORION_WIDE_STATS == 1
#define WIDE_STAT(k, v) (stat[k] += (v))
#else
#define WIDE_STAT(k, v) ((void)0)     // ORION_WIDE_STATS=2: the per-wave times only; 3: start and end only
#endif
#else
#define WIDE_STAT(k, v) ((void)0)
#endif

// The loop of a wave that draws rays: load(i, o, d) fetches ray i, store(i, h) takes its result.
template <bool ANY, bool LISTED, typename Load, typename Store>
__device__ __forceinline__ void wide_draw_rays(const SceneView& S, uint32_t n, uint32_t grab, uint32_t* __restrict__ work, const uint32_t* __restrict__ chunks,
                                               Load load, Store store) {
    const uint32_t lane = lane_id(), tid = threadIdx.x;
    const uint32_t n_listed = LISTED ? work[1] : 0u;
    const uint32_t n_grabs = LISTED ? (n_listed + grab / 64u - 1u) / (grab / 64u) : (n + grab - 1u) / grab;
    const uint32_t part_grabs = (n_grabs + kWorkParts - 1u) / kWorkParts;   // grabs per partition
    uint32_t part = blockIdx.x % kWorkParts, parts_done = 0u;                // (wave-uniform)
    uint32_t g_next = 0u, g_end = 0u;    // the rays of the wave's current grab not handed out yet (wave-uniform)
    uint32_t l_next = 0u, l_end = 0u;    // LISTED: the listed chunks of the grab not started yet
    bool pool_empty = false;             // (wave-uniform)

    uint32_t ray = kInvalid;
    V3 o = splat(0.0f), d = splat(1.0f), inv = splat(1.0f);
    HitRec h = miss_rec();
    WideState w; w.cur = kWideEmpty; w.sp = 0u; w.best_slot = 0u;
    bool quick_det = false, slow = false;
    WideGuard guard_words; guard_words.cell = kGuardNoCell;
    // Sharing (see wide_share): the lane whose ray this lane walks a part of, and how many lanes walk parts of this lane's.
    uint32_t owner = kNoOwner, lent = 0u;
#if ORION_WIDE_STATS
    uint32_t stat[12] = {};
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_empty = 0ull;
#endif
    for (;;) {
        WIDE_STAT(0, 1u);
#if ORION_WIDE_SHARE
        if (pool_empty) wide_deliver(w, h, slow, owner, lent);
#endif
        if (ray != kInvalid && w.cur == kWideEmpty && !slow && lent == 0u) { store(ray, h); ray = kInvalid; }
        const bool idle = w.cur == kWideEmpty && !slow && ray == kInvalid && owner == kNoOwner;
        const uint64_t mi = __ballot(idle);
        const uint64_t busy = ~mi & __ballot(true);
        if (!pool_empty && mi != 0ull && ((uint32_t)__popcll(mi) >= kRefillMin || busy == 0ull)) {
            if (g_next == g_end && l_next == l_end) {   // a new grab
                uint32_t g = n_grabs;
                while (parts_done < kWorkParts) {
                    uint32_t k = 0u;
                    if (lane == 0u) k = atomicAdd(&work[kWorkCounters + 16u * part], 1u);
                    k = __builtin_amdgcn_readfirstlane(k);
                    if (k < part_grabs && part * part_grabs + k < n_grabs) { g = part * part_grabs + k; break; }
                    part = (part + 1u) % kWorkParts; parts_done++;   // this range is used up: on to the next
                }
                if (g >= n_grabs) {
                    pool_empty = true;
#if ORION_WIDE_STATS && ORION_WIDE_STATS != 3
                    t_empty = __builtin_amdgcn_s_memrealtime();
#endif
                }
                else if (LISTED) { l_next = g * (grab / 64u); l_end = min(n_listed, l_next + grab / 64u); }
                else { g_next = g * grab; g_end = min(n, g_next + grab); }
            }
            if (LISTED && g_next == g_end && l_next != l_end) { g_next = chunks[l_next] * 64u; g_end = min(n, g_next + 64u); l_next++; }
            if (!pool_empty) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u));
                const uint32_t i = g_next + rank;
                WIDE_STAT(7, 1u); WIDE_STAT(8, (uint32_t)__popcll(__ballot(idle && i < g_end)));
                if (idle && i < g_end) {
                    load(i, o, d);
                    bool d_small, d_normal;
                    dir_class(d, S.dir_bound, d_small, d_normal);
                    quick_det = d_small;
                    // Only rays that pass wide_tame use these reciprocals, and their direction components are normal
                    // numbers between 2^-40 and 2^40: the short form returns the division's bits for them.
                    inv = ray_inverse_quick(d);
                    h = miss_rec();
                    ray = i;
                    if (wide_tame(S, o, d)) { w.cur = 0u; w.sp = 0u; w.best_slot = 0u; guard_words = wide_guard_of(S, d); }
                    else slow = true;
                }
                g_next = min(g_end, g_next + (uint32_t)__popcll(mi));
            }
        }
        bool sharing = false;   // (wave-uniform) lanes are free to take parts of other lanes' rays
#if ORION_WIDE_SHARE
        if (pool_empty) {
            const bool had = w.cur != kWideEmpty;
            sharing = wide_share(w, o, d, inv, h, quick_det, slow, idle, owner, lent);
            if (!had && w.cur != kWideEmpty) guard_words = wide_guard_of(S, d);   // a part of another lane's ray: its direction
        }
#endif
        if (sharing) WIDE_STAT(9, 1u);
        if (__ballot(w.cur != kWideEmpty || slow) == 0ull) {
            if (pool_empty && __ballot(ray != kInvalid || owner != kNoOwner) == 0ull) break;
            continue;
        }
        // rays the wide walk does not take, and lanes whose stack overflowed: the unculled walk
        if (__any(slow)) {
            WIDE_STAT(10, 1u);
            if (slow) {
                h = miss_rec(); traverse_lane<ANY>(S, o, d, h); slow = false; w.cur = kWideEmpty;
                w.best_slot = h.gtri != kInvalid ? S.slot_of[h.gtri] : 0u;   // (the result rule of shared rays compares slots)
            }
        }
        // Node steps until nobody is at a node -- or until enough lanes wait at a leaf, or enough have run
        // out of work to make fetching worthwhile: they would sit idle while the others walk on.
        const bool guard = __any(w.cur != kWideEmpty && guard_words.cell != kGuardNoCell);
        for (;;) {
            const bool at_node = wide_at_node(w);
            if (!__any(at_node)) break;
            if ((uint32_t)__popcll(__ballot(wide_at_leaf(w))) >= kLeafQuorumWide) break;
            if (!pool_empty && (uint32_t)__popcll(__ballot(w.cur == kWideEmpty && !slow)) >= kRefillMin) break;
            WIDE_STAT(1, 1u); WIDE_STAT(2, (uint32_t)__popcll(__ballot(at_node)));
            const bool over = at_node && !wide_node_step(S, o, inv, h.t, w, tid, guard_words, guard);
            if (__any(over)) {   // a stack that is full: that node's subtree goes to the unculled walk
                if (over && !wide_overflow<ANY>(S, o, d, inv, false, h, w, tid)) slow = true;
            }
            if (sharing && __any(w.sp != 0u && w.cur != kWideEmpty)) break;   // somebody has a part to hand out
        }
        const bool at_leaf = wide_at_leaf(w);
        const bool all_quick = __all(!at_leaf || quick_det);
#if ORION_WIDE_STATS == 1
        if (__any(at_leaf)) {
            const uint32_t cnt = at_leaf ? (w.cur >> 25) & 63u : 0u;
            uint32_t mx = cnt, sum = cnt;
            for (int sft = 1; sft < 64; sft <<= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, sft)); sum += (uint32_t)__shfl_xor((int)sum, sft); }
            stat[3] += 1u; stat[4] += (uint32_t)__popcll(__ballot(at_leaf)); stat[5] += mx; stat[6] += sum;
        }
#endif
        if (at_leaf) {
            if (all_quick) wide_leaf_step<ANY, true>(S, o, d, h, w, tid);
            else wide_leaf_step<ANY, false>(S, o, d, h, w, tid);
        }
    }
#if ORION_WIDE_STATS
    stat[11] = 1u;
    if (lane == 0u) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
#if ORION_WIDE_STATS == 1
        for (int k = 0; k < 12; k++) atomicAdd(&g_wide_stats[k], (unsigned long long)stat[k]);   // (thousands of waves on twelve addresses: this lengthens the launch)
#else
        atomicAdd(&g_wide_stats[11], 1ull);
#endif
        const uint32_t wv = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if (wv < 8192u) { g_wide_times[wv][0] = t_start; g_wide_times[wv][1] = t_empty; g_wide_times[wv][2] = t_end; g_wide_times[wv][3] = stat[8]; }
    }
#endif
}

template <bool ANY, bool LISTED>
__global__ __launch_bounds__(kBlock) ORION_DRAW_ATTR void k_query_wide(SceneView S, const float* __restrict__ org, const float* __restrict__ dir, uint32_t n,
                                                       OrionHit* __restrict__ out, uint8_t* __restrict__ occ,
                                                       uint32_t grab, uint32_t* __restrict__ work, const uint32_t* __restrict__ chunks) {
    wide_draw_rays<ANY, LISTED>(S, n, grab, work, chunks,
                                [&](uint32_t i, V3& o, V3& d) { o = ld3(org + 3ull * i); d = ld3(dir + 3ull * i); },
                                [&](uint32_t i, const HitRec& h) { query_store<ANY>(S, i, h, out, occ); });
}

// --- the wide walk's guard table -------------------------------------------------
// Four bits per node and direction cell (wide_guard.h).  One thread per cell walks the node array from the back -- a
// node's children stand behind it, so an inner slot's bit is "the child's four bits are not all zero", read from what this
// thread wrote earlier -- and tests a leaf slot's triangles against its cell.  The node and band records are the same for
// all threads at any moment (scalar loads).  The table must be zero when the kernel starts.
// The first n_virtual nodes are the virtual ones above the meshes' roots, whose bits mean something else (wide_walk.h: "the
// nodes above the meshes' roots"): slot k must be visited whatever its box says -- a leaf (a mesh whose tree is one leaf)
// with an ill-conditioned triangle, or a virtual node with such a slot; a mesh's own inner node never.
__global__ __launch_bounds__(kBlock) void k_guard_cells(const WideNode* __restrict__ nodes, uint32_t n_nodes, uint32_t n_virtual,
                                                        const GuardBand* __restrict__ bands, uint32_t G, uint8_t* table) {
    const uint32_t cell = blockIdx.x * kBlock + threadIdx.x, n_cells = G * G;
    if (cell >= n_cells) return;
    float c[3], sr, cr;
    guard_cell_geometry(G, cell, c, sr, cr);
    for (uint32_t i = n_nodes; i-- > 0u;) {
        const WideNode n = nodes[i];
        const uint32_t refs[4] = {n.ref0, n.ref1, n.ref2, n.ref3};
        uint32_t bits = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t ref = refs[k];
            if (ref == kWideEmpty) continue;
            bool a = false;
            if (ref & kWideLeaf) {
                const uint32_t first = ref & 0x01FFFFFFu, count = (ref >> 25) & 63u;
                for (uint32_t s2 = 0; s2 < count; s2++) a = a || guard_band_meets_cell(bands[first + s2], c, sr, cr);
            } else if (i >= n_virtual || ref < n_virtual) {
                a = (((uint32_t)table[guard_table_offset(ref, cell, n_cells)] >> ((ref & 1u) << 2)) & 15u) != 0u;
            }
            bits |= (a ? 1u : 0u) << k;
        }
        uint8_t* at = table + guard_table_offset(i, cell, n_cells);    // (the byte's other half belongs to node i ^ 1: this thread's too)
        *at = (uint8_t)(*at | (bits << ((i & 1u) << 2)));
    }
}

// --- path tracing --------------------------------------------------------------
struct Queues {
    float4* o;    // origin.xyz, path id (index of the path's radiance slot)
    float4* d;    // direction.xyz, pixel index (RNG counter word 0)
    float4* T;    // throughput.xyz, absolute sample index (RNG counter word 1)
};

// The vertex records of a batch's FIRST bounce hold the path id only (the rest follows from it): a packed array of words in
// the memory of the origin records.
__device__ __forceinline__ uint32_t* first_vertex_ids(const Queues& q) { return reinterpret_cast<uint32_t*>(q.o); }

// Path ids of a batch run pixel by pixel, the batch's samples of one pixel side by side: the 64 lanes of a wave
// that generates camera rays hold (up to) 64 samples of ONE pixel, rays that differ by a fraction of a pixel.  The
// wave-synchronous walk costs the union of its lanes' paths, and so do the shade kernel's in-place shadow rays:
// on a 19 k-triangle model camera rays laid out this way walk 2.9x as fast as 64 neighbouring pixels of one sample
// (tools/ab/sample_order.py), on the Cornell box 8 % faster.
__device__ __forceinline__ void path_pixel(const FrameParams& F, uint32_t pid, uint32_t& pixel, uint32_t& sample, uint32_t& px, uint32_t& py) {
    const uint32_t p = pid / F.n_samples, k = pid - p * F.n_samples;
    py = F.tile_y0 + p / F.tile_w;
    px = F.tile_x0 + p % F.tile_w;
    pixel = py * F.width + px;
    sample = F.spp_begin + k;
}

// Per-sample jitter shared by all pixels (orion/raytracer.cpp:53-63).
__device__ __forceinline__ void sample_jitter(const FrameParams& F, uint32_t sample, float& jx, float& jy) {
    const Words4 r = philox4x32_10(kPatternPixel, sample, 0u, kBlockPattern, F.seed_lo, F.seed_hi);
    const float pixelX = 2.0f / (float)F.width, pixelY = 2.0f / (float)F.height;
    jx = pixelX * u01(r.w[0]); jy = pixelY * u01(r.w[1]);
}

// Primary ray through pixel (px, py) with the sample's jitter, as its three queue records
// (orion/raytracer.cpp:64-72).
__device__ __forceinline__ void primary_ray_at(const FrameParams& F, uint32_t pid, uint32_t px, uint32_t py, uint32_t sample,
                                               float jx, float jy, float4& qo, float4& qd, float4& qT) {
    const float x = 2 * ((float)px / (float)F.width) - 1;
    const float y = 2 * ((float)py / (float)F.height) - 1;
    const float xs = x + jx, ys = -(y + jy);
    const V3 dir = (ld3(F.front) + ld3(F.right) * xs) + ld3(F.up) * ys;
    qo = make_float4(F.eye[0], F.eye[1], F.eye[2], __uint_as_float(pid));
    qd = make_float4(dir.x, dir.y, dir.z, __uint_as_float(py * F.width + px));
    qT = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(sample));
}

// Primary ray of path `pid`, the general way: three integer divisions and one Philox block.
__device__ __forceinline__ void primary_ray(const FrameParams& F, uint32_t pid, float4& qo, float4& qd, float4& qT) {
    uint32_t pixel, sample, px, py;
    path_pixel(F, pid, pixel, sample, px, py);
    float jx, jy;
    sample_jitter(F, sample, jx, jy);
    primary_ray_at(F, pid, px, py, sample, jx, jy, qo, qd, qT);
}

// Primary ray of path `pid` with the sample's jitter taken from the batch's table (the same numbers as primary_ray's).
__device__ __forceinline__ void primary_ray_table(const FrameParams& F, const float2* __restrict__ jitter, uint32_t pid, float4& qo, float4& qd, float4& qT) {
    uint32_t pixel, sample, px, py;
    path_pixel(F, pid, pixel, sample, px, py);
    const float2 j = jitter[sample - F.spp_begin];
    primary_ray_at(F, pid, px, py, sample, j.x, j.y, qo, qd, qT);
}

// The batch's jitters, one per sample (k_jitter fills the table before the batch's first launch).
__global__ __launch_bounds__(kBlock) void k_jitter(FrameParams F, float2* __restrict__ jitter) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= F.n_samples) return;
    float jx, jy;
    sample_jitter(F, F.spp_begin + k, jx, jy);
    jitter[k] = make_float2(jx, jy);
}

// a / b and a % b for a < 2^24 through one float multiplication and a fix-up (inv_b = 1.0f / b): the product is
// off by less than one, so the quotient is exact or one off.
__device__ __forceinline__ void small_divmod(uint32_t a, uint32_t b, float inv_b, uint32_t& q, uint32_t& r) {
    q = (uint32_t)((float)a * inv_b);
    int32_t rem = (int32_t)(a - q * b);
    if (rem < 0) { q--; rem += (int32_t)b; } else if (rem >= (int32_t)b) { q++; rem -= (int32_t)b; }
    r = (uint32_t)rem;
}

// The same for a tile of consecutive path ids: the tile's first pixel and sample are found once per block, a path's
// sample and pixel take two of the short divisions above and its jitter comes from the batch's table -- instead of
// three integer divisions and a Philox block per ray, which were a third of the first bounce's instructions.
struct TileGen {
    bool fast;             // every quantity below 2^24: the short divisions apply
    uint32_t p0, k0;       // pixel and sample (within the batch) of the tile's first path
    float inv_ns, inv_w;
};
__device__ __forceinline__ TileGen tile_gen(const FrameParams& F, uint32_t tile0, uint32_t tile_size) {
    TileGen g{};
    const uint32_t P = F.tile_w * F.tile_h;
    g.fast = F.n_samples + tile_size < (1u << 24) && P < (1u << 24);
    if (g.fast) {
        g.p0 = tile0 / F.n_samples; g.k0 = tile0 - g.p0 * F.n_samples;
        g.inv_ns = 1.0f / (float)F.n_samples;
        g.inv_w = 1.0f / (float)F.tile_w;
    }
    return g;
}
__device__ __forceinline__ void primary_ray_tile(const FrameParams& F, const TileGen& g, const float2* __restrict__ jitter, uint32_t pid, uint32_t slot,
                                                 float4& qo, float4& qd, float4& qT) {
    if (!g.fast) { primary_ray(F, pid, qo, qd, qT); return; }
    uint32_t dp, k, y, x;
    small_divmod(g.k0 + slot, F.n_samples, g.inv_ns, dp, k);
    small_divmod(g.p0 + dp, F.tile_w, g.inv_w, y, x);
    const float2 j = jitter[k];
    primary_ray_at(F, pid, F.tile_x0 + x, F.tile_y0 + y, F.spp_begin + k, j.x, j.y, qo, qd, qT);
}

// Rays per thread of k_extend: kRaysWide while the queue is long (larger tiles group better and
// amortise the block's bookkeeping), 1 once a bounce no longer fills the machine, where the
// launch time is one thread's serial work and more blocks are the only way to shorten it.
constexpr uint32_t kRaysWide = 4;
constexpr uint32_t kTileWide = kBlock * kRaysWide;
#ifndef ORION_NARROW_BELOW
#define ORION_NARROW_BELOW (1u << 21)
#endif
constexpr uint32_t kNarrowBelow = ORION_NARROW_BELOW;   // rays; about one ray per thread slot of the chip x 4
constexpr uint32_t kMaxBins = 64;

// Closest hit of every queued ray (orion/model.hpp:52-62).  Rays that miss end here: a miss
// contributes nothing (orion/raytracer.cpp:113-114).  Rays that hit leave as complete path-vertex
// records (origin, direction, throughput, ids + hit; the first bounce: path id + hit, from which the
// shade launch computes the rest again), compacted into the block's partition of `qout`/`hits`, and inside
// each block's tile of kTile rays they are grouped by the triangle they hit (a counting sort
// through LDS).  The shade kernel that consumes them therefore runs on waves whose vertices lie
// on the same surface: its light-sample rays start on one plane and aim at one light, so the
// wave-synchronous traversal's early exits hold for whole waves, and its material fetches are
// uniform.  n_bins = number of triangle bins (<= kMaxBins; larger ids share the last bin), 1 = no grouping.
// The ray count is read on the device (prev[1], written by the previous bounce's shade launch;
// prev == null: n_bound itself): the host sizes the grid from an older, larger count (n_bound)
// and never has to wait for the exact one.  `clear` is a counter slot of a later bounce, zeroed
// here.  The previous bounce's counters are final by now; one lane stores them to the host's
// pinned mirror (`publish`), where the host polls for them.  Each count travels with the tag in
// one 64-bit store, so the host needs no ordering between the two (and the kernel no fence,
// which at system scope would write back and invalidate the L2).
// Published per bounce: the vertices and the rays of all partitions together, and the largest partition's rays
// (the bound the host sizes the next grids from).
__device__ __forceinline__ void publish_counters(const unsigned long long* cnt, unsigned long long* publish, uint32_t tag, uint32_t part_log2) {
    const unsigned long long hi = (unsigned long long)tag << 32;
    uint32_t vertices = 0, rays = 0, most = 0;
    for (uint32_t k = 0; k < (1u << part_log2); k++) {
        const uint32_t r = (uint32_t)cnt[k * kPartWords + 1];
        vertices += (uint32_t)cnt[k * kPartWords]; rays += r; most = max(most, r);
    }
    __hip_atomic_store(publish, hi | vertices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(publish + 1, hi | rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(publish + 2, hi | most, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_publish(const unsigned long long* cnt, unsigned long long* publish, uint32_t tag, uint32_t part_log2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) publish_counters(cnt, publish, tag, part_log2);
}

// The closest hits of a bounce's rays on a large scene, found the way the ray queries find theirs: the
// waves of a launch that just fills the chip draw rays from the queue as their lanes finish
// (wide_draw_rays).  Hits go to pre_hits[i], ray by ray; k_extend<MODE_PRE> then compacts them as it
// compacts its own.  work: the grab counters, zero at launch (k_extend<MODE_PRE> resets them).
__global__ __launch_bounds__(kBlock) ORION_DRAW_ATTR void k_extend_wide(SceneView S, Queues qin, const unsigned long long* __restrict__ prev, uint32_t n_bound,
                                                        float4* __restrict__ pre_hits, uint32_t grab, uint32_t* __restrict__ work) {
    uint32_t n = n_bound;
    if (prev) { const unsigned long long m = prev[1]; if (m < n) n = (uint32_t)m; }
    wide_draw_rays<false, false>(S, n, grab, work, nullptr,
                                 [&](uint32_t i, V3& o, V3& d) { const float4 a = qin.o[i], b = qin.d[i]; o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); },
                                 [&](uint32_t i, const HitRec& h) { pre_hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.gtri)); });
}

// GEN: the first bounce.  The rays are the primary rays, computed here from the path index
// instead of being read from a queue (no generate pass, no 64 B/path round trip through HBM), and
// a bit per path says whether it hit anything: a path that misses contributes nothing and its radiance slot is never
// touched, one that hits gets its slot written by its first vertex's shade launch.
template <int MODE, uint32_t kRaysPerThread, bool GEN>
__global__ __launch_bounds__(kBlock) void k_extend(SceneView S, FrameParams F, float4* __restrict__ rad,
                                                   Queues qin, const unsigned long long* __restrict__ prev, uint32_t n_bound,
                                                   Queues qout, HitRec* __restrict__ hits, unsigned long long* __restrict__ counter,
                                                   unsigned long long* __restrict__ clear, unsigned long long* publish, uint32_t tag, uint32_t n_bins,
                                                   const float4* __restrict__ pre_hits, uint32_t* __restrict__ pre_work, const float2* __restrict__ jitter,
                                                   unsigned long long* __restrict__ hit_mask) {
    constexpr uint32_t kTile = kBlock * kRaysPerThread;   // rays per block
    __shared__ float4 s_hit[kTile];
    __shared__ uint16_t s_rank[kTile];
    __shared__ uint32_t s_cnt[kMaxBins];
    __shared__ uint32_t s_off[kMaxBins];
    __shared__ uint32_t s_base;
    // The block's partition: its input (unless the rays are generated here, from path ids that run through the whole
    // batch), its counters and its output.  n_bound and the counts are per partition (GEN: the batch's paths).
    const uint32_t part = blockIdx.x & ((1u << F.part_log2) - 1u);
    const uint32_t qoff = part * F.part_stride;
    const uint32_t tile0 = (GEN ? blockIdx.x : blockIdx.x >> F.part_log2) * kTile;
    counter += part * kPartWords;
    uint32_t n = n_bound;
    if (prev) { const unsigned long long m = prev[part * kPartWords + 1]; if (m < n) n = (uint32_t)m; }
    // Block 0's housekeeping stores come after the traversal on purpose: a store ahead of the
    // scene loads would keep the compiler from issuing them as scalar loads.
    auto housekeeping = [&]() {
        if (blockIdx.x != 0) return;
        if (threadIdx.x < (2u << F.part_log2)) clear[(threadIdx.x >> 1) * kPartWords + (threadIdx.x & 1u)] = 0ull;
        if (prev && threadIdx.x == 0) publish_counters(prev, publish, tag, F.part_log2);
        if (MODE == MODE_PRE && threadIdx.x < kWorkParts) pre_work[kWorkCounters + 16u * threadIdx.x] = 0u;   // k_extend_wide's grab counters, for its next launch
    };
    if (tile0 >= n) { housekeeping(); return; }
    if (threadIdx.x < kMaxBins) s_cnt[threadIdx.x] = 0;
    TileGen gen{};
    if (GEN) gen = tile_gen(F, tile0, kTile);
    __syncthreads();
    for (uint32_t k = 0; k < kRaysPerThread; k++) {
        const uint32_t slot = k * kBlock + threadIdx.x;
        const uint32_t i = tile0 + slot;
        const bool active = i < n;
        V3 o = splat(0.0f), d = splat(1.0f);
        if (active && MODE != MODE_PRE) {
            float4 a, b, c;
            if (GEN) primary_ray_tile(F, gen, jitter, i, slot, a, b, c);
            else { a = qin.o[qoff + i]; b = qin.d[qoff + i]; }
            o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z);
        }
        HitRec h = miss_rec();
        if (MODE == MODE_PRE) {
            if (active) { const float4 p = pre_hits[i]; h.t = p.x; h.u = p.y; h.v = p.z; h.gtri = __float_as_uint(p.w); }
        } else {
            traverse<MODE, false>(S, o, d, h, active);
        }
        s_hit[slot] = make_float4(h.t, h.u, h.v, __uint_as_float(h.gtri));
        if (GEN) {   // which of the wave's 64 paths hit something: all k_resolve needs to know about the others
            const unsigned long long m = __ballot(h.gtri != kInvalid);
            if (lane_id() == 0u && i < n) hit_mask[i >> 6] = m;   // (lane 0 holds the wave's first path)
        }
        if (h.gtri != kInvalid) {
            const uint32_t bin = min(h.gtri, n_bins - 1u);
            s_rank[slot] = (uint16_t)atomicAdd(&s_cnt[bin], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // exclusive scan of the bin counts by the first wave (kMaxBins == 64 lanes)
        const uint32_t c = s_cnt[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t up = __shfl_up(incl, sft);
            if ((int)threadIdx.x >= sft) incl += up;
        }
        s_off[threadIdx.x] = incl - c;
        if (threadIdx.x == 63) s_base = incl ? (uint32_t)atomicAdd(counter, (unsigned long long)incl) : 0u;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t k = 0; k < kRaysPerThread; k++) {
        const uint32_t slot = k * kBlock + threadIdx.x;
        const float4 h = s_hit[slot];
        const uint32_t gtri = __float_as_uint(h.w);
        if (gtri != kInvalid) {
            const uint32_t i = tile0 + slot;
            const uint32_t dst = qoff + base + s_off[min(gtri, n_bins - 1u)] + s_rank[slot];
            if (GEN) {
                // A first vertex's ray, throughput and ids follow from its path id (= i): 4 bytes instead of 48 written
                // here and read by the shade launch, which computes them again (first_vertex_ids).
                first_vertex_ids(qout)[dst] = i;
            } else {
                qout.o[dst] = qin.o[qoff + i];
                qout.d[dst] = qin.d[qoff + i];
                qout.T[dst] = qin.T[qoff + i];
            }
            reinterpret_cast<float4*>(hits)[dst] = h;
        }
    }
    housekeeping();
}

// One bounce of RayTracer::trace after the closest hit is known
// (orion/raytracer.cpp:116-206).  POINT selects the point-light branch.  Adds the vertex's light
// to rad[pid] (lanes with `writer` set do the store) and returns whether the path goes on, with
// the next ray's origin (the biased hit point), direction and throughput.
// SHADOW: the walk of the point lights' shadow rays (MODE unless the caller knows better).
// SOLID: the scene has solid colours only and lights whose vertices share one normal (OrionScene::solid): what the
// general form reaches through chains of dependent loads comes from the mesh records (Mesh::kd and below).
// PARK: the throughput and the path id, which nothing needs while the light samples are traced, wait in LDS meanwhile
// (a slot per thread of the block) instead of in registers the walks would push out to scratch memory.
// DEPTH0: 1 if the caller knows that depth is 0, 0 if it knows that it is not, -1 if it does not know.
template <int MODE, bool POINT, int SHADOW = MODE, bool SOLID = false, bool PARK = false, int DEPTH0 = -1>
__device__ __forceinline__ bool shade_core(const SceneView& S, const FrameParams& F, uint32_t depth, bool live, bool writer,
                                           V3 o, V3 d, V3 T, uint32_t pid, uint32_t pixel, uint32_t sample, const HitRec& h,
                                           float4* __restrict__ rad, V3& biased, V3& next_d, V3& next_T) {
    __shared__ float4 s_park[PARK ? kBlock : 1];
    if (PARK) s_park[threadIdx.x] = make_float4(T.x, T.y, T.z, __uint_as_float(pid));
    const float bias = 1e-3f;
    V3 normal = splat(0.0f), hitPos = splat(0.0f), color = splat(0.0f);
    biased = splat(0.0f);
    float tu = 0.0f, tv = 0.0f;
    uint32_t mesh_id = 0;
    if (live) {
        const Surface sf = surface_at(S, h.gtri, h.u, h.v);
        mesh_id = sf.mesh; tu = sf.tu; tv = sf.tv;
        normal = normalize(sf.normal);
        hitPos = o + d * h.t;
        biased = hitPos + normal * bias;
        if (DEPTH0 == 1 || (DEPTH0 == -1 && depth == 0)) {   // primary ray sees an emissive surface (orion/raytracer.cpp:127-128)
            const Mesh& m = S.meshes[mesh_id];
            const float dn = dot(normalize(d), -normal);
            const V3 ke = SOLID ? ld3(m.ke) : tex_fetch(S, m.tex[kTexEmissive], tu, tv);
            color = (ke * m.area) * dn;
        }
    }

    bool cont = false;
    next_d = splat(0.0f); next_T = splat(0.0f);

    if (!POINT) {
        // Direct light: light_samples points on every emissive mesh (orion/raytracer.cpp:133-159)
        V3 kd = splat(0.0f);
        if (live) {
            const Mesh& m = S.meshes[mesh_id];
            kd = SOLID ? ld3(m.kd) : tex_fetch(S, m.tex[kTexDiffuse], tu, tv);
        }
        const float inv_ls = F.inv_light_samples;
        for (uint32_t e = 0; e < S.n_emissive; e++) {
            const uint32_t lmesh = S.emissive[e];
            const Mesh lm = S.meshes[lmesh];
            V3 contrib = splat(0.0f);
            for (uint32_t l = 0; l < F.light_samples; l++) {
                V3 target = splat(0.0f), sd = splat(1.0f);
                float lint = 0.0f;
                if (live) {
                    // (the counter words pass through an empty asm: left alone, the compiler computes the first round's
                    // products of the loop's Philox blocks ahead of the loop and then spills them around the walks)
                    uint32_t c0 = pixel, c1 = sample;
                    asm volatile("" : "+v"(c0), "+v"(c1));
                    const Words4 r = philox4x32_10(c0, c1, depth, e * F.light_samples + l, F.seed_lo, F.seed_hi);
                    light_point(S, lm, r, target, lint);
                    sd = target - hitPos;
                }
                // A sample whose term is surely zero is not traced.  The term is a product of finite numbers with
                // max(f1 * f2, 0), f1 = dot(normal, l), f2 = dot(light's normal at the hit, -l), l = sd normalised
                // (color_brdf).  On a light with one normal N for all its vertices (Mesh::flat) the interpolated,
                // normalised normal is N's unit vector to within 1e-6 per component, so with a = dot(normal, sd) and
                // b = dot(N, sd) both beyond 1e-4 |sd| -- a hundred times the rounding of either dot product -- f1 has
                // a's sign and f2 the opposite of b's: if a and b agree in sign the product is negative, the term is a
                // zero vector and adding it leaves the sum's bits alone.  (NaN or overflow anywhere fails the
                // comparisons and the ray is traced.)
                bool need = live;
                if (lm.flat) {
                    const float a = fmaf(normal.z, sd.z, fmaf(normal.y, sd.y, normal.x * sd.x));
                    const float b = fmaf(lm.flat_n[2], sd.z, fmaf(lm.flat_n[1], sd.y, lm.flat_n[0] * sd.x));
                    const float lim = 1e-8f * fmaf(sd.z, sd.z, fmaf(sd.y, sd.y, sd.x * sd.x));
                    if (a * b > 0.0f && fminf(a * a, b * b) > lim) need = false;
                }
                HitRec h2 = miss_rec();
                traverse<MODE, false>(S, biased, sd, h2, need);
                if (need && h2.gtri != kInvalid) {   // the sample is visible iff the closest hit is the light itself
                    if (SOLID) {
                        // A light of one colour whose vertices share one normal: the hit's mesh follows from its triangle
                        // id, the normal there is surface_at's weighted sum of that one vector, and nothing is loaded.
                        if (h2.gtri - lm.tri_base < lm.n_tri) {
                            const V3 N = ld3(lm.raw_n);
                            const float w = 1.0f - h2.u - h2.v;
                            const V3 n2 = (N * w + N * h2.u) + N * h2.v;
                            contrib = contrib + color_brdf(kd, normal, hitPos, target, ld3(lm.ke), lint, normalize(n2));
                        }
                    } else {
                        const Surface s2 = surface_at(S, h2.gtri, h2.u, h2.v);
                        if (s2.mesh == lmesh) {
                            const V3 lcol = tex_fetch(S, lm.tex[kTexEmissive], s2.tu, s2.tv);
                            contrib = contrib + color_brdf(kd, normal, hitPos, target, lcol, lint, normalize(s2.normal));
                        }
                    }
                }
            }
            color = color + contrib * inv_ls;
        }
        if (PARK) { const float4 k = s_park[threadIdx.x]; T = mk(k.x, k.y, k.z); pid = __float_as_uint(k.w); }
        if (live) {
#if ORION_EXPERIMENT_NO_RAD == 1
            if (writer && color.x == 12345.0f) rad[pid] = make_float4(color.x, 0.0f, 0.0f, 0.0f);   // timing experiment: no radiance traffic
#elif ORION_EXPERIMENT_NO_RAD == 2
            if (writer) { const V3 acc = T * color; rad[blockIdx.x * kBlock + threadIdx.x] = make_float4(acc.x, acc.y, acc.z, 0.0f); }   // timing experiment: coalesced write, no read
#elif ORION_EXPERIMENT_NO_RAD == 3
            if (writer) {   // timing experiment: three hardware float atomics, nothing to wait for
                const V3 acc = T * color;
                float* rp = reinterpret_cast<float*>(rad + pid);
                unsafeAtomicAdd(rp, acc.x); unsafeAtomicAdd(rp + 1, acc.y); unsafeAtomicAdd(rp + 2, acc.z);
            }
#else
            const V3 term = T * color;
            // A path's first vertex finds no sum yet (nobody reads the slots of the paths that miss).  A later vertex that
            // gathers no light -- every sample occluded or facing away -- would add zeros to a sum that is no negative
            // zero, since it began as 0 + x: the slot keeps its bits untouched, 32 bytes of scattered traffic less.
            const bool first = DEPTH0 == 1 || (DEPTH0 == -1 && depth == 0);
            if (writer && (first || term.x != 0.0f || term.y != 0.0f || term.z != 0.0f)) {
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (!first) r0 = rad[pid];
                const V3 acc = mk(r0.x, r0.y, r0.z) + term;
                rad[pid] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            }
#endif
            if (depth < F.max_depth) {
                // Russian roulette on max(kd), then a cosine-weighted direction (orion/raytracer.cpp:164-194)
                const float cc = tpl_max(tpl_max(kd.x, kd.y), kd.z);
                const Words4 r = philox4x32_10(pixel, sample, depth, kBlockBounce, F.seed_lo, F.seed_hi);
                if (!(u01(r.w[0]) > cc)) {
                    const float sin_theta = sqrt_quick(u01(r.w[1]));   // u01 lies in (2^-24, 1): always in the short form's range
                    const float cos_theta = sqrt_exact(1 - sin_theta * sin_theta);
                    const float psi = u01(r.w[2]) * 2.0f * 3.1415926535f;
                    V3 tangent = cross(normal, mk(0.0f, 1.0f, 0.0f));
                    if (dot(tangent, tangent) == 0.0f) tangent = cross(normal, mk(0.0f, 0.0f, 1.0f));
                    const V3 bitangent = cross(normal, tangent);
                    float sin_psi, cos_psi;
                    sincosf(psi, &sin_psi, &cos_psi);
                    const float a = sin_theta * cos_psi, b = sin_theta * sin_psi, c = cos_theta;
                    next_d = (tangent * a + bitangent * b) + normal * c;
                    next_T = (T * kd) / cc;
                    cont = true;
                }
            }
        }
    } else {
        // Point lights: any hit on the unbounded shadow ray occludes (orion/raytracer.cpp:196-201)
        V3 ka = splat(0.0f), kd = splat(0.0f), ks = splat(0.0f);
        float shininess = 0.0f;
        if (live) {
            const Mesh m = S.meshes[mesh_id];
            ka = tex_fetch(S, m.tex[kTexAmbient], tu, tv);
            kd = tex_fetch(S, m.tex[kTexDiffuse], tu, tv);
            ks = tex_fetch(S, m.tex[kTexSpecular], tu, tv);
            shininess = m.shininess;
        }
        for (uint32_t li = 0; li < F.n_lights; li++) {
            const V3 lpos = ld3(F.lights[li].position);
            V3 sd = splat(1.0f);
            if (live) sd = lpos - hitPos;
            HitRec h2 = miss_rec();
            traverse<SHADOW, true>(S, biased, sd, h2, live);
            if (live && h2.gtri == kInvalid)
                color = color + color_phong(ka, kd, ks, shininess, d, normal, hitPos, lpos, ld3(F.lights[li].color), F.lights[li].intensity);
        }
        if (live) {
            if (writer) {
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (depth != 0) r0 = rad[pid];
                const V3 acc = mk(r0.x, r0.y, r0.z) + T * color;
                rad[pid] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            }
            if (depth < F.max_depth) {   // mirror bounce weighted by the specular texel (orion/raytracer.cpp:203-206)
                next_d = reflect(d, normal);
                next_T = T * ks;
                cont = true;
            }
        }
    }

    return cont;
}

// Input: the `*n_live` vertex records k_extend left (every one of them a hit); the grid is sized
// for the ray count, blocks past the live count leave at once.
// FIRST: the batch's first bounce, whose records are path ids (k_extend<.., GEN>); `jitter` is the batch's table.
template <int MODE, bool POINT, int SHADOW, bool FIRST, bool SOLID = false, bool PARK = false>
__device__ __forceinline__ void shade_block(const SceneView& S, const FrameParams& F, uint32_t depth, Queues qin,
                                            const unsigned long long* __restrict__ n_live, const HitRec* __restrict__ hits,
                                            Queues qout, unsigned long long* __restrict__ out_count, float4* __restrict__ rad,
                                            const float2* __restrict__ jitter) {
    const uint32_t part = blockIdx.x & ((1u << F.part_log2) - 1u), blk = blockIdx.x >> F.part_log2;   // (see k_extend)
    const uint32_t qoff = part * F.part_stride;
    const uint32_t n = (uint32_t)n_live[part * kPartWords];
    if (blk * kBlock >= n) return;
    const uint32_t i = blk * kBlock + threadIdx.x;
    const bool live = i < n;
    V3 o = splat(0.0f), d = splat(1.0f), T = splat(0.0f);
    uint32_t pid = 0, pixel = 0, sample = 0;
    HitRec h = miss_rec();
    if (live) {
        float4 a, b, c;
        if (FIRST) primary_ray_table(F, jitter, first_vertex_ids(qin)[qoff + i], a, b, c);
        else { a = qin.o[qoff + i]; b = qin.d[qoff + i]; c = qin.T[qoff + i]; }
        o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); T = mk(c.x, c.y, c.z);
        pid = __float_as_uint(a.w); pixel = __float_as_uint(b.w); sample = __float_as_uint(c.w);
        const float4 hr = reinterpret_cast<const float4*>(hits)[qoff + i];
        h.t = hr.x; h.u = hr.y; h.v = hr.z; h.gtri = __float_as_uint(hr.w);
    }
    V3 biased, next_d, next_T;
    const bool cont = shade_core<MODE, POINT, SHADOW, SOLID, PARK, FIRST ? 1 : 0>(S, F, depth, live, true, o, d, T, pid, pixel, sample, h, rad, biased, next_d, next_T);
    const uint32_t slot = qoff + queue_slot(cont, false, out_count + part * kPartWords);
    if (cont) {
        qout.o[slot] = make_float4(biased.x, biased.y, biased.z, __uint_as_float(pid));
        qout.d[slot] = make_float4(next_d.x, next_d.y, next_d.z, __uint_as_float(pixel));
        qout.T[slot] = make_float4(next_T.x, next_T.y, next_T.z, __uint_as_float(sample));
    }
}

template <int MODE, bool POINT, bool FIRST>
__global__ __launch_bounds__(kBlock) void k_shade(SceneView S, FrameParams F, uint32_t depth, Queues qin,
                                                  const unsigned long long* __restrict__ n_live,
                                                  const HitRec* __restrict__ hits, Queues qout, unsigned long long* __restrict__ out_count,
                                                  float4* __restrict__ rad, const float2* __restrict__ jitter) {
    shade_block<MODE, POINT, MODE, FIRST>(S, F, depth, qin, n_live, hits, qout, out_count, rad, jitter);
}

// The first vertices of a point-light frame on a large scene: a wave holds samples of one pixel (path_pixel), points a
// fraction of a pixel apart that all look at the same light, and the wave-synchronous walk of such shadow rays costs
// little more than one ray's (nanosuit frame: this launch 1.63 -> 1.10 ms per 64 spp against the per-lane wide walk).
#ifndef ORION_SHADOW0_MODE
#define ORION_SHADOW0_MODE MODE_WAVE
#endif
__global__ __launch_bounds__(kBlock) void k_shade_wide_point0(SceneView S, FrameParams F, uint32_t depth, Queues qin, const unsigned long long* __restrict__ n_live,
                                                              const HitRec* __restrict__ hits, Queues qout, unsigned long long* __restrict__ out_count,
                                                              float4* __restrict__ rad, const float2* __restrict__ jitter) {
    shade_block<MODE_WIDE, true, ORION_SHADOW0_MODE, true>(S, F, depth, qin, n_live, hits, qout, out_count, rad, jitter);
}

// The wave-synchronous BRDF form, the kernel a Cornell-class frame spends two thirds of its time
// in, is held to 72 VGPRs: 7 waves per SIMD instead of 6 is worth 12 % of the step, and with the
// hand-scheduled walk the compiler reaches it without spilling a vector register.
#ifndef ORION_SHADE_WAVES
#define ORION_SHADE_WAVES 7
#endif
template <bool FIRST, bool SOLID>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ORION_SHADE_WAVES, ORION_SHADE_WAVES)))
void k_shade_wave_brdf(SceneView S, FrameParams F, uint32_t depth, Queues qin, const unsigned long long* __restrict__ n_live,
                       const HitRec* __restrict__ hits, Queues qout, unsigned long long* __restrict__ out_count, float4* __restrict__ rad,
                       const float2* __restrict__ jitter) {
    shade_block<MODE_WAVE, false, MODE_WAVE, FIRST, SOLID, true>(S, F, depth, qin, n_live, hits, qout, out_count, rad, jitter);
}

// The last bounces of a batch, one PATH per wave.  Once a few thousand rays are left, a bounce's
// two launches last as long as one ray's serial walk plus one vertex's serial shading, dozens of
// times in a row while the chip idles.  Here every wave takes one of the remaining rays and
// follows its path to the end: all 64 lanes hold the same path, the closest hit is one parallel
// step (traverse_one), the shading runs redundantly on all lanes with lane 0 storing, and no
// queue, count or launch stands between two bounces.  Same operations per path as
// k_extend + k_shade, hence the same radiance bit for bit.
// `prev`: counter slot of the bounce that produced the rays (prev[1] = how many); depth0 = the
// depth of those rays.  The rays traced here are not counted in the profiling statistics (a few
// per mille of a frame's rays; counting them would cost an atomic per wave).
template <bool POINT>
__global__ __launch_bounds__(kBlock) void k_tail(SceneView S, FrameParams F, uint32_t depth0, Queues qin,
                                                 const unsigned long long* __restrict__ prev, float4* __restrict__ rad) {
    uint32_t path = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);   // counted through the partitions' rays, one partition after the other
    uint32_t part = 0;
    for (; part < (1u << F.part_log2); part++) {
        const uint32_t c = (uint32_t)prev[part * kPartWords + 1];
        if (path < c) break;
        path -= c;
    }
    if (part == (1u << F.part_log2)) return;   // whole waves leave: every lane of a wave has the same path
    path += part * F.part_stride;
    const float4 a = qin.o[path], b = qin.d[path], c = qin.T[path];
    V3 o = mk(a.x, a.y, a.z), d = mk(b.x, b.y, b.z), T = mk(c.x, c.y, c.z);
    const uint32_t pid = __float_as_uint(a.w), pixel = __float_as_uint(b.w), sample = __float_as_uint(c.w);
    const bool writer = lane_id() == 0;
    for (uint32_t depth = depth0; depth <= F.max_depth; depth++) {
        HitRec h = miss_rec();
        traverse_one<false>(S, o, d, h);
        if (h.gtri == kInvalid) break;
        V3 biased, next_d, next_T;
        if (!shade_core<MODE_SAME, POINT>(S, F, depth, true, writer, o, d, T, pid, pixel, sample, h, rad, biased, next_d, next_T)) break;
        o = biased; d = next_d; T = next_T;
    }
}

// image[i][j] += tracedColor for the batch's samples in ascending order (orion/raytracer.cpp:84).  A pixel's samples
// lie side by side (path_pixel); the slots of paths that missed everything (hit_mask: a bit per path, k_extend's
// first launch) hold nothing and count as the zero they stand for, which leaves a sum as it is.
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams F, const float4* __restrict__ rad, float4* __restrict__ accum,
                                                    const uint32_t* __restrict__ hit_mask) {
    const uint32_t P = F.tile_w * F.tile_h;
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const uint32_t py = F.tile_y0 + p / F.tile_w, px = F.tile_x0 + p % F.tile_w;
    float4 acc = accum[(size_t)py * F.width + px];
    const size_t first = (size_t)p * F.n_samples;
    uint32_t k = 0;
    for (; k + 16 <= F.n_samples; k += 16) {   // sixteen samples at a time: their loads are under way together
        const size_t i = first + k;
        const uint32_t sh = (uint32_t)(i & 31u);
        const uint32_t w0 = hit_mask[i >> 5], w1 = sh > 16u ? hit_mask[(i >> 5) + 1] : 0u;
        const uint32_t bits = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> sh) & 0xFFFFu;
        float4 c[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { c[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); if (bits >> j & 1u) c[j] = rad[i + j]; }
#pragma unroll
        for (int j = 0; j < 16; j++) { acc.x += c[j].x; acc.y += c[j].y; acc.z += c[j].z; acc.w += 1.0f; }
    }
    for (; k + 8 <= F.n_samples; k += 8) {   // ... then eight
        const size_t i = first + k;
        const uint32_t sh = (uint32_t)(i & 31u);
        const uint32_t w0 = hit_mask[i >> 5], w1 = sh > 24u ? hit_mask[(i >> 5) + 1] : 0u;
        const uint32_t bits = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> sh) & 0xFFu;
        float4 c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { c[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); if (bits >> j & 1u) c[j] = rad[i + j]; }
#pragma unroll
        for (int j = 0; j < 8; j++) { acc.x += c[j].x; acc.y += c[j].y; acc.z += c[j].z; acc.w += 1.0f; }
    }
    for (; k < F.n_samples; k++) {
        const size_t i = first + k;
        if (hit_mask[i >> 5] >> (i & 31u) & 1u) {
            const float4 c = rad[i];
            acc.x += c.x; acc.y += c.y; acc.z += c.z;
        }
        acc.w += 1.0f;
    }
    accum[(size_t)py * F.width + px] = acc;
}

} // namespace
