// a-2, LDS-tiled: CIC / TSC scatter-add through LDS-resident grid tiles.
//
// Random global float atomics run ~17x below the coalesced atomic rate on
// MI355X (MI355X_MICROARCH.md "Global float atomics"), so the deposit is
// restructured so that all 8 / 27 updates of a particle land in LDS:
//
//   A1 count   one thread per particle (coalesced position loads): tile id of
//              its base cell.  Inside a wave, consecutive particles of the same
//              tile form a run; only the run head bumps the tile's particle
//              count (one atomic per run, not per particle).
//   B  scan    exclusive prefix sum of the per-tile counts.
//   A2 fill    same walk; the run head reserves `len` slots of the tile's
//              segment of a particle-INDEX list and the run's lanes write
//              their 4-byte indices there (coalesced for coherent input).
//              Particle data itself is never moved.
//   D  deposit one workgroup per tile: zero an LDS tile (+ window halo), walk
//              the tile's index segment with all 256 lanes, gather positions,
//              ds_add_f32/f64 into LDS, flush the tile to HBM once.
//
// Spatially coherent input (lattice / Morton / slab ordered snapshots) gives
// long runs: few atomics in A1/A2 and coalesced gathers in D.  Fully shuffled
// input degrades to one run per particle — still correct.
//
// Default is the SINGLE-PASS variant: A1 and B are skipped, every tile gets a
// fixed-capacity segment of the index list (twice the mean occupancy) and A2
// reserves slots with the same one-atomic-per-(interval, tile) scheme; particles
// that do not fit go to an overflow list and are deposited with global atomics
// by a small extra kernel.  Exact for any input; the two-pass variant
// (AST_PAINT_TWO_PASS) is kept for strongly clustered data.
#include "ast_common.h"
#include "paint_tile_geom.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

using ast::Window;

using ast::TX;
using ast::TY;
using ast::TZ;
using ast::RingMap;
#ifndef PAINT_ABLATE
#define PAINT_ABLATE 0        // perf experiments only: 1 no flush stores, 2 no LDS atomics, 32 no gather,
                              // 64 no flush barriers, 128 no flush, 256 no deposit arithmetic
#endif
constexpr int ablate = PAINT_ABLATE;

// -DPAINT_STAMPS: s_memtime stamps of wave 0 of 64 sample workgroups of the column deposit kernel
// (scripts/perf_timeline.py reads them through ast_debug_stamps)
#ifdef PAINT_STAMPS
__device__ unsigned long long g_stamps[64 * 4096];
__device__ unsigned g_nstamp[64];
#define STAMP_DECL unsigned nst_ = 0
#define STAMP(id)                                                                                     \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x % 257 == 3 && blockIdx.x / 257 < 64 && nst_ < 4096)         \
            g_stamps[(blockIdx.x / 257) * 4096 + nst_++] = (__builtin_amdgcn_s_memtime() << 8) | (id); \
    } while (0)
#define STAMP_END                                                                                     \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x % 257 == 3 && blockIdx.x / 257 < 64) g_nstamp[blockIdx.x / 257] = nst_; \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(id) do {} while (0)
#define STAMP_END do {} while (0)
#endif
#if defined(PAINT_STAMPS) && PAINT_STAMPS == 2      // stamps in the index kernel instead of the deposit kernel
#define FSTAMP_DECL STAMP_DECL
#define FSTAMP(id) STAMP(id)
#define FSTAMP_END STAMP_END
#define DSTAMP_DECL
#define DSTAMP(id) do {} while (0)
#define DSTAMP_END do {} while (0)
#else
#define FSTAMP_DECL
#define FSTAMP(id) do {} while (0)
#define FSTAMP_END do {} while (0)
#define DSTAMP_DECL STAMP_DECL
#define DSTAMP(id) STAMP(id)
#define DSTAMP_END STAMP_END
#endif

struct TileGeom {
    int n, x_start, nx_alloc;
    int ntx, nty, ntz;
    double inv_dx;
    double shift;        // added to every coordinate in grid units (interlacing paints at +1/2 cell)
    int off_lo, off_hi;  // buffer planes [off_lo, off_hi) get `offset` subtracted (a slab's ghost planes must not)
};

// position -> grid units: x * n/L + shift as ONE fma (for shift = 0 exactly the rounded product)
template <typename T> __device__ inline double grid_coord(T x, const TileGeom& g) { return __fma_rn((double)x, g.inv_dx, g.shift); }

// base cell (window centre for TSC, lower corner for CIC) -> tile id, or 0xffffffff when the
// base plane is outside the buffer.  The common case is a position inside the box: cell =
// (int)floor(s) needs no reduction at all, and only then does the deposit kernel's unreduced
// lookup agree, so everything else (one box length out: locate_rel would do, but the flag is
// what matters) takes the general path and marks the tile's column in col_flags.
// PLAINX: the buffer is the whole periodic grid (x_start = 0, nx_alloc = n).
template <typename T, int W, bool PLAINX>
__device__ inline uint32_t tile_of(T x, T y, T z, const TileGeom& g, uint32_t* __restrict__ col_flags) {
    const double sx = grid_coord(x, g), sy = grid_coord(y, g), sz = grid_coord(z, g);
    int cx = (int)floor(W == 2 ? sx : sx + 0.5);
    int cy = (int)floor(W == 2 ? sy : sy + 0.5);
    int cz = (int)floor(W == 2 ? sz : sz + 0.5);
    const bool far = max(max((unsigned)cx, (unsigned)cy), (unsigned)cz) >= (unsigned)g.n;
    if (far) {
        double f;
        cx = ast::locate<W>(sx, g.n, f);
        cy = ast::locate<W>(sy, g.n, f);
        cz = ast::locate<W>(sz, g.n, f);
    }
    unsigned bx = (unsigned)cx;
    if (!PLAINX) {
        int b = cx - g.x_start;
        if (b < 0) b += g.n;
        if (b >= g.nx_alloc) return 0xffffffffu;
        bx = (unsigned)b;
    }
    // 24-bit multiplies (v_mad_u32_u24, full rate; a 32-bit multiply-add compiles to the quarter-rate v_mad_u64_u32):
    // tile rows, columns and tiles per column are all below 2^24 (tiled_geometry checks)
    const uint32_t col = __umul24(bx / TX, (unsigned)g.nty) + (unsigned)cy / TY;
    // (tested with a device-scope load first: all far particles of a lattice plane - half of its 2^20 particles under the
    // TSC rule, whose nearest cell is n for the last half cell - raise the flags of the same 128 columns, and 500 000 atomics
    // queueing at 128 addresses cost the TSC grouping 1.4 of its 5.3 ms at 1024^3; once a flag is up nobody else writes it)
    if (far && col_flags && !(__hip_atomic_load(&col_flags[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u))
        atomicOr(&col_flags[col], 1u);
    return __umul24(col, (unsigned)g.ntz) + (unsigned)cz / TZ;
}

// Run structure of one wave's 64 consecutive particles.
struct WaveRuns {
    bool head;       // this lane starts a run
    int len;         // run length (valid on head lanes)
    int head_lane;   // lane of the head of the run this lane belongs to (live lanes)
};

__device__ inline WaveRuns wave_runs(uint32_t key, bool live, int lane) {
    WaveRuns r;
    // key of lane - 1 in ONE instruction (v_mov_b32_dpp wave_shr:1; lane 0 keeps the fill value)
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)key, 0x138, 0xf, 0xf, false);
    r.head = live && (lane == 0 || prev != key);
    const unsigned long long hmask = __ballot(r.head);
    const unsigned long long dmask = __ballot(!live);
    const unsigned long long above = ~((2ull << lane) - 1ull);           // bits > lane (lane 63: none)
    const unsigned long long stop = (hmask | dmask) & above;
    const int end = stop ? __ffsll((long long)stop) - 1 : 64;
    r.len = end - lane;
    const unsigned long long at_or_below = hmask & ~above;
    r.head_lane = at_or_below ? 63 - __clzll((long long)at_or_below) : 0;
    return r;
}

constexpr int IDX_UNROLL = 4;      // particles per thread per trip: keeps 12 loads in flight
#ifndef AGG_TRIPS_N
#define AGG_TRIPS_N 4
#endif
constexpr int AGG_TRIPS = AGG_TRIPS_N;       // trips of 1024 particles between flushes of the LDS table
constexpr int AGG_SLOTS = 256;     // direct-mapped by (tile id & 255)
constexpr uint32_t SLOT_EMPTY = 0xffffffffu;
constexpr uint32_t CODE_DONE = 0xffffffffu;

// A run of consecutive particle ids placed by one thread: slots from the tile's counter, then the
// ids.  Deliberately NOT inlined: its returning atomics and stores would otherwise share registers
// with the caller's common path, and the compiler then guards that path with s_waitcnt vmcnt(0),
// i.e. waits for the prefetched positions at every particle.
template <int MODE>
__device__ __noinline__ void place_run_slow(uint32_t key, uint32_t p1, uint32_t len, const uint32_t* __restrict__ tile_off,
                                            uint32_t* __restrict__ tile_fill, uint32_t* __restrict__ index, uint32_t cap,
                                            uint32_t* __restrict__ ovf, unsigned long long* __restrict__ ovf_count) {
    const uint32_t b = atomicAdd(&tile_fill[key], len);
    for (uint32_t i = 0; i < len; ++i) {
        if (MODE == 1) index[(size_t)tile_off[key] + b + i] = p1 + i;
        else if (b + i < cap) index[(size_t)key * cap + b + i] = p1 + i;
        else ovf[atomicAdd(ovf_count, 1ull)] = p1 + i;
    }
}

// Both passes walk the particle array in contiguous intervals of 4096 particles,
// one interval per workgroup.  Global atomics are the bottleneck of a naive version
// (one per run: ~2/3 of the kernel time), so run heads first combine into a small
// LDS table keyed by a hash of the tile id; the table is flushed once per interval
// with ONE global atomic per distinct tile.  Runs whose slot is taken by another tile
// park in an LDS miss list that is placed with the flush.  In the FILL pass the final
// position of a particle is only known after the flush, so each thread parks
// (slot, offset-in-interval) codes of its 16 particles in LDS meanwhile.
// s_memtime timeline of a workgroup: trips (cell lookup, runs, table: vector-issue
// bound) 53 %, scatter of the parked codes (store-issue bound) 28 %, flush 12 %.
// MODE 0: count (two-pass A1)   1: fill at exact offsets (two-pass A2)
// MODE 2: single pass — tile t owns index[t*cap, (t+1)*cap); overflow goes to ovf[]
// STRIDE = 4, np_dev: the "particles" are the {x, y, z, m} records of the scatter path's late list, as many as *np_dev says
// (at most np; fewer than np_min: nothing to do here, the list is deposited with global atomics instead).
template <typename T, int W, int MODE, bool PLAINX, int STRIDE = 3>
__global__ void __launch_bounds__(256)
tile_index_kernel(const T* __restrict__ pos, size_t np, TileGeom g, uint32_t* __restrict__ tile_count,
                  const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_fill,
                  uint32_t* __restrict__ index, uint32_t cap, uint32_t* __restrict__ ovf,
                  unsigned long long* __restrict__ ovf_count, uint32_t* __restrict__ col_flags,
                  unsigned long long* dropped, const unsigned long long* __restrict__ np_dev = nullptr,
                  unsigned long long np_min = 0) {
    constexpr bool FILL = MODE != 0;
    if (np_dev != nullptr) {                          // uniform, before any barrier
        const unsigned long long nd = *np_dev;
        if (nd < np_min || nd == 0) return;
        np = nd < np ? (size_t)nd : np;
    }
    // skey/scnt are re-armed by each thread as soon as it leaves the scatter loop, so the
    // scatter reads the interval's results from sdst/sroom, which only the next flush rewrites:
    // sdst = first index element of the slot's tile segment part, sroom = elements left in it
    __shared__ uint32_t skey[AGG_SLOTS], scnt[AGG_SLOTS], sroom[AGG_SLOTS];
    __shared__ unsigned long long sdst[AGG_SLOTS];
    __shared__ uint32_t codes[FILL ? AGG_TRIPS * IDX_UNROLL : 1][256];
    constexpr uint32_t MISS_CAP = FILL ? 512 : 1;
    __shared__ uint32_t smiss_key[MISS_CAP], smiss_p[MISS_CAP], smiss_len[MISS_CAP], smiss_n;
    FSTAMP_DECL;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    skey[tid] = SLOT_EMPTY;
    scnt[tid] = 0;
    if (tid == 0) smiss_n = 0;
    __syncthreads();
    const size_t per_trip = 256 * IDX_UNROLL;
    const size_t per_interval = per_trip * AGG_TRIPS;
    const size_t nintervals = (np + per_interval - 1) / per_interval;
    unsigned long long ndrop = 0;
    // positions are fetched one trip ahead (two register sets, used alternately; the last trip of an
    // interval fetches the first trip of the workgroup's next interval, across its flush)
    auto fetch = [&](size_t pbase, T (&x)[IDX_UNROLL], T (&y)[IDX_UNROLL], T (&z)[IDX_UNROLL]) {
#pragma unroll
        for (int u = 0; u < IDX_UNROLL; ++u) {
            // unconditional (lanes past the end re-load the last particle): a predicated load
            // costs a branch and a full s_waitcnt each, i.e. IDX_UNROLL serial round trips
            const size_t p = min(pbase + (size_t)u * 256 + tid, np - 1);
            x[u] = pos[STRIDE * p + 0];
            y[u] = pos[STRIDE * p + 1];
            z[u] = pos[STRIDE * p + 2];
        }
    };
    auto place_run = [&](uint32_t key, uint32_t p1, uint32_t len) {
        place_run_slow<MODE>(key, p1, len, tile_off, tile_fill, index, cap, ovf, ovf_count);
    };
    static_assert(AGG_TRIPS % 2 == 0, "two register sets");
    T xa[IDX_UNROLL], ya[IDX_UNROLL], za[IDX_UNROLL], xb[IDX_UNROLL], yb[IDX_UNROLL], zb[IDX_UNROLL];
    if ((size_t)blockIdx.x < nintervals) fetch((size_t)blockIdx.x * per_interval, xa, ya, za);
    for (size_t interval = blockIdx.x; interval < nintervals; interval += gridDim.x) {
        const size_t p0 = interval * per_interval;
        const bool full = p0 + per_interval <= np;            // uniform: no lane of the interval is past the end
        auto process = [&](int trip, const T (&x)[IDX_UNROLL], const T (&y)[IDX_UNROLL], const T (&z)[IDX_UNROLL]) {
#pragma unroll
            for (int u = 0; u < IDX_UNROLL; ++u) {
                const size_t p = p0 + (size_t)trip * per_trip + (size_t)u * 256 + tid;
                const bool valid = full || p < np;
                uint32_t key = tile_of<T, W, PLAINX>(x[u], y[u], z[u], g, MODE != 1 ? col_flags : nullptr);
                if (!valid) key = 0xffffffffu;
                const bool live = key != 0xffffffffu;
                if (!PLAINX && valid && !live) ++ndrop;
                const WaveRuns r = wave_runs(key, live, lane);
                bool parked = false;            // run handled outside the table (miss list, or placed by its head lane)
                uint32_t val = 0;               // slot << 16 | offset in the interval
                if (r.head) {
                    const uint32_t slot = (key * 2654435761u) >> 24;      // 256 slots; a plain mask maps the x / y neighbour columns (tile id +- 4096, +- 32) onto the column's own slots
                    const uint32_t old = atomicCAS(&skey[slot], SLOT_EMPTY, key);
                    if (old == SLOT_EMPTY || old == key) {
                        val = (slot << 16) | atomicAdd(&scnt[slot], (uint32_t)r.len);
                    } else if (MODE == 0) {
                        atomicAdd(&tile_count[key], (uint32_t)r.len);
                    } else {
                        // the slot belongs to another tile.  A returning global atomic here would park the
                        // whole wave for a memory round trip (with half-cell jitter ~85 % of the waves
                        // hold a stray of a neighbour column), so the run - consecutive particle ids -
                        // goes to a list that is placed once per interval, all atomics in flight together.
                        // Nothing the common path reads may depend on a vector-memory result: the
                        // compiler would wait for the prefetched positions with it.
                        parked = true;
                        const uint32_t mi = atomicAdd(&smiss_n, 1u);
                        if (mi < MISS_CAP) {
                            smiss_key[mi] = key;
                            smiss_p[mi] = (uint32_t)p;
                            smiss_len[mi] = (uint32_t)r.len;
                        } else {
                            place_run(key, (uint32_t)p, (uint32_t)r.len);      // list full (scattered input)
                        }
                    }
                }
                if (FILL) {
                    const uint32_t valb = __shfl(val, r.head_lane, 64) + (uint32_t)(lane - r.head_lane);
                    uint32_t code = live ? valb : CODE_DONE;
                    if (__any(parked)) {                  // uniform
                        if (__shfl((int)parked, r.head_lane, 64)) code = CODE_DONE;
                    }
                    codes[trip * IDX_UNROLL + u][tid] = code;
                }
            }
        };
#pragma unroll
        for (int trip = 0; trip < AGG_TRIPS; trip += 2) {
            FSTAMP(1);
            fetch(p0 + (size_t)(trip + 1) * per_trip, xb, yb, zb);
            process(trip, xa, ya, za);
            FSTAMP(2);
            fetch(trip + 2 < AGG_TRIPS ? p0 + (size_t)(trip + 2) * per_trip : (interval + gridDim.x) * per_interval, xa, ya, za);
            process(trip + 1, xb, yb, zb);
            FSTAMP(2);
        }
        __syncthreads();
        FSTAMP(3);
        if (skey[tid] != SLOT_EMPTY) {
            const uint32_t t = skey[tid], c = scnt[tid];
            if (MODE == 0) atomicAdd(&tile_count[t], c);
            else if (MODE == 1) { sdst[tid] = (unsigned long long)tile_off[t] + atomicAdd(&tile_fill[t], c); sroom[tid] = 0xffffffffu; }
            else {
                const uint32_t base = min(atomicAdd(&tile_fill[t], c), cap);
                sdst[tid] = (unsigned long long)t * cap + base;
                sroom[tid] = cap - base;
            }
        }
        if (FILL) {
            const uint32_t nm = min(smiss_n, MISS_CAP);
            for (uint32_t e = tid; e < nm; e += 256) {
                place_run(smiss_key[e], smiss_p[e], smiss_len[e]);
            }
        }
        FSTAMP(4);
        __syncthreads();
        FSTAMP(5);
        if (FILL) {
            if (tid == 0) smiss_n = 0;
#pragma unroll 4
            for (int j = 0; j < AGG_TRIPS * IDX_UNROLL; ++j) {
                const uint32_t c = codes[j][tid];
                if (c == CODE_DONE) continue;
                const uint32_t p = (uint32_t)(p0 + (size_t)j * 256 + tid);
                const uint32_t slot = c >> 16, at = c & 0xffffu;
                if (at < sroom[slot]) index[sdst[slot] + at] = p;
                else ovf[atomicAdd(ovf_count, 1ull)] = p;
            }
        }
        skey[tid] = SLOT_EMPTY;
        scnt[tid] = 0;
        FSTAMP(6);
        __syncthreads();
        FSTAMP(7);
    }
    FSTAMP_END;
    if (MODE != 1 && dropped && ndrop) atomicAdd(dropped, ndrop);
}

// ------------------------------------------------------------------------------------
// Single pass, COMPACT lists (the format of the overwrite column walk): per tile
//   * group records {first, mask}: the particles first + b, for every set bit b of mask, of one
//     32-particle window (first is a multiple of 32) - 8 bytes for up to 32 particles.  The walk
//     reads a window's positions with one fully coalesced, line-aligned 384-byte access, so a
//     spatially coherent input costs ~0.3 B per particle of list traffic instead of a 4-byte id,
//     and neither the list nor the gather depends on where the runs are interrupted;
//   * stray records {x, y, z, m}: a particle with fewer than MINPOP companions of its tile in its
//     window is COPIED (the index pass has it in registers anyway... it re-reads it from L2 when
//     the destination is known): 16 bytes written and read back contiguously, instead of a 4-byte
//     id plus a 64/128-byte line per stray in the walk's gather.  Fully scattered input turns
//     into a binned copy of the particles.
// Slots are reserved like in tile_index_kernel: an LDS table keyed by tile (one returning 64-bit
// global atomic per distinct tile and interval: run count in the high word, stray count in the low),
// a miss list for hash collisions, overflow list for what does not fit a tile's segments.
constexpr int MINPOP = 8;              // records with fewer particles would waste the walk's lanes
#ifndef GROUP_ITERS_N
#define GROUP_ITERS_N 3
#endif
#ifndef GROUP_CAND
#define GROUP_CAND 16, 8, 24, 0
#endif
constexpr int GROUP_ITERS = GROUP_ITERS_N;         // tiles that can get a record per 32-particle window
constexpr uint32_t LREC_CAP = 512;     // records parked in LDS per interval (128 windows x up to 3; rest: slow path)
struct GroupRec { uint32_t first, mask; };

// a stray copy is {x, y, z, m}, or {x, y, z} when the paint has no masses (has_mass false: the walk reads it as FMT 2)
template <typename T>
__device__ inline void store_stray(T* __restrict__ strays, size_t slot, T x, T y, T z, T m, bool has_mass) {
    typedef T vec4_t __attribute__((ext_vector_type(4)));
    struct Rec3 { T x, y, z; };
    if (has_mass) *reinterpret_cast<vec4_t*>(strays + 4 * slot) = vec4_t{x, y, z, m};
    else *reinterpret_cast<Rec3*>(strays + 3 * slot) = Rec3{x, y, z};
}

// One group or one stray placed by one thread (hash collision in the LDS table, or LDS lists full).
// NOT inlined, for the reason given at place_run_slow.
template <typename T>
__device__ __noinline__ void place_group_slow(uint32_t key, uint32_t a, uint32_t mask, const T* __restrict__ pos,
                                              const T* __restrict__ mass, unsigned long long* __restrict__ fill64,
                                              GroupRec* __restrict__ recs, uint32_t rcap, T* __restrict__ strays,
                                              uint32_t scap, uint32_t* __restrict__ ovf,
                                              unsigned long long* __restrict__ ovf_count) {
    if (mask) {
        const uint32_t r = (uint32_t)(atomicAdd(&fill64[key], 1ull << 32) >> 32);
        if (r < rcap) { recs[(size_t)key * rcap + r] = GroupRec{a, mask}; return; }
        for (uint32_t m = mask; m; m &= m - 1) ovf[atomicAdd(ovf_count, 1ull)] = a + (uint32_t)__ffs((int)m) - 1u;
    } else {
        const uint32_t sidx = (uint32_t)atomicAdd(&fill64[key], 1ull);
        if (sidx < scap) store_stray(strays, (size_t)key * scap + sidx, pos[3 * (size_t)a], pos[3 * (size_t)a + 1],
                                     pos[3 * (size_t)a + 2], mass ? mass[a] : (T)1, mass != nullptr);
        else ovf[atomicAdd(ovf_count, 1ull)] = a;
    }
}

// Phases of one interval of 4096 contiguous particles (one workgroup):
//   1 trips    cell lookup, grouping by ballots; group records and strays (position, tile) are APPENDED to
//              LDS lists - no table lookups here, so the loop carries no dependent LDS round trips;
//   2 slots    one thread per list entry: tile -> table slot (CAS) and the entry's offset in the slot;
//   3 reserve  one thread per used slot: ONE returning 64-bit global atomic per distinct tile;
//   4 scatter  one thread per entry: 8-byte records and 16-byte stray copies to their final place.
// A table slot taken by another tile, or a full LDS list, sends the entry through place_group_slow
// (its global atomics are issued by independent threads, all in flight together).
constexpr uint32_t LST_CAP = 768;      // strays parked in LDS per interval (the bench input has ~330)

// One late particle (AST_PAINT_XSORTED: its tile row has been handed to the column walk already) goes to the overflow
// list.  NOT inlined, for the reason given at place_run_slow.
__device__ __noinline__ void place_late_slow(uint32_t p, uint32_t* __restrict__ ovf, unsigned long long* __restrict__ ovf_count) {
    ovf[atomicAdd(ovf_count, 1ull)] = p;
}

// The kernel handles the particles [p_begin, p_end) (p_begin a multiple of 32: the records' windows are aligned in
// GLOBAL particle ids).  Tiles closed_lo <= id < closed_lo + closed_n are closed: their particles go to the overflow list.
template <typename T, int W, bool PLAINX>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 5 : 1, sizeof(T) == 4 ? 5 : 8)))      // fp32: 96 VGPRs, five workgroups per CU (what the 31 KB of LDS allow); the slab variant spills 3 VGPRs for it and still gains (0.57 -> 0.52 ms for one rank of eight)
tile_group_kernel(const T* __restrict__ pos, const T* __restrict__ mass, size_t p_begin, size_t np, TileGeom g,
                  unsigned long long* __restrict__ fill64, GroupRec* __restrict__ recs, uint32_t rcap,
                  T* __restrict__ strays, uint32_t scap, uint32_t* __restrict__ ovf,
                  unsigned long long* __restrict__ ovf_count, uint32_t* __restrict__ col_flags,
                  unsigned long long* dropped, uint32_t closed_lo, uint32_t closed_n, uint32_t closed_mod = 0u,
                  uint32_t xcd_map = 0u) {
    __shared__ uint32_t skey[AGG_SLOTS], srun[AGG_SLOTS], sstray[AGG_SLOTS], sroom_run[AGG_SLOTS], sroom_stray[AGG_SLOTS];
    __shared__ unsigned long long sdst_run[AGG_SLOTS], sdst_stray[AGG_SLOTS];
    // group records: first, mask, tile key (phase 2 turns the key into slot << 16 | offset, DST_NONE = placed already)
    __shared__ uint32_t lrec_first[LREC_CAP], lrec_mask[LREC_CAP], lrec_key[LREC_CAP], lrec_n;
    // strays: position, particle id, tile key (same conversion)
    __shared__ T lst_x[LST_CAP], lst_y[LST_CAP], lst_z[LST_CAP];
    __shared__ uint32_t lst_p[LST_CAP], lst_key[LST_CAP], lst_n;
    constexpr uint32_t DST_NONE = 0xffffffffu;
    FSTAMP_DECL;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    skey[tid] = SLOT_EMPTY;
    srun[tid] = 0;
    sstray[tid] = 0;
    if (tid == 0) { lst_n = 0; lrec_n = 0; }
    __syncthreads();
    const size_t per_trip = 256 * IDX_UNROLL;
    const size_t per_interval = per_trip * AGG_TRIPS;
    const size_t nintervals = (np - p_begin + per_interval - 1) / per_interval;      // np: END of the range
    unsigned long long ndrop = 0;
    // pbase is UNIFORM: the trip's base address stays in scalar registers and a lane adds a 32-bit offset (a 64-bit
    // lane address cost two quarter-rate v_mad_u64_u32 and a 64-bit compare / select per load: ~9 of the kernel's ~93
    // vector instructions per particle).  Unconditional loads: lanes past the end re-read the last particle.
    auto fetch = [&](size_t pbase, T (&x)[IDX_UNROLL], T (&y)[IDX_UNROLL], T (&z)[IDX_UNROLL]) {
        const size_t pb = min(pbase, np - 1);
        const T* __restrict__ bp = pos + 3 * pb;
        const uint32_t last = (uint32_t)min((size_t)(256 * IDX_UNROLL - 1), np - 1 - pb);
#pragma unroll
        for (int u = 0; u < IDX_UNROLL; ++u) {
            const uint32_t rel = 3u * min((uint32_t)(u * 256 + tid), last);
            x[u] = bp[rel + 0];
            y[u] = bp[rel + 1];
            z[u] = bp[rel + 2];
        }
    };
    auto slow = [&](uint32_t key, uint32_t a, uint32_t mask) {
        place_group_slow<T>(key, a, mask, pos, mass, fill64, recs, rcap, strays, scap, ovf, ovf_count);
    };
    // the table slot of a tile, or -1 when it belongs to another tile of this interval
    auto slot_of = [&](uint32_t key) -> int {
        const uint32_t slot = (key * 2654435761u) >> 24;
        const uint32_t old = atomicCAS(&skey[slot], SLOT_EMPTY, key);
        return (old == SLOT_EMPTY || old == key) ? (int)slot : -1;
    };
    static_assert(AGG_TRIPS % 2 == 0, "two register sets");
    T xa[IDX_UNROLL], ya[IDX_UNROLL], za[IDX_UNROLL], xb[IDX_UNROLL], yb[IDX_UNROLL], zb[IDX_UNROLL];
    // xcd_map: workgroup b runs on XCD b mod 8 (round-robin dispatch: observed, nothing depends on it for correctness); with
    // the map each XCD takes a CONTIGUOUS eighth of the intervals, so that the workgroups which append to one tile's
    // segments - neighbouring intervals, and the same rows 256 intervals later, plane after plane - share an L2
    size_t vb = blockIdx.x;
    if (xcd_map && gridDim.x % 8u == 0u) vb = (size_t)(blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u;
    if (vb < nintervals) fetch(p_begin + vb * per_interval, xa, ya, za);
    for (size_t interval = vb; interval < nintervals; interval += gridDim.x) {
        const size_t p0 = p_begin + interval * per_interval;
        const bool full = p0 + per_interval <= np;
        const uint32_t p0_32 = (uint32_t)p0, left = (uint32_t)min(np - p0, per_interval);      // particle ids fit 32 bits (checked by the host)
        auto process = [&](int trip, const T (&x)[IDX_UNROLL], const T (&y)[IDX_UNROLL], const T (&z)[IDX_UNROLL]) {
#pragma unroll
            for (int u = 0; u < IDX_UNROLL; ++u) {
                const uint32_t prel = (uint32_t)(trip * (int)per_trip + u * 256 + tid);
                const uint32_t p = p0_32 + prel;
                const bool valid = full || prel < left;
                uint32_t key = tile_of<T, W, PLAINX>(x[u], y[u], z[u], g, col_flags);
                if (!valid) key = 0xffffffffu;
                if (!PLAINX && valid && key == 0xffffffffu) ++ndrop;
                // closed tiles: [closed_lo, closed_lo + closed_n), taken modulo closed_mod tiles when that is set (slab
                // buffers, staged paint: the rows that hold the ghost planes are closed first, then rows 0, 1, ... - one
                // range that wraps around the end of the buffer).  A particle for a closed tile of a STAGED paint cannot
                // be deposited any more - its rows may have left the GPU already: it is counted as dropped, which is what
                // tells the caller that the input was not ordered as promised.
                uint32_t dkey = key - closed_lo;
                if (!PLAINX && closed_mod && key < closed_lo) dkey += closed_mod;
                if (dkey < closed_n && key != 0xffffffffu) {       // (closed_n = 0: never)
                    // x-sorted single-call paint: a late particle goes through the overflow list, deposited at the end of the
                    // call (the result never depends on the order that was assumed).  STAGED paint: dropped cleanly - on the
                    // overflow list a later FOLD would deposit the part of its window that falls into rows not yet folded,
                    // a partial, order-dependent deposit; it is counted, and the caller's check() raises
                    if (!PLAINX && closed_mod) ++ndrop;
                    else place_late_slow(p, ovf, ovf_count);
                    key = 0xffffffffu;
                }
                const bool live = key != 0xffffffffu;
                // up to GROUP_ITERS tiles per 32-lane window get a group record.  The candidates are the tiles of
                // three fixed lanes of the window (two v_readlane each, no cross-lane search): in a spatially
                // coherent input nearly every window is one tile plus a few strays, and the first candidate
                // settles it.  A candidate with fewer than MINPOP takers leaves them as strays.
                bool pending = live, stray = false;
#pragma unroll
                for (int it = 0; it < GROUP_ITERS; ++it) {
                    constexpr int cand[4] = {GROUP_CAND};
                    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, cand[it]);
                    const uint32_t k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, cand[it] + 32);
                    const bool match = pending && key == (half ? k1 : k0);
                    const unsigned long long mb = __ballot(match);
                    const uint32_t mlo = (uint32_t)mb, mhi = (uint32_t)(mb >> 32);
                    const uint32_t mm = half ? mhi : mlo;
                    const bool big = half ? __popc(mhi) >= MINPOP : __popc(mlo) >= MINPOP;
                    if (big) {
                        if (match && (lane & 31) == __ffs((int)mm) - 1) {          // the group's first lane
                            const uint32_t first = (uint32_t)p - (uint32_t)(lane & 31);
                            const uint32_t k = atomicAdd(&lrec_n, 1u);
                            if (k < LREC_CAP) {
                                lrec_first[k] = first;
                                lrec_mask[k] = mm;
                                lrec_key[k] = key;
                            } else {
                                slow(key, first, mm);
                            }
                        }
                    } else {
                        stray = stray || match;
                    }
                    pending = pending && !match;
                    const unsigned long long pend = __ballot(pending);
                    if (__popc((uint32_t)pend) < MINPOP && __popc((uint32_t)(pend >> 32)) < MINPOP) break;      // uniform
                }
                stray = stray || pending;
                // strays append themselves to the LDS list: one atomic per wave (the first stray lane reserves
                // for all of them)
                const unsigned long long sm = __ballot(stray);
                if (sm) {                                                      // uniform
                    const int lead = __ffsll((long long)sm) - 1;
                    uint32_t base = 0;
                    if (lane == lead) base = atomicAdd(&lst_n, (uint32_t)__popcll(sm));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
                    if (stray) {
                        const uint32_t k = base + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                        if (k < LST_CAP) {
                            lst_x[k] = x[u];
                            lst_y[k] = y[u];
                            lst_z[k] = z[u];
                            lst_p[k] = (uint32_t)p;
                            lst_key[k] = key;
                        } else {
                            slow(key, (uint32_t)p, 0u);                        // list full (scattered input)
                        }
                    }
                }
            }
        };
        FSTAMP(1);
#pragma unroll
        for (int trip = 0; trip < AGG_TRIPS; trip += 2) {
            fetch(p0 + (size_t)(trip + 1) * per_trip, xb, yb, zb);
            process(trip, xa, ya, za);
            fetch(trip + 2 < AGG_TRIPS ? p0 + (size_t)(trip + 2) * per_trip : p_begin + (interval + gridDim.x) * per_interval, xa, ya, za);
            process(trip + 1, xb, yb, zb);
        }
        FSTAMP(2);
        __syncthreads();
        FSTAMP(3);
        // phase 2: slots
        const uint32_t nr = min(lrec_n, LREC_CAP), ns = min(lst_n, LST_CAP);
        for (uint32_t k = tid; k < nr; k += 256) {
            const uint32_t key = lrec_key[k];
            const int slot = slot_of(key);
            if (slot < 0) { slow(key, lrec_first[k], lrec_mask[k]); lrec_key[k] = DST_NONE; }
            else lrec_key[k] = ((uint32_t)slot << 16) | atomicAdd(&srun[slot], 1u);
        }
        for (uint32_t k = tid; k < ns; k += 256) {
            const uint32_t key = lst_key[k];
            const int slot = slot_of(key);
            if (slot < 0) { slow(key, lst_p[k], 0u); lst_key[k] = DST_NONE; }
            else lst_key[k] = ((uint32_t)slot << 16) | atomicAdd(&sstray[slot], 1u);
        }
        FSTAMP(4);
        __syncthreads();
        FSTAMP(5);
        // phase 3: reserve
        if (skey[tid] != SLOT_EMPTY) {
            const uint32_t t = skey[tid], cr = srun[tid], cs = sstray[tid];
            const unsigned long long old = atomicAdd(&fill64[t], ((unsigned long long)cr << 32) | cs);
            const uint32_t br = min((uint32_t)(old >> 32), rcap), bs = min((uint32_t)old, scap);
            sdst_run[tid] = (unsigned long long)t * rcap + br;
            sroom_run[tid] = rcap - br;
            sdst_stray[tid] = (unsigned long long)t * scap + bs;
            sroom_stray[tid] = scap - bs;
        }
        FSTAMP(6);
        __syncthreads();
        FSTAMP(7);
        // phase 4: scatter
        for (uint32_t k = tid; k < nr; k += 256) {
            const uint32_t d = lrec_key[k];
            if (d == DST_NONE) continue;
            const uint32_t slot = d >> 16, at = d & 0xffffu;
            if (at < sroom_run[slot]) recs[sdst_run[slot] + at] = GroupRec{lrec_first[k], lrec_mask[k]};
            else for (uint32_t m = lrec_mask[k]; m; m &= m - 1) ovf[atomicAdd(ovf_count, 1ull)] = lrec_first[k] + (uint32_t)__ffs((int)m) - 1u;
        }
        for (uint32_t k = tid; k < ns; k += 256) {
            const uint32_t d = lst_key[k];
            if (d == DST_NONE) continue;
            const uint32_t slot = d >> 16, at = d & 0xffffu;
            if (at < sroom_stray[slot]) store_stray(strays, (size_t)(sdst_stray[slot] + at), lst_x[k], lst_y[k], lst_z[k],
                                                    mass ? mass[lst_p[k]] : (T)1, mass != nullptr);
            else ovf[atomicAdd(ovf_count, 1ull)] = lst_p[k];
        }
        FSTAMP(8);
        __syncthreads();                    // everyone is done with the lists and the table before they are re-armed
        skey[tid] = SLOT_EMPTY;
        srun[tid] = 0;
        sstray[tid] = 0;
        if (tid == 0) { lst_n = 0; lrec_n = 0; }
        __syncthreads();
        FSTAMP(9);
    }
    FSTAMP_END;
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

// ------------------------------------------------------------------------------------
// AST_PAINT_SCATTERED: particles WITHOUT spatial order in memory.  The grouping kernel above degenerates there
// (every particle is a stray of a tile no neighbour shares: one returning global atomic and one 16-byte scattered
// store per particle, 74 ms at 1024^3).  Two radix levels instead, every workgroup handling 4096 records so that a
// destination receives a run of records (4 at the first level, 8 at the second, at 1024^3) for ONE reservation atomic:
//   level A   particle -> 16-byte record {x, y, z, m} in its coarse bucket's (nb buckets of tpb consecutive tiles)
//             fixed-capacity segment of a staging array (no counting pass, see scatter_level_a_kernel);
//   level B   bucket by bucket, record -> the stray segment of its tile (the format the column walk reads), slots
//             reserved per (workgroup, tile) from the tile's fill64 counter.
// The walk then finds only stray copies (no group records) and reads them contiguously.  A record that does not fit
// its tile's segment (strongly clustered input) is deposited on the spot with global atomics.
// Workgroup shape per level (threads, records per thread).  A workgroup's phases - load, rank, reserve, scan, staged store -
// run one after the other between barriers, so what is idle in one phase can only be used by ANOTHER workgroup of the CU:
// both levels run 512 x 8 with two workgroups per CU (<= 80 KB of LDS each).  Measured and without effect (the loads are
// not what the phases wait for): prefetching the next chunk's records during the staged store, in either level; three
// level-A workgroups per CU (80 VGPRs, 8 spilled).
// Level B used to run 1024 x 8, one workgroup per CU, because halving its runs into the tile segments cost more than the
// overlap gained (8.9 -> 10.7 ms at 1024^3).  That was a symptom: the workgroups of one bucket were spread over the whole
// launch (grid (bucket, 32)), so the lines a run left partly written met their other pieces long after they had left L2.
// With a bucket's workgroups running at the SAME time on ONE XCD (round 4) those pieces meet in L2: 9.2 -> 6.4 ms as it
// was, 5.7 ms with 512 x 8.
#ifndef SCA_NT
#define SCA_NT 512
#endif
#ifndef SCB_NT
#define SCB_NT 512
#endif
constexpr int SC_PER_THREAD = 8;                                 // (16 per thread spills at the 128 VGPRs either shape allows)
constexpr int SCA_THREADS = SCA_NT, SCB_THREADS = SCB_NT;
// Most buckets (level A's LDS tables) and most tiles per bucket (level B's; its scan takes 4 entries per thread).  The
// number of buckets is chosen per call (scatter_buckets): about the square root of the number of tiles, so that both
// levels write runs of similar length.
constexpr uint32_t SC_BUCKETS_MAX = 1024, SC_TPB_MAX = 4 * SCB_NT;
// Every bucket's range of the staging array is cut into SC_GROUPS sub-ranges, and chunk c writes into sub-range
// c mod 8.  Blocks b and b + 8 share an XCD (observed placement; nothing here depends on it for correctness: the
// sub-ranges are fixed segments, one per (bucket, label)), so the 96-byte runs that complete a 128-byte line come
// from workgroups behind ONE L2 and the line leaves it whole.  With one cursor per bucket the pieces of a line sat in
// different XCDs' L2s and went to HBM as partial writes: level A's stores ran at 2 TB/s.
constexpr uint32_t SC_GROUPS = 8;
// Level B: workgroups per bucket.  They take consecutive slots of ONE XCD (see the kernel), 2 per CU x 32 CUs.
#ifndef SCB_WGS
#define SCB_WGS 64
#endif
static_assert(SCB_WGS % SC_GROUPS == 0 && SCB_WGS >= SC_GROUPS, "level B: every (bucket, label) segment gets SCB_WGS / SC_GROUPS workgroups");

// records staged per round: four (float) or two (double) per thread - 96 KB of LDS per 1024 threads
template <typename T, int NT> constexpr uint32_t sc_round() { return (sizeof(T) == 4 ? 4u : 2u) * (uint32_t)NT; }

// Exclusive scan of cnt[0 .. table) into lstart[] by the whole workgroup (NT threads, table <= 4 * NT: thread t takes
// the entries [K t, K t + K), K = 2 or 4); returns the total.  wsum: 16 words of LDS.
template <int NT>
__device__ inline uint32_t block_exclusive_scan(const uint32_t* cnt, uint32_t* lstart, uint32_t table, uint32_t* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K = table > 2u * NT ? 4u : 2u, i0 = K * tid;
    uint32_t v[4], own = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) { v[k] = (k < K && i0 + k < table) ? cnt[i0 + k] : 0u; own += v[k]; }
    uint32_t inc = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int k = 0; k < NT / 64; ++k) { if (k < wave) woff += wsum[k]; total += wsum[k]; }
    uint32_t ex = woff + inc - own;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) { if (k < K && i0 + k < table) lstart[i0 + k] = ex; ex += v[k]; }
    __syncthreads();
    return total;
}

// The records of one workgroup leave through LDS in DESTINATION order, a quarter at a time: consecutive lanes then
// store consecutive 16-byte records of the same run, so a run reaches L2 as whole lines in one instruction.  (Stored
// straight from the registers, each record is a lone 16-byte piece of a line that other workgroups' pieces reach much
// later: with ~500 workgroups x 1024 open runs the lines leave L2 partly filled - 0.75 TB/s.)
// where[u] = table entry << 16 | rank of record u among the workgroup's records of that entry (0xffffffff: none);
// lstart[]: the entries' first slots in the workgroup's destination order, base[]: their first index in `out`,
// room[] (or null): how many records of an entry `out` still takes - the rest is not stored (the caller deposits them).
// SW: words per record - 4 {x, y, z, m}, or 3 {x, y, z} when there are no masses (a quarter less staging traffic).
template <typename T, int SW, int NT>
__device__ inline void staged_store(const T (&rx)[SC_PER_THREAD], const T (&ry)[SC_PER_THREAD], const T (&rz)[SC_PER_THREAD],
                                    const T (&rm)[SC_PER_THREAD], const uint32_t (&where)[SC_PER_THREAD],
                                    const uint32_t* lstart, const unsigned long long* base, const uint32_t* room,
                                    uint32_t total, T* __restrict__ out, T* stage /* [SC_ROUND * 4] */,
                                    unsigned long long* sidx /* [SC_ROUND] */) {
    typedef T vec4_t __attribute__((ext_vector_type(4)));
    struct Rec3 { T x, y, z; };
    constexpr uint32_t ROUND = sc_round<T, NT>();
    for (uint32_t q0 = 0; q0 < total; q0 += ROUND) {
#pragma unroll
        for (int u = 0; u < SC_PER_THREAD; ++u) {
            if (where[u] == 0xffffffffu) continue;
            const uint32_t e = where[u] >> 16, r = where[u] & 0xffffu;
            const uint32_t sl = lstart[e] + r - q0;           // unsigned: slots of other rounds fail the test
            if (sl < ROUND) {
                if (SW == 4) reinterpret_cast<vec4_t*>(stage)[sl] = vec4_t{rx[u], ry[u], rz[u], rm[u]};
                else reinterpret_cast<Rec3*>(stage)[sl] = Rec3{rx[u], ry[u], rz[u]};
                sidx[sl] = (room == nullptr || r < room[e]) ? base[e] + r : ~0ull;
            }
        }
        __syncthreads();
        const uint32_t here = min(ROUND, total - q0);
        for (uint32_t i = threadIdx.x; i < here; i += NT) {
            const unsigned long long d = sidx[i];
            if (d == ~0ull) continue;
            if (SW == 4) *reinterpret_cast<vec4_t*>(out + 4 * (size_t)d) = reinterpret_cast<const vec4_t*>(stage)[i];
            else *reinterpret_cast<Rec3*>(out + 3 * (size_t)d) = reinterpret_cast<const Rec3*>(stage)[i];
        }
        __syncthreads();
    }
}

// No counting pass: every (bucket, group) owns a FIXED segment of cap_bg records of the staging array (twice the mean: a
// read of all positions - 2.5 ms at 1024^3 - just to size the segments exactly was a tenth of the unordered paint).  A record
// that does not fit its segment (strongly clustered input) goes to the late list like one that does not fit its tile's
// stray segment in level B, and is deposited with global atomics; device.paint sees the count and repaints two-pass.
// Append to a list with ONE returning global atomic per wave: every lane with `want` gets a slot of its own.  (Clustered
// input sends millions of records to the late list; one atomic per record on the list's single counter serialised them:
// level B took 48 ms at 1024^3 where the uniform set takes 5.7.)  Call in convergent control flow.
__device__ inline unsigned long long wave_append(unsigned long long* __restrict__ counter, bool want) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0ull;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0ull;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
}

template <typename T, int W, bool PLAINX, int SW>
__global__ void __launch_bounds__(SCA_THREADS)
scatter_level_a_kernel(const T* __restrict__ pos, const T* __restrict__ mass, size_t np, TileGeom g, uint32_t tpb, uint32_t nb,
                       unsigned long long* __restrict__ cursor, T* __restrict__ staging, uint32_t cap_bg,
                       uint32_t* __restrict__ col_flags, T* __restrict__ late_list, unsigned long long late_cap,
                       unsigned long long* __restrict__ late, unsigned long long* dropped) {
    __shared__ uint32_t cnt[SC_BUCKETS_MAX], lstart[SC_BUCKETS_MAX], room[SC_BUCKETS_MAX], wsum[16];
    __shared__ unsigned long long base[SC_BUCKETS_MAX];
    extern __shared__ unsigned long long dyn[];          // stage: 4096 records, then their 4096 destinations
    T* stage = reinterpret_cast<T*>(dyn);
    constexpr int NT = SCA_THREADS, SC_CHUNK = NT * SC_PER_THREAD;
    unsigned long long* sidx = dyn + sc_round<T, NT>() * SW * sizeof(T) / sizeof(unsigned long long);
    typedef T vec4_t __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    for (uint32_t b = tid; b < nb; b += NT) cnt[b] = 0;
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * SC_CHUNK;
    T x[SC_PER_THREAD], y[SC_PER_THREAD], z[SC_PER_THREAD], m[SC_PER_THREAD];
    uint32_t where[SC_PER_THREAD];                   // bucket << 16 | rank inside the workgroup's run (< 16384)
    const T* __restrict__ bp = pos + 3 * p0;                       // uniform base, 32-bit lane offsets
    const uint32_t last = (uint32_t)min(np - 1 - p0, (size_t)(SC_CHUNK - 1));
#pragma unroll
    for (int u = 0; u < SC_PER_THREAD; ++u) {
        const uint32_t rel = min((uint32_t)(u * NT + tid), last);      // unconditional loads
        x[u] = bp[3 * rel];
        y[u] = bp[3 * rel + 1];
        z[u] = bp[3 * rel + 2];
        m[u] = SW == 4 && mass ? mass[p0 + rel] : (T)1;
    }
    unsigned long long ndrop = 0;
#pragma unroll
    for (int u = 0; u < SC_PER_THREAD; ++u) {
        where[u] = 0xffffffffu;
        if ((uint32_t)(u * NT + tid) <= last) {
            const uint32_t key = tile_of<T, W, PLAINX>(x[u], y[u], z[u], g, col_flags);
            if (key != 0xffffffffu) {
                const uint32_t b = key / tpb;
                where[u] = (b << 16) | atomicAdd(&cnt[b], 1u);
            } else {
                ++ndrop;                                  // base plane outside the buffer (slab buffers only)
            }
        }
    }
    __syncthreads();
    const uint32_t grp = blockIdx.x % SC_GROUPS;
    __shared__ int any_full;
    if (tid == 0) any_full = 0;
    __syncthreads();
    for (uint32_t b = tid; b < nb; b += NT) {
        if (!cnt[b]) continue;
        const unsigned long long old = atomicAdd(&cursor[grp * nb + b], (unsigned long long)cnt[b]);
        const uint32_t bs = (uint32_t)min(old, (unsigned long long)cap_bg);
        base[b] = ((unsigned long long)b * SC_GROUPS + grp) * cap_bg + bs;
        room[b] = cap_bg - bs;
        if (cnt[b] > cap_bg - bs) any_full = 1;
    }
    const uint32_t total = block_exclusive_scan<NT>(cnt, lstart, nb, wsum);          // (barriers inside: any_full is settled)
    const bool full = any_full != 0;                      // uniform; a full segment is rare (strongly clustered input)
    if (full) {
#pragma unroll
        for (int u = 0; u < SC_PER_THREAD; ++u) {         // the records that do not fit go to the late list
            const bool over = where[u] != 0xffffffffu && (where[u] & 0xffffu) >= room[where[u] >> 16];
            const unsigned long long k = wave_append(late, over);
            if (over) {
                if (k < late_cap) reinterpret_cast<vec4_t*>(late_list)[k] = vec4_t{x[u], y[u], z[u], m[u]};
                else ++ndrop;
            }
        }
    }
    staged_store<T, SW, NT>(x, y, z, m, where, lstart, base, full ? room : nullptr, total, staging, stage, sidx);
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

// one record deposited with global atomics (a full tile segment): overflow_deposit_kernel's body
template <typename T, int W>
__device__ __noinline__ void deposit_record_global(T x, T y, T z, T m, const TileGeom& g, double scale, T* __restrict__ grid,
                                                   unsigned long long* dropped, int x_lo = 0, int x_hi = 0x7fffffff) {
    constexpr int LO = Window<W>::LO;
    double fx, fy, fz;
    const int bx = ast::locate<W>(grid_coord(x, g), g.n, fx);
    const int by = ast::locate<W>(grid_coord(y, g), g.n, fy);
    const int bz = ast::locate<W>(grid_coord(z, g), g.n, fz);
    T wx[W], wy[W], wz[W];
    Window<W>::weights(fx, wx);
    Window<W>::weights(fy, wy);
    Window<W>::weights(fz, wz);
    const T mm = (T)((double)m * scale);
    for (int a = 0; a < W; ++a) {
        int px = ast::wrap1(bx - LO + a, g.n) - g.x_start;
        if (px < 0) px += g.n;
        if (px >= g.nx_alloc) { if (dropped && x_hi >= g.nx_alloc) atomicAdd(dropped, 1ull); continue; }
        if (px < x_lo || px >= x_hi) continue;
        for (int b = 0; b < W; ++b) {
            T* row = grid + ((size_t)px * g.n + ast::wrap1(by - LO + b, g.n)) * g.n;
            for (int c = 0; c < W; ++c) atomicAdd(row + ast::wrap1(bz - LO + c, g.n), mm * wx[a] * wy[b] * wz[c]);
        }
    }
}

// the late list of level B (records whose tile segment was full), deposited with global atomics
template <typename T, int W>
__global__ void __launch_bounds__(256)
late_deposit_kernel(const T* __restrict__ late_list, const unsigned long long* __restrict__ late, unsigned long long late_cap,
                    TileGeom g, double scale, T* __restrict__ grid, unsigned long long* dropped, int x_lo, int x_hi,
                    unsigned long long n_below = ~0ull) {
    const unsigned long long n = min(*late, late_cap);
    if (n >= n_below) return;                         // a long list goes through LDS tiles (run_tiled, scattered branch)
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        deposit_record_global<T, W>(late_list[4 * i], late_list[4 * i + 1], late_list[4 * i + 2], late_list[4 * i + 3], g, scale,
                                    grid, dropped, x_lo, x_hi);
}

// Level B's grid is one-dimensional: workgroup L runs on XCD L mod 8 (round-robin dispatch; observed placement, nothing
// depends on it for correctness), and the SCB_WGS workgroups of a bucket take consecutive slots of ONE XCD - so they run
// at the same time behind one L2, where the lines one workgroup's run leaves partly written are completed by the next
// run into the same tile segment before they are written back.  (nb is a multiple of 8.)
template <typename T, int W, bool PLAINX, int SW>
__global__ void __launch_bounds__(SCB_THREADS)
scatter_level_b_kernel(const T* __restrict__ staging, const unsigned long long* __restrict__ cursor, uint32_t cap_bg, TileGeom g, uint32_t tpb,
                       uint32_t nb, unsigned long long* __restrict__ fill64, T* __restrict__ strays, uint32_t scap,
                       T* __restrict__ late_list, unsigned long long late_cap, unsigned long long* __restrict__ late,
                       unsigned long long* dropped) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t late_n, any_full;
    __shared__ unsigned long long late_base;
    extern __shared__ unsigned long long dyn[];          // base[tpb] | stage | sidx | cnt[tpb] room[tpb] lstart[tpb]
    constexpr int NT = SCB_THREADS, SC_CHUNK = NT * SC_PER_THREAD;
    const uint32_t tp = (tpb + 1u) & ~1u;                // (the staged records stay 16-byte aligned)
    unsigned long long* base = dyn;
    T* stage = reinterpret_cast<T*>(dyn + tp);
    unsigned long long* sidx = dyn + tp + sc_round<T, NT>() * SW * sizeof(T) / sizeof(unsigned long long);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sidx + sc_round<T, NT>());
    uint32_t* room = cnt + tp;
    uint32_t* lstart = room + tp;
    const int tid = threadIdx.x;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t bucket = ((slot / SCB_WGS) << 3) | xcd, by = slot % SCB_WGS;
    // the bucket's SC_GROUPS segments of the staging array, SCB_WGS / SC_GROUPS workgroups each
    const uint32_t grp = by % SC_GROUPS, sub = by / SC_GROUPS, nsub = SCB_WGS / SC_GROUPS;
    const size_t b0 = ((size_t)bucket * SC_GROUPS + grp) * cap_bg;
    const size_t b1 = b0 + (size_t)min(cursor[grp * nb + bucket], (unsigned long long)cap_bg);
    typedef T vec4_t __attribute__((ext_vector_type(4)));
    struct Rec3 { T x, y, z; };
    for (size_t c0 = b0 + (size_t)sub * SC_CHUNK; c0 < b1; c0 += (size_t)nsub * SC_CHUNK) {
        for (uint32_t t = tid; t < tpb; t += NT) cnt[t] = 0;
        if (tid == 0) { late_n = 0u; any_full = 0u; }
        __syncthreads();
        T x[SC_PER_THREAD], y[SC_PER_THREAD], z[SC_PER_THREAD], m[SC_PER_THREAD];
        uint32_t where[SC_PER_THREAD];               // tile in bucket << 16 | rank (< 16384)
#pragma unroll
        for (int u = 0; u < SC_PER_THREAD; ++u) {
            const size_t ri = min(c0 + (size_t)u * NT + tid, b1 - 1);
            if (SW == 4) {
                const vec4_t r = reinterpret_cast<const vec4_t*>(staging)[ri];
                x[u] = r.x; y[u] = r.y; z[u] = r.z; m[u] = r.w;
            } else {
                const Rec3 r = reinterpret_cast<const Rec3*>(staging)[ri];
                x[u] = r.x; y[u] = r.y; z[u] = r.z; m[u] = (T)1;
            }
        }
#pragma unroll
        for (int u = 0; u < SC_PER_THREAD; ++u) {
            where[u] = 0xffffffffu;
            if (c0 + (size_t)u * NT + tid < b1) {
                const uint32_t lt = tile_of<T, W, PLAINX>(x[u], y[u], z[u], g, nullptr) - bucket * tpb;
                where[u] = (lt << 16) | atomicAdd(&cnt[lt], 1u);
            }
        }
        __syncthreads();
        for (uint32_t t = tid; t < tpb; t += NT) {
            if (!cnt[t]) continue;
            const uint32_t tile = bucket * tpb + t;
            const uint32_t old = (uint32_t)atomicAdd(&fill64[tile], (unsigned long long)cnt[t]);
            const uint32_t bs = min(old, scap);
            base[t] = (unsigned long long)tile * scap + bs;
            room[t] = scap - bs;
            if (cnt[t] > scap - bs) any_full = 1u;
        }
        const uint32_t total = block_exclusive_scan<NT>(cnt, lstart, tpb, wsum);         // (barriers inside: any_full is settled)
        // a full tile segment: the record goes to the late list.  Rare for a uniform set; a clustered one sends millions -
        // and the list has ONE counter, where same-address atomics are served one after the other (one per wave, as the
        // compiler already arranges, still left level B at 48 ms for the clustered 1024^3 set against 5.7 for the uniform
        // one): the workgroup counts its late records in LDS and reserves their slots with ONE global atomic per chunk
        if (any_full != 0u) {                          // (uniform over the workgroup; never taken for a uniform particle set)
            uint32_t lslot[SC_PER_THREAD];
#pragma unroll
            for (int u = 0; u < SC_PER_THREAD; ++u) {
                const bool over = where[u] != 0xffffffffu && (where[u] & 0xffffu) >= room[where[u] >> 16];
                lslot[u] = over ? atomicAdd(&late_n, 1u) : 0xffffffffu;
            }
            __syncthreads();
            if (tid == 0 && late_n) late_base = atomicAdd(late, (unsigned long long)late_n);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SC_PER_THREAD; ++u) {
                if (lslot[u] == 0xffffffffu) continue;
                const unsigned long long k = late_base + lslot[u];
                if (k < late_cap) reinterpret_cast<vec4_t*>(late_list)[k] = vec4_t{x[u], y[u], z[u], m[u]};
                else if (dropped) atomicAdd(dropped, 1ull);            // more than a quarter of all particles: reported, not lost silently
            }
        }
        staged_store<T, SW, NT>(x, y, z, m, where, lstart, base, room, total, strays, stage, sidx);
    }
}

// Particles that did not fit their tile's segment in the single-pass variant: plain
// global-atomic deposit (the direct kernel's inner loop over an index list).
template <typename T, int W>
__global__ void __launch_bounds__(256)
overflow_deposit_kernel(const T* __restrict__ pos, const T* __restrict__ mass, const uint32_t* __restrict__ ovf,
                        const unsigned long long* __restrict__ ovf_count, TileGeom g, double scale,
                        T* __restrict__ grid, unsigned long long* dropped, int x_lo, int x_hi) {
    // [x_lo, x_hi): the buffer planes this launch deposits into (the staged paint runs it tile row by tile row; a
    // deposit outside the buffer is counted by the launch that holds the buffer's last plane)
    constexpr int LO = Window<W>::LO;
    const unsigned long long n = *ovf_count;
    unsigned long long ndrop = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const size_t p = ovf[i];
        double fx, fy, fz;
        const int bx = ast::locate<W>(grid_coord(pos[3 * p + 0], g), g.n, fx);
        const int by = ast::locate<W>(grid_coord(pos[3 * p + 1], g), g.n, fy);
        const int bz = ast::locate<W>(grid_coord(pos[3 * p + 2], g), g.n, fz);
        T wx[W], wy[W], wz[W];
        Window<W>::weights(fx, wx);
        Window<W>::weights(fy, wy);
        Window<W>::weights(fz, wz);
        const T m = (T)((mass ? (double)mass[p] : 1.0) * scale);
#pragma unroll
        for (int a = 0; a < W; ++a) {
            int px = ast::wrap1(bx - LO + a, g.n) - g.x_start;
            if (px < 0) px += g.n;
            if (px >= g.nx_alloc) { ndrop += x_hi >= g.nx_alloc; continue; }
            if (px < x_lo || px >= x_hi) continue;
#pragma unroll
            for (int b = 0; b < W; ++b) {
                T* row = grid + ((size_t)px * g.n + ast::wrap1(by - LO + b, g.n)) * g.n;
#pragma unroll
                for (int c = 0; c < W; ++c) atomicAdd(row + ast::wrap1(bz - LO + c, g.n), m * wx[a] * wy[b] * wz[c]);
            }
        }
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

// ---- exclusive scan of tile_count (3 kernels, 1024 items per block) ----
__global__ void __launch_bounds__(256)
scan_blocks_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ block_sums, uint32_t n) {
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < wave; ++i) woff += wsum[i];
    uint32_t ex = woff + inc - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ void __launch_bounds__(256)
scan_sums_kernel(uint32_t* block_sums, uint32_t nblocks) {
    // single workgroup, serial over 256-wide strips (nblocks <= 32768)
    __shared__ uint32_t carry, wsum[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = carry;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        if (i < nblocks) block_sums[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = woff + inc;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
scan_add_kernel(uint32_t* out, const uint32_t* __restrict__ block_sums, uint32_t n) {
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) out[base + i] += add;
}

template <typename T, int W, int STRIDE = 3>
__global__ void __launch_bounds__(256)
tile_deposit_kernel(const T* __restrict__ pos, const T* __restrict__ mass, TileGeom g, double scale,
                    const uint32_t* __restrict__ index, const uint32_t* __restrict__ tile_off,
                    const uint32_t* __restrict__ tile_count, uint32_t cap, T* __restrict__ grid,
                    unsigned long long* dropped, uint32_t ntiles_loop = 0) {
    constexpr int LX = TX + W - 1, LY = TY + W - 1, LZ = TZ + W - 1;
    constexpr int LO = Window<W>::LO;
    // ds_add_f32 retires ~0.33 lanes/clk/CU on gfx950 against ~7 for ds_add_f64
    // (scripts/micro/lds_atomics.hip), so the LDS tile accumulates in double for
    // both grid dtypes and is rounded to T once, at the flush.
    __shared__ double tile[LX * LY * LZ];
    // ntiles_loop = 0: tile = workgroup.  Otherwise the workgroups stride over the ntiles_loop tiles (the late list's
    // deposit: nearly all tiles are empty there, and half a million workgroups that only find that out cost 0.1 ms)
    const uint32_t t_end = ntiles_loop ? ntiles_loop : blockIdx.x + 1u;
  for (uint32_t t = blockIdx.x; t < t_end; t += gridDim.x) {
    // cap != 0: single-pass layout (fixed segments, tile_count holds the slots REQUESTED)
    const uint32_t cnt = cap ? min(tile_count[t], cap) : tile_count[t];
    if (cnt == 0) continue;                     // uniform for the workgroup
    const size_t off = cap ? (size_t)t * cap : (size_t)tile_off[t];
    __syncthreads();                            // (the previous tile's flush has read the LDS tile)
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) tile[i] = 0.0;
    __syncthreads();

    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int ox = tx * TX, oy = ty * TY, oz = tz * TZ;   // owned origin (buffer plane / global y, z)
#ifndef DEP_U
#define DEP_U 4
#endif
    constexpr int U = DEP_U;
    for (uint32_t i0 = 0; i0 < cnt; i0 += 256 * U) {
        T px[U], py[U], pz[U], pm[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * 256 + threadIdx.x;
            on[u] = i < cnt;
            const size_t p = on[u] ? index[off + i] : 0;
            px[u] = on[u] ? pos[STRIDE * p + 0] : (T)0;
            py[u] = on[u] ? pos[STRIDE * p + 1] : (T)0;
            pz[u] = on[u] ? pos[STRIDE * p + 2] : (T)0;
            pm[u] = STRIDE == 4 ? (on[u] ? pos[STRIDE * p + 3] : (T)1) : (on[u] && mass) ? mass[p] : (T)1;       // (records carry their mass)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            double fx, fy, fz;
            int bx = ast::locate<W>(grid_coord(px[u], g), g.n, fx) - g.x_start;
            if (bx < 0) bx += g.n;
            const int lx = bx - ox;                               // 0..TX-1 by construction of the index
            const int ly = ast::locate<W>(grid_coord(py[u], g), g.n, fy) - oy;
            const int lz = ast::locate<W>(grid_coord(pz[u], g), g.n, fz) - oz;
            T wx[W], wy[W], wz[W];
            Window<W>::weights(fx, wx);
            Window<W>::weights(fy, wy);
            Window<W>::weights(fz, wz);
            const T m = (T)((double)pm[u] * scale);
#pragma unroll
            for (int a = 0; a < W; ++a) {
                const T ma = m * wx[a];
#pragma unroll
                for (int b = 0; b < W; ++b) {
                    const T mab = ma * wy[b];
                    double* row = &tile[((lx + a) * LY + (ly + b)) * LZ + lz];
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        if (ablate & 2) { T v = mab * wz[c]; asm volatile("" ::"v"(v), "v"(row)); }
                        else atomicAdd(row + c, (double)(mab * wz[c]));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (ablate & 1) continue;

    // flush: LDS cell (a, b, c) is buffer plane ox + a - LO, global (oy + b - LO, oz + c - LO)
    unsigned long long ndrop = 0;
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) {
        const T v = (T)tile[i];
        if (v == (T)0) continue;
        const int c = i % LZ, b = (i / LZ) % LY, a = i / (LZ * LY);
        int px = ox + a - LO;
        if (g.nx_alloc == g.n) px = ast::wrap1(px, g.n);
        else if (px < 0 || px >= g.nx_alloc) { ++ndrop; continue; }
        const int gy = ast::wrap1(oy + b - LO, g.n), gz = ast::wrap1(oz + c - LO, g.n);
        atomicAdd(&grid[((size_t)px * g.n + gy) * g.n + gz], v);
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
  }
}

// ------------------------------------------------------------------------------------
// OVERWRITE flush without global atomics: the column walk.
//
// One workgroup owns a whole z-column of tiles (fixed tx, ty) and walks it with the LDS
// tile kept resident: after a tile's particles are in, the TZ lowest planes are final and
// leave as PLAIN stores — owned (x, y) cells to the grid, the x/y halo ring to the column's
// halo record — and the top W-1 planes (the z halo) are carried down to become the bottom
// planes of the next tile, so z halos never leave the CU.  A second kernel lets every
// column fold its neighbours' halo records into its own border rows (plain read-modify-
// write, contiguous in z).  Every grid cell is written, so no zero-fill is needed, and the
// 1.3 TB/s global float-atomic flush (4 ms of the 13 ms tile kernel at 1024^3) is gone.
// The few planes that wrap around the periodic z edge are added atomically at the end.
// The particle lists a column walk reads.  FMT 0: 4-byte particle ids per tile (exact offsets of the two-pass
// variant, tile_off / tile_count).  FMT 2: as FMT 1 with 3-word stray copies {x, y, z} (paints without masses).
// FMT 1: the compact lists of tile_group_kernel (group records + stray copies
// in fixed-capacity segments, counts in fill64: records in the high word, strays in the low).
// (The list pointers are separate __restrict__ kernel arguments on purpose: read through a struct member the
// per-tile counts become VECTOR loads followed by s_waitcnt vmcnt(0) - which also waits for every prefetched
// position - instead of scalar loads.)
// quantum of the fixed-point tiles (see column_deposit_kernel): 2 * cmax terms of at most mass_bound * |scale| each
// stay below 2^SUM_BITS, every term below 2^50
template <bool RAW>
__device__ inline double paint_inv_quantum(uint32_t cmax, double mass_bound, double scale) {
    constexpr int SUM_BITS = RAW ? 47 : 62;
    const int bits = 33 - __clz((int)min(cmax, 0x3fffffffu));            // 2 * cmax < 2^bits
    const double vmax = mass_bound * fabs(scale) > 0.0 ? mass_bound * fabs(scale) : 1.0;
    return exp2((double)min(50, SUM_BITS - bits)) / vmax;
}

// where the z line of LDS column (a, b) of tile column `col` goes (see column_deposit_kernel)
template <typename T, int W>
__device__ inline unsigned long long column_line_dest(int a, int b, int col, int ox, int oy, const TileGeom& g, T* grid, T* rec) {
    constexpr int LO = Window<W>::LO;
    using RM = RingMap<W>;
    const int px = ox + a - LO;
    if (RM::owned(a, TX) && RM::owned(b, TY))
        return px < g.nx_alloc ? ((unsigned long long)(grid + ((size_t)px * g.n + oy + b - LO) * g.n) |
                                  (px >= g.off_lo && px < g.off_hi ? 2ull : 0ull)) : 0ull;
    unsigned long long d = (unsigned long long)(rec + ((size_t)col * RM::COUNT + RM::cell(a, b)) * ast::rec_pitch((size_t)g.n));
    if (g.nx_alloc != g.n && (px < 0 || px >= g.nx_alloc)) d |= 1ull;
    return d;
}

// a fixed-point cell sum -> grid value: (sum * q - sub) rounded once.  RAW (fp32 grids): the low 48 bits hold the sum + BIAS
template <typename T>
__device__ inline T fixed_to_value(unsigned long long raw, double q, double sub, bool& any) {
    if (sizeof(T) == 4) {
        const unsigned long long low = raw & 0x0000ffffffffffffull;
        any = low != (1ull << 47);
        return (T)((__longlong_as_double((long long)(low | 0x4330000000000000ull)) - 4644337115725824.0) * q - sub);      // 2^52 + 2^47
    }
    any = raw != 0ull;
    return (T)((double)(long long)raw * q - sub);
}

struct WalkCaps {
    uint32_t rcap, scap, cap;
    size_t np;
};

template <typename T, int W, bool HAS_MASS, int FMT>
__global__ void __launch_bounds__(256)
column_deposit_kernel(const T* __restrict__ pos, const T* __restrict__ mass, TileGeom g, double scale,
                      const uint32_t* __restrict__ wl_index, const uint32_t* __restrict__ wl_tile_off,
                      const uint32_t* __restrict__ wl_tile_count, const unsigned long long* __restrict__ wl_fill64,
                      const GroupRec* __restrict__ wl_recs, const T* __restrict__ wl_strays, WalkCaps wl, double mass_bound,
                      const uint32_t* __restrict__ col_flags, T* __restrict__ grid, T* __restrict__ rec,
                      double offset, unsigned long long* dropped, int col0, int nseg, unsigned long long* __restrict__ zrec) {
    constexpr int LX = TX + W - 1, LY = TY + W - 1, LZ = TZ + W - 1;
    constexpr int LO = Window<W>::LO, H = W - 1;
    // The LDS tile accumulates in 64-bit FIXED POINT: ds_add_u64 retires ~1.9x the lanes per
    // clock of ds_add_f64 on scattered addresses (scripts/micro/lds_atomics.hip), integer sums
    // do not depend on arrival order (the whole paint is bit-reproducible), and the quantum
    // chosen per column below sits far under the rounding of the output type.
    //
    // double -> integer is one add with M = 1.5 * 2^52: bits(x + M) = 0x4338 << 48 + round(x).
    //  * T = double: the 0x4338 is stripped per term and sums may use 62 bits.
    //  * T = float: the raw bits are added.  K terms leave K * (0x4338 << 48) + sum in the cell,
    //    so the low 48 bits are the sum mod 2^48; cells start at BIAS = 2^47, which keeps those
    //    48 bits non-negative, and the flush turns them into a double by or-ing them under the
    //    exponent of 2^52 (one v_and_or_b32) and subtracting 2^52 + 2^47.
    constexpr bool RAW = sizeof(T) == 4;
    constexpr unsigned long long BIAS = RAW ? (1ull << 47) : 0ull;
    // The z planes of the tile form a RING of LZ slots: local plane c of tile tz lives in slot
    // (c + tz * TZ) mod LZ, so the W-1 halo planes carried from one tile to the next stay where
    // they are and the flush only has to re-arm the TZ slots it stored.
    // Row pitch PZ >= LZ cells (WALK_PZ, default LZ).  The walk is bound by its LDS atomics (rocprofv3: LDS 77 % busy, 59 %
    // of its cycles bank conflicts: the 32 z-consecutive particles of a half wave fill all 64 banks, so every particle
    // the jitter moved to a neighbouring (x, y) row collides with one that stayed).  Measured and NOT adopted: PZ = 48 -
    // rows half of the banks apart - runs the bare atomics 1.29x faster (scripts/micro/lds_tile_atomics.hip: 335 -> 433 G
    // particles/s; 32-bit cells 543, a 32-bit lo plane with returning adds + carry into a hi plane 377), but the 31 KB
    // tile leaves 5 workgroups per CU instead of 7 and the walk gets slower: CIC 4.62 -> 4.71 ms (TSC 13.1 -> 12.3).
#ifndef WALK_PZ
#define WALK_PZ 0
#endif
    // WALK_ODD_PITCH (measured, not adopted): row pitches odd, in cells, for both neighbour directions (y: PZ, x: LYP * PZ) so
    // that, with the lanes of a half wave split even / odd (load_pos), a particle moved by one row lands on the banks of a
    // cell of the other parity.  CIC's 33 and 9 * 33 are odd as they are; TSC (34, 10 rows) padded to 35 and 11 rows costs
    // a workgroup of occupancy (33 KB) and gains nothing: 11.5 -> 11.6 ms.
#ifndef WALK_ODD_PITCH
#define WALK_ODD_PITCH 0
#endif
    constexpr int PZ = WALK_PZ > LZ ? WALK_PZ : (WALK_ODD_PITCH ? (LZ | 1) : LZ);
    constexpr int LYP = WALK_ODD_PITCH ? (LY | 1) : LY;      // rows per x-plane of the tile (>= LY)
    __shared__ unsigned long long tile[LX * LYP * PZ];       // ((a * LYP + b) * PZ + slot), slot fastest
    auto trow = [](int ab) { return LYP == LY ? ab : (ab / LY) * LYP + ab % LY; };      // logical (a, b) pair -> row of the LDS tile
    // the launch walks the columns col0 .. col0 + gridDim.x - 1 (mod the column count: the x-sorted pipeline's last
    // launch wraps around to the tile rows it held back)
    // nseg > 1: the column is cut into nseg z-segments of ntz / nseg tiles, one workgroup each (more, shorter
    // workgroups: thin launches of the x-sorted pipeline and thin slab buffers still fill the chip).  Where two
    // segments meet, the lower one's top halo planes and the upper one's first planes leave as EXACT sums
    // (zrec) and z_seam_kernel writes their one correctly rounded total.
    const int seg = nseg > 1 ? (int)(blockIdx.x % (unsigned)nseg) : 0;
    int col = col0 + (int)(nseg > 1 ? blockIdx.x / (unsigned)nseg : blockIdx.x);
    col = col >= g.ntx * g.nty ? col - g.ntx * g.nty : col;
    const int nsteps = g.ntz / nseg;                            // tiles this workgroup walks
    DSTAMP_DECL;
    const int ty = col % g.nty, tx = col / g.nty;
    const int ox = tx * TX, oy = ty * TY;
    const bool x_periodic = g.nx_alloc == g.n;
    unsigned long long ndrop = 0;
    for (int i = threadIdx.x; i < LX * LYP * PZ; i += 256) tile[i] = BIAS;
    // where the z line of LDS column (a, b) goes: the owned cells' grid line (bit 1 set: `offset`
    // is subtracted there, in double, before the one rounding to T) or the halo ring's record line
    // (bit 0: a halo that points outside a slab buffer, counted as dropped when non-zero);
    // 0 = an owned plane the buffer does not hold (last x tile of a slab buffer with
    // nx_alloc % TX != 0): deposits there are counted as dropped
    __shared__ unsigned long long dest[LX * LY];
    // exact sums of the first H planes the walk flushes: the periodic wrap at the end of the walk lands on them
    __shared__ unsigned long long first_planes[LX * LY * H];
    for (int ab = threadIdx.x; ab < LX * LY; ab += 256)
        dest[ab] = column_line_dest<T, W>(ab / LY, ab % LY, col, ox, oy, g, grid, rec);
    // quantum: a cell collects at most the particles of two consecutive tiles, each contribution
    // is <= mass_bound * |scale|; keep every sum below 2^SUM_BITS and every term below 2^50
    // (FMT 1: the bound is the segments' capacity, the same for every column and independent of how the
    // particles are ordered in memory - the painted grid must not depend on that order)
    uint32_t cmax = 1;
    if (FMT == 0) {
        for (int tz = 0; tz < g.ntz; ++tz) cmax = max(cmax, wl_tile_count[(uint32_t)((tx * g.nty + ty) * g.ntz + tz)]);
    } else {
        cmax = (uint32_t)min(5ull * wl.cap, 0x3fffffffull);      // 32 * rcap + the largest scap = 5 * cap
    }
    const double invq = paint_inv_quantum<RAW>(cmax, mass_bound, scale);
    const double q = 1.0 / invq;
    // tile-relative cell lookup (paint_window.h locate_rel); x also carries the slab offset
    int orx = g.x_start + ox;
    if (orx >= g.n) orx -= g.n;
    const unsigned ex = (unsigned)min(TX, g.n), ey = (unsigned)min(TY, g.n), ez = (unsigned)min(TZ, g.n);
    __syncthreads();

    // The walk is software-pipelined over batches of 256*U particles, across tile boundaries:
    // while batch k is deposited, the positions of batch k+1 and the indices of batch k+2 are in
    // flight, so the gather latency (and the flush between two tiles) is covered by work of the
    // same wave.  All loads are unconditional (lanes past the end of a batch re-load its last
    // particle): a predicated load costs a branch and a full s_waitcnt each.
    // slots per thread and batch.  A tile's last batch is partly empty (72 entries per tile of the bench input):
    // 2 wastes less than 4 (deposit 4.49 vs 4.72 ms at 1024^3), 3 is in between
#ifndef COL_U
#define COL_U 2
#endif
    constexpr int U = COL_U;
    // A tile's work is a list of ENTRIES.  FMT 0: particle ids, one per thread slot.  FMT 1: 32-lane entries, one
    // per half wave - the tile's nrec group records followed by ceil(nst / 32) blocks of 32 consecutive stray
    // copies, so records and strays share batches (72 entries, i.e. 4.5 batches, per tile of the bench input).
    // off: FMT 0 first id of the tile's list; FMT 1 the tile id.  tz == nsteps: past the end
    using list_off_t = std::conditional_t<FMT != 0, uint32_t, size_t>;
    struct Batch { int tz; uint32_t i0, cnt; list_off_t off; uint32_t nrec, nst; };
    // The walk starts at a column-dependent tile and wraps around the periodic z edge (the ring
    // does not care), so concurrently running columns are at different z: in lockstep all of
    // them would store to / gather from addresses a large power of two apart.
    int tz0 = (ablate & 512 ? 0 : (int)(((unsigned)col * 2654435761u >> 16) % (unsigned)g.ntz)) + seg * nsteps;
    tz0 = tz0 >= g.ntz ? tz0 - g.ntz : tz0;                     // first tile of this workgroup's walk
    auto phys = [&](int step) { const int t = tz0 + step; return t >= g.ntz ? t - g.ntz : t; };   // step -> tile
    constexpr uint32_t BATCH = FMT != 0 ? 8u * U : 256u * U;      // entries per batch
    auto next_batch = [&](Batch bt) -> Batch {
        if (bt.tz >= 0 && bt.tz < nsteps && bt.i0 + BATCH < bt.cnt) { bt.i0 += BATCH; return bt; }
        for (int nt = bt.tz + 1; nt < nsteps; ++nt) {
            const uint32_t t = (uint32_t)((tx * g.nty + ty) * g.ntz + phys(nt));
            if (FMT == 0) {
                const uint32_t cnt = wl_tile_count[t];
                if (cnt) return Batch{nt, 0u, cnt, (list_off_t)wl_tile_off[t], 0u, 0u};
            } else {
                const unsigned long long f = wl_fill64[t];
                const uint32_t nrec = min((uint32_t)(f >> 32), wl.rcap), nst = min((uint32_t)f, wl.scap);
                if (nrec + nst) return Batch{nt, 0u, nrec + (nst + 31u) / 32u, (list_off_t)t, nrec, nst};
            }
        }
        return Batch{nsteps, 0u, bt.cnt, bt.off, bt.nrec, bt.nst};           // keeps a loadable span
    };
    // stage 1 of a batch: LOAD ONLY - the id (FMT 0) or the group record (FMT 1) of each of this thread's U
    // slots.  Nothing may be computed from the loaded values here: that would wait for them on the spot, and with
    // them for everything else in flight.  (A stray block re-reads the tile's last record, or the segment's first
    // slot when the tile has none: unconditional loads.)
#ifndef WALK_LANE_PERM
#define WALK_LANE_PERM 1
#endif
    // which of the batch's 32-particle entries (0 .. 7 per slot u) and which of its particles a lane takes:
    //   0  half wave h = entry, lane = particle
    //   1  half wave h = entry, lanes 0-15 the even particles, 16-31 the odd ones
    //   2  the wave's two entries interleaved over its four 16-lane groups: group g holds the particles = g (mod 4) of both
    const uint32_t lane6 = threadIdx.x & 63u, wave2 = (threadIdx.x >> 6) << 1;
    const uint32_t my_entry = WALK_LANE_PERM == 2 ? wave2 + ((lane6 >> 3) & 1u) : (threadIdx.x >> 5);
    const uint32_t my_part = WALK_LANE_PERM == 2 ? 4u * (lane6 & 7u) + (lane6 >> 4)
                           : WALK_LANE_PERM == 1 ? (((lane6 & 15u) << 1) | ((lane6 >> 4) & 1u)) : (lane6 & 31u);
    auto load_idx = [&](const Batch& bt, GroupRec (&rec)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (FMT == 0) {
                const uint32_t i = bt.i0 + u * 256 + threadIdx.x;
                rec[u].first = wl_index[bt.off + min(i, bt.cnt - 1)];
            } else {
                const uint32_t r = bt.i0 + u * 8 + my_entry;
                const uint32_t rr = min(min(r, bt.cnt - 1), bt.nrec ? bt.nrec - 1 : 0u);
                rec[u] = wl_recs[(size_t)bt.off * wl.rcap + rr];
            }
        }
    };
    // stage 2, one batch later: ids and occupancy from the records (bit u of act: slot u holds a particle), then
    // positions and masses - a gather through the ids, or the stray copies themselves; one unconditional load per
    // component through a per-lane base pointer
    auto load_pos = [&](const Batch& bt, const GroupRec (&rec)[U], uint32_t& act, T (&p)[3 * U], T (&m)[U]) {
        act = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (FMT != 0) {
                constexpr int SW = FMT == 2 ? 3 : 4;                            // stray copies: {x, y, z, m} or {x, y, z}
                // lane -> particle of the 32-particle entry: lanes 0-15 take the even particles, 16-31 the odd ones.  In
                // z-ordered input the particles of a window sit in consecutive cells; a particle that the jitter moved
                // to a neighbouring (x, y) row lands one cell (y) or nine (x) further in the banks - on the banks of
                // an ODD neighbour, which is now in the other half of the lanes (WALK_LANE_PERM 0: identity)
                const uint32_t r = bt.i0 + u * 8 + my_entry, b = my_part;
                const bool st = r >= bt.nrec;                                   // uniform per half wave
                const uint32_t sidx = (r - bt.nrec) * 32u + b;
                const bool on = st ? (r < bt.cnt && sidx < bt.nst) : ((rec[u].mask >> b) & 1u) != 0;
                act |= (uint32_t)on << u;
                const uint32_t idx = st ? min(sidx, bt.nst - 1) : min(rec[u].first + b, (uint32_t)(wl.np - 1));    // np < 2^32 - 64: no wrap
                const T* const sbase = wl_strays + SW * ((size_t)bt.off * wl.scap);
                const T* const src = st ? sbase + SW * (size_t)idx : pos + 3 * (size_t)idx;
                p[3 * u + 0] = src[0];
                p[3 * u + 1] = src[1];
                p[3 * u + 2] = src[2];
                if (HAS_MASS) { const T* const msrc = st ? src + 3 : mass + idx; m[u] = *msrc; }
                else m[u] = (T)1;
                continue;
            }
            act |= (uint32_t)(bt.i0 + u * 256 + threadIdx.x < bt.cnt) << u;
            const uint32_t idx = rec[u].first;
            const size_t q3 = (ablate & 32) ? (size_t)((idx % 1000000u) * 3) : (size_t)idx * 3;
            if (ablate & 1024) { p[3 * u + 0] = p[3 * u + 1] = p[3 * u + 2] = (T)idx; continue; }
            p[3 * u + 0] = pos[q3 + 0];
            p[3 * u + 1] = pos[q3 + 1];
            p[3 * u + 2] = pos[q3 + 2];
            m[u] = HAS_MASS ? mass[idx] : (T)1;         // compile time: a run-time choice would merge the
        }                                               // loaded value with a constant and wait for it
    };

    Batch nxt = next_batch(Batch{-1, 0u, 0u, 0, 0u, 0u});       // becomes batch k + 1 below
    int cur_tz = nxt.tz;                                        // tile of batch k, the one being deposited
    int sh = 0;                             // (tz * TZ) mod LZ
    int oz = 0;
    // deposit one batch: positions pc / masses mc of batch `cur`.  CAREFUL = false: every particle
    // of the column lies inside the box and the tile does not straddle the periodic x edge, so
    // the unreduced cell (int)floor(s) minus the tile origin is the LDS coordinate (the same
    // expression decided the particle's tile in tile_of(); it raised col_flags otherwise).
    // mass * scale / quantum in double (no intermediate rounding to T); unit masses: a constant
    const double scale_invq = scale * invq, m_unit = scale_invq;
    auto deposit = [&](const T (&pc)[3 * U], const T (&mc)[U], uint32_t act, auto careful_tag) {
        constexpr bool CAREFUL = decltype(careful_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!((act >> u) & 1u)) continue;
            if (ablate & 256) { asm volatile("" ::"v"(pc[3 * u]), "v"(pc[3 * u + 1]), "v"(pc[3 * u + 2])); continue; }
            double fx, fy, fz;
            int lx, ly, lz;
            if (!CAREFUL) {
                const double sx = grid_coord(pc[3 * u + 0], g), sy = grid_coord(pc[3 * u + 1], g),
                             sz = grid_coord(pc[3 * u + 2], g);
                const double flx = floor(W == 2 ? sx : sx + 0.5), fly = floor(W == 2 ? sy : sy + 0.5),
                             flz = floor(W == 2 ? sz : sz + 0.5);
                fx = sx - flx;
                fy = sy - fly;
                fz = sz - flz;
                lx = (int)flx - orx;
                ly = (int)fly - oy;
                lz = (int)flz - oz;
            } else {
                lx = ast::locate_rel<W>(grid_coord(pc[3 * u + 0], g), g.n, orx, fx);
                ly = ast::locate_rel<W>(grid_coord(pc[3 * u + 1], g), g.n, oy, fy);
                lz = ast::locate_rel<W>(grid_coord(pc[3 * u + 2], g), g.n, oz, fz);
                if ((unsigned)lx >= ex || (unsigned)ly >= ey || (unsigned)lz >= ez) {
                    // more than a box length outside the box: the general reduction
                    int bx = ast::locate<W>(grid_coord(pc[3 * u + 0], g), g.n, fx) - g.x_start;
                    if (bx < 0) bx += g.n;
                    lx = bx - ox;
                    ly = ast::locate<W>(grid_coord(pc[3 * u + 1], g), g.n, fy) - oy;
                    lz = ast::locate<W>(grid_coord(pc[3 * u + 2], g), g.n, fz) - oz;
                    if ((unsigned)lx >= ex || (unsigned)ly >= ey || (unsigned)lz >= ez) continue;   // not in this tile: cannot happen
                }
            }
            // weights and products in double, pre-scaled by 1/quantum; the last product and the
            // magic-number add are one fma (a single rounding of the exact product to the quantum)
            double wx[W], wy[W], wz[W];
            Window<W>::weights(fx, wx);
            Window<W>::weights(fy, wy);
            Window<W>::weights(fz, wz);
            const double m = HAS_MASS ? (double)mc[u] * scale_invq : m_unit;
            unsigned long long* slot[W];          // (lx, ly) row at the ring slots of planes lz + c
            int sl = lz + sh;
            sl = sl >= LZ ? sl - LZ : sl;
            // (24-bit multiplies: the plain expression compiles to a quarter-rate v_mad_u64_u32 and v_mul_lo_u32 per particle)
            unsigned long long* const row0 = &tile[__umul24(__umul24((unsigned)lx, (unsigned)LYP) + (unsigned)ly, (unsigned)PZ)];
#pragma unroll
            for (int c = 0; c < W; ++c) {
                slot[c] = row0 + sl;
                sl = sl + 1 == LZ ? 0 : sl + 1;
            }
#pragma unroll
            for (int a = 0; a < W; ++a) {
                const double mwa = m * wx[a];
#pragma unroll
                for (int b = 0; b < W; ++b) {
                    const double mab = mwa * wy[b];
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        long long v = __double_as_longlong(__fma_rn(mab, wz[c], 6755399441055744.0));   // |mab wz| < 2^50
                        if (!RAW) v -= 0x4338000000000000ll;
                        unsigned long long* cell = slot[c] + (a * LYP + b) * PZ;
                        if (ablate & 2) asm volatile("" ::"v"(v), "v"(cell)); else atomicAdd(cell, (unsigned long long)v);
                    }
                }
            }
        }
    };
    const bool careful = col_flags[col] != 0 || orx + TX > g.n;       // uniform
    // Two register sets, used alternately (a copy at the end of a step would have to wait for
    // the loads it copies): pos/mass A|B of batch k|k+1, indices X|Y of batch k+1|k+2.
    T pA[3 * U], mA[U], pB[3 * U], mB[U];
    GroupRec iX[U], iY[U];
    uint32_t aA = 0, aB = 0;                // occupied slots of the position sets
    if (cur_tz < nsteps) {                  // uniform: the column holds particles
        load_idx(nxt, iY);
        load_pos(nxt, iY, aA, pA, mA);
        nxt = next_batch(nxt);
        load_idx(nxt, iX);
    }
    // One loop, two phases with the register sets swapped, so no set is ever copied; tiles are
    // flushed (all the empty ones too: every grid cell gets written) before the first batch of a
    // later tile is deposited.
    int ftz = 0;                            // next tile to flush; the ring is positioned for it
    auto flush_until = [&](int tz_end) {
        for (; ftz < tz_end; ++ftz) {
            DSTAMP(5);
            if (!(ablate & 64)) __syncthreads();
            DSTAMP(6);

            // planes c = 0..TZ-1 are final (z = ftz * TZ - LO + c): store them and re-arm their slots.
            // A thread keeps its VW consecutive planes (one 16-byte store: the flush is bound by the
            // rate at which a CU issues stores, and dword stores move a quarter of the bytes per
            // instruction) and walks the (a, b) columns; everything but the values comes from the
            // per-column table.
            constexpr int VW = 16 / (int)sizeof(T), LPC = TZ / VW, CPI = 256 / LPC;    // values per lane, lanes per column, columns per sweep
            static_assert(TZ % VW == 0 && 256 % LPC == 0, "flush mapping");
            typedef T vec_t __attribute__((ext_vector_type(VW), aligned(4)));
            const int c0 = (threadIdx.x % LPC) * VW;
            int sl[VW];
            sl[0] = c0 + sh;
            sl[0] = sl[0] >= LZ ? sl[0] - LZ : sl[0];
#pragma unroll
            for (int i = 1; i < VW; ++i) sl[i] = sl[i - 1] + 1 == LZ ? 0 : sl[i - 1] + 1;
            const int z0 = phys(ftz) * TZ - LO + c0;
            const bool straight = z0 >= 0 && z0 + VW <= g.n;          // no periodic wrap inside the vector
#pragma unroll
            for (int k = 0; k < (LX * LY + CPI - 1) / CPI; ++k) {
                const int ab = threadIdx.x / LPC + k * CPI;
                if (ab >= LX * LY || (ablate & 128)) break;
                const unsigned long long d = dest[ab];
                const double sub = (d & 2ull) ? offset : 0.0;
                vec_t v;
                bool any = false;
#pragma unroll
                for (int i = 0; i < VW; ++i) {
                    const unsigned long long raw = tile[trow(ab) * PZ + sl[i]];
                    tile[trow(ab) * PZ + sl[i]] = BIAS;
                    any |= raw != BIAS;
                    if (ftz == 0 && c0 + i < H) first_planes[ab * H + c0 + i] = raw;     // (stored below, rewritten at the end)
                    if (RAW) {
                        const unsigned long long dbits = (raw & 0x0000ffffffffffffull) | 0x4330000000000000ull;
                        v[i] = (T)((__longlong_as_double((long long)dbits) - 4644337115725824.0) * q - sub);      // 2^52 + 2^47
                    } else {
                        v[i] = (T)((double)(long long)raw * q - sub);
                    }
                }
                if (d == 0ull) { ndrop += any; continue; }
                if (ablate & 1) continue;
                // (global address space spelled out: a pointer rebuilt from an integer is "flat")
                typedef __attribute__((address_space(1))) T gT;
                typedef __attribute__((address_space(1))) vec_t gvec_t;
                gT* const line = (gT*)(d & ~3ull);
                if (straight) {
                    *(gvec_t*)(line + z0) = v;
                } else {
#pragma unroll
                    for (int i = 0; i < VW; ++i) line[ast::wrap1(z0 + i, g.n)] = v[i];
                }
                if (!x_periodic && (d & 1ull)) {
#pragma unroll
                    for (int i = 0; i < VW; ++i) ndrop += v[i] != (T)0;
                }
            }
            sh += TZ;
            sh = sh >= LZ ? sh - LZ : sh;
            DSTAMP(7);
            if (!(ablate & 64)) __syncthreads();
            DSTAMP(8);
        }
    };
    DSTAMP(0);
    for (;;) {
        flush_until(cur_tz);
        if (cur_tz >= nsteps) break;
        oz = phys(cur_tz) * TZ;
        {
            DSTAMP(1);
            const Batch nn = next_batch(nxt);
            DSTAMP(2);
            load_pos(nxt, iX, aB, pB, mB);                // batch k+1 (a harmless re-load at the end)
            load_idx(nn, iY);                             // batch k+2
            DSTAMP(3);
            if (careful) deposit(pA, mA, aA, std::true_type{}); else deposit(pA, mA, aA, std::false_type{});
            DSTAMP(4);
            cur_tz = nxt.tz;
            nxt = nn;
        }
        flush_until(cur_tz);
        if (cur_tz >= nsteps) break;
        oz = phys(cur_tz) * TZ;
        {
            DSTAMP(1);
            const Batch nn = next_batch(nxt);
            DSTAMP(2);
            load_pos(nxt, iY, aA, pA, mA);
            load_idx(nn, iX);
            DSTAMP(3);
            if (careful) deposit(pB, mB, aB, std::true_type{}); else deposit(pB, mB, aB, std::false_type{});
            DSTAMP(4);
            cur_tz = nxt.tz;
            nxt = nn;
        }
    }
    DSTAMP(9);
    DSTAMP_END;

    // The H planes still in the ring hold z = n - LO + k (k < H): the periodic wrap onto the first planes this
    // workgroup flushed.  Their exact sums were kept in first_planes, so the cells are REWRITTEN with the one
    // correctly rounded total (exact integer sum of both parts, offset subtracted in double) - plain stores, no
    // read-modify-write.  The earlier stores to the same cells are this workgroup's own: draining them (a
    // workgroup-scope release; a __threadfence() here is a buffer_wbl2 + buffer_inv of the whole L2 per wave)
    // and the barrier order the two.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (nseg > 1) {
        // the seam is another workgroup's: both parts leave as exact sums, [column][segment][first | top][ab][k]
        unsigned long long* const zf = zrec + ((size_t)col * nseg + seg) * 2 * (LX * LY * H);
        for (int i = threadIdx.x; i < LX * LY * H; i += 256) {
            const int k = i % H, ab = i / H;
            int sl = k + sh;
            sl = sl >= LZ ? sl - LZ : sl;
            zf[i] = first_planes[i];
            zf[LX * LY * H + i] = tile[trow(ab) * PZ + sl];
        }
        if (dropped && ndrop) atomicAdd(dropped, ndrop);
        return;
    }
    for (int i = threadIdx.x; i < LX * LY * H; i += 256) {
        const int k = i % H, ab = i / H;
        int sl = k + sh;
        sl = sl >= LZ ? sl - LZ : sl;
        const unsigned long long raw = first_planes[i] + tile[trow(ab) * PZ + sl] - BIAS;
        const unsigned long long d = dest[ab];
        bool any;
        const T v = fixed_to_value<T>(raw, q, (d & 2ull) ? offset : 0.0, any);
        if (d == 0ull) { ndrop += any; continue; }      // owned plane the buffer does not hold
        typedef __attribute__((address_space(1))) T gT;
        ((gT*)(d & ~3ull))[ast::wrap1(tz0 * TZ + k - LO, g.n)] = v;
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

// nseg > 1: the seams between the z-segments of a column.  Segment s starts at tile (t0(col) + s * ntz / nseg) mod ntz;
// its first H planes are the sum of what it deposited there itself (first) and the top halo of the segment below
// (top): both exact, so the cell gets the one correctly rounded total - the same value the unsegmented walk stores.
template <typename T, int W>
__global__ void __launch_bounds__(256)
z_seam_kernel(const unsigned long long* __restrict__ zrec, TileGeom g, double scale, uint32_t cmax, double mass_bound,
              T* __restrict__ grid, T* __restrict__ rec, double offset, unsigned long long* dropped, int col0, int nseg) {
    constexpr int LX = TX + W - 1, LY = TY + W - 1;
    constexpr int LO = Window<W>::LO, H = W - 1;
    constexpr bool RAW = sizeof(T) == 4;
    constexpr unsigned long long BIAS = RAW ? (1ull << 47) : 0ull;
    int col = col0 + (int)blockIdx.x;
    col = col >= g.ntx * g.nty ? col - g.ntx * g.nty : col;
    const int ty = col % g.nty, tx = col / g.nty;
    const int ox = tx * TX, oy = ty * TY;
    const int nsteps = g.ntz / nseg;
    const int t00 = (int)(((unsigned)col * 2654435761u >> 16) % (unsigned)g.ntz);
    const double q = 1.0 / paint_inv_quantum<RAW>(cmax, mass_bound, scale);
    unsigned long long ndrop = 0;
    for (int i = threadIdx.x; i < nseg * LX * LY * H; i += 256) {
        const int seg = i / (LX * LY * H), j = i % (LX * LY * H);
        const int k = j % H, ab = j / H;
        const int below = seg == 0 ? nseg - 1 : seg - 1;
        const unsigned long long raw = zrec[((size_t)col * nseg + seg) * 2 * (LX * LY * H) + j] +
                                       zrec[((size_t)col * nseg + below) * 2 * (LX * LY * H) + LX * LY * H + j] - BIAS;
        const unsigned long long d = column_line_dest<T, W>(ab / LY, ab % LY, col, ox, oy, g, grid, rec);
        bool any;
        const T v = fixed_to_value<T>(raw, q, (d & 2ull) ? offset : 0.0, any);
        if (d == 0ull) { ndrop += any; continue; }      // owned plane the buffer does not hold
        int tz0 = t00 + seg * nsteps;
        tz0 = tz0 >= g.ntz ? tz0 - g.ntz : tz0;
        typedef __attribute__((address_space(1))) T gT;
        ((gT*)(d & ~3ull))[ast::wrap1(tz0 * TZ + k - LO, g.n)] = v;
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

#ifdef PAINT_WALK_TEST
// perf experiment: the plainest possible walk over a column's index lists and positions
template <typename T>
__global__ void __launch_bounds__(256)
walk_test_kernel(const T* __restrict__ pos, TileGeom g, const uint32_t* __restrict__ index,
                 const uint32_t* __restrict__ tile_count, uint32_t cap, T* out) {
    __shared__ T pad[5400];
    T acc = 0;
    for (int tz = 0; tz < g.ntz; ++tz) {
        const uint32_t t = blockIdx.x * g.ntz + tz;
        const uint32_t cnt = min(tile_count[t], cap);
        const size_t off = (size_t)t * cap;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 1024) {
            uint32_t idx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) idx[u] = index[off + min(i0 + u * 256 + threadIdx.x, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t q = (size_t)idx[u] * 3;
                acc += pos[q] + pos[q + 1] + pos[q + 2];
            }
        }
    }
    if (acc == (T)1234.5) { out[0] = acc; pad[threadIdx.x] = acc; out[1] = pad[(threadIdx.x * 7) % 5400]; }
}
#endif

// Every column adds its neighbours' halo records into its own border rows.  Only the border cells
// have anything to add (15 of 64 for CIC, 28 for TSC), each from at most 3 neighbours: a small
// table of (grid line, up to 3 record lines) is built in LDS and then walked with 16-byte
// loads/stores along z, sources added in a fixed order.
template <typename T, int W>
__global__ void __launch_bounds__(256)
column_fold_kernel(const T* __restrict__ rec, TileGeom g, T* __restrict__ grid, int col0) {
    constexpr int LO = Window<W>::LO;
    using RM = RingMap<W>;
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));          // lines are 16-byte aligned (n % 32 == 0)
    __shared__ unsigned long long line_dst[TX * TY], line_src[TX * TY][3];
    __shared__ int nlines;
    const int col = col0 + (int)blockIdx.x;
    const int ty = col % g.nty, tx = col / g.nty;
    const int ox = tx * TX, oy = ty * TY;
    const bool x_periodic = g.nx_alloc == g.n;
    if (threadIdx.x == 0) nlines = 0;
    __syncthreads();
    if (threadIdx.x < TX * TY) {
        const int ao = threadIdx.x / TY, bo = threadIdx.x % TY;       // owned cell of this column
        unsigned long long src[3] = {0ull, 0ull, 0ull};
        int ns = 0;
        if (ox + ao < g.nx_alloc) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    if (dx == 0 && dy == 0) continue;
                    const int a = ao + LO - dx * TX, b = bo + LO - dy * TY;   // this cell in the neighbour's LDS frame
                    if (a < 0 || a >= RM::LX || b < 0 || b >= RM::LY) continue;
                    int ntx = tx + dx;
                    if (x_periodic) ntx = ast::wrap1(ntx, g.ntx);
                    else if (ntx < 0 || ntx >= g.ntx) continue;
                    const int nty_ = ast::wrap1(ty + dy, g.nty);
                    if (ns < 3) src[ns++] = (unsigned long long)(rec + ((size_t)(ntx * g.nty + nty_) * RM::COUNT + RM::cell(a, b)) * ast::rec_pitch((size_t)g.n));
                }
            }
        }
        if (ns) {
            const int k = atomicAdd(&nlines, 1);
            line_dst[k] = (unsigned long long)(grid + ((size_t)(ox + ao) * g.n + oy + bo) * g.n);
            line_src[k][0] = src[0];
            line_src[k][1] = src[1];
            line_src[k][2] = src[2];
        }
    }
    __syncthreads();
    const int per_line = g.n / VW;
    const int items = nlines * per_line;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int k = it / per_line, zc = (it % per_line) * VW;
        typedef __attribute__((address_space(1))) vec_t gvec_t;
        gvec_t* const dst = (gvec_t*)((__attribute__((address_space(1))) T*)line_dst[k] + zc);
        // sum of the neighbours first, then onto the cell (the slot order of the table is not
        // deterministic, the order within a cell is)
        vec_t sum = *(const gvec_t*)((__attribute__((address_space(1))) T*)line_src[k][0] + zc);
        if (line_src[k][1]) sum += *(const gvec_t*)((__attribute__((address_space(1))) T*)line_src[k][1] + zc);
        if (line_src[k][2]) sum += *(const gvec_t*)((__attribute__((address_space(1))) T*)line_src[k][2] + zc);
        *dst = *dst + sum;
    }
}

// Second stream of the x-sorted pipeline and the events that order it against the caller's stream (per device, made once).
struct SidePipe {
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev;
    std::mutex run;          // held for a whole pipelined paint: one set of events and one side stream per device
};
SidePipe* side_pipe(int nevents) {
    static std::mutex mu;
    static SidePipe pipes[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    SidePipe& p = pipes[dev];
    if (!p.side && hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) != hipSuccess) { p.side = nullptr; return nullptr; }
    while ((int)p.ev.size() < nevents) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        p.ev.push_back(e);
    }
    return &p;
}

// ast_paint_tiled_stage: which part of the single-pass overwrite paint a call runs (tile rows [row0, row0 + nrows))
struct StageSel {
    int stage = AST_PAINT_STAGE_ALL, row0 = 0, nrows = 0;
    int closed_row0 = 0, closed_nrows = 0;        // GROUP_PART: tile rows already walked (a range that may wrap around)
};

// How the single-pass overwrite paint is cut up: z-segments per column (walk workgroups = columns x nseg) and, with
// AST_PAINT_XSORTED, chunks of `chunk_planes` buffer planes' worth of particles.  A function of the geometry, np and
// the flags only (the workspace is sized from it); AST_PAINT_ZSEG / AST_PAINT_XCHUNK_MB / AST_PAINT_XMARGIN /
// AST_PAINT_XSTREAMS override it for experiments.
struct WalkPlan {
    int nseg;            // z-segments per column (divides ntz)
    int chunks;          // 1: not chunked
    int chunk_planes;    // planes of particles per chunk
    int margin_planes;   // a tile row is walked once the chunks cover its planes plus this many
    int streams;         // 2: the walks run on a second stream beside the grouping of the next chunks
};
inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
inline int seg_count(int want, int ntz) {            // largest divisor of ntz that is a power of two <= want
    int n = 1;
    while (n * 2 <= want && ntz % (n * 2) == 0) n *= 2;
    return n;
}
inline WalkPlan walk_plan(const TileGeom& g, size_t np, int flags, size_t esz) {
    WalkPlan p{1, 1, g.nx_alloc, 0, 1};
    if ((flags & AST_PAINT_TWO_PASS) || !(flags & AST_PAINT_OVERWRITE)) return p;
    const unsigned ncols = (unsigned)(g.ntx * g.nty);
    int rows_per_launch = g.ntx;
    if ((flags & AST_PAINT_XSORTED) && !(flags & AST_PAINT_SCATTERED)) {
        // Default: FOUR chunks, walks on a second stream beside the grouping of the next chunk (8.0 against 8.24 ms
        // at 1024^3: both kernels are bound by the bytes they move, overlapping them buys little).  Measured and not
        // adopted: chunks of 24-96 MB of positions (AST_PAINT_XCHUNK_MB), so that a chunk would still sit in the
        // 256 MB Infinity Cache when the walk gathers it - launches of ~1000 workgroups are bound by their tails:
        // 13.5-20.7 ms, grouping 5.8-10.8 and walks 6.0 where the two big launches take 3.7 + 4.7.
        const double plane_bytes = (double)np * 3.0 * (double)esz / (double)g.nx_alloc;
        const int mb = env_int("AST_PAINT_XCHUNK_MB", 0);
        int cp = mb > 0 ? (int)(1048576.0 * (double)mb / plane_bytes + 0.5) : (g.nx_alloc + 3) / 4;
        cp = std::max(2, std::min(cp, g.nx_alloc));
        const int K = (g.nx_alloc + cp - 1) / cp;
        if (K >= 2 && g.ntx >= 4 * K / 2 && np >= (size_t)K * 16 * 4096) {
            p.chunks = K;
            p.chunk_planes = cp;
            p.margin_planes = env_int("AST_PAINT_XMARGIN", g.nx_alloc == g.n ? 4 : 12);
            p.streams = env_int("AST_PAINT_XSTREAMS", mb > 0 ? 1 : 2);
            rows_per_launch = std::max(1, cp / TX);
        }
    }
    // enough workgroups per walk launch to fill 256 CUs a few times over
    const unsigned per_launch = (unsigned)rows_per_launch * (unsigned)g.nty;
    int want = 1;
    if (p.chunks > 1) { while (per_launch * (unsigned)want < 2048u && want < 16) want *= 2; }
    else if (ncols < 4096u) { while (ncols * (unsigned)want < 8192u && want < 8) want *= 2; }
    p.nseg = seg_count(env_int("AST_PAINT_ZSEG", want), g.ntz);
    return p;
}

struct Workspace {
    unsigned long long* ovf_count;   // single pass: particles in the overflow list
    uint32_t* col_flags;             // per tile column: 1 = holds a particle outside the box (general cell lookup)
    uint32_t* tile_count;            // two pass: exact counts; single pass: unused
    uint32_t* tile_fill;             // slots requested per tile
    uint32_t* tile_off;              // two pass: exclusive scan of tile_count
    uint32_t* block_sums;
    uint32_t* index;                 // particle indices, tile-major
    uint32_t* ovf;                   // single pass: indices that did not fit their tile's segment
    void* rec;                       // OVERWRITE flush: per-column halo records [column][ring cell][z]
    unsigned long long* zrec;        // z-segmented walk: exact seam sums [column][segment][first | top][LX * LY * H]
    uint32_t cap;                    // single pass: index slots per tile
    // single pass + OVERWRITE: the compact lists of tile_group_kernel instead of `index`
    unsigned long long* fill64;      // per tile: group records requested << 32 | strays requested
    GroupRec* recs;                  // [tile][rcap]
    void* strays;                    // [tile][scap] x {x, y, z, m}
    uint32_t rcap, scap;
    // AST_PAINT_SCATTERED: two-level bucket scatter
    unsigned long long* bcursor;     // [SC_GROUPS][nb] level A write cursors = records per (chunk label, coarse bucket)
    uint32_t cap_bg;                 // records per (bucket, label) segment of the staging array
    unsigned long long* late;        // records that found their tile's segment full (deposited on the spot)
    void* staging;                   // [np] x {x, y, z, m}, bucket-major
    uint32_t tpb;                    // tiles per bucket (0: the scatter path does not apply)
    uint32_t nb;                     // buckets (scatter_buckets)
    uint32_t* late_index;            // record numbers of the late list, tile-major (its deposit through LDS tiles)
    size_t bytes;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Single pass: every tile gets room for twice the mean occupancy (at least mean + 512).
inline uint32_t tile_capacity(size_t np, uint32_t ntiles) {
    const size_t mean = (np + ntiles - 1) / ntiles;
    size_t cap = 2 * mean > mean + 512 ? 2 * mean : mean + 512;
    cap = (cap + 63) / 64 * 64;
    return (uint32_t)(cap > 0x7fffffffull ? 0x7fffffffull : cap);
}

// flags decide the list format: two pass -> exact id lists; single pass -> fixed-capacity id segments, or
// (with AST_PAINT_OVERWRITE) the compact group / stray lists.  A tile's group segment holds cap / MINPOP records
// (enough for `cap` particles however they are grouped), its stray segment cap / 4 copies - or `cap` with
// AST_PAINT_SCATTERED, for input without spatial order where every particle is a stray.
// Buckets of the two-level scatter: the power of two nearest to sqrt(ntiles) from above, in [64, SC_BUCKETS_MAX] - both
// levels then split into about equally many destinations (1024^3: 524288 tiles -> 1024 buckets of 512 tiles; 512 buckets of
// 1024 measure the same within 1 %) - and enough of them that a bucket's tiles fit level B's tables.
// AST_PAINT_SC_BUCKETS overrides (experiments; a multiple of 8).
uint32_t scatter_buckets(uint32_t ntiles) {
    static const uint32_t forced = [] { const char* v = getenv("AST_PAINT_SC_BUCKETS"); return v ? (uint32_t)atoi(v) : 0u; }();
    if (forced >= 8 && forced <= SC_BUCKETS_MAX && forced % 8 == 0) return forced;
    uint32_t nb = 64;
    while (nb < SC_BUCKETS_MAX && ((unsigned long long)nb * nb < ntiles || (ntiles + nb - 1) / nb > SC_TPB_MAX)) nb <<= 1;
    return nb;
}

Workspace carve(void* base, size_t np, uint32_t ntiles, uint32_t ncols, int flags, size_t esz, size_t rec_bytes, size_t zrec_bytes = 0) {
    const bool two_pass = (flags & AST_PAINT_TWO_PASS) != 0;
    const bool compact = !two_pass && (flags & AST_PAINT_OVERWRITE);
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = (char*)base + off; off += align256(bytes); return p; };
    w.ovf_count = (unsigned long long*)take(8);
    w.col_flags = (uint32_t*)take((size_t)ncols * 4);
    w.tile_count = (uint32_t*)take((size_t)ntiles * 4);
    w.tile_fill = (uint32_t*)take((size_t)ntiles * 4);
    w.fill64 = (unsigned long long*)take(compact ? (size_t)ntiles * 8 : 0);
    const bool scattered = compact && (flags & AST_PAINT_SCATTERED);
    w.nb = scatter_buckets(ntiles);
    w.tpb = scattered ? (ntiles + w.nb - 1) / w.nb : 0;
    if (w.tpb > SC_TPB_MAX || np < (size_t)8192) w.tpb = 0;      // huge grids / tiny inputs: the grouping kernel does it
    w.bcursor = (unsigned long long*)take(w.tpb ? (size_t)w.nb * SC_GROUPS * 8 : 0);      // (inside the part run_tiled zeroes)
    // a (bucket, label) segment holds twice its mean share of the particles
    w.cap_bg = w.tpb ? (uint32_t)((2 * ((np + (size_t)w.nb * SC_GROUPS - 1) / ((size_t)w.nb * SC_GROUPS)) + 1024 + 63) / 64 * 64) : 0;
    w.late = (unsigned long long*)take(w.tpb ? 8 : 0);
    w.tile_off = (uint32_t*)take((size_t)ntiles * 4);
    w.block_sums = (uint32_t*)take((size_t)((ntiles + 1023) / 1024 + 1) * 4);
    w.cap = two_pass ? 0 : tile_capacity(np, ntiles);
    w.rcap = compact && !w.tpb ? (w.cap + MINPOP - 1) / MINPOP : 0;       // the scatter path writes stray copies only
    w.scap = compact ? ((flags & AST_PAINT_SCATTERED) ? w.cap : (w.cap + 3) / 4) : 0;
    {   // At 1024^3 a tile's record segment is 512 x 8 bytes and its stray segment 1024 x 12: strides of 4 KB and 12 KB, so every
        // segment starts at the same offset inside a 4-KB block and the grouping kernel's stores crowd a few channels - how
        // badly depends on where the workspace's pages lie (the step's 12.6-13.1 ms spread).  16 records / 32 copies more per
        // segment make the strides odd multiples of 128 bytes: grouping 3.61 -> 3.46 ms, step 12.95 -> 12.79 (means over ten
        // placements each, scripts/micro/placement_step.py; AST_PAINT_SEG_SKEW=0 restores the old strides).
        static const int skew = getenv("AST_PAINT_SEG_SKEW") ? atoi(getenv("AST_PAINT_SEG_SKEW")) : 1;
        if (skew && w.rcap) w.rcap += 16 * skew;
        if (skew && w.scap) w.scap += 32 * skew;
        // (the scatter path: level B's stores into the stray segments 5.7 -> 5.5 ms; level A's staging segments - 3 MB apart -
        // measured worse with such a skew, 6.6 -> 6.8, and keep their stride)
    }
    w.index = (uint32_t*)take(two_pass ? np * 4 : compact ? 0 : (size_t)ntiles * w.cap * 4);
    w.recs = (GroupRec*)take((size_t)ntiles * w.rcap * sizeof(GroupRec));
    w.strays = take((size_t)ntiles * w.scap * 4 * esz);
    w.staging = take(w.tpb ? (size_t)w.nb * SC_GROUPS * w.cap_bg * 4 * esz : 0);
    w.ovf = (uint32_t*)take(two_pass ? 0 : np * 4);
    w.late_index = (uint32_t*)take(w.tpb ? np / 4 * sizeof(uint32_t) / esz * 4 : 0);      // one id per record the list has room for
    w.rec = take(rec_bytes);
    w.zrec = (unsigned long long*)take(zrec_bytes);
    w.bytes = off;
    return w;
}

bool tiled_geometry(int nmesh, int nx_alloc, TileGeom& g, uint32_t& ntiles) {
    if (nmesh % TY || nmesh % TZ) return false;
    g.n = nmesh;
    g.nx_alloc = nx_alloc;
    g.ntx = (nx_alloc + TX - 1) / TX;
    g.nty = nmesh / TY;
    g.ntz = nmesh / TZ;
    const unsigned long long nt = (unsigned long long)g.ntx * g.nty * g.ntz;
    if (nt >= (1ull << 31) || (unsigned long long)g.ntx * g.nty >= (1ull << 24)) return false;      // (24-bit multiplies in tile_of)
    ntiles = (uint32_t)nt;
    return true;
}

inline size_t seam_bytes(int window, const TileGeom& g, size_t np, size_t esz, int flags) {
    const WalkPlan p = walk_plan(g, np, flags, esz);
    if (p.nseg <= 1) return 0;
    const size_t cells = window == AST_WIN_TSC ? (size_t)(TX + 2) * (TY + 2) * 2 : (size_t)(TX + 1) * (TY + 1);
    return (size_t)g.ntx * g.nty * p.nseg * 2 * cells * sizeof(unsigned long long);
}

inline size_t record_bytes(int window, const TileGeom& g, size_t esz, int flags) {
    if (!(flags & AST_PAINT_OVERWRITE)) return 0;
    const size_t ring = window == AST_WIN_TSC ? RingMap<3>::COUNT : RingMap<2>::COUNT;
    return (size_t)g.ntx * g.nty * ring * ast::rec_pitch((size_t)g.n) * esz;
}

template <typename T, int W>
int run_tiled(const T* pos, const T* mass, size_t np, TileGeom g, uint32_t ntiles, double scale, T* grid,
              void* workspace, unsigned long long* dropped, int flags, double mass_bound, double offset, hipStream_t s,
              StageSel sel = StageSel{}) {
    const bool two_pass = (flags & AST_PAINT_TWO_PASS) != 0;
    const bool overwrite = (flags & AST_PAINT_OVERWRITE) != 0;
    const unsigned ncols = (unsigned)(g.ntx * g.nty);
    const WalkPlan plan = walk_plan(g, np, flags, sizeof(T));
    Workspace w = carve(workspace, np, ntiles, ncols, flags, sizeof(T),
                        record_bytes(W == 3 ? AST_WIN_TSC : AST_WIN_CIC, g, sizeof(T), flags),
                        seam_bytes(W == 3 ? AST_WIN_TSC : AST_WIN_CIC, g, np, sizeof(T), flags));
    // ovf_count, col_flags, tile_count, tile_fill and fill64 are contiguous at the front of the workspace
    if (sel.stage == AST_PAINT_STAGE_ALL || sel.stage == AST_PAINT_STAGE_GROUP || sel.stage == AST_PAINT_STAGE_RESET)
        AST_CHECK_HIP(hipMemsetAsync(w.ovf_count, 0, (size_t)((char*)w.tile_off - (char*)w.ovf_count), s));
    const size_t per_interval = (size_t)256 * IDX_UNROLL * AGG_TRIPS;
    const size_t nintervals = (np + per_interval - 1) / per_interval;
    // grid of the index pass: ONE interval per workgroup.  Measured at 1024^3: 8192 workgroups of 32
    // intervals 4.72 ms, 65536 x 4: 4.40, 262144 x 1: 4.28; persistent grids of a few per CU are the
    // slowest (4.95-5.0): short workgroups keep the CUs' phases mixed and leave no ragged last round.
    // (The kernel still loops, for inputs beyond 2^31 intervals and for AST_PAINT_INDEX_GRID experiments.)
    const size_t want = getenv("AST_PAINT_INDEX_GRID") ? (size_t)atol(getenv("AST_PAINT_INDEX_GRID")) : (size_t)0x7fffffff;
    const unsigned ga = (unsigned)(nintervals > want ? want : nintervals);
    const bool plainx = g.x_start == 0 && g.nx_alloc == g.n;
    auto index_pass = [&](auto mode, uint32_t* tile_count, const uint32_t* tile_off, uint32_t* tile_fill, uint32_t* index,
                          uint32_t cap, uint32_t* ovf, unsigned long long* ovf_count, unsigned long long* drop) {
        constexpr int MODE = decltype(mode)::value;
        if (plainx)
            tile_index_kernel<T, W, MODE, true><<<ga, 256, 0, s>>>(pos, np, g, tile_count, tile_off, tile_fill, index, cap, ovf,
                                                                   ovf_count, w.col_flags, drop);
        else
            tile_index_kernel<T, W, MODE, false><<<ga, 256, 0, s>>>(pos, np, g, tile_count, tile_off, tile_fill, index, cap, ovf,
                                                                    ovf_count, w.col_flags, drop);
    };
    // the column walk over the columns col0 .. col0 + ncol - 1 (mod ncols) on stream st
    auto walk_pass = [&](const uint32_t* tile_off, const uint32_t* tile_count, int col0, unsigned ncol, hipStream_t st) {
        AST_PROF("paint_tiled.deposit", st);
        WalkCaps wl{w.rcap, w.scap, w.cap, np};
        using I2 = std::integral_constant<int, 2>;
        const int nseg = two_pass ? 1 : plan.nseg;
        auto launch = [&](auto has_mass, auto fmt) {
            column_deposit_kernel<T, W, decltype(has_mass)::value, decltype(fmt)::value><<<ncol * (unsigned)nseg, 256, 0, st>>>(
                pos, mass, g, scale, w.index, tile_off, tile_count, w.fill64, w.recs, (const T*)w.strays, wl,
                mass ? mass_bound : 1.0, w.col_flags, grid, (T*)w.rec, offset, dropped, col0, nseg, w.zrec);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if (two_pass) { if (mass) launch(std::true_type{}, I0{}); else launch(std::false_type{}, I0{}); }
        else if (mass) launch(std::true_type{}, I1{});
        else launch(std::false_type{}, I2{});                     // no masses: 3-word stray copies
    };
    // the seams of a z-segmented walk (all columns, after their walks), then the x / y halo fold
    auto fold_pass = [&]() {
        if (!two_pass && plan.nseg > 1) {
            AST_PROF("paint_tiled.seams", s);
            z_seam_kernel<T, W><<<ncols, 256, 0, s>>>(w.zrec, g, scale, (uint32_t)std::min<unsigned long long>(5ull * w.cap, 0x3fffffffull),
                                                       mass ? mass_bound : 1.0, grid, (T*)w.rec, offset, dropped, 0, plan.nseg);
        }
        if (!(flags & AST_PAINT_DEFER_FOLD)) {
            AST_PROF("paint_tiled.fold", s);
            column_fold_kernel<T, W><<<ncols, 256, 0, s>>>((const T*)w.rec, g, grid, 0);
        }
    };
    auto deposit_pass = [&](const uint32_t* tile_off, const uint32_t* tile_count, uint32_t cap) {
        if (overwrite) {
            walk_pass(tile_off, tile_count, 0, ncols, s);
            fold_pass();
        } else {
            AST_PROF("paint_tiled.deposit", s);
            tile_deposit_kernel<T, W><<<ntiles, 256, 0, s>>>(pos, mass, g, scale, w.index, tile_off, tile_count, cap, grid, dropped);
        }
    };
    // AST_PAINT_SCATTERED: the two bucket levels (particles -> staging -> the tiles' stray segments)
    auto scatter_pass = [&]() -> int {
    {
        const unsigned nchunks = (unsigned)((np + SCA_THREADS * SC_PER_THREAD - 1) / (SCA_THREADS * SC_PER_THREAD));
        auto run = [&](auto px, auto sw) -> int {
            constexpr bool PX = decltype(px)::value;
            constexpr int SW = decltype(sw)::value;
            const size_t lds_a = sc_round<T, SCA_THREADS>() * (SW * sizeof(T) + sizeof(unsigned long long));
            const size_t lds_b = sc_round<T, SCB_THREADS>() * (SW * sizeof(T) + sizeof(unsigned long long)) + (size_t)((w.tpb + 1u) & ~1u) * 20;
            const size_t lds_b_max = sc_round<T, SCB_THREADS>() * (SW * sizeof(T) + sizeof(unsigned long long)) + (size_t)SC_TPB_MAX * 20;
            static ast::PerDeviceOnce attr_once;
            if (attr_once.need()) {
                AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_level_a_kernel<T, W, PX, SW>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
                AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_level_b_kernel<T, W, PX, SW>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b_max));
                attr_once.mark();
            }
            const unsigned long long late_cap = np / 4 * sizeof(uint32_t) / sizeof(T);       // the overflow list's room
            {
                AST_PROF("paint_tiled.level_a", s);
                scatter_level_a_kernel<T, W, PX, SW><<<nchunks, SCA_THREADS, lds_a, s>>>(
                    pos, mass, np, g, w.tpb, w.nb, w.bcursor, (T*)w.staging, w.cap_bg, w.col_flags, (T*)w.ovf, late_cap, w.late, dropped);
            }
            AST_PROF("paint_tiled.level_b", s);
            scatter_level_b_kernel<T, W, PX, SW><<<w.nb * SCB_WGS, SCB_THREADS, lds_b, s>>>(
                (const T*)w.staging, w.bcursor, w.cap_bg, g, w.tpb, w.nb, w.fill64, (T*)w.strays, w.scap, (T*)w.ovf, late_cap, w.late, dropped);
            return AST_OK;
        };
        using S3 = std::integral_constant<int, 3>;
        using S4 = std::integral_constant<int, 4>;
        const int rc = mass ? (plainx ? run(std::true_type{}, S4{}) : run(std::false_type{}, S4{}))
                            : (plainx ? run(std::true_type{}, S3{}) : run(std::false_type{}, S3{}));
        if (rc != AST_OK) return rc;
    }
        return AST_OK;
    };
    auto group_pass = [&](size_t pb, size_t pe, uint32_t closed_lo, uint32_t closed_n, uint32_t closed_mod = 0u) {
        AST_PROF("paint_tiled.fill", s);
        const size_t nint = (pe - pb + per_interval - 1) / per_interval;
        static const unsigned xcd_map = getenv("AST_PAINT_GROUP_XCD") ? (unsigned)atoi(getenv("AST_PAINT_GROUP_XCD")) : 0u;
        unsigned gg = (unsigned)(nint > want ? want : nint);
        if (xcd_map && nint <= want && nint >= 64) gg = (gg + 7u) / 8u * 8u;          // (the map needs a multiple of 8; the surplus workgroups find no interval)
        if (plainx)
            tile_group_kernel<T, W, true><<<gg, 256, 0, s>>>(pos, mass, pb, pe, g, w.fill64, w.recs, w.rcap, (T*)w.strays, w.scap,
                                                             w.ovf, w.ovf_count, w.col_flags, dropped, closed_lo, closed_n, 0u, xcd_map);
        else
            tile_group_kernel<T, W, false><<<gg, 256, 0, s>>>(pos, mass, pb, pe, g, w.fill64, w.recs, w.rcap, (T*)w.strays, w.scap,
                                                              w.ovf, w.ovf_count, w.col_flags, dropped, closed_lo, closed_n, closed_mod, xcd_map);
    };
    if (sel.stage != AST_PAINT_STAGE_ALL) {
        // the single-pass overwrite paint in three parts, so that a caller can interleave tile rows with what consumes
        // them (the slab pipeline: walk -> fold -> transform -> send, plane range by plane range)
        if (two_pass || !overwrite || (flags & (AST_PAINT_DEFER_FOLD | AST_PAINT_XSORTED))) {
            ast::set_error("ast_paint_tiled_stage: needs AST_PAINT_OVERWRITE without TWO_PASS / DEFER_FOLD / XSORTED");
            return AST_ERR_ARG;
        }
        if (sel.stage == AST_PAINT_STAGE_RESET) return AST_OK;                 // (the counters, above)
        if (sel.stage == AST_PAINT_STAGE_GROUP) {
            if (w.tpb) { const int rc = scatter_pass(); if (rc != AST_OK) return rc; }
            else group_pass(0, np, 0u, 0u);
            AST_CHECK_LAUNCH();
            return AST_OK;
        }
        if (sel.stage == AST_PAINT_STAGE_GROUP_PART) {
            // part row0 of nrows equal parts of the particle array (boundaries on whole grouping intervals, as in the
            // x-sorted pipeline); tile rows [closed_row0, closed_row0 + closed_nrows) (mod ntx) have been walked already
            // (nrows = parts | (span - 1) << 16: `span` CONSECUTIVE parts, row0 .. row0 + span - 1, in one launch - stages of
            // unequal size out of equal parts)
            const int k = sel.row0, K = sel.nrows & 0xffff, span = (sel.nrows >> 16) + 1;
            if (w.tpb || plainx || K < 1 || k < 0 || span < 1 || k + span > K || sel.closed_nrows < 0 || sel.closed_nrows > g.ntx ||
                sel.closed_row0 < 0 || sel.closed_row0 >= g.ntx) {
                ast::set_error("ast_paint_tiled_stage: GROUP_PART needs a slab buffer (not the scattered path), 0 <= part, part + span <= parts and closed rows inside the buffer");
                return AST_ERR_ARG;
            }
            const size_t pb = (size_t)((double)k / K * (double)np) / per_interval * per_interval;
            const size_t pe = k + span == K ? np : (size_t)((double)(k + span) / K * (double)np) / per_interval * per_interval;
            const uint32_t rs = (uint32_t)(g.nty * g.ntz);             // tiles per row
            if (pe > pb) group_pass(pb, pe, (uint32_t)sel.closed_row0 * rs, (uint32_t)sel.closed_nrows * rs, ntiles);
            AST_CHECK_LAUNCH();
            return AST_OK;
        }
        if (sel.row0 < 0 || sel.nrows < 1 || sel.row0 + sel.nrows > g.ntx) {
            ast::set_error("ast_paint_tiled_stage: tile rows [%d, %d) outside [0, %d)", sel.row0, sel.row0 + sel.nrows, g.ntx);
            return AST_ERR_ARG;
        }
        const int col0 = sel.row0 * g.nty;
        const unsigned ncol = (unsigned)(sel.nrows * g.nty);
        if (sel.stage == AST_PAINT_STAGE_WALK) {
            walk_pass(nullptr, nullptr, col0, ncol, s);
            if (plan.nseg > 1) {
                AST_PROF("paint_tiled.seams", s);
                z_seam_kernel<T, W><<<ncol, 256, 0, s>>>(w.zrec, g, scale, (uint32_t)std::min<unsigned long long>(5ull * w.cap, 0x3fffffffull),
                                                          mass ? mass_bound : 1.0, grid, (T*)w.rec, offset, dropped, col0, plan.nseg);
            }
        } else if (sel.stage == AST_PAINT_STAGE_FOLD || sel.stage == AST_PAINT_STAGE_LATE) {
            if (sel.stage == AST_PAINT_STAGE_FOLD) {
                AST_PROF("paint_tiled.fold", s);
                column_fold_kernel<T, W><<<ncol, 256, 0, s>>>((const T*)w.rec, g, grid, col0);
            }
            // the overflow / late list's deposits into these rows' planes (the lists are complete after the GROUP stage)
            AST_PROF("paint_tiled.overflow", s);
            const int x_lo = sel.row0 * TX, x_hi = std::min(g.nx_alloc, (sel.row0 + sel.nrows) * TX);
            if (w.tpb)
                late_deposit_kernel<T, W><<<128, 256, 0, s>>>((const T*)w.ovf, w.late, np / 4 * sizeof(uint32_t) / sizeof(T), g, scale, grid, dropped, x_lo, x_hi);
            else
                overflow_deposit_kernel<T, W><<<128, 256, 0, s>>>(pos, mass, w.ovf, w.ovf_count, g, scale, grid, dropped, x_lo, x_hi);
        } else {
            ast::set_error("ast_paint_tiled_stage: unknown stage %d", sel.stage);
            return AST_ERR_ARG;
        }
        AST_CHECK_LAUNCH();
        return AST_OK;
    }
    if (two_pass) {
        {
            AST_PROF("paint_tiled.count", s);
            index_pass(std::integral_constant<int, 0>{}, w.tile_count, nullptr, nullptr, nullptr, 0, nullptr, nullptr, dropped);
        }
        const uint32_t nblk = (ntiles + 1023) / 1024;
        {
            AST_PROF("paint_tiled.scan", s);
            scan_blocks_kernel<<<nblk, 256, 0, s>>>(w.tile_count, w.tile_off, w.block_sums, ntiles);
            scan_sums_kernel<<<1, 256, 0, s>>>(w.block_sums, nblk);
            scan_add_kernel<<<nblk, 256, 0, s>>>(w.tile_off, w.block_sums, ntiles);
        }
        {
            AST_PROF("paint_tiled.fill", s);
            index_pass(std::integral_constant<int, 1>{}, nullptr, w.tile_off, w.tile_fill, w.index, 0, nullptr, nullptr, nullptr);
        }
        deposit_pass(w.tile_off, w.tile_count, 0);
    } else if (overwrite && w.tpb) {
        { const int rc = scatter_pass(); if (rc != AST_OK) return rc; }
        deposit_pass(nullptr, nullptr, 0);
        AST_PROF("paint_tiled.overflow", s);
        // The late list (records whose tile - or bucket - segment was full).  A few stragglers: global atomics.  A long list
        // (clustered input without order in memory: a tenth of the particles, in a few hundred hot tiles) is a small paint
        // of its own - counted per tile, scanned, its record numbers filled tile-major, then ONE workgroup per tile adds its
        // records up in LDS and flushes the tile onto the grid: the two-pass variant with the list's records as particles.
        // The list's length stays on the device: all kernels are launched, each looks at the count.
        const unsigned long long late_cap = np / 4 * sizeof(uint32_t) / sizeof(T);
        const unsigned long long lds_min = getenv("AST_PAINT_LATE_LDS_MIN") ? strtoull(getenv("AST_PAINT_LATE_LDS_MIN"), nullptr, 10) : 262144ull;
        late_deposit_kernel<T, W><<<1024, 256, 0, s>>>((const T*)w.ovf, w.late, late_cap, g, scale, grid, dropped, 0, g.nx_alloc, lds_min);
        if (lds_min != ~0ull && late_cap > 0) {
            const size_t lint = (late_cap + per_interval - 1) / per_interval;
            const unsigned gl = (unsigned)(lint > 4096 ? 4096 : lint);
            const uint32_t nblk = (ntiles + 1023) / 1024;
            auto late_index_pass = [&](auto mode) {
                constexpr int MODE = decltype(mode)::value;
                if (plainx)
                    tile_index_kernel<T, W, MODE, true, 4><<<gl, 256, 0, s>>>((const T*)w.ovf, (size_t)late_cap, g, w.tile_count, w.tile_off, w.tile_fill,
                                                                              w.late_index, 0, nullptr, nullptr, w.col_flags, nullptr, w.late, lds_min);
                else
                    tile_index_kernel<T, W, MODE, false, 4><<<gl, 256, 0, s>>>((const T*)w.ovf, (size_t)late_cap, g, w.tile_count, w.tile_off, w.tile_fill,
                                                                               w.late_index, 0, nullptr, nullptr, w.col_flags, nullptr, w.late, lds_min);
            };
            late_index_pass(std::integral_constant<int, 0>{});
            scan_blocks_kernel<<<nblk, 256, 0, s>>>(w.tile_count, w.tile_off, w.block_sums, ntiles);
            scan_sums_kernel<<<1, 256, 0, s>>>(w.block_sums, nblk);
            scan_add_kernel<<<nblk, 256, 0, s>>>(w.tile_off, w.block_sums, ntiles);
            late_index_pass(std::integral_constant<int, 1>{});
            tile_deposit_kernel<T, W, 4><<<ntiles < 8192u ? ntiles : 8192u, 256, 0, s>>>((const T*)w.ovf, nullptr, g, scale, w.late_index, w.tile_off,
                                                                                       w.tile_count, 0, grid, dropped, ntiles);
        }
    } else if (overwrite) {
        // AST_PAINT_XSORTED: the particles come in ascending x (buffer planes).  They are grouped chunk by chunk
        // (plan.chunk_planes planes' worth each); a tile row is walked as soon as the chunks cover its planes plus a
        // margin - while the chunk's positions are still in the Infinity Cache, so the walk's gather does not go to
        // HBM a second time.  A particle that arrives for a row already handed to the walk ("late": the input was not
        // sorted after all) goes to the overflow list and is deposited with global atomics at the end - the result
        // never depends on the assumption, only the speed.  A periodic grid's row 0 also takes the wrap of the LAST
        // particles: it is walked last.  plan.streams == 2: the walks run on a second stream, each beside the grouping
        // of the following chunks (which wait for the walk before the one they overlap: the grouping must not run
        // ahead, or the positions are gone from the cache before they are gathered).
        const int K = plan.chunks;
        if (K == 1) {
            group_pass(0, np, 0u, 0u);
            deposit_pass(nullptr, nullptr, 0);
        } else {
            const bool two_streams = plan.streams > 1;
            SidePipe* sp = two_streams ? side_pipe(2 * g.ntx + 4) : nullptr;
            if (two_streams && !sp) { ast::set_error("ast_paint_tiled: no side stream"); return AST_ERR_HIP; }
            AST_PROF("paint_tiled.pipeline", s);
            // two host threads painting on one device share the side stream and its events: the pipeline is serialised
            // per device; and whatever happens in the loop, `s` is joined with the side stream before this
            // function returns - the caller may free or reuse the workspace on `s` right after
            std::unique_lock<std::mutex> pipe_lock;
            if (two_streams) pipe_lock = std::unique_lock<std::mutex>(sp->run);
            const bool x_periodic = g.nx_alloc == g.n;
            const int m0 = x_periodic ? 1 : 0;                         // rows [0, m0) are held back to the end
            const uint32_t rs = (uint32_t)(g.nty * g.ntz);             // tiles per row
            int prev = m0, nev = 0;
            hipEvent_t walk_done[2] = {nullptr, nullptr};              // the last two walks on the side stream
            auto chunk_loop = [&]() -> int {
            for (int k = 0; k < K; ++k) {
                const size_t pb = (size_t)((double)k / K * (double)np) / per_interval * per_interval;
                const size_t pe = k + 1 == K ? np : (size_t)((double)(k + 1) / K * (double)np) / per_interval * per_interval;
                if (two_streams && walk_done[1]) AST_CHECK_HIP(hipStreamWaitEvent(s, walk_done[1], 0));
                if (pe > pb) group_pass(pb, pe, (uint32_t)m0 * rs, (uint32_t)(prev - m0) * rs);
                // rows whose planes (plus the margin) the chunks 0..k cover
                const int covered = (int)((long long)(k + 1) * g.nx_alloc / K) - plan.margin_planes;
                const int r = k + 1 == K ? g.ntx : std::max(prev, std::min(g.ntx, covered / TX));
                if (r > prev || k + 1 == K) {
                    const int rows = r - prev + (k + 1 == K ? m0 : 0);      // the last launch wraps around to row 0
                    hipStream_t ws = s;
                    if (two_streams) {
                        AST_CHECK_HIP(hipEventRecord(sp->ev[nev], s));
                        AST_CHECK_HIP(hipStreamWaitEvent(sp->side, sp->ev[nev], 0));
                        ++nev;
                        ws = sp->side;
                    }
                    if (rows > 0) walk_pass(nullptr, nullptr, prev * g.nty, (unsigned)(rows * g.nty), ws);
                    if (two_streams) {
                        AST_CHECK_HIP(hipEventRecord(sp->ev[nev], sp->side));
                        walk_done[1] = walk_done[0];
                        walk_done[0] = sp->ev[nev];
                        ++nev;
                    }
                    prev = r;
                }
            }
            return AST_OK;
            };
            const int loop_rc = chunk_loop();
            if (two_streams) {
                if (loop_rc != AST_OK) {
                    // an enqueue failed half way: whatever already runs on the side stream still reads the workspace
                    (void)hipStreamSynchronize(sp->side);
                    return loop_rc;
                }
                if (walk_done[0]) AST_CHECK_HIP(hipStreamWaitEvent(s, walk_done[0], 0));
            } else if (loop_rc != AST_OK) {
                return loop_rc;
            }
            fold_pass();
        }
        AST_PROF("paint_tiled.overflow", s);
        overflow_deposit_kernel<T, W><<<1024, 256, 0, s>>>(pos, mass, w.ovf, w.ovf_count, g, scale, grid, dropped, 0, g.nx_alloc);
    } else {
        {
            AST_PROF("paint_tiled.fill", s);
            index_pass(std::integral_constant<int, 2>{}, nullptr, nullptr, w.tile_fill, w.index, w.cap, w.ovf, w.ovf_count, dropped);
        }
        deposit_pass(nullptr, w.tile_fill, w.cap);
#ifdef PAINT_WALK_TEST
        {
            AST_PROF("paint_tiled.walktest", s);
            walk_test_kernel<T><<<ncols, 256, 0, s>>>(pos, g, w.index, w.tile_fill, w.cap, (T*)w.rec);
        }
#endif
        AST_PROF("paint_tiled.overflow", s);
        overflow_deposit_kernel<T, W><<<1024, 256, 0, s>>>(pos, mass, w.ovf, w.ovf_count, g, scale, grid, dropped, 0, g.nx_alloc);
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}

}  // namespace

#ifdef PAINT_STAMPS
extern "C" int ast_debug_stamps(unsigned long long* stamps_host, unsigned* counts_host) {
    AST_CHECK_HIP(hipDeviceSynchronize());
    AST_CHECK_HIP(hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 4096));
    AST_CHECK_HIP(hipMemcpyFromSymbol(counts_host, HIP_SYMBOL(g_nstamp), sizeof(unsigned) * 64));
    unsigned zero[64] = {0};
    AST_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_nstamp), zero, sizeof(zero)));
    return AST_OK;
}
#endif

extern "C" size_t ast_paint_tiled_workspace_bytes(int window, int dtype, size_t np, int nmesh, int nx_alloc, int flags) {
    TileGeom g;
    uint32_t ntiles = 0;
    if (nmesh <= 0 || nx_alloc <= 0 || !tiled_geometry(nmesh, nx_alloc, g, ntiles)) return 0;
    return carve(nullptr, np, ntiles, (uint32_t)(g.ntx * g.nty), flags, dtype == AST_F32 ? 4 : 8,
                 record_bytes(window, g, dtype == AST_F32 ? 4 : 8, flags),
                 seam_bytes(window, g, np, dtype == AST_F32 ? 4 : 8, flags)).bytes;
}

// Where ast_paint_tiled(... AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD) left the halo records of a paint with
// these parameters (inside `workspace`), for ast_fft_tile_power_3d_halo.
extern "C" int ast_paint_tiled_halo(void* workspace, int window, int dtype, size_t np, int nmesh, int nx_alloc, int flags,
                                    void** rec_out) {
    AST_CHECK_ARG(workspace && rec_out && (flags & AST_PAINT_OVERWRITE));
    AST_CHECK_ARG(window == AST_WIN_CIC || window == AST_WIN_TSC);
    TileGeom g;
    uint32_t ntiles = 0;
    AST_CHECK_ARG(nmesh > 0 && nx_alloc > 0 && tiled_geometry(nmesh, nx_alloc, g, ntiles));
    const size_t esz = dtype == AST_F32 ? 4 : 8;
    *rec_out = carve(workspace, np, ntiles, (uint32_t)(g.ntx * g.nty), flags, esz, record_bytes(window, g, esz, flags)).rec;
    return AST_OK;
}

// out[0] = group records kept, out[1] = stray copies kept, out[2] = particles on the overflow list,
// out[3] = largest number of strays any tile asked for
__global__ void __launch_bounds__(256)
list_stats_kernel(const unsigned long long* __restrict__ fill64, uint32_t ntiles, uint32_t rcap, uint32_t scap,
                  const unsigned long long* __restrict__ ovf_count, unsigned long long* out) {
    unsigned long long nr = 0, ns = 0, smax = 0;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
        const unsigned long long f = fill64[t];
        nr += min((uint32_t)(f >> 32), rcap);
        ns += min((uint32_t)f, scap);
        smax = max(smax, (unsigned long long)(uint32_t)f);
    }
    atomicAdd(&out[0], nr);
    atomicAdd(&out[1], ns);
    atomicMax(&out[3], smax);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = *ovf_count;
}

extern "C" int ast_paint_tiled_list_stats(void* workspace, int window, int dtype, size_t np, int nmesh, int nx_alloc,
                                          int flags, unsigned long long* out, void* stream) {
    AST_CHECK_ARG(workspace && out && (flags & AST_PAINT_OVERWRITE) && !(flags & AST_PAINT_TWO_PASS));
    TileGeom g;
    uint32_t ntiles = 0;
    AST_CHECK_ARG(nmesh > 0 && nx_alloc > 0 && tiled_geometry(nmesh, nx_alloc, g, ntiles));
    const size_t esz = dtype == AST_F32 ? 4 : 8;
    const Workspace w = carve(workspace, np, ntiles, (uint32_t)(g.ntx * g.nty), flags, esz, record_bytes(window, g, esz, flags));
    hipStream_t s = ast::as_stream(stream);
    AST_CHECK_HIP(hipMemsetAsync(out, 0, 4 * sizeof(unsigned long long), s));
    list_stats_kernel<<<256, 256, 0, s>>>(w.fill64, ntiles, w.rcap, w.scap, w.tpb ? w.late : w.ovf_count, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ---- order probe: does the input have spatial order in memory? ----
// One 64-lane workgroup per run of 32 consecutive particles (lanes 32-63 idle): the tile (TX x TY x TZ cells) of every
// particle's cell, a ballot against the tile of the run's 16th particle; a run with >= 8 takers is what the grouping
// kernel turns into a group record.  *groupable += 1 for such a run.  (A heuristic on plain floor(x n / L + shift): the
// window only shifts the base cell by one.)
template <typename T>
__global__ void __launch_bounds__(64)
order_probe_kernel(const T* __restrict__ pos, size_t np, int n, double inv_dx, double shift, int windows,
                   unsigned* __restrict__ groupable) {
    const size_t w = blockIdx.x;
    // run starts as device.sample_run_starts: evenly spread, multiples of 32, at most np - 32 (64-bit integers)
    size_t start = (w * (np - 32) / (size_t)(windows - 1)) / 32 * 32;
    if (start > np - 32) start = np - 32;
    const int lane = threadIdx.x;
    unsigned long long key = ~0ull;
    if (lane < 32) {
        const T* p = pos + 3 * (start + lane);
        long long c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double sc = floor(__fma_rn((double)p[a], inv_dx, shift));
            long long ci = (long long)fmod(sc, (double)n);       // (|sc| < 2^53: exact)
            if (ci < 0) ci += n;
            c[a] = ci;
        }
        key = ((unsigned long long)(c[0] / ast::TX) * (unsigned long long)n + (unsigned long long)(c[1] / ast::TY)) * (unsigned long long)n +
              (unsigned long long)(c[2] / ast::TZ);
    }
    const unsigned long long k15 = __shfl(key, 15, 64);
    const unsigned long long same = __ballot(lane < 32 && key == k15);
    if (lane == 0 && __popcll(same) >= 8) atomicAdd(groupable, 1u);
}

extern "C" int ast_paint_order_probe(const void* pos, int dtype, size_t np, int nmesh, double boxsize, double shift_cells,
                                     int windows, unsigned* groupable, void* stream) {
    AST_CHECK_ARG(pos != nullptr && groupable != nullptr && (dtype == AST_F32 || dtype == AST_F64));
    AST_CHECK_ARG(np >= 32 && nmesh > 0 && boxsize > 0.0 && windows >= 2 && windows <= 65536);
    hipStream_t s = ast::as_stream(stream);
    AST_CHECK_HIP(hipMemsetAsync(groupable, 0, sizeof(unsigned), s));
    const double inv_dx = (double)nmesh / boxsize;
    if (dtype == AST_F32) order_probe_kernel<float><<<windows, 64, 0, s>>>((const float*)pos, np, nmesh, inv_dx, shift_cells, windows, groupable);
    else order_probe_kernel<double><<<windows, 64, 0, s>>>((const double*)pos, np, nmesh, inv_dx, shift_cells, windows, groupable);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ---- occupancy probe: will the single-pass paint's fixed tile segments overflow? ----
// The single pass gives every tile room for max(2 mean, mean + 512) particles (tile_capacity); what does not fit goes through a
// global-atomic list - fine for a few stragglers, a crawl for clustered input (evolved snapshots, halo catalogues:
// stats_subfind.py:125-131 paints exactly such a set).  Finding that out from the overflow list costs a wasted paint, and only
// callers that synchronise ever looked.  This probe estimates it from a SAMPLE before anything is painted: `samples`
// particles, one from every stride of np / samples (position inside the stride from a hash of the stride's index, so that a
// lattice-ordered file is not sampled on a sub-lattice), counted per tile (TX x TY x TZ cells of the periodic grid) in
// counts_d (ntiles x 4 B, zeroed here); then per tile the sample count is scaled by np / samples and compared with the
// capacity.  out_d[0] = estimated particles beyond capacity, out_d[1] = largest estimated tile occupancy.  With samples >= 8
// per tile on average the Poisson noise of a uniform input stays below 0.1 % of np; callers compare with np / 64, the
// level at which device.paint used to repeat the paint.
__device__ inline uint64_t probe_mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ void __launch_bounds__(256)
occupancy_sample_kernel(const T* __restrict__ pos, size_t np, int n, double inv_dx, double shift, size_t samples,
                        unsigned* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= samples) return;
    const size_t lo = (size_t)(((unsigned __int128)i * np) / samples), hi = (size_t)(((unsigned __int128)(i + 1) * np) / samples);
    const size_t j = lo + (hi > lo ? probe_mix(i) % (hi - lo) : 0);
    const T* p = pos + 3 * j;
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double sc = floor(__fma_rn((double)p[a], inv_dx, shift));
        long long ci = (long long)fmod(sc, (double)n);
        if (ci < 0) ci += n;
        c[a] = (unsigned)ci;
    }
    const unsigned tile = ((c[0] / ast::TX) * (unsigned)(n / ast::TY) + c[1] / ast::TY) * (unsigned)(n / ast::TZ) + c[2] / ast::TZ;
    atomicAdd(&counts[tile], 1u);
}
__global__ void __launch_bounds__(256)
occupancy_excess_kernel(const unsigned* __restrict__ counts, unsigned ntiles, double per_sample, double cap,
                        unsigned long long* __restrict__ out) {
    __shared__ double ex[256];
    __shared__ double mx[256];
    double e = 0.0, m = 0.0;
    for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
        const double occ = (double)counts[t] * per_sample;
        if (occ > cap) e += occ - cap;
        m = occ > m ? occ : m;
    }
    ex[threadIdx.x] = e;
    mx[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            ex[threadIdx.x] += ex[threadIdx.x + st];
            mx[threadIdx.x] = mx[threadIdx.x + st] > mx[threadIdx.x] ? mx[threadIdx.x + st] : mx[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], (unsigned long long)(ex[0] + 0.5));
        atomicMax(&out[1], (unsigned long long)(mx[0] + 0.5));
    }
}

extern "C" size_t ast_paint_occupancy_probe_bytes(int nmesh) {
    if (nmesh <= 0 || nmesh % TY || nmesh % TZ || nmesh % TX) return 0;
    return (size_t)(nmesh / TX) * (nmesh / TY) * (nmesh / TZ) * sizeof(unsigned);
}

extern "C" int ast_paint_occupancy_probe(const void* pos, int dtype, size_t np, int nmesh, double boxsize, double shift_cells,
                                         size_t samples, void* counts, size_t counts_bytes, unsigned long long* out, void* stream) {
    AST_CHECK_ARG(pos != nullptr && counts != nullptr && out != nullptr && (dtype == AST_F32 || dtype == AST_F64));
    AST_CHECK_ARG(np >= 1 && boxsize > 0.0 && samples >= 1 && samples <= np && samples < (1ull << 40));
    const size_t need = ast_paint_occupancy_probe_bytes(nmesh);
    AST_CHECK_ARG(need != 0 && counts_bytes >= need);
    const unsigned ntiles = (unsigned)(need / sizeof(unsigned));
    hipStream_t s = ast::as_stream(stream);
    AST_CHECK_HIP(hipMemsetAsync(counts, 0, need, s));
    AST_CHECK_HIP(hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), s));
    const double inv_dx = (double)nmesh / boxsize;
    const unsigned blocks = (unsigned)((samples + 255) / 256);
    if (dtype == AST_F32) occupancy_sample_kernel<float><<<blocks, 256, 0, s>>>((const float*)pos, np, nmesh, inv_dx, shift_cells, samples, (unsigned*)counts);
    else occupancy_sample_kernel<double><<<blocks, 256, 0, s>>>((const double*)pos, np, nmesh, inv_dx, shift_cells, samples, (unsigned*)counts);
    occupancy_excess_kernel<<<std::min(1024u, (ntiles + 255) / 256), 256, 0, s>>>((const unsigned*)counts, ntiles, (double)np / (double)samples,
                                                                                 (double)tile_capacity(np, ntiles), out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

static int paint_tiled_impl(int window, int dtype, const void* pos, const void* mass, size_t np, int nmesh,
                            double boxsize, double scale, int x_start, int nx_alloc, void* grid,
                            void* workspace, size_t workspace_bytes, unsigned long long* dropped,
                            int flags, double mass_bound, double offset, int offset_start, int offset_count,
                            double shift_cells, void* stream, StageSel sel) {
    AST_CHECK_ARG(window == AST_WIN_CIC || window == AST_WIN_TSC);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nmesh > 0 && boxsize > 0.0);
    AST_CHECK_ARG(x_start >= 0 && x_start < nmesh && nx_alloc > 0 && nx_alloc <= nmesh);
    AST_CHECK_ARG(grid != nullptr);
    AST_CHECK_ARG(np < 0xffffffffull - 64);
    if (np == 0) {
        if ((flags & AST_PAINT_OVERWRITE) && (sel.stage == AST_PAINT_STAGE_ALL || sel.stage == AST_PAINT_STAGE_GROUP))
            AST_CHECK_HIP(hipMemsetAsync(grid, 0, (size_t)nx_alloc * nmesh * nmesh * (dtype == AST_F32 ? 4 : 8),
                                         ast::as_stream(stream)));
        return AST_OK;
    }
    AST_CHECK_ARG(pos != nullptr && workspace != nullptr);
    AST_CHECK_ARG(!(flags & AST_PAINT_OVERWRITE) || mass == nullptr || mass_bound > 0.0);
    AST_CHECK_ARG(!(flags & AST_PAINT_DEFER_FOLD) || ((flags & AST_PAINT_OVERWRITE) && x_start == 0 && nx_alloc == nmesh));
    AST_CHECK_ARG(offset == 0.0 || (flags & AST_PAINT_OVERWRITE));
    AST_CHECK_ARG(offset_start >= 0 && offset_start <= nx_alloc && (offset_count < 0 || offset_start + offset_count <= nx_alloc));
    TileGeom g;
    uint32_t ntiles = 0;
    if (!tiled_geometry(nmesh, nx_alloc, g, ntiles)) {
        ast::set_error("ast_paint_tiled: nmesh must be a multiple of %d", TZ);
        return AST_ERR_ARG;
    }
    g.x_start = x_start;
    g.inv_dx = (double)nmesh / boxsize;
    g.shift = shift_cells;
    g.off_lo = offset_start;
    g.off_hi = offset_start + (offset_count < 0 ? nx_alloc : offset_count);
    const size_t need = carve(nullptr, np, ntiles, (uint32_t)(g.ntx * g.nty), flags, dtype == AST_F32 ? 4 : 8,
                              record_bytes(window, g, dtype == AST_F32 ? 4 : 8, flags),
                              seam_bytes(window, g, np, dtype == AST_F32 ? 4 : 8, flags)).bytes;
    if (workspace_bytes < need) {
        ast::set_error("ast_paint_tiled: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
        return AST_ERR_WORKSPACE;
    }
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32) {
        if (window == AST_WIN_CIC)
            return run_tiled<float, 2>((const float*)pos, (const float*)mass, np, g, ntiles, scale, (float*)grid, workspace, dropped, flags, mass_bound, offset, s, sel);
        return run_tiled<float, 3>((const float*)pos, (const float*)mass, np, g, ntiles, scale, (float*)grid, workspace, dropped, flags, mass_bound, offset, s, sel);
    }
    if (window == AST_WIN_CIC)
        return run_tiled<double, 2>((const double*)pos, (const double*)mass, np, g, ntiles, scale, (double*)grid, workspace, dropped, flags, mass_bound, offset, s, sel);
    return run_tiled<double, 3>((const double*)pos, (const double*)mass, np, g, ntiles, scale, (double*)grid, workspace, dropped, flags, mass_bound, offset, s, sel);
}

extern "C" int ast_paint_tiled(int window, int dtype, const void* pos, const void* mass, size_t np, int nmesh,
                               double boxsize, double scale, int x_start, int nx_alloc, void* grid,
                               void* workspace, size_t workspace_bytes, unsigned long long* dropped,
                               int flags, double mass_bound, double offset, int offset_start, int offset_count,
                               double shift_cells, void* stream) {
    return paint_tiled_impl(window, dtype, pos, mass, np, nmesh, boxsize, scale, x_start, nx_alloc, grid, workspace,
                            workspace_bytes, dropped, flags, mass_bound, offset, offset_start, offset_count, shift_cells,
                            stream, StageSel{});
}

// One part of the single-pass overwrite paint (see include/astrild_hip.h): GROUP once, then WALK and FOLD per range of
// tile rows, every call with the SAME arguments as the others.
extern "C" int ast_paint_tiled_stage(int window, int dtype, const void* pos, const void* mass, size_t np, int nmesh,
                                     double boxsize, double scale, int x_start, int nx_alloc, void* grid,
                                     void* workspace, size_t workspace_bytes, unsigned long long* dropped,
                                     int flags, double mass_bound, double offset, int offset_start, int offset_count,
                                     double shift_cells, int stage, int row0, int nrows, int closed_row0, int closed_nrows,
                                     void* stream) {
    AST_CHECK_ARG(stage == AST_PAINT_STAGE_GROUP || stage == AST_PAINT_STAGE_WALK || stage == AST_PAINT_STAGE_FOLD ||
                  stage == AST_PAINT_STAGE_GROUP_PART || stage == AST_PAINT_STAGE_RESET || stage == AST_PAINT_STAGE_LATE);
    StageSel sel;
    sel.stage = stage;
    sel.row0 = row0;
    sel.nrows = nrows;
    sel.closed_row0 = closed_row0;
    sel.closed_nrows = closed_nrows;
    return paint_tiled_impl(window, dtype, pos, mass, np, nmesh, boxsize, scale, x_start, nx_alloc, grid, workspace,
                            workspace_bytes, dropped, flags, mass_bound, offset, offset_start, offset_count, shift_cells,
                            stream, sel);
}

extern "C" int ast_paint_tile_rows(int nx_alloc) { return (nx_alloc + TX - 1) / TX; }
extern "C" int ast_paint_tile_row_planes(void) { return TX; }
