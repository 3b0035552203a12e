// Device-side arithmetic shared by the registration kernels.
//
// Everything here is strict IEEE float32 evaluated in a fixed order, because
// comparisons between these values decide which point index wins: the file
// must be compiled with -ffp-contract=off and without fast-math, so that
// hipcc neither fuses a*b+c nor replaces sqrtf or '/' by approximations
// (both are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PCP_BIG 1.23e20f          // FunctionalBase.glsl:146,204
#define PCP_LIDAR_OFF_Y 0.127f    // 0.254 / 2.0, FunctionalBase.glsl:35-36
#define PCP_NVAL 27               // 21 upper-triangle JtJ entries + 6 Jte entries
#define PCP_PSTRIDE 32            // floats per partial record

#ifdef PCP_STATS
// Instrumented build only (make stats): [0] ring visits by lanes, [1] ring passes executed by waves,
// [2] wave searches, for hinted searches; [4..6] the same for searches over all rings.
__device__ unsigned long long pcp_stats[8];
// first_ring<K>: [0] waves, [1] loop rounds, [2] rounds in which some lane searched a ring, [3] ring searches by lanes,
// [4] queries, [8 + r] waves that ran r searching rounds (r = 0..6, 7 = more)
__device__ unsigned long long pcp_stats2[16];
#endif

namespace pcp {

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    s = s + a.w * b.w;
    return s;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    float s = ax * bx;
    s = s + ay * by;
    s = s + az * bz;
    return s;
}

// distsq, FunctionalBase.glsl:106-109
__device__ __forceinline__ float distsq(const float4 a, const float4 b) {
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    return dot4(d, d);
}

// projectSpherePoint / the normalisation inside mcos, FunctionalBase.glsl:34-48.
// origin is the 4-vector as uploaded ([ox,oy,oz,ox]); the offset is (0, 0.127, 0, 0).
__device__ __forceinline__ float4 unit_sphere(const float4 p, const float4 o) {
    float4 v = make_float4(p.x - o.x, (p.y - o.y) - PCP_LIDAR_OFF_Y, p.z - o.z, p.w - o.w);
    const float len = sqrtf(dot4(v, v));
    return make_float4(v.x / len, v.y / len, v.z / len, v.w / len);
}

#ifndef PCP_WIN_GROUP
#define PCP_WIN_GROUP 7           // window loads in flight per lane
#endif
#ifndef PCP_WRAP_GROUP
#define PCP_WRAP_GROUP 4          // the same for a window that wraps around its ring's end (each slot needs its own index)
#endif
#ifndef PCP_SCALAR_LEVELS
#define PCP_SCALAR_LEVELS 1       // search levels at which the whole wave stands at one place read it through the scalar cache
#endif
#ifndef PCP_AZ_BINS
#define PCP_AZ_BINS 128           // azimuth bins of the per-ring elevation and range tables.  With the range bands finer bins
                                  // pay in the batched hinted launches (64 / 128 / 256 bins: 2.34 / 2.24 / 2.21 ms per launch of
                                  // 256 pairs) and cost a single pair in flight, whose pre-pass waits for the larger table
                                  // (the driver chain: 0.462 / 0.475 ms per frame at 64 / 256 bins)
// 0.999 * (sin(0.99 * 2 pi / PCP_AZ_BINS) - 1e-5)^2, rounded down: squared sine of the smallest
// azimuth gap between a query and a target point two or more bins away, with the margins of
// ring_lower_bound (64 bins: 0.0094050f, 256: 0.00058921f).
#define PCP_AZ_FAR 0.0023564f
#endif

// Arrays the search gathers from are addressed as GLOBAL memory explicitly: pointers that reach a
// kernel through a descriptor in device memory are generic to the compiler, and generic (flat) loads
// can neither use the scalar-base addressing mode nor skip the aperture check.
#define PCP_GLOBAL __attribute__((address_space(1)))
typedef float pcp_f4 __attribute__((ext_vector_type(4)));
typedef const pcp_f4 PCP_GLOBAL* gf4_t;

__device__ __forceinline__ gf4_t as_global(const float4* p) { return (gf4_t)(const void*)p; }
__device__ __forceinline__ float4 ld4(const gf4_t p, const size_t i) {
    const pcp_f4 v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
// Element i >= 0 of an array of fewer than 2^28 elements: the byte offset is formed in 32 bits, which
// lets the load use the (uniform 64-bit base + 32-bit lane offset) addressing mode instead of a
// 64-bit address computed per lane.
__device__ __forceinline__ float4 ld4i(const gf4_t p, const int i) {
    const unsigned off = (unsigned)i << 4;
    const pcp_f4 v = *(gf4_t)((const char PCP_GLOBAL*)p + off);
    return make_float4(v.x, v.y, v.z, v.w);
}

struct Target {
    gf4_t a;                         // target points
    gf4_t ua;                        // unit_sphere(a[i])
    const int2* scan;                // (begin, end_inclusive) per ring, in LDS
    const float4* stat;              // per ring (sin, cos) of its lowest and highest elevation, in LDS
    gf4_t bins;                      // [azimuth bin][ring]: the same for the ring's points in bins b-1, b, b+1; behind these
                                     // PCP_AZ_BINS x Z entries as many with the range band (lowest, highest) of the same points
    int n_a;
    int Z;                           // lens_data.z
    int Zs;                          // rings [Zs, Z) are copies of ring Zs - 1 (the padding rows of the table)
    float4 o;
};

// binaryScanCircleSearch (FunctionalBase.glsl:69-87) followed by the +-CHECK
// window of binsearchAndCheck (:142-156) / closestDistinctOnLine
// (NormalShader.glsl:18-31).  Index arithmetic is the shader's modulo
// arithmetic with the one integer division hoisted: the absolute midpoint
// (begin+end)/2 is reduced mod L once, every later step stays inside (-L, 2L)
// and wraps with a compare.  exclude < 0 disables the 'test != pindex' clause.
// One level of binaryScanCircleSearch for a lane whose ring has L >= 2 points starting at beg.
__device__ __forceinline__ void search_level(const Target& T, const float4 ref, const int beg, const int L, int& idx, int& half,
                                             bool& together) {
    float c1, c2;
    if (__any(idx == 0 || idx + 1 == L)) {        // some lane of the wave is at an end of its ring
        const int i1 = (idx == 0) ? L - 1 : idx - 1;
        const int i2 = (idx + 1 == L) ? 0 : idx + 1;
        c1 = dot4(ld4i(T.ua, beg + i1), ref) + 1.0f;
        c2 = dot4(ld4i(T.ua, beg + i2), ref) + 1.0f;
    } else if (together && (together = __all(beg + idx == __builtin_amdgcn_readfirstlane(beg + idx)))) {
        // every lane of the wave stands at the same place of the same ring (the first levels of neighbouring queries on
        // one ring): the two neighbours are read once, through the scalar cache, instead of by a 64-lane gather.  Lanes
        // that have parted do not meet again: `together` (the same in all lanes) ends the asking.
        typedef const pcp_f4 __attribute__((address_space(4)))* cf4_t;
        const cf4_t m = (cf4_t)(const void*)((const char PCP_GLOBAL*)T.ua + ((size_t)(unsigned)__builtin_amdgcn_readfirstlane(beg + idx) << 4));
        const pcp_f4 v1 = m[-1], v2 = m[1];
        c1 = dot4(make_float4(v1.x, v1.y, v1.z, v1.w), ref) + 1.0f;
        asm volatile("" : "+v"(c1));
        c2 = dot4(make_float4(v2.x, v2.y, v2.z, v2.w), ref) + 1.0f;
    } else {                                      // both neighbours sit beside idx: one address
        const gf4_t m = (gf4_t)((const char PCP_GLOBAL*)T.ua + ((unsigned)(beg + idx) << 4));
        const pcp_f4 v1 = m[-1], v2 = m[1];
        c1 = dot4(make_float4(v1.x, v1.y, v1.z, v1.w), ref) + 1.0f;
        asm volatile("" : "+v"(c1));              // keeps the two sums from being interleaved lane-pairwise
        c2 = dot4(make_float4(v2.x, v2.y, v2.z, v2.w), ref) + 1.0f;
    }
    // idx -+ half lies in (-L, 2L): of t, t + L and t - L exactly one is in [0, L), and read as
    // unsigned numbers it is the smallest of the three
    const unsigned t = (unsigned)((c1 > c2) ? idx - half : idx + half);
    idx = (int)min(min(t, t + (unsigned)L), t - (unsigned)L);
    half = (half + 1) >> 1;
}

// The +-CHECK window around position idx of a ring of L >= 2 points (binsearchAndCheck :142-156 /
// closestDistinctOnLine, NormalShader.glsl:18-31); exclude < 0 disables the 'test != pindex' clause.
template <int CHECK, bool EXCL>
__device__ __forceinline__ void window_min(const Target& T, const float4 q, const int beg, const int L, const int idx,
                                           const int exclude, int& best, float& mind) {
    best = beg + idx;
    if (idx >= CHECK && idx + CHECK <= L) {
        // The window does not wrap (all but the lanes within CHECK points of the ring's ends): its
        // 2*CHECK points are consecutive, so one address serves them all and the winner is kept as
        // a slot number.  Same slots, same order, same strict '<' as the general walk below.
        const int first = beg + idx - CHECK;
        const gf4_t w = (gf4_t)((const char PCP_GLOBAL*)T.a + ((unsigned)first << 4));
        const int skip = EXCL ? exclude - first : -1;
        int slot = -1;
        // PCP_WIN_GROUP loads are issued together before their distances are evaluated: the window
        // then costs ceil(2*CHECK / PCP_WIN_GROUP) memory round trips of the wave instead of one per
        // few slots, within the register budget of 8 resident waves per SIMD.
#pragma unroll
        for (int g = 0; g < 2 * CHECK; g += PCP_WIN_GROUP) {
            pcp_f4 p[PCP_WIN_GROUP];
#pragma unroll
            for (int i = 0; i < PCP_WIN_GROUP; ++i)
                if (g + i < 2 * CHECK) p[i] = w[g + i];
#pragma unroll
            for (int i = 0; i < PCP_WIN_GROUP; ++i)
                if (g + i < 2 * CHECK) asm volatile("" : "+v"(p[i]));      // all of the group's loads precede its arithmetic
#pragma unroll
            for (int i = 0; i < PCP_WIN_GROUP; ++i) {
                if (g + i < 2 * CHECK) {
                    const float d = distsq(q, make_float4(p[i].x, p[i].y, p[i].z, p[i].w));
                    if (d < mind && (!EXCL || g + i != skip)) { mind = d; slot = g + i; }
                }
            }
        }
        if (slot >= 0) best = first + slot;
    } else if (L >= CHECK) {
        // The window wraps around an end of the ring.  Few lanes come here, but the whole wave waits for them (the
        // rings all begin at the same azimuth, so the waves that straddle it pay this in every ring they visit): the
        // loads are grouped exactly as above, each slot's index with its own wrap.
        auto slot_index = [&](const int k) {
            int t = idx + k - CHECK;
            t = (t < 0) ? t + L : ((t >= L) ? t - L : t);
            return beg + t;
        };
#pragma unroll
        for (int g = 0; g < 2 * CHECK; g += PCP_WRAP_GROUP) {
            pcp_f4 p[PCP_WRAP_GROUP];
#pragma unroll
            for (int i = 0; i < PCP_WRAP_GROUP; ++i)
                if (g + i < 2 * CHECK) p[i] = *(gf4_t)((const char PCP_GLOBAL*)T.a + ((unsigned)slot_index(g + i) << 4));
#pragma unroll
            for (int i = 0; i < PCP_WRAP_GROUP; ++i)
                if (g + i < 2 * CHECK) asm volatile("" : "+v"(p[i]));
#pragma unroll
            for (int i = 0; i < PCP_WRAP_GROUP; ++i) {
                if (g + i < 2 * CHECK) {
                    const int test = slot_index(g + i);
                    const float d = distsq(q, make_float4(p[i].x, p[i].y, p[i].z, p[i].w));
                    if (d < mind && (!EXCL || test != exclude)) { mind = d; best = test; }
                }
            }
        }
    } else {
        for (int i = -CHECK; i < CHECK; ++i) {
            int t = (idx + i + L) % L;
            if (t < 0) t += L;
            const int test = beg + t;
            const float d = distsq(q, ld4i(T.a, test));
            if (d < mind && (!EXCL || test != exclude)) { mind = d; best = test; }
        }
    }
}

// A ring of a single point: runs == 0, the search returns begin + (begin+end)/2 untouched, and all
// 2*CHECK window slots are the ring's single point.
template <bool EXCL>
__device__ __forceinline__ void single_point_ring(const Target& T, const float4 q, const int beg, const int end,
                                                  const int exclude, int& best, float& mind) {
    const int closest = beg + (beg + end) / 2;
    best = min(closest, T.n_a - 1);
    const float d = distsq(q, ld4i(T.a, beg));
    if (d < mind && (!EXCL || beg != exclude)) { mind = d; best = beg; }
}

template <int CHECK, bool EXCL>
__device__ __forceinline__ void ring_window(const Target& T, const float4 q, const float4 ref,
                                            const int beg, const int end, const int exclude,
                                            int& best, float& mind) {
    const int L = end - beg + 1;
    mind = PCP_BIG;
    if (L == 1) {
        single_point_ring<EXCL>(T, q, beg, end, exclude, best, mind);
        return;
    }
    int idx = ((beg + end) / 2) % L;
    int half = L / 2;
    const int runs = 32 - __clz(L - 1);           // exact ceil(log2(L)), L >= 2
    bool together = PCP_SCALAR_LEVELS != 0;
    for (int i = 0; i < runs; ++i) search_level(T, ref, beg, L, idx, half, together);
    window_min<CHECK, EXCL>(T, q, beg, L, idx, exclude, best, mind);
}

// ring_window on N rings for the same query at once: the N binary searches advance level by level
// together in straight-line code, so their gathers are in flight together and the dependent chain is
// that of the longest search instead of the sum; the windows follow one after the other.  Per ring the
// same loads feed the same arithmetic as in ring_window, hence the same result.  live[k] == false
// skips ring k (best[k] = -1); its (begin, end) must still name a valid ring (its loads are issued and
// their results dropped, as are those of a lane whose search has fewer levels than the wave's longest).
template <int N, int CHECK, bool EXCL>
__device__ __forceinline__ void ring_windows(const Target& T, const float4 q, const float4 ref, const int2 (&se)[N],
                                             const bool (&live)[N], const int exclude, int (&best)[N], float (&mind)[N]) {
    int L[N], idx[N], half[N], runs[N];
    int most = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        L[k] = se[k].y - se[k].x + 1;
        mind[k] = PCP_BIG;
        best[k] = -1;
        const bool search = live[k] && L[k] >= 2;
        idx[k] = search ? ((se[k].x + se[k].y) / 2) % L[k] : 0;
        half[k] = L[k] / 2;
        runs[k] = search ? 32 - __clz(L[k] - 1) : 0;
        most = max(most, runs[k]);
    }
    for (int m = 1; m < 64; m <<= 1) most = max(most, __shfl_xor(most, m, 64));      // wave-uniform trip count
    bool together[N];
#pragma unroll
    for (int k = 0; k < N; ++k) together[k] = PCP_SCALAR_LEVELS != 0;
    for (int i = 0; i < most; ++i) {
        bool edge = false;
#pragma unroll
        for (int k = 0; k < N; ++k) edge = edge || idx[k] == 0 || idx[k] + 1 >= L[k];
        pcp_f4 v1[N], v2[N];
        if (__any(edge)) {                            // some lane is at an end of one of its rings
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int i1 = (idx[k] == 0) ? L[k] - 1 : idx[k] - 1;
                const int i2 = (idx[k] + 1 >= L[k]) ? 0 : idx[k] + 1;
                v1[k] = *(gf4_t)((const char PCP_GLOBAL*)T.ua + ((unsigned)(se[k].x + i1) << 4));
                v2[k] = *(gf4_t)((const char PCP_GLOBAL*)T.ua + ((unsigned)(se[k].x + i2) << 4));
            }
        } else {                                      // both neighbours sit beside idx: one address per ring
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int pos = se[k].x + idx[k];
                if (together[k] && (together[k] = __all(pos == __builtin_amdgcn_readfirstlane(pos)))) {    // as in search_level
                    typedef const pcp_f4 __attribute__((address_space(4)))* cf4_t;
                    const cf4_t m = (cf4_t)(const void*)((const char PCP_GLOBAL*)T.ua + ((size_t)(unsigned)__builtin_amdgcn_readfirstlane(pos) << 4));
                    v1[k] = m[-1];
                    v2[k] = m[1];
                } else {
                    const gf4_t m = (gf4_t)((const char PCP_GLOBAL*)T.ua + ((unsigned)pos << 4));
                    v1[k] = m[-1];
                    v2[k] = m[1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float c1 = dot4(make_float4(v1[k].x, v1[k].y, v1[k].z, v1[k].w), ref) + 1.0f;
            asm volatile("" : "+v"(c1));
            const float c2 = dot4(make_float4(v2[k].x, v2[k].y, v2[k].z, v2[k].w), ref) + 1.0f;
            const unsigned t = (unsigned)((c1 > c2) ? idx[k] - half[k] : idx[k] + half[k]);
            const int next = (int)min(min(t, t + (unsigned)L[k]), t - (unsigned)L[k]);
            if (i < runs[k]) { idx[k] = next; half[k] = (half[k] + 1) >> 1; }
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (!live[k]) continue;
        if (L[k] == 1) single_point_ring<EXCL>(T, q, se[k].x, se[k].y, exclude, best[k], mind[k]);
        else window_min<CHECK, EXCL>(T, q, se[k].x, L[k], idx[k], exclude, best[k], mind[k]);
    }
}

// Lower bound on distsq(q, a) over EVERY point a of a ring, from the ring's
// elevation band.  With u = unit_sphere(a) and the pole e = (0,1,0,0), the angle
// between u and the query direction is at least the difference of their
// elevations (triangle inequality on the unit sphere, any dimension), and the
// distance from q' = q - o' to a ray at angle D from it is at least |q'| sin D.
// st = (sin, cos) of the band's lowest and (sin, cos) of its highest elevation;
// (sq, cq) the query's.  The 1e-5 and 0.999 margins are orders of magnitude
// above the float32 rounding of every quantity involved, so a ring whose
// bound exceeds the running minimum cannot hold the minimum nor tie it.
// NaN statistics (set for rings too close to the pole or with undefined unit
// vectors) make the comparison false: such rings are never skipped.
__device__ __forceinline__ float ring_lower_bound(const float4 st, const float sq, const float cq,
                                                  const float rho2) {
    float sd = 0.0f;
    if (sq > st.z) sd = sq * st.w - cq * st.z;
    else if (sq < st.x) sd = st.x * cq - st.y * sq;
    sd = sd - 1e-5f;
    return (sd > 0.0f) ? (rho2 * 0.999f) * (sd * sd) : 0.0f;
}

// Azimuth bin of a unit vector, from its (x, z) components.  Targets and queries go through this
// one function, so a target point and a query whose bins differ by two or more (circularly) are
// at least 0.99 bin widths apart in true azimuth whatever atan2f rounds to.
__device__ __forceinline__ int az_bin(const float x, const float z) {
    const float t = (atan2f(z, x) + 3.14159265f) * ((float)PCP_AZ_BINS / 6.28318531f);
    const int b = (int)t;
    return min(max(b, 0), PCP_AZ_BINS - 1);
}

// Lower bound on distsq(q, a) over every point a of ring r, sharper than the whole ring's band:
// the ring's points in the query's azimuth bin and its two neighbours are bounded by THEIR elevation
// band (st, with st.y < 0 marking "no such point"), all others are at least 0.99 bin widths away in
// azimuth, hence at least |q'_xz| sin(that) away in the (x, z) plane.
__device__ __forceinline__ float ring_bin_lower_bound(const float4 st, const float sq, const float cq,
                                                      const float rho2, const float far) {
    const float near = (st.y < 0.0f) ? PCP_BIG : ring_lower_bound(st, sq, cq, rho2);
    return fminf(near, far);
}

// The bound a search tests a ring with before it visits it: the ring's points in the query's azimuth neighbourhood (bins
// b-1 .. b+1) are bounded by their elevation band and by their range band -- | |q'| - |a'| | <= |q' - a'|, both lengths from
// the same expressions (ring_stats_body) -- all others are at least `far` away.  The range band is only fetched when the
// elevation band alone does not rule the ring out.  2e-5 of the two ranges and 1e-5 m are far above the float32 rounding
// of the three square roots involved.  Used where a ring visit saved is worth the second look-up: the block-shared hinted
// search (2.50 -> 2.35 ms per 256 pairs); in the all-rings first dispatch, whose time is its bound tests, it cost more than
// it saved (5.5 -> 6.2 ms), and a single pair in flight did not notice it.
__device__ __forceinline__ float ring_bound(const Target& T, const gf4_t my_bins, const int r, const float sq, const float cq,
                                            const float rho2, const float len, const float far, const float mind) {
    const float4 st = ld4(my_bins, r);
    float near = (st.y < 0.0f) ? PCP_BIG : ring_lower_bound(st, sq, cq, rho2);
    if (near <= mind && far > mind) {
        const float4 rb = ld4(my_bins + (size_t)PCP_AZ_BINS * T.Z, r);
        const float gap = fmaxf(len - rb.y, rb.x - len) - ((len + rb.y) * 2e-5f + 1e-5f);
        if (gap > 0.0f && rb.y < PCP_BIG) near = fmaxf(near, (gap * gap) * 0.999f);
    }
    return fminf(near, far);
}

// The ring an all-rings search starts at: one whose band is near the query's elevation.  Which ring that is only
// decides the order of the visits, never the result, so it need not be THE nearest band: neighbouring scan rings are
// neighbours in elevation, so every fourth ring's bound is looked at and then the rings around the best of those
// (65 bounds per query before; the first launch of a 64-pair step 1.51 -> 1.42 ms).
__device__ __forceinline__ int start_ring(const Target& T, const float sq, const float cq, const float rho2, const int from,
                                          const int to) {
    int centre = from;
    float best = PCP_BIG;
    for (int r = from; r < to; r += 4) {
        const float lb = ring_lower_bound(T.stat[r], sq, cq, rho2);
        if (lb < best) { best = lb; centre = r; }
    }
    const int c0 = centre;
    for (int r = max(from, c0 - 3); r < min(to, c0 + 4); ++r) {
        const float lb = ring_lower_bound(T.stat[r], sq, cq, rho2);
        if (lb < best || (lb == best && r < centre)) { best = lb; centre = r; }
    }
    return centre;
}

// findClosestPoint, FunctionalBase.glsl:203-231.  hint is corr[id].w, updated in place.
// The shader walks rings [from, to) in ascending order and keeps the strict
// minimum, i.e. the smallest distance, then the lowest ring, then the first
// window slot.  That triple does not depend on the order of the visits, so this
// routine starts at the most promising ring (the hinted one, else the one whose
// elevation band is nearest to the query), works outwards from it so that the
// running minimum tightens early, and skips every ring whose lower bound exceeds
// that minimum.
__device__ __forceinline__ int find_closest(const Target& T, const float4 q, float& hint) {
    const float4 v = make_float4(q.x - T.o.x, (q.y - T.o.y) - PCP_LIDAR_OFF_Y, q.z - T.o.z, q.w - T.o.w);
    const float rho2 = dot4(v, v);
    const float len = sqrtf(rho2);
    const float4 ref = make_float4(v.x / len, v.y / len, v.z / len, v.w / len);   // == unit_sphere(q, o)
    const float sq = ref.y;
    const float cq = sqrtf(ref.x * ref.x + ref.z * ref.z + ref.w * ref.w);
    const bool can_prune = cq > 0.05f;          // near the pole the elevation is ill-conditioned
    const gf4_t my_bins = T.bins + (size_t)az_bin(ref.x, ref.z) * T.Z;   // this query's row
    const float far = (v.x * v.x + v.z * v.z) * PCP_AZ_FAR;
    float mind = PCP_BIG;
    int glob = 0, ring = -1;
    // An all-rings search stops at Zs: a copy of a ring returns the same point at the same distance,
    // and of equal results the lowest ring wins, so the copies can never change (point, ring).
    int from = 0, to = T.Zs, centre = 0;
    if (hint != -1.0f) {
        const int h = (int)hint;
        from = max(0, h - 2);
        to = min(T.Z, h + 2);
        centre = min(max(h, from), to - 1);
    } else if (can_prune) {
        centre = start_ring(T, sq, cq, rho2, from, to);
    }
    const int last = 2 * max(centre - from, to - 1 - centre);       // offset index of this lane's farthest ring
#ifdef PCP_STATS
    const int sbase = (hint != -1.0f) ? 0 : 4;
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&pcp_stats[sbase + 2], 1ull);
#endif
    // All lanes step through the offsets 0, +1, -1, +2, ... together: lanes of a wave mostly share
    // their centre ring, so a pass gathers from one ring and coalesces.  (Letting each lane jump to
    // its own next unpruned ring saves 10 % of the passes and costs 30 % in time: measured.)
    for (int k = 0; k <= last; ++k) {
        const int r = centre + ((k & 1) ? (k + 1) / 2 : -(k / 2));      // c, c+1, c-1, c+2, c-2, ...
        if (r < from || r >= to) continue;
        if (can_prune && mind < PCP_BIG) {
            if (ring_lower_bound(T.stat[r], sq, cq, rho2) > mind) continue;
            if (ring_bin_lower_bound(ld4(my_bins, r), sq, cq, rho2, far) > mind) continue;
        }
#ifdef PCP_STATS
        {
            const unsigned long long act = __ballot(true);
            if (__ffsll((long long)act) - 1 == (int)(threadIdx.x & 63)) {
                atomicAdd(&pcp_stats[sbase + 0], (unsigned long long)__popcll(act));
                atomicAdd(&pcp_stats[sbase + 1], 1ull);
            }
        }
#endif
        const int2 se = T.scan[r];
        int best;
        float d;
        ring_window<10, false>(T, q, ref, se.x, se.y, -1, best, d);
        if (d < mind || (d == mind && ring >= 0 && r < ring)) { mind = d; glob = best; ring = r; }
    }
    if (ring >= 0) hint = (float)ring;
    return glob;
}

// find_closest for a whole wave, with the ring searches behind the first one run as dense passes.  In the
// lock-step walk above a ring pass costs the wave the same whether one lane or all 64 need it, and beyond the first
// offset only some lanes do (all rings: a query far from every target point cannot rule rings out, its neighbours can;
// hinted: 2.2 of a query's four rings survive their bounds, not the same ones for every lane).  Here the
// wave walks the first PCP_POOL_LOCKSTEP offsets together as before; for the offsets behind them every lane only TESTS
// its ring's bounds and puts the rings it cannot rule out on a list the wave shares (LDS; offset by offset and lane by
// lane, so entries that lie together are neighbouring queries on the same or a neighbouring ring and their gathers
// still fall into few lines).  Whenever 64 entries are waiting, the wave searches them in one pass, one entry per lane,
// the query's point and direction fetched from the lane that owns it; the results meet in a per-query 64-bit word
// (distance bits, ring, index) by an LDS minimum, which is the shader's order: smallest distance, then lowest ring.
// The owners read their word back after every pass, so later bounds are tested against the tightened minimum.
// A ring that is searched although the lock-step walk would have ruled it out by then can neither hold nor tie the
// minimum, so the result is the same triple (point, ring, distance).  Needs Z <= 256 and n_a <= 2^24.  Every lane of
// the wave must call (live = false for lanes without a query).
#ifndef PCP_POOL_LOCKSTEP
#define PCP_POOL_LOCKSTEP 1
#endif
__device__ __forceinline__ unsigned long long pool_pack(const float d, const int ring, const int glob) {
    return ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)(unsigned)ring << 24) | (unsigned long long)(unsigned)glob;
}

__device__ __forceinline__ int find_closest_pooled(const Target& T, const float4 q, const bool live, float& hint,
                                                   unsigned long long* slot, unsigned short* items) {
    const int lane = threadIdx.x & 63;
    const float4 v = make_float4(q.x - T.o.x, (q.y - T.o.y) - PCP_LIDAR_OFF_Y, q.z - T.o.z, q.w - T.o.w);
    const float rho2 = dot4(v, v);
    const float len = sqrtf(rho2);
    const float4 ref = make_float4(v.x / len, v.y / len, v.z / len, v.w / len);   // == unit_sphere(q, o)
    const float sq = ref.y;
    const float cq = sqrtf(ref.x * ref.x + ref.z * ref.z + ref.w * ref.w);
    const bool can_prune = cq > 0.05f;
    const gf4_t my_bins = T.bins + (size_t)az_bin(ref.x, ref.z) * T.Z;
    const float far = (v.x * v.x + v.z * v.z) * PCP_AZ_FAR;
    float mind = PCP_BIG;
    int glob = 0, ring = -1;
    int from = 0, to = T.Zs, centre = 0;          // the same ranges and starting ring as find_closest
    if (hint != -1.0f) {
        const int h = (int)hint;
        from = max(0, h - 2);
        to = min(T.Z, h + 2);
        centre = min(max(h, from), to - 1);
    } else if (can_prune) {
        centre = start_ring(T, sq, cq, rho2, from, to);
    }
    const int last = live ? 2 * max(centre - from, to - 1 - centre) : -1;
    // ---- the first offsets in lock step, as find_closest walks them
    for (int k = 0; k < PCP_POOL_LOCKSTEP; ++k) {
        const int r = centre + ((k & 1) ? (k + 1) / 2 : -(k / 2));
        if (k > last || r < from || r >= to) continue;
        if (can_prune && mind < PCP_BIG) {
            if (ring_lower_bound(T.stat[r], sq, cq, rho2) > mind) continue;
            if (ring_bin_lower_bound(ld4(my_bins, r), sq, cq, rho2, far) > mind) continue;
        }
        const int2 se = T.scan[r];
        int best;
        float d;
        ring_window<10, false>(T, q, ref, se.x, se.y, -1, best, d);
        if (d < mind || (d == mind && ring >= 0 && r < ring)) { mind = d; glob = best; ring = r; }
    }
    // ---- the other offsets: listed, and searched 64 at a time
    slot[lane] = (ring >= 0) ? pool_pack(mind, ring, glob) : ~0ull;
    int kmax = last;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) kmax = max(kmax, __shfl_xor(kmax, m, 64));
    int count = 0;                                   // entries waiting (the same in every lane)
    auto run_pass = [&](const int n_items) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const bool act = lane < n_items;
        const int item = act ? (int)items[lane] : 0;
        const int l = item & 63, r = item >> 6;
        const float4 oq = make_float4(__shfl(q.x, l, 64), __shfl(q.y, l, 64), __shfl(q.z, l, 64), __shfl(q.w, l, 64));
        const float4 oref = make_float4(__shfl(ref.x, l, 64), __shfl(ref.y, l, 64), __shfl(ref.z, l, 64), __shfl(ref.w, l, 64));
        if (act) {
            const int2 se = T.scan[r];
            int best;
            float d;
            ring_window<10, false>(T, oq, oref, se.x, se.y, -1, best, d);
            if (d < PCP_BIG) atomicMin(&slot[l], pool_pack(d, r, best));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // entries behind the ones just searched move to the front
        const int rest = count - n_items;
        unsigned short keep = 0;
        if (lane < rest) keep = items[n_items + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < rest) items[lane] = keep;
        count = rest;
        const unsigned long long mine = slot[lane];
        if (mine != ~0ull) mind = fminf(mind, __uint_as_float((unsigned)(mine >> 32)));
    };
    for (int k = PCP_POOL_LOCKSTEP; k <= kmax; ++k) {
        const int r = centre + ((k & 1) ? (k + 1) / 2 : -(k / 2));
        bool want = k <= last && r >= from && r < to;
        if (want && can_prune && mind < PCP_BIG) {
            if (ring_lower_bound(T.stat[r], sq, cq, rho2) > mind) want = false;
            else if (ring_bin_lower_bound(ld4(my_bins, r), sq, cq, rho2, far) > mind) want = false;
        }
        const unsigned long long b = __ballot(want);
        if (b) {
            if (want) items[count + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)(lane | (r << 6));
            count += __popcll(b);
            if (count >= 64) run_pass(64);
        }
    }
    if (count > 0) run_pass(count);
    const unsigned long long mine = slot[lane];
    if (mine != ~0ull) {
        ring = (int)((mine >> 24) & 255ull);
        glob = (int)(mine & 0xFFFFFFull);
    }
    if (ring >= 0) hint = (float)ring;
    return glob;
}

// The hinted search of find_closest for a whole block of PCP_LS_THREADS queries: every query's hinted ring in lock step
// (all lanes busy), then the other rings of its range that the bounds cannot rule out against that first distance go,
// offset by offset and thread by thread, onto a list the block shares, and the block works the list off one entry per
// thread (a wave's entries are neighbouring queries on one ring, mostly).  Results meet in a per-query word as in
// find_closest_pooled: smallest distance, then lowest ring.  A ring searched although the lock-step walk, with its
// tightening minimum, would have skipped it can neither hold nor tie the minimum: same (point, ring, distance).
// Every thread of the block must call (live = false for threads without a hinted query); Z <= 256, n_a <= 2^24.
struct BlockPool {
    unsigned long long slot[PCP_LS_THREADS];
    float4 q[PCP_LS_THREADS], ref[PCP_LS_THREADS];
    unsigned items[3 * PCP_LS_THREADS];
    int count[3][PCP_LS_THREADS / 64];
};

__device__ __forceinline__ int find_closest_block(const Target& T, const float4 q, const bool live, float& hint, BlockPool& P) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float4 v = make_float4(q.x - T.o.x, (q.y - T.o.y) - PCP_LIDAR_OFF_Y, q.z - T.o.z, q.w - T.o.w);
    const float rho2 = dot4(v, v);
    const float len = sqrtf(rho2);
    const float4 ref = make_float4(v.x / len, v.y / len, v.z / len, v.w / len);   // == unit_sphere(q, o)
    const float sq = ref.y;
    const float cq = sqrtf(ref.x * ref.x + ref.z * ref.z + ref.w * ref.w);
    const bool can_prune = cq > 0.05f;
    const gf4_t my_bins = T.bins + (size_t)az_bin(ref.x, ref.z) * T.Z;
    const float far = (v.x * v.x + v.z * v.z) * PCP_AZ_FAR;
    float mind = PCP_BIG;
    int glob = 0, ring = -1, from = 0, to = 0, centre = 0;
    if (live) {
        const int h = (int)hint;
        from = max(0, h - 2);
        to = min(T.Z, h + 2);
        centre = min(max(h, from), to - 1);
        const int2 se = T.scan[centre];
        int best;
        float d;
        ring_window<10, false>(T, q, ref, se.x, se.y, -1, best, d);
        if (d < mind) { mind = d; glob = best; ring = centre; }
    }
    P.slot[t] = (ring >= 0) ? pool_pack(mind, ring, glob) : ~0ull;
    P.q[t] = q;
    P.ref[t] = ref;
    // the other rings of the range in find_closest's order (centre + 1, centre - 1, centre - 2; centre + 2 lies outside)
    int r3[3];
    bool want[3];
    unsigned long long b3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r3[j] = (j == 0) ? centre + 1 : centre - j;
        want[j] = live && r3[j] >= from && r3[j] < to;
        if (want[j] && can_prune && mind < PCP_BIG) {
            if (ring_lower_bound(T.stat[r3[j]], sq, cq, rho2) > mind) want[j] = false;
            else if (ring_bound(T, my_bins, r3[j], sq, cq, rho2, len, far, mind) > mind) want[j] = false;
        }
        b3[j] = __ballot(want[j]);
        if (lane == 0) P.count[j][wave] = __popcll(b3[j]);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int before = total;
#pragma unroll
        for (int w = 0; w < PCP_LS_THREADS / 64; ++w) {
            const int c = P.count[j][w];
            if (w < wave) before += c;
            total += c;
        }
        if (want[j]) P.items[before + __popcll(b3[j] & ((1ull << lane) - 1ull))] = (unsigned)t | ((unsigned)r3[j] << 16);
    }
    __syncthreads();
    for (int base = 0; base < total; base += PCP_LS_THREADS) {
        if (base + t < total) {
            const int item = (int)P.items[base + t];
            const int l = item & 0xffff, r = item >> 16;
            const float4 oq = P.q[l], oref = P.ref[l];
            const int2 se = T.scan[r];
            int best;
            float d;
            ring_window<10, false>(T, oq, oref, se.x, se.y, -1, best, d);
            if (d < PCP_BIG) atomicMin(&P.slot[l], pool_pack(d, r, best));
        }
    }
    __syncthreads();
    const unsigned long long mine = P.slot[t];
    if (mine != ~0ull) {
        ring = (int)((mine >> 24) & 255ull);
        glob = (int)(mine & 0xFFFFFFull);
    }
    if (live && ring >= 0) hint = (float)ring;
    return glob;
}

// The all-rings search (hint -1) of find_closest with K lanes per query, K a power of two <= 64 and the
// K lanes adjacent: lane `sub` takes the offsets k = sub, sub + K, ... of the same outward walk, the K
// running minima are shared after every round (they only serve the bound tests, and a ring whose lower
// bound exceeds ANY distance already found can neither hold nor tie the minimum), and the results are
// merged by the same order (smaller distance, then lower ring).  Returns the ring find_closest would
// leave in the hint, -1 if no ring produced a comparable distance.  `live` lanes hold a query; every
// lane of the wave must call.
template <int K>
__device__ __forceinline__ int first_ring(const Target& T, const bool live, const float4 q, const int sub) {
    const float4 v = make_float4(q.x - T.o.x, (q.y - T.o.y) - PCP_LIDAR_OFF_Y, q.z - T.o.z, q.w - T.o.w);
    const float rho2 = dot4(v, v);
    const float len = sqrtf(rho2);
    const float4 ref = make_float4(v.x / len, v.y / len, v.z / len, v.w / len);
    const float sq = ref.y;
    const float cq = sqrtf(ref.x * ref.x + ref.z * ref.z + ref.w * ref.w);
    const bool can_prune = cq > 0.05f;
    const gf4_t my_bins = T.bins + (size_t)az_bin(ref.x, ref.z) * T.Z;
    const float far = (v.x * v.x + v.z * v.z) * PCP_AZ_FAR;
    const int from = 0, to = T.Zs;
    int centre = 0;
    if (can_prune) {                 // the K lanes share the scan for the most promising ring (first of the smallest bounds)
        float lo = PCP_BIG;
        for (int r = from + sub; r < to; r += K) {
            const float lb = ring_lower_bound(T.stat[r], sq, cq, rho2);
            if (lb < lo) { lo = lb; centre = r; }
        }
#pragma unroll
        for (int s = 1; s < K; s <<= 1) {
            const float olo = __shfl_xor(lo, s, 64);
            const int oc = __shfl_xor(centre, s, 64);
            if (olo < lo || (olo == lo && oc < centre)) { lo = olo; centre = oc; }
        }
    }
    const int last = live ? 2 * max(centre - from, to - 1 - centre) : -1;
    float mind = PCP_BIG, shared = PCP_BIG;
    int ring = -1;
#ifdef PCP_STATS
    int st_rounds = 0, st_active = 0;
    const bool st_leader = __ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63);
    if (st_leader) { atomicAdd(&pcp_stats2[0], 1ull); atomicAdd(&pcp_stats2[4], (unsigned long long)(__popcll(__ballot(live)) / K)); }
#endif
    for (int k0 = 0; __any(k0 <= last); k0 += K) {
        const int k = k0 + sub;
        const int r = centre + ((k & 1) ? (k + 1) / 2 : -(k / 2));
        bool visit = k <= last && r >= from && r < to;
        if (visit && can_prune && shared < PCP_BIG) {
            if (ring_lower_bound(T.stat[r], sq, cq, rho2) > shared) visit = false;
            else if (ring_bin_lower_bound(ld4(my_bins, r), sq, cq, rho2, far) > shared) visit = false;
        }
#ifdef PCP_STATS
        {
            const unsigned long long act = __ballot(visit);
            ++st_rounds;
            if (act) ++st_active;
            if (st_leader) atomicAdd(&pcp_stats2[3], (unsigned long long)__popcll(act));
        }
#endif
        if (visit) {
            const int2 se = T.scan[r];
            int best;
            float d;
            ring_window<10, false>(T, q, ref, se.x, se.y, -1, best, d);
            if (d < mind || (d == mind && ring >= 0 && r < ring)) { mind = d; ring = r; }
        }
        shared = fminf(shared, mind);
#pragma unroll
        for (int s = 1; s < K; s <<= 1) shared = fminf(shared, __shfl_xor(shared, s, 64));
    }
#ifdef PCP_STATS
    if (st_leader) {
        atomicAdd(&pcp_stats2[1], (unsigned long long)st_rounds);
        atomicAdd(&pcp_stats2[2], (unsigned long long)st_active);
        atomicAdd(&pcp_stats2[8 + (st_active < 7 ? st_active : 7)], 1ull);
    }
#endif
#pragma unroll
    for (int s = 1; s < K; s <<= 1) {
        const float od = __shfl_xor(mind, s, 64);
        const int orr = __shfl_xor(ring, s, 64);
        if (orr >= 0 && (ring < 0 || od < mind || (od == mind && orr < ring))) { mind = od; ring = orr; }
    }
    return ring;
}

// One ring of the hinted search of find_closest (rings [max(0, h-2), min(Z, h+2))): ring from + sub, its
// window minimum in (mind, glob), ring = -1 when the ring is outside the range or yields no distance.
// The caller merges sub = 0..3 in that order with a strict '<', which is the shader's loop; no ring is
// skipped, which changes nothing (a skipped ring can neither hold nor tie the minimum).
__device__ __forceinline__ void hinted_ring(const Target& T, const bool live, const float4 q, const int sub,
                                            const float hint, float& mind, int& glob, int& ring) {
    const float4 v = make_float4(q.x - T.o.x, (q.y - T.o.y) - PCP_LIDAR_OFF_Y, q.z - T.o.z, q.w - T.o.w);
    const float len = sqrtf(dot4(v, v));
    const float4 ref = make_float4(v.x / len, v.y / len, v.z / len, v.w / len);   // == unit_sphere(q, o)
    const int h = (int)hint;
    const int r = max(0, h - 2) + sub;
    mind = PCP_BIG;
    glob = 0;
    ring = -1;
    if (live && r < min(T.Z, h + 2)) {
        const int2 se = T.scan[r];
        int best;
        float d;
        ring_window<10, false>(T, q, ref, se.x, se.y, -1, best, d);
        if (d < mind) { mind = d; glob = best; ring = r; }
    }
}

__device__ __forceinline__ float nan0(const float v) { return (v != v) ? 0.0f : v; }

// Upper-triangle index of (i, j), i <= j, in the 27-value record.
__host__ __device__ constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// LeastSquaresPoint.glsl:87-113.  v[tri(i,j)] = Hs[i][j], v[21+i] = Hs[i][6].
__device__ __forceinline__ void ls_point_values(const float4 pa, const float4 pb, const float dist,
                                                float (&v)[PCP_NVAL]) {
    const float valid = (dist < 0.1f) ? 1.0f : 0.0f;
    const float e[3] = {pa.x - pb.x, pa.y - pb.y, pa.z - pb.z};
    const float x = pb.x, y = pb.y, z = pb.z;
    const float J[3][6] = {{1.0f, 0.0f, 0.0f, 0.0f, -z, y},
                           {0.0f, 1.0f, 0.0f, z, 0.0f, -x},
                           {0.0f, 0.0f, 1.0f, -y, x, 0.0f}};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s = s + J[k][i] * J[k][j];
            v[tri(i, j)] = nan0(s * valid);
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s = s + J[k][i] * e[k];
        v[21 + i] = nan0(s * valid);
    }
}

// LeastSquaresPlane.glsl:91-112 with getJ :1-16 (point = the TARGET point).
__device__ __forceinline__ void ls_plane_values(const float4 pa, const float4 pb, const float4 n,
                                                const float dist, float (&v)[PCP_NVAL]) {
    const float valid = (dist < 0.2f) ? 1.0f : 0.0f;
    const float err = dot3(n.x, n.y, n.z, pa.x - pb.x, pa.y - pb.y, pa.z - pb.z);
    const float J[6] = {n.x, n.y, n.z,
                        n.y * pa.z - n.z * pa.y,
                        n.z * pa.x - n.x * pa.z,
                        n.x * pa.y - n.y * pa.x};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) v[tri(i, j)] = nan0((J[i] * J[j]) * valid);
        v[21 + i] = nan0((J[i] * err) * valid);
    }
}

// The pairwise tree of ParalellHBSum.glsl:158-164 over the 64 lanes of a wave, for all PCP_NVAL values at
// once: x[l] += x[l + s] for the lanes l < s, s = 32, 16, 8, 4, 2, 1, after which lane 0 holds sdata[0] of the
// shader.  Returns for lane k < PCP_NVAL the sum of value k, 0 for the other lanes.
// Only the lanes below s matter at step s, and they read lanes below 2s, so what the other lanes receive is
// irrelevant.  Steps 32 and 16 are the gfx950 half-wave and row swaps: the swap that step 32 needs can exchange
// the halves of TWO values, after which one addition performs step 32 for both (each left in one half of the
// register), and the row swap of step 16 does the same with two such registers.  Seven registers then hold four
// values each, one per row of 16 lanes, and go through steps 8 ... 1 together as row shifts applied inside the
// addition (DPP).  Every value sees exactly the additions of the shader's tree; only unrelated values share a
// register.  No LDS traffic but the final gather of the 27 sums.
__device__ __forceinline__ float wave_tree_sum_all(const float (&v)[PCP_NVAL], const int lane) {
    typedef unsigned pcp_u2 __attribute__((ext_vector_type(2)));
    static_assert(PCP_NVAL <= 28, "seven registers of four values");
    float h[14];                    // h[m] = values 2m (lanes 0-31) and 2m + 1 (lanes 32-63) after step 32
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const float lo = v[2 * m];
        const float hi = (2 * m + 1 < PCP_NVAL) ? v[2 * m + 1] : 0.0f;
        const pcp_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        h[m] = __uint_as_float(r.x) + __uint_as_float(r.y);       // [lo.low + lo.high | hi.low + hi.high]
    }
    float mine = 0.0f;
    const int row_of = ((lane & 1) << 1) | ((lane >> 1) & 1);      // value 4n + c sits in row {0, 2, 1, 3}[c] of register n
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        // rows before: h[2n] = [a0 a1 | b0 b1], h[2n+1] = [c0 c1 | d0 d1]; after the swap [a0 c0 b0 d0] and [a1 c1 b1 d1]
        const pcp_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[2 * n]), __float_as_uint(h[2 * n + 1]), false, false);
        float x = __uint_as_float(r.x) + __uint_as_float(r.y);
        x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x108, 0xf, 0xf, true));   // row_shl:8
        x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0xf, true));   // row_shl:4
        x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x102, 0xf, 0xf, true));   // row_shl:2
        x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xf, 0xf, true));   // row_shl:1
        const float got = __shfl(x, row_of << 4, 64);             // first lane of the row that holds this lane's value
        mine = ((lane >> 2) == n && lane < PCP_NVAL) ? got : mine;
    }
    return mine;
}

}  // namespace pcp
