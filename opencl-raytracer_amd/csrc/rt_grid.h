// rt_grid.h - conservative uniform grid over the objects' bounding spheres, for the large-scene trace kernels.
//
// The reference tests every ray against every object. The grid does not change WHAT is computed for an
// object - a candidate still goes through the reference's exact test on the same object-space ray - it only
// skips objects that cannot pass that test, so the result (winner index, t, visibility) is bit-identical to
// the brute-force loops. "Cannot pass" has to hold for the reference's fp32 arithmetic, not for exact geometry.
//
// Bound (u = 2^-24; A, b = rows x,y,z of mvInverse; c = -A^-1 b the centre, R = r0 sigma_max(A^-1) the bounding
// radius with r0 = 1 for the unit sphere and sqrt(0.75) for the unit box; kappa = cond(A); dist = |start - c|):
//  * the object-space origin o^ = fl(A start + b) is off by |do| <= 6u |A|_F (|start| + |c|) (the sum cancels: the
//    error scales with the coordinates, not with dist), the direction by a relative 5.2 u kappa;
//  * on those computed vectors, radical = B^2 - 4AC differs from the exact 4A(1 - b*^2) (b* = distance of the
//    computed line from the object's origin) by at most 4A u (14 |o^|^2 + 8), so `radical >= 0` needs
//    b*^2 <= 1 + u (14 |o^|^2 + 8); the box's slab test accepts only if some point of the line has all three
//    |p_i| <= 0.5 + 2u(|o^_i| + 0.5), which is inside the same expression with r0^2 in place of 1;
//  * back in view space (|o^| <= kappa dist / sigma_max(A^-1), distances scale by at most sigma_max(A^-1)):
//        the line passes c within  sqrt(R^2 (1 + 8u) + 14 u kappa^2 dist^2) + 10.4 u kappa (|start| + |c|) + 9 u kappa^2 dist.
// build_grid (rt_api.cpp) evaluates this with u_eff = 2e-7 (3.3 u; also covers the unfused flavour's extra
// roundings), in double precision, twice: with the farthest possible ray origin for the radius an object is
// REGISTERED with (+ 0.01 cell for the walk's own fp32 arithmetic), and as a function of the ray's actual distance
// for the PRE-TEST below. Objects whose bounds are not finite or as large as the scene sit in an "always" list that
// every ray tests. A reported hit additionally needs t >= 0: the pre-test's "entirely behind" rejection keeps a
// 1e-5 relative margin on |start - c|^2 - w^2, eight times what the cancellation in (root - B) / 2A can move.
// tests/ compare the grid path with the brute-force path bit for bit (RT_FLAG_NO_GRID forces the latter).
//
// Order: cells are visited front to back and objects repeat across cells, so the closest-hit update uses the
// order-free form of the reference's sequential tie rules (Q3): among the candidates with the smallest t the
// winner is the highest-index sphere if there is one, else the lowest-index box. The walk keeps going for two
// more cells after the current best hit.
#pragma once

#include "rt_device.h"

namespace rt {

// How far past the current best hit (closest) or past t = 1 (shadow) a walk continues, in cell edges. An object
// whose computed t is <= the current best has the point start + t dir inside its registration sphere (the sphere
// holds the surface plus the fp32 error of t, which is ~1e-6 of the distances involved), i.e. it is registered in
// the cell that contains that point - a cell the walk enters at a parameter <= t. The slack only has to cover the
// walk's own rounding; half a cell is 3-4 orders of magnitude more than that.
#ifndef RT_WALK_SLACK
#define RT_WALK_SLACK 0.5f
#endif
constexpr float kWalkSlackCells = RT_WALK_SLACK;

struct GridDesc {
    float lox, loy, loz;        // grid origin (view space)
    float inv_cell;             // 1 / cell edge
    float cell;                 // cell edge
    int nx, ny, nz;
    const uint2* __restrict__ cell_range;     // nx*ny*nz x {first entry, number of entries}; an EMPTY cell's first word is
                                              // the number of further walk steps that are sure to stay in empty cells
    const float4* __restrict__ cell_rec;      // nx*ny*nz x 32 bytes for the persistent walk: [0] the bounding sphere of the cell's FIRST
                                              // entry, [1] bits {first entry, number of entries, object of the first entry, 0} - the
                                              // range and the first candidate arrive with ONE request (RT_CELL_INLINE)
    const uint32_t* __restrict__ entries;     // object indices, ascending inside a cell
    const float4* __restrict__ entry_sphere;  // parallel to entries: the object's inflated bounding sphere (centre, R_grid)
    const uint32_t* __restrict__ always;      // objects every ray must test
    uint32_t n_always;
    float pretest_alpha;   // distance-dependent term of the pre-test radius (misses_bounding_sphere)
    float own_shrink;      // first-cell rule (entered_inside): how much smaller than an entry's |w| the ball is taken (usually 0)
    uint32_t has_triangles;  // the scene holds type-2 records (selects the kernel variants that know them)
    uint32_t enabled;
    // The unified walk (rt_wavefront.hip: walk_segment; scenes without triangles). ONE table of 32-byte records
    //   [0] {cx, cy, cz, w^2}   bounding sphere of a candidate (centre, SQUARED pre-test radius; -inf: no candidate here)
    //   [1] {object, next, key, 0}   next = index of the record to look at after this one; 0 = the list ends, step to the next cell
    // indexed as: [0, walk_cells) one head per cell of the grid PADDED by two empty cells on every side (strides walk_nx,
    // walk_nxy; cell (ix, iy, iz) of the grid proper is head ((iz + 2) walk_ny + iy + 2) walk_nx + ix + 2) - the head holds the
    // cell's first entry, so range and first candidate still arrive with one request; then the 2nd, 3rd ... entries of every
    // cell, consecutive; then the light tiles' entries (LightTiles::walk_base; key = the entry's distance key, -inf elsewhere).
    // Every trip of a walking lane is the same: fetch the record under its cursor, pre-test it, move the cursor. The empty
    // border replaces the per-axis step budgets: a walk ends when it passes its exit parameter (+ a quarter step), which
    // it does inside the border. A record without a candidate points at object `walk_none`, a never-hit dummy behind the
    // last HotObject, so that even a pre-test that lets it through (overflow to NaN) does no harm.
    const float4* __restrict__ walk_rec;
    uint32_t walk_cells, walk_nx, walk_nxy;
    uint32_t walk_none;
};

// The block walk (rt_wavefront.hip: block_segment; closest-hit rays of scenes without triangles). What bounds a grid walk
// on this chip is neither instruction issue nor latency but the number of cache LINES its lanes pull per second: a
// dependent random fetch costs a CU ~5 cycles per line that misses L2 and ~3 per line that hits, whether 16 or 64 of
// its bytes are used and whether 4 or 8 waves per SIMD are resident (tools/ubench/gather_rate.hip). So a fetch should
// carry as many candidates as a line can hold. A SECOND, coarser grid over the same box, one 32-byte block per cell:
//   word 0      header: next block of the cell's chain (bits 0-23, 0 = none), lattice scale s (bits 27-28); an EMPTY cell's head:
//               how many further steps of a walk are sure to stay in empty cells (bits 24-26 and 29-31: 0..63)
//   words 1-7   one candidate each, its bounding sphere quantised to 8 bits per field {x, y, z, r} on a lattice of 256
//               steps centred on the cell: step = 2^s cell / 128, i.e. scale 0 reaches half a cell beyond the cell on
//               every side (unused slots: all zero - a sphere of radius 0 at a lattice corner; should a ray ever pass
//               its pre-test, its id is the never-hit dummy object)
// and a parallel array of object ids (8 per block), read only for the few entries that pass the pre-test. The pre-test
// is the one of misses_bounding_sphere, evaluated in lattice coordinates (a similarity transform: the ray origin is
// moved there once per trip). The host (rt_api.cpp: build_walk_blocks) rounds every sphere OUTWARDS: it computes the
// lattice the way the device does (same fp32 fma), knows each entry's exact quantisation error and adds it - plus the
// transform's rounding and the cross term of the distance-dependent tolerance - to the radius; a sphere that does not
// fit any scale becomes "the whole cell". Seven candidates per line instead of one: cells can be ~1.7x larger, a ray
// pulls ~14 lines instead of ~34 and the table fits an XCD's L2. The array carries two empty cells on every side (as
// GridDesc::walk_rec: walks end by parameter, inside the border).
struct BlockGrid {
    float lox, loy, loz;       // origin of the grid proper (view space)
    float cell, inv_cell;
    int nx, ny, nz;            // cells of the grid proper
    float c0x, c0y, c0z;       // centre of padded cell (0, 0, 0); padded cell f has its centre at fma(f, cell, c0)
    float inv_step;            // lattice steps per unit at scale 0 (128 / cell)
    uint32_t wnx, wny;         // padded array: cells per row, rows per slab
    uint32_t n_cells;          // head blocks (one per padded cell); chain blocks follow
    const uint4* __restrict__ blocks;   // 2 x uint4 per block
    const uint32_t* __restrict__ ids;   // 8 per block (slot e of block b: ids[8 b + e]; slot 7 unused)
    uint32_t none;             // the never-hit dummy object
    uint32_t enabled;
    uint32_t take_skips;       // the walk takes the empty-space steps of the headers (always set; RT_BLOCK_SKIPS=0 is the measurement knob:
                               // cfg4 12.2 -> 12.4 ms, cfg5 40 -> 48 ms without them)
};
constexpr uint32_t kBlockBorder = 2;
#ifndef RT_BLOCK_ENTRIES
#define RT_BLOCK_ENTRIES 7   // candidates a block holds (<= 7); 4 / 5 / 6 with their best cell edge: 13.11 / 12.86 / 12.74 vs 12.67 ms per cfg4 frame
#endif
constexpr uint32_t kBlockEntries = RT_BLOCK_ENTRIES;
static_assert(kBlockEntries >= 1 && kBlockEntries <= 7, "a block is a header word and up to seven 4-byte candidates");

// Light-tile blocks (LightTiles::blocks) are addressed by 24-bit indices: a walk that has to come back to a block keeps the
// entry to resume at in bits 24+ of its cursor. build_light_tiles refuses the block form for tables beyond that.
constexpr uint32_t kLtBlockIndexBits = 24;
constexpr uint32_t kLtBlockIndexMask = (1u << kLtBlockIndexBits) - 1u;
__host__ __device__ constexpr bool light_tile_blocks_fit(uint64_t n_blocks) { return n_blocks < (1ull << kLtBlockIndexBits); }
static_assert(light_tile_blocks_fit((1ull << 24) - 1) && !light_tile_blocks_fit(1ull << 24) && !light_tile_blocks_fit(9000000ull * 2),
              "the last index a cursor can carry is 2^24 - 1");

// Element `index` of a read-only table smaller than 4 GiB (build_grid / build_light_tiles refuse larger ones): the byte
// offset is formed in 32 bits, so the load takes the scalar-base + 32-bit-vector-offset form instead of a 64-bit vector
// address computed with two more vector instructions per load - the walk does three such loads per trip.
template <typename T>
__device__ __forceinline__ T table_at(const T* __restrict__ base, uint32_t index) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint32_t)(index * (uint32_t)sizeof(T)));
}

// screen tiles for pinhole primary rays (built on the host per camera, rt_api.cpp: build_screen_tiles)
struct ScreenTiles {
    const uint32_t* __restrict__ tile_start;  // tiles_x * tiles_y + 1 offsets
    const uint32_t* __restrict__ entries;     // object indices, ascending inside a tile
    uint32_t tiles_x;                         // tiles are 8 rows tall and 1 << col_shift pixels wide:
    uint32_t col_shift;                       //   3 (8 x 8, exactly a wave's block when work-items walk 8 x 8 blocks) or 6 (64 x 8: a wave is 64 pixels of a row)
    uint32_t global_begin, n_global;          // entries[global_begin ..): objects whose projection is the whole screen
    uint32_t enabled;
};

// Light tiles: shadow rays towards ONE positional light (the last light - the one shade_and_reflect's colour comes from)
// all lie on lines through that light, so "which objects can this ray meet" is a 2-D question in the light's own
// perspective: objects are binned by the rectangle their registration sphere covers in gnomonic coordinates
// (u, v) = (x', y') / -z' of the light-local frame (x', y', z' = a signed permutation of p - L in which every object lies
// at z' < 0), and a ray looks up the one tile its origin falls in - no cell walk. Built on the host (rt_api.cpp:
// build_light_tiles) when every object lies strictly on one side of an axis-aligned plane through the light; every
// candidate still goes through the pre-test and the reference's exact test, so the answer is the brute-force loop's.
struct LightTiles {
    const uint2* __restrict__ tile_range;      // tiles_u * tiles_v x {first entry, count}
    const float4* __restrict__ records;        // two float4 per entry, side by side (one or two cache lines per ray instead of three tables):
                                               //   [0] centre (view space) + pre-test radius, as GridDesc::entry_sphere
                                               //   [1].x distance from the light to the nearest point of the registration sphere - a tile's entries
                                               //        are sorted by it, and a ray's list ENDS at the first key beyond its own origin;
                                               //   [1].y object index (bits)
    float lx, ly, lz;                          // the light
    float u0, v0, inv_du, inv_dv;              // tile (iu, iv) covers u0 + iu / inv_du ...
    uint32_t tiles_u, tiles_v;
    uint32_t ax, ay, az;                       // light-local x', y', z' = component ax / ay / az of p - L ...
    float sx, sy, sz;                          // ... times this sign
    uint32_t light;                            // index of the light the structure is for
    uint32_t enabled;
    uint32_t walk_base;                        // entry e of a tile = record walk_base + e of GridDesc::walk_rec (0: not built)
    float cut_pad;                             // absolute slack of the distance cut: the rounding of origin - light (scales with the coordinates)
    // The same lists as 32-byte BLOCKS of three candidates (scenes without triangles; round 3): a shadow ray looks at 2.4 entries of
    // its tile on average, each a 32-byte record in a table of tens of MB - 2.4 line fills from beyond L2 per ray, which is what the
    // shadow walk was paying for (rt_grid.h: BlockGrid has the cost model). Block t (t < tiles_u tiles_v) is the HEAD of tile t's
    // chain and sits in a table small enough for L2 (2 MB at 256 x 256 tiles); further blocks follow behind the heads.
    //   word 0   next block (0: the chain ends)        word 1   unused
    //   words 2-7   three candidates, two words each: {x16 | y16 << 16, z16 | r8 << 16 | k8 << 24} - the centre on a 16-bit lattice
    //               over the grid box (lat_lo + q lat_step), the pre-test radius r8 rstep rounded UP (quantisation error of the
    //               centre included, build_light_tiles), the distance key k8 kstep rounded DOWN; an empty slot has k8 = 255, r8 = 0
    // block_ids: 4 per block (slot 3 unused), read only for candidates that pass the pre-test.
    const uint4* __restrict__ blocks;
    const uint32_t* __restrict__ block_ids;
    float lat_lox, lat_loy, lat_loz, lat_step, rstep, kstep;
    uint32_t blocks_enabled;
};

// tile of the ray whose ORIGIN is `s` (any point of the line through the light does): false = no object in that direction
__device__ __forceinline__ bool light_tile_of(const LightTiles& lt, float sx, float sy, float sz, uint32_t& tile, float& dist) {
    const float p[3] = {sx - lt.lx, sy - lt.ly, sz - lt.lz};
    dist = __builtin_sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) * 1.000001f + lt.cut_pad;  // origin <-> light, rounded up
    const float qx = lt.sx * (lt.ax == 0u ? p[0] : (lt.ax == 1u ? p[1] : p[2]));
    const float qy = lt.sy * (lt.ay == 0u ? p[0] : (lt.ay == 1u ? p[1] : p[2]));
    const float qz = lt.sz * (lt.az == 0u ? p[0] : (lt.az == 1u ? p[1] : p[2]));
    if (!(qz < 0.f)) return false;  // (never for an origin on an object: they all lie at z' < 0, a margin away; NaN: caller)
    const float inv = -1.0f / qz;
    const float fu = (qx * inv - lt.u0) * lt.inv_du, fv = (qy * inv - lt.v0) * lt.inv_dv;
    if (!(fu >= 0.f && fv >= 0.f && fu < (float)lt.tiles_u && fv < (float)lt.tiles_v)) return false;
    tile = (uint32_t)fv * lt.tiles_u + (uint32_t)fu;
    return true;
}

// order-free closest-hit update (see header comment); `cur_sphere` = the current winner is a sphere. Triangles
// (extension) tie like boxes: the earlier object wins.
__device__ __forceinline__ void closest_take(float t, int k, bool sphere, float& T, int& index, bool& cur_sphere) {
    bool take;
    if (t < T) take = true;
    else if (t == T) take = sphere ? (!cur_sphere || k > index) : (!cur_sphere && k < index);
    else take = false;
    if (take) { T = t; index = k; cur_sphere = sphere; }
}

template <bool FUSED>
__device__ __forceinline__ void closest_update_unordered(uint32_t type, float sx, float sy, float sz, float dx, float dy,
                                                         float dz, int k, float& T, int& index, bool& cur_sphere) {
    float t;
    bool cand = false;
    if (type == 0u) cand = sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t);
    else if (type == 1u) cand = box_candidate(sx, sy, sz, dx, dy, dz, t);
    if (cand) closest_take(t, k, type == 0u, T, index, cur_sphere);
}

// Cheap conservative rejection of a grid candidate before its 52-byte matrix is fetched. s = (centre, w): the
// reference's fp32 test can only accept a ray whose line passes the centre within sqrt(w^2 + 6e-6 K^2 |oc|^2)
// (build_grid, rt_api.cpp: the bound is a function of the ray's ACTUAL distance |oc| from the object, which for
// secondary rays is a small fraction of the scene size the cell registration has to assume). `alpha` adds 8e-6
// for this test's own fp32 rounding (~1e-6 |oc|^2 A on the discriminant). A negative w marks an entry whose
// radius already contains the worst-case distance term (it gets the same tolerance: more than it needs). Second test: the sphere is entirely behind the origin.
__device__ __forceinline__ bool misses_bounding_sphere(const float4 s, const Ray& ray, float dd, float alpha) {
    // (bookkeeping arithmetic, not the reference's: explicit fused multiply-adds - this file is compiled without
    // contraction - and one tolerance for both kinds of entry: alpha >= 1.4e-5 covers the 8e-6 an entry with w < 0 needs;
    // 19 vector instructions instead of 28, at 13-17 pre-tests per closest-hit ray)
    const float ox = s.x - ray.sx, oy = s.y - ray.sy, oz = s.z - ray.sz;
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float od = __builtin_fmaf(oz, ray.dz, __builtin_fmaf(oy, ray.dy, ox * ray.dx));
    const float c = oo - s.w * s.w;
    // od^2 - dd c < -alpha oo dd   <=>   od^2 - dd (c - alpha oo) < 0
    const float disc = __builtin_fmaf(od, od, -(dd * __builtin_fmaf(-alpha, oo, c)));
    if (disc < 0.f) return true;                  // the line misses the sphere
    if (od < 0.f && c > 1.0e-5f * oo) return true;  // centre behind the origin and the origin clearly outside
    return false;                                 // (NaNs compare false: the candidate is tested)
}

// First-cell rule of the walk (meshes: a triangle is registered in every cell its guard sphere reaches, ~5 of them, and a
// ray that crosses several of those cells would run the exact test in each). True when the point at which the ray ENTERED
// the current cell (parameter t_enter >= 0; the first cell of a walk carries a negative one) lies inside the entry's ball
// of radius |w| - own_shrink. Then the cell the walk was in before - whose wall that point is on, up to the DDA's 1e-4 cell -
// comes within that radius of the centre, hence holds the object too (own_shrink is chosen on the host such that
// |w| - own_shrink + 1e-3 cell + this function's own rounding <= the object's registration radius, for every object), the
// same pre-test passed there (it does not depend on the cell), and by induction the exact test ran - or is parked - in
// the first cell of that chain, whose entry point is outside the ball or which is the walk's first. Skipping the entry
// here therefore only drops a REPEATED test: closest_take is order-free and idempotent, an any-hit walk has ended if
// the test succeeded.
__device__ __forceinline__ bool entered_inside(const float4 s, const Ray& ray, float t_enter, float own_shrink) {
    const float px = __builtin_fmaf(t_enter, ray.dx, ray.sx), py = __builtin_fmaf(t_enter, ray.dy, ray.sy), pz = __builtin_fmaf(t_enter, ray.dz, ray.sz);
    const float ox = s.x - px, oy = s.y - py, oz = s.z - pz;
    const float a = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float r = __builtin_fabsf(s.w) - own_shrink;
    return t_enter >= 0.f && r > 0.f && a < r * r;  // (NaN: false - the entry is tested)
}

// per-lane object test: the HotObject arrives through ordinary (divergent) vector loads
template <bool FUSED, bool DW0>
__device__ __forceinline__ void lane_object_space(const HotObject* __restrict__ o, const Ray& ray, float& sx, float& sy,
                                                  float& sz, float& dx, float& dy, float& dz, uint32_t& type) {
    const float4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
    type = o->type;
    sx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.sx, ray.sy, ray.sz, ray.sw);
    sy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.sx, ray.sy, ray.sz, ray.sw);
    sz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.sx, ray.sy, ray.sz, ray.sw);
    if constexpr (DW0) {
        dx = row3<FUSED>(r0.x, r0.y, r0.z, ray.dx, ray.dy, ray.dz);
        dy = row3<FUSED>(r1.x, r1.y, r1.z, ray.dx, ray.dy, ray.dz);
        dz = row3<FUSED>(r2.x, r2.y, r2.z, ray.dx, ray.dy, ray.dz);
    } else {
        dx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.dx, ray.dy, ray.dz, ray.dw);
    }
}

// the reference's exact test of one object (any type) for one lane: candidate or not, and its t
template <bool FUSED, bool DW0, bool TRI = true>
__device__ __forceinline__ bool lane_candidate(const HotObject* __restrict__ o, const Ray& ray, float& t, bool& sphere) {
    const uint32_t type = o->type;
    sphere = (type == 0u);
    if (TRI && type == 2u) {  // TRI = false: the caller knows the scene holds no triangles (saves registers in the walk)
        const float4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
        return triangle_candidate(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z, r0.w, r1.w, r2.w,
                                  __uint_as_float(o->pad[0]), ray, t);
    }
    float sx, sy, sz, dx, dy, dz;
    uint32_t ty;
    lane_object_space<FUSED, DW0>(o, ray, sx, sy, sz, dx, dy, dz, ty);
    if (type == 0u) return sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t);
    if (type == 1u) return box_candidate(sx, sy, sz, dx, dy, dz, t);
    return false;
}

// ... of an object whose record is in registers already (materialise: ObjRows)
template <bool FUSED, bool DW0>
__device__ __forceinline__ bool rows_candidate(const ObjRows& o, const Ray& ray, float& t, bool& sphere) {
    sphere = (o.type == 0u);
    if (o.type == 2u)
        return triangle_candidate(o.r0.x, o.r0.y, o.r0.z, o.r1.x, o.r1.y, o.r1.z, o.r2.x, o.r2.y, o.r2.z, o.r0.w, o.r1.w, o.r2.w,
                                  __uint_as_float(o.pad0), ray, t);
    const float sx = row4<FUSED>(o.r0.x, o.r0.y, o.r0.z, o.r0.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sy = row4<FUSED>(o.r1.x, o.r1.y, o.r1.z, o.r1.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sz = row4<FUSED>(o.r2.x, o.r2.y, o.r2.z, o.r2.w, ray.sx, ray.sy, ray.sz, ray.sw);
    float dx, dy, dz;
    if constexpr (DW0) {
        dx = row3<FUSED>(o.r0.x, o.r0.y, o.r0.z, ray.dx, ray.dy, ray.dz);
        dy = row3<FUSED>(o.r1.x, o.r1.y, o.r1.z, ray.dx, ray.dy, ray.dz);
        dz = row3<FUSED>(o.r2.x, o.r2.y, o.r2.z, ray.dx, ray.dy, ray.dz);
    } else {
        dx = row4<FUSED>(o.r0.x, o.r0.y, o.r0.z, o.r0.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dy = row4<FUSED>(o.r1.x, o.r1.y, o.r1.z, o.r1.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dz = row4<FUSED>(o.r2.x, o.r2.y, o.r2.z, o.r2.w, ray.dx, ray.dy, ray.dz, ray.dw);
    }
    if (o.type == 0u) return sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t);
    if (o.type == 1u) return box_candidate(sx, sy, sz, dx, dy, dz, t);
    return false;
}

// A ray with a NaN in it walks no cells; what the reference's loop ends with for it is nan_ray_outcome() (rt_device.h),
// patched in where the walk's result is produced (shadow rays) or consumed (closest_result, rt_wavefront.hip).
__device__ __forceinline__ bool nan_shadow_blocked(const Scene& S, const Ray& ray) {
    return nan_ray_outcome(ray, S.nan_winner, S.nan_winner_sphere) == kNanRayTimeNaN;
}

// 3-D DDA state for one ray. All of it is plain fp32 bookkeeping about WHICH cells to look at; it never feeds
// the intersection arithmetic.
struct Walk {
    int ix, iy, iz;
    int stepx, stepy, stepz;
    float tx, ty, tz;      // ray parameter at which the walk crosses the next x / y / z cell wall
    float dtx, dty, dtz;   // parameter advance per cell
    float t_enter;         // parameter at which the current cell was entered
    float t_exit;          // parameter at which the ray leaves the grid box (or reaches t_limit)
    bool alive;
};

// The walk's own arithmetic uses the hardware reciprocal (1 ulp) instead of IEEE division: t values are only used
// to order cell crossings, and their error (~1e-7 relative, i.e. < 1e-4 of a cell over the whole scene) sits far
// inside the 0.01-cell slack every registered radius carries for exactly this purpose.
template <typename GRID>
__device__ __forceinline__ Walk walk_begin(const GRID& g, const Ray& ray, float t_limit) {
    Walk w;
    w.alive = false;
    // the ray as a 3-D segment (the reference divides nothing by w here; start.w scales nothing in view space:
    // secondary rays have w = 1, pinhole primaries too; other values are handled by the caller)
    const float ox = ray.sx, oy = ray.sy, oz = ray.sz;
    const float dx = ray.dx, dy = ray.dy, dz = ray.dz;
    {   // NaN anywhere: no walk (fmin / fmax below would swallow it); callers give such rays the reference's result
        const float chk = ((ox + oy) + oz) + ((dx + dy) + dz);
        if (!(chk == chk)) return w;
    }
    const float hix = g.lox + g.cell * (float)g.nx, hiy = g.loy + g.cell * (float)g.ny, hiz = g.loz + g.cell * (float)g.nz;
    // slab clip against the grid box, [t0, t1] subset of [0, t_limit]
    float t0 = 0.f, t1 = t_limit;
    const float big = 3.0e38f;
    const float invx = __builtin_amdgcn_rcpf(dx), invy = __builtin_amdgcn_rcpf(dy), invz = __builtin_amdgcn_rcpf(dz);
    {
        float a = (g.lox - ox) * invx, b = (hix - ox) * invx;
        if (dx == 0.f) { a = (ox < g.lox || ox > hix) ? big : -big; b = (ox < g.lox || ox > hix) ? -big : big; }
        t0 = __builtin_fmaxf(t0, __builtin_fminf(a, b));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(a, b));
    }
    {
        float a = (g.loy - oy) * invy, b = (hiy - oy) * invy;
        if (dy == 0.f) { a = (oy < g.loy || oy > hiy) ? big : -big; b = (oy < g.loy || oy > hiy) ? -big : big; }
        t0 = __builtin_fmaxf(t0, __builtin_fminf(a, b));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(a, b));
    }
    {
        float a = (g.loz - oz) * invz, b = (hiz - oz) * invz;
        if (dz == 0.f) { a = (oz < g.loz || oz > hiz) ? big : -big; b = (oz < g.loz || oz > hiz) ? -big : big; }
        t0 = __builtin_fmaxf(t0, __builtin_fminf(a, b));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(a, b));
    }
    if (!(t0 <= t1)) return w;  // misses the grid box (or NaN): only the always-list applies
    const float px = ox + t0 * dx, py = oy + t0 * dy, pz = oz + t0 * dz;
    int ix = (int)__builtin_floorf((px - g.lox) * g.inv_cell);
    int iy = (int)__builtin_floorf((py - g.loy) * g.inv_cell);
    int iz = (int)__builtin_floorf((pz - g.loz) * g.inv_cell);
    ix = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= g.nz ? g.nz - 1 : iz);
    w.ix = ix; w.iy = iy; w.iz = iz;
    w.stepx = dx > 0.f ? 1 : -1;
    w.stepy = dy > 0.f ? 1 : -1;
    w.stepz = dz > 0.f ? 1 : -1;
    const float wallx = g.lox + g.cell * (float)(ix + (dx > 0.f ? 1 : 0));
    const float wally = g.loy + g.cell * (float)(iy + (dy > 0.f ? 1 : 0));
    const float wallz = g.loz + g.cell * (float)(iz + (dz > 0.f ? 1 : 0));
    w.tx = dx != 0.f ? (wallx - ox) * invx : big;
    w.ty = dy != 0.f ? (wally - oy) * invy : big;
    w.tz = dz != 0.f ? (wallz - oz) * invz : big;
    w.dtx = dx != 0.f ? g.cell * __builtin_fabsf(invx) : big;
    w.dty = dy != 0.f ? g.cell * __builtin_fabsf(invy) : big;
    w.dtz = dz != 0.f ? g.cell * __builtin_fabsf(invz) : big;
    w.t_enter = t0;
    w.t_exit = t1;
    w.alive = true;
    return w;
}

// step to the next cell; false when the walk leaves the grid
__device__ __forceinline__ bool walk_next(const GridDesc& g, Walk& w) {
    if (w.tx <= w.ty && w.tx <= w.tz) {
        w.ix += w.stepx; w.t_enter = w.tx; w.tx += w.dtx;
        return (unsigned)w.ix < (unsigned)g.nx;
    } else if (w.ty <= w.tz) {
        w.iy += w.stepy; w.t_enter = w.ty; w.ty += w.dty;
        return (unsigned)w.iy < (unsigned)g.ny;
    } else {
        w.iz += w.stepz; w.t_enter = w.tz; w.tz += w.dtz;
        return (unsigned)w.iz < (unsigned)g.nz;
    }
}

// The same walk in the form the persistent kernel carries per lane: a running cell index and per-axis step
// budgets instead of three coordinates, and a branch-free step (same axis choice, ties x before y before z, so the
// cells and their order are those of walk_next).
struct LeanWalk {
    int c;                 // current cell (linear index)
    int rx, ry, rz;        // steps left along each axis before the walk leaves the grid
    float tx, ty, tz;      // ray parameter at the next x / y / z cell wall
    float dtx, dty, dtz;   // parameter advance per cell (>= 0)
    int sx, sy, sz;        // what a step along x / y / z adds to the cell index (+-1, +-nx, +-nx ny): three registers that
                           // save three compares and three selects with scalar operands per step (the walk has them
                           // to spare at 6 waves per SIMD; a compare or a select costs ~2x a plain vector instruction)
    float t_enter;
};

__device__ __forceinline__ LeanWalk lean_walk(const GridDesc& g, const Walk& w) {
    LeanWalk k;
    k.c = (w.iz * g.ny + w.iy) * g.nx + w.ix;
    k.rx = w.stepx > 0 ? g.nx - 1 - w.ix : w.ix;
    k.ry = w.stepy > 0 ? g.ny - 1 - w.iy : w.iy;
    k.rz = w.stepz > 0 ? g.nz - 1 - w.iz : w.iz;
    k.tx = w.tx; k.ty = w.ty; k.tz = w.tz;
    k.dtx = w.dtx; k.dty = w.dty; k.dtz = w.dtz;
    k.sx = w.stepx > 0 ? 1 : -1;
    k.sy = w.stepy > 0 ? g.nx : -g.nx;
    k.sz = w.stepz > 0 ? g.nx * g.ny : -(g.nx * g.ny);
    k.t_enter = -1.0f;  // "no cell before this one" (entered_inside); the walk only reads t_enter of cells it has stepped into
    return k;
}

__device__ __forceinline__ bool lean_next(const GridDesc& g, LeanWalk& k) {
    // written with plain selects of 0 / value so that the fields stay in registers (an indexed pick of
    // tx/ty/tz makes the compiler move the struct to LDS)
    const float tmin = __builtin_fminf(__builtin_fminf(k.tx, k.ty), k.tz);
    const bool ax = (k.tx <= k.ty) && (k.tx <= k.tz);
    const bool ay = !ax && (k.ty <= k.tz);
    const bool az = !ax && !ay;
    k.t_enter = tmin;
    k.tx += ax ? k.dtx : 0.f;
    k.ty += ay ? k.dty : 0.f;
    k.tz += az ? k.dtz : 0.f;
    k.c += ax ? k.sx : (ay ? k.sy : k.sz);
    k.rx -= ax ? 1 : 0;
    k.ry -= ay ? 1 : 0;
    k.rz -= az ? 1 : 0;
    return (k.rx | k.ry | k.rz) >= 0;
}

// Closest hit through the grid. Same (T, index) as closest_hit() over all objects.
template <bool FUSED, bool DW0>
__device__ __forceinline__ void closest_hit_grid(const GridDesc& g, const HotObject* __restrict__ hot, const Ray& ray, float& T,
                                                 int& index, uint32_t& tested) {
    bool cur_sphere = false;
    tested = g.n_always;
    for (uint32_t a = 0; a < g.n_always; ++a) {
        const int k = (int)g.always[a];
        float t;
        bool sphere;
        if (lane_candidate<FUSED, DW0>(hot + k, ray, t, sphere)) closest_take(t, k, sphere, T, index, cur_sphere);
    }
    Walk w = walk_begin(g, ray, 3.0e38f);
    if (!w.alive) return;
    // two more cells after the best hit's cell (the candidate's own t is exact; the slack covers objects that
    // start in the next cells but whose computed t the reference may place marginally earlier)
    const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    const float len = __builtin_sqrtf(dd);
    const float slack = len > 0.f ? kWalkSlackCells * g.cell / len : 3.0e38f;
    for (;;) {
        const uint32_t c = ((uint32_t)w.iz * (uint32_t)g.ny + (uint32_t)w.iy) * (uint32_t)g.nx + (uint32_t)w.ix;
        const uint2 range = g.cell_range[c];
        const uint32_t e0 = range.x, e1 = range.y ? range.x + range.y : range.x;
        for (uint32_t e = e0; e < e1; ++e) {
            if (DW0 && misses_bounding_sphere(g.entry_sphere[e], ray, dd, g.pretest_alpha)) continue;  // the walk kernels' 16-byte pre-test (conservative)
            const int k = (int)g.entries[e];
            float t;
            bool sphere;
            ++tested;
            if (lane_candidate<FUSED, DW0>(hot + k, ray, t, sphere)) closest_take(t, k, sphere, T, index, cur_sphere);
        }
        // an empty cell's offset word says how many FURTHER steps are sure to land in empty cells (build_grid): those cells are
        // stepped through without fetching them
        uint32_t steps = range.y ? 1u : 1u + range.x;
        bool stop = false;
        const float limit = T + slack;  // T is +MAX until something is hit
        while (steps-- != 0u && !stop) stop = !walk_next(g, w) || w.t_enter > limit;
        if (stop) break;
    }
}

// Any hit with t < 1 through the grid, one thread per ray (the tail of a frame: wf_finish). Same answer as the
// brute-force any-hit loops: every candidate that can occlude is registered in a cell the walk visits.
template <bool FUSED>
__device__ __forceinline__ bool any_hit_grid(const GridDesc& g, const Scene& S, const Ray& ray, uint32_t& tested) {
    const HotObject* __restrict__ hot = S.hot;
    for (uint32_t a = 0; a < g.n_always; ++a) {
        float t;
        bool sphere;
        ++tested;
        if (lane_candidate<FUSED, true>(hot + (int)g.always[a], ray, t, sphere) && !(t >= 1.f)) return true;
    }
    const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    const float len = __builtin_sqrtf(dd);
    const float slack = len > 0.f ? kWalkSlackCells * g.cell / len : 3.0e38f;
    Walk w = walk_begin(g, ray, 1.0f + slack);
    if (!w.alive) return nan_shadow_blocked(S, ray);  // off the grid: nothing in the way - unless the ray is a NaN ray
    for (;;) {
        const uint32_t c = ((uint32_t)w.iz * (uint32_t)g.ny + (uint32_t)w.iy) * (uint32_t)g.nx + (uint32_t)w.ix;
        const uint2 range = g.cell_range[c];
        for (uint32_t e = range.x; e < range.x + range.y; ++e) {
            if (misses_bounding_sphere(g.entry_sphere[e], ray, dd, g.pretest_alpha)) continue;
            float t;
            bool sphere;
            ++tested;
            if (lane_candidate<FUSED, true>(hot + (int)g.entries[e], ray, t, sphere) && !(t >= 1.f)) return true;
        }
        uint32_t steps = range.y ? 1u : 1u + range.x;  // (as in closest_hit_grid)
        bool stop = false;
        while (steps-- != 0u && !stop) stop = !walk_next(g, w) || w.t_enter > 1.0f + slack;
        if (stop) break;
    }
    return false;
}

}  // namespace rt

