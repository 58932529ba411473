// mrtx_device.h -- data shared by the host side of libmoonrt.so and its gfx950 kernels.
//
// Per-launch constants, derived once on the host in float64 from the calls the reference makes on its renderer
// object (moon_renderer.py:570-650 for the static scene, :824-871 for the per-time-step update).
//   FrameC    -- what the march loops touch, passed to the kernel BY VALUE -> SGPRs (wave-uniform scalar
//                loads, no VGPR cost);
//   FrameCold -- what a sample touches once (camera / moon-frame matrices in float64, light, Sun disk, colour and
//                environment grids), kept in device memory behind FrameC::cold and read with scalar loads at the
//                point of use.  Keeping these out of the by-value block stops the compiler from holding ~150
//                SGPRs live from kernel entry (it spilled 81 of them into VGPR lanes and moved them back and forth
//                with v_readlane / v_writelane around every sample).
#pragma once
#include <stdint.h>

struct GridC {          // equirectangular grid: row 0 = +90 deg, column 0 = -180 deg
    int32_t h, w;       // (renderer_navigation.py:575-593)
    float row_scale, row_off, col_scale, col_off, wf;
};

// DEM layout in HBM (see dem_march() in mrtx_kernels.hip): 1 = row pairs, element (r, c) = float2 (D[r][c], D[r+1][c]),
// so the 2x2 footprint of a bilinear evaluation is ONE 16-byte load; 0 = plain padded float32, two 8-byte loads.
#ifndef MRTX_DEM_PAIRS
#define MRTX_DEM_PAIRS 1
#endif
#define MRTX_DEM_ELEM_BYTES (MRTX_DEM_PAIRS ? 8 : 4)

// Tile numbering.  Tile t belongs to rank t % world, and t counts the tiles in raster order with a cyclic shift of
// `shift` columns per tile row: t = ty * tiles_x + (tx + shift * ty) % tiles_x.  With plain raster numbering and
// tiles_x a multiple of world (120 columns at 3840 px, world 2/4/8) every rank would own whole tile COLUMNS, which line
// up with vertical features (the terminator at the default orientation): 12 % rank imbalance at world 8.  The shift
// -- the smallest odd number >= 3 coprime to world -- turns the ownership into a 2-D lattice.
static inline __host__ __device__ void mrtx_tile_xy(int t, int tiles_x, int shift, int& tx, int& ty) {
    ty = t / tiles_x;
    const int c = t % tiles_x - (int)(((long long)shift * ty) % tiles_x);
    tx = c < 0 ? c + tiles_x : c;
}
static inline __host__ __device__ int mrtx_tile_id(int tx, int ty, int tiles_x, int shift) {
    return ty * tiles_x + (int)((tx + (long long)shift * ty) % tiles_x);
}
static inline int mrtx_tile_shift(int world) {
    for (int s = 3;; s += 2) {
        int a = s, b = world;
        while (b) { const int r = a % b; a = b; b = r; }
        if (a == 1) return s;
    }
}

// path_kernel's medium-mip step mask (mrtx_kernels.hip: step_mask): 1 builds and uses it, 0 (default: it does not pay, see there) neither
#ifndef MRTX_PATH_MIP2
#define MRTX_PATH_MIP2 0
#endif
// The medium mip also cuts the skip interval of a segment from one end, with tests that stop at the first inconclusive step
// (first_kept_step / last_kept_step in mrtx_kernels.hip; round 4): bit 2 (4) camera rays, from the front: render 15.4 -> 13.9 ms at cfg3;
// bit 1 (2) shadow rays in render_kernel, from the end: -> 13.8; bit 3 (8) path_kernel's marches, from the end: path stage 5.0 -> 4.65;
// bit 0 (1) the trial segment, from the end: +0.17 ms, off.  The host builds the medium mip when either switch is set.
#ifndef MRTX_M2_DELTA
#define MRTX_M2_DELTA 2      // medium-mip cell = max-mip cell >> MRTX_M2_DELTA (never below 4 texels)
#endif
#ifndef MRTX_SEG_MASK
#define MRTX_SEG_MASK 14
#endif

struct FrameCold {
    // D1 pinhole camera (moon_renderer.py:627-635)
    float Wd[3], Ux[3], Vy[3], two_over_w, two_over_h;
    double oc[3], cq;       // eye - centre, |oc|^2 - R^2
    double M[3][3];         // scene -> moon frame, rows = (east 90, lon 0, north)
    float Mf[3][3], centerf[3], eyef[3];
    // D5 light (moon_renderer.py:640-641, :859-860)
    float Lb[3], rL2, rad2;
    // D8 Sun disk (moon_renderer.py:647-650)
    int32_t sun_on;
    float sc[3], sun_cq, sun_rad;
    float Sb[3], sun_r2;    // disk centre in the moon frame (relative to the Moon centre), radius^2: continuation rays see it
    float eps, scene_eps;
    float dlat_scale, dlon_scale;
    GridC gc;
    int32_t bg_h, bg_w;
    float bg_row_scale, bg_row_off, bg_col_scale, bg_col_off;
    uint32_t key0;
    uint32_t path_seg_min, path_seg_max;   // D6: path length in segments (camera segment = 1); max 1 = direct only
    float const_albedo[3];
    const uint8_t* color;   // RGBA8 or null
    const uint8_t* bg;      // RGBA8 or null
    // D11 overlay tubes: 12 floats per capsule (a - centre, r, b - centre, 0, colour, 0) + per-local-tile CSR bins
    const float* caps;
    const int32_t* caps_off;   // n_local_tiles + 1
    const int32_t* caps_idx;
    int32_t n_caps;
    // output buffers: touched once per pixel, at the end of a wave's life -- kept out of the by-value block so that they do
    // not occupy SGPRs (and get spilled) across the march loops
    float* accum;           // W*H float4: running sums r,g,b,coverage
    float* hits;            // W*H float4: x,y,z,d of sample 0 of the last block
    unsigned long long* stats;  // 10 counters, see MrtxStats (render_kernel's; [15] = path_kernel's watchdog)
    unsigned long long* stats_paths;   // the same counters as path_kernel adds them (MrtxStats::camera_* = the render kernel's share)
    // horizon mip (horizon_kend in mrtx_kernels.hip): hm_h x hm_w cells of 2^hm_shift texels, or null
    const float* hmip;
    int32_t hm_h, hm_w, hm_shift;
    float hm_cell;                  // cell size in texels
    float hm_krow, hm_kcol;         // angle -> texel rows (h / pi); angle -> texel columns (1.05 w / 2 pi)
    // medium max-mip (path_kernel's step mask, round 4): cells of 2^m2_shift texels (a quarter of the max-mip's), plain floats with a
    // one-cell border like the max-mip before pairing -- (m2_h + 2) x (m2_w + 2) -- or null
    const float* mip2;
    int32_t m2_pitch, m2_h, m2_w, m2_shift;
};

struct FrameC {
    int32_t W, H;
    float Rf, R2f;
    // D2 march (moon_renderer.py:586-588)
    float step, inv_step;
    float polar_rho2, row_hi, col_hi;   // segment fallback threshold (0.04 R^2); largest floats below h / w
    int32_t nbis, kmax;
    GridC gd;
    const float* dem;       // PADDED (h+4) x (w+4): element [0] is (row -2, col -2); see dem_march()
    int32_t dem_pitch, dem_wide;   // pitch = w+4 floats; wide = byte offsets need 64 bits (> 4 GiB)
    uint32_t dem_maxidx;           // (h+2)*pitch + (w+2): last padded index a 2x2 tap may start at
    const float* mip;       // max-mip, (mip_h+2) x (mip_w+2) incl. its border, or null (skipping disabled)
    int32_t mip_pitch, mip_h, mip_w, mip_shift;   // cell = 2^mip_shift texels
    const FrameCold* cold;  // device memory
    // image-tile sharding (new) + accumulation state
    int32_t tile_w, tile_h, tiles_x, tiles_y, rank, world, n_local_tiles;
    int32_t tile_shift;         // see mrtx_tile_xy()
    const int32_t* tile_list;   // local tile indices to render (sky tiles culled on the host), or null = all
    int32_t n_active;           // entries of tile_list (== n_local_tiles when null)

    int32_t xcd_share;      // 1: every tile is shared by the 8 XCDs (few tiles per launch), 0: whole tiles per XCD
    uint32_t first_block, n_blocks;
};

// Hand-over between render_kernel<MODE 2> (camera ray, first vertex, its direct light, the decision to go on) and
// path_kernel (everything after), one RECORD per lane of every wave-job ("chunk") of the render launch:
//   ray0/1/2  float4 each, only for the samples whose path goes on, compacted to the front of the chunk (npaths[chunk] of
//             them, lane_of[] says whose): continuation-ray origin (3), direction (3), path throughput (3), exact DEM texel
//             coordinates (2) of the point the march goes on from, RNG key of the sample (1)
//   lane_of   bits 0-5 the lane (sample) of the chunk the record belongs to, and what render_kernel's trial segment found
//             out about the ray: bit 6 = still marching after segment 1 (bits 8-31 = its horizon bound kend, the texel
//             coordinates are those of the END of segment 1), bit 7 = hit inside segment 1 (bits 8-31 = the step k that
//             landed at/below the surface, the coordinates are the origin's); neither = march from the origin
//   c4 (c0/1/2 with MRTX_C_AOS = 0)  the sample's radiance so far (direct term / Sun disk / environment / overlay colour); path_kernel writes
//             the final value back when the path adds light, resolve_paths_kernel sums the 64 lanes in the butterfly
//             order of the spec
//   meta      per chunk: bit 31 = the chunk was deferred, bits 0-14 / 15-29 = pixel (x0, y0) of the wave's pixel block
// A wave writes 1 KB (ray*) / 256 B (c*) contiguous per array.
struct PathQ {
    float4* ray0; float4* ray1; float4* ray2;
    float* c0; float* c1; float* c2;
    float4* c4;                 // MRTX_C_AOS 1: the three of them as one float4 per sample (one 16-byte access instead of three sectors); 2: packed, 12 bytes per sample
    uint32_t* lane_of;          // per ray record: lane + the state of its march after the trial segment (see above)
    uint8_t* npaths;            // per chunk: ray records it holds (0 for a chunk that was not deferred): zero before the launch
    uint32_t* meta;
    uint32_t n_chunks;          // wave-jobs of the render launch (grid x jobs per wave)
    uint32_t grid_a;            // blocks of the render launch; chunk = block * jobs + job
    int32_t njobs_log2;         // jobs (pixel blocks) per render wave: 1 or 2
    uint32_t* counters;         // path_kernel's work counters: 8 XCDs x n_sub, zero before the launch
    int32_t n_sub, grp_log2;    // counters per XCD; render blocks per group handed out = 1 << grp_log2
    uint32_t gs_base;           // global sample index of sample 0 of this block (first_block * S)
    int32_t s_log2, pw_log2;    // S = 1 << s_log2 samples per pixel in a wave, pixel block PW x PH, PW = 1 << pw_log2
    int32_t refill_min;         // path_kernel refills its idle lanes when at least this many are idle
    int32_t seg_min;            // ... sets up march segments when at least this many lanes need one
    int32_t rare_min;           // ... and runs the rare steps (a continuation ray hit terrain; a vertex got its direct
                                //     term) when at least this many lanes wait for them
};
// The running radiance of a sample as ONE float4 per sample (round 3): a path that adds light reads and writes one 64-byte sector
// instead of three in three arrays -- path stage 5.41 -> 5.19 ms at cfg3, render and resolve unchanged (0 = the three float arrays)
// 2 (round 4): the three floats PACKED, 12 bytes per sample -- resolve_paths_kernel streams a quarter less, the render kernel writes a
// quarter less, a path's read-modify-write still touches one sector five times in eight: path stage 4.35 -> 4.28 ms, frame -0.12 ms
#ifndef MRTX_C_AOS
#define MRTX_C_AOS 2
#endif
#define MRTX_PATH_REC_BYTES (MRTX_C_AOS == 1 ? 68 : 64)  // per record: 3 x float4 + 3 x float (or one float4) + 1 word
#define MRTX_REC_RESUME 64u
#define MRTX_REC_HIT 128u
