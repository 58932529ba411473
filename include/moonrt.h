/*
 * moonrt.h -- C ABI of libmoonrt.so, the MI355X (gfx950) renderer that replaces the
 * PlotOptiX/OptiX backend behind MoonRTX's `self.rt` object.
 *
 * The reference has no C FFI: its renderer boundary is the Python object created at
 * moonrtx/moon_renderer.py:571-575 (`TkOptiX(width, height, on_launch_finished=...)`).
 * Every entry point below cites the reference call(s) on that object which it serves; the
 * Python facade `moonrtx_amd/tkoptix.py` maps those calls 1:1 onto this ABI through ctypes.
 *
 * Conventions
 *   - plain C types only; host pointers are borrowed for the duration of the call;
 *   - every function returns 0 on success or a negative MRTX_E_* code; the text of the last
 *     failure on a context is available from mrtx_last_error();
 *   - no C++ exception crosses the ABI and nothing aborts the process;
 *   - one context is driven by one thread at a time (the facade's render thread holds the
 *     `_padlock` while it calls in, moon_renderer.py:849-852).
 */
#ifndef MOONRT_H
#define MOONRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRTX_ABI_VERSION 7

enum {
    MRTX_OK = 0,
    MRTX_E_INVALID = -1,   /* bad argument                                   */
    MRTX_E_DEVICE = -2,    /* a HIP call failed (text in mrtx_last_error)    */
    MRTX_E_STATE = -3,     /* call made in the wrong state (e.g. no DEM yet) */
    MRTX_E_NOMEM = -4
};

typedef struct mrtx_ctx mrtx_ctx;

/* Frame + sharding description.  rank/world shard the image in tiles of tile_w x tile_h
 * pixels: tile t belongs to rank (t % world), where t numbers the tiles in raster order with a cyclic shift of s
 * columns per tile row -- t = ty * tiles_x + (tx + s * ty) % tiles_x, s = the smallest odd number >= 3 coprime to
 * world -- so that a rank's tiles form a 2-D lattice, not whole columns.  world == 1 renders everything. */
typedef struct MrtxConfig {
    int32_t device;          /* HIP device ordinal                                             */
    int32_t width, height;   /* TkOptiX(width=, height=)            moon_renderer.py:571-573   */
    int32_t rank, world;     /* image-tile sharding (new; the reference is single-GPU)        */
    int32_t tile_w, tile_h;  /* sharding / culling tile (multiples of 16), 0 = default: 16 x 16 for world <= 2, 32 x 32 else */
} MrtxConfig;

/* String-keyed knobs of the reference collapsed into one POD.
 *   set_float("scene_epsilon" | "marching_step" | "marching_step_eps")  moon_renderer.py:586-588
 *   set_float("tonemap_exposure" | "tonemap_gamma")                      moon_renderer.py:598-599, :367
 *   set_uint("path_seg_range", 2, 4)                                      moon_renderer.py:583
 *   set_param(min_accumulation_step=, max_accumulation_frames=)           moon_renderer.py:578, :475, :487 */
typedef struct MrtxParams {
    float scene_epsilon;
    float marching_step;
    float marching_step_eps;
    float tonemap_exposure;
    float tonemap_gamma;
    uint32_t path_seg_min, path_seg_max;   /* path length in ray segments, camera segment = 1; max <= 1 traces direct
                                              light only; beyond `min` the path survives Russian roulette            */
    uint32_t spp_per_launch;   /* samples per pixel per accumulation block: 1,2,4,...,64 */
    uint32_t max_spp;          /* max_accumulation_frames (informational for the ABI)   */
    uint32_t seed;
    float const_albedo[3];     /* reflectance used when no colour texture is bound       */
    uint32_t flags;            /* MRTX_F_*                                                */
} MrtxParams;

#define MRTX_F_COUNT_STATS 1u  /* maintain the deterministic sample counters of MrtxStats */
#define MRTX_F_NO_SKIP     4u  /* evaluate every march step (disable the result-preserving max-mip skip) */
#define MRTX_F_NO_CULL     8u  /* dispatch every tile (disable the host-side sky-tile cull) */
#define MRTX_F_NO_SORT     16u /* dispatch tiles in raster order (disable the limb-ring-first launch order) */
#define MRTX_F_FORCE_WIDE  2u  /* test hook: use the 64-bit DEM addressing path (normally only for DEMs > 4 GiB) */
#define MRTX_F_INWAVE_PATHS 32u /* path_seg_max > 1: keep the whole path inside the wave that traced the camera ray instead
                                   of continuing it behind a queue in path_kernel (same result bit for bit; A/B switch) */

typedef struct MrtxStats {
    uint64_t primary_rays;        /* camera samples (pixel x spp), the headline "ray"      */
    uint64_t primary_hits;        /* camera samples that hit the Moon                      */
    uint64_t shadow_rays;         /* light-sample rays marched                             */
    uint64_t height_samples;      /* DEM bilinear evaluations the march DEFINES (16 B each): every step of every
                                     segment until hit / exit, bisection, normal taps -- equals the oracle's count */
    uint64_t colour_fetches;      /* colour bilinear fetches (16 B each)                   */
    uint64_t background_fetches;  /* environment texel fetches (4 B each)                  */
    uint64_t dem_fetches;         /* DEM bilinear evaluations actually PERFORMED (steps the max-mip bound proves to
                                     be above the terrain are skipped; results are unchanged)                  */
    uint64_t mip_fetches;         /* max-mip texels read for those bounds (4 B each)                           */
    uint64_t bounce_rays;         /* D6 path-continuation rays marched (0 when path_seg_max <= 1)              */
    uint64_t bounce_sun_hits;     /* ... of which left the Moon and ended on the visible Sun disk              */
    double kernel_ms;             /* HIP-event time of the kernels of this call = primary_ms + paths_ms        */
    double primary_ms;            /* render_kernel: camera ray, first vertex, its direct light                 */
    double paths_ms;              /* path_kernel + resolve_paths_kernel: everything after the first vertex
                                     (0 unless path_seg_max > 1 and the queue-based path stage is in use)      */
    uint32_t launches;
    uint32_t reserved;
    /* ABI 7: the share of the counters above that render_kernel (the camera-ray stage: camera ray, first vertex, its shadow ray,
     * and with the path queue the FIRST segment of the continuation ray) performed itself; the rest was performed by path_kernel.
     * With the paths inside the render wave (or path_seg_max <= 1) these equal the totals.  bench.py prices each kernel's
     * algorithmic bytes with its own counts. */
    uint64_t camera_height_samples, camera_dem_fetches, camera_mip_fetches, camera_colour_fetches, camera_background_fetches;
} MrtxStats;

/* TkOptiX(width, height, ...) -- moon_renderer.py:571-575.  Allocates accumulation + hit buffers. */
int mrtx_create(const MrtxConfig* cfg, mrtx_ctx** out);
/* rt.close() -- moon_renderer.py:880-884 */
void mrtx_destroy(mrtx_ctx* ctx);
const char* mrtx_last_error(mrtx_ctx* ctx);
int mrtx_abi_version(void);
/* The configuration the context runs with: what mrtx_create was given, with the defaults filled in (tile_w / tile_h: 16 x 16
 * for world <= 2, 32 x 32 from four ranks up) -- so that a caller reports the tiling that was actually used. */
int mrtx_get_config(mrtx_ctx* ctx, MrtxConfig* out);

/* rt.set_displacement("moon", elevation, refresh=False) -- moon_renderer.py:624.
 * `host` is the float32 (h, w) array load_elevation_data returns (data_loader.py:166-247):
 * equirectangular, row 0 = +90 deg, column 0 = -180 deg, max exactly 1.0. */
int mrtx_upload_dem(mrtx_ctx* ctx, const float* host, int32_t h, int32_t w);
/* Same, from a device pointer (synthetic / device-built DEMs).  Either way the context keeps its OWN copy,
 * re-laid-out in row pairs -- element (r, c) = float2 (D[r][c], D[r+1][c]) -- with a two-texel border (wrap in longitude,
 * clamp in latitude), so a bilinear evaluation on the march path is ONE unconditional 16-byte load; 8 bytes per texel.
 * The caller may free its buffer when the call returns. */
int mrtx_bind_dem_device(mrtx_ctx* ctx, const void* dev_f32, int32_t h, int32_t w);

/* rt.set_texture_2d("moon_color", rgba_u8) + update_material("diffuse", {"ColorTextures": [...]})
 * -- moon_renderer.py:613-617.  NULL => const_albedo of MrtxParams. */
int mrtx_upload_color(mrtx_ctx* ctx, const uint8_t* rgba, int32_t h, int32_t w);
/* Same, from a device pointer; the context keeps its own row-pair copy (8 bytes per texel), the caller's array is read once. */
int mrtx_bind_color_device(mrtx_ctx* ctx, const void* dev_rgba8, int32_t h, int32_t w);

/* rt.set_background_mode("TextureEnvironment"); rt.set_background(star_map, gamma=, rt_format="UByte4")
 * -- moon_renderer.py:604-609.  NULL => black (`set_background(0)`). Texels are RGBA8, already in the
 * renderer's linear space (the facade applies the gamma of set_background on the host). */
int mrtx_upload_background(mrtx_ctx* ctx, const uint8_t* rgba, int32_t h, int32_t w);

/* add_postproc("Overlay") + set_texture_2d("frame_overlay", rgba, filter_mode="Nearest") -- renderer_video.py:137-144:
 * a frame-sized RGBA8 texture alpha-blended over the tone-mapped image in mrtx_read_rgba8 (exact: 50 % black over 46
 * reads 23, renderer_video.py:21-25).  NULL removes it. */
int mrtx_upload_overlay(mrtx_ctx* ctx, const uint8_t* rgba, int32_t h, int32_t w);

/* set_float / set_uint / set_param / set_ambient / add_postproc -- moon_renderer.py:578-600 */
int mrtx_set_params(mrtx_ctx* ctx, const MrtxParams* p);
void mrtx_default_params(MrtxParams* p);

/* setup_camera / update_camera / _optix.set_camera_fov -- moon_renderer.py:627-635,
 * renderer_navigation.py:73,150,224,521.  Pinhole; vfov is the vertical field of view, degrees. */
int mrtx_set_camera(mrtx_ctx* ctx, const double eye[3], const double target[3], const double up[3],
                    double vfov_deg);

/* set_data("moon", geom="ParticleSetTextured", geom_attr="DisplacedSurface", pos, u, v, r) and
 * update_data("moon", u=, v=) -- moon_renderer.py:620-621, :854.  u = north pole, v = direction of
 * longitude 0 (renderer_navigation.py:47-53, :486-490), both in scene coordinates. */
int mrtx_set_moon_frame(mrtx_ctx* ctx, const double center[3], double radius, const double u[3],
                        const double v[3]);

/* setup_light("sun", color=, radius=, in_geometry=False) / update_light(pos=, color=, radius=)
 * -- moon_renderer.py:640-641, :347, :859-860.  `radiance` is the light colour (scalar, white). */
int mrtx_set_light(mrtx_ctx* ctx, const double pos[3], double radius, double radiance);

/* set_data("sun_disk", geom="ParticleSet", mat="flat", pos, r, c=2.0) / update_data(...)
 * -- moon_renderer.py:647-650, :855.  radius <= 0 disables the disk. */
int mrtx_set_sun_disk(mrtx_ctx* ctx, const double pos[3], double radius, double radiance);

/* set_graph / update_graph / delete_geometry -- renderer_labels.py:295-300, :367-373, renderer_pins.py:54: ALL overlay
 * graphs flattened into capsules, 12 floats each (ax ay az r  bx by bz 0  cr cg cb 0), scene coordinates, flat colour;
 * they never shadow and are invisible to shadow / continuation rays (renderer_labels.py:132-139).  n = 0 removes them. */
int mrtx_set_capsules(mrtx_ctx* ctx, const float* caps12, int32_t n);

/* rt.refresh_scene() -- moon_renderer.py:488, :871: restart the accumulation cycle. */
int mrtx_reset_accum(mrtx_ctx* ctx);

/* One or more accumulation blocks (what the PlotOptiX render thread does between two
 * on_launch_finished callbacks, moon_renderer.py:574; renderer_status.py:239).  Each block adds
 * spp_per_launch samples to every pixel this rank owns.  Blocking.  `out` may be NULL. */
int mrtx_render(mrtx_ctx* ctx, int32_t n_blocks, MrtxStats* out);
/* The same block of samples, one part of the tile list at a time (parts 0 .. n_parts-1 in order; the sample counter
 * advances with the last one), so that the exchange can move part k while part k+1 renders: see mrtx_pack_part. */
int mrtx_render_part(mrtx_ctx* ctx, int32_t n_blocks, int32_t part, int32_t n_parts, MrtxStats* out);

/* Read-back.  All are full-frame W*H arrays, caller-allocated; on a sharded context pixels of
 * other ranks read as zero.
 *   linear : float32 RGBA, mean linear radiance, A = 1 where any sample hit geometry
 *   rgba8  : the "Gamma" post-process (exposure * L)^(1/gamma) -> 8 bit, moon_renderer.py:598-600
 *   hits   : float32 (x, y, z, d) in scene coordinates, d <= 0 == miss -- rt._get_hit_at(x, y),
 *            moon_renderer.py:1138, renderer_navigation.py:195-203 */
int mrtx_read_linear(mrtx_ctx* ctx, float* rgba_out);
int mrtx_read_rgba8(mrtx_ctx* ctx, uint8_t* out);
/* rt.save_image(path, bps="Bps16") -- renderer_dialogs.py:1222-1224 (".tiff" is saved with 16 bits per sample): the same
 * exposure + "Gamma" post-process at 16 bits, W*H*3 uint16 (RGB interleaved), caller-owned.  Both tone-mapped read-backs are
 * exact: level = round(N * (exposure * mean)^(1/gamma)) decided by comparing with N float32 thresholds
 * (float)pow((j - 0.5) / N, gamma) built on the host, so they equal the oracle's bytes. */
int mrtx_read_rgb16(mrtx_ctx* ctx, uint16_t* out);
int mrtx_read_hits(mrtx_ctx* ctx, float* xyzd_out);
/* One texel of the hit buffer (16 bytes over PCIe): what rt._get_hit_at(x, y) needs per mouse event
 * (moon_renderer.py:1137-1142) without pulling the 133 MB buffer of a 4K frame after every launch. */
int mrtx_read_hit(mrtx_ctx* ctx, int32_t x, int32_t y, float xyzd_out[4]);
int mrtx_samples_done(mrtx_ctx* ctx, uint32_t* out);

/* ---- multi-GPU exchange step (new: the reference is single-GPU) -------------------------------
 * A rank packs the tiles it owns (linear float4 radiance followed by float4 hits) into a compact
 * device buffer the caller provides (e.g. a torch tensor handed to an RCCL gather), and rank 0
 * scatters the gathered buffers back into frame order. */
/* Whether the hit buffer travels with the radiance (default 1: a packed slot is one tile of float4 sums followed by one tile of
 * float4 hits, 32 B per pixel).  0: sums only, 16 B per pixel -- the exchange moves the final linear framebuffer and nothing
 * else; the root's hit buffer then holds its own tiles only, and a pick (rt._get_hit_at(x, y), one texel per mouse event,
 * moon_renderer.py:1137-1142) is served by mrtx_read_hit on the rank that owns the pixel (moonrtx_amd/dist.py: FrameGather.hit_at).
 * Must be set alike on every rank; changes mrtx_shard_bytes* and the layout pack / unpack use. */
int mrtx_set_gather_hits(mrtx_ctx* ctx, int32_t on);
int mrtx_shard_bytes(mrtx_ctx* ctx, int32_t rank, uint64_t* out);   /* upper bound of a packed buffer (all tiles) */
/* Bytes the exchange moves per rank for the scene as it stands.  While the host-side sky cull is in force (no
 * environment map, no overlay geometry, MRTX_F_NO_CULL clear) only the tiles the cull keeps travel: every rank
 * derives every rank's tile list from its own copy of the scene -- identical on all ranks by contract -- so the
 * layout is never negotiated.  Equal on every rank (padded to the longest list), <= mrtx_shard_bytes().
 * pack/unpack below use this layout; a buffer of mrtx_shard_bytes() is always large enough. */
int mrtx_shard_bytes_active(mrtx_ctx* ctx, uint64_t* out);
int mrtx_pack_shard(mrtx_ctx* ctx, void* dev_dst, void* hip_stream);
/* The shard in parts.  A packed shard is [slot][sums tile, hits tile], so a range of slots is one contiguous piece;
 * part k of n_parts covers the same slot numbers on every rank (the tiles mrtx_render_part(.., k, n_parts) rendered),
 * is written at byte_off .. byte_off + byte_len of dev_dst, and can be handed to the collective while part k+1
 * still renders.  mrtx_shard_parts() says how many parts the scene allows (1 while the full layout is in force). */
int mrtx_shard_parts(mrtx_ctx* ctx, int32_t wanted, int32_t* out);
int mrtx_pack_part(mrtx_ctx* ctx, void* dev_dst, int32_t part, int32_t n_parts, uint64_t* byte_off, uint64_t* byte_len,
                   void* hip_stream);
int mrtx_unpack_shard(mrtx_ctx* ctx, int32_t src_rank, const void* dev_src, void* hip_stream);
/* Same for every peer at once: dev_srcs[r] is rank r's packed buffer (the own rank's entry is ignored); one
 * synchronisation.  Tiles of peers that held an earlier view's data and are sky in this one are zeroed. */
int mrtx_unpack_all(mrtx_ctx* ctx, const void* const* dev_srcs, int32_t n);

/* Raw device pointers of the context's buffers (for zero-copy wrapping by the host side). */
/* MRTX_BUF_DEM is the context's own copy of the displacement map in its march layout: (h+4) x (w+4) elements of
 * float2 (D[r][c], D[r+1][c]), two-texel border (rows clamp, columns wrap) -- not the array that was uploaded.
 * MRTX_BUF_COLOR likewise: (h+1) x (w+4) elements of two RGBA8 texels (T[r][c], T[r+1][c]).  mrtx_bind_dem_device and
 * mrtx_bind_color_device read the caller's device array once; the caller may free it afterwards. */
enum { MRTX_BUF_ACCUM = 0, MRTX_BUF_HITS = 1, MRTX_BUF_DEM = 2, MRTX_BUF_COLOR = 3 };
int mrtx_device_ptr(mrtx_ctx* ctx, int32_t which, void** out, uint64_t* bytes);

/* ---- ingest kernels (data_loader.py:166-247, the step before set_displacement) ----------------
 * Device-side restatement of load_elevation_data: int16 LDEM units -> block mean (two-stage f32,
 * axis 4 then axis 2) -> * 0.5/1737400 -> + 1 -> / max.  src_dev is (h*d, w*d) int16 on the device,
 * dst_dev is (h, w) float32 on the device.  radius_scale receives the pre-normalisation maximum. */
int mrtx_dem_from_ldem(int32_t device, const void* src_dev_i16, int32_t h, int32_t w, int32_t downscale,
                       void* dst_dev_f32, float* radius_scale, char* err, int32_t err_len);
/* Seeded synthetic LDEM-like int16 source written straight into device memory (bench input). */
int mrtx_synth_ldem(int32_t device, void* dst_dev_i16, int32_t h, int32_t w, uint32_t seed,
                    char* err, int32_t err_len);
/* Seeded synthetic RGBA8 albedo texture (already through the 0.2+0.75v LUT range), device memory. */
int mrtx_synth_color(int32_t device, void* dst_dev_rgba8, int32_t h, int32_t w, uint32_t seed,
                     char* err, int32_t err_len);
/* Plain device allocation helpers so the host side needs no other GPU runtime for inputs. */
int mrtx_dev_alloc(int32_t device, uint64_t bytes, void** out);
int mrtx_dev_free(int32_t device, void* p);
int mrtx_dev_download(int32_t device, void* host_dst, const void* dev_src, uint64_t bytes);
int mrtx_dev_upload(int32_t device, void* dev_dst, const void* host_src, uint64_t bytes);

/* Profiling aid: streams a `bytes`-sized buffer `repeats` times with 8-byte-per-lane loads (the render kernel's
 * access width) so rocprofv3's FETCH_SIZE can be calibrated against a known byte count. */
int mrtx_probe_stream(int32_t device, uint64_t bytes, int32_t repeats);

/* Math conformance probe: evaluates the renderer's own (lat, lon) primitive -- polynomial atan2 pair sharing
 * one reciprocal -- on the device for n moon-frame points (tests compare it with the oracle bit for bit). */
int mrtx_probe_latlon(int32_t device, const float* a, const float* b, const float* c, float* lat, float* lon,
                      int32_t n);

/* Math conformance probe (ABI 7): the kernels' domain-restricted reciprocal (v_rcp_f32 + Newton steps) and square root (v_sqrt_f32 + a
 * +-1 ulp residual fix) against the compiler's IEEE expansions of 1.0f / x and sqrtf(x), ON THE DEVICE, for the n float bit patterns
 * from lo_bits on: which = 0 one Newton step, 1 two steps (what the kernels use), 2 the square root.  mismatches = how many differ;
 * first_bad_bits = the smallest bit pattern that does (0 if none).  The whole domain is 2^32 patterns: seconds. */
int mrtx_probe_cr(int32_t device, int32_t which, uint32_t lo_bits, uint64_t n, uint64_t* mismatches, uint32_t* first_bad_bits);

#ifdef __cplusplus
}
#endif
#endif /* MOONRT_H */
