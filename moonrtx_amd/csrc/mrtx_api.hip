// mrtx_api.hip -- host side of libmoonrt.so: the C ABI declared in include/moonrt.h.
//
// The context keeps the scene the way the reference describes it to its renderer object (float64
// camera / moon frame / light / Sun disk, moon_renderer.py:570-650 and :824-871) and derives the
// float32 per-launch constant block (FrameC) from it right before every launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/moonrt.h"
#include "mrtx_device.h"

hipError_t mrtx_launch_render(const FrameC& f, int S, bool stats, int mode, bool overlay, const PathQ* pq, hipStream_t st);
uint64_t mrtx_path_chunks(const FrameC& f, int S, uint32_t* grid_a, int* njobs_log2);
int mrtx_path_waves(bool stats, bool wide, int* out);
hipError_t mrtx_launch_paths(const FrameC& f, const PathQ& pq, int S, bool stats, int n_waves, hipStream_t st);
hipError_t mrtx_launch_resolve_linear(const float* accum, float* out, int64_t npix, uint32_t ns, hipStream_t st);
hipError_t mrtx_launch_resolve_rgba8(const float* accum, uint32_t* out, int64_t npix, uint32_t ns, float expo,
                                     const float* T8, const uint32_t* overlay, hipStream_t st);
hipError_t mrtx_launch_resolve_rgb16(const float* accum, uint16_t* out, int64_t npix, uint32_t ns, float expo,
                                     const float* T16, hipStream_t st);
hipError_t mrtx_launch_pack(const float* accum, const float* hits, void* dst, int W, int H, int tw, int th,
                            int tiles_x, int n_tiles, int rank, int world, int slot0, int slots, const int32_t* list, int shift,
                            int with_hits, hipStream_t st);
hipError_t mrtx_launch_unpack(float* accum, float* hits, const void* src, int W, int H, int tw, int th,
                              int tiles_x, int n_tiles, int src_rank, int world, int slots, const int32_t* list, int shift,
                              int with_hits, hipStream_t st);
hipError_t mrtx_launch_unpack_all(float* accum, float* hits, const void* const* srcs, int W, int H, int tw, int th, int tiles_x,
                                  int n_tiles, int self, int world, int slots, const int32_t* list_all, int shift, int with_hits,
                                  hipStream_t st);
hipError_t mrtx_launch_zero_tiles(float* accum, float* hits, const int32_t* tiles, int n, int W, int H, int tw, int th,
                                  int tiles_x, int shift, hipStream_t st);
hipError_t mrtx_launch_ldem(const int16_t* src, float* dst, int h, int w, int d, unsigned int* max_bits,
                            hipStream_t st);
hipError_t mrtx_launch_synth_ldem(int16_t* dst, int h, int w, uint32_t seed, hipStream_t st);
hipError_t mrtx_launch_synth_color(uint32_t* dst, int h, int w, uint32_t seed, hipStream_t st);
hipError_t mrtx_launch_probe_latlon(const float* a, const float* b, const float* c, float* lat, float* lon, int n,
                                    hipStream_t st);
hipError_t mrtx_launch_pad_dem(const float* src, float* dst, int h, int w, hipStream_t st);
hipError_t mrtx_launch_mip(const float* dem_padded, int h, int w, float* mip, int mh, int mw, int shift, hipStream_t st);
hipError_t mrtx_launch_color_pairs(const uint32_t* src, void* dst, int h, int w, hipStream_t st);
hipError_t mrtx_launch_mip_pairs(const float* mip, float* out_pairs, int rows, int pitch, hipStream_t st);
hipError_t mrtx_launch_hmip(const float* mip, int mh, int mw, int shift, float* out, int hh, int hw, int hshift, int h, int w,
                            hipStream_t st);
hipError_t mrtx_launch_probe_stream(const void* src, int64_t n_pairs, float* out, hipStream_t st);
hipError_t mrtx_launch_probe_cr(uint32_t lo, uint64_t n, int which, unsigned long long* out2, hipStream_t st);

struct mrtx_ctx {
    MrtxConfig cfg{};
    MrtxParams prm{};
    int tiles_x = 0, tiles_y = 0, n_tiles = 0, n_local = 0, slots = 0, tile_shift = 3;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evs;         // stage boundaries of a deferred-path call (3 per block)
    // hand-over records of the deferred path stage (PathQ): 6 float arrays of 64 x chunks + one word per chunk
    float* path_rec = nullptr; uint32_t* path_meta = nullptr; uint8_t* path_npaths = nullptr; uint64_t path_cap = 0;
    uint32_t* path_ctr = nullptr;        // 8 x 16 work counters of path_kernel
    unsigned long long* wd_host = nullptr;   // pinned: path_kernel's watchdog word
    uint64_t path_budget_bytes = 64ull << 30;   // hand-over buffers: frames that need more are rendered in sub-parts (MOONRT_PATH_MAX_GB).  Sized for 288 GB of HBM:
                                                // a 4K frame with every pixel on the Moon (34 GB) stays ONE part -- 38.9 ms against 39.2 in two parts of 24 GB
    bool path_budget_env = false;               // the budget was given explicitly: take it as it is (else: at most half of the free memory)
    int path_nsub = 4, path_grp_log2 = 3;   // measured at cfg3: (0,1) 37 ms, (1,1) 20.5, (1,4) 16.1, (2,4) 16.3, (3,4) 16.6
    int path_waves[4] = {0, 0, 0, 0};    // persistent waves of path_kernel<stats, wide>, 0 = not asked yet
    // launches with fewer samples than this keep their paths inside the render wave (same result, bit for bit): the three
    // kernels + 5 120 persistent waves of the queue do not pay below ~8 M samples (4K: 1 spp 1.34 ms against 1.48, 2 spp 2.28 /
    // 2.41, 4 spp 3.42 / 3.16; cfg1 0.33 / 0.53; tools/spp_sweep.py).  MOONRT_PATH_QUEUE_MIN overrides (0 = always the queue).
    uint64_t path_queue_min = 8000000ull;
    bool path_fallback_said = false;     // the fall-back to in-wave paths (no memory for the records) was reported
    // overlapped path stage (MOONRT_PATH_OVERLAP = sub-parts, 0 = off): path_kernel + resolve of sub-part i run on stream2 beside
    // render_kernel of sub-part i+1 (two sets of hand-over buffers, ping-pong); MOONRT_PATH_OVERLAP_WAVES sizes the persistent
    // launch while it shares the chip (every path wave holds 96 VGPRs, a render wave 64)
    int path_overlap = 0, path_overlap_waves = 2048;
    int path_sets = 1;                   // buffer sets the allocations hold
    hipStream_t stream2 = nullptr;
    hipEvent_t ov_done[2] = {nullptr, nullptr}, ov_join = nullptr;
    uint64_t path_nomem_chunks = 0;      // a hand-over allocation of this many chunks failed: not retried until less is needed or mrtx_reset_accum
    bool gather_hits = true;             // mrtx_set_gather_hits: the hit buffer travels with the radiance
    bool path_alloc_fail_test = false;   // MOONRT_TEST_PATH_NOMEM=1: test hook, the hand-over allocation "fails"
    int path_refill = 32, path_segmin = 16, path_hitmin = 16, path_waves_env = 0;   // cfg3 sweep: (8,24,16) 16.2 ms, (24,24,16) 14.7, (32,16,16) 14.5, (48,24,16) 21.5
    float* accum = nullptr;
    float* hits = nullptr;
    void* scratch = nullptr;  // W*H*16 bytes, resolve target for read-back
    // "Gamma" post-process tables (tone_table): 256 / 65536 thresholds for the gamma they were built for (0 = not built)
    float* tone8_dev = nullptr; float tone8_gamma = 0.0f;
    float* tone16_dev = nullptr; float tone16_gamma = 0.0f;
    float* dem = nullptr; int dem_h = 0, dem_w = 0;   // padded (h+4) x (w+4) copy, always owned
    float* hmip = nullptr; int hm_h = 0, hm_w = 0, hm_shift = 0;       // horizon mip: cells of 8 max-mip cells, dilated by one cell (see horizon_kend)
    float* mip2 = nullptr; int m2_h = 0, m2_w = 0, m2_shift = 0;       // medium max-mip: cells a quarter of the max-mip's (path_kernel's step mask)
    float* mip = nullptr; int mip_h = 0, mip_w = 0, mip_shift = 0;   // max-mip of it, cell 2^mip_shift texels (+ one-cell border)
    uint8_t* color = nullptr; bool color_owned = false; int color_h = 0, color_w = 0;
    uint8_t* bg = nullptr; int bg_h = 0, bg_w = 0;
    uint32_t* overlay = nullptr;   // D12: frame-sized RGBA8 blended over the tone-mapped image, or null
    unsigned long long* stats_dev = nullptr;
    FrameCold* cold_dev = nullptr;
    FrameCold cold_uploaded;             // what cold_dev holds (valid once cold_valid): re-uploaded only when it changes
    bool cold_valid = false;
    // D11 overlay capsules: host copy in scene coordinates; device copies relative to the Moon centre + tile bins
    std::vector<float> caps_host;            // 12 floats per capsule
    float* caps_dev = nullptr; int32_t* caps_off_dev = nullptr; int32_t* caps_idx_dev = nullptr;
    size_t caps_dev_n = 0, caps_idx_cap = 0;
    std::vector<int32_t> caps_off_host;      // per local tile, valid for caps_version
    uint64_t caps_version = 0;               // scene_version the bins were built for
    int32_t* tile_list_dev = nullptr;   // n_local entries
    std::vector<int32_t> keep_uploaded;  // what tile_list_dev holds
    std::vector<int32_t> keep_cached;    // cull result for scene_version == cull_version
    uint64_t culled_px_cached = 0;
    int keep_front_cached = 0;           // leading entries of keep_cached that can see the Moon / Sun / an overlay (the rest is sky)
    uint64_t scene_version = 1, cull_version = 0;   // bumped by every setter that can change the cull
    std::vector<uint8_t> tile_dirty;     // local tiles written since the buffers were last zeroed
    // gather layout: with the sky cull in force only ACTIVE tiles travel (every rank derives every rank's list from the
    // scene, which is identical on all ranks by contract); act_slots = the longest list, shorter ones are padded with -1
    std::vector<std::vector<int32_t>> act_lists;
    uint64_t act_version = 0;
    bool act_on = false;
    int act_slots = 0;
    int32_t* act_dev = nullptr;          // world x act_slots, uploaded for act_uploaded_version
    size_t act_dev_cap = 0;
    uint64_t act_uploaded_version = 0;
    std::vector<uint8_t> peer_written;   // root: global tiles of other ranks that hold unpacked data
    int32_t* stale_dev = nullptr;
    size_t stale_cap = 0;
    uint32_t blocks_done = 0;
    double eye[3] = {0, -300, 0}, target[3] = {0, 0, 0}, up[3] = {0, 0, 1}, vfov = 4.2421875;
    double center[3] = {0, 0, 0}, radius = 10.0, u[3] = {0, 0, 1}, v[3] = {0, -1, 0};
    double light_pos[3] = {0, -21460, 0}, light_radius = 100.0, light_radiance = 0.0;
    double sun_pos[3] = {0, 3100, 0}, sun_radius = 0.0, sun_radiance = 2.0;
    std::string err;
};

// Host -> device copies that feed kernels of this context go through the context's OWN stream (hipStreamNonBlocking: the null
// stream does not order it) and are waited for before the host buffer is released, so copy and consumer are ordered by the
// stream itself.  (Introduced in round 3 on suspicion while a one-off fuzz mismatch was open; round 4 showed that mismatch to
// be a changed INPUT -- one word of the caller's colour-map array decremented by one before the upload,
// profiles/r04_seed601_case49.md -- so this is plain hygiene, not a fix of anything observed.)
static hipError_t h2d(mrtx_ctx* c, void* dst, const void* src, size_t bytes) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e;
}

namespace {

// Streams and events are POOLED per device and never destroyed while the process lives (MOONRT_POOL_STREAMS=0 switches back to
// create / destroy per context).  A context is cheap and short-lived in the tests (hundreds per process); the pool saves two
// runtime calls per object and, more to the point, keeps the HIP runtime's completion threads from ever working on a stream or
// event whose owner has gone (round 4: the harness's input guard caught a word of a host array being decremented by one while
// only the runtime's own threads could have been writing, profiles/r04_seed601_case49.md).
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> streams[16];
    std::vector<hipEvent_t> events[16];
};
StreamPool& pool() { static StreamPool* p = new StreamPool(); return *p; }     // leaked on purpose: outlives every context
// MOONRT_POOL_STREAMS: unset / 1 = pool both (default), 0 = neither, 2 = streams only, 3 = events only (the last two: diagnosis)
int pool_mode() { static const int m = std::getenv("MOONRT_POOL_STREAMS") ? std::atoi(std::getenv("MOONRT_POOL_STREAMS")) : 1; return m; }
bool pool_streams() { return pool_mode() == 1 || pool_mode() == 2; }
bool pool_events() { return pool_mode() == 1 || pool_mode() == 3; }

hipError_t get_stream(int dev, hipStream_t* out) {
    if (pool_streams() && dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> g(pool().mu);
        auto& v = pool().streams[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void put_stream(int dev, hipStream_t st) {
    if (!st) return;
    if (pool_streams() && dev >= 0 && dev < 16) { std::lock_guard<std::mutex> g(pool().mu); pool().streams[dev].push_back(st); }
    else (void)hipStreamDestroy(st);
}
hipError_t get_event(int dev, hipEvent_t* out) {
    if (pool_events() && dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> g(pool().mu);
        auto& v = pool().events[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipEventCreate(out);
}
void put_event(int dev, hipEvent_t ev) {
    if (!ev) return;
    if (pool_events() && dev >= 0 && dev < 16) { std::lock_guard<std::mutex> g(pool().mu); pool().events[dev].push_back(ev); }
    else (void)hipEventDestroy(ev);
}

int fail(mrtx_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), MRTX_E_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

const double kPiD = 3.14159265358979323846;

void unit3(double v[3]) {
    const double l = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
}
void cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
uint32_t mix32h(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
void set_grid(GridC& g, int h, int w) {
    g.h = h; g.w = w;
    g.row_scale = (float)(-(double)h / kPiD);
    g.row_off = (float)(0.5 * (double)h - 0.5);
    g.col_scale = (float)((double)w / (2.0 * kPiD));
    g.col_off = (float)(0.5 * (double)w - 0.5);
    g.wf = (float)w;
}

// Scene (float64) -> per-launch constants.  DESIGN.md section 3.1 lists every formula; the oracle
// derives the same block on its own and tests compare the two float for float.
void build_frame(const mrtx_ctx* c, FrameC& f, FrameCold& k) {
    std::memset(&f, 0, sizeof f);
    std::memset(&k, 0, sizeof k);
    f.W = c->cfg.width; f.H = c->cfg.height;
    double wv[3], uv[3], vv[3];
    for (int i = 0; i < 3; i++) wv[i] = c->target[i] - c->eye[i];
    unit3(wv);
    cross(wv, c->up, uv); unit3(uv);
    cross(uv, wv, vv);
    const double th = std::tan(c->vfov * kPiD / 360.0);
    const double aspect = (double)f.W / (double)f.H;
    for (int i = 0; i < 3; i++) {
        k.Wd[i] = (float)wv[i];
        k.Ux[i] = (float)(uv[i] * (th * aspect));
        k.Vy[i] = (float)(vv[i] * th);
        k.oc[i] = c->eye[i] - c->center[i];
        k.centerf[i] = (float)c->center[i];
        k.eyef[i] = (float)c->eye[i];
    }
    k.two_over_w = (float)(2.0 / (double)f.W);
    k.two_over_h = (float)(2.0 / (double)f.H);
    k.cq = ((k.oc[0] * k.oc[0] + k.oc[1] * k.oc[1]) + k.oc[2] * k.oc[2]) - c->radius * c->radius;
    // moon frame rows: east-90, lon-0, north.  u = north pole, v = longitude-0 direction
    // (moon_renderer.py:621, :844-845; renderer_navigation.py:47-53)
    double ez[3] = {c->u[0], c->u[1], c->u[2]}, v0[3], ex[3];
    unit3(ez);
    const double dp = (c->v[0] * ez[0] + c->v[1] * ez[1]) + c->v[2] * ez[2];
    for (int i = 0; i < 3; i++) v0[i] = c->v[i] - dp * ez[i];
    unit3(v0);
    cross(ez, v0, ex);
    for (int j = 0; j < 3; j++) { k.M[0][j] = ex[j]; k.M[1][j] = v0[j]; k.M[2][j] = ez[j]; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) k.Mf[i][j] = (float)k.M[i][j];
    f.Rf = (float)c->radius;
    f.R2f = f.Rf * f.Rf;
    double lr[3];
    for (int i = 0; i < 3; i++) lr[i] = c->light_pos[i] - c->center[i];
    for (int i = 0; i < 3; i++) k.Lb[i] = (float)((k.M[i][0] * lr[0] + k.M[i][1] * lr[1]) + k.M[i][2] * lr[2]);
    k.rL2 = (float)(c->light_radius * c->light_radius);
    k.rad2 = (float)(2.0 * c->light_radiance);
    k.sun_on = c->sun_radius > 0.0 ? 1 : 0;
    double sr[3];
    for (int i = 0; i < 3; i++) { sr[i] = c->sun_pos[i] - c->eye[i]; k.sc[i] = (float)sr[i]; }
    k.sun_cq = (float)(((sr[0] * sr[0] + sr[1] * sr[1]) + sr[2] * sr[2]) - c->sun_radius * c->sun_radius);
    k.sun_rad = (float)c->sun_radiance;
    {
        double sm[3];
        for (int i = 0; i < 3; i++) sm[i] = c->sun_pos[i] - c->center[i];
        for (int i = 0; i < 3; i++) k.Sb[i] = (float)((k.M[i][0] * sm[0] + k.M[i][1] * sm[1]) + k.M[i][2] * sm[2]);
        k.sun_r2 = (float)(c->sun_radius * c->sun_radius);
    }
    f.step = c->prm.marching_step;
    k.eps = c->prm.marching_step_eps;
    k.scene_eps = c->prm.scene_epsilon;
    f.nbis = 0;
    for (double wdt = (double)f.step; wdt > (double)k.eps && f.nbis < 24; wdt *= 0.5) f.nbis++;
    f.kmax = (((int)(2.0 * c->radius / (double)f.step) + 8) + 15) & ~15;   // multiple of the 16-step segment
    f.inv_step = 1.0f / f.step;
    f.polar_rho2 = (float)(0.04 * c->radius * c->radius);
    f.row_hi = std::nextafterf((float)c->dem_h, 0.0f);
    f.col_hi = std::nextafterf((float)c->dem_w, 0.0f);
    set_grid(f.gd, c->dem_h, c->dem_w);
    k.dlat_scale = (float)((double)c->dem_h / (2.0 * kPiD));
    k.dlon_scale = (float)((double)c->dem_w / (4.0 * kPiD));
    if (c->color) set_grid(k.gc, c->color_h, c->color_w);
    if (c->bg) {
        k.bg_h = c->bg_h; k.bg_w = c->bg_w;
        k.bg_row_scale = (float)(-(double)c->bg_h / kPiD);
        k.bg_row_off = (float)(0.5 * (double)c->bg_h);
        k.bg_col_scale = (float)((double)c->bg_w / (2.0 * kPiD));
        k.bg_col_off = (float)(0.5 * (double)c->bg_w);
    }
    k.key0 = mix32h(c->prm.seed ^ 0x9E3779B9u);
    k.path_seg_min = c->prm.path_seg_min; k.path_seg_max = c->prm.path_seg_max < 1 ? 1 : c->prm.path_seg_max;
    for (int i = 0; i < 3; i++) k.const_albedo[i] = c->prm.const_albedo[i];
    f.dem = c->dem; k.color = c->color; k.bg = c->bg;
    k.hmip = (c->prm.flags & MRTX_F_NO_SKIP) ? nullptr : c->hmip; k.hm_h = c->hm_h; k.hm_w = c->hm_w; k.hm_shift = c->hm_shift;
    k.hm_cell = (float)(1 << c->hm_shift);
    k.mip2 = (c->prm.flags & MRTX_F_NO_SKIP) ? nullptr : c->mip2; k.m2_pitch = c->m2_w + 2; k.m2_h = c->m2_h; k.m2_w = c->m2_w; k.m2_shift = c->m2_shift;
    k.hm_krow = (float)((double)c->dem_h / kPiD);
    k.hm_kcol = (float)(1.05 * (double)c->dem_w / (2.0 * kPiD));
    f.mip = c->mip; f.mip_pitch = c->mip_w + 2; f.mip_h = c->mip_h; f.mip_w = c->mip_w; f.mip_shift = c->mip_shift;
    f.dem_pitch = c->dem_w + 4;
#if MRTX_DEM_PAIRS
    f.dem_maxidx = (uint32_t)((uint64_t)(c->dem_h + 4) * (uint64_t)(c->dem_w + 4) - 2);   // elements idx and idx+1 are read
#else
    f.dem_maxidx = (uint32_t)((uint64_t)(c->dem_h + 2) * (uint64_t)(c->dem_w + 4) + (uint64_t)(c->dem_w + 2));
#endif
    f.dem_wide = ((uint64_t)(c->dem_h + 4) * (uint64_t)(c->dem_w + 4) * MRTX_DEM_ELEM_BYTES > 0xFFFFFFFFull) ? 1 : 0;
    f.tile_w = c->cfg.tile_w; f.tile_h = c->cfg.tile_h;
    f.tiles_x = c->tiles_x; f.tiles_y = c->tiles_y; f.tile_shift = c->tile_shift;
    f.rank = c->cfg.rank; f.world = c->cfg.world; f.n_local_tiles = c->n_local;
    k.accum = c->accum; k.hits = c->hits; k.stats = c->stats_dev; k.stats_paths = c->stats_dev + 16;
    f.tile_list = nullptr; f.n_active = c->n_local;
}

// D11: (re)build the device copy of the overlay capsules (relative to the Moon centre) and their per-tile bins.
// A capsule is binned into every local tile its bounding sphere can project into (conservative: perspective
// stretch factor 1 + tan^2, +2 px), so the kernel's nearest-hit search over a bin equals the search over all.
int build_capsule_bins(mrtx_ctx* c) {
    const size_t n = c->caps_host.size() / 12;
    c->caps_off_host.assign((size_t)c->n_local + 1, 0);
    if (n == 0) return MRTX_OK;
    const int W = c->cfg.width, H = c->cfg.height;
    double wv[3], uv[3], vv[3];
    for (int i = 0; i < 3; i++) wv[i] = c->target[i] - c->eye[i];
    unit3(wv);
    cross(wv, c->up, uv); unit3(uv);
    cross(uv, wv, vv);
    const double th = std::tan(c->vfov * kPiD / 360.0), aspect = (double)W / (double)H;
    std::vector<float> rel(n * 12);
    std::vector<std::vector<int32_t>> bins((size_t)c->n_local);
    for (size_t k = 0; k < n; k++) {
        const float* p = &c->caps_host[12 * k];
        float* q = &rel[12 * k];
        for (int i = 0; i < 3; i++) {
            q[i] = (float)((double)p[i] - c->center[i]);
            q[4 + i] = (float)((double)p[4 + i] - c->center[i]);
        }
        q[3] = p[3]; q[7] = 0.0f; q[8] = p[8]; q[9] = p[9]; q[10] = p[10]; q[11] = 0.0f;
        double m[3], half = 0.0;
        for (int i = 0; i < 3; i++) { m[i] = 0.5 * ((double)p[i] + (double)p[4 + i]) - c->eye[i]; const double d = (double)p[4 + i] - (double)p[i]; half += d * d; }
        const double rho = 0.5 * std::sqrt(half) + (double)p[3];
        const double z = (m[0] * wv[0] + m[1] * wv[1]) + m[2] * wv[2];
        if (z + rho <= 1.0e-9) continue;                       // entirely behind the eye
        int tx0 = 0, tx1 = c->tiles_x - 1, ty0 = 0, ty1 = c->tiles_y - 1;
        if (z - rho > 1.0e-6) {
            const double x = (m[0] * uv[0] + m[1] * uv[1]) + m[2] * uv[2], y = (m[0] * vv[0] + m[1] * vv[1]) + m[2] * vv[2];
            const double tx = x / z, ty = y / z;               // tangents of the view angles
            const double px = (tx / (th * aspect) + 1.0) * 0.5 * W, py = (1.0 - ty / th) * 0.5 * H;
            const double pr = rho / (z - rho) / th * (0.5 * H) * (1.0 + tx * tx + ty * ty) * 1.05 + 2.0;
            const double fx0 = std::floor((px - pr) / c->cfg.tile_w), fx1 = std::floor((px + pr) / c->cfg.tile_w);
            const double fy0 = std::floor((py - pr) / c->cfg.tile_h), fy1 = std::floor((py + pr) / c->cfg.tile_h);
            if (fx1 < 0 || fy1 < 0 || fx0 > c->tiles_x - 1 || fy0 > c->tiles_y - 1) continue;
            tx0 = (int)std::fmax(fx0, 0.0); tx1 = (int)std::fmin(fx1, (double)c->tiles_x - 1);
            ty0 = (int)std::fmax(fy0, 0.0); ty1 = (int)std::fmin(fy1, (double)c->tiles_y - 1);
        }
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int t = mrtx_tile_id(tx, ty, c->tiles_x, c->tile_shift);
                if (t % c->cfg.world == c->cfg.rank) bins[(size_t)(t / c->cfg.world)].push_back((int32_t)k);
            }
    }
    std::vector<int32_t> idx;
    for (int lt = 0; lt < c->n_local; lt++) {
        c->caps_off_host[(size_t)lt] = (int32_t)idx.size();
        idx.insert(idx.end(), bins[(size_t)lt].begin(), bins[(size_t)lt].end());
    }
    c->caps_off_host[(size_t)c->n_local] = (int32_t)idx.size();
    if (c->caps_dev_n < n) {
        if (c->caps_dev) { HIPCHK(c, hipFree(c->caps_dev)); c->caps_dev = nullptr; }
        HIPCHK(c, hipMalloc((void**)&c->caps_dev, n * 12 * sizeof(float)));
        c->caps_dev_n = n;
    }
    if (!c->caps_off_dev) HIPCHK(c, hipMalloc((void**)&c->caps_off_dev, ((size_t)c->n_local + 1) * sizeof(int32_t)));
    if (c->caps_idx_cap < idx.size() + 1) {
        if (c->caps_idx_dev) { HIPCHK(c, hipFree(c->caps_idx_dev)); c->caps_idx_dev = nullptr; }
        c->caps_idx_cap = idx.size() * 2 + 64;
        HIPCHK(c, hipMalloc((void**)&c->caps_idx_dev, c->caps_idx_cap * sizeof(int32_t)));
    }
    HIPCHK(c, h2d(c, c->caps_dev, rel.data(), n * 12 * sizeof(float)));
    HIPCHK(c, h2d(c, c->caps_off_dev, c->caps_off_host.data(), c->caps_off_host.size() * sizeof(int32_t)));
    if (!idx.empty()) HIPCHK(c, h2d(c, c->caps_idx_dev, idx.data(), idx.size() * sizeof(int32_t)));
    return MRTX_OK;
}

// Host-side sky cull (exact): with no environment texture a pixel whose samples all miss the Moon's bounding
// sphere and the Sun disk is identically zero (radiance, coverage and hit record).  A tile is kept unless the
// cone of its view directions (tile-centre direction, half-angle = largest corner angle + 5 % + 1e-4 rad, pixel
// extents included so every jittered sample is inside) is disjoint from both objects' cones as seen from the eye.
// Returns the local tile indices to render and the number of pixels culled.
// With an environment texture bound nothing is culled, but the classification still pays: the tiles that can only see
// the sky go to the END of the list (n_front = how many can see more), so that the deferred-path pipeline -- buffers sized
// per wave-job, persistent waves walking every group of blocks -- is set up for the Moon's tiles only and the sky gets one
// plain launch (cfg3 with the reference's star map: 8 100 tiles, 3 056 of them with the Moon in view).
void cull_tiles(const mrtx_ctx* c, int rank, std::vector<int32_t>& keep, uint64_t& culled_px, int* n_front = nullptr) {
    keep.clear(); culled_px = 0;
    std::vector<int32_t> sky;
    const int n_local = (c->n_tiles - rank + c->cfg.world - 1) / c->cfg.world;
    const bool own = rank == c->cfg.rank;   // overlay bins exist for the own tiles only (gather_layout() switches off with overlays)
    const int W = c->cfg.width, H = c->cfg.height;
    double wv[3], uv[3], vv[3];
    for (int i = 0; i < 3; i++) wv[i] = c->target[i] - c->eye[i];
    unit3(wv);
    cross(wv, c->up, uv); unit3(uv);
    cross(uv, wv, vv);
    const double th = std::tan(c->vfov * kPiD / 360.0), aspect = (double)W / (double)H;
    struct Cone { double ax[3]; double half; bool all; bool on; };
    Cone cones[2];
    const double* centres[2] = {c->center, c->sun_pos};
    const double radii[2] = {c->radius, c->sun_radius};
    for (int k = 0; k < 2; k++) {
        Cone& cn = cones[k];
        cn.on = radii[k] > 0.0; cn.all = false; cn.half = 0.0;
        double d[3] = {centres[k][0] - c->eye[0], centres[k][1] - c->eye[1], centres[k][2] - c->eye[2]};
        const double dist = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        if (!cn.on) continue;
        if (dist <= radii[k] * 1.0001) { cn.all = true; continue; }
        for (int i = 0; i < 3; i++) cn.ax[i] = d[i] / dist;
        cn.half = std::asin(radii[k] / dist);
    }
    const bool everything = cones[0].all || cones[1].all;
    const bool sky_rendered = c->bg != nullptr;
    auto dir = [&](double px, double py, double o[3]) {
        const double sx = px * 2.0 / W - 1.0, sy = 1.0 - py * 2.0 / H;
        for (int i = 0; i < 3; i++) o[i] = wv[i] + sx * (uv[i] * (th * aspect)) + sy * (vv[i] * th);
        unit3(o);
    };
    auto ang = [](const double a[3], const double b[3]) {
        double dp = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
        dp = dp > 1.0 ? 1.0 : (dp < -1.0 ? -1.0 : dp);
        return std::acos(dp);
    };
    // Longest first: a pixel on the limb marches ~20x longer than one at the disc centre, and a launch ends with its
    // slowest late wave, so tiles on the limb ring (0.85 .. 1 of the disc's angular radius) go to the FRONT of the
    // list; raster order is kept inside both classes (neighbouring tiles share DEM lines in L2 / Infinity Cache).
    std::vector<int32_t> ring;
    const bool sort_ring = !(c->prm.flags & MRTX_F_NO_SORT) && cones[0].on && !cones[0].all;
    for (int lt = 0; lt < n_local; lt++) {
        const int t = lt * c->cfg.world + rank;
        int ttx, tty;
        mrtx_tile_xy(t, c->tiles_x, c->tile_shift, ttx, tty);
        const int x0 = ttx * c->cfg.tile_w, y0 = tty * c->cfg.tile_h;
        const int x1 = x0 + c->cfg.tile_w < W ? x0 + c->cfg.tile_w : W, y1 = y0 + c->cfg.tile_h < H ? y0 + c->cfg.tile_h : H;
        bool need = everything;
        if (!need) {
            double d0[3], dc[3];
            dir(0.5 * (x0 + x1), 0.5 * (y0 + y1), d0);
            double gamma = 0.0;
            const double cx[4] = {(double)x0, (double)x1, (double)x0, (double)x1}, cy[4] = {(double)y0, (double)y0, (double)y1, (double)y1};
            for (int k = 0; k < 4; k++) { dir(cx[k], cy[k], dc); const double a = ang(d0, dc); gamma = a > gamma ? a : gamma; }
            gamma = gamma * 1.05 + 1.0e-4;
            for (int k = 0; k < 2; k++)
                if (cones[k].on && ang(d0, cones[k].ax) <= cones[k].half + gamma) need = true;
        }
        if (!need && own && !c->caps_off_host.empty() && c->caps_off_host[(size_t)lt + 1] > c->caps_off_host[(size_t)lt]) need = true;   // overlay tubes here
        if (need) {
            bool on_ring = false;
            if (sort_ring) {
                double d0[3];
                dir(0.5 * (x0 + x1), 0.5 * (y0 + y1), d0);
                const double rho = ang(d0, cones[0].ax) / cones[0].half;
                on_ring = rho >= 0.85 && rho <= 1.05;
            }
            (on_ring ? ring : keep).push_back(lt);
        } else if (sky_rendered) {
            sky.push_back(lt);
        } else {
            culled_px += (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
        }
    }
    keep.insert(keep.begin(), ring.begin(), ring.end());
    if (n_front) *n_front = (int)keep.size();
    keep.insert(keep.end(), sky.begin(), sky.end());
}

int check_vec(const double* p) {
    if (!p) return 0;
    for (int i = 0; i < 3; i++)
        if (!std::isfinite(p[i])) return 0;
    return 1;
}

}  // namespace

extern "C" {

int mrtx_abi_version(void) { return MRTX_ABI_VERSION; }

int mrtx_get_config(mrtx_ctx* c, MrtxConfig* out) {
    if (!c || !out) return MRTX_E_INVALID;
    *out = c->cfg;      // mrtx_create filled the defaults in
    return MRTX_OK;
}

void mrtx_default_params(MrtxParams* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->scene_epsilon = 1.0e-4f;      // moon_renderer.py:99
    p->marching_step = 5.0e-3f;      // :100
    p->marching_step_eps = 3.0e-4f;  // :101
    p->tonemap_exposure = 0.9f;      // :598
    p->tonemap_gamma = 2.2f;
    p->path_seg_min = 2; p->path_seg_max = 4;  // :583
    p->spp_per_launch = 64; p->max_spp = 64;   // :130
    p->seed = 1;
    p->const_albedo[0] = p->const_albedo[1] = p->const_albedo[2] = 75.0f / 255.0f;  // lut[128], gamma 2.2
    p->flags = 0;   // production kernels; MRTX_F_COUNT_STATS selects the counting instantiations (2-3x slower)
}

int mrtx_create(const MrtxConfig* cfg, mrtx_ctx** out) {
    if (!cfg || !out) return MRTX_E_INVALID;
    *out = nullptr;
    if (cfg->width < 1 || cfg->height < 1 || cfg->width > 32768 || cfg->height > 32768) return MRTX_E_INVALID;
    if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world) return MRTX_E_INVALID;
    mrtx_ctx* c = new (std::nothrow) mrtx_ctx();
    if (!c) return MRTX_E_NOMEM;
    c->cfg = *cfg;
    // default tile: 16 x 16 on one GPU -- the sky cull and the limb-first launch order work at tile granularity, and the finer
    // tile keeps ~2 % of cfg3's waves (all-sky pixel blocks along the limb) off the GPU: 23.7 ms against 24.2 with 32 x 32 (direct
    // frame 12.7 / 13.2; gpurun_out/r3m/tile1.log), and for two ranks (slowest rank 12.28 / 12.49 ms); 32 x 32 from four ranks up
    // (world 4: 6.41 against 6.60 ms, world 8: 3.51 against 3.76: a rank's lattice of small tiles loses L2 locality)
    const int dflt = c->cfg.world > 2 ? 32 : 16;
    if (c->cfg.tile_w <= 0) c->cfg.tile_w = dflt;
    if (c->cfg.tile_h <= 0) c->cfg.tile_h = dflt;
    if ((c->cfg.tile_w & 15) || (c->cfg.tile_h & 15)) { delete c; return MRTX_E_INVALID; }
    mrtx_default_params(&c->prm);
    {   // tuning knobs of the deferred path stage (measurement aid; the defaults are what bench.py times)
        const char* e;
        if ((e = std::getenv("MOONRT_PATH_REFILL")) && std::atoi(e) >= -64 && std::atoi(e) <= 64) c->path_refill = std::atoi(e);
        if ((e = std::getenv("MOONRT_PATH_SEGMIN")) && std::atoi(e) >= -64 && std::atoi(e) <= 64) c->path_segmin = std::atoi(e);
        if ((e = std::getenv("MOONRT_PATH_HITMIN")) && std::atoi(e) >= -64 && std::atoi(e) <= 64) c->path_hitmin = std::atoi(e);
        if ((e = std::getenv("MOONRT_PATH_NSUB")) && std::atoi(e) >= 1 && std::atoi(e) <= 16) c->path_nsub = std::atoi(e);
        if ((e = std::getenv("MOONRT_PATH_GRP")) && std::atoi(e) >= 0 && std::atoi(e) <= 5) c->path_grp_log2 = std::atoi(e);   // + njobs_log2 (<= 1) <= 6: 64 chunks per group
        if ((e = std::getenv("MOONRT_PATH_MAX_GB")) && std::atof(e) > 0.0) { c->path_budget_bytes = (uint64_t)(std::atof(e) * 1073741824.0); c->path_budget_env = true; }
        if ((e = std::getenv("MOONRT_PATH_WAVES")) && std::atoi(e) >= 8 * c->path_nsub) c->path_waves_env = std::atoi(e) / 8 * 8;   // every work counter needs a consumer (read after MOONRT_PATH_NSUB)
        if ((e = std::getenv("MOONRT_PATH_QUEUE_MIN")) && std::atof(e) >= 0.0) c->path_queue_min = (uint64_t)std::atof(e);
        if ((e = std::getenv("MOONRT_TEST_PATH_NOMEM")) && std::atoi(e) == 1) c->path_alloc_fail_test = true;
        if ((e = std::getenv("MOONRT_PATH_OVERLAP")) && std::atoi(e) >= 0 && std::atoi(e) <= 64) c->path_overlap = std::atoi(e) == 1 ? 0 : std::atoi(e);
        if ((e = std::getenv("MOONRT_PATH_OVERLAP_WAVES")) && std::atoi(e) >= 8 * c->path_nsub) c->path_overlap_waves = std::atoi(e) / 8 * 8;
    }
    c->tiles_x = (cfg->width + c->cfg.tile_w - 1) / c->cfg.tile_w;
    c->tiles_y = (cfg->height + c->cfg.tile_h - 1) / c->cfg.tile_h;
    c->n_tiles = c->tiles_x * c->tiles_y;
    c->tile_shift = mrtx_tile_shift(cfg->world);
    c->slots = (c->n_tiles + cfg->world - 1) / cfg->world;
    c->n_local = (c->n_tiles - cfg->rank + cfg->world - 1) / cfg->world;
    *out = c;  // from here on the caller can read mrtx_last_error and must destroy
    const size_t fb = (size_t)cfg->width * cfg->height * 16;
    HIPCHK(c, hipSetDevice(cfg->device));
    HIPCHK(c, get_stream(cfg->device, &c->stream));
    HIPCHK(c, get_event(cfg->device, &c->ev0));
    HIPCHK(c, get_event(cfg->device, &c->ev1));
    HIPCHK(c, hipMalloc((void**)&c->accum, fb));
    HIPCHK(c, hipMalloc((void**)&c->hits, fb));
    HIPCHK(c, hipMalloc(&c->scratch, fb));
    HIPCHK(c, hipMalloc((void**)&c->stats_dev, 32 * sizeof(unsigned long long)));   // [0..15] render_kernel (+ watchdog), [16..31] path_kernel
    HIPCHK(c, hipMalloc((void**)&c->cold_dev, sizeof(FrameCold)));
    HIPCHK(c, hipMalloc((void**)&c->tile_list_dev, (size_t)(c->n_local > 0 ? c->n_local : 1) * sizeof(int32_t)));
    HIPCHK(c, hipMemsetAsync(c->accum, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->hits, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->stats_dev, 0, 32 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}

void mrtx_destroy(mrtx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->accum) (void)hipFree(c->accum);
    if (c->hits) (void)hipFree(c->hits);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->tone8_dev) (void)hipFree(c->tone8_dev);
    if (c->tone16_dev) (void)hipFree(c->tone16_dev);
    if (c->stats_dev) (void)hipFree(c->stats_dev);
    if (c->cold_dev) (void)hipFree(c->cold_dev);
    if (c->caps_dev) (void)hipFree(c->caps_dev);
    if (c->caps_off_dev) (void)hipFree(c->caps_off_dev);
    if (c->caps_idx_dev) (void)hipFree(c->caps_idx_dev);
    if (c->tile_list_dev) (void)hipFree(c->tile_list_dev);
    if (c->act_dev) (void)hipFree(c->act_dev);
    if (c->stale_dev) (void)hipFree(c->stale_dev);
    if (c->path_rec) (void)hipFree(c->path_rec);
    if (c->path_meta) (void)hipFree(c->path_meta);
    if (c->path_npaths) (void)hipFree(c->path_npaths);
    if (c->path_ctr) (void)hipFree(c->path_ctr);
    if (c->wd_host) (void)hipHostFree(c->wd_host);
    for (hipEvent_t e : c->evs) put_event(c->cfg.device, e);
    if (c->dem) (void)hipFree(c->dem);
    if (c->mip) (void)hipFree(c->mip);
    if (c->hmip) (void)hipFree(c->hmip);
    if (c->mip2) (void)hipFree(c->mip2);
    if (c->color) (void)hipFree(c->color);
    if (c->bg) (void)hipFree(c->bg);
    if (c->overlay) (void)hipFree(c->overlay);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); put_stream(c->cfg.device, c->stream2); }
    put_event(c->cfg.device, c->ov_done[0]); put_event(c->cfg.device, c->ov_done[1]); put_event(c->cfg.device, c->ov_join);
    put_event(c->cfg.device, c->ev0);
    put_event(c->cfg.device, c->ev1);
    put_stream(c->cfg.device, c->stream);      // synchronised above
    delete c;
}

const char* mrtx_last_error(mrtx_ctx* c) { return c ? c->err.c_str() : "null context"; }

// The context keeps its own PADDED copy of the DEM (see dem_march() in mrtx_kernels.hip).
static int ingest_dem(mrtx_ctx* c, const float* dev_src, int32_t h, int32_t w) {
    if (c->dem) { HIPCHK(c, hipFree(c->dem)); }
    c->dem = nullptr; c->dem_h = c->dem_w = 0;
    const size_t bytes = (size_t)(h + 4) * (size_t)(w + 4) * MRTX_DEM_ELEM_BYTES;
    HIPCHK(c, hipMalloc((void**)&c->dem, bytes));
    HIPCHK(c, mrtx_launch_pad_dem(dev_src, c->dem, h, w, c->stream));
    if (c->mip) { HIPCHK(c, hipFree(c->mip)); }
    c->mip = nullptr; c->mip_shift = 0;   // (re)built by mrtx_render for the march step in force
    if (c->hmip) { HIPCHK(c, hipFree(c->hmip)); }
    c->hmip = nullptr;
    if (c->mip2) { HIPCHK(c, hipFree(c->mip2)); }
    c->mip2 = nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->dem_h = h; c->dem_w = w;
    return MRTX_OK;
}
// The max-mip cell must cover a whole 16-step segment (plus tap margins) so that a segment's footprint touches at
// most 2 x 2 cells: cell = 2^shift >= 16*step/texel + 3.5, clamped to [16, 1024] texels.
static int ensure_mip(mrtx_ctx* c) {
    const double texel = std::fmin(kPiD * c->radius / (double)c->dem_h, 2.0 * kPiD * c->radius / (double)c->dem_w);
    const double span = 16.0 * (double)c->prm.marching_step / texel + 3.5;   // + the (-1, +2) tap margins of seg_setup
    int shift = 4;
    while ((1 << shift) < span && shift < 10) shift++;
    if (c->mip && c->mip_shift == shift) return MRTX_OK;
    if (c->mip) { HIPCHK(c, hipFree(c->mip)); }
    c->mip = nullptr;
    if (c->hmip) { HIPCHK(c, hipFree(c->hmip)); }
    c->hmip = nullptr;
    if (c->mip2) { HIPCHK(c, hipFree(c->mip2)); }
    c->mip2 = nullptr;
    const int cell = 1 << shift;
    c->mip_h = (c->dem_h + cell - 1) / cell; c->mip_w = (c->dem_w + cell - 1) / cell;
    const size_t cells = (size_t)(c->mip_h + 2) * (c->mip_w + 2);
    float* plain = nullptr;
    HIPCHK(c, hipMalloc((void**)&plain, cells * sizeof(float)));
    hipError_t e = hipMalloc((void**)&c->mip, (cells + 1) * 2 * sizeof(float));   // row pairs + one element of slack for the 16-byte load
    if (e == hipSuccess) e = mrtx_launch_mip(c->dem, c->dem_h, c->dem_w, plain, c->mip_h, c->mip_w, shift, c->stream);
    if (e == hipSuccess) e = mrtx_launch_mip_pairs(plain, c->mip, c->mip_h + 2, c->mip_w + 2, c->stream);
    c->hm_shift = shift + 3;
    c->hm_h = (c->dem_h + (cell << 3) - 1) / (cell << 3); c->hm_w = (c->dem_w + (cell << 3) - 1) / (cell << 3);
    if (e == hipSuccess) e = hipMalloc((void**)&c->hmip, (size_t)c->hm_h * c->hm_w * sizeof(float));
    if (e == hipSuccess) e = mrtx_launch_hmip(plain, c->mip_h, c->mip_w, shift, c->hmip, c->hm_h, c->hm_w, c->hm_shift, c->dem_h, c->dem_w, c->stream);
    // medium max-mip: the same cell maxima (dilated by the two-texel tap border) at a quarter of the cell size -- 16 texels at cfg3,
    // 17 MB: what path_kernel tests the steps of a segment against before it fetches the DEM for them
    c->m2_shift = shift - MRTX_M2_DELTA < 2 ? 2 : shift - MRTX_M2_DELTA;
    if ((MRTX_PATH_MIP2 | MRTX_SEG_MASK) != 0) {
        const int c2 = 1 << c->m2_shift;
        c->m2_h = (c->dem_h + c2 - 1) / c2; c->m2_w = (c->dem_w + c2 - 1) / c2;
        if (e == hipSuccess) e = hipMalloc((void**)&c->mip2, (size_t)(c->m2_h + 2) * (size_t)(c->m2_w + 2) * sizeof(float));
        if (e == hipSuccess) e = mrtx_launch_mip(c->dem, c->dem_h, c->dem_w, c->mip2, c->m2_h, c->m2_w, c->m2_shift, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(plain);
    HIPCHK(c, e);
    c->mip_shift = shift;
    return MRTX_OK;
}

int mrtx_upload_dem(mrtx_ctx* c, const float* host, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    if (!host || h < 2 || w < 2) return fail(c, MRTX_E_INVALID, "DEM must be a float32 (h>=2, w>=2) array");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    float* stage = nullptr;
    const size_t bytes = (size_t)h * w * sizeof(float);
    HIPCHK(c, hipMalloc((void**)&stage, bytes));
    hipError_t e = h2d(c, stage, host, bytes);
    int rc = e == hipSuccess ? ingest_dem(c, stage, h, w) : fail(c, MRTX_E_DEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
    (void)hipFree(stage);
    return rc;
}
int mrtx_bind_dem_device(mrtx_ctx* c, const void* dev, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    if (!dev || h < 2 || w < 2) return fail(c, MRTX_E_INVALID, "bad device DEM");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return ingest_dem(c, (const float*)dev, h, w);
}
// The context keeps the colour map in ROW PAIRS (color_pair_kernel): 8 bytes per texel, (h+1) x (w+4) elements + slack.
static int ingest_color(mrtx_ctx* c, const void* dev_src, int32_t h, int32_t w) {
    if (c->color) { HIPCHK(c, hipFree(c->color)); }
    c->color = nullptr; c->color_h = c->color_w = 0;
    if (!dev_src) return MRTX_OK;
    const size_t elems = (size_t)(h + 1) * (size_t)(w + 4) + 1;
    HIPCHK(c, hipMalloc((void**)&c->color, elems * 8));
    HIPCHK(c, mrtx_launch_color_pairs((const uint32_t*)dev_src, c->color, h, w, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->color_h = h; c->color_w = w;
    return MRTX_OK;
}
int mrtx_upload_color(mrtx_ctx* c, const uint8_t* rgba, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!rgba) return ingest_color(c, nullptr, 0, 0);
    if (h < 2 || w < 2) return fail(c, MRTX_E_INVALID, "colour texture must be (h>=2, w>=2, 4) uint8");
    const size_t bytes = (size_t)h * w * 4;
    void* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, bytes));
    hipError_t e = h2d(c, tmp, rgba, bytes);
    int rc = MRTX_OK;
    if (e == hipSuccess) rc = ingest_color(c, tmp, h, w);
    (void)hipFree(tmp);
    HIPCHK(c, e);
    return rc;
}
int mrtx_bind_color_device(mrtx_ctx* c, const void* dev, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (dev && (h < 2 || w < 2)) return fail(c, MRTX_E_INVALID, "bad device colour texture");
    return ingest_color(c, dev, h, w);   // the caller's buffer is read once; the context keeps its own row-pair copy
}
int mrtx_upload_background(mrtx_ctx* c, const uint8_t* rgba, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->bg) { HIPCHK(c, hipFree(c->bg)); }
    c->bg = nullptr; c->bg_h = c->bg_w = 0;
    c->scene_version++;
    if (!rgba) return MRTX_OK;
    if (h < 1 || w < 1) return fail(c, MRTX_E_INVALID, "background must be (h>=1, w>=1, 4) uint8");
    const size_t bytes = (size_t)h * w * 4;
    HIPCHK(c, hipMalloc((void**)&c->bg, bytes));
    HIPCHK(c, h2d(c, c->bg, rgba, bytes));
    c->bg_h = h; c->bg_w = w;
    c->scene_version++;
    return MRTX_OK;
}

int mrtx_upload_overlay(mrtx_ctx* c, const uint8_t* rgba, int32_t h, int32_t w) {
    if (!c) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!rgba) {
        if (c->overlay) { HIPCHK(c, hipFree(c->overlay)); }
        c->overlay = nullptr;
        return MRTX_OK;
    }
    if (h != c->cfg.height || w != c->cfg.width) return fail(c, MRTX_E_INVALID, "the overlay texture must match the frame (%d x %d)", c->cfg.width, c->cfg.height);
    const size_t bytes = (size_t)h * w * 4;
    if (!c->overlay) HIPCHK(c, hipMalloc((void**)&c->overlay, bytes));
    HIPCHK(c, h2d(c, c->overlay, rgba, bytes));
    return MRTX_OK;
}

int mrtx_set_params(mrtx_ctx* c, const MrtxParams* p) {
    if (!c || !p) return MRTX_E_INVALID;
    const uint32_t S = p->spp_per_launch;
    if (S == 0 || S > 64 || (S & (S - 1))) return fail(c, MRTX_E_INVALID, "spp_per_launch must be 1,2,4,...,64");
    if (!(p->marching_step > 0.0f) || !(p->marching_step_eps > 0.0f) || !(p->scene_epsilon >= 0.0f))
        return fail(c, MRTX_E_INVALID, "marching_step, marching_step_eps must be > 0 and scene_epsilon >= 0");
    if (!(p->tonemap_gamma > 0.0f)) return fail(c, MRTX_E_INVALID, "tonemap_gamma must be > 0");
    if (S != c->prm.spp_per_launch && c->blocks_done != 0)
        return fail(c, MRTX_E_STATE, "spp_per_launch cannot change inside an accumulation cycle; reset first");
    if (p->flags != c->prm.flags) c->scene_version++;   // MRTX_F_NO_CULL changes the cull and the gather layout
    c->prm = *p;
    return MRTX_OK;
}

int mrtx_set_camera(mrtx_ctx* c, const double eye[3], const double target[3], const double up[3], double vfov) {
    if (!c) return MRTX_E_INVALID;
    if (!check_vec(eye) || !check_vec(target) || !check_vec(up) || !(vfov > 0.0 && vfov < 180.0))
        return fail(c, MRTX_E_INVALID, "bad camera");
    double w[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]}, x[3];
    cross(w, up, x);
    if (!((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2] > 0.0)) return fail(c, MRTX_E_INVALID, "camera up is parallel to the view axis");
    for (int i = 0; i < 3; i++) { c->eye[i] = eye[i]; c->target[i] = target[i]; c->up[i] = up[i]; }
    c->vfov = vfov;
    c->scene_version++;
    return MRTX_OK;
}
int mrtx_set_moon_frame(mrtx_ctx* c, const double center[3], double radius, const double u[3], const double v[3]) {
    if (!c) return MRTX_E_INVALID;
    if (!check_vec(center) || !check_vec(u) || !check_vec(v) || !(radius > 0.0)) return fail(c, MRTX_E_INVALID, "bad moon frame");
    double x[3];
    cross(u, v, x);
    if (!((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2] > 0.0)) return fail(c, MRTX_E_INVALID, "moon u and v are parallel");
    for (int i = 0; i < 3; i++) { c->center[i] = center[i]; c->u[i] = u[i]; c->v[i] = v[i]; }
    c->radius = radius;
    c->scene_version++;
    return MRTX_OK;
}
int mrtx_set_light(mrtx_ctx* c, const double pos[3], double radius, double radiance) {
    if (!c) return MRTX_E_INVALID;
    if (!check_vec(pos) || !(radius >= 0.0) || !(radiance >= 0.0)) return fail(c, MRTX_E_INVALID, "bad light");
    for (int i = 0; i < 3; i++) c->light_pos[i] = pos[i];
    c->light_radius = radius; c->light_radiance = radiance;
    return MRTX_OK;
}
int mrtx_set_sun_disk(mrtx_ctx* c, const double pos[3], double radius, double radiance) {
    if (!c) return MRTX_E_INVALID;
    if (!check_vec(pos) || !std::isfinite(radius) || !(radiance >= 0.0)) return fail(c, MRTX_E_INVALID, "bad sun disk");
    for (int i = 0; i < 3; i++) c->sun_pos[i] = pos[i];
    c->sun_radius = radius; c->sun_radiance = radiance;
    c->scene_version++;
    return MRTX_OK;
}

int mrtx_set_capsules(mrtx_ctx* c, const float* caps12, int32_t n) {
    if (!c || n < 0 || (n > 0 && !caps12)) return MRTX_E_INVALID;
    for (int64_t i = 0; i < (int64_t)n * 12; i++)
        if (!std::isfinite(caps12[i])) return fail(c, MRTX_E_INVALID, "non-finite value in capsule %lld", (long long)(i / 12));
    c->caps_host.assign(caps12, caps12 + (size_t)n * 12);
    c->scene_version++;
    return MRTX_OK;
}

int mrtx_reset_accum(mrtx_ctx* c) {
    if (!c) return MRTX_E_INVALID;
    c->blocks_done = 0;
    c->path_nomem_chunks = 0;            // a new accumulation cycle may try the hand-over allocation again
    return MRTX_OK;
}

static int gather_layout(mrtx_ctx* c);
// slots [a, b) of part `part` of `n_parts` over n entries: contiguous, sizes differ by at most one
static void part_range(int n, int part, int n_parts, int& a, int& b) {
    a = (int)((int64_t)n * part / n_parts);
    b = (int)((int64_t)n * (part + 1) / n_parts);
}

int mrtx_render(mrtx_ctx* c, int32_t n_blocks, MrtxStats* out) { return mrtx_render_part(c, n_blocks, 0, 1, out); }

int mrtx_render_part(mrtx_ctx* c, int32_t n_blocks, int32_t part, int32_t n_parts, MrtxStats* out) {
    if (!c) return MRTX_E_INVALID;
    if (n_blocks < 1) return fail(c, MRTX_E_INVALID, "n_blocks must be >= 1");
    if (n_parts < 1 || part < 0 || part >= n_parts) return fail(c, MRTX_E_INVALID, "bad part %d of %d", part, n_parts);
    if (n_parts > 1 && (c->prm.flags & MRTX_F_NO_CULL)) return fail(c, MRTX_E_STATE, "parts need the tile list (MRTX_F_NO_CULL is set)");
    if (!c->dem) return fail(c, MRTX_E_STATE, "no displacement map: call mrtx_upload_dem first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!(c->prm.flags & MRTX_F_NO_SKIP)) { const int rc_ = ensure_mip(c); if (rc_ != MRTX_OK) return rc_; }
    if (c->caps_version != c->scene_version) {   // overlay bins follow the camera / moon frame / capsule list
        const int rc_ = build_capsule_bins(c);
        if (rc_ != MRTX_OK) return rc_;
        c->caps_version = c->scene_version;
        c->cull_version = 0;   // the sky cull depends on the bins
    }
    FrameC f;
    FrameCold cold;
    std::memset(&cold, 0, sizeof cold);      // padding included: the block is compared bytewise below
    build_frame(c, f, cold);
    f.cold = c->cold_dev;
    cold.caps = c->caps_dev; cold.caps_off = c->caps_off_dev; cold.caps_idx = c->caps_idx_dev;
    cold.n_caps = (int32_t)(c->caps_host.size() / 12);
    if (!c->cold_valid || std::memcmp(&cold, &c->cold_uploaded, sizeof cold) != 0) {
        std::memcpy(&c->cold_uploaded, &cold, sizeof cold);
        HIPCHK(c, hipMemcpyAsync(c->cold_dev, &c->cold_uploaded, sizeof cold, hipMemcpyHostToDevice, c->stream));
        c->cold_valid = true;
    }
    f.first_block = c->blocks_done;
    f.n_blocks = (uint32_t)n_blocks;
    const bool stats = (c->prm.flags & MRTX_F_COUNT_STATS) != 0;
    if (c->prm.flags & MRTX_F_FORCE_WIDE) f.dem_wide = 1;
    if (c->prm.flags & MRTX_F_NO_SKIP) f.mip = nullptr;   // (build_frame leaves the horizon mip out as well)
    uint64_t culled_px = 0;
    if (c->tile_dirty.size() != (size_t)c->n_local) c->tile_dirty.assign((size_t)c->n_local, 0);
    const bool overlay = !c->caps_host.empty();
    bool culling = false;
    if (!(c->prm.flags & MRTX_F_NO_CULL)) {
        if (c->cull_version != c->scene_version) {
            cull_tiles(c, c->cfg.rank, c->keep_cached, c->culled_px_cached, &c->keep_front_cached);
            c->cull_version = c->scene_version;
        }
        culled_px = c->culled_px_cached;
        culling = true;   // the list also carries the launch ORDER, so it is used even when it holds every tile
    }
    const std::vector<int32_t>& keep = c->keep_cached;
    int moon_n = -1;                     // leading tiles of this launch's list that can see more than sky (-1: no split known)
    if (culling) {
        if (keep != c->keep_uploaded) {
            if (!keep.empty()) {
                HIPCHK(c, hipMemcpyAsync(c->tile_list_dev, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));   // `keep` is pageable host memory
            }
            c->keep_uploaded = keep;
        }
        int pa = 0, pb = (int)keep.size();
        if (n_parts > 1) {   // cut where the shard parts cut: at slot numbers of the COMMON (longest) list
            const int rc_ = gather_layout(c);
            if (rc_ != MRTX_OK) return rc_;
            if (!c->act_on) return fail(c, MRTX_E_STATE, "parts need the active gather layout (mrtx_shard_parts)");
            part_range(c->act_slots, part, n_parts, pa, pb);
            pa = std::min(pa, (int)keep.size()); pb = std::min(pb, (int)keep.size());
        }
        f.tile_list = c->tile_list_dev + pa;
        f.n_active = pb - pa;
        if (c->bg != nullptr) moon_n = std::max(0, std::min(pb, c->keep_front_cached) - pa);
        // culled tiles are all-zero by construction: clear only if an earlier view rendered into one of them
        std::vector<uint8_t> kept((size_t)c->n_local, 0);
        for (int32_t lt : keep) kept[(size_t)lt] = 1;
        bool stale = false;
        for (int lt = 0; lt < c->n_local && !stale; lt++) stale = c->tile_dirty[(size_t)lt] && !kept[(size_t)lt];
        if (stale) {
            const size_t fb = (size_t)c->cfg.width * c->cfg.height * 16;
            HIPCHK(c, hipMemsetAsync(c->accum, 0, fb, c->stream));
            HIPCHK(c, hipMemsetAsync(c->hits, 0, fb, c->stream));
            c->tile_dirty.assign((size_t)c->n_local, 0);
        }
        for (int32_t lt : keep) c->tile_dirty[(size_t)lt] = 1;
    } else {
        culled_px = 0;
        c->tile_dirty.assign((size_t)c->n_local, 1);
    }
    if (stats) HIPCHK(c, hipMemsetAsync(c->stats_dev, 0, 32 * sizeof(unsigned long long), c->stream));
    // D6 (path_seg_max > 1): by default the path continues in path_kernel behind a queue (mode 2);
    // MRTX_F_INWAVE_PATHS keeps it inside the render wave (mode 1) -- same result bit for bit, slower.
    const int S = (int)c->prm.spp_per_launch;
    int mode = c->prm.path_seg_max > 1 ? ((c->prm.flags & MRTX_F_INWAVE_PATHS) ? 1 : 2) : 0;
    if (mode == 2) {
        const uint64_t tiles = f.tile_list ? (uint64_t)(moon_n >= 0 ? moon_n : f.n_active) : (uint64_t)c->n_local;
        if (tiles * (uint64_t)(f.tile_w * f.tile_h) * (uint64_t)S * (uint64_t)n_blocks < c->path_queue_min) mode = 1;
    }
    // With an environment map the sky-only tiles at the end of the list get a launch of their own (render_kernel<MODE 3>: a
    // sample is its environment texel): a small kernel at full occupancy, and the path pipeline is set up for the rest only.
    FrameC fsky = f;
    bool have_sky = false;
    if (moon_n >= 0 && moon_n < f.n_active) {
        fsky.tile_list = f.tile_list + moon_n; fsky.n_active = f.n_active - moon_n;
        fsky.first_block = c->blocks_done; fsky.n_blocks = (uint32_t)n_blocks;
        f.n_active = moon_n;
        have_sky = true;
    }
    // Deferred paths need hand-over buffers and a 24-bit step count per record.  When either is not to be had, the frame is
    // rendered with the paths inside the render wave instead (mode 1: no buffers, the same frame bit for bit, slower) -- a
    // render call does not fail where an identical result exists.  Said once per context on stderr.
    std::vector<FrameC> fs;
    std::vector<PathQ> pqs;
    int n_sub = 1;
    bool overlap = false;                // path stage of sub-part i beside the render kernel of sub-part i + 1 (MOONRT_PATH_OVERLAP)
    auto fall_back = [&](const char* why) {
        if (!c->path_fallback_said) {
            std::fprintf(stderr, "libmoonrt: %s -- frames of this size keep their paths inside the render wave (same result, slower); "
                                 "retried after mrtx_reset_accum or when a frame needs less\n", why);
            c->path_fallback_said = true;
        }
        mode = 1;
    };
    if (mode == 2 && f.kmax >= (1 << 24)) fall_back("marching_step is too small for the 24-bit step count of a hand-over record");
    if (mode == 2) {
    // The hand-over buffers are sized for the worst case (64 bytes for each of the 64 lanes of every wave-job of the
    // launch: 12 GB for the whole-disc cfg3 frame, 129 GB for a cfg4 frame whose every pixel is on the Moon).  A frame
    // that needs more than path_budget_bytes is rendered in SUB-PARTS of its tile list, one after the other through the
    // same buffers: render(sub) -> paths(sub) -> resolve(sub); sub-parts cover disjoint pixels.
    // ... and never more than half of what the device has free right now (unless MOONRT_PATH_MAX_GB says otherwise): a second
    // context, a torch process or a 34 GB cfg4 DEM on the same GPU must not turn the fixed budget into an allocation failure
    uint64_t budget = c->path_budget_bytes;
    const uint64_t full_chunks = mrtx_path_chunks(f, S, nullptr, nullptr);
    if (!c->path_budget_env && full_chunks > c->path_cap) {      // asked only when the buffers have to grow (a driver call: not per frame)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t held = (uint64_t)c->path_cap * (uint64_t)c->path_sets * 64ull * MRTX_PATH_REC_BYTES;   // what this context already holds counts as free
            budget = std::min<uint64_t>(budget, std::max<uint64_t>(1ull << 28, ((uint64_t)free_b + held) / 2));
        } else {
            (void)hipGetLastError();
        }
    }
    const uint64_t cap_chunks = std::max<uint64_t>(std::max<uint64_t>(4096, budget / (64ull * MRTX_PATH_REC_BYTES)), c->path_cap);
    n_sub = 1;
    {
        const uint64_t full = full_chunks;
        if (full > cap_chunks && f.tile_list != nullptr) n_sub = (int)std::min<uint64_t>((uint64_t)f.n_active, (full + cap_chunks - 1) / cap_chunks);
    }
    overlap = c->path_overlap > 1 && f.tile_list != nullptr && f.n_active >= 8 * c->path_overlap && !stats;
    if (overlap) n_sub = std::max(n_sub, c->path_overlap);
    const int sets = overlap ? 2 : 1;
    fs.assign((size_t)n_sub, f);
    pqs.assign((size_t)n_sub, PathQ());
    uint64_t chunks = 0;
    for (int s = 0; s < n_sub; s++) {
        int a_, b_;
        part_range(f.n_active, s, n_sub, a_, b_);
        if (n_sub > 1) { fs[(size_t)s].tile_list = f.tile_list + a_; fs[(size_t)s].n_active = b_ - a_; }
        std::memset(&pqs[(size_t)s], 0, sizeof(PathQ));
        chunks = std::max(chunks, mrtx_path_chunks(fs[(size_t)s], S, &pqs[(size_t)s].grid_a, &pqs[(size_t)s].njobs_log2));
    }
    if (chunks * 64ull > 0xFFFFFFFFull) {
        fall_back("the frame holds more wave-jobs than one deferred-path launch can index");
    } else if (chunks > c->path_cap && c->path_nomem_chunks != 0 && chunks >= c->path_nomem_chunks) {
        mode = 1;      // an allocation of this size failed before: do not free and retry three multi-gigabyte hipMallocs per frame
    } else if (chunks > c->path_cap || sets > c->path_sets) {
        if (c->path_rec) { HIPCHK(c, hipFree(c->path_rec)); c->path_rec = nullptr; }
        if (c->path_meta) { HIPCHK(c, hipFree(c->path_meta)); c->path_meta = nullptr; }
        if (c->path_npaths) { HIPCHK(c, hipFree(c->path_npaths)); c->path_npaths = nullptr; }
        c->path_cap = 0;
        if (c->path_alloc_fail_test ||
            hipMalloc((void**)&c->path_rec, (size_t)sets * (size_t)chunks * 64 * MRTX_PATH_REC_BYTES) != hipSuccess ||
            hipMalloc((void**)&c->path_meta, (size_t)sets * (size_t)chunks * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&c->path_npaths, (size_t)sets * ((size_t)chunks + 64)) != hipSuccess) {
            (void)hipGetLastError();                                    // clear the allocation error: the frame is still rendered
            if (c->path_rec) { (void)hipFree(c->path_rec); c->path_rec = nullptr; }
            if (c->path_meta) { (void)hipFree(c->path_meta); c->path_meta = nullptr; }
            if (c->path_npaths) { (void)hipFree(c->path_npaths); c->path_npaths = nullptr; }
            c->path_nomem_chunks = chunks;                              // latched until less is needed or mrtx_reset_accum
            fall_back("no device memory for the hand-over records");
        } else {
            c->path_cap = chunks;
            c->path_sets = sets;
            c->path_nomem_chunks = 0;
        }
    }
    const size_t n = (size_t)c->path_cap * 64;
    if (mode == 2 && !c->path_ctr) HIPCHK(c, hipMalloc((void**)&c->path_ctr, 2 * 8 * 16 * sizeof(uint32_t)));
    for (int s = 0; mode == 2 && s < n_sub; s++) {
        PathQ& pq = pqs[(size_t)s];
        const size_t set = overlap ? (size_t)(s & 1) : 0;     // ping-pong: sub-part s hands over through buffer set s % 2
        pq.ray0 = reinterpret_cast<float4*>(reinterpret_cast<char*>(c->path_rec) + set * n * MRTX_PATH_REC_BYTES); pq.ray1 = pq.ray0 + n; pq.ray2 = pq.ray1 + n;
#if MRTX_C_AOS == 2
        pq.c4 = pq.ray2 + n; pq.c0 = pq.c1 = pq.c2 = nullptr;
        pq.lane_of = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(pq.c4) + 3 * n);     // 12 bytes per sample
#elif MRTX_C_AOS
        pq.c4 = pq.ray2 + n; pq.c0 = pq.c1 = pq.c2 = nullptr;
        pq.lane_of = reinterpret_cast<uint32_t*>(pq.c4 + n);
#else
        pq.c0 = reinterpret_cast<float*>(pq.ray2 + n); pq.c1 = pq.c0 + n; pq.c2 = pq.c1 + n; pq.c4 = nullptr;
        pq.lane_of = reinterpret_cast<uint32_t*>(pq.c2 + n);
#endif
        pq.npaths = c->path_npaths + set * ((size_t)c->path_cap + 64);
        pq.meta = c->path_meta + set * (size_t)c->path_cap;
        pq.counters = c->path_ctr + set * 8 * 16; pq.n_sub = c->path_nsub; pq.grp_log2 = c->path_grp_log2;
        pq.n_chunks = (uint32_t)((uint64_t)pq.grid_a << pq.njobs_log2);
        pq.s_log2 = 0;
        while ((1 << pq.s_log2) < S) pq.s_log2++;
        { const int P = 64 / S; const int PW = P >= 32 ? 8 : P >= 8 ? 4 : P >= 2 ? 2 : 1; pq.pw_log2 = PW == 8 ? 3 : PW == 4 ? 2 : PW == 2 ? 1 : 0; }
        pq.refill_min = c->path_refill; pq.seg_min = c->path_segmin; pq.rare_min = c->path_hitmin;
    }
    }
    double primary_ms = 0.0, paths_ms = 0.0;
    uint32_t launches = 0;
    if (mode != 2) {
        // one launch per block of S samples (round 4: a kernel that knows it traces ONE block keeps nothing alive from block to
        // block -- 25 to 48 VGPRs less, see render_kernel; a launch costs ~10 us, a block of a 4K frame milliseconds)
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        for (int32_t b = 0; b < n_blocks; b++) {
            FrameC fb = f, fsb = fsky;
            fb.first_block = fsb.first_block = c->blocks_done + (uint32_t)b;
            fb.n_blocks = fsb.n_blocks = 1;
            if (f.n_active > 0 || !f.tile_list) HIPCHK(c, mrtx_launch_render(fb, S, stats, mode, overlay, nullptr, c->stream));
            if (have_sky) HIPCHK(c, mrtx_launch_render(fsb, S, stats, 3, false, nullptr, c->stream));
        }
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        primary_ms = ms;
        launches = (((f.n_active > 0 || !f.tile_list) ? 1u : 0u) + (have_sky ? 1u : 0u)) * (uint32_t)n_blocks;
    } else {
        const int wi = (stats ? 2 : 0) + (f.dem_wide ? 1 : 0);
        if (c->path_waves[wi] == 0) {
            int nw = 0;
            if (mrtx_path_waves(stats, f.dem_wide != 0, &nw) != 0 || nw < 8) return fail(c, MRTX_E_DEVICE, "cannot size the path_kernel launch");
            c->path_waves[wi] = nw;
        }
        const int nw = c->path_waves_env ? c->path_waves_env : c->path_waves[wi];
        const size_t n_ev = (size_t)n_blocks * (size_t)n_sub * 3;
        while (c->evs.size() < n_ev) { hipEvent_t e; HIPCHK(c, get_event(c->cfg.device, &e)); c->evs.push_back(e); }
        if (overlap) {
            if (!c->stream2) HIPCHK(c, get_stream(c->cfg.device, &c->stream2));
            for (int i = 0; i < 2; i++) if (!c->ov_done[i]) HIPCHK(c, get_event(c->cfg.device, &c->ov_done[i]));
            if (!c->ov_join) HIPCHK(c, get_event(c->cfg.device, &c->ov_join));
        }
        size_t ov_idx = 0;                 // sub-part launches so far in this call (overlap: which buffer set / which `done` event)
        for (int32_t b = 0; b < n_blocks && f.n_active > 0; b++) {
            for (int s = 0; s < n_sub; s++) {
                hipEvent_t* ev = &c->evs[((size_t)b * (size_t)n_sub + (size_t)s) * 3];
                FrameC fb = fs[(size_t)s];
                fb.first_block = c->blocks_done + (uint32_t)b;
                fb.n_blocks = 1;
                PathQ& pq = pqs[(size_t)s];
                pq.gs_base = fb.first_block * (uint32_t)S;
                if (overlap) {
                    // render(i) on the context's stream, paths(i) + resolve(i) on stream2 beside render(i + 1); buffer set i % 2 is
                    // free again when paths(i - 2) has finished
                    const int set = s & 1;
                    if (ov_idx >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ov_done[set], 0));
                    HIPCHK(c, hipMemsetAsync(pq.meta, 0, (size_t)pq.n_chunks * sizeof(uint32_t), c->stream));
                    HIPCHK(c, hipMemsetAsync(pq.npaths, 0, (size_t)pq.n_chunks, c->stream));
                    HIPCHK(c, hipMemsetAsync(pq.counters, 0, 8 * 16 * sizeof(uint32_t), c->stream));
                    HIPCHK(c, hipEventRecord(ev[0], c->stream));
                    HIPCHK(c, mrtx_launch_render(fb, S, stats, 2, overlay, &pq, c->stream));
                    HIPCHK(c, hipEventRecord(ev[1], c->stream));
                    HIPCHK(c, hipStreamWaitEvent(c->stream2, ev[1], 0));
                    const bool last = (b == n_blocks - 1) && (s == n_sub - 1);
                    HIPCHK(c, mrtx_launch_paths(fb, pq, S, stats, last ? nw : std::min(nw, c->path_overlap_waves), c->stream2));
                    HIPCHK(c, hipEventRecord(c->ov_done[set], c->stream2));
                    HIPCHK(c, hipEventRecord(ev[2], c->stream2));
                    ov_idx++;
                    continue;
                }
                HIPCHK(c, hipMemsetAsync(pq.meta, 0, (size_t)pq.n_chunks * sizeof(uint32_t), c->stream));
                HIPCHK(c, hipMemsetAsync(pq.npaths, 0, (size_t)pq.n_chunks, c->stream));
                HIPCHK(c, hipMemsetAsync(pq.counters, 0, 8 * 16 * sizeof(uint32_t), c->stream));
                HIPCHK(c, hipEventRecord(ev[0], c->stream));
                HIPCHK(c, mrtx_launch_render(fb, S, stats, 2, overlay, &pq, c->stream));
                HIPCHK(c, hipEventRecord(ev[1], c->stream));
                HIPCHK(c, mrtx_launch_paths(fb, pq, S, stats, nw, c->stream));
                HIPCHK(c, hipEventRecord(ev[2], c->stream));
            }
        }
        if (overlap && ov_idx > 0) {       // the context's stream goes on (sky tiles, watchdog read-back) when stream2 has drained
            HIPCHK(c, hipEventRecord(c->ov_join, c->stream2));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ov_join, 0));
        }
        if (have_sky) {   // nothing but the environment can be seen from these tiles: no paths, no records
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            for (int32_t b = 0; b < n_blocks; b++) {
                FrameC fsb = fsky;
                fsb.first_block = c->blocks_done + (uint32_t)b; fsb.n_blocks = 1;
                HIPCHK(c, mrtx_launch_render(fsb, S, stats, 3, false, nullptr, c->stream));
            }
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        }
        // path_kernel's watchdog (stats[15]): a wave gave up after 2^24 iterations -- the frame is incomplete.  Read through
        // pinned memory on the stream, so that the one synchronisation below covers it (no second blocking copy per call).
        if (!c->wd_host) HIPCHK(c, hipHostMalloc((void**)&c->wd_host, sizeof(unsigned long long), hipHostMallocDefault));
        HIPCHK(c, hipMemcpyAsync(c->wd_host, c->stats_dev + 15, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        {
            const unsigned long long wd = *c->wd_host;
            if (wd != 0) {
                HIPCHK(c, hipMemset(c->stats_dev + 15, 0, sizeof wd));
                return fail(c, MRTX_E_DEVICE, "path_kernel watchdog: %llu wave(s) did not finish their paths", wd);
            }
        }
        if (have_sky) {
            float a = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&a, c->ev0, c->ev1));
            primary_ms += a;
        }
        const size_t n_launched = f.n_active > 0 ? (size_t)n_blocks * (size_t)n_sub : 0;
        if (overlap && n_launched > 0) {
            // the two stages run side by side: primary_ms = first render start .. last render end, paths_ms = what of the path
            // stage is still running after that (the EXPOSED part)
            float a = 0.0f, p = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&a, c->evs[0], c->evs[(n_launched - 1) * 3 + 1]));
            HIPCHK(c, hipEventElapsedTime(&p, c->evs[(n_launched - 1) * 3 + 1], c->evs[(n_launched - 1) * 3 + 2]));
            primary_ms += a; paths_ms += p;
        }
        for (size_t i = 0; !overlap && i < n_launched; i++) {
            float a = 0.0f, p = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&a, c->evs[i * 3], c->evs[i * 3 + 1]));
            HIPCHK(c, hipEventElapsedTime(&p, c->evs[i * 3 + 1], c->evs[i * 3 + 2]));
            primary_ms += a; paths_ms += p;
        }
        launches = (f.n_active > 0 ? 3u * (uint32_t)n_blocks * (uint32_t)n_sub : 0u) + (have_sky ? (uint32_t)n_blocks : 0u);
    }
    if (part == n_parts - 1) c->blocks_done += (uint32_t)n_blocks;
    if (out) {
        std::memset(out, 0, sizeof *out);
        out->kernel_ms = primary_ms + paths_ms;
        out->primary_ms = primary_ms;
        out->paths_ms = paths_ms;
        out->launches = launches;
        if (stats) {
            unsigned long long h[32];
            HIPCHK(c, hipMemcpy(h, c->stats_dev, sizeof h, hipMemcpyDeviceToHost));
            out->camera_height_samples = h[3]; out->camera_dem_fetches = h[6]; out->camera_mip_fetches = h[7];
            out->camera_colour_fetches = h[4]; out->camera_background_fetches = h[5];
            for (int i = 0; i < 15; i++) h[i] += h[16 + i];          // totals = render_kernel's + path_kernel's
            out->primary_rays = h[0] + (part == 0 ? culled_px : 0) * (uint64_t)c->prm.spp_per_launch * (uint64_t)n_blocks;
            out->primary_hits = h[1]; out->shadow_rays = h[2];
            out->height_samples = h[3]; out->colour_fetches = h[4]; out->background_fetches = h[5];
            out->dem_fetches = h[6]; out->mip_fetches = h[7]; out->bounce_rays = h[8]; out->bounce_sun_hits = h[9];
            if (std::getenv("MOONRT_DEBUG_STATS"))   // measurement builds (-DMRTX_PROF_MARGIN): the spare counter slots
                std::fprintf(stderr, "libmoonrt debug stats[10..14]: %llu %llu %llu %llu %llu\n", h[10], h[11], h[12], h[13], h[14]);
        }
    }
    return MRTX_OK;
}

int mrtx_samples_done(mrtx_ctx* c, uint32_t* out) {
    if (!c || !out) return MRTX_E_INVALID;
    *out = c->blocks_done * c->prm.spp_per_launch;
    return MRTX_OK;
}

// Thresholds of the exact "Gamma" post-process: T[0] = 0 (never read as a threshold),
// T[j] = (float)pow((j - 0.5) / n, gamma) for j = 1..n, float64 on the host, rounded once; n + 1 entries on the device,
// rebuilt when tonemap_gamma changes.  Level = #{j : exposure * mean >= T[j]} = round(n * (exposure * mean)^(1/gamma)).
static int tone_table(mrtx_ctx* c, int n, float** dev, float* built_for) {
    if (*dev && *built_for == c->prm.tonemap_gamma) return MRTX_OK;
    std::vector<float> T((size_t)n + 1);
    const double g = (double)c->prm.tonemap_gamma;
    T[0] = 0.0f;
    for (int j = 1; j <= n; j++) T[(size_t)j] = (float)pow(((double)j - 0.5) / (double)n, g);
    if (!*dev) HIPCHK(c, hipMalloc((void**)dev, T.size() * sizeof(float)));
    HIPCHK(c, hipMemcpyAsync(*dev, T.data(), T.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // T is a local
    *built_for = c->prm.tonemap_gamma;
    return MRTX_OK;
}

int mrtx_read_linear(mrtx_ctx* c, float* out) {
    if (!c || !out) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int64_t npix = (int64_t)c->cfg.width * c->cfg.height;
    HIPCHK(c, mrtx_launch_resolve_linear(c->accum, (float*)c->scratch, npix, c->blocks_done * c->prm.spp_per_launch, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->scratch, (size_t)npix * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}
int mrtx_read_rgba8(mrtx_ctx* c, uint8_t* out) {
    if (!c || !out) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int64_t npix = (int64_t)c->cfg.width * c->cfg.height;
    const int rc = tone_table(c, 255, &c->tone8_dev, &c->tone8_gamma);
    if (rc != MRTX_OK) return rc;
    HIPCHK(c, mrtx_launch_resolve_rgba8(c->accum, (uint32_t*)c->scratch, npix, c->blocks_done * c->prm.spp_per_launch,
                                        c->prm.tonemap_exposure, c->tone8_dev, c->overlay, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->scratch, (size_t)npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}
int mrtx_read_rgb16(mrtx_ctx* c, uint16_t* out) {
    if (!c || !out) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int64_t npix = (int64_t)c->cfg.width * c->cfg.height;
    const int rc = tone_table(c, 65535, &c->tone16_dev, &c->tone16_gamma);
    if (rc != MRTX_OK) return rc;
    HIPCHK(c, mrtx_launch_resolve_rgb16(c->accum, (uint16_t*)c->scratch, npix, c->blocks_done * c->prm.spp_per_launch,
                                        c->prm.tonemap_exposure, c->tone16_dev, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->scratch, (size_t)npix * 6, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}
int mrtx_read_hits(mrtx_ctx* c, float* out) {
    if (!c || !out) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t bytes = (size_t)c->cfg.width * c->cfg.height * 16;
    HIPCHK(c, hipMemcpyAsync(out, c->hits, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}

int mrtx_read_hit(mrtx_ctx* c, int32_t x, int32_t y, float out[4]) {
    if (!c || !out) return MRTX_E_INVALID;
    if (x < 0 || y < 0 || x >= c->cfg.width || y >= c->cfg.height) return fail(c, MRTX_E_INVALID, "pixel (%d, %d) outside the frame", x, y);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpyAsync(out, c->hits + 4 * ((size_t)y * (size_t)c->cfg.width + (size_t)x), 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}

// The layout both ends of the gather use for the scene as it stands: active tiles only while the sky cull is in force
// (no environment map, no overlay geometry, culling not disabled), the full layout otherwise.  Every rank derives
// every rank's list from its own copy of the scene -- identical on all ranks by contract -- so nothing is negotiated.
static int gather_layout(mrtx_ctx* c) {
    if (c->act_version == c->scene_version) return MRTX_OK;
    c->act_on = false;
    c->act_slots = c->slots;
    const bool can = c->cfg.world > 1 && !(c->prm.flags & MRTX_F_NO_CULL) && c->bg == nullptr && c->caps_host.empty();
    if (can) {
        c->act_lists.assign((size_t)c->cfg.world, std::vector<int32_t>());
        int longest = 0;
        uint64_t px = 0;
        for (int r = 0; r < c->cfg.world; r++) {
            cull_tiles(c, r, c->act_lists[(size_t)r], px);
            longest = std::max(longest, (int)c->act_lists[(size_t)r].size());
        }
        if (longest < c->slots) { c->act_on = true; c->act_slots = longest; }
    }
    c->act_version = c->scene_version;
    return MRTX_OK;
}
static int upload_layout(mrtx_ctx* c) {
    if (!c->act_on || c->act_uploaded_version == c->act_version) return MRTX_OK;
    const size_t n = (size_t)c->cfg.world * (size_t)std::max(1, c->act_slots);
    std::vector<int32_t> flat(n, -1);
    for (int r = 0; r < c->cfg.world; r++)
        std::copy(c->act_lists[(size_t)r].begin(), c->act_lists[(size_t)r].end(), flat.begin() + (size_t)r * (size_t)c->act_slots);
    if (c->act_dev_cap < n) {
        if (c->act_dev) { HIPCHK(c, hipFree(c->act_dev)); c->act_dev = nullptr; }
        HIPCHK(c, hipMalloc((void**)&c->act_dev, n * sizeof(int32_t)));
        c->act_dev_cap = n;
    }
    HIPCHK(c, h2d(c, c->act_dev, flat.data(), n * sizeof(int32_t)));
    c->act_uploaded_version = c->act_version;
    return MRTX_OK;
}
static const int32_t* layout_list(const mrtx_ctx* c, int rank) {
    return c->act_on ? c->act_dev + (size_t)rank * (size_t)c->act_slots : nullptr;
}
// root: tiles of other ranks that still hold an earlier view's data and are sky now must read as zero
static int clear_stale_peer_tiles(mrtx_ctx* c) {
    if (c->peer_written.size() != (size_t)c->n_tiles) c->peer_written.assign((size_t)c->n_tiles, 0);
    if (!c->act_on) return MRTX_OK;
    std::vector<uint8_t> active((size_t)c->n_tiles, 0);
    for (int r = 0; r < c->cfg.world; r++)
        for (int32_t lt : c->act_lists[(size_t)r]) active[(size_t)lt * (size_t)c->cfg.world + (size_t)r] = 1;
    std::vector<int32_t> stale;
    for (int t = 0; t < c->n_tiles; t++)
        if (t % c->cfg.world != c->cfg.rank && c->peer_written[(size_t)t] && !active[(size_t)t]) { stale.push_back(t); c->peer_written[(size_t)t] = 0; }
    if (stale.empty()) return MRTX_OK;
    if (c->stale_cap < stale.size()) {
        if (c->stale_dev) { HIPCHK(c, hipFree(c->stale_dev)); c->stale_dev = nullptr; }
        HIPCHK(c, hipMalloc((void**)&c->stale_dev, (size_t)c->n_tiles * sizeof(int32_t)));
        c->stale_cap = (size_t)c->n_tiles;
    }
    HIPCHK(c, h2d(c, c->stale_dev, stale.data(), stale.size() * sizeof(int32_t)));
    HIPCHK(c, mrtx_launch_zero_tiles(c->accum, c->hits, c->stale_dev, (int)stale.size(), c->cfg.width, c->cfg.height,
                                     c->cfg.tile_w, c->cfg.tile_h, c->tiles_x, c->tile_shift, c->stream));
    return MRTX_OK;
}
static void mark_peer_written(mrtx_ctx* c, int src_rank) {
    if (c->peer_written.size() != (size_t)c->n_tiles) c->peer_written.assign((size_t)c->n_tiles, 0);
    if (c->act_on) {
        for (int32_t lt : c->act_lists[(size_t)src_rank]) c->peer_written[(size_t)lt * (size_t)c->cfg.world + (size_t)src_rank] = 1;
    } else {
        for (int t = src_rank; t < c->n_tiles; t += c->cfg.world) c->peer_written[(size_t)t] = 1;
    }
}

int mrtx_set_gather_hits(mrtx_ctx* c, int32_t on) {
    if (!c) return MRTX_E_INVALID;
    c->gather_hits = on != 0;
    return MRTX_OK;
}
int mrtx_shard_bytes(mrtx_ctx* c, int32_t rank, uint64_t* out) {
    if (!c || !out || rank < 0 || rank >= c->cfg.world) return MRTX_E_INVALID;
    *out = (uint64_t)c->slots * c->cfg.tile_w * c->cfg.tile_h * (c->gather_hits ? 32ull : 16ull);  // equal for every rank (padded): the upper bound
    return MRTX_OK;
}
int mrtx_shard_bytes_active(mrtx_ctx* c, uint64_t* out) {
    if (!c || !out) return MRTX_E_INVALID;
    const int rc = gather_layout(c);
    if (rc != MRTX_OK) return rc;
    *out = (uint64_t)c->act_slots * c->cfg.tile_w * c->cfg.tile_h * (c->gather_hits ? 32ull : 16ull);
    return MRTX_OK;
}
int mrtx_pack_shard(mrtx_ctx* c, void* dev_dst, void* hip_stream) { return mrtx_pack_part(c, dev_dst, 0, 1, nullptr, nullptr, hip_stream); }

// Part `part` of `n_parts` of the shard: the slots whose tiles mrtx_render_part(.., part, n_parts) rendered (own
// tiles first come first; the padding up to the common slot count goes with the last part).
int mrtx_pack_part(mrtx_ctx* c, void* dev_dst, int32_t part, int32_t n_parts, uint64_t* byte_off, uint64_t* byte_len,
                   void* hip_stream) {
    if (!c || !dev_dst || n_parts < 1 || part < 0 || part >= n_parts) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = gather_layout(c);
    if (rc == MRTX_OK) rc = upload_layout(c);
    if (rc != MRTX_OK) return rc;
    if (n_parts > 1 && !c->act_on) return fail(c, MRTX_E_STATE, "the shard moves in parts only in the active layout (mrtx_shard_parts)");
    // every rank cuts at the SAME slot numbers (parts of the longest list), so part k of every peer lands at one offset
    int a, b;
    part_range(c->act_slots, part, n_parts, a, b);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HIPCHK(c, mrtx_launch_pack(c->accum, c->hits, dev_dst, c->cfg.width, c->cfg.height, c->cfg.tile_w, c->cfg.tile_h,
                               c->tiles_x, c->n_tiles, c->cfg.rank, c->cfg.world, a, b - a, layout_list(c, c->cfg.rank),
                               c->tile_shift, c->gather_hits ? 1 : 0, st));
    const uint64_t slot_bytes = (uint64_t)c->cfg.tile_w * c->cfg.tile_h * (c->gather_hits ? 32ull : 16ull);
    if (byte_off) *byte_off = (uint64_t)a * slot_bytes;
    if (byte_len) *byte_len = (uint64_t)(b - a) * slot_bytes;
    if (!hip_stream) HIPCHK(c, hipStreamSynchronize(st));
    return MRTX_OK;
}
// How many parts the exchange may use for the scene as it stands (1 when the full layout is in force), and which
// render part produces the tiles of shard part k: shard parts cut the COMMON slot range, render parts must cover them.
int mrtx_shard_parts(mrtx_ctx* c, int32_t wanted, int32_t* out) {
    if (!c || !out || wanted < 1) return MRTX_E_INVALID;
    const int rc = gather_layout(c);
    if (rc != MRTX_OK) return rc;
    *out = (c->act_on && c->act_slots >= wanted) ? wanted : 1;
    return MRTX_OK;
}
int mrtx_unpack_shard(mrtx_ctx* c, int32_t src_rank, const void* dev_src, void* hip_stream) {
    if (!c || !dev_src || src_rank < 0 || src_rank >= c->cfg.world) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = gather_layout(c);
    if (rc == MRTX_OK) rc = upload_layout(c);
    if (rc == MRTX_OK) rc = clear_stale_peer_tiles(c);
    if (rc != MRTX_OK) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (hip_stream) HIPCHK(c, hipStreamSynchronize(c->stream));   // the stale-tile clear ran on the context's stream
    HIPCHK(c, mrtx_launch_unpack(c->accum, c->hits, dev_src, c->cfg.width, c->cfg.height, c->cfg.tile_w,
                                 c->cfg.tile_h, c->tiles_x, c->n_tiles, src_rank, c->cfg.world, c->act_slots,
                                 layout_list(c, src_rank), c->tile_shift, c->gather_hits ? 1 : 0, st));
    mark_peer_written(c, src_rank);
    if (!hip_stream) HIPCHK(c, hipStreamSynchronize(st));
    return MRTX_OK;
}

int mrtx_unpack_all(mrtx_ctx* c, const void* const* dev_srcs, int32_t n) {
    if (!c || !dev_srcs || n != c->cfg.world) return MRTX_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = gather_layout(c);
    if (rc == MRTX_OK) rc = upload_layout(c);
    if (rc == MRTX_OK) rc = clear_stale_peer_tiles(c);
    if (rc != MRTX_OK) return rc;
    for (int r = 0; r < n; r++)
        if (r != c->cfg.rank && !dev_srcs[r]) return fail(c, MRTX_E_INVALID, "missing shard of rank %d", r);
    if (n <= 16) {      // one launch for all peers (pointers by value in the kernel arguments)
        HIPCHK(c, mrtx_launch_unpack_all(c->accum, c->hits, dev_srcs, c->cfg.width, c->cfg.height, c->cfg.tile_w, c->cfg.tile_h,
                                         c->tiles_x, c->n_tiles, c->cfg.rank, c->cfg.world, c->act_slots,
                                         c->act_on ? c->act_dev : nullptr, c->tile_shift, c->gather_hits ? 1 : 0, c->stream));
        for (int r = 0; r < n; r++)
            if (r != c->cfg.rank) mark_peer_written(c, r);
    } else {
        for (int r = 0; r < n; r++) {
            if (r == c->cfg.rank) continue;
            HIPCHK(c, mrtx_launch_unpack(c->accum, c->hits, dev_srcs[r], c->cfg.width, c->cfg.height, c->cfg.tile_w,
                                         c->cfg.tile_h, c->tiles_x, c->n_tiles, r, c->cfg.world, c->act_slots,
                                         layout_list(c, r), c->tile_shift, c->gather_hits ? 1 : 0, c->stream));
            mark_peer_written(c, r);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MRTX_OK;
}

int mrtx_device_ptr(mrtx_ctx* c, int32_t which, void** out, uint64_t* bytes) {
    if (!c || !out) return MRTX_E_INVALID;
    const uint64_t fb = (uint64_t)c->cfg.width * c->cfg.height * 16;
    switch (which) {
        case MRTX_BUF_ACCUM: *out = c->accum; if (bytes) *bytes = fb; break;
        case MRTX_BUF_HITS: *out = c->hits; if (bytes) *bytes = fb; break;
        case MRTX_BUF_DEM: *out = c->dem; if (bytes) *bytes = c->dem ? (uint64_t)(c->dem_h + 4) * (c->dem_w + 4) * MRTX_DEM_ELEM_BYTES : 0; break;
        case MRTX_BUF_COLOR: *out = c->color; if (bytes) *bytes = c->color ? ((uint64_t)(c->color_h + 1) * (c->color_w + 4) + 1) * 8 : 0; break;
        default: return fail(c, MRTX_E_INVALID, "unknown buffer id %d", which);
    }
    return MRTX_OK;
}

// ---- context-free helpers ---------------------------------------------------------------------
static int efail(char* err, int32_t n, const char* what, hipError_t e) {
    if (err && n > 0) snprintf(err, (size_t)n, "%s: %s", what, hipGetErrorString(e));
    return MRTX_E_DEVICE;
}
#define HIPCHK2(call)                                       \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return efail(err, err_len, #call, e_); \
    } while (0)

int mrtx_dem_from_ldem(int32_t device, const void* src, int32_t h, int32_t w, int32_t d, void* dst,
                       float* radius_scale, char* err, int32_t err_len) {
    if (!src || !dst || h < 1 || w < 1 || d < 1 || d > 64) return MRTX_E_INVALID;
    HIPCHK2(hipSetDevice(device));
    unsigned int* mb = nullptr;
    HIPCHK2(hipMalloc((void**)&mb, sizeof(unsigned int)));
    HIPCHK2(hipMemset(mb, 0, sizeof(unsigned int)));
    hipError_t e = mrtx_launch_ldem((const int16_t*)src, (float*)dst, h, w, d, mb, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned int bits = 0;
    if (e == hipSuccess) e = hipMemcpy(&bits, mb, sizeof bits, hipMemcpyDeviceToHost);
    (void)hipFree(mb);
    if (e != hipSuccess) return efail(err, err_len, "ldem kernels", e);
    if (radius_scale) std::memcpy(radius_scale, &bits, sizeof(float));
    return MRTX_OK;
}
int mrtx_synth_ldem(int32_t device, void* dst, int32_t h, int32_t w, uint32_t seed, char* err, int32_t err_len) {
    if (!dst || h < 1 || w < 1) return MRTX_E_INVALID;
    HIPCHK2(hipSetDevice(device));
    HIPCHK2(mrtx_launch_synth_ldem((int16_t*)dst, h, w, seed, nullptr));
    HIPCHK2(hipDeviceSynchronize());
    return MRTX_OK;
}
int mrtx_synth_color(int32_t device, void* dst, int32_t h, int32_t w, uint32_t seed, char* err, int32_t err_len) {
    if (!dst || h < 1 || w < 1) return MRTX_E_INVALID;
    HIPCHK2(hipSetDevice(device));
    HIPCHK2(mrtx_launch_synth_color((uint32_t*)dst, h, w, seed, nullptr));
    HIPCHK2(hipDeviceSynchronize());
    return MRTX_OK;
}
int mrtx_dev_alloc(int32_t device, uint64_t bytes, void** out) {
    if (!out || bytes == 0) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    return hipMalloc(out, (size_t)bytes) == hipSuccess ? MRTX_OK : MRTX_E_NOMEM;
}
int mrtx_dev_free(int32_t device, void* p) {
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    return hipFree(p) == hipSuccess ? MRTX_OK : MRTX_E_DEVICE;
}
int mrtx_dev_download(int32_t device, void* host_dst, const void* dev_src, uint64_t bytes) {
    if (!host_dst || !dev_src) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    return hipMemcpy(host_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToHost) == hipSuccess ? MRTX_OK : MRTX_E_DEVICE;
}
int mrtx_dev_upload(int32_t device, void* dev_dst, const void* host_src, uint64_t bytes) {
    if (!dev_dst || !host_src) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    return hipMemcpy(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice) == hipSuccess ? MRTX_OK : MRTX_E_DEVICE;
}
int mrtx_probe_stream(int32_t device, uint64_t bytes, int32_t repeats) {
    if (bytes < 1024 || repeats < 1) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    void* buf = nullptr; float* out = nullptr;
    int rc = MRTX_OK;
    if (hipMalloc(&buf, (size_t)bytes) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) rc = MRTX_E_NOMEM;
    if (rc == MRTX_OK && hipMemset(buf, 1, (size_t)bytes) != hipSuccess) rc = MRTX_E_DEVICE;
    for (int i = 0; i < repeats && rc == MRTX_OK; i++)
        if (mrtx_launch_probe_stream(buf, (int64_t)(bytes / 8), out, nullptr) != hipSuccess) rc = MRTX_E_DEVICE;
    if (rc == MRTX_OK && hipDeviceSynchronize() != hipSuccess) rc = MRTX_E_DEVICE;
    if (buf) (void)hipFree(buf);
    if (out) (void)hipFree(out);
    return rc;
}
int mrtx_probe_cr(int32_t device, int32_t which, uint32_t lo_bits, uint64_t n, uint64_t* mismatches, uint32_t* first_bad_bits) {
    if (which < 0 || which > 2 || n == 0 || !mismatches) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    unsigned long long* d = nullptr;
    unsigned long long h[2] = {0ull, ~0ull};
    int rc = MRTX_OK;
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return MRTX_E_NOMEM;
    if (hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess ||
        mrtx_launch_probe_cr(lo_bits, n, which, d, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
        rc = MRTX_E_DEVICE;
    (void)hipFree(d);
    *mismatches = h[0];
    if (first_bad_bits) *first_bad_bits = h[1] == ~0ull ? 0u : (uint32_t)(h[1] - 1ull);
    return rc;
}
int mrtx_probe_latlon(int32_t device, const float* a, const float* b, const float* c, float* lat, float* lon,
                      int32_t n) {
    if (!a || !b || !c || !lat || !lon || n < 1) return MRTX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MRTX_E_DEVICE;
    float* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const float* in[3] = {a, b, c};
    const size_t bytes = (size_t)n * sizeof(float);
    int rc = MRTX_OK;
    for (int i = 0; i < 5 && rc == MRTX_OK; i++)
        if (hipMalloc((void**)&d[i], bytes) != hipSuccess) rc = MRTX_E_NOMEM;
    for (int i = 0; i < 3 && rc == MRTX_OK; i++)
        if (hipMemcpy(d[i], in[i], bytes, hipMemcpyHostToDevice) != hipSuccess) rc = MRTX_E_DEVICE;
    if (rc == MRTX_OK && (mrtx_launch_probe_latlon(d[0], d[1], d[2], d[3], d[4], n, nullptr) != hipSuccess ||
                          hipMemcpy(lat, d[3], bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                          hipMemcpy(lon, d[4], bytes, hipMemcpyDeviceToHost) != hipSuccess))
        rc = MRTX_E_DEVICE;
    for (int i = 0; i < 5; i++)
        if (d[i]) (void)hipFree(d[i]);
    return rc;
}

}  // extern "C"
