/*
 * mrtx_oracle.c -- CPU oracle for the MoonRTX hot path (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library.  The shipped renderer (moonrtx_amd/, libmoonrt.so) never links, imports or calls it.
 *
 * PARITY UNPINNED against PlotOptiX: the reference delegates every pixel to the closed third-party
 * package plotoptix>=0.19.2 (requirements.txt:3; call sites moon_renderer.py:571-650, :854-871),
 * which is not in /root/reference, not installed and not fetchable, and the reference ships no
 * tests and no golden frames (the images/ jpgs are git-LFS pointers).  What IS pinned, by golden
 * vectors captured from the reference modules that import here (tests/golden/, made by
 * tests/golden/make_golden.py):
 *   - the DEM <-> sphere mapping and bilinear convention  (renderer_navigation.py:575-593)
 *   - the body frame / hit-buffer convention              (renderer_navigation.py:43-57, :452-492)
 * Everything else restates the scene the reference *describes* through its renderer calls:
 *   D1 pinhole camera, vertical fov            moon_renderer.py:627-635, renderer_navigation.py:330-332
 *   D2 displaced-sphere height-field march     moon_renderer.py:620-624, :586-588 (marching_step,
 *      marching_step_eps, scene_epsilon), surface inside the radius-R sphere (data_loader.py:240-242)
 *   D3 bilinear equirect DEM fetch             renderer_navigation.py:575-593
 *   D4 Lambert x RGBA8 colour texture          moon_renderer.py:613-617
 *   D5 spherical light, one shadow ray/sample  moon_renderer.py:640-641, :859-860, :65-71, :89-93
 *   D7 environment texel / black on a miss     moon_renderer.py:604-609
 *   D8 flat emissive Sun-disk sphere           moon_renderer.py:647-650
 *   D9 accumulation (running mean, linear)     moon_renderer.py:578
 *   D10 hit-position buffer                    moon_renderer.py:1138, renderer_navigation.py:195-203
 *
 * The arithmetic below is the SPEC: float32 with explicit fmaf, float64 for
 * the per-ray sphere entry, own polynomial atan/sin/cos (no libm transcendental on the per-sample
 * path), counter-based integer RNG.  Compile with -ffp-contract=off.  The HIP kernels follow the
 * same spec independently; tests require bit-exact agreement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct OrcScene {
    int32_t width, height;
    float scene_epsilon, marching_step, marching_step_eps;
    uint32_t spp_per_block, seed;
    uint32_t path_seg_min, path_seg_max;   /* set_uint("path_seg_range", 2, 4), moon_renderer.py:583; max 1 = direct light only */
    float const_albedo[3];
    double eye[3], target[3], up[3], vfov_deg;
    double center[3], radius, u[3], v[3];
    double light_pos[3], light_radius, light_radiance;
    double sun_pos[3], sun_radius, sun_radiance;
    const float* dem;
    int32_t dem_h, dem_w;
    const uint8_t* color;
    int32_t color_h, color_w;
    const uint8_t* bg;
    int32_t bg_h, bg_w;
    /* D11 overlay tubes (set_graph, renderer_labels.py:295-300): n_caps x 12 floats, scene coordinates:
     * ax ay az r  bx by bz 0  cr cg cb 0 -- flat colour, never shadow, invisible to shadow / continuation rays */
    const float* caps;
    int32_t n_caps;
} OrcScene;

/* indices of the stats array */
enum { ST_PRIMARY = 0, ST_HITS, ST_SHADOW, ST_HEIGHT, ST_COLOUR, ST_BG, ST_BOUNCE, ST_SUNHIT, ST_N };

/* ------------------------------------------------------------------ spec constants */
#define PI_D 3.14159265358979323846
static const float PI_F = 3.14159274101257324f;
static const float HALF_PI_F = 1.57079637050628662f;
static const float INV255 = 0.003921568859368563f;   /* (float)(1/255) */
/* atan(q) ~= q * P(q^2) on [0,1], |err| <= 1.3e-7 */
static const float AT0 = 0.9999993443489075f, AT1 = -0.33329859375953674f, AT2 = 0.19946560263633728f,
                   AT3 = -0.1390860229730606f, AT4 = 0.0964212492108345f, AT5 = -0.05591127648949623f,
                   AT6 = 0.02186218835413456f, AT7 = -0.004054343327879906f;
/* sin / cos on [0, pi/2] */
static const float SN0 = 1.0f, SN1 = -0.16666647791862488f, SN2 = 0.008332899771630764f,
                   SN3 = -0.00019800894369836897f, SN4 = 2.590481244624243e-06f;
static const float CS0 = 0.9999999403953552f, CS1 = -0.4999990463256836f, CS2 = 0.04166358336806297f,
                   CS3 = -0.001385370153002441f, CS4 = 2.3153859729063697e-05f;

static inline float atan_poly(float q) {
    float s = q * q;
    float p = AT7;
    p = fmaf(p, s, AT6);
    p = fmaf(p, s, AT5);
    p = fmaf(p, s, AT4);
    p = fmaf(p, s, AT3);
    p = fmaf(p, s, AT2);
    p = fmaf(p, s, AT1);
    p = fmaf(p, s, AT0);
    return p * q;
}

/* (a, b, c) -> lat = atan2(c, rho), lon = atan2(a, b), rho = sqrt(max(a^2+b^2, 1e-28)).
 * Both min/max ratios come from ONE reciprocal: t = 1/(m1*m2), q1 = n1*(t*m2), q2 = n2*(t*m1).
 * Octant reflection when the "y" magnitude is >= the "x" magnitude; signs copied from c and a. */
void orc_latlon(float a, float b, float c, float* lat, float* lon) {
    float rho2 = fmaf(b, b, a * a);
    rho2 = rho2 < 1.0e-28f ? 1.0e-28f : rho2;
    float rho = sqrtf(rho2);
    float aa = fabsf(a), ab = fabsf(b), ac = fabsf(c);
    float m1 = rho > ac ? rho : ac, n1 = rho > ac ? ac : rho;
    float m2 = ab > aa ? ab : aa, n2 = ab > aa ? aa : ab;
    float den = m1 * m2;
    den = den < 1.0e-37f ? 1.0e-37f : den;
    float t = 1.0f / den;
    float r1 = atan_poly(n1 * (t * m2));
    float r2 = atan_poly(n2 * (t * m1));
    if (ac >= rho) r1 = HALF_PI_F - r1;
    if (aa >= ab) r2 = HALF_PI_F - r2;
    if (b < 0.0f) r2 = PI_F - r2;
    *lat = copysignf(r1, c);
    *lon = copysignf(r2, a);
}

static inline void sincos_quadrant(float u, float* cs, float* sn) {
    /* angle = 2*pi*u, u in [0,1): quadrant + polynomial on [0, pi/2) */
    float t4 = u * 4.0f;
    float qf = floorf(t4);
    float a = (t4 - qf) * HALF_PI_F;
    float a2 = a * a;
    float sp = SN4;
    sp = fmaf(sp, a2, SN3);
    sp = fmaf(sp, a2, SN2);
    sp = fmaf(sp, a2, SN1);
    sp = fmaf(sp, a2, SN0);
    float s1 = sp * a;
    float cp = CS4;
    cp = fmaf(cp, a2, CS3);
    cp = fmaf(cp, a2, CS2);
    cp = fmaf(cp, a2, CS1);
    cp = fmaf(cp, a2, CS0);
    float c1 = cp;
    int qi = (int)qf;
    if (qi == 0) { *cs = c1; *sn = s1; }
    else if (qi == 1) { *cs = -s1; *sn = c1; }
    else if (qi == 2) { *cs = -c1; *sn = -s1; }
    else { *cs = s1; *sn = -c1; }
}

static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
static inline float u01(uint32_t key, uint32_t dim) {
    uint32_t r = mix32(key + (dim + 1u) * 0x9E3779B9u);
    return (float)(r >> 8) * 5.9604644775390625e-08f; /* 2^-24 */
}

/* ------------------------------------------------------------------ equirect grid */
typedef struct Grid {
    int32_t h, w;
    float row_scale, row_off, col_scale, col_off, wf;
} Grid;

static void grid_init(Grid* g, int32_t h, int32_t w) {
    g->h = h; g->w = w;
    g->row_scale = (float)(-(double)h / PI_D);
    g->row_off = (float)(0.5 * (double)h - 0.5);
    g->col_scale = (float)((double)w / (2.0 * PI_D));
    g->col_off = (float)(0.5 * (double)w - 0.5);
    g->wf = (float)w;
}

typedef struct Tap { int64_t i00, i01, i10, i11; float fr, fc; } Tap;

/* texel coordinates -> the four taps (renderer_navigation.py:581-588 restated on floor()):
 * r0 = floor(row), fr = row - r0, rows r0 and r0+1 each clamped to [0, h-1]  (row -1 == row 0, row h == row h-1)
 * c0 = floor(col), fc = col - c0, columns c0 and c0+1 wrapped into [0, w)      (col -1 == col w-1, col w == col 0) */
static inline int32_t wrapc(int32_t c, int32_t w) {
    if (c < 0) c += w;
    if (c >= w) c -= w;
    if (c < 0) c += w;
    if (c >= w) c -= w;
    return c;
}
static inline void grid_tap(const Grid* g, float rowf, float colf, Tap* t) {
    float rfl = floorf(rowf), cfl = floorf(colf);
    int32_t r0 = (int32_t)rfl, c0 = (int32_t)cfl;
    r0 = r0 < -1 ? -1 : (r0 > g->h - 1 ? g->h - 1 : r0);
    c0 = c0 < -2 ? -2 : (c0 > g->w ? g->w : c0);
    int32_t ra = r0 < 0 ? 0 : r0, rb = r0 + 1 > g->h - 1 ? g->h - 1 : r0 + 1;
    int32_t ca = wrapc(c0, g->w), cb = wrapc(c0 + 1, g->w);
    t->i00 = (int64_t)ra * g->w + ca; t->i01 = (int64_t)ra * g->w + cb;
    t->i10 = (int64_t)rb * g->w + ca; t->i11 = (int64_t)rb * g->w + cb;
    t->fr = rowf - rfl; t->fc = colf - cfl;
}
static inline void grid_rc(const Grid* g, float lat, float lon, float* rowf, float* colf) {
    *rowf = fmaf(lat, g->row_scale, g->row_off);
    *colf = fmaf(lon, g->col_scale, g->col_off);
}
static inline float lerp2(float e00, float e01, float e10, float e11, float fr, float fc) {
    float top = fmaf(fc, e01 - e00, e00);
    float bot = fmaf(fc, e11 - e10, e10);
    return fmaf(fr, bot - top, top);
}
static inline float dem_at(const float* dem, const Grid* g, float rowf, float colf) {
    Tap t;
    grid_tap(g, rowf, colf, &t);
    return lerp2(dem[t.i00], dem[t.i01], dem[t.i10], dem[t.i11], t.fr, t.fc);
}

/* bilinear DEM value at (lat, lon) in radians: the spec's D(lat, lon).  Exposed so tests can pin
 * it to NavigationMixin.get_elevation_m golden vectors. */
float orc_dem_bilinear(const float* dem, int32_t h, int32_t w, float lat, float lon) {
    Grid g; grid_init(&g, h, w);
    float rowf, colf; grid_rc(&g, lat, lon, &rowf, &colf);
    return dem_at(dem, &g, rowf, colf);
}

/* ------------------------------------------------------------------ derived frame constants */
typedef struct Frame {
    float Wd[3], Ux[3], Vy[3], two_over_w, two_over_h;
    double oc[3], cq, M[3][3];
    float Mf[3][3], centerf[3], eyef[3];
    float Rf, R2f;
    float Lb[3], rL2, rad2;
    int sun_on; float sc[3], sun_cq, sun_rad;
    float Sb[3], sun_r2;   /* Sun-disk centre in the moon frame (relative to the Moon centre), radius^2: continuation rays */
    float step, eps, inv_step; int nbis, kmax;
    float polar_rho2, row_hi, col_hi;
    float dlat_scale, dlon_scale;
    Grid gd, gc, gb;
    float bg_row_scale, bg_row_off, bg_col_scale, bg_col_off;
    uint32_t key0;
} Frame;

static void normalize3(double v[3]) {
    double l = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
}
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static void frame_init(Frame* f, const OrcScene* s) {
    int i, j;
    double wv[3], uv[3], vv[3];
    for (i = 0; i < 3; i++) wv[i] = s->target[i] - s->eye[i];
    normalize3(wv);
    cross3(wv, s->up, uv); normalize3(uv);
    cross3(uv, wv, vv);
    double th = tan(s->vfov_deg * PI_D / 360.0);
    double aspect = (double)s->width / (double)s->height;
    for (i = 0; i < 3; i++) {
        f->Wd[i] = (float)wv[i];
        f->Ux[i] = (float)(uv[i] * (th * aspect));
        f->Vy[i] = (float)(vv[i] * th);
        f->oc[i] = s->eye[i] - s->center[i];
        f->centerf[i] = (float)s->center[i];
        f->eyef[i] = (float)s->eye[i];
    }
    f->two_over_w = (float)(2.0 / (double)s->width);
    f->two_over_h = (float)(2.0 / (double)s->height);
    f->cq = ((f->oc[0] * f->oc[0] + f->oc[1] * f->oc[1]) + f->oc[2] * f->oc[2]) - s->radius * s->radius;
    /* moon frame: rows = (east 90, lon 0, north) */
    double ez[3], v0[3], ex[3];
    for (i = 0; i < 3; i++) ez[i] = s->u[i];
    normalize3(ez);
    double dp = (s->v[0] * ez[0] + s->v[1] * ez[1]) + s->v[2] * ez[2];
    for (i = 0; i < 3; i++) v0[i] = s->v[i] - dp * ez[i];
    normalize3(v0);
    cross3(ez, v0, ex);
    for (j = 0; j < 3; j++) { f->M[0][j] = ex[j]; f->M[1][j] = v0[j]; f->M[2][j] = ez[j]; }
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) f->Mf[i][j] = (float)f->M[i][j];
    f->Rf = (float)s->radius;
    f->R2f = f->Rf * f->Rf;
    double lr[3];
    for (i = 0; i < 3; i++) lr[i] = s->light_pos[i] - s->center[i];
    for (i = 0; i < 3; i++) f->Lb[i] = (float)((f->M[i][0] * lr[0] + f->M[i][1] * lr[1]) + f->M[i][2] * lr[2]);
    f->rL2 = (float)(s->light_radius * s->light_radius);
    f->rad2 = (float)(2.0 * s->light_radiance);
    f->sun_on = s->sun_radius > 0.0;
    double sr[3];
    for (i = 0; i < 3; i++) { sr[i] = s->sun_pos[i] - s->eye[i]; f->sc[i] = (float)sr[i]; }
    f->sun_cq = (float)(((sr[0] * sr[0] + sr[1] * sr[1]) + sr[2] * sr[2]) - s->sun_radius * s->sun_radius);
    f->sun_rad = (float)s->sun_radiance;
    {
        double sm[3];
        for (i = 0; i < 3; i++) sm[i] = s->sun_pos[i] - s->center[i];
        for (i = 0; i < 3; i++) f->Sb[i] = (float)((f->M[i][0] * sm[0] + f->M[i][1] * sm[1]) + f->M[i][2] * sm[2]);
        f->sun_r2 = (float)(s->sun_radius * s->sun_radius);
    }
    f->step = s->marching_step;
    f->eps = s->marching_step_eps;
    f->nbis = 0;
    { double wdt = (double)f->step; while (wdt > (double)f->eps && f->nbis < 24) { wdt *= 0.5; f->nbis++; } }
    f->kmax = (((int)(2.0 * s->radius / (double)f->step) + 8) + 15) & ~15;   /* multiple of SEG_N */
    f->inv_step = 1.0f / f->step;
    f->polar_rho2 = (float)(0.04 * s->radius * s->radius);
    f->row_hi = nextafterf((float)s->dem_h, 0.0f);
    f->col_hi = nextafterf((float)s->dem_w, 0.0f);
    grid_init(&f->gd, s->dem_h, s->dem_w);
    f->dlat_scale = (float)((double)s->dem_h / (2.0 * PI_D));
    f->dlon_scale = (float)((double)s->dem_w / (4.0 * PI_D));
    if (s->color) grid_init(&f->gc, s->color_h, s->color_w);
    if (s->bg) {
        f->bg_row_scale = (float)(-(double)s->bg_h / PI_D);
        f->bg_row_off = (float)(0.5 * (double)s->bg_h);
        f->bg_col_scale = (float)((double)s->bg_w / (2.0 * PI_D));
        f->bg_col_off = (float)(0.5 * (double)s->bg_w);
    }
    f->key0 = mix32(s->seed ^ 0x9E3779B9u);
}

/* exposes the float frame constants for host-logic parity tests (46 floats) */
void orc_frame_floats(const OrcScene* s, float* out) {
    Frame f; frame_init(&f, s);
    int k = 0, i, j;
    for (i = 0; i < 3; i++) out[k++] = f.Wd[i];
    for (i = 0; i < 3; i++) out[k++] = f.Ux[i];
    for (i = 0; i < 3; i++) out[k++] = f.Vy[i];
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) out[k++] = f.Mf[i][j];
    for (i = 0; i < 3; i++) out[k++] = f.Lb[i];
    out[k++] = f.rL2; out[k++] = f.rad2; out[k++] = f.R2f;
    for (i = 0; i < 3; i++) out[k++] = f.sc[i];
    out[k++] = f.sun_cq; out[k++] = (float)f.nbis; out[k++] = (float)f.kmax;
    out[k++] = f.gd.row_scale; out[k++] = f.gd.row_off; out[k++] = f.gd.col_scale; out[k++] = f.gd.col_off;
    out[k++] = f.dlat_scale; out[k++] = f.dlon_scale;
    out[k++] = (float)f.cq; out[k++] = f.two_over_w; out[k++] = f.two_over_h;
}

/* ------------------------------------------------------------------ the march */
/* Texel coordinates along a ray are smooth: per SEG_N-step segment the exact (row, col) is evaluated at the
 * segment's start, middle and end, and the steps in between use the quadratic through those three
 * (|error| <= ~1e-3 row / 7e-3 column texels for rho >= 0.2 R, i.e. at the float32 resolution of the
 * coordinate itself).  Segments that touch the polar cap (rho < 0.2 R) or straddle the +/-180 seam evaluate
 * every step exactly. */
#define SEG_N 16
static int orc_force_exact = 0;              /* orc_set_exact(1): every step evaluates (lat, lon) -> (row, col) exactly */
static uint64_t orc_quad_out_of_range = 0;   /* must stay 0; tests read it through orc_debug_counter() */
typedef struct Seg { float sa, ra, r1, r2, ca, c1, c2; int exact; } Seg;
typedef struct Ray { float oa, ob, oc, da, db, dc; } Ray;

static inline void exact_rowcol(const Frame* f, float pa, float pb, float pc, float* rowf, float* colf, float* rho2) {
    float lat, lon;
    *rho2 = fmaf(pb, pb, pa * pa);
    orc_latlon(pa, pb, pc, &lat, &lon);
    grid_rc(&f->gd, lat, lon, rowf, colf);
}

/* anchors of the segment starting at step ka; (rowA, colA, q2A) are the exact values at its start */
static void seg_setup(const Frame* f, const Ray* r, int ka, float rowA, float colA, float q2A, Seg* sg,
                      float* rowB, float* colB, float* q2B) {
    float sm = (float)(ka + SEG_N / 2) * f->step, sb = (float)(ka + SEG_N) * f->step;
    float rM, cM, q2M, rB, cB, qB;
    exact_rowcol(f, fmaf(sm, r->da, r->oa), fmaf(sm, r->db, r->ob), fmaf(sm, r->dc, r->oc), &rM, &cM, &q2M);
    exact_rowcol(f, fmaf(sb, r->da, r->oa), fmaf(sb, r->db, r->ob), fmaf(sb, r->dc, r->oc), &rB, &cB, &qB);
    float hw = 0.5f * f->gd.wf;
    float qmin = fminf(q2A, fminf(q2M, qB));
    sg->exact = orc_force_exact || (fabsf(cM - colA) > hw) || (fabsf(cB - colA) > hw) || (qmin < f->polar_rho2);
    sg->sa = (float)ka * f->step;
    sg->ra = rowA; sg->ca = colA;
    sg->r2 = (fmaf(-2.0f, rM, rowA) + rB) * 0.0078125f;               /* (A - 2M + B) / (N^2/2) */
    sg->r1 = fmaf(-16.0f, sg->r2, (rB - rowA) * 0.0625f);             /* (B - A)/N - c2 N        */
    sg->c2 = (fmaf(-2.0f, cM, colA) + cB) * 0.0078125f;
    sg->c1 = fmaf(-16.0f, sg->c2, (cB - colA) * 0.0625f);
    *rowB = rB; *colB = cB; *q2B = qB;
}

static inline int below_seg(const OrcScene* s, const Frame* f, const Seg* sg, float sk, float pa, float pb,
                            float pc, uint64_t* st) {
    float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
    float rowf, colf, q2;
    if (sg->exact) {
        exact_rowcol(f, pa, pb, pc, &rowf, &colf, &q2);
    } else {
        float u = (sk - sg->sa) * f->inv_step;
        rowf = fmaf(u, fmaf(u, sg->r2, sg->r1), sg->ra);
        colf = fmaf(u, fmaf(u, sg->c2, sg->c1), sg->ca);
        /* a non-seam, non-polar segment keeps the quadratic well inside [-1, h) x [-1, w): counted, never clamped */
        if (!(rowf >= -1.0f && rowf <= f->row_hi && colf >= -1.0f && colf <= f->col_hi)) {
#pragma omp atomic
            orc_quad_out_of_range++;
        }
    }
    float d = dem_at(s->dem, &f->gd, rowf, colf);
    st[ST_HEIGHT]++;
    float surf = f->Rf * d;
    return r2 <= surf * surf;
}

/* Coarse march s_k = k*step, k = 1, 2, ...: returns 1 and *k_hit at the first sample at/below the surface.
 * primary: stop when s_k > smax (left the bounding sphere);  shadow: stop when r^2 > R^2. */
static int march(const OrcScene* s, const Frame* f, const Ray* r, int primary, float smax, Seg* sg, int* k_hit,
                 uint64_t* st) {
    float rowA, colA, q2A;
    int ka = 0;
    exact_rowcol(f, r->oa, r->ob, r->oc, &rowA, &colA, &q2A);
    for (;;) {
        float rowB, colB, q2B;
        int j;
        seg_setup(f, r, ka, rowA, colA, q2A, sg, &rowB, &colB, &q2B);
        for (j = 1; j <= SEG_N; j++) {
            int k = ka + j;
            float sk = (float)k * f->step;
            float pa = fmaf(sk, r->da, r->oa), pb = fmaf(sk, r->db, r->ob), pc = fmaf(sk, r->dc, r->oc);
            int in;
            if (primary) in = (sk <= smax) && (k <= f->kmax);
            else in = (fmaf(pc, pc, fmaf(pb, pb, pa * pa)) <= f->R2f) && (k <= f->kmax);
            if (!in) return 0;
            if (below_seg(s, f, sg, sk, pa, pb, pc, st)) { *k_hit = k; return 1; }
        }
        ka += SEG_N; rowA = rowB; colA = colB; q2A = q2B;
    }
}

typedef struct Sample { float c[3]; float hitflag; float hit[4]; } Sample;
typedef struct Vertex { float p[3], n[3], alb[3]; } Vertex;

/* Duff et al., "Building an Orthonormal Basis, Revisited" */
static inline void duff_basis(const float n[3], float b1[3], float b2[3]) {
    float sg = n[2] >= 0.0f ? 1.0f : -1.0f;
    float aa = -1.0f / (sg + n[2]);
    float bb = (n[0] * n[1]) * aa;
    b1[0] = fmaf(sg, (n[0] * n[0]) * aa, 1.0f); b1[1] = sg * bb; b1[2] = -sg * n[0];
    b2[0] = bb; b2[1] = fmaf(n[1] * n[1], aa, sg); b2[2] = -n[1];
}

/* D7: nearest environment texel along a scene-frame direction */
static inline void env_lookup(const OrcScene* s, const Frame* f, float dx, float dy, float dz, float out[3], uint64_t* st) {
    float el, az;
    orc_latlon(dx, dy, dz, &el, &az);
    float rowf = fmaf(el, f->bg_row_scale, f->bg_row_off);
    float colf = fmaf(az, f->bg_col_scale, f->bg_col_off);
    int r = (int)floorf(rowf), c = (int)floorf(colf);
    r = r < 0 ? 0 : (r > s->bg_h - 1 ? s->bg_h - 1 : r);
    if (c >= s->bg_w) c -= s->bg_w;
    if (c < 0) c = 0;
    const uint8_t* px = s->bg + 4 * ((int64_t)r * s->bg_w + c);
    out[0] = (float)px[0] * INV255;
    out[1] = (float)px[1] * INV255;
    out[2] = (float)px[2] * INV255;
    st[ST_BG]++;
}

/* surface point -> normal (central differences of D one texel either side) and albedo (D4) */
static void hit_vertex(const OrcScene* s, const Frame* f, float ha, float hb, float hc, Vertex* v, uint64_t* st) {
    float rho2 = fmaf(hb, hb, ha * ha);
    float r2 = fmaf(hc, hc, rho2);
    float rho = sqrtf(rho2);
    float r = sqrtf(r2);
    float lat, lon;
    orc_latlon(ha, hb, hc, &lat, &lon);
    float rowf, colf;
    grid_rc(&f->gd, lat, lon, &rowf, &colf);
    float dn = dem_at(s->dem, &f->gd, rowf - 1.0f, colf);
    float ds = dem_at(s->dem, &f->gd, rowf + 1.0f, colf);
    float de = dem_at(s->dem, &f->gd, rowf, colf + 1.0f);
    float dw = dem_at(s->dem, &f->gd, rowf, colf - 1.0f);
    st[ST_HEIGHT] += 4;
    float dlat = (dn - ds) * f->dlat_scale;
    float dlon = (de - dw) * f->dlon_scale;
    float rhoc = rho > 1.0e-6f ? rho : 1.0e-6f;
    float inv_r = 1.0f / r, inv_rho = 1.0f / rhoc;
    float sphi = hc * inv_r, cphi = rhoc * inv_r;
    float slam = ha * inv_rho, clam = hb * inv_rho;
    float glat = (f->Rf * inv_r) * dlat;
    float glon = (f->Rf * inv_rho) * dlon;
    float na = fmaf(-glon, clam, fmaf(glat, sphi * slam, ha * inv_r));
    float nb = fmaf(glon, slam, fmaf(glat, sphi * clam, hb * inv_r));
    float nc = fmaf(-glat, cphi, hc * inv_r);
    float inv_nl = 1.0f / sqrtf(fmaf(nc, nc, fmaf(nb, nb, na * na)));
    v->p[0] = ha; v->p[1] = hb; v->p[2] = hc;
    v->n[0] = na * inv_nl; v->n[1] = nb * inv_nl; v->n[2] = nc * inv_nl;
    if (s->color) {
        float rc, cc; Tap t; int ch;
        grid_rc(&f->gc, lat, lon, &rc, &cc);
        grid_tap(&f->gc, rc, cc, &t);
        for (ch = 0; ch < 3; ch++) {
            float val = lerp2((float)s->color[4 * t.i00 + ch], (float)s->color[4 * t.i01 + ch],
                              (float)s->color[4 * t.i10 + ch], (float)s->color[4 * t.i11 + ch], t.fr, t.fc);
            v->alb[ch] = val * INV255;
        }
        st[ST_COLOUR]++;
    } else {
        v->alb[0] = s->const_albedo[0]; v->alb[1] = s->const_albedo[1]; v->alb[2] = s->const_albedo[2];
    }
}

static int march(const OrcScene* s, const Frame* f, const Ray* r, int primary, float smax, Seg* sg, int* k_hit,
                 uint64_t* st);

/* D5: one sample of the spherical light from a vertex; returns radiance * solid angle / pi * cos * visibility */
static float direct_light(const OrcScene* s, const Frame* f, const Vertex* v, float u2, float u3, uint64_t* st) {
    float eps = s->scene_epsilon;
    float oa = fmaf(eps, v->n[0], v->p[0]), ob = fmaf(eps, v->n[1], v->p[1]), occ = fmaf(eps, v->n[2], v->p[2]);
    float ta = f->Lb[0] - oa, tb = f->Lb[1] - ob, tc = f->Lb[2] - occ;
    float d2 = fmaf(tc, tc, fmaf(tb, tb, ta * ta));
    float inv_dist = 1.0f / sqrtf(d2);
    float l[3] = { ta * inv_dist, tb * inv_dist, tc * inv_dist };
    float sin2 = f->rL2 * (inv_dist * inv_dist);
    if (sin2 > 1.0f) sin2 = 1.0f;
    float cosmax = sqrtf(1.0f - sin2);
    float omc = sin2 / (1.0f + cosmax);
    float av = u2 * omc;
    float cost = 1.0f - av;
    float sint = sqrtf(av * (2.0f - av));
    float cph, sph;
    sincos_quadrant(u3, &cph, &sph);
    float b1[3], b2[3];
    duff_basis(l, b1, b2);
    float ca = sint * cph, sa = sint * sph;
    float wa = fmaf(cost, l[0], fmaf(sa, b2[0], ca * b1[0]));
    float wb = fmaf(cost, l[1], fmaf(sa, b2[1], ca * b1[1]));
    float wc = fmaf(cost, l[2], fmaf(sa, b2[2], ca * b1[2]));
    float cosi = fmaf(v->n[2], wc, fmaf(v->n[1], wb, v->n[0] * wa));
    if (!(cosi > 0.0f)) return 0.0f;
    st[ST_SHADOW]++;
    Ray sray = { oa, ob, occ, wa, wb, wc };
    Seg ssg;
    int kk = 0;
    if (march(s, f, &sray, 0, 0.0f, &ssg, &kk, st)) return 0.0f;
    return (f->rad2 * omc) * cosi;
}

/* Nearest overlay capsule along the primary ray.  ro = ray point closest to the Moon centre (relative to the
 * centre), d = unit direction; per capsule the origin is re-centred once more at the capsule's first endpoint
 * (ta = (a - ro).d, oa = (ro - a) + ta d) so that the quadratic stays well-conditioned for radii ~1e-2 at a
 * 300-unit eye distance.  Returns the parameter relative to `ro` (or -1e30) and the index of the capsule. */
static float nearest_capsule(const OrcScene* s, const Frame* f, const float ro[3], float dx, float dy, float dz, float smin, int* which) {
    float best = 1.0e30f;
    int k;
    *which = -1;
    for (k = 0; k < s->n_caps; k++) {
        const float* c = s->caps + 12 * k;
        float a0 = (float)((double)c[0] - s->center[0]), a1 = (float)((double)c[1] - s->center[1]), a2 = (float)((double)c[2] - s->center[2]);
        float b0 = (float)((double)c[4] - s->center[0]), b1 = (float)((double)c[5] - s->center[1]), b2 = (float)((double)c[6] - s->center[2]);
        float r = c[3];
        float e0 = a0 - ro[0], e1 = a1 - ro[1], e2 = a2 - ro[2];
        float ta = fmaf(e2, dz, fmaf(e1, dy, e0 * dx));
        float o0 = fmaf(ta, dx, -e0), o1 = fmaf(ta, dy, -e1), o2 = fmaf(ta, dz, -e2);
        float ba0 = b0 - a0, ba1 = b1 - a1, ba2 = b2 - a2;
        float baba = fmaf(ba2, ba2, fmaf(ba1, ba1, ba0 * ba0));
        float bard = fmaf(ba2, dz, fmaf(ba1, dy, ba0 * dx));
        float baoa = fmaf(ba2, o2, fmaf(ba1, o1, ba0 * o0));
        float rdoa = fmaf(dz, o2, fmaf(dy, o1, dx * o0));
        float oaoa = fmaf(o2, o2, fmaf(o1, o1, o0 * o0));
        float r2 = r * r;
        float A = fmaf(-bard, bard, baba);
        float B = fmaf(baba, rdoa, -(baoa * bard));
        float C = fmaf(baba, oaoa, -(baoa * baoa)) - r2 * baba;
        float h = fmaf(B, B, -(A * C));
        float cand = -1.0f;
        int have = 0;
        float y = baoa;
        if (A > 0.0f && h >= 0.0f) {
            float t = (-B - sqrtf(h)) / A;
            y = fmaf(t, bard, baoa);
            if (y > 0.0f && y < baba) { cand = t; have = 1; }
        }
        if (!have) {   /* rounded end nearest to where the axis test left the segment */
            float q0 = o0, q1 = o1, q2 = o2;
            if (!(y <= 0.0f)) { q0 = o0 - ba0; q1 = o1 - ba1; q2 = o2 - ba2; }
            float B2 = fmaf(dz, q2, fmaf(dy, q1, dx * q0));
            float C2 = fmaf(q2, q2, fmaf(q1, q1, q0 * q0)) - r2;
            float h2 = fmaf(B2, B2, -C2);
            if (h2 > 0.0f) { cand = -B2 - sqrtf(h2); have = 1; }
        }
        if (have) {
            float sc = ta + cand;
            if (sc > smin && sc < best) { best = sc; *which = k; }   /* in front of the eye only */
        }
    }
    (void)f;
    return *which >= 0 ? best : -1.0e30f;
}

static void trace_sample(const OrcScene* s, const Frame* f, int x, int y, uint32_t gs, Sample* o,
                         uint64_t* st) {
    uint32_t pix = (uint32_t)y * (uint32_t)s->width + (uint32_t)x;
    uint32_t kp = mix32(pix + f->key0);
    uint32_t ks = mix32(kp ^ (gs * 0x85EBCA6Bu + 1u));
    float u0 = u01(ks, 0), u1 = u01(ks, 1), u2 = u01(ks, 2), u3 = u01(ks, 3);
    o->c[0] = o->c[1] = o->c[2] = 0.0f; o->hitflag = 0.0f;
    o->hit[0] = o->hit[1] = o->hit[2] = o->hit[3] = 0.0f;
    st[ST_PRIMARY]++;

    float fx = (float)x + u0, fy = (float)y + u1;
    float sx = fmaf(fx, f->two_over_w, -1.0f);
    float sy = fmaf(-fy, f->two_over_h, 1.0f);
    float dx = fmaf(sy, f->Vy[0], fmaf(sx, f->Ux[0], f->Wd[0]));
    float dy = fmaf(sy, f->Vy[1], fmaf(sx, f->Ux[1], f->Wd[1]));
    float dz = fmaf(sy, f->Vy[2], fmaf(sx, f->Ux[2], f->Wd[2]));
    float inv_len = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    dx = dx * inv_len; dy = dy * inv_len; dz = dz * inv_len;

    /* float64 entry into the bounding sphere */
    double Dx = (double)dx, Dy = (double)dy, Dz = (double)dz;
    double a = (Dx * Dx + Dy * Dy) + Dz * Dz;
    double b = (f->oc[0] * Dx + f->oc[1] * Dy) + f->oc[2] * Dz;
    double disc = b * b - a * f->cq;
    int on_sphere = 0;
    double t0 = 0.0, t1 = 0.0;
    if (disc > 0.0) {
        /* products and sums in float64 (the eye is 30 radii away); the root and the 1/a scale in float32:
         * their rounding moves the entry point ALONG the ray only (<= 2e-5 units), which the march absorbs */
        double sq = (double)sqrtf((float)disc);
        double inva = (double)(1.0f / (float)a);
        t0 = (-b - sq) * inva;
        t1 = (-b + sq) * inva;
        if (t1 > 0.0) { on_sphere = 1; if (t0 < 0.0) t0 = 0.0; }
    }
    /* D11: overlay tubes live outside the bounding sphere (r = 1.025 R, moon_grid.py:188): one wins the sample if
     * it is hit in front of the sphere entry, or if the ray finds no terrain at all */
    int cap = -1;
    float cap_s = 0.0f, cro[3] = { 0.0f, 0.0f, 0.0f };
    double tc0 = 0.0;
    if (s->n_caps > 0) {
        tc0 = -b * (double)(1.0f / (float)a);   /* closest approach to the Moon centre (float32 reciprocal: a ~ 1) */
        cro[0] = (float)(f->oc[0] + tc0 * Dx); cro[1] = (float)(f->oc[1] + tc0 * Dy); cro[2] = (float)(f->oc[2] + tc0 * Dz);
        cap_s = nearest_capsule(s, f, cro, dx, dy, dz, (float)(-tc0), &cap);   /* the nearest one in front of the eye */
    }
    int cap_front = cap >= 0 && (!on_sphere || cap_s < (float)(t0 - tc0));
    int hit = 0;
    float pa = 0, pb = 0, pc = 0, da = 0, db = 0, dc = 0, lo = 0.0f;
    if (on_sphere && !cap_front) {
        double pe0 = f->oc[0] + t0 * Dx, pe1 = f->oc[1] + t0 * Dy, pe2 = f->oc[2] + t0 * Dz;
        pa = (float)((f->M[0][0] * pe0 + f->M[0][1] * pe1) + f->M[0][2] * pe2);
        pb = (float)((f->M[1][0] * pe0 + f->M[1][1] * pe1) + f->M[1][2] * pe2);
        pc = (float)((f->M[2][0] * pe0 + f->M[2][1] * pe1) + f->M[2][2] * pe2);
        da = (float)((f->M[0][0] * Dx + f->M[0][1] * Dy) + f->M[0][2] * Dz);
        db = (float)((f->M[1][0] * Dx + f->M[1][1] * Dy) + f->M[1][2] * Dz);
        dc = (float)((f->M[2][0] * Dx + f->M[2][1] * Dy) + f->M[2][2] * Dz);
        float smax = (float)(t1 - t0);
        Ray ray = { pa, pb, pc, da, db, dc };
        Seg sg;
        int k = 0;
        hit = march(s, f, &ray, 1, smax, &sg, &k, st);
        if (hit) {
            int i;
            float hi = (float)k * f->step;
            lo = (float)(k - 1) * f->step;
            for (i = 0; i < f->nbis; i++) {
                float mid = 0.5f * (lo + hi);
                if (below_seg(s, f, &sg, mid, fmaf(mid, da, pa), fmaf(mid, db, pb), fmaf(mid, dc, pc), st)) hi = mid;
                else lo = mid;
            }
        }
    }

    if (!hit && cap >= 0) {
        const float* c = s->caps + 12 * cap;
        o->c[0] = c[8]; o->c[1] = c[9]; o->c[2] = c[10];
        o->hitflag = 1.0f;
        o->hit[0] = f->centerf[0] + fmaf(cap_s, dx, cro[0]);
        o->hit[1] = f->centerf[1] + fmaf(cap_s, dy, cro[1]);
        o->hit[2] = f->centerf[2] + fmaf(cap_s, dz, cro[2]);
        o->hit[3] = (float)tc0 + cap_s;
        return;
    }
    if (!hit) {
        /* D8 Sun disk, then D7 environment */
        if (f->sun_on) {
            float bq = fmaf(f->sc[2], dz, fmaf(f->sc[1], dy, f->sc[0] * dx));
            float dq = fmaf(bq, bq, -f->sun_cq);
            if (bq > 0.0f && dq > 0.0f) {
                float t = bq - sqrtf(dq);
                o->c[0] = o->c[1] = o->c[2] = f->sun_rad;
                o->hitflag = 1.0f;
                o->hit[0] = fmaf(t, dx, f->eyef[0]);
                o->hit[1] = fmaf(t, dy, f->eyef[1]);
                o->hit[2] = fmaf(t, dz, f->eyef[2]);
                o->hit[3] = t;
                return;
            }
        }
        if (s->bg) env_lookup(s, f, dx, dy, dz, o->c, st);
        return;
    }

    /* ---- the path: vertex 1 is the primary hit; D6 continues it (set_uint("path_seg_range", min, max)) */
    st[ST_HITS]++;
    Vertex v;
    hit_vertex(s, f, fmaf(lo, da, pa), fmaf(lo, db, pb), fmaf(lo, dc, pc), &v, st);
    o->hitflag = 1.0f;
    o->hit[0] = f->centerf[0] + fmaf(v.p[2], f->Mf[2][0], fmaf(v.p[1], f->Mf[1][0], v.p[0] * f->Mf[0][0]));
    o->hit[1] = f->centerf[1] + fmaf(v.p[2], f->Mf[2][1], fmaf(v.p[1], f->Mf[1][1], v.p[0] * f->Mf[0][1]));
    o->hit[2] = f->centerf[2] + fmaf(v.p[2], f->Mf[2][2], fmaf(v.p[1], f->Mf[1][2], v.p[0] * f->Mf[0][2]));
    o->hit[3] = (float)t0 + lo;

    float thr[3] = { 1.0f, 1.0f, 1.0f };
    uint32_t seg = 1, max_seg = s->path_seg_max < 1 ? 1 : s->path_seg_max;
    float ul1 = u2, ul2 = u3;
    for (;;) {
        float wgt = direct_light(s, f, &v, ul1, ul2, st);
        o->c[0] = fmaf(thr[0] * v.alb[0], wgt, o->c[0]);
        o->c[1] = fmaf(thr[1] * v.alb[1], wgt, o->c[1]);
        o->c[2] = fmaf(thr[2] * v.alb[2], wgt, o->c[2]);
        if (seg >= max_seg) break;
        /* continue with segment seg+1: cosine-weighted direction about the normal => throughput *= albedo */
        uint32_t d0 = 4u + 5u * (seg - 1u);
        thr[0] *= v.alb[0]; thr[1] *= v.alb[1]; thr[2] *= v.alb[2];
        if (seg + 1u > s->path_seg_min) {          /* Russian roulette beyond the guaranteed segments */
            float pc_ = v.alb[0] > v.alb[1] ? v.alb[0] : v.alb[1];
            pc_ = pc_ > v.alb[2] ? pc_ : v.alb[2];
            pc_ = pc_ > 1.0f ? 1.0f : pc_;
            if (!(u01(ks, d0) < pc_)) break;
            float ip = 1.0f / pc_;
            thr[0] *= ip; thr[1] *= ip; thr[2] *= ip;
        }
        float uh1 = u01(ks, d0 + 1u), uh2 = u01(ks, d0 + 2u);
        ul1 = u01(ks, d0 + 3u); ul2 = u01(ks, d0 + 4u);
        float rr = sqrtf(uh1), zz = sqrtf(1.0f - uh1), cph, sph;
        sincos_quadrant(uh2, &cph, &sph);
        float b1[3], b2[3];
        duff_basis(v.n, b1, b2);
        float xx = rr * cph, yy = rr * sph;
        Ray br;
        float eps = s->scene_epsilon;
        br.oa = fmaf(eps, v.n[0], v.p[0]); br.ob = fmaf(eps, v.n[1], v.p[1]); br.oc = fmaf(eps, v.n[2], v.p[2]);
        br.da = fmaf(zz, v.n[0], fmaf(yy, b2[0], xx * b1[0]));
        br.db = fmaf(zz, v.n[1], fmaf(yy, b2[1], xx * b1[1]));
        br.dc = fmaf(zz, v.n[2], fmaf(yy, b2[2], xx * b1[2]));
        st[ST_BOUNCE]++;
        Seg bsg;
        int bk = 0;
        if (!march(s, f, &br, 0, 0.0f, &bsg, &bk, st)) {
            /* The path leaves the Moon.  The flat Sun-disk sphere IS visible to continuation rays (the reference keeps
             * the light it bounces onto the Moon small by its radiance 2.0 and by parking it, moon_renderer.py:109-111,
             * :757-760): distance of the disk centre from the ray, in the moon frame, float32. */
            if (f->sun_on) {
                float sa = f->Sb[0] - br.oa, sb = f->Sb[1] - br.ob, sc = f->Sb[2] - br.oc;
                float bq = fmaf(sc, br.dc, fmaf(sb, br.db, sa * br.da));
                float pa_ = fmaf(-bq, br.da, sa), pb_ = fmaf(-bq, br.db, sb), pc_ = fmaf(-bq, br.dc, sc);
                float d2 = fmaf(pc_, pc_, fmaf(pb_, pb_, pa_ * pa_));
                if (bq > 0.0f && d2 < f->sun_r2) {
                    o->c[0] = fmaf(thr[0], f->sun_rad, o->c[0]);
                    o->c[1] = fmaf(thr[1], f->sun_rad, o->c[1]);
                    o->c[2] = fmaf(thr[2], f->sun_rad, o->c[2]);
                    st[ST_SUNHIT]++;
                    break;
                }
            }
            if (s->bg) {   /* environment radiance along its direction (scene frame) */
                float ex = fmaf(br.dc, f->Mf[2][0], fmaf(br.db, f->Mf[1][0], br.da * f->Mf[0][0]));
                float ey = fmaf(br.dc, f->Mf[2][1], fmaf(br.db, f->Mf[1][1], br.da * f->Mf[0][1]));
                float ez = fmaf(br.dc, f->Mf[2][2], fmaf(br.db, f->Mf[1][2], br.da * f->Mf[0][2]));
                float env[3];
                env_lookup(s, f, ex, ey, ez, env, st);
                o->c[0] = fmaf(thr[0], env[0], o->c[0]);
                o->c[1] = fmaf(thr[1], env[1], o->c[1]);
                o->c[2] = fmaf(thr[2], env[2], o->c[2]);
            }
            break;
        }
        float bhi = (float)bk * f->step, blo = (float)(bk - 1) * f->step;
        int i;
        for (i = 0; i < f->nbis; i++) {
            float mid = 0.5f * (blo + bhi);
            if (below_seg(s, f, &bsg, mid, fmaf(mid, br.da, br.oa), fmaf(mid, br.db, br.ob), fmaf(mid, br.dc, br.oc), st)) bhi = mid;
            else blo = mid;
        }
        hit_vertex(s, f, fmaf(blo, br.da, br.oa), fmaf(blo, br.db, br.ob), fmaf(blo, br.dc, br.oc), &v, st);
        seg++;
    }
}

/* Render n_blocks accumulation blocks of spp_per_block samples for pixels [x0,x1) x [y0,y1).
 * accum (W*H*4, in/out): running sums (r, g, b, hit-coverage).  hits (W*H*4, out): sample 0 of the
 * last block.  stats: ST_N counters, added to.  first_block = number of blocks already accumulated. */
int orc_render(const OrcScene* s, int x0, int y0, int x1, int y1, uint32_t first_block, uint32_t n_blocks,
               float* accum, float* hits, uint64_t* stats) {
    Frame f;
    uint32_t S = s->spp_per_block;
    if (S == 0 || S > 64 || (S & (S - 1)) != 0) return -1;
    if (!s->dem || s->dem_h < 2 || s->dem_w < 2) return -1;
    frame_init(&f, s);
    int y;
    uint64_t tot[ST_N];
    memset(tot, 0, sizeof tot);
#pragma omp parallel
    {
        uint64_t st[ST_N];
        memset(st, 0, sizeof st);
#pragma omp for schedule(dynamic, 1)
        for (y = y0; y < y1; y++) {
            int x;
            for (x = x0; x < x1; x++) {
                float* acc = accum + 4 * ((int64_t)y * s->width + x);
                float sum[4] = { acc[0], acc[1], acc[2], acc[3] };
                uint32_t blk;
                for (blk = 0; blk < n_blocks; blk++) {
                    float v[64][4];
                    uint32_t i, stride;
                    for (i = 0; i < S; i++) {
                        Sample smp;
                        trace_sample(s, &f, x, y, (first_block + blk) * S + i, &smp, st);
                        v[i][0] = smp.c[0]; v[i][1] = smp.c[1]; v[i][2] = smp.c[2]; v[i][3] = smp.hitflag;
                        if (i == 0 && blk == n_blocks - 1) {
                            float* hp = hits + 4 * ((int64_t)y * s->width + x);
                            hp[0] = smp.hit[0]; hp[1] = smp.hit[1]; hp[2] = smp.hit[2]; hp[3] = smp.hit[3];
                        }
                    }
                    /* pairwise tree: the order an xor-butterfly over the lanes produces */
                    for (stride = 1; stride < S; stride *= 2)
                        for (i = 0; i + stride < S; i += 2 * stride) {
                            v[i][0] += v[i + stride][0]; v[i][1] += v[i + stride][1];
                            v[i][2] += v[i + stride][2]; v[i][3] += v[i + stride][3];
                        }
                    sum[0] += v[0][0]; sum[1] += v[0][1]; sum[2] += v[0][2]; sum[3] += v[0][3];
                }
                acc[0] = sum[0]; acc[1] = sum[1]; acc[2] = sum[2]; acc[3] = sum[3];
            }
        }
        int i;
#pragma omp critical
        for (i = 0; i < ST_N; i++) tot[i] += st[i];
    }
    if (stats) { int i; for (i = 0; i < ST_N; i++) stats[i] += tot[i]; }
    return 0;
}

/* linear = accum / n  (the resolve step of the spec) */
void orc_resolve_linear(const float* accum, int64_t npix, uint32_t n_samples, float* out) {
    int64_t i;
    float n = (float)n_samples;
    for (i = 0; i < 4 * npix; i++) out[i] = n_samples ? accum[i] / n : 0.0f;
}

/* "Gamma" post-process (add_postproc("Gamma") with tonemap_exposure / tonemap_gamma, moon_renderer.py:598-600) -> the
 * 8-bit image the GUI shows and save_image writes (renderer_dialogs.py:1222-1224), or 16 bits per sample.
 * Spec: x = exposure * (sum / n) in float32; level = the number of thresholds
 * T[j] = (float)pow((j - 0.5) / N, gamma), j = 1..N (float64 pow, rounded once), that are <= x -- i.e.
 * round(N * x^(1/gamma)) clamped to [0, N], decided by comparisons only.  N = 255 or 65535. */
void orc_tone_table(double gamma, int32_t n, float* T) {
    int32_t j;
    T[0] = 0.0f;
    for (j = 1; j <= n; j++) T[j] = (float)pow(((double)j - 0.5) / (double)n, gamma);
}
static uint32_t tone_level(float x, const float* T, int32_t n) {
    /* upper bound: first j in [1, n+1) with x < T[j]; level = j - 1.  NaN compares false -> level 0. */
    int32_t lo = 1, hi = n + 1;
    if (!(x >= T[1])) return 0;
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (x >= T[mid]) lo = mid + 1; else hi = mid;
    }
    return (uint32_t)(lo - 1);
}
/* D12 compositing (renderer_video.py:15-27, :137-144): out = round((src*(255-a) + ov*a) / 255) per channel */
static uint32_t blend8(uint32_t src, uint32_t ov, uint32_t a) { return (src * (255u - a) + ov * a + 127u) / 255u; }
void orc_resolve_rgba8(const float* accum, int64_t npix, uint32_t n_samples, float exposure, float gamma,
                       const uint8_t* overlay, uint8_t* out) {
    float T[256];
    int64_t i; int k;
    const float n = (float)n_samples;
    orc_tone_table((double)gamma, 255, T);
    for (i = 0; i < npix; i++) {
        for (k = 0; k < 3; k++) {
            uint32_t v = n_samples ? tone_level(exposure * (accum[4 * i + k] / n), T, 255) : 0u;
            if (overlay) v = blend8(v, overlay[4 * i + k], overlay[4 * i + 3]);
            out[4 * i + k] = (uint8_t)v;
        }
        out[4 * i + 3] = 255;
    }
}
void orc_resolve_rgb16(const float* accum, int64_t npix, uint32_t n_samples, float exposure, float gamma, uint16_t* out) {
    float* T = (float*)malloc(65536 * sizeof(float));
    int64_t i; int k;
    const float n = (float)n_samples;
    orc_tone_table((double)gamma, 65535, T);
    for (i = 0; i < npix; i++)
        for (k = 0; k < 3; k++)
            out[3 * i + k] = (uint16_t)(n_samples ? tone_level(exposure * (accum[4 * i + k] / n), T, 65535) : 0u);
    free(T);
}

/* data_loader.py:166-247 restated: int16 LDEM (h*d, w*d) -> float32 (h, w) displacement factors.
 * Stage 1 (axis 4): exact integer sum of d int16 as float32, / d.  Stage 2 (axis 2): sequential
 * float32 sum over the d rows, / d.  Then * (0.5/1737400), + 1, / max. */
float orc_dem_from_ldem(const int16_t* src, int32_t h, int32_t w, int32_t d, float* dst) {
    const float scale = (float)(0.5 / 1737400.0);
    int64_t W = (int64_t)w * d;
    float mx = -INFINITY;
    int32_t r, c, i, j;
    for (r = 0; r < h; r++)
        for (c = 0; c < w; c++) {
            float v;
            if (d == 1) {
                v = (float)src[(int64_t)r * W + c] * scale;
            } else {
                float acc2 = 0.0f;
                for (i = 0; i < d; i++) {
                    const int16_t* p = src + ((int64_t)r * d + i) * W + (int64_t)c * d;
                    float acc = 0.0f;
                    for (j = 0; j < d; j++) acc += (float)p[j];
                    acc = acc / (float)d;
                    acc2 = (i == 0) ? acc : acc2 + acc;
                }
                v = (acc2 / (float)d) * scale;
            }
            v += 1.0f;
            dst[(int64_t)r * w + c] = v;
            if (v > mx) mx = v;
        }
    for (r = 0; r < h; r++)
        for (c = 0; c < w; c++) dst[(int64_t)r * w + c] /= mx;
    return mx;
}

/* Test switch: 1 = drop the per-segment quadratic of the texel coordinates and evaluate every march step and bisection
 * point exactly (what the spec approximates); used to MEASURE the approximation's effect on radiance.  Returns the old value. */
int orc_set_exact(int on) { int old = orc_force_exact; orc_force_exact = on ? 1 : 0; return old; }

int orc_sizeof_scene(void) { return (int)sizeof(OrcScene); }
uint64_t orc_debug_counter(void) { return orc_quad_out_of_range; }

/* number of OpenMP threads orc_render uses from now on; returns the count in effect */
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
