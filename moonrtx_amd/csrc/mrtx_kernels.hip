// mrtx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libmoonrt.so.
//
// What runs here replaces the device programs PlotOptiX launches for MoonRTX's one scene (SURVEY.md
// section 2.1, D1-D10): pinhole ray generation, the height-field march of the displaced sphere, the
// bilinear equirect DEM / colour fetches, Lambert shading against one sample of the spherical Sun
// light with a marched shadow ray, Sun-disk / environment on a miss, accumulation and the hit buffer.
//
// MI355X mapping:
//   * a 64-lane wavefront IS the 64 samples of one pixel (spp_per_launch = 64): the 64 rays differ
//     only by sub-pixel jitter, so they walk the same DEM cache lines and leave the march loops on
//     nearly the same step -- the wave-wide `while (any lane active)` the compiler emits for the
//     march (s_cbranch_execnz on the ballot of live lanes) wastes almost nothing, and the sample
//     mean is an in-register xor-butterfly, not a read-modify-write on HBM;
//   * smaller spp_per_launch packs 64/S neighbouring pixels into the wave the same way;
//   * workgroups are dealt to image tiles through an XCD-aware remap so that the ~8 workgroups an
//     XCD runs at a time work inside one 32x32-pixel tile and share its DEM footprint in that
//     XCD's 4 MiB L2;
//   * FrameC comes by value -> SGPRs; no LDS, no MFMA (nothing here is a dense contraction).
//
// Arithmetic follows the spec of DESIGN.md section 3 operation by operation: explicit fmaf, build
// with -ffp-contract=off, correctly rounded / and sqrt (hipcc default), own polynomial atan/sin/cos.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "mrtx_device.h"

// The cold constants are read-only for the whole launch: address them through the constant address space so the
// (wave-uniform) reads become scalar loads (s_load_*) instead of per-lane vector loads.
#define CF(f) ((const __attribute__((address_space(4))) FrameCold*)(f).cold)

#ifndef MRTX_TRIAL_BATCH
#define MRTX_TRIAL_BATCH 1     // steps fetched together in the trial segment.  Round 4 (8 waves per SIMD, VALU issue 0.65): 1 step
#endif                         // 16.20 ms against 16.34 with 2 -- the 64 continuation rays are incoherent, a speculated second step is
                               // mostly thrown away; the coherent marches keep MRTX_STEP_BATCH = 2 (1: 16.64 ms, 3: 16.68)
#ifndef MRTX_ENV_PREFETCH
#define MRTX_ENV_PREFETCH 0    // 1: render_kernel<MODE 2> fetches the environment texel of a continuation ray BEFORE its trial segment, so
#endif                         // that the DRAM miss hides behind the march.  Bit-exact; measured on the star-map frame (gpurun_out/r4z): render
                               // 17.28 ms against 17.06 -- the 19 % extra look-ups (rays that hit or march on) cost more than the hidden round: off
#ifndef MRTX_TRIAL_SEGMENT
#define MRTX_TRIAL_SEGMENT 1   // 0 = hand every continuation ray to path_kernel unmarched (A/B switch, see trace_sample)
#endif

namespace mrtx {

__device__ constexpr float kPi = 3.14159274101257324f;
__device__ constexpr float kHalfPi = 1.57079637050628662f;
__device__ constexpr float kInv255 = 0.003921568859368563f;

// -DMRTX_PROF (tools/build_variant.sh prof -DMRTX_PROF): s_memtime section timers, summed per wave into g_prof.
// A measurement build only; the shipped library never defines it.
#ifdef MRTX_PROF
__device__ unsigned long long g_prof[16];
#ifdef MRTX_PROF_FULLIV       // this measurement build uses the timers' slots for its own counts
#define PROF_BEGIN(i)
#define PROF_END(i)
#else
#define PROF_BEGIN(i) const unsigned long long _pt##i = __builtin_readcyclecounter()
#define PROF_END(i) cnt[i] += (uint32_t)(__builtin_readcyclecounter() - _pt##i)
#endif
#else
#define PROF_BEGIN(i)
#define PROF_END(i)
#endif

#ifdef MRTX_PROF_MARGIN   // measurement build: how far above the surface are the steps path_kernel evaluates? (slots 10-13 of stats[])
enum { ST_PRIMARY = 0, ST_HITS, ST_SHADOW, ST_HEIGHT, ST_COLOUR, ST_BG, ST_FETCH, ST_MIP, ST_BOUNCE, ST_SUNHIT, ST_MALL, ST_M1, ST_M2, ST_M3, ST_N };
#else
enum { ST_PRIMARY = 0, ST_HITS, ST_SHADOW, ST_HEIGHT, ST_COLOUR, ST_BG, ST_FETCH, ST_MIP, ST_BOUNCE, ST_SUNHIT, ST_N };
#endif

// atan(q) ~= q * P(q^2) on [0,1], |err| <= 1.3e-7
__device__ __forceinline__ float atan_poly(float q) {
    const float s = q * q;
    float p = -0.004054343327879906f;
    p = fmaf(p, s, 0.02186218835413456f);
    p = fmaf(p, s, -0.05591127648949623f);
    p = fmaf(p, s, 0.0964212492108345f);
    p = fmaf(p, s, -0.1390860229730606f);
    p = fmaf(p, s, 0.19946560263633728f);
    p = fmaf(p, s, -0.33329859375953674f);
    p = fmaf(p, s, 0.9999993443489075f);
    return p * q;
}

// Correctly rounded sqrt for x in [2^-96, 2^96]: the raw v_sqrt_f32 (<= 1 ulp) plus the +-1 ulp residual
// test LLVM uses, without the denormal pre-scaling and class checks the general expansion carries
// (callers clamp x into the domain; tests compare against the host's IEEE sqrtf bit for bit).
__device__ __forceinline__ float sqrt_cr(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fmaf(-s_dn, s, x);
    const float r_up = fmaf(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}

// Correctly rounded 1/x for every NORMAL x whose reciprocal is normal (|x| in [2^-126, 2^126)): v_rcp_f32 (<= 1 ulp) + one Newton
// step with an exact fma residual -- 3 VALU instead of the 10 of the general division expansion (v_div_scale x 2, v_rcp, four fmas,
// v_div_fmas, v_div_fixup), which exists to survive operands and quotients at the ends of the exponent range.  Equality with the
// IEEE quotient 1.0f / x is not argued, it is CHECKED: mrtx_probe_cr() compares the two on the device for EVERY normal float of
// either sign (2^32 bit patterns, under a second; profiles/r04_probe_cr.txt): 0 mismatches for exponents -126 .. 125, all of them
// where 1/x is subnormal or x = 0 (tests/test_gpu_parity.py::test_domain_restricted_reciprocal_and_sqrt_are_ieee_exact).
// sqrt_cr() likewise: 0 mismatches for x = 0 and exponents -104 .. 127.  Callers keep their arguments inside these domains (each
// call site says why); the oracle uses the C compiler's IEEE division and sqrtf throughout.
#ifndef MRTX_RCP_STEPS
#define MRTX_RCP_STEPS 1      // measured exhaustively: ONE step is already exact wherever x and 1/x are normal (profiles/r04_probe_cr.txt)
#endif
#ifndef MRTX_FAST_CR
#define MRTX_FAST_CR 1      // 0 = the compiler's IEEE expansions everywhere (A/B switch; the results are the same bits)
#endif
template <int STEPS = 2>
__device__ __forceinline__ float rcp_nr(float x) {
    float y = __builtin_amdgcn_rcpf(x);
#pragma unroll
    for (int i = 0; i < STEPS; i++) { const float e = fmaf(-x, y, 1.0f); y = fmaf(e, y, y); }
    return y;
}
__device__ __forceinline__ float rcp_cr(float x) {
#if MRTX_FAST_CR
    return rcp_nr<MRTX_RCP_STEPS>(x);
#else
    return 1.0f / x;
#endif
}
// sqrt for the shading code: sqrt_cr where the argument is inside its domain (or exactly zero, which it returns as zero)
__device__ __forceinline__ float sqrt_sh(float x) {
#if MRTX_FAST_CR
    return sqrt_cr(x);
#else
    return sqrtf(x);
#endif
}

// (a, b, c) -> lat = atan2(c, rho), lon = atan2(a, b), rho = sqrt(max(a^2+b^2, 1e-28)).  The two min/max
// ratios share ONE correctly rounded reciprocal (a v_div_scale/v_rcp/fma/v_div_fixup chain is ~12 VALU);
// min/max instead of compare+select keeps VCC hazards (s_nop) out of the loop.
__device__ __forceinline__ void latlon(float a, float b, float c, float rho2, float& lat, float& lon) {
    const float rho = sqrt_cr(fmaxf(rho2, 1.0e-28f));
    const float aa = fabsf(a), ab = fabsf(b), ac = fabsf(c);
    const float m1 = fmaxf(rho, ac), n1 = fminf(rho, ac);
    const float m2 = fmaxf(ab, aa), n2 = fminf(ab, aa);
    const float den = fmaxf(m1 * m2, 1.0e-37f);
    const float t = rcp_cr(den);            // den in [1e-37, ~1e7]: normal, reciprocal normal
    float r1 = atan_poly(n1 * (t * m2));
    float r2 = atan_poly(n2 * (t * m1));
    r1 = ac >= rho ? kHalfPi - r1 : r1;
    r2 = aa >= ab ? kHalfPi - r2 : r2;
    r2 = b < 0.0f ? kPi - r2 : r2;
    lat = copysignf(r1, c);
    lon = copysignf(r2, a);
}

// cos / sin of 2*pi*u, u in [0,1): quadrant split + polynomials on [0, pi/2)
__device__ __forceinline__ void sincos_turn(float u, float& cs, float& sn) {
    const float t4 = u * 4.0f;
    const float qf = floorf(t4);
    const float a = (t4 - qf) * kHalfPi;
    const float a2 = a * a;
    float sp = 2.590481244624243e-06f;
    sp = fmaf(sp, a2, -0.00019800894369836897f);
    sp = fmaf(sp, a2, 0.008332899771630764f);
    sp = fmaf(sp, a2, -0.16666647791862488f);
    sp = fmaf(sp, a2, 1.0f);
    const float s1 = sp * a;
    float cp = 2.3153859729063697e-05f;
    cp = fmaf(cp, a2, -0.001385370153002441f);
    cp = fmaf(cp, a2, 0.04166358336806297f);
    cp = fmaf(cp, a2, -0.4999990463256836f);
    cp = fmaf(cp, a2, 0.9999999403953552f);
    const float c1 = cp;
    const int qi = (int)qf;
    cs = qi == 0 ? c1 : (qi == 1 ? -s1 : (qi == 2 ? -c1 : s1));
    sn = qi == 0 ? s1 : (qi == 1 ? c1 : (qi == 2 ? -s1 : -c1));
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t key, uint32_t dim) {
    const uint32_t r = mix32(key + (dim + 1u) * 0x9E3779B9u);
    return (float)(r >> 8) * 5.9604644775390625e-08f;
}

__device__ __forceinline__ float lerp2(float e00, float e01, float e10, float e11, float fr, float fc) {
    const float top = fmaf(fc, e01 - e00, e00);
    const float bot = fmaf(fc, e11 - e10, e10);
    return fmaf(fr, bot - top, top);
}

// Bilinear taps, floor() form of renderer_navigation.py:581-588: r0 = floor(row), c0 = floor(col); rows r0 and
// r0+1 clamp to [0,h-1], columns c0 and c0+1 wrap into [0,w).
//
// The DEM lives in HBM PADDED by two texels on every side (rows -2,-1 = row 0, rows h,h+1 = row h-1, columns
// -2,-1 = columns w-2,w-1, columns w,w+1 = columns 0,1; pitch = w+4), so a bilinear evaluation -- on the march
// path, where floor() lands in [-1,h-1] x [-1,w-1], and one texel either side of it for the normal -- is two
// unconditional 8-byte loads: no clamp, no wrap, no seam branch.
// The hand-over records between render_kernel<MODE 2>, path_kernel and resolve_paths_kernel are written once and read once,
// 12 GB per cfg-3 frame: NON-TEMPORAL accesses keep them from sweeping the DEM neighbourhoods out of the 4 MB L2s
// (with plain stores render_kernel<MODE 2> took 17.7 ms against 13.4 without the stores at all).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* p, float a, float b, float c, float d) {
    v4f v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// a sample's running radiance in the hand-over buffers (PathQ::c0/c1/c2, or one float4 per sample with MRTX_C_AOS)
typedef float v3f __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) Tri { float x, y, z; };
__device__ __forceinline__ void c_load(const PathQ& pq, uint32_t e, float& c0, float& c1, float& c2) {
#if MRTX_C_AOS == 2
    const Tri v = reinterpret_cast<const Tri*>(pq.c4)[e]; c0 = v.x; c1 = v.y; c2 = v.z;
#elif MRTX_C_AOS
    const float4 v = pq.c4[e]; c0 = v.x; c1 = v.y; c2 = v.z;
#else
    c0 = pq.c0[e]; c1 = pq.c1[e]; c2 = pq.c2[e];
#endif
}
__device__ __forceinline__ void c_store(const PathQ& pq, uint32_t e, float c0, float c1, float c2) {
#if MRTX_C_AOS == 2
    { Tri v; v.x = c0; v.y = c1; v.z = c2; reinterpret_cast<Tri*>(pq.c4)[e] = v; }
#elif MRTX_C_AOS
    pq.c4[e] = make_float4(c0, c1, c2, 0.0f);
#else
    pq.c0[e] = c0; pq.c1[e] = c1; pq.c2[e] = c2;
#endif
}
struct __attribute__((packed, aligned(4))) Pair { float x, y; };
struct __attribute__((packed, aligned(8))) Quad { float a, b, c, d; };
struct __attribute__((packed, aligned(8))) UQuad { uint32_t a, b, c, d; };

// CP: cache policy of the load -- 0 plain, 1 non-temporal (a line the wave will not touch again: the incoherent marches)
#ifndef MRTX_TRIAL_CP
#define MRTX_TRIAL_CP 0
#endif
#ifndef MRTX_PATH_CP
#define MRTX_PATH_CP 0
#endif
template <bool WIDE, int CP = 0>
__device__ __forceinline__ float dem_march(const FrameC& f, float rowf, float colf) {
    const float rfl = floorf(rowf), cfl = floorf(colf);
    const float fr = rowf - rfl, fc = colf - cfl;
    // padded index of (r0, c0) = (r0+2)*pitch + (c0+2); both factors < 2^24 -> one v_mad_u32_u24.  A single
    // unsigned min keeps any garbage (NaN position) inside the array; it never bites for a valid (lat, lon).
    const uint32_t r0p = (uint32_t)((int)rfl + 2), c0p = (uint32_t)((int)cfl + 2);
    const uint32_t idx = min(__umul24(r0p, (uint32_t)f.dem_pitch) + c0p, f.dem_maxidx);
    const char* base = reinterpret_cast<const char*>(f.dem);
#if MRTX_DEM_PAIRS
    // row-pair layout: element (r, c) = (D[r][c], D[r+1][c]); elements (r0, c0) and (r0, c0+1) are adjacent, so the
    // whole 2x2 footprint is ONE 16-byte load -- half the gather instructions and L1 tag look-ups of two row loads
    Quad q;
    if (CP == 2) {
        // TIMING-ONLY A/B (results wrong; -DMRTX_PATH_CP=2): what would path_kernel gain from a DEM of half the bytes per texel
        // (16-bit codes: round-3 verdict item 6)?  The same gather, 8 bytes from a buffer addressed at 4 bytes per element, so
        // that a 128-byte line covers 32 columns x 2 rows: an upper bound on the traffic effect, without any decode cost.
        const Pair h = *reinterpret_cast<const Pair*>(WIDE ? base + ((uint64_t)idx << 2) : base + (idx << 2));
        q.a = h.x; q.b = h.y; q.c = h.x; q.d = h.y;
    } else if (CP == 1) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(WIDE ? base + ((uint64_t)idx << 3) : base + (idx << 3)));
        q.a = v.x; q.b = v.y; q.c = v.z; q.d = v.w;
    } else if (WIDE) q = *reinterpret_cast<const Quad*>(base + ((uint64_t)idx << 3));
    else q = *reinterpret_cast<const Quad*>(base + (idx << 3));
    return lerp2(q.a, q.c, q.b, q.d, fr, fc);
#else
    Pair t, u;
    if (WIDE) {
        const char* p = base + ((uint64_t)idx << 2);
        t = *reinterpret_cast<const Pair*>(p);
        u = *reinterpret_cast<const Pair*>(p + ((uint64_t)(uint32_t)f.dem_pitch << 2));
    } else {
        const uint32_t off0 = idx << 2, off1 = off0 + ((uint32_t)f.dem_pitch << 2);
        t = *reinterpret_cast<const Pair*>(base + off0);
        u = *reinterpret_cast<const Pair*>(base + off1);
    }
    return lerp2(t.x, t.y, u.x, u.y, fr, fc);
#endif
}

__device__ __forceinline__ int32_t wrapc(int32_t c, int32_t w) {
    if (c < 0) c += w;
    if (c >= w) c -= w;
    if (c < 0) c += w;
    if (c >= w) c -= w;
    return c;
}
// ---- D2/D3: the march.
// Texel coordinates are smooth along a ray, while the exact (lat, lon) -> (row, col) costs ~65 VALU (sqrt,
// reciprocal, two degree-15 polynomials, octant logic) and this kernel is VALU-issue bound.  Per SEG_N-step
// segment the exact coordinates are evaluated at the segment's start, middle and end only; the steps in
// between use the quadratic through those three (|error| <= ~1e-3 row / 7e-3 column texels for rho >= 0.2 R,
// i.e. the float32 resolution of the coordinate itself).  Segments that touch the polar cap or straddle
// the +/-180 seam evaluate every step exactly.  DEM evaluations, hit tests and counters are unchanged.
constexpr int SEG_N = 16;
struct Seg {
    float sa, ra, r1, r2, ca, c1, c2;
    int jlo, jhi;   // steps of this segment that can possibly be at/below the surface (see seg_setup)
#ifdef MRTX_PROF_FULLIV
    int why;        // measurement only: why the max-mip gave no interval (1 rows, 2 columns, 3 map edge; 0 = it did)
#endif
    bool exact;
};
// per-march constants of r^2(s) = q0 + 2 b s + a s^2
struct RayQ { float q0, b, a; };

__device__ __forceinline__ void exact_rowcol(const FrameC& f, float pa, float pb, float pc, float& rowf, float& colf,
                                             float& rho2) {
    rho2 = fmaf(pb, pb, pa * pa);
    float lat, lon;
    latlon(pa, pb, pc, rho2, lat, lon);
    rowf = fmaf(lat, f.gd.row_scale, f.gd.row_off);
    colf = fmaf(lon, f.gd.col_scale, f.gd.col_off);
}

// Anchors + quadratic of one segment, and the RESULT-PRESERVING skip interval:
// the max-mip (64x64-texel cell maxima, ~1 MB, cache resident) bounds D over the footprint of the three
// anchors (+1 texel for the bilinear tap and the quadratic's bulge): D <= Dmax there.  A step can only be at or
// below the surface if r^2(s) <= (R Dmax)^2; r^2(s) is a parabola in s, so those steps form one interval
// [jlo, jhi] (widened by a step each side and by 1e-5 in the bound, which dwarfs every rounding involved,
// so approximate v_sqrt/v_rcp are fine here).  Steps outside it cannot hit and are not evaluated; the ray's
// termination test is monotone, so it is enough to apply it at evaluated steps and at the segment end.
// seg_setup in two phases, so that a caller can put OTHER work between the max-mip fetch and its use (fused_first_segment):
// seg_anchors = the anchors, the quadratic and WHERE the max-mip is to be read; seg_interval = the skip interval from the cells.
#ifndef MRTX_WIDE_TAP
#define MRTX_WIDE_TAP 1       // a footprint of up to MRTX_TAP_COLS cells along the columns still gets its skip interval (see seg_anchors)
#endif
#ifndef MRTX_TAP_COLS
#define MRTX_TAP_COLS 4
#endif
struct MipTap { uint32_t off; bool usable, two_r, two_c; int ncol; };
__device__ __forceinline__ void seg_anchors(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                            int ka, float rowA, float colA, float q2A, Seg& sg,
                                            float& rowB, float& colB, float& q2B, MipTap& tap) {
    const float sm = (float)(ka + SEG_N / 2) * f.step, sb = (float)(ka + SEG_N) * f.step;
    float rM, cM, q2M;
    exact_rowcol(f, fmaf(sm, da, oa), fmaf(sm, db, ob), fmaf(sm, dc, oc), rM, cM, q2M);
    exact_rowcol(f, fmaf(sb, da, oa), fmaf(sb, db, ob), fmaf(sb, dc, oc), rowB, colB, q2B);
    const float hw = 0.5f * f.gd.wf;
    const float qmin = fminf(q2A, fminf(q2M, q2B));
    sg.exact = (fabsf(cM - colA) > hw) || (fabsf(colB - colA) > hw) || (qmin < f.polar_rho2);
    sg.sa = (float)ka * f.step;
    sg.ra = rowA; sg.ca = colA;
    sg.r2 = (fmaf(-2.0f, rM, rowA) + rowB) * 0.0078125f;
    sg.r1 = fmaf(-16.0f, sg.r2, (rowB - rowA) * 0.0625f);
    sg.c2 = (fmaf(-2.0f, cM, colA) + colB) * 0.0078125f;
    sg.c1 = fmaf(-16.0f, sg.c2, (colB - colA) * 0.0625f);

    sg.jlo = 1; sg.jhi = SEG_N;
    tap.usable = false; tap.two_r = tap.two_c = false; tap.off = 0u; tap.ncol = 1;
    if (f.mip != nullptr) {
        const int i0 = ((int)floorf(fminf(rowA, fminf(rM, rowB))) - 1) >> f.mip_shift;
        const int i1 = ((int)floorf(fmaxf(rowA, fmaxf(rM, rowB))) + 2) >> f.mip_shift;
        const int j0 = ((int)floorf(fminf(colA, fminf(cM, colB))) - 1) >> f.mip_shift;
        const int j1 = ((int)floorf(fmaxf(colA, fmaxf(cM, colB))) + 2) >> f.mip_shift;
        // Columns shrink with cos(latitude): a ray that travels east-west at 30 degrees of latitude already covers more columns in
        // 16 steps than a cell is wide, and round 4 found 23 % of all camera and shadow segments of the cfg3 frame WITHOUT a skip
        // interval for that reason alone.  Up to four cells along the columns are therefore allowed (a second 16-byte load).
        tap.usable = !sg.exact & (i1 - i0 <= 1) & (j1 - j0 <= (MRTX_WIDE_TAP ? MRTX_TAP_COLS - 1 : 1)) & (i0 >= -1) & (i1 <= f.mip_h) & (j0 >= -1) &
                     (j1 <= f.mip_w);
        tap.two_r = i1 > i0; tap.two_c = j1 > j0; tap.ncol = j1 - j0 + 1;
#ifdef MRTX_PROF_FULLIV
        sg.why = sg.exact ? 4 : (i1 - i0 > 1) ? 1 : (j1 - j0 > MRTX_TAP_COLS - 1) ? 2 : tap.usable ? 0 : 3;
#endif
        // the mip is stored in row pairs as well (element (i, j) = (m[i][j], m[i+1][j])): one 16-byte load brings the
        // 2x2 cells at (i0, j0); the ones the footprint does not reach are ignored, so the bound is the old one
        tap.off = tap.usable ? ((uint32_t)((i0 + 1) * f.mip_pitch + j0 + 1) << 3) : 0u;
    }
}
__device__ __forceinline__ Quad mip_fetch(const FrameC& f, const MipTap& tap) {
    return *reinterpret_cast<const Quad*>(reinterpret_cast<const char*>(f.mip) + tap.off);
}
template <bool STATS>
__device__ __forceinline__ void seg_interval(const FrameC& f, const RayQ& rq, Seg& sg, const MipTap& tap, const Quad& q,
                                             uint32_t* cnt, float dmax_more = 0.0f) {
    const float dmax = fmaxf(fmaxf(fmaxf(q.a, tap.two_r ? q.b : q.a), fmaxf(tap.two_c ? q.c : q.a, (tap.two_r & tap.two_c) ? q.d : q.a)), dmax_more);
    if (STATS) cnt[ST_MIP] += 4;
    const float rd = f.Rf * dmax;
    const float T = (rd * rd) * 1.00001f;
    const float disc = fmaf(rq.b, rq.b, -rq.a * (rq.q0 - T));
    if (disc < 0.0f) {
        sg.jlo = SEG_N + 1; sg.jhi = SEG_N;      // the whole segment stays above Dmax
    } else {
        const float sq = __builtin_amdgcn_sqrtf(disc), inva = __builtin_amdgcn_rcpf(rq.a);
        const float u1 = fminf(fmaxf(((-rq.b - sq) * inva - sg.sa) * f.inv_step, -4.0f), 64.0f);
        const float u2 = fminf(fmaxf(((-rq.b + sq) * inva - sg.sa) * f.inv_step, -4.0f), 64.0f);
        sg.jlo = min(SEG_N + 1, max(1, (int)floorf(u1) - 1));
        sg.jhi = min(SEG_N, (int)ceilf(u2) + 1);
    }
}
template <bool STATS>
__device__ __forceinline__ void seg_setup(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                          const RayQ& rq, int ka, float rowA, float colA, float q2A, Seg& sg,
                                          float& rowB, float& colB, float& q2B, uint32_t* cnt) {
    MipTap tap;
    seg_anchors(f, oa, ob, oc, da, db, dc, ka, rowA, colA, q2A, sg, rowB, colB, q2B, tap);
    if (tap.usable) {
        const Quad q = mip_fetch(f, tap);
        float more = 0.0f;       // D > 0 everywhere: zero is neutral for the maximum
        if (MRTX_WIDE_TAP) {
#pragma unroll
            for (int c = 2; c < MRTX_TAP_COLS; c += 2) {      // columns j0 + c (and j0 + c + 1): the next two cells of the same row pair
                if (tap.ncol > c) {
                    const Quad q2 = *reinterpret_cast<const Quad*>(reinterpret_cast<const char*>(f.mip) + tap.off + 8u * (uint32_t)c);
                    more = fmaxf(more, fmaxf(q2.a, tap.two_r ? q2.b : q2.a));
                    if (tap.ncol > c + 1) more = fmaxf(more, fmaxf(q2.c, tap.two_r ? q2.d : q2.c));
                    if (STATS) cnt[ST_MIP] += 4;
                }
            }
        }
        seg_interval<STATS>(f, rq, sg, tap, q, cnt, more);
    }
}

// is the point at or below the displaced surface?  r^2 <= (R * D(row, col))^2
// EXACTABLE = false: the caller knows (by ballot) that no lane of the wave is in an exact-fallback segment.
// The quadratic needs no clamp: a non-seam, non-polar segment keeps (row, col) >= 0.5 texel inside
// [-1, h) x [-1, w), and dem_march()'s unsigned index clamp keeps even a NaN inside the allocation.
template <bool WIDE, bool EXACTABLE, int CP = 0>
__device__ __forceinline__ bool below_seg(const FrameC& f, const Seg& sg, float sk, float pa, float pb, float pc,
                                          float r2) {
    const float u = (sk - sg.sa) * f.inv_step;
    float rowf = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra);
    float colf = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
    if (EXACTABLE && sg.exact) {
        float q2;
        exact_rowcol(f, pa, pb, pc, rowf, colf, q2);
    }
    const float surf = f.Rf * dem_march<WIDE, CP>(f, rowf, colf);
    return r2 <= surf * surf;
}

// The steps jlo..jhi of one segment, per lane.  Branch-free body: the DEM is sampled even on the step that
// turns out to lie outside (its result is discarded), so the only control flow is the loop-back on the
// ballot of lanes still stepping.
//
// The kernel is bound by DEPENDENT-LOAD LATENCY (each round trip ~1-2 k cycles under load, five waves per SIMD
// to hide it), so the plain-quadratic variant evaluates MRTX_STEP_BATCH consecutive steps per iteration: all
// their DEM loads are issued back to back, then the steps are tested in march order and everything after the
// first terminating one is discarded.  Same evaluations, same order, same result; a few wasted fetches.
// Measured at cfg3: batch 1 15.19 ms, 2 14.83, 4 16.57 (+11 % fetches), 8 18.55.
#ifndef MRTX_STEP_BATCH
#define MRTX_STEP_BATCH 2
#endif
#ifndef MRTX_SHADOW_BATCH
#define MRTX_SHADOW_BATCH MRTX_STEP_BATCH     // the first vertex's shadow march in render_kernel<MODE 2> (A/B: 1 / 2 / 3)
#endif
#ifndef MRTX_STEP_BATCH_BOUNCE
#define MRTX_STEP_BATCH_BOUNCE 1
#endif
template <bool WIDE, bool PRIMARY, bool STATS, bool EXACTABLE, int BATCH, int CP = 0>
__device__ __forceinline__ void step_loop_from(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                               float smax, const Seg& sg, int ka, int j, bool more, bool& go, bool& hit,
                                               float& sk_out, uint32_t* cnt) {
    if (EXACTABLE || BATCH == 1) {
        while (more) {
            const int k = ka + j;
            const float sk = (float)k * f.step;
            const float pa = fmaf(sk, da, oa), pb = fmaf(sk, db, ob), pc = fmaf(sk, dc, oc);
            const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
            const bool in = (PRIMARY ? (sk <= smax) : (r2 <= f.R2f)) & (k <= f.kmax);
            const bool bel = below_seg<WIDE, EXACTABLE, CP>(f, sg, sk, pa, pb, pc, r2);
            if (STATS) { cnt[ST_HEIGHT] += in ? 1u : 0u; cnt[ST_FETCH]++; }
#ifdef MRTX_PROF
            cnt[11] += 1;                                    // wave-level step iterations
            cnt[12] += (uint32_t)__popcll(__ballot(true));   // lanes evaluating in them
#endif
            hit = in & bel;
            go = in & !bel;
            sk_out = sk;
            j++;
            more = go & (j <= sg.jhi);
        }
    } else {
        constexpr int B = BATCH;
        while (more) {
            float surf[B];
#pragma unroll
            for (int i = 0; i < B; i++) {
                // steps past jhi are evaluated at the segment's last step instead (inside the quadratic's range)
                const float u = ((float)(ka + min(j + i, SEG_N)) * f.step - sg.sa) * f.inv_step;   // as below_seg()
                const float rowf = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra);
                const float colf = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
                surf[i] = f.Rf * dem_march<WIDE, CP>(f, rowf, colf);
            }
#ifdef MRTX_PROF
            cnt[11] += 1;
            cnt[12] += (uint32_t)__popcll(__ballot(true));
#endif
            bool act = true;
#pragma unroll
            for (int i = 0; i < B; i++) {
                const int k = ka + j + i;
                const float sk = (float)k * f.step;
                const float pa = fmaf(sk, da, oa), pb = fmaf(sk, db, ob), pc = fmaf(sk, dc, oc);
                const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
                const bool in = (PRIMARY ? (sk <= smax) : (r2 <= f.R2f)) & (k <= f.kmax);
                const bool bel = r2 <= surf[i] * surf[i];
                if (STATS) { cnt[ST_HEIGHT] += (act & in) ? 1u : 0u; cnt[ST_FETCH] += act ? 1u : 0u; }   // speculative fetches are not credited
                hit = act ? (in & bel) : hit;
                go = act ? (in & !bel) : go;
                sk_out = act ? sk : sk_out;
                act = act & go & (j + i + 1 <= sg.jhi);
            }
            j += B;
            more = act;
        }
    }
}

template <bool WIDE, bool PRIMARY, bool STATS, bool EXACTABLE, int BATCH, int CP = 0>
__device__ __forceinline__ void step_loop(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                          float smax, const Seg& sg, int ka, bool& go, bool& hit, float& sk_out,
                                          uint32_t* cnt) {
    step_loop_from<WIDE, PRIMARY, STATS, EXACTABLE, BATCH, CP>(f, oa, ob, oc, da, db, dc, smax, sg, ka, sg.jlo, sg.jlo <= sg.jhi,
                                                           go, hit, sk_out, cnt);
}

// STATS builds only: the spec counts a DEM evaluation at every step that is still inside; add the skipped ones.
template <bool PRIMARY>
__device__ __forceinline__ uint32_t count_in_steps(const FrameC& f, float oa, float ob, float oc, float da, float db,
                                                   float dc, float smax, int ka, int j_from, int j_to) {
    uint32_t n = 0;
    for (int j = j_from; j <= j_to; j++) {
        const int k = ka + j;
        const float sk = (float)k * f.step;
        const float pa = fmaf(sk, da, oa), pb = fmaf(sk, db, ob), pc = fmaf(sk, dc, oc);
        const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
        const bool in = (PRIMARY ? (sk <= smax) : (r2 <= f.R2f)) & (k <= f.kmax);
        if (!in) break;
        n++;
    }
    return n;
}

// Per-ray march state between segments: the ray, the coefficients of r^2(s) and the exact texel coordinates at the
// start of the next segment (step ka).
struct MarchState {
    float oa, ob, oc, da, db, dc;
    RayQ rq;
    float rowA, colA, q2A;
    int ka;
    int kend;   // no step beyond this one can be at/below the surface (horizon_kend); kmax when nothing is known
};

// RESULT-PRESERVING end of a shadow / continuation march: once an ASCENDING ray (b = o.d >= 0, so r^2(s) grows
// monotonically) is above everything its remaining ground track can reach, no later step can be at/below the surface,
// and the march can stop there instead of stepping -- or setting up empty segments -- until it leaves the bounding sphere.
// "Everything it can reach" comes from the HORIZON MIP: cells of Cc = 8 fine-mip cells (512 texels at cfg 3), each holding
// the maximum of D over the cell DILATED by Cc texels on every side (rows clamp, columns wrap), so one look-up at the
// ray's origin bounds D over any ground track that stays within Cc texels of it.  The track's extent is bounded from the
// chord to the sphere exit L: it subtends phi <= 1.03 L / r0 at the centre (the ray stays above its origin radius r0), at most
// phi * h/pi rows and asin(sin phi / cos(lat_max)) * w/2pi <= 1.05 phi / (cos(lat0) - phi) * w/2pi columns; the test needs
// both (+4 texels for taps and the quadratic's bulge) inside Cc, otherwise nothing is cut.  Then the last step that can
// matter is where r^2(s) reaches (R Dc)^2 (1 + 1e-5).  Approximate v_sqrt / v_rcp are fine: every bound is padded.
// Radiance, hits and the spec counters are unchanged (MRTX_F_NO_SKIP switches this off together with the max-mip).
// the horizon-mip cell of a march origin at texel (rowA, colA): one look-up serves every ray that starts there
__device__ __forceinline__ float horizon_cell(const FrameC& f, float rowA, float colA) {
    int i = (int)floorf(rowA) >> CF(f)->hm_shift, j = (int)floorf(colA) >> CF(f)->hm_shift;
    i = max(0, min(i, CF(f)->hm_h - 1)); j = max(0, min(j, CF(f)->hm_w - 1));
    return CF(f)->hmip[i * CF(f)->hm_w + j];
}
// PRE: the caller has fetched horizon_cell(f, m.rowA, m.colA) already (`cell_pre`): same bound, no load here
template <bool PRE = false>
__device__ __forceinline__ int horizon_kend(const FrameC& f, const MarchState& m, float cell_pre = 0.0f) {
    int kend = f.kmax;
    const float* hm = CF(f)->hmip;
    if (hm != nullptr && m.rq.b >= 0.0f) {
        const float a = m.rq.a, b = m.rq.b, q0 = m.rq.q0;
        const float c = f.R2f - q0;                        // >= 0: the origin is inside the bounding sphere
        const float L = c * __builtin_amdgcn_rcpf(b + __builtin_amdgcn_sqrtf(fmaf(a, c, b * b)) + 1.0e-30f) * 1.02f;
        const float inv_cos = __builtin_amdgcn_sqrtf(q0 * __builtin_amdgcn_rcpf(fmaxf(m.q2A, 1.0e-30f)));   // r0 / rho0
        const float phi = L * __builtin_amdgcn_rsqf(q0) * 1.03f;   // 2 asin(L / 2 r0) <= 1.003 L / r0 for L <= r0 / 4; r(s) >= r0
        const float den = 1.0f - phi * inv_cos;            // cos(lat0) - phi, in units of cos(lat0)
        const float drow = fmaf(phi, CF(f)->hm_krow, 4.0f);
        const float dcol = fmaf(phi * CF(f)->hm_kcol, inv_cos * __builtin_amdgcn_rcpf(fmaxf(den, 0.25f)), 4.0f);
        const float cell = CF(f)->hm_cell;
        if ((c >= 0.0f) & (phi <= 0.25f) & (den >= 0.5f) & (drow <= cell) & (dcol <= cell)) {
            const float rd = f.Rf * (PRE ? cell_pre : horizon_cell(f, m.rowA, m.colA));
            const float d = (rd * rd) * 1.00001f - q0;
            if (d <= 0.0f) kend = 0;
            else {
                const float sc = d * __builtin_amdgcn_rcpf(b + __builtin_amdgcn_sqrtf(fmaf(a, d, b * b))) * 1.001f;
                kend = min(f.kmax, (int)(sc * f.inv_step) + 2);
            }
        }
    }
    return kend;
}
// horizon_kend() asked again from the start of segment m.ka (texel coordinates m.rowA / m.colA, rho^2 = m.q2A): the parabola's
// coefficients moved to that point.  Steps are counted from there; f.kmax = nothing known.
#ifndef MRTX_HORIZON_RETRY
#define MRTX_HORIZON_RETRY 2     // bit 0: render_kernel's shadow marches (measured: +0.55 ms, the test runs for the whole wave), bit 1: path_kernel (-0.13 ms)
#endif
__device__ __forceinline__ int horizon_retry(const FrameC& f, const MarchState& m) {
    const float s = (float)m.ka * f.step;
    MarchState t;
    t.rq.a = m.rq.a;
    t.rq.b = fmaf(m.rq.a, s, m.rq.b);
    t.rq.q0 = fmaf(s, fmaf(s, m.rq.a, m.rq.b + m.rq.b), m.rq.q0);
    t.q2A = m.q2A; t.rowA = m.rowA; t.colA = m.colA;
    return horizon_kend(f, t);
}
// STATS builds: the steps the spec evaluates after a march was cut at kend (every step while the ray is inside)
__device__ __forceinline__ uint32_t steps_after(const FrameC& f, const MarchState& m, int k_from) {
    uint32_t n = 0;
    for (int k = k_from; k <= f.kmax; k++) {
        const float sk = (float)k * f.step;
        const float pa = fmaf(sk, m.da, m.oa), pb = fmaf(sk, m.db, m.ob), pc = fmaf(sk, m.dc, m.oc);
        if (!(fmaf(pc, pc, fmaf(pb, pb, pa * pa)) <= f.R2f)) break;
        n++;
    }
    return n;
}

// Start of a march: exact coordinates at the origin, r^2(s) coefficients; returns `go` (false: the march is over before
// its first step).
// ... with the exact texel coordinates of the origin already known (m.rowA, m.colA)
// LAZY_KEND (path_kernel): the horizon bound is left open (m.kend = -1) and looked up by the first segment set-up, in the
// same memory round as that segment's max-mip fetch, instead of costing a round of its own here.
template <bool PRIMARY, bool STATS, bool LAZY_KEND = false, bool PRE_CELL = false>
__device__ __forceinline__ bool march_begin_at(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                               MarchState& m, uint32_t* cnt, float cell_pre = 0.0f) {
    m.oa = oa; m.ob = ob; m.oc = oc; m.da = da; m.db = db; m.dc = dc;
    m.q2A = fmaf(ob, ob, oa * oa);
    m.rq.q0 = fmaf(oc, oc, m.q2A);
    m.rq.b = fmaf(oc, dc, fmaf(ob, db, oa * da));
    m.rq.a = fmaf(dc, dc, fmaf(db, db, da * da));
    m.ka = 0;
    bool go = true;
    if (!PRIMARY) {
        // The skip below relies on "once outside, always outside".  r^2(s) is convex, so that holds from the first
        // step that is inside -- but an origin lifted by scene_epsilon off a D = 1 texel can sit just outside R and head
        // inward: the march ends at step 1 (spec), and must not resume where the parabola dips back inside.
        const float s1 = f.step;
        const float pa = fmaf(s1, da, oa), pb = fmaf(s1, db, ob), pc = fmaf(s1, dc, oc);
        go = fmaf(pc, pc, fmaf(pb, pb, pa * pa)) <= f.R2f;
        if (LAZY_KEND) {
            m.kend = -1;
        } else {
            m.kend = horizon_kend<PRE_CELL>(f, m, cell_pre);
            if (go && m.kend < 1) {          // already above everything in reach: no step can hit
                if (STATS) cnt[ST_HEIGHT] += steps_after(f, m, 1);
                go = false;
            }
        }
    } else {
        m.kend = f.kmax;
    }
    return go;
}
template <bool PRIMARY, bool STATS, bool LAZY_KEND = false>
__device__ __forceinline__ bool march_begin(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                            MarchState& m, uint32_t* cnt) {
    float q2;
    exact_rowcol(f, oa, ob, oc, m.rowA, m.colA, q2);
    return march_begin_at<PRIMARY, STATS, LAZY_KEND>(f, oa, ob, oc, da, db, dc, m, cnt);
}

// End of a segment whose steps are through: a ray that is still marching (`go`) may have ended inside the skipped tail, or
// is cut by its horizon bound; the march state moves on to the next segment.
template <bool PRIMARY, bool STATS>
__device__ __forceinline__ void segment_tail(const FrameC& f, MarchState& m, float smax, const Seg& sg, bool& go, float rowB,
                                             float colB, float q2B, uint32_t* cnt) {
    const float oa = m.oa, ob = m.ob, oc = m.oc, da = m.da, db = m.db, dc = m.dc;
    const int ka = m.ka;
    if (go) {
        // still marching after the last evaluated step: did the ray end inside the skipped tail?
        if (STATS) cnt[ST_HEIGHT] += count_in_steps<PRIMARY>(f, oa, ob, oc, da, db, dc, smax, ka, max(sg.jhi + 1, 1), SEG_N);
        const int k = ka + SEG_N;
        const float sk = (float)k * f.step;
        const float pa = fmaf(sk, da, oa), pb = fmaf(sk, db, ob), pc = fmaf(sk, dc, oc);
        go = (PRIMARY ? (sk <= smax) : (fmaf(pc, pc, fmaf(pb, pb, pa * pa)) <= f.R2f)) & (k < f.kmax);
        if (!PRIMARY && go && k >= m.kend) {           // cut by the horizon bound: the rest of the ray is above the terrain
            if (STATS) cnt[ST_HEIGHT] += steps_after(f, m, k + 1);
            go = false;
        }
    }
    m.ka = ka + SEG_N; m.rowA = rowB; m.colA = colB; m.q2A = q2B;
}

// Camera rays: the FIRST step of the skip interval [jlo, jhi] that the medium max-mip cannot prove above the surface (jhi + 1 when it
// proves them all).  A camera ray descends onto the terrain and its march ends at the first step at or below it, so only the front of
// the interval matters: the steps are tested in march order, four per memory round, and a lane stops at its first inconclusive one.
// The test itself only has to be CONSERVATIVE, not the spec's arithmetic: the step's texel position from the segment's quadratic at
// u = j (the spec's u differs by < 3e-4, a thousandth of a texel; a cell's maximum covers two texels more than its own rows and
// columns on the low side and one more than a bilinear tap needs on the high side, mip_build_kernel), r^2 from the ray's parabola
// (its terms are ~R^2 each and s reaches 2R: good to ~1e-6 relative, a tenth of the comparison's 1e-5 margin -- the margin seg_interval
// has relied on since round 1; the evaluation's own r^2 is as close to the true value).  Result-preserving like every other skip.
#ifndef MRTX_PMASK_Q
#define MRTX_PMASK_Q 4        // tests per memory round in render_kernel's marches (cfg3: 2 -> 14.0 ms, 3 -> 13.9, 4 -> 13.75, 6 -> 13.9)
#endif
#ifndef MRTX_PATH_MASK_Q
#define MRTX_PATH_MASK_Q 8    // ... and in path_kernel, which is bound by its dependent memory rounds (4 -> 4.84 ms, 6 -> 4.75, 8 -> 4.64, 16 -> 6.4: spills)
#endif
// ... and the mirror image for shadow and continuation rays, which LEAVE the terrain: their first steps are close to the surface,
// the later ones far above it, so the interval is cut from its END -- the steps are tested backwards from jhi and a lane stops at
// the first one the medium mip cannot prove above the surface: that is the new jhi (jlo - 1 when every step is proven above).
template <bool STATS, int Q>
__device__ __forceinline__ int last_kept_step(const FrameC& f, const MarchState& m, const Seg& sg, uint32_t* cnt) {
    const float* m2 = CF(f)->mip2;
    const int pitch = CF(f)->m2_pitch, sh = CF(f)->m2_shift;
    const uint32_t maxidx = (uint32_t)((CF(f)->m2_h + 2) * pitch - 1);
    const float two_b = m.rq.b + m.rq.b;
    int j = sg.jhi;
    int last = sg.jhi;
    bool open = (sg.jlo <= sg.jhi) & !sg.exact;
    if (open) last = sg.jlo - 1;
    while (open) {
        float mv[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const float u = (float)max(j - q, sg.jlo);
            const float rowf = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra), colf = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
            const int i = (int)floorf(rowf) >> sh, c = (int)floorf(colf) >> sh;
            mv[q] = m2[min((uint32_t)((i + 1) * pitch + c + 1), maxidx)];
        }
        bool found = false;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int jj = j - q;
            const float sk = fmaf((float)max(jj, sg.jlo), f.step, sg.sa);
            const float r2 = fmaf(sk, fmaf(sk, m.rq.a, two_b), m.rq.q0);
            const float rd = f.Rf * mv[q];
            const bool kept = !(r2 > (rd * rd) * 1.00001f);
            if (STATS) cnt[ST_MIP] += (!found && jj >= sg.jlo) ? 1u : 0u;
            if (!found && jj >= sg.jlo && kept) { last = jj; found = true; }
        }
        j -= Q;
        open = !found && j >= sg.jlo;
    }
    return last;
}
template <bool STATS, int Q>
__device__ __forceinline__ int first_kept_step(const FrameC& f, const MarchState& m, const Seg& sg, uint32_t* cnt) {
    const float* m2 = CF(f)->mip2;
    const int pitch = CF(f)->m2_pitch, sh = CF(f)->m2_shift;
    const uint32_t maxidx = (uint32_t)((CF(f)->m2_h + 2) * pitch - 1);
    const float two_b = m.rq.b + m.rq.b;
    int j = sg.jlo;
    int first = sg.jlo;
    bool open = (sg.jlo <= sg.jhi) & !sg.exact;
    if (open) first = sg.jhi + 1;
    while (open) {
        float mv[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const float u = (float)min(j + q, sg.jhi);
            const float rowf = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra), colf = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
            const int i = (int)floorf(rowf) >> sh, c = (int)floorf(colf) >> sh;
            mv[q] = m2[min((uint32_t)((i + 1) * pitch + c + 1), maxidx)];      // one-cell border; the clamp never bites for a valid segment
        }
        bool found = false;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int jj = j + q;
            const float sk = fmaf((float)min(jj, sg.jhi), f.step, sg.sa);
            const float r2 = fmaf(sk, fmaf(sk, m.rq.a, two_b), m.rq.q0);
            const float rd = f.Rf * mv[q];
            const bool kept = !(r2 > (rd * rd) * 1.00001f);
            if (STATS) cnt[ST_MIP] += (!found && jj <= sg.jhi) ? 1u : 0u;
            if (!found && jj <= sg.jhi && kept) { first = jj; found = true; }
        }
        j += Q;
        open = !found && j <= sg.jhi;
    }
    return first;
}
// ONE 16-step segment of a march (the lanes that call it are still marching): anchors + skip interval, the steps
// that can be at/below the surface, the termination test at the segment end.  `hit` / `sk_hit` are set by the step
// that lands at/below the surface, `go` says whether the ray continues with the next segment.
template <bool WIDE, bool PRIMARY, bool STATS, int BATCH, int CP = 0, int SCAN = PRIMARY ? 1 : 0>
__device__ __forceinline__ void march_segment(const FrameC& f, MarchState& m, float smax, Seg& sg, bool& go, bool& hit,
                                              float& sk_hit, uint32_t* cnt) {
    const float oa = m.oa, ob = m.ob, oc = m.oc, da = m.da, db = m.db, dc = m.dc;
    const int ka = m.ka;
    float rowB, colB, q2B;
    if (!PRIMARY && (MRTX_HORIZON_RETRY & 1) != 0 && ka > 0 && m.kend >= f.kmax) {
        // The horizon bound was out of reach at the ray's origin (its ground track to the sphere exit is longer than the horizon
        // cell's dilation: low rays, and any east-west ray at high latitude, where columns shrink).  The ray has climbed since:
        // asked again from HERE, the remaining track is shorter and the bound may apply.
        const int ke = horizon_retry(f, m);
        if (ke < 1) {                        // above everything in reach already: the march ends before this segment
            if (STATS) cnt[ST_HEIGHT] += steps_after(f, m, ka + 1);
            go = false;
            return;
        }
        m.kend = min(f.kmax, ka + ke);       // ke == kmax: still unknown
    }
    PROF_BEGIN(6);
    seg_setup<STATS>(f, oa, ob, oc, da, db, dc, m.rq, ka, m.rowA, m.colA, m.q2A, sg, rowB, colB, q2B, cnt);
#ifdef MRTX_PROF_FULLIV   // measurement only: how many lanes get NO skip interval from the max-mip (footprint over more than 2 x 2 cells, or a true full interval)
    { const bool full = (sg.jlo == 1) & (sg.jhi == SEG_N) & !sg.exact;
      cnt[13] += (uint32_t)__popcll(__ballot(full)); cnt[14] += (uint32_t)__popcll(__ballot(sg.exact));
      cnt[5] += (uint32_t)__popcll(__ballot(full && sg.why == 1)); cnt[6] += (uint32_t)__popcll(__ballot(full && sg.why == 2));
      cnt[7] += (uint32_t)__popcll(__ballot(full && sg.why == 3)); }
#endif
    if (!PRIMARY) sg.jhi = max(min(sg.jhi, m.kend - ka), sg.jlo - 1);   // steps beyond kend cannot be at/below the surface
    // the medium max-mip cuts the interval once more (first_kept_step / last_kept_step above): camera rays from the front, shadow rays
    // from the end; MRTX_SEG_MASK bits 2 / 1 switch the two off (A/B)
    if (SCAN == 1 && (MRTX_SEG_MASK & 4) != 0 && CF(f)->mip2 != nullptr) sg.jlo = first_kept_step<STATS, MRTX_PMASK_Q>(f, m, sg, cnt);
    if (SCAN == 2 && (MRTX_SEG_MASK & 2) != 0 && CF(f)->mip2 != nullptr) sg.jhi = last_kept_step<STATS, MRTX_PMASK_Q>(f, m, sg, cnt);
    if (SCAN == 3 && (MRTX_SEG_MASK & 1) != 0 && CF(f)->mip2 != nullptr) sg.jhi = last_kept_step<STATS, MRTX_PMASK_Q>(f, m, sg, cnt);   // A/B: the trial segment
    PROF_END(6);
    PROF_BEGIN(7);
    if (STATS) cnt[ST_HEIGHT] += count_in_steps<PRIMARY>(f, oa, ob, oc, da, db, dc, smax, ka, 1, sg.jlo - 1);
    if (__ballot(sg.exact) != 0ull)
        step_loop<WIDE, PRIMARY, STATS, true, 1, CP>(f, oa, ob, oc, da, db, dc, smax, sg, ka, go, hit, sk_hit, cnt);
    else
        step_loop<WIDE, PRIMARY, STATS, false, BATCH, CP>(f, oa, ob, oc, da, db, dc, smax, sg, ka, go, hit, sk_hit, cnt);
    PROF_END(7);
#ifdef MRTX_PROF
#if !defined(MRTX_PROF_SPREAD) && !defined(MRTX_PROF_TRIAL) && !defined(MRTX_PROF_FULLIV)
    cnt[8] += 1;                                     // wave-level segments
    cnt[9] += (uint32_t)__popcll(__ballot(true));    // lanes alive in them
    cnt[PRIMARY ? 13 : 14] += (__ballot(sg.jlo <= sg.jhi) == 0ull) ? 1u : 0u;   // wave-level segments nobody steps in
    cnt[15] += PRIMARY ? 1u : 0u;
#endif
#ifdef MRTX_PROF_FULLIV
    cnt[8] += 1; cnt[9] += (uint32_t)__popcll(__ballot(true));
#endif
#endif
    segment_tail<PRIMARY, STATS>(f, m, smax, sg, go, rowB, colB, q2B, cnt);
}

// Coarse march s_k = k*step, k = 1, 2, ...; returns true and s_k at the first sample at/below the surface.
// PRIMARY: stop when s_k > smax (left the bounding sphere); shadow rays: stop when r^2 > R^2.
// A lane drops out of the exec mask when it hits or leaves, and the wave leaves the loop when no lane is still
// marching.  f.kmax is a multiple of SEG_N.
template <bool WIDE, bool PRIMARY, bool STATS, int BATCH, int SCAN = PRIMARY ? 1 : 0>
__device__ __forceinline__ bool march(const FrameC& f, float oa, float ob, float oc, float da, float db, float dc,
                                      float smax, Seg& sg, float& sk_hit, uint32_t* cnt) {
    MarchState m;
    bool hit = false;
    bool go = march_begin<PRIMARY, STATS>(f, oa, ob, oc, da, db, dc, m, cnt);
    while (go) march_segment<WIDE, PRIMARY, STATS, BATCH, 0, SCAN>(f, m, smax, sg, go, hit, sk_hit, cnt);
    return hit;
}

// D3 refinement: nbis bisections of (lo, hi) on below().  (Two levels per round -- the three mid-points evaluated
// together, 3 dependent rounds instead of 5 -- measured no faster: 13.90 vs 13.85 ms.)
template <bool WIDE>
__device__ __forceinline__ void refine(const FrameC& f, const Seg& sg, float oa, float ob, float oc, float da, float db,
                                       float dc, float& lo, float& hi) {
    auto below = [&](float s) {
        const float ma = fmaf(s, da, oa), mb = fmaf(s, db, ob), mc = fmaf(s, dc, oc);
        return below_seg<WIDE, true>(f, sg, s, ma, mb, mc, fmaf(mc, mc, fmaf(mb, mb, ma * ma)));
    };
    for (int i = 0; i < f.nbis; i++) {
        const float mid = 0.5f * (lo + hi);
        const bool bel = below(mid);
        hi = bel ? mid : hi;
        lo = bel ? lo : mid;
    }
}

struct Vertex {
    float pa, pb, pc;      // surface point (moon frame)
    float na, nb, nc;      // unit normal
    float al0, al1, al2;   // reflectance
};

// Duff et al., "Building an Orthonormal Basis, Revisited"
__device__ __forceinline__ void duff_basis(float na, float nb, float nc, float& b1a, float& b1b, float& b1c, float& b2a,
                                           float& b2b, float& b2c) {
    const float sg = nc >= 0.0f ? 1.0f : -1.0f;
    const float aa = -rcp_cr(sg + nc);       // |sg + nc| in [1, 2]; -(1/x) == (-1)/x bit for bit
    const float bb = (na * nb) * aa;
    b1a = fmaf(sg, (na * na) * aa, 1.0f); b1b = sg * bb; b1c = -sg * na;
    b2a = bb; b2b = fmaf(nb * nb, aa, sg); b2c = -nb;
}

// D7: nearest environment texel along a scene-frame direction
// ... in two halves, so that a caller can issue the fetch long before it needs the texel: the texel's index, and its decoding
__device__ __forceinline__ int64_t env_texel_index(const FrameC& f, float dx, float dy, float dz) {
    float el, az;
    latlon(dx, dy, dz, fmaf(dy, dy, dx * dx), el, az);
    const float rowf = fmaf(el, CF(f)->bg_row_scale, CF(f)->bg_row_off);
    const float colf = fmaf(az, CF(f)->bg_col_scale, CF(f)->bg_col_off);
    int r = (int)floorf(rowf), c = (int)floorf(colf);
    r = r < 0 ? 0 : (r > CF(f)->bg_h - 1 ? CF(f)->bg_h - 1 : r);
    if (c >= CF(f)->bg_w) c -= CF(f)->bg_w;
    if (c < 0) c = 0;
    return (int64_t)r * CF(f)->bg_w + c;
}
__device__ __forceinline__ void env_decode(uint32_t px, float& e0, float& e1, float& e2) {
    e0 = (float)(px & 255u) * kInv255;
    e1 = (float)((px >> 8) & 255u) * kInv255;
    e2 = (float)((px >> 16) & 255u) * kInv255;
}
template <bool STATS>
__device__ __forceinline__ void env_lookup(const FrameC& f, float dx, float dy, float dz, float& e0, float& e1, float& e2,
                                           uint32_t* cnt) {
    env_decode(reinterpret_cast<const uint32_t*>(CF(f)->bg)[env_texel_index(f, dx, dy, dz)], e0, e1, e2);
    if (STATS) cnt[ST_BG]++;
}
// moon-frame direction -> scene frame (the environment map is addressed in scene coordinates)
__device__ __forceinline__ void to_scene_dir(const FrameC& f, float bda, float bdb, float bdc, float& ex, float& ey, float& ez) {
    ex = fmaf(bdc, CF(f)->Mf[2][0], fmaf(bdb, CF(f)->Mf[1][0], bda * CF(f)->Mf[0][0]));
    ey = fmaf(bdc, CF(f)->Mf[2][1], fmaf(bdb, CF(f)->Mf[1][1], bda * CF(f)->Mf[0][1]));
    ez = fmaf(bdc, CF(f)->Mf[2][2], fmaf(bdb, CF(f)->Mf[1][2], bda * CF(f)->Mf[0][2]));
}

// surface point -> normal (central differences of D one texel either side of it) and albedo (D4)
template <bool STATS, bool WIDE>
__device__ __forceinline__ void hit_vertex(const FrameC& f, float ha, float hb, float hc, Vertex& v, uint32_t* cnt) {
    const float rho2 = fmaf(hb, hb, ha * ha);
    const float r2 = fmaf(hc, hc, rho2);
    const float rho = sqrt_sh(rho2);         // only used through rhoc = max(rho, 1e-6): a rho2 below 2^-104 cannot matter
    const float r = sqrt_sh(r2);             // r2 ~ R^2
    float lat, lon;
    latlon(ha, hb, hc, rho2, lat, lon);
    const float rowf = fmaf(lat, f.gd.row_scale, f.gd.row_off);
    const float colf = fmaf(lon, f.gd.col_scale, f.gd.col_off);
    // the two-texel border makes the +-1 texel taps plain two-load evaluations as well
    const float dn = dem_march<WIDE>(f, rowf - 1.0f, colf);
    const float ds = dem_march<WIDE>(f, rowf + 1.0f, colf);
    const float de = dem_march<WIDE>(f, rowf, colf + 1.0f);
    const float dw = dem_march<WIDE>(f, rowf, colf - 1.0f);
    if (STATS) { cnt[ST_HEIGHT] += 4; cnt[ST_FETCH] += 4; }
    const float dlat = (dn - ds) * CF(f)->dlat_scale;
    const float dlon = (de - dw) * CF(f)->dlon_scale;
    const float rhoc = rho > 1.0e-6f ? rho : 1.0e-6f;
    const float inv_r = rcp_cr(r), inv_rho = rcp_cr(rhoc);   // r ~ R, rhoc in [1e-6, R]
    const float sphi = hc * inv_r, cphi = rhoc * inv_r;
    const float slam = ha * inv_rho, clam = hb * inv_rho;
    const float glat = (f.Rf * inv_r) * dlat;
    const float glon = (f.Rf * inv_rho) * dlon;
    const float na = fmaf(-glon, clam, fmaf(glat, sphi * slam, ha * inv_r));
    const float nb = fmaf(glon, slam, fmaf(glat, sphi * clam, hb * inv_r));
    const float nc = fmaf(-glat, cphi, hc * inv_r);
    const float inv_nl = rcp_cr(sqrt_sh(fmaf(nc, nc, fmaf(nb, nb, na * na))));   // |n|^2 >= ~1 (unit radial part + gradient)
    v.pa = ha; v.pb = hb; v.pc = hc;
    v.na = na * inv_nl; v.nb = nb * inv_nl; v.nc = nc * inv_nl;
    if (CF(f)->color) {  // D4: bilinear RGBA8
        const float rc = fmaf(lat, CF(f)->gc.row_scale, CF(f)->gc.row_off);
        const float cc = fmaf(lon, CF(f)->gc.col_scale, CF(f)->gc.col_off);
        GridC gcl;   // scalar-load the colour grid constants (member-wise: no copy constructor across address spaces)
        gcl.h = CF(f)->gc.h; gcl.w = CF(f)->gc.w; gcl.row_scale = CF(f)->gc.row_scale; gcl.row_off = CF(f)->gc.row_off;
        gcl.col_scale = CF(f)->gc.col_scale; gcl.col_off = CF(f)->gc.col_off; gcl.wf = CF(f)->gc.wf;
        // row-pair layout as for the DEM (color_pair_kernel): element (r, c) = (T[max(r,0)][wrap(c)], T[min(r+1,h-1)][wrap(c)])
        // for r in [-1, h-1], c in [-2, w+1]: the 2x2 RGBA8 footprint is one 16-byte load instead of four gathers
        const float rfl = floorf(rc), cfl = floorf(cc);
        int32_t r0 = (int32_t)rfl, c0 = (int32_t)cfl;
        r0 = r0 < -1 ? -1 : (r0 > gcl.h - 1 ? gcl.h - 1 : r0);
        c0 = c0 < -2 ? -2 : (c0 > gcl.w ? gcl.w : c0);
        const float tfr = rc - rfl, tfc = cc - cfl;
        const uint64_t ci = (uint64_t)(uint32_t)(r0 + 1) * (uint64_t)(uint32_t)(gcl.w + 4) + (uint64_t)(uint32_t)(c0 + 2);
        const UQuad cq = *reinterpret_cast<const UQuad*>(reinterpret_cast<const char*>(CF(f)->color) + (ci << 3));
        const uint32_t p00 = cq.a, p10 = cq.b, p01 = cq.c, p11 = cq.d;
        v.al0 = lerp2((float)(p00 & 255u), (float)(p01 & 255u), (float)(p10 & 255u), (float)(p11 & 255u), tfr, tfc) * kInv255;
        v.al1 = lerp2((float)((p00 >> 8) & 255u), (float)((p01 >> 8) & 255u), (float)((p10 >> 8) & 255u),
                      (float)((p11 >> 8) & 255u), tfr, tfc) * kInv255;
        v.al2 = lerp2((float)((p00 >> 16) & 255u), (float)((p01 >> 16) & 255u), (float)((p10 >> 16) & 255u),
                      (float)((p11 >> 16) & 255u), tfr, tfc) * kInv255;
        if (STATS) cnt[ST_COLOUR]++;
    } else {
        v.al0 = CF(f)->const_albedo[0]; v.al1 = CF(f)->const_albedo[1]; v.al2 = CF(f)->const_albedo[2];
    }
}

// D5: one sample of the spherical light from a vertex: the shadow ray (origin lifted by scene_epsilon, direction
// uniform in the cone the light subtends) and what it carries if it arrives, radiance * solid angle / pi * cos(theta_i);
// false when the sampled direction lies below the surface (no shadow ray, no contribution).
__device__ __forceinline__ bool light_sample(const FrameC& f, const Vertex& v, float u2, float u3, float& oa, float& ob,
                                             float& oc, float& wa, float& wb, float& wc, float& carried) {
    const float eps = CF(f)->scene_eps;
    oa = fmaf(eps, v.na, v.pa); ob = fmaf(eps, v.nb, v.pb); oc = fmaf(eps, v.nc, v.pc);
    const float ta = CF(f)->Lb[0] - oa, tb = CF(f)->Lb[1] - ob, tc = CF(f)->Lb[2] - oc;
    const float d2 = fmaf(tc, tc, fmaf(tb, tb, ta * ta));
    const float inv_dist = rcp_cr(sqrt_sh(d2));   // distance to the light: ~2e4 R
    const float la = ta * inv_dist, lb = tb * inv_dist, lc = tc * inv_dist;
    float sin2 = CF(f)->rL2 * (inv_dist * inv_dist);
    if (sin2 > 1.0f) sin2 = 1.0f;
    const float cosmax = sqrt_sh(1.0f - sin2);     // 0 or >= 2^-24
    const float omc = sin2 / (1.0f + cosmax);
    const float av = u2 * omc;
    const float cost = 1.0f - av;
    const float sint = sqrt_sh(av * (2.0f - av));  // 0 (u2 = 0 or a point light) or >= ~2^-24 * omc
    float cph, sph;
    sincos_turn(u3, cph, sph);
    float b1a, b1b, b1c, b2a, b2b, b2c;
    duff_basis(la, lb, lc, b1a, b1b, b1c, b2a, b2b, b2c);
    const float ca = sint * cph, sa = sint * sph;
    wa = fmaf(cost, la, fmaf(sa, b2a, ca * b1a));
    wb = fmaf(cost, lb, fmaf(sa, b2b, ca * b1b));
    wc = fmaf(cost, lc, fmaf(sa, b2c, ca * b1c));
    const float cosi = fmaf(v.nc, wc, fmaf(v.nb, wb, v.na * wa));
    carried = (CF(f)->rad2 * omc) * cosi;
    return cosi > 0.0f;
}

// the light sample with its shadow ray marched through the same height field: carried radiance * visibility
template <bool STATS, bool WIDE, int BATCH>
__device__ __forceinline__ float direct_light(const FrameC& f, const Vertex& v, float u2, float u3, uint32_t* cnt) {
    float oa, ob, oc, wa, wb, wc, carried;
    if (!light_sample(f, v, u2, u3, oa, ob, oc, wa, wb, wc, carried)) return 0.0f;
    if (STATS) cnt[ST_SHADOW]++;
    Seg ssg;
    float sk_occ;
    if (march<WIDE, false, STATS, BATCH, 2>(f, oa, ob, oc, wa, wb, wc, 0.0f, ssg, sk_occ, cnt)) return 0.0f;
    return carried;
}

// D6: what happens to a path after the direct term of its vertex number `seg` (camera segment = 1): stop at
// path_seg_max, Russian roulette beyond path_seg_min, else the cosine-weighted continuation ray from the lifted
// vertex.  Returns false when the path ends; thr is the path throughput (updated).
__device__ __forceinline__ bool continue_path(const FrameC& f, const Vertex& v, uint32_t ks, uint32_t seg, float& t0r,
                                              float& t1r, float& t2r, float& boa, float& bob, float& boc, float& bda,
                                              float& bdb, float& bdc) {
    if (seg >= CF(f)->path_seg_max) return false;
    // segment seg+1: cosine-weighted direction about the normal => throughput *= albedo
    const uint32_t d0 = 4u + 5u * (seg - 1u);
    t0r *= v.al0; t1r *= v.al1; t2r *= v.al2;
    if (seg + 1u > CF(f)->path_seg_min) {   // Russian roulette beyond the guaranteed segments
        float pcont = v.al0 > v.al1 ? v.al0 : v.al1;
        pcont = pcont > v.al2 ? pcont : v.al2;
        pcont = pcont > 1.0f ? 1.0f : pcont;
        if (!(u01(ks, d0) < pcont)) return false;
        const float ip = rcp_cr(pcont);           // pcont in (0, 1]: a reflectance (u01 < pcont held, so pcont > 0), never subnormal
        t0r *= ip; t1r *= ip; t2r *= ip;
    }
    const float uh1 = u01(ks, d0 + 1u), uh2 = u01(ks, d0 + 2u);
    const float rr = sqrt_sh(uh1), zz = sqrt_sh(1.0f - uh1);   // uh1 = m * 2^-24: 0 or >= 2^-24
    float cph, sph;
    sincos_turn(uh2, cph, sph);
    float b1a, b1b, b1c, b2a, b2b, b2c;
    duff_basis(v.na, v.nb, v.nc, b1a, b1b, b1c, b2a, b2b, b2c);
    const float xx = rr * cph, yy = rr * sph;
    const float eps = CF(f)->scene_eps;
    boa = fmaf(eps, v.na, v.pa); bob = fmaf(eps, v.nb, v.pb); boc = fmaf(eps, v.nc, v.pc);
    bda = fmaf(zz, v.na, fmaf(yy, b2a, xx * b1a));
    bdb = fmaf(zz, v.nb, fmaf(yy, b2b, xx * b1b));
    bdc = fmaf(zz, v.nc, fmaf(yy, b2c, xx * b1c));
    return true;
}

// A continuation ray that left the Moon: the flat Sun-disk sphere IS visible to it (the reference keeps the light the
// disk bounces onto the Moon small through its radiance 2.0 and by parking it, moon_renderer.py:109-111, :757-760),
// then the environment texel along its direction; adds throughput x radiance.  Moon frame, float32: distance of the
// disk centre from the ray.
// PRE: the caller fetched the environment texel of this direction already (`pre_px`; render_kernel<MODE 2> issues the fetch before the
// trial segment so that its latency hides behind the march): same texel, same result, no load here
template <bool STATS, bool PRE = false>
__device__ __forceinline__ bool escaped_radiance(const FrameC& f, float boa, float bob, float boc, float bda, float bdb,
                                                 float bdc, float& e0, float& e1, float& e2, uint32_t* cnt, uint32_t pre_px = 0u) {
    if (CF(f)->sun_on) {
        const float sa = CF(f)->Sb[0] - boa, sb = CF(f)->Sb[1] - bob, sc = CF(f)->Sb[2] - boc;
        const float bq = fmaf(sc, bdc, fmaf(sb, bdb, sa * bda));
        const float qa = fmaf(-bq, bda, sa), qb = fmaf(-bq, bdb, sb), qc = fmaf(-bq, bdc, sc);
        const float d2 = fmaf(qc, qc, fmaf(qb, qb, qa * qa));
        if (bq > 0.0f && d2 < CF(f)->sun_r2) {
            e0 = e1 = e2 = CF(f)->sun_rad;
            if (STATS) cnt[ST_SUNHIT]++;
            return true;
        }
    }
    if (CF(f)->bg) {   // environment radiance along its direction (scene frame)
        if (PRE) {
            env_decode(pre_px, e0, e1, e2);
            if (STATS) cnt[ST_BG]++;
        } else {
            float ex, ey, ez;
            to_scene_dir(f, bda, bdb, bdc, ex, ey, ez);
            env_lookup<STATS>(f, ex, ey, ez, e0, e1, e2, cnt);
        }
        return true;
    }
    return false;
}
template <bool STATS, bool PRE = false>
__device__ __forceinline__ void escaped_path(const FrameC& f, float boa, float bob, float boc, float bda, float bdb,
                                             float bdc, float t0r, float t1r, float t2r, float& c0, float& c1,
                                             float& c2, uint32_t* cnt, uint32_t pre_px = 0u) {
    float e0, e1, e2;
    if (escaped_radiance<STATS, PRE>(f, boa, bob, boc, bda, bdb, bdc, e0, e1, e2, cnt, pre_px)) {
        c0 = fmaf(t0r, e0, c0); c1 = fmaf(t1r, e1, c1); c2 = fmaf(t2r, e2, c2);
    }
}

// D11: nearest overlay capsule of this tile's bin along the primary ray.  ro = ray point closest to the Moon
// centre (relative to the centre), d = unit direction; per capsule the origin is re-centred once more at the
// capsule's first endpoint so the quadratic stays well-conditioned for radii ~1e-2 at a 300-unit eye distance.
// The loop is wave-uniform (every lane walks the same bin), capsule data come through scalar loads.
// Only intersections IN FRONT OF THE EYE count (parameter > smin = the eye's parameter on this ray): a tube behind a camera
// that sits inside the shell of tubes must not shadow the one in view (the host's per-tile bins never hold it anyway).
__device__ __forceinline__ float nearest_capsule(const FrameC& f, int lt, float r0, float r1, float r2v, float dx,
                                                 float dy, float dz, float smin, int& which) {
    typedef const __attribute__((address_space(4))) float* CFloat;
    typedef const __attribute__((address_space(4))) int32_t* CInt;
    const CInt off = (CInt)CF(f)->caps_off;
    const CInt idx = (CInt)CF(f)->caps_idx;
    const CFloat caps = (CFloat)CF(f)->caps;
    float best = 1.0e30f;
    which = -1;
    const int i0 = off[lt], i1 = off[lt + 1];
    for (int i = i0; i < i1; i++) {
        const int k = idx[i];
        const CFloat c = caps + 12 * k;
        const float a0 = c[0], a1 = c[1], a2 = c[2], r = c[3];
        const float e0 = a0 - r0, e1 = a1 - r1, e2 = a2 - r2v;
        const float ta = fmaf(e2, dz, fmaf(e1, dy, e0 * dx));
        const float o0 = fmaf(ta, dx, -e0), o1 = fmaf(ta, dy, -e1), o2 = fmaf(ta, dz, -e2);
        const float ba0 = c[4] - a0, ba1 = c[5] - a1, ba2 = c[6] - a2;
        const float baba = fmaf(ba2, ba2, fmaf(ba1, ba1, ba0 * ba0));
        const float bard = fmaf(ba2, dz, fmaf(ba1, dy, ba0 * dx));
        const float baoa = fmaf(ba2, o2, fmaf(ba1, o1, ba0 * o0));
        const float rdoa = fmaf(dz, o2, fmaf(dy, o1, dx * o0));
        const float oaoa = fmaf(o2, o2, fmaf(o1, o1, o0 * o0));
        const float rr = r * r;
        const float A = fmaf(-bard, bard, baba);
        const float B = fmaf(baba, rdoa, -(baoa * bard));
        const float C = fmaf(baba, oaoa, -(baoa * baoa)) - rr * baba;
        const float h = fmaf(B, B, -(A * C));
        float cand = -1.0f;
        bool have = false;
        float y = baoa;
        if (A > 0.0f && h >= 0.0f) {
            const float t = (-B - sqrtf(h)) / A;
            y = fmaf(t, bard, baoa);
            if (y > 0.0f && y < baba) { cand = t; have = true; }
        }
        if (!have) {   // rounded end nearest to where the axis test left the segment
            const bool far_end = !(y <= 0.0f);
            const float q0 = far_end ? o0 - ba0 : o0, q1 = far_end ? o1 - ba1 : o1, q2 = far_end ? o2 - ba2 : o2;
            const float B2 = fmaf(dz, q2, fmaf(dy, q1, dx * q0));
            const float C2 = fmaf(q2, q2, fmaf(q1, q1, q0 * q0)) - rr;
            const float h2 = fmaf(B2, B2, -C2);
            if (h2 > 0.0f) { cand = -B2 - sqrtf(h2); have = true; }
        }
        if (have) {
            const float sc = ta + cand;
            if (sc > smin && sc < best) { best = sc; which = k; }
        }
    }
    return which >= 0 ? best : -1.0e30f;
}

// ---- render_kernel<MODE 2>, an experiment that is NOT the shipped order (round 3; kept as an A/B switch with its measurement):
// the shadow ray of the first vertex and the FIRST segment of its continuation ray marched TOGETHER.  Both rays leave the same
// point (vertex + scene_epsilon * normal: light_sample and continue_path build it with the same expression), so its exact texel
// coordinates and its horizon-mip cell are evaluated / fetched once; the two segment set-ups are issued back to back (both
// max-mip fetches in flight together) and the step loop evaluates the next steps of BOTH rays per iteration, so that the trial
// segment rides the memory rounds of the shadow march instead of adding ~4 dependent rounds of its own after it.  Per ray the
// evaluations, their order and every counter are those of march_begin_at + march_segment: bit-identical (tools/quick_parity.py).
// Measured at cfg3, S1, (2,4) (gpurun_out/r3b, profiles/r03_mode2_ab.md): render_kernel<MODE 2> 18.9 ms as shipped;
//   MRTX_FUSED_TRIAL=1: 22.7 ms (147 VGPRs: 3 waves per SIMD instead of 4), 23.6 ms when held to 128 VGPRs (16 spilled);
//   MRTX_FUSED_TRIAL=2 (only the origin's coordinates and horizon cell shared, marches in sequence): 19.0 ms -- 65 VALU per
//   sample less buy nothing;
// and the shipped kernel under an occupancy cap (tools/occ_sweep.sh): 3 waves per SIMD 23.4 ms, 4 (as shipped) 18.9 ms, while a
// launch bound of 5 / 6 waves (96 / 80 VGPRs, 22+ spilled) gives 22.2 / 23.3 ms.  The kernel is bound by the dependent-load
// rounds a wave goes through times the waves a SIMD can hold; two more live march states cost a wave slot, which is worth
// more than the four rounds the fusion hides.
#ifndef MRTX_FUSED_TRIAL
#define MRTX_FUSED_TRIAL 0     // 1 = fused first segment, 2 = shared origin only, 0 = one march after the other (ships)
#endif
#if MRTX_FUSED_TRIAL == 1
#if MRTX_WIDE_TAP
#error "fused_first_segment reads one max-mip element pair per ray: build it with -DMRTX_WIDE_TAP=0"
#endif
template <bool WIDE, bool STATS>
__device__ __forceinline__ void fused_first_segment(const FrameC& f, MarchState& ms, MarchState& mt, Seg& sgs, Seg& sgt,
                                                    bool& go_s, bool& go_t, bool& hit_s, bool& hit_t, float& sk_s, float& sk_t,
                                                    uint32_t* cnt) {
    constexpr int B = 2;
    float rBs, cBs, qBs, rBt, cBt, qBt;
    MipTap tap_s, tap_t;
    seg_anchors(f, ms.oa, ms.ob, ms.oc, ms.da, ms.db, ms.dc, ms.ka, ms.rowA, ms.colA, ms.q2A, sgs, rBs, cBs, qBs, tap_s);
    seg_anchors(f, mt.oa, mt.ob, mt.oc, mt.da, mt.db, mt.dc, mt.ka, mt.rowA, mt.colA, mt.q2A, sgt, rBt, cBt, qBt, tap_t);
    tap_s.usable &= go_s; tap_t.usable &= go_t;
    if (f.mip != nullptr) {                                  // wave-uniform
        tap_s.off = tap_s.usable ? tap_s.off : 0u; tap_t.off = tap_t.usable ? tap_t.off : 0u;
        const Quad qs = mip_fetch(f, tap_s), qt = mip_fetch(f, tap_t);     // unconditional: both in flight together
        if (tap_s.usable) seg_interval<STATS>(f, ms.rq, sgs, tap_s, qs, cnt);
        if (tap_t.usable) seg_interval<STATS>(f, mt.rq, sgt, tap_t, qt, cnt);
    }
    sgs.jhi = max(min(sgs.jhi, ms.kend - ms.ka), sgs.jlo - 1);   // steps beyond kend cannot be at/below the surface
    sgt.jhi = max(min(sgt.jhi, mt.kend - mt.ka), sgt.jlo - 1);
    if (STATS) {
        if (go_s) cnt[ST_HEIGHT] += count_in_steps<false>(f, ms.oa, ms.ob, ms.oc, ms.da, ms.db, ms.dc, 0.0f, ms.ka, 1, sgs.jlo - 1);
        if (go_t) cnt[ST_HEIGHT] += count_in_steps<false>(f, mt.oa, mt.ob, mt.oc, mt.da, mt.db, mt.dc, 0.0f, mt.ka, 1, sgt.jlo - 1);
    }
    int js = sgs.jlo, jt = sgt.jlo;
    bool more_s = go_s & (js <= sgs.jhi), more_t = go_t & (jt <= sgt.jhi);
    if (__ballot((go_s & sgs.exact) | (go_t & sgt.exact)) == 0ull) {
        // both rays step while both have lanes stepping; what is left of either one afterwards runs in its own loop below
        while (__ballot(more_s) != 0ull && __ballot(more_t) != 0ull) {
            float surf_s[B], surf_t[B];
#pragma unroll
            for (int i = 0; i < B; i++) {
                // steps past jhi are evaluated at the segment's last step instead (inside the quadratic's range), as in step_loop
                const float us = ((float)(ms.ka + min(js + i, SEG_N)) * f.step - sgs.sa) * f.inv_step;
                surf_s[i] = f.Rf * dem_march<WIDE>(f, fmaf(us, fmaf(us, sgs.r2, sgs.r1), sgs.ra), fmaf(us, fmaf(us, sgs.c2, sgs.c1), sgs.ca));
                const float ut = ((float)(mt.ka + min(jt + i, SEG_N)) * f.step - sgt.sa) * f.inv_step;
                surf_t[i] = f.Rf * dem_march<WIDE>(f, fmaf(ut, fmaf(ut, sgt.r2, sgt.r1), sgt.ra), fmaf(ut, fmaf(ut, sgt.c2, sgt.c1), sgt.ca));
            }
            bool act = more_s;
#pragma unroll
            for (int i = 0; i < B; i++) {
                const int k = ms.ka + js + i;
                const float sk = (float)k * f.step;
                const float pa = fmaf(sk, ms.da, ms.oa), pb = fmaf(sk, ms.db, ms.ob), pc = fmaf(sk, ms.dc, ms.oc);
                const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
                const bool in = (r2 <= f.R2f) & (k <= f.kmax);
                const bool bel = r2 <= surf_s[i] * surf_s[i];
                if (STATS) { cnt[ST_HEIGHT] += (act & in) ? 1u : 0u; cnt[ST_FETCH] += act ? 1u : 0u; }
                hit_s = act ? (in & bel) : hit_s;
                go_s = act ? (in & !bel) : go_s;
                sk_s = act ? sk : sk_s;
                act = act & go_s & (js + i + 1 <= sgs.jhi);
            }
            js += B; more_s = act;
            act = more_t;
#pragma unroll
            for (int i = 0; i < B; i++) {
                const int k = mt.ka + jt + i;
                const float sk = (float)k * f.step;
                const float pa = fmaf(sk, mt.da, mt.oa), pb = fmaf(sk, mt.db, mt.ob), pc = fmaf(sk, mt.dc, mt.oc);
                const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
                const bool in = (r2 <= f.R2f) & (k <= f.kmax);
                const bool bel = r2 <= surf_t[i] * surf_t[i];
                if (STATS) { cnt[ST_HEIGHT] += (act & in) ? 1u : 0u; cnt[ST_FETCH] += act ? 1u : 0u; }
                hit_t = act ? (in & bel) : hit_t;
                go_t = act ? (in & !bel) : go_t;
                sk_t = act ? sk : sk_t;
                act = act & go_t & (jt + i + 1 <= sgt.jhi);
            }
            jt += B; more_t = act;
        }
        if (__ballot(more_s) != 0ull)
            step_loop_from<WIDE, false, STATS, false, MRTX_STEP_BATCH>(f, ms.oa, ms.ob, ms.oc, ms.da, ms.db, ms.dc, 0.0f, sgs, ms.ka, js, more_s, go_s, hit_s, sk_s, cnt);
        if (__ballot(more_t) != 0ull)
            step_loop_from<WIDE, false, STATS, false, MRTX_TRIAL_BATCH>(f, mt.oa, mt.ob, mt.oc, mt.da, mt.db, mt.dc, 0.0f, sgt, mt.ka, jt, more_t, go_t, hit_t, sk_t, cnt);
    } else {                                                 // a lane needs exact texel coordinates at every step (seam / pole)
        step_loop_from<WIDE, false, STATS, true, 1>(f, ms.oa, ms.ob, ms.oc, ms.da, ms.db, ms.dc, 0.0f, sgs, ms.ka, js, more_s, go_s, hit_s, sk_s, cnt);
        step_loop_from<WIDE, false, STATS, true, 1>(f, mt.oa, mt.ob, mt.oc, mt.da, mt.db, mt.dc, 0.0f, sgt, mt.ka, jt, more_t, go_t, hit_t, sk_t, cnt);
    }
    segment_tail<false, STATS>(f, ms, 0.0f, sgs, go_s, rBs, cBs, qBs, cnt);
    segment_tail<false, STATS>(f, mt, 0.0f, sgt, go_t, rBt, cBt, qBt, cnt);
}

#endif   // MRTX_FUSED_TRIAL == 1

// ---- MODE 3 (sky-only tiles, environment map bound): the pixel-uniform shortcut.
// A 4K pixel of the default view subtends 1/11 of a texel of the reference's 16k star map, so the 64 samples of most sky pixels
// read ONE texel.  env_rowcol() evaluates the (row, col) a sample at image-plane position (fx, fy) gets -- the very arithmetic of
// trace_sample + env_lookup up to the floor -- and sky_pixel_uniform() does that at the four corners of a pixel (every jittered
// position fx = x + u0 lies in [x, x + 1], corners included): when the corners' coordinates, widened by a margin that covers the
// rounding of the per-sample evaluation (~1.6e-7 x map width texels) and the curvature of the mapping inside the pixel
// (<= tan(el) theta_px^2, < 2e-3 texels away from the poles, where the corners disagree anyway), all fall into one texel away
// from the map's seam and edges, every sample of the pixel reads that texel.  The S samples of a block then add up to S x texel
// exactly (a pairwise tree over S equal floats), so the pixel needs one look-up instead of S.  Result-preserving: radiance and
// counters equal the per-sample evaluation (tests/test_gpu_paths.py, the fuzz's environment cases).
__device__ __forceinline__ void env_rowcol(const FrameC& f, float fx, float fy, float& rowf, float& colf) {
    const float sx = fmaf(fx, CF(f)->two_over_w, -1.0f);
    const float sy = fmaf(-fy, CF(f)->two_over_h, 1.0f);
    float dx = fmaf(sy, CF(f)->Vy[0], fmaf(sx, CF(f)->Ux[0], CF(f)->Wd[0]));
    float dy = fmaf(sy, CF(f)->Vy[1], fmaf(sx, CF(f)->Ux[1], CF(f)->Wd[1]));
    float dz = fmaf(sy, CF(f)->Vy[2], fmaf(sx, CF(f)->Ux[2], CF(f)->Wd[2]));
    const float inv_len = rcp_cr(sqrt_sh(fmaf(dz, dz, fmaf(dy, dy, dx * dx))));   // |Wd + sx Ux + sy Vy|^2 in [1, 1 + tan^2]
    dx = dx * inv_len; dy = dy * inv_len; dz = dz * inv_len;
    float el, az;
    latlon(dx, dy, dz, fmaf(dy, dy, dx * dx), el, az);
    rowf = fmaf(el, CF(f)->bg_row_scale, CF(f)->bg_row_off);
    colf = fmaf(az, CF(f)->bg_col_scale, CF(f)->bg_col_off);
}
__device__ __forceinline__ bool sky_pixel_uniform(const FrameC& f, int x, int y, float& e0, float& e1, float& e2) {
    float r00, c00, r10, c10, r01, c01, r11, c11;
    const float fx = (float)x, fy = (float)y;
    env_rowcol(f, fx, fy, r00, c00);
    env_rowcol(f, fx + 1.0f, fy, r10, c10);
    env_rowcol(f, fx, fy + 1.0f, r01, c01);
    env_rowcol(f, fx + 1.0f, fy + 1.0f, r11, c11);
    const float m = fmaf(1.0e-6f, (float)max(CF(f)->bg_w, CF(f)->bg_h), 0.02f);
    const float rlo = fminf(fminf(r00, r10), fminf(r01, r11)) - m, rhi = fmaxf(fmaxf(r00, r10), fmaxf(r01, r11)) + m;
    const float clo = fminf(fminf(c00, c10), fminf(c01, c11)) - m, chi = fmaxf(fmaxf(c00, c10), fmaxf(c01, c11)) + m;
    const float rf = floorf(rlo), cf = floorf(clo);
    // one texel, no clamp (rows), no wrap (columns); NaNs compare false
    const bool uni = rf == floorf(rhi) && cf == floorf(chi) && rlo >= 0.0f && clo >= 0.0f &&
                     rhi < (float)CF(f)->bg_h && chi < (float)CF(f)->bg_w;
    e0 = e1 = e2 = 0.0f;
    if (uni) {
        const uint32_t px = reinterpret_cast<const uint32_t*>(CF(f)->bg)[(int64_t)(int)rf * CF(f)->bg_w + (int)cf];
        e0 = (float)(px & 255u) * kInv255;
        e1 = (float)((px >> 8) & 255u) * kInv255;
        e2 = (float)((px >> 16) & 255u) * kInv255;
    }
    return uni;
}

// ---- LDS as the spill space of render_kernel<MODE 2> (round 4, MRTX_LDS_PARK).
// The kernel's time follows the waves a SIMD holds (1 -> 2 -> 3 -> 4 waves: 52 -> 29 -> 23 -> 19 ms, profiles/r03_mode2_ab.md) and the
// fifth wave needs <= 96 VGPRs; the compiler's own 96-VGPR code spills 17 registers to scratch and loses its load clustering.  What
// is COLD while a march runs -- the first vertex (point, normal, albedo: 9), the sample's RNG key and the radiance the light sample
// carries across the shadow march; the running radiance, throughput and key across the trial segment -- is parked in LDS instead:
// lane-private 16-byte slots (ds_write_b128 / ds_read_b128, consecutive lanes = consecutive 16-byte slots: conflict-free), three
// slots = 3 KB per wave, 60 KB per CU at 20 waves.  The values come back bit for bit, so nothing in the arithmetic changes.
// A compiler-level memory barrier on either side keeps LLVM from forwarding the stored values to the loads (which would keep them
// in registers after all).
#ifndef MRTX_LDS_PARK
#define MRTX_LDS_PARK 1
#endif
#ifndef MRTX_PARK_SHARE_ORIGIN
#define MRTX_PARK_SHARE_ORIGIN 1     // the shadow ray and the continuation ray leave the SAME point (vertex + scene_epsilon * normal): its exact
#endif                               // texel coordinates and horizon cell are evaluated once and sit out the shadow march in a fourth slot
#define MRTX_PARK_SLOTS (MRTX_PARK_SHARE_ORIGIN ? 4 : 3)
#ifndef MRTX_WG_WAVES
#define MRTX_WG_WAVES 1       // waves per workgroup of render_kernel (see there)
#endif
__device__ __forceinline__ void park_put(v4f* park, int slot, float a, float b, float c, float d) {
    const v4f v = {a, b, c, d};
    park[slot * (64 * MRTX_WG_WAVES) + threadIdx.x] = v;
}
__device__ __forceinline__ v4f park_get(const v4f* park, int slot) { return park[slot * (64 * MRTX_WG_WAVES) + threadIdx.x]; }
__device__ __forceinline__ void park_fence() { asm volatile("" ::: "memory"); }

struct SampleOut {
    float c0, c1, c2, hitflag;
    float h0, h1, h2, h3;
    // DEFER: the continuation ray of the path (origin, direction, throughput, exact texel coordinates of the origin)
    // and the sample's RNG key, for path_kernel
    float oa, ob, oc, da, db, dc, t0, t1, t2, row, col;
    uint32_t ks, aux;   // aux: MRTX_REC_RESUME | kend << 8, MRTX_REC_HIT | k << 8, or 0 (PathQ::lane_of)
    bool path;
};

// MODE: 0 = direct light only, 1 = the whole path inside this wave (BOUNCE), 2 = direct light + hand-over of the
// vertex to path_kernel (DEFER), 3 = SKY: the host has proven that no sample of this tile can meet the Moon's bounding
// sphere, the Sun disk or an overlay tube (cull_tiles), so a sample is its environment texel and nothing else
template <bool STATS, bool WIDE, int MODE, bool OVERLAY>
__device__ __forceinline__ void trace_sample(const FrameC& f, int lt, int x, int y, uint32_t gs, SampleOut& o,
                                             uint32_t* cnt, bool lead = false, v4f* park = nullptr) {
    constexpr bool BOUNCE = MODE == 1, DEFER = MODE == 2;
    constexpr bool PARK = DEFER && MRTX_LDS_PARK != 0;
    // (writing the hit record from here, as soon as it is known, instead of carrying it to the end of the wave was measured for the
    // direct kernel: 13.33 ms against 13.21 -- its single late store stays.  render_kernel<MODE 2> with MRTX_LDS_PARK is held to 96
    // VGPRs, where four registers carried through three marches do count: its launches hold ONE block, so the record of sample 0
    // (`lead`) IS the frame's hit record and goes to memory right here)
    auto emit_hit = [&](float h0, float h1, float h2, float h3) {
        if (PARK) {
            if (lead) reinterpret_cast<float4*>(CF(f)->hits)[(int64_t)y * f.W + x] = make_float4(h0, h1, h2, h3);
        } else {
            o.h0 = h0; o.h1 = h1; o.h2 = h2; o.h3 = h3;
        }
    };
    constexpr int BATCH = BOUNCE ? MRTX_STEP_BATCH_BOUNCE : MRTX_STEP_BATCH;   // incoherent bounce rays waste the speculative fetches
    const uint32_t pix = (uint32_t)y * (uint32_t)f.W + (uint32_t)x;
    const uint32_t kp = mix32(pix + CF(f)->key0);
    const uint32_t ks = mix32(kp ^ (gs * 0x85EBCA6Bu + 1u));
    const float u0 = u01(ks, 0), u1 = u01(ks, 1), u2 = u01(ks, 2), u3 = u01(ks, 3);
    o.c0 = o.c1 = o.c2 = 0.0f; o.hitflag = 0.0f;
    o.h0 = o.h1 = o.h2 = o.h3 = 0.0f;
    o.path = false;
    // the hand-over fields mean something only while o.path is set: (re)initialised HERE, per sample, they are constants until
    // a path assigns them -- initialised once outside the caller's block loop they were loop-carried values, thirteen registers
    // held through every march of the sample for nothing
    if (DEFER) { o.oa = o.ob = o.oc = o.da = o.db = o.dc = o.t0 = o.t1 = o.t2 = o.row = o.col = 0.0f; o.ks = 0u; o.aux = 0u; }
    if (STATS) cnt[ST_PRIMARY]++;
    PROF_BEGIN(1);

    // D1: jittered pinhole ray
    const float fx = (float)x + u0, fy = (float)y + u1;
    const float sx = fmaf(fx, CF(f)->two_over_w, -1.0f);
    const float sy = fmaf(-fy, CF(f)->two_over_h, 1.0f);
    float dx = fmaf(sy, CF(f)->Vy[0], fmaf(sx, CF(f)->Ux[0], CF(f)->Wd[0]));
    float dy = fmaf(sy, CF(f)->Vy[1], fmaf(sx, CF(f)->Ux[1], CF(f)->Wd[1]));
    float dz = fmaf(sy, CF(f)->Vy[2], fmaf(sx, CF(f)->Ux[2], CF(f)->Wd[2]));
    const float inv_len = rcp_cr(sqrt_sh(fmaf(dz, dz, fmaf(dy, dy, dx * dx))));   // |Wd + sx Ux + sy Vy|^2 in [1, 1 + tan^2]
    dx = dx * inv_len; dy = dy * inv_len; dz = dz * inv_len;
    if (MODE == 3) {
        if (CF(f)->bg) env_lookup<STATS>(f, dx, dy, dz, o.c0, o.c1, o.c2, cnt);  // D7
        emit_hit(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }

    // float64 entry into the bounding sphere: the eye sits ~30 radii away, float32 would cost metres
    const double Dx = (double)dx, Dy = (double)dy, Dz = (double)dz;
    const double a = (Dx * Dx + Dy * Dy) + Dz * Dz;
    const double b = (CF(f)->oc[0] * Dx + CF(f)->oc[1] * Dy) + CF(f)->oc[2] * Dz;
    const double disc = b * b - a * CF(f)->cq;
    bool on_sphere = false;
    double t0 = 0.0, t1 = 0.0;
    if (disc > 0.0) {
        // products/sums in float64; the root and the 1/a scale in float32 (f64 sqrt and division are ~35
        // instructions each): their rounding moves the entry point along the ray only
        const double sq = (double)sqrtf((float)disc);
        const double inva = (double)rcp_cr((float)a);      // a = |d|^2 = 1 +- rounding
        t0 = (-b - sq) * inva;
        t1 = (-b + sq) * inva;
        if (t1 > 0.0) { on_sphere = true; if (t0 < 0.0) t0 = 0.0; }
    }
    // D11: overlay tubes live outside the bounding sphere (r = 1.025 R, moon_grid.py:188): one wins the sample if it
    // is hit in front of the sphere entry, or if the ray finds no terrain at all
    int cap = -1;
    float cap_s = 0.0f, cr0 = 0.0f, cr1 = 0.0f, cr2 = 0.0f;
    double tc0 = 0.0;
    bool cap_front = false;
    if (OVERLAY) {
        tc0 = -b * (double)rcp_cr((float)a);
        cr0 = (float)(CF(f)->oc[0] + tc0 * Dx); cr1 = (float)(CF(f)->oc[1] + tc0 * Dy); cr2 = (float)(CF(f)->oc[2] + tc0 * Dz);
        cap_s = nearest_capsule(f, lt, cr0, cr1, cr2, dx, dy, dz, (float)(-tc0), cap);   // nearest one in front of the eye
        cap_front = cap >= 0 && (!on_sphere || cap_s < (float)(t0 - tc0));
    }
    bool hit = false;
    float pa = 0.f, pb = 0.f, pc = 0.f, da = 0.f, db = 0.f, dc = 0.f, lo = 0.0f;
    PROF_END(1);
    if (on_sphere && !cap_front) {
        const double pe0 = CF(f)->oc[0] + t0 * Dx, pe1 = CF(f)->oc[1] + t0 * Dy, pe2 = CF(f)->oc[2] + t0 * Dz;
        pa = (float)((CF(f)->M[0][0] * pe0 + CF(f)->M[0][1] * pe1) + CF(f)->M[0][2] * pe2);
        pb = (float)((CF(f)->M[1][0] * pe0 + CF(f)->M[1][1] * pe1) + CF(f)->M[1][2] * pe2);
        pc = (float)((CF(f)->M[2][0] * pe0 + CF(f)->M[2][1] * pe1) + CF(f)->M[2][2] * pe2);
        da = (float)((CF(f)->M[0][0] * Dx + CF(f)->M[0][1] * Dy) + CF(f)->M[0][2] * Dz);
        db = (float)((CF(f)->M[1][0] * Dx + CF(f)->M[1][1] * Dy) + CF(f)->M[1][2] * Dz);
        dc = (float)((CF(f)->M[2][0] * Dx + CF(f)->M[2][1] * Dy) + CF(f)->M[2][2] * Dz);
        const float smax = (float)(t1 - t0);
        Seg sg;
        float hi = 0.0f;
        PROF_BEGIN(2);
        hit = march<WIDE, true, STATS, BATCH>(f, pa, pb, pc, da, db, dc, smax, sg, hi, cnt);
        PROF_END(2);
        PROF_BEGIN(3);
#ifdef MRTX_PROF_SPREAD   // measurement only: spread of the lanes' texel coordinates at the primary hit (would an LDS tile cover the wave?)
        {
            const float u = (hi - sg.sa) * f.inv_step;
            float rw = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra), cl = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
            float rmin = hit ? rw : 1e30f, rmax = hit ? rw : -1e30f, cmin = hit ? cl : 1e30f, cmax = hit ? cl : -1e30f;
            for (int m = 1; m < 64; m <<= 1) {
                rmin = fminf(rmin, __shfl_xor(rmin, m, 64)); rmax = fmaxf(rmax, __shfl_xor(rmax, m, 64));
                cmin = fminf(cmin, __shfl_xor(cmin, m, 64)); cmax = fmaxf(cmax, __shfl_xor(cmax, m, 64));
            }
            const float er = rmax - rmin, ec = cmax - cmin;
            if (__ballot(hit) != 0ull) {
                const float e = fmaxf(er, ec);
                cnt[13] += (e <= 8.0f) ? 1u : 0u;
                cnt[14] += (e <= 16.0f) ? 1u : 0u;
                cnt[15] += (e <= 28.0f) ? 1u : 0u;
                cnt[9] += 1u;
            }
        }
#endif
        if (hit) {
            // hi = (float)k * step of the first sample below; k recovered exactly (|k*step/step - k| << 0.5)
            const int k = (int)rintf(hi * f.inv_step);
            lo = (float)(k - 1) * f.step;
            refine<WIDE>(f, sg, pa, pb, pc, da, db, dc, lo, hi);
            if (STATS) { cnt[ST_HEIGHT] += (uint32_t)f.nbis; cnt[ST_FETCH] += (uint32_t)f.nbis; }
        }
        PROF_END(3);
    }

    if (OVERLAY && !hit && cap >= 0) {
        typedef const __attribute__((address_space(4))) float* CFloat;
        const CFloat c = (CFloat)CF(f)->caps + 12 * cap;
        o.c0 = c[8]; o.c1 = c[9]; o.c2 = c[10];
        o.hitflag = 1.0f;
        emit_hit(CF(f)->centerf[0] + fmaf(cap_s, dx, cr0), CF(f)->centerf[1] + fmaf(cap_s, dy, cr1),
                 CF(f)->centerf[2] + fmaf(cap_s, dz, cr2), (float)tc0 + cap_s);
        return;
    }
    if (!hit) {
        if (CF(f)->sun_on) {  // D8
            const float bq = fmaf(CF(f)->sc[2], dz, fmaf(CF(f)->sc[1], dy, CF(f)->sc[0] * dx));
            const float dq = fmaf(bq, bq, -CF(f)->sun_cq);
            if (bq > 0.0f && dq > 0.0f) {
                const float t = bq - sqrtf(dq);
                o.c0 = o.c1 = o.c2 = CF(f)->sun_rad;
                o.hitflag = 1.0f;
                emit_hit(fmaf(t, dx, CF(f)->eyef[0]), fmaf(t, dy, CF(f)->eyef[1]), fmaf(t, dz, CF(f)->eyef[2]), t);
                return;
            }
        }
        if (CF(f)->bg) env_lookup<STATS>(f, dx, dy, dz, o.c0, o.c1, o.c2, cnt);  // D7
        emit_hit(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }

    // ---- the path: vertex 1 is the primary hit; BOUNCE continues it (D6, set_uint("path_seg_range", min, max))
    if (STATS) cnt[ST_HITS]++;
    Vertex v;
    PROF_BEGIN(4);
    hit_vertex<STATS, WIDE>(f, fmaf(lo, da, pa), fmaf(lo, db, pb), fmaf(lo, dc, pc), v, cnt);
    PROF_END(4);
    o.hitflag = 1.0f;
    emit_hit(CF(f)->centerf[0] + fmaf(v.pc, CF(f)->Mf[2][0], fmaf(v.pb, CF(f)->Mf[1][0], v.pa * CF(f)->Mf[0][0])),
             CF(f)->centerf[1] + fmaf(v.pc, CF(f)->Mf[2][1], fmaf(v.pb, CF(f)->Mf[1][1], v.pa * CF(f)->Mf[0][1])),
             CF(f)->centerf[2] + fmaf(v.pc, CF(f)->Mf[2][2], fmaf(v.pb, CF(f)->Mf[1][2], v.pa * CF(f)->Mf[0][2])),
             (float)t0 + lo);

    float t0r = 1.0f, t1r = 1.0f, t2r = 1.0f;   // path throughput
    uint32_t seg = 1;
    float ul1 = u2, ul2 = u3;
#if MRTX_FUSED_TRIAL && MRTX_TRIAL_SEGMENT && !defined(MRTX_PROF)
    if (DEFER) {
        // Vertex 1: its light sample's shadow ray and the first segment of its continuation ray share their origin and are
        // marched together (fused_first_segment); then the rest of the shadow march.  Same evaluations, results and counters
        // as direct_light() followed by the trial segment below (MRTX_FUSED_TRIAL = 0).
        float so_a, so_b, so_c, sw_a, sw_b, sw_c, carried = 0.0f;
        const bool have_s = light_sample(f, v, ul1, ul2, so_a, so_b, so_c, sw_a, sw_b, sw_c, carried);
        if (STATS && have_s) cnt[ST_SHADOW]++;
        const float k0 = t0r * v.al0, k1 = t1r * v.al1, k2 = t2r * v.al2;      // the direct term's weights: throughput BEFORE the bounce
        const bool have_c = continue_path(f, v, ks, 1u, t0r, t1r, t2r, o.oa, o.ob, o.oc, o.da, o.db, o.dc);
        if (STATS && have_c) cnt[ST_BOUNCE]++;
        float row0, col0, q2o;
        exact_rowcol(f, so_a, so_b, so_c, row0, col0, q2o);                     // ONE evaluation for both rays' origin
        const bool hm_on = CF(f)->hmip != nullptr;                              // wave-uniform
        const float cell = hm_on ? horizon_cell(f, row0, col0) : 0.0f;         // ... and one horizon-mip look-up
        MarchState ms, mt;
        ms.rowA = mt.rowA = row0; ms.colA = mt.colA = col0;
        // an absent ray is a ray of length zero from the same origin: its set-up arithmetic stays finite and is discarded
        // (b = 0 and a = 0 make horizon_kend's chord bound infinite, so it never cuts and never counts anything)
        bool go_s = march_begin_at<false, STATS, false, true>(f, so_a, so_b, so_c, have_s ? sw_a : 0.0f, have_s ? sw_b : 0.0f,
                                                              have_s ? sw_c : 0.0f, ms, cnt, cell) & have_s;
        bool go_t = march_begin_at<false, STATS, false, true>(f, so_a, so_b, so_c, have_c ? o.da : 0.0f, have_c ? o.db : 0.0f,
                                                              have_c ? o.dc : 0.0f, mt, cnt, cell) & have_c;
        bool hit_s = false, hit_t = false;
        float sk_s = 0.0f, sk_t = 0.0f;
        Seg sgs, sgt;
#if MRTX_FUSED_TRIAL == 2     // A/B: shared origin coordinates and horizon cell only, the two marches one after the other
        while (go_s) march_segment<WIDE, false, STATS, BATCH, 0, 2>(f, ms, 0.0f, sgs, go_s, hit_s, sk_s, cnt);
        if (go_t) march_segment<WIDE, false, STATS, MRTX_TRIAL_BATCH>(f, mt, 0.0f, sgt, go_t, hit_t, sk_t, cnt);
#else
        const bool any_s = __ballot(go_s) != 0ull, any_t = __ballot(go_t) != 0ull;
        if (any_s && any_t) {
            fused_first_segment<WIDE, STATS>(f, ms, mt, sgs, sgt, go_s, go_t, hit_s, hit_t, sk_s, sk_t, cnt);
        } else if (any_t) {
            if (go_t) march_segment<WIDE, false, STATS, MRTX_TRIAL_BATCH>(f, mt, 0.0f, sgt, go_t, hit_t, sk_t, cnt);
        }
        while (go_s) march_segment<WIDE, false, STATS, BATCH, 0, 2>(f, ms, 0.0f, sgs, go_s, hit_s, sk_s, cnt);
#endif
        const float wgt = (have_s && !hit_s) ? carried : 0.0f;
        o.c0 = fmaf(k0, wgt, o.c0);
        o.c1 = fmaf(k1, wgt, o.c1);
        o.c2 = fmaf(k2, wgt, o.c2);
        if (have_c) {
            o.row = row0; o.col = col0;
            o.t0 = t0r; o.t1 = t1r; o.t2 = t2r;
            o.ks = ks; o.aux = 0u;
            if (!go_t && !hit_t) {
                escaped_path<STATS>(f, o.oa, o.ob, o.oc, o.da, o.db, o.dc, t0r, t1r, t2r, o.c0, o.c1, o.c2, cnt);
            } else {
                o.path = true;
                if (hit_t) {
                    o.aux = MRTX_REC_HIT | ((uint32_t)(int)rintf(sk_t * f.inv_step) << 8);
                } else {
                    o.aux = MRTX_REC_RESUME | ((uint32_t)mt.kend << 8);
                    o.row = mt.rowA; o.col = mt.colA;
                }
            }
        }
        return;
    }
#endif
    uint32_t ksl = ks;      // the sample's RNG key as the continuation uses it (PARK: the copy that came back from LDS)
    float org_row = 0.0f, org_col = 0.0f, org_cell = 0.0f;   // PARK + MRTX_PARK_SHARE_ORIGIN: the lifted vertex on the DEM grid
    for (;;) {
        PROF_BEGIN(5);
        float wgt;
        if (PARK) {
            // direct_light() with the vertex, the key and the carried radiance parked in LDS while the shadow ray marches
            float oa, ob, oc, wa, wb, wc, carried = 0.0f;
            const bool have_s = light_sample(f, v, ul1, ul2, oa, ob, oc, wa, wb, wc, carried);
#if MRTX_PARK_SHARE_ORIGIN
            // the origin's exact texel coordinates and horizon-mip cell: ONE evaluation / look-up for the shadow ray and, below,
            // for the continuation ray (continue_path builds its origin with the very expression light_sample uses)
            float q2o;
            exact_rowcol(f, oa, ob, oc, org_row, org_col, q2o);
            org_cell = CF(f)->hmip != nullptr ? horizon_cell(f, org_row, org_col) : 0.0f;
            park_put(park, 3, org_row, org_col, org_cell, 0.0f);
#endif
            park_put(park, 0, v.pa, v.pb, v.pc, v.na);
            park_put(park, 1, v.nb, v.nc, v.al0, v.al1);
            park_put(park, 2, v.al2, __uint_as_float(ksl), carried, 0.0f);
            park_fence();
            bool occluded = false;
            if (have_s) {
                if (STATS) cnt[ST_SHADOW]++;
                Seg ssg;
                float sk_occ;
#if MRTX_PARK_SHARE_ORIGIN
                MarchState ms;
                ms.rowA = org_row; ms.colA = org_col;
                bool sgo = march_begin_at<false, STATS, false, true>(f, oa, ob, oc, wa, wb, wc, ms, cnt, org_cell);
                while (sgo) march_segment<WIDE, false, STATS, MRTX_SHADOW_BATCH, 0, 2>(f, ms, 0.0f, ssg, sgo, occluded, sk_occ, cnt);
#else
                occluded = march<WIDE, false, STATS, BATCH, 2>(f, oa, ob, oc, wa, wb, wc, 0.0f, ssg, sk_occ, cnt);
#endif
            }
            park_fence();
#if MRTX_PARK_SHARE_ORIGIN
            { const v4f p3 = park_get(park, 3); org_row = p3.x; org_col = p3.y; org_cell = p3.z; }
#endif
            const v4f p0 = park_get(park, 0), p1 = park_get(park, 1), p2 = park_get(park, 2);
            v.pa = p0.x; v.pb = p0.y; v.pc = p0.z; v.na = p0.w;
            v.nb = p1.x; v.nc = p1.y; v.al0 = p1.z; v.al1 = p1.w;
            v.al2 = p2.x; ksl = __float_as_uint(p2.y);
            wgt = (have_s && !occluded) ? p2.z : 0.0f;
        } else {
            wgt = direct_light<STATS, WIDE, BATCH>(f, v, ul1, ul2, cnt);
        }
        PROF_END(5);
        o.c0 = fmaf(t0r * v.al0, wgt, o.c0);
        o.c1 = fmaf(t1r * v.al1, wgt, o.c1);
        o.c2 = fmaf(t2r * v.al2, wgt, o.c2);
        if (DEFER) {
            // The path goes on in path_kernel.  What is still coherent -- the 64 samples of a pixel decide together
            // whether to continue, build their continuation rays and locate the ray origins on the DEM grid -- is done
            // here at full occupancy; the queue record is a ray that is ready to march.
            if (continue_path(f, v, ksl, 1u, t0r, t1r, t2r, o.oa, o.ob, o.oc, o.da, o.db, o.dc)) {
                constexpr bool SHARED = PARK && MRTX_PARK_SHARE_ORIGIN != 0;
                if (SHARED) {
                    o.row = org_row; o.col = org_col;
                } else {
                    float q2;
                    exact_rowcol(f, o.oa, o.ob, o.oc, o.row, o.col, q2);
                }
                o.t0 = t0r; o.t1 = t1r; o.t2 = t2r;
                o.ks = ksl; o.aux = 0u;
                if (STATS) cnt[ST_BOUNCE]++;
#if MRTX_TRIAL_SEGMENT
                // The FIRST segment of the continuation ray is marched right here: the 64 rays of the pixel still start
                // within a texel of each other, every lane is busy, and four rays in five end inside it without touching
                // the terrain again (they have cleared everything in reach, horizon_kend, or left the shell) -- those paths
                // are finished in this wave.  A ray that hits, or is still marching after 16 steps, goes to path_kernel WITH
                // what the segment found out: the step that landed below the surface, or the state at the segment's end.
#ifdef MRTX_PROF
                uint32_t tcnt[16];
#pragma unroll
                for (int i = 0; i < 16; i++) tcnt[i] = 0;
#else
                uint32_t tcnt_store[STATS ? ST_N : 1];
                uint32_t* const tcnt = STATS ? tcnt_store : nullptr;
#endif
                if (STATS) {
#pragma unroll
                    for (int i = 0; i < ST_N; i++) tcnt[i] = 0;
                }
                MarchState tm;
                tm.rowA = o.row; tm.colA = o.col;
                // With an environment map bound (the reference's default, moon_renderer.py:604-607) 84 % of the continuation rays
                // end in this segment by leaving the Moon, and each then reads one texel of a 537 MB map -- a DRAM miss at the very
                // end of the wave's life.  The texel depends on the ray's direction alone: its fetch is issued HERE and lands while
                // the segment is marched (a wasted 4-byte fetch for the rays that hit or march on).
                constexpr bool ENVPRE = MRTX_ENV_PREFETCH != 0;
                uint32_t env_px = 0u;
                if (ENVPRE && CF(f)->bg != nullptr) {          // wave-uniform
                    float ex, ey, ez;
                    to_scene_dir(f, o.da, o.db, o.dc, ex, ey, ez);
                    env_px = reinterpret_cast<const uint32_t*>(CF(f)->bg)[env_texel_index(f, ex, ey, ez)];
                }
                if (PARK) {     // the radiance so far, the throughput and the key sit out the trial segment in LDS
                    park_put(park, 0, o.c0, o.c1, o.c2, o.t0);
                    park_put(park, 1, o.t1, o.t2, __uint_as_float(o.ks), 0.0f);
                    park_fence();
                }
                bool tgo = SHARED ? march_begin_at<false, STATS, false, true>(f, o.oa, o.ob, o.oc, o.da, o.db, o.dc, tm, tcnt, org_cell)
                                  : march_begin_at<false, STATS>(f, o.oa, o.ob, o.oc, o.da, o.db, o.dc, tm, tcnt);
                bool thit = false;
                Seg tsg;
                float tsk = 0.0f;
                if (tgo) march_segment<WIDE, false, STATS, MRTX_TRIAL_BATCH, MRTX_TRIAL_CP, 3>(f, tm, 0.0f, tsg, tgo, thit, tsk, tcnt);
                if (PARK) {
                    park_fence();
                    const v4f p0 = park_get(park, 0), p1 = park_get(park, 1);
                    o.c0 = p0.x; o.c1 = p0.y; o.c2 = p0.z; o.t0 = p0.w;
                    o.t1 = p1.x; o.t2 = p1.y; o.ks = __float_as_uint(p1.z);
                    t0r = o.t0; t1r = o.t1; t2r = o.t2;
                }
#ifdef MRTX_PROF_TRIAL   // measurement only: the trial's step iterations and the lanes evaluating in them (slots 13 / 14), its cycles (15)
                cnt[13] += tcnt[11]; cnt[14] += tcnt[12]; cnt[15] += tcnt[6] + tcnt[7];
#endif
                if (STATS) {
#pragma unroll
                    for (int i = 0; i < ST_N; i++) cnt[i] += tcnt[i];
                }
                if (!tgo && !thit) {
                    if (ENVPRE) escaped_path<STATS, true>(f, o.oa, o.ob, o.oc, o.da, o.db, o.dc, t0r, t1r, t2r, o.c0, o.c1, o.c2, cnt, env_px);
                    else escaped_path<STATS>(f, o.oa, o.ob, o.oc, o.da, o.db, o.dc, t0r, t1r, t2r, o.c0, o.c1, o.c2, cnt);
                } else {
                    o.path = true;
                    if (thit) {
                        o.aux = MRTX_REC_HIT | ((uint32_t)(int)rintf(tsk * f.inv_step) << 8);
                    } else {
                        o.aux = MRTX_REC_RESUME | ((uint32_t)tm.kend << 8);
                        o.row = tm.rowA; o.col = tm.colA;
                    }
                }
#else
                o.path = true;
#endif
            }
        }
        if (!BOUNCE) break;
        float boa, bob, boc, bda, bdb, bdc;
        if (!continue_path(f, v, ks, seg, t0r, t1r, t2r, boa, bob, boc, bda, bdb, bdc)) break;
        const uint32_t d0 = 4u + 5u * (seg - 1u);
        ul1 = u01(ks, d0 + 3u); ul2 = u01(ks, d0 + 4u);
        if (STATS) cnt[ST_BOUNCE]++;
        Seg bsg;
        float bhi = 0.0f;
        if (!march<WIDE, false, STATS, BATCH>(f, boa, bob, boc, bda, bdb, bdc, 0.0f, bsg, bhi, cnt)) {
            escaped_path<STATS>(f, boa, bob, boc, bda, bdb, bdc, t0r, t1r, t2r, o.c0, o.c1, o.c2, cnt);
            break;
        }
        const int bk = (int)rintf(bhi * f.inv_step);
        float blo = (float)(bk - 1) * f.step;
        refine<WIDE>(f, bsg, boa, bob, boc, bda, bdb, bdc, blo, bhi);
        if (STATS) { cnt[ST_HEIGHT] += (uint32_t)f.nbis; cnt[ST_FETCH] += (uint32_t)f.nbis; }
        hit_vertex<STATS, WIDE>(f, fmaf(blo, bda, boa), fmaf(blo, bdb, bob), fmaf(blo, bdc, boc), v, cnt);
        seg++;
    }
}

template <int S>
__device__ __forceinline__ float tree_sum(float v) {
#pragma unroll
    for (int m = 1; m < S; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// One wave = 64 (pixel, sample) pairs: P = 64/S pixels (PW x PH block) x S samples in adjacent lanes.
// One workgroup = MRTX_WG_WAVES waves over a (MRTX_WG_TILE x MRTX_WG_TILE)-pixel sub-tile of a sharding
// tile (never smaller than one wave's pixel block).  Measured at cfg3 (profiles/r01g_wgtile.txt):
// 4 waves x 16x16 px 21.1 ms, 8x8 18.5, 4x4 17.2, 2x2 15.7; 1 wave x 1 px 15.15 ms.  Per-pixel march cost
// varies ~20x between disc centre and limb, so the finest grain lets the dispatcher balance the CUs and
// no wave slot waits on a slower sibling of its workgroup.
#ifndef MRTX_WG_TILE
#define MRTX_WG_TILE 1
#endif
#ifndef MRTX_WG_WAVES
#define MRTX_WG_WAVES 1
#endif
#ifndef MRTX_SKY_WG_TILE
#define MRTX_SKY_WG_TILE 8    // MODE 3: a wave walks an 8x8-pixel block (a sky sample is ~150 instructions: one pixel per
#endif                        // workgroup is bound by the dispatcher, 1.9 ms for the 5 000 sky tiles of cfg3 + star map)
#ifndef MRTX_MIN_WAVES
#define MRTX_MIN_WAVES 4   // an upper bound on registers only: since a launch traces one block (round 4) the direct kernel needs 69 VGPRs
                           // whatever the bound says -- 7 waves per SIMD (rounds 1-3: 94 VGPRs, 13.57 ms at cfg3 with 4, 13.87 with 5)
#endif
#ifndef MRTX_XCD_SHARE
#define MRTX_XCD_SHARE 1   // 0 = always whole tiles per XCD (A/B switch, see the remap in render_kernel)
#endif
#ifndef MRTX_XCD_TILE_RUN
#define MRTX_XCD_TILE_RUN 1
#endif
#ifndef MRTX_XCD_SHARE_BELOW
#define MRTX_XCD_SHARE_BELOW 1200   // tiles per launch under which the XCDs share every tile
#endif
#ifndef MRTX_MIN_WAVES_BOUNCE
#define MRTX_MIN_WAVES_BOUNCE 5   // 8 spilled VGPRs at 5 waves/SIMD still beat 4 waves without spills (36.1 vs 39.8 ms)
#endif
// The COUNTING in-wave variants carry the counter array on top of the path state and want ~180-200 VGPRs; they only render
// the counted frame, so they get the registers they ask for (2 waves per SIMD, no VGPR spills) -- a performance choice.
// What is known about the round-1 report that one of them (<64, STATS, !WIDE, in-wave, OVERLAY>) miscomputed a sample when
// held to 72 VGPRs (~270 spilled): the failing case was not kept; the configuration has been rebuilt since (make spilltest)
// and 540 fuzz scenes through exactly that kernel equal the oracle in radiance, hits and counters; the march state was
// restructured in round 2 (every field initialised before use); the oracle is clean under ASan / UBSan.  No miscompute has
// been observed since, and tests/test_gpu_fuzz.py re-runs the spilled build every round.
#ifndef MRTX_BOUNCE_STATS_WAVES
#define MRTX_BOUNCE_STATS_WAVES 2   // tools/spill_repro.py builds with 7 to bring the spilled configuration back
#endif
// The in-wave path kernels that actually ship are the preview launches (1 / 2 samples per pixel; launches below ~8 M samples keep
// their paths in the wave): at 4 waves per SIMD they need no scratch and run as fast (4K, 1 spp: 1.302 ms against 1.312 with
// 17 spilled registers at 5 waves; 2 spp 2.256 / 2.270); the larger S (an A/B path and the no-memory fall-back) keep 5.
#define MRTX_BOUNCE_WAVES(STATS, SV) ((STATS) ? MRTX_BOUNCE_STATS_WAVES : ((SV) <= 2 ? 4 : MRTX_MIN_WAVES_BOUNCE))
template <int S, bool STATS, bool WIDE, int MODE, bool OVERLAY>
#ifndef MRTX_MIN_WAVES_DEFER
#define MRTX_MIN_WAVES_DEFER 3   // no longer what sets the occupancy: with one block per launch and its cold state parked in LDS the kernel
                                 // needs 62 VGPRs under any bound up to 8 -- 8 waves per SIMD (rounds 2-3: 121-125 VGPRs, 4 waves; a bound of
                                 // 5 then meant 17 spilled registers and 22.2 ms against 18.9)
#endif
__global__ void __launch_bounds__(64 * MRTX_WG_WAVES, MODE == 1 ? MRTX_BOUNCE_WAVES(STATS, S) : MODE == 2 ? MRTX_MIN_WAVES_DEFER : MRTX_MIN_WAVES)
render_kernel(const FrameC f, const PathQ pq) {
    constexpr bool DEFER = MODE == 2;
    constexpr int P = 64 / S;
    constexpr int PW = P >= 64 ? 8 : P >= 32 ? 8 : P >= 16 ? 4 : P >= 8 ? 4 : P >= 4 ? 2 : P >= 2 ? 2 : 1;
    constexpr int PH = P / PW;
    // workgroup tile edge in pixels: MRTX_WG_TILE, but at least two jobs wide so the 4 waves all have work
    constexpr int WGMIN = MRTX_WG_WAVES > 2 ? 2 * PW : MRTX_WG_WAVES > 1 ? 2 * PH : PW;
    constexpr int WGTILE = MODE == 3 ? MRTX_SKY_WG_TILE : MRTX_WG_TILE;
    constexpr int WGT = (WGMIN > WGTILE) ? WGMIN : WGTILE;
    constexpr int WGS = WGT == 16 ? 4 : WGT == 8 ? 3 : WGT == 4 ? 2 : WGT == 2 ? 1 : 0;
    constexpr int JX = WGT / PW, JY = WGT / PH, NJOBS = JX * JY;
    __shared__ unsigned int lds_cnt[ST_N];
    constexpr bool PARK = DEFER && MRTX_LDS_PARK != 0;
    __shared__ v4f park_lds[PARK ? MRTX_PARK_SLOTS * 64 * MRTX_WG_WAVES : 1];
    v4f* const park = PARK ? park_lds : nullptr;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware remap: consecutive blockIdx values go to XCDs round-robin, so block b and b+8 share an XCD (and its
    // L2).  Two deals, chosen per launch on the host (f.xcd_share):
    //  0: XCD x gets tiles x, x+8, ... whole -- best L2 locality; 3056 tiles at cfg3 on one GPU average out (14.87 ms
    //     against 14.99 for the other deal);
    //  1: every tile is SHARED by the eight XCDs -- XCD x takes the x-th contiguous eighth of its sub-tiles (4 pixel
    //     rows of a 32x32 tile at S = 64) and all XCDs walk the tile list in step.  Per-tile cost varies ~20x (limb,
    //     terminator), so a rank holding a few hundred tiles is otherwise bound by its unluckiest XCD: 1/8 of cfg3
    //     takes 2.24 ms instead of 2.60.
    const int subs_x = f.tile_w >> WGS, subs = subs_x * (f.tile_h >> WGS);
    const int b = blockIdx.x, xcd = b & 7, g = b >> 3;
    int li, sub;
    if (f.xcd_share) {
        const int per = subs >> 3;
        li = g / per; sub = xcd * per + g % per;
    } else {
        // an XCD takes MRTX_XCD_TILE_RUN consecutive tiles of the list at a time (raster neighbours: one L2 working set)
        constexpr int K = MRTX_XCD_TILE_RUN;
        const int rem = g % (subs * K);
        li = ((g / (subs * K)) * 8 + xcd) * K + rem / subs; sub = rem % subs;
    }
    if (li >= f.n_active) return;
    const int lt = f.tile_list ? f.tile_list[li] : li;
    const int t = lt * f.world + f.rank;
    int tx, ty;
    mrtx_tile_xy(t, f.tiles_x, f.tile_shift, tx, ty);
    const int px0 = tx * f.tile_w + (sub % subs_x) * WGT, py0 = ty * f.tile_h + (sub / subs_x) * WGT;
    if (px0 >= f.W || py0 >= f.H) return;

#ifdef MRTX_PROF
    uint32_t cnt[16];
    if (!STATS) {
#pragma unroll
        for (int i = 0; i < 16; i++) cnt[i] = 0;
    }
#else
    // the production kernels carry no counters: a null pointer instead of a dead array (which still cost every wave a
    // private segment: 36 bytes of scratch that no instruction touched)
    uint32_t cnt_store[STATS ? ST_N : 1];
    uint32_t* const cnt = STATS ? cnt_store : nullptr;
#endif
    if (STATS) {
#pragma unroll
        for (int i = 0; i < ST_N; i++) cnt[i] = 0;
        if (threadIdx.x < ST_N) lds_cnt[threadIdx.x] = 0;
        __syncthreads();
    }

    const int p = lane / S, s = lane % S;
    // MODE 3: lane l looks at pixel (l % 8, l / 8) of the wave's 8 x 8 block first -- one texel for the whole pixel? (see
    // sky_pixel_uniform) -- and the jobs below take a pixel's verdict and texel from the lane that holds them
    constexpr bool SKY_UNI = MODE == 3 && WGT == 8 && MRTX_WG_WAVES == 1;
    uint64_t sky_nu = ~0ull;             // lanes (= pixels of the 8 x 8 block) whose samples do NOT all read one texel
    auto sky_jobmask = [&](int jx_, int jy_) {   // the lanes that hold the PW x PH pixels of job (jx_, jy_)
        uint64_t mk = 0ull;
#pragma unroll
        for (int r = 0; r < PH; r++) mk |= (uint64_t)((1u << PW) - 1u) << ((jy_ * PH + r) * 8 + jx_ * PW);
        return mk;
    };
    if (SKY_UNI && CF(f)->bg != nullptr) {
        // lane l = pixel (l % 8, l / 8): a pixel whose whole job is uniform is finished right here, one lane per pixel -- per
        // block the S samples add up to S x texel exactly (the tree sum of S equal values), blocks in sequence as always
        const int ux = px0 + (lane & 7), uy = py0 + (lane >> 3);
        const bool uin = ux < f.W && uy < f.H;
        float sky0, sky1, sky2;
        const bool uni = sky_pixel_uniform(f, ux, uy, sky0, sky1, sky2);
        sky_nu = __ballot(uin && !uni);
        if (uin && (sky_nu & sky_jobmask((lane & 7) / PW, (lane >> 3) / PH)) == 0ull) {
            const int64_t upix = (int64_t)uy * f.W + ux;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f.first_block != 0) a = reinterpret_cast<const float4*>(CF(f)->accum)[upix];
            const float b0 = (float)S * sky0, b1 = (float)S * sky1, b2 = (float)S * sky2;
            a.x += b0; a.y += b1; a.z += b2; a.w += 0.0f;     // one block per launch
            reinterpret_cast<float4*>(CF(f)->accum)[upix] = a;
            reinterpret_cast<float4*>(CF(f)->hits)[upix] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (STATS) { cnt[ST_PRIMARY] += (uint32_t)S; cnt[ST_BG] += (uint32_t)S; }
        }
    }
    for (int job = wv; job < NJOBS; job += MRTX_WG_WAVES) {
        const int jx = job % JX, jy = job / JX;
        const int x = px0 + jx * PW + (p % PW), y = py0 + jy * PH + (p / PW);
        if (px0 + jx * PW >= f.W || py0 + jy * PH >= f.H) continue;  // wave-uniform
        if (SKY_UNI && (sky_nu & sky_jobmask(jx, jy)) == 0ull) continue;   // every pixel of the job was finished above
        const bool inb = x < f.W && y < f.H;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        SampleOut o;
        o.h0 = o.h1 = o.h2 = o.h3 = 0.f;
        o.path = false;
        o.oa = o.ob = o.oc = o.da = o.db = o.dc = o.t0 = o.t1 = o.t2 = o.row = o.col = 0.f; o.ks = 0u; o.aux = 0u;
        bool deferred = false;
        // A launch carries ONE block of S samples per pixel (mrtx_launch_render checks it; mrtx_render_part launches block after
        // block).  Round 4: with the block count a run-time value the compiler kept everything a sample can hand to the next
        // iteration -- running sums, hit record, hand-over fields, hoisted per-pixel invariants -- alive through every march:
        // render_kernel<MODE 2> 121 VGPRs (4 waves per SIMD); with a trip count of one 73 (6 waves), 62 with MRTX_LDS_PARK (8).
        constexpr uint32_t n_blk = 1u;
        for (uint32_t blk = 0; blk < n_blk; blk++) {
            o.c0 = o.c1 = o.c2 = o.hitflag = 0.f;
            PROF_BEGIN(0);
            if (inb) trace_sample<STATS, WIDE, MODE, OVERLAY>(f, lt, x, y, (f.first_block + blk) * (uint32_t)S + (uint32_t)s, o, cnt, s == 0, park);
            PROF_END(0);
            if (blk == 0 && f.first_block != 0 && inb && s == 0) {
                // the running sums of earlier launches are fetched HERE, not before the first block was traced (four registers
                // less while it is); the order of the additions is the spec's: ((prev + b0) + b1) + ...
                const float4 prev = reinterpret_cast<const float4*>(CF(f)->accum)[(int64_t)y * f.W + x];
                s0 = prev.x; s1 = prev.y; s2 = prev.z; s3 = prev.w;
            }
            if (DEFER) deferred = __ballot(o.path) != 0ull;
            if (DEFER && deferred) {
                // Some path of this wave continues: every lane hands its sample to path_kernel / resolve_paths_kernel
                // (the radiance sum of a pixel needs all S final values, in the spec's order); coverage is final now.
                const uint32_t chunk = (uint32_t)blockIdx.x * (uint32_t)NJOBS + (uint32_t)job;
                const uint32_t e = chunk * 64u + (uint32_t)lane;
                // ray records: only the lanes whose path goes on, COMPACTED to the front of the chunk (npaths says how many;
                // lane_of maps a record back to its sample); the running radiance of all 64 samples in lane order
                const uint64_t pm = __ballot(o.path);
                const uint32_t es = chunk * 64u + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
#ifndef MRTX_AB_NOSTORE   // A/B only: how much of render_kernel<MODE 2> is the hand-over traffic (results are wrong without it)
                if (o.path) {
                    nt_store4(pq.ray0 + es, o.oa, o.ob, o.oc, o.da);
                    nt_store4(pq.ray1 + es, o.db, o.dc, o.t0, o.t1);
                    nt_store4(pq.ray2 + es, o.t2, o.row, o.col, __uint_as_float(o.ks));
                    pq.lane_of[es] = (uint32_t)lane | o.aux;
                }
#if MRTX_C_AOS == 2
                { float* const cp = reinterpret_cast<float*>(pq.c4) + 3u * (size_t)e;
                  __builtin_nontemporal_store(o.c0, cp); __builtin_nontemporal_store(o.c1, cp + 1); __builtin_nontemporal_store(o.c2, cp + 2); }
#elif MRTX_C_AOS
                nt_store4(pq.c4 + e, o.c0, o.c1, o.c2, 0.0f);
#else
                __builtin_nontemporal_store(o.c0, pq.c0 + e);
                __builtin_nontemporal_store(o.c1, pq.c1 + e);
                __builtin_nontemporal_store(o.c2, pq.c2 + e);
#endif
#endif
                if (lane == 0) pq.npaths[chunk] = (uint8_t)__popcll(pm);
                if (lane == 0) pq.meta[chunk] = 0x80000000u | (uint32_t)(px0 + jx * PW) | ((uint32_t)(py0 + jy * PH) << 15);
            } else {
                s0 += tree_sum<S>(o.c0);
                s1 += tree_sum<S>(o.c1);
                s2 += tree_sum<S>(o.c2);
            }
            s3 += tree_sum<S>(o.hitflag);
        }
        if (inb && s == 0) reinterpret_cast<float4*>(CF(f)->accum)[(int64_t)y * f.W + x] = make_float4(s0, s1, s2, s3);
        if (!PARK && inb && s == 0) reinterpret_cast<float4*>(CF(f)->hits)[(int64_t)y * f.W + x] = make_float4(o.h0, o.h1, o.h2, o.h3);   // PARK: trace_sample stored it
    }

#ifdef MRTX_PROF
    if (!STATS && lane == 0 && (((uint32_t)(blockIdx.x >> 3) * 2654435761u) >> 26) == 0u) {   // 1/64 of the waves
        cnt[10] = 1;
#pragma unroll
        for (int i = 0; i < 16; i++) atomicAdd(&g_prof[i], (unsigned long long)cnt[i]);
    }
#endif
    if (STATS) {
#pragma unroll
        for (int i = 0; i < ST_N; i++) {
            uint32_t v = cnt[i];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) atomicAdd(&lds_cnt[i], v);
        }
        __syncthreads();
        if (threadIdx.x < ST_N) atomicAdd(&CF(f)->stats[threadIdx.x], (unsigned long long)lds_cnt[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------
// D6 behind a queue: what is left of a path after render_kernel<MODE 2> (set_uint("path_seg_range", 2, 4),
// moon_renderer.py:583).
//
// Why not inside render_kernel: a continuation ray is cosine-distributed about the normal; the 64 rays of a pixel need
// between one and ten segments, and the few paths that hit terrain again (~8 %) drag the wave through bisection, vertex,
// shadow march and the next bounce at that lane occupancy (round 1: 35.4 ms against 13.5 ms for direct light).
// render_kernel<MODE 2> therefore only does what is still coherent -- roulette, the continuation ray, its FIRST segment,
// which ends 84 % of the paths -- and hands the surviving rays over, compacted per chunk.  Here a wave is PERSISTENT and
// keeps 64 marches in flight whatever path they belong to: every lane is a little state machine, a lane whose path is
// finished takes the next record of its wave's current group, and the rare heavy steps (a continuation ray that hit
// terrain: bisection + vertex + light sample; a vertex that got its direct term: roulette + next ray) wait until enough
// lanes need them.  Shadow and continuation rays share the one march.  No workgroup barrier; the only atomics hand out
// groups of chunks (see the kernel body); a watchdog instead of a hang.
// Every arithmetic step is the one trace_sample<MODE 1> performs for the same (pixel, sample), in the same order, so
// the three implementations (this, the in-wave loop, the oracle) agree bit for bit.
#ifndef MRTX_PATH_WAVES
#define MRTX_PATH_WAVES 5
#endif
#ifndef MRTX_PATH_STEPS
#define MRTX_PATH_STEPS 2
#endif
#ifndef MRTX_PATH_WIDE
#define MRTX_PATH_WIDE 0      // 1 = wide stepping while a wave drains (see the kernel body): bit-exact, measured SLOWER (path stage 5.44 ms
                              // against 5.13 on one GPU, 0.98 / 0.91 ms for a rank of eight): the block costs three more spilled registers
                              // in the main phase, more than the shorter drain gives back -- an A/B switch, off
#endif
#ifdef MRTX_PATH_PROF   // measurement build only (tools/path_prof.py): block executions and lane counts of path_kernel
__device__ unsigned long long g_pprof[16];
__device__ unsigned long long g_pprof_end[8192];   // end time of every persistent wave (tools/path_prof.py: the shape of the tail)
__device__ unsigned long long g_pprof_t[16];   // [0..7] latest wave end per label, [8] earliest wave start (wall_clock64 ticks, 100 MHz), [9] waves
#endif
enum { PS_IDLE = 0, PS_NEEDSEG, PS_STEP, PS_BISECT, PS_ENDED, PS_HITWAIT, PS_SHADE, PS_ESCAPED };

// ---- path_kernel's step mask (round 4, MRTX_PATH_MIP2).  The kernel is bound by 128-byte line fills for 16-byte footprints, and half
// of its L2 misses are march steps -- 85 % of which turn out to be more than 500 m above the terrain (round 3's margin histogram): the
// max-mip that bounds a whole 16-step segment has cells of 64 texels at cfg3, far too coarse to see that.  The MEDIUM max-mip has
// cells a quarter of that size (16 texels, 17 MB: L2 / Infinity-Cache resident, and consecutive steps share its lines); at segment
// set-up every step the segment's skip interval kept is tested against the cell its footprint lies in -- r^2(s_k) > (R m)^2 (1 + 1e-5)
// proves it above the surface, exactly as the coarse test does -- and only the steps that survive are evaluated (`todo`, one bit
// per step).  Result-preserving like the other skips: radiance, hits and the spec counters are unchanged, MRTX_F_NO_SKIP switches
// it off with the rest.  Fetched four steps at a time (one memory round per four tests).
// MEASURED (cfg3, gpurun_out/r4w, prof_ab_m2on / m2off): the mask drops 74 % of path_kernel's step evaluations (151 M -> 39 M), its
// L2 misses 215 M -> 121 M and its traffic beyond the L2s 26 -> 14 GB per frame -- and the kernel takes 4.53 ms instead of 4.35.  So
// path_kernel is NOT bound by the bytes it moves (rounds 2-3 read "0.74 of the HBM peak" that way): what bounds it is the chain of
// dependent memory rounds a wave goes through (the mask trades ~1.6 step rounds per set-up for ~1.5 medium-mip rounds + 0.6), with
// the memory system merely close to saturation at the same time.  Off by default (MRTX_PATH_MIP2 in mrtx_device.h); kept as a switch
// because it is bit-exact and halves the path stage's HBM traffic, which a bandwidth-starved configuration might want.
#if MRTX_PATH_MIP2 && MRTX_PATH_WIDE
#error "MRTX_PATH_WIDE walks j .. jhi contiguously: build it with -DMRTX_PATH_MIP2=0"
#endif
template <bool STATS>
__device__ __forceinline__ uint32_t step_mask(const FrameC& f, const MarchState& m, const Seg& sg, uint32_t todo, uint32_t* cnt) {
    const float* m2 = CF(f)->mip2;
    const int pitch = CF(f)->m2_pitch, sh = CF(f)->m2_shift, mh = CF(f)->m2_h, mw = CF(f)->m2_w;
    uint32_t rem = todo, keep = 0u;
    while (rem != 0u) {
        int jj[4]; bool on[4]; float mv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            on[q] = rem != 0u;
            jj[q] = on[q] ? (int)__builtin_ctz(rem) + 1 : 1;
            rem &= rem - 1u;                                     // 0 stays 0
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float u = ((float)(m.ka + jj[q]) * f.step - sg.sa) * f.inv_step;      // as below_seg()
            const float rowf = fmaf(u, fmaf(u, sg.r2, sg.r1), sg.ra), colf = fmaf(u, fmaf(u, sg.c2, sg.c1), sg.ca);
            int i = ((int)floorf(rowf) >> sh) + 1, c = ((int)floorf(colf) >> sh) + 1;     // + 1: the one-cell border (floor = -1 -> cell -1)
            i = max(0, min(i, mh + 1)); c = max(0, min(c, mw + 1));                      // never bites for a valid segment
            mv[q] = on[q] ? m2[i * pitch + c] : 0.0f;
        }
        if (STATS) { cnt[ST_MIP] += (on[0] ? 1u : 0u) + (on[1] ? 1u : 0u) + (on[2] ? 1u : 0u) + (on[3] ? 1u : 0u); }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float sk = (float)(m.ka + jj[q]) * f.step;
            const float pa = fmaf(sk, m.da, m.oa), pb = fmaf(sk, m.db, m.ob), pc = fmaf(sk, m.dc, m.oc);
            const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
            const float rd = f.Rf * mv[q];
            if (on[q] && r2 <= (rd * rd) * 1.00001f) keep |= 1u << (jj[q] - 1);
        }
    }
    return keep;
}

// The march inside path_kernel is cut at STEP granularity, not at segment granularity: within one 16-step segment
// the rays of a wave need anything from 0 to 16 dependent DEM fetches (mean ~4), and a wave that steps a whole
// segment per iteration waits for its slowest lane -- up to 16 memory round trips per iteration (measured: 17.6 ms
// for the path stage, nearly inversely proportional to the number of waves in flight: pure latency).  So a lane is in
// one of two march states: NEEDSEG (the next segment's anchors, quadratic and skip interval are to be set up: ~300
// VALU + the max-mip fetch) and STEP (its next step of the current segment is to be evaluated: ~45 VALU + one DEM
// fetch); every iteration evaluates ONE step for all stepping lanes, and the set-up block runs when enough lanes
// need it (or nobody is stepping).  Same evaluations, same order per ray as march_segment().
// MRTX_PATH_PARK (round 4): the lanes' COLD state -- the current vertex (9 floats), the radiance its light sample carries, the path
// throughput (3) and the sample's radiance so far (3) -- lives in LDS, four lane-private 16-byte slots (4 KB per wave), and is touched
// only by the blocks that need it (refill, march over, the rare blocks).  path_kernel is bound by its waves' chains of dependent memory
// rounds times the waves a SIMD holds (section 4.6 of DESIGN.md), so sixteen registers less looked like the price of a sixth wave.
// MEASURED (cfg3, gpurun_out/r4x/path_park.log; path stage ms incl. resolve, two rounds): as shipped 4.98; parked at 5 waves (94
// VGPRs, no scratch instead of 96 + 2 spilled) 4.98-5.04; at launch bounds 6 / 7 (80 / 72 VGPRs, 12 / 20 spilled: the set-up and
// vertex blocks' temporaries are what fills the file, not the cold state) 5.36 / 6.10; with the step mask (MRTX_PATH_MIP2) on top
// 4.97 / 5.00 / 4.96 at 5 / 6 / 7 waves.  Neither more resident waves, nor 46 % less traffic, nor both move the kernel: off (an A/B
// switch; bit-exact, tools/quick_parity.py).
#ifndef MRTX_PATH_PARK
#define MRTX_PATH_PARK 0
#endif
template <bool STATS, bool WIDE>
__global__ void __launch_bounds__(64, STATS ? 2 : MRTX_PATH_WAVES) path_kernel(const FrameC f, const PathQ pq) {
    const uint32_t lane = threadIdx.x;
#if MRTX_PATH_PARK
    __shared__ float pk_lds[4 * 64 * 4];
    float* const pk = pk_lds + lane * 4u;              // slot s, component c of this lane: pk[s * 256 + c]
#define PKS(s, c) pk[(s) * 256 + (c)]
    // slot 0: v.pa pb pc na | 1: v.nb nc al0 al1 | 2: v.al2, carried, t0r, t1r | 3: t2r, c0, c1, c2
#define COLD_GET_T(a, b, c) do { a = PKS(2, 2); b = PKS(2, 3); c = PKS(3, 0); } while (0)
#define COLD_SET_T(a, b, c) do { PKS(2, 2) = a; PKS(2, 3) = b; PKS(3, 0) = c; } while (0)
#define COLD_GET_C(a, b, c) do { a = PKS(3, 1); b = PKS(3, 2); c = PKS(3, 3); } while (0)
#define COLD_SET_C(a, b, c) do { PKS(3, 1) = a; PKS(3, 2) = b; PKS(3, 3) = c; } while (0)
#endif
    // Work distribution.  The records of render block b (b % 8 = a group label: blocks with one label ran on one XCD and
    // cover neighbouring pixels of the same tiles, see the remap in render_kernel) are taken in GROUPS of G consecutive
    // blocks of one label, handed out by atomic counters in device memory -- one set of counters per label of THIS
    // kernel's blocks, so that at any time the ~640 waves of an XCD march rays that start within about one
    // 32x32-pixel tile of each other and share its DEM neighbourhood in that XCD's L2.  (A static deal -- wave w owns
    // chunks w, w + NW, ... -- lets the waves drift apart by dozens of tiles: 46 % L2 hits, 80 GB of HBM reads.)
    // Which wave marches which ray has no influence on any result.
    const uint32_t label = blockIdx.x & 7u;   // which blocks share an XCD (observed round-robin dispatch); correctness does not depend on it
    const uint32_t sub = (blockIdx.x >> 3) % (uint32_t)pq.n_sub;
    uint32_t* const ctr = pq.counters + label * (uint32_t)pq.n_sub + sub;
    const uint32_t npos = (pq.grid_a + 7u) >> 3;                     // render blocks per label
    const uint32_t cpg = 1u << (pq.grp_log2 + pq.njobs_log2);         // chunks in a group (<= 64)
    // current group: first block position, chunk being read, records taken from it / it holds; lane i < cpg keeps the
    // record count of the group's chunk i.  The NEXT group's number is fetched (atomic) while this one is consumed.
    uint32_t grp_first = 0, cidx = cpg, pos = 0, ncur = 0, gcnt = 0;
    bool more = npos > 0;                                             // groups may be left
    uint32_t next_g = 0;
    if (lane == 0 && more) next_g = atomicAdd(ctr, 1u);
    uint32_t iterations = 0;
    uint32_t cnt_store[STATS ? ST_N : 1];
    uint32_t* const cnt = STATS ? cnt_store : nullptr;
    if (STATS) {
#pragma unroll
        for (int i = 0; i < ST_N; i++) cnt[i] = 0;
    }
#ifdef MRTX_PATH_PROF
    uint32_t pf[16];   // wave-uniform: iterations, then (executions, lanes) of refill / set-up / step / rare
#pragma unroll
    for (int i = 0; i < 16; i++) pf[i] = 0;
    if (lane == 0) atomicMin(&g_pprof_t[8], (unsigned long long)wall_clock64());
#endif

    // per-lane path state
    int state = PS_IDLE;
    bool shadow = false;          // which ray the lane is marching: the shadow ray of its vertex or the continuation ray
    bool hit = false, have_c = false;
    uint32_t e = 0, ks = 0, seg = 1;
    MarchState m;
    Seg sg;
    int j = 1;                    // next step of the current segment (STEP lanes)
#if MRTX_PATH_MIP2
    uint32_t todo = 0u;           // the steps of the current segment still to be evaluated, bit jj - 1 for step jj >= j
#endif
    float sk_hit = 0.0f;
#if MRTX_PATH_PARK
    float wgt = 0.0f;
#else
    Vertex v;
    float t0r = 1.0f, t1r = 1.0f, t2r = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, carried = 0.0f, wgt = 0.0f;
#endif
    // BISECT lanes (a continuation ray that hit: D3's bisection, one level per DEM fetch like any other step) keep the
    // bracket in registers that are dead meanwhile: lo in wgt, hi in sk_hit, the levels left in j
    float& bis_lo = wgt;
    float& bis_hi = sk_hit;
    m.oa = m.ob = m.oc = m.da = m.db = m.dc = 0.0f; m.rq.q0 = m.rq.b = m.rq.a = 0.0f; m.rowA = m.colA = m.q2A = 0.0f; m.ka = 0; m.kend = 0;
    sg.sa = sg.ra = sg.r1 = sg.r2 = sg.ca = sg.c1 = sg.c2 = 0.0f; sg.jlo = 1; sg.jhi = 0; sg.exact = false;
#if !MRTX_PATH_PARK
    v.pa = v.pb = v.pc = v.na = v.nb = v.nc = v.al0 = v.al1 = v.al2 = 0.0f;
#endif

    for (;;) {
        // Every lane waits for exactly one of four blocks -- refill, segment set-up, step, rare -- and the wave decides
        // per iteration which of them to run (wave-uniform flags), so that the expensive ones execute with many lanes:
        // stepping is cheap and runs whenever a lane can step; the others wait for their thresholds, or until nothing
        // cheaper can make progress.
        const uint64_t idle_m = __ballot(state == PS_IDLE);
        const int nidle = __popcll(idle_m);
        const bool can_refill = more;
        if (nidle == 64 && !can_refill) break;   // every path of this wave is finished and no work is left: the regular exit
        if (++iterations > (1u << 24)) {         // watchdog: far beyond any real frame (~1e4 iterations); never hang the GPU
            if (lane == 0) atomicAdd(&CF(f)->stats[15], 1ull);
            break;
        }
        const int n_seg = __popcll(__ballot(state == PS_NEEDSEG)), n_step = __popcll(__ballot(state == PS_STEP || state == PS_BISECT));
        const int n_rare = __popcll(__ballot(state == PS_HITWAIT || state == PS_SHADE || state == PS_ESCAPED));
        // stepping is cheap and runs whenever a lane can step; the others wait for their thresholds, or until nothing cheaper
        // can make progress.  (Running only the block most lanes wait for was measured: more iterations, 18.9 ms against 17.7.)
        // (Dropping the thresholds once the queue is empty -- no new ray will fill a block any more -- was measured: the waves of a
        // launch still end spread over ~1.2 ms, on average 0.6 ms before the last one (tools/path_prof.py): that spread is the
        // length of the longest of a wave's last 64 paths, 40-200 iterations of ~5 us, not time spent waiting for a threshold.)
        const bool do_step = n_step > 0;
        const int seg_min = pq.seg_min, rare_min = pq.rare_min;
        const bool do_seg = n_seg > 0 && (n_seg >= seg_min || n_step == 0);
        const bool do_refill = can_refill && nidle > 0 && (nidle >= pq.refill_min || (n_step == 0 && !do_seg));
        const bool do_rare = n_rare > 0 && (n_rare >= rare_min || (n_step == 0 && !do_seg && !do_refill));
#ifdef MRTX_PATH_PROF
        pf[0]++;
        if (do_refill) { pf[1]++; pf[2] += (uint32_t)nidle; }
        if (do_seg) { pf[3]++; pf[4] += (uint32_t)n_seg; }
        if (do_step) { pf[5]++; pf[6] += (uint32_t)n_step; }
        if (do_rare) { pf[7]++; pf[8] += (uint32_t)n_rare; }
#define PPROF_T(i) { const unsigned long long _t = __builtin_readcyclecounter(); pf[i] += (uint32_t)(_t - pt); pt = _t; }
        unsigned long long pt = __builtin_readcyclecounter();
#else
#define PPROF_T(i)
#endif

        bool segend = false;
        if (do_refill) {
            // ---- refill: idle lanes take the next ray records of the wave's current group, chunk after chunk (a chunk
            // holds npaths[chunk] of them, compacted); wave-uniform bookkeeping, one fetch round for the lanes that got one
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
            uint32_t need = (uint32_t)nidle, assigned = 0;
            bool fill_l = false;
            while (need > 0 && more) {
                if (pos >= ncur) {                       // next chunk, or next group
                    cidx++;
                    if (cidx >= cpg) {
                        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_g);
                        grp_first = (g * (uint32_t)pq.n_sub + sub) << pq.grp_log2;
                        more = grp_first < npos;
                        if (!more) break;
                        if (lane == 0) next_g = atomicAdd(ctr, 1u);
                        gcnt = 0;
                        if (lane < cpg) {
                            const uint32_t blk = ((grp_first + (lane >> pq.njobs_log2)) << 3) + label;
                            if (blk < pq.grid_a) gcnt = pq.npaths[(blk << pq.njobs_log2) + (lane & ((1u << pq.njobs_log2) - 1u))];
                        }
                        cidx = 0;
                    }
                    ncur = (uint32_t)__builtin_amdgcn_readlane((int)gcnt, (int)cidx);
                    pos = 0;
                    continue;
                }
                const uint32_t take = min(need, ncur - pos);
                if (state == PS_IDLE && !fill_l && r >= assigned && r < assigned + take) {
                    const uint32_t blk = ((grp_first + (cidx >> pq.njobs_log2)) << 3) + label;
                    e = (((blk << pq.njobs_log2) + (cidx & ((1u << pq.njobs_log2) - 1u))) << 6) + pos + (r - assigned);
                    fill_l = true;
                }
                pos += take; assigned += take; need -= take;
            }
            if (fill_l) {
                const float4 r0 = nt_load4(pq.ray0 + e), r1 = nt_load4(pq.ray1 + e), r2 = nt_load4(pq.ray2 + e);
                const uint32_t aux = __builtin_nontemporal_load(pq.lane_of + e);
                e = (e & ~63u) | (aux & 63u);            // from here on: the sample's slot in c0/c1/c2
                m.rowA = r2.y; m.colA = r2.z;
                bool go = true;
                j = 1;
                if (aux & (MRTX_REC_RESUME | MRTX_REC_HIT)) {
                    m.oa = r0.x; m.ob = r0.y; m.oc = r0.z; m.da = r0.w; m.db = r1.x; m.dc = r1.y;
                    m.q2A = fmaf(m.ob, m.ob, m.oa * m.oa);
                    m.rq.q0 = fmaf(m.oc, m.oc, m.q2A);
                    m.rq.b = fmaf(m.oc, m.dc, fmaf(m.ob, m.db, m.oa * m.da));
                    m.rq.a = fmaf(m.dc, m.dc, fmaf(m.db, m.db, m.da * m.da));
                    if (aux & MRTX_REC_RESUME) {
                        // render_kernel marched segment 1 and the ray is still going: take the march up at step 16 (the
                        // record carries the texel coordinates there and the horizon bound; the rest follows from the ray)
                        const float sb = (float)SEG_N * f.step;
                        const float pa = fmaf(sb, m.da, m.oa), pb = fmaf(sb, m.db, m.ob);
                        m.q2A = fmaf(pb, pb, pa * pa);
                        m.ka = SEG_N;
                        m.kend = (int)(aux >> 8);
                    } else {
                        // ... or found the step that lands below the surface: only the segment's quadratic is needed again
                        // (for the bisection); j < 0 tells the set-up block
                        m.ka = 0;
                        m.kend = f.kmax;
                        j = -(int)(aux >> 8);
                    }
                } else {
                    go = march_begin_at<false, STATS, true>(f, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, m, cnt);
                }
#if MRTX_PATH_PARK
                COLD_SET_T(r1.z, r1.w, r2.x);
#else
                t0r = r1.z; t1r = r1.w; t2r = r2.x;
#endif
                ks = __float_as_uint(r2.w);
                hit = false; shadow = false; have_c = false;
                seg = 1;
                state = go ? PS_NEEDSEG : PS_ENDED;
            }
        }
        PPROF_T(9);
        if (do_seg) {
            // ---- segment set-up for the lanes that need one
            if (state == PS_NEEDSEG) {
                float rowB, colB, q2B;
                const int known_hit = j;                    // < 0: render_kernel stepped this segment already, the hit is at step -j
                const uint32_t mip0 = STATS ? cnt[ST_MIP] : 0u;
                const bool open_kend = m.kend < 0;          // first segment of a march begun in this kernel: the horizon
                if (open_kend) m.kend = horizon_kend(f, m); // bound is looked up beside the max-mip (one memory round)
                else if ((MRTX_HORIZON_RETRY & 2) != 0 && known_hit >= 0 && m.ka > 0 && m.kend >= f.kmax) {
                    const int ke = horizon_retry(f, m);     // out of reach at the origin: asked again from here (see march_segment)
                    m.kend = ke < f.kmax ? m.ka + max(ke, 0) : f.kmax;
                }
                seg_setup<STATS>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, m.rq, m.ka, m.rowA, m.colA, m.q2A, sg, rowB, colB, q2B, cnt);
                if (open_kend && m.kend < 1) {
                    // already above everything in reach: no step can hit, the march ends before its first segment
                    if (STATS) { cnt[ST_MIP] = mip0; cnt[ST_HEIGHT] += steps_after(f, m, 1); }
                    state = PS_ENDED;
                } else if (known_hit < 0) {
                    if (STATS) cnt[ST_MIP] = mip0;          // counted where the segment was marched
                    hit = true; sk_hit = (float)(-known_hit) * f.step;
                    j = 1;
                    state = PS_ENDED;
                } else {
                    if (STATS) cnt[ST_HEIGHT] += count_in_steps<false>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, 0.0f, m.ka, 1, sg.jlo - 1);
                    m.rowA = rowB; m.colA = colB; m.q2A = q2B;   // the next segment starts where this one ends
                    sg.jhi = max(min(sg.jhi, m.kend - m.ka), sg.jlo - 1);   // steps beyond kend cannot be at/below the surface
                    if ((MRTX_SEG_MASK & 8) != 0 && CF(f)->mip2 != nullptr) sg.jhi = last_kept_step<STATS, MRTX_PATH_MASK_Q>(f, m, sg, cnt);
                    j = sg.jlo;
#if MRTX_PATH_MIP2
                    todo = j <= sg.jhi ? (((2u << (sg.jhi - 1)) - 1u) & ~((1u << (j - 1)) - 1u)) : 0u;      // steps j .. jhi
                    if (CF(f)->mip2 != nullptr && !sg.exact) todo = step_mask<STATS>(f, m, sg, todo, cnt);
                    if (todo != 0u) state = PS_STEP; else segend = true;
#else
                    if (j <= sg.jhi) state = PS_STEP; else segend = true;
#endif
                }
            }
        }
        PPROF_T(10);
#if MRTX_PATH_WIDE
        // ---- DRAIN, wide stepping: once the queue is empty a wave ends when its last path does, and a marching path needs an
        // iteration per MRTX_PATH_STEPS steps while most lanes idle.  With at most four lanes stepping, each of them borrows a
        // quarter of the wave: lane 16 g + i evaluates step j + i of the g-th stepping lane's segment, and the owner takes the
        // first step (in march order) that ends the march -- the same evaluations in the same order as one step at a time, a
        // whole segment per iteration.  Results and counters unchanged (the steps behind the terminating one are not counted).
        bool wide_done = false;
        {
            const uint64_t stepm = __ballot(state == PS_STEP);
            const int nst = __popcll(stepm);
            if (!more && nst > 0 && nst <= 4) {            // wave-uniform
                const uint32_t grp = lane >> 4, sub16 = lane & 15u;
                int src = -1;
                {
                    uint64_t mm = stepm;
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        const int bpos = mm ? (int)__builtin_ctzll(mm) : -1;
                        src = (i == grp) ? bpos : src;
                        mm &= mm - 1ull;
                    }
                }
                const bool has = src >= 0;
                const int sl = has ? src : 0;
                Seg wsg;
                const float woa = __shfl(m.oa, sl, 64), wob = __shfl(m.ob, sl, 64), woc = __shfl(m.oc, sl, 64);
                const float wda = __shfl(m.da, sl, 64), wdb = __shfl(m.db, sl, 64), wdc = __shfl(m.dc, sl, 64);
                wsg.sa = __shfl(sg.sa, sl, 64); wsg.ra = __shfl(sg.ra, sl, 64); wsg.r1 = __shfl(sg.r1, sl, 64); wsg.r2 = __shfl(sg.r2, sl, 64);
                wsg.ca = __shfl(sg.ca, sl, 64); wsg.c1 = __shfl(sg.c1, sl, 64); wsg.c2 = __shfl(sg.c2, sl, 64);
                wsg.exact = __shfl((int)sg.exact, sl, 64) != 0;
                wsg.jlo = 1; wsg.jhi = SEG_N;
                const int wka = __shfl(m.ka, sl, 64), wj = __shfl(j, sl, 64), wjhi = __shfl(sg.jhi, sl, 64);
                const int jj = wj + (int)sub16;
                const bool valid = has && jj <= wjhi;
                const int kk = wka + min(jj, wjhi);
                const float wsk = (float)kk * f.step;
                const float wpa = fmaf(wsk, wda, woa), wpb = fmaf(wsk, wdb, wob), wpc = fmaf(wsk, wdc, woc);
                const float wr2 = fmaf(wpc, wpc, fmaf(wpb, wpb, wpa * wpa));
                const bool win = (wr2 <= f.R2f) & (kk <= f.kmax);
                const bool wbel = below_seg<WIDE, true, MRTX_PATH_CP>(f, wsg, wsk, wpa, wpb, wpc, wr2);
                const uint64_t hm = __ballot(valid & win & wbel), om = __ballot(valid & !win);
                if (state == PS_STEP) {
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(stepm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)stepm, 0u));   // < 4
                    const uint32_t h16 = (uint32_t)(hm >> (16u * rk)) & 0xFFFFu, o16 = (uint32_t)(om >> (16u * rk)) & 0xFFFFu;
                    const int nvalid = min(16, sg.jhi - j + 1);
                    const uint32_t term = h16 | o16;
                    if (term != 0u) {
                        const int t = (int)__builtin_ctz(term);
                        const bool is_hit = ((h16 >> t) & 1u) != 0u;
                        if (STATS) { cnt[ST_HEIGHT] += (uint32_t)t + (is_hit ? 1u : 0u); cnt[ST_FETCH] += (uint32_t)t + 1u; }
                        if (is_hit) { hit = true; sk_hit = (float)(m.ka + j + t) * f.step; }
                        j += t + 1;
                        state = PS_ENDED;
                    } else {
                        if (STATS) { cnt[ST_HEIGHT] += (uint32_t)nvalid; cnt[ST_FETCH] += (uint32_t)nvalid; }
                        j += nvalid;
                        if (j > sg.jhi) segend = true;
                    }
                }
                wide_done = true;
            }
        }
#else
        const bool wide_done = false;
#endif
        if (do_step) {
            // ---- the next MRTX_PATH_STEPS steps of every stepping lane: their DEM footprints are fetched together (one
            // memory round trip per iteration is what bounds this kernel), then the steps are tested in march order and
            // whatever follows the one that ends the march or the segment is dropped (its fetch is wasted).  Same
            // evaluations, same order, same results as one step at a time.  cfg3: 1 step 14.5 ms, 2 steps 13.3.
            if ((state == PS_STEP && !wide_done) || state == PS_BISECT) {
                // A BISECT lane evaluates the mid-point of its bracket in slot 0 and, speculatively, the mid-point of the
                // lower half in slot 1 (the next level if slot 0 turns out below the surface): the bisection of D3 rides the
                // fetch rounds of the march instead of adding five of its own to the rare block.
                const bool bis = state == PS_BISECT;
                bool bel[MRTX_PATH_STEPS], in[MRTX_PATH_STEPS];
                float sks[MRTX_PATH_STEPS];
#ifdef MRTX_PROF_MARGIN
                float mrg[MRTX_PATH_STEPS];
#endif
                const float mid0 = 0.5f * (bis_lo + bis_hi);
#if MRTX_PATH_MIP2
                // the next MRTX_PATH_STEPS steps the mask kept (a lane with fewer left repeats its last one: a wasted fetch)
                int js[MRTX_PATH_STEPS]; bool have[MRTX_PATH_STEPS];
                {
                    uint32_t t = todo;
#pragma unroll
                    for (int i = 0; i < MRTX_PATH_STEPS; i++) {
                        have[i] = t != 0u;
                        js[i] = have[i] ? (int)__builtin_ctz(t) + 1 : (i ? js[i - 1] : 1);
                        t &= t - 1u;
                    }
                }
#endif
#pragma unroll
                for (int i = 0; i < MRTX_PATH_STEPS; i++) {
#if MRTX_PATH_MIP2
                    const int k = m.ka + js[i];
#else
                    const int k = m.ka + min(j + i, sg.jhi);                 // a step past jhi is read at jhi instead
#endif
                    const float sk = bis ? (i == 1 ? 0.5f * (bis_lo + mid0) : mid0) : (float)k * f.step;
                    const float pa = fmaf(sk, m.da, m.oa), pb = fmaf(sk, m.db, m.ob), pc = fmaf(sk, m.dc, m.oc);
                    const float r2 = fmaf(pc, pc, fmaf(pb, pb, pa * pa));
                    in[i] = (r2 <= f.R2f) & (k <= f.kmax);
                    bel[i] = below_seg<WIDE, true, MRTX_PATH_CP>(f, sg, sk, pa, pb, pc, r2);
                    sks[i] = sk;
#ifdef MRTX_PROF_MARGIN
                    {
                        const float uu = (sk - sg.sa) * f.inv_step;
                        float rw = fmaf(uu, fmaf(uu, sg.r2, sg.r1), sg.ra), cl = fmaf(uu, fmaf(uu, sg.c2, sg.c1), sg.ca);
                        if (sg.exact) { float q2; exact_rowcol(f, pa, pb, pc, rw, cl, q2); }
                        mrg[i] = sqrtf(r2) / (f.Rf * dem_march<WIDE>(f, rw, cl)) - 1.0f;
                    }
#endif
                }
                bool act = !bis;
                if (bis) {
                    bis_hi = bel[0] ? mid0 : bis_hi;
                    bis_lo = bel[0] ? bis_lo : mid0;
                    j--;
                    if (MRTX_PATH_STEPS > 1 && j > 0 && bel[0]) {           // the speculated level was the right one
                        const float mid1 = sks[1];                          // == 0.5f * (lo + hi) of the new bracket
                        bis_hi = bel[1] ? mid1 : bis_hi;
                        bis_lo = bel[1] ? bis_lo : mid1;
                        j--;
                    }
                    if (j <= 0) state = PS_HITWAIT;
                }
#pragma unroll
                for (int i = 0; i < MRTX_PATH_STEPS; i++) {
#if MRTX_PATH_MIP2
                    if (act && have[i]) {
                        // the steps the mask dropped between the last evaluated one and this one are spec steps all the same
                        if (STATS) { cnt[ST_HEIGHT] += count_in_steps<false>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, 0.0f, m.ka, j, js[i] - 1); }
                        if (STATS) { cnt[ST_HEIGHT] += in[i] ? 1u : 0u; cnt[ST_FETCH]++; }
                        todo &= todo - 1u;
                        j = js[i] + 1;
                        if (in[i] & bel[i]) { hit = true; sk_hit = sks[i]; state = PS_ENDED; act = false; }
                        else if (!in[i]) { state = PS_ENDED; act = false; }
                        else if (todo == 0u) { segend = true; act = false; }
                    }
#else
                    if (act) {
                        if (STATS) { cnt[ST_HEIGHT] += in[i] ? 1u : 0u; cnt[ST_FETCH]++; }
#ifdef MRTX_PROF_MARGIN
                        if (STATS) { cnt[ST_MALL]++; cnt[ST_M1] += mrg[i] > 1.0e-4f; cnt[ST_M2] += mrg[i] > 3.0e-4f; cnt[ST_M3] += mrg[i] > 1.0e-3f; }
#endif
                        j++;
                        if (in[i] & bel[i]) { hit = true; sk_hit = sks[i]; state = PS_ENDED; act = false; }
                        else if (!in[i]) { state = PS_ENDED; act = false; }
                        else if (j > sg.jhi) { segend = true; act = false; }
                    }
#endif
                }
            }
        }

        PPROF_T(11);
        // ---- end of a segment that is neither hit nor left: did the ray end inside the skipped tail?
        if (segend) {
            // STATS: the steps after the last evaluated one (j = the step after it, or jlo when none was evaluated)
            if (STATS) cnt[ST_HEIGHT] += count_in_steps<false>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, 0.0f, m.ka, max(MRTX_PATH_MIP2 ? j : sg.jhi + 1, 1), SEG_N);
            const int k = m.ka + SEG_N;
            const float sk = (float)k * f.step;
            const float pa = fmaf(sk, m.da, m.oa), pb = fmaf(sk, m.db, m.ob), pc = fmaf(sk, m.dc, m.oc);
            bool go = (fmaf(pc, pc, fmaf(pb, pb, pa * pa)) <= f.R2f) & (k < f.kmax);
            if (go && k >= m.kend) {                         // cut by the horizon bound (see horizon_kend)
                if (STATS) cnt[ST_HEIGHT] += steps_after(f, m, k + 1);
                go = false;
            }
            m.ka = k;
            state = go ? PS_NEEDSEG : PS_ENDED;
        }

        // ---- march over
        if (state == PS_ENDED) {
#if MRTX_PATH_PARK
            if (shadow) {
                wgt = hit ? 0.0f : PKS(2, 1);
                state = PS_SHADE;
            } else if (hit) {
                if (!have_c) { float a, b, c_; c_load(pq, e, a, b, c_); COLD_SET_C(a, b, c_); have_c = true; }
                const int bk = (int)rintf(sk_hit * f.inv_step);
                bis_lo = (float)(bk - 1) * f.step;       // bis_hi is sk_hit already
                j = f.nbis;
                if (STATS) { cnt[ST_HEIGHT] += (uint32_t)f.nbis; cnt[ST_FETCH] += (uint32_t)f.nbis; }
                state = f.nbis > 0 ? PS_BISECT : PS_HITWAIT;
            } else if (CF(f)->bg) {
                state = PS_ESCAPED;
            } else {
                float e0, e1, e2, a = 0.0f, b = 0.0f, c_ = 0.0f;
                bool in_reg = false;
                if (escaped_radiance<STATS>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, e0, e1, e2, cnt)) {   // the Sun disk
                    if (!have_c) { c_load(pq, e, a, b, c_); have_c = true; } else { COLD_GET_C(a, b, c_); }
                    float t0, t1, t2;
                    COLD_GET_T(t0, t1, t2);
                    a = fmaf(t0, e0, a); b = fmaf(t1, e1, b); c_ = fmaf(t2, e2, c_);
                    in_reg = true;
                }
                if (have_c) { if (!in_reg) COLD_GET_C(a, b, c_); c_store(pq, e, a, b, c_); }
                state = PS_IDLE;
            }
#else
            if (shadow) {
                wgt = hit ? 0.0f : carried;
                state = PS_SHADE;
            } else if (hit) {
                // D3: bracket the crossing between the last step above and the first step at/below the surface; the sample's
                // radiance so far is fetched now, to arrive while the bisection runs
                if (!have_c) { c_load(pq, e, c0, c1, c2); have_c = true; }
                const int bk = (int)rintf(sk_hit * f.inv_step);
                bis_lo = (float)(bk - 1) * f.step;       // bis_hi is sk_hit already
                j = f.nbis;
                if (STATS) { cnt[ST_HEIGHT] += (uint32_t)f.nbis; cnt[ST_FETCH] += (uint32_t)f.nbis; }
                state = f.nbis > 0 ? PS_BISECT : PS_HITWAIT;
            } else if (CF(f)->bg) {
                // the path left the Moon and an environment map is bound: its look-up (a dependent fetch from a large
                // texture) waits with the rare steps instead of stalling every iteration
                state = PS_ESCAPED;
            } else {
                float e0, e1, e2;
                if (escaped_radiance<STATS>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, e0, e1, e2, cnt)) {   // the Sun disk
                    if (!have_c) { c_load(pq, e, c0, c1, c2); have_c = true; }
                    c0 = fmaf(t0r, e0, c0); c1 = fmaf(t1r, e1, c1); c2 = fmaf(t2r, e2, c2);
                }
                if (have_c) c_store(pq, e, c0, c1, c2);
                state = PS_IDLE;
            }
#endif
        }

        // ---- the rare steps (~8 % of the paths reach them): a continuation ray that hit terrain gets its vertex and
        // light sample (~600 VALU); a vertex whose shadow ray is through gets its direct term, and the path is
        // continued or ended (~250 VALU).  They wait until enough lanes need them -- or nothing is marching.
        PPROF_T(12);
        if (do_rare) {
#if MRTX_PATH_PARK
            if (state == PS_ESCAPED) {
                float e0, e1, e2, a = 0.0f, b = 0.0f, c_ = 0.0f;
                bool in_reg = false;
                if (escaped_radiance<STATS>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, e0, e1, e2, cnt)) {
                    float t0, t1, t2;
                    COLD_GET_T(t0, t1, t2);
                    // a black texel adds nothing (fmaf(t, 0, c) == c for finite t): the sample's radiance need not be touched
                    const bool nothing = (e0 == 0.0f) & (e1 == 0.0f) & (e2 == 0.0f) & ((t0 + t1 + t2) < __builtin_inff());
                    if (!nothing) {
                        if (!have_c) { c_load(pq, e, a, b, c_); have_c = true; } else { COLD_GET_C(a, b, c_); }
                        a = fmaf(t0, e0, a); b = fmaf(t1, e1, b); c_ = fmaf(t2, e2, c_);
                        in_reg = true;
                    }
                }
                if (have_c) { if (!in_reg) COLD_GET_C(a, b, c_); c_store(pq, e, a, b, c_); }
                state = PS_IDLE;
            }
            if (state == PS_HITWAIT) {
                const float blo = bis_lo;                  // the bisected bracket's upper side (PS_BISECT)
                Vertex v;
                hit_vertex<STATS, WIDE>(f, fmaf(blo, m.da, m.oa), fmaf(blo, m.db, m.ob), fmaf(blo, m.dc, m.oc), v, cnt);
                PKS(0, 0) = v.pa; PKS(0, 1) = v.pb; PKS(0, 2) = v.pc; PKS(0, 3) = v.na;
                PKS(1, 0) = v.nb; PKS(1, 1) = v.nc; PKS(1, 2) = v.al0; PKS(1, 3) = v.al1; PKS(2, 0) = v.al2;
                seg++;
                const uint32_t d0 = 4u + 5u * (seg - 2u);   // the dimensions drawn when this segment was started
                const float ul1 = u01(ks, d0 + 3u), ul2 = u01(ks, d0 + 4u);
                float soa, sob, soc, swa, swb, swc, carried;
                if (light_sample(f, v, ul1, ul2, soa, sob, soc, swa, swb, swc, carried)) {
                    PKS(2, 1) = carried;
                    if (STATS) cnt[ST_SHADOW]++;
                    const bool go = march_begin<false, STATS, true>(f, soa, sob, soc, swa, swb, swc, m, cnt);
                    hit = false; shadow = true;
                    state = go ? PS_NEEDSEG : PS_ENDED;
                } else {
                    wgt = 0.0f;
                    state = PS_SHADE;
                }
            }
            park_fence();       // the vertex a lane parked above is READ BACK below (not kept in registers across the blocks)
            if (state == PS_SHADE) {
                Vertex v;
                v.pa = PKS(0, 0); v.pb = PKS(0, 1); v.pc = PKS(0, 2); v.na = PKS(0, 3);
                v.nb = PKS(1, 0); v.nc = PKS(1, 1); v.al0 = PKS(1, 2); v.al1 = PKS(1, 3); v.al2 = PKS(2, 0);
                float t0r, t1r, t2r, c0, c1, c2;
                COLD_GET_T(t0r, t1r, t2r);
                COLD_GET_C(c0, c1, c2);
                c0 = fmaf(t0r * v.al0, wgt, c0);
                c1 = fmaf(t1r * v.al1, wgt, c1);
                c2 = fmaf(t2r * v.al2, wgt, c2);
                float boa, bob, boc, bda, bdb, bdc;
                if (continue_path(f, v, ks, seg, t0r, t1r, t2r, boa, bob, boc, bda, bdb, bdc)) {
                    COLD_SET_T(t0r, t1r, t2r);
                    COLD_SET_C(c0, c1, c2);
                    if (STATS) cnt[ST_BOUNCE]++;
                    const bool go = march_begin<false, STATS, true>(f, boa, bob, boc, bda, bdb, bdc, m, cnt);
                    hit = false; shadow = false;
                    state = go ? PS_NEEDSEG : PS_ENDED;
                } else {
                    c_store(pq, e, c0, c1, c2);
                    state = PS_IDLE;
                }
            }
#else
            if (state == PS_ESCAPED) {
                float e0, e1, e2;
                if (escaped_radiance<STATS>(f, m.oa, m.ob, m.oc, m.da, m.db, m.dc, e0, e1, e2, cnt)) {
                    // a black texel adds nothing (fmaf(t, 0, c) == c for finite t): the sample's radiance need not be touched
                    const bool nothing = (e0 == 0.0f) & (e1 == 0.0f) & (e2 == 0.0f) & ((t0r + t1r + t2r) < __builtin_inff());
                    if (!nothing) {
                        if (!have_c) { c_load(pq, e, c0, c1, c2); have_c = true; }
                        c0 = fmaf(t0r, e0, c0); c1 = fmaf(t1r, e1, c1); c2 = fmaf(t2r, e2, c2);
                    }
                }
                if (have_c) c_store(pq, e, c0, c1, c2);
                state = PS_IDLE;
            }
            if (state == PS_HITWAIT) {
                const float blo = bis_lo;                  // the bisected bracket's upper side (PS_BISECT)
                hit_vertex<STATS, WIDE>(f, fmaf(blo, m.da, m.oa), fmaf(blo, m.db, m.ob), fmaf(blo, m.dc, m.oc), v, cnt);
                seg++;
                const uint32_t d0 = 4u + 5u * (seg - 2u);   // the dimensions drawn when this segment was started
                const float ul1 = u01(ks, d0 + 3u), ul2 = u01(ks, d0 + 4u);
                float soa, sob, soc, swa, swb, swc;
                if (light_sample(f, v, ul1, ul2, soa, sob, soc, swa, swb, swc, carried)) {
                    if (STATS) cnt[ST_SHADOW]++;
                    const bool go = march_begin<false, STATS, true>(f, soa, sob, soc, swa, swb, swc, m, cnt);
                    hit = false; shadow = true;
                    state = go ? PS_NEEDSEG : PS_ENDED;
                } else {
                    wgt = 0.0f;
                    state = PS_SHADE;
                }
            }
            if (state == PS_SHADE) {
                c0 = fmaf(t0r * v.al0, wgt, c0);
                c1 = fmaf(t1r * v.al1, wgt, c1);
                c2 = fmaf(t2r * v.al2, wgt, c2);
                float boa, bob, boc, bda, bdb, bdc;
                if (continue_path(f, v, ks, seg, t0r, t1r, t2r, boa, bob, boc, bda, bdb, bdc)) {
                    if (STATS) cnt[ST_BOUNCE]++;
                    const bool go = march_begin<false, STATS, true>(f, boa, bob, boc, bda, bdb, bdc, m, cnt);
                    hit = false; shadow = false;
                    state = go ? PS_NEEDSEG : PS_ENDED;
                } else {
                    c_store(pq, e, c0, c1, c2);
                    state = PS_IDLE;
                }
            }
#endif
        }
        PPROF_T(13);
    }

#ifdef MRTX_PATH_PROF
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) atomicAdd(&g_pprof[i], (unsigned long long)pf[i]);
        atomicMax(&g_pprof_t[label], (unsigned long long)wall_clock64());
        if (blockIdx.x < 8192u) g_pprof_end[blockIdx.x] = (unsigned long long)wall_clock64();
    }
#endif
    if (STATS) {
#pragma unroll
        for (int i = 0; i < ST_N; i++) {
            uint32_t t = cnt[i];
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) t += __shfl_xor(t, k, 64);
            if (lane == 0 && t != 0u) atomicAdd(&CF(f)->stats_paths[i], (unsigned long long)t);
        }
    }
}

// The radiance sums of the deferred pixels: the S final sample values of a pixel added in the butterfly order of the
// spec (block sum), then onto the running sum -- exactly what render_kernel does in registers.
#ifndef MRTX_RESOLVE_U
#define MRTX_RESOLVE_U 4      // chunks a wave keeps in flight (4 KB of sample values); the launch sizes its grid with it
#endif
template <int S>
__global__ void __launch_bounds__(256) resolve_paths_kernel(const FrameC f, const PathQ pq) {
    // A wave takes U consecutive chunks at a time: their meta words in one round, then every load of the deferred ones (sample
    // values and running sums) in a second -- the kernel is a pure stream, bound by the bytes it keeps in flight.
    constexpr uint32_t U = MRTX_RESOLVE_U;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t pp = lane >> pq.s_log2, ss = lane & ((1u << pq.s_log2) - 1u);
    for (uint32_t base = (blockIdx.x * 4u + wv) * U; base < pq.n_chunks; base += gridDim.x * 4u * U) {
        uint32_t mt[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) mt[u] = base + u < pq.n_chunks ? pq.meta[base + u] : 0u;
        float a0[U], a1[U], a2[U];
        float4 acc[U];
        float4* ap[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            ap[u] = nullptr;
            a0[u] = a1[u] = a2[u] = 0.0f;
            acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mt[u] & 0x80000000u) {                       // wave-uniform
                const uint32_t e = (base + u) * 64u + lane;
#if MRTX_C_AOS == 2
                { const float* const cp = reinterpret_cast<const float*>(pq.c4) + 3u * (size_t)e;
                  a0[u] = __builtin_nontemporal_load(cp); a1[u] = __builtin_nontemporal_load(cp + 1); a2[u] = __builtin_nontemporal_load(cp + 2); }
#elif MRTX_C_AOS
                { const float4 v = nt_load4(pq.c4 + e); a0[u] = v.x; a1[u] = v.y; a2[u] = v.z; }
#else
                a0[u] = __builtin_nontemporal_load(pq.c0 + e);
                a1[u] = __builtin_nontemporal_load(pq.c1 + e);
                a2[u] = __builtin_nontemporal_load(pq.c2 + e);
#endif
                const uint32_t x = (mt[u] & 0x7FFFu) + (pp & ((1u << pq.pw_log2) - 1u));
                const uint32_t y = ((mt[u] >> 15) & 0x7FFFu) + (pp >> pq.pw_log2);
                if (ss == 0u && x < (uint32_t)f.W && y < (uint32_t)f.H) {
                    ap[u] = reinterpret_cast<float4*>(CF(f)->accum) + ((int64_t)y * f.W + x);
                    acc[u] = *ap[u];
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (mt[u] & 0x80000000u) {
                const float t0 = tree_sum<S>(a0[u]), t1 = tree_sum<S>(a1[u]), t2 = tree_sum<S>(a2[u]);
                if (ap[u]) {
                    float4 t = acc[u];
                    t.x += t0; t.y += t1; t.z += t2;
                    *ap[u] = t;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// resolve: running sums -> mean linear radiance; "Gamma" post-process -> RGBA8
__global__ void resolve_linear_kernel(const float4* __restrict__ accum, float4* __restrict__ out, int64_t n,
                                      uint32_t nsamples) {
    const float ns = (float)nsamples;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = accum[i];
        out[i] = nsamples ? make_float4(a.x / ns, a.y / ns, a.z / ns, a.w / ns) : make_float4(0, 0, 0, 0);
    }
}
// "Gamma" post-process, exact: level = number of thresholds T[1..N] (N = 255 or 65535) that are
// <= x = exposure * mean, with T[j] = (float)pow((j - 0.5) / N, gamma) computed once on the host in float64 -- the level
// round(N * x^(1/gamma)) without a device transcendental, so the byte the GUI shows is the oracle's byte.  T is
// non-decreasing; a NaN or negative x compares false everywhere (level 0).  BITS steps of a branch-free bisection.
template <int BITS>
__device__ __forceinline__ uint32_t tone_level(float x, const float* T) {
    uint32_t k = 0;
#pragma unroll
    for (int b = BITS - 1; b >= 0; b--) {
        const uint32_t j = k + (1u << b);
        k = (x >= T[j]) ? j : k;
    }
    return k;
}
// D12 "Overlay" post-process (renderer_video.py:15-27, :137-144): a frame-sized RGBA8 texture blended over the
// tone-mapped 8-bit image with exact alpha compositing, out = round((src*(255-a) + ov*a) / 255)
// -- an opaque black patch gives 0, a 50 % (a = 128) black patch over 46 gives 23.
__device__ __forceinline__ uint32_t blend8(uint32_t src, uint32_t ov, uint32_t a) {
    return (src * (255u - a) + ov * a + 127u) / 255u;
}
__global__ void resolve_rgba8_kernel(const float4* __restrict__ accum, uint32_t* __restrict__ out, int64_t n,
                                     uint32_t nsamples, float expo, const float* __restrict__ T8,
                                     const uint32_t* __restrict__ overlay) {
    __shared__ float T[256];
    T[threadIdx.x & 255] = T8[threadIdx.x & 255];       // blockDim.x == 256
    __syncthreads();
    const float ns = (float)nsamples;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = accum[i];
        uint32_t r = 0, g = 0, b = 0;
        if (nsamples) {
            r = tone_level<8>(expo * (a.x / ns), T); g = tone_level<8>(expo * (a.y / ns), T); b = tone_level<8>(expo * (a.z / ns), T);
        }
        if (overlay) {
            const uint32_t o = overlay[i], al = o >> 24;
            r = blend8(r, o & 255u, al); g = blend8(g, (o >> 8) & 255u, al); b = blend8(b, (o >> 16) & 255u, al);
        }
        out[i] = 0xFF000000u | r | (g << 8) | (b << 16);
    }
}
// save_image(..., bps="Bps16") (renderer_dialogs.py:1222-1224): the same post-process at 16 bits per sample, RGB interleaved;
// T16 has 65536 entries (256 KB, L2 resident); no overlay at 16 bits (the reference composites video frames at 8 bits only).
__global__ void resolve_rgb16_kernel(const float4* __restrict__ accum, uint16_t* __restrict__ out, int64_t n,
                                     uint32_t nsamples, float expo, const float* __restrict__ T16) {
    const float ns = (float)nsamples;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = accum[i];
        uint32_t r = 0, g = 0, b = 0;
        if (nsamples) {
            r = tone_level<16>(expo * (a.x / ns), T16); g = tone_level<16>(expo * (a.y / ns), T16); b = tone_level<16>(expo * (a.z / ns), T16);
        }
        out[3 * i] = (uint16_t)r; out[3 * i + 1] = (uint16_t)g; out[3 * i + 2] = (uint16_t)b;
    }
}

// ------------------------------------------------------------------------------------------------
// multi-GPU exchange: pack the tiles a rank owns into a dense buffer / scatter a peer's buffer back.
// Layout of a packed shard: per slot one tile of float4 sums followed by one tile of float4 hits (so any range of slots
// is one contiguous piece: the exchange moves the shard in parts while later parts still render); slot k of
// rank r is local tile k (full layout, list == nullptr) or local tile list[k] (active layout: only tiles the sky
// cull kept, -1 = padding); local tile lt of rank r is tile lt*world + r.  Whole float4 (16 B/lane, coalesced).
__global__ void pack_shard_kernel(const float4* __restrict__ accum, const float4* __restrict__ hits,
                                  float4* __restrict__ dst, int W, int H, int tile_w, int tile_h, int tiles_x,
                                  int n_tiles, int rank, int world, int slot0, int slots, const int32_t* __restrict__ list,
                                  int shift, int with_hits) {
    const int tile_px = tile_w * tile_h;
    const int64_t total = (int64_t)slots * tile_px;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = slot0 + (int)(i / tile_px), r = (int)(i % tile_px);
        const int lt = list ? list[slot] : slot;
        const int t = lt * world + rank;
        float4 a = make_float4(0, 0, 0, 0), h = a;
        if (lt >= 0 && t < n_tiles) {
            int tx, ty;
            mrtx_tile_xy(t, tiles_x, shift, tx, ty);
            const int x = tx * tile_w + r % tile_w, y = ty * tile_h + r / tile_w;
            if (x < W && y < H) { a = accum[(int64_t)y * W + x]; if (with_hits) h = hits[(int64_t)y * W + x]; }
        }
        const int64_t o = (int64_t)slot * (with_hits ? 2 : 1) * tile_px + r;     // a slot = sums tile [+ hits tile]
        dst[o] = a;
        if (with_hits) dst[o + tile_px] = h;
    }
}
__global__ void unpack_shard_kernel(float4* __restrict__ accum, float4* __restrict__ hits,
                                    const float4* __restrict__ src, int W, int H, int tile_w, int tile_h,
                                    int tiles_x, int n_tiles, int src_rank, int world, int slots,
                                    const int32_t* __restrict__ list, int shift, int with_hits) {
    const int tile_px = tile_w * tile_h;
    const int64_t total = (int64_t)slots * tile_px;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i / tile_px), r = (int)(i % tile_px);
        const int lt = list ? list[slot] : slot;
        const int t = lt * world + src_rank;
        if (lt < 0 || t >= n_tiles) continue;
        int tx, ty;
        mrtx_tile_xy(t, tiles_x, shift, tx, ty);
        const int x = tx * tile_w + r % tile_w, y = ty * tile_h + r / tile_w;
        const int64_t o = (int64_t)slot * (with_hits ? 2 : 1) * tile_px + r;
        if (x < W && y < H) { accum[(int64_t)y * W + x] = src[o]; if (with_hits) hits[(int64_t)y * W + x] = src[o + tile_px]; }
    }
}

// ... every peer's shard in ONE launch (the root of an 8-GPU job has seven to scatter, 6 MB each: seven launches are ~50 us of gaps
// in a 3.3 ms step).  The source pointers travel by value in the kernel arguments (no upload); list_all = world x slots tile lists
// (active layout) or null.
struct UnpackSrcs { const float4* p[16]; };
__global__ void unpack_all_kernel(float4* __restrict__ accum, float4* __restrict__ hits, const UnpackSrcs srcs, int W, int H, int tile_w,
                                  int tile_h, int tiles_x, int n_tiles, int self, int world, int slots,
                                  const int32_t* __restrict__ list_all, int shift, int with_hits) {
    const int tile_px = tile_w * tile_h;
    const int64_t per = (int64_t)slots * tile_px, total = per * world;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int src_rank = (int)(i / per);
        if (src_rank == self) continue;
        const int64_t k = i - (int64_t)src_rank * per;
        const int slot = (int)(k / tile_px), r = (int)(k % tile_px);
        const int lt = list_all ? list_all[(int64_t)src_rank * slots + slot] : slot;
        const int t = lt * world + src_rank;
        if (lt < 0 || t >= n_tiles) continue;
        int tx, ty;
        mrtx_tile_xy(t, tiles_x, shift, tx, ty);
        const int x = tx * tile_w + r % tile_w, y = ty * tile_h + r / tile_w;
        const float4* src = srcs.p[src_rank];
        const int64_t o = (int64_t)slot * (with_hits ? 2 : 1) * tile_px + r;
        if (x < W && y < H) { accum[(int64_t)y * W + x] = src[o]; if (with_hits) hits[(int64_t)y * W + x] = src[o + tile_px]; }
    }
}

// rank 0, active layout: tiles of other ranks that held data of an earlier view and are sky in this one
__global__ void zero_tiles_kernel(float4* __restrict__ accum, float4* __restrict__ hits, const int32_t* __restrict__ tiles,
                                  int n, int W, int H, int tile_w, int tile_h, int tiles_x, int shift) {
    const int tile_px = tile_w * tile_h;
    const int64_t total = (int64_t)n * tile_px;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = tiles[i / tile_px], r = (int)(i % tile_px);
        int tx, ty;
        mrtx_tile_xy(t, tiles_x, shift, tx, ty);
        const int x = tx * tile_w + r % tile_w, y = ty * tile_h + r / tile_w;
        if (x < W && y < H) { accum[(int64_t)y * W + x] = make_float4(0, 0, 0, 0); hits[(int64_t)y * W + x] = make_float4(0, 0, 0, 0); }
    }
}

// ------------------------------------------------------------------------------------------------
// ingest: data_loader.py:166-247 on the device.  One thread per output texel; the d x d int16 block
// is read as d row segments (for d <= 8 a segment is <= 16 B).  Integer sums of <= 8 int16 are exact
// in float32, so stage 1 is order-free; stage 2 adds the d row means in row order.
__global__ void ldem_block_mean_kernel(const int16_t* __restrict__ src, float* __restrict__ dst, int h, int w,
                                       int d, unsigned int* __restrict__ max_bits) {
    const float scale = (float)(0.5 / 1737400.0);
    const int64_t n = (int64_t)h * w, Wsrc = (int64_t)w * d;
    float mx = 0.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / w), c = (int)(i % w);
        float v;
        if (d == 1) {
            v = (float)src[i] * scale;
        } else {
            float acc2 = 0.0f;
            for (int ii = 0; ii < d; ii++) {
                const int16_t* p = src + ((int64_t)r * d + ii) * Wsrc + (int64_t)c * d;
                float acc = 0.0f;
                for (int j = 0; j < d; j++) acc += (float)p[j];
                acc = acc / (float)d;
                acc2 = (ii == 0) ? acc : acc2 + acc;
            }
            v = (acc2 / (float)d) * scale;
        }
        v += 1.0f;
        dst[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(mx));  // v > 0: uint order == float order
}
__global__ void scale_by_inv_kernel(float* __restrict__ dst, int64_t n, const unsigned int* __restrict__ max_bits) {
    const float mx = __uint_as_float(*max_bits);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = dst[i] / mx;
}

// ------------------------------------------------------------------------------------------------
// synthetic inputs (bench / test data): fractal value noise on the sphere +
// hashed crater field, in LDEM units (0.5 m), clipped to the LOLA-like range.
__device__ __forceinline__ float hash01(int x, int y, int z, uint32_t seed) {
    const uint32_t h = mix32((uint32_t)x * 0x8da6b343u ^ (uint32_t)y * 0xd8163841u ^ (uint32_t)z * 0xcb1ab31fu ^ seed);
    return (float)(h >> 8) * 5.9604644775390625e-08f;
}
__device__ float vnoise3(float x, float y, float z, uint32_t seed) {
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = x - fx, ty = y - fy, tz = z - fz;
    tx = tx * tx * (3.f - 2.f * tx); ty = ty * ty * (3.f - 2.f * ty); tz = tz * tz * (3.f - 2.f * tz);
    float c[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) c[i][j][k] = hash01(ix + i, iy + j, iz + k, seed);
    const float x00 = c[0][0][0] + tx * (c[1][0][0] - c[0][0][0]), x10 = c[0][1][0] + tx * (c[1][1][0] - c[0][1][0]);
    const float x01 = c[0][0][1] + tx * (c[1][0][1] - c[0][0][1]), x11 = c[0][1][1] + tx * (c[1][1][1] - c[0][1][1]);
    const float y0 = x00 + ty * (x10 - x00), y1 = x01 + ty * (x11 - x01);
    return 2.0f * (y0 + tz * (y1 - y0)) - 1.0f;
}
__device__ float synth_height_km(float px, float py, float pz, int octaves, int levels, uint32_t seed) {
    // LOLA-like hypsometry: hemispheric dichotomy (far-side highlands / near-side lowlands), rare high
    // massifs reaching ~ +10 km, fractal relief with std ~ 2 km, craters; mean near 0 km.
    float hkm = 0.9f + 1.9f * vnoise3(px * 1.1f + 5.3f, py * 1.1f + 1.7f, pz * 1.1f - 8.1f, seed ^ 0xA511E9B3u);
    {
        const float m = vnoise3(px * 3.3f - 2.2f, py * 3.3f + 6.1f, pz * 3.3f + 0.4f, seed ^ 0x63D83595u);
        const float mp = fmaxf(m, 0.0f);
        hkm += 14.0f * mp * mp * mp;
    }
    float amp = 2.0f, fr = 2.0f;
    for (int o = 0; o < octaves; o++) {
        hkm += amp * vnoise3(px * fr + 17.3f, py * fr - 4.1f, pz * fr + 9.7f, seed + 101u * (uint32_t)o);
        fr *= 2.0f; amp *= 0.56f;
    }
    float cell = 0.5f;
    for (int l = 0; l < levels; l++) {
        const float inv = 1.0f / cell;
        const int cx = (int)floorf(px * inv), cy = (int)floorf(py * inv), cz = (int)floorf(pz * inv);
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int ix = cx + dx, iy = cy + dy, iz = cz + dz;
                    const uint32_t s2 = seed ^ (0x51ed27u * (uint32_t)(l + 1));
                    if (hash01(ix, iy, iz, s2) > 0.55f) continue;
                    float qx = ((float)ix + hash01(ix, iy, iz, s2 + 1u)) * cell;
                    float qy = ((float)iy + hash01(ix, iy, iz, s2 + 2u)) * cell;
                    float qz = ((float)iz + hash01(ix, iy, iz, s2 + 3u)) * cell;
                    const float ql = sqrtf(qx * qx + qy * qy + qz * qz);
                    if (fabsf(ql - 1.0f) > 0.5f * cell) continue;
                    qx /= ql; qy /= ql; qz /= ql;
                    const float rad = cell * (0.12f + 0.28f * hash01(ix, iy, iz, s2 + 4u));
                    const float ex = px - qx, ey = py - qy, ez = pz - qz;
                    const float tt = sqrtf(ex * ex + ey * ey + ez * ez) / rad;
                    if (tt >= 1.4f) continue;
                    const float rad_km = rad * 1737.4f;
                    const float depth = fminf(0.4f * rad_km, 2.5f + 0.006f * rad_km);
                    const float rim = 0.22f * depth;
                    if (tt < 1.0f) hkm += -depth * (1.0f - tt * tt) + rim * tt * tt;
                    else { const float e = (1.4f - tt) * 2.5f; hkm += rim * e * e; }
                }
        cell *= 0.5f;
    }
    return hkm;
}
__global__ void synth_ldem_kernel(int16_t* __restrict__ dst, int h, int w, int octaves, int levels, uint32_t seed) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / w), c = (int)(i % w);
        const float lat = (0.5f - ((float)r + 0.5f) / (float)h) * 3.14159265f;
        const float lon = (((float)c + 0.5f) / (float)w - 0.5f) * 6.2831853f;
        const float cl = cosf(lat);
        const float hk = synth_height_km(cl * sinf(lon), -cl * cosf(lon), sinf(lat), octaves, levels, seed);
        float units = hk * 2000.0f;
        units = fminf(fmaxf(units, -18200.0f), 21600.0f);
        dst[i] = (int16_t)(int)rintf(units);
    }
}
__global__ void synth_color_kernel(uint32_t* __restrict__ dst, int h, int w, uint32_t seed) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / w), c = (int)(i % w);
        const float lat = (0.5f - ((float)r + 0.5f) / (float)h) * 3.14159265f;
        const float lon = (((float)c + 0.5f) / (float)w - 0.5f) * 6.2831853f;
        const float cl = cosf(lat);
        const float px = cl * sinf(lon), py = -cl * cosf(lon), pz = sinf(lat);
        float v = 0.0f, amp = 0.5f, fr = 1.5f;
        for (int o = 0; o < 9; o++) {
            v += amp * vnoise3(px * fr + 3.1f, py * fr + 7.7f, pz * fr - 2.9f, seed + 977u * (uint32_t)o);
            fr *= 2.1f; amp *= 0.6f;
        }
        // source byte 0..255 -> the reference's albedo range 0.2 + 0.75 v/255, then ^gamma 2.2, as bytes
        // (data_loader.py:261-287); small per-channel tint so the three channels differ.
        const float sv = fminf(fmaxf(0.5f + 0.55f * v, 0.0f), 1.0f);
        const float alb = 0.2f + 0.75f * sv;
        const uint32_t R = (uint32_t)(255.0f * powf(alb, 2.2f));
        const uint32_t G = (uint32_t)(255.0f * powf(alb * 0.985f, 2.2f));
        const uint32_t B = (uint32_t)(255.0f * powf(alb * 0.955f, 2.2f));
        dst[i] = R | (G << 8) | (B << 16) | 0xFF000000u;
    }
}

__global__ void probe_latlon_kernel(const float* a, const float* b, const float* c, float* lat, float* lon, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) latlon(a[i], b[i], c[i], fmaf(b[i], b[i], a[i] * a[i]), lat[i], lon[i]);
}

// Exhaustive check of the domain-restricted primitives against the compiler's IEEE expansions, on the device, for the bit patterns
// [lo, lo + n): which = 0: rcp_nr<1>(x) vs 1.0f / x; 1: rcp_nr<2>(x); 2: sqrt_cr(x) vs sqrtf(x).  out[0] = mismatches, out[1] = the
// smallest mismatching bit pattern + 1 (0 = none).
__global__ void probe_cr_kernel(uint32_t lo, uint64_t n, int which, unsigned long long* __restrict__ out) {
    unsigned long long bad = 0, first = ~0ull;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = lo + (uint32_t)i;
        const float x = __uint_as_float(bits);
        float a, b;
        if (which == 2) { a = sqrtf(x); b = sqrt_cr(x); }
        else { a = 1.0f / x; b = which == 0 ? rcp_nr<1>(x) : rcp_nr<2>(x); }
        if (__float_as_uint(a) != __float_as_uint(b)) { bad++; first = first < bits ? first : (unsigned long long)bits; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        bad += __shfl_xor(bad, m, 64);
        const unsigned long long o = __shfl_xor(first, m, 64);
        first = first < o ? first : o;
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(&out[0], bad);
        if (first != ~0ull) atomicMin(&out[1], first + 1ull);
    }
}

// DEM (h, w) row-major -> padded (h+4, w+4): rows clamp, columns wrap (see dem_march)
__global__ void pad_dem_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w) {
    const int pitch = w + 4;
    const int64_t n = (int64_t)(h + 4) * pitch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / pitch) - 2, c = (int)(i % pitch) - 2;
        const int r1 = r + 1 < 0 ? 0 : (r + 1 > h - 1 ? h - 1 : r + 1);
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        c = c < 0 ? c + w : (c > w - 1 ? c - w : c);
#if MRTX_DEM_PAIRS
        reinterpret_cast<float2*>(dst)[i] = make_float2(src[(int64_t)r * w + c], src[(int64_t)r1 * w + c]);
#else
        (void)r1;
        dst[i] = src[(int64_t)r * w + c];
#endif
    }
}

// colour map (h, w) RGBA8 -> row pairs (h+1, w+4): element (r, c) = (T[max(r,0)][wrap(c)], T[min(r+1,h-1)][wrap(c)]) for
// r in [-1, h-1], c in [-2, w+1] -- exactly the taps the bilinear colour fetch of hit_vertex() defines
__global__ void color_pair_kernel(const uint32_t* __restrict__ src, uint2* __restrict__ dst, int h, int w) {
    const int pitch = w + 4;
    const int64_t n = (int64_t)(h + 1) * pitch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / pitch) - 1;
        const int c = wrapc((int)(i % pitch) - 2, w);
        const int ra = r < 0 ? 0 : r, rb = r + 1 > h - 1 ? h - 1 : r + 1;
        dst[i] = make_uint2(src[(int64_t)ra * w + c], src[(int64_t)rb * w + c]);
    }
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section): stream a buffer once with this kernel's own access width
// (one 8-byte load per lane) so the PMC reading can be compared with a known byte count.
__global__ void probe_stream_kernel(const Pair* __restrict__ src, int64_t n_pairs, float* __restrict__ out) {
    float acc = 0.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        const Pair p = src[i];
        acc += p.x + p.y;
    }
    if (acc == 123456.789f) out[0] = acc;   // keep the loads alive
}

// max-mip of the padded DEM with cells of C = 2^shift texels: cell (i, j) = max over texel rows [C i-2, C i+C+1]
// (clamped) x columns [C j-2, C j+C+1] (wrapped) -- dilated by the two-texel border so that any bilinear tap whose
// indices land in a cell is covered.  Stored with a one-cell border of its own: (mh+2) x (mw+2), rows clamp,
// columns wrap.  C is chosen on the host so that a 16-step segment spans at most two cells per axis.
__global__ void mip_build_kernel(const float* __restrict__ dem_padded, int h, int w, float* __restrict__ mip, int mh,
                                 int mw, int shift) {
    const int pitch = w + 4, mp = mw + 2, C = 1 << shift;
    const int n = (mh + 2) * mp;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        int i = t / mp - 1, j = t % mp - 1;
        i = i < 0 ? 0 : (i > mh - 1 ? mh - 1 : i);
        j = j < 0 ? mw - 1 : (j > mw - 1 ? 0 : j);
        const int r0 = max(C * i - 2, -2), r1 = min(C * i + C + 1, h + 1);
        const int c0 = max(C * j - 2, -2), c1 = min(C * j + C + 1, w + 1);
        float m = 0.0f;
        for (int r = r0; r <= r1; r++) {
            const float* row = dem_padded + ((int64_t)(r + 2) * pitch + 2) * (MRTX_DEM_ELEM_BYTES / 4);
            for (int c = c0; c <= c1; c++) m = fmaxf(m, row[c * (MRTX_DEM_ELEM_BYTES / 4)]);
        }
        mip[t] = m;
    }
}

// Horizon mip (horizon_kend): cell (i, j) of Cc = C << 3 texels holds the maximum of D over texel rows [Cc i - Cc, Cc i + 2 Cc)
// (clamped) x columns [Cc j - Cc, Cc j + 2 Cc) (wrapped), taken from the plain max-mip of C-texel cells (whose cells are
// themselves dilated by the two-texel tap border).  (hh) x (hw) floats, a few thousand cells.
__global__ void hmip_build_kernel(const float* __restrict__ mip, int mh, int mw, int shift, float* __restrict__ out,
                                  int hh, int hw, int hshift, int h, int w) {
    const int mp = mw + 2, Cc = 1 << hshift;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < hh * hw; t += gridDim.x * blockDim.x) {
        const int i = t / hw, j = t % hw;
        const int r0 = max(Cc * i - Cc, 0), r1 = min(Cc * i + 2 * Cc - 1, h - 1);
        const int c0 = Cc * j - Cc, c1 = Cc * j + 2 * Cc - 1;
        float m = 0.0f;
        // fine cells by INDEX, per unwrapped piece of the column range: stepping texel columns by C across the -180/+180
        // seam skips the last, partial fine cell mw - 1 whenever w is not a multiple of C (round-2 advisor finding)
        int a0 = 0, a1 = mw - 1, b0 = 1, b1 = 0;        // up to two inclusive fine-cell ranges; the second one empty
        if (c1 - c0 + 1 < w) {                           // else: the dilated range wraps around the whole map
            if (c0 < 0)       { a0 = (c0 + w) >> shift; a1 = mw - 1; b0 = 0; b1 = c1 >> shift; }
            else if (c1 >= w) { a0 = c0 >> shift; a1 = mw - 1; b0 = 0; b1 = (c1 - w) >> shift; }
            else              { a0 = c0 >> shift; a1 = c1 >> shift; }
        }
        for (int fi = r0 >> shift; fi <= (r1 >> shift); fi++) {
            for (int fj = a0; fj <= a1; fj++) m = fmaxf(m, mip[(fi + 1) * mp + fj + 1]);
            for (int fj = b0; fj <= b1; fj++) m = fmaxf(m, mip[(fi + 1) * mp + fj + 1]);
        }
        out[t] = m;
    }
}

// plain max-mip (mh+2) x (mw+2) -> row pairs: element (i, j) = (m[i][j], m[i+1][j]), the last row paired with itself
__global__ void mip_pair_kernel(const float* __restrict__ mip, float2* __restrict__ out, int rows, int pitch) {
    const int n = rows * pitch;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int r = t / pitch;
        out[t] = make_float2(mip[t], mip[r + 1 < rows ? t + pitch : t]);
    }
}

}  // namespace mrtx

#ifdef MRTX_DEV_ONE
// resource-usage / disassembly builds (tools/one_kernel.sh): ONE instantiation, device code only, seconds instead of minutes
#ifndef MRTX_DEV_ONE_MODE
#define MRTX_DEV_ONE_MODE 2
#endif
#if MRTX_DEV_ONE_MODE == 9
template __global__ void mrtx::path_kernel<false, true>(const FrameC, const PathQ);
#else
template __global__ void mrtx::render_kernel<64, false, true, MRTX_DEV_ONE_MODE, false>(const FrameC, const PathQ);
#endif
#else
// ------------------------------------------------------------------------------------------------
// launch wrappers (called from mrtx_api.hip)
extern "C++" {
// geometry of a render launch for S samples per pixel per wave: wave-jobs ("chunks") = grid x jobs per wave
static void render_geometry(const FrameC& f, int S, int& xcd_share, unsigned& grid, int& njobs, int& pw_log2, int mode = 0) {
    const int P = 64 / S;
    const int PW = P >= 32 ? 8 : P >= 8 ? 4 : P >= 2 ? 2 : 1;
    const int PH = P / PW;
    const int wgmin = MRTX_WG_WAVES > 2 ? 2 * PW : MRTX_WG_WAVES > 1 ? 2 * PH : PW;
    const int wgtile = mode == 3 ? MRTX_SKY_WG_TILE : MRTX_WG_TILE;
    const int wgt = (wgmin > wgtile) ? wgmin : wgtile;
    const int subs = (f.tile_w / wgt) * (f.tile_h / wgt);
    xcd_share = (MRTX_XCD_SHARE && (subs & 7) == 0 && f.n_active < MRTX_XCD_SHARE_BELOW) ? 1 : 0;   // see the remap in render_kernel
    const int groups = xcd_share ? f.n_active : (f.n_active + 8 * MRTX_XCD_TILE_RUN - 1) / (8 * MRTX_XCD_TILE_RUN) * (8 * MRTX_XCD_TILE_RUN);
    grid = (unsigned)(groups * subs);
    njobs = (wgt / PW) * (wgt / PH);
    pw_log2 = PW == 8 ? 3 : PW == 4 ? 2 : PW == 2 ? 1 : 0;
}
// records a MODE 2 (deferred-path) launch of this frame block hands over: one per lane of every chunk
uint64_t mrtx_path_chunks(const FrameC& f, int S, uint32_t* grid_a, int* njobs_log2) {
    int share, njobs, pwl; unsigned grid;
    render_geometry(f, S, share, grid, njobs, pwl);
    if (grid_a) *grid_a = grid;
    if (njobs_log2) *njobs_log2 = njobs == 2 ? 1 : 0;
    return (uint64_t)grid * (uint64_t)njobs;
}
// mode: 0 = direct light only, 1 = whole paths inside the wave, 2 = direct light + hand-over to path_kernel (pq),
// 3 = sky tiles (environment texel only)
hipError_t mrtx_launch_render(const FrameC& f, int S, bool stats, int mode, bool overlay, const PathQ* pq, hipStream_t st) {
    FrameC fr = f;
    int njobs, pwl; unsigned gx;
    render_geometry(f, S, fr.xcd_share, gx, njobs, pwl, mode);
    const dim3 grid(gx), block(64 * MRTX_WG_WAVES);
    if (f.n_blocks != 1) return hipErrorInvalidValue;   // one block of S samples per launch (see render_kernel)
    if (grid.x == 0) return hipSuccess;
    PathQ q;
    memset(&q, 0, sizeof q);
    if (mode == 2) {
        if (!pq || pq->n_chunks != (uint64_t)gx * (uint64_t)njobs || pq->grid_a != gx || (1 << pq->njobs_log2) != njobs)
            return hipErrorInvalidValue;
        q = *pq;
    }
    const bool wide = f.dem_wide != 0;
    // measurement aid: MOONRT_DEV_LDS=<bytes> of dynamic LDS per workgroup caps the waves a CU holds (13312 -> 12 = 3 per SIMD)
    static const unsigned dev_lds = getenv("MOONRT_DEV_LDS") ? (unsigned)atoi(getenv("MOONRT_DEV_LDS")) : 0u;
#define MRTX_LAUNCH(SV, ST, WD, MD, OV) hipLaunchKernelGGL((mrtx::render_kernel<SV, ST, WD, MD, OV>), grid, block, dev_lds, st, fr, q)
#define MRTX_CASE3(SV, MD, OV)                                                                                     \
        if (wide) { if (stats) MRTX_LAUNCH(SV, true, true, MD, OV); else MRTX_LAUNCH(SV, false, true, MD, OV); }   \
        else { if (stats) MRTX_LAUNCH(SV, true, false, MD, OV); else MRTX_LAUNCH(SV, false, false, MD, OV); }
#define MRTX_CASE2(SV, MD) if (overlay) { MRTX_CASE3(SV, MD, true) } else { MRTX_CASE3(SV, MD, false) }
#define MRTX_CASE(SV)                                                  \
    case SV:                                                           \
        if (mode == 1) { MRTX_CASE2(SV, 1) } else if (mode == 2) { MRTX_CASE2(SV, 2) }                              \
        else if (mode == 3) { if (stats) MRTX_LAUNCH(SV, true, false, 3, false); else MRTX_LAUNCH(SV, false, false, 3, false); }   \
        else { MRTX_CASE2(SV, 0) }   \
        break;
    switch (S) {
#ifndef MRTX_DEV_ONLY_S64      // tools/build_variant.sh ... -DMRTX_DEV_ONLY_S64: A/B builds for bench.py compile in a fifth of the time
        MRTX_CASE(1) MRTX_CASE(2) MRTX_CASE(4) MRTX_CASE(8) MRTX_CASE(16) MRTX_CASE(32)
#endif
        MRTX_CASE(64)
        default: return hipErrorInvalidValue;
    }
#undef MRTX_CASE
#undef MRTX_CASE2
#undef MRTX_CASE3
#undef MRTX_LAUNCH
    return hipGetLastError();
}

// persistent waves the device holds at once for path_kernel (a multiple of 8: see the chunk deal in the kernel)
int mrtx_path_waves(bool stats, bool wide, int* out) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
#define MRTX_OCC(ST, WD) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mrtx::path_kernel<ST, WD>, 64, 0)
    if (stats) e = wide ? MRTX_OCC(true, true) : MRTX_OCC(true, false);
    else e = wide ? MRTX_OCC(false, true) : MRTX_OCC(false, false);
#undef MRTX_OCC
    if (e != hipSuccess) return (int)e;
    if (per_cu < 1) per_cu = 1;
    int n = per_cu * prop.multiProcessorCount;
    n = (n + 7) / 8 * 8;
    *out = n;
    return 0;
}
hipError_t mrtx_launch_paths(const FrameC& f, const PathQ& pq, int S, bool stats, int n_waves, hipStream_t st) {
    if (pq.n_chunks == 0) return hipSuccess;
    // every (label, sub) counter needs a consumer: blocks b with (b >> 3) % n_sub == sub exist only if n_waves >= 8 * n_sub;
    // a group's chunk counts live in the lanes of one wave: at most 64 chunks per group
    if (n_waves < 8 * pq.n_sub || (n_waves & 7) || !pq.counters || pq.n_sub < 1 || pq.n_sub > 16 || pq.grp_log2 < 0 ||
        pq.grp_log2 + pq.njobs_log2 > 6)
        return hipErrorInvalidValue;
    const bool wide = f.dem_wide != 0;
    const dim3 grid((unsigned)n_waves), block(64);
    if (stats) { if (wide) hipLaunchKernelGGL((mrtx::path_kernel<true, true>), grid, block, 0, st, f, pq);
                 else hipLaunchKernelGGL((mrtx::path_kernel<true, false>), grid, block, 0, st, f, pq); }
    else { if (wide) hipLaunchKernelGGL((mrtx::path_kernel<false, true>), grid, block, 0, st, f, pq);
           else hipLaunchKernelGGL((mrtx::path_kernel<false, false>), grid, block, 0, st, f, pq); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    unsigned rb = (pq.n_chunks + 4u * MRTX_RESOLVE_U - 1u) / (4u * MRTX_RESOLVE_U);   // 4 waves x U chunks per block and turn
    if (rb > 65536u) rb = 65536u;
    const dim3 rgrid(rb), rblock(256);
    switch (S) {
#define MRTX_RES(SV) case SV: hipLaunchKernelGGL((mrtx::resolve_paths_kernel<SV>), rgrid, rblock, 0, st, f, pq); break;
        MRTX_RES(1) MRTX_RES(2) MRTX_RES(4) MRTX_RES(8) MRTX_RES(16) MRTX_RES(32) MRTX_RES(64)
#undef MRTX_RES
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifdef MRTX_PROF
extern "C" __attribute__((visibility("default"))) int mrtx_prof_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mrtx::g_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mrtx::g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef MRTX_PATH_PROF
extern "C" __attribute__((visibility("default"))) int mrtx_pprof_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mrtx::g_pprof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mrtx::g_pprof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int mrtx_pprof_ends(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mrtx::g_pprof_end), (size_t)(n < 8192 ? n : 8192) * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int mrtx_pprof_times(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mrtx::g_pprof_t), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        z[8] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(mrtx::g_pprof_t), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

static inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
hipError_t mrtx_launch_resolve_linear(const float* accum, float* out, int64_t npix, uint32_t ns, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::resolve_linear_kernel, dim3(grid_for(npix)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(accum), reinterpret_cast<float4*>(out), npix, ns);
    return hipGetLastError();
}
hipError_t mrtx_launch_resolve_rgba8(const float* accum, uint32_t* out, int64_t npix, uint32_t ns, float expo,
                                     const float* T8, const uint32_t* overlay, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::resolve_rgba8_kernel, dim3(grid_for(npix)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(accum), out, npix, ns, expo, T8, overlay);
    return hipGetLastError();
}
hipError_t mrtx_launch_resolve_rgb16(const float* accum, uint16_t* out, int64_t npix, uint32_t ns, float expo,
                                     const float* T16, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::resolve_rgb16_kernel, dim3(grid_for(npix)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(accum), out, npix, ns, expo, T16);
    return hipGetLastError();
}
hipError_t mrtx_launch_pack(const float* accum, const float* hits, void* dst, int W, int H, int tw, int th,
                            int tiles_x, int n_tiles, int rank, int world, int slot0, int slots, const int32_t* list, int shift,
                            int with_hits, hipStream_t st) {
    if (slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrtx::pack_shard_kernel, dim3(grid_for((int64_t)slots * tw * th)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(accum), reinterpret_cast<const float4*>(hits),
                       reinterpret_cast<float4*>(dst), W, H, tw, th, tiles_x, n_tiles, rank, world, slot0, slots, list, shift, with_hits);
    return hipGetLastError();
}
hipError_t mrtx_launch_unpack(float* accum, float* hits, const void* src, int W, int H, int tw, int th,
                              int tiles_x, int n_tiles, int src_rank, int world, int slots, const int32_t* list, int shift,
                              int with_hits, hipStream_t st) {
    if (slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrtx::unpack_shard_kernel, dim3(grid_for((int64_t)slots * tw * th)), dim3(256), 0, st,
                       reinterpret_cast<float4*>(accum), reinterpret_cast<float4*>(hits),
                       reinterpret_cast<const float4*>(src), W, H, tw, th, tiles_x, n_tiles, src_rank, world, slots, list, shift, with_hits);
    return hipGetLastError();
}
hipError_t mrtx_launch_unpack_all(float* accum, float* hits, const void* const* srcs, int W, int H, int tw, int th, int tiles_x,
                                  int n_tiles, int self, int world, int slots, const int32_t* list_all, int shift, int with_hits,
                                  hipStream_t st) {
    if (slots <= 0 || world < 2) return hipSuccess;
    if (world > 16) return hipErrorInvalidValue;
    mrtx::UnpackSrcs tab;
    for (int r = 0; r < 16; r++) tab.p[r] = r < world ? reinterpret_cast<const float4*>(srcs[r]) : nullptr;
    hipLaunchKernelGGL(mrtx::unpack_all_kernel, dim3(grid_for((int64_t)world * slots * tw * th)), dim3(256), 0, st,
                       reinterpret_cast<float4*>(accum), reinterpret_cast<float4*>(hits), tab, W, H, tw, th, tiles_x, n_tiles, self, world,
                       slots, list_all, shift, with_hits);
    return hipGetLastError();
}
hipError_t mrtx_launch_zero_tiles(float* accum, float* hits, const int32_t* tiles, int n, int W, int H, int tw, int th,
                                  int tiles_x, int shift, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrtx::zero_tiles_kernel, dim3(grid_for((int64_t)n * tw * th)), dim3(256), 0, st,
                       reinterpret_cast<float4*>(accum), reinterpret_cast<float4*>(hits), tiles, n, W, H, tw, th, tiles_x, shift);
    return hipGetLastError();
}
hipError_t mrtx_launch_ldem(const int16_t* src, float* dst, int h, int w, int d, unsigned int* max_bits,
                            hipStream_t st) {
    const int64_t n = (int64_t)h * w;
    hipLaunchKernelGGL(mrtx::ldem_block_mean_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, dst, h, w, d, max_bits);
    hipLaunchKernelGGL(mrtx::scale_by_inv_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, n, max_bits);
    return hipGetLastError();
}
hipError_t mrtx_launch_synth_ldem(int16_t* dst, int h, int w, uint32_t seed, hipStream_t st) {
    int octaves = 1, levels = 1;
    while ((8 << octaves) < w && octaves < 13) octaves++;
    while ((64 << levels) < w && levels < 10) levels++;
    hipLaunchKernelGGL(mrtx::synth_ldem_kernel, dim3(grid_for((int64_t)h * w)), dim3(256), 0, st, dst, h, w,
                       octaves, levels, seed);
    return hipGetLastError();
}
hipError_t mrtx_launch_synth_color(uint32_t* dst, int h, int w, uint32_t seed, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::synth_color_kernel, dim3(grid_for((int64_t)h * w)), dim3(256), 0, st, dst, h, w, seed);
    return hipGetLastError();
}
hipError_t mrtx_launch_probe_latlon(const float* a, const float* b, const float* c, float* lat, float* lon, int n,
                                    hipStream_t st) {
    hipLaunchKernelGGL(mrtx::probe_latlon_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, c, lat, lon, n);
    return hipGetLastError();
}
hipError_t mrtx_launch_probe_cr(uint32_t lo, uint64_t n, int which, unsigned long long* out2, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::probe_cr_kernel, dim3(8192), dim3(256), 0, st, lo, n, which, out2);
    return hipGetLastError();
}
hipError_t mrtx_launch_probe_stream(const void* src, int64_t n_pairs, float* out, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::probe_stream_kernel, dim3(8192), dim3(256), 0, st, reinterpret_cast<const mrtx::Pair*>(src), n_pairs, out);
    return hipGetLastError();
}
hipError_t mrtx_launch_mip(const float* dem_padded, int h, int w, float* mip, int mh, int mw, int shift, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::mip_build_kernel, dim3(grid_for((int64_t)(mh + 2) * (mw + 2))), dim3(64), 0, st, dem_padded,
                       h, w, mip, mh, mw, shift);
    return hipGetLastError();
}
hipError_t mrtx_launch_hmip(const float* mip, int mh, int mw, int shift, float* out, int hh, int hw, int hshift, int h, int w,
                            hipStream_t st) {
    hipLaunchKernelGGL(mrtx::hmip_build_kernel, dim3(grid_for((int64_t)hh * hw)), dim3(64), 0, st, mip, mh, mw, shift, out, hh, hw,
                       hshift, h, w);
    return hipGetLastError();
}
hipError_t mrtx_launch_mip_pairs(const float* mip, float* out_pairs, int rows, int pitch, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::mip_pair_kernel, dim3(grid_for((int64_t)rows * pitch)), dim3(256), 0, st, mip,
                       reinterpret_cast<float2*>(out_pairs), rows, pitch);
    return hipGetLastError();
}
hipError_t mrtx_launch_color_pairs(const uint32_t* src, void* dst, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::color_pair_kernel, dim3(grid_for((int64_t)(h + 1) * (w + 4))), dim3(256), 0, st, src,
                       reinterpret_cast<uint2*>(dst), h, w);
    return hipGetLastError();
}
hipError_t mrtx_launch_pad_dem(const float* src, float* dst, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(mrtx::pad_dem_kernel, dim3(grid_for((int64_t)(h + 4) * (w + 4))), dim3(256), 0, st, src, dst, h, w);
    return hipGetLastError();
}
}
#endif   // MRTX_DEV_ONE
