/*
 * restir_oracle.c — CPU restatement (plain C) of the reference's ReSTIR DI hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (cedec_2024_rt_amd/, include/,
 * app/) may include, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the reported
 * CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout). Arithmetic is IEEE binary32 in the reference's operation
 * order; compile with -ffp-contract=off and WITHOUT -ffast-math.
 *
 * Pinning (DESIGN.md "Oracle"):
 *   - integer functions against the KATs of SURVEY.md §8(c) (tests/test_oracle_kat.py);
 *   - in MATH_LIBM mode, bit-for-bit against oracle/_ref (the reference's own sources
 *     compiled in place, see oracle/Makefile and oracle/ref_driver.cpp) and against
 *     the committed fixtures under tests/golden/ generated from it;
 *   - ray/scene intersection: the reference delegates to HIPRT (closed binary, device
 *     code absent from the checkout) => PARITY UNPINNED for that step; it is pinned
 *     BY DEFINITION to the brute-force closest hit of the reference's own
 *     intersect_ray_triangle (common/core.hpp:91-136) with the 04_ao tie rule
 *     (examples/04_ao/04_ao.cu:8-29: `t <= tmax` => on equal t the later index wins).
 *
 * Math modes:
 *   MATH_LIBM     log/cos/sin/exp/pow = glibc (what a host build of the reference does)
 *   MATH_PORTABLE the deterministic functions of portable_math.h, which the HIP
 *                 kernels use as well => CPU and GPU agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../cedec_2024_rt_amd/csrc/portable_math.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

enum { MATH_LIBM = 0, MATH_PORTABLE = 1 };
static int g_math_mode = MATH_PORTABLE;

ORACLE_API void o_set_math_mode(int mode) { g_math_mode = mode; }
ORACLE_API int o_get_math_mode(void) { return g_math_mode; }
ORACLE_API void o_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
ORACLE_API int o_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float m_log(float x) { return g_math_mode == MATH_LIBM ? logf(x) : pm_logf(x); }
static inline float m_cos(float x) { return g_math_mode == MATH_LIBM ? cosf(x) : pm_cosf(x); }
static inline float m_sin(float x) { return g_math_mode == MATH_LIBM ? sinf(x) : pm_sinf(x); }
static inline float m_exp(float x) { return g_math_mode == MATH_LIBM ? expf(x) : pm_expf(x); }
static inline float m_pow8(float x)
{
    return g_math_mode == MATH_LIBM ? powf(x, 8.0f) : pm_pow8f(x);
}
static inline float m_pow(float x, float y)
{
    return g_math_mode == MATH_LIBM ? powf(x, y) : pm_powf_pos(x, y);
}

/* ------------------------------------------------------------------ types */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

/* common/core.hpp:38-43, 60 bytes */
typedef struct { v3 v[3]; v3 color; v3 emissive; } OTriangle;
/* common/core.hpp:167-172, 16 bytes */
typedef struct { v2 uv; int index; int pad; } OVisibility;
/* common/reservoir.hpp:5-13, 64 bytes */
typedef struct
{
    v3 origin_position, origin_normal, hit_position, hit_normal, radiance;
    uint8_t visibility;
    uint8_t pad[3];
} OSample;
/* common/reservoir.hpp:15-38, 76 bytes */
typedef struct { OSample sample; float w_sum; float ucw; int M; } OReservoir;
/* common/options.hpp:4-22, 48 bytes */
typedef struct
{
    uint8_t accumulate;
    int max_depth;
    v3 sky_color;
    int ris_sample_count;
    float rejection_heuristics_threshold;
    uint8_t use_temporal_resampling;
    uint8_t use_spatial_resampling;
    int spatial_resampling_sample_count;
    float spatial_resampling_radius;
    int spatial_resampling_passes;
    uint8_t use_shadowed_target_function;
    uint8_t use_visibility_reuse;
} OOptions;
/* common/camera.hpp:5-9, 36 bytes */
typedef struct { v3 origin, right, up; } ORayGen;
typedef struct { float x, y, z, w; } v4;

ORACLE_API int o_sizeof(int what)
{
    switch (what)
    {
        case 0: return (int)sizeof(OTriangle);
        case 1: return (int)sizeof(OVisibility);
        case 2: return (int)sizeof(OSample);
        case 3: return (int)sizeof(OReservoir);
        case 4: return (int)sizeof(OOptions);
        case 5: return (int)sizeof(ORayGen);
        default: return -1;
    }
}

/* ------------------------------------------------------------- vector ops */
/* HIP vector operators are component-wise (hip_vector_types.h); the helpers in
 * common/math.hpp:109-130 are restated with the same association. */

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* common/math.hpp:109-113 */
static inline v3 cross(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* common/math.hpp:114-117 */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* common/math.hpp:118-119 */
static inline float length3(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { return divs(a, length3(a)); }
/* common/math.hpp:120-123 */
static inline v3 mix(v3 a, v3 b, float t) { return add(a, muls(sub(b, a), t)); }
/* common/math.hpp:125-130 */
static inline float luminance(v3 a)
{
    return dot(a, V3(0.1762044f, 0.8129847f, 0.0108109f));
}
#define O_PI 3.14159265358979323846f
#define O_FLT_MAX 3.402823466e+38f
/* device max(float,float) is fmaxf-like: a NaN operand yields the other one */
static inline float fmax_dev(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }

/* -------------------------------------------------------------------- rng */
/* common/rng.hpp:8-40 */
typedef struct { uint64_t state, inc; } PCG;
static inline uint32_t pcg_uniform(PCG* r)
{
    const uint64_t old = r->state;
    r->state = old * 6364136223846793005ULL + r->inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
static inline PCG pcg_init(uint64_t seed, uint64_t sequence)
{
    PCG r;
    r.state = 0u;
    r.inc = (sequence << 1u) | 1u;
    pcg_uniform(&r);
    r.state += seed;
    pcg_uniform(&r);
    return r;
}
static inline float pcg_uniformf(PCG* r)
{
    const uint32_t bits = (pcg_uniform(r) >> 9) | 0x3f800000u;
    float v;
    memcpy(&v, &bits, 4);
    return v - 1.0f;
}
/* common/rng.hpp:43-58 */
static inline uint32_t hashPCG(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
static inline uint32_t hashPCG3(uint32_t x, uint32_t y, uint32_t z)
{
    return hashPCG(hashPCG(hashPCG(x) + y) + z);
}
static inline uint32_t hashPCG4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    return hashPCG(hashPCG(hashPCG(hashPCG(x) + y) + z) + w);
}

ORACLE_API uint32_t o_hashPCG(uint32_t v) { return hashPCG(v); }
ORACLE_API uint32_t o_hashPCG3(uint32_t x, uint32_t y, uint32_t z) { return hashPCG3(x, y, z); }
ORACLE_API uint32_t o_hashPCG4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    return hashPCG4(x, y, z, w);
}
/* out_u[n_u] raw draws, then out_f[n_f] float draws; final state/inc returned */
ORACLE_API void o_pcg_sequence(uint64_t seed, uint64_t sequence, int n_u, uint32_t* out_u,
                               int n_f, float* out_f, uint64_t* state_inc)
{
    PCG r = pcg_init(seed, sequence);
    for (int i = 0; i < n_u; ++i) out_u[i] = pcg_uniform(&r);
    for (int i = 0; i < n_f; ++i) out_f[i] = pcg_uniformf(&r);
    state_inc[0] = r.state;
    state_inc[1] = r.inc;
}

/* ----------------------------------------------------------------- camera */
/* common/camera.hpp:11-25 (host side; tan() of a float argument is tanf) */
ORACLE_API void o_raygen_lookat(ORayGen* rg, const float* eye, const float* center,
                                const float* up, float fovy, int width, int height)
{
    const v3 e = V3(eye[0], eye[1], eye[2]);
    const v3 c = V3(center[0], center[1], center[2]);
    const v3 upv = V3(up[0], up[1], up[2]);
    const v3 f = normalize(sub(c, e));
    const v3 s = normalize(cross(f, upv));
    const v3 u = cross(s, f);
    const float tanThetaY = tanf(fovy * 0.5f);
    const float tanThetaX = tanThetaY / (float)height * (float)width;
    rg->origin = e;
    rg->right = muls(s, tanThetaX);
    rg->up = muls(u, tanThetaY);
}
/* common/camera.hpp:27-35 */
static inline void raygen_shoot(const ORayGen* rg, v3* ro, v3* rd, float u, float v)
{
    const v3 from = rg->origin;
    const v3 forward = normalize(cross(rg->up, rg->right));
    const v3 to = add(add(add(rg->origin, forward), mix(neg(rg->right), rg->right, u)),
                      mix(rg->up, neg(rg->up), v));
    *ro = from;
    *rd = normalize(sub(to, from));
}
ORACLE_API void o_raygen_shoot(const ORayGen* rg, float u, float v, float* ro, float* rd)
{
    v3 o, d;
    raygen_shoot(rg, &o, &d, u, v);
    ro[0] = o.x; ro[1] = o.y; ro[2] = o.z;
    rd[0] = d.x; rd[1] = d.y; rd[2] = d.z;
}

/* The examples' interactive camera: CameraControl::cursorPosCallback, common/misc.hpp:129-205, one drag
 * event of (dx, dy) pixels with ONE button held (0 = left: orbit :147-181, 1 = right: dolly :183-190,
 * 2 = middle: pan :192-205). Host code in the reference (glibc sinf/cosf whatever the math mode).
 * eye / lookat are updated in place; returns m_updated (always 1 for a known button, as in the reference). */
ORACLE_API int o_camera_control(float* eye3, float* lookat3, int button, float dx, float dy)
{
    v3 orig = V3(eye3[0], eye3[1], eye3[2]), lookat = V3(lookat3[0], lookat3[1], lookat3[2]);
    v3 cameraLocal = sub(orig, lookat);
    const float r = length3(cameraLocal);
    int updated = 0;
    if (button == 0)
    {
        const float sensitivity = 0.004f;
        {
            const float sinTheta = sinf(dx * sensitivity);
            const float cosTheta = cosf(dx * sensitivity);
            const float new_x = cosTheta * cameraLocal.x - sinTheta * cameraLocal.z;
            const float new_z = sinTheta * cameraLocal.x + cosTheta * cameraLocal.z;
            cameraLocal.x = new_x;
            cameraLocal.z = new_z;
        }
        {
            const float xz = sqrtf(cameraLocal.x * cameraLocal.x + cameraLocal.z * cameraLocal.z);
            const float sinTheta = sinf(dy * sensitivity);
            const float cosTheta = cosf(dy * sensitivity);
            const float new_xz = cosTheta * xz - sinTheta * cameraLocal.y;
            const float new_y = sinTheta * xz + cosTheta * cameraLocal.y;
            if (-r + r * 0.01f < new_y && new_y < r - r * 0.01f)
            {
                cameraLocal.x = cameraLocal.x * (new_xz / xz);
                cameraLocal.z = cameraLocal.z * (new_xz / xz);
                cameraLocal.y = new_y;
            }
        }
        orig = add(lookat, cameraLocal);
        updated = 1;
    }
    if (button == 1)
    {
        const float sensitivity = 0.002f;
        const float new_r = fmaxf(r - r * sensitivity * dy, 0.01f);
        const float s = new_r / r;
        orig = add(lookat, muls(cameraLocal, s));
        updated = 1;
    }
    if (button == 2)
    {
        const float sensitivity = 0.001f;
        const v3 forward = normalize(sub(lookat, orig));
        const v3 right = normalize(cross(forward, V3(0.0f, 1.0f, 0.0f)));
        const v3 up = cross(right, forward);
        const float amount = fmaxf(r * sensitivity, 0.01f);
        const v3 delta = add(muls(muls(neg(right), dx), amount), muls(muls(up, dy), amount));
        orig = add(orig, delta);
        lookat = add(lookat, delta);
        updated = 1;
    }
    eye3[0] = orig.x; eye3[1] = orig.y; eye3[2] = orig.z;
    lookat3[0] = lookat.x; lookat3[1] = lookat.y; lookat3[2] = lookat.z;
    return updated;
}

/* ---------------------------------------------------------- triangle math */
/* common/core.hpp:45-68 */
static inline v3 a_tangent_of(const OTriangle* t) { return normalize(sub(t->v[1], t->v[0])); }
static inline v3 normal_of(const OTriangle* t)
{
    const v3 e0 = sub(t->v[1], t->v[0]);
    const v3 e1 = sub(t->v[2], t->v[0]);
    return normalize(cross(e0, e1));
}
static inline float area_of(const OTriangle* t)
{
    const v3 e0 = sub(t->v[1], t->v[0]);
    const v3 e1 = sub(t->v[2], t->v[0]);
    return 0.5f * length3(cross(e0, e1));
}
static inline int has_emission(const OTriangle* t)
{
    return t->emissive.x > 0.0f || t->emissive.y > 0.0f || t->emissive.z > 0.0f;
}

/* common/core.hpp:76-89 */
static inline v3 sample_hemisphere(float r0, float r1, float r2)
{
    const float theta = r0 * 2.0f * O_PI;
    float radius = r1 + r2;
    if (1.0f < radius) { radius = 2.0f - radius; }
    const float x = m_cos(theta) * radius;
    const float z = m_sin(theta) * radius;
    const float a = 1.0f - radius * radius;
    const float y = sqrtf((a < 0.0f) ? 0.0f : a);
    return V3(x, y, z);
}

/* common/core.hpp:91-136 */
static inline int intersect_ray_triangle(float* tOut, float* uOut, float* vOut, v3 ro, v3 rd,
                                         float tmin, float tmax, v3 v0, v3 v1, v3 v2)
{
    const v3 e0 = sub(v1, v0);
    const v3 e1 = sub(v2, v1);
    const v3 e2 = sub(v0, v2);
    const v3 n = cross(e0, e1);
    const float t = dot(sub(v0, ro), n) / dot(n, rd);
    if (tmin <= t && t <= tmax)
    {
        const v3 p = add(ro, muls(rd, t));
        const float a0 = dot(n, cross(e0, sub(p, v0)));
        const float a1 = dot(n, cross(e1, sub(p, v1)));
        const float a2 = dot(n, cross(e2, sub(p, v2)));
        if (a0 < 0.0f || a1 < 0.0f || a2 < 0.0f) { return 0; }
        const float a = a0 + a1 + a2;
        const float bW = a0 / a;
        const float bU = a1 / a;
        const float bV = a2 / a;
        (void)bU;
        *tOut = t;
        *uOut = bV;
        *vOut = bW;
        return 1;
    }
    return 0;
}

/* common/core.hpp:237-252 */
static inline v2 warp_unit_triangle(float x, float y)
{
    if (y > x) { x *= 0.5f; y -= x; }
    else { y *= 0.5f; x -= y; }
    v2 r = {x, y};
    return r;
}

/* common/core.hpp:287-295 */
static inline float geometry_term(v3 p0, v3 n0, v3 p1, v3 n1)
{
    v3 v = sub(p1, p0);
    const float sqr_dist = dot(v, v);
    v = normalize(v);
    return fabsf(dot(v, n0)) * fabsf(dot(neg(v), n1)) / sqr_dist;
}

/* ------------------------------------------------------------- the scene */

typedef struct { float lo[3], hi[3]; int left, right; int first, count; } ONode;

typedef struct
{
    OTriangle* tris;
    int n_tris;
    uint32_t* lights;
    int n_lights;
    /* CPU BVH (own construction; results must equal brute force) */
    ONode* nodes;
    int n_nodes;
    int* prim; /* permutation */
    int use_bvh;
} OScene;

typedef struct { float t; v2 uv; int index; } OHit; /* common/core.hpp:138-143 */

/* The pinned definition of raytrace() (common/raytrace.hpp:18-43): closest hit in
 * [tmin,tmax] over all triangles with core.hpp:91-136, ties -> highest index. */
static int closest_hit_brute(const OScene* s, v3 ro, v3 rd, float tmin, float tmax, OHit* hit)
{
    float best = tmax;
    int index = -1;
    float bu = 0.0f, bv = 0.0f;
    for (int i = 0; i < s->n_tris; ++i)
    {
        const OTriangle* tri = &s->tris[i];
        float t, u, v;
        if (intersect_ray_triangle(&t, &u, &v, ro, rd, tmin, best, tri->v[0], tri->v[1], tri->v[2]))
        {
            best = t; bu = u; bv = v; index = i;
        }
    }
    if (index < 0) return 0;
    hit->t = best; hit->uv.x = bu; hit->uv.y = bv; hit->index = index;
    return 1;
}

/* ---- CPU BVH: binned-SAH top-down build, conservative slab test ---- */

typedef struct { float lo[3], hi[3], c[3]; } PrimBox;

static void box_init(float* lo, float* hi)
{
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
}
static void box_grow(float* lo, float* hi, const float* l2, const float* h2)
{
    for (int a = 0; a < 3; ++a)
    {
        if (l2[a] < lo[a]) lo[a] = l2[a];
        if (h2[a] > hi[a]) hi[a] = h2[a];
    }
}
static float box_area(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

#define BVH_LEAF_MAX 4
#define BVH_BINS 16

static int bvh_build_rec(OScene* s, const PrimBox* pb, int first, int count)
{
    const int me = s->n_nodes++;
    ONode* node = &s->nodes[me];
    box_init(node->lo, node->hi);
    float clo[3], chi[3];
    box_init(clo, chi);
    for (int i = first; i < first + count; ++i)
    {
        const PrimBox* b = &pb[s->prim[i]];
        box_grow(node->lo, node->hi, b->lo, b->hi);
        box_grow(clo, chi, b->c, b->c);
    }
    node->left = node->right = -1;
    node->first = first;
    node->count = count;
    if (count <= BVH_LEAF_MAX) return me;

    int best_axis = -1, best_split = -1;
    float best_cost = INFINITY;
    for (int a = 0; a < 3; ++a)
    {
        const float ext = chi[a] - clo[a];
        if (!(ext > 0.0f)) continue;
        float blo[BVH_BINS][3], bhi[BVH_BINS][3];
        int bcnt[BVH_BINS];
        for (int b = 0; b < BVH_BINS; ++b) { box_init(blo[b], bhi[b]); bcnt[b] = 0; }
        const float scale = (float)BVH_BINS / ext;
        for (int i = first; i < first + count; ++i)
        {
            const PrimBox* p = &pb[s->prim[i]];
            int b = (int)((p->c[a] - clo[a]) * scale);
            if (b < 0) b = 0;
            if (b >= BVH_BINS) b = BVH_BINS - 1;
            box_grow(blo[b], bhi[b], p->lo, p->hi);
            bcnt[b]++;
        }
        float rarea[BVH_BINS];
        int rcnt[BVH_BINS];
        float lo[3], hi[3];
        box_init(lo, hi);
        int c = 0;
        for (int b = BVH_BINS - 1; b > 0; --b)
        {
            box_grow(lo, hi, blo[b], bhi[b]);
            c += bcnt[b];
            rarea[b] = box_area(lo, hi);
            rcnt[b] = c;
        }
        box_init(lo, hi);
        c = 0;
        for (int b = 0; b < BVH_BINS - 1; ++b)
        {
            box_grow(lo, hi, blo[b], bhi[b]);
            c += bcnt[b];
            if (c == 0 || rcnt[b + 1] == 0) continue;
            const float cost = box_area(lo, hi) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
        }
    }
    int mid;
    if (best_axis < 0)
    {
        mid = first + count / 2; /* all centroids coincide: split by order */
    }
    else
    {
        const float ext = chi[best_axis] - clo[best_axis];
        const float scale = (float)BVH_BINS / ext;
        int i = first, j = first + count - 1;
        while (i <= j)
        {
            const PrimBox* p = &pb[s->prim[i]];
            int b = (int)((p->c[best_axis] - clo[best_axis]) * scale);
            if (b < 0) b = 0;
            if (b >= BVH_BINS) b = BVH_BINS - 1;
            if (b <= best_split) { ++i; }
            else
            {
                const int tmp = s->prim[i]; s->prim[i] = s->prim[j]; s->prim[j] = tmp;
                --j;
            }
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    const int l = bvh_build_rec(s, pb, first, mid - first);
    const int r = bvh_build_rec(s, pb, mid, first + count - mid);
    s->nodes[me].left = l;
    s->nodes[me].right = r;
    return me;
}

static void bvh_build(OScene* s)
{
    const int n = s->n_tris;
    PrimBox* pb = (PrimBox*)malloc(sizeof(PrimBox) * (size_t)(n > 0 ? n : 1));
    float slo[3], shi[3];
    box_init(slo, shi);
    for (int i = 0; i < n; ++i)
    {
        const OTriangle* t = &s->tris[i];
        box_init(pb[i].lo, pb[i].hi);
        for (int k = 0; k < 3; ++k)
        {
            const float p[3] = {t->v[k].x, t->v[k].y, t->v[k].z};
            box_grow(pb[i].lo, pb[i].hi, p, p);
        }
        box_grow(slo, shi, pb[i].lo, pb[i].hi);
    }
    /* conservative padding: the accepted hit point of core.hpp:91-136 lies within
     * rounding distance of the triangle, not exactly on it */
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a)
    {
        const float m = fmaxf(fabsf(slo[a]), fabsf(shi[a]));
        if (m > ext) ext = m;
    }
    const float pad = 4e-5f * (ext > 1.0f ? ext : 1.0f);
    for (int i = 0; i < n; ++i)
    {
        for (int a = 0; a < 3; ++a)
        {
            pb[i].c[a] = 0.5f * (pb[i].lo[a] + pb[i].hi[a]);
            pb[i].lo[a] -= pad;
            pb[i].hi[a] += pad;
        }
    }
    s->prim = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) s->prim[i] = i;
    s->nodes = (ONode*)malloc(sizeof(ONode) * (size_t)(2 * n + 1));
    s->n_nodes = 0;
    if (n > 0) bvh_build_rec(s, pb, 0, n);
    free(pb);
}

/* conservative slab test: returns 1 when the ray may touch the box within [t0,t1] */
static inline int slab(const ONode* nd, v3 ro, const float* inv, float t0, float t1, float* tnear)
{
    const float o[3] = {ro.x, ro.y, ro.z};
    float tn = t0, tf = t1;
    for (int a = 0; a < 3; ++a)
    {
        float ta = (nd->lo[a] - o[a]) * inv[a];
        float tb = (nd->hi[a] - o[a]) * inv[a];
        if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
        /* NaN (0*inf) compares false => the bound is left unchanged (conservative) */
        /* relative slack as a product so that +-inf (ray parallel to the slab) stays inf */
        ta = ta * (1.0f - 4e-7f);
        tb = tb * (1.0f + 4e-7f);
        if (ta > tn) tn = ta;
        if (tb < tf) tf = tb;
    }
    *tnear = tn;
    return tn <= tf;
}

static int closest_hit_bvh(const OScene* s, v3 ro, v3 rd, float tmin, float tmax, OHit* hit,
                           int any_hit)
{
    if (s->n_tris == 0) return 0;
    const float inv[3] = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    float best = tmax;
    int index = -1;
    float bu = 0.0f, bv = 0.0f;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0)
    {
        const ONode* nd = &s->nodes[stack[--sp]];
        float tn;
        if (!slab(nd, ro, inv, tmin, best, &tn)) continue;
        if (nd->left < 0)
        {
            for (int i = nd->first; i < nd->first + nd->count; ++i)
            {
                const int pi = s->prim[i];
                const OTriangle* tri = &s->tris[pi];
                float t, u, v;
                /* range test against the ORIGINAL interval; tie rule applied explicitly */
                if (intersect_ray_triangle(&t, &u, &v, ro, rd, tmin, tmax, tri->v[0], tri->v[1],
                                           tri->v[2]))
                {
                    if (index < 0 || t < best || (t == best && pi > index))
                    {
                        best = t; bu = u; bv = v; index = pi;
                        if (any_hit) { hit->t = t; hit->uv.x = u; hit->uv.y = v; hit->index = pi; return 1; }
                    }
                }
            }
        }
        else
        {
            float tl, tr;
            const int hl = slab(&s->nodes[nd->left], ro, inv, tmin, best, &tl);
            const int hr = slab(&s->nodes[nd->right], ro, inv, tmin, best, &tr);
            if (hl && hr)
            {
                if (tl <= tr) { stack[sp++] = nd->right; stack[sp++] = nd->left; }
                else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
            }
            else if (hl) { stack[sp++] = nd->left; }
            else if (hr) { stack[sp++] = nd->right; }
        }
    }
    if (index < 0) return 0;
    hit->t = best; hit->uv.x = bu; hit->uv.y = bv; hit->index = index;
    return 1;
}

static inline int raytrace(const OScene* s, v3 ro, v3 rd, float tmin, float tmax, OHit* hit)
{
    if (s->use_bvh) return closest_hit_bvh(s, ro, rd, tmin, tmax, hit, 0);
    return closest_hit_brute(s, ro, rd, tmin, tmax, hit);
}

/* common/core.hpp:32-36 + common/raytrace.hpp:45-52. Only the boolean is consumed,
 * so an any-hit early out is observationally identical to the closest-hit query. */
static inline float check_visibility(const OScene* s, v3 p0, v3 n0, v3 p1)
{
    const v3 org = add(p0, muls(n0, 0.001f));
    const v3 dir = sub(p1, p0);
    OHit h;
    int hit;
    if (s->use_bvh) hit = closest_hit_bvh(s, org, dir, 0.0f, 0.99f, &h, 1);
    else hit = closest_hit_brute(s, org, dir, 0.0f, 0.99f, &h);
    return hit ? 0.0f : 1.0f;
}

ORACLE_API OScene* o_scene_create(const OTriangle* tris, int n, int use_bvh)
{
    OScene* s = (OScene*)calloc(1, sizeof(OScene));
    s->tris = (OTriangle*)malloc(sizeof(OTriangle) * (size_t)(n > 0 ? n : 1));
    memcpy(s->tris, tris, sizeof(OTriangle) * (size_t)n);
    s->n_tris = n;
    /* examples/10_restir_di/10_restir_di.cpp:196-205 */
    s->lights = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    s->n_lights = 0;
    for (int i = 0; i < n; ++i)
        if (has_emission(&s->tris[i])) s->lights[s->n_lights++] = (uint32_t)i;
    s->use_bvh = use_bvh;
    if (use_bvh) bvh_build(s);
    return s;
}
ORACLE_API void o_scene_destroy(OScene* s)
{
    if (!s) return;
    free(s->tris); free(s->lights); free(s->nodes); free(s->prim); free(s);
}
ORACLE_API int o_scene_num_lights(const OScene* s) { return s->n_lights; }
ORACLE_API void o_scene_lights(const OScene* s, uint32_t* out)
{
    memcpy(out, s->lights, sizeof(uint32_t) * (size_t)s->n_lights);
}
ORACLE_API void o_scene_set_bvh(OScene* s, int use_bvh)
{
    if (use_bvh && !s->nodes) bvh_build(s);
    s->use_bvh = use_bvh;
}

/* rays: n x {ox,oy,oz,dx,dy,dz,tmin,tmax}; hits: n x {t,u,v,index(as int bits)} */
ORACLE_API void o_trace_closest(const OScene* s, const float* rays, int n, float* hits, int force_brute)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i)
    {
        const float* r = rays + 8 * (size_t)i;
        OHit h = {0.0f, {0.0f, 0.0f}, -1};
        const v3 ro = V3(r[0], r[1], r[2]), rd = V3(r[3], r[4], r[5]);
        if (force_brute) closest_hit_brute(s, ro, rd, r[6], r[7], &h);
        else raytrace(s, ro, rd, r[6], r[7], &h);
        float* o = hits + 4 * (size_t)i;
        o[0] = h.t; o[1] = h.uv.x; o[2] = h.uv.y;
        memcpy(&o[3], &h.index, 4);
    }
}

/* -------------------------------------------------------- surface / lights */

typedef struct { v3 p, n; } Surf;
/* common/core.hpp:189-207 */
static inline Surf make_surface_info_eye(const OVisibility* vis, const OTriangle* tris, v3 eye)
{
    const OTriangle* t = &tris[vis->index];
    Surf r;
    r.p = add(add(muls(t->v[0], 1.0f - vis->uv.x - vis->uv.y), muls(t->v[1], vis->uv.x)),
              muls(t->v[2], vis->uv.y));
    r.n = normal_of(t);
    const v3 view = normalize(sub(eye, r.p));
    if (dot(view, r.n) < 0.0f) { r.n = neg(r.n); }
    return r;
}

/* common/core.hpp:152-165: surface info from a ray hit (07_pt / 08_nee / 09_ris) */
static inline Surf make_surface_info_ray(const OTriangle* tri, v3 ro, v3 rd, float t)
{
    Surf r;
    r.p = add(ro, muls(rd, t));
    r.n = normal_of(tri);
    if (dot(neg(rd), r.n) < 0.0f) { r.n = neg(r.n); }
    return r;
}
/* common/core.hpp:216-235: make_tangent_basis(n, index, triangles) then local_to_world(v, basis) */
static inline v3 tangent_to_world(const OTriangle* tri, v3 n, v3 v)
{
    const v3 t = a_tangent_of(tri);
    const v3 b = normalize(cross(t, n));
    return add(add(muls(t, v.x), muls(n, v.y)), muls(b, v.z));
}

typedef struct { v3 p, n; int index; } LightSample;
/* common/core.hpp:261-285 */
static inline LightSample sample_light(const OScene* s, float rv0, float rv1, float rv2)
{
    const float fl = rv0 * (float)(size_t)s->n_lights;
    uint32_t nth = (uint32_t)fl;
    if (nth == (uint32_t)s->n_lights) { nth = (uint32_t)s->n_lights - 1u; }
    LightSample r;
    r.index = (int)s->lights[nth];
    const OTriangle* lt = &s->tris[r.index];
    const v2 b = warp_unit_triangle(rv1, rv2);
    r.p = add(add(muls(lt->v[0], 1.0f - b.x - b.y), muls(lt->v[1], b.x)), muls(lt->v[2], b.y));
    r.n = normal_of(lt);
    return r;
}

/* common/reservoir.hpp:42-59 */
static inline float evaluate_target_function(const OScene* s, v3 op, v3 on, v3 hp, v3 hn, v3 rad,
                                             int is_shadowed, long* rays)
{
    const float brdf = 1.0f / O_PI;
    const float G = geometry_term(op, on, hp, hn);
    if (is_shadowed)
    {
        const float V = check_visibility(s, op, on, hp);
        ++*rays;
        return brdf * G * V * luminance(rad);
    }
    return brdf * G * luminance(rad);
}
/* common/reservoir.hpp:61-87 */
static inline float normal_rejection_heuristics(v3 n0, v3 n1)
{
    return m_pow8(fmax_dev(dot(n0, n1), 0.0f));
}
static inline float depth_rejection_heuristics(v3 p0, v3 p1, v3 eye)
{
    const float d0 = length3(sub(p0, eye));
    const float d1 = length3(sub(p1, eye));
    const float diff = (d1 - d0) * (d1 - d0) / d0;
    return m_exp(-32.0f * diff);
}
static inline float rejection_heuristics(const OReservoir* r0, const OReservoir* r1, v3 eye)
{
    float w = 1.0f;
    w *= depth_rejection_heuristics(r0->sample.origin_position, r1->sample.origin_position, eye);
    w *= normal_rejection_heuristics(r0->sample.origin_normal, r1->sample.origin_normal);
    return w;
}
/* common/reservoir.hpp:89-95 */
static inline v2 sample_2d_gaussian(float rv0, float rv1)
{
    const float radius = sqrtf(fmax_dev(-2.0f * m_log(rv0), 0.0f));
    const float phi = 2.0f * O_PI * rv1;
    v2 r = {radius * m_cos(phi), radius * m_sin(phi)};
    return r;
}
/* int <- float conversion as the AMD GPU does it (v_cvt_i32_f32): NaN -> 0,
 * saturating. C leaves the out-of-range cases undefined. */
static inline int f2i_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
/* `M *= w` with int M, float w (10_restir_di.cu:211-212, 362-363) */
static inline int scale_M(int M, float w) { return f2i_sat((float)M * w); }

/* common/reservoir.hpp:22-37 */
static inline void reservoir_update(OReservoir* r, const OSample* s, float weight, float u)
{
    r->w_sum += weight;
    r->M += 1;
    if (u < weight / r->w_sum) { r->sample = *s; }
}
static inline void reservoir_merge(OReservoir* r, const OReservoir* o, float weight, float u)
{
    r->w_sum += weight;
    r->M += o->M;
    if (u < weight / r->w_sum) { r->sample = o->sample; }
}
static inline float ucw_of(const OReservoir* r, float p_hat)
{
    return p_hat > 0.0f ? r->w_sum / ((float)r->M * p_hat) : 0.0f;
}

/* Function-level exports for pinning against oracle/_ref (bulk, n items each). */
/* the arithmetic of resolve, examples/10_restir_di/10_restir_di.cu:433-458, V (the shadow ray's answer, :443-444) given:
 * brdf = 1/PI * Kd (:437); G (:439-441); radiance = brdf * G * V * sample.radiance * ucw, left to right (:447);
 * accumulation += / = {radiance, 1} (:451-458). Used by o_resolve and, for the check against the reference's own functions
 * (oracle/ref_driver.cpp cmd_fn id 12), by o_fn_bulk. */
static inline v4 resolve_arithmetic(v3 color, v3 surf_p, v3 surf_n, v3 hit_position, v3 hit_normal, v3 sample_radiance, float ucw, float V,
                                    int accumulate, v4 accumulation)
{
    const v3 brdf = muls(color, 1.0f / O_PI);
    const float G = geometry_term(surf_p, surf_n, hit_position, hit_normal);
    const v3 radiance = muls(mulv(muls(muls(brdf, G), V), sample_radiance), ucw);
    if (accumulate)
    {
        accumulation.x += radiance.x; accumulation.y += radiance.y;
        accumulation.z += radiance.z; accumulation.w += 1.0f;
    }
    else
    {
        const v4 o = {radiance.x, radiance.y, radiance.z, 1.0f};
        accumulation = o;
    }
    return accumulation;
}

ORACLE_API void o_fn_bulk(int fn, const float* in, float* out, int n)
{
    for (int i = 0; i < n; ++i)
    {
        switch (fn)
        {
            case 0: { /* warp_unit_triangle: in 2 -> out 2 */
                v2 r = warp_unit_triangle(in[2 * i], in[2 * i + 1]);
                out[2 * i] = r.x; out[2 * i + 1] = r.y; break; }
            case 1: { /* sample_hemisphere: in 3 -> out 3 */
                v3 r = sample_hemisphere(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
                out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
            case 2: { /* sample_2d_gaussian: in 2 -> out 2 */
                v2 r = sample_2d_gaussian(in[2 * i], in[2 * i + 1]);
                out[2 * i] = r.x; out[2 * i + 1] = r.y; break; }
            case 3: { /* geometry_term: in 12 -> out 1 */
                const float* a = in + 12 * i;
                out[i] = geometry_term(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]),
                                       V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11])); break; }
            case 4: { /* intersect_ray_triangle: in 17 (ro,rd,tmin,tmax,v0,v1,v2) -> out 4 (hit,t,u,v) */
                const float* a = in + 17 * i;
                float t = 0, u = 0, v = 0;
                const int h = intersect_ray_triangle(&t, &u, &v, V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]),
                                                     a[6], a[7], V3(a[8], a[9], a[10]),
                                                     V3(a[11], a[12], a[13]), V3(a[14], a[15], a[16]));
                out[4 * i] = (float)h; out[4 * i + 1] = t; out[4 * i + 2] = u; out[4 * i + 3] = v; break; }
            case 5: { /* luminance: in 3 -> out 1 */
                out[i] = luminance(V3(in[3 * i], in[3 * i + 1], in[3 * i + 2])); break; }
            case 6: { /* normal_rejection_heuristics: in 6 -> out 1 */
                const float* a = in + 6 * i;
                out[i] = normal_rejection_heuristics(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5])); break; }
            case 7: { /* depth_rejection_heuristics: in 9 -> out 1 */
                const float* a = in + 9 * i;
                out[i] = depth_rejection_heuristics(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]),
                                                    V3(a[6], a[7], a[8])); break; }
            case 8: { /* triangle normal/area/tangent: in 9 -> out 7 */
                OTriangle t; const float* a = in + 9 * i;
                t.v[0] = V3(a[0], a[1], a[2]); t.v[1] = V3(a[3], a[4], a[5]); t.v[2] = V3(a[6], a[7], a[8]);
                const v3 nn = normal_of(&t); const v3 tg = a_tangent_of(&t);
                float* o = out + 7 * i;
                o[0] = nn.x; o[1] = nn.y; o[2] = nn.z; o[3] = area_of(&t); o[4] = tg.x; o[5] = tg.y; o[6] = tg.z;
                break; }
            case 9: { /* aces: in 1 -> out 1 (common/kernels/common.cu:19-28) */
                const float x = in[i];
                const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
                out[i] = (x * (a * x + b)) / (x * (c * x + d) + e); break; }
            case 10: { /* make_surface_info(ray, isect, triangles) core.hpp:152-165: in 16 (tri9, ro3, rd3, t) -> out 6 */
                OTriangle t; const float* a = in + 16 * i;
                t.v[0] = V3(a[0], a[1], a[2]); t.v[1] = V3(a[3], a[4], a[5]); t.v[2] = V3(a[6], a[7], a[8]);
                const Surf sf = make_surface_info_ray(&t, V3(a[9], a[10], a[11]), V3(a[12], a[13], a[14]), a[15]);
                float* o = out + 6 * i;
                o[0] = sf.p.x; o[1] = sf.p.y; o[2] = sf.p.z; o[3] = sf.n.x; o[4] = sf.n.y; o[5] = sf.n.z;
                break; }
            case 11: { /* make_tangent_basis + local_to_world core.hpp:216-235: in 15 (tri9, n3, v3) -> out 3 */
                OTriangle t; const float* a = in + 15 * i;
                t.v[0] = V3(a[0], a[1], a[2]); t.v[1] = V3(a[3], a[4], a[5]); t.v[2] = V3(a[6], a[7], a[8]);
                const v3 w = tangent_to_world(&t, V3(a[9], a[10], a[11]), V3(a[12], a[13], a[14]));
                float* o = out + 3 * i;
                o[0] = w.x; o[1] = w.y; o[2] = w.z;
                break; }
            case 12: { /* resolve's arithmetic, 10_restir_di.cu:433-458: in 25 (Kd3 p3 n3 hit_p3 hit_n3 Le3 ucw V accumulate prev4) -> out 4 */
                const float* a = in + 25 * i;
                const v4 prev = {a[21], a[22], a[23], a[24]};
                const v4 r = resolve_arithmetic(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11]),
                                                V3(a[12], a[13], a[14]), V3(a[15], a[16], a[17]), a[18], a[19], a[20] != 0.0f, prev);
                float* o = out + 4 * i;
                o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
                break; }
            /* raw math functions in the current mode: in 1 -> out 1 */
            case 20: out[i] = m_log(in[i]); break;
            case 21: out[i] = m_cos(in[i]); break;
            case 22: out[i] = m_sin(in[i]); break;
            case 28: { float sn, cs; pm_sincosf(in[i], &sn, &cs); out[i] = sn; break; } /* fused form of the device kernels */
            case 29: { float sn, cs; pm_sincosf(in[i], &sn, &cs); out[i] = cs; break; }
            case 23: out[i] = m_exp(in[i]); break;
            case 24: out[i] = m_pow8(in[i]); break;
            case 25: out[i] = m_pow(in[i], 1.0f / 2.2f); break;
            case 26: out[i] = in[2 * i] / in[2 * i + 1]; break; /* IEEE div (GPU check) */
            case 27: out[i] = sqrtf(in[i]); break;            /* IEEE sqrt (GPU check) */
            default: out[i] = 0.0f;
        }
    }
}

/* -------------------------------------------------------------- counters */
typedef struct
{
    long rays;           /* raytrace() invocations */
    long shaded_pixels;  /* pixels that hit a non-emissive triangle */
    long spatial_bytes;  /* SURVEY.md §8(d) algorithmic bytes of the spatial pass */
    long spatial_accepted; /* neighbours that passed on-screen/not-self tests */
    long spatial_merged;   /* neighbours that reached merge() */
} OCounters;

/* ---------------------------------------------------------------- kernels */
/* All kernels follow the reference's indexing: tid -> (xi = tid % W, yi = tid / W),
 * pixel_idx = xi + (H - yi - 1) * W. [row0,row1) restricts the *storage* rows
 * processed (row = H - yi - 1), for the multi-rank strip tests; pass 0,H for all. */

/* examples/10_restir_di/10_restir_di.cu:9-34 */
ORACLE_API void o_raycast(const OScene* s, int W, int H, const ORayGen* rg, OVisibility* vis,
                          int row0, int row1, OCounters* cnt)
{
    long rays = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            v3 ro, rd;
            raygen_shoot(rg, &ro, &rd, (float)xi / (float)W, (float)yi / (float)H);
            OHit h = {0.0f, {0.0f, 0.0f}, -1};
            raytrace(s, ro, rd, 0.0f, O_FLT_MAX, &h);
            ++rays;
            vis[pixel_idx].uv = h.uv;
            vis[pixel_idx].index = h.index;
            vis[pixel_idx].pad = 0;
        }
    }
    if (cnt) cnt->rays += rays;
}

static inline OReservoir reservoir_zero(void)
{
    OReservoir r;
    memset(&r, 0, sizeof(r));
    return r;
}

/* examples/10_restir_di/10_restir_di.cu:36-135 */
ORACLE_API void o_generate_candidate(const OScene* s, int W, int H, int frame, const OVisibility* vis,
                                     const float* eye3, const OOptions* opt, OReservoir* res,
                                     int row0, int row1, OCounters* cnt)
{
    const v3 eye = V3(eye3[0], eye3[1], eye3[2]);
    long rays = 0, shaded = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, shaded)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            const OVisibility v = vis[pixel_idx];
            if (v.index == -1) { res[pixel_idx] = reservoir_zero(); continue; }
            if (has_emission(&s->tris[v.index])) { res[pixel_idx] = reservoir_zero(); continue; }
            ++shaded;
            PCG rng = pcg_init(hashPCG4((uint32_t)xi, (uint32_t)yi, (uint32_t)frame, 0u), 0);
            const Surf surf = make_surface_info_eye(&v, s->tris, eye);
            OReservoir r = reservoir_zero();
            for (int i = 0; i < opt->ris_sample_count; ++i)
            {
                OSample smp;
                memset(&smp, 0, sizeof(smp));
                smp.origin_position = surf.p;
                smp.origin_normal = surf.n;
                /* draw order = left-to-right argument evaluation (clang/hipcc) */
                const float rv0 = pcg_uniformf(&rng);
                const float rv1 = pcg_uniformf(&rng);
                const float rv2 = pcg_uniformf(&rng);
                const LightSample ls = sample_light(s, rv0, rv1, rv2);
                smp.hit_position = ls.p;
                smp.hit_normal = ls.n;
                const OTriangle* lt = &s->tris[ls.index];
                smp.radiance = lt->emissive;
                const float light_pdf = 1.0f / (float)(size_t)s->n_lights * 1.0f / area_of(lt);
                long dummy = 0;
                const float p_hat = evaluate_target_function(s, surf.p, surf.n, smp.hit_position,
                                                             smp.hit_normal, smp.radiance, 0, &dummy);
                const float weight = p_hat / light_pdf;
                const float u = pcg_uniformf(&rng);
                reservoir_update(&r, &smp, weight, u);
            }
            {
                const float p_hat = evaluate_target_function(
                    s, surf.p, surf.n, r.sample.hit_position, r.sample.hit_normal, r.sample.radiance,
                    opt->use_shadowed_target_function, &rays);
                r.ucw = ucw_of(&r, p_hat);
            }
            if (opt->use_visibility_reuse)
            {
                const float V = check_visibility(s, surf.p, surf.n, r.sample.hit_position);
                ++rays;
                r.sample.visibility = (uint8_t)(V != 0.0f);
            }
            res[pixel_idx] = r;
        }
    }
    if (cnt) { cnt->rays += rays; cnt->shaded_pixels += shaded; }
}

/* examples/10_restir_di/10_restir_di.cu:137-237 */
ORACLE_API void o_temporal_resampling(const OScene* s, int W, int H, int frame, const OVisibility* vis,
                                      const float* eye3, const OOptions* opt, const OReservoir* prev,
                                      OReservoir* res, int row0, int row1, OCounters* cnt)
{
    const v3 eye = V3(eye3[0], eye3[1], eye3[2]);
    long rays = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            const OVisibility v = vis[pixel_idx];
            if (v.index == -1) continue;
            if (has_emission(&s->tris[v.index])) continue;
            if (!opt->use_temporal_resampling) continue;
            PCG rng = pcg_init(hashPCG4((uint32_t)xi, (uint32_t)yi, (uint32_t)frame, 1u), 0);
            const Surf surf = make_surface_info_eye(&v, s->tris, eye);
            OReservoir pr = prev[pixel_idx];
            OReservoir r = res[pixel_idx];
            const int cap = 20 * opt->ris_sample_count;
            pr.M = pr.M < cap ? pr.M : cap;
            float weight;
            {
                float p_hat_y = evaluate_target_function(
                    s, surf.p, surf.n, pr.sample.hit_position, pr.sample.hit_normal, pr.sample.radiance,
                    opt->use_shadowed_target_function, &rays);
                if (opt->use_visibility_reuse) { p_hat_y *= (float)pr.sample.visibility; }
                pr.M = scale_M(pr.M, rejection_heuristics(&r, &pr, eye));
                weight = p_hat_y * pr.ucw * (float)pr.M;
            }
            reservoir_merge(&r, &pr, weight, pcg_uniformf(&rng));
            {
                const float p_hat = evaluate_target_function(
                    s, surf.p, surf.n, r.sample.hit_position, r.sample.hit_normal, r.sample.radiance,
                    opt->use_shadowed_target_function, &rays);
                r.ucw = ucw_of(&r, p_hat);
            }
            res[pixel_idx] = r;
        }
    }
    if (cnt) cnt->rays += rays;
}

/* examples/10_restir_di/10_restir_di.cu:239-254 (call site 10_restir_di.cpp:314-321:
 * src = reservoir0, dst = temporal buffer) */
ORACLE_API void o_save_temporal_reservoir(int W, int H, const OReservoir* src, OReservoir* dst,
                                          int row0, int row1)
{
    for (int row = row0; row < row1; ++row)
        memcpy(dst + (size_t)row * W, src + (size_t)row * W, sizeof(OReservoir) * (size_t)W);
    (void)H;
}

/* examples/10_restir_di/10_restir_di.cu:256-388 */
ORACLE_API void o_spatial_resampling(const OScene* s, int W, int H, int frame, int pass,
                                     const OVisibility* vis, const float* eye3, const OOptions* opt,
                                     const OReservoir* in, OReservoir* out, int row0, int row1,
                                     OCounters* cnt)
{
    const v3 eye = V3(eye3[0], eye3[1], eye3[2]);
    long rays = 0, bytes = 0, accepted = 0, merged = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, bytes, accepted, merged)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            const OVisibility v = vis[pixel_idx];
            bytes += 16;
            if (v.index == -1) continue;
            if (has_emission(&s->tris[v.index])) continue;
            PCG rng = pcg_init(hashPCG4((uint32_t)xi, (uint32_t)yi, (uint32_t)frame, (uint32_t)(2 + pass)), 0);
            const Surf surf = make_surface_info_eye(&v, s->tris, eye);
            OReservoir r = in[pixel_idx];
            bytes += 76 + 76; /* own reservoir in + out */
            if (!opt->use_spatial_resampling) { out[pixel_idx] = r; continue; }
            for (int k = 0; k < opt->spatial_resampling_sample_count; ++k)
            {
                const float rv0 = pcg_uniformf(&rng);
                const float rv1 = pcg_uniformf(&rng);
                const v2 g = sample_2d_gaussian(rv0, rv1);
                const float fx = (float)xi + opt->spatial_resampling_radius / 1.96f * g.x;
                const float fy = (float)yi + opt->spatial_resampling_radius / 1.96f * g.y;
                /* float -> int is UB out of range; the GPU saturates, both are off-screen */
                const int x = f2i_sat(fx);
                const int y = f2i_sat(fy);
                if (x < 0 || x >= W || y < 0 || y >= H) continue;
                if (x == xi && y == yi) continue;
                const int pid = x + (H - y - 1) * W;
                const OVisibility nv = vis[pid];
                bytes += 16;
                ++accepted;
                if (nv.index == -1) continue;
                if (has_emission(&s->tris[nv.index])) continue;
                OReservoir nr = in[pid];
                bytes += 76;
                float weight;
                {
                    float p_hat_y = evaluate_target_function(
                        s, surf.p, surf.n, nr.sample.hit_position, nr.sample.hit_normal,
                        nr.sample.radiance, opt->use_shadowed_target_function, &rays);
                    if (opt->use_visibility_reuse) { p_hat_y *= (float)nr.sample.visibility; }
                    nr.M = scale_M(nr.M, rejection_heuristics(&r, &nr, eye));
                    weight = p_hat_y * nr.ucw * (float)nr.M;
                }
                reservoir_merge(&r, &nr, weight, pcg_uniformf(&rng));
                ++merged;
            }
            {
                const float p_hat = evaluate_target_function(
                    s, surf.p, surf.n, r.sample.hit_position, r.sample.hit_normal, r.sample.radiance,
                    opt->use_shadowed_target_function, &rays);
                r.ucw = ucw_of(&r, p_hat);
            }
            out[pixel_idx] = r;
        }
    }
    if (cnt)
    {
        cnt->rays += rays; cnt->spatial_bytes += bytes;
        cnt->spatial_accepted += accepted; cnt->spatial_merged += merged;
    }
}

/* examples/10_restir_di/10_restir_di.cu:390-459 */
ORACLE_API void o_resolve(const OScene* s, v4* accum, int W, int H, const OVisibility* vis,
                          const float* eye3, const OOptions* opt, const OReservoir* res, int row0,
                          int row1, OCounters* cnt)
{
    const v3 eye = V3(eye3[0], eye3[1], eye3[2]);
    long rays = 0;
    (void)H;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            const OVisibility v = vis[pixel_idx];
            if (v.index == -1)
            {
                const v4 z = {0.0f, 0.0f, 0.0f, 1.0f};
                accum[pixel_idx] = z;
                continue;
            }
            const OTriangle* tri = &s->tris[v.index];
            if (has_emission(tri))
            {
                const v4 e = {tri->emissive.x, tri->emissive.y, tri->emissive.z, 1.0f};
                accum[pixel_idx] = e;
                continue;
            }
            const Surf surf = make_surface_info_eye(&v, s->tris, eye);
            const OReservoir* r = &res[pixel_idx];
            const float V = check_visibility(s, surf.p, surf.n, r->sample.hit_position);
            ++rays;
            accum[pixel_idx] = resolve_arithmetic(tri->color, surf.p, surf.n, r->sample.hit_position, r->sample.hit_normal, r->sample.radiance,
                                                  r->ucw, V, opt->accumulate, accum[pixel_idx]);
        }
    }
    if (cnt) cnt->rays += rays;
}

/* common/kernels/common.cu:4-17 */
ORACLE_API void o_clear(v4* accum, int W, int H, int row0, int row1)
{
    for (int row = row0; row < row1; ++row)
        memset(accum + (size_t)row * W, 0, sizeof(v4) * (size_t)W);
    (void)H;
}

static inline float aces_tone_mapping(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return (x * (a * x + b)) / (x * (c * x + d) + e);
}
static inline uint8_t to_u8(float v)
{
    /* (uint8_t)clamp(v, 0, 255) with device min/max NaN behaviour */
    const float c = fminf(fmax_dev(v, 0.0f), 255.0f);
    return (uint8_t)(int)c;
}
/* common/kernels/common.cu:30-74 */
ORACLE_API void o_tone_mapping(uint8_t* pixels, const v4* accum, int W, int H, int row0, int row1)
{
    (void)H;
    for (int row = row0; row < row1; ++row)
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            const v4 a = accum[pixel_idx];
            v3 rad = V3(a.x / a.w, a.y / a.w, a.z / a.w);
            rad = muls(rad, 1.0f);
            rad.x = aces_tone_mapping(rad.x);
            rad.y = aces_tone_mapping(rad.y);
            rad.z = aces_tone_mapping(rad.z);
            const float gamma = 1.0f / 2.2f;
            rad.x = m_pow(rad.x, gamma);
            rad.y = m_pow(rad.y, gamma);
            rad.z = m_pow(rad.z, gamma);
            pixels[4 * pixel_idx + 0] = to_u8(rad.x * 255.0f);
            pixels[4 * pixel_idx + 1] = to_u8(rad.y * 255.0f);
            pixels[4 * pixel_idx + 2] = to_u8(rad.z * 255.0f);
            pixels[4 * pixel_idx + 3] = 255;
        }
}

/* One frame exactly as examples/10_restir_di/10_restir_di.cpp:257-379 sequences it.
 * r0, r1, temporal: the three reservoir buffers of :113-122. The buffer that resolve
 * read (r1 for an odd pass count) holds the final reservoirs. */
ORACLE_API void o_frame(const OScene* s, int W, int H, int frame, const ORayGen* rg, const float* eye3,
                        const OOptions* opt, OVisibility* vis, OReservoir* r0, OReservoir* r1,
                        OReservoir* temporal, v4* accum, uint8_t* pixels, OCounters* cnt)
{
    o_raycast(s, W, H, rg, vis, 0, H, cnt);
    o_generate_candidate(s, W, H, frame, vis, eye3, opt, r0, 0, H, cnt);
    o_temporal_resampling(s, W, H, frame, vis, eye3, opt, temporal, r0, 0, H, cnt);
    o_save_temporal_reservoir(W, H, r0, temporal, 0, H);
    OReservoir* in = r0;
    OReservoir* out = r1;
    for (int k = 0; k < opt->spatial_resampling_passes; ++k)
    {
        if (k != 0) { OReservoir* t = in; in = out; out = t; }
        o_spatial_resampling(s, W, H, frame, k, vis, eye3, opt, in, out, 0, H, cnt);
    }
    o_resolve(s, accum, W, H, vis, eye3, opt, out, 0, H, cnt);
    if (pixels) o_tone_mapping(pixels, accum, W, H, 0, H);
}

/* ------------------------------------------------- config #1: 04_ao kernel */
/* examples/04_ao/04_ao.cu:31-88 (brute force by construction, :8-29) */
ORACLE_API void o_ao_04(const OScene* s, uint8_t* pixels, const ORayGen* rg, int W, int H)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int yi = 0; yi < H; ++yi)
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixelIdx = xi + (H - yi - 1) * W;
            PCG rng = pcg_init(0, hashPCG3((uint32_t)xi, (uint32_t)yi, 42u));
            v3 ro, rd;
            raygen_shoot(rg, &ro, &rd, (float)xi / (float)W, (float)yi / (float)H);
            OHit h;
            if (raytrace(s, ro, rd, 0.0f, O_FLT_MAX, &h))
            {
                const OTriangle* tri = &s->tris[h.index];
                v3 n = normal_of(tri);
                if (0.0f < dot(n, rd)) { n = neg(n); }
                const v3 tangent0 = a_tangent_of(tri);
                const v3 tangent1 = cross(tangent0, n);
                const v3 p_hit = add(ro, muls(rd, h.t));
                const v3 ao_ro = add(p_hit, muls(n, 0.0001f));
                const int N_Rays = 64;
                int n_visible = 0;
                for (int i = 0; i < N_Rays; ++i)
                {
                    const float r0 = pcg_uniformf(&rng);
                    const float r1 = pcg_uniformf(&rng);
                    const float r2 = pcg_uniformf(&rng);
                    const v3 sm = sample_hemisphere(r0, r1, r2);
                    const v3 ao_rd = add(add(muls(tangent0, sm.x), muls(tangent1, sm.z)), muls(n, sm.y));
                    OHit ah;
                    if (!raytrace(s, ao_ro, ao_rd, 0.0f, O_FLT_MAX, &ah)) { n_visible++; }
                }
                const float ao = (float)n_visible / (float)N_Rays;
                const uint8_t c = (uint8_t)(int)(m_pow(ao, 1.0f / 2.2f) * 255.0f);
                pixels[pixelIdx * 4 + 0] = c; pixels[pixelIdx * 4 + 1] = c;
                pixels[pixelIdx * 4 + 2] = c; pixels[pixelIdx * 4 + 3] = 255;
            }
            else
            {
                pixels[pixelIdx * 4 + 0] = 32; pixels[pixelIdx * 4 + 1] = 32;
                pixels[pixelIdx * 4 + 2] = 32; pixels[pixelIdx * 4 + 3] = 255;
            }
        }
}

/* ------------------------------------- configs #2 / #3: 07_pt and 09_ris path tracers */
/* examples/07_pt/07_pt.cu:11-90 */
ORACLE_API void o_path_trace_07(const OScene* s, int W, int H, int frame, const ORayGen* rg,
                                const OOptions* opt, v4* accum, int row0, int row1, OCounters* cnt)
{
    long rays = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            PCG rng = pcg_init(hashPCG3((uint32_t)xi, (uint32_t)yi, (uint32_t)frame), 0);
            v3 ro, rd;
            raygen_shoot(rg, &ro, &rd, (float)xi / (float)W, (float)yi / (float)H);
            v3 radiance = V3(0.0f, 0.0f, 0.0f), throughput = V3(1.0f, 1.0f, 1.0f);
            for (int depth = 0; depth < opt->max_depth; ++depth)
            {
                OHit h;
                ++rays;
                if (!raytrace(s, ro, rd, 0.0f, O_FLT_MAX, &h))
                {
                    radiance = add(radiance, mulv(throughput, opt->sky_color));
                    break;
                }
                const OTriangle* tri = &s->tris[h.index];
                if (has_emission(tri))
                {
                    radiance = add(radiance, mulv(throughput, tri->emissive));
                    break;
                }
                const Surf surf = make_surface_info_ray(tri, ro, rd, h.t);
                const float r0 = pcg_uniformf(&rng);
                const float r1 = pcg_uniformf(&rng);
                const float r2 = pcg_uniformf(&rng);
                const v3 wo = tangent_to_world(tri, surf.n, sample_hemisphere(r0, r1, r2));
                throughput = mulv(throughput, tri->color);
                ro = add(surf.p, muls(surf.n, 0.001f));
                rd = wo;
            }
            if (opt->accumulate)
            {
                accum[pixel_idx].x += radiance.x; accum[pixel_idx].y += radiance.y;
                accum[pixel_idx].z += radiance.z; accum[pixel_idx].w += 1.0f;
            }
            else
            {
                const v4 o = {radiance.x, radiance.y, radiance.z, 1.0f};
                accum[pixel_idx] = o;
            }
        }
    }
    if (cnt) cnt->rays += rays;
}

/* examples/08_nee/08_nee.cu:11-140: path tracing with next-event estimation (one uniformly picked
 * light sample and one shadow ray per bounce; emission counted at depth 0 only) */
ORACLE_API void o_path_trace_08(const OScene* s, int W, int H, int frame, const ORayGen* rg,
                                const OOptions* opt, v4* accum, int row0, int row1, OCounters* cnt)
{
    long rays = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            PCG rng = pcg_init(hashPCG3((uint32_t)xi, (uint32_t)yi, (uint32_t)frame), 0); /* :27 */
            v3 ro, rd;
            raygen_shoot(rg, &ro, &rd, (float)xi / (float)W, (float)yi / (float)H);
            v3 radiance = V3(0.0f, 0.0f, 0.0f), throughput = V3(1.0f, 1.0f, 1.0f);
            for (int depth = 0; depth < opt->max_depth; ++depth)
            {
                OHit h;
                ++rays;
                if (!raytrace(s, ro, rd, 0.0f, O_FLT_MAX, &h)) break; /* :43-49, the sky is not a light here */
                const OTriangle* tri = &s->tris[h.index];
                if (has_emission(tri))
                {
                    if (depth == 0) radiance = add(radiance, mulv(throughput, tri->emissive)); /* :53-61 */
                    break;
                }
                const Surf surf = make_surface_info_ray(tri, ro, rd, h.t);
                /* :67-69: the three draws are function arguments, evaluated left to right (clang) */
                const float rv0 = pcg_uniformf(&rng);
                const float rv1 = pcg_uniformf(&rng);
                const float rv2 = pcg_uniformf(&rng);
                const LightSample ls = sample_light(s, rv0, rv1, rv2);
                {
                    const OTriangle* lt = &s->tris[ls.index];
                    const float V = check_visibility(s, surf.p, surf.n, ls.p); /* :76-77 */
                    ++rays;
                    const v3 brdf = muls(tri->color, 1.0f / O_PI);
                    const float G = geometry_term(surf.p, surf.n, ls.p, ls.n);
                    const float light_pdf = 1.0f / (float)(size_t)s->n_lights * 1.0f / area_of(lt);
                    /* :89-90: ((((throughput * brdf) * G) * V) * Le) / pdf */
                    const v3 c = divs(mulv(muls(muls(mulv(throughput, brdf), G), V), lt->emissive), light_pdf);
                    radiance = add(radiance, c);
                }
                const float r0 = pcg_uniformf(&rng);
                const float r1 = pcg_uniformf(&rng);
                const float r2 = pcg_uniformf(&rng);
                const v3 wo = tangent_to_world(tri, surf.n, sample_hemisphere(r0, r1, r2));
                throughput = mulv(throughput, tri->color);
                ro = add(surf.p, muls(surf.n, 0.001f));
                rd = wo;
            }
            if (opt->accumulate)
            {
                accum[pixel_idx].x += radiance.x; accum[pixel_idx].y += radiance.y;
                accum[pixel_idx].z += radiance.z; accum[pixel_idx].w += 1.0f;
            }
            else
            {
                const v4 o = {radiance.x, radiance.y, radiance.z, 1.0f};
                accum[pixel_idx] = o;
            }
        }
    }
    if (cnt) cnt->rays += rays;
}

/* examples/09_ris/09_ris.cu:11-166 */
ORACLE_API void o_path_trace_09(const OScene* s, int W, int H, int frame, const ORayGen* rg,
                                const OOptions* opt, v4* accum, int row0, int row1, OCounters* cnt)
{
    long rays = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays)
    for (int row = row0; row < row1; ++row)
    {
        const int yi = H - 1 - row;
        for (int xi = 0; xi < W; ++xi)
        {
            const int pixel_idx = xi + row * W;
            PCG rng = pcg_init(hashPCG3((uint32_t)xi, (uint32_t)yi, (uint32_t)frame), 0);
            v3 ro, rd;
            raygen_shoot(rg, &ro, &rd, (float)xi / (float)W, (float)yi / (float)H);
            v3 radiance = V3(0.0f, 0.0f, 0.0f), throughput = V3(1.0f, 1.0f, 1.0f);
            for (int depth = 0; depth < opt->max_depth; ++depth)
            {
                OHit h;
                ++rays;
                if (!raytrace(s, ro, rd, 0.0f, O_FLT_MAX, &h)) break;
                const OTriangle* tri = &s->tris[h.index];
                if (has_emission(tri))
                {
                    if (depth == 0) radiance = add(radiance, mulv(throughput, tri->emissive));
                    break;
                }
                const Surf surf = make_surface_info_ray(tri, ro, rd, h.t);
                OReservoir r = reservoir_zero();
                for (int i = 0; i < opt->ris_sample_count; ++i)
                {
                    OSample smp;
                    memset(&smp, 0, sizeof(smp));
                    smp.origin_position = surf.p;
                    smp.origin_normal = surf.n;
                    const float rv0 = pcg_uniformf(&rng);
                    const float rv1 = pcg_uniformf(&rng);
                    const float rv2 = pcg_uniformf(&rng);
                    const LightSample ls = sample_light(s, rv0, rv1, rv2);
                    smp.hit_position = ls.p;
                    smp.hit_normal = ls.n;
                    const OTriangle* lt = &s->tris[ls.index];
                    smp.radiance = lt->emissive;
                    const float light_pdf = 1.0f / (float)(size_t)s->n_lights * 1.0f / area_of(lt);
                    const float p_hat = evaluate_target_function(s, surf.p, surf.n, smp.hit_position, smp.hit_normal,
                                                                 smp.radiance, opt->use_shadowed_target_function, &rays);
                    const float weight = p_hat / light_pdf;
                    reservoir_update(&r, &smp, weight, pcg_uniformf(&rng));
                }
                {
                    const v3 brdf = muls(tri->color, 1.0f / O_PI);
                    const float G = geometry_term(surf.p, surf.n, r.sample.hit_position, r.sample.hit_normal);
                    const float V = check_visibility(s, surf.p, surf.n, r.sample.hit_position);
                    ++rays;
                    const float p_hat = evaluate_target_function(s, surf.p, surf.n, r.sample.hit_position,
                                                                 r.sample.hit_normal, r.sample.radiance,
                                                                 opt->use_shadowed_target_function, &rays);
                    const float ucw = ucw_of(&r, p_hat);
                    const v3 c = muls(mulv(muls(muls(mulv(throughput, brdf), G), V), r.sample.radiance), ucw);
                    radiance = add(radiance, c);
                }
                const float r0 = pcg_uniformf(&rng);
                const float r1 = pcg_uniformf(&rng);
                const float r2 = pcg_uniformf(&rng);
                const v3 wo = tangent_to_world(tri, surf.n, sample_hemisphere(r0, r1, r2));
                throughput = mulv(throughput, tri->color);
                ro = add(surf.p, muls(surf.n, 0.001f));
                rd = wo;
            }
            if (opt->accumulate)
            {
                accum[pixel_idx].x += radiance.x; accum[pixel_idx].y += radiance.y;
                accum[pixel_idx].z += radiance.z; accum[pixel_idx].w += 1.0f;
            }
            else
            {
                const v4 o = {radiance.x, radiance.y, radiance.z, 1.0f};
                accum[pixel_idx] = o;
            }
        }
    }
    if (cnt) cnt->rays += rays;
}
