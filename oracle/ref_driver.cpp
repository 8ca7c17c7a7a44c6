/*
 * ref_driver.cpp — host harness around the REFERENCE'S OWN kernel sources.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/restir_oracle.c header).
 *
 * This file contains no restated algorithm. It #includes the reference's files where
 * they lie under $(REF) (= /root/reference) and calls their functions/kernels on the
 * host, so that oracle/restir_oracle.c (MATH_LIBM mode) can be checked bit for bit
 * against the real thing and golden fixtures can be generated from it
 * (tests/golden/make_golden.py). Built by oracle/Makefile into oracle/_ref/ref_kernels
 * with the ROCm clang++ (clang evaluates call arguments left to right like hipcc; the
 * reference relies on that order for its RNG draws, 10_restir_di.cu:88-89,310).
 *
 * What is NOT here: any stand-in for HIPRT or Orochi. The reference's HIPRT headers are
 * used as shipped (declarations only). The HIPRT device implementation is a missing
 * binary blob, so raytrace() (common/raytrace.hpp:18-43) cannot run: the executable is
 * linked with --unresolved-symbols=ignore-all and this driver only ever runs code paths
 * that never reach raytrace():
 *   - every inline function of common/{rng,core,camera,math,reservoir}.hpp;
 *   - examples/04_ao/04_ao.cu kernelMain (brute force, HIPRT-free)            [config #1]
 *   - generate_candidate with use_visibility_reuse = use_shadowed_target_function = false
 *   - temporal_resampling / spatial_resampling with use_shadowed_target_function = false
 *   - save_temporal_reservoir, clear, tone_mapping.
 * raycast, the ray of resolve and the visibility-reuse ray of generate_candidate stay unpinned by
 * the reference (DESIGN.md "Oracle"); resolve's ARITHMETIC is pinned through the reference's own functions with the ray's
 * answer V as an input (cmd_fn id 12, ref::resolve_arithmetic below).
 *
 * Kernel parameters of type TypedBuffer<T> are declared by value in the reference; the
 * type has a deleted copy constructor, so the Itanium C++ ABI passes it by invisible
 * reference. The casts below spell that out.
 */
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_vector_types.h>

/* device-style min/max (fmaxf/fminf semantics for float) */
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline int max(int a, int b) { return a < b ? b : a; }
static inline int min(int a, int b) { return a < b ? a : b; }

#define __device__
#define __host__
#define __global__
#ifndef __shared__
#define __shared__ static thread_local
#endif
struct ref_dim3 { unsigned x, y, z; };
static thread_local ref_dim3 threadIdx, blockIdx, blockDim;

#define __HIPCC__ 1
namespace ref
{
#include "examples/10_restir_di/10_restir_di.cu"
}
namespace ref
{
/* The ARITHMETIC of the reference's resolve kernel (examples/10_restir_di/10_restir_di.cu:433-458) on given operands, through
 * the reference's own functions, constants and vector operators, statement for statement and operand for operand in the kernel's
 * order (`1.0f / PI * color`, `brdf * G * V * radiance * ucw`, the float4 `+=` / `=` of the accumulation). The kernel itself
 * cannot run here: its V = check_visibility(...) reaches raytrace(), i.e. HIPRT, a missing binary — so V is an INPUT. This is what
 * pins the expression order of the oracle's o_resolve. */
static float4 resolve_arithmetic(float3 color, float3 surf_p, float3 surf_n, float3 hit_position, float3 hit_normal, float3 sample_radiance,
                                 float ucw, float V, bool accumulate, float4 accumulation)
{
    float3 radiance;
    {
        const float3 brdf = 1.0f / PI * color;
        const float G = geometry_term(surf_p, surf_n, hit_position, hit_normal);
        radiance = brdf * G * V * sample_radiance * ucw;
    }
    if (accumulate) { accumulation += {radiance.x, radiance.y, radiance.z, 1.0f}; }
    else { accumulation = {radiance.x, radiance.y, radiance.z, 1.0f}; }
    return accumulation;
}
}  // namespace ref
namespace ref_ao
{
/* 04_ao.cu re-includes the same pragma-once headers; only its two functions are new */
using namespace ref;
#include "examples/04_ao/04_ao.cu"
}  // namespace ref_ao

/* ------------------------------------------------------------ blob files */
typedef std::map<std::string, std::vector<uint8_t>> Blobs;

static Blobs read_blobs(const char* path)
{
    Blobs b;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    for (;;)
    {
        char name[32];
        uint64_t n;
        if (fread(name, 1, 32, f) != 32) break;
        if (fread(&n, 8, 1, f) != 1) break;
        std::vector<uint8_t> d(n);
        if (n && fread(d.data(), 1, n, f) != n) { fprintf(stderr, "short blob\n"); exit(2); }
        name[31] = 0;
        b[name] = std::move(d);
    }
    fclose(f);
    return b;
}
static void write_blob(FILE* f, const char* name, const void* data, uint64_t n)
{
    char nm[32];
    memset(nm, 0, 32);
    strncpy(nm, name, 31);
    fwrite(nm, 1, 32, f);
    fwrite(&n, 8, 1, f);
    if (n) fwrite(data, 1, n, f);
}
template <class T>
static T* ptr(Blobs& b, const char* name)
{
    auto it = b.find(name);
    if (it == b.end()) { fprintf(stderr, "missing blob %s\n", name); exit(2); }
    return reinterpret_cast<T*>(it->second.data());
}
static int geti(Blobs& b, const char* name) { return *ptr<int>(b, name); }
template <class T>
static size_t count(Blobs& b, const char* name) { return b[name].size() / sizeof(T); }

template <class T>
static void view(ref::TypedBuffer<T>& tb, T* data, size_t n)
{
    tb.m_data = data;
    tb.m_size = n;
}

template <class K>
static void run_grid(int n_threads, K&& body)
{
    blockDim.x = 256; blockDim.y = blockDim.z = 1;
    const int blocks = (n_threads + 255) / 256;
    for (int b = 0; b < blocks; ++b)
    {
        blockIdx.x = b; blockIdx.y = blockIdx.z = 0;
        for (int t = 0; t < 256; ++t)
        {
            threadIdx.x = t; threadIdx.y = threadIdx.z = 0;
            body();
        }
    }
}

using TBTri = ref::TypedBuffer<ref::Triangle>;
using TBVis = ref::TypedBuffer<ref::Visibility>;
using TBRes = ref::TypedBuffer<ref::Reservoir>;
using TBU32 = ref::TypedBuffer<uint32_t>;
using TBF4 = ref::TypedBuffer<float4>;
using TBU8 = ref::TypedBuffer<uint8_t>;

typedef void (*gen_fn)(int, int, int, ref::hiprtGeometry, const TBTri&, const TBVis&, float3, const TBU32&,
                       ref::Options, TBRes&);
typedef void (*temporal_fn)(int, int, int, ref::hiprtGeometry, const TBTri&, const TBVis&, float3, ref::Options,
                            const TBRes&, TBRes&);
typedef void (*save_fn)(int, int, const TBRes&, TBRes&);
typedef void (*spatial_fn)(int, int, int, int, ref::hiprtGeometry, const TBTri&, const TBVis&, float3,
                           ref::Options, const TBRes&, TBRes&);
typedef void (*clear_fn)(TBF4&, int, int);
typedef void (*tone_fn)(TBU8&, TBF4&, int, int);
typedef void (*ao_fn)(TBU8&, ref::RayGenerator, int, int, const TBTri&);

static int cmd_kat()
{
    printf("{\n");
    printf("\"hashPCG\": [%u, %u, %u],\n", ref::hashPCG(0), ref::hashPCG(1), ref::hashPCG(42));
    printf("\"hashPCG3_3_5_42\": %u,\n", ref::hashPCG3(3, 5, 42));
    printf("\"hashPCG4_3_5_1_0\": %u,\n", ref::hashPCG4(3, 5, 1, 0));
    printf("\"hashPCG4_1919_1079_1_4\": %u,\n", ref::hashPCG4(1919, 1079, 1, 4));
    {
        ref::PCG r(2157792022u, 0);
        const unsigned a = r.uniform(), b = r.uniform();
        const float c = r.uniformf(), d = r.uniformf();
        printf("\"pcg_a\": [%u, %u, %.9g, %.9g, %llu, %llu],\n", a, b, c, d,
               (unsigned long long)r.state, (unsigned long long)r.inc);
    }
    {
        ref::PCG r(0, 2946961066u);
        const unsigned long long inc = r.inc;
        const unsigned a = r.uniform();
        const float c = r.uniformf();
        printf("\"pcg_b\": [%llu, %u, %.9g],\n", inc, a, c);
    }
    printf("\"sizeof\": {\"Triangle\": %zu, \"Visibility\": %zu, \"ReservoirSample\": %zu, "
           "\"Reservoir\": %zu, \"Options\": %zu, \"RayGenerator\": %zu, \"TypedBuffer\": %zu}\n",
           sizeof(ref::Triangle), sizeof(ref::Visibility), sizeof(ref::ReservoirSample),
           sizeof(ref::Reservoir), sizeof(ref::Options), sizeof(ref::RayGenerator), sizeof(TBTri));
    printf("}\n");
    return 0;
}

static float3 f3(const float* a) { return make_float3(a[0], a[1], a[2]); }

static void cmd_fn(Blobs& b, FILE* out)
{
    const int fn = geti(b, "fn");
    const float* in = ptr<float>(b, "in");
    const int n = geti(b, "n");
    std::vector<float> o;
    for (int i = 0; i < n; ++i)
    {
        switch (fn)
        {
            case 0: { float2 r = ref::warp_unit_triangle(in[2 * i], in[2 * i + 1]); o.push_back(r.x); o.push_back(r.y); break; }
            case 1: { float3 r = ref::sample_hemisphere(in[3 * i], in[3 * i + 1], in[3 * i + 2]); o.push_back(r.x); o.push_back(r.y); o.push_back(r.z); break; }
            case 2: { float2 r = ref::sample_2d_gaussian(in[2 * i], in[2 * i + 1]); o.push_back(r.x); o.push_back(r.y); break; }
            case 3: { const float* a = in + 12 * i; o.push_back(ref::geometry_term(f3(a), f3(a + 3), f3(a + 6), f3(a + 9))); break; }
            case 4: {
                const float* a = in + 17 * i; float t = 0, u = 0, v = 0;
                const bool h = ref::intersect_ray_triangle(&t, &u, &v, f3(a), f3(a + 3), a[6], a[7], f3(a + 8), f3(a + 11), f3(a + 14));
                o.push_back(h ? 1.0f : 0.0f); o.push_back(t); o.push_back(u); o.push_back(v); break; }
            case 5: { o.push_back(ref::luminance(f3(in + 3 * i))); break; }
            case 6: { const float* a = in + 6 * i; o.push_back(ref::normal_rejection_heuristics(f3(a), f3(a + 3))); break; }
            case 7: { const float* a = in + 9 * i; o.push_back(ref::depth_rejection_heuristics(f3(a), f3(a + 3), f3(a + 6))); break; }
            case 8: {
                ref::Triangle t; const float* a = in + 9 * i;
                t.vertices[0] = f3(a); t.vertices[1] = f3(a + 3); t.vertices[2] = f3(a + 6);
                float3 nn = ref::normal_of(t), tg = ref::a_tangent_of(t);
                o.push_back(nn.x); o.push_back(nn.y); o.push_back(nn.z); o.push_back(ref::area_of(t));
                o.push_back(tg.x); o.push_back(tg.y); o.push_back(tg.z); break; }
            case 9: { o.push_back(ref::aces_tone_mapping(in[i])); break; }
            case 10: { /* make_surface_info(ray, isect, triangles), common/core.hpp:152-165 */
                const float* a = in + 16 * i;
                ref::Triangle t; t.vertices[0] = f3(a); t.vertices[1] = f3(a + 3); t.vertices[2] = f3(a + 6);
                TBTri tb; view(tb, &t, 1);
                ref::Ray ray = ref::make_ray(f3(a + 9), f3(a + 12));
                ref::Intersection is; is.t = a[15]; is.index = 0;
                ref::SurfaceInfo sf = ref::make_surface_info(ray, is, tb);
                o.push_back(sf.p.x); o.push_back(sf.p.y); o.push_back(sf.p.z); o.push_back(sf.n.x); o.push_back(sf.n.y); o.push_back(sf.n.z);
                break; }
            case 11: { /* make_tangent_basis + local_to_world, common/core.hpp:216-235 */
                const float* a = in + 15 * i;
                ref::Triangle t; t.vertices[0] = f3(a); t.vertices[1] = f3(a + 3); t.vertices[2] = f3(a + 6);
                TBTri tb; view(tb, &t, 1);
                ref::TangentBasis b = ref::make_tangent_basis(f3(a + 9), 0, tb);
                float3 w = ref::local_to_world(f3(a + 12), b);
                o.push_back(w.x); o.push_back(w.y); o.push_back(w.z);
                break; }
            case 12: { /* resolve's arithmetic, 10_restir_di.cu:433-458: in 25 (Kd3 p3 n3 hit_p3 hit_n3 Le3 ucw V accumulate prev4) -> out 4 */
                const float* a = in + 25 * i;
                const float4 r = ref::resolve_arithmetic(f3(a), f3(a + 3), f3(a + 6), f3(a + 9), f3(a + 12), f3(a + 15), a[18], a[19], a[20] != 0.0f,
                                                         make_float4(a[21], a[22], a[23], a[24]));
                o.push_back(r.x); o.push_back(r.y); o.push_back(r.z); o.push_back(r.w);
                break; }
            default: o.push_back(0.0f);
        }
    }
    write_blob(out, "out", o.data(), o.size() * 4);
}

static void cmd_camera(Blobs& b, FILE* out)
{
    const float* p = ptr<float>(b, "cam"); /* eye3 center3 up3 fovy */
    const int W = geti(b, "W"), H = geti(b, "H");
    ref::RayGenerator rg;
    rg.lookat(f3(p), f3(p + 3), f3(p + 6), p[9], W, H);
    write_blob(out, "raygen", &rg, sizeof(rg));
    const float* uv = ptr<float>(b, "uv");
    const size_t n = count<float>(b, "uv") / 2;
    std::vector<float> o;
    for (size_t i = 0; i < n; ++i)
    {
        float3 ro, rd;
        rg.shoot(&ro, &rd, uv[2 * i], uv[2 * i + 1]);
        o.push_back(ro.x); o.push_back(ro.y); o.push_back(ro.z);
        o.push_back(rd.x); o.push_back(rd.y); o.push_back(rd.z);
    }
    write_blob(out, "rays", o.data(), o.size() * 4);
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_kernels <cmd> [in.bin out.bin]\n"); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "kat") return cmd_kat();
    if (argc < 4) { fprintf(stderr, "need in/out files\n"); return 2; }
    Blobs b = read_blobs(argv[2]);
    FILE* out = fopen(argv[3], "wb");
    if (!out) return 2;

    if (cmd == "fn") { cmd_fn(b, out); fclose(out); return 0; }
    if (cmd == "camera") { cmd_camera(b, out); fclose(out); return 0; }

    const int W = geti(b, "W"), H = geti(b, "H");
    const int N = W * H;

    if (cmd == "clear")
    {
        TBF4 acc; view(acc, ptr<float4>(b, "accum"), (size_t)N);
        auto f = reinterpret_cast<clear_fn>(&ref::clear);
        run_grid(N, [&] { f(acc, W, H); });
        write_blob(out, "accum", acc.m_data, (uint64_t)N * 16);
    }
    else if (cmd == "tone_mapping")
    {
        TBF4 acc; view(acc, ptr<float4>(b, "accum"), (size_t)N);
        std::vector<uint8_t> px((size_t)N * 4, 0);
        TBU8 pix; view(pix, px.data(), px.size());
        auto f = reinterpret_cast<tone_fn>(&ref::tone_mapping);
        run_grid(N, [&] { f(pix, acc, W, H); });
        write_blob(out, "pixels", px.data(), px.size());
    }
    else if (cmd == "ao04")
    {
        TBTri tris; view(tris, ptr<ref::Triangle>(b, "tris"), count<ref::Triangle>(b, "tris"));
        ref::RayGenerator rg = *ptr<ref::RayGenerator>(b, "raygen");
        std::vector<uint8_t> px((size_t)N * 4, 0);
        TBU8 pix; view(pix, px.data(), px.size());
        auto f = reinterpret_cast<ao_fn>(&ref_ao::kernelMain);
        run_grid(N, [&] { f(pix, rg, W, H, tris); });
        write_blob(out, "pixels", px.data(), px.size());
    }
    else
    {
        TBTri tris; view(tris, ptr<ref::Triangle>(b, "tris"), count<ref::Triangle>(b, "tris"));
        TBVis vis; view(vis, ptr<ref::Visibility>(b, "vis"), (size_t)N);
        ref::Options opt = *ptr<ref::Options>(b, "options");
        const float3 eye = f3(ptr<float>(b, "eye"));
        const int frame = geti(b, "frame");
        if (opt.use_shadowed_target_function)
        {
            fprintf(stderr, "shadowed target function needs HIPRT: not runnable\n");
            return 3;
        }
        if (cmd == "generate_candidate")
        {
            if (opt.use_visibility_reuse)
            {
                fprintf(stderr, "visibility reuse needs HIPRT: not runnable\n");
                return 3;
            }
            TBU32 lights; view(lights, ptr<uint32_t>(b, "lights"), count<uint32_t>(b, "lights"));
            std::vector<ref::Reservoir> r((size_t)N);
            memset((void*)r.data(), 0xCD, sizeof(ref::Reservoir) * (size_t)N);
            TBRes res; view(res, r.data(), r.size());
            auto f = reinterpret_cast<gen_fn>(&ref::generate_candidate);
            run_grid(N, [&] { f(W, H, frame, nullptr, tris, vis, eye, lights, opt, res); });
            write_blob(out, "res", r.data(), sizeof(ref::Reservoir) * (size_t)N);
        }
        else if (cmd == "temporal_resampling")
        {
            TBRes prev; view(prev, ptr<ref::Reservoir>(b, "prev"), (size_t)N);
            TBRes res; view(res, ptr<ref::Reservoir>(b, "res"), (size_t)N);
            auto f = reinterpret_cast<temporal_fn>(&ref::temporal_resampling);
            run_grid(N, [&] { f(W, H, frame, nullptr, tris, vis, eye, opt, prev, res); });
            write_blob(out, "res", res.m_data, sizeof(ref::Reservoir) * (size_t)N);
        }
        else if (cmd == "save_temporal_reservoir")
        {
            TBRes src; view(src, ptr<ref::Reservoir>(b, "res"), (size_t)N);
            std::vector<ref::Reservoir> d((size_t)N);
            TBRes dst; view(dst, d.data(), d.size());
            auto f = reinterpret_cast<save_fn>(&ref::save_temporal_reservoir);
            run_grid(N, [&] { f(W, H, src, dst); });
            write_blob(out, "res", d.data(), sizeof(ref::Reservoir) * (size_t)N);
        }
        else if (cmd == "spatial_resampling")
        {
            const int pass = geti(b, "pass");
            TBRes in; view(in, ptr<ref::Reservoir>(b, "res"), (size_t)N);
            std::vector<ref::Reservoir> o((size_t)N);
            memset((void*)o.data(), 0xCD, sizeof(ref::Reservoir) * (size_t)N);
            TBRes outb; view(outb, o.data(), o.size());
            auto f = reinterpret_cast<spatial_fn>(&ref::spatial_resampling);
            run_grid(N, [&] { f(W, H, frame, pass, nullptr, tris, vis, eye, opt, in, outb); });
            write_blob(out, "res", o.data(), sizeof(ref::Reservoir) * (size_t)N);
        }
        else { fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 2; }
    }
    fclose(out);
    return 0;
}
