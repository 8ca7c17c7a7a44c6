/*
 * host_path.h — the host-side (CPU, no GPU call) part of the product: the reference's RayGenerator
 * (common/camera.hpp:5-36) and BASELINE config #1, the `kernelMain` of examples/04_ao/04_ao.cu:31-88 as a C++
 * loop over image rows ("04_ao cornellbox1.obj 256x256 1spp as host-side C++ loop over common/raytrace.hpp
 * headers (no GPU, plumbing)").
 *
 * Built on the same [parity] functions the device kernels use (rt_device.h: PCG, hashPCG3, vector ops,
 * intersect_ray_triangle, tri_normal) compiled as plain C++; transcendental functions are the host's libm
 * (cosf / sinf / powf / tanf), which is what the reference's host-compiled kernel calls — this path is compared
 * byte for byte with the reference's own kernel run on the host (tests/golden/ref_ao04_256.npz), not with the GPU.
 * Compile with -ffp-contract=off. The test infrastructure (the CPU checker directory) is not used here in any form.
 */
#pragma once
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/restir_rt.h"
#include "rt_device.h"

namespace rt_host
{
using namespace rt;

/* RayGenerator::lookat, common/camera.hpp:11-25 (tan of a float argument = tanf) */
inline void raygen_lookat(rt_raygen* rg, const float eye[3], const float center[3], const float up[3], float fovy, int W, int H)
{
    const f3 e = F3(eye[0], eye[1], eye[2]), ce = F3(center[0], center[1], center[2]), u0 = F3(up[0], up[1], up[2]);
    const f3 f = normalize(ce - e);
    const f3 s = normalize(cross(f, u0));
    const f3 u = cross(s, f);
    const float tanThetaY = tanf(fovy * 0.5f);
    const float tanThetaX = tanThetaY / (float)H * (float)W;
    const f3 r = s * tanThetaX, uu = u * tanThetaY;
    rg->origin[0] = e.x; rg->origin[1] = e.y; rg->origin[2] = e.z;
    rg->right[0] = r.x; rg->right[1] = r.y; rg->right[2] = r.z;
    rg->up[0] = uu.x; rg->up[1] = uu.y; rg->up[2] = uu.z;
}
/* RayGenerator::shoot, common/camera.hpp:27-35 */
inline void raygen_shoot(const rt_raygen& rg, float u, float v, f3& ro, f3& rd)
{
    const f3 o = F3(rg.origin[0], rg.origin[1], rg.origin[2]), right = F3(rg.right[0], rg.right[1], rg.right[2]),
             up = F3(rg.up[0], rg.up[1], rg.up[2]);
    const f3 forward = normalize(cross(up, right));
    const f3 to = o + forward + mix(-right, right, u) + mix(up, -up, v);
    ro = o;
    rd = normalize(to - o);
}

/* closeset_hit, examples/04_ao/04_ao.cu:8-29: every triangle in index order, the interval's far end follows the
 * closest hit so far and is inclusive — of two hits at the same t the later triangle wins */
inline bool closest_hit_all(const rt_triangle* tris, uint32_t n, f3 ro, f3 rd, float& t_out, int& index_out)
{
    float t = kFltMax;
    int index = -1;
    for (uint32_t i = 0; i < n; ++i)
    {
        const rt_triangle& T = tris[i];
        float u, v;
        if (intersect_ray_triangle(t, u, v, ro, rd, 0.0f, t, F3(T.v[0][0], T.v[0][1], T.v[0][2]), F3(T.v[1][0], T.v[1][1], T.v[1][2]),
                                   F3(T.v[2][0], T.v[2][1], T.v[2][2])))
            index = (int)i;
    }
    if (index < 0) return false;
    t_out = t;
    index_out = index;
    return true;
}

/* sample_hemisphere, common/core.hpp:76-89, with the host's libm as the reference's host build has it */
inline f3 sample_hemisphere_libm(float r0, float r1, float r2)
{
    const float theta = r0 * 2.0f * kPI;
    float radius = r1 + r2;
    if (1.0f < radius) radius = 2.0f - radius;
    const float x = cosf(theta) * radius;
    const float z = sinf(theta) * radius;
    const float a = 1.0f - radius * radius;
    const float y = sqrtf(a < 0.0f ? 0.0f : a);
    return F3(x, y, z);
}

/* one image row of kernelMain (04_ao.cu:31-88): thread order is top-down (yi), storage bottom-up (pixelIdx) */
inline void ao04_row(const rt_triangle* tris, uint32_t n, const rt_raygen& rg, int W, int H, int yi, uint8_t* pixels)
{
    for (int xi = 0; xi < W; ++xi)
    {
        const size_t pixelIdx = (size_t)xi + (size_t)(H - yi - 1) * (size_t)W;
        PCG random = pcg_init(0, hashPCG3((uint32_t)xi, (uint32_t)yi, 42u)); /* seed 0, SEQUENCE = the hash (:42) */
        f3 ro, rd;
        raygen_shoot(rg, (float)xi / (float)W, (float)yi / (float)H, ro, rd);
        float t;
        int index;
        uint8_t* px = pixels + 4 * pixelIdx;
        if (!closest_hit_all(tris, n, ro, rd, t, index))
        {
            px[0] = 32; px[1] = 32; px[2] = 32; px[3] = 255;
            continue;
        }
        const rt_triangle& T = tris[index];
        const f3 v0 = F3(T.v[0][0], T.v[0][1], T.v[0][2]), v1 = F3(T.v[1][0], T.v[1][1], T.v[1][2]), v2 = F3(T.v[2][0], T.v[2][1], T.v[2][2]);
        f3 nrm = tri_normal(v0, v1, v2); /* normal_of, common/core.hpp:50-55 */
        if (0.0f < dot(nrm, rd)) nrm = -nrm;
        const f3 tangent0 = normalize(v1 - v0); /* a_tangent_of, common/core.hpp:45-48 */
        const f3 tangent1 = cross(tangent0, nrm);
        const f3 p_hit = ro + rd * t;
        const f3 ao_ro = p_hit + nrm * 0.0001f;
        const int N_Rays = 64;
        int n_visible = 0;
        for (int i = 0; i < N_Rays; ++i)
        {
            /* three draws in argument order (left to right: clang / hipcc, and the braces below fix it here) */
            const float r0 = random.uniformf();
            const float r1 = random.uniformf();
            const float r2 = random.uniformf();
            const f3 s = sample_hemisphere_libm(r0, r1, r2);
            const f3 ao_rd = tangent0 * s.x + tangent1 * s.z + nrm * s.y;
            float at;
            int ai;
            if (!closest_hit_all(tris, n, ao_ro, ao_rd, at, ai)) ++n_visible;
        }
        const float ao = (float)n_visible / (float)N_Rays;
        const uint8_t c = (uint8_t)(powf(ao, 1.0f / 2.2f) * 255.0f);
        px[0] = c; px[1] = c; px[2] = c; px[3] = 255;
    }
}

/* the whole image on `threads` host threads (0 = hardware_concurrency), rows dealt out dynamically;
 * pixels: W * H RGBA8 in the reference's storage order */
inline void ao04_image(const rt_triangle* tris, uint32_t n, const rt_raygen& rg, int W, int H, int threads, uint8_t* pixels)
{
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    if (threads > H) threads = H;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (int yi = next.fetch_add(1); yi < H; yi = next.fetch_add(1)) ao04_row(tris, n, rg, W, H, yi, pixels);
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < threads; ++k) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

}  // namespace rt_host
