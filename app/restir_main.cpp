/*
 * restir_main.cpp — headless C++ host of the hot path: the frame loop of the reference's
 * examples/10_restir_di/10_restir_di.cpp:23-416 without GLFW/OpenGL/Orochi/HIPRT, written
 * against the C-ABI only: include/restir_rt.h for everything the example's frame loop does; restir_rt_internal.h
 * supplies what its measurement switches add (--mirror / --shm stand-in transports, --cost-strips' rt_row_shaded,
 * the path tracers' ray counter).
 *
 *   restir_app [--obj scene.obj | --tris scene.tris] [--size W H] [--frames N]
 *              [--eye x y z] [--lookat x y z] [--temporal 0|1] [--spatial 0|1]
 *              [--shadowed 0|1] [--visreuse 0|1] [--accumulate 0|1] [--by-kernel]
 *              [--example 10|7|8|9|4] [--ppm out.ppm] [--png out.png] [--pfm out.pfm] [--rgba out.raw] [--dump-tris out.tris]
 *              [--ranks N [--mirror | --shm] [--bounds 0,a,b,...,H | --cost-strips]] [--threads N]
 *
 * --example 4: BASELINE config #1 — the `kernelMain` of examples/04_ao/04_ao.cu:31-88 as a host C++ loop over the
 * image rows (cedec_2024_rt_amd/csrc/host_path.h: brute-force closest hit, 64 ambient-occlusion rays per pixel,
 * host libm), on --threads host threads (default: all). No GPU call is made: it runs on a machine without one.
 * Defaults then follow 04_ao.cpp (256x256 is the BASELINE size; camera (8,8,8) -> (0,0,0), common/misc.hpp:217-218);
 * --rgba writes the W*H RGBA8 bytes in the reference's storage order (what its pixel buffer holds).
 *
 * --ranks N: the multi-GPU frame loop: N processes, forked before anything touches a GPU,
 * rank r on device r, each rendering one row strip through the native strip driver (rt_mg_*: sparse
 * reservoir halos over RCCL send/recv with rank +-1). Strip heights: equal rows by default; --bounds gives the
 * edges explicitly (e.g. a cut measured by tools/strip_overhead.py, profiles/strip_cuts.json); --cost-strips weights
 * rows by their shaded pixels (round 2's model; measured worse than equal rows at 3840x2160, docs/MEASUREMENT_LOG_r01_r03.md). --pfm then receives every rank's rows (one file, written in
 * place). --mirror: all ranks on device 0 with the MIRROR transport (1-GPU boxes; timing/launch smoke
 * run, the image is not a frame). --shm: all ranks on device 0 with the host-staged shared-memory transport
 * (exact image, slow).
 * --example 7|8|9 runs the `path_trace` kernel of examples/07_pt, 08_nee or 09_ris instead of the
 * ReSTIR DI frame (one sample per pixel and frame; use --accumulate 1 to average frames).
 * Keys 1,2,3,4,A of the example (10_restir_di.cpp:143-174) are the --temporal/--spatial/
 * --shadowed/--visreuse/--accumulate flags; key S (screenshot, an RGB PNG written top row first,
 * common/misc.hpp:226-245) is --png (or --ppm). `--tris` reads a raw
 * array of 60-byte Triangle records (cedec_2024_rt_amd.scenes can write one); `--obj` uses the
 * OBJ/MTL subset the reference's loader consumes (common/loader.hpp:11-66: positions, faces as
 * triangle fans, per-face usemtl -> Kd/Ke).
 */
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../include/restir_rt.h"
#include "../include/restir_rt_internal.h" /* --mirror, --shm, --cost-strips, rt_path_trace_rays */
#include "../cedec_2024_rt_amd/csrc/host_path.h"

/* RGB8 PNG, top row first, as saveScreenshot writes it (common/misc.hpp:226-245 via stbi_write_png);
 * the deflate stream uses stored blocks only (no compression: no zlib dependency). */
static void write_png(const char* path, int W, int H, const uint8_t* rgb /* W*H*3, top-down */)
{
    static uint32_t crc_table[256];
    if (!crc_table[1])
        for (uint32_t n = 0; n < 256; ++n)
        {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
    auto crc = [&](const std::vector<uint8_t>& d, size_t from) {
        uint32_t c = 0xffffffffu;
        for (size_t i = from; i < d.size(); ++i) c = crc_table[(c ^ d[i]) & 0xffu] ^ (c >> 8);
        return c ^ 0xffffffffu;
    };
    auto be32 = [](std::vector<uint8_t>& d, uint32_t v) { for (int s = 24; s >= 0; s -= 8) d.push_back((uint8_t)(v >> s)); };
    std::vector<uint8_t> raw; /* filter byte 0 + RGB per scanline */
    raw.reserve((size_t)H * (1 + 3 * (size_t)W));
    for (int y = 0; y < H; ++y)
    {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * W * 3, rgb + (size_t)(y + 1) * W * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0; /* adler32 */
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535)
    {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        for (size_t i = off; i < off + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        if (raw.empty()) break;
    }
    be32(z, (b << 16) | a);
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    auto chunk = [&](const char* type, const std::vector<uint8_t>& payload) {
        std::vector<uint8_t> d;
        be32(d, (uint32_t)payload.size());
        d.insert(d.end(), type, type + 4);
        d.insert(d.end(), payload.begin(), payload.end());
        be32(d, crc(d, 4));
        fwrite(d.data(), 1, d.size(), f);
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)W); be32(ihdr, (uint32_t)H);
    const uint8_t tail[5] = {8, 2, 0, 0, 0}; /* 8 bits, colour type 2 (RGB), deflate, no filter, no interlace */
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    fclose(f);
}

static void die(rt_ctx* c, const char* what, int rc)
{
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, c ? rt_last_error(c) : "");
    exit(1);
}
#define CK(call)                                  \
    do                                            \
    {                                             \
        int _rc = (call);                         \
        if (_rc != RT_OK) die(ctx, #call, _rc);   \
    } while (0)

struct Mtl { float kd[3] = {0, 0, 0}, ke[3] = {0, 0, 0}; };

static std::map<std::string, Mtl> load_mtl(const std::string& path)
{
    std::map<std::string, Mtl> m;
    std::ifstream f(path);
    std::string line, cur;
    while (std::getline(f, line))
    {
        std::istringstream ss(line);
        std::string k;
        ss >> k;
        if (k == "newmtl") { ss >> cur; m[cur] = Mtl(); }
        else if (k == "Kd" && !cur.empty()) ss >> m[cur].kd[0] >> m[cur].kd[1] >> m[cur].kd[2];
        else if (k == "Ke" && !cur.empty()) ss >> m[cur].ke[0] >> m[cur].ke[1] >> m[cur].ke[2];
    }
    return m;
}

static std::vector<rt_triangle> load_obj(const std::string& path)
{
    std::vector<rt_triangle> out;
    std::ifstream f(path);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
    const std::string dir = path.find('/') == std::string::npos ? "." : path.substr(0, path.rfind('/'));
    std::vector<float> v;
    std::map<std::string, Mtl> mats;
    Mtl cur;
    std::string line;
    while (std::getline(f, line))
    {
        std::istringstream ss(line);
        std::string k;
        ss >> k;
        if (k == "v") { float x, y, z; ss >> x >> y >> z; v.push_back(x); v.push_back(y); v.push_back(z); }
        else if (k == "mtllib") { std::string n; ss >> n; auto m = load_mtl(dir + "/" + n); mats.insert(m.begin(), m.end()); }
        else if (k == "usemtl") { std::string n; ss >> n; cur = mats.count(n) ? mats[n] : Mtl(); }
        else if (k == "f")
        {
            std::vector<int> idx;
            std::string w;
            while (ss >> w)
            {
                const int i = atoi(w.substr(0, w.find('/')).c_str());
                idx.push_back(i > 0 ? i - 1 : (int)(v.size() / 3) + i);
            }
            for (size_t t = 2; t < idx.size(); ++t) /* triangle fan, tiny_obj_loader.h:908-931 */
            {
                rt_triangle tri;
                const int id[3] = {idx[0], idx[t - 1], idx[t]};
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) tri.v[a][c] = v[3 * (size_t)id[a] + c];
                memcpy(tri.color, cur.kd, 12);
                memcpy(tri.emissive, cur.ke, 12);
                out.push_back(tri);
            }
        }
    }
    return out;
}

static std::vector<rt_triangle> load_tris(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
    const size_t bytes = (size_t)f.tellg();
    std::vector<rt_triangle> t(bytes / sizeof(rt_triangle));
    f.seekg(0);
    f.read((char*)t.data(), (std::streamsize)(t.size() * sizeof(rt_triangle)));
    return t;
}

/* ---- --ranks N: one process per GPU over the native strip driver ---- */
struct Shared /* anonymous shared mapping made by the parent before fork */
{
    std::atomic<int> arrived[8];
    char uid[128];
    std::atomic<int> uid_ready;
    double ms_per_frame[64];
    unsigned long long rays[64];
    uint32_t row_cost[1]; /* H entries */
};
static void shared_barrier(Shared* sh, int which, int ranks)
{
    sh->arrived[which].fetch_add(1);
    for (long spins = 0; sh->arrived[which].load() < ranks; ++spins)
    {
        if (spins > 1500000L) { fprintf(stderr, "a rank did not reach barrier %d within 300 s\n", which); _exit(3); }
        usleep(200);
    }
}
static int rank_main(int rank, int ranks, bool mirror, bool shm, bool equal_strips, const std::vector<int>& given_bounds, Shared* sh,
                     const std::vector<rt_triangle>& triangles, int W, int H, int frames, const float* eye, const float* lookat, const rt_options& opt,
                     const std::string& pfm)
{
    const float up[3] = {0, 1, 0};
    const int halo = 87, device = (mirror || shm) ? 0 : rank;
    if (rank == 0 && shm) snprintf(sh->uid, sizeof(sh->uid), "rtmg_app_%d_%llx", (int)getppid(),
                                  (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count()); /* per-run nonce */
    if (rank == 0 && !mirror && !shm)
    {
        if (rt_mg_unique_id(sh->uid) != RT_OK) { fprintf(stderr, "rank 0: %s\n", rt_mg_load_error()); return 1; }
    }
    if (rank == 0) sh->uid_ready.store(1);
    while (!sh->uid_ready.load()) usleep(200);
    std::vector<int> bounds((size_t)ranks + 1);
    if (rt_mg_partition(H, ranks, halo, nullptr, bounds.data()) != RT_OK) { fprintf(stderr, "%d rows cannot be cut into %d strips of >= %d rows\n", H, ranks, halo); return 1; }
    if (!given_bounds.empty())
    {
        bool ok = (int)given_bounds.size() == ranks + 1 && given_bounds.front() == 0 && given_bounds.back() == H;
        for (int r = 0; ok && r < ranks; ++r) ok = given_bounds[(size_t)r + 1] - given_bounds[(size_t)r] >= halo;
        if (!ok) { fprintf(stderr, "--bounds: %d + 1 edges from 0 to %d, strips of >= %d rows\n", ranks, H, halo); return 1; }
        bounds = given_bounds;
    }
    rt_ctx* ctx = nullptr;
    auto make_ctx = [&]() -> int {
        int rc = rt_create(device, W, H, bounds[(size_t)rank], bounds[(size_t)rank + 1], halo, &ctx);
        if (rc != RT_OK) return rc;
        rc = rt_scene_set(ctx, triangles.data(), (uint32_t)triangles.size());
        if (rc == RT_OK) rc = rt_camera_lookat(ctx, eye, lookat, up, 3.14159265358979323846f / 4.0f);
        if (rc == RT_OK) rc = rt_options_set(ctx, &opt);
        return rc;
    };
    int rc = make_ctx();
    if (rc != RT_OK) die(ctx, "strip context", rc);
    if (!equal_strips && given_bounds.empty())
    {
        /* cost-weighted heights: shaded pixels per row from one raycast of the equal partition */
        CK(rt_raycast(ctx));
        CK(rt_row_shaded(ctx, sh->row_cost + bounds[(size_t)rank]));
        shared_barrier(sh, 0, ranks);
        std::vector<uint32_t> cost((size_t)H);
        for (int i = 0; i < H; ++i) cost[(size_t)i] = sh->row_cost[i] * 7u + (uint32_t)W;
        std::vector<int> nb((size_t)ranks + 1);
        if (rt_mg_partition(H, ranks, halo, cost.data(), nb.data()) == RT_OK && nb != bounds)
        {
            CK(rt_destroy(ctx));
            bounds = nb;
            rc = make_ctx();
            if (rc != RT_OK) die(ctx, "strip context", rc);
        }
    }
    rt_mg* mg = nullptr;
    rc = rt_mg_create(ctx, rank, ranks, bounds.data(), shm ? (int)RT_MG_TRANSPORT_SHM : (mirror ? (int)RT_MG_TRANSPORT_MIRROR : (int)RT_MG_TRANSPORT_RCCL),
                      mirror ? nullptr : sh->uid, 0, &mg);
    if (rc != RT_OK) { fprintf(stderr, "rank %d: rt_mg_create failed (%d): %s\n", rank, rc, mg ? rt_mg_last_error(mg) : ""); return 1; }
    CK(rt_clear(ctx));
    const int warm = frames > 4 ? 2 : 0;
    struct timespec t0, t1;
    for (int frame = 1; frame <= frames; ++frame)
    {
        if (frame == warm + 1) { CK(rt_sync(ctx)); shared_barrier(sh, 1, ranks); clock_gettime(CLOCK_MONOTONIC, &t0); }
        rc = rt_mg_frame(mg, frame, 0);
        if (rc != RT_OK) { fprintf(stderr, "rank %d frame %d: %s\n", rank, frame, rt_mg_last_error(mg)); return 1; }
    }
    CK(rt_sync(ctx));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    sh->ms_per_frame[rank] = ((double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6) / (double)(frames - warm);
    uint64_t rays = 0, shaded = 0;
    CK(rt_ray_count(ctx, &rays, &shaded));
    sh->rays[rank] = rays;
    rt_mg_stats st;
    rt_mg_get_stats(mg, &st);
    printf("rank %d: rows [%d,%d) %.3f ms/frame, host %.0f us/frame, %llu cold frame(s), %.2f MB sent/frame\n", rank, bounds[(size_t)rank],
           bounds[(size_t)rank + 1], sh->ms_per_frame[rank], (double)st.host_ns / 1e3 / (double)st.frames, st.cold_frames,
           (double)st.bytes_sent / 1e6 / (double)st.frames);
    if (!pfm.empty())
    {
        int l0 = 0, ln = 0;
        CK(rt_local_rows(ctx, &l0, &ln));
        std::vector<float> acc((size_t)W * (size_t)ln * 4);
        CK(rt_download(ctx, RT_BUF_ACCUMULATION, acc.data(), acc.size() * 4));
        char hdr[64];
        const int hl = snprintf(hdr, sizeof(hdr), "PF\n%d %d\n-1.0\n", W, H);
        const int fd = open(pfm.c_str(), O_WRONLY);
        if (fd < 0) { fprintf(stderr, "cannot write %s\n", pfm.c_str()); return 1; }
        std::vector<float> rgb((size_t)W * 3);
        for (int row = bounds[(size_t)rank]; row < bounds[(size_t)rank + 1]; ++row)
        {
            const float* a = &acc[(size_t)(row - l0) * W * 4];
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) rgb[3 * (size_t)x + c] = a[4 * (size_t)x + c] / a[4 * (size_t)x + 3];
            if (pwrite(fd, rgb.data(), rgb.size() * 4, (off_t)hl + (off_t)row * W * 12) < 0) return 1;
        }
        close(fd);
    }
    shared_barrier(sh, 2, ranks);
    if (rank == 0)
    {
        double worst = 0.0;
        unsigned long long total = 0;
        for (int r = 0; r < ranks; ++r) { worst = sh->ms_per_frame[r] > worst ? sh->ms_per_frame[r] : worst; total += sh->rays[r]; }
        printf("%d ranks: %.3f ms/frame (slowest rank), %llu rays/frame, %.0f Mray/s\n", ranks, worst, total, (double)total / worst / 1e3);
    }
    rt_mg_destroy(mg);
    CK(rt_destroy(ctx));
    return 0;
}

int main(int argc, char** argv)
{
    setenv("GPU_MAX_HW_QUEUES", "8", 0); /* five streams side by side per rank: before the first HIP call */
    int W = 1920, H = 1080, frames = 8; /* 10_restir_di.cpp:26-27 */
    /* camera "blocks_restir.obj 1", 10_restir_di.cpp:188-189 */
    float eye[3] = {-0.579885f, 22.194597f, -6.567105f}, lookat[3] = {5.224952f, 20.847435f, 1.431192f};
    const float up[3] = {0, 1, 0};
    std::string obj, tris_path, ppm, png, pfm, dump, rgba;
    bool by_kernel = false, mirror = false, shm = false, equal_strips = true, size_set = false, cam_set = false;
    std::vector<int> given_bounds;
    int example = 10, ranks = 1, threads = 0;
    rt_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.max_depth = 6; opt.ris_sample_count = 32; opt.rejection_heuristics_threshold = 0.2f;
    opt.spatial_resampling_sample_count = 5; opt.spatial_resampling_radius = 30.0f;
    opt.spatial_resampling_passes = 3; opt.use_visibility_reuse = 1; /* common/options.hpp:6-22 */
    opt.use_temporal_resampling = 1; opt.use_spatial_resampling = 1;  /* keys 1, 2 */
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto f = [&](int k) { return (float)atof(argv[i + k]); };
        if (a == "--obj") obj = argv[++i];
        else if (a == "--tris") tris_path = argv[++i];
        else if (a == "--size") { W = atoi(argv[i + 1]); H = atoi(argv[i + 2]); i += 2; size_set = true; }
        else if (a == "--frames") frames = atoi(argv[++i]);
        else if (a == "--eye") { eye[0] = f(1); eye[1] = f(2); eye[2] = f(3); i += 3; cam_set = true; }
        else if (a == "--lookat") { lookat[0] = f(1); lookat[1] = f(2); lookat[2] = f(3); i += 3; cam_set = true; }
        else if (a == "--temporal") opt.use_temporal_resampling = (uint8_t)atoi(argv[++i]);
        else if (a == "--spatial") opt.use_spatial_resampling = (uint8_t)atoi(argv[++i]);
        else if (a == "--shadowed") opt.use_shadowed_target_function = (uint8_t)atoi(argv[++i]);
        else if (a == "--visreuse") opt.use_visibility_reuse = (uint8_t)atoi(argv[++i]);
        else if (a == "--accumulate") opt.accumulate = (uint8_t)atoi(argv[++i]);
        else if (a == "--by-kernel") by_kernel = true;
        else if (a == "--example") example = atoi(argv[++i]);
        else if (a == "--ranks") ranks = atoi(argv[++i]);
        else if (a == "--mirror") mirror = true;
        else if (a == "--shm") shm = true;
        else if (a == "--equal-strips") equal_strips = true;
        else if (a == "--cost-strips") equal_strips = false;
        else if (a == "--bounds")
        {
            std::stringstream ss(argv[++i]);
            std::string tok;
            while (std::getline(ss, tok, ',')) given_bounds.push_back(atoi(tok.c_str()));
        }
        else if (a == "--dump-tris") dump = argv[++i]; /* write the loaded triangle array and exit (no GPU needed) */
        else if (a == "--ppm") ppm = argv[++i];
        else if (a == "--png") png = argv[++i];
        else if (a == "--pfm") pfm = argv[++i];
        else if (a == "--rgba") rgba = argv[++i];
        else if (a == "--threads") threads = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    std::vector<rt_triangle> triangles = !obj.empty() ? load_obj(obj) : load_tris(tris_path);
    if (triangles.empty()) { fprintf(stderr, "no triangles (use --obj or --tris)\n"); return 2; }
    if (!dump.empty())
    {
        FILE* f = fopen(dump.c_str(), "wb");
        fwrite(triangles.data(), sizeof(rt_triangle), triangles.size(), f);
        fclose(f);
        printf("triangles: %zu\n", triangles.size());
        return 0;
    }

    if (example == 4)
    {
        /* BASELINE config #1: 04_ao as a host loop (host_path.h); no context, no GPU */
        if (!size_set) { W = 256; H = 256; }
        if (!cam_set) { eye[0] = eye[1] = eye[2] = 8.0f; lookat[0] = lookat[1] = lookat[2] = 0.0f; } /* common/misc.hpp:217-218 */
        if (ranks > 1) { fprintf(stderr, "--example 4 is a host loop: --threads, not --ranks\n"); return 2; }
        rt_raygen rg;
        rt_host::raygen_lookat(&rg, eye, lookat, up, 3.14159265358979323846f / 4.0f, W, H); /* 04_ao.cpp:124-135 */
        std::vector<uint8_t> px((size_t)W * H * 4);
        const auto t0 = std::chrono::steady_clock::now();
        rt_host::ao04_image(triangles.data(), (uint32_t)triangles.size(), rg, W, H, threads, px.data());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const unsigned hw = std::thread::hardware_concurrency();
        size_t hit = 0;
        for (size_t i = 0; i < (size_t)W * H; ++i) hit += px[4 * i] != 32 || px[4 * i + 1] != 32;
        printf("triangles: %zu\n04_ao %dx%d: %.1f ms on %d host thread(s), %zu of %zu pixels hit\n", triangles.size(), W, H, ms,
               threads > 0 ? threads : (int)(hw ? hw : 1), hit, (size_t)W * H);
        if (!rgba.empty())
        {
            FILE* f = fopen(rgba.c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", rgba.c_str()); return 1; }
            fwrite(px.data(), 1, px.size(), f);
            fclose(f);
        }
        if (!ppm.empty())
        {
            FILE* f = fopen(ppm.c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", ppm.c_str()); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", W, H);
            for (int y = H - 1; y >= 0; --y)
                for (int x = 0; x < W; ++x) fwrite(&px[4 * ((size_t)y * W + x)], 1, 3, f);
            fclose(f);
        }
        if (!png.empty())
        {
            std::vector<uint8_t> rgb((size_t)W * H * 3);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) memcpy(&rgb[3 * ((size_t)y * W + x)], &px[4 * ((size_t)(H - 1 - y) * W + x)], 3);
            write_png(png.c_str(), W, H, rgb.data());
        }
        return 0;
    }
    if (example != 10 && example != 7 && example != 8 && example != 9) { fprintf(stderr, "--example 10, 7, 8, 9 or 4\n"); return 2; }

    if (ranks > 1)
    {
        /* one process per GPU, forked BEFORE any HIP call (a process that has initialised the GPU must not fork) */
        if (ranks > 64 || example != 10) { fprintf(stderr, "--ranks: 2..64 ranks of the ReSTIR DI frame\n"); return 2; }
        const size_t bytes = sizeof(Shared) + (size_t)H * 4;
        Shared* sh = (Shared*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (sh == MAP_FAILED) { perror("mmap"); return 1; }
        memset((void*)sh, 0, bytes);
        if (!pfm.empty())
        {
            FILE* f = fopen(pfm.c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", pfm.c_str()); return 1; }
            const int hl = fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
            fclose(f);
            if (truncate(pfm.c_str(), (off_t)hl + (off_t)W * H * 12) != 0) { perror("truncate"); return 1; }
        }
        std::vector<pid_t> kids;
        for (int r = 0; r < ranks; ++r)
        {
            const pid_t pid = fork();
            if (pid == 0)
            {
                const int code = rank_main(r, ranks, mirror, shm, equal_strips, given_bounds, sh, triangles, W, H, frames, eye, lookat, opt, pfm);
                fflush(stdout);
                fflush(stderr);
                _exit(code); /* no atexit handlers of the parent's image in a forked child */
            }
            kids.push_back(pid);
        }
        int worst = 0;
        for (pid_t k : kids)
        {
            int status = 0;
            waitpid(k, &status, 0);
            const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 1;
            worst = code > worst ? code : worst;
        }
        return worst;
    }

    rt_ctx* ctx = nullptr;
    int rc = rt_create(0, W, H, 0, H, 0, &ctx);
    if (rc != RT_OK) die(ctx, "rt_create", rc);
    CK(rt_scene_set(ctx, triangles.data(), (uint32_t)triangles.size()));
    uint32_t nt, nl, bh;
    CK(rt_scene_info(ctx, &nt, &nl, &bh));
    printf("triangles: %u\nlights: %u\nbvh height: %u\n", nt, nl, bh); /* cf. 10_restir_di.cpp:206 */
    CK(rt_camera_lookat(ctx, eye, lookat, up, 3.14159265358979323846f / 4.0f)); /* :242-251 */
    CK(rt_options_set(ctx, &opt));
    CK(rt_timing_enable(ctx, 1));
    CK(rt_clear(ctx)); /* :222-226 */

    for (int frame = 1; frame <= frames; ++frame) /* frame++ before the first launch, :233-234 */
    {
        if (example != 10)
        {
            /* examples/07_pt/07_pt.cpp:206-222: path_trace, then tone_mapping */
            CK(rt_path_trace(ctx, example, frame));
            CK(rt_tone_mapping(ctx));
        }
        else if (!by_kernel) { CK(rt_frame(ctx, frame, 0, nullptr)); }
        else
        {
            /* the launch sequence of 10_restir_di.cpp:270-379, one C-ABI call per kernel */
            CK(rt_raycast(ctx));
            CK(rt_generate_candidate(ctx, frame, RT_RES_0));
            CK(rt_temporal_resampling(ctx, frame, RT_RES_TEMPORAL, RT_RES_0));
            CK(rt_save_temporal_reservoir(ctx, RT_RES_0, RT_RES_TEMPORAL));
            int in = RT_RES_0, out = RT_RES_1;
            for (int k = 0; k < opt.spatial_resampling_passes; ++k)
            {
                if (k != 0) { const int t = in; in = out; out = t; }
                CK(rt_spatial_resampling(ctx, frame, k, in, out));
            }
            CK(rt_resolve(ctx, out));
            CK(rt_tone_mapping(ctx));
        }
        CK(rt_sync(ctx));
        if (!by_kernel && example == 10)
        {
            float ms[9];
            CK(rt_timing(ctx, ms));
            printf("frame %d kernel: %.3f ms (raycast %.3f, candidates %.3f, spatial %.3f+%.3f+%.3f, resolve %.3f)\n", frame,
                   ms[8], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6]); /* cf. the overlay of :410 */
        }
    }
    uint64_t rays = 0, shaded = 0;
    if (example == 10)
    {
        CK(rt_ray_count(ctx, &rays, &shaded));
        printf("rays/frame: %llu (shaded pixels %llu)\n", (unsigned long long)rays, (unsigned long long)shaded);
    }
    else
    {
        CK(rt_path_trace_rays(ctx, &rays));
        printf("rays in the last frame: %llu\n", (unsigned long long)rays);
    }

    if (!ppm.empty())
    {
        std::vector<uint8_t> px((size_t)W * H * 4);
        CK(rt_download(ctx, RT_BUF_PIXELS, px.data(), px.size()));
        FILE* f = fopen(ppm.c_str(), "wb");
        fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int y = H - 1; y >= 0; --y) /* storage is bottom-up (pixel_idx = x + (H-yi-1)*W) */
            for (int x = 0; x < W; ++x) fwrite(&px[4 * ((size_t)y * W + x)], 1, 3, f);
        fclose(f);
    }
    if (!png.empty())
    {
        std::vector<uint8_t> px((size_t)W * H * 4), rgb((size_t)W * H * 3);
        CK(rt_download(ctx, RT_BUF_PIXELS, px.data(), px.size()));
        for (int y = 0; y < H; ++y) /* storage is bottom-up */
            for (int x = 0; x < W; ++x)
                memcpy(&rgb[3 * ((size_t)y * W + x)], &px[4 * ((size_t)(H - 1 - y) * W + x)], 3);
        write_png(png.c_str(), W, H, rgb.data());
    }
    if (!pfm.empty())
    {
        std::vector<float> acc((size_t)W * H * 4);
        CK(rt_download(ctx, RT_BUF_ACCUMULATION, acc.data(), acc.size() * 4));
        FILE* f = fopen(pfm.c_str(), "wb");
        fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i)
        {
            const float w = acc[4 * i + 3];
            const float rgb[3] = {acc[4 * i] / w, acc[4 * i + 1] / w, acc[4 * i + 2] / w};
            fwrite(rgb, 4, 3, f);
        }
        fclose(f);
    }
    CK(rt_destroy(ctx));
    return 0;
}
