/*
 * strip_mg.cpp — native multi-GPU driver of the frame: one process (or, in tests, one context) per
 * GPU, the image cut into row strips, reservoir halos exchanged with rank +-1 between the spatial
 * passes. The reference is single-GPU (examples/10_restir_di/10_restir_di.cpp:35,
 * 231-383): this file is the frame loop of :257-379 for one strip, built on the C-ABI of
 * include/restir_rt.h + restir_rt_internal.h only (it is a client of rt_frame_stage_* / rt_halo_*), plus HIP streams/events
 * and RCCL point-to-point over xGMI.
 *
 * What it adds over the Python StripFrame of round 1 (cedec_2024_rt_amd/strips.py):
 *   - no host wait in a steady frame: the sparse-halo plan of frame f+1 (which neighbour records
 *     each rank will gather = a pure function of the RNG and the shaded flags) is marked during frame
 *     f on the second lane, its bitmaps ride along with frame f's last halo message, and the record
 *     counts (= the message sizes of frame f+1) reach the host by an asynchronous copy that frame f+1
 *     finds finished. A frame whose plan is not there (first frame, camera / option / scene change,
 *     non-consecutive frame number) builds it on the spot with one synchronisation ("cold" frame);
 *   - dense halos are sent from and received into the reservoir buffers themselves (no staging);
 *   - RCCL is called directly (ncclSend/ncclRecv grouped per exchange; on the main stream whenever that stream would
 *     only wait for the exchange anyway, see comm_on_main; on a communication stream otherwise),
 *     loaded with dlopen so that librestir_rt.so has no link-time dependency on it;
 *   - cost-weighted strip heights (rt_mg_partition) and the boundary/interior row bands.
 *
 * Transports: RCCL (product) and LOCAL (several contexts of ONE process on one GPU, driven in
 * lock-step by rt_mg_frame_step; device-to-device copies ordered by events) — the latter exists
 * because the development boxes have one GPU and RCCL refuses two ranks on one device.
 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <atomic>
#include <cstdlib>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include "../../include/restir_rt_internal.h"

/* ------------------------------------------------------------------ RCCL, resolved at run time */
namespace
{
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId; /* NCCL_UNIQUE_ID_BYTES, rccl.h:40-43 */
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 }; /* rccl.h:459-460 */

struct Rccl
{
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr; /* optional (NCCL >= 2.13): the library's own text for the last failure */
    std::string err;
    /* "<generic error string>[; RCCL says: <ncclGetLastError>]" for a failed call */
    std::string describe(int rc, ncclComm_t comm) const
    {
        std::string s = GetErrorString ? GetErrorString(rc) : "?";
        const char* last = GetLastError ? GetLastError(comm) : nullptr;
        if (last && *last) { s += "; RCCL says: "; s += last; }
        return s;
    }

    bool load()
    {
        if (lib) return true;
        /* torch ships its own librccl.so (same SONAME): if it is already mapped the loader returns it */
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
#define RCCL_SYM(field, name)                                                       \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));                    \
    if (!field) { err = std::string("RCCL lacks ") + name; lib = nullptr; return false; }
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        RCCL_SYM(CommInitRank, "ncclCommInitRank")
        RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RCCL_SYM(Send, "ncclSend")
        RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
        GetLastError = reinterpret_cast<decltype(GetLastError)>(dlsym(lib, "ncclGetLastError"));
        return true;
    }
};
Rccl g_rccl;

/* ------------------------------------------------------------------ transports */
struct Part { const void* send; size_t send_bytes; void* recv; size_t recv_bytes; };
struct Exchange { int peer; std::vector<Part> parts; };

/* LOCAL transport: mailbox shared by the contexts of one process */
struct LocalMsg
{
    std::vector<std::pair<const void*, size_t>> parts;
    hipEvent_t ready = nullptr, consumed = nullptr;
    bool consumed_recorded = false;
    ~LocalMsg()
    {
        if (ready) hipEventDestroy(ready);
        if (consumed) hipEventDestroy(consumed);
    }
};
/* SHM transport: N processes of one node (e.g. N ranks sharing ONE GPU on a development box), host-staged through a
 * POSIX shared-memory segment. One mailbox per directed neighbour pair: the sender copies its parts device -> mailbox
 * and bumps `posted`, the receiver waits for it, copies mailbox -> device and bumps `consumed`. Blocking and slow by
 * design: it exists to run the real multi-process code paths (bench.py --gpus N, restir_app --ranks N) with exact
 * images where RCCL cannot (two ranks on one device). */
struct ShmMailbox
{
    std::atomic<unsigned long long> posted, consumed;
    unsigned long long bytes[8];
    unsigned int n_parts, pad;
};
struct ShmSegment
{
    void* base = nullptr;
    size_t bytes = 0, slot = 0;
    int world = 0;
    std::string name;
    /* mailbox of messages src -> dst (|src - dst| == 1): index 2*min + (src > dst) */
    ShmMailbox* box(int src, int dst) const
    {
        const int lo = src < dst ? src : dst;
        return reinterpret_cast<ShmMailbox*>((char*)base + (size_t)(2 * lo + (src > dst ? 1 : 0)) * slot);
    }
    char* data(int src, int dst) const { return (char*)box(src, dst) + 256; }
};
struct LocalHub
{
    int world = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<LocalMsg>>> box; /* (src, dst) -> FIFO */
};
}  // namespace

enum { HP_N_ = 7 };
struct HostProfile { bool on = false; unsigned long long ns[HP_N_] = {}, calls[HP_N_] = {}; };
struct rt_mg
{
    HostProfile hp;
    /* Halo plans live in three slots (slot = frame mod 3): while frame f runs with plan f, plan f+1 (marked during frame
     * f-1) rides on frame f's first halo message, and plan f+2 is being marked on the prep stream. Marking two frames
     * ahead (r03) takes the mark kernel off the critical path: since the next frame's stage 0 runs beside this frame's
     * passes (rt_tuning key 14), a frame starts with its first exchange, and a plan marked in the same frame would have
     * to be waited for right there. */
    static constexpr int NSLOT = 3;
    static int slot_of(long long frame) { return (int)(((frame % NSLOT) + NSLOT) % NSLOT); }
    rt_ctx* ctx = nullptr;
    int rank = 0, world = 1, W = 0, H = 0, halo = 0, a = 0, b = 0;
    std::vector<int> bounds;
    int transport = RT_MG_TRANSPORT_RCCL;
    bool sparse = true, two_lanes = true;
    /* RCCL: grouped send / recv on the MAIN stream instead of the communication stream. A cross-stream dependency costs
     * ~10 us on this GPU (tools/stream_hops.hip: 2.5 us per link on one stream, 13.5 across two), and an exchange on the
     * communication stream puts two of them on the frame's critical chain (pass -> send, recv -> next pass): 6 per frame.
     * The main stream has nothing else to do while an exchange is in flight whenever the interior rows run on the second
     * lane or there are none, so the exchange goes there then (the MIRROR transport, which the compute-side bounds are
     * measured with, always copied on the main stream). RT_MG_COMM_STREAM=1 in the environment keeps the separate stream. */
    bool comm_on_main = false, pending_on_main = false;
    bool interior_late = false; /* RT_MG_INTERIOR_LATE=1 */
    double wire_gbs = 153.0, wire_lat_us = 5.0; /* WIRE_MODEL: one xGMI link per neighbour (bytes per ns = GB/s), fixed latency */
    int wire_slot = 0;
    unsigned long long mirror_wire_ns = 0; /* MIRROR_WIRE: the exchange in flight */
    int mirror_wire_slot = 0;
    unsigned long long stats_wire_ns = 0; /* modelled wire time of all exchanges since rt_mg_reset_stats */
    bool counts_on_comm = false; /* RT_MG_COUNTS_ON_COMM=1 (A/B) */
    bool fuse_halos = true; /* sparse halos packed / unpacked by the spatial passes themselves (rt_halo_fuse_set, r03) */
    std::string err;

    /* neighbours: side 0 = the strip below (rank - 1, smaller rows), side 1 = the strip above */
    struct Side
    {
        int peer = -1, side = 0;
        int send_row0 = 0, recv_row0 = 0, n_rows = 0;
        size_t bm_words = 0;                                /* rt_halo_bitmap_words(n_rows): count, bits, prefix */
        uint32_t* need_bm[NSLOT] = {nullptr, nullptr, nullptr}; /* [plan slot] passes x bm_words: what I gather from the peer */
        uint32_t* give_bm[NSLOT] = {nullptr, nullptr, nullptr}; /* what the peer gathers from me */
        uint32_t* cnt_h[NSLOT] = {nullptr, nullptr, nullptr};   /* pinned host, [slot][0..P) need counts, [P..2P) give counts */
        char* send_buf[2] = {nullptr, nullptr};             /* sparse record lists, by exchange parity */
        char* recv_buf = nullptr;
        uint8_t *flags_send = nullptr, *flags_recv = nullptr;
        std::shared_ptr<LocalMsg> last_send[2], last_bm[NSLOT], last_flags; /* LOCAL: who may still be reading a buffer */
    };
    std::vector<Side> sides;
    /* one arena per plan slot: [need side A][need side B][give side A][give side B], each max_passes bitmaps of
     * bm_stride words, so that one memset clears the needs, one launch marks them and ONE strided copy brings every
     * count (word 0 of each bitmap) to cnt_all; Side::need_bm / give_bm / cnt_h point into these */
    uint32_t* bm_arena[NSLOT] = {nullptr, nullptr, nullptr};
    uint32_t* cnt_all[NSLOT] = {nullptr, nullptr, nullptr}; /* pinned host: [need A][need B][give A][give B] x max_passes */
    size_t bm_stride = 0;
    int bnd[2][2] = {{0, 0}, {0, 0}}, n_bnd = 0;   /* boundary row ranges (needed by a neighbour), computed first */
    int itr[2][2] = {{0, 0}, {0, 0}}, n_itr = 0;   /* interior row ranges, computed while halos travel */

    hipStream_t comm = nullptr, prep = nullptr; /* prep: the halo marks of the frames to come, beside this frame's passes */
    bool own_prep = false;
    hipEvent_t ev_packed = nullptr, ev_arrived = nullptr, ev_arrived2 = nullptr, ev_plan[NSLOT] = {nullptr, nullptr, nullptr}, ev_gbuf = nullptr,
               ev_marked[NSLOT] = {nullptr, nullptr, nullptr}, ev_carried = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr; /* GPU-side clock of the frame loop: start of the first / latest frame since rt_mg_reset_stats */
    unsigned long long frames_timed = 0;
    ncclComm_t nccl = nullptr;
    LocalHub* hub = nullptr;
    ShmSegment shm;

    /* cached halo plans, slot = frame mod 3: marked_* = need-bitmaps marked on the device (event ev_marked), plan_* =
     * bitmaps exchanged and counts on their way to the host (event ev_plan) */
    int plan_passes = 0, max_passes = 0;
    long long plan_frame[NSLOT] = {-1, -1, -1}, marked_frame[NSLOT] = {-1, -1, -1};
    uint64_t plan_epoch[NSLOT] = {0, 0, 0}, marked_epoch[NSLOT] = {0, 0, 0}, flags_epoch = 0;
    int marked_passes[NSLOT] = {0, 0, 0};

    /* frame state machine (rt_mg_frame_begin / _step) */
    enum Seg { SEG_IDLE, SEG_RAYCAST, SEG_COLD_FLAGS, SEG_COLD_MARK, SEG_COLD_BITMAPS, SEG_GENERATE, SEG_PASS, SEG_FINAL };
    Seg seg = SEG_IDLE;
    int frame = 0, clear_first = 0, stage = 0, passes = 0;
    bool warm = false, use_sparse = false, pending = false, pending_carries_plan = false;
    int pending_buf = 0, pending_k = 0;
    std::vector<Exchange> pending_x;
    std::vector<std::shared_ptr<LocalMsg>> pending_local;

    rt_mg_stats stats;
};

/* RT_MG_HOST_PROFILE=1: where the host time of rt_mg_frame_step goes, printed by rt_mg_destroy (stderr). Scoped timers around the
 * driver's calls into the C-ABI, HIP and RCCL; what is left is the driver's own bookkeeping. */
enum { HP_STAGE = 0, HP_LAUNCH, HP_RCCL, HP_MARK, HP_PACK, HP_EVENT, HP_COUNTS, HP_N };
static_assert(HP_N == HP_N_, "HostProfile size");
static const char* const HP_NAME[HP_N] = {"rt_frame_stage_begin/_end/_fork", "rt_frame_stage_run_* (kernel launches)", "RCCL group (send/recv)", "rt_halo_mark_sides (memset + 2-3 launches)",
                                          "pack / unpack / fuse_set / wire delay", "hipEventRecord / hipStreamWaitEvent", "counts copy + plan bookkeeping"};
struct HpScope
{
    HostProfile* p; int k; std::chrono::steady_clock::time_point t0;
    HpScope(HostProfile* p_, int k_) : p(p_ && p_->on ? p_ : nullptr), k(k_) { if (p) t0 = std::chrono::steady_clock::now(); }
    ~HpScope() { if (p) { p->ns[k] += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); p->calls[k] += 1; } }
};
#define HP(m, k) HpScope _hp(&(m)->hp, (k))
#define MG_FAIL(m, code, ...)                     \
    do                                            \
    {                                             \
        char _b[512];                             \
        snprintf(_b, sizeof(_b), __VA_ARGS__);    \
        (m)->err = _b;                            \
        return (code);                            \
    } while (0)
#define MG_HIP(m, call)                                                                             \
    do                                                                                              \
    {                                                                                               \
        hipError_t _e = (call);                                                                     \
        if (_e != hipSuccess) MG_FAIL(m, RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)
#define MG_RT(m, call)                                                                    \
    do                                                                                    \
    {                                                                                     \
        int _rc = (call);                                                                 \
        if (_rc != RT_OK) MG_FAIL(m, _rc, "%s: %s", #call, rt_last_error((m)->ctx));      \
    } while (0)
#define MG_NCCL(m, call)                                                                                   \
    do                                                                                                     \
    {                                                                                                      \
        int _r = (call);                                                                                   \
        if (_r != ncclSuccess) MG_FAIL(m, RT_ERR_COMM, "rank %d of %d: %s failed with code %d: %s", (m)->rank, (m)->world, #call, _r, g_rccl.describe(_r, (m)->nccl).c_str()); \
    } while (0)

static bool is_self(const rt_mg* m) { return m->transport == RT_MG_TRANSPORT_RCCL_SELF || m->transport == RT_MG_TRANSPORT_WIRE_MODEL; }
static bool is_rccl(const rt_mg* m) { return m->transport == RT_MG_TRANSPORT_RCCL || is_self(m); }
static hipStream_t main_stream(rt_mg* m)
{
    void* s = nullptr;
    rt_get_stream(m->ctx, &s);
    return (hipStream_t)s;
}

/* ------------------------------------------------------------------ partition / bands (pure host) */
extern "C" {

/* Contiguous strips of at least `halo` rows each (a halo must not reach beyond the adjacent strip).
 * row_cost == NULL: near-equal heights, the first H % world strips one row taller. Otherwise the
 * partition that minimises the largest strip cost (binary search on the bound, greedy fill). */
int rt_mg_partition(int height, int world, int halo, const uint32_t* row_cost, int* bounds)
{
    if (height <= 0 || world <= 0 || !bounds || halo < 0) return RT_ERR_ARG;
    if (world > 1 && height / world < halo) return RT_ERR_ARG;
    if (!row_cost || world == 1)
    {
        const int base = height / world, extra = height % world;
        bounds[0] = 0;
        for (int r = 0; r < world; ++r) bounds[r + 1] = bounds[r] + base + (r < extra ? 1 : 0);
        return RT_OK;
    }
    std::vector<unsigned long long> pre((size_t)height + 1, 0);
    for (int i = 0; i < height; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + (unsigned long long)row_cost[i] + 1ull; /* +1: every row costs something */
    const int min_rows = halo > 1 ? halo : 1;
    auto fill = [&](unsigned long long T, int* out) -> bool {
        int at = 0;
        out[0] = 0;
        for (int r = 0; r < world; ++r)
        {
            const int left = world - 1 - r;               /* strips still to place after this one */
            int lo = at + min_rows, hi = height - left * min_rows;
            if (lo > hi) return false;
            if (left == 0) { if (pre[(size_t)height] - pre[(size_t)at] > T) return false; out[r + 1] = height; return true; }
            /* the largest end <= hi with cost <= T, but at least lo rows */
            int end = lo;
            if (pre[(size_t)lo] - pre[(size_t)at] > T) return false;
            int l = lo, h = hi;
            while (l <= h)
            {
                const int mid = (l + h) / 2;
                if (pre[(size_t)mid] - pre[(size_t)at] <= T) { end = mid; l = mid + 1; }
                else h = mid - 1;
            }
            out[r + 1] = end;
            at = end;
        }
        return true;
    };
    unsigned long long lo = 0, hi = pre[(size_t)height];
    std::vector<int> tmp((size_t)world + 1);
    while (lo < hi)
    {
        const unsigned long long mid = lo + (hi - lo) / 2;
        if (fill(mid, tmp.data())) hi = mid; else lo = mid + 1;
    }
    if (!fill(lo, bounds)) return RT_ERR_ARG;
    return RT_OK;
}

/* Row bands of one strip: `boundary` = owned rows a neighbour's spatial pass can reach (within `halo`
 * rows of a strip edge that has a neighbour) — computed and sent first; `interior` = the rest,
 * computed while the halos travel. Each list holds up to two [row0,row1) pairs. */
int rt_mg_bands(const int* bounds, int world, int rank, int halo, int* boundary, int* n_boundary, int* interior, int* n_interior)
{
    if (!bounds || rank < 0 || rank >= world || !boundary || !n_boundary || !interior || !n_interior) return RT_ERR_ARG;
    const int a = bounds[rank], b = bounds[rank + 1];
    int nb = 0, ni = 0;
    int c[2][2];
    int nc = 0;
    if (rank > 0) { c[nc][0] = a; c[nc][1] = a + halo < b ? a + halo : b; ++nc; }
    if (rank + 1 < world) { c[nc][0] = b - halo > a ? b - halo : a; c[nc][1] = b; ++nc; }
    if (nc == 2 && c[1][0] <= c[0][1]) { c[0][1] = c[1][1] > c[0][1] ? c[1][1] : c[0][1]; nc = 1; }
    int cur = a;
    for (int i = 0; i < nc; ++i)
    {
        boundary[2 * nb] = c[i][0]; boundary[2 * nb + 1] = c[i][1]; ++nb;
        if (cur < c[i][0]) { interior[2 * ni] = cur; interior[2 * ni + 1] = c[i][0]; ++ni; }
        cur = c[i][1];
    }
    if (cur < b) { interior[2 * ni] = cur; interior[2 * ni + 1] = b; ++ni; }
    *n_boundary = nb;
    *n_interior = ni;
    return RT_OK;
}

/* ------------------------------------------------------------------ creation */
int rt_mg_unique_id(void* id128)
{
    if (!id128) return RT_ERR_ARG;
    if (!g_rccl.load()) return RT_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return RT_ERR_COMM;
    memcpy(id128, &id, sizeof(id));
    return RT_OK;
}
const char* rt_mg_load_error(void) { return g_rccl.err.c_str(); }

int rt_mg_hub_create(int world, void** hub)
{
    if (!hub || world <= 0) return RT_ERR_ARG;
    LocalHub* h = new LocalHub();
    h->world = world;
    *hub = h;
    return RT_OK;
}
int rt_mg_hub_destroy(void* hub)
{
    delete (LocalHub*)hub;
    return RT_OK;
}

const char* rt_mg_last_error(rt_mg* m) { return m ? m->err.c_str() : "null rt_mg"; }

static int alloc_sides(rt_mg* m)
{
    const size_t ns = m->sides.size();
    if (ns == 0) return RT_OK;
    for (auto& s : m->sides)
    {
        s.bm_words = rt_halo_bitmap_words(m->ctx, s.n_rows);
        if (s.bm_words > m->bm_stride) m->bm_stride = s.bm_words;
    }
    for (auto& s : m->sides)
        if (s.bm_words != m->bm_stride) MG_FAIL(m, RT_ERR_STATE, "halo regions of different heights on the two sides");
    const size_t per = m->bm_stride * (size_t)m->max_passes; /* words of one side's need (or give) bitmaps */
    for (int k = 0; k < rt_mg::NSLOT; ++k)
    {
        MG_HIP(m, hipMalloc(&m->bm_arena[k], per * 2 * ns * 4));
        MG_HIP(m, hipMemset(m->bm_arena[k], 0, per * 2 * ns * 4));
        MG_HIP(m, hipHostMalloc(&m->cnt_all[k], (size_t)m->max_passes * 2 * ns * 4, hipHostMallocDefault));
        memset(m->cnt_all[k], 0, (size_t)m->max_passes * 2 * ns * 4);
    }
    for (size_t i = 0; i < ns; ++i)
    {
        auto& s = m->sides[i];
        const size_t list_bytes = rt_halo_bytes(m->ctx, s.n_rows) + 256;
        for (int k = 0; k < rt_mg::NSLOT; ++k)
        {
            s.need_bm[k] = m->bm_arena[k] + per * i;
            s.give_bm[k] = m->bm_arena[k] + per * (ns + i);
            s.cnt_h[k] = m->cnt_all[k] + (size_t)m->max_passes * i; /* give counts: + max_passes * ns */
        }
        for (int k = 0; k < 2; ++k) MG_HIP(m, hipMalloc(&s.send_buf[k], list_bytes));
        MG_HIP(m, hipMalloc(&s.recv_buf, list_bytes));
        MG_HIP(m, hipMalloc(&s.flags_send, rt_halo_flags_bytes(m->ctx, s.n_rows) + 16));
        MG_HIP(m, hipMalloc(&s.flags_recv, rt_halo_flags_bytes(m->ctx, s.n_rows) + 16));
    }
    return RT_OK;
}

/* A stalled exchange must not hang the caller for ever: host waits on events of the exchange chain poll with a
 * deadline (RT_MG_TIMEOUT_S seconds, default 30) and fail with the rank, the peer(s) and the step that stalled. The
 * process cannot recover from that (the stream is stuck behind the collective): the caller reports and exits. */
static double mg_timeout_s()
{
    const char* e = getenv("RT_MG_TIMEOUT_S");
    const double t = e ? atof(e) : 0.0;
    return t > 0.0 ? t : 30.0;
}
static int wait_event_deadline(rt_mg* m, hipEvent_t ev, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = mg_timeout_s();
    for (long spins = 0;; ++spins)
    {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return RT_OK;
        if (e != hipErrorNotReady) MG_FAIL(m, RT_ERR_HIP, "hipEventQuery failed while waiting for %s: %s", what, hipGetErrorString(e));
        if (spins > 2000) usleep(spins > 20000 ? 200 : 20);
        if ((spins & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            MG_FAIL(m, RT_ERR_COMM, "rank %d of %d (rows %d..%d): %s did not complete within %.0f s — the exchange with rank %d%s stalled (frame %d, stage %d)",
                    m->rank, m->world, m->a, m->b, what, limit, m->sides.empty() ? -1 : m->sides[0].peer,
                    m->sides.size() > 1 ? (std::string(" / ") + std::to_string(m->sides[1].peer)).c_str() : "", m->frame, m->stage);
    }
}

/* first contact with the neighbours over RCCL: one grouped 16-byte send/recv with rank +-1 on the communication stream,
 * checked on the host with a deadline. Proves, before any frame is enqueued, that the communicator, the grouped
 * point-to-point path and the stream ordering work between THESE ranks; each rank also learns its neighbours' view of
 * the partition (the row they believe the shared edge is at) and refuses a mismatch. */
static int rccl_handshake(rt_mg* m)
{
    if (m->sides.empty()) return RT_OK;
    /* host buffers are PINNED: a device -> pageable-host copy completes synchronously on the host, so a stalled first
     * exchange would block inside hipMemcpyAsync and never reach the deadline below; on a timeout both the
     * device and the host buffer are leaked on purpose — the stuck stream may still write them */
    int* d = nullptr;
    int (*h)[2][4] = nullptr; /* h[0] = what I send, h[1] = what arrived */
    MG_HIP(m, hipMalloc(&d, 2 * 2 * 16));
    MG_HIP(m, hipHostMalloc((void**)&h, 2 * sizeof(*h), hipHostMallocDefault));
    memset(h, 0, 2 * sizeof(*h));
    int (&h_send)[2][4] = h[0];
    int (&h_recv)[2][4] = h[1];
    for (size_t i = 0; i < m->sides.size(); ++i)
    {
        const auto& s = m->sides[i];
        h_send[i][0] = 0x52544d47; /* "RTMG" */
        h_send[i][1] = m->rank; h_send[i][2] = s.side == 0 ? m->a : m->b; /* the edge shared with this peer */
        h_send[i][3] = m->halo;
    }
    MG_HIP(m, hipMemcpyAsync(d, h_send, sizeof(h_send), hipMemcpyHostToDevice, m->comm));
    MG_HIP(m, hipMemsetAsync(d + 8, 0, 32, m->comm));
    MG_NCCL(m, g_rccl.GroupStart());
    for (size_t i = 0; i < m->sides.size(); ++i)
    {
        MG_NCCL(m, g_rccl.Send(d + 4 * i, 16, ncclUint8, m->sides[i].peer, m->nccl, m->comm));
        MG_NCCL(m, g_rccl.Recv(d + 8 + 4 * i, 16, ncclUint8, m->sides[i].peer, m->nccl, m->comm));
    }
    MG_NCCL(m, g_rccl.GroupEnd());
    MG_HIP(m, hipMemcpyAsync(h_recv, d + 8, sizeof(h_recv), hipMemcpyDeviceToHost, m->comm));
    { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_arrived, m->comm)); }
    int rc = wait_event_deadline(m, m->ev_arrived, "the RCCL handshake (16-byte grouped send/recv with the neighbours)");
    if (rc != RT_OK) return rc; /* d and h are leaked on purpose: the stream may still own them */
    int got[2][4];
    memcpy(got, h_recv, sizeof(got));
    hipFree(d);
    hipHostFree(h);
    for (size_t i = 0; i < m->sides.size(); ++i)
    {
        const auto& s = m->sides[i];
        if (got[i][0] != 0x52544d47 || got[i][1] != s.peer || got[i][2] != (s.side == 0 ? m->a : m->b) || got[i][3] != m->halo)
            MG_FAIL(m, RT_ERR_COMM, "RCCL handshake: rank %d expected {rank %d, edge row %d, halo %d} from its neighbour, received {magic %08x, rank %d, edge row %d, halo %d}",
                    m->rank, s.peer, s.side == 0 ? m->a : m->b, m->halo, (unsigned)got[i][0], got[i][1], got[i][2], got[i][3]);
    }
    return RT_OK;
}

/* ctx: the strip context of this rank, created with rows bounds[rank]..bounds[rank+1] and a halo of at
 * least the reach of the spatial pass (87 rows for the default radius). `arg`: RCCL: the 128-byte unique
 * id of rt_mg_unique_id (made by one rank, distributed by the caller); LOCAL: the hub. */
int rt_mg_create(rt_ctx* ctx, int rank, int world, const int* bounds, int transport, const void* arg, int flags, rt_mg** out)
{
    if (!ctx || !out || !bounds || world <= 0 || rank < 0 || rank >= world) return RT_ERR_ARG;
    rt_mg* m = new rt_mg();
    *out = m;
    m->ctx = ctx; m->rank = rank; m->world = world; m->transport = transport;
    m->bounds.assign(bounds, bounds + world + 1);
    m->sparse = !(flags & RT_MG_DENSE);
    m->two_lanes = !(flags & RT_MG_ONE_LANE);
    {
        const char* e = getenv("RT_MG_COMM_STREAM");
        m->comm_on_main = !(e && e[0] == '1');
    }
    m->fuse_halos = !(flags & RT_MG_SEPARATE_PACK);
    { const char* e = getenv("RT_MG_COUNTS_ON_COMM"); m->counts_on_comm = e && e[0] == '1'; }
    {
        const char* e = getenv("RT_MG_INTERIOR_LATE");
        m->interior_late = e && e[0] == '1';
    }
    memset(&m->stats, 0, sizeof(m->stats));
    { const char* e = getenv("RT_MG_HOST_PROFILE"); m->hp.on = e && e[0] == '1'; }
    int ra = 0, rb = 0;
    MG_RT(m, rt_geometry(ctx, &m->W, &m->H, &ra, &rb, &m->halo));
    m->a = bounds[rank]; m->b = bounds[rank + 1];
    if (ra != m->a || rb != m->b) MG_FAIL(m, RT_ERR_ARG, "context owns rows [%d,%d), the partition gives rank %d [%d,%d)", ra, rb, rank, m->a, m->b);
    if (bounds[0] != 0 || bounds[world] != m->H) MG_FAIL(m, RT_ERR_ARG, "bounds must cover rows 0..%d", m->H);
    for (int r = 0; r < world; ++r)
        if (world > 1 && bounds[r + 1] - bounds[r] < m->halo)
            MG_FAIL(m, RT_ERR_ARG, "strip %d has %d rows, fewer than the %d-row halo", r, bounds[r + 1] - bounds[r], m->halo);
    if (world > 1 && m->halo <= 0) MG_FAIL(m, RT_ERR_ARG, "strip contexts of a multi-rank frame need halo rows");
    int bl[4], il[4];
    rt_mg_bands(bounds, world, rank, m->halo, bl, &m->n_bnd, il, &m->n_itr);
    for (int i = 0; i < m->n_bnd; ++i) { m->bnd[i][0] = bl[2 * i]; m->bnd[i][1] = bl[2 * i + 1]; }
    for (int i = 0; i < m->n_itr; ++i) { m->itr[i][0] = il[2 * i]; m->itr[i][1] = il[2 * i + 1]; }
    if (rank > 0)
    {
        rt_mg::Side s;
        s.peer = rank - 1; s.side = 0; s.n_rows = m->halo;
        s.send_row0 = m->a; s.recv_row0 = m->a - m->halo;
        m->sides.push_back(s);
    }
    if (rank + 1 < world)
    {
        rt_mg::Side s;
        s.peer = rank + 1; s.side = 1; s.n_rows = m->halo;
        s.send_row0 = m->b - m->halo; s.recv_row0 = m->b;
        m->sides.push_back(s);
    }
    m->max_passes = 8;
    int dev = 0;
    MG_HIP(m, hipGetDevice(&dev));
    MG_HIP(m, hipStreamCreateWithFlags(&m->comm, hipStreamNonBlocking));
    {
        /* the halo plans are marked on the context's tail stream (behind the previous frame's resolve + tone mapping, which
         * are short and long done when the marks are enqueued): HIP multiplexes streams onto few hardware queues, and every
         * stream this driver does not create is one chance less for the main stream and the pipelined stage 0 to share one */
        void* ts = nullptr;
        MG_RT(m, rt_side_stream(ctx, 0, &ts));
        m->prep = (hipStream_t)ts;
        if (!m->prep) { MG_HIP(m, hipStreamCreateWithFlags(&m->prep, hipStreamNonBlocking)); m->own_prep = true; }
    }
    MG_HIP(m, hipEventCreateWithFlags(&m->ev_gbuf, hipEventDisableTiming));
    for (auto& e : m->ev_marked) MG_HIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    MG_HIP(m, hipEventCreateWithFlags(&m->ev_packed, hipEventDisableTiming));
    MG_HIP(m, hipEventCreateWithFlags(&m->ev_arrived, hipEventDisableTiming));
    MG_HIP(m, hipEventCreateWithFlags(&m->ev_arrived2, hipEventDisableTiming));
    for (auto& e : m->ev_plan) MG_HIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    MG_HIP(m, hipEventCreateWithFlags(&m->ev_carried, hipEventDisableTiming));
    MG_HIP(m, hipEventCreate(&m->ev_t0));
    MG_HIP(m, hipEventCreate(&m->ev_t1));
    int rc = alloc_sides(m);
    if (rc != RT_OK) return rc;
    if (world > 1 && transport == RT_MG_TRANSPORT_RCCL)
    {
        if (!arg) MG_FAIL(m, RT_ERR_ARG, "RCCL transport needs the 128-byte unique id");
        if (!g_rccl.load()) MG_FAIL(m, RT_ERR_COMM, "%s", g_rccl.err.c_str());
        ncclUniqueId id;
        memcpy(&id, arg, sizeof(id));
        MG_NCCL(m, g_rccl.CommInitRank(&m->nccl, world, id, rank));
        rc = rccl_handshake(m);
        if (rc != RT_OK) return rc;
    }
    else if (world > 1 && transport == RT_MG_TRANSPORT_LOCAL)
    {
        if (!arg) MG_FAIL(m, RT_ERR_ARG, "LOCAL transport needs a hub (rt_mg_hub_create)");
        m->hub = (LocalHub*)arg;
        if (m->hub->world != world) MG_FAIL(m, RT_ERR_ARG, "hub was created for %d ranks", m->hub->world);
    }
    else if (world > 1 && transport == RT_MG_TRANSPORT_SHM)
    {
        if (!arg) MG_FAIL(m, RT_ERR_ARG, "SHM transport needs a segment name shared by all ranks");
        /* capacity of a mailbox: the largest message = a dense halo band + the bitmaps of a plan */
        size_t cap = rt_halo_bytes(ctx, m->halo) + 256;
        for (auto& sd : m->sides) cap = std::max(cap, rt_halo_bytes(ctx, sd.n_rows) + sd.bm_words * 4 * (size_t)m->max_passes + 4096);
        m->shm.slot = ((cap + 256 + 4095) / 4096) * 4096;
        m->shm.world = world;
        m->shm.bytes = m->shm.slot * 2 * (size_t)(world - 1);
        m->shm.name = std::string("/") + (const char*)arg;
        /* rank 0 creates the segment exclusively (a stale one of a crashed run with the same name is removed first, so
         * every mailbox starts zero-filled: posted = consumed = 0); the other ranks wait until it exists at its full size.
         * Callers put a per-run nonce in the name (bench.py, restir_app), so a rank never maps a predecessor's segment. */
        int fd = -1;
        if (rank == 0)
        {
            shm_unlink(m->shm.name.c_str());
            fd = shm_open(m->shm.name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) MG_FAIL(m, RT_ERR_COMM, "shm_open(%s, O_CREAT | O_EXCL) failed", m->shm.name.c_str());
            if (ftruncate(fd, (off_t)m->shm.bytes) != 0) { close(fd); MG_FAIL(m, RT_ERR_COMM, "ftruncate of the shared segment failed"); }
        }
        else
        {
            for (int tries = 0; tries < 60000 && fd < 0; ++tries) /* <= 60 s */
            {
                fd = shm_open(m->shm.name.c_str(), O_RDWR, 0600);
                struct stat sb;
                if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size != m->shm.bytes)) { close(fd); fd = -1; }
                if (fd < 0) usleep(1000);
            }
            if (fd < 0) MG_FAIL(m, RT_ERR_COMM, "rank %d: the shared segment %s never appeared (rank 0 creates it)", rank, m->shm.name.c_str());
        }
        m->shm.base = mmap(nullptr, m->shm.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m->shm.base == MAP_FAILED) { m->shm.base = nullptr; MG_FAIL(m, RT_ERR_COMM, "mmap of the shared segment failed"); }
    }
    else if (world > 1 && (transport == RT_MG_TRANSPORT_RCCL_SELF || transport == RT_MG_TRANSPORT_WIRE_MODEL))
    {
        if (transport == RT_MG_TRANSPORT_WIRE_MODEL)
        {
            const char* g = getenv("RT_MG_WIRE_GBS");
            const char* l = getenv("RT_MG_WIRE_LAT_US");
            if (g && atof(g) > 0.0) m->wire_gbs = atof(g);
            if (l && atof(l) >= 0.0) m->wire_lat_us = atof(l);
        }
        /* one rank alone with a communicator of its own: the peers of the partition do not exist, every message goes to self */
        if (!g_rccl.load()) MG_FAIL(m, RT_ERR_COMM, "%s", g_rccl.err.c_str());
        ncclUniqueId id;
        MG_NCCL(m, g_rccl.GetUniqueId(&id));
        MG_NCCL(m, g_rccl.CommInitRank(&m->nccl, 1, id, 0));
    }
    else if (world > 1 && transport == RT_MG_TRANSPORT_MIRROR_WIRE)
    {
        const char* g = getenv("RT_MG_WIRE_GBS");
        const char* l = getenv("RT_MG_WIRE_LAT_US");
        if (g && atof(g) > 0.0) m->wire_gbs = atof(g);
        if (l && atof(l) >= 0.0) m->wire_lat_us = atof(l);
    }
    else if (world > 1 && transport != RT_MG_TRANSPORT_MIRROR) MG_FAIL(m, RT_ERR_ARG, "unknown transport %d", transport);
    return RT_OK;
}

int rt_mg_destroy(rt_mg* m)
{
    if (!m) return RT_ERR_ARG;
    if (m->hp.on && m->stats.frames)
    {
        unsigned long long sum = 0;
        for (int k = 0; k < HP_N; ++k) sum += m->hp.ns[k];
        fprintf(stderr, "rt_mg host profile, rank %d: %llu frames since the last reset, %.1f us per frame in rt_mg_frame_step\n", m->rank, m->stats.frames,
                (double)m->stats.host_ns / (double)m->stats.frames / 1e3);
        for (int k = 0; k < HP_N; ++k)
            fprintf(stderr, "  %-46s %7.1f us per frame, %5.1f calls per frame\n", HP_NAME[k], (double)m->hp.ns[k] / (double)m->stats.frames / 1e3,
                    (double)m->hp.calls[k] / (double)m->stats.frames);
        fprintf(stderr, "  %-46s %7.1f us per frame\n", "the driver's own bookkeeping (rest)", ((double)m->stats.host_ns - (double)sum) / (double)m->stats.frames / 1e3);
    }
    if (m->ctx) rt_sync(m->ctx);
    if (m->comm) hipStreamSynchronize(m->comm);
    if (m->prep) hipStreamSynchronize(m->prep);
    m->pending_local.clear();
    for (auto& s : m->sides)
    {
        for (int k = 0; k < 2; ++k) { hipFree(s.send_buf[k]); s.last_send[k].reset(); }
        for (auto& l : s.last_bm) l.reset();
        s.last_flags.reset();
        hipFree(s.recv_buf); hipFree(s.flags_send); hipFree(s.flags_recv);
    }
    for (int k = 0; k < rt_mg::NSLOT; ++k)
    {
        hipFree(m->bm_arena[k]);
        if (m->cnt_all[k]) hipHostFree(m->cnt_all[k]);
    }
    if (m->shm.base)
    {
        munmap(m->shm.base, m->shm.bytes);
        if (m->rank == 0) shm_unlink(m->shm.name.c_str());
    }
    if (m->nccl) g_rccl.CommDestroy(m->nccl);
    if (m->ev_packed) hipEventDestroy(m->ev_packed);
    if (m->ev_arrived) hipEventDestroy(m->ev_arrived);
    if (m->ev_arrived2) hipEventDestroy(m->ev_arrived2);
    for (auto& e : m->ev_plan) if (e) hipEventDestroy(e);
    if (m->ev_gbuf) hipEventDestroy(m->ev_gbuf);
    for (auto& e : m->ev_marked) if (e) hipEventDestroy(e);
    if (m->ev_carried) hipEventDestroy(m->ev_carried);
    if (m->ev_t0) hipEventDestroy(m->ev_t0);
    if (m->ev_t1) hipEventDestroy(m->ev_t1);
    if (m->comm) hipStreamDestroy(m->comm);
    if (m->prep && m->own_prep) hipStreamDestroy(m->prep);
    delete m;
    return RT_OK;
}

int rt_mg_get_stats(rt_mg* m, rt_mg_stats* out)
{
    if (!m || !out) return RT_ERR_ARG;
    m->stats.gpu_ns_per_frame = 0;
    if (m->frames_timed >= 2)
    {
        /* GPU clock between the starts of the first and the latest frame since the reset */
        float ms = 0.0f;
        if (hipEventSynchronize(m->ev_t1) == hipSuccess && hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1) == hipSuccess)
            m->stats.gpu_ns_per_frame = (unsigned long long)((double)ms * 1e6 / (double)(m->frames_timed - 1));
    }
    m->stats.wire_ns = m->stats_wire_ns;
    *out = m->stats;
    return RT_OK;
}
int rt_mg_reset_stats(rt_mg* m)
{
    if (!m) return RT_ERR_ARG;
    memset(&m->stats, 0, sizeof(m->stats));
    m->stats_wire_ns = 0;
    memset(m->hp.ns, 0, sizeof(m->hp.ns)); memset(m->hp.calls, 0, sizeof(m->hp.calls));
    m->frames_timed = 0;
    return RT_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------ exchange = post + complete */
/* LOCAL: before a buffer that a peer may still be copying from is overwritten */
static int local_guard(rt_mg* m, std::shared_ptr<LocalMsg>& last, hipStream_t writer)
{
    if (last && last->consumed_recorded) MG_HIP(m, hipStreamWaitEvent(writer, last->consumed, 0));
    last.reset();
    return RT_OK;
}

/* post: everything the sends read has been enqueued on the main stream (or lanes joined into it) */
static int post(rt_mg* m, std::vector<Exchange>&& xs)
{
    hipStream_t ms = main_stream(m);
    m->pending_x = std::move(xs);
    m->pending = true;
    for (auto& x : m->pending_x)
        for (auto& p : x.parts) { m->stats.bytes_sent += p.send_bytes; m->stats.messages += 1; }
    if (m->transport == RT_MG_TRANSPORT_MIRROR) return RT_OK;
    if (m->transport == RT_MG_TRANSPORT_MIRROR_WIRE)
    {
        /* the data is ready HERE (the send would be posted now): note the GPU clock; the copy and the wait for the modelled link
         * follow where the exchange completes, so that whatever runs in between hides the wire as it would on real links */
        size_t most = 0;
        for (auto& x : m->pending_x)
        {
            size_t out = 0, in = 0;
            for (auto& p : x.parts) { out += p.send_bytes; in += p.recv_bytes; }
            most = std::max(most, std::max(out, in));
        }
        m->mirror_wire_ns = (unsigned long long)((double)most / m->wire_gbs + m->wire_lat_us * 1e3);
        m->stats_wire_ns += m->mirror_wire_ns;
        m->mirror_wire_slot = m->wire_slot & 3; /* main stream: slots 0-3 */
        m->wire_slot = (m->wire_slot + 1) & 7;
        int wrc; { HP(m, HP_PACK); wrc = rt_wire_delay(m->ctx, 0, m->mirror_wire_slot, 0); }
        if (wrc != RT_OK) MG_FAIL(m, wrc, "rt_wire_delay: %s", rt_last_error(m->ctx));
        return RT_OK;
    }
    if (m->transport == RT_MG_TRANSPORT_SHM)
    {
        MG_HIP(m, hipStreamSynchronize(ms)); /* what the parts hold must be final before the host copies it */
        for (auto& x : m->pending_x)
        {
            ShmMailbox* b = m->shm.box(m->rank, x.peer);
            for (long spins = 0; b->consumed.load(std::memory_order_acquire) != b->posted.load(std::memory_order_relaxed); ++spins)
            {
                if (spins > 600000000L) MG_FAIL(m, RT_ERR_COMM, "SHM transport: rank %d never consumed the previous message of rank %d", x.peer, m->rank);
                if ((spins & 1023) == 1023) usleep(50);
            }
            if (x.parts.size() > 8) MG_FAIL(m, RT_ERR_STATE, "SHM transport: more than 8 parts in a message");
            size_t off = 0;
            for (size_t i = 0; i < x.parts.size(); ++i)
            {
                if (off + x.parts[i].send_bytes + 256 > m->shm.slot) MG_FAIL(m, RT_ERR_STATE, "SHM transport: message larger than the mailbox");
                MG_HIP(m, hipMemcpy(m->shm.data(m->rank, x.peer) + off, x.parts[i].send, x.parts[i].send_bytes, hipMemcpyDeviceToHost));
                b->bytes[i] = x.parts[i].send_bytes;
                off += (x.parts[i].send_bytes + 255) & ~(size_t)255;
            }
            b->n_parts = (unsigned int)x.parts.size();
            b->posted.fetch_add(1, std::memory_order_release);
        }
        return RT_OK;
    }
    if (is_rccl(m))
    {
        const bool self = is_self(m);
        const bool wire = m->transport == RT_MG_TRANSPORT_WIRE_MODEL;
        const bool on_main = m->comm_on_main && (m->n_itr == 0 || m->two_lanes); /* one lane with interior rows: they follow the boundary rows on the main stream */
        hipStream_t cs = on_main ? ms : m->comm;
        m->pending_on_main = on_main;
        if (!on_main)
        {
            { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_packed, ms)); }
            { HP(m, HP_EVENT); MG_HIP(m, hipStreamWaitEvent(cs, m->ev_packed, 0)); }
        }
        unsigned long long wire_ns = 0;
        if (wire)
        {
            /* the data of this exchange is ready when stream `cs` gets here: note the GPU clock (rt_wire_delay runs on the
             * context's current stream) */
            size_t most = 0;
            for (auto& x : m->pending_x)
            {
                size_t out = 0, in = 0;
                for (auto& p : x.parts) { out += p.send_bytes; in += p.recv_bytes; }
                most = std::max(most, std::max(out, in)); /* one full-duplex link per neighbour */
            }
            wire_ns = (unsigned long long)((double)most / m->wire_gbs + m->wire_lat_us * 1e3);
            m->stats_wire_ns += wire_ns;
            if (cs != ms) MG_RT(m, rt_set_stream(m->ctx, cs));
            /* slots per stream: in-order execution of ONE stream keeps a slot's next stamp behind the wait that reads it */
            const int wslot = (on_main ? 0 : 4) + (m->wire_slot & 3);
            int wrc; { HP(m, HP_PACK); wrc = rt_wire_delay(m->ctx, 0, wslot, 0); }
            if (cs != ms) rt_set_stream(m->ctx, ms);
            if (wrc != RT_OK) MG_FAIL(m, wrc, "rt_wire_delay: %s", rt_last_error(m->ctx));
        }
        {
        HP(m, HP_RCCL);
        MG_NCCL(m, g_rccl.GroupStart());
        for (auto& x : m->pending_x)
            for (auto& p : x.parts)
            {
                if (self && p.send_bytes != p.recv_bytes) MG_FAIL(m, RT_ERR_STATE, "RCCL_SELF transport: %zu bytes out, %zu in", p.send_bytes, p.recv_bytes);
                MG_NCCL(m, g_rccl.Send(p.send, p.send_bytes, ncclUint8, self ? 0 : x.peer, m->nccl, cs));
                MG_NCCL(m, g_rccl.Recv(p.recv, p.recv_bytes, ncclUint8, self ? 0 : x.peer, m->nccl, cs));
            }
        MG_NCCL(m, g_rccl.GroupEnd());
        }
        if (wire)
        {
            if (cs != ms) MG_RT(m, rt_set_stream(m->ctx, cs));
            const int wslot = (on_main ? 0 : 4) + (m->wire_slot & 3);
            int wrc; { HP(m, HP_PACK); wrc = rt_wire_delay(m->ctx, 1, wslot, wire_ns); }
            if (cs != ms) rt_set_stream(m->ctx, ms);
            if (wrc != RT_OK) MG_FAIL(m, wrc, "rt_wire_delay: %s", rt_last_error(m->ctx));
            m->wire_slot = (m->wire_slot + 1) & 7;
        }
        if (!on_main) MG_HIP(m, hipEventRecord(m->ev_arrived, cs));
        return RT_OK;
    }
    m->pending_local.clear();
    for (auto& x : m->pending_x)
    {
        auto msg = std::make_shared<LocalMsg>();
        for (auto& p : x.parts) msg->parts.push_back({p.send, p.send_bytes});
        MG_HIP(m, hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming));
        MG_HIP(m, hipEventCreateWithFlags(&msg->consumed, hipEventDisableTiming));
        { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(msg->ready, ms)); }
        m->hub->box[{m->rank, x.peer}].push_back(msg);
        m->pending_local.push_back(msg);
    }
    return RT_OK;
}

/* complete: after it, the received bytes are visible to work enqueued on the main stream */
static int complete(rt_mg* m)
{
    if (!m->pending) return RT_OK;
    hipStream_t ms = main_stream(m);
    m->pending = false;
    if (is_rccl(m))
    {
        if (!m->pending_on_main) MG_HIP(m, hipStreamWaitEvent(ms, m->ev_arrived, 0));
        return RT_OK;
    }
    if (m->transport == RT_MG_TRANSPORT_SHM)
    {
        MG_HIP(m, hipStreamSynchronize(ms)); /* whatever still reads the receive buffers is done */
        for (auto& x : m->pending_x)
        {
            ShmMailbox* b = m->shm.box(x.peer, m->rank);
            for (long spins = 0; b->posted.load(std::memory_order_acquire) == b->consumed.load(std::memory_order_relaxed); ++spins)
            {
                if (spins > 600000000L) MG_FAIL(m, RT_ERR_COMM, "SHM transport: rank %d never posted to rank %d", x.peer, m->rank);
                if ((spins & 1023) == 1023) usleep(50);
            }
            if (b->n_parts != x.parts.size()) MG_FAIL(m, RT_ERR_STATE, "SHM transport: message shape mismatch between ranks %d and %d", x.peer, m->rank);
            size_t off = 0;
            for (size_t i = 0; i < x.parts.size(); ++i)
            {
                if (b->bytes[i] != x.parts[i].recv_bytes)
                    MG_FAIL(m, RT_ERR_STATE, "SHM transport: rank %d sends %llu bytes, rank %d expects %zu", x.peer, b->bytes[i], m->rank, x.parts[i].recv_bytes);
                MG_HIP(m, hipMemcpy(x.parts[i].recv, m->shm.data(x.peer, m->rank) + off, x.parts[i].recv_bytes, hipMemcpyHostToDevice));
                off += (x.parts[i].recv_bytes + 255) & ~(size_t)255;
            }
            b->consumed.fetch_add(1, std::memory_order_release);
        }
        return RT_OK;
    }
    if (m->transport == RT_MG_TRANSPORT_MIRROR || m->transport == RT_MG_TRANSPORT_MIRROR_WIRE)
    {
        /* every rank receives what it sent (a neighbour that mirrors it): same launches and bytes as a
         * real exchange with no peer to wait for — the overhead measurements of tools/strip_overhead.py */
        /* one launch for the whole group, as one grouped ncclSend/ncclRecv is */
        const void* src[8];
        void* dst[8];
        size_t nb[8];
        int n = 0;
        for (auto& x : m->pending_x)
            for (auto& p : x.parts)
            {
                if (p.send_bytes != p.recv_bytes) MG_FAIL(m, RT_ERR_STATE, "MIRROR transport: %zu bytes out, %zu in", p.send_bytes, p.recv_bytes);
                if (n == 8) { MG_RT(m, rt_copy_parts(m->ctx, n, src, dst, nb)); n = 0; }
                src[n] = p.send; dst[n] = p.recv; nb[n] = p.recv_bytes; ++n;
            }
        MG_RT(m, rt_copy_parts(m->ctx, n, src, dst, nb));
        if (m->transport == RT_MG_TRANSPORT_MIRROR_WIRE)
        {
            int wrc; { HP(m, HP_PACK); wrc = rt_wire_delay(m->ctx, 1, m->mirror_wire_slot, m->mirror_wire_ns); }
            if (wrc != RT_OK) MG_FAIL(m, wrc, "rt_wire_delay: %s", rt_last_error(m->ctx));
        }
        return RT_OK;
    }
    for (auto& x : m->pending_x)
    {
        auto& q = m->hub->box[{x.peer, m->rank}];
        if (q.empty()) MG_FAIL(m, RT_ERR_STATE, "LOCAL transport: rank %d has not posted to rank %d yet (drive all ranks in lock-step)", x.peer, m->rank);
        auto msg = q.front();
        q.pop_front();
        if (msg->parts.size() != x.parts.size()) MG_FAIL(m, RT_ERR_STATE, "LOCAL transport: message shape mismatch between ranks %d and %d", x.peer, m->rank);
        { HP(m, HP_EVENT); MG_HIP(m, hipStreamWaitEvent(ms, msg->ready, 0)); }
        if (x.parts.size() > 8) MG_FAIL(m, RT_ERR_STATE, "LOCAL transport: more than 8 parts in a message");
        const void* src[8];
        void* dst[8];
        size_t nb[8];
        for (size_t i = 0; i < x.parts.size(); ++i)
        {
            if (msg->parts[i].second != x.parts[i].recv_bytes)
                MG_FAIL(m, RT_ERR_STATE, "LOCAL transport: rank %d sends %zu bytes, rank %d expects %zu", x.peer, msg->parts[i].second, m->rank, x.parts[i].recv_bytes);
            src[i] = msg->parts[i].first; dst[i] = x.parts[i].recv; nb[i] = x.parts[i].recv_bytes;
        }
        MG_RT(m, rt_copy_parts(m->ctx, (int)x.parts.size(), src, dst, nb));
        { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(msg->consumed, ms)); }
        msg->consumed_recorded = true;
    }
    return RT_OK;
}

/* ------------------------------------------------------------------ the frame */
static size_t list_bytes(uint32_t count) { return (size_t)(count > 0 ? count : 1) * 80; }

/* record counts of the plan in `slot`: word 0 of every bitmap of the arena -> pinned host, one strided copy */
static int fetch_counts(rt_mg* m, int slot, hipStream_t st)
{
    const size_t rows = (size_t)m->max_passes * 2 * m->sides.size();
    HP(m, HP_COUNTS);
    MG_HIP(m, hipMemcpy2DAsync(m->cnt_all[slot], 4, m->bm_arena[slot], m->bm_stride * 4, 4, rows, hipMemcpyDeviceToHost, st));
    return RT_OK;
}
static uint32_t need_count(const rt_mg*, const rt_mg::Side& s, int slot, int k) { return s.cnt_h[slot][k]; }
static uint32_t give_count(const rt_mg* m, const rt_mg::Side& s, int slot, int k) { return s.cnt_h[slot][(size_t)m->max_passes * m->sides.size() + k]; }

/* need-bitmaps of `frame` for both neighbours (RNG replay on the device): one memset, one mark launch, one scan */
static int mark_plan(rt_mg* m, int frame, int slot, hipStream_t writer)
{
    void* bm[2] = {nullptr, nullptr};
    for (auto& s : m->sides)
    {
        if (m->transport == RT_MG_TRANSPORT_LOCAL) { int rc = local_guard(m, s.last_bm[slot], writer); if (rc != RT_OK) return rc; }
        bm[s.side] = s.need_bm[slot];
    }
    HP(m, HP_MARK);
    MG_RT(m, rt_halo_mark_sides(m->ctx, frame, 0, m->passes, bm[0], bm[1]));
    return RT_OK;
}

/* the bitmap part of an exchange: my need-bitmaps go to the owner of those rows, its need-bitmaps
 * (= what I must give) come back */
static void add_bitmap_parts(rt_mg* m, int slot, std::vector<Exchange>& xs)
{
    for (size_t i = 0; i < m->sides.size(); ++i)
    {
        auto& s = m->sides[i];
        const size_t bytes = s.bm_words * 4 * (size_t)m->passes;
        xs[i].parts.push_back({s.need_bm[slot], bytes, s.give_bm[slot], bytes});
    }
}

/* up to two row ranges, ONE launch per kernel (a small launch lasts as long as its slowest wavefront:
 * two boundary bands back to back would pay that twice) */
static int run_rows(rt_mg* m, int stage, int part, const int (*ranges)[2], int n, bool second_lane)
{
    if (n <= 0) return RT_OK;
    HP(m, HP_LAUNCH);
    MG_RT(m, rt_frame_stage_run_ranges(m->ctx, m->frame, stage, part, n, &ranges[0][0], second_lane ? 1 : 0));
    return RT_OK;
}

/* halo exchange of the buffer stage `stage` has just written on the boundary rows (input of spatial
 * pass `stage`); with_plan: the bitmaps of the next frame's plan ride along */
static int post_halo(rt_mg* m, int stage, int buf, bool with_plan)
{
    hipStream_t ms = main_stream(m);
    const int slot = rt_mg::slot_of(m->frame), nslot = rt_mg::slot_of((long long)m->frame + 1), k = stage;
    std::vector<Exchange> xs(m->sides.size());
    if (m->use_sparse && !(m->fuse_halos && k >= 1))
    {
        /* the marked records of both boundary bands -> two dense lists, one launch (exchange 0: the candidates come from
         * the pipelined stage 0; later exchanges: the spatial pass that produced the records has filled the lists) */
        int row0[2], nrows[2];
        const void* bms[2];
        void* dsts[2];
        for (size_t i = 0; i < m->sides.size(); ++i)
        {
            auto& s = m->sides[i];
            if (m->transport == RT_MG_TRANSPORT_LOCAL) { int rc = local_guard(m, s.last_send[k & 1], ms); if (rc != RT_OK) return rc; }
            row0[i] = s.send_row0; nrows[i] = s.n_rows;
            bms[i] = s.give_bm[slot] + (size_t)k * s.bm_words;
            dsts[i] = s.send_buf[k & 1];
        }
        HP(m, HP_PACK);
        MG_RT(m, rt_halo_pack_sparse_ranges(m->ctx, buf, (int)m->sides.size(), row0, nrows, bms, dsts));
    }
    for (size_t i = 0; i < m->sides.size(); ++i)
    {
        auto& s = m->sides[i];
        xs[i].peer = s.peer;
        if (m->use_sparse)
        {
            const uint32_t give = give_count(m, s, slot, k), need = need_count(m, s, slot, k);
            xs[i].parts.push_back({s.send_buf[k & 1], list_bytes(give), s.recv_buf, list_bytes(need)});
            m->stats.records_sent += give;
        }
        else
        {
            void *srec, *srad, *rrec, *rrad;
            size_t nrec, nrad;
            MG_RT(m, rt_res_region(m->ctx, buf, s.send_row0, s.n_rows, &srec, &nrec, &srad, &nrad));
            MG_RT(m, rt_res_region(m->ctx, buf, s.recv_row0, s.n_rows, &rrec, &nrec, &rrad, &nrad));
            xs[i].parts.push_back({srec, nrec, rrec, nrec});
            xs[i].parts.push_back({srad, nrad, rrad, nrad});
            m->stats.records_sent += (unsigned long long)s.n_rows * m->W;
        }
    }
    if (with_plan) add_bitmap_parts(m, nslot, xs);
    int rc = post(m, std::move(xs));
    if (rc != RT_OK) return rc;
    if (m->transport == RT_MG_TRANSPORT_LOCAL)
        for (size_t i = 0; i < m->sides.size(); ++i)
        {
            if (m->use_sparse) m->sides[i].last_send[k & 1] = m->pending_local[i];
            if (with_plan) m->sides[i].last_bm[nslot] = m->pending_local[i];
        }
    m->pending_buf = buf; m->pending_k = k; m->pending_carries_plan = with_plan;
    return RT_OK;
}

static int finish_halo(rt_mg* m)
{
    hipStream_t ms = main_stream(m);
    const bool carried = m->pending_carries_plan;
    int rc = complete(m);
    if (rc != RT_OK) return rc;
    const bool fused = m->use_sparse && m->fuse_halos; /* the consuming pass reads the received lists itself */
    const int slot = rt_mg::slot_of(m->frame), nslot = rt_mg::slot_of((long long)m->frame + 1), k = m->pending_k;
    if (m->use_sparse && !fused)
    {
        int row0[2], nrows[2];
        const void *bms[2], *srcs[2];
        for (size_t i = 0; i < m->sides.size(); ++i)
        {
            auto& s = m->sides[i];
            row0[i] = s.recv_row0; nrows[i] = s.n_rows;
            bms[i] = s.need_bm[slot] + (size_t)k * s.bm_words;
            srcs[i] = s.recv_buf;
        }
        HP(m, HP_PACK);
        MG_RT(m, rt_halo_unpack_sparse_ranges(m->ctx, m->pending_buf, (int)m->sides.size(), row0, nrows, bms, srcs));
    }
    if (carried)
    {
        /* the plan of frame + 1 is complete on the device: its counts travel to the host behind it, on the prep (= tail) stream
         * (the spatial passes of this frame do not wait for that copy). There the copy queues behind the previous frame's resolve and
         * the marks of frame + 2, so the host — which cannot size the messages of frame + 1 without the counts — wakes up late, and
         * with it the look-ahead stage 0 of frame + 2 it enqueues. r05 A/B, RT_MG_COUNTS_ON_COMM=1 (the idle communication stream:
         * counts 400 us earlier at 4K): SLOWER, 0.868 -> 1.19 ms at 4K and 0.293 -> 0.365 at 1080p (rank 4 of 8, MIRROR) — with
         * stage 0 of frame + 2 started that early, generate and resolve hold the wavefront slots and the main stream's chain of
         * small launches (pack, exchange, boundary pass, ...) crawls behind them: 205 us for the pack that takes 13 alone
         * (profiles/r05_strip_timelines.txt). The late counts are the throttle that keeps the chain fed. */
        hipStream_t cs = m->counts_on_comm ? m->comm : m->prep;
        { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_carried, ms)); }
        { HP(m, HP_EVENT); MG_HIP(m, hipStreamWaitEvent(cs, m->ev_carried, 0)); }
        rc = fetch_counts(m, nslot, cs);
        if (rc != RT_OK) return rc;
        { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_plan[nslot], cs)); }
        m->plan_frame[nslot] = (long long)m->frame + 1;
        rt_state_epoch(m->ctx, &m->plan_epoch[nslot]);
        m->plan_passes = m->passes;
        m->pending_carries_plan = false;
    }
    return RT_OK;
}

extern "C" {

int rt_mg_frame_begin(rt_mg* m, int frame, int clear_first)
{
    if (!m) return RT_ERR_ARG;
    if (m->seg != rt_mg::SEG_IDLE) MG_FAIL(m, RT_ERR_STATE, "the previous frame has not been stepped to its end");
    rt_options o;
    MG_RT(m, rt_options_get(m->ctx, &o));
    m->passes = o.spatial_resampling_passes;
    if (m->passes > m->max_passes) MG_FAIL(m, RT_ERR_UNSUPPORTED, "more than %d spatial passes", m->max_passes);
    m->frame = frame; m->clear_first = clear_first;
    const bool exchanges = !m->sides.empty() && m->passes > 0;
    m->use_sparse = exchanges && m->sparse && o.use_spatial_resampling;
    m->warm = false;
    if (m->use_sparse)
    {
        uint64_t epoch = 0;
        rt_state_epoch(m->ctx, &epoch);
        const int slot = rt_mg::slot_of(frame);
        if (m->plan_frame[slot] == (long long)frame && m->plan_epoch[slot] == epoch && m->plan_passes == m->passes)
        {
            /* recorded in the middle of the previous frame: in a steady loop this returns at once */
            const auto t0 = std::chrono::steady_clock::now();
            { const int wrc = wait_event_deadline(m, m->ev_plan[slot], "the halo plan of this frame (carried by the previous frame's first exchange)"); if (wrc != RT_OK) return wrc; }
            m->stats.plan_wait_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            m->warm = true;
        }
    }
    m->seg = rt_mg::SEG_RAYCAST;
    { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->frames_timed == 0 ? m->ev_t0 : m->ev_t1, main_stream(m))); }
    m->frames_timed += 1;
    m->stats.frames += 1;
    if (m->use_sparse && !m->warm) m->stats.cold_frames += 1;
    return RT_OK;
}

/* one segment of the frame: [finish the exchange posted by the previous segment,] compute, [post the
 * next exchange]. *more = 0 after the last one. All ranks take the same sequence of segments. */
static int frame_step(rt_mg* m, int* more);
int rt_mg_frame_step(rt_mg* m, int* more)
{
    if (!m || !more) return RT_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = frame_step(m, more);
    m->stats.host_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
static int frame_step(rt_mg* m, int* more)
{
    hipStream_t ms = main_stream(m);
    const int slot = rt_mg::slot_of(m->frame), nslot = rt_mg::slot_of((long long)m->frame + 1), P = m->passes;
    const bool exchanges = !m->sides.empty() && P > 0;
    const int all[1][2] = {{m->a, m->b}};
    *more = 1;
    int rc = RT_OK;
    switch (m->seg)
    {
        case rt_mg::SEG_IDLE: MG_FAIL(m, RT_ERR_STATE, "rt_mg_frame_begin first");
        case rt_mg::SEG_RAYCAST:
        {
            { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_begin(m->ctx, m->frame, 0, m->clear_first)); }
            /* [clear,] raycast of all owned rows: the marks and the second lane need the G-buffer */
            { HP(m, HP_LAUNCH); MG_RT(m, rt_frame_stage_run_part(m->ctx, m->frame, 0, 1, m->a, m->b)); }
            if (!m->use_sparse || m->warm) { m->seg = rt_mg::SEG_GENERATE; return frame_step(m, more); }
            uint64_t epoch = 0;
            rt_state_epoch(m->ctx, &epoch);
            if (m->flags_epoch == epoch) { m->seg = rt_mg::SEG_COLD_MARK; return frame_step(m, more); }
            /* cold frame, 1: shaded flags of the neighbours' boundary rows (valid until the camera moves) */
            std::vector<Exchange> xs(m->sides.size());
            for (size_t i = 0; i < m->sides.size(); ++i)
            {
                auto& s = m->sides[i];
                if (m->transport == RT_MG_TRANSPORT_LOCAL) { rc = local_guard(m, s.last_flags, ms); if (rc != RT_OK) return rc; }
                MG_RT(m, rt_halo_flags_pack(m->ctx, s.send_row0, s.n_rows, s.flags_send));
                const size_t fb = rt_halo_flags_bytes(m->ctx, s.n_rows);
                xs[i].peer = s.peer;
                xs[i].parts.push_back({s.flags_send, fb, s.flags_recv, fb});
            }
            rc = post(m, std::move(xs));
            if (rc != RT_OK) return rc;
            if (m->transport == RT_MG_TRANSPORT_LOCAL)
                for (size_t i = 0; i < m->sides.size(); ++i) m->sides[i].last_flags = m->pending_local[i];
            m->seg = rt_mg::SEG_COLD_FLAGS;
            return RT_OK;
        }
        case rt_mg::SEG_COLD_FLAGS:
        {
            rc = complete(m);
            if (rc != RT_OK) return rc;
            for (auto& s : m->sides) MG_RT(m, rt_halo_flags_unpack(m->ctx, s.recv_row0, s.n_rows, s.flags_recv));
            rt_state_epoch(m->ctx, &m->flags_epoch);
            m->seg = rt_mg::SEG_COLD_MARK;
            return frame_step(m, more);
        }
        case rt_mg::SEG_COLD_MARK:
        {
            /* cold frame, 2: what I will gather from each neighbour in every pass of THIS frame. A plan the previous
             * frame carried for this slot may still be on its way to the host on the prep stream (counts copy into
             * cnt_all[slot], reading bm_arena[slot]): the main stream waits for it before it re-marks the slot. */
            { HP(m, HP_EVENT); MG_HIP(m, hipStreamWaitEvent(ms, m->ev_plan[slot], 0)); }
            /* stale look-ahead marks (other epoch) still queued on the prep stream: of this slot — they write its bitmaps — and of
             * the other slots too: every mark may rebuild the context's shaded-bit rows, which the mark below reads */
            for (auto& e : m->ev_marked) MG_HIP(m, hipStreamWaitEvent(ms, e, 0));
            m->marked_frame[slot] = -1;
            rc = mark_plan(m, m->frame, slot, ms);
            if (rc != RT_OK) return rc;
            std::vector<Exchange> xs(m->sides.size());
            for (size_t i = 0; i < m->sides.size(); ++i) xs[i].peer = m->sides[i].peer;
            add_bitmap_parts(m, slot, xs);
            rc = post(m, std::move(xs));
            if (rc != RT_OK) return rc;
            if (m->transport == RT_MG_TRANSPORT_LOCAL)
                for (size_t i = 0; i < m->sides.size(); ++i) m->sides[i].last_bm[slot] = m->pending_local[i];
            m->seg = rt_mg::SEG_COLD_BITMAPS;
            return RT_OK;
        }
        case rt_mg::SEG_COLD_BITMAPS:
        {
            rc = complete(m);
            if (rc != RT_OK) return rc;
            rc = fetch_counts(m, slot, ms);
            if (rc != RT_OK) return rc;
            /* the one host wait of a cold frame: message sizes */
            { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_arrived2, ms)); }
            rc = wait_event_deadline(m, m->ev_arrived2, "the cold frame's bitmap exchange");
            if (rc != RT_OK) return rc;
            m->plan_frame[slot] = m->frame;
            rt_state_epoch(m->ctx, &m->plan_epoch[slot]);
            m->plan_passes = P;
            m->seg = rt_mg::SEG_GENERATE;
            return frame_step(m, more);
        }
        case rt_mg::SEG_GENERATE:
        {
            const bool lanes = m->two_lanes && exchanges && m->n_itr > 0;
            if (lanes)
            {
                { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_fork(m->ctx)); /* the second lane starts behind the raycast */ }
                rc = run_rows(m, 0, 2, m->itr, m->n_itr, true);
                if (rc != RT_OK) return rc;
            }
            if (m->use_sparse)
            {
                /* Plans of the frames to come (valid while the camera stays), marked on a stream of their own from this
                 * frame's G-buffer: frame + 2 in every frame; frame + 1 too if the previous frame did not mark it (a cold
                 * frame: then the first exchange below waits for that mark, once). */
                uint64_t epoch = 0;
                rt_state_epoch(m->ctx, &epoch);
                { HP(m, HP_EVENT); MG_HIP(m, hipEventRecord(m->ev_gbuf, ms)); }
                { HP(m, HP_EVENT); MG_HIP(m, hipStreamWaitEvent(m->prep, m->ev_gbuf, 0)); }
                MG_RT(m, rt_set_stream(m->ctx, m->prep));
                for (int ahead = 1; ahead <= 2 && rc == RT_OK; ++ahead)
                {
                    const long long tf = (long long)m->frame + ahead;
                    const int ts = rt_mg::slot_of(tf);
                    if (ahead == 1 && m->marked_frame[ts] == tf && m->marked_epoch[ts] == epoch && m->marked_passes[ts] == P) continue;
                    rc = mark_plan(m, (int)tf, ts, m->prep);
                    if (rc != RT_OK) break;
                    if (hipEventRecord(m->ev_marked[ts], m->prep) != hipSuccess) { rc = RT_ERR_HIP; m->err = "hipEventRecord(ev_marked) failed"; break; }
                    m->marked_frame[ts] = tf; m->marked_epoch[ts] = epoch; m->marked_passes[ts] = P;
                }
                rt_set_stream(m->ctx, ms);
                if (rc != RT_OK) return rc;
            }
            if (exchanges)
            {
                rc = run_rows(m, 0, 2, m->bnd, m->n_bnd, false);
                if (rc != RT_OK) return rc;
                if (!lanes) { rc = run_rows(m, 0, 2, m->itr, m->n_itr, false); if (rc != RT_OK) return rc; }
            }
            else { rc = run_rows(m, 0, 2, all, 1, false); if (rc != RT_OK) return rc; }
            if (exchanges)
            {
                /* the next frame's bitmaps ride on this first message */
                const bool carry = m->use_sparse;
                int buf = 0;
                MG_RT(m, rt_frame_stage_output(m->ctx, 0, &buf));
                if (carry) MG_HIP(m, hipStreamWaitEvent(ms, m->ev_marked[nslot], 0)); /* marked during the previous frame: long done */
                rc = post_halo(m, 0, RT_RES_PHYS + buf, carry);
                if (rc != RT_OK) return rc;
                { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_end(m->ctx, 0)); }
            }
            else { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_end(m->ctx, 0)); }
            m->stage = 1;
            m->seg = P > 0 ? rt_mg::SEG_PASS : rt_mg::SEG_FINAL;
            return exchanges ? RT_OK : frame_step(m, more);
        }
        case rt_mg::SEG_PASS:
        {
            const int s = m->stage; /* spatial pass s - 1 */
            const bool lanes = m->two_lanes && exchanges && m->n_itr > 0;
            { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_begin(m->ctx, m->frame, s, 0)); }
            /* interior rows on the second lane: beside the exchange in flight (default: on real links they hide the wire), or —
             * RT_MG_INTERIOR_LATE=1 — only once the exchange has completed: its kernel then does not queue for wavefront slots
             * behind them (A/B on the stand-in transports, profiles/r04_strip_interior_late.txt) */
            if (lanes && !m->interior_late) { rc = run_rows(m, s, 0, m->itr, m->n_itr, true); if (rc != RT_OK) return rc; }
            if (exchanges) { rc = finish_halo(m); if (rc != RT_OK) return rc; }
            if (lanes && m->interior_late)
            {
                { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_fork(m->ctx)); /* the lane starts behind the completed exchange */ }
                rc = run_rows(m, s, 0, m->itr, m->n_itr, true);
                if (rc != RT_OK) return rc;
            }
            if (exchanges && m->use_sparse && m->fuse_halos)
            {
                /* this pass gathers its halo records from the lists that just arrived and fills the lists of the next
                 * exchange as it writes its boundary rows: no unpack launch before it, no pack launch after it */
                rt_halo_fuse f;
                memset(&f, 0, sizeof(f));
                for (auto& sd : m->sides)
                {
                    f.need_bitmap[sd.side] = sd.need_bm[slot] + (size_t)(s - 1) * sd.bm_words;
                    f.recv_list[sd.side] = sd.recv_buf;
                    if (s < P)
                    {
                        if (m->transport == RT_MG_TRANSPORT_LOCAL) { rc = local_guard(m, sd.last_send[s & 1], ms); if (rc != RT_OK) return rc; }
                        f.give_bitmap[sd.side] = sd.give_bm[slot] + (size_t)s * sd.bm_words;
                        f.send_list[sd.side] = sd.send_buf[s & 1];
                    }
                }
                { HP(m, HP_PACK); MG_RT(m, rt_halo_fuse_set(m->ctx, &f)); }
            }
            if (exchanges)
            {
                rc = run_rows(m, s, 0, m->bnd, m->n_bnd, false);
                if (rc != RT_OK) return rc;
                if (!lanes) { rc = run_rows(m, s, 0, m->itr, m->n_itr, false); if (rc != RT_OK) return rc; }
            }
            else { rc = run_rows(m, s, 0, all, 1, false); if (rc != RT_OK) return rc; }
            bool posted = false;
            if (exchanges && s < P)
            {
                int buf = 0;
                MG_RT(m, rt_frame_stage_output(m->ctx, s, &buf));
                rc = post_halo(m, s, RT_RES_PHYS + buf, false);
                if (rc != RT_OK) return rc;
                posted = true;
            }
            { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_end(m->ctx, s)); }
            m->stage = s + 1;
            if (s == P) m->seg = rt_mg::SEG_FINAL;
            return posted ? RT_OK : frame_step(m, more);
        }
        case rt_mg::SEG_FINAL:
        {
            { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_begin(m->ctx, m->frame, P + 1, 0)); }
            { HP(m, HP_LAUNCH); MG_RT(m, rt_frame_stage_run(m->ctx, m->frame, P + 1, m->a, m->b)); }
            { HP(m, HP_STAGE); MG_RT(m, rt_frame_stage_end(m->ctx, P + 1)); }
            m->seg = rt_mg::SEG_IDLE;
            *more = 0;
            return RT_OK;
        }
    }
    return RT_OK;
}

int rt_mg_frame(rt_mg* m, int frame, int clear_first)
{
    int rc = rt_mg_frame_begin(m, frame, clear_first);
    int more = 1;
    while (rc == RT_OK && more) rc = rt_mg_frame_step(m, &more);
    if (rc != RT_OK && m)
    {
        m->seg = rt_mg::SEG_IDLE;
        m->pending = false; m->pending_carries_plan = false;
        m->pending_x.clear(); m->pending_local.clear();
    }
    return rc;
}

/* RCCL smoke test for boxes with one GPU: a communicator of ONE rank sends a buffer to itself through
 * the same grouped ncclSend/ncclRecv path the halos use (dlopen, types, stream ordering). */
int rt_mg_selftest_rccl(size_t bytes)
{
    if (!g_rccl.load()) return RT_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return RT_ERR_COMM;
    ncclComm_t comm = nullptr;
    if (g_rccl.CommInitRank(&comm, 1, id, 0) != ncclSuccess) return RT_ERR_COMM;
    unsigned char *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    int rc = RT_ERR_HIP;
    if (hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess && hipStreamCreate(&st) == hipSuccess)
    {
        std::vector<unsigned char> h(bytes), g(bytes);
        for (size_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 131u + 7u);
        hipMemcpyAsync(a, h.data(), bytes, hipMemcpyHostToDevice, st);
        hipMemsetAsync(b, 0, bytes, st);
        bool ok = g_rccl.GroupStart() == ncclSuccess;
        ok = ok && g_rccl.Send(a, bytes, ncclUint8, 0, comm, st) == ncclSuccess;
        ok = ok && g_rccl.Recv(b, bytes, ncclUint8, 0, comm, st) == ncclSuccess;
        ok = ok && g_rccl.GroupEnd() == ncclSuccess;
        hipMemcpyAsync(g.data(), b, bytes, hipMemcpyDeviceToHost, st);
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        rc = ok ? (memcmp(h.data(), g.data(), bytes) == 0 ? RT_OK : RT_ERR_STATE) : RT_ERR_COMM;
    }
    if (st) hipStreamDestroy(st);
    hipFree(a); hipFree(b);
    g_rccl.CommDestroy(comm);
    return rc;
}

}  // extern "C"
