// adf_api.hip -- host side of the C-ABI declared in include/adf_wls.h.
//
// Orchestrates DisparityWLSFilterImpl::filter (DF.cpp:219-298) and
// FastGlobalSmootherFilterImpl::{init,filter} (FGS.cpp:141-233) as a fixed sequence of HIP kernel
// launches on the caller's stream.  No host<->device synchronisation happens on the device-pointer
// path after the workspace exists, so a caller may capture it into a hipGraph.
#include "adf_internal.h"
#include "../../include/adf_wls.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace adf;

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// error reporting for the other translation units (declared in adf_internal.h)
namespace adf {
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? ADF_ENOMEM : ADF_EHIP, "%s failed: %s",   \
                        #expr, hipGetErrorString(e_));                                        \
    } while (0)

extern "C" int adf_version(void) { return ADF_VERSION; }
extern "C" const char* adf_last_error(void) { return g_err; }
extern "C" int adf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int adf_device_pci_bus_id(int device, char* buf, int len)
{
    if (!buf || len < 16) return fail(ADF_EBADARG, "adf_device_pci_bus_id: buffer of at least 16 bytes required");
    buf[0] = 0;
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return ADF_OK;
}

// ----------------------------------------------------------------------------------------------
// shared pieces
// ----------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static Geom make_geom(int W, int H, int rx, int ry, int rw, int rh)
{
    Geom g;
    g.W = W; g.H = H; g.rx = rx; g.ry = ry; g.rw = rw; g.rh = rh;
    g.pw = round_up(rw, 64);
    g.ph = round_up(rh, 64);
    size_t a = (size_t)round_up(rh, ADF_TILE_ROWS) * g.pw, b = (size_t)rw * g.ph;   // (the wave solver's pair plane holds rows in tiles)
    g.plane = ((a > b ? a : b) + 63) / 64 * 64;
    g.frame = (size_t)W * H;
    // confidence plane: ROI column 0 16-byte aligned, >= 3 zero floats behind a row, pitch a multiple of 4 (adf_internal.h)
    g.cx0 = (4 - (rx & 3)) & 3;
    g.cpitch = round_up(W + g.cx0 + 3, 4);
    g.cframe = (size_t)g.cpitch * H;
    return g;
}

// Confidence plane as a plain W-pitch frame (the low-resolution scratch map of the down-scaled path).
static Geom plain_conf_layout(Geom g) { g.cx0 = 0; g.cpitch = g.W; g.cframe = g.frame; return g; }

// RAII: run on the handle's device, restore the caller's on exit.
struct DeviceScope {
    int prev = -1; bool switched = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() { if (switched) hipSetDevice(prev); }
};

static bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

// Growable device buffer (never shrinks; freed with the handle).
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int reserve(size_t need, hipStream_t st)
    {
        if (need <= bytes) return ADF_OK;
        if (p) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(p)); p = nullptr; bytes = 0; }
        need = (need + 255) / 256 * 256;
        HIP_TRY(adf::device_malloc(&p, need));
        bytes = need;
        // deterministic padding lanes: the sweeps read (and discard) pitch padding
        HIP_TRY(hipMemsetAsync(p, 0, need, st));
        return ADF_OK;
    }
    void release() { if (p) hipFree(p); p = nullptr; bytes = 0; }
};

// Weight LUTs (FGS.cpp:150-154, 663-675), built on the host with libm, one immutable device table per sigma seen
// (up to LUT_CACHE of them per handle).  Round 3: a table is never rewritten, so coming back to a sigma used before --
// the common way callers vary it -- is a pointer switch with no device work and no synchronisation (capturable into a
// hipGraph), and a NEW sigma no longer drains the stream: its table goes into a fresh buffer no kernel in flight can be
// reading.  Only that first upload is a synchronous copy; a caller that captures filter calls must have used every
// sigma it switches between once before the capture (include/adf_wls.h).
//
// Tables are shared by every handle of the process on the same device (LutStore): the one-shot function
// fastGlobalSmootherFilter (EF.hpp:413) and the reference's own perf test (perf_fgs_filter.cpp:70-76) create a filter
// per call, and 3*256*256 libm calls cost ~1 ms on one core -- ten times the 720p filter call itself.  A table seen
// before is a look-up; a new one is built by a few threads (each entry is the same scalar libm expression as before:
// same bits).
struct LutTable {
    int device = 0; float sigma = 0; float* dev = nullptr;
    LutTable() = default;
    LutTable(const LutTable&) = delete;
    LutTable& operator=(const LutTable&) = delete;
    ~LutTable() { if (dev) { DeviceScope ds(device); hipFree(dev); } }
};

struct LutStore {
    static constexpr size_t CAP = 16;
    std::mutex m;
    std::vector<std::shared_ptr<LutTable>> tables;               // most recently used last
    static LutStore& get() { static LutStore* s = new LutStore; return *s; }   // (never destroyed: no HIP calls at exit)
    std::shared_ptr<LutTable> find(int device, float sigma)
    {
        std::lock_guard<std::mutex> lk(m);
        for (size_t k = 0; k < tables.size(); k++)
            if (tables[k]->device == device && tables[k]->sigma == sigma) {
                auto t = tables[k];
                tables.erase(tables.begin() + (ptrdiff_t)k); tables.push_back(t);
                return t;
            }
        return nullptr;
    }
    void add(const std::shared_ptr<LutTable>& t)
    {
        std::lock_guard<std::mutex> lk(m);
        // (a table dropped here lives on while a handle still refers to it; with no handle left nothing can be reading it)
        if (tables.size() >= CAP) tables.erase(tables.begin());
        tables.push_back(t);
    }
    void clear() { std::lock_guard<std::mutex> lk(m); tables.clear(); }
};

static void lut_build_host(float s, float* host)
{
    auto span = [&](int a, int b) { for (int i = a; i < b; i++) host[i] = -expf(-sqrtf((float)i) / s); };
    // Where the argument is at or below -110 the float exponential is +0 -- exp(-110) = 1.7e-48 lies 400 times below half
    // the smallest denormal, so every libm returns zero there, and the entry is -0.0f.  With the filter's usual sigma
    // (1..2) that is nine tenths of the table: those entries are stored, not computed.  The argument falls
    // monotonically with i (sqrtf and the division are monotone), so the first such index bounds the computed part.
    int n = ADF_LUT_LEVELS;
    if (s > 0.0f) {
        const double lim = 110.0 * (double)s;
        if (lim * lim * 1.001 + 2.0 < (double)ADF_LUT_LEVELS) {
            int i0 = (int)(lim * lim * 1.001) + 2;
            while (i0 < ADF_LUT_LEVELS && !(-sqrtf((float)i0) / s <= -110.0f)) i0++;   // (a check, not a search: the margin covers it)
            n = i0;
        }
    }
    for (int i = n; i < ADF_LUT_LEVELS; i++) host[i] = -0.0f;
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw >= 16 ? 8 : hw >= 4 ? hw / 2 : 1);
    if (n < 32768) nt = 1;                                        // a thread costs more to start than such a share to compute
    if (nt <= 1) { span(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    bool ok = true;
    int done = std::min(per, n);                                  // the caller's own share is [0, per)
    for (int t = 1; t < nt && ok; t++) {
        const int a = t * per, b = std::min(n, a + per);
        if (a >= b) break;
        try { th.emplace_back(span, a, b); done = b; } catch (...) { ok = false; }
    }
    span(0, std::min(per, n));
    for (auto& t : th) t.join();
    if (done < n) span(done, n);                                  // threads that could not be started
}

struct Lut {
    static constexpr int LUT_CACHE = 8;
    struct Entry { std::shared_ptr<LutTable> t; unsigned long long used; };
    std::vector<Entry> tables;
    const float* cur = nullptr;
    unsigned long long tick = 0;
    size_t bytes() const { return tables.size() * sizeof(float) * ADF_LUT_LEVELS; }
    int ensure(float s, hipStream_t st)
    {
        for (auto& e : tables)
            if (e.t->sigma == s) { e.used = ++tick; cur = e.t->dev; return ADF_OK; }
        if ((int)tables.size() >= LUT_CACHE) {                 // drop the least recently used table: kernels of
            size_t lru = 0;                                    // earlier calls on `st` may still read it
            for (size_t k = 1; k < tables.size(); k++) if (tables[k].used < tables[lru].used) lru = k;
            HIP_TRY(hipStreamSynchronize(st));
            tables.erase(tables.begin() + (ptrdiff_t)lru);
        }
        int device = 0;
        HIP_TRY(hipGetDevice(&device));
        std::shared_ptr<LutTable> t = LutStore::get().find(device, s);
        if (!t) {
            std::vector<float> host(ADF_LUT_LEVELS);
            lut_build_host(s, host.data());
            float* d = nullptr;
            HIP_TRY(hipMalloc(&d, sizeof(float) * ADF_LUT_LEVELS));
            hipError_t e = hipMemcpy(d, host.data(), sizeof(float) * ADF_LUT_LEVELS, hipMemcpyHostToDevice);
            if (e != hipSuccess) { hipFree(d); return fail(ADF_EHIP, "LUT upload failed: %s", hipGetErrorString(e)); }
            t = std::make_shared<LutTable>();
            t->device = device; t->sigma = s; t->dev = d;
            LutStore::get().add(t);
        }
        tables.push_back(Entry{t, ++tick});
        cur = t->dev;
        return ADF_OK;
    }
    // (the caller has made sure no kernel still reads the tables: handle destruction synchronises first)
    void release() { tables.clear(); cur = nullptr; }
};

// Device blocks of short-lived handles (adf_fgs: the planes and the staged image of ONE image), kept for the next
// handle instead of going back to the driver: hipMalloc + hipFree of a 4K handle's 300 MB cost more than its filter
// call, and hipFree waits for the whole device.  A block comes back with the event behind its last user; whoever takes
// it makes its own stream wait for that event first, so nobody synchronises the host.
struct BlockCache {
    struct Ent { int device; void* p; size_t bytes; hipEvent_t ready; };
    static constexpr size_t CAP_BYTES = (size_t)3 << 30;
    static constexpr size_t CAP_ENTRIES = 8;
    std::mutex m;
    std::vector<Ent> ents;                                         // oldest first
    size_t total = 0;
    static BlockCache& get() { static BlockCache* c = new BlockCache; return *c; }
    static void drop(const Ent& e)
    {
        DeviceScope ds(e.device);
        if (e.ready) { hipEventSynchronize(e.ready); hipEventDestroy(e.ready); }
        hipFree(e.p);
    }
    // a cached block of at least `need` bytes (and not wastefully larger), ordered into `st`; null if there is none
    void* take(int device, size_t need, hipStream_t st, size_t* bytes)
    {
        Ent hit{};
        {
            std::lock_guard<std::mutex> lk(m);
            size_t best = ents.size();
            for (size_t k = 0; k < ents.size(); k++)
                if (ents[k].device == device && ents[k].bytes >= need && ents[k].bytes <= need + need / 4 + ((size_t)1 << 20) &&
                    (best == ents.size() || ents[k].bytes < ents[best].bytes))
                    best = k;
            if (best == ents.size()) return nullptr;
            hit = ents[best];
            ents.erase(ents.begin() + (ptrdiff_t)best);
            total -= hit.bytes;
        }
        if (hit.ready) {
            const hipError_t e = hipStreamWaitEvent(st, hit.ready, 0);
            if (e != hipSuccess) hipEventSynchronize(hit.ready);
            hipEventDestroy(hit.ready);
        }
        *bytes = hit.bytes;
        return hit.p;
    }
    void give(int device, void* p, size_t bytes, hipEvent_t ready)
    {
        std::vector<Ent> out;
        {
            std::lock_guard<std::mutex> lk(m);
            ents.push_back(Ent{device, p, bytes, ready});
            total += bytes;
            while (!ents.empty() && (total > CAP_BYTES || ents.size() > CAP_ENTRIES)) {
                out.push_back(ents.front());
                total -= ents.front().bytes;
                ents.erase(ents.begin());
            }
        }
        for (auto& e : out) drop(e);
    }
    void clear()
    {
        std::vector<Ent> out;
        { std::lock_guard<std::mutex> lk(m); out.swap(ents); total = 0; }
        for (auto& e : out) drop(e);
    }
};

// Per-launch HIP-event timing (adf_wls_profile_*).  Events are pooled and reused.
enum KClass { K_FILL = 0, K_WEIGHTS, K_DISC, K_LRC, K_PROLOGUE, K_PASS_H_FIRST, K_PASS_H, K_PASS_V, K_PASS_V_LAST, K_RESIZE, K_COUNT };
static const char* const kclass_names[K_COUNT] = {"fill_outside", "weights", "discontinuity", "lrc_confidence",
                                                  "plain_prologue", "pass_h_first", "pass_h", "pass_v", "pass_v_last", "resize"};
struct Profiler {
    bool on = false;
    struct Rec { int cls; hipEvent_t a, b; double alg, moved; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void clear() { for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); } recs.clear(); }
    void destroy() { clear(); for (auto e : pool) hipEventDestroy(e); pool.clear(); }
};
// Brackets one launch: start event in the constructor, stop event in the destructor.
struct ProfScope {
    Profiler* p; hipStream_t st; Profiler::Rec r{};
    ProfScope(Profiler* prof, int cls, double alg, double moved, hipStream_t s) : p(prof && prof->on ? prof : nullptr), st(s)
    {
        if (!p) return;
        r.cls = cls; r.alg = alg; r.moved = moved; r.a = p->get(); r.b = p->get();
        if (r.a) hipEventRecord(r.a, st);
    }
    ~ProfScope()
    {
        if (!p) return;
        if (r.b) hipEventRecord(r.b, st);
        if (r.a && r.b) p->recs.push_back(r);
    }
};

// The six (2*num_iter) solve passes of FGS.cpp:207-212 on planes already resident on the device.
// `A` holds the right-hand sides in the orientation the first (horizontal) pass wants.
struct SolvePlanes {
    float* CH; float* CV;       // weights (orientation depends on the solver)
    float* D; float* F0; float* F1;
    float* A0; float* A1;       // ping
    float* B0; float* B1;       // pong
};

struct FinalOut {
    int epilogue; void* out; ptrdiff_t stride, pair_stride; int x0, y0, cn, c;
};

static int run_passes_exact(const Geom& g, const SolvePlanes& p, int n_rhs, float lambda, float atten,
                            int num_iter, const FinalOut& fo, int n_pairs, hipStream_t st, Profiler* prof = nullptr)
{
    float lam = lambda;
    const double px = (double)g.rw * g.rh * n_pairs;
    const double alg = (4.0 + 8.0 * n_rhs) * px;      // read weight + R rhs, write R rhs
    const double moved = (12.0 + 16.0 * n_rhs) * px;  // exact solver: D and the eliminated rhs round-trip
    for (int it = 0; it < num_iter; it++) {
        PassArgs h{};
        h.C = p.CH; h.U0 = p.A0; h.U1 = p.A1; h.D = p.D; h.F0 = p.F0; h.F1 = p.F1;
        h.O0 = p.B0; h.O1 = p.B1;
        h.nscan = g.rh; h.len = g.rw; h.pitch_in = g.ph; h.pitch_out = g.pw;
        h.plane = g.plane; h.lambda = lam;
        {
            ProfScope ps(prof, K_PASS_H, alg, moved, st);
            HIP_TRY(launch_exact_pass(h, n_rhs, EPI_PLANES, n_pairs, st)); // FGS.cpp:209
        }

        const bool last = (it == num_iter - 1);
        PassArgs v{};
        v.C = p.CV; v.U0 = p.B0; v.U1 = p.B1; v.D = p.D; v.F0 = p.F0; v.F1 = p.F1;
        v.O0 = p.A0; v.O1 = p.A1;
        v.nscan = g.rw; v.len = g.rh; v.pitch_in = g.pw; v.pitch_out = g.ph;
        v.plane = g.plane; v.lambda = lam;
        if (last) {
            v.out = fo.out; v.out_stride = fo.stride; v.out_pair_stride = fo.pair_stride;
            v.out_x0 = fo.x0; v.out_y0 = fo.y0; v.out_cn = fo.cn; v.out_c = fo.c;
        }
        {
            // the fused epilogue writes 2 (int16) instead of 4R bytes per pixel
            const double out_b = last ? (fo.epilogue == EPI_F32 ? 4.0 : fo.epilogue == EPI_U8 ? 1.0 : 2.0) * px : 4.0 * n_rhs * px;
            ProfScope ps(prof, last ? K_PASS_V_LAST : K_PASS_V, (4.0 + 4.0 * n_rhs) * px + out_b,
                         (12.0 + 12.0 * n_rhs) * px + out_b, st);
            HIP_TRY(launch_exact_pass(v, n_rhs, last ? fo.epilogue : EPI_PLANES, n_pairs, st)); // FGS.cpp:210
        }
        lam *= atten;                                                      // FGS.cpp:211 (float)
    }
    return ADF_OK;
}

// Same six passes with the on-chip partitioned solver: row-major planes, in place, algorithmic traffic.
static int run_passes_wave(const Geom& g, const SolvePlanes& p, int n_rhs, float lambda, float atten,
                           int num_iter, const FinalOut& fo, int n_pairs, hipStream_t st, Profiler* prof = nullptr,
                           const WavePassArgs* fuse_first = nullptr)
{
    float lam = lambda;
    const double px = (double)g.rw * g.rh * n_pairs;
    const double alg = (4.0 + 8.0 * n_rhs) * px;
    for (int it = 0; it < num_iter; it++) {
        WavePassArgs h{};
        h.C = p.CH; h.U0 = p.A0; h.U1 = p.A1;
        h.nscan = g.rh; h.len = g.rw; h.pitch = g.pw; h.plane = g.plane; h.lambda = lam;
        const bool fused = (it == 0 && fuse_first);
        if (fused) {   // U1 = conf, U0 = conf*float(dL) formed in the pass (DF.cpp:288-290)
            h.conf_in = fuse_first->conf_in; h.conf_frame = fuse_first->conf_frame; h.conf_pitch = fuse_first->conf_pitch;
            h.conf_x0 = fuse_first->conf_x0; h.conf_y0 = fuse_first->conf_y0;
            h.dl_in = fuse_first->dl_in; h.dl_stride = fuse_first->dl_stride; h.dl_pair_stride = fuse_first->dl_pair_stride;
            h.dl_x0 = fuse_first->dl_x0; h.dl_y0 = fuse_first->dl_y0;
            // ... or interpolated from the low-resolution maps in the pass (down-scaled path, DF.cpp:272-274)
            h.lo_conf = fuse_first->lo_conf; h.lo_conf_stride = fuse_first->lo_conf_stride; h.lo_conf_pair = fuse_first->lo_conf_pair;
            h.lo_dl = fuse_first->lo_dl; h.lo_dl_stride = fuse_first->lo_dl_stride; h.lo_dl_pair = fuse_first->lo_dl_pair;
            h.lo_w = fuse_first->lo_w; h.lo_h = fuse_first->lo_h; h.hi_x0 = fuse_first->hi_x0; h.hi_y0 = fuse_first->hi_y0;
            h.lo_scale_x = fuse_first->lo_scale_x; h.lo_scale_y = fuse_first->lo_scale_y; h.lo_post_scale = fuse_first->lo_post_scale;
            h.lo_zero_outside = fuse_first->lo_zero_outside; h.lo_vx0 = fuse_first->lo_vx0; h.lo_vy0 = fuse_first->lo_vy0;
            h.lo_vx1 = fuse_first->lo_vx1; h.lo_vy1 = fuse_first->lo_vy1; h.lo_taps = fuse_first->lo_taps; h.lo_half = fuse_first->lo_half;
        }
        {
            // C + conf + dL read, U0/U1 written (low-resolution maps: their bytes per view pixel, each row counted once)
            const double lo_b = (fused && fuse_first->lo_conf) ? 6.0 * fuse_first->lo_scale_x * fuse_first->lo_scale_y : 6.0;
            const double b = fused ? (4.0 + lo_b + 8.0) * px : alg;
            ProfScope ps(prof, fused ? K_PASS_H_FIRST : K_PASS_H, b, b, st);
            HIP_TRY(launch_wave_hpass(h, n_rhs, n_pairs, st));             // FGS.cpp:209
        }
        const bool last = (it == num_iter - 1);
        WavePassArgs v{};
        v.C = p.CV; v.U0 = p.A0; v.U1 = p.A1;
        v.nscan = g.rw; v.len = g.rh; v.pitch = g.pw; v.plane = g.plane; v.lambda = lam;
        if (last) {
            v.out = fo.out; v.out_stride = fo.stride; v.out_pair_stride = fo.pair_stride;
            v.out_x0 = fo.x0; v.out_y0 = fo.y0; v.out_cn = fo.cn; v.out_c = fo.c;
        }
        {
            const double out_b = last ? (fo.epilogue == EPI_F32 ? 4.0 : fo.epilogue == EPI_U8 ? 1.0 : 2.0) * px : 4.0 * n_rhs * px;
            const double bytes = (4.0 + 4.0 * n_rhs) * px + out_b;
            ProfScope ps(prof, last ? K_PASS_V_LAST : K_PASS_V, bytes, bytes, st);
            HIP_TRY(launch_wave_vpass(v, n_rhs, last ? fo.epilogue : EPI_PLANES, n_pairs, st)); // FGS.cpp:210
        }
        lam *= atten;                                                      // FGS.cpp:211 (float)
    }
    return ADF_OK;
}

static bool wave_fits(const Geom& g)
{
    return g.rw >= 2 && g.rh >= 2 && g.rw <= wave_max_row_len() && g.rh <= wave_max_col_len();
}

// ----------------------------------------------------------------------------------------------
// DisparityWLSFilter
// ----------------------------------------------------------------------------------------------
struct adf_wls {
    int device = 0;
    // DF.cpp:142-159
    int left_offset = 0, right_offset = 0, top_offset = 0, bottom_offset = 0;
    int min_disp = 0;
    bool use_confidence = true;
    double lambda = 8000.0, sigma_color = 1.0;
    int lrc_thresh = 24, disc_radius = 5;
    float roll_off = 0.001f;
    // EF.hpp:393 defaults used by DF.cpp:292
    double atten = 0.25; int num_iter = 3;
    int solver = ADF_SOLVER_WAVE;      // default: the throughput path (see include/adf_wls.h)
    int last_solver = ADF_SOLVER_WAVE;
    // state of the last call
    adf_rect roi{0, 0, 0, 0};
    int last_W = 0, last_H = 0, last_pairs = 0;
    int last_cpitch = 0, last_cx0 = 0;       // layout of the confidence planes of the last call (Geom::cpitch, cx0)
    long long conf_sig[4] = {0, 0, 0, 0};    // (W, H, cx0, pairs) the confidence planes were last zeroed for
    int last_path = 0;                       // ADF_PATH_* bits of the last call (adf_wls_get_last_path)
    // device memory
    Lut lut;
    DevBuf ws;    // per-chunk workspace
    DevBuf conf;  // confidence maps of the last call (n_pairs full frames)
    DevBuf stage; // host-pointer path staging
    DevBuf scaled; // down-scaled path: resized disparity + low-resolution confidence scratch
    size_t ws_limit = (size_t)64 << 30;
    Profiler prof;
    // (geometry, solver) the workspace planes were last laid out for; a change re-zeroes them so that
    // pitch padding is zero again (the wave solver treats it as identity rows without masking)
    long long ws_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // side stream: the weight kernel (guide only) runs beside the confidence kernels (disparity maps only)
    // of the same call, forked from and joined back into the caller's stream with events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = true;
    // Down-scaled call whose first row pass interpolated the maps itself: the view-sized confidence map of
    // getConfidenceMap() (DF.cpp:274) has not been materialised; adf_wls_get_confidence_* runs the float resize then.
    struct LazyConf {
        bool pending = false;
        const float* clo = nullptr; int dW = 0, dH = 0; adf_rect rlo{0, 0, 0, 0}; Geom ghi{}; bool band_map = false; int n_pairs = 0;
    } lazy_conf;
    bool scaled_half = true; // ADF_LO_HALF=0: never the half-width form of the fused low-resolution prologue (A/B, tests)
    bool scaled_fuse = true; // ADF_SCALED_FUSE=0: the down-scaled path through the two resize kernels (A/B measurements)
    bool conf_band = true;   // ADF_CONF_BAND=0: the two-kernel confidence stage (A/B measurements)
    bool merge_small = true; // ADF_MERGE_SMALL=0: never the merged preparation launch (A/B measurements)
    size_t conf_lds_floor = 0; // ADF_CONF_LDS_FLOOR_KB: see ConfBandArgs::lds_floor (A/B measurements)
    // HIP maps the streams of ONE priority level onto a small pool of hardware queues (4 by default) and two streams that
    // share a queue run one after the other: a side stream of the caller's priority lost the overlap for about one caller
    // stream in four (tools/batch_cpp.cpp: 13.5-13.6 ms per 64 x 4K call instead of 12.8-13.2).  Each priority level has a pool
    // of its own, so the side stream is created on a level the caller's stream is NOT on.  ADF_SIDE_PRIORITY overrides (A/B).
    static bool null_caller_of(hipStream_t s) { return s == nullptr || s == hipStreamLegacy || s == hipStreamPerThread; }
    int ensure_side(hipStream_t caller)
    {
        if (side) return ADF_OK;
        int least = 0, greatest = 0, prio = 0, cp = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
        if (null_caller_of(caller) || stream_is_capturing(caller) || hipStreamGetPriority(caller, &cp) != hipSuccess) { cp = 0; (void)hipGetLastError(); }
        prio = (cp != greatest) ? greatest : (greatest < least ? greatest + 1 : greatest);   // the highest level, or the one below it
        // (the NULL stream is the exception: beside a side stream of another level its calls took 13.9-14.0 ms, with one of
        // its own level 12.8-13.0 -- torch's default stream is the NULL stream)
        const bool null_caller = null_caller_of(caller);
        const char* e = getenv("ADF_SIDE_PRIORITY");
        if (e) prio = atoi(e);
        if (prio < greatest) prio = greatest;
        if (prio > least) prio = least;
        if (null_caller && !e) HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));   // the default level, as rounds 1-3 did
        else HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio));
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        return ADF_OK;
    }
};

extern "C" int adf_wls_create(adf_wls_t** out, int use_confidence, int l, int r, int t, int b, int min_disp)
{
    if (!out) return fail(ADF_EBADARG, "adf_wls_create: out is NULL");
    *out = nullptr;
    if (l < 0 || r < 0 || t < 0 || b < 0) return fail(ADF_EBADARG, "adf_wls_create: negative offset");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(ADF_ENODEV, "adf_wls_create: no HIP device visible");
    adf_wls* h = new (std::nothrow) adf_wls();
    if (!h) return fail(ADF_ENOMEM, "adf_wls_create: out of host memory");
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(ADF_EHIP, "hipGetDevice failed"); }
    h->use_confidence = use_confidence != 0;
    h->left_offset = l; h->right_offset = r; h->top_offset = t; h->bottom_offset = b;
    h->min_disp = 0; (void)min_disp; // DF.cpp:146 then :149
    if (const char* e = getenv("ADF_WS_LIMIT_GB")) {
        double gb = atof(e);
        if (gb > 0) h->ws_limit = (size_t)(gb * (double)((size_t)1 << 30));
    }
    if (const char* e = getenv("ADF_NO_OVERLAP")) h->overlap = atoi(e) == 0;   // measurement knob
    if (const char* e = getenv("ADF_CONF_BAND")) h->conf_band = atoi(e) != 0;    // measurement knob
    if (const char* e = getenv("ADF_SCALED_FUSE")) h->scaled_fuse = atoi(e) != 0;  // measurement knob
    if (const char* e = getenv("ADF_LO_HALF")) h->scaled_half = atoi(e) != 0;
    if (const char* e = getenv("ADF_MERGE_SMALL")) h->merge_small = atoi(e) != 0;   // measurement knob
    if (const char* e = getenv("ADF_CONF_LDS_FLOOR_KB")) h->conf_lds_floor = (size_t)atoi(e) * 1024;
    *out = h;
    return ADF_OK;
}

extern "C" void adf_wls_destroy(adf_wls_t* h)
{
    if (!h) return;
    DeviceScope ds(h->device);
    h->lut.release(); h->ws.release(); h->conf.release(); h->stage.release(); h->scaled.release();
    h->prof.destroy();
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    delete h;
}

#define NEED_HANDLE(h) do { if (!(h)) return fail(ADF_EBADARG, "%s: handle is NULL", __func__); } while (0)

extern "C" int adf_wls_set_lambda(adf_wls_t* h, double v) { NEED_HANDLE(h); h->lambda = v; return ADF_OK; }
extern "C" int adf_wls_get_lambda(const adf_wls_t* h, double* v) { NEED_HANDLE(h); if (v) *v = h->lambda; return ADF_OK; }
extern "C" int adf_wls_set_sigma_color(adf_wls_t* h, double v) { NEED_HANDLE(h); h->sigma_color = v; return ADF_OK; }
extern "C" int adf_wls_get_sigma_color(const adf_wls_t* h, double* v) { NEED_HANDLE(h); if (v) *v = h->sigma_color; return ADF_OK; }
extern "C" int adf_wls_set_lrc_thresh(adf_wls_t* h, int v) { NEED_HANDLE(h); h->lrc_thresh = v; return ADF_OK; }
extern "C" int adf_wls_get_lrc_thresh(const adf_wls_t* h, int* v) { NEED_HANDLE(h); if (v) *v = h->lrc_thresh; return ADF_OK; }
extern "C" int adf_wls_set_depth_discontinuity_radius(adf_wls_t* h, int v) { NEED_HANDLE(h); h->disc_radius = v; return ADF_OK; }
extern "C" int adf_wls_get_depth_discontinuity_radius(const adf_wls_t* h, int* v) { NEED_HANDLE(h); if (v) *v = h->disc_radius; return ADF_OK; }

extern "C" int adf_wls_set_fgs_params(adf_wls_t* h, double atten, int num_iter)
{
    NEED_HANDLE(h);
    if (num_iter < 1) return fail(ADF_EBADARG, "num_iter must be >= 1 (FGS.cpp:143)");
    h->atten = atten; h->num_iter = num_iter;
    return ADF_OK;
}

extern "C" int adf_wls_set_solver(adf_wls_t* h, int solver)
{
    NEED_HANDLE(h);
    if (solver != ADF_SOLVER_EXACT && solver != ADF_SOLVER_WAVE) return fail(ADF_EBADARG, "unknown solver %d", solver);
    h->solver = solver;
    return ADF_OK;
}
extern "C" int adf_wls_get_solver(const adf_wls_t* h, int* v) { NEED_HANDLE(h); if (v) *v = h->solver; return ADF_OK; }
extern "C" int adf_wls_get_last_solver(const adf_wls_t* h, int* v) { NEED_HANDLE(h); if (v) *v = h->last_solver; return ADF_OK; }
extern "C" int adf_wls_get_last_path(const adf_wls_t* h, int* v) { NEED_HANDLE(h); if (v) *v = h->last_path; return ADF_OK; }

extern "C" int adf_wls_get_device(const adf_wls_t* h, int* device) { NEED_HANDLE(h); if (device) *device = h->device; return ADF_OK; }
extern "C" int adf_wls_get_roi(const adf_wls_t* h, adf_rect* roi) { NEED_HANDLE(h); if (roi) *roi = h->roi; return ADF_OK; }
extern "C" size_t adf_wls_workspace_bytes(const adf_wls_t* h)
{
    return h ? h->ws.bytes + h->conf.bytes + h->stage.bytes + h->scaled.bytes + h->lut.bytes() : 0;
}

extern "C" int adf_wls_sync(adf_wls_t* h, void* stream)
{
    NEED_HANDLE(h);
    DeviceScope ds(h->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ADF_OK;
}

// The confidence planes of a call: n_pairs planes in Geom's cpitch layout.  Whatever lies outside the ROI -- frame
// pixels and row padding alike -- must read as zero; frame pixels are written by the kernels of every call, the padding
// never is, so the buffer is cleared whenever the layout it was last used with changes.
static int ensure_conf_planes(adf_wls* h, const Geom& g, int n_pairs, hipStream_t st)
{
    int rc = h->conf.reserve(g.cframe * sizeof(float) * (size_t)n_pairs, st);
    if (rc) return rc;
    const long long sig[4] = {g.W, g.H, g.cx0, n_pairs};
    if (memcmp(sig, h->conf_sig, sizeof(sig)) != 0) {
        HIP_TRY(hipMemsetAsync(h->conf.p, 0, h->conf.bytes, st));
        memcpy(h->conf_sig, sig, sizeof(sig));
    }
    return ADF_OK;
}

static size_t wls_pair_ws_bytes(const Geom& g, bool conf, bool wave, bool disc_maps)
{
    // exact: ROI planes CH CV D F0 A0 B0 (+ F1 A1 B1 with confidence); wave: CH CV A0 (+ A1), in place
    // full frames: cL cR, the depth-discontinuity maps (not needed when the one-sweep confidence kernel runs)
    size_t planes = wave ? (conf ? 4 : 3) : (conf ? 9 : 6);
    return planes * g.plane * sizeof(float) + (conf && disc_maps ? 2 * g.frame * sizeof(float) : 0);
}

// What the down-scaled path queues where a same-size call runs its confidence kernels: the low-resolution confidence
// map and the two resizes, for ALL pairs of the call, on the caller's stream -- i.e. beside the weight kernel, which
// wls_filter_impl has forked to the side stream by then.
struct ScaledStage;
static int run_scaled_stage(const ScaledStage& s, hipStream_t st, Profiler* prof, int part);
// fills `fuse` with the low-resolution form of the fused first row pass for the chunk starting at pair `first`, when the
// stage asked for it; false = the resize kernels run
static bool scaled_fuse_lo(const ScaledStage& s, int first, const Geom& g, WavePassArgs& fuse);
static void scaled_note_lazy_conf(const ScaledStage& s);
static int scaled_resize_conf_now(const ScaledStage& s, hipStream_t st, Profiler* prof);

// conf_given: the down-scaled path (DF.cpp:274): the confidence kernels are skipped and dispR is not used; `scaled`
// (may be null) fills h->conf with the view-sized confidence planes of all pairs and produces dispL itself.
static int wls_filter_impl(adf_wls_t* h, int n_pairs,
                           const int16_t* dispL, ptrdiff_t sL, ptrdiff_t psL,
                           const uint8_t* view, ptrdiff_t sG, ptrdiff_t psG, int gch, int W, int H,
                           int16_t* out, ptrdiff_t sO, ptrdiff_t psO,
                           const int16_t* dispR, ptrdiff_t sR, ptrdiff_t psR,
                           const adf_rect* roi_in, bool conf_given, hipStream_t st, const ScaledStage* scaled = nullptr)
{
    NEED_HANDLE(h);
    // DF.cpp:221-222
    if (!dispL || W <= 0 || H <= 0) return fail(ADF_EBADARG, "disparity_map_left is empty");
    if (!view || (gch != 1 && gch != 3)) return fail(ADF_EBADARG, "left_view must be CV_8UC1 or CV_8UC3");
    if (!out) return fail(ADF_EBADARG, "filtered_disparity_map is NULL");
    if (n_pairs < 1) return fail(ADF_EBADARG, "n_pairs must be >= 1");
    if (sL < (ptrdiff_t)W * 2 || sO < (ptrdiff_t)W * 2 || sG < (ptrdiff_t)W * gch)
        return fail(ADF_ESIZE, "row stride smaller than a row");
    if (h->use_confidence && !conf_given) { // DF.cpp:262-264
        if (!dispR) return fail(ADF_EBADARG, "disparity_map_right is required with use_confidence");
        if (sR < (ptrdiff_t)W * 2) return fail(ADF_ESIZE, "right disparity stride smaller than a row");
    }
    if (h->lambda < 0 || h->sigma_color < 0) return fail(ADF_EBADARG, "lambda and sigma_color must be >= 0 (FGS.cpp:143)");
    // DF.cpp:228-233
    adf_rect roi;
    if (roi_in && roi_in->width * roi_in->height != 0) roi = *roi_in;
    else roi = adf_rect{h->left_offset, h->top_offset, W - h->left_offset - h->right_offset,
                        H - h->top_offset - h->bottom_offset};
    if (roi.width <= 0 || roi.height <= 0 || roi.x < 0 || roi.y < 0 || roi.x + roi.width > W ||
        roi.y + roi.height > H)
        return fail(ADF_ESIZE, "ROI (%d,%d,%d,%d) does not fit a %dx%d map", roi.x, roi.y, roi.width,
                    roi.height, W, H);
    if (h->use_confidence && (h->disc_radius < 0 || h->disc_radius > max_disc_radius()))
        return fail(ADF_EBADARG, "depth discontinuity radius %d outside [0,%d]", h->disc_radius, max_disc_radius());

    DeviceScope ds(h->device);
    const Geom g = make_geom(W, H, roi.x, roi.y, roi.width, roi.height);
    h->lazy_conf.pending = false;
    h->roi = roi; h->last_W = W; h->last_H = H; h->last_pairs = n_pairs;
    h->last_cpitch = g.cpitch; h->last_cx0 = g.cx0; h->last_path = 0;

    int rc = h->lut.ensure((float)h->sigma_color, st);
    if (rc) return rc;
    const bool conf = h->use_confidence;
    // sizes outside the register-resident kernels' range fall back to the exact solver
    const bool wave = h->solver == ADF_SOLVER_WAVE && wave_fits(g);
    h->last_solver = wave ? ADF_SOLVER_WAVE : ADF_SOLVER_EXACT;
    // does the one-sweep confidence kernel run (both views' maps stay on chip)?  It needs the fused first row pass,
    // whose alignment conditions depend only on the geometry, the strides and the pointers known here.
    bool band = false;
    if (conf && !conf_given && wave && h->conf_band && conf_band_fits(g, h->disc_radius)) {
        WavePassArgs probe{};
        probe.conf_in = (const float*)h->conf.p; probe.conf_frame = g.cframe; probe.conf_pitch = g.cpitch; probe.conf_x0 = g.cx0 + roi.x; probe.conf_y0 = roi.y;
        probe.dl_in = dispL; probe.dl_stride = sL; probe.dl_pair_stride = psL; probe.dl_x0 = roi.x; probe.dl_y0 = roi.y;
        probe.len = g.rw;
        // (h->conf.p may still be null or about to be re-allocated: hipMalloc returns 256-byte aligned memory either way)
        if (!probe.conf_in) probe.conf_in = reinterpret_cast<const float*>(uintptr_t(256));
        band = wave_hpass_can_fuse(probe);
    }
    const size_t per_pair = wls_pair_ws_bytes(g, conf, wave, !band);
    int chunk = (int)(h->ws_limit / per_pair);
    if (chunk < 1) chunk = 1;
    if (chunk > n_pairs) chunk = n_pairs;
    if ((rc = h->ws.reserve(per_pair * (size_t)chunk, st))) return rc;
    if (conf && !conf_given && (rc = ensure_conf_planes(h, g, n_pairs, st))) return rc;

    {
        const long long sig[8] = {W, H, roi.x, roi.y, roi.width, roi.height, (long long)band * 4 + (long long)wave * 2 + conf, chunk};
        if (memcmp(sig, h->ws_sig, sizeof(sig)) != 0) {
            HIP_TRY(hipMemsetAsync(h->ws.p, 0, h->ws.bytes, st));
            memcpy(h->ws_sig, sig, sizeof(sig));
        }
    }
    // carve the workspace
    float* base = (float*)h->ws.p;
    auto take = [&](size_t elems) { float* p = base; base += elems * (size_t)chunk; return p; };
    SolvePlanes p{};
    p.CH = take(g.plane); p.CV = take(g.plane); p.A0 = take(g.plane);
    if (!wave) { p.D = take(g.plane); p.F0 = take(g.plane); p.B0 = take(g.plane); }
    float *cL = nullptr, *cR = nullptr;
    if (conf) {
        p.A1 = take(g.plane);                     // wave: directly behind A0 (the pair plane spans both)
        if (!wave) { p.F1 = take(g.plane); p.B1 = take(g.plane); }
        if (!band) { cL = take(g.frame); cR = take(g.frame); }
    }
    // exact: the horizontal pass wants the row index fastest (T); wave: row-major (N), except that two
    // right-hand sides share one interleaved pair plane (A0 and A1 are adjacent: 2*plane floats per
    // image starting at A0) and Cvert is strip-major -- see fgs_wave_common.h
    const int orient_h = wave ? ORIENT_N : ORIENT_T;
    const int orient_u2 = wave ? ORIENT_PAIR : ORIENT_T;     // the two right-hand sides of a confidence-mode call
    const int orient_cv = wave ? ORIENT_STRIP : ORIENT_N;

    for (int first = 0; first < n_pairs; first += chunk) {
        const int n = (n_pairs - first < chunk) ? n_pairs - first : chunk;
        const int16_t* dL = (const int16_t*)((const char*)dispL + (ptrdiff_t)first * psL);
        const uint8_t* gv = view + (ptrdiff_t)first * psG;
        int16_t* o = (int16_t*)((char*)out + (ptrdiff_t)first * psO);

        const double F = (double)g.frame * n, P = (double)g.rw * g.rh * n;
        Profiler* prof = &h->prof;
        const int16_t fill = (int16_t)(16 * (h->min_disp - 1));            // DF.cpp:254,284
        if (!conf) {                                                       // with confidence the LRC kernel fills
            OutsideArgs oa{o, sO, psO, fill, nullptr, g};
            ProfScope ps(prof, K_FILL, 2.0 * (F - P), 2.0 * (F - P), st);
            HIP_TRY(launch_outside(oa, n, st));
            if (scaled && first == 0 && (rc = run_scaled_stage(*scaled, st, prof, 1))) return rc;
        }
        WeightArgs wa{gv, sG, psG, gch, h->lut.cur, p.CH, p.CV, orient_h, orient_cv, g,
                      wave ? nullptr : p.B0};   // exact: B0 is free until the first pass writes its output there
        // confidence mode: the weights depend on the guide only and the confidence kernels on the disparity
        // maps only -- one is bound by memory latency, the others lean on the vector ALUs -- so the weight
        // kernel is forked onto the side stream and joined before the first solve pass
        // one small frame per call: weights, confidence map and fill in ONE launch on the caller's stream (no fork)
        const bool merged = band && h->merge_small && prep_small_fits(g, h->disc_radius, gch, n) && prep_small_guide_fits(g, sG, gch);
        const bool fork_weights = conf && h->overlap && !merged;
        hipStream_t wst = st;
        if (fork_weights) {
            if ((rc = h->ensure_side(st))) return rc;
            HIP_TRY(hipEventRecord(h->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            wst = h->side;
        }
        // the fill of everything outside the ROI (DF.cpp:284, :187-190) touches no pixel any other kernel of the call
        // touches: on the wave path it rides the side stream too instead of sitting between the confidence kernel and
        // the first solve pass (one dependent launch less on the critical path of a single-pair call) -- and it goes
        // FIRST there: alone it takes 0.08 ms of a 64 x 4K step on the StereoBM factory's ROI, but queued behind the
        // weight kernel it starts when the confidence kernel's workgroups hold nearly every register of every CU and
        // crawls through 0.8 ms as the call's tail (round 3)
        // (down-scaled path: the resized confidence map covers the whole frame, only the output is filled)
        const bool outside_on_side = fork_weights && wave && (conf_given || h->disc_radius <= conf_left_max_radius());
        if (outside_on_side) {
            OutsideArgs oa{o, sO, psO, fill, conf_given ? nullptr : (float*)h->conf.p + (size_t)first * g.cframe, g};
            const double ob = (conf_given ? 2.0 : 6.0) * (F - P);
            ProfScope ps(prof, K_FILL, ob, ob, wst);
            HIP_TRY(launch_outside(oa, n, wst));
        }
        if (!merged) {
            ProfScope ps(prof, K_WEIGHTS, (gch + 8.0) * P, (gch + 8.0) * P, wst);
            HIP_TRY(launch_weights(wa, n, wst));                           // FGS.cpp:163-172
        }
        if (fork_weights) HIP_TRY(hipEventRecord(h->ev_join, h->side));

        if (conf) {
            const int16_t* dRp = (const int16_t*)((const char*)dispR + (ptrdiff_t)first * psR);
            const int rrx = W - (roi.x + roi.width);                       // DF.cpp:202
            float* confp = (float*)h->conf.p + (size_t)first * g.cframe;
            const int thresh = (int)(1.0f * h->lrc_thresh);                // DF.cpp:318 (resize_factor 1)
            DiscArgs da{};
            da.disp[0] = dL; da.stride[0] = sL; da.pair_stride[0] = psL; da.rx[0] = roi.x; da.dst[0] = cL;
            da.disp[1] = dRp; da.stride[1] = sR; da.pair_stride[1] = psR; da.rx[1] = rrx; da.dst[1] = cR;
            da.ry = roi.y; da.rw = roi.width; da.rh = roi.height; da.radius = h->disc_radius;
            da.roll_off = h->roll_off; da.W = W; da.frame = g.frame; da.only_view = -1;
            WavePassArgs fuse{};                                            // inputs of a fused first pass
            if (conf_given) {
                // confidence resized to the view (DF.cpp:274): only the prologue remains (DF.cpp:286-290)
                if (!outside_on_side) {
                    OutsideArgs oa{o, sO, psO, fill, nullptr, g};
                    ProfScope ps(prof, K_FILL, 2.0 * (F - P), 2.0 * (F - P), st);
                    HIP_TRY(launch_outside(oa, n, st));                    // DF.cpp:284
                }
                const bool lo_fused = scaled && wave && scaled_fuse_lo(*scaled, first, g, fuse);
                if (scaled && first == 0 && ((rc = run_scaled_stage(*scaled, st, prof, 0)) || (!lo_fused && (rc = run_scaled_stage(*scaled, st, prof, 1))))) return rc;
                if (lo_fused) {
                    // the first row pass taps the low-resolution maps itself: no resize launch, no view-sized planes
                    h->last_path |= ADF_PATH_SCALED_FUSED;
                    if (wave_hpass_lo_half(fuse)) h->last_path |= ADF_PATH_SCALED_HALF;
                    if (first == 0) {
                        if (stream_is_capturing(st)) {
                            // a call captured into a graph is replayed without this host code: the view-sized confidence
                            // maps are made inside the call (the graph), so getConfidenceMap() stays current after replays
                            if ((rc = scaled_resize_conf_now(*scaled, st, prof))) return rc;
                        } else scaled_note_lazy_conf(*scaled);
                    }
                } else {
                fuse.conf_in = confp; fuse.conf_frame = g.cframe; fuse.conf_pitch = g.cpitch; fuse.conf_x0 = g.cx0 + roi.x; fuse.conf_y0 = roi.y;
                fuse.dl_in = dL; fuse.dl_stride = sL; fuse.dl_pair_stride = psL; fuse.dl_x0 = roi.x; fuse.dl_y0 = roi.y;
                fuse.len = g.rw;
                }
                if (!lo_fused && !(wave && wave_hpass_can_fuse(fuse))) {
                    fuse = WavePassArgs{};
                    PlainPrologueArgs pa{dL, sL, psL, ADF_16S, 1, 0, p.A0, g, orient_u2, confp, p.A1};
                    ProfScope ps(prof, K_PROLOGUE, 14.0 * P, 14.0 * P, st);
                    HIP_TRY(launch_plain_prologue(pa, n, st));
                }
            } else if (wave && h->disc_radius <= conf_left_max_radius()) {
                // wave path: right map, then left map + LRC + x255 in one sweep (cL never hits memory);
                // the first horizontal pass forms conf*disp itself when alignment allows
                da.only_view = 1;
                fuse.conf_in = confp; fuse.conf_frame = g.cframe; fuse.conf_pitch = g.cpitch; fuse.conf_x0 = g.cx0 + roi.x; fuse.conf_y0 = roi.y;
                fuse.dl_in = dL; fuse.dl_stride = sL; fuse.dl_pair_stride = psL; fuse.dl_x0 = roi.x; fuse.dl_y0 = roi.y;
                fuse.len = g.rw;
                const bool fused_h = wave_hpass_can_fuse(fuse);
                if (!fused_h) fuse = WavePassArgs{};
                if (band && !fused_h) return fail(ADF_EHIP, "internal: confidence kernel selection and first-pass fusion disagree");
                if (band) h->last_path |= ADF_PATH_CONF_BAND;
                if (merged) {
                    h->last_path |= ADF_PATH_MERGED_PREP;
                    ConfBandArgs ba{dL, sL, psL, dRp, sR, psR, confp, g, rrx, thresh, h->disc_radius, h->roll_off, 0};
                    OutsideArgs oa{o, sO, psO, fill, confp, g};
                    const double b = (8.0 + gch + 8.0) * P + 6.0 * (F - P);
                    ProfScope ps(prof, K_LRC, b, b, st);
                    HIP_TRY(launch_prep_small(ba, wa, oa, n, st));         // FGS.cpp:163-172 + DF.cpp:197-210 + :284
                } else if (band) {
                    // both views, LRC and x255 in one band sweep: the right view's map lives in LDS only
                    ConfBandArgs ba{dL, sL, psL, dRp, sR, psR, confp, g, rrx, thresh, h->disc_radius, h->roll_off, 0, fork_weights ? h->conf_lds_floor : 0};
                    ProfScope ps(prof, K_LRC, 8.0 * P, 8.0 * P, st);     // dL 2 + dR 2 read, conf 4 written
                    HIP_TRY(launch_conf_band(ba, n, st));                  // DF.cpp:197-210
                } else {
                    {
                        ProfScope ps(prof, K_DISC, 2.0 * P, 6.0 * P, st);
                        HIP_TRY(launch_discontinuity(da, n, st));          // DF.cpp:204 (right view)
                    }
                    ConfLeftArgs ca{dL, sL, psL, dRp, sR, psR, cR, confp, fused_h ? nullptr : p.A0, fused_h ? nullptr : p.A1,
                                    g, rrx, thresh, h->disc_radius, h->roll_off};
                    // alg: conf (4P); moved: dL 2 + dR 2 + cR 4 reads, conf 4 (+8 when U0/U1 are materialised)
                    const double wu = fused_h ? 0.0 : 8.0;
                    ProfScope ps(prof, K_LRC, (4.0 + wu) * P, (12.0 + wu) * P, st);
                    HIP_TRY(launch_conf_left(ca, n, st));                  // DF.cpp:204-209 (+288-290)
                }
                if (!outside_on_side && !merged) {
                    OutsideArgs oa{o, sO, psO, fill, confp, g};
                    ProfScope ps(prof, K_FILL, 6.0 * (F - P), 6.0 * (F - P), st);
                    HIP_TRY(launch_outside(oa, n, st));                    // DF.cpp:284, :187-190
                }
            } else {
                {   // reads the int16 ROIs, writes the float maps (the maps themselves are not algorithmic I/O)
                    ProfScope ps(prof, K_DISC, 4.0 * P, 12.0 * P, st);
                    HIP_TRY(launch_discontinuity(da, n, st));              // DF.cpp:204
                }
                LrcArgs la{dL, sL, psL, dRp, sR, psR, cL, cR, confp, o, sO, psO, fill, p.A0, p.A1, g, rrx, thresh, orient_u2};
                {   // alg: confidence map out (4F) + the two rhs planes (8P); moved adds dL,dR,cL,cR reads
                    ProfScope ps(prof, K_LRC, 4.0 * F + 8.0 * P + 2.0 * (F - P), 4.0 * F + 20.0 * P + 2.0 * (F - P), st);
                    HIP_TRY(launch_lrc_prologue(la, n, st));               // DF.cpp:208-209,288-290
                }
            }
            if (fork_weights) HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
            if (wave && (fuse.conf_in || fuse.lo_conf)) h->last_path |= ADF_PATH_FUSED_FIRST_PASS;
            FinalOut fo{EPI_WLS_CONF, o, sO, psO, roi.x, roi.y, 1, 0};
            rc = wave ? run_passes_wave(g, p, 2, (float)h->lambda, (float)h->atten, h->num_iter, fo, n, st, prof,
                                        (fuse.conf_in || fuse.lo_conf) ? &fuse : nullptr)
                      : run_passes_exact(g, p, 2, (float)h->lambda, (float)h->atten, h->num_iter, fo, n, st, prof);
            if (rc) return rc;                                             // DF.cpp:292-296
        } else {
            PlainPrologueArgs pa{dL, sL, psL, ADF_16S, 1, 0, p.A0, g, orient_h};
            {
                ProfScope ps(prof, K_PROLOGUE, 6.0 * P, 6.0 * P, st);
                HIP_TRY(launch_plain_prologue(pa, n, st));                 // FGS.cpp:203-205
            }
            FinalOut fo{EPI_I16, o, sO, psO, roi.x, roi.y, 1, 0};
            rc = wave ? run_passes_wave(g, p, 1, (float)h->lambda, (float)h->atten, h->num_iter, fo, n, st, prof)
                      : run_passes_exact(g, p, 1, (float)h->lambda, (float)h->atten, h->num_iter, fo, n, st, prof);
            if (rc) return rc;                                             // DF.cpp:257-258
        }
    }
    return ADF_OK;
}

extern "C" int adf_wls_filter_device(adf_wls_t* h, int n_pairs,
                                     const int16_t* dispL, ptrdiff_t sL, ptrdiff_t psL,
                                     const uint8_t* view, ptrdiff_t sG, ptrdiff_t psG, int gch, int W, int H,
                                     int16_t* out, ptrdiff_t sO, ptrdiff_t psO,
                                     const int16_t* dispR, ptrdiff_t sR, ptrdiff_t psR,
                                     const adf_rect* roi_in, void* stream)
{
    return wls_filter_impl(h, n_pairs, dispL, sL, psL, view, sG, psG, gch, W, H, out, sO, psO, dispR, sR, psR, roi_in,
                           false, (hipStream_t)stream);
}

struct ScaledStage {
    adf_wls_t* h; int n_pairs;
    const int16_t* dispL; ptrdiff_t sL, psL;
    const int16_t* dispR; ptrdiff_t sR, psR;
    int dW, dH, W, H;
    adf_rect rlo; Geom ghi;
    float resize_factor, x_ratio;
    char* dhi; size_t dhi_bytes;
    float *cl, *cr, *clo;
    bool conf;
    bool fuse_lo;   // the first row pass interpolates (decided by adf_wls_filter_scaled_device: dhi is not allocated then)
    float* taps;    // ... with this scratch for the columns' source coordinates
};

static bool scaled_band_map(const ScaledStage& s)
{
    return s.conf && s.h->conf_band &&
           conf_band_fits(plain_conf_layout(make_geom(s.dW, s.dH, s.rlo.x, s.rlo.y, s.rlo.width, s.rlo.height)), s.h->disc_radius);
}

static void scaled_lo_args(const ScaledStage& s, int first, const Geom& g, WavePassArgs& f)
{
    const size_t lo = (size_t)s.dW * s.dH;
    f = WavePassArgs{};
    f.lo_conf = s.clo + (size_t)first * lo; f.lo_conf_stride = s.dW; f.lo_conf_pair = (ptrdiff_t)lo;
    f.lo_dl = (const int16_t*)((const char*)s.dispL + (ptrdiff_t)first * s.psL); f.lo_dl_stride = s.sL; f.lo_dl_pair = s.psL;
    f.lo_w = s.dW; f.lo_h = s.dH; f.hi_x0 = g.rx; f.hi_y0 = g.ry;
    f.lo_scale_x = (double)s.dW / s.W; f.lo_scale_y = (double)s.dH / s.H; f.lo_post_scale = s.x_ratio;
    if (scaled_band_map(s)) {
        f.lo_zero_outside = 1; f.lo_vx0 = s.rlo.x; f.lo_vy0 = s.rlo.y; f.lo_vx1 = s.rlo.x + s.rlo.width; f.lo_vy1 = s.rlo.y + s.rlo.height;
    }
    f.len = g.rw; f.lo_taps = s.taps; f.lo_half = s.h->scaled_half ? 1 : 0;
}

static bool scaled_fuse_lo(const ScaledStage& s, int first, const Geom& g, WavePassArgs& fuse)
{
    if (!s.fuse_lo) return false;
    scaled_lo_args(s, first, g, fuse);
    return true;     // (adf_wls_filter_scaled_device has checked wave_hpass_can_fuse_lo on the same arguments)
}

static void scaled_note_lazy_conf(const ScaledStage& s)
{
    adf_wls::LazyConf& z = s.h->lazy_conf;
    z.pending = true; z.clo = s.clo; z.dW = s.dW; z.dH = s.dH; z.rlo = s.rlo; z.ghi = s.ghi; z.band_map = scaled_band_map(s); z.n_pairs = s.n_pairs;
}

// cv::resize of the low-resolution confidence maps into the handle's view-sized planes (DF.cpp:274)
static int resize_conf_planes(adf_wls_t* h, const float* clo, int dW, int dH, const adf_rect& rlo, const Geom& ghi, bool band_map,
                              int n_pairs, hipStream_t st, Profiler* prof)
{
    const size_t lo = (size_t)dW * dH;
    const double Fhi = (double)ghi.W * ghi.H * n_pairs;
    ResizeArgs rc32{clo, (ptrdiff_t)dW * 4, (ptrdiff_t)(lo * 4), dW, dH, (float*)h->conf.p + ghi.cx0, (ptrdiff_t)ghi.cpitch * 4,
                    (ptrdiff_t)(ghi.cframe * 4), ghi.W, ghi.H, (double)dW / ghi.W, (double)dH / ghi.H, 1.0f, 0};
    if (band_map) { rc32.zero_outside = 1; rc32.vx0 = rlo.x; rc32.vy0 = rlo.y; rc32.vx1 = rlo.x + rlo.width; rc32.vy1 = rlo.y + rlo.height; }
    ProfScope ps(prof, K_RESIZE, 4.0 * Fhi + 4.0 * (double)lo * n_pairs, 4.0 * Fhi + 4.0 * (double)lo * n_pairs, st);
    HIP_TRY(launch_resize_linear(rc32, n_pairs, st));
    return ADF_OK;
}

static int scaled_resize_conf_now(const ScaledStage& s, hipStream_t st, Profiler* prof)
{
    return resize_conf_planes(s.h, s.clo, s.dW, s.dH, s.rlo, s.ghi, scaled_band_map(s), s.n_pairs, st, prof);
}

// part 0: the low-resolution confidence map; part 1: the two resizes.  Both are queued beside the weight kernel (after
// its fork).  Measured at 64 x 4K views / 1080p maps: everything after the fork 13.8-14.1 / 13.8-14.0 ms per call (radius 2 / 5;
// the band kernel crawls beside the weight kernel's small workgroups, 0.44 -> 0.9-1.7 ms, but the resizes then run
// alone), part 0 before the fork 14.20 / 14.51 (two memory-bound kernels side by side gain nothing), no overlap at all
// 14.49 / 14.69.
static int run_scaled_stage(const ScaledStage& s, hipStream_t st, Profiler* prof, int part)
{
    adf_wls_t* h = s.h;
    const int n_pairs = s.n_pairs, dW = s.dW, dH = s.dH, W = s.W, H = s.H;
    const size_t lo = (size_t)dW * dH;
    const adf_rect& rlo = s.rlo;
    const double Plo = (double)rlo.width * rlo.height * n_pairs, Fhi = (double)W * H * n_pairs;
    const bool band_map = s.conf && h->conf_band && conf_band_fits(plain_conf_layout(make_geom(dW, dH, rlo.x, rlo.y, rlo.width, rlo.height)), h->disc_radius);
    if (s.conf && part == 0) {
        const Geom glo = plain_conf_layout(make_geom(dW, dH, rlo.x, rlo.y, rlo.width, rlo.height));
        const int rrx = dW - (rlo.x + rlo.width);                          // DF.cpp:202
        DiscArgs da{};
        da.disp[0] = s.dispL; da.stride[0] = s.sL; da.pair_stride[0] = s.psL; da.rx[0] = rlo.x; da.dst[0] = s.cl;
        da.disp[1] = s.dispR; da.stride[1] = s.sR; da.pair_stride[1] = s.psR; da.rx[1] = rrx; da.dst[1] = s.cr;
        da.ry = rlo.y; da.rw = rlo.width; da.rh = rlo.height; da.radius = h->disc_radius;
        da.roll_off = h->roll_off / (s.resize_factor * s.resize_factor);    // DF.cpp:359
        da.W = dW; da.frame = lo; da.only_view = -1;
        const int thresh_lo = (int)(s.resize_factor * h->lrc_thresh);       // DF.cpp:318
        if (band_map) {
            // the one-sweep kernel at the maps' resolution: ROI pixels from the band kernel, zeros outside (DF.cpp:187-190)
            ConfBandArgs ba{s.dispL, s.sL, s.psL, s.dispR, s.sR, s.psR, s.clo, glo, rrx, thresh_lo, h->disc_radius, da.roll_off, 0};
            h->last_path |= ADF_PATH_CONF_BAND;
            {
                ProfScope ps(prof, K_LRC, 8.0 * Plo, 8.0 * Plo, st);
                HIP_TRY(launch_conf_band(ba, n_pairs, st));                // DF.cpp:197-210
            }
            // (zeros outside the ROI: the resize's window)
        } else {
            {
                ProfScope ps(prof, K_DISC, 4.0 * Plo, 12.0 * Plo, st);
                HIP_TRY(launch_discontinuity(da, n_pairs, st));            // DF.cpp:204
            }
            LrcArgs la{s.dispL, s.sL, s.psL, s.dispR, s.sR, s.psR, s.cl, s.cr, s.clo, nullptr, 0, 0, 0, nullptr, nullptr, glo, rrx, thresh_lo, ORIENT_N};
            ProfScope ps(prof, K_LRC, 4.0 * (double)lo * n_pairs, 4.0 * (double)lo * n_pairs + 12.0 * Plo, st);
            HIP_TRY(launch_lrc_prologue(la, n_pairs, st));                 // DF.cpp:208-209
        }
    }
    if (part == 0) return ADF_OK;
    if (s.conf) {
        int rc = resize_conf_planes(h, s.clo, dW, dH, rlo, s.ghi, band_map, n_pairs, st, prof);   // DF.cpp:274
        if (rc) return rc;
    }
    ResizeArgs r16{s.dispL, s.sL, s.psL, dW, dH, s.dhi, (ptrdiff_t)W * 2, (ptrdiff_t)s.dhi_bytes, W, H, (double)dW / W, (double)dH / H, s.x_ratio, 1};
    ProfScope ps(prof, K_RESIZE, 2.0 * Fhi + 2.0 * (double)lo * n_pairs, 2.0 * Fhi + 2.0 * (double)lo * n_pairs, st);
    HIP_TRY(launch_resize_linear(r16, n_pairs, st));                       // DF.cpp:243-244, 272-273
    return ADF_OK;
}

// Down-scaled disparity path (DF.cpp:224-227, 239-247, 268-277): disparity maps of dW x dH, view and
// output of W x H.  ROI is in disparity-map coordinates, like the reference's.
extern "C" int adf_wls_filter_scaled_device(adf_wls_t* h, int n_pairs,
                                            const int16_t* dispL, ptrdiff_t sL, ptrdiff_t psL, int dW, int dH,
                                            const uint8_t* view, ptrdiff_t sG, ptrdiff_t psG, int gch, int W, int H,
                                            int16_t* out, ptrdiff_t sO, ptrdiff_t psO,
                                            const int16_t* dispR, ptrdiff_t sR, ptrdiff_t psR,
                                            const adf_rect* roi_in, void* stream)
{
    NEED_HANDLE(h);
    hipStream_t st = (hipStream_t)stream;
    if (dW == W && dH == H)                                                // DF.cpp:224-227: same size, resize_factor 1
        return wls_filter_impl(h, n_pairs, dispL, sL, psL, view, sG, psG, gch, W, H, out, sO, psO, dispR, sR, psR, roi_in, false, st);
    if (!dispL || dW <= 0 || dH <= 0 || W <= 0 || H <= 0) return fail(ADF_EBADARG, "disparity_map_left is empty");
    if (n_pairs < 1) return fail(ADF_EBADARG, "n_pairs must be >= 1");
    if (sL < (ptrdiff_t)dW * 2) return fail(ADF_ESIZE, "row stride smaller than a row");
    const bool conf = h->use_confidence;
    if (conf) {
        if (!dispR) return fail(ADF_EBADARG, "disparity_map_right is required with use_confidence");
        if (sR < (ptrdiff_t)dW * 2) return fail(ADF_ESIZE, "right disparity stride smaller than a row");
        if (h->disc_radius < 0 || h->disc_radius > max_disc_radius())
            return fail(ADF_EBADARG, "depth discontinuity radius %d outside [0,%d]", h->disc_radius, max_disc_radius());
    }
    adf_rect rlo;                                                          // DF.cpp:228-233, disparity-map coordinates
    if (roi_in && roi_in->width * roi_in->height != 0) rlo = *roi_in;
    else rlo = adf_rect{h->left_offset, h->top_offset, dW - h->left_offset - h->right_offset, dH - h->top_offset - h->bottom_offset};
    if (rlo.width <= 0 || rlo.height <= 0 || rlo.x < 0 || rlo.y < 0 || rlo.x + rlo.width > dW || rlo.y + rlo.height > dH)
        return fail(ADF_ESIZE, "ROI (%d,%d,%d,%d) does not fit a %dx%d map", rlo.x, rlo.y, rlo.width, rlo.height, dW, dH);
    const float resize_factor = dW / (float)W;                             // DF.cpp:225
    const float x_ratio = W / (float)dW, y_ratio = H / (float)dH;          // DF.cpp:241-242,270-271
    adf_rect rhi{(int)(rlo.x * x_ratio), (int)(rlo.y * y_ratio), (int)(rlo.width * x_ratio), (int)(rlo.height * y_ratio)};
    if (rhi.width <= 0 || rhi.height <= 0 || rhi.x + rhi.width > W || rhi.y + rhi.height > H)
        return fail(ADF_ESIZE, "scaled ROI (%d,%d,%d,%d) does not fit the %dx%d view", rhi.x, rhi.y, rhi.width, rhi.height, W, H);

    DeviceScope ds(h->device);
    h->lazy_conf.pending = false;    // (the previous call's low-resolution maps are about to be overwritten or freed)
    const size_t lo = (size_t)dW * dH, hi = (size_t)W * H;
    const Geom ghi = make_geom(W, H, rhi.x, rhi.y, rhi.width, rhi.height);   // the geometry wls_filter_impl will derive
    // Can the first row pass interpolate the maps itself (fgs_wave_h.hip, FUSE_LO)?  Confidence mode on the wave solver,
    // scale factors within the staging buffer's reach.  Then neither the resized disparity map nor -- until
    // getConfidenceMap() asks for it -- the resized confidence map is ever written.
    bool fuse_lo = false;
    if (conf && h->scaled_fuse && h->solver == ADF_SOLVER_WAVE && wave_fits(ghi)) {
        ScaledStage probe{h, n_pairs, dispL, sL, psL, dispR, sR, psR, dW, dH, W, H, rlo, ghi, resize_factor, x_ratio,
                          nullptr, 0, nullptr, nullptr, reinterpret_cast<float*>(uintptr_t(256)), conf, true, nullptr};
        WavePassArgs f;
        scaled_lo_args(probe, 0, ghi, f);
        fuse_lo = wave_hpass_can_fuse_lo(f);
    }
    // scratch: resized disparity (int16, view size; not with fuse_lo) + low-resolution cL, cR, conf (float)
    const size_t dhi_bytes = fuse_lo ? 0 : (hi * 2 + 255) / 256 * 256;
    const size_t maps_bytes = ((size_t)n_pairs * (dhi_bytes + (conf ? 3 * lo * sizeof(float) : 0)) + 255) / 256 * 256;
    const size_t need = maps_bytes + (fuse_lo ? 2 * ((size_t)rhi.width + 4) * sizeof(float) : 0);   // + the columns' taps
    int rc = h->scaled.reserve(need, st);
    if (rc) return rc;
    char* dhi = (char*)h->scaled.p;
    float* cl = (float*)(dhi + (size_t)n_pairs * dhi_bytes);
    float* cr = cl + (size_t)n_pairs * lo;
    float* clo = cr + (size_t)n_pairs * lo;
    float* taps = fuse_lo ? (float*)((char*)h->scaled.p + maps_bytes) : nullptr;
    if (conf && (rc = ensure_conf_planes(h, ghi, n_pairs, st))) return rc;
    ScaledStage stage{h, n_pairs, dispL, sL, psL, dispR, sR, psR, dW, dH, W, H, rlo, ghi, resize_factor, x_ratio,
                      dhi, dhi_bytes, cl, cr, clo, conf, fuse_lo, taps};
    // (without confidence the stage is the disparity resize alone; either way wls_filter_impl queues it after it has
    // forked the weight kernel, which needs the view only)
    // (fuse_lo: wls_filter_impl never dereferences its dispL -- the caller's low-resolution map stands in, with its own strides)
    rc = fuse_lo ? wls_filter_impl(h, n_pairs, dispL, (ptrdiff_t)W * 2, 0, view, sG, psG, gch, W, H,
                                   out, sO, psO, nullptr, 0, 0, &rhi, conf, st, &stage)
                 : wls_filter_impl(h, n_pairs, (const int16_t*)dhi, (ptrdiff_t)W * 2, (ptrdiff_t)dhi_bytes, view, sG, psG, gch, W, H,
                                   out, sO, psO, nullptr, 0, 0, &rhi, conf, st, &stage);
    h->roi = rlo;                                                          // getROI(): valid_disp_ROI (DF.cpp:139)
    return rc;
}

extern "C" int adf_wls_filter_scaled_host(adf_wls_t* h, int n_pairs,
                                          const int16_t* dispL, ptrdiff_t sL, ptrdiff_t psL, int dW, int dH,
                                          const uint8_t* view, ptrdiff_t sG, ptrdiff_t psG, int gch, int W, int H,
                                          int16_t* out, ptrdiff_t sO, ptrdiff_t psO,
                                          const int16_t* dispR, ptrdiff_t sR, ptrdiff_t psR,
                                          const adf_rect* roi)
{
    NEED_HANDLE(h);
    if (!dispL || !view || !out || W <= 0 || H <= 0 || dW <= 0 || dH <= 0 || n_pairs < 1)
        return fail(ADF_EBADARG, "adf_wls_filter_host: empty input");
    if (gch != 1 && gch != 3) return fail(ADF_EBADARG, "left_view must be CV_8UC1 or CV_8UC3");
    if (h->use_confidence && !dispR) return fail(ADF_EBADARG, "disparity_map_right is required with use_confidence");
    DeviceScope ds(h->device);
    hipStream_t st = nullptr;
    // dense device copies: [dispL | dispR | out | view] per batch
    const size_t dbytes = (size_t)dW * dH * 2, obytes = (size_t)W * H * 2, gbytes = (size_t)W * H * gch;
    const size_t dpad = (dbytes + 255) / 256 * 256, opad = (obytes + 255) / 256 * 256, gpad = (gbytes + 255) / 256 * 256;
    const size_t need = (size_t)n_pairs * (2 * dpad + opad + gpad);
    int rc = h->stage.reserve(need, st);
    if (rc) return rc;
    char* dLd = (char*)h->stage.p;
    char* dRd = dLd + (size_t)n_pairs * dpad;
    char* od = dRd + (size_t)n_pairs * dpad;
    char* gd = od + (size_t)n_pairs * opad;
    for (int k = 0; k < n_pairs; k++) {
        HIP_TRY(hipMemcpy2DAsync(dLd + k * dpad, (size_t)dW * 2, (const char*)dispL + (ptrdiff_t)k * psL, sL,
                                 (size_t)dW * 2, dH, hipMemcpyHostToDevice, st));
        if (dispR)
            HIP_TRY(hipMemcpy2DAsync(dRd + k * dpad, (size_t)dW * 2, (const char*)dispR + (ptrdiff_t)k * psR, sR,
                                     (size_t)dW * 2, dH, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(gd + k * gpad, (size_t)W * gch, view + (ptrdiff_t)k * psG, sG,
                                 (size_t)W * gch, H, hipMemcpyHostToDevice, st));
    }
    rc = adf_wls_filter_scaled_device(h, n_pairs, (const int16_t*)dLd, (ptrdiff_t)dW * 2, (ptrdiff_t)dpad, dW, dH,
                                      (const uint8_t*)gd, (ptrdiff_t)W * gch, (ptrdiff_t)gpad, gch, W, H,
                                      (int16_t*)od, (ptrdiff_t)W * 2, (ptrdiff_t)opad,
                                      dispR ? (const int16_t*)dRd : nullptr, (ptrdiff_t)dW * 2, (ptrdiff_t)dpad, roi, st);
    if (rc) return rc;
    for (int k = 0; k < n_pairs; k++)
        HIP_TRY(hipMemcpy2DAsync((char*)out + (ptrdiff_t)k * psO, sO, od + k * opad, (size_t)W * 2,
                                 (size_t)W * 2, H, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ADF_OK;
}

extern "C" int adf_wls_filter_host(adf_wls_t* h, int n_pairs,
                                   const int16_t* dispL, ptrdiff_t sL, ptrdiff_t psL,
                                   const uint8_t* view, ptrdiff_t sG, ptrdiff_t psG, int gch, int W, int H,
                                   int16_t* out, ptrdiff_t sO, ptrdiff_t psO,
                                   const int16_t* dispR, ptrdiff_t sR, ptrdiff_t psR,
                                   const adf_rect* roi)
{
    return adf_wls_filter_scaled_host(h, n_pairs, dispL, sL, psL, W, H, view, sG, psG, gch, W, H, out, sO, psO,
                                      dispR, sR, psR, roi);
}

extern "C" int adf_wls_profile_enable(adf_wls_t* h, int on)
{
    NEED_HANDLE(h);
    DeviceScope ds(h->device);
    h->prof.clear();
    h->prof.on = on != 0;
    return ADF_OK;
}

extern "C" int adf_wls_profile_read(adf_wls_t* h, adf_kernel_time* out, int capacity, int* count)
{
    NEED_HANDLE(h);
    if (!out || !count || capacity < 1) return fail(ADF_EBADARG, "adf_wls_profile_read: bad output buffer");
    DeviceScope ds(h->device);
    adf_kernel_time acc[K_COUNT];
    memset(acc, 0, sizeof(acc));
    for (int k = 0; k < K_COUNT; k++) snprintf(acc[k].name, sizeof(acc[k].name), "%s", kclass_names[k]);
    for (auto& r : h->prof.recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        acc[r.cls].launches++; acc[r.cls].total_ms += ms; acc[r.cls].alg_bytes += r.alg; acc[r.cls].moved_bytes += r.moved;
    }
    int n = 0;
    for (int k = 0; k < K_COUNT && n < capacity; k++)
        if (acc[k].launches) out[n++] = acc[k];
    *count = n;
    return ADF_OK;
}

static int conf_copy(adf_wls_t* h, int pair, float* dst, ptrdiff_t stride, hipMemcpyKind kind, hipStream_t st)
{
    if (!dst) return fail(ADF_EBADARG, "confidence destination is NULL");
    if (!h->use_confidence || !h->conf.p || h->last_pairs == 0)
        return fail(ADF_EBADARG, "no confidence map: filter() has not run with use_confidence");
    if (pair < 0 || pair >= h->last_pairs) return fail(ADF_EBADARG, "pair %d out of range [0,%d)", pair, h->last_pairs);
    if (stride < (ptrdiff_t)h->last_W * 4) return fail(ADF_ESIZE, "confidence stride smaller than a row");
    if (h->lazy_conf.pending) {
        // a down-scaled call whose first row pass interpolated the maps itself: resize the low-resolution confidence
        // maps of the call now (DF.cpp:274; all pairs, one launch, outside the filter call), once
        const adf_wls::LazyConf& z = h->lazy_conf;
        int rc = resize_conf_planes(h, z.clo, z.dW, z.dH, z.rlo, z.ghi, z.band_map, z.n_pairs, st, nullptr);
        if (rc) return rc;
        h->lazy_conf.pending = false;
    }
    const float* src = (const float*)h->conf.p + (size_t)pair * h->last_cpitch * h->last_H + h->last_cx0;
    HIP_TRY(hipMemcpy2DAsync(dst, stride, src, (size_t)h->last_cpitch * 4, (size_t)h->last_W * 4, h->last_H, kind, st));
    return ADF_OK;
}

extern "C" int adf_wls_get_confidence_device(adf_wls_t* h, int pair, float* dst, ptrdiff_t stride, void* stream)
{
    NEED_HANDLE(h);
    DeviceScope ds(h->device);
    return conf_copy(h, pair, dst, stride, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

extern "C" int adf_wls_get_confidence_host(adf_wls_t* h, int pair, float* dst, ptrdiff_t stride)
{
    NEED_HANDLE(h);
    DeviceScope ds(h->device);
    int rc = conf_copy(h, pair, dst, stride, hipMemcpyDeviceToHost, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return ADF_OK;
}

// ----------------------------------------------------------------------------------------------
// FastGlobalSmootherFilter
// ----------------------------------------------------------------------------------------------
struct adf_fgs {
    int device = 0;
    int w = 0, h = 0;
    float lambda = 0, sigma = 0, atten = 0.25f; int num_iter = 3; int solver = ADF_SOLVER_EXACT;
    Geom g{};
    Lut lut;
    // one device block (BlockCache): the planes CH CV D F0 A0 B0, then the src / dst image staging
    void* block = nullptr; size_t block_bytes = 0;
    DevBuf planes, io;            // views into `block` (never released on their own)
    // Recorded behind the last thing the handle queued, on whatever stream that was.  Every call first makes its stream
    // wait for it: a filter call overwrites `io` (where a device-guide create staged the guide for its weight kernel,
    // and where the previous call's result may still be copied out on another stream).  At destruction the block goes
    // to the cache together with this event.
    hipEvent_t busy = nullptr;
    bool in_capture = false;      // a call was captured into a graph at some point: replays may be in flight that `busy` does not cover
};

// A call that is being CAPTURED into a graph neither waits for nor records the handle's event (an event recorded
// outside the capture has no place inside it, and one recorded inside is a graph node, not a marker on a stream);
// every other call does both -- also after a capture: the state is not latched (round 4).  `in_capture` only remembers
// that replays the library cannot see may exist, for adf_fgs_destroy.  include/adf_wls.h: a captured handle is used
// on ONE stream.
static int fgs_begin(adf_fgs* f, hipStream_t st)
{
    if (f->busy && !stream_is_capturing(st)) HIP_TRY(hipStreamWaitEvent(st, f->busy, 0));
    return ADF_OK;
}

static int fgs_end(adf_fgs* f, hipStream_t st)
{
    if (stream_is_capturing(st)) { f->in_capture = true; return ADF_OK; }
    if (f->busy) HIP_TRY(hipEventRecord(f->busy, st));
    return ADF_OK;
}

// guide_on_device: `guide` is a HIP device pointer (copied into the handle on `st`, no host round trip).
static int fgs_create_impl(adf_fgs_t** out, const uint8_t* guide, ptrdiff_t gstride, int gch, int w, int hgt,
                           double lambda, double sigma_color, double atten, int num_iter, int solver,
                           bool guide_on_device, hipStream_t st)
{
    if (!out) return fail(ADF_EBADARG, "adf_fgs_create: out is NULL");
    *out = nullptr;
    // FGS.cpp:143-144
    if (!guide || w <= 0 || hgt <= 0) return fail(ADF_EBADARG, "guide is empty");
    if (lambda < 0 || sigma_color < 0 || num_iter < 1) return fail(ADF_EBADARG, "lambda>=0, sigma_color>=0, num_iter>=1 required");
    if (gch != 1 && gch != 3) return fail(ADF_EBADARG, "guide must be CV_8UC1 or CV_8UC3");
    if (gstride < (ptrdiff_t)w * gch) return fail(ADF_ESIZE, "guide stride smaller than a row");
    if (solver != ADF_SOLVER_EXACT && solver != ADF_SOLVER_WAVE) return fail(ADF_EBADARG, "unknown solver %d", solver);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ADF_ENODEV, "adf_fgs_create: no HIP device visible");
    adf_fgs* f = new (std::nothrow) adf_fgs();
    if (!f) return fail(ADF_ENOMEM, "out of host memory");
    hipGetDevice(&f->device);
    f->w = w; f->h = hgt;
    f->lambda = (float)lambda; f->sigma = (float)sigma_color; f->atten = (float)atten; // FGS.cpp:145-147
    f->num_iter = num_iter;
    f->g = make_geom(w, hgt, 0, 0, w, hgt);
    f->solver = (solver == ADF_SOLVER_WAVE && wave_fits(f->g)) ? ADF_SOLVER_WAVE : ADF_SOLVER_EXACT;
    int rc = f->lut.ensure(f->sigma, st);
    if (rc) { adf_fgs_destroy(f); return rc; }
    const size_t gbytes = (size_t)w * hgt * gch;
    const size_t planes_bytes = (6 * f->g.plane * sizeof(float) + 255) / 256 * 256;
    const size_t io_bytes = ((gbytes > (size_t)w * hgt * 16 ? gbytes : (size_t)w * hgt * 16) + 255) / 256 * 256;
    hipError_t e = hipSuccess;
    f->block = BlockCache::get().take(f->device, planes_bytes + io_bytes, st, &f->block_bytes);
    if (!f->block) {
        e = adf::device_malloc(&f->block, planes_bytes + io_bytes);  // (clears the cache and retries when the driver refuses)
        if (e != hipSuccess) {
            f->block = nullptr; adf_fgs_destroy(f);
            return fail(e == hipErrorOutOfMemory ? ADF_ENOMEM : ADF_EHIP, "adf_fgs_create: %s", hipGetErrorString(e));
        }
        f->block_bytes = planes_bytes + io_bytes;
    }
    f->planes.p = f->block; f->planes.bytes = planes_bytes;
    f->io.p = (char*)f->block + planes_bytes; f->io.bytes = io_bytes;
    // deterministic padding lanes: the sweeps read (and discard) pitch padding
    e = hipMemsetAsync(f->block, 0, planes_bytes + io_bytes, st);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->busy, hipEventDisableTiming);
    if (e == hipSuccess) e = guide_on_device
        ? hipMemcpy2DAsync(f->io.p, (size_t)w * gch, guide, gstride, (size_t)w * gch, hgt, hipMemcpyDeviceToDevice, st)
        : hipMemcpy2D(f->io.p, (size_t)w * gch, guide, gstride, (size_t)w * gch, hgt, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        float* base = (float*)f->planes.p;
        WeightArgs wa{(const uint8_t*)f->io.p, (ptrdiff_t)w * gch, 0, gch, f->lut.cur,
                      base, base + f->g.plane, f->solver == ADF_SOLVER_WAVE ? ORIENT_N : ORIENT_T,
                      f->solver == ADF_SOLVER_WAVE ? ORIENT_STRIP : ORIENT_N, f->g,
                      f->solver == ADF_SOLVER_WAVE ? nullptr : base + 5 * f->g.plane};   // B0
        e = launch_weights(wa, 1, st);
    }
    // host guide: the weights are finished when create returns, like the reference's init (FGS.cpp:163-172);
    // device guide: they are queued on `st`; an event behind them orders every later filter call -- whatever stream it
    // is on -- after the kernel that still reads the staged guide
    if (e == hipSuccess && !guide_on_device) e = hipStreamSynchronize(st);
    // (a failed create may have queued work on `st` that the event was never recorded behind: drain it before the block
    // goes back to the cache)
    if (e != hipSuccess) { hipStreamSynchronize(st); adf_fgs_destroy(f); return fail(ADF_EHIP, "adf_fgs_create: %s", hipGetErrorString(e)); }
    if ((rc = fgs_end(f, st))) { hipStreamSynchronize(st); adf_fgs_destroy(f); return rc; }
    *out = f;
    return ADF_OK;
}

extern "C" int adf_fgs_create(adf_fgs_t** out, const uint8_t* guide, ptrdiff_t gstride, int gch, int w, int hgt,
                              double lambda, double sigma_color, double atten, int num_iter, int solver)
{
    return fgs_create_impl(out, guide, gstride, gch, w, hgt, lambda, sigma_color, atten, num_iter, solver, false, nullptr);
}

extern "C" int adf_fgs_create_device(adf_fgs_t** out, const uint8_t* guide, ptrdiff_t gstride, int gch, int w, int hgt,
                                     double lambda, double sigma_color, double atten, int num_iter, int solver, void* stream)
{
    return fgs_create_impl(out, guide, gstride, gch, w, hgt, lambda, sigma_color, atten, num_iter, solver, true,
                           (hipStream_t)stream);
}

extern "C" int adf_weight_table_host(float sigma_color, float* table, int levels)
{
    if (!table || levels != ADF_LUT_LEVELS) return fail(ADF_EBADARG, "table must hold %d floats", ADF_LUT_LEVELS);
    if (!(sigma_color >= 0.0f)) return fail(ADF_EBADARG, "sigma_color must be >= 0 (FGS.cpp:143)");
    lut_build_host(sigma_color, table);
    return ADF_OK;
}

namespace adf {
hipError_t device_malloc(void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        BlockCache::get().clear();                                   // the cache may be what fills the memory
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) *p = nullptr;
    return e;
}
} // namespace adf

extern "C" void adf_release_cached_memory(void)
{
    BlockCache::get().clear();
    LutStore::get().clear();
}

extern "C" int adf_fgs_get_device(const adf_fgs_t* f, int* device) { NEED_HANDLE(f); if (device) *device = f->device; return ADF_OK; }

extern "C" void adf_fgs_destroy(adf_fgs_t* f)
{
    if (!f) return;
    DeviceScope ds(f->device);
    if (f->block) {
        if (f->busy && !f->in_capture) {
            BlockCache::get().give(f->device, f->block, f->block_bytes, f->busy);    // (the event goes with the block)
            f->busy = nullptr;
        } else {
            hipDeviceSynchronize();
            hipFree(f->block);
        }
    }
    if (f->busy) hipEventDestroy(f->busy);
    f->lut.release();
    delete f;
}

// FastGlobalSmootherFilter::filter (FGS.cpp:200-221: channels one by one) from the device image `src` into the device
// image `dst` -- the same buffer (the host path's staging area; a caller filtering in place) or two that do not overlap.
static int fgs_filter_run(adf_fgs* f, int depth, int channels, const void* src, ptrdiff_t sstride, void* dst, ptrdiff_t dstride,
                          hipStream_t st)
{
    float* base = (float*)f->planes.p;
    const Geom& g = f->g;
    SolvePlanes p{};
    p.CH = base; p.CV = base + g.plane; p.D = base + 2 * g.plane; p.F0 = base + 3 * g.plane;
    p.A0 = base + 4 * g.plane; p.B0 = base + 5 * g.plane;
    const int epi = depth == ADF_8U ? EPI_U8 : depth == ADF_16S ? EPI_I16 : EPI_F32;
    const bool wave = f->solver == ADF_SOLVER_WAVE;
    for (int c = 0; c < channels;) {
        // The reference filters the channels one by one with the same weights (FGS.cpp:200-221); the wave
        // solver takes them two at a time as the two right-hand sides of one factorisation (its pair plane
        // spans A0 and B0, which are adjacent) -- the same arithmetic per channel, half the passes.
        const int nr = (wave && c + 1 < channels) ? 2 : 1;
        PlainPrologueArgs pa{src, sstride, 0, depth, channels, c, p.A0, g,
                             wave ? (nr == 2 ? ORIENT_PAIR : ORIENT_N) : ORIENT_T};
        pa.pair2 = nr == 2; pa.c2 = c + 1;
        HIP_TRY(launch_plain_prologue(pa, 1, st));
        // the epilogue of channel c overwrites only channel c of the image, which later
        // channels never read (they read their own channel), so filtering in place is safe
        FinalOut fo{epi, dst, dstride, 0, 0, 0, channels, c};
        int rc = wave ? run_passes_wave(g, p, nr, f->lambda, f->atten, f->num_iter, fo, 1, st)
                      : run_passes_exact(g, p, 1, f->lambda, f->atten, f->num_iter, fo, 1, st);
        if (rc) return rc;
        c += nr;
    }
    return ADF_OK;
}

static int fgs_check_args(adf_fgs* f, const void* src, ptrdiff_t sstride, void* dst, ptrdiff_t dstride, int depth,
                          int channels, size_t* rowb)
{
    // FGS.cpp:184
    if (!src || !dst) return fail(ADF_EBADARG, "src/dst is empty");
    if (depth != ADF_8U && depth != ADF_16S && depth != ADF_32F) return fail(ADF_EBADARG, "src depth must be CV_8U, CV_16S or CV_32F");
    if (channels < 1 || channels > 4) return fail(ADF_EBADARG, "src must have 1..4 channels");
    const size_t esz = depth == ADF_8U ? 1 : depth == ADF_16S ? 2 : 4;
    *rowb = (size_t)f->w * channels * esz;
    if (sstride < (ptrdiff_t)*rowb || dstride < (ptrdiff_t)*rowb)
        return fail(ADF_ESIZE, "Size of the filtered image must be equal to the size of the guide image"); // FGS.cpp:187
    return ADF_OK;
}

extern "C" int adf_fgs_filter_host(adf_fgs_t* f, const void* src, ptrdiff_t sstride, void* dst, ptrdiff_t dstride,
                                   int depth, int channels)
{
    NEED_HANDLE(f);
    size_t rowb = 0;
    int rc = fgs_check_args(f, src, sstride, dst, dstride, depth, channels, &rowb);
    if (rc) return rc;
    DeviceScope ds(f->device);
    hipStream_t st = nullptr;
    if ((rc = fgs_begin(f, st))) return rc;
    HIP_TRY(hipMemcpy2DAsync(f->io.p, rowb, src, sstride, rowb, f->h, hipMemcpyHostToDevice, st));
    if ((rc = fgs_filter_run(f, depth, channels, f->io.p, (ptrdiff_t)rowb, f->io.p, (ptrdiff_t)rowb, st))) return rc;
    HIP_TRY(hipMemcpy2DAsync(dst, dstride, f->io.p, rowb, rowb, f->h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return fgs_end(f, st);
}

extern "C" int adf_fgs_filter_device(adf_fgs_t* f, const void* src, ptrdiff_t sstride, void* dst, ptrdiff_t dstride,
                                     int depth, int channels, void* stream)
{
    NEED_HANDLE(f);
    size_t rowb = 0;
    int rc = fgs_check_args(f, src, sstride, dst, dstride, depth, channels, &rowb);
    if (rc) return rc;
    DeviceScope ds(f->device);
    hipStream_t st = (hipStream_t)stream;
    // the images are filtered where they are (round 3: no staging copies): the first kernel of a channel reads `src`,
    // the last one writes `dst`; dst == src (same pointer and stride) filters in place, anything else must not overlap
    const char* s0 = (const char*)src; const char* d0 = (const char*)dst;
    const size_t sspan = (size_t)sstride * (f->h - 1) + rowb, dspan = (size_t)dstride * (f->h - 1) + rowb;
    if (!(src == dst && sstride == dstride) && s0 < d0 + dspan && d0 < s0 + sspan)
        return fail(ADF_EBADARG, "dst must be src itself (same stride) or must not overlap it");
    if ((rc = fgs_begin(f, st))) return rc;
    if ((rc = fgs_filter_run(f, depth, channels, src, sstride, dst, dstride, st))) return rc;
    return fgs_end(f, st);
}
