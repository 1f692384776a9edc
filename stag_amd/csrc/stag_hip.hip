// libstag_hip.so: C ABI (include/stag_hip.h) + host side of the MI355X classify path.
// gfx950 only; no CPU fallback: every entry point fails with STAG_E_NOGPU/STAG_E_HIP when
// the GPU is not usable.
#include "stag_hip.h"
#include "stag_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace stagk;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? STAG_E_NOMEM : STAG_E_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

// ---- the bounds-checked build (see stag_kernels.hip.h): the table of device allocations the kernels may touch.
// Persistent entries are the library's own buffers (DevBuf registers them), temporary ones the buffers of the
// call at hand (each launching helper names what its kernels are given).  Every launch of such a build is
// followed by a device synchronisation and a look at g_violation (launch_check), so the table only changes while
// nothing runs.  Fresh buffers are filled with 0xff, so that a read of memory nobody wrote shows up the same way
// on every run.  In the product build all of this is empty.
#ifdef STAG_CHECKED
struct CheckRegistry {
    std::mutex mu;
    std::map<uintptr_t, size_t> persistent;
    std::vector<std::pair<uintptr_t, size_t>> temp;
    bool dirty = true;
} g_chk;
void chk_register(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_chk.mu);
    g_chk.persistent[(uintptr_t)p] = bytes;
    g_chk.dirty = true;
}
void chk_unregister(const void *p) {
    std::lock_guard<std::mutex> lock(g_chk.mu);
    g_chk.persistent.erase((uintptr_t)p);
    g_chk.dirty = true;
}
void chk_temp_clear() {
    std::lock_guard<std::mutex> lock(g_chk.mu);
    g_chk.temp.clear();
    g_chk.dirty = true;
}
void chk_temp(const void *p, size_t bytes) {
    if (!p || !bytes) return;
    std::lock_guard<std::mutex> lock(g_chk.mu);
    g_chk.temp.emplace_back((uintptr_t)p, bytes);
    g_chk.dirty = true;
}
int chk_upload() {   // the current device's copy of the table (cheap enough to send before every launch)
    static CheckTable t;
    std::lock_guard<std::mutex> lock(g_chk.mu);
    t.n = 0;
    auto add = [&](uintptr_t a, size_t b) {
        if (t.n < (uint32_t)CHECK_MAX) { t.lo[t.n] = a; t.hi[t.n] = a + b; }
        ++t.n;
    };
    for (auto &e : g_chk.persistent) add(e.first, e.second);
    for (auto &e : g_chk.temp) add(e.first, e.second);
    if (t.n > (uint32_t)CHECK_MAX) return fail(STAG_E_INVALID, "checked build: %u allocations, the table holds %d", t.n, CHECK_MAX);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_check), &t, sizeof t));
    return STAG_OK;
}
int chk_collect(const char *what) {
    unsigned long long v[4] = {0, 0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_violation), sizeof v));
    if (!v[0]) return STAG_OK;
    const unsigned long long zero[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_violation), zero, sizeof zero));
    const long long line = (long long)v[1];
    if (line < 0)
        return fail(STAG_E_INVALID, "checked build: %s: %llu violations, the first a failed condition at stag_kernels.hip.h:%lld (values %llu, %llu)",
                    what, v[0], -line, v[2], v[3]);
    return fail(STAG_E_INVALID, "checked build: %s: %llu accesses outside every known allocation, the first at stag_kernels.hip.h:%lld, address %#llx (%llu bytes)",
                what, v[0], line, v[2], v[3]);
}
#define CHK_TEMP_CLEAR() chk_temp_clear()
#define CHK_TEMP(p, bytes) chk_temp((p), (size_t)(bytes))
#define CHK_UPLOAD()                         \
    do {                                     \
        const int rc_ = chk_upload();        \
        if (rc_) return rc_;                 \
    } while (0)
#else
#define CHK_TEMP_CLEAR() do { } while (0)
#define CHK_TEMP(p, bytes) do { } while (0)
#define CHK_UPLOAD() do { } while (0)
#endif

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    int reserve(size_t n) {
        if (n <= cap) return STAG_OK;
        release();
        HIP_TRY(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
        cap = n;
#ifdef STAG_CHECKED
        chk_register(p, std::max<size_t>(n, 1) * sizeof(T));
        HIP_TRY(hipMemset(p, 0xff, std::max<size_t>(n, 1) * sizeof(T)));
#endif
        return STAG_OK;
    }
    void release() {
#ifdef STAG_CHECKED
        if (p) chk_unregister(p);
#endif
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ProfRec {
    hipEvent_t a, b;
    int cls;
};

// Everything one call needs on the device beyond its inputs and outputs.  A handle fills it from
// its own (growing) buffers; stag_classify_a2m_noalloc_dev carves it from `scratch` below.
struct Work {
    int32_t *node, *pos, *count, *task_off, *n_tasks;
    SeqRef *order;
    LevelRec *lvl;
    Task *tasks;
    uint32_t *win;
    uint32_t *bits;                       // encode's packed rows (text entry points only)
    int32_t *enc_n_char, *enc_status;     // likewise
    bool prof;                            // bracket launches with events (allocates: never on the no-allocation path)
};

size_t walk_max_tasks(int64_t n, int max_level_nodes) { return (size_t)n / TILE + max_level_nodes + 2; }

}  // namespace

// ---- stag_classify_a2m_noalloc_dev: the workspace of callers that cannot allocate.  One copy per
// GPU, held until the process ends; in host code this NAME is only a stand-in for the symbol, the
// device address comes from hipGetSymbolAddress (scratch_of) and is what kernels and memsets get.
static __device__ __align__(256) unsigned char scratch[STAG_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {};
// the stream of the last stag_classify_a2m_noalloc_dev call per device: a call on ANOTHER stream while that one
// still has work queued would share the array with it (calls on one stream are ordered and may follow each other)
hipStream_t g_scratch_stream[MAX_DEVICES] = {};
bool g_scratch_used[MAX_DEVICES] = {};

// Device address of `scratch` on `device`, which must be the current device.  Looked up once per device.
int scratch_of(int device, unsigned char **out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(STAG_E_INVALID, "device %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(STAG_E_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
#ifdef STAG_CHECKED
        chk_register(p, STAG_NOALLOC_SCRATCH_BYTES);
#endif
    }
    *out = g_scratch_ptr[device];
    return STAG_OK;
}

// Streams and events of the host pipeline (stag_classify_a2m) come from a per-device pool and go back to it when
// their handle is destroyed; the pool never destroys them.  A HIP stream owns (a share of) a hardware queue, and
// the runtime creates and destroys that queue with the first / last stream that uses it: a process that opens
// and closes many handles (the differential fuzz: thousands) would otherwise create and destroy hardware queues
// thousands of times, which is what the GPU faults of that tool coincided with.
struct PipePool {
    std::mutex mu;
    std::vector<hipStream_t> streams[MAX_DEVICES];
    std::vector<hipEvent_t> events[MAX_DEVICES];
    hipError_t stream(int device, hipStream_t *out) {
        std::lock_guard<std::mutex> lock(mu);
        auto &v = streams[device];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    hipError_t event(int device, hipEvent_t *out) {
        std::lock_guard<std::mutex> lock(mu);
        auto &v = events[device];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        return hipEventCreateWithFlags(out, hipEventDisableTiming);
    }
    void give_back(int device, hipStream_t st) {   // st is idle
        std::lock_guard<std::mutex> lock(mu);
        streams[device].push_back(st);
    }
    void give_back(int device, hipEvent_t ev) {
        std::lock_guard<std::mutex> lock(mu);
        events[device].push_back(ev);
    }
} g_pipe_pool;
}  // namespace

constexpr int PIPE_SLOTS = 3;    // sub-batches of the host pipeline that are on the device at a time

struct stag_db {
    int device = 0;
    int n_nodes = 0, L = 0, W = 0, nchunks = 0, max_depth = 0, n_keys = 0, n_tf = 0;
    std::vector<int> level_start;  // nodes of level d are [level_start[d], level_start[d+1])
    int max_level_nodes = 1;
    int64_t table_bytes = 0;
    // tables
    DevBuf<int32_t> child_first;
    DevBuf<uint8_t> node_err;
    DevBuf<double> intercept;
    DevBuf<uint32_t> blk_off;
    DevBuf<uint4> blocks;
    DevBuf<double> tf;
    // workspace (grown by reserve)
    int64_t cap_n = 0;
    DevBuf<int32_t> node, pos, count, task_off, n_tasks;
    DevBuf<SeqRef> order;
    DevBuf<LevelRec> lvl;
    DevBuf<Task> tasks;
    DevBuf<uint32_t> bits, win;
    DevBuf<double> gappre, gapsuf;
    // tables of the matrix-core walk (k_walk_mfma): per parent node the union of its children's features per
    // chunk and the children's coefficients as signed base-256 digits of a fixed-point number
    DevBuf<uint32_t> pf_off, pfeat;
    DevBuf<int8_t> dig;
    DevBuf<uint64_t> dig_base;
    DevBuf<uint32_t> dig_ld;
    DevBuf<double> qscale;
    DevBuf<uint8_t> mfma_ok;
    std::vector<uint8_t> level_has_mfma, level_has_gather;   // per level: are there tasks for either kernel?
    int walk_mode = 0;                                         // 0 auto, 1 gather only (STAG_WALK=gather)
    int64_t parsed_rows = 0;                                   // rows stag_onehot_text_parse left in `bits`
    DevBuf<unsigned long long> dbg;                            // phase stamps of STAG_ABL == 10 builds
    DevBuf<uint8_t> text, gene_text, bt_text;
    DevBuf<int64_t> offs, gene_offs, bt_begin, bt_end;
    DevBuf<int32_t> bt_status;
    DevBuf<int32_t> enc_n_char, enc_status;
    // results (used when the caller passes NULL or host pointers)
    DevBuf<int32_t> r_path_node, r_path_len, r_lead, r_selected, r_n_aligned, r_status;
    DevBuf<double> r_path_prob, r_level_prob;
    // host pipeline of stag_classify_a2m (pipe_setup): three slots of device text / offsets / results, a copy-in,
    // a compute and a copy-out stream.  Slot j % 3 holds sub-batch j: its text goes up while sub-batch j - 1 is
    // computed and the results of j - 2 come down (with two slots the upload of j could not be queued before the
    // download of j - 2 had ended, which left the link idle between uploads).
    struct Slot {
        DevBuf<uint8_t> text;
        DevBuf<int64_t> offs;
        DevBuf<int32_t> lead_trail;
        DevBuf<int32_t> path_node, path_len, lead, selected, n_aligned, status, n_char;
        DevBuf<double> path_prob, level_prob;
        DevBuf<uint32_t> bits;
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;   // H2D finished / kernels finished / D2H finished
        bool used = false;
    } slot[PIPE_SLOTS];
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;

    void free_all() {
        child_first.release(); node_err.release(); intercept.release(); blk_off.release();
        blocks.release(); tf.release(); node.release(); pos.release(); order.release();
        pf_off.release(); pfeat.release(); dig.release(); dig_base.release(); dig_ld.release(); qscale.release(); mfma_ok.release();
        count.release(); lvl.release(); task_off.release(); n_tasks.release(); tasks.release();
        bits.release(); win.release(); gappre.release(); gapsuf.release(); text.release(); offs.release();
        gene_text.release(); bt_text.release(); gene_offs.release(); bt_begin.release(); bt_end.release(); bt_status.release(); enc_n_char.release(); enc_status.release();
        r_path_node.release(); r_path_len.release(); r_lead.release(); r_selected.release();
        r_n_aligned.release(); r_status.release(); r_path_prob.release(); r_level_prob.release();
        for (auto &r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        recs.clear();
        for (auto &sl : slot) {
            sl.text.release(); sl.offs.release(); sl.lead_trail.release(); sl.path_node.release(); sl.path_len.release(); sl.lead.release();
            sl.selected.release(); sl.n_aligned.release(); sl.status.release(); sl.n_char.release();
            sl.path_prob.release(); sl.level_prob.release(); sl.bits.release();
            for (hipEvent_t *e : {&sl.up, &sl.done, &sl.down})
                if (*e) { g_pipe_pool.give_back(device, *e); *e = nullptr; }
        }
        for (hipStream_t *st : {&s_up, &s_run, &s_down})
            if (*st) { (void)hipStreamSynchronize(*st); g_pipe_pool.give_back(device, *st); *st = nullptr; }
    }
};

namespace {

struct Prof {  // brackets one launch with events when profiling is on
    stag_db *db;
    hipStream_t st;
    ProfRec r{};
    bool on;
    Prof(stag_db *d, hipStream_t s, int cls, bool enabled = true) : db(d), st(s), on(d && d->prof && enabled) {
        if (!on) return;
        r.cls = cls;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.a, st);
    }
    ~Prof() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        db->recs.push_back(r);
    }
};

int check_device(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(STAG_E_NOGPU, "no HIP device visible (libstag_hip has no CPU path)");
    if (device < 0 || device >= cnt) return fail(STAG_E_INVALID, "device %d out of range (0..%d)", device, cnt - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(STAG_E_NOGPU, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    return STAG_OK;
}

// STAG_TRACE_FILE=<path>: every kernel launch is announced in that file (flushed) after the stream's earlier work
// has been waited for, so that after a GPU fault the file's last line names the kernel that was running.
// A debugging aid (it serialises host and device); unset, it costs one branch per launch.
FILE *g_trace = nullptr;
bool g_trace_init = false;
void trace_launch(hipStream_t st, const char *fmt, ...) {
    if (!g_trace_init) {
        g_trace_init = true;
        const char *path = getenv("STAG_TRACE_FILE");
        if (path && *path) g_trace = fopen(path, "a");
    }
    if (!g_trace) return;
    const hipError_t e = hipStreamSynchronize(st);
    fprintf(g_trace, "  earlier work: %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    va_list ap;
    va_start(ap, fmt);
    vfprintf(g_trace, fmt, ap);
    va_end(ap);
    fputc('\n', g_trace);
    fflush(g_trace);
}

int launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(STAG_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
#ifdef STAG_CHECKED
    return chk_collect(what);
#else
    return STAG_OK;
#endif
}

int encode_dev(stag_db *db_for_prof, int L, const char *d_text, int64_t text_bytes, const int64_t *d_offs, int64_t n,
               double min_perc, uint32_t *d_bits, int32_t *d_n_char, int32_t *d_n_aligned, int32_t *d_status,
               uint32_t *d_win, hipStream_t st, const int64_t *d_ends = nullptr, const int32_t *d_pre_status = nullptr,
               bool full_rows = true, const int32_t *d_lead_trail = nullptr) {
    if (n == 0) return STAG_OK;
    const int W = stag_words_per_row(L);
    // (a kernel handed a null output would write to address 0: a GPU memory fault, not an error code)
    if (!d_text || !d_offs || !d_bits || !d_n_char || !d_n_aligned || !d_status)
        return fail(STAG_E_INVALID, "encode: null buffer (text %p, offsets %p, rows %p, n_char %p, n_aligned %p, status %p)",
                    (const void *)d_text, (const void *)d_offs, (void *)d_bits, (void *)d_n_char, (void *)d_n_aligned, (void *)d_status);
    if (((uintptr_t)d_text & 15) || ((uintptr_t)d_bits & 15))
        return fail(STAG_E_INVALID, "d_text and d_bits must be 16-byte aligned");
    if (text_bytes & 15)
        return fail(STAG_E_INVALID, "text_bytes (the size of the text buffer) must be a multiple of 16: pad the buffer");
    CHK_TEMP_CLEAR();
    CHK_TEMP(d_text, text_bytes); CHK_TEMP(d_offs, (n + (d_ends ? 0 : 1)) * 8); CHK_TEMP(d_ends, n * 8);
    CHK_TEMP(d_bits, (size_t)n * W * 4); CHK_TEMP(d_n_char, n * 4); CHK_TEMP(d_n_aligned, n * 4); CHK_TEMP(d_status, n * 4);
    CHK_TEMP(d_win, n * 4); CHK_TEMP(d_pre_status, n * 4); CHK_TEMP(d_lead_trail, n * 8);
    CHK_UPLOAD();
    // lanes per record: the candidate that wastes the fewest lane slots on an average record
    // (ties go to the wider group: fewer, longer-lived rows in LDS)
    const int64_t avg = std::max<int64_t>(1, text_bytes / n);
    int G = 64;
    int64_t best_slots = -1;
    for (int g : {64, 32, 16}) {
        const int64_t slots = ((avg + 32 * g - 1) / (32 * g)) * g;      // a lane takes 32 bytes per step
        const size_t need = (size_t)4 * (64 / g) * W * sizeof(uint32_t);
        if (need > 160 * 1024) continue;
        if (best_slots < 0 || slots < best_slots) { best_slots = slots; G = g; }
    }
    if (best_slots < 0) return fail(STAG_E_INVALID, "n_features %d too large for the encode kernel's LDS row buffers", L);
    const int rpw = 64 / G;
    const size_t lds = (size_t)4 * rpw * W * sizeof(uint32_t);
    const int64_t grid = (n + 4 * rpw - 1) / (4 * rpw);
    if (grid > 0x7fffffff) return fail(STAG_E_INVALID, "batch too large");
    auto launch = [&](auto kernel) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        Prof p(db_for_prof, st, 0);
        trace_launch(st, "k_encode_a2m G=%d full=%d n=%lld L=%d W=%d text_bytes=%lld grid=%lld lds=%zu", G, (int)(full_rows || !d_win), (long long)n, L, W, (long long)text_bytes, (long long)grid, lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, st, (const uint8_t *)d_text, text_bytes,
                           d_offs, n, L, W, min_perc, d_bits, d_n_char, d_n_aligned, d_status, d_win, d_ends, d_pre_status, d_lead_trail);
        return STAG_OK;
    };
    // full_rows: the caller reads the rows (stag_encode_a2m*, bits_out); otherwise only the walk does, and it
    // needs no more than each row's window chunks
    int rc;
    if (full_rows || !d_win)
        rc = G == 64 ? launch(k_encode_a2m<64, true>) : G == 32 ? launch(k_encode_a2m<32, true>) : launch(k_encode_a2m<16, true>);
    else
        rc = G == 64 ? launch(k_encode_a2m<64, false>) : G == 32 ? launch(k_encode_a2m<32, false>) : launch(k_encode_a2m<16, false>);
    if (rc) return rc;
    return launch_check("k_encode_a2m");
}

// Resolved device pointers for one classify call.
struct Out {
    int32_t *path_node, *path_len, *lead, *selected, *n_aligned, *status;
    double *path_prob, *level_prob;
};

int reserve_results(stag_db *db, int64_t n) {
    const size_t D = std::max(db->max_depth, 1), K = std::max(db->n_keys, 1);
    int rc;
    if ((rc = db->r_path_node.reserve(n * D))) return rc;
    if ((rc = db->r_path_prob.reserve(n * D))) return rc;
    if ((rc = db->r_path_len.reserve(n))) return rc;
    if ((rc = db->r_lead.reserve(n))) return rc;
    if ((rc = db->r_level_prob.reserve(n * K))) return rc;
    if ((rc = db->r_selected.reserve(n))) return rc;
    if ((rc = db->r_n_aligned.reserve(n))) return rc;
    if ((rc = db->r_status.reserve(n))) return rc;
    return STAG_OK;
}

int reserve_walk(stag_db *db, int64_t n) {
    int rc;
    if ((rc = db->node.reserve(n))) return rc;
    if ((rc = db->pos.reserve(n))) return rc;
    const size_t max_tasks = walk_max_tasks(n, db->max_level_nodes);
    if ((rc = db->order.reserve(max_tasks * TILE))) return rc;   // 64 slots per task
    if ((rc = db->lvl.reserve((size_t)std::max(db->max_depth, 1) * n))) return rc;
    if ((rc = db->win.reserve(n))) return rc;
    if ((rc = db->tasks.reserve(max_tasks))) return rc;
    if ((rc = db->count.reserve(db->n_nodes))) return rc;
    if ((rc = db->task_off.reserve(db->n_nodes))) return rc;
    if ((rc = db->n_tasks.reserve(std::max(db->max_depth, 1)))) return rc;
#if STAG_ABL == 10
    if (!db->dbg.p) { if ((rc = db->dbg.reserve(4096 * 16))) return rc; (void)hipMemset(db->dbg.p, 0, 4096 * 16 * sizeof(unsigned long long)); }
#endif
    if ((rc = reserve_results(db, n))) return rc;
    db->cap_n = std::max(db->cap_n, n);
    return STAG_OK;
}

Work work_of(stag_db *db) {
    Work w;
    w.node = db->node.p; w.pos = db->pos.p; w.count = db->count.p; w.task_off = db->task_off.p; w.n_tasks = db->n_tasks.p;
    w.order = db->order.p; w.lvl = db->lvl.p; w.tasks = db->tasks.p; w.win = db->win.p;
    w.bits = db->bits.p; w.enc_n_char = db->enc_n_char.p; w.enc_status = db->enc_status.p;
    w.prof = true;
    return w;
}

Out resolve(stag_db *db, const stag_results_t *o) {
    Out r;
    r.path_node = o && o->path_node ? o->path_node : db->r_path_node.p;
    r.path_prob = o && o->path_prob ? o->path_prob : db->r_path_prob.p;
    r.path_len = o && o->path_len ? o->path_len : db->r_path_len.p;
    r.lead = o && o->n_lead_single ? o->n_lead_single : db->r_lead.p;
    r.level_prob = o && o->level_prob ? o->level_prob : db->r_level_prob.p;
    r.selected = o && o->selected ? o->selected : db->r_selected.p;
    r.n_aligned = o && o->n_aligned ? o->n_aligned : db->r_n_aligned.p;
    r.status = o && o->status ? o->status : db->r_status.p;
    return r;
}

// The level-synchronous walk + rank cut on device rows.  n_aligned must already be filled
// (encode) unless count_aligned is set.
int walk_dev(stag_db *db, const Work &wk, const uint32_t *d_bits, int64_t n, const int32_t *d_status_in, const Out &o,
             bool count_aligned, hipStream_t st) {
    if (n == 0) return STAG_OK;
    if (n > 0x7fffffff - 1024) return fail(STAG_E_INVALID, "batch of %lld sequences is too large; split it", (long long)n);
    if ((uintptr_t)d_bits & 15) return fail(STAG_E_INVALID, "d_bits must be 16-byte aligned");
    // (a null buffer handed to a kernel is a GPU memory fault, not an error code: refuse it here)
    if (!d_bits || !wk.node || !wk.pos || !wk.count || !wk.task_off || !wk.n_tasks || !wk.order || !wk.lvl || !wk.tasks ||
        !wk.win || !o.path_node || !o.path_prob || !o.path_len || !o.lead || !o.level_prob || !o.selected || !o.n_aligned ||
        !o.status)
        return fail(STAG_E_INVALID, "walk: a workspace or result buffer is null (the handle's workspace was not reserved)");
    const int D = std::max(db->max_depth, 1);
    const unsigned g256 = (unsigned)((n + 255) / 256);
    CHK_TEMP_CLEAR();
    CHK_TEMP(d_bits, (size_t)n * db->W * 4); CHK_TEMP(d_status_in, n * 4);
    CHK_TEMP(o.path_node, (size_t)n * D * 4); CHK_TEMP(o.path_prob, (size_t)n * D * 8); CHK_TEMP(o.path_len, n * 4);
    CHK_TEMP(o.lead, n * 4); CHK_TEMP(o.level_prob, (size_t)n * std::max(db->n_keys, 1) * 8); CHK_TEMP(o.selected, n * 4);
    CHK_TEMP(o.n_aligned, n * 4); CHK_TEMP(o.status, n * 4);
    CHK_UPLOAD();
    if (count_aligned) {
        Prof p(db, st, 4, wk.prof);
        trace_launch(st, "k_count_aligned n=%lld L=%d W=%d", (long long)n, db->L, db->W);
        hipLaunchKernelGGL(k_count_aligned, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_bits, n, db->L, db->W,
                           o.n_aligned, wk.win);
    }
    {
        Prof p(db, st, 2, wk.prof);
        HIP_TRY(hipMemsetAsync(wk.count, 0, sizeof(int32_t) * db->n_nodes, st));
        if (db->max_depth == 0) {
            trace_launch(st, "k_init_status n=%lld", (long long)n);
            hipLaunchKernelGGL(k_init_status, dim3(g256), dim3(256), 0, st, n, d_status_in, o.status);
        }
    }
    int rc;
    if ((rc = launch_check("k_init_status"))) return rc;
    HintTables hints;
    hints.blk_off = db->blk_off.p; hints.NC = db->nchunks;
    hints.mfma_ok = db->walk_mode != 1 ? db->mfma_ok.p : nullptr;
    hints.pf_off = db->pf_off.p; hints.dig_base = db->dig_base.p; hints.dig_ld = db->dig_ld.p; hints.qscale = db->qscale.p;
    for (int d = 0; d < db->max_depth; ++d) {
        const int lo = db->level_start[d], hi = db->level_start[d + 1];
        LevelRec *cur = wk.lvl + (size_t)d * n;
        if (d == 0) {
            Prof p(db, st, 2, wk.prof);
            trace_launch(st, "k_root_level n=%lld", (long long)n);
            hipLaunchKernelGGL(k_root_level, dim3(g256), dim3(256), 0, st, n, d_status_in, o.status, wk.win, cur,
                               db->child_first.p, db->node_err.p, wk.order, wk.tasks, wk.n_tasks, hints);
        } else {
            Prof p(db, st, 2, wk.prof);
            trace_launch(st, "k_assign+k_scan_level+k_scatter d=%d n=%lld lo=%d hi=%d", d, (long long)n, lo, hi);
            hipLaunchKernelGGL(k_assign, dim3((unsigned)((n + ASSIGN_SPAN - 1) / ASSIGN_SPAN)), dim3(ASSIGN_THREADS), 0, st,
                               n, lo, hi, d ? cur - n : (const LevelRec *)nullptr, cur, d_status_in, o.status, wk.node,
                               wk.count, wk.pos);
            hipLaunchKernelGGL(k_scan_level, dim3(1), dim3(1024), 0, st, lo, hi, wk.count, db->child_first.p,
                               db->node_err.p, wk.task_off, wk.n_tasks + d);
            hipLaunchKernelGGL(k_scatter, dim3(g256), dim3(256), 0, st, n, lo, hi, wk.node, wk.pos, wk.count,
                               wk.task_off, db->child_first.p, db->node_err.p, wk.win, wk.order, wk.tasks, cur, o.status, hints);
        }
        if ((rc = launch_check("grouping kernels"))) return rc;
        WalkArgs a;
        a.bits = d_bits; a.W = db->W; a.NC = db->nchunks;
        a.n_tasks = wk.n_tasks + d; a.tasks = wk.tasks; a.order = wk.order;
        a.intercept = db->intercept.p;
        a.blk_off = db->blk_off.p; a.blocks = db->blocks.p;
        a.lvl = cur; a.status = o.status;
        a.gappre = db->gappre.p; a.gapsuf = db->gapsuf.p;
        a.dbg = db->dbg.p;
        const int64_t max_tasks = std::min<int64_t>(n, n / TILE + (hi - lo) + 1);
        // nodes with matrix-core tables are walked by k_walk_mfma, the others (children beyond its column blocks,
        // coefficients that are not finite, STAG_WALK=gather) by k_walk_level; a level launches what it has
        const bool has_mfma = db->level_has_mfma[d], has_gather = db->level_has_gather[d];
        a.mfma_ok = has_mfma ? db->mfma_ok.p : nullptr;
        if (has_mfma) {
            MfmaArgs m;
            m.pf_off = db->pf_off.p; m.pfeat = db->pfeat.p; m.dig = db->dig.p; m.dig_base = db->dig_base.p;
            m.dig_ld = db->dig_ld.p; m.qscale = db->qscale.p; m.mfma_ok = db->mfma_ok.p;
            Prof p(db, st, 1, wk.prof);
            trace_launch(st, "k_walk_mfma d=%d grid=%lld W=%d NC=%d", d, (long long)max_tasks, db->W, db->nchunks);
            hipLaunchKernelGGL(k_walk_mfma, dim3((unsigned)max_tasks), dim3(NW * 64), 0, st, a, m);
        }
        if (has_gather) {
            Prof p(db, st, 1, wk.prof);
            trace_launch(st, "k_walk_level d=%d grid=%lld W=%d NC=%d", d, (long long)max_tasks, db->W, db->nchunks);
            hipLaunchKernelGGL(k_walk_level, dim3((unsigned)max_tasks), dim3(NW * 64), 0, st, a);
        }
        if ((rc = launch_check("k_walk_level"))) return rc;
    }
    {
        Prof p(db, st, 3, wk.prof);
        const size_t lds = (size_t)256 * ((D + db->n_keys) * sizeof(double) + D * sizeof(int32_t));
        if (lds > 160 * 1024) return fail(STAG_E_INVALID, "tree depth %d + %d rank-cut keys exceed the rank-cut kernel's LDS tile", D, db->n_keys);
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_rankcut, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        trace_launch(st, "k_rankcut n=%lld D=%d K=%d", (long long)n, D, db->n_keys);
        hipLaunchKernelGGL(k_rankcut, dim3(g256), dim3(256), lds, st, n, D, db->max_depth, db->n_keys, db->n_tf, db->tf.p,
                           wk.lvl, o.path_node, o.path_prob, o.path_len, o.lead, o.level_prob, o.selected, o.status);
    }
    return launch_check("k_rankcut");
}

// Host record offsets must be non-negative and non-decreasing: an interior offset outside its neighbours would
// send the encode kernel before or past the text buffer.
int check_offsets(const int64_t *offs, int64_t n, const char *what) {
    if (offs[0] < 0) return fail(STAG_E_INVALID, "%s[0] is negative", what);
    for (int64_t i = 0; i < n; ++i)
        if (offs[i + 1] < offs[i])
            return fail(STAG_E_INVALID, "%s must be non-decreasing (%s[%lld] = %lld > %s[%lld] = %lld)", what, what,
                        (long long)i, (long long)offs[i], what, (long long)(i + 1), (long long)offs[i + 1]);
    return STAG_OK;
}

// Records per sub-batch of the host pipeline (STAG_PIPE_RECORDS overrides it for tests): about 64 MB of text, but
// no fewer than 65,536 records (smaller launches use the GPU badly: per million reads the walk takes 0.62 ms in
// launches of 500k, 0.81 ms in launches of 250k and several ms in launches of 65k) and no more than 262,144 (the
// first upload and the last download of a call overlap with nothing).  Full-width amplicon text (1.5 kB per
// read) therefore goes in sub-batches of 65,536, trimmed records (0.27 kB) in sub-batches of about 250,000.
int64_t pipe_records(int64_t n, int64_t text_bytes) {
    const char *e = getenv("STAG_PIPE_RECORDS");
    const long long v = e ? atoll(e) : 0;
    if (v > 0) return v;
    const int64_t per_record = std::max<int64_t>(text_bytes / std::max<int64_t>(n, 1), 1);
    return std::min<int64_t>(std::max<int64_t>(((int64_t)64 << 20) / per_record, 65536), 262144);
}

template <typename T>
int d2h(T *host, const T *dev, size_t count, hipStream_t st) {
    if (!host || count == 0) return STAG_OK;
    HIP_TRY(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, st));
    return STAG_OK;
}

int copy_results_to_host(stag_db *db, const stag_results_t *out, int64_t n, hipStream_t st) {
    const size_t D = std::max(db->max_depth, 1), K = db->n_keys;
    int rc;
    if ((rc = d2h(out->path_node, db->r_path_node.p, n * D, st))) return rc;
    if ((rc = d2h(out->path_prob, db->r_path_prob.p, n * D, st))) return rc;
    if ((rc = d2h(out->path_len, db->r_path_len.p, n, st))) return rc;
    if ((rc = d2h(out->n_lead_single, db->r_lead.p, n, st))) return rc;
    if ((rc = d2h(out->level_prob, db->r_level_prob.p, n * K, st))) return rc;
    if ((rc = d2h(out->selected, db->r_selected.p, n, st))) return rc;
    if ((rc = d2h(out->n_aligned, db->r_n_aligned.p, n, st))) return rc;
    if ((rc = d2h(out->status, db->r_status.p, n, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return STAG_OK;
}

}  // namespace

extern "C" {

const char *stag_last_error(void) { return g_err.c_str(); }

int stag_device_count(int32_t *count) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    if (count) *count = c;
    return STAG_OK;
}

int32_t stag_words_per_row(int32_t n_features) {
    const int32_t w = (n_features + 31) / 32;
    return std::max((w + 3) / 4 * 4, 4);
}

int stag_db_create(const stag_tables_t *t, int32_t device, stag_db_t **out) {
    if (!t || !out) return fail(STAG_E_INVALID, "null argument");
    *out = nullptr;
    if (t->n_nodes < 1 || t->n_features < 1) return fail(STAG_E_INVALID, "n_nodes and n_features must be positive");
    if (!t->child_first || !t->row_kind || !t->intercept || !t->row_off || (t->n_keys > 0 && !t->tf))
        return fail(STAG_E_INVALID, "null table pointer");
    if (t->n_keys < 0 || t->n_tf < 0) return fail(STAG_E_INVALID, "negative rank-cut shape");
    if (t->row_off[t->n_nodes] > 0 && (!t->row_feat || !t->row_w))
        return fail(STAG_E_INVALID, "row_off announces %lld coefficients but row_feat / row_w is null", (long long)t->row_off[t->n_nodes]);
    const int n = t->n_nodes;
    if (t->child_first[n] != n || (n > 1 && t->child_first[0] != 1) || (n == 1 && t->child_first[0] != 1))
        return fail(STAG_E_INVALID, "child_first is not a breadth-first numbering (child_first[0]=%d, child_first[n]=%d, n=%d)",
                    t->child_first[0], t->child_first[n], n);
    for (int i = 0; i < n; ++i)
        if (t->child_first[i + 1] < t->child_first[i] || t->child_first[i] <= i)
            return fail(STAG_E_INVALID, "child_first not monotone / not breadth-first at node %d", i);
    int rc = check_device(device);
    if (rc) return rc;
    {   // per-device set-up: the address of this GPU's copy of `scratch` (device is current here)
        unsigned char *sp = nullptr;
        if ((rc = scratch_of(device, &sp))) return rc;
    }

    stag_db *db = new (std::nothrow) stag_db;
    if (!db) return fail(STAG_E_NOMEM, "out of host memory");
    db->device = device;
    db->n_nodes = n;
    db->L = t->n_features;
    db->W = stag_words_per_row(t->n_features);
    db->nchunks = (db->W + CHUNK_WORDS - 1) / CHUNK_WORDS;
    db->n_keys = t->n_keys;
    db->n_tf = t->n_tf;

    // levels: [0,1) is the root; the children of a level's nodes are the next level
    db->level_start = {0, 1};
    for (;;) {
        const int hi = db->level_start.back();
        const int nhi = t->child_first[hi];  // == n when hi == n
        if (hi >= n || nhi == hi) break;
        db->level_start.push_back(nhi);
    }
    if (db->level_start.back() != n) {
        delete db;
        return fail(STAG_E_INVALID, "tree numbering does not cover all %d nodes (reached %d)", n, db->level_start.back());
    }
    db->max_depth = (int)db->level_start.size() - 2;  // levels whose nodes can have children
    for (int d = 0; d < db->max_depth; ++d)
        db->max_level_nodes = std::max(db->max_level_nodes, db->level_start[d + 1] - db->level_start[d]);

    // per-parent error status and the set of rows that can ever be evaluated
    std::vector<uint8_t> node_err(n, 0), evaluated(n, 0);
    for (int p = 0; p < n; ++p) {
        const int c0 = t->child_first[p], c1 = t->child_first[p + 1];
        if (c1 - c0 < 2) continue;
        for (int c = c0; c < c1; ++c) {
            if (t->row_kind[c] == 1) { evaluated[c] = 1; continue; }
            node_err[p] = t->row_kind[c] == 0 ? STAG_SEQ_MISSING_CLASSIFIER : STAG_SEQ_STRING_CLASSIFIER;
            break;
        }
        if (node_err[p]) for (int c = c0; c < c1; ++c) evaluated[c] = 0;
    }

    // coefficient blocks per (row, 16-word chunk) and, per chunk, the sum of the coefficients on
    // "others" bits (feature % 5 == 0): what the all-gaps row R0 scores there
    const int nch = db->nchunks;
    std::vector<uint32_t> blk_off((size_t)n * nch + 1, 0);
    std::vector<double> gappre((size_t)n * (nch + 1), 0.0), gapsuf((size_t)n * (nch + 1), 0.0), gaps(nch, 0.0);
    std::vector<Entry> entries;
    for (int c = 0; c < n; ++c) {
        int64_t e = t->row_off[c];
        const int64_t e1 = t->row_off[c + 1];
        if (e1 < e) { delete db; return fail(STAG_E_INVALID, "row_off not monotone at node %d", c); }
        int prev = -1;
        std::fill(gaps.begin(), gaps.end(), 0.0);
        for (int ch = 0; ch < nch; ++ch) {
            blk_off[(size_t)c * nch + ch] = (uint32_t)(entries.size() / BLOCK_ENTRIES);
            if (!evaluated[c]) continue;
            double gap = 0.0;   // same additions, same order, as the kernel does for a row equal to R0
            const int64_t f_end = (int64_t)(ch + 1) * CHUNK_WORDS * 32;
            while (e < e1 && t->row_feat[e] < f_end) {
                const int f = t->row_feat[e];
                if (f <= prev || f >= t->n_features) {
                    delete db;
                    return fail(STAG_E_INVALID, "row %d: feature %d out of order or out of range", c, f);
                }
                prev = f;
                Entry en;
                en.off = 4u * (uint32_t)(f >> 5);
                en.bit = (uint32_t)(f & 31);
                en.w = t->row_w[e];
                if (f % 5 == 0) gap += en.w;   // the others contribute +0.0 in the kernel
                entries.push_back(en);
                ++e;
            }
            const uint32_t pad_off = 4u * (uint32_t)(ch * CHUNK_WORDS);   // padding reads a staged word
            while (entries.size() % BLOCK_ENTRIES) entries.push_back(Entry{pad_off, 0u, 0.0});
            gaps[ch] = gap;
        }
        // the two chains the kernel adds to a row's own-window sum (fixed orders: up from chunk 0,
        // down from the last chunk)
        double *pre = &gappre[(size_t)c * (nch + 1)], *suf = &gapsuf[(size_t)c * (nch + 1)];
        for (int ch = 0; ch < nch; ++ch) pre[ch + 1] = pre[ch] + gaps[ch];
        for (int ch = nch - 1; ch >= 0; --ch) suf[ch] = suf[ch + 1] + gaps[ch];
        if (evaluated[c] && e != e1) { delete db; return fail(STAG_E_INVALID, "row %d: feature out of range", c); }
        if (entries.size() / BLOCK_ENTRIES > 0x7fffffffull) { delete db; return fail(STAG_E_INVALID, "coefficient table too large"); }
    }
    blk_off[(size_t)n * nch] = (uint32_t)(entries.size() / BLOCK_ENTRIES);

    // ---- matrix-core tables.  For a parent p whose k children are all scored (2 <= k <= MFMA_MAX_K), chunk by
    // chunk: U = the sorted union of the children's features in the chunk, padded to a multiple of 16 (a padding
    // feature reads the chunk's first word and has all-zero digits); every coefficient w becomes the integer
    // v = round(w * 2^q) (q per parent: the largest with max |v| < 2^46) written as 6 signed base-256 digits.
    // dig holds, per parent, a matrix [columns][features of the parent]: column 16 (j / 2) + 8 (j % 2) + s = digit s
    // of child j (two children per block of 16 columns, columns 6 and 7 of each half are zero).  The kernel multiplies the 0/1 feature bytes of 64 rows with it on
    // the matrix cores (v_mfma_i32_16x16x64_i8): integer arithmetic, exact and order-independent.
    std::vector<uint32_t> pf_off((size_t)n * nch + 1, 0), pfeat, dig_ld(n, 0);
    std::vector<uint64_t> dig_base(n, 0);
    std::vector<double> qscale(n, 0.0);
    std::vector<uint8_t> mfma_ok(n, 0);
    std::vector<int8_t> dig;
    // STAG_WALK: "gather" = every node through k_walk_level, "mfma" = every node that can through k_walk_mfma,
    // otherwise (auto) k_walk_mfma for the nodes whose rows are dense enough for it to win (MFMA_MIN_DENSITY,
    // measured: profiles/r2_walk_density.md)
    const char *walk_env = getenv("STAG_WALK");
    db->walk_mode = walk_env && !strcmp(walk_env, "gather") ? 1 : walk_env && !strcmp(walk_env, "mfma") ? 2 : 0;
    {
        std::vector<int32_t> ufeat;                    // union features of one chunk
        std::vector<size_t> at;                        // per child: next coefficient
        for (int p = 0; p < n; ++p) {
            const int c0 = t->child_first[p], k = t->child_first[p + 1] - c0;
            bool ok = db->walk_mode != 1 && k >= 2 && k <= MFMA_MAX_K && !node_err[p];
            if (ok && db->walk_mode == 0) {      // auto: only where the children's rows are dense
                int64_t nnz = 0;
                for (int c = c0; c < c0 + k; ++c) nnz += t->row_off[c + 1] - t->row_off[c];
                ok = (double)nnz >= MFMA_MIN_DENSITY * (double)k * (double)t->n_features;
            }
            double wmax = 0.0;
            if (ok)
                for (int c = c0; c < c0 + k && ok; ++c) {
                    if (!(std::fabs(t->intercept[c]) < 1e300)) { ok = false; break; }
                    for (int64_t e = t->row_off[c]; e < t->row_off[c + 1]; ++e) {
                        const double w = t->row_w[e];
                        if (!(std::fabs(w) < 1e30)) { ok = false; break; }     // inf / NaN: the gather walk keeps their semantics
                        wmax = std::max(wmax, std::fabs(w));
                    }
                }
            const size_t first_feat = pfeat.size();
            if (ok) {
                int q = 80;
                if (wmax > 0.0) { int ex; std::frexp(wmax, &ex); q = std::min(80, 46 - ex); }   // wmax < 2^ex
                const double scale = std::ldexp(1.0, q);
                at.assign(k, 0);
                for (int j = 0; j < k; ++j) at[j] = (size_t)t->row_off[c0 + j];
                // pass 1: the union per chunk -> descriptors; remember each chunk's range
                std::vector<std::pair<size_t, size_t>> chunk_range(nch);
                for (int ch = 0; ch < nch; ++ch) {
                    pf_off[(size_t)p * nch + ch] = (uint32_t)pfeat.size();
                    const int64_t f_end = (int64_t)(ch + 1) * CHUNK_WORDS * 32;
                    ufeat.clear();
                    for (int j = 0; j < k; ++j) {
                        size_t e = at[j];
                        const size_t e1 = (size_t)t->row_off[c0 + j + 1];
                        while (e < e1 && t->row_feat[e] < f_end) ufeat.push_back(t->row_feat[e++]);
                        at[j] = e;
                    }
                    std::sort(ufeat.begin(), ufeat.end());
                    ufeat.erase(std::unique(ufeat.begin(), ufeat.end()), ufeat.end());
                    const size_t b = pfeat.size();
                    for (int32_t f : ufeat) pfeat.push_back(4u * (uint32_t)(f >> 5) | ((uint32_t)(f & 31) << 16));
                    while ((pfeat.size() - first_feat) % 16) pfeat.push_back(4u * (uint32_t)(ch * CHUNK_WORDS));
                    chunk_range[ch] = {b, pfeat.size()};
                }
                const size_t U = pfeat.size() - first_feat;
                const size_t ncolp = (size_t)((k + 1) / 2) * 16;      // two children per block of 16 columns
                if (U > 0xffffff0u || pfeat.size() > 0xfffffff0u) { delete db; return fail(STAG_E_INVALID, "feature table too large"); }
                dig_base[p] = dig.size();
                dig_ld[p] = (uint32_t)U;
                qscale[p] = std::ldexp(1.0, -q);
                dig.resize(dig.size() + ncolp * U, 0);
                int8_t *M = dig.data() + dig_base[p];
                // pass 2: digits.  Features of a chunk are sorted, so a child's coefficient is found by merging.
                for (int j = 0; j < k; ++j) {
                    size_t e = (size_t)t->row_off[c0 + j];
                    const size_t e1 = (size_t)t->row_off[c0 + j + 1];
                    for (int ch = 0; ch < nch && e < e1; ++ch) {
                        size_t u = chunk_range[ch].first;
                        const int64_t f_end = (int64_t)(ch + 1) * CHUNK_WORDS * 32;
                        while (e < e1 && t->row_feat[e] < f_end) {
                            const uint32_t want = 4u * (uint32_t)(t->row_feat[e] >> 5) | ((uint32_t)(t->row_feat[e] & 31) << 16);
                            while (pfeat[u] != want) ++u;       // present by construction
                            long long v = std::llround(t->row_w[e] * scale);
                            for (int sl = 0; sl < MFMA_SLICES; ++sl) {
                                const long long d = ((v + 128) & 255) - 128;
                                M[((size_t)(j >> 1) * 16 + (size_t)(j & 1) * 8 + sl) * U + (u - first_feat)] = (int8_t)d;
                                v = (v - d) >> 8;
                            }
                            if (v != 0) { ok = false; }          // cannot happen with |v| < 2^46
                            ++e;
                        }
                    }
                }
            }
            if (!ok) {      // no tables for this node: it is walked by the gather kernel
                pfeat.resize(first_feat);
                for (int ch = 0; ch < nch; ++ch) pf_off[(size_t)p * nch + ch] = (uint32_t)first_feat;
                if (dig_ld[p]) { dig.resize(dig_base[p]); dig_ld[p] = 0; }
            }
            mfma_ok[p] = ok ? 1 : 0;
        }
        pf_off[(size_t)n * nch] = (uint32_t)pfeat.size();
        for (int i = 0; i < 256; ++i) pfeat.push_back(0u);     // the kernel reads descriptors / digits a block at a time
        dig.resize(dig.size() + 4096, 0);
    }
    db->level_has_mfma.assign(db->max_depth + 1, 0);
    db->level_has_gather.assign(db->max_depth + 1, 0);
    for (int d = 0; d < db->max_depth; ++d)
        for (int p = db->level_start[d]; p < db->level_start[d + 1]; ++p) {
            const int k = t->child_first[p + 1] - t->child_first[p];
            if (k < 2 || node_err[p]) continue;
            (mfma_ok[p] ? db->level_has_mfma : db->level_has_gather)[d] = 1;
        }

#define UP(buf, vec, T)                                                                              \
    do {                                                                                             \
        if ((rc = db->buf.reserve((vec).size()))) { db->free_all(); delete db; return rc; }          \
        if (!(vec).empty() && hipMemcpy(db->buf.p, (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { \
            db->free_all(); delete db; return fail(STAG_E_HIP, "upload of " #buf " failed");         \
        }                                                                                            \
        db->table_bytes += (int64_t)((vec).size() * sizeof(T));                                      \
    } while (0)
    std::vector<int32_t> cf(t->child_first, t->child_first + n + 1);
    std::vector<double> ic(t->intercept, t->intercept + n);
    std::vector<double> tfv(t->tf ? t->tf : nullptr, t->tf ? t->tf + (size_t)t->n_keys * (1 + t->n_tf) : nullptr);
    UP(gappre, gappre, double);
    UP(gapsuf, gapsuf, double);
    UP(pf_off, pf_off, uint32_t);
    UP(pfeat, pfeat, uint32_t);
    UP(dig, dig, int8_t);
    UP(dig_base, dig_base, uint64_t);
    UP(dig_ld, dig_ld, uint32_t);
    UP(qscale, qscale, double);
    UP(mfma_ok, mfma_ok, uint8_t);
    UP(child_first, cf, int32_t);
    UP(node_err, node_err, uint8_t);
    UP(intercept, ic, double);
    UP(blk_off, blk_off, uint32_t);
    UP(tf, tfv, double);
    static_assert(sizeof(Entry) == sizeof(uint4), "Entry must be 16 bytes");
    if ((rc = db->blocks.reserve(entries.size()))) { db->free_all(); delete db; return rc; }
    if (!entries.empty() && hipMemcpy(db->blocks.p, entries.data(), entries.size() * sizeof(Entry), hipMemcpyHostToDevice) != hipSuccess) {
        db->free_all(); delete db; return fail(STAG_E_HIP, "upload of coefficient blocks failed");
    }
    db->table_bytes += (int64_t)(entries.size() * sizeof(Entry));
#undef UP
    *out = db;
    return STAG_OK;
}

void stag_db_destroy(stag_db_t *db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
#if STAG_ABL == 10
    {
        unsigned long long h[16] = {};
        std::vector<unsigned long long> all(4096 * 16, 0);
        (void)hipDeviceSynchronize();
        if (db->dbg.p && hipMemcpy(all.data(), db->dbg.p, all.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t i = 0; i < all.size(); ++i) h[i & 15] += all[i];
        }
        if (h[15]) {
            fprintf(stderr, "k_walk_mfma phases, cycles per workgroup over %llu workgroups:", h[15]);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " [%d] %.0f", i, (double)h[i] / (double)h[15]);
            fprintf(stderr, "\n");
        }
    }
#endif
    db->free_all();
    delete db;
}

int stag_db_info(const stag_db_t *db, int32_t *max_depth, int32_t *words_per_row, int32_t *n_keys, int64_t *table_bytes) {
    if (!db) return fail(STAG_E_INVALID, "null db");
    if (max_depth) *max_depth = std::max(db->max_depth, 1);
    if (words_per_row) *words_per_row = db->W;
    if (n_keys) *n_keys = db->n_keys;
    if (table_bytes) *table_bytes = db->table_bytes;
    return STAG_OK;
}

int stag_db_reserve(stag_db_t *db, int64_t n, int64_t text_bytes) {
    if (!db || n < 0 || text_bytes < 0) return fail(STAG_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    if ((rc = reserve_walk(db, n))) return rc;
    if (text_bytes > 0) {
        if ((rc = db->bits.reserve((size_t)n * db->W))) return rc;
        if ((rc = db->enc_n_char.reserve(n))) return rc;
        if ((rc = db->enc_status.reserve(n))) return rc;
    }
    return STAG_OK;
}

int stag_encode_a2m_dev(int32_t n_features, const char *d_text, int64_t text_bytes, const int64_t *d_offs, int64_t n,
                        double min_perc_state, uint32_t *d_bits, int32_t *d_n_char, int32_t *d_n_aligned,
                        int32_t *d_status, void *stream) {
    if (n < 0 || n_features < 1 || (n > 0 && (!d_text || !d_offs || !d_bits || !d_n_char || !d_n_aligned || !d_status)))
        return fail(STAG_E_INVALID, "bad argument");
    return encode_dev(nullptr, n_features, d_text, text_bytes, d_offs, n, min_perc_state, d_bits, d_n_char, d_n_aligned,
                      d_status, nullptr, (hipStream_t)stream);
}

int stag_encode_a2m(int32_t device, int32_t n_features, const char *text, const int64_t *offs, int64_t n,
                    double min_perc_state, uint32_t *bits, int32_t *n_char, int32_t *n_aligned, int32_t *status) {
    if (n < 0 || n_features < 1 || (n > 0 && (!text || !offs || !bits || !n_char || !n_aligned || !status)))
        return fail(STAG_E_INVALID, "bad argument");
    int rc = check_device(device);
    if (rc || n == 0) return rc;
    const int W = stag_words_per_row(n_features);
    if ((rc = check_offsets(offs, n, "offs"))) return rc;
    const int64_t text_bytes = offs[n];
    DevBuf<uint8_t> d_text;
    DevBuf<int64_t> d_offs;
    DevBuf<uint32_t> d_bits;
    DevBuf<int32_t> d_i;
    auto done = [&](int code) { d_text.release(); d_offs.release(); d_bits.release(); d_i.release(); return code; };
    if ((rc = d_text.reserve(text_bytes + 16)) || (rc = d_offs.reserve(n + 1)) || (rc = d_bits.reserve((size_t)n * W)) ||
        (rc = d_i.reserve(3 * n)))
        return done(rc);
    if (hipMemcpy(d_text.p, text, text_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_offs.p, offs, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(STAG_E_HIP, "host to device copy failed"));
    rc = encode_dev(nullptr, n_features, (const char *)d_text.p, (text_bytes + 15) / 16 * 16, d_offs.p, n, min_perc_state, d_bits.p, d_i.p,
                    d_i.p + n, d_i.p + 2 * n, nullptr, nullptr);
    if (rc) return done(rc);
    if (hipMemcpy(bits, d_bits.p, (size_t)n * W * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(n_char, d_i.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(n_aligned, d_i.p + n, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(status, d_i.p + 2 * n, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(STAG_E_HIP, "device to host copy failed: %s", hipGetErrorString(hipGetLastError())));
    return done(STAG_OK);
}

int stag_classify_bits_dev(stag_db_t *db, const uint32_t *d_bits, int64_t n, const int32_t *d_status_in,
                           const stag_results_t *d_out, void *stream) {
    if (!db || n < 0 || (n > 0 && !d_bits)) return fail(STAG_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(db->device));
    int rc = reserve_walk(db, n);
    if (rc) return rc;
    return walk_dev(db, work_of(db), d_bits, n, d_status_in, resolve(db, d_out), true, (hipStream_t)stream);
}

int stag_classify_bits(stag_db_t *db, const uint32_t *bits, int64_t n, const int32_t *status_in,
                       const stag_results_t *out) {
    if (!db || n < 0 || !out || (n > 0 && (!bits || !out->status))) return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    if ((rc = reserve_walk(db, n)) || (rc = db->bits.reserve((size_t)n * db->W)) || (rc = db->enc_status.reserve(n)))
        return rc;
    HIP_TRY(hipMemcpy(db->bits.p, bits, (size_t)n * db->W * 4, hipMemcpyHostToDevice));
    if (status_in) HIP_TRY(hipMemcpy(db->enc_status.p, status_in, n * 4, hipMemcpyHostToDevice));
    if ((rc = walk_dev(db, work_of(db), db->bits.p, n, status_in ? db->enc_status.p : nullptr, resolve(db, nullptr), true, nullptr)))
        return rc;
    return copy_results_to_host(db, out, n, nullptr);
}

int stag_parse_onehot_dev(int32_t n_features, const char *d_text, int64_t text_bytes, const int64_t *d_body_begin,
                          const int64_t *d_body_end, int64_t n, uint32_t *d_bits, int32_t *d_n_fields, int32_t *d_flag,
                          void *stream) {
    if (n < 0 || n_features < 1 || text_bytes < 0 ||
        (n > 0 && (!d_text || !d_body_begin || !d_body_end || !d_bits || !d_n_fields || !d_flag)))
        return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    if ((text_bytes & 3) || ((uintptr_t)d_text & 3))
        return fail(STAG_E_INVALID, "d_text must be 4-byte aligned and text_bytes (the size of the buffer) a multiple of 4: pad the buffer");
    if ((n + 3) / 4 > 0x7fffffff) return fail(STAG_E_INVALID, "batch too large");
    CHK_TEMP_CLEAR();
    CHK_TEMP(d_text, text_bytes); CHK_TEMP(d_body_begin, n * 8); CHK_TEMP(d_body_end, n * 8);
    CHK_TEMP(d_bits, (size_t)n * stag_words_per_row(n_features) * 4); CHK_TEMP(d_n_fields, n * 4); CHK_TEMP(d_flag, n * 4);
    CHK_UPLOAD();
    trace_launch((hipStream_t)stream, "k_parse_onehot n=%lld L=%d", (long long)n, n_features);
    hipLaunchKernelGGL(k_parse_onehot, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_text, text_bytes, d_body_begin, d_body_end, n, n_features,
                       stag_words_per_row(n_features), d_bits, d_n_fields, d_flag);
    return launch_check("k_parse_onehot");
}

// The lines' rows stay in the handle (db->bits) for stag_onehot_text_classify.
int stag_onehot_text_parse(stag_db_t *db, const char *text, int64_t text_bytes, const int64_t *body_begin,
                           const int64_t *body_end, int64_t n, int32_t *n_fields, int32_t *flag) {
    if (!db || n < 0 || text_bytes < 0 || (n > 0 && (!text || !body_begin || !body_end || !n_fields || !flag)))
        return fail(STAG_E_INVALID, "bad argument");
    db->parsed_rows = 0;
    if (n == 0) return STAG_OK;
    for (int64_t i = 0; i < n; ++i)
        if (body_begin[i] < 0 || body_end[i] < body_begin[i] || body_end[i] > text_bytes)
            return fail(STAG_E_INVALID, "line %lld: body [%lld, %lld) is not inside the text buffer of %lld bytes", (long long)i,
                        (long long)body_begin[i], (long long)body_end[i], (long long)text_bytes);
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    const int64_t padded = (text_bytes + 15) / 16 * 16;
    if ((rc = reserve_walk(db, n)) || (rc = db->bits.reserve((size_t)n * db->W)) || (rc = db->enc_status.reserve(n)) ||
        (rc = db->enc_n_char.reserve(n)) || (rc = db->text.reserve(padded + 16)) || (rc = db->bt_begin.reserve(n)) ||
        (rc = db->bt_end.reserve(n)))
        return rc;
    HIP_TRY(hipMemcpyAsync(db->text.p, text, text_bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemsetAsync(db->text.p + text_bytes, 0, padded - text_bytes, nullptr));
    HIP_TRY(hipMemcpyAsync(db->bt_begin.p, body_begin, n * sizeof(int64_t), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(db->bt_end.p, body_end, n * sizeof(int64_t), hipMemcpyHostToDevice, nullptr));
    // n_fields -> enc_n_char, flag -> enc_status (overwritten with the caller's status by stag_onehot_text_classify)
    if ((rc = stag_parse_onehot_dev(db->L, (const char *)db->text.p, padded, db->bt_begin.p, db->bt_end.p, n, db->bits.p,
                                    db->enc_n_char.p, db->enc_status.p, nullptr)))
        return rc;
    HIP_TRY(hipMemcpyAsync(n_fields, db->enc_n_char.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(flag, db->enc_status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    db->parsed_rows = n;
    return STAG_OK;
}

int stag_onehot_text_classify(stag_db_t *db, int64_t n, const int64_t *patch_index, const uint32_t *patch_rows,
                              int64_t n_patch, const int32_t *status_in, const stag_results_t *out) {
    if (!db || n < 0 || n_patch < 0 || !out || (n > 0 && !out->status) || (n_patch > 0 && (!patch_index || !patch_rows)))
        return fail(STAG_E_INVALID, "bad argument");
    if (n != db->parsed_rows)
        return fail(STAG_E_INVALID, "stag_onehot_text_classify: %lld rows asked for, stag_onehot_text_parse left %lld",
                    (long long)n, (long long)db->parsed_rows);
    db->parsed_rows = 0;
    if (n == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    for (int64_t j = 0; j < n_patch; ++j) {
        if (patch_index[j] < 0 || patch_index[j] >= n) return fail(STAG_E_INVALID, "patch_index[%lld] out of range", (long long)j);
        HIP_TRY(hipMemcpyAsync(db->bits.p + (size_t)patch_index[j] * db->W, patch_rows + (size_t)j * db->W,
                               (size_t)db->W * 4, hipMemcpyHostToDevice, nullptr));
    }
    if (status_in) HIP_TRY(hipMemcpyAsync(db->enc_status.p, status_in, n * 4, hipMemcpyHostToDevice, nullptr));
    int rc;
    if ((rc = walk_dev(db, work_of(db), db->bits.p, n, status_in ? db->enc_status.p : nullptr, resolve(db, nullptr), true, nullptr)))
        return rc;
    return copy_results_to_host(db, out, n, nullptr);
}

int stag_classify_a2m_trimmed_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs,
                                  const int32_t *d_lead_trail, int64_t n, double min_perc_state,
                                  const stag_results_t *d_out, void *stream) {
    if (!db || n < 0 || (n > 0 && (!d_text || !d_offs))) return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    if ((rc = stag_db_reserve(db, n, std::max<int64_t>(text_bytes, 1)))) return rc;
    const Out o = resolve(db, d_out);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = encode_dev(db, db->L, d_text, text_bytes, d_offs, n, min_perc_state, db->bits.p, db->enc_n_char.p,
                         o.n_aligned, db->enc_status.p, db->win.p, st, nullptr, nullptr, /*full_rows=*/false, d_lead_trail)))
        return rc;
    return walk_dev(db, work_of(db), db->bits.p, n, db->enc_status.p, o, false, st);
}

int stag_classify_a2m_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs, int64_t n,
                          double min_perc_state, const stag_results_t *d_out, void *stream) {
    return stag_classify_a2m_trimmed_dev(db, d_text, text_bytes, d_offs, nullptr, n, min_perc_state, d_out, stream);
}

// Same kernels as stag_classify_a2m_dev; the workspace and any output the caller left NULL come out
// of this GPU's `scratch` instead of the handle's hipMalloc'ed buffers.  No hipMalloc / hipFree /
// event creation on this path.
int stag_classify_a2m_noalloc_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs, int64_t n,
                                  double min_perc_state, const stag_results_t *d_out, void *stream) {
    if (!db || n < 0 || (n > 0 && (!d_text || !d_offs))) return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    if (n > 0x7fffffff - 1024) return fail(STAG_E_INVALID, "batch of %lld sequences is too large; split it", (long long)n);
    HIP_TRY(hipSetDevice(db->device));
    unsigned char *base = nullptr;
    hipStream_t st = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        base = db->device < MAX_DEVICES ? g_scratch_ptr[db->device] : nullptr;
        if (base) {
            // overlap on one GPU is refused where it can be seen: the previous call went to another stream and
            // that stream has not drained (hipStreamQuery allocates nothing)
            if (g_scratch_used[db->device] && g_scratch_stream[db->device] != st &&
                hipStreamQuery(g_scratch_stream[db->device]) == hipErrorNotReady)
                return fail(STAG_E_INVALID, "stag_classify_a2m_noalloc_dev: the previous call on this GPU went to another "
                                            "stream that still has work queued; all such calls share one scratch array "
                                            "(synchronise that stream first, or use stag_classify_a2m_dev)");
            (void)hipGetLastError();
            g_scratch_stream[db->device] = st;
            g_scratch_used[db->device] = true;
        }
    }
    if (!base) return fail(STAG_E_HIP, "the scratch array of device %d was not set up (stag_db_create does it)", db->device);
    const size_t D = std::max(db->max_depth, 1), K = std::max(db->n_keys, 1);
    const size_t max_tasks = walk_max_tasks(n, db->max_level_nodes);
    size_t used = 0;
    auto carve = [&](size_t bytes) {   // 256-byte aligned pieces, one after the other
        unsigned char *p = base + used;
        used += (bytes + 255) / 256 * 256;
        return p;
    };
    Work wk;
    wk.bits = (uint32_t *)carve((size_t)n * db->W * 4);
    wk.lvl = (LevelRec *)carve(D * n * sizeof(LevelRec));
    wk.order = (SeqRef *)carve(max_tasks * TILE * sizeof(SeqRef));
    wk.tasks = (Task *)carve(max_tasks * sizeof(Task));
    wk.node = (int32_t *)carve(n * 4);
    wk.pos = (int32_t *)carve(n * 4);
    wk.win = (uint32_t *)carve(n * 4);
    wk.enc_n_char = (int32_t *)carve(n * 4);
    wk.enc_status = (int32_t *)carve(n * 4);
    wk.count = (int32_t *)carve((size_t)db->n_nodes * 4);
    wk.task_off = (int32_t *)carve((size_t)db->n_nodes * 4);
    wk.n_tasks = (int32_t *)carve(D * 4);
    wk.prof = false;
    Out o;   // outputs the caller does not want still have to land somewhere
    o.path_node = d_out && d_out->path_node ? d_out->path_node : (int32_t *)carve(n * D * 4);
    o.path_prob = d_out && d_out->path_prob ? d_out->path_prob : (double *)carve(n * D * 8);
    o.path_len = d_out && d_out->path_len ? d_out->path_len : (int32_t *)carve(n * 4);
    o.lead = d_out && d_out->n_lead_single ? d_out->n_lead_single : (int32_t *)carve(n * 4);
    o.level_prob = d_out && d_out->level_prob ? d_out->level_prob : (double *)carve(n * K * 8);
    o.selected = d_out && d_out->selected ? d_out->selected : (int32_t *)carve(n * 4);
    o.n_aligned = d_out && d_out->n_aligned ? d_out->n_aligned : (int32_t *)carve(n * 4);
    o.status = d_out && d_out->status ? d_out->status : (int32_t *)carve(n * 4);
    if (used > (size_t)STAG_NOALLOC_SCRATCH_BYTES)
        return fail(STAG_E_INVALID,
                    "%lld sequences of this database need %lld bytes of workspace; the static scratch of "
                    "stag_classify_a2m_noalloc_dev holds %lld (split the batch, or use stag_classify_a2m_dev)",
                    (long long)n, (long long)used, (long long)STAG_NOALLOC_SCRATCH_BYTES);
    int rc;
    if ((rc = encode_dev(nullptr, db->L, d_text, text_bytes, d_offs, n, min_perc_state, wk.bits, wk.enc_n_char, o.n_aligned,
                         wk.enc_status, wk.win, st, nullptr, nullptr, /*full_rows=*/false)))
        return rc;
    return walk_dev(db, wk, wk.bits, n, wk.enc_status, o, false, st);
}

int stag_classify_a2m(stag_db_t *db, const char *text, const int64_t *offs, int64_t n, double min_perc_state,
                      const stag_results_t *out, uint32_t *bits_out, int32_t *n_char_out) {
    return stag_classify_a2m_trimmed(db, text, offs, nullptr, n, min_perc_state, out, bits_out, n_char_out);
}

int stag_classify_a2m_trimmed(stag_db_t *db, const char *text, const int64_t *offs, const int32_t *lead_trail, int64_t n,
                              double min_perc_state, const stag_results_t *out, uint32_t *bits_out, int32_t *n_char_out) {
    if (!db || n < 0 || !out || (n > 0 && (!text || !offs || !out->status))) return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    int rc;
    // The records are validated sub-batch by sub-batch, each right before its upload is queued (a pass over a
    // million offsets takes as long as uploading a fifth of them: done up front it was a fifth of the call; done
    // here it runs while the device works on the sub-batches before).  A record that fails stops the call with
    // STAG_E_INVALID as before; the output arrays may then hold the results of the sub-batches before it.
    auto validate = [&](int64_t b0, int64_t b1) -> int {
        int r = check_offsets(offs + b0, b1 - b0, "offs");
        if (r) return r;
        if (lead_trail)
            for (int64_t i = b0; i < b1; ++i) {
                const int64_t cols = (int64_t)lead_trail[2 * i] + lead_trail[2 * i + 1] + (offs[i + 1] - offs[i]);
                if (lead_trail[2 * i] < 0 || lead_trail[2 * i + 1] < 0 || cols > 0x7fffffff)
                    return fail(STAG_E_INVALID, "lead_trail of record %lld: counts must be non-negative and the record shorter than 2^31 columns", (long long)i);
            }
        return STAG_OK;
    };
    if (offs[0] < 0 || offs[n] < offs[0]) return (rc = check_offsets(offs, n, "offs")) ? rc : fail(STAG_E_INVALID, "bad offsets");
    HIP_TRY(hipSetDevice(db->device));
    // ---- sub-batches: cut at record boundaries, at most PIPE_RECORDS records / PIPE_BYTES bytes each
    const int64_t PIPE_RECORDS = pipe_records(n, offs[n] - offs[0]), PIPE_BYTES = (int64_t)96 << 20;
    const size_t D = std::max(db->max_depth, 1), K = std::max(db->n_keys, 1);
    if (!db->slot[PIPE_SLOTS - 1].down) {   // (the last one taken: a handle whose set-up failed half way takes the rest)
        for (hipStream_t *st : {&db->s_up, &db->s_run, &db->s_down})
            if (!*st) HIP_TRY(g_pipe_pool.stream(db->device, st));
        for (auto &sl : db->slot)
            for (hipEvent_t *e : {&sl.up, &sl.done, &sl.down})
                if (!*e) HIP_TRY(g_pipe_pool.event(db->device, e));
    }
    for (auto &sl : db->slot) sl.used = false;
    auto fin = [&](int code) {   // on any exit nothing of this call may still be in flight
        (void)hipStreamSynchronize(db->s_up);
        (void)hipStreamSynchronize(db->s_run);
        (void)hipStreamSynchronize(db->s_down);
        return code;
    };
#define PIPE_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fin(fail(e_ == hipErrorOutOfMemory ? STAG_E_NOMEM : STAG_E_HIP, "%s: %s (%s:%d)", #expr, \
                            hipGetErrorString(e_), __FILE__, __LINE__));                                \
    } while (0)
    // the cuts first, so that everything is reserved ONCE, for the largest sub-batch, before any work is queued
    // (growing a buffer frees it, and the previous sub-batch's kernels would still be using it)
    // The last sub-batches get smaller (each about half of what is left, down to PIPE_TAIL records): the last
    // one's kernels and download overlap with nothing, so it should be short.  (The offsets at the cuts are looked at
    // before they have been validated: a cut is clamped to (b0, n], and a span that makes no sense is reported by the
    // full check.)
    const int64_t PIPE_TAIL = std::min<int64_t>(32768, PIPE_RECORDS);
    std::vector<int64_t> cuts{0};
    int64_t max_m = 0, max_span = 0;
    while (cuts.back() < n) {
        const int64_t b0 = cuts.back(), left = n - b0;
        int64_t b1 = b0 + std::min(PIPE_RECORDS, left <= PIPE_TAIL + PIPE_TAIL / 2 ? left : std::max(left / 2, PIPE_TAIL));
        if (offs[b1] - offs[b0] > PIPE_BYTES) {   // long records: fewer of them (at least one)
            b1 = std::upper_bound(offs + b0 + 1, offs + b1 + 1, offs[b0] + PIPE_BYTES) - offs - 1;
            b1 = std::min(std::max(b1, b0 + 1), n);
        }
        if (offs[b1] < offs[b0]) return (rc = check_offsets(offs, n, "offs")) ? rc : fail(STAG_E_INVALID, "bad offsets");
        cuts.push_back(b1);
        max_m = std::max(max_m, b1 - b0);
        // the 16-byte aligned stretch of the host text that holds records b0 .. b1 - 1
        max_span = std::max(max_span, (offs[b1] - (offs[b0] & ~(int64_t)15) + 15) / 16 * 16);
    }
    if ((rc = validate(cuts[0], cuts[1]))) return rc;      // the first sub-batch before anything is reserved or queued
    for (size_t q = 0; q < std::min<size_t>(PIPE_SLOTS, cuts.size() - 1); ++q) {
        stag_db::Slot &sl = db->slot[q];
        if ((rc = sl.text.reserve(max_span + 16)) || (rc = sl.offs.reserve(max_m + 1)) || (rc = sl.path_node.reserve(max_m * D)) ||
            (rc = sl.path_prob.reserve(max_m * D)) || (rc = sl.path_len.reserve(max_m)) || (rc = sl.lead.reserve(max_m)) ||
            (rc = sl.level_prob.reserve(max_m * K)) || (rc = sl.selected.reserve(max_m)) || (rc = sl.n_aligned.reserve(max_m)) ||
            (rc = sl.status.reserve(max_m)) || (rc = sl.n_char.reserve(max_m)) ||
            (lead_trail && (rc = sl.lead_trail.reserve(2 * max_m))) ||
            (bits_out && (rc = sl.bits.reserve((size_t)max_m * db->W))))
            return fin(rc);
    }
    // (at least one byte of text: a sub-batch made of empty records only still needs the encode buffers.  With
    // a span of 0 they used to stay unallocated and the encode kernel wrote its status to address 0 -- the GPU
    // fault of the differential fuzz, one configuration in a few thousand: one empty record as the whole batch)
    if ((rc = stag_db_reserve(db, max_m, std::max<int64_t>(max_span, 1)))) return fin(rc);
    for (size_t j = 0; j + 1 < cuts.size(); ++j) {
        const int64_t b0 = cuts[j], b1 = cuts[j + 1];
        const int64_t m = b1 - b0;
        stag_db::Slot &sl = db->slot[j % PIPE_SLOTS];
        if (j > 0 && (rc = validate(b0, b1))) return fin(rc);   // (while the sub-batches before are on the device)
        // the slot's previous sub-batch has left the device before its buffers are overwritten
        if (sl.used) PIPE_TRY(hipEventSynchronize(sl.down));
        // text: the 16-byte aligned stretch of the host text that holds records b0 .. b1 - 1 lands at the start of
        // the slot's buffer; the kernels get a base pointer shifted by that stretch's offset, so the record
        // offsets need no rebasing
        const int64_t a0 = offs[b0] & ~(int64_t)15, a1 = offs[b1];
        const int64_t span = (a1 - a0 + 15) / 16 * 16;
        PIPE_TRY(hipMemcpyAsync(sl.text.p, text + a0, a1 - a0, hipMemcpyHostToDevice, db->s_up));
        PIPE_TRY(hipMemcpyAsync(sl.offs.p, offs + b0, (m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, db->s_up));
        if (lead_trail)
            PIPE_TRY(hipMemcpyAsync(sl.lead_trail.p, lead_trail + 2 * b0, 2 * m * sizeof(int32_t), hipMemcpyHostToDevice, db->s_up));
        PIPE_TRY(hipEventRecord(sl.up, db->s_up));
        // kernels: one compute stream, one workspace (the handle's); results into the slot's own buffers
        PIPE_TRY(hipStreamWaitEvent(db->s_run, sl.up, 0));
        Work wk = work_of(db);
        if (bits_out) wk.bits = sl.bits.p;      // rows the caller wants leave through the slot, not the shared workspace
        wk.enc_n_char = sl.n_char.p;
        Out o;
        o.path_node = sl.path_node.p; o.path_prob = sl.path_prob.p; o.path_len = sl.path_len.p; o.lead = sl.lead.p;
        o.level_prob = sl.level_prob.p; o.selected = sl.selected.p; o.n_aligned = sl.n_aligned.p; o.status = sl.status.p;
        const char *vbase = (const char *)sl.text.p - a0;      // record i of the sub-batch = vbase[offs[b0 + i] ..)
        if ((rc = encode_dev(db, db->L, vbase, a0 + span, sl.offs.p, m, min_perc_state, wk.bits, wk.enc_n_char, o.n_aligned,
                             wk.enc_status, wk.win, db->s_run, nullptr, nullptr, /*full_rows=*/bits_out != nullptr,
                             lead_trail ? sl.lead_trail.p : nullptr)) ||
            (rc = walk_dev(db, wk, wk.bits, m, wk.enc_status, o, false, db->s_run)))
            return fin(rc);
        PIPE_TRY(hipEventRecord(sl.done, db->s_run));
        // results: straight into the caller's arrays at the sub-batch's position
        PIPE_TRY(hipStreamWaitEvent(db->s_down, sl.done, 0));
        auto down = [&](void *host, const void *dev, size_t bytes_per_seq) -> hipError_t {
            if (!host) return hipSuccess;
            return hipMemcpyAsync((char *)host + (size_t)b0 * bytes_per_seq, dev, (size_t)m * bytes_per_seq,
                                  hipMemcpyDeviceToHost, db->s_down);
        };
        PIPE_TRY(down(out->path_node, sl.path_node.p, D * 4));
        PIPE_TRY(down(out->path_prob, sl.path_prob.p, D * 8));
        PIPE_TRY(down(out->path_len, sl.path_len.p, 4));
        PIPE_TRY(down(out->n_lead_single, sl.lead.p, 4));
        PIPE_TRY(down(out->level_prob, sl.level_prob.p, (size_t)db->n_keys * 8));
        PIPE_TRY(down(out->selected, sl.selected.p, 4));
        PIPE_TRY(down(out->n_aligned, sl.n_aligned.p, 4));
        PIPE_TRY(down(out->status, sl.status.p, 4));
        PIPE_TRY(down(bits_out, sl.bits.p, (size_t)db->W * 4));
        PIPE_TRY(down(n_char_out, sl.n_char.p, 4));
        PIPE_TRY(hipEventRecord(sl.down, db->s_down));
        sl.used = true;
    }
#undef PIPE_TRY
    hipError_t e = hipStreamSynchronize(db->s_down);
    if (e != hipSuccess) return fin(fail(STAG_E_HIP, "device to host copy failed: %s", hipGetErrorString(e)));
    return fin(STAG_OK);
}

int stag_classify_concat(stag_db_t *db, int32_t n_genes, const int32_t *gene_len, const uint32_t *const *rows,
                         const int32_t *words_per_row, const int64_t *n_rows, const int32_t *src_row, int64_t n_genomes,
                         const stag_results_t *out, uint32_t *bits_out) {
    if (!db || n_genes < 1 || n_genomes < 0 || !gene_len || !rows || !words_per_row || !n_rows || !out ||
        (n_genomes > 0 && (!src_row || !out->status)))
        return fail(STAG_E_INVALID, "bad argument");
    std::vector<int64_t> goff(n_genes + 1, 0), base(n_genes, 0);
    int64_t arena_words = 0;
    for (int p = 0; p < n_genes; ++p) {
        if (gene_len[p] < 0 || n_rows[p] < 0 || words_per_row[p] < 1 || (int64_t)words_per_row[p] * 32 < gene_len[p] ||
            (n_rows[p] > 0 && !rows[p]))
            return fail(STAG_E_INVALID, "gene %d: bad length / row shape", p);
        goff[p + 1] = goff[p] + gene_len[p];
        base[p] = arena_words;
        arena_words += n_rows[p] * words_per_row[p];
    }
    if (goff[n_genes] != db->L)
        return fail(STAG_E_INVALID, "gene lengths add up to %lld features, the database has %d", (long long)goff[n_genes], db->L);
    for (int64_t i = 0; i < n_genomes * n_genes; ++i)
        if (src_row[i] >= n_rows[i % n_genes]) return fail(STAG_E_INVALID, "src_row[%lld] out of range", (long long)i);
    if (n_genomes == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    if ((rc = reserve_walk(db, n_genomes)) || (rc = db->bits.reserve((size_t)n_genomes * db->W))) return rc;
    DevBuf<uint32_t> arena;
    DevBuf<int64_t> d_tab;      // goff [n_genes+1], base [n_genes]
    DevBuf<int32_t> d_itab;     // words_per_row [n_genes], src_row [n_genomes * n_genes]
    auto done = [&](int code) { arena.release(); d_tab.release(); d_itab.release(); return code; };
    if ((rc = arena.reserve(arena_words + 1)) || (rc = d_tab.reserve(2 * n_genes + 1)) ||
        (rc = d_itab.reserve(n_genes + n_genomes * n_genes)))
        return done(rc);
    bool ok = true;
    for (int p = 0; p < n_genes && ok; ++p)
        if (n_rows[p])
            ok = hipMemcpy(arena.p + base[p], rows[p], (size_t)n_rows[p] * words_per_row[p] * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(d_tab.p, goff.data(), (n_genes + 1) * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_tab.p + n_genes + 1, base.data(), n_genes * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_itab.p, words_per_row, n_genes * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_itab.p + n_genes, src_row, (size_t)n_genomes * n_genes * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) return done(fail(STAG_E_HIP, "host to device copy failed"));
    if (n_genomes > 0x7fffffff || (db->W + 255) / 256 > 65535) return done(fail(STAG_E_INVALID, "batch too large"));
    CHK_TEMP_CLEAR();
    CHK_UPLOAD();
    trace_launch(nullptr, "k_concat_bits n=%lld", (long long)n_genomes);
    hipLaunchKernelGGL(k_concat_bits, dim3((unsigned)n_genomes, (unsigned)((db->W + 255) / 256)), dim3(256), 0, nullptr, n_genomes,
                       n_genes, d_tab.p, d_tab.p + n_genes + 1, d_itab.p, d_itab.p + n_genes, arena.p, db->bits.p, db->W);
    if ((rc = launch_check("k_concat_bits"))) return done(rc);
    if ((rc = walk_dev(db, work_of(db), db->bits.p, n_genomes, nullptr, resolve(db, nullptr), true, nullptr))) return done(rc);
    if ((rc = copy_results_to_host(db, out, n_genomes, nullptr))) return done(rc);
    if (bits_out && hipMemcpy(bits_out, db->bits.p, (size_t)n_genomes * db->W * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(STAG_E_HIP, "device to host copy failed"));
    return done(STAG_OK);
}

int stag_backtranslate_dev(const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs, const char *d_gene,
                           const int64_t *d_gene_offs, int64_t n, int32_t check_length, char *d_out, int64_t out_bytes,
                           int64_t *d_out_begin, int64_t *d_out_end, int32_t *d_status, void *stream) {
    if (n < 0 || prot_bytes < 0 || (n > 0 && (!d_prot || !d_prot_offs || !d_gene || !d_gene_offs || !d_out || !d_out_begin ||
                                               !d_out_end || !d_status)))
        return fail(STAG_E_INVALID, "bad argument");
    if (out_bytes < 3 * prot_bytes) return fail(STAG_E_INVALID, "d_out must hold 3 * prot_bytes = %lld bytes", (long long)(3 * prot_bytes));
    if (n == 0) return STAG_OK;
    if ((n + 3) / 4 > 0x7fffffff) return fail(STAG_E_INVALID, "batch too large");
    CHK_TEMP_CLEAR();
    CHK_TEMP(d_prot, prot_bytes); CHK_TEMP(d_prot_offs, (n + 1) * 8); CHK_TEMP(d_gene_offs, (n + 1) * 8);
    CHK_TEMP(d_out, out_bytes); CHK_TEMP(d_out_begin, n * 8); CHK_TEMP(d_out_end, n * 8); CHK_TEMP(d_status, n * 4);
#ifdef STAG_CHECKED
    {   // the gene text has no size argument: it ends where the last gene does
        int64_t gene_bytes = 0;
        HIP_TRY(hipMemcpy(&gene_bytes, d_gene_offs + n, sizeof gene_bytes, hipMemcpyDeviceToHost));
        CHK_TEMP(d_gene, gene_bytes);
    }
#endif
    CHK_UPLOAD();
    trace_launch((hipStream_t)stream, "k_backtranslate n=%lld prot_bytes=%lld", (long long)n, (long long)prot_bytes);
    hipLaunchKernelGGL(k_backtranslate, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_prot, d_prot_offs, (const uint8_t *)d_gene, d_gene_offs, n, (int)check_length,
                       (uint8_t *)d_out, d_out_begin, d_out_end, d_status);
    return launch_check("k_backtranslate");
}

static int classify_protein_dev(stag_db_t *db, const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs,
                                const char *d_gene, const int64_t *d_gene_offs, int64_t n, double min_perc_state,
                                const stag_results_t *d_out, void *stream, bool full_rows);

int stag_classify_protein_a2m_dev(stag_db_t *db, const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs,
                                  const char *d_gene, const int64_t *d_gene_offs, int64_t n, double min_perc_state,
                                  const stag_results_t *d_out, void *stream) {
    return classify_protein_dev(db, d_prot, prot_bytes, d_prot_offs, d_gene, d_gene_offs, n, min_perc_state, d_out, stream,
                                /*full_rows=*/false);
}

static int classify_protein_dev(stag_db_t *db, const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs,
                                const char *d_gene, const int64_t *d_gene_offs, int64_t n, double min_perc_state,
                                const stag_results_t *d_out, void *stream, bool full_rows) {
    if (!db || n < 0 || prot_bytes < 0 || (n > 0 && (!d_prot || !d_prot_offs || !d_gene || !d_gene_offs)))
        return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    // (4 bytes of room per alignment character: every record's codon alignment then starts on a dword, which is what
    // lets k_backtranslate_fast move everything in dwords)
    const int64_t bt_bytes = (4 * prot_bytes + 31) / 16 * 16;
    int rc;
    if ((rc = stag_db_reserve(db, n, bt_bytes)) || (rc = db->bt_text.reserve(bt_bytes)) || (rc = db->bt_begin.reserve(n)) ||
        (rc = db->bt_end.reserve(n)) || (rc = db->bt_status.reserve(n)))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    {
        if ((n + 3) / 4 > 0x7fffffff) return fail(STAG_E_INVALID, "batch too large");
        CHK_TEMP_CLEAR();
        CHK_TEMP(d_prot, prot_bytes); CHK_TEMP(d_prot_offs, (n + 1) * 8); CHK_TEMP(d_gene_offs, (n + 1) * 8);
#ifdef STAG_CHECKED
        {   // the gene text has no size argument: it ends where the last gene does
            int64_t gene_bytes = 0;
            HIP_TRY(hipMemcpy(&gene_bytes, d_gene_offs + n, sizeof gene_bytes, hipMemcpyDeviceToHost));
            CHK_TEMP(d_gene, gene_bytes);
        }
#endif
        CHK_UPLOAD();
        Prof p(db, st, 4);
        trace_launch(st, "k_backtranslate_fast n=%lld prot_bytes=%lld", (long long)n, (long long)prot_bytes);
        hipLaunchKernelGGL(k_backtranslate_fast, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const uint8_t *)d_prot,
                           prot_bytes, d_prot_offs, (const uint8_t *)d_gene, d_gene_offs, n, db->bt_text.p, db->bt_begin.p,
                           db->bt_end.p, db->bt_status.p);
        if ((rc = launch_check("k_backtranslate_fast"))) return rc;
    }
    const Out o = resolve(db, d_out);
    if ((rc = encode_dev(db, db->L, (const char *)db->bt_text.p, bt_bytes, db->bt_begin.p, n, min_perc_state, db->bits.p,
                         db->enc_n_char.p, o.n_aligned, db->enc_status.p, db->win.p, st, db->bt_end.p, db->bt_status.p,
                         full_rows)))
        return rc;
    return walk_dev(db, work_of(db), db->bits.p, n, db->enc_status.p, o, false, st);
}

int stag_classify_protein_a2m(stag_db_t *db, const char *prot, const int64_t *prot_offs, const char *gene,
                              const int64_t *gene_offs, int64_t n, double min_perc_state, const stag_results_t *out,
                              uint32_t *bits_out, int32_t *n_char_out) {
    if (!db || n < 0 || !out || (n > 0 && (!prot || !prot_offs || !gene || !gene_offs || !out->status)))
        return fail(STAG_E_INVALID, "bad argument");
    if (n == 0) return STAG_OK;
    HIP_TRY(hipSetDevice(db->device));
    int rc;
    if ((rc = check_offsets(prot_offs, n, "prot_offs")) || (rc = check_offsets(gene_offs, n, "gene_offs"))) return rc;
    const int64_t pb = prot_offs[n], gb = gene_offs[n];
    if ((rc = db->text.reserve(pb + 16)) || (rc = db->offs.reserve(n + 1)) || (rc = db->gene_text.reserve(gb + 16)) ||
        (rc = db->gene_offs.reserve(n + 1)))
        return rc;
    HIP_TRY(hipMemcpy(db->text.p, prot, pb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db->offs.p, prot_offs, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db->gene_text.p, gene, gb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db->gene_offs.p, gene_offs, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if ((rc = classify_protein_dev(db, (const char *)db->text.p, pb, db->offs.p, (const char *)db->gene_text.p,
                                   db->gene_offs.p, n, min_perc_state, nullptr, nullptr, /*full_rows=*/bits_out != nullptr)))
        return rc;
    if ((rc = copy_results_to_host(db, out, n, nullptr))) return rc;
    if (bits_out) HIP_TRY(hipMemcpy(bits_out, db->bits.p, (size_t)n * db->W * 4, hipMemcpyDeviceToHost));
    if (n_char_out) HIP_TRY(hipMemcpy(n_char_out, db->enc_n_char.p, n * 4, hipMemcpyDeviceToHost));
    return STAG_OK;
}

int stag_backtranslate(const char *prot, int64_t prot_len, const char *gene, int64_t gene_len, int32_t check_length,
                       char *out, int64_t *out_len) {
    if (!prot || !gene || !out || !out_len || prot_len < 0 || gene_len < 0) return fail(STAG_E_INVALID, "bad argument");
    int64_t residues = 0;
    for (int64_t i = 0; i < prot_len; ++i) residues += prot[i] != '-';
    if (check_length && gene_len != 3 * residues && gene_len - 3 != 3 * residues) {
        *out_len = -1;
        return STAG_OK;
    }
    int64_t g = 0, o = 0;
    for (int64_t i = 0; i < prot_len; ++i) {
        const unsigned char c = (unsigned char)prot[i];
        if (c == '-') {
            out[o++] = '-'; out[o++] = '-'; out[o++] = '-';
        } else if (c >= 'A' && c <= 'Z') {
            for (int64_t j = g; j < g + 3 && j < gene_len; ++j) out[o++] = gene[j];
            g += 3;
        } else if (c >= 'a' && c <= 'z') {
            g += 3;
        }
    }
    *out_len = o;
    return STAG_OK;
}

void *stag_host_alloc(int64_t bytes) {
    void *p = nullptr;
    if (bytes < 0) { fail(STAG_E_INVALID, "negative size"); return nullptr; }
    const hipError_t e = hipHostMalloc(&p, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) { fail(STAG_E_NOMEM, "hipHostMalloc(%lld): %s", (long long)bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

void stag_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int stag_profile_enable(stag_db_t *db, int32_t on) {
    if (!db) return fail(STAG_E_INVALID, "null db");
    db->prof = on != 0;
    return STAG_OK;
}

int stag_profile_read(stag_db_t *db, double *ms_per_class, int64_t *launches_per_class) {
    if (!db || !ms_per_class || !launches_per_class) return fail(STAG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(db->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < STAG_PROF_CLASSES; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
    for (auto &r : db->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ms_per_class[r.cls] += ms;
            launches_per_class[r.cls] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    db->recs.clear();
    return STAG_OK;
}

}  // extern "C"
