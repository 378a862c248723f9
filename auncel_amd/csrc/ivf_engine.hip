// Host side of the gfx950 IVF-Flat engine: device-resident index, work-list construction,
// round scheduling for the adaptive (Auncel) search, and the extern "C" boundary declared in
// include/auncel_amd.h.  No CPU compute path exists here: distances and selection always run in
// the kernels of ivf_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/auncel_amd.h"
#include "ivf_kernels.h"
#include "kmeans_host.h"

using namespace amdivf;

namespace {

thread_local std::string g_last_error;

struct EngineError : std::runtime_error {  // what the reference raises as FaissException
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr); \
    } while (0)

// ------------------------------------------------------------------------------------ options
// Policy and tuning choices of an index, set through the ABI (amd_ivf_set_option / amd_ivf_get_option; the reference exposes
// such choices as index fields, Auncel/IndexIVF.h:97-143, and through ParameterSpace / c_api/IndexIVF_c.h:82-85).  A value that
// was never set is OPT_UNSET: the debugging environment variable of the same meaning decides then, and after it the built-in
// default -- the environment is a debugging aid, never the only way to a behaviour.
constexpr double OPT_UNSET = -1e300;
constexpr double ROW_LISTS_DEFAULT = -1;
constexpr uint32_t CL_ARENA = 4u << 20;  // entries of CompactArgs::arena (32 MiB)
enum OptId {
    OPT_COARSE_TIES,    // order inside runs of bit-equal coarse distances: 0 centroid number, 1 the reference's heap, 2 "redo"
                        // (search again the queries that read such a run); unset: heap for calls of < 20 queries, else 0
    OPT_SELECT,         // 0 the reference's heap replayed for every query, 1 sorted arrays + tie_fix_kernel (k <= 128)
    OPT_TIE_FIX,        // 0 one pass at the end, 1 behind every round on a side stream; unset: by call size and concurrency
    OPT_FILTER,         // fp32 threshold rounds: matrix-core filter + exact rescoring over fp16 (2) or fp32 (1) copies of the lists, or the
                        // vector ALU throughout (0)
    OPT_FIXED_ROUNDS,   // fixed-nprobe searches: 1 one dense round, 2 dense + threshold round; unset: by nprobe
    OPT_ROUND_FIRST,    // adaptive search: probes of the first round (12)
    OPT_ROUND_GROW,     // ... factor by which later rounds grow (12 byte codes, 6 fp32 filter, 3.5 fp32)
    OPT_ROUND_INC,      // ... probes a later round adds at least
    OPT_DIRECT_OUT,     // results written straight into page-locked caller buffers (1) or copied at the end (0)
    OPT_SCAN_PIPELINED, // byte-code scan: bit 0 dense, bit 1 threshold rounds through scan_mfma_thr_kernel, bit 2 threshold rounds with 64 queries
                        // per item through scan_mfma_pair_kernel (7: all; 0: scan_mfma_kernel)
    OPT_PLAN_FUSED,     // round planning in three launches (1) or seven (0)
    OPT_COARSE_PICK,    // exact coarse top-nprobe of large calls from matrix-core distances (2: fp16 operands, 1: fp32) + exact recomputation of the candidates, or
                        // from exact distances to every centroid (0)
    OPT_PHASE_TIMING,   // HIP events around every phase of a search (amd_ivf_last_timing): 1 always, 0 never; unset: calls of >= 20 queries
    OPT_PINNED_IO,      // per-call inputs / outputs through one page-locked block read and written by kernels (1) or by copies (0)
    OPT_ROW_LISTS,      // threshold rounds of calls of >= 256 queries: the rows' marked candidates compacted into short lists before the
                        // selection (1, compact_rows_kernel) or found by the selection itself in the masks (0); unset: lists when no
                        // other search of the index is running
    OPT_LANES,          // fp32 dense rounds from a lane-ordered copy of the lists (1, scan_lanes_kernel: one coalesced KiB per 64 vectors and
                        // step, no LDS staging) or from the rows (0, scan_tiles_kernel); the copy costs the lists' bytes once more
    OPT_FP32_IN_FLIGHT, // large fp32 searches of one index that run at a time (4, the measured optimum: 1.43 / 1.42 / 1.37 / 1.33 M queries/s
                        // at 4 / 5 / 6 / 8 at a time); the others wait inside their calls
    OPT_COALESCE,       // asynchronous adaptive searches: queued tickets over adjacent resident ranges (same parameters, adjacent result
                        // buffers) that one pass over the lists may serve together (1: every ticket its own pass)
    N_OPT
};
struct OptSpec {
    const char* key;
    const char* env;
    const char* words;  // "name=value,..." for environment variables that hold a word
};
const OptSpec OPT_TABLE[N_OPT] = {
    {"coarse_ties", "AUNCEL_AMD_COARSE_TIES", "id=0,heap=1,redo=2"},
    {"select", "AUNCEL_AMD_SELECT", "heap=0,sorted=1"},
    {"tie_fix", "AUNCEL_AMD_TIE_FIX", "final=0,eager=1"},
    {"filter", "AUNCEL_AMD_FILTER", nullptr},
    {"fixed_rounds", "AUNCEL_AMD_FIXED_ROUNDS", nullptr},
    {"round_first", "AUNCEL_AMD_ROUND_FIRST", nullptr},
    {"round_grow", "AUNCEL_AMD_ROUND_GROW", nullptr},
    {"round_inc", "AUNCEL_AMD_ROUND_INC", nullptr},
    {"direct_out", "AUNCEL_AMD_DIRECT_OUT", nullptr},
    {"scan_pipelined", "AUNCEL_AMD_SCAN_PIPELINED", nullptr},
    {"plan_fused", "AUNCEL_AMD_PLAN_FUSED", nullptr},
    {"coarse_pick", "AUNCEL_AMD_COARSE_PICK", nullptr},
    {"phase_timing", "AUNCEL_AMD_PHASE_TIMING", nullptr},
    {"pinned_io", "AUNCEL_AMD_PINNED_IO", nullptr},
    {"row_lists", "AUNCEL_AMD_ROW_LISTS", nullptr},
    {"lanes", "AUNCEL_AMD_LANES", nullptr},
    {"fp32_in_flight", "AUNCEL_AMD_FP32_IN_FLIGHT", nullptr},
    {"coalesce", "AUNCEL_AMD_COALESCE", nullptr},
};
struct Options {
    // (atomic: amd_ivf_set_option on the owner may run while search contexts cloned from it are searching; a search reads the
    // options that shape its launches ONCE -- run_rounds_device's snapshot -- so a change takes effect with the next search)
    std::atomic<double> v[N_OPT];
    Options() {
        for (auto& x : v) x.store(OPT_UNSET, std::memory_order_relaxed);
    }
    // the value set through the ABI, else the environment's (read per call: the tests flip it inside one process), else `dflt`
    double get(OptId id, double dflt) const {
        const double set = v[id].load(std::memory_order_relaxed);
        if (set != OPT_UNSET) return set;
        const char* e = getenv(OPT_TABLE[id].env);
        if (!e || !*e) return dflt;
        if (const char* w = OPT_TABLE[id].words) {
            const size_t n = strlen(e);
            for (const char* p = w; *p;) {
                const char* eq = strchr(p, '=');
                if ((size_t)(eq - p) == n && !strncmp(p, e, n)) return atof(eq + 1);
                const char* c = strchr(eq, ',');
                if (!c) break;
                p = c + 1;
            }
            return dflt;  // an unknown word
        }
        return atof(e);
    }
    bool is_set(OptId id) const {
        return v[id].load(std::memory_order_relaxed) != OPT_UNSET || (getenv(OPT_TABLE[id].env) && *getenv(OPT_TABLE[id].env));
    }
};

// Range of a set of fp32 values when all of them are integers (else ok = false).  When every operand of a
// scan is an integer of magnitude <= 4095, x - y and (x - y)^2 (or x * y) are exactly representable, so
// fma(t, t, acc) and acc + t * t round identically: the scan kernel may then fuse (2 VALU ops per element
// instead of 3) without changing a single bit of any distance.
struct IntRange {
    bool ok = true;
    float lo = 0.f, hi = 0.f;
    void add(const float* x, size_t n) {
        if (!ok) return;
        float l = lo, h = hi;
        bool good = true;
        for (size_t i = 0; i < n; i++) {
            const float v = x[i];
            good &= (v == (float)(int)v) & (v >= -4095.f) & (v <= 4095.f);
            l = v < l ? v : l;
            h = v > h ? v : h;
        }
        ok = good;
        lo = l;
        hi = h;
    }
    void merge(const IntRange& o) {
        ok = ok && o.ok;
        lo = std::min(lo, o.lo);
        hi = std::max(hi, o.hi);
    }
    // Inner product: |x|, |y| <= 4095 keeps every product below 2^24.  L2 squares the difference, so it is the spread of the
    // two ranges together that must stay within 4096 (|x - y| <= 4096, (x - y)^2 <= 2^24): operands of opposite signs
    // up to 4095 each would give differences up to 8190, whose squares fp32 no longer holds exactly.
    bool fusable_with(const IntRange& o, int metric) const {
        if (!(ok && o.ok)) return false;
        const float l = std::min(lo, o.lo), h = std::max(hi, o.hi);
        if (metric == METRIC_L2) return h - l <= 4096.f;
        return h <= 4095.f && l >= -4095.f;
    }
    // Byte-code scan (scan_tiles_kernel, ARITH 2): both sides hold integers 0..255 and no sum of d products can pass
    // 2^24, so the reference's fp32 partial sums are exact integers in any order and integer arithmetic gives the
    // same fp32 distance bit for bit.
    bool bytes() const { return ok && lo >= 0.f && hi <= 255.f; }
    bool bytes_with(const IntRange& o, size_t d) const {
        const double m = std::max(hi, o.hi);
        return bytes() && o.bytes() && (double)d * m * m <= 16777216.0;
    }
};

// the most row space the rounds of a search shape have wanted so far, for the last few shapes (a caller that alternates between
// shapes -- byte codes and fp32, a large call and the small second pass of the exact tie order -- would otherwise size every
// search as if it were the first of its kind, and grow its workspace again right after)
struct WantHistory {
    static constexpr int N = 8;
    uint64_t sig[N] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t want[N] = {0, 0, 0, 0, 0, 0, 0, 0};
    int next = 0;
    size_t get(uint64_t s) const {
        for (int i = 0; i < N; i++)
            if (sig[i] == s && want[i]) return want[i];
        return 0;
    }
    void raise(uint64_t s, size_t w) {
        for (int i = 0; i < N; i++)
            if (sig[i] == s && want[i]) {
                want[i] = std::max(want[i], w);
                return;
            }
        sig[next] = s;
        want[next] = w;
        next = (next + 1) % N;
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    uint64_t touch = 0;  // calls of ensure(): every writer of a workspace sizes it first (coarse_dev's tail book-keeping)
    void ensure(size_t bytes) {
        touch++;
        if (bytes <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    // the address kernels use for this memory (page-locked host memory is mapped into the device's address space)
    void* dev() const {
        void* d = nullptr;
        HIP_CHECK(hipHostGetDevicePointer(&d, p, 0));
        return d;
    }
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
};

struct EventTimer {
    struct Span {
        hipEvent_t a, b;
        int cat;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    // off: a call of a few queries is all launch latency, and a pair of event records around every phase is a dozen stream
    // operations more (the "phase_timing" option; amd_ivf_last_timing then reports zeros for the phases)
    bool off = false;
    static constexpr size_t NONE = ~(size_t)0;
    size_t begin(int cat, hipStream_t s) {
        if (off) return NONE;
        Span sp{get(), get(), cat};
        HIP_CHECK(hipEventRecord(sp.a, s));
        spans.push_back(sp);
        return spans.size() - 1;
    }
    void end(size_t i, hipStream_t s) {
        if (i != NONE) HIP_CHECK(hipEventRecord(spans[i].b, s));
    }
    // call after the stream has been synchronised
    void collect(double* ms_by_cat, int ncat, double* launches_by_cat) {
        for (int c = 0; c < ncat; c++) ms_by_cat[c] = 0, launches_by_cat[c] = 0;
        for (auto& sp : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
                ms_by_cat[sp.cat] += ms;
                launches_by_cat[sp.cat] += 1;
            }
            pool.push_back(sp.a);
            pool.push_back(sp.b);
        }
        spans.clear();
    }
    ~EventTimer() {
        for (auto& sp : spans) {
            (void)hipEventDestroy(sp.a);
            (void)hipEventDestroy(sp.b);
        }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// phases timed by HIP events on the streams they run on (amd_ivf_last_timing: coarse, scan = dense + threshold rounds,
// select = both selections + tie_fix_kernel; amd_ivf_last_timing_detail: every phase by itself)
enum { CAT_COARSE = 0, CAT_SCAN = 1, CAT_SELECT = 2, CAT_SCAN_THR = 3, CAT_SELECT_THR = 4, CAT_TIE_FIX = 5, CAT_PLAN = 6, NCAT = 7 };

}  // namespace

struct amd_ivf;
static void async_shutdown(amd_ivf* h);  // (stops the handle's worker threads and frees their contexts)

struct amd_ivf {
    int d = 0, dpad = 0, metric = METRIC_L2, device = 0;
    size_t nlist = 0, ntotal = 0;
    hipStream_t stream = nullptr;

    // host mirror of the inverted lists (ArrayInvertedLists layout) + device CSR copy
    std::vector<std::vector<float>> h_codes;  // rows padded to dpad
    std::vector<std::vector<int64_t>> h_ids;
    std::vector<uint64_t> h_list_off;
    bool lists_dirty = true;
    DevBuf d_codes, d_ids, d_list_off, d_centroids, d_centroid_norms;
    std::vector<float> h_centroids;  // nlist x dpad
    bool have_centroids = false;

    // resident queries
    DevBuf d_resident;
    size_t n_resident = 0;
    IntRange db_range, centroid_range, resident_range, call_range;  // see IntRange
    int allow_fused = 1;
    int allow_bytes = 1;
    Options opt;  // on the index owner (search contexts read their owner's): amd_ivf_set_option
    // small device-to-host copies go through page-locked staging and are handed out after the call's synchronisation: a copy
    // into pageable memory blocks the host for ~20 us each, and a search ends with up to nine of them (d2h_small / flush_small)
    PinnedBuf p_small;
    size_t small_used = 0;
    struct SmallCopy {
        void* dst;
        size_t off, bytes;
        const void* src;  // device source still to be gathered into the staging block by copy_segs_kernel (null: copied already)
    };
    std::vector<SmallCopy> small;
    // host-to-device counterpart: inputs of a call are packed into page-locked staging and scattered by one kernel (h2d_small)
    PinnedBuf p_stage;
    size_t stage_used = 0;
    CopySegs h2d_pending{};
    std::vector<std::function<void()>> after_flush;  // run by sync_and_flush behind the synchronisation (deferred epilogues)
    bool want_first_tie = false;        // adaptive_slice: also leave the start of each ranking's first run of equal distances
    size_t first_tie_nreal = 0;
    DevBuf w_first_tie;
    std::vector<uint32_t> first_tie_host;
    // AUNCEL_AMD_COARSE_TIES=redo: the rankings that may need the reference's heap order (first run of equal distances within
    // the window a query can read in its first two rounds) are set aside and re-ranked on a side stream WHILE the first pass
    // searches; the second pass takes its rankings from those slots instead of running the heap (2.75 ms at nlist 4096) itself
    DevBuf w_spec_full, w_spec_dis, w_spec_keys, w_spec_count, w_spec_slot, w_spec_pick, w_redo_idx, w_spec_query, w_spec_scratch, w_split;
    uint32_t spec_cap = 512;        // slots of this search (adaptive_slice)
    bool spec_inline = false;       // the heap's order was applied to the first pass itself (launch_tie_patch): only what it could not fix is searched again
    uint32_t tie_patched_host = 0;  // rankings of the last first pass that the heap's order changed
    hipStream_t spec_stream = nullptr;  // (= bg_stream)
    hipEvent_t ev_spec_go = nullptr, ev_spec_done = nullptr;
    bool spec_wanted = false;  // set by adaptive_redo_ties around its first pass (small calls repeat as a whole: no slots)
    bool spec_done = false;   // run_rounds_device: the search ended at its first look and the caller's read-backs came with it
    // a call of at most four queries: init_state / byte_queries record their launches here instead of making them, and adaptive_slice
    // makes them as one (launch_small_state)
    struct SmallFuse {
        bool active = false, have_init = false, have_bytes = false;
        InitStateArgs init{};
        const float* bx = nullptr;
        int8_t* bout = nullptr;
        int32_t* bcx = nullptr;
    } fuse;
    bool spec_valid = false;  // the slots of the last first pass are (being) re-ranked
    bool spec_use = false;    // second pass: ranking row j comes from slot w_spec_pick[j]
    size_t spec_ncopy = 0;    // leading entries of a ranking the slots hold
    std::vector<int32_t> spec_slot_host;
    uint64_t tie_rows_host = 0;  // rankings taken from re-ranked slots (amd_ivf_coarse_tie_rows adds them)
    int ties_override = -1;  // coarse_dev: -1 as AUNCEL_AMD_COARSE_TIES / the call size say, 0 centroid-number order, 1 the reference's heap
    int last_arith = 0;  // scan arithmetic of the last search: 0 reference order, 1 fused, 2 byte codes
    // byte copy of the lists in MFMA fragment order (ivf_kernels.h) + per-slot constants, kept while the data qualifies
    // (IntRange::bytes); block_off[l] = first 32-vector block of list l
    DevBuf d_frag, d_cy, d_block_off;
    std::vector<uint64_t> h_block_off;
    bool have_codes8 = false;
    // fp32 copy of the lists in fragment order + |y|^2 / |y| per slot (ivf_filter.hip), kept when the data does not qualify for
    // byte codes: threshold rounds run as a matrix-core filter over it, the exact distance only for what the filter keeps
    DevBuf d_frag32, d_yn;
    std::mutex fp32_gate_mu;               // large fp32 searches running on this index (option fp32_in_flight)
    std::condition_variable fp32_gate_cv;
    int fp32_running = 0;
    uint64_t fp32_gate_next = 0, fp32_gate_serving = 0;  // (first come, first served)
    uint64_t async_served[2] = {0, 0};  // tickets / passes of asynchronous pools that were shut down (amd_ivf_async_counts)
    double probed_len = 0;                 // expected length of the list a query probes: sum(len^2) / sum(len) (upload_lists)
    DevBuf d_lanes;                        // the fp32 lists in lane order (ScanArgs::lanes), built by the first fp32 dense round
    std::atomic<int> lanes_state{0};       // 0 not tried, 1 there, -1 not possible
    std::atomic<bool> have_frag32{false};  // (read without the lock by search contexts: ensure_frag32's double-checked creation)
    bool frag32_possible = false;
    // the fp16 form of the filter's list copy (ivf_filter.hip): scaled halves in fragment order + the range they were scaled by
    DevBuf d_frag16, d_yinfo;
    std::atomic<int> frag16_state{0};      // 0 not built yet, 1 there, -1 the lists have no usable scale (the fp32 form serves)
    int allow_filter = 1;
    DevBuf w_xf, w_xn, w_surv, w_surv_cnt;  // packed queries + norms of the current search, the filter's survivors
    DevBuf w_qinfo, w_fparams;              // fp16 form: the queries' range, the search's FilterParams

    // Auncel state
    DevBuf d_interdis;
    bool have_interdis = false;
    DevBuf d_arcos, d_trace_off, d_trace_x, d_trace_y, d_trace_std;
    size_t tuner_max_topk = 0, tuner_ntraces = 0, tuner_trace_cap = 0;
    bool have_tuner = false;

    // workspaces (grow only)
    DevBuf w_qtile, w_group_p0, w_group_cnt, w_xnorms;
    PinnedBuf p_group_p0, p_group_cnt, p_counters;
    DevBuf w_pl_cnt, w_pl_need, w_pl_dist_base, w_pl_lcount, w_pl_lstart, w_pl_gbase, w_pl_ibase, w_pl_fill, w_pl_counters, w_seg_slot;
    DevBuf w_x8, w_xnorm8;  // signed byte copy of the current queries + per-query constants (launch_sbytes_from_f32)
    const float* x8_src = nullptr;  // what w_x8 holds when it is a slice of the resident queries (byte_queries)
    size_t x8_n = 0;
    uint64_t x8_gen = 0, resident_gen = 1;
    DevBuf w_rcount, w_roff, w_rlab, w_rdis;  // range search: per-query counts / output offsets / results of a round
    std::vector<size_t> r_lims;              // results of the last range search (amd_ivf_range_results)
    std::vector<int64_t> r_labels;
    std::vector<uint32_t> r_part_pos;  // amd_ivf_scan_codes_range: positions / distances of the last call
    std::vector<float> r_part_dis;
    std::vector<float> r_dist;
    DevBuf w_thr, w_mask, w_pl_pad, w_cl_cnt, w_cl_ent, w_cl_arena, w_cl_cursor;  // threshold mode of the device-planned rounds: heap tops, candidate bit masks
    DevBuf w_x, w_dist, w_items, w_pair_query, w_pair_out, w_seg_off, w_seg_list, w_seg_count, w_qsel;
    DevBuf w_heap_val, w_heap_ref, w_stage, w_nscan, w_done, w_pre_val, w_stoped, w_dtb, w_D, w_I;
    DevBuf w_cdis, w_ckeys, w_stats, w_error, w_misc, w_misc2, w_misc3, w_rawptrs, w_sub_off;
    struct CoarseTail {  // what the last prefix ranking left behind the prefix of every row of w_cdis / w_ckeys (coarse_dev)
        bool valid = false;
        const void* dis = nullptr;
        const void* keys = nullptr;
        size_t prefix = 0, nprobe = 0, rows = 0;
        int metric = 0;
        uint64_t touch_dis = 0, touch_keys = 0;
    } coarse_tail;
    DevBuf c_heap_val, c_heap_ref, c_stage, c_nscan, c_done, c_seg_off, c_seg_list, c_seg_count;
    float centroid_norm_max = 0.f;  // max |c|^2 over the centroids, rounded up
    DevBuf d_cinfo;                 // range of the centroid table (launch_amax): the fp16 form of the approximate coarse ranking
    bool in_coarse_pick = false;    // (the exact re-run of flagged queries is inside coarse_dev)
    size_t coarse_picked = 0;       // rankings of the last coarse_dev call that came from coarse_pick_kernel
    DevBuf c_pick_flag, c_pick_x, c_pick_dis, c_pick_keys;  // coarse_pick_kernel: flagged queries and their exact re-run
    PinnedBuf p_pick;
    DevBuf c_pair_query, c_pair_out, c_items, c_group_p0, c_group_cnt;  // the coarse quantiser's work list (coarse_dev), kept
    uint64_t coarse_sig = 0;                                             // while its signature (chunk, queries, nlist) repeats
    bool coarse_sig_valid = false;
    PinnedBuf p_items, p_pair_query, p_pair_out, p_seg_off, p_seg_list, p_seg_count, p_qsel, p_seg_begin;
    DevBuf w_seg_begin;
    DevBuf w_log, w_log_cnt, w_amb, w_tie_flag;  // sorted-array selection: admission logs, ambiguity marks, tie_fix flags
    DevBuf w_qstat;     // per query: lists scanned, heap updates (what a query searched again takes out of the statistics)
    DevBuf w_redo_x;    // rows of the queries searched again with the reference's coarse tie order (adaptive_core)
    DevBuf w_log_snap, w_fin_round, w_fix_pos, w_fix_val, w_fix_ref;  // tie_fix_kernel: per-round log counts, heaps replayed so far
    hipStream_t fix_stream = nullptr;            // tie_fix_kernel runs here, under the next round (= bg_stream)
    hipStream_t bg_stream = nullptr;             // the context's background stream: tie replay beside the next round, the heap order of coarse ties
    hipEvent_t ev_sel = nullptr, ev_fix[2] = {nullptr, nullptr};
    size_t last_state_n = 0;                     // queries of the last search (amd_ivf_last_tie_fixed reads their flags)
    // chained rounds: the planning counters of every round of the last search (grid hints for the next one of the same shape)
    DevBuf w_pl_hist;
    PinnedBuf p_hist;
    std::vector<uint32_t> round_hint;  // [round][16]
    uint64_t hint_sig = 0;
    uint64_t last_tie_redone = 0;  // queries the last adaptive call searched again for the coarse tie order (AUNCEL_AMD_COARSE_TIES=redo)
    uint64_t last_tie_patched = 0;  // ... rankings whose order the heap changed while the call's one pass was under way
    uint32_t row_align_now = 1024;  // run_rounds_device: what the rows of the search under way are padded to (the tie patch moves rows)
    // run_rounds_device: the selection kernels write (D, I) straight into the caller's buffers when those are page-locked and
    // device-visible (rows leave as their queries finish, under the later rounds; no copy at the end); null: w_D / w_I + a copy
    float* out_D = nullptr;
    int64_t* out_I = nullptr;
    int last_direct_out = 0;  // whether the last search did
    uint64_t filter_launches = 0;  // last search: threshold rounds that went through the matrix-core filter (ivf_filter.hip)
    uint64_t hinted_rounds = 0, short_rounds = 0;  // last search: scan launches sized by a hint / of those, grids smaller than the work
    std::atomic<int> live_contexts{1};  // on the index owner: itself + its clones (amd_ivf_clone / amd_ivf_destroy)
    std::atomic<float> tie_rate{-1.f};  // on the index owner: share of the last search's queries in which equal distances met (-1: none yet)
    std::atomic<int> active_searches{0};  // on the index owner: searches inside run_rounds_device right now
    bool force_heap_select = false;  // (set while a search is repeated after ERR_LOG_OVERFLOW)
    // tune / train search over a coarse ranking the caller supplies (amd_ivf_search_adaptive_pre, amd_ivf_train_samples_pre):
    // host rows of this call's (or slice's) queries, given_nprobe entries each; null: the engine ranks the centroids itself
    const int64_t* given_keys = nullptr;
    const float* given_dis = nullptr;
    size_t given_nprobe = 0;
    DevBuf w_limit;  // time-bounded search: per-slot end of the probe loop (plan_counts_kernel -> replay_kernel)
    DevBuf w_tie_rows;  // rankings re-run through the reference's heap because of equal distances (launch_heap_tie_order)

    size_t dist_budget_floats = (size_t)768 << 20;  // 3 GiB of distances per scan launch
    WantHistory dist_want;                           // run_rounds_device: row space the rounds of the last searches wanted, by shape
    // ... and on the index owner, shared by every context cloned from it: a context that has not met the largest slice yet is sized
    // for it all the same (a workspace that grows frees device memory, which waits for every stream of the device: 20-40 ms in
    // which nothing of any context runs -- two or three of those were a tenth of a 20-step timed region)
    std::mutex want_mu;
    WantHistory shared_want;
    size_t stats_host[4] = {0, 0, 0, 0};
    double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double timing_detail[2 * 7 + 2] = {0};  // (ms, launches) per phase of the last search (CAT_*) | min bytes of dense / threshold rounds
    double last_min_bytes = 0;
    double scan_bytes = 0, scan_slots = 0, scan_useful = 0, scan_min_bytes = 0, scan_min_bytes_thr = 0;
    EventTimer timer;

    // Query lanes: a large adaptive batch is cut into slices that run their rounds concurrently, each on
    // its own stream with its own workspaces (kids borrow the index data of `parent`), so that one
    // slice's latency-bound selection and host-side round planning overlap another slice's VALU-bound scan.
    struct AsyncPool* async = nullptr;  // owner only: internal search contexts + worker threads of amd_ivf_submit_* (below)
    std::vector<amd_ivf*> async_ctx;    // those contexts (statistics and settings of the owner cover them)
    int async_depth = 4;
    amd_ivf* parent = nullptr;
    bool is_clone = false;  // made by amd_ivf_clone: a search context of its own over the parent's index data
    std::mutex upload_mu;   // owner only: serialises the first upload of the lists
    std::mutex async_mu;    // owner only: serialises the creation of the asynchronous pool
    std::vector<std::unique_ptr<amd_ivf>> kids;
    // side streams for the sparse tile shapes of a round (fork / join around the dense launch)
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};

    ~amd_ivf() {
        async_shutdown(this);
        kids.clear();
        for (int i = 0; i < 4; i++) {
            if (aux[i]) (void)hipStreamDestroy(aux[i]);
            if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (bg_stream) (void)hipStreamDestroy(bg_stream);
        if (ev_spec_go) (void)hipEventDestroy(ev_spec_go);
        if (ev_spec_done) (void)hipEventDestroy(ev_spec_done);
        if (ev_sel) (void)hipEventDestroy(ev_sel);
        for (int i = 0; i < 2; i++)
            if (ev_fix[i]) (void)hipEventDestroy(ev_fix[i]);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static inline amd_ivf* ix(amd_ivf* h) { return h->parent ? h->parent : h; }
static inline const amd_ivf* ix(const amd_ivf* h) { return h->parent ? h->parent : h; }
static inline double opt(const amd_ivf* h, OptId id, double dflt) { return ix(h)->opt.get(id, dflt); }

namespace {

void use_device(const amd_ivf* h) { HIP_CHECK(hipSetDevice(h->device)); }

// Host waits.  hipStreamSynchronize spins on the completion signal: lowest wake-up latency, one busy core per waiting thread.
// AUNCEL_AMD_BLOCKING_SYNC=1 waits on a blocking event instead (the thread sleeps until the interrupt), for hosts where the
// calling threads outnumber the cores they may use.
hipError_t stream_sync(hipStream_t s) {
    static const bool blocking = [] {
        const char* e = getenv("AUNCEL_AMD_BLOCKING_SYNC");
        return e && *e && *e != '0';
    }();
    if (!blocking) return hipStreamSynchronize(s);
    thread_local hipEvent_t ev = nullptr;
    thread_local int ev_dev = -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!ev || ev_dev != dev) {
        e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess) return e;
        ev_dev = dev;
    }
    e = hipEventRecord(ev, s);
    if (e != hipSuccess) return e;
    return hipEventSynchronize(ev);
}

// The main stream of a context carries its latency-bound work (round planning, ordered selection, transfers); with
// several contexts on one GPU it gets the high priority so that those kernels are not queued behind another context's
// scan workgroups (the scans themselves run on the normal-priority side streams).
hipStream_t make_main_stream() {
    static const bool flat = getenv("AUNCEL_AMD_FLAT_PRIORITY") != nullptr;
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    if (!flat && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) {
        HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    } else {
        HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    return s;
}

// Streams of a context's long background kernels (tie replay beside the next round, the heap order of coarse ties), created at
// first use.  Normal priority: at the lowest the tie replay fell behind and the next search waited for it (2.7 against 2.95 M q/s);
// scans at the high priority of the main streams were worse still (2.1), and so were these streams at it (2.35 against 3.0).  Which hardware queue such a stream shares with which
// other context's scan stream is the runtime's choice at creation time (the least used of GPU_MAX_HW_QUEUES) and depends on every
// stream the process created -- and destroyed -- before: bench.py's side legs (exact_tie_order, fp32_path) move by a factor of two
// with the order they run in; creating these streams with the context did not make that steadier.
hipStream_t make_background_stream() {
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

// The streams of a search context, all of them when the context is made: main (made by the caller; high priority), scan (low),
// background (normal).  The runtime keeps GPU_MAX_HW_QUEUES hardware queues per priority class and hands a new stream the least
// used one of its class: with the three kinds of stream in three classes and the 8 queues per class the engine asks for (below),
// every stream of up to eight contexts has a hardware queue to itself, whatever was created before -- no kernel ever waits behind
// another stream's kernel in a shared queue.  (Round 4 created side streams on first use, all in the normal class: which stream
// shared a queue with which followed from which search happened to need what first, a scan queued behind another context's
// 1.5 ms heap waited for it, and the figures of a bench leg moved by a factor of two with the legs run before it.)  The classes
// are also the order the work should be dispatched in when the chip is full: the latency-bound chain (planning, selection) first,
// the background kernels the chain will wait for next, the bandwidth-bound scans -- grids of thousands of workgroups -- with
// whatever is left.  Measured (profiles/r05_experiments.txt I, K, V): the byte-code scans have since moved to the main stream, so for
// them this stream carries nothing -- and its mere presence is worth 8 % of the headline (3.11-3.17 against 2.81-2.92 M q/s, twice on
// one box: with queues in all three classes the hardware's arbitration between the high-priority main queues and the rest
// changes, and the runtime offers no other handle on it).  Making it on first use and in the normal class instead was tried: the
// headline lost those 8 %, and four fp32 searches in flight -- five normal-class streams each then, twenty on eight queues, with
// fork / join events between streams that share a queue -- collapsed to 25 ms a step.  The fp32 paths keep it as the stream of the
// widest tile shape and of the filter passes, where it starts behind the other shapes (cfg 5: DESIGN.md 3.2b).
void ensure_context_streams(amd_ivf* h) {
    if (h->bg_stream) return;
    static const char* scan_prio_env = getenv("AUNCEL_AMD_SCAN_PRIO");  // ("normal": the scan stream in the background streams' class)
    static const char* scan_prio = scan_prio_env ? scan_prio_env : "low";
    if (!h->ev_fork) {
        HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < 4; i++) HIP_CHECK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    int lo = 0, hi = 0;
    if (scan_prio && !strcmp(scan_prio, "low") && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) {
        HIP_CHECK(hipStreamCreateWithPriority(&h->aux[3], hipStreamNonBlocking, lo));
    } else {
        HIP_CHECK(hipStreamCreateWithFlags(&h->aux[3], hipStreamNonBlocking));
    }
    static const char* bg_prio = getenv("AUNCEL_AMD_BG_PRIO");  // (experiment: "high" = the background stream at the main streams' priority)
    if (bg_prio && (!strcmp(bg_prio, "high") || !strcmp(bg_prio, "low")) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) {
        HIP_CHECK(hipStreamCreateWithPriority(&h->bg_stream, hipStreamNonBlocking, !strcmp(bg_prio, "high") ? hi : lo));
    } else {
        h->bg_stream = make_background_stream();
    }
    h->fix_stream = h->spec_stream = h->bg_stream;
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_sel, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIP_CHECK(hipEventCreateWithFlags(&h->ev_fix[i], hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_spec_go, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_spec_done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(h->ev_spec_done, h->bg_stream));  // (a search waits for the previous search's heap: nothing to wait for yet)
}

// The engine's streams are laid out for 8 hardware queues per priority class (ROCm's default is 4): with 4, the scan and background
// streams of four searches in flight share queues two by two, and a scan queued behind another context's 1.5 ms heap waits for it.
// Read by the HIP runtime when it starts: set here, when the library is loaded, unless the process has chosen a value itself.
// The library does NOT touch the environment (a constructor that called setenv changed the host application's and torch's runtime
// behind their backs, and was without effect anyway once HIP had started): the process that wants several searches in flight
// exports GPU_MAX_HW_QUEUES=8 before anything starts the HIP runtime (bench.py, tests/conftest.py).  What the engine
// does itself is not run more searches at a time than there are queues in a class (async_running_limit below).
static int hw_queues_per_class() {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e && *e ? atoi(e) : 4;  // (ROCm's default)
    return v > 0 ? v : 4;
}

// Side streams of a context, created one by one on first use.  The runtime attaches a new stream to the hardware queue with the
// fewest streams (GPU_MAX_HW_QUEUES of them): a context that creates its four side streams together puts stream i on queue i, so
// the stream the byte-code scans use -- aux[3] -- of EVERY context lands on the same hardware queue, where the scans of four
// searches in flight wait for each other (the dense launch's event span swung between 1.1 and 5 ms from run to run).  Created on
// demand, a byte-code context has one side stream, and four contexts spread over four queues.
void ensure_aux(amd_ivf* h, int lo, int hi) {
    ensure_context_streams(h);
    for (int i = lo; i <= hi; i++)
        if (!h->aux[i]) HIP_CHECK(hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking));
}

// ------------------------------------------------------------------------------------ lists
void upload_lists(amd_ivf* h) {
    h = ix(h);
    std::lock_guard<std::mutex> lock(h->upload_mu);
    if (!h->lists_dirty) return;
    use_device(h);
    h->h_list_off.assign(h->nlist + 1, 0);
    for (size_t l = 0; l < h->nlist; l++) h->h_list_off[l + 1] = h->h_list_off[l] + h->h_ids[l].size();
    size_t nt = h->h_list_off[h->nlist];
    // (a tile's first vector travels in the low SCAN_VB_BITS bits of ScanItem::vec_base: scan_vec_base)
    if ((uint64_t)nt >= (1ull << SCAN_VB_BITS)) throw EngineError("too many vectors for one index");
    h->d_codes.ensure(std::max<size_t>(nt, 1) * h->dpad * sizeof(float));
    h->d_ids.ensure(std::max<size_t>(nt, 1) * sizeof(int64_t));
    h->d_list_off.ensure((h->nlist + 1) * sizeof(uint64_t));
    for (size_t l = 0; l < h->nlist; l++) {
        size_t n = h->h_ids[l].size();
        if (!n) continue;
        HIP_CHECK(hipMemcpyAsync(h->d_codes.as<float>() + h->h_list_off[l] * h->dpad, h->h_codes[l].data(),
                                 n * h->dpad * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipMemcpyAsync(h->d_ids.as<int64_t>() + h->h_list_off[l], h->h_ids[l].data(), n * sizeof(int64_t),
                                 hipMemcpyHostToDevice, h->stream));
    }
    HIP_CHECK(hipMemcpyAsync(h->d_list_off.p, h->h_list_off.data(), (h->nlist + 1) * sizeof(uint64_t),
                             hipMemcpyHostToDevice, h->stream));
    h->have_codes8 = h->allow_bytes && nt > 0 && h->db_range.bytes() && (double)h->d * 255.0 * 255.0 < 2147483648.0;
    // (byte-valued lists get their fp32 fragment copy only when a search asks for the fp32 path: ensure_frag32)
    // (the filter's copy of the lists -- fp16 or fp32 fragment order, by the "filter" option -- is built by the first fp32 search
    // that wants it: ensure_frag16 / ensure_frag32)
    h->have_frag32 = false;
    h->frag16_state = 0;
    h->lanes_state = 0;
    h->frag32_possible = h->allow_filter && nt > 0 && nt < 0xffffffffull;
    {
        double s1 = 0, s2 = 0;
        for (size_t l = 0; l < h->nlist; l++) {
            const double len = (double)h->h_ids[l].size();
            s1 += len;
            s2 += len * len;
        }
        h->probed_len = s1 > 0 ? s2 / s1 : 0.0;
    }
    h->h_block_off.clear();
    h->d_lanes.release();
    if (nt > 0) {  // (the block table of every padded copy: byte codes, the filter's fragments, the dense rounds' lane order)
        h->h_block_off.assign(h->nlist + 1, 0);
        for (size_t l = 0; l < h->nlist; l++) h->h_block_off[l + 1] = h->h_block_off[l] + mfma_list_blocks(h->h_ids[l].size());
        h->d_block_off.ensure((h->nlist + 1) * sizeof(uint64_t));
        HIP_CHECK(hipMemcpyAsync(h->d_block_off.p, h->h_block_off.data(), (h->nlist + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    }
    h->d_frag32.release();
    h->d_frag16.release();
    h->d_yn.release();
    if (h->have_codes8) {
        const uint64_t nblk = h->h_block_off[h->nlist];
        h->d_frag.ensure(nblk * mfma_ksteps(h->d) * 1024);
        h->d_cy.ensure(nblk * 32 * sizeof(int32_t));
        launch_frag_from_f32(h->d_codes.as<float>(), h->d_list_off.as<uint64_t>(), h->d_block_off.as<uint64_t>(), (uint32_t)h->nlist, nblk, h->d,
                             h->dpad, h->metric, h->d_frag.as<uint8_t>(), h->d_cy.as<int32_t>(), h->stream);
    }
    HIP_CHECK(stream_sync(h->stream));
    h->lists_dirty = false;
}

// Byte view of n query rows (row stride dpad floats) when the lists have one and the queries qualify; fills
// ws->w_x8 / w_xnorm8 with the same row numbering.  Returns false when the search has to run in fp32.
bool byte_queries(amd_ivf* ws, const amd_ivf* index, const float* d_x, size_t n, const IntRange& qr) {
    if (!index->have_codes8 || !ws->allow_bytes || n == 0) return false;
    if (!index->db_range.bytes_with(qr, (size_t)index->d)) return false;
    // a slice of the resident query matrix that was converted by the previous call is still there
    const float* r0 = ws->d_resident.as<float>();
    const bool resident = ws->n_resident && d_x >= r0 && d_x + n * (size_t)ws->dpad <= r0 + ws->n_resident * (size_t)ws->dpad;
    if (resident && ws->x8_src == d_x && ws->x8_n == n && ws->x8_gen == ws->resident_gen) return true;
    ws->w_x8.ensure(n * (size_t)mfma_ksteps(index->d) * 32);
    ws->w_xnorm8.ensure(n * sizeof(int32_t));
    if (ws->fuse.active) {
        ws->fuse.bx = d_x;
        ws->fuse.bout = ws->w_x8.as<int8_t>();
        ws->fuse.bcx = ws->w_xnorm8.as<int32_t>();
        ws->fuse.have_bytes = true;
    } else {
        launch_sbytes_from_f32(d_x, n, index->d, index->dpad, index->metric, ws->w_x8.as<int8_t>(), ws->w_xnorm8.as<int32_t>(), ws->stream);
    }
    ws->x8_src = resident ? d_x : nullptr;
    ws->x8_n = n;
    ws->x8_gen = ws->resident_gen;
    return true;
}

// fp32 searches: threshold rounds as matrix-core filter + exact rescoring (ivf_filter.hip) when the index keeps the
// fragment-ordered fp32 copy.  AUNCEL_AMD_FILTER=0 (read per search: the tests run both ways) keeps scan_tiles_kernel throughout.
bool filter_available(const amd_ivf* ws, const amd_ivf* index, bool bytes) {
    if (index->opt.get(OPT_FILTER, 2) == 0) return false;
    return !bytes && (index->have_frag32 || index->frag32_possible) && ws->allow_filter;
}
// ... in fp16 (option "filter" = 2, the default): half the bytes of a pass and fp16 matrix-core rates; false where the lists have
// no usable scale (a non-finite element, magnitudes outside 2^+-24, a row tiny beside the largest) or the fp32 form was asked for
bool ensure_frag16(amd_ivf* index) {
    if (index->opt.get(OPT_FILTER, 2) != 2) return false;
    if (index->frag16_state != 0) return index->frag16_state > 0;
    std::lock_guard<std::mutex> lock(index->upload_mu);
    if (index->frag16_state != 0) return index->frag16_state > 0;
    if (!index->frag32_possible) return false;
    use_device(index);
    const uint64_t nblk = index->h_block_off[index->nlist];
    const size_t nt = index->h_list_off[index->nlist];
    index->d_yinfo.ensure(16);
    HIP_CHECK(hipMemsetAsync(index->d_yinfo.p, 0, 16, index->stream));
    launch_amax(index->d_codes.as<float>(), nt, index->dpad, index->d_yinfo.as<uint32_t>(), index->stream);
    uint32_t info[4] = {0, 0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(info, index->d_yinfo.p, 16, hipMemcpyDeviceToHost, index->stream));
    HIP_CHECK(stream_sync(index->stream));
    if (filter_half_scale(info, index->d) == 0.f) {
        index->frag16_state = -1;
        return false;
    }
    index->d_frag16.ensure(nblk * filter_steps16(index->d) * 1024);
    index->d_yn.ensure(nblk * 32 * sizeof(float));
    launch_frag16_from_f32(index->d_codes.as<float>(), index->d_list_off.as<uint64_t>(), index->d_block_off.as<uint64_t>(), (uint32_t)index->nlist,
                           nblk, index->d, index->dpad, index->metric, index->d_yinfo.as<uint32_t>(), index->d_frag16.as<float>(),
                           index->d_yn.as<float>(), index->stream);
    HIP_CHECK(stream_sync(index->stream));
    index->frag16_state = 1;
    return true;
}
// the fragment-ordered fp32 copy of lists that also have byte codes is built the first time an fp32 search runs over them
void ensure_frag32(amd_ivf* index) {
    if (index->have_frag32) return;
    std::lock_guard<std::mutex> lock(index->upload_mu);
    if (index->have_frag32 || !index->frag32_possible) return;
    use_device(index);
    const uint64_t nblk = index->h_block_off[index->nlist];
    index->d_frag32.ensure(nblk * filter_steps(index->d) * 1024);
    index->d_yn.ensure(nblk * 32 * sizeof(float));
    launch_frag32_from_f32(index->d_codes.as<float>(), index->d_list_off.as<uint64_t>(), index->d_block_off.as<uint64_t>(), (uint32_t)index->nlist,
                           nblk, index->d, index->dpad, index->metric, index->d_frag32.as<float>(), index->d_yn.as<float>(), index->stream);
    HIP_CHECK(stream_sync(index->stream));
    index->have_frag32 = true;
}
// the lane-ordered copy for the dense rounds of an fp32 search (option "lanes"); null where the block table is missing or the block
// numbers do not fit an item's upper bits (scan_vec_base)
const float* ensure_lanes(amd_ivf* index) {
    if (index->opt.get(OPT_LANES, 1) == 0) return nullptr;
    int st = index->lanes_state.load(std::memory_order_acquire);
    if (st == 0) {
        std::lock_guard<std::mutex> lock(index->upload_mu);
        st = index->lanes_state.load(std::memory_order_acquire);
        if (st == 0) {
            const uint64_t nt = index->h_list_off[index->nlist];
            const uint64_t nblk64 = index->h_block_off.empty() ? 0 : index->h_block_off[index->nlist] / 2;
            if (nblk64 == 0 || nblk64 >= (1ull << (64 - SCAN_VB_BITS)) || nt >= (1ull << SCAN_VB_BITS)) {
                st = -1;
            } else {
                // The copy is the fp32 lists once more.  An index that fits without it must keep searching (scan_tiles_kernel reads
                // the rows): the copy is made only where it leaves a quarter of itself + 1 GiB free for the searches' workspaces,
                // and a failed allocation means "no copy" for good (lanes_state -1), not a search that throws every time.
                use_device(index);
                const uint64_t bytes = nblk64 * (uint64_t)index->dpad * 64 * sizeof(float);
                const uint64_t want = bytes + bytes / 8 + 256;  // (DevBuf's slack)
                size_t free_b = 0, total_b = 0;
                const bool fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= want + want / 4 + (1ull << 30) &&
                                  !getenv("AUNCEL_AMD_LANES_NOFIT");  // (tests: the search of an index whose copy does not fit)
                st = -1;
                if (fits) {
                    try {
                        index->d_lanes.ensure(bytes);
                        launch_lanes_from_f32(index->d_codes.as<float>(), index->d_list_off.as<uint64_t>(), index->d_block_off.as<uint64_t>(),
                                              (uint32_t)index->nlist, nblk64, index->dpad, index->d_lanes.as<float>(), index->stream);
                        HIP_CHECK(stream_sync(index->stream));
                        st = 1;
                    } catch (const std::exception&) {
                        index->d_lanes.release();
                        (void)hipGetLastError();  // (the sticky error of the failed allocation)
                    }
                } else {
                    (void)hipGetLastError();
                }
                if (st < 0 && getenv("AUNCEL_AMD_VERBOSE"))
                    fprintf(stderr, "[auncel_amd] no lane-ordered copy of the lists (%.1f GiB wanted, %.1f GiB free): dense fp32 rounds read the rows\n",
                            want / 1073741824.0, free_b / 1073741824.0);
            }
            index->lanes_state.store(st, std::memory_order_release);
        }
    }
    return st > 0 ? index->d_lanes.as<float>() : nullptr;
}
// Dense probes ahead of the filter: the k-th best of the first f lists is the threshold everything else is filtered with, and
// about k / (f x length of a probed list) of the later candidates get under it; f is chosen to keep that near 3 % (the survivors are
// rescored one lane each; measured at nprobe 32 -- cfg 5: f = 1 / 2 / 3 / 4 -> 2.10 / 2.03 / 1.90 / 1.65 M queries/s, cfg 3:
// 2.31 / 2.38 / 2.35-2.47 / 2.27 -- a dense probe costs exact arithmetic per distance, a filtered one its list's bytes)
size_t filter_first_probes(const amd_ivf* index, size_t k, size_t nprobe, double survivors = 0.03) {
    // (the length of a list a query lands in: queries fall like the data, so a list is probed in proportion to its length and the
    // expected length of a probed list is sum(len^2) / sum(len) -- the plain mean where lists are even, three times it on cfg 5's
    // blobs, whose dense round then computed three times the distances the rule meant it to)
    const double mean_len = std::max<double>(1.0, index->probed_len);
    if (const char* e = getenv("AUNCEL_AMD_FILTER_FIRST"))  // (experiments: the dense probes of an fp32 search, whatever the rule says)
        if (*e) return std::max<size_t>(1, std::min<size_t>((size_t)atoi(e), nprobe));
    const size_t f = (size_t)std::ceil((double)k / (survivors * mean_len));
    return std::max<size_t>(1, std::min<size_t>(f, std::max<size_t>(1, nprobe / 4)));
}

// copy n x d host rows into a device matrix with row stride dpad (zero padded)
void upload_rows(amd_ivf* h, float* dst, const float* src, size_t n) {
    if (n == 0) return;
    if (h->d == h->dpad) {
        HIP_CHECK(hipMemcpyAsync(dst, src, n * h->d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_CHECK(hipMemsetAsync(dst, 0, n * h->dpad * sizeof(float), h->stream));
        HIP_CHECK(hipMemcpy2DAsync(dst, h->dpad * sizeof(float), src, h->d * sizeof(float), h->d * sizeof(float), n,
                                   hipMemcpyHostToDevice, h->stream));
    }
}

// ------------------------------------------------------------------------------------ state
struct State {  // per query slot
    float* heap_val;
    int64_t* heap_ref;
    uint32_t* stage;
    unsigned long long* nscan;
    uint32_t* done;
};

void init_state(amd_ivf* h, size_t n, size_t k, bool tune_or_train) {
    h->w_heap_val.ensure(n * k * sizeof(float));
    h->w_heap_ref.ensure(n * k * sizeof(int64_t));
    h->w_stage.ensure(n * 4);
    h->w_nscan.ensure(n * 8);
    h->w_done.ensure(n * 4);
    h->w_pre_val.ensure(n * 4);
    h->w_stoped.ensure(n * 4);
    h->w_D.ensure(n * k * sizeof(float));
    h->w_I.ensure(n * k * sizeof(int64_t));
    h->w_stats.ensure(STATS_ROWS * 4 * 8);
    h->w_error.ensure(4);
    h->w_thr.ensure(n * sizeof(float));
    h->w_log_cnt.ensure(n * 4);
    h->w_amb.ensure(n * 4);
    h->w_tie_flag.ensure(n * 4);
    h->w_log_snap.ensure(2 * n * 4);
    h->w_fin_round.ensure(n * 4);
    h->w_fix_pos.ensure(n * 4);
    h->w_qstat.ensure(n * sizeof(uint2));
    const bool fix_state = k <= 128 && !h->force_heap_select;  // (the heaps tie_fix_kernel replays: sorted-array selection only)
    if (fix_state) {
        h->w_fix_val.ensure(n * k * sizeof(float));
        h->w_fix_ref.ensure(n * k * sizeof(int64_t));
    }
    h->last_state_n = n;
    InitStateArgs ia{};
    ia.n = n;
    ia.k = k;
    ia.neutral = h->metric == METRIC_L2 ? FLT_MAX : -FLT_MAX;
    ia.heap_val = h->w_heap_val.as<float>();
    ia.heap_ref = h->w_heap_ref.as<int64_t>();
    ia.thr = h->w_thr.as<float>();
    ia.stage = h->w_stage.as<uint32_t>();
    ia.nscan = h->w_nscan.as<unsigned long long>();
    ia.done = h->w_done.as<uint32_t>();
    ia.pre_val = h->w_pre_val.as<float>();
    ia.stoped = h->w_stoped.as<uint32_t>();
    ia.stats = h->w_stats.as<unsigned long long>();
    ia.error = h->w_error.as<uint32_t>();
    ia.log_cnt = h->w_log_cnt.as<uint32_t>();
    ia.amb = h->w_amb.as<uint32_t>();
    ia.tie_flag = h->w_tie_flag.as<uint32_t>();
    ia.log_snap = h->w_log_snap.as<uint32_t>();
    ia.fin_round = h->w_fin_round.as<uint32_t>();
    ia.fix_pos = h->w_fix_pos.as<uint32_t>();
    ia.qstat = h->w_qstat.as<uint2>();
    if (fix_state) {
        ia.fix_val = h->w_fix_val.as<float>();
        ia.fix_ref = h->w_fix_ref.as<int64_t>();
    }
    if (h->fuse.active) {
        h->fuse.init = ia;
        h->fuse.have_init = true;
    } else {
        launch_init_state(ia, h->stream);
    }
    if (tune_or_train) h->w_dtb.ensure(n * (h->nlist / 8 + 20) * sizeof(float));
}

// ------------------------------------------------------------------------------------ query tiles
// Groups of 8 consecutive pairs (never crossing a list) get their query rows gathered and interleaved into
// the layout the scan kernel's scalar loads want.  `ranges`: (first pair, pair count) of every run of pairs
// that must start a new group (one per list); returns the group index of each run's first group.
std::vector<uint32_t> pack_query_tiles(amd_ivf* h, const float* d_queries, const std::vector<std::pair<uint32_t, uint32_t>>& ranges,
                                       int row_words = 0) {
    if (!row_words) row_words = h->dpad;  // 4-byte words per query row (byte rows: d / 4)
    std::vector<uint32_t> first(ranges.size());
    size_t ng = 0;
    for (size_t i = 0; i < ranges.size(); i++) {
        first[i] = (uint32_t)ng;
        ng += (ranges[i].second + SCAN_RQ - 1) / SCAN_RQ;
    }
    h->p_group_p0.ensure(std::max<size_t>(ng, 1) * 4);
    h->p_group_cnt.ensure(std::max<size_t>(ng, 1) * 4);
    uint32_t* gp = h->p_group_p0.as<uint32_t>();
    uint32_t* gc = h->p_group_cnt.as<uint32_t>();
    size_t g = 0;
    for (auto& r : ranges)
        for (uint32_t o = 0; o < r.second; o += SCAN_RQ) {
            gp[g] = r.first + o;
            gc[g] = std::min<uint32_t>(SCAN_RQ, r.second - o);
            g++;
        }
    h->w_group_p0.ensure(std::max<size_t>(ng, 1) * 4);
    h->w_group_cnt.ensure(std::max<size_t>(ng, 1) * 4);
    h->w_qtile.ensure(std::max<size_t>(ng, 1) * (size_t)row_words * SCAN_RQ * sizeof(float));
    if (ng) {
        HIP_CHECK(hipMemcpyAsync(h->w_group_p0.p, gp, ng * 4, hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipMemcpyAsync(h->w_group_cnt.p, gc, ng * 4, hipMemcpyHostToDevice, h->stream));
        launch_pack_queries(d_queries, h->w_pair_query.as<uint32_t>(), h->w_group_p0.as<uint32_t>(), h->w_group_cnt.as<uint32_t>(), ng,
                            row_words, h->w_qtile.as<float>(), h->stream);
    }
    return first;
}

// ------------------------------------------------------------------------------------ rounds
struct RoundSpec {
    // participating query slots and, for each, the probes [p0, p0+cnt) to run this round
    std::vector<uint32_t> slot;
    std::vector<uint32_t> p0, cnt;
    const int64_t* keys = nullptr;  // host, [.. x key_stride], row = slot
    size_t key_stride = 0;
    int k = 0;
    int store_pairs = 0;
    size_t max_codes = 0;
    int finalize_all = 0;
    uint32_t total_nprobe = 0;
    uint64_t id_offset = 0;
    const float* d_x = nullptr;  // device queries, row = slot
    bool bytes = false;          // scan the byte copies (ws->w_x8 / w_xnorm8 hold these queries)
    bool fixed_two = false;      // fixed nprobe: after round 0 one more round takes all remaining probes (threshold mode)
    bool range = false;          // range search: every round in threshold mode (thr = radius), entries collected instead of a replay
    TunerDev tuner{};
    TrainDev train{};
    const float* d_cdis = nullptr;     // device coarse arrays for set_online (row = slot)
    const int64_t* d_ckeys = nullptr;
    uint32_t coarse_stride = 0;
    int raw_heap_out = 0;
    int fused = 0;
    // scanner API over a part of a list (amd_ivf_scan_codes_at / _range; exec_round only, one query, one probe, key 0): the round sees
    // an index of ONE list whose vectors are [sub_base, sub_base + sub_n) of the packed matrix -- positions count from sub_base, as
    // the reference's scanner counts from the pointer it is handed (IndexIVFFlat.cpp:117-137)
    uint64_t sub_base = 0;
    size_t sub_n = (size_t)-1;
    long long pair_list = -1;  // store_pairs labels of a list part carry this list number
    // ... and, instead of the heap replay, the row's entries inside `collect_radius` in position order (scan_codes_range)
    bool collect = false;
    float collect_radius = 0.f;
    // time-bounded search: budgets in ms (device, by absolute id) and the host clock (us) the budgets count from
    const float* d_budget_ms = nullptr;
    double t_start_us = 0;
    bool caller_checks_error = false;  // the caller reads the error word back together with its results (finish_results)
    // a call of a few queries: queues the caller's own read-backs (results, statistics, error word) so that the look after the first
    // round can bring them along -- three calls in four end there, and then end with that one synchronisation
    std::function<void()> spec_finish;
    // AUNCEL_AMD_COARSE_TIES=redo with the heap's order arriving while the first round is scanned (adaptive_slice): enqueued between
    // the scan and the selection of round 0 -- waits for the heap, patches the ranking and the round's rows (launch_tie_patch),
    // then derives the stop rule's boundary distances from the patched ranking.  run_ties: rounds take whole runs of equal
    // coarse distances (PlanArgs::run_dis).
    std::function<void()> before_first_select;
    bool run_ties = false;
    // ... and, in the chained rounds, the first selection in two launches: the queries whose ranking does not wait for the heap are
    // selected while it still runs (split_prepare: the two query lists + the boundary distances of everybody), then the hook above
    // (split_between: wait, patch, the boundary distances of the patched rankings again), then the waiting ones.  The heap takes
    // 1.6 ms a row beside a dense round of 0.5: a lone batch stood idle for 1.07 ms of its 3.5 (profiles/r05_timeline_exact_ties.txt).
    std::function<void()> split_prepare, split_between;
    const uint32_t* split_free = nullptr;   // query lists and their counts (device)
    const uint32_t* split_wait = nullptr;
    const uint32_t* split_counts = nullptr; // [0] free, [1] waiting
    uint32_t split_wait_cap = 0;
};

static bool dbg_timing() {
    static const bool on = getenv("AUNCEL_AMD_DEBUG_TIMING") != nullptr;
    return on;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// AUNCEL_AMD_DEBUG_TIMING: host-side time stamps of a search (what the calling thread spends between its launches)
static thread_local std::vector<std::pair<const char*, double>> g_stamps;
static inline void host_stamp(const char* what) {
    if (dbg_timing()) g_stamps.emplace_back(what, now_us());
}
static void print_stamps() {
    if (!dbg_timing() || g_stamps.empty()) return;
    fprintf(stderr, "[host]");
    for (size_t i = 1; i < g_stamps.size(); i++) fprintf(stderr, " %s +%.0f", g_stamps[i].first, g_stamps[i].second - g_stamps[i - 1].second);
    fprintf(stderr, " | total %.0f us\n", g_stamps.back().second - g_stamps.front().second);
    g_stamps.clear();
}

// queries per full block of a list (ivf_kernels.h): 64 for the byte-code scan, 32 for the fp32 scans
static uint32_t scan_qblock(bool bytes) {
    static const int env = getenv("AUNCEL_AMD_QBLOCK") ? atoi(getenv("AUNCEL_AMD_QBLOCK")) : 0;
    if (env == 32 || env == 64) return (uint32_t)env;
    return bytes ? 64u : 32u;
}

static void print_replay_dbg(amd_ivf* h, size_t mb, hipStream_t s) {
    std::vector<unsigned long long> dbg(mb * 8);
    HIP_CHECK(hipMemcpyAsync(dbg.data(), h->w_misc.p, mb * 64, hipMemcpyDeviceToHost, s));
    HIP_CHECK(stream_sync(s));
    static const char* nm_replay[8] = {"wave cycles", "heap updates", "candidates", "rule evaluations", "stream cycles | rows walked", "rule cycles", "masked chunks", "probes"};
    const char* const* nm = nm_replay;
    for (int c = 0; c < 8; c++) {
        std::vector<unsigned long long> v(mb);
        for (size_t i = 0; i < mb; i++) v[i] = dbg[i * 8 + c];
        std::sort(v.begin(), v.end());
        double sum = 0;
        for (auto x : v) sum += x;
        fprintf(stderr, "[replay] %-16s mean %.0f p50 %llu p90 %llu p99 %llu max %llu\n", nm[c], sum / mb, v[mb / 2], v[mb * 9 / 10], v[mb * 99 / 100], v[mb - 1]);
    }
}

void exec_round(amd_ivf* h, const RoundSpec& r) {
    const size_t m = r.slot.size();
    if (m == 0) return;
    const uint32_t qblock = scan_qblock(r.bytes);
    const double t_enter = now_us();
    const bool sub = r.sub_n != (size_t)-1;
    const std::vector<uint64_t> sub_off{r.sub_base, r.sub_base + (sub ? r.sub_n : 0)};
    const size_t nlist = sub ? 1 : h->nlist;
    const std::vector<uint64_t>& off = sub ? sub_off : ix(h)->h_list_off;
    if (sub) {
        if (m != 1 || r.cnt[0] != 1 || r.bytes) throw EngineError("a list part is scanned for one query, in fp32");
        h->w_sub_off.ensure(32);  // (two offsets; the third word: scan_codes_range's count)
        HIP_CHECK(hipMemcpyAsync(h->w_sub_off.p, sub_off.data(), 16, hipMemcpyHostToDevice, h->stream));
    }
    size_t q0 = 0;
    std::vector<uint32_t> lcount(nlist + 1);
    while (q0 < m) {
        // ---- sub-batch [q0, q1) bounded by the distance-buffer budget
        size_t q1 = q0, total = 0;
        uint32_t rp = 0;
        while (q1 < m) {
            size_t need = 0;
            for (uint32_t p = 0; p < r.cnt[q1]; p++) {
                int64_t key = r.keys[(size_t)r.slot[q1] * r.key_stride + r.p0[q1] + p];
                if (key < 0) continue;
                if ((size_t)key >= nlist) throw EngineError("Invalid key=" + std::to_string(key) + " nlist=" + std::to_string(nlist));
                need += off[key + 1] - off[key];
            }
            if (q1 > q0 && total + need > h->dist_budget_floats) break;
            total += need;
            rp = std::max(rp, r.cnt[q1]);
            q1++;
        }
        const size_t mb = q1 - q0;
        (void)rp;
        // ---- segments (query-major distance rows, CSR over the queries) and pairs grouped by list
        size_t nseg = 0;
        for (size_t i = 0; i < mb; i++) nseg += r.cnt[q0 + i];
        h->p_seg_off.ensure(std::max<size_t>(nseg, 1) * 8);
        h->p_seg_list.ensure(std::max<size_t>(nseg, 1) * 4);
        h->p_seg_count.ensure(mb * 4);
        h->p_seg_begin.ensure(mb * 4);
        h->p_qsel.ensure(mb * 4);
        uint64_t* seg_off = h->p_seg_off.as<uint64_t>();
        int32_t* seg_list = h->p_seg_list.as<int32_t>();
        uint32_t* seg_count = h->p_seg_count.as<uint32_t>();
        uint32_t* seg_begin = h->p_seg_begin.as<uint32_t>();
        uint32_t* qsel = h->p_qsel.as<uint32_t>();
        std::fill(lcount.begin(), lcount.end(), 0u);
        size_t npairs = 0;
        uint64_t cursor = 0;
        double bytes = 0;
        {
            uint32_t sp = 0;
            for (size_t i = 0; i < mb; i++) {
                const size_t qi = q0 + i;
                seg_count[i] = r.cnt[qi];
                seg_begin[i] = sp;
                qsel[i] = r.slot[qi];
                const int64_t* kq = r.keys + (size_t)r.slot[qi] * r.key_stride + r.p0[qi];
                for (uint32_t p = 0; p < r.cnt[qi]; p++, sp++) {
                    const int64_t key = kq[p];
                    seg_list[sp] = (int32_t)key;
                    seg_off[sp] = cursor;
                    if (key >= 0) {
                        const size_t sz = off[key + 1] - off[key];
                        if (sz) {
                            lcount[key + 1]++;
                            npairs++;
                            cursor += sz;
                            bytes += (double)sz * h->d * 4.0;
                        }
                    }
                }
            }
        }
        h->scan_bytes += bytes;
        for (size_t l = 0; l < nlist; l++) lcount[l + 1] += lcount[l];
        h->p_pair_query.ensure(std::max<size_t>(npairs, 1) * 4);
        h->p_pair_out.ensure(std::max<size_t>(npairs, 1) * 8);
        uint32_t* pair_query = h->p_pair_query.as<uint32_t>();
        uint64_t* pair_out = h->p_pair_out.as<uint64_t>();
        {
            std::vector<uint32_t> fill(lcount.begin(), lcount.end() - 1);
            for (size_t i = 0; i < mb; i++) {
                const size_t qi = q0 + i;
                const uint32_t sb = seg_begin[i];
                for (uint32_t p = 0; p < r.cnt[qi]; p++) {
                    int32_t key = seg_list[sb + p];
                    if (key < 0 || off[key + 1] == off[key]) continue;
                    uint32_t pos = fill[key]++;
                    pair_query[pos] = r.slot[qi];
                    pair_out[pos] = seg_off[sb + p];
                }
            }
        }
        // ---- tiles, grouped by workgroup shape (qg 1 | 2 | 4 | 8) so that each shape gets its own launch.  The c queries
        // of a list go into blocks of 64 (qg 8: 8 waves x 8 queries over one 128-vector tile); the remainder block takes
        // the narrowest shape that holds it, so that no wave runs without queries.
        size_t n_qg[4] = {0, 0, 0, 0};
        std::vector<uint32_t> gbase(nlist, 0);
        std::vector<std::pair<uint32_t, uint32_t>> qranges;
        {
            uint32_t g = 0;
            for (size_t l = 0; l < nlist; l++) {
                uint32_t c = lcount[l + 1] - lcount[l];
                if (!c) continue;
                gbase[l] = g;
                g += (c + SCAN_RQ - 1) / SCAN_RQ;
                qranges.emplace_back(lcount[l], c);
            }
        }
        for (size_t l = 0; l < nlist; l++) {
            uint32_t c = lcount[l + 1] - lcount[l];
            if (!c) continue;
            const size_t sz = off[l + 1] - off[l];
            if (r.bytes) {  // scan_mfma_kernel items: (chunk of the list) x (block of 32 queries)
                n_qg[3] += (size_t)((c + MFMA_QBLOCK - 1) / MFMA_QBLOCK) * ((sz + mfma_chunk() - 1) / mfma_chunk());
                continue;
            }
            const uint32_t full = c / qblock, rem = c % qblock;
            if (full) {
                const uint32_t qg = scan_shape_of(qblock);
                const size_t tv = scan_tile_vecs(qg);
                n_qg[scan_qg_class(qg)] += (size_t)full * ((sz + tv - 1) / tv);
            }
            if (rem) {
                uint32_t qg = scan_shape_of(rem);
                size_t tv = scan_tile_vecs(qg);
                n_qg[scan_qg_class(qg)] += (sz + tv - 1) / tv;
            }
        }
        const size_t nitems = n_qg[0] + n_qg[1] + n_qg[2] + n_qg[3];
        h->p_items.ensure(std::max<size_t>(nitems, 1) * sizeof(ScanItem));
        ScanItem* items = h->p_items.as<ScanItem>();
        size_t cur[4] = {0, n_qg[0], n_qg[0] + n_qg[1], n_qg[0] + n_qg[1] + n_qg[2]};
        for (size_t l = 0; l < nlist; l++) {
            uint32_t c = lcount[l + 1] - lcount[l];
            if (!c) continue;
            const uint32_t sz = (uint32_t)(off[l + 1] - off[l]);
            if (r.bytes) {
                size_t& ni = cur[3];
                for (uint32_t vb = 0; vb < sz; vb += mfma_chunk())
                    for (uint32_t qb = 0; qb < c; qb += MFMA_QBLOCK) {
                        ScanItem& it = items[ni++];
                        it.vec_base = ix(h)->h_block_off[l] + vb / MFMA_BLOCK;
                        it.nvec = std::min<uint32_t>(mfma_chunk(), sz - vb);
                        it.vec_off = vb;
                        it.pair_begin = lcount[l] + qb;
                        it.npair = std::min<uint32_t>(MFMA_QBLOCK, c - qb);
                        it.qg = 0;
                        it.qgroup = 0;
                        h->scan_slots += (double)MFMA_QBLOCK * (((it.nvec + 63) / 64) * 64);
                        h->scan_useful += (double)it.npair * it.nvec;
                    }
                continue;
            }
            for (uint32_t qb = 0; qb < c; qb += qblock) {
                const uint32_t nq_blk = std::min<uint32_t>(qblock, c - qb);
                const uint32_t qg = scan_shape_of(nq_blk);
                const uint32_t tv = scan_tile_vecs(qg);
                size_t& ni = cur[scan_qg_class(qg)];
                for (uint32_t vb = 0; vb < sz; vb += tv) {
                    ScanItem& it = items[ni++];
                    // (a list part starts anywhere: no block of the lane-ordered copy belongs to it, the tiles read the rows)
                    it.vec_base = sub ? off[l] + vb : scan_vec_base(off[l] + vb, ix(h)->h_block_off.empty() ? 0ull : ix(h)->h_block_off[l], vb);
                    it.nvec = std::min(tv, sz - vb);
                    it.vec_off = vb;
                    it.pair_begin = lcount[l] + qb;
                    it.npair = nq_blk;
                    it.qg = qg;
                    it.qgroup = gbase[l] + qb / SCAN_RQ;
                    // bookkeeping: (query, vector) slots the waves that run will compute vs pairs wanted
                    h->scan_slots += (double)((nq_blk + SCAN_RQ - 1) / SCAN_RQ) * SCAN_RQ * tv;
                    h->scan_useful += (double)nq_blk * it.nvec;
                }
            }
        }
        // ---- upload + launch
        h->w_dist.ensure(std::max<uint64_t>(cursor, 1) * sizeof(float));
        h->w_seg_off.ensure(std::max<size_t>(nseg, 1) * 8);
        h->w_seg_list.ensure(std::max<size_t>(nseg, 1) * 4);
        h->w_seg_count.ensure(mb * 4);
        h->w_seg_begin.ensure(mb * 4);
        h->w_qsel.ensure(mb * 4);
        h->w_pair_query.ensure(std::max<size_t>(npairs, 1) * 4);
        h->w_pair_out.ensure(std::max<size_t>(npairs, 1) * 8);
        h->w_items.ensure(std::max<size_t>(nitems, 1) * sizeof(ScanItem));
        hipStream_t s = h->stream;
        if (nseg) {
            HIP_CHECK(hipMemcpyAsync(h->w_seg_off.p, seg_off, nseg * 8, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->w_seg_list.p, seg_list, nseg * 4, hipMemcpyHostToDevice, s));
        }
        HIP_CHECK(hipMemcpyAsync(h->w_seg_count.p, seg_count, mb * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(h->w_seg_begin.p, seg_begin, mb * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(h->w_qsel.p, qsel, mb * 4, hipMemcpyHostToDevice, s));
        if (npairs) {
            HIP_CHECK(hipMemcpyAsync(h->w_pair_query.p, pair_query, npairs * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->w_pair_out.p, pair_out, npairs * 8, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->w_items.p, items, nitems * sizeof(ScanItem), hipMemcpyHostToDevice, s));
        }
        const double t_prep = now_us();
        if (npairs && !r.bytes) pack_query_tiles(h, r.d_x, qranges);
        ScanArgs sa{};
        sa.qtile = h->w_qtile.as<float>();
        sa.codes = ix(h)->d_codes.as<float>();
        sa.lanes = r.bytes || sub ? nullptr : ensure_lanes(ix(h));
        sa.queries = r.d_x;
        sa.items = h->w_items.as<ScanItem>();
        sa.pair_query = h->w_pair_query.as<uint32_t>();
        sa.pair_out = h->w_pair_out.as<uint64_t>();
        sa.dist = h->w_dist.as<float>();
        sa.d = h->dpad;
        sa.metric = h->metric;
        sa.fused = r.fused;
        MfmaScanArgs ma{};
        ma.codes_frag = ix(h)->d_frag.as<uint8_t>();
        ma.code_cy = ix(h)->d_cy.as<int32_t>();
        ma.queries8 = h->w_x8.as<int8_t>();
        ma.query_cx = h->w_xnorm8.as<int32_t>();
        ma.items = h->w_items.as<ScanItem>();
        ma.pair_query = h->w_pair_query.as<uint32_t>();
        ma.pair_out = h->w_pair_out.as<uint64_t>();
        ma.dist = h->w_dist.as<float>();
        ma.d = h->d;
        ma.metric = h->metric;
        ma.nitems = (uint32_t)nitems;
        if (nitems && r.bytes) {
            size_t t = h->timer.begin(CAT_SCAN, s);
            launch_scan_mfma(ma, s);
            h->timer.end(t, s);
        } else if (nitems) {
            ensure_aux(h, 3, 3);
            size_t t = h->timer.begin(CAT_SCAN, s);
            const bool fork = true;  // the shapes on the side stream (see make_main_stream; one stream: run_rounds_device's note)
            if (fork) {
                HIP_CHECK(hipEventRecord(h->ev_fork, s));
                HIP_CHECK(hipStreamWaitEvent(h->aux[3], h->ev_fork, 0));
                launch_scan(sa, n_qg, h->aux[3], h->aux[3], h->aux[3], h->aux[3]);
                HIP_CHECK(hipEventRecord(h->ev_join[3], h->aux[3]));
                HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[3], 0));
            } else {
                launch_scan(sa, n_qg, s);
            }
            h->timer.end(t, s);
        }
        if (r.collect) {
            // scan_codes_range: the row's entries inside the radius, in position order (count, positions, distances: w_D / w_I hold them)
            h->w_D.ensure(std::max<uint64_t>(cursor, 1) * sizeof(float));
            h->w_I.ensure(std::max<uint64_t>(cursor, 1) * sizeof(uint32_t));
            if (!sub) throw EngineError("collect is the scanner API's (a list part)");
            launch_range_collect(h->w_dist.as<float>(), (uint32_t)cursor, r.collect_radius, h->metric, h->w_sub_off.as<uint32_t>() + 4,
                                 h->w_I.as<uint32_t>(), h->w_D.as<float>(), s);
            HIP_CHECK(stream_sync(s));
            q0 = q1;
            continue;
        }
        ReplayArgs ra{};
        ra.metric = h->metric;
        ra.k = r.k;
        ra.nlist = (uint32_t)nlist;
        ra.nq = (uint32_t)mb;
        ra.qsel = h->w_qsel.as<uint32_t>();
        ra.total_nprobe = r.total_nprobe;
        ra.round_probes = 0;
        ra.seg_begin = h->w_seg_begin.as<uint32_t>();
        ra.id_offset = r.id_offset;
        ra.dist = h->w_dist.as<float>();
        ra.seg_off = h->w_seg_off.as<uint64_t>();
        ra.seg_list = h->w_seg_list.as<int32_t>();
        ra.seg_count = h->w_seg_count.as<uint32_t>();
        ra.list_off = sub ? h->w_sub_off.as<uint64_t>() : ix(h)->d_list_off.as<uint64_t>();
        ra.ids = ix(h)->d_ids.as<int64_t>();
        ra.store_pairs = r.store_pairs;
        ra.identity_ids = 0;
        ra.max_codes = r.max_codes;
        ra.finalize_all = r.finalize_all;
        ra.heap_val = h->w_heap_val.as<float>();
        ra.heap_ref = h->w_heap_ref.as<int64_t>();
        ra.stage = h->w_stage.as<uint32_t>();
        ra.nscan = h->w_nscan.as<unsigned long long>();
        ra.done = h->w_done.as<uint32_t>();
        ra.pre_val = h->w_pre_val.as<float>();
        ra.stoped = h->w_stoped.as<uint32_t>();
        ra.dtb = h->w_dtb.as<float>();
        ra.coarse_dis = r.d_cdis;
        ra.coarse_keys = r.d_ckeys;
        ra.coarse_stride = r.coarse_stride;
        ra.trace_cap = (uint32_t)ix(h)->tuner_trace_cap;
        ra.D = h->w_D.as<float>();
        ra.I = h->w_I.as<int64_t>();
        ra.stats = h->w_stats.as<unsigned long long>();
        ra.error = h->w_error.as<uint32_t>();
        ra.raw_heap_out = r.raw_heap_out;
        ra.pair_list = r.pair_list;
        ra.tuner = r.tuner;
        ra.train = r.train;
        static const bool dbg_replay = getenv("AUNCEL_AMD_DEBUG_REPLAY") != nullptr;
        if (dbg_replay) {
            h->w_misc.ensure(mb * 64);
            HIP_CHECK(hipMemsetAsync(h->w_misc.p, 0, mb * 64, s));
            ra.dbg = h->w_misc.as<unsigned long long>();
        }
        {
            size_t t = h->timer.begin(CAT_SELECT, s);
            launch_replay(ra, s);
            h->timer.end(t, s);
        }
        if (dbg_replay) print_replay_dbg(h, mb, s);
        // the pinned staging buffers are reused by the next sub-batch
        const double t_launched = now_us();
        HIP_CHECK(stream_sync(s));
        if (dbg_timing())
            fprintf(stderr, "[round] queries %zu pairs %zu items %zu: host prep %.0f us, launch %.0f us, gpu wait %.0f us\n", mb, npairs, nitems,
                    t_prep - t_enter, t_launched - t_prep, now_us() - t_launched);
        q0 = q1;
    }
}

static bool pinned_io(const amd_ivf* h) { return opt(h, OPT_PINNED_IO, 1) != 0; }

static void d2h_small(amd_ivf* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    constexpr size_t LIMIT = (size_t)64 << 10, CAP = (size_t)1 << 20;
    const size_t off = (h->small_used + 63) & ~(size_t)63;
    if (bytes == 0) return;
    if (bytes > LIMIT || off + bytes > CAP) {
        HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        return;
    }
    h->p_small.ensure(CAP);
    if (h->small.empty()) h->small_used = 0;
    // (page-locked staging either way; with pinned_io the gathers of a whole batch are ONE kernel, launched by sync_and_flush)
    const bool by_kernel = pinned_io(h) && bytes % 4 == 0;
    if (!by_kernel) HIP_CHECK(hipMemcpyAsync(h->p_small.as<unsigned char>() + off, src, bytes, hipMemcpyDeviceToHost, s));
    h->small.push_back({dst, off, bytes, by_kernel ? src : nullptr});
    h->small_used = off + bytes;
}
// the gathers recorded by d2h_small, on the stream their sources are produced on
static void launch_small_gathers(amd_ivf* h, hipStream_t s) {
    CopySegs c{};
    unsigned char* base = nullptr;
    for (auto& e : h->small) {
        if (!e.src) continue;
        if (!base) base = static_cast<unsigned char*>(h->p_small.dev());
        c.src[c.n] = e.src;
        c.dst[c.n] = base + e.off;
        c.words[c.n] = (uint32_t)(e.bytes / 4);
        e.src = nullptr;
        if (++c.n == CopySegs::MAX) {
            launch_copy_segs(c, s);
            c.n = 0;
        }
    }
    launch_copy_segs(c, s);
}
// after the stream the copies were queued on has been synchronised
static void flush_small(amd_ivf* h) {
    for (const auto& c : h->small) memcpy(c.dst, h->p_small.as<unsigned char>() + c.off, c.bytes);
    h->small.clear();
    h->small_used = 0;
    h->stage_used = 0;  // (the scatter kernels that read the host-to-device staging have run)
}

// a failed synchronisation must not leave copies pending: their destinations may be the caller's stack
static void sync_and_flush(amd_ivf* h, hipStream_t s) {
    hipError_t e = hipSuccess;
    try {
        launch_small_gathers(h, s);
    } catch (...) {
        h->small.clear();
        h->small_used = 0;
        h->after_flush.clear();
        throw;
    }
    e = stream_sync(s);
    if (e != hipSuccess) {
        h->small.clear();
        h->small_used = 0;
        h->after_flush.clear();
        HIP_CHECK(e);
    }
    flush_small(h);
    // epilogues that were waiting for this synchronisation (they may throw: the first one that does ends the call)
    std::vector<std::function<void()>> fs;
    fs.swap(h->after_flush);
    for (auto& f : fs) f();
}

// Host-to-device counterpart of d2h_small: the bytes are packed into page-locked staging now and scattered to their device
// destinations by one kernel (flush_h2d), instead of one blit per array.  The staging is reused once a synchronisation has
// shown the scatter to be done (flush_small).
static void flush_h2d(amd_ivf* h, hipStream_t s) {
    launch_copy_segs(h->h2d_pending, s);
    h->h2d_pending.n = 0;
}
static void h2d_small(amd_ivf* h, void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    constexpr size_t LIMIT = (size_t)256 << 10, CAP = (size_t)2 << 20;
    if (bytes == 0) return;
    const size_t off = (h->stage_used + 63) & ~(size_t)63;
    if (!pinned_io(h) || bytes % 4 != 0 || bytes > LIMIT || off + bytes > CAP) {
        HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    h->p_stage.ensure(CAP);
    memcpy(h->p_stage.as<unsigned char>() + off, src_host, bytes);
    CopySegs& c = h->h2d_pending;
    c.src[c.n] = static_cast<unsigned char*>(h->p_stage.dev()) + off;
    c.dst[c.n] = dst_dev;
    c.words[c.n] = (uint32_t)(bytes / 4);
    h->stage_used = off + bytes;
    if (++c.n == CopySegs::MAX) flush_h2d(h, s);
}

// a query admitted more candidates than its admission log holds: the search is repeated with the heap kernels
struct SelectLogOverflow : std::runtime_error {
    SelectLogOverflow() : std::runtime_error("selection log overflow") {}
};

static void throw_device_error(uint32_t err) {
    if (err == ERR_ARCOS_DOMAIN) throw EngineError("arcos's domain definition is [-1, 1]");
    if (err == ERR_COSINE_PRECOND) throw EngineError("cosine theorem's prerequisites");
    if (err == ERR_INVALID_KEY) throw EngineError("Invalid key");
    if (err == ERR_ITEM_OVERFLOW) throw std::runtime_error("tile list overflow");
    if (err == ERR_LOG_OVERFLOW) throw SelectLogOverflow();
    if (err) throw EngineError("device-side error " + std::to_string(err));
}

// Per-handle state of a call that must not outlive it when the call is left by an exception: the phase timers' switch, scatter
// entries still pending in the host-to-device staging (they name device buffers the next call may have reallocated), the staging
// cursor, epilogues queued behind a synchronisation that will not happen, and -- results going straight into the caller's
// page-locked buffers -- kernels that may still be writing there (the selection on the main stream, the tie replay on the
// background stream): the caller gets its buffers back with the error only once they are quiet.
struct CallScope {
    amd_ivf* h;
    int entered;
    explicit CallScope(amd_ivf* hh) : h(hh), entered(std::uncaught_exceptions()) {}
    ~CallScope() {
        h->timer.off = false;
        if (std::uncaught_exceptions() > entered) {
            h->h2d_pending.n = 0;
            h->stage_used = 0;
            h->small.clear();
            h->small_used = 0;
            h->after_flush.clear();
            if (h->last_direct_out) {
                (void)hipStreamSynchronize(h->stream);
                if (h->bg_stream) (void)hipStreamSynchronize(h->bg_stream);
            }
        }
    }
};

// copies queued with d2h_small name caller memory (often stack variables): if the sequence is left by an exception before
// its sync_and_flush, they must not stay pending for the next call's flush
struct SmallCopies {
    amd_ivf* h;
    explicit SmallCopies(amd_ivf* hh) : h(hh) {}
    ~SmallCopies() {
        h->small.clear();
        h->small_used = 0;
        h->after_flush.clear();
    }
};

void check_device_error(amd_ivf* h) {
    uint32_t err = 0;
    SmallCopies guard(h);
    d2h_small(h, &err, h->w_error.p, 4, h->stream);
    sync_and_flush(h, h->stream);
    throw_device_error(err);
}

// body(): a whole search on h (state initialised inside, outputs written only on success).  ERR_LOG_OVERFLOW from the
// sorted-array selection -> once more with the reference's heap as the selection.
template <class F> static void with_select_fallback(amd_ivf* h, F&& body) {
    try {
        body();
    } catch (const SelectLogOverflow&) {
        h->force_heap_select = true;
        try {
            body();
        } catch (...) {
            h->force_heap_select = false;
            throw;
        }
        h->force_heap_select = false;
    }
}

// the device statistics of a call: the sum of the per-XCD rows (STATS_ROWS)
static void sum_stat_rows(const unsigned long long* rows, unsigned long long out[4]) {
    for (int k = 0; k < 4; k++) out[k] = 0;
    for (uint32_t r = 0; r < STATS_ROWS; r++)
        for (int k = 0; k < 4; k++) out[k] += rows[4 * r + k];
}
void fold_stats(amd_ivf* h, size_t nq) {
    unsigned long long rows[4 * STATS_ROWS], st[4];
    SmallCopies guard(h);
    d2h_small(h, rows, h->w_stats.p, sizeof rows, h->stream);
    sync_and_flush(h, h->stream);
    sum_stat_rows(rows, st);
    if (nq) ix(h)->tie_rate.store((float)((double)st[3] / (double)nq));
    h->stats_host[0] += nq;
    h->stats_host[1] += st[0];
    h->stats_host[2] += st[1];
    h->stats_host[3] += st[2];
}

// device view of a page-locked host buffer (hipHostMalloc / hipHostRegister, e.g. torch's pin_memory), else null
static void* device_view(const void* host) {
    if (!host) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) {
        (void)hipGetLastError();  // (pageable memory: not an error of ours)
        return nullptr;
    }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}
struct DirectOut {
    amd_ivf* h;
    DirectOut(amd_ivf* h_, float* D, int64_t* I) : h(h_) {
        void* d = opt(h, OPT_DIRECT_OUT, 1) != 0 ? device_view(D) : nullptr;
        void* i = d ? device_view(I) : nullptr;
        h->out_D = i ? static_cast<float*>(d) : nullptr;
        h->out_I = i ? static_cast<int64_t*>(i) : nullptr;
        h->last_direct_out = h->out_D != nullptr;
    }
    // Leaving by an exception (a failed HIP call, ERR_LOG_OVERFLOW from the selection) while kernels that store into the caller's
    // buffers may still be in flight: wait for them, the caller is about to get its buffers back with an error.
    ~DirectOut() {
        if (h->out_D && std::uncaught_exceptions() > 0) {
            (void)hipStreamSynchronize(h->stream);
            if (h->fix_stream) (void)hipStreamSynchronize(h->fix_stream);
        }
        h->out_D = nullptr, h->out_I = nullptr;
    }
};

// Final read-back of a search: results, error word and counters behind one synchronisation.  An error is raised after the
// copies (the reference throws in the middle of its loop and leaves partial output behind as well).
// defer: the caller has more to read back and synchronises (sync_and_flush) itself; error check and statistics then run behind
// that synchronisation (amd_ivf::after_flush), and the caller guards the pending copies (SmallCopies).
void finish_results(amd_ivf* h, size_t n, size_t k, float* D, int64_t* I, uint32_t* stage_out = nullptr, bool defer = false) {
    struct Back {
        uint32_t err = 0;
        unsigned long long rows[4 * STATS_ROWS];
        unsigned long long st[4] = {0, 0, 0, 0};
    };
    auto back = std::make_shared<Back>();
    auto epilogue = [h, n, back]() {
        throw_device_error(back->err);
        sum_stat_rows(back->rows, back->st);
        if (n) ix(h)->tie_rate.store((float)((double)back->st[3] / (double)n));
        h->stats_host[0] += n;
        h->stats_host[1] += back->st[0];
        h->stats_host[2] += back->st[1];
        h->stats_host[3] += back->st[2];
    };
    auto queue = [&] {
        d2h_small(h, &back->err, h->w_error.p, 4, h->stream);
        d2h_small(h, back->rows, h->w_stats.p, sizeof back->rows, h->stream);
        if (stage_out) d2h_small(h, stage_out, h->w_stage.p, n * 4, h->stream);
        if (!h->out_D) {  // (else the kernels wrote the caller's buffers themselves)
            d2h_small(h, D, h->w_D.p, n * k * sizeof(float), h->stream);
            d2h_small(h, I, h->w_I.p, n * k * sizeof(int64_t), h->stream);
        }
    };
    if (defer) {
        queue();
        h->after_flush.push_back(epilogue);
        return;
    }
    SmallCopies guard(h);
    queue();
    sync_and_flush(h, h->stream);
    epilogue();
}

static void fill_timing(amd_ivf* h, const double* ms, const double* ln) {
    h->timing[0] = ms[CAT_COARSE];
    h->timing[1] = ms[CAT_SCAN] + ms[CAT_SCAN_THR];
    h->timing[2] = ms[CAT_SELECT] + ms[CAT_SELECT_THR] + ms[CAT_TIE_FIX];
    h->timing[4] = ln[CAT_SCAN] + ln[CAT_SCAN_THR];
    h->timing[7] = ln[CAT_SELECT] + ln[CAT_SELECT_THR] + ln[CAT_TIE_FIX];
    for (int c = 0; c < NCAT; c++) h->timing_detail[2 * c] = ms[c], h->timing_detail[2 * c + 1] = ln[c];
    h->timing_detail[2 * NCAT] = h->scan_min_bytes - h->scan_min_bytes_thr;
    h->timing_detail[2 * NCAT + 1] = h->scan_min_bytes_thr;
}
void finish_timing(amd_ivf* h, double wall_ms) {
    double ms[NCAT], ln[NCAT];
    h->timer.collect(ms, NCAT, ln);
    fill_timing(h, ms, ln);
    h->timing[3] = wall_ms;
    h->timing[5] = h->scan_bytes;
    h->last_min_bytes = h->scan_min_bytes;
    h->timing[6] = h->scan_slots > 0 ? h->scan_useful / h->scan_slots : 0;
}

struct WallClock {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    double t0 = 0;
    // host_only: the calling thread's clock (a search whose phase timers are off ends with a synchronisation anyway; an event pair
    // costs it two creations, two records and a second wait: 15 us of a 0.3 ms call)
    explicit WallClock(hipStream_t st, bool host_only = false) : s(st) {
        if (host_only) {
            t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
            return;
        }
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, s));
    }
    double stop() {
        if (!a) return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
        HIP_CHECK(hipEventRecord(b, s));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
    ~WallClock() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// ------------------------------------------------------------------------------------ coarse
// distances of n device queries (row stride dpad) to every centroid -> sorted top-nprobe on device
// prefix != 0: the caller reads only the first `prefix` entries of every ranking (launch_sort_rows)
void coarse_dev(amd_ivf* h, const float* d_x, size_t n, size_t nprobe, int mode, float* d_out_dis, int64_t* d_out_keys, int fused,
                size_t prefix = 0) {
    if (!ix(h)->have_centroids) throw EngineError("quantizer has no centroids");
    // mode 0: exact per-pair kernel; 1: GEMM formulation on the matrix cores; -1: the reference's own switch
    // (utils.cpp:624-655: exact for fewer than 20 queries when d % 4 == 0, BLAS otherwise)
    const bool gemm = mode == 1 || (mode < 0 && !(n < 20 && h->d % 4 == 0));
    const size_t nlist = h->nlist;
    hipStream_t s = h->stream;
    // Exact rankings of a large call that reads few entries of each (fixed nprobe): the matrix cores rank approximately, the
    // candidates that can be among the nprobe best are recomputed exactly (coarse_pick_kernel); queries in which exactly equal
    // distances meet -- where the reference's order is its heap's history -- come back flagged and take the path below.
    if (!gemm && !h->in_coarse_pick && opt(h, OPT_COARSE_PICK, 2) != 0 && n >= 256 && nprobe <= 128 && nlist <= 4096 && nlist >= 4 * nprobe + 64 &&
        prefix == 0 && h->ties_override < 0 && n * nlist <= h->dist_budget_floats) {
        size_t t = h->timer.begin(CAT_COARSE, s);
        h->w_xnorms.ensure(n * sizeof(float));
        h->w_dist.ensure(n * nlist * sizeof(float));
        static const bool dbg_pick = getenv("AUNCEL_AMD_DEBUG_PICK") != nullptr;  // why rankings are flagged (five counters behind the list)
        h->c_pick_flag.ensure((n + 1 + 8 + 8) * 4);  // (+ the padding of the flagged list, + the debugging counters)
        h->p_pick.ensure((n + 1 + 8) * 4);
        HIP_CHECK(hipMemsetAsync(h->c_pick_flag.p, 0, 4, s));
        if (dbg_pick) HIP_CHECK(hipMemsetAsync(h->c_pick_flag.as<uint32_t>() + n + 1 + 8, 0, 32, s));
        launch_row_norms(d_x, n, h->dpad, h->w_xnorms.as<float>(), s);
        const FilterParams* prm = nullptr;
        if (opt(h, OPT_COARSE_PICK, 2) >= 2 && ix(h)->d_cinfo.p) {
            // the approximate distances from fp16 operands (option coarse_pick = 2, the default): scales and error constant from the
            // two matrices' ranges, on the device; without a usable scale every query comes back flagged (the exact path below)
            h->w_qinfo.ensure(16);
            h->w_fparams.ensure(2 * sizeof(FilterParams));
            HIP_CHECK(hipMemsetAsync(h->w_qinfo.p, 0, 16, s));
            launch_amax(d_x, n, h->dpad, h->w_qinfo.as<uint32_t>(), s);
            FilterParams* p = h->w_fparams.as<FilterParams>() + 1;  // (slot 0: the list filter's, of the same search)
            launch_half_params(h->w_qinfo.as<uint32_t>(), ix(h)->d_cinfo.as<uint32_t>(), h->d, p, s);
            launch_coarse_gemm16(h->metric, d_x, ix(h)->d_centroids.as<float>(), h->w_xnorms.as<float>(), ix(h)->d_centroid_norms.as<float>(),
                                 (int)n, (int)nlist, h->dpad, h->w_dist.as<float>(), p, s);
            prm = p;
        } else {
            launch_coarse_gemm(h->metric, d_x, ix(h)->d_centroids.as<float>(), h->w_xnorms.as<float>(), ix(h)->d_centroid_norms.as<float>(),
                               (int)n, (int)nlist, h->dpad, h->w_dist.as<float>(), s);
        }
        launch_coarse_pick(h->metric, h->w_dist.as<float>(), d_x, ix(h)->d_centroids.as<float>(), h->w_xnorms.as<float>(), ix(h)->centroid_norm_max,
                           (uint32_t)n, (uint32_t)nlist, (uint32_t)nprobe, h->dpad, d_out_dis, d_out_keys, h->c_pick_flag.as<uint32_t>(),
                           h->c_pick_flag.as<uint32_t>() + 1, s, prm, dbg_pick ? h->c_pick_flag.as<uint32_t>() + n + 1 + 8 : nullptr);
        h->timer.end(t, s);
        // (the flagged queries: how many, which)
        CopySegs c{};
        c.src[0] = h->c_pick_flag.p;
        c.dst[0] = h->p_pick.dev();
        c.words[0] = (uint32_t)(n + 1);
        c.n = 1;
        launch_copy_segs(c, s);
        HIP_CHECK(stream_sync(s));
        const uint32_t m = h->p_pick.as<uint32_t>()[0];
        if (dbg_pick) {
            uint32_t why[5];
            HIP_CHECK(hipMemcpy(why, h->c_pick_flag.as<uint32_t>() + n + 1 + 8, sizeof(why), hipMemcpyDeviceToHost));
            fprintf(stderr, "[pick] %u of %zu rankings flagged: threshold/scale %u, too many candidates %u, too few %u, not finite %u, equal distances %u\n", m, n,
                    why[0], why[1], why[2], why[3], why[4]);
        }
        if (const uint32_t m_flagged = m) {
            // The flagged rankings go the exact way as a call of their own, whose work list is cached by its size (coarse_dev below):
            // a handful of them -- 3, 7, 5 from one batch to the next -- would upload a new list every time.  They are padded to a
            // multiple of eight (one query group: the same waves) with copies of the first, which are ranked and scattered twice.
            const uint32_t m = (m_flagged + 7u) & ~7u;
            if (m != m_flagged) {
                uint32_t* list = h->p_pick.as<uint32_t>() + 1;
                for (uint32_t i = m_flagged; i < m; i++) list[i] = list[0];
                HIP_CHECK(hipMemcpyAsync(h->c_pick_flag.as<uint32_t>() + 1 + m_flagged, list + m_flagged, (m - m_flagged) * 4, hipMemcpyHostToDevice, s));
            }
            h->c_pick_x.ensure((size_t)m * h->dpad * sizeof(float));
            h->c_pick_dis.ensure((size_t)m * nprobe * 4);
            h->c_pick_keys.ensure((size_t)m * nprobe * 8);
            launch_gather_rows(d_x, h->c_pick_flag.as<uint32_t>() + 1, m, (uint32_t)h->dpad, h->c_pick_x.as<float>(), s);
            struct Guard {
                bool& f;
                ~Guard() { f = false; }
            } guard{h->in_coarse_pick};
            h->in_coarse_pick = true;
            coarse_dev(h, h->c_pick_x.as<float>(), m, nprobe, mode, h->c_pick_dis.as<float>(), h->c_pick_keys.as<int64_t>(), fused, 0);
            launch_scatter_rows(h->c_pick_dis.p, h->c_pick_flag.as<uint32_t>() + 1, m, (uint32_t)nprobe, d_out_dis, s);
            launch_scatter_rows(h->c_pick_keys.p, h->c_pick_flag.as<uint32_t>() + 1, m, (uint32_t)(2 * nprobe), d_out_keys, s);
        }
        h->coarse_picked = n - m;
        return;
    }
    h->coarse_picked = 0;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, h->dist_budget_floats / std::max<size_t>(nlist, 1)));
    const bool use_heap = nprobe <= 128;
    // Longer rankings are sorted; between exactly equal distances the reference's order is an artefact of its heap's
    // history, reproduced by re-running that heap for the rows concerned (launch_heap_tie_order).  AUNCEL_AMD_COARSE_TIES:
    // "heap" always, "id" never (such runs stay in centroid-number order), default: calls of fewer than 20 queries, the
    // regime in which the reference ranks exact distances at all (utils.cpp:624-655; from 20 queries on it ranks sgemm
    // output, whose low bits -- and with them which distances coincide -- belong to the BLAS library).
    const int ties_opt = (int)opt(h, OPT_COARSE_TIES, -1);  // 0 centroid number, 1 heap, 2 redo (a matter of the adaptive search), -1 unset
    const bool heap_ties = !use_heap && (h->ties_override >= 0 ? h->ties_override == 1 : ties_opt == 1 || (ties_opt != 0 && n < 20));
    for (size_t c0 = 0; c0 < n; c0 += chunk) {
        const size_t m = std::min(chunk, n - c0);
        // pairs: every query of the chunk against the single "list" = centroid table
        const uint32_t qg = scan_shape_of((uint32_t)std::min<size_t>(m, scan_qblock(false)));
        const uint32_t tv = scan_tile_vecs(qg);
        const size_t nitems = ((m + qg * SCAN_RQ - 1) / (qg * SCAN_RQ)) * ((nlist + tv - 1) / tv);
        // The work list of a chunk (every query of the chunk against the one "list" of centroids) depends on (c0, m, nlist)
        // only: it lives in buffers of its own (the rounds' planning kernels write the w_* ones) and is uploaded when that
        // signature changes -- a caller that searches batch after batch of one size pays for it once (it was five blits and,
        // for the staging buffers' sake, a host synchronisation per call).
        const uint64_t csig = ((uint64_t)c0 << 40) ^ ((uint64_t)m << 20) ^ (uint64_t)nlist ^ ((uint64_t)qg << 60) ^ (use_heap ? 1ull << 59 : 0ull);
        const size_t ng = (m + SCAN_RQ - 1) / SCAN_RQ;
        const bool cached = h->coarse_sig_valid && h->coarse_sig == csig;
        if (!cached) {
            h->p_pair_query.ensure(m * 4);
            h->p_pair_out.ensure(m * 8);
            h->p_items.ensure(nitems * sizeof(ScanItem));
            h->p_group_p0.ensure(ng * 4);
            h->p_group_cnt.ensure(ng * 4);
            uint32_t* pq = h->p_pair_query.as<uint32_t>();
            uint64_t* po = h->p_pair_out.as<uint64_t>();
            ScanItem* items = h->p_items.as<ScanItem>();
            for (size_t i = 0; i < m; i++) {
                pq[i] = (uint32_t)(c0 + i);
                po[i] = (uint64_t)i * nlist;
            }
            size_t ni = 0;
            for (uint32_t vb = 0; vb < nlist; vb += tv)
                for (uint32_t qb = 0; qb < m; qb += qg * SCAN_RQ) {
                    ScanItem& it = items[ni++];
                    it.vec_base = vb;
                    it.nvec = std::min<uint32_t>(tv, (uint32_t)nlist - vb);
                    it.vec_off = vb;
                    it.pair_begin = qb;
                    it.npair = std::min<uint32_t>(qg * SCAN_RQ, (uint32_t)m - qb);
                    it.qg = qg;
                    it.qgroup = qb / SCAN_RQ;
                }
            for (size_t g = 0; g < ng; g++) {
                h->p_group_p0.as<uint32_t>()[g] = (uint32_t)(g * SCAN_RQ);
                h->p_group_cnt.as<uint32_t>()[g] = (uint32_t)std::min<size_t>(SCAN_RQ, m - g * SCAN_RQ);
            }
            h->c_pair_query.ensure(m * 4);
            h->c_pair_out.ensure(m * 8);
            h->c_items.ensure(nitems * sizeof(ScanItem));
            h->c_group_p0.ensure(ng * 4);
            h->c_group_cnt.ensure(ng * 4);
            HIP_CHECK(hipMemcpyAsync(h->c_pair_query.p, pq, m * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->c_pair_out.p, po, m * 8, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->c_items.p, items, nitems * sizeof(ScanItem), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->c_group_p0.p, h->p_group_p0.p, ng * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(h->c_group_cnt.p, h->p_group_cnt.p, ng * 4, hipMemcpyHostToDevice, s));
            h->coarse_sig_valid = false;  // (valid once the uploads are known to have left the staging buffers: below)
        }
        h->w_dist.ensure(m * nlist * sizeof(float));
        h->w_qtile.ensure(std::max<size_t>(ng, 1) * (size_t)h->dpad * SCAN_RQ * sizeof(float));
        launch_pack_queries(d_x, h->c_pair_query.as<uint32_t>(), h->c_group_p0.as<uint32_t>(), h->c_group_cnt.as<uint32_t>(), ng, h->dpad,
                            h->w_qtile.as<float>(), s);
        ScanArgs sa{};
        sa.qtile = h->w_qtile.as<float>();
        sa.codes = ix(h)->d_centroids.as<float>();
        sa.queries = d_x;
        sa.items = h->c_items.as<ScanItem>();
        sa.pair_query = h->c_pair_query.as<uint32_t>();
        sa.pair_out = h->c_pair_out.as<uint64_t>();
        sa.dist = h->w_dist.as<float>();
        sa.d = h->dpad;
        sa.metric = h->metric;
        sa.fused = fused;
        size_t t = h->timer.begin(CAT_COARSE, s);
        if (gemm) {
            h->w_xnorms.ensure(m * sizeof(float));
            launch_row_norms(d_x + c0 * h->dpad, m, h->dpad, h->w_xnorms.as<float>(), s);
            launch_coarse_gemm(h->metric, d_x + c0 * h->dpad, ix(h)->d_centroids.as<float>(), h->w_xnorms.as<float>(),
                               ix(h)->d_centroid_norms.as<float>(), (int)m, (int)nlist, h->dpad, h->w_dist.as<float>(), s);
        } else {
            size_t n_qg[4] = {0, 0, 0, 0};
            n_qg[scan_qg_class(qg)] = nitems;
            launch_scan(sa, n_qg, s);
        }
        if (use_heap) {
            // the reference's own selection: a heap of nprobe over centroids 0..nlist-1 (utils.cpp:454-490)
            const size_t k = nprobe;
            h->c_heap_val.ensure(m * k * 4);
            h->c_heap_ref.ensure(m * k * 8);
            h->c_stage.ensure(m * 4);
            h->c_nscan.ensure(m * 8);
            h->c_done.ensure(m * 4);
            h->c_seg_off.ensure(m * 8);
            h->c_seg_list.ensure(m * 4);
            h->c_seg_count.ensure(m * 4);
            // (throw-away statistics: one row per XCD as every selection launch adds to them, and the error word behind the rows)
            constexpr size_t MISC_BYTES = STATS_ROWS * 4 * 8 + 8;
            h->w_misc.ensure(MISC_BYTES);
            if (!cached) {  // (the heap form's one-row-a-query segment table is part of the cached work list)
                h->p_seg_off.ensure(m * 8);
                h->p_seg_list.ensure(m * 4);
                h->p_seg_count.ensure(m * 4);
                for (size_t i = 0; i < m; i++) {
                    h->p_seg_off.as<uint64_t>()[i] = (uint64_t)i * nlist;
                    h->p_seg_list.as<int32_t>()[i] = 0;
                    h->p_seg_count.as<uint32_t>()[i] = 1;
                }
                HIP_CHECK(hipMemcpyAsync(h->c_seg_off.p, h->p_seg_off.p, m * 8, hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync(h->c_seg_list.p, h->p_seg_list.p, m * 4, hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync(h->c_seg_count.p, h->p_seg_count.p, m * 4, hipMemcpyHostToDevice, s));
            }
            launch_fill_f32(h->c_heap_val.as<float>(), m * k, h->metric == METRIC_L2 ? FLT_MAX : -FLT_MAX, s);
            launch_fill_i64(h->c_heap_ref.as<int64_t>(), m * k, -1, s);
            HIP_CHECK(hipMemsetAsync(h->c_stage.p, 0, m * 4, s));
            HIP_CHECK(hipMemsetAsync(h->c_nscan.p, 0, m * 8, s));
            HIP_CHECK(hipMemsetAsync(h->c_done.p, 0, m * 4, s));
            HIP_CHECK(hipMemsetAsync(h->w_misc.p, 0, MISC_BYTES, s));
            ReplayArgs ra{};
            ra.metric = h->metric;
            ra.k = (int)k;
            ra.nlist = (uint32_t)nlist;
            ra.nq = (uint32_t)m;
            ra.round_probes = 1;
            ra.dist = h->w_dist.as<float>();
            ra.seg_off = h->c_seg_off.as<uint64_t>();
            ra.seg_list = h->c_seg_list.as<int32_t>();
            ra.seg_count = h->c_seg_count.as<uint32_t>();
            ra.identity_ids = 1;
            ra.finalize_all = 1;
            ra.heap_val = h->c_heap_val.as<float>();
            ra.heap_ref = h->c_heap_ref.as<int64_t>();
            ra.stage = h->c_stage.as<uint32_t>();
            ra.nscan = h->c_nscan.as<unsigned long long>();
            ra.done = h->c_done.as<uint32_t>();
            ra.D = d_out_dis + c0 * nprobe;
            ra.I = d_out_keys + c0 * nprobe;
            ra.stats = h->w_misc.as<unsigned long long>();
            ra.error = reinterpret_cast<uint32_t*>(h->w_misc.as<unsigned long long>() + STATS_ROWS * 4);
            launch_replay(ra, s);
        } else {
            // The entries behind the prefix come out as (neutral, -1) -- and stay so: when the ranking of the search before went into
            // the same buffers with the same prefix and nothing else has sized (i.e. may have written) them since, they are not written
            // again.  (Writers inside the prefix -- the heap's order, the patch -- do not matter.)
            bool tail_neutral = false;
            if (sort_rows_ranks_a_prefix((uint32_t)nlist, (uint32_t)nprobe, (uint32_t)prefix) && c0 == 0 && m == n && d_out_dis == h->w_cdis.p &&
                d_out_keys == h->w_ckeys.p) {
                amd_ivf::CoarseTail& ct = h->coarse_tail;
                tail_neutral = ct.valid && ct.dis == d_out_dis && ct.keys == d_out_keys && ct.prefix == prefix && ct.nprobe == nprobe && ct.rows >= n &&
                               ct.metric == h->metric && ct.touch_dis + 1 == h->w_cdis.touch && ct.touch_keys + 1 == h->w_ckeys.touch &&
                               !getenv("AUNCEL_AMD_WRITE_TAILS");
                ct = amd_ivf::CoarseTail{true, d_out_dis, d_out_keys, prefix, nprobe, tail_neutral ? ct.rows : n, h->metric, h->w_cdis.touch, h->w_ckeys.touch};
            } else {
                h->coarse_tail.valid = false;
            }
            launch_sort_rows(h->w_dist.as<float>(), (uint32_t)m, (uint32_t)nlist, (uint32_t)nprobe, h->metric,
                             d_out_dis + c0 * nprobe, d_out_keys + c0 * nprobe, s, (uint32_t)prefix, tail_neutral);
            if (heap_ties) {
                if (!h->w_tie_rows.p) {
                    h->w_tie_rows.ensure(8);
                    HIP_CHECK(hipMemsetAsync(h->w_tie_rows.p, 0, 8, s));
                }
                const size_t nout = prefix && prefix < nprobe ? prefix : nprobe;
                const bool ran = launch_heap_tie_order(h->w_dist.as<float>(), (uint32_t)m, (uint32_t)nlist, (uint32_t)nprobe, (uint32_t)nout,
                                                       h->metric, d_out_dis + c0 * nprobe, d_out_keys + c0 * nprobe,
                                                       h->w_tie_rows.as<unsigned long long>(), s);
                // the heap and a row must fit one workgroup's LDS (nlist up to ~13 000): beyond that the reference's tie order is
                // not available -- say so when it was asked for explicitly instead of silently ranking by centroid number
                if (!ran && (h->ties_override == 1 || ties_opt == 1))
                    throw EngineError("coarse tie order by the reference's heap needs nlist x 12 bytes of LDS (nlist <= ~13 000)");
            }
        }
        h->timer.end(t, s);
        // (the staging buffers of this chunk are rewritten by the next chunk / call unless the work list is the cached one; the
        // heap form also stages per call)
        if (!cached || c0 + m < n) {
            HIP_CHECK(stream_sync(s));
            if (c0 == 0 && m == n) {
                h->coarse_sig = csig;
                h->coarse_sig_valid = true;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ searches
void run_rounds_device(amd_ivf* h, const RoundSpec& base, size_t n, size_t first_round, size_t total_nprobe,
                       const unsigned long long* d_np_abs);

// Fixed nprobe, planned on the device from keys that are already there (n x nprobe).  From 16 probes on they are split
// in two rounds: the first nprobe / 8 fill the heap, the rest run in threshold mode (the scan stores and the selection
// reads only what can still enter it; a dense round is bound by those 8 bytes per distance).  Below that one dense
// round is cheaper than a second pass over the lists (5000 queries, 10M x 128, nprobe 8: 2.7 vs 2.1 M queries/s;
// nprobe 32: 1.1 vs 1.3).
void search_fixed_device(amd_ivf* h, const float* d_x, size_t n, size_t k, size_t nprobe, const int64_t* d_keys, float* D,
                         int64_t* I, int store_pairs, size_t max_codes, const IntRange& qr) {
    CallScope call_scope(h);
    upload_lists(h);
    init_state(h, n, k, false);
    RoundSpec base;
    base.k = (int)k;
    base.store_pairs = store_pairs;
    base.max_codes = max_codes;
    base.d_x = d_x;
    base.d_ckeys = d_keys;
    base.coarse_stride = (uint32_t)nprobe;
    base.fused = h->allow_fused && ix(h)->db_range.fusable_with(qr, h->metric);
    base.bytes = byte_queries(h, ix(h), d_x, n, qr);
    ix(h)->last_arith = base.bytes ? 2 : base.fused ? 1 : 0;
    const int two_env = (int)opt(h, OPT_FIXED_ROUNDS, 0);
    // a handful of queries: the second round's planning + synchronisation costs more than threshold mode saves
    // (fp32 lists with the matrix-core filter: a threshold round costs its list bytes, not its distances -- one dense probe gives
    // the thresholds, everything else goes through the filter)
    const bool filt = filter_available(h, ix(h), base.bytes);
    const bool two = two_env ? two_env == 2 : filt ? (nprobe >= 4 && n * nprobe >= 1024) : (nprobe >= 16 && n * nprobe >= 4096);
    base.fixed_two = two;
    const size_t first = two ? (filt ? filter_first_probes(ix(h), k, nprobe) : std::max<size_t>(1, nprobe / 8)) : nprobe;
    base.caller_checks_error = true;
    with_select_fallback(h, [&] {
        if (h->force_heap_select) init_state(h, n, k, false);
        DirectOut direct(h, D, I);
        run_rounds_device(h, base, n, first, nprobe, nullptr);
        finish_results(h, n, k, D, I);
    });
}

void search_fixed_core(amd_ivf* h, const float* d_x, size_t n, size_t k, size_t nprobe, const int64_t* keys, float* D,
                       int64_t* I, int store_pairs, size_t max_codes, const IntRange& qr) {
    upload_lists(h);
    init_state(h, n, k, false);
    RoundSpec r;
    r.slot.resize(n);
    r.p0.assign(n, 0);
    r.cnt.assign(n, (uint32_t)nprobe);
    for (size_t i = 0; i < n; i++) r.slot[i] = (uint32_t)i;
    if (max_codes) {
        // the probe loop stops after the list that brings the visited codes to max_codes (IndexIVF.cpp:541)
        for (size_t i = 0; i < n; i++) {
            size_t nscan = 0;
            for (size_t p = 0; p < nprobe; p++) {
                int64_t key = keys[i * nprobe + p];
                if (key >= 0 && (size_t)key < h->nlist) nscan += ix(h)->h_list_off[key + 1] - ix(h)->h_list_off[key];
                if (nscan >= max_codes) {
                    r.cnt[i] = (uint32_t)(p + 1);
                    break;
                }
            }
        }
    }
    r.keys = keys;
    r.key_stride = nprobe;
    r.k = (int)k;
    r.store_pairs = store_pairs;
    r.max_codes = max_codes;
    r.finalize_all = 1;
    r.d_x = d_x;
    r.fused = h->allow_fused && ix(h)->db_range.fusable_with(qr, h->metric);
    r.bytes = byte_queries(h, ix(h), d_x, n, qr);
    ix(h)->last_arith = r.bytes ? 2 : r.fused ? 1 : 0;
    exec_round(h, r);
    check_device_error(h);
    HIP_CHECK(hipMemcpyAsync(D, h->w_D.p, n * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(I, h->w_I.p, n * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    fold_stats(h, n);
}

void search_full(amd_ivf* h, const float* d_x, size_t n, size_t k, size_t nprobe, int coarse_mode, float* D, int64_t* I, const IntRange& qr) {
    h->w_cdis.ensure(n * nprobe * 4);
    h->w_ckeys.ensure(n * nprobe * 8);
    coarse_dev(h, d_x, n, nprobe, coarse_mode, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(),
               h->allow_fused && ix(h)->centroid_range.fusable_with(qr, h->metric));
    static const bool host_plan = getenv("AUNCEL_AMD_HOST_PLAN") != nullptr;
    if (!host_plan) {
        search_fixed_device(h, d_x, n, k, nprobe, h->w_ckeys.as<int64_t>(), D, I, 0, 0, qr);
        return;
    }
    std::vector<int64_t> keys(n * nprobe);
    HIP_CHECK(hipMemcpyAsync(keys.data(), h->w_ckeys.p, n * nprobe * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    search_fixed_core(h, d_x, n, k, nprobe, keys.data(), D, I, 0, 0, qr);
}

TunerDev make_tuner(amd_ivf* h, size_t query_topk, float multipler, float std_m, const float* d_req, const float* d_gt,
                    unsigned long long* d_np, float* d_tr, int profile) {
    TunerDev t{};
    t.enabled = 1;
    t.profile = profile & 1;
    t.overhead = (profile & 2) ? 1 : 0;
    t.max_topk = (uint32_t)ix(h)->tuner_max_topk;
    t.query_topk = (uint32_t)query_topk;
    t.ntraces = (uint32_t)ix(h)->tuner_ntraces;
    t.multipler = multipler;
    t.std_m = std_m;
    t.interdis = ix(h)->d_interdis.as<float>();
    t.arcos = ix(h)->d_arcos.as<float>();
    t.trace_off = ix(h)->d_trace_off.as<uint32_t>();
    t.trace_x = ix(h)->d_trace_x.as<float>();
    t.trace_y = ix(h)->d_trace_y.as<float>();
    t.trace_std = ix(h)->d_trace_std.as<float>();
    t.require_acc = d_req;
    t.gt_D = d_gt;
    t.my_nprobe = d_np;
    t.t_recalls = d_tr;
    return t;
}

// Multi-round driver with the round planning on the device (ivf_plan.hip).
//
// Chained (the default): the host enqueues plan -> scan -> selection for several rounds back to back without reading
// anything back.  Every launch of a round takes its size from the counters the planning kernels leave on the device: the
// scans are resident grids walking a device-side item count, the selection is sized by the number of queries and its
// surplus waves leave at once.  After a batch of rounds the next round is planned and only then does the host look (one
// 96-byte read-back): no active query left -> done.  A fixed-nprobe search whose rounds provably fit the buffers needs no
// look at all.
// Synchronous (range search, time-bounded search, AUNCEL_AMD_SYNC_ROUNDS=1): one read-back per round; the range search lays
// out its results and the time-bounded search reads the clock between rounds.
void run_rounds_device(amd_ivf* h, const RoundSpec& base, size_t n, size_t first_round_in, size_t total_nprobe,
                       const unsigned long long* d_np_abs /* may be null */) {
    amd_ivf* I = ix(h);
    size_t first_round = first_round_in;
    const size_t nlist = h->nlist;
    hipStream_t s = h->stream;
    static const bool sync_env = getenv("AUNCEL_AMD_SYNC_ROUNDS") != nullptr || getenv("AUNCEL_AMD_DEBUG_REPLAY") != nullptr;
    const bool chained = !(base.range || base.d_budget_ms || sync_env);
    // over-scan against rounds: the byte-code scan is bound by its one pass over the lists, not by the pairs it computes, so
    // its rounds grow fast (12 -> 144 -> all); the fp32 scans pay for every distance (12 -> 42 -> 147)
    const double grow_env = opt(h, OPT_ROUND_GROW, 0.0);
    // fp32 lists with the matrix-core filter (ivf_filter.hip): threshold rounds cost their list bytes too, and the first round --
    // the only one computed on the vector ALU in the reference's rounding sequence -- shrinks to one probe per query
    const bool filter_ok = filter_available(h, I, base.bytes) && !base.range;
    const bool filter_half = filter_ok && ensure_frag16(I);
    if (filter_ok && !filter_half) ensure_frag32(I);
    const float* lanes = base.bytes ? nullptr : ensure_lanes(I);  // (fp32 tiles: scan_lanes_kernel over the lane-ordered copy)
    // the engine's own limit on fp32 searches that share the device (callers may have any number of searches in flight)
    struct Fp32Gate {
        amd_ivf* ix;
        bool held;
        // (Callers with threads of their own may block here on other callers' searches, in the order they arrive; the limit is read
        // again whenever a waiter is woken, so changing the option while searches wait takes effect.  The handle's own asynchronous
        // searches never park here: the pool does not start more fp32 searches than the limit, async_running_limit.)
        Fp32Gate(amd_ivf* index, bool wanted) : ix(index), held(false) {
            if (!wanted || (int)index->opt.get(OPT_FP32_IN_FLIGHT, 4) <= 0) return;
            std::unique_lock<std::mutex> lk(ix->fp32_gate_mu);
            const uint64_t mine = ix->fp32_gate_next++;
            ix->fp32_gate_cv.wait(lk, [&] {
                const int limit = (int)ix->opt.get(OPT_FP32_IN_FLIGHT, 4);
                return mine == ix->fp32_gate_serving && (limit <= 0 || ix->fp32_running < limit);
            });
            ix->fp32_gate_serving++;
            ix->fp32_running++;
            held = true;
            lk.unlock();
            ix->fp32_gate_cv.notify_all();  // (the next in line may fit as well)
        }
        ~Fp32Gate() {
            if (!held) return;
            {
                std::lock_guard<std::mutex> lk(ix->fp32_gate_mu);
                ix->fp32_running--;
            }
            ix->fp32_gate_cv.notify_all();
        }
    } fp32_gate(I, !base.bytes && n >= 256);
    // (an adaptive search ends most queries in its first rounds: a first round of 2 probes instead of 3 sent more of them through the
    // threshold rounds and cost the bench workload's fp32 form 9 % -- it keeps the 1.5 % rule)
    if (filter_ok && base.tuner.enabled) first_round = filter_first_probes(I, (size_t)base.k, 64, 0.015);
    // (byte codes: a round is bound by its one pass over the lists, x 12; the fp32 filter's rounds are bound by matrix-core issue
    // from ~50 queries per list on, so what a round scans past the queries' stop points is paid for: x 6 measured best --
    // fp32_path 0.69 / 0.88 / 0.90 / 0.84 M q/s at x 12 / 8 / 6 / 4)
    const double grow = grow_env > 0 ? grow_env : base.bytes ? 12.0 : filter_ok ? 6.0 : 3.5;
    // pairs of a round: the packed query tiles of the fp32 scans (8 queries x dpad floats per group) must fit 4 GiB
    size_t seg_cap = (size_t)2 << 20;
    if (!base.bytes) {
        const size_t groups = ((size_t)4 << 30) / ((size_t)h->dpad * SCAN_RQ * sizeof(float));
        seg_cap = std::min(seg_cap, std::max<size_t>((size_t)64 << 10, (groups > nlist ? groups - nlist : 0) * SCAN_RQ));
    }
    size_t maxlist = 0;
    for (size_t l = 0; l < nlist; l++) maxlist = std::max<size_t>(maxlist, I->h_list_off[l + 1] - I->h_list_off[l]);
    const size_t item_cap = (seg_cap / 32 + nlist) * ((maxlist + SCAN_WAVE_VECS - 1) / SCAN_WAVE_VECS) + nlist * 4 + 16;
    // rows are padded to multiples of 64 floats (one mask word covers 64 candidates of one row)
    // A query's rows of a round are consecutive, one (padded) list length per probe, whether the scan stores every distance
    // (round 0) or the < 1 % that beat the threshold: after round 0 the buffer is address space more than traffic, and a
    // round that does not fit is cut (plan_prefix_kernel defers the remaining queries: another pass over the lists).  8 GiB
    // hold 2500 unfinished queries x 144 probes of the bench workload; small searches take what they can ever need.
    static const size_t budget_env = getenv("AUNCEL_AMD_DIST_BUDGET_MB") ? (size_t)atol(getenv("AUNCEL_AMD_DIST_BUDGET_MB")) << 18 : 0;
    const size_t padded_max = (maxlist + 1023) & ~(size_t)1023;
    const double all_rows = (double)n * (double)std::min<size_t>(total_nprobe, nlist) * (double)padded_max;
    // The workspace follows the round schedule instead of a flat 8 GiB per context: what the rounds of the last
    // search of this shape wanted (plan counters[12], + an eighth), and before any history the first round's rows -- a round
    // that wants more than it gets defers the queries that do not fit (another pass, same results) and the next search of the
    // shape is sized for it.  Upper limit as before: 2^31 floats.
    size_t budget = budget_env ? budget_env : std::max<size_t>(h->dist_budget_floats, (size_t)2 << 30);
    const uint64_t bsig = (uint64_t)n * 1000003u ^ (uint64_t)first_round_in * 10007u ^ (uint64_t)total_nprobe * 101u ^ (base.bytes ? 1u : 0u) ^
                          (base.tuner.enabled ? 2u : 0u) ^ (base.train.enabled ? 4u : 0u) ^ ((uint64_t)base.k << 40);
    // (adaptive searches: a fixed-nprobe search knows its rows exactly and takes them -- its rounds are enqueued without a look
    // only if nothing can be deferred)
    if (!budget_env && chained && base.tuner.enabled) {
        size_t want;
        size_t known = h->dist_want.get(bsig);
        {
            amd_ivf* owner = ix(h);
            std::lock_guard<std::mutex> lock(owner->want_mu);
            known = std::max(known, owner->shared_want.get(bsig));
        }
        if (known) {
            want = known + known / 4;
        } else {
            // no history: the first round's rows, and for the round behind it a quarter of the queries going on for `grow` times as
            // many probes (what the bench workload does; a shape that wants more defers queries once and is sized for it next time)
            const double first = (double)n * (double)std::min<size_t>(std::max<size_t>(first_round_in, 1), std::min<size_t>(total_nprobe, nlist)) * (double)padded_max;
            want = (size_t)std::min<double>(first * (1.0 + 0.25 * grow), (double)budget);
        }
        budget = std::min(budget, std::max<size_t>(want, (size_t)16 << 20));
    }
    if (all_rows < (double)budget) budget = (size_t)all_rows + 64;
    budget = std::max<size_t>(budget, I->h_list_off[nlist] + 1024 * nlist + 1024);
    if (filter_ok && budget > ((size_t)1 << 31)) budget = (size_t)1 << 31;  // (the filter's survivor entries hold 32-bit row positions)
    if (base.bytes && budget > ((size_t)1 << 31)) throw std::runtime_error("distance rows beyond 2^31 floats (byte-code scan offsets are 32-bit)");
    static const bool no_thr = getenv("AUNCEL_AMD_NO_THRESHOLD") != nullptr;
    h->w_pl_pad.ensure(n * 4);
    h->w_mask.ensure((budget / 64 + 2 + 256) * 8);  // (+ the words the selection's stream requests past a region's end)
    // (1 / 0: always / never; unset: when no other search of this index is running -- alone, a search is the latency of its chain
    // and the lists take 0.06 ms off it; among others it is the sum of the work, to which compact_rows_kernel adds: -2 % at four
    // batches in flight.  Decided below, where the searches are counted.)
    const int row_lists_opt = (int)opt(h, OPT_ROW_LISTS, ROW_LISTS_DEFAULT);
    static const size_t row_lists_min = getenv("AUNCEL_AMD_ROW_LISTS_MIN") ? (size_t)atoi(getenv("AUNCEL_AMD_ROW_LISTS_MIN")) : 256;
    const bool row_lists = row_lists_opt != 0 && n >= row_lists_min;
    if (row_lists) {
        h->w_cl_cnt.ensure(seg_cap * 4);
        h->w_cl_ent.ensure((seg_cap + 64) * CL_CAP * sizeof(uint2));  // (+ a probe window's worth behind a query's last row)
        h->w_cl_arena.ensure((size_t)CL_ARENA * sizeof(uint2));
        h->w_cl_cursor.ensure(8 * 32 * 4);
    }
    h->w_pl_cnt.ensure(n * 4);
    h->w_pl_need.ensure(n * 8);
    h->w_seg_begin.ensure(n * 4);
    h->w_pl_dist_base.ensure(n * 8);
    h->w_qsel.ensure(n * 4);
    h->w_seg_list.ensure(seg_cap * 4);
    h->w_seg_off.ensure(seg_cap * 8);
    h->w_pair_query.ensure(seg_cap * 4);
    h->w_pair_out.ensure(seg_cap * 8);
    h->w_group_p0.ensure(seg_cap * 4);
    h->w_group_cnt.ensure(seg_cap * 4);
    h->w_items.ensure(item_cap * sizeof(ScanItem));
    h->w_pl_lcount.ensure(nlist * 4);
    h->w_pl_lstart.ensure(nlist * 4);
    h->w_pl_gbase.ensure(nlist * 4);
    h->w_pl_ibase.ensure(4 * nlist * 4);
    h->w_pl_fill.ensure(8 * nlist * 4);  // (per-XCD pair counts)
    h->w_seg_slot.ensure(seg_cap * 4);
    // 16 uint32 counters | double bytes | 2 x u64 slot bookkeeping | double min_bytes | per-round unfinished | double min_bytes of threshold rounds
    constexpr size_t CNT_WORDS = 24 + PLAN_MAX_ROUNDS * 8 + 2;  // (... | per round and XCD: unfinished | ...)
    constexpr size_t STAT_WORDS = STATS_ROWS * 4 * 2;            // (the statistics rows ride along with a look: speculative finish)
    h->w_pl_counters.ensure(CNT_WORDS * 4);
    h->p_counters.ensure((CNT_WORDS + STAT_WORDS) * 4);
    h->w_dist.ensure((budget + 4096) * sizeof(float));  // (+ the blocks the selection's stream requests past a region's end)
    // (the counters need no memset: the first planning pass of the search zeroes what accumulates, PlanArgs::first_plan)
    // sorted-array selection: global positions must fit 32 bits; a query's admission log holds 32 k entries (k (1 + ln(N / k))
    // are expected: ~5 k for a million candidates), beyond which the call is repeated with the heap kernels
    const size_t log_cap = std::min<size_t>(4096, (std::max<size_t>(256, 32 * (size_t)base.k) + 63) & ~(size_t)63);
    const bool sorted_ok = !base.range && !base.train.enabled && !base.raw_heap_out && base.k <= 128 && I->h_list_off[nlist] < 0xffffffffull &&
                           !h->force_heap_select && opt(h, OPT_SELECT, 1) != 0;
    if (sorted_ok) h->w_log.ensure(n * log_cap * 8);
    // groups of 8 pairs never cross a list: at most pairs / 8 + one partial group per list
    const size_t group_cap = seg_cap / SCAN_RQ + nlist;
    if (chained && !base.bytes) h->w_qtile.ensure(group_cap * (size_t)h->dpad * SCAN_RQ * sizeof(float));

    PlanArgs pa{};
    pa.nq = (uint32_t)n;
    pa.nlist = (uint32_t)nlist;
    pa.total_nprobe = (uint32_t)total_nprobe;
    pa.key_stride = base.coarse_stride;
    pa.slot_base = 0;
    pa.first_round = (uint32_t)first_round;
    const size_t inc_env = (size_t)opt(h, OPT_ROUND_INC, 0);
    pa.min_inc = (uint32_t)(inc_env ? inc_env : filter_ok && base.tuner.enabled ? 12 : first_round);
    pa.tune = base.tuner.enabled;
    pa.d = h->d;
    pa.multipler = base.tuner.multipler;
    pa.grow = std::max<double>(base.tuner.multipler, grow);
    pa.id_offset = base.id_offset;
    pa.dist_budget = budget;
    pa.seg_cap = (uint32_t)seg_cap;
    pa.keys = base.d_ckeys;
    pa.list_off = I->d_list_off.as<uint64_t>();
    pa.stage = h->w_stage.as<uint32_t>();
    pa.done = h->w_done.as<uint32_t>();
    pa.my_nprobe = d_np_abs;
    pa.cnt = h->w_pl_cnt.as<uint32_t>();
    pa.need = h->w_pl_need.as<unsigned long long>();
    pa.pad = h->w_pl_pad.as<uint32_t>();
    pa.row_align = sorted_ok ? 1024 : 64;  // (select_sorted_kernel reads dense rows in groups of four blocks of 256 candidates)
    h->row_align_now = pa.row_align;
    pa.qblock = scan_qblock(base.bytes);
    pa.mfma_qblock = MFMA_QBLOCK;
    static const int item_order_env = getenv("AUNCEL_AMD_ITEM_ORDER") ? atoi(getenv("AUNCEL_AMD_ITEM_ORDER")) : 0;
    pa.item_order = item_order_env;
    pa.lane_block_off = lanes ? I->d_block_off.as<uint64_t>() : nullptr;
    if (base.bytes) {
        pa.mfma_chunk = mfma_chunk();
        pa.block_off = I->d_block_off.as<uint64_t>();
    }
    constexpr size_t SURV_CAP = (size_t)16 << 20;
    if (filter_ok) {
        h->w_xf.ensure(n * (size_t)filter_steps(h->d) * 8 * sizeof(float));  // (the fp16 rows are shorter)
        h->w_xn.ensure(n * sizeof(float));
        h->w_surv.ensure(SURV_CAP * sizeof(uint4));
        h->w_surv_cnt.ensure(4);
        if (filter_half) {
            h->w_qinfo.ensure(16);
            h->w_fparams.ensure(2 * sizeof(FilterParams));
            HIP_CHECK(hipMemsetAsync(h->w_qinfo.p, 0, 16, s));
            launch_amax(base.d_x, n, h->dpad, h->w_qinfo.as<uint32_t>(), s);
            launch_filter_queries16(base.d_x, n, h->d, h->dpad, h->metric, h->w_qinfo.as<uint32_t>(), I->d_yinfo.as<uint32_t>(),
                                    h->w_xf.as<float>(), h->w_xn.as<float>(), h->w_fparams.as<FilterParams>(), s);
        } else {
            launch_filter_queries(base.d_x, n, h->d, h->dpad, h->metric, h->w_xf.as<float>(), h->w_xn.as<float>(), s);
        }
    }
    pa.seg_begin = h->w_seg_begin.as<uint32_t>();
    pa.dist_base = h->w_pl_dist_base.as<unsigned long long>();
    pa.qsel = h->w_qsel.as<uint32_t>();
    pa.seg_list = h->w_seg_list.as<int32_t>();
    pa.seg_off = h->w_seg_off.as<uint64_t>();
    pa.lcount = h->w_pl_lcount.as<uint32_t>();
    pa.lstart = h->w_pl_lstart.as<uint32_t>();
    pa.gbase = h->w_pl_gbase.as<uint32_t>();
    pa.ibase = h->w_pl_ibase.as<uint32_t>();
    pa.xcount = h->w_pl_fill.as<uint32_t>();
    pa.seg_slot = h->w_seg_slot.as<uint32_t>();
    pa.pair_query = h->w_pair_query.as<uint32_t>();
    pa.pair_out = h->w_pair_out.as<uint64_t>();
    pa.group_p0 = h->w_group_p0.as<uint32_t>();
    pa.group_cnt = h->w_group_cnt.as<uint32_t>();
    pa.items = h->w_items.as<ScanItem>();
    pa.item_cap = (uint32_t)item_cap;
    pa.counters = h->w_pl_counters.as<uint32_t>();
    pa.cl_cursor = row_lists ? h->w_cl_cursor.as<uint32_t>() : nullptr;
    pa.bytes = reinterpret_cast<double*>(h->w_pl_counters.as<uint32_t>() + 16);
    pa.acc64 = reinterpret_cast<unsigned long long*>(h->w_pl_counters.as<uint32_t>() + 18);
    pa.min_bytes = reinterpret_cast<double*>(h->w_pl_counters.as<uint32_t>() + 22);
    pa.min_bytes_thr = reinterpret_cast<double*>(h->w_pl_counters.as<uint32_t>() + 24 + PLAN_MAX_ROUNDS * 8);
    pa.row_bytes = base.bytes ? mfma_ksteps(h->d) * 32 : (uint32_t)h->dpad * 4;
    pa.error = h->w_error.as<uint32_t>();
    uint32_t* const d_unfinished = h->w_pl_counters.as<uint32_t>() + 24;  // [PLAN_MAX_ROUNDS][8 XCDs]
    pa.round_unfinished = d_unfinished;
    if (base.d_budget_ms) {
        h->w_limit.ensure(n * 4);
        pa.budget_ms = base.d_budget_ms;
        pa.limit = h->w_limit.as<uint32_t>();
    }

    uint32_t* hc = h->p_counters.as<uint32_t>();
    const uint32_t* dcnt = h->w_pl_counters.as<uint32_t>();
    // grid hints: what each round of the previous search of this shape needed
    constexpr size_t MAX_HIST = 32;
    const uint64_t sig = (uint64_t)n * 1000003u ^ (uint64_t)first_round * 10007u ^ (uint64_t)total_nprobe * 101u ^ (base.bytes ? 1u : 0u) ^
                         (base.tuner.enabled ? 2u : 0u) ^ (base.train.enabled ? 4u : 0u) ^ ((uint64_t)base.k << 40);
    if (chained) {
        h->w_pl_hist.ensure(MAX_HIST * 64);
        h->p_hist.ensure(MAX_HIST * 64);
        if (h->hint_sig != sig) h->round_hint.clear();
        h->hint_sig = sig;
    }
    auto hint_of = [&](size_t round, int counter) -> uint32_t {
        return chained && (round + 1) * 16 <= h->round_hint.size() ? h->round_hint[round * 16 + counter] : 0u;
    };
    const std::vector<uint32_t> hints_used = chained ? h->round_hint : std::vector<uint32_t>();  // (compared with what the rounds needed)
    h->hinted_rounds = h->short_rounds = 0;
    h->filter_launches = 0;
    size_t planned_rounds = 0;  // plans launched so far (round r's counters reach history[r] when round r + 1 is planned)
    // searches of this index that were running when this one started (the tie replay, the candidate lists and the scan's stream
    // are chosen by it: alone, a search is the latency of its chain; among others, the sum of the work)
    struct Active {
        std::atomic<int>& c;
        int before;
        explicit Active(std::atomic<int>& cc) : c(cc), before(cc.fetch_add(1)) {}
        ~Active() { c.fetch_sub(1); }
    } active(I->active_searches);
    // the form of the byte-code scan shapes the planner's items (queries per item) AND picks the kernel that walks them: read once
    const int scan_pipelined = (int)opt(h, OPT_SCAN_PIPELINED, 7);
    auto plan_round = [&](size_t round_len) {
        pa.round_len = (uint32_t)round_len;
        pa.dense_round = !(base.range || (planned_rounds > 0 && !no_thr));
        if (base.bytes) {
            pa.mfma_chunk = pa.dense_round || base.range ? mfma_chunk() : mfma_chunk_thr();
            pa.mfma_qblock = pa.dense_round ? MFMA_QBLOCK : mfma_thr_qblock(h->d, scan_pipelined, base.range);
        }
        if (filter_ok) {  // threshold rounds of an fp32 search: items in the matrix-core form (a chunk x a block of 32 queries)
            const bool mf = !pa.dense_round;
            pa.mfma_chunk = mf ? filter_item_vectors(h->d) : 0;
            pa.mfma_qblock = mf ? filter_item_queries(h->d) : MFMA_QBLOCK;
            pa.block_off = mf ? I->d_block_off.as<uint64_t>() : nullptr;
            pa.qblock = scan_qblock(mf);
            pa.row_bytes = mf && filter_half ? filter_steps16(h->d) * 32u : (uint32_t)h->dpad * 4;  // (what a pass streams per vector)
        }
        pa.history = chained && planned_rounds >= 1 && planned_rounds <= MAX_HIST ? h->w_pl_hist.as<uint32_t>() + (planned_rounds - 1) * 16 : nullptr;
        pa.first_plan = planned_rounds == 0;
        // (only while the ranking can still change: the rows of round 0 are reordered in place, later rounds are planned on the patched keys)
        pa.run_dis = base.run_ties && planned_rounds == 0 ? base.d_cdis : nullptr;
        size_t t = h->timer.begin(CAT_PLAN, s);
        launch_plan(pa, s);
        h->timer.end(t, s);
        planned_rounds++;
    };
    static const bool xcd_off = getenv("AUNCEL_AMD_NO_XCD_CHUNKS") != nullptr;
    // side streams of an fp32 round's tile shapes.  Round 4 gave every shape its own (each filled the chip's LDS by itself); the
    // lane-ordered kernel holds no LDS, the shapes share the chip from one stream just as well (cfg 3 / 5 and a lone fp32 batch: the
    // same within a run's spread at 4 / 2 / 1) -- and with four low-class streams a search, the fifth fp32 search in flight pushed
    // them three to a hardware queue: 0.31 M q/s at five at a time, 0.20 with a round growth of 5, against 1.41 and 1.42 with one
    // (profiles/r05_experiments.txt Y2)
    static const int nstreams = getenv("AUNCEL_AMD_SCAN_STREAMS") ? atoi(getenv("AUNCEL_AMD_SCAN_STREAMS")) : 1;

    // ---- the scan of a planned round.  counts == nullptr: sizes on the device (chained); else the counters read back.
    auto enqueue_scan = [&](bool thr_mode, const uint32_t* counts, size_t round) {
        if (counts && counts[CNT_PAIRS] == 0) return;
        size_t t = h->timer.begin(thr_mode ? CAT_SCAN_THR : CAT_SCAN, s);
        if (base.bytes) {
            // byte codes: one launch of scan_mfma_kernel (no query packing: the A operand is gathered from the query matrix)
            MfmaScanArgs ma{};
            ma.codes_frag = I->d_frag.as<uint8_t>();
            ma.code_cy = I->d_cy.as<int32_t>();
            ma.queries8 = h->w_x8.as<int8_t>();
            ma.query_cx = h->w_xnorm8.as<int32_t>();
            ma.items = h->w_items.as<ScanItem>();
            ma.pair_query = h->w_pair_query.as<uint32_t>();
            ma.pair_out = h->w_pair_out.as<uint64_t>();
            ma.dist = h->w_dist.as<float>();
            ma.d = h->d;
            ma.metric = h->metric;
            ma.xcd_chunks = xcd_off ? 0 : 1;
            ma.nitems = counts ? counts[CNT_QG8] : 0;
            ma.dev_nitems = counts ? nullptr : dcnt + CNT_QG8;
            ma.hint_nitems = hint_of(round, CNT_QG8);
            ma.pipelined = scan_pipelined;
            static const int scan_debug = getenv("AUNCEL_AMD_SCAN_DEBUG") ? atoi(getenv("AUNCEL_AMD_SCAN_DEBUG")) : 0;
            ma.debug = scan_debug;
            if (thr_mode) {
                ma.thr = h->w_thr.as<float>();
                ma.mask = h->w_mask.as<unsigned long long>();
                ma.exact_mask = base.range;  // (range search counts the mask bits)
            }
            // the scan runs on a normal-priority side stream (see make_main_stream)
            // (a handful of queries: the launch is a few microseconds of work, the fork and join around it two event waits of
            // ~13 us each -- it stays on the search's own stream)
            // (so does a search that has the index to itself: there is nobody whose selection the side stream's lower priority would
            // let pass)
            // Round 5: on the search's own stream always.  With six searches in flight through the engine's own contexts, every stream
            // on a hardware queue of its own and the background work (heap order of coarse ties) beside them, the fork and join around
            // a side stream cost more than letting other searches' selections pass the scans bought: 2.7 -> 3.05 M q/s in the exact
            // tie regime, 3.2 -> 3.5 with runs in centroid-number order (profiles/r05_experiments.txt).  AUNCEL_AMD_SCAN_ON_MAIN=0: the
            // side stream (low priority class) for calls of 20 queries and more.
            static const int scan_on_main = getenv("AUNCEL_AMD_SCAN_ON_MAIN") ? atoi(getenv("AUNCEL_AMD_SCAN_ON_MAIN")) : 1;
            if (scan_on_main > 0 || n < 20) {
                launch_scan_mfma(ma, s);
            } else {
                ensure_aux(h, 3, 3);
                HIP_CHECK(hipEventRecord(h->ev_fork, s));
                HIP_CHECK(hipStreamWaitEvent(h->aux[3], h->ev_fork, 0));
                launch_scan_mfma(ma, h->aux[3]);
                HIP_CHECK(hipEventRecord(h->ev_join[3], h->aux[3]));
                HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[3], 0));
            }
        } else if (filter_ok && thr_mode) {
            FilterScanArgs fa{};
            fa.codes_frag = filter_half ? I->d_frag16.as<float>() : I->d_frag32.as<float>();
            fa.half = filter_half ? 1 : 0;
            fa.params = filter_half ? h->w_fparams.as<float>() : nullptr;
            fa.yn = I->d_yn.as<float>();
            fa.xf = h->w_xf.as<float>();
            fa.xn = h->w_xn.as<float>();
            fa.codes = I->d_codes.as<float>();
            fa.queries = base.d_x;
            fa.items = h->w_items.as<ScanItem>();
            fa.pair_query = h->w_pair_query.as<uint32_t>();
            fa.pair_out = h->w_pair_out.as<uint64_t>();
            fa.dist = h->w_dist.as<float>();
            fa.thr = h->w_thr.as<float>();
            fa.mask = h->w_mask.as<unsigned long long>();
            fa.surv = h->w_surv.as<uint4>();
            fa.surv_count = h->w_surv_cnt.as<uint32_t>();
            fa.surv_cap = (uint32_t)SURV_CAP;
            fa.d = h->d;
            fa.dpad = h->dpad;
            fa.metric = h->metric;
            fa.xcd_chunks = xcd_off ? 0 : 1;
            fa.nitems = counts ? counts[CNT_QG8] : 0;
            fa.dev_nitems = counts ? nullptr : dcnt + CNT_QG8;
            fa.hint_nitems = hint_of(round, CNT_QG8);
            HIP_CHECK(hipMemsetAsync(h->w_surv_cnt.p, 0, 4, s));
            ensure_aux(h, 3, 3);
            HIP_CHECK(hipEventRecord(h->ev_fork, s));
            HIP_CHECK(hipStreamWaitEvent(h->aux[3], h->ev_fork, 0));
            launch_scan_filter(fa, h->aux[3]);
            h->filter_launches++;
            HIP_CHECK(hipEventRecord(h->ev_join[3], h->aux[3]));
            HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[3], 0));
        } else {
            const size_t ngroups = counts ? counts[CNT_GROUPS] : group_cap;
            if (counts) h->w_qtile.ensure(ngroups * (size_t)h->dpad * SCAN_RQ * sizeof(float));
            launch_pack_queries(base.d_x, h->w_pair_query.as<uint32_t>(), h->w_group_p0.as<uint32_t>(), h->w_group_cnt.as<uint32_t>(), ngroups,
                                h->dpad, h->w_qtile.as<float>(), s, counts ? nullptr : dcnt + CNT_GROUPS, hint_of(round, CNT_GROUPS));
            ScanArgs sa{};
            sa.qtile = h->w_qtile.as<float>();
            sa.codes = I->d_codes.as<float>();
            sa.lanes = lanes;
            sa.queries = base.d_x;
            sa.items = h->w_items.as<ScanItem>();
            sa.pair_query = h->w_pair_query.as<uint32_t>();
            sa.pair_out = h->w_pair_out.as<uint64_t>();
            sa.dist = h->w_dist.as<float>();
            sa.d = h->dpad;
            sa.metric = h->metric;
            sa.fused = base.fused;
            sa.xcd_chunks = xcd_off ? 0 : 1;  // measured: 3 % off the scan launches of the bench workload
            sa.dev_counts = counts ? nullptr : dcnt;
            sa.hint_qg[0] = hint_of(round, CNT_QG1);
            sa.hint_qg[1] = hint_of(round, CNT_QG2);
            sa.hint_qg[2] = hint_of(round, CNT_QG4);
            sa.hint_qg[3] = hint_of(round, CNT_QG8);
            sa.hint_valid = chained && (round + 1) * 16 <= h->round_hint.size();
            if (thr_mode) {
                sa.thr = h->w_thr.as<float>();
                sa.mask = h->w_mask.as<unsigned long long>();
            }
            size_t n_qg[4] = {0, 0, 0, 0};
            if (counts) {
                n_qg[0] = counts[CNT_QG1];
                n_qg[1] = counts[CNT_QG2];
                n_qg[2] = counts[CNT_QG4];
                n_qg[3] = counts[CNT_QG8];
            }
            // the shapes on the side stream(s) (see make_main_stream); streams that carry nothing are not made at all
            const bool use_aux[4] = {nstreams >= 2, nstreams >= 3, nstreams >= 3, true};
            ensure_aux(h, nstreams >= 2 ? 0 : 3, 3);
            HIP_CHECK(hipEventRecord(h->ev_fork, s));
            for (int i = 0; i < 4; i++)
                if (use_aux[i]) HIP_CHECK(hipStreamWaitEvent(h->aux[i], h->ev_fork, 0));
            if (nstreams <= 1) launch_scan(sa, n_qg, h->aux[3], h->aux[3], h->aux[3], h->aux[3]);
            else if (nstreams == 2) launch_scan(sa, n_qg, h->aux[3], h->aux[0], h->aux[0], h->aux[3]);  // 8,4 | 2,1
            else launch_scan(sa, n_qg, h->aux[3], h->aux[0], h->aux[1], h->aux[2]);
            for (int i = 0; i < 4; i++)
                if (use_aux[i]) {
                    HIP_CHECK(hipEventRecord(h->ev_join[i], h->aux[i]));
                    HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[i], 0));
                }
        }
        h->timer.end(t, s);
    };

    bool fix_pending = false, sorted_any = false;
    size_t last_fix_round = 0;
    // tie_fix_kernel: behind every round on a side stream when this search has the index to itself (latency), one pass at the end
    // when other searches are running on it (their kernels fill the GPU while it runs; more streams would only crowd the hardware
    // queues)
    const int fix_opt = (int)opt(h, OPT_TIE_FIX, -1);
    // ... and only where equal distances are common (integer-valued data: a third of the bench workload's queries): replaying
    // every unfinished query's admissions after every round is then cheaper than replaying the flagged queries' whole logs at the
    // end.  Where they are rare (float data: a query in a few hundred) the end pass is a handful of waves and the per-round one
    // would only compete with the next round's planning and scan (DEEP-like configuration: 1.2 ms of it per search).
    const float seen_rate = I->tie_rate.load();
    const bool ties_common = seen_rate >= 0.f ? seen_rate > 0.02f : (base.bytes || base.fused);
    // (a handful of queries: the per-round replay is one wave's serial work on the critical path of a search that is all latency)
    const bool eager_fix = fix_opt >= 0 ? fix_opt == 1 : (active.before == 0 && ties_common && n >= 512);
    bool fix_due = false;
    size_t fix_due_round = 0;
    auto tie_fix_args = [&](uint32_t round, int final_pass) {
        TieFixArgs ta{};
        ta.metric = h->metric;
        ta.k = base.k;
        ta.nq = (uint32_t)n;
        ta.nlist = (uint32_t)nlist;
        ta.log = h->w_log.as<uint2>();
        ta.log_cap = (uint32_t)log_cap;
        ta.round = round;
        ta.final_pass = final_pass;
        ta.log_cnt = h->w_log_cnt.as<uint32_t>();
        ta.log_snap = h->w_log_snap.as<uint32_t>();
        ta.fin_round = h->w_fin_round.as<uint32_t>();
        ta.fix_val = h->w_fix_val.as<float>();
        ta.fix_ref = h->w_fix_ref.as<int64_t>();
        ta.fix_pos = h->w_fix_pos.as<uint32_t>();
        ta.tie_flag = h->w_tie_flag.as<uint32_t>();
        ta.list_off = I->d_list_off.as<uint64_t>();
        ta.ids = I->d_ids.as<int64_t>();
        ta.store_pairs = base.store_pairs;
        ta.D = h->out_D ? h->out_D : h->w_D.as<float>();
        ta.I = h->out_I ? h->out_I : h->w_I.as<int64_t>();
        return ta;
    };
    // ---- ordered selection of a scanned round.  nact: active queries (sync) or the bound n with the count on the device.
    auto enqueue_replay = [&](bool thr_mode, uint32_t nact, bool on_device, size_t round, const uint32_t* only = nullptr,
                              const uint32_t* only_count = nullptr) {
        ReplayArgs ra{};
        ra.metric = h->metric;
        ra.k = base.k;
        ra.nlist = (uint32_t)nlist;
        ra.nq = nact;
        ra.nq_dev = only ? only_count : on_device ? dcnt + CNT_ACTIVE : nullptr;
        ra.nq_hint = on_device && round > 0 ? std::max<uint32_t>(1, nact / 3) : nact;
        ra.qsel = only ? only : h->w_qsel.as<uint32_t>();  // (only: a part of the round's queries -- RoundSpec::split_*)
        ra.total_nprobe = (uint32_t)total_nprobe;
        ra.round_probes = 0;
        ra.id_offset = base.id_offset;
        ra.dist = h->w_dist.as<float>();
        ra.mask = thr_mode ? h->w_mask.as<unsigned long long>() : nullptr;
        ra.thr = no_thr ? nullptr : h->w_thr.as<float>();
        ra.seg_off = h->w_seg_off.as<uint64_t>();
        ra.seg_list = h->w_seg_list.as<int32_t>();
        ra.seg_count = h->w_pl_cnt.as<uint32_t>();
        ra.seg_begin = h->w_seg_begin.as<uint32_t>();
        ra.seg_by_slot = 1;
        ra.list_off = I->d_list_off.as<uint64_t>();
        ra.ids = I->d_ids.as<int64_t>();
        ra.store_pairs = base.store_pairs;
        ra.max_codes = base.max_codes;
        ra.heap_val = h->w_heap_val.as<float>();
        ra.heap_ref = h->w_heap_ref.as<int64_t>();
        ra.stage = h->w_stage.as<uint32_t>();
        ra.nscan = h->w_nscan.as<unsigned long long>();
        ra.done = h->w_done.as<uint32_t>();
        ra.pre_val = h->w_pre_val.as<float>();
        ra.stoped = h->w_stoped.as<uint32_t>();
        ra.dtb = h->w_dtb.as<float>();
        ra.coarse_dis = base.d_cdis;
        ra.coarse_keys = base.d_ckeys;
        ra.coarse_stride = base.coarse_stride;
        ra.trace_cap = (uint32_t)I->tuner_trace_cap;
        ra.D = h->out_D ? h->out_D : h->w_D.as<float>();
        ra.I = h->out_I ? h->out_I : h->w_I.as<int64_t>();
        ra.stats = h->w_stats.as<unsigned long long>();
        ra.error = h->w_error.as<uint32_t>();
        ra.tuner = base.tuner;
        ra.train = base.train;
        ra.limit = base.d_budget_ms ? h->w_limit.as<uint32_t>() : nullptr;
        ra.qstat = h->w_qstat.as<uint2>();
        ra.unfinished = round < PLAN_MAX_ROUNDS ? d_unfinished + round * 8 : nullptr;
        if (sorted_ok) {
            ra.log = h->w_log.as<uint2>();
            ra.log_cap = (uint32_t)log_cap;
            ra.log_cnt = h->w_log_cnt.as<uint32_t>();
            ra.amb = h->w_amb.as<uint32_t>();
            ra.tie_flag = h->w_tie_flag.as<uint32_t>();
            ra.round = (uint32_t)round;
            ra.log_snap = h->w_log_snap.as<uint32_t>();
            ra.nq_total = (uint32_t)n;
            ra.fin_round = h->w_fin_round.as<uint32_t>();
        }
        const bool sorted_now = replay_sorted_applies(ra);
        // (tie_fix_kernel of round - 2 read the log counts this round's selection is about to overwrite)
        if (sorted_now && eager_fix && round >= 2) HIP_CHECK(hipStreamWaitEvent(s, h->ev_fix[round & 1], 0));
        static const bool dbg_replay_dev = getenv("AUNCEL_AMD_DEBUG_REPLAY") != nullptr;
        if (dbg_replay_dev) {
            h->w_misc.ensure((size_t)nact * 64);
            HIP_CHECK(hipMemsetAsync(h->w_misc.p, 0, (size_t)nact * 64, s));
            ra.dbg = h->w_misc.as<unsigned long long>();
        }
        {
            size_t t = h->timer.begin(thr_mode ? CAT_SELECT_THR : CAT_SELECT, s);
            if (thr_mode && sorted_now && row_lists && (row_lists_opt > 0 || active.before == 0)) {
                CompactArgs ca{};
                ca.nseg_dev = dcnt + CNT_SEGMENTS;
                ca.nseg_hint = (uint32_t)std::min<size_t>(seg_cap, (size_t)ra.nq_hint * std::min<size_t>(total_nprobe, 160));
                ca.nlist = (uint32_t)nlist;
                ca.seg_list = ra.seg_list;
                ca.seg_off = ra.seg_off;
                ca.list_off = ra.list_off;
                ca.dist = ra.dist;
                ca.mask = ra.mask;
                ca.cl_cnt = h->w_cl_cnt.as<uint32_t>();
                ca.cl_ent = h->w_cl_ent.as<uint2>();
                ca.arena = h->w_cl_arena.as<uint2>();
                const char* arena_env = getenv("AUNCEL_AMD_CL_ARENA");  // (tests: a small arena, so that rows fall back to the mask walk)
                ca.arena_per_xcd = (arena_env ? std::min<uint32_t>((uint32_t)atoi(arena_env), CL_ARENA) : CL_ARENA) / 8;
                ca.cursor = h->w_cl_cursor.as<uint32_t>();
                launch_compact_rows(ca, s);
                ra.cl_cnt = ca.cl_cnt;
                ra.cl_ent = ca.cl_ent;
                ra.cl_arena = ca.arena;
            }
            launch_replay(ra, s);
            h->timer.end(t, s);
        }
        sorted_any = sorted_any || sorted_now;
        if (sorted_now && eager_fix) {
            fix_due = true;  // (launch_due_fix: behind the next round's planning)
            fix_due_round = round;
        }
        if (dbg_replay_dev) {
            fprintf(stderr, "[replay] round %zu: %u queries\n", round, nact);
            print_replay_dbg(h, nact, s);
        }
    };
    // The reference's heap over what a round admitted (and the results of the flagged queries that finished in it), on a side
    // stream: it runs under the next round's scan and selection.  It starts behind the next round's PLANNING: those kernels are
    // single workgroups of 1024 threads on the search's critical path, and with the replay's workgroups already on every CU they
    // waited for room (round 1's planning: 0.10 ms against round 0's 0.06).
    auto launch_due_fix = [&]() {
        if (!fix_due) return;
        fix_due = false;
        const size_t round = fix_due_round;
        ensure_context_streams(h);
        TieFixArgs ta = tie_fix_args((uint32_t)round, 0);
        HIP_CHECK(hipEventRecord(h->ev_sel, s));
        HIP_CHECK(hipStreamWaitEvent(h->fix_stream, h->ev_sel, 0));
        size_t t = h->timer.begin(CAT_TIE_FIX, h->fix_stream);
        launch_tie_fix(ta, h->fix_stream);
        h->timer.end(t, h->fix_stream);
        HIP_CHECK(hipEventRecord(h->ev_fix[round & 1], h->fix_stream));
        fix_pending = true;
        last_fix_round = round;
    };
    auto next_round_len = [&](size_t round_len) { return base.fixed_two ? total_nprobe : std::min<size_t>(round_len * 2, 64); };
    // the planning counters (and, at the end, the per-round history) into their page-locked mirrors: one kernel, or blits
    auto fetch_counters = [&](size_t nhist, bool with_stats = false) {
        if (pinned_io(h)) {
            CopySegs c{};
            c.src[0] = h->w_pl_counters.p;
            c.dst[0] = h->p_counters.dev();
            c.words[0] = (uint32_t)CNT_WORDS;
            c.n = 1;
            if (nhist) {
                c.src[c.n] = h->w_pl_hist.p;
                c.dst[c.n] = h->p_hist.dev();
                c.words[c.n] = (uint32_t)(nhist * 16);
                c.n++;
            }
            if (with_stats) {
                c.src[c.n] = h->w_stats.p;
                c.dst[c.n] = static_cast<unsigned char*>(h->p_counters.dev()) + CNT_WORDS * 4;
                c.words[c.n] = (uint32_t)STAT_WORDS;
                c.n++;
            }
            launch_copy_segs(c, s);
        } else {
            HIP_CHECK(hipMemcpyAsync(hc, h->w_pl_counters.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, s));
            if (nhist) HIP_CHECK(hipMemcpyAsync(h->p_hist.p, h->w_pl_hist.p, nhist * 64, hipMemcpyDeviceToHost, s));
        }
    };
    auto plan_and_look = [&](size_t round_len) {
        plan_round(round_len);
        fetch_counters(0);
        HIP_CHECK(stream_sync(s));
    };
    // what is done with the counters of the last planned round once they are on the host (the end of the search, below)
    auto epilogue = [h, hc, chained, hints_used, bsig](size_t nh) {
        if (chained) {
            h->round_hint.assign(h->p_hist.as<uint32_t>(), h->p_hist.as<uint32_t>() + nh * 16);
            h->round_hint.insert(h->round_hint.end(), hc, hc + 16);  // the last planned round
            static const bool dbg_rounds = getenv("AUNCEL_AMD_DEBUG_ROUNDS") != nullptr;  // what every round of the search planned
            if (dbg_rounds)
                for (size_t r = 0; (r + 1) * 16 <= h->round_hint.size(); r++) {
                    const uint32_t* c = &h->round_hint[r * 16];
                    fprintf(stderr, "[rounds] %zu: queries %u segments %u pairs %u groups %u items qg1 %u qg2 %u qg4 %u qg8 %u\n", r, c[CNT_ACTIVE],
                            c[CNT_SEGMENTS], c[CNT_PAIRS], c[CNT_GROUPS], c[CNT_QG1], c[CNT_QG2], c[CNT_QG4], c[CNT_QG8]);
                }
            // a scan grid is its hint + 12 %; a round that needed more still covers its items (the workgroups stride over the
            // device-side count), only with fewer workgroups than it would have been given
            for (size_t r = 0; (r + 1) * 16 <= hints_used.size() && (r + 1) * 16 <= h->round_hint.size(); r++)
                for (int c : {CNT_QG1, CNT_QG2, CNT_QG4, CNT_QG8}) {
                    const uint32_t used = hints_used[r * 16 + c], need = h->round_hint[r * 16 + c];
                    if (!used) continue;
                    h->hinted_rounds++;
                    if (need > used + used / 8 + 8) h->short_rounds++;
                }
        }
        if (chained) {  // (the most row space a round of this shape has wanted so far)
            uint32_t want_mi = hc[12];
            for (size_t r = 0; r < nh; r++) want_mi = std::max(want_mi, h->p_hist.as<uint32_t>()[r * 16 + 12]);
            size_t want = (size_t)want_mi << 20;
            h->dist_want.raise(bsig, want);
            amd_ivf* owner = ix(h);
            std::lock_guard<std::mutex> lock(owner->want_mu);
            owner->shared_want.raise(bsig, want);
        }
        h->scan_bytes += *reinterpret_cast<double*>(hc + 16);
        h->scan_min_bytes += *reinterpret_cast<double*>(hc + 22);
        h->scan_min_bytes_thr += *reinterpret_cast<double*>(hc + 24 + PLAN_MAX_ROUNDS * 8);
        const unsigned long long* acc = reinterpret_cast<const unsigned long long*>(hc + 18);
        h->scan_slots += (double)acc[0];
        h->scan_useful += (double)acc[1];
    };
    static const bool no_skip = getenv("AUNCEL_AMD_NO_LAST_PLAN_SKIP") != nullptr;
    size_t round_len = first_round;

    if (chained) {
        // a fixed-nprobe search ends with its last planned round when no query can be deferred (rows and pairs fit the buffers)
        const size_t padded = (maxlist + 1023) & ~(size_t)1023;
        const bool fits = (double)n * (double)total_nprobe * (double)padded <= (double)budget && n * total_nprobe <= seg_cap;
        bool fixed_complete = !base.tuner.enabled && !base.train.enabled && fits && (base.fixed_two || first_round >= total_nprobe);
        static const size_t ahead_env = getenv("AUNCEL_AMD_ROUNDS_AHEAD") ? (size_t)atoi(getenv("AUNCEL_AMD_ROUNDS_AHEAD")) : 0;
        // (a handful of queries: three in four are done after the first round, and looking costs less than a round of empty launches)
        size_t batch = ahead_env ? ahead_env : base.train.enabled ? 4 : base.tuner.enabled ? (n < 20 ? 1 : 2) : (base.fixed_two ? 2 : 1);
        if (fixed_complete) batch = base.fixed_two && first_round < total_nprobe ? 2 : 1;
        bool planned = false;  // the round at hand has been planned (and looked at) already
        for (size_t round = 0;;) {
            for (size_t b = 0; b < batch; b++, round++) {
                const bool thr_mode = round > 0 && !no_thr;
                if (!planned) plan_round(round_len);
                planned = false;
                launch_due_fix();
                enqueue_scan(thr_mode, nullptr, round);
                // (alone on the index only: 3.50 -> 3.36 ms a lone batch; among five other searches the two extra launches and the
                // second pass over the boundary distances cost more than the wait they fill: 3.40 -> 3.30 M q/s)
                if (round == 0 && base.split_prepare && !thr_mode && active.before == 0) {
                    base.split_prepare();
                    enqueue_replay(thr_mode, (uint32_t)n, true, round, base.split_free, base.split_counts);
                    base.split_between();
                    enqueue_replay(thr_mode, base.split_wait_cap, true, round, base.split_wait, base.split_counts + 1);
                } else {
                    if (round == 0 && base.before_first_select) base.before_first_select();
                    enqueue_replay(thr_mode, (uint32_t)n, true, round);
                }
                round_len = next_round_len(round_len);
            }
            launch_due_fix();
            if (fixed_complete) break;
            // Is anything left?  The selection of the last round counted the queries it left unfinished, the planning of that
            // round the ones it deferred: one look at those two numbers, and a search that has ended (the common case after
            // two rounds) needs no further planning pass -- round 3 planned the next round first and looked at its count.
            static const bool plan_look = getenv("AUNCEL_AMD_PLAN_LOOK") != nullptr;
            if (!plan_look && round >= 1 && round - 1 < PLAN_MAX_ROUNDS) {
                // A call of a few queries ends here three times in four: the caller's read-backs (results, statistics, error word)
                // are queued with the look, and if nothing is left -- and no query needs its ties replayed -- this synchronisation
                // is the call's last (else what was queued is dropped and the search goes on as before).
                const bool speculate = base.spec_finish && pinned_io(h) && round == 1 && planned_rounds == 1 && !fix_pending && !eager_fix;
                const size_t small_mark = h->small.size(), af_mark = h->after_flush.size(), used_mark = h->small_used;
                fetch_counters(0, speculate);
                if (speculate) {
                    base.spec_finish();
                    launch_small_gathers(h, s);
                }
                host_stamp("enqueued");
                HIP_CHECK(stream_sync(s));
                host_stamp("look-sync");
                uint32_t unf = 0;
                for (int x = 0; x < 8; x++) unf += hc[24 + (round - 1) * 8 + x];
                const uint32_t left = unf + hc[11];
                if (speculate) {
                    const unsigned long long* rows = reinterpret_cast<const unsigned long long*>(hc + CNT_WORDS);
                    unsigned long long flagged = 0;
                    for (uint32_t r = 0; r < STATS_ROWS; r++) flagged += rows[4 * r + 3];
                    if (left == 0 && flagged == 0) {
                        flush_small(h);
                        std::vector<std::function<void()>> fs;
                        fs.swap(h->after_flush);
                        h->spec_done = true;  // (the caller skips its own read-backs and synchronisation)
                        for (auto& f : fs) f();
                        epilogue(0);
                        return;
                    }
                    h->small.resize(small_mark);
                    h->after_flush.resize(af_mark);
                    h->small_used = used_mark;
                }
                if (dbg_timing()) fprintf(stderr, "[rounds/chained] after round %zu: unfinished %u deferred %u\n", round, unf, hc[11]);
                if (left == 0) break;
                plan_round(round_len);
            } else {
                plan_and_look(round_len);
                if (dbg_timing())
                    fprintf(stderr, "[rounds/chained] after round %zu: active %u pairs %u items %u may-continue %u MiB %u\n", round, hc[CNT_ACTIVE],
                            hc[CNT_PAIRS], hc[CNT_QG1] + hc[CNT_QG2] + hc[CNT_QG4] + hc[CNT_QG8], hc[10], hc[7]);
                if (hc[CNT_ACTIVE] == 0) break;
            }
            planned = true;
            batch = ahead_env ? ahead_env : 2;
        }
    } else {
        // Round 0 starts from empty heaps (every distance is wanted: dense rows); later rounds run in threshold mode.
        for (size_t round = 0;; round++) {
            const bool thr_mode = base.range || (round > 0 && !no_thr);
            const double t0 = now_us();
            if (base.d_budget_ms) {  // the clock is read once the previous round has finished
                HIP_CHECK(stream_sync(s));
                pa.elapsed_ms = (float)((now_us() - base.t_start_us) * 1e-3);
            }
            plan_and_look(round_len);
            const double t1 = now_us();
            const uint32_t nact = hc[CNT_ACTIVE];
            if (nact == 0) break;
            enqueue_scan(thr_mode, hc, round);
            if (base.range) {
                // ---- range search: count, lay out, fill (the scan left masks of the entries inside the radius)
                RangeArgs ga{};
                ga.nq = nact;
                ga.nlist = (uint32_t)nlist;
                ga.qsel = h->w_qsel.as<uint32_t>();
                ga.seg_count = h->w_pl_cnt.as<uint32_t>();
                ga.seg_begin = h->w_seg_begin.as<uint32_t>();
                ga.seg_list = h->w_seg_list.as<int32_t>();
                ga.seg_off = h->w_seg_off.as<uint64_t>();
                ga.list_off = I->d_list_off.as<uint64_t>();
                ga.ids = I->d_ids.as<int64_t>();
                ga.dist = h->w_dist.as<float>();
                ga.mask = h->w_mask.as<unsigned long long>();
                h->w_rcount.ensure(n * 4);
                h->w_roff.ensure(n * 8);
                ga.counts = h->w_rcount.as<uint32_t>();
                ga.out_off = h->w_roff.as<unsigned long long>();
                ga.stage = h->w_stage.as<uint32_t>();
                ga.done = h->w_done.as<uint32_t>();
                ga.stats = h->w_stats.as<unsigned long long>() + 4 * (STATS_ROWS - 1);  // (ordinary atomics: the last row)
                ga.error = h->w_error.as<uint32_t>();
                size_t t = h->timer.begin(CAT_SELECT, s);
                launch_range_count(ga, s);
                std::vector<uint32_t> cnts(n), probes(n);
                HIP_CHECK(hipMemcpyAsync(cnts.data(), h->w_rcount.p, n * 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(probes.data(), h->w_pl_cnt.p, n * 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(stream_sync(s));
                check_device_error(h);
                // the queries of a round are a run of consecutive slots (everything unfinished before the budget cut), so
                // appending the rounds keeps the results in query order
                std::vector<unsigned long long> off(n, 0);
                size_t tot = 0;
                for (size_t i = 0; i < n; i++)
                    if (probes[i]) {
                        off[i] = tot;
                        tot += cnts[i];
                        h->r_lims[i + 1] = cnts[i];
                    }
                if (tot) {
                    h->w_rlab.ensure(tot * 8);
                    h->w_rdis.ensure(tot * 4);
                    ga.out_labels = h->w_rlab.as<int64_t>();
                    ga.out_dist = h->w_rdis.as<float>();
                    HIP_CHECK(hipMemcpyAsync(h->w_roff.p, off.data(), n * 8, hipMemcpyHostToDevice, s));
                    launch_range_fill(ga, s);
                    const size_t at = h->r_labels.size();
                    h->r_labels.resize(at + tot);
                    h->r_dist.resize(at + tot);
                    HIP_CHECK(hipMemcpyAsync(h->r_labels.data() + at, h->w_rlab.p, tot * 8, hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipMemcpyAsync(h->r_dist.data() + at, h->w_rdis.p, tot * 4, hipMemcpyDeviceToHost, s));
                }
                h->timer.end(t, s);
                HIP_CHECK(stream_sync(s));
                continue;
            }
            if (round == 0 && base.before_first_select) base.before_first_select();
            enqueue_replay(thr_mode, nact, false, round);
            launch_due_fix();
            if (dbg_timing())
                fprintf(stderr, "[round/dev] active %u pairs %u groups %u: plan+readback %.0f us, launches %.0f us\n", nact, hc[CNT_PAIRS],
                        hc[CNT_GROUPS], t1 - t0, now_us() - t1);
            // every query of this round ends with it and none was deferred: no further planning pass (and no read-back) is needed
            if (!no_skip && !base.train.enabled && hc[10] == 0) break;
            round_len = next_round_len(round_len);
        }
    }
    launch_due_fix();
    if (fix_pending) HIP_CHECK(hipStreamWaitEvent(s, h->ev_fix[last_fix_round & 1], 0));  // the last round's tie_fix_kernel
    if (sorted_any && !eager_fix) {
        size_t t = h->timer.begin(CAT_TIE_FIX, s);
        launch_tie_fix(tie_fix_args(0, 1), s);
        h->timer.end(t, s);
    }
    if (!base.caller_checks_error) check_device_error(h);
    // The bookkeeping counters of the last round (bytes, slots) have not been read yet; nor has the history.  What is done with
    // them needs a synchronisation: the caller's, when it is about to read its results back anyway (caller_checks_error: the
    // epilogue below then runs behind that one, sync_and_flush), else one of our own.
    const size_t nh = chained ? std::min(planned_rounds ? planned_rounds - 1 : 0, MAX_HIST) : 0;
    if (chained) fetch_counters(nh);
    if (chained && base.caller_checks_error) {
        h->after_flush.push_back([epilogue, nh]() { epilogue(nh); });
    } else {
        if (chained) HIP_CHECK(stream_sync(s));
        epilogue(nh);
    }
}

// multi-round driver shared by the adaptive search and the trace training (host-side planning: kept for reference /
// AUNCEL_AMD_HOST_PLAN=1)
void run_rounds(amd_ivf* h, RoundSpec& base, size_t n, size_t first_round, size_t total_nprobe,
                const unsigned long long* d_np_abs /* may be null */, size_t start) {
    // keys for probes [0, have) of every slot are kept on the host and extended on demand
    size_t have = 0;
    std::vector<int64_t> hkeys;
    size_t stride = 0;
    std::vector<uint32_t> stage(n, 0), done(n, 0);
    std::vector<unsigned long long> np(n, 0);
    size_t round_len = first_round;
    for (;;) {
        const double t_plan0 = now_us();
        // plan: every unfinished query runs either up to its known my_nprobe or one more block of round_len
        std::vector<uint32_t> slot, p0, cnt;
        size_t need_cols = 0;
        for (size_t i = 0; i < n; i++) {
            if (done[i]) continue;
            size_t target = stage[i] + round_len;
            if (base.tuner.enabled) {
                // an unfired query at stage s cannot stop before floor((s+1) * multipler): that many probes
                // are waste-free; beyond it allow growth-1 of over-scan to keep the number of rounds small
                static const double grow_env = getenv("AUNCEL_AMD_ROUND_GROW") ? atof(getenv("AUNCEL_AMD_ROUND_GROW")) : 3.5;
                const double grow = std::max<double>(base.tuner.multipler, grow_env);
                const size_t safe = (size_t)((float)(stage[i] + 1) * base.tuner.multipler);
                target = std::max<size_t>({safe, (size_t)(stage[i] * grow), stage[i] + first_round});
            }
            if (np[i] != 0) target = std::max<size_t>(np[i], stage[i] + 1);
            target = std::min(target, total_nprobe);
            if (target <= stage[i]) target = std::min<size_t>(stage[i] + 1, total_nprobe);
            slot.push_back((uint32_t)i);
            p0.push_back(stage[i]);
            cnt.push_back((uint32_t)(target - stage[i]));
            need_cols = std::max(need_cols, target);
        }
        if (slot.empty()) break;
        if (need_cols > have) {
            size_t new_have = std::min(total_nprobe, std::max(need_cols, have * 2));
            std::vector<int64_t> nk(n * new_have);
            HIP_CHECK(hipMemcpy2DAsync(nk.data(), new_have * 8, base.d_ckeys, (size_t)base.coarse_stride * 8, new_have * 8, n,
                                       hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(stream_sync(h->stream));
            hkeys.swap(nk);
            have = new_have;
            stride = new_have;
        }
        const double t_plan1 = now_us();
        RoundSpec r = base;
        r.slot = slot;
        r.p0 = p0;
        r.cnt = cnt;
        r.keys = hkeys.data();
        r.key_stride = stride;
        r.total_nprobe = (uint32_t)total_nprobe;
        exec_round(h, r);
        const double t_rb0 = now_us();
        check_device_error(h);
        HIP_CHECK(hipMemcpyAsync(stage.data(), h->w_stage.p, n * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(done.data(), h->w_done.p, n * 4, hipMemcpyDeviceToHost, h->stream));
        if (d_np_abs)
            HIP_CHECK(hipMemcpyAsync(np.data(), d_np_abs + start, n * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(stream_sync(h->stream));
        if (dbg_timing()) fprintf(stderr, "[rounds] plan+keys %.0f us, readback %.0f us\n", t_plan1 - t_plan0, now_us() - t_rb0);
        round_len = std::min<size_t>(round_len * 2, 64);
    }
}

}  // namespace

// ============================================================================================
// extern "C" boundary
// ============================================================================================
#define API_BEGIN try {
#define API_END                                  \
    return 0;                                    \
    }                                            \
    catch (const EngineError& e) {               \
        g_last_error = e.what();                 \
        return -2;                               \
    }                                            \
    catch (const std::exception& e) {            \
        g_last_error = e.what();                 \
        return -4;                               \
    }                                            \
    catch (...) {                                \
        g_last_error = "unknown error";          \
        return -1;                               \
    }

namespace amdivf {
void set_last_error(const std::string& msg) { g_last_error = msg; }  // (dataset_io.cpp)
}

extern "C" {

const char* amd_ivf_last_error(void) { return g_last_error.c_str(); }

int amd_ivf_device_count(int* count) {
    API_BEGIN
    HIP_CHECK(hipGetDeviceCount(count));
    API_END
}

// The per-XCD counters under contention (ivf_plan.hip: xcd_check_kernel).  out[0] adds made, out[1] sum of the counters, out[2]
// (XCD, address) pairs whose returned values are not exactly 0 .. count - 1, out[3] mask of the XCD numbers seen.
int amd_ivf_self_check(int device, uint64_t out[4]) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    const uint32_t n = 1u << 20, nkeys = 1024;
    DevBuf c, sl, xo;
    c.ensure((size_t)8 * nkeys * 4);
    sl.ensure((size_t)n * 4);
    xo.ensure((size_t)n * 4);
    HIP_CHECK(hipMemset(c.p, 0, (size_t)8 * nkeys * 4));
    launch_xcd_check(c.as<uint32_t>(), nkeys, sl.as<uint32_t>(), xo.as<uint32_t>(), n, nullptr);
    std::vector<uint32_t> hs(n), hx(n), hc((size_t)8 * nkeys);
    HIP_CHECK(hipMemcpy(hs.data(), sl.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hx.data(), xo.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hc.data(), c.p, hc.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::vector<uint32_t>> got((size_t)8 * nkeys);
    uint64_t mask = 0, bad = 0, total = 0;
    for (uint32_t i = 0; i < n; i++) {
        mask |= 1ull << hx[i];
        if (hx[i] >= 8) {
            bad++;
            continue;
        }
        got[(size_t)hx[i] * nkeys + (i * 2654435761u) % nkeys].push_back(hs[i]);
    }
    for (size_t k = 0; k < got.size(); k++) {
        auto& v = got[k];
        std::sort(v.begin(), v.end());
        total += hc[k];
        bool ok = v.size() == hc[k];
        for (size_t j = 0; ok && j < v.size(); j++) ok = v[j] == j;
        bad += ok ? 0 : 1;
    }
    out[0] = n;
    out[1] = total;
    out[2] = bad;
    out[3] = mask;
    API_END
}

int amd_ivf_create(int d, size_t nlist, int metric, int device, amd_ivf_t** out) {
    API_BEGIN
    if (d <= 0 || nlist == 0) throw EngineError("bad dimension / nlist");
    if (metric != METRIC_L2 && metric != METRIC_IP) throw EngineError("unsupported metric");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw std::runtime_error("no such HIP device (this engine has no CPU path)");
    std::unique_ptr<amd_ivf> h(new amd_ivf);
    h->d = d;
    h->dpad = (d + 3) & ~3;
    h->nlist = nlist;
    h->metric = metric;
    h->device = device;
    HIP_CHECK(hipSetDevice(device));
    {
        // once per device and process: the per-XCD counters the searches end on (xcd_local_add) behave as assumed -- 8 XCDs at most,
        // adds exact under contention -- or the engine refuses to run rather than end a search early
        static std::mutex mu;
        static std::map<int, bool> checked;
        std::lock_guard<std::mutex> lock(mu);
        if (!checked.count(device)) {
            uint64_t r[4] = {0, 0, 0, 0};
            if (amd_ivf_self_check(device, r) != 0) throw std::runtime_error(std::string("self-check of the per-XCD counters: ") + g_last_error);
            checked[device] = r[1] == r[0] && r[2] == 0 && (r[3] >> 8) == 0;
        }
        if (!checked[device]) throw std::runtime_error("per-XCD counters are not exact on this device (amd_ivf_self_check): the engine's round planning relies on them");
    }
    h->stream = make_main_stream();
    ensure_context_streams(h.get());
    h->h_codes.resize(nlist);
    h->h_ids.resize(nlist);
    h->h_list_off.assign(nlist + 1, 0);
    if (getenv("AUNCEL_AMD_NO_FUSED")) h->allow_fused = 0;
    if (getenv("AUNCEL_AMD_NO_BYTES")) h->allow_bytes = 0;
    *out = h.release();
    API_END
}

// Entry points that change or read back index data are for the owning handle only.
#define OWNER_ONLY(h) \
    if ((h)->is_clone) throw EngineError("not available on a search context made by amd_ivf_clone: use the owning handle")

int amd_ivf_clone(amd_ivf_t* h, amd_ivf_t** out) {
    API_BEGIN
    amd_ivf* owner = ix(h);
    use_device(owner);
    upload_lists(owner);
    std::unique_ptr<amd_ivf> c(new amd_ivf);
    c->parent = owner;
    c->is_clone = true;
    c->d = owner->d;
    c->dpad = owner->dpad;
    c->nlist = owner->nlist;
    c->metric = owner->metric;
    c->device = owner->device;
    c->dist_budget_floats = owner->dist_budget_floats;
    c->allow_fused = owner->allow_fused;
    c->allow_bytes = owner->allow_bytes;
    c->stream = make_main_stream();
    ensure_context_streams(c.get());
    owner->live_contexts.fetch_add(1);
    *out = c.release();
    API_END
}

int amd_ivf_destroy(amd_ivf_t* h) {
    API_BEGIN
    if (h) {
        use_device(h);
        if (h->is_clone && h->parent) h->parent->live_contexts.fetch_sub(1);
        delete h;
    }
    API_END
}

int amd_ivf_set_centroids(amd_ivf_t* h, const float* centroids) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    h->h_centroids.assign(h->nlist * h->dpad, 0.f);
    for (size_t i = 0; i < h->nlist; i++) memcpy(&h->h_centroids[i * h->dpad], centroids + i * h->d, h->d * sizeof(float));
    h->d_centroids.ensure(h->nlist * h->dpad * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(h->d_centroids.p, h->h_centroids.data(), h->nlist * h->dpad * sizeof(float),
                             hipMemcpyHostToDevice, h->stream));
    h->d_centroid_norms.ensure(h->nlist * sizeof(float));
    launch_row_norms(h->d_centroids.as<float>(), h->nlist, h->dpad, h->d_centroid_norms.as<float>(), h->stream);
    h->d_cinfo.ensure(16);
    HIP_CHECK(hipMemsetAsync(h->d_cinfo.p, 0, 16, h->stream));
    launch_amax(h->d_centroids.as<float>(), h->nlist, h->dpad, h->d_cinfo.as<uint32_t>(), h->stream);
    HIP_CHECK(stream_sync(h->stream));
    h->have_centroids = true;
    h->have_interdis = false;
    {
        double mx = 0;  // max |c|^2: the error bound of the approximate coarse ranking (coarse_pick_kernel)
        for (size_t i = 0; i < h->nlist; i++) {
            double sq = 0;
            for (int c = 0; c < h->d; c++) sq += (double)h->h_centroids[i * h->dpad + c] * (double)h->h_centroids[i * h->dpad + c];
            mx = std::max(mx, sq);
        }
        h->centroid_norm_max = (float)(mx * (1.0 + 1e-6));
    }
    h->centroid_range = IntRange();
    h->centroid_range.add(centroids, h->nlist * (size_t)h->d);
    API_END
}

int amd_ivf_set_lists(amd_ivf_t* h, const size_t* sizes, const float* const* codes, const int64_t* const* ids) {
    API_BEGIN
    OWNER_ONLY(h);
    size_t nt = 0;
    h->db_range = IntRange();
    for (size_t l = 0; l < h->nlist; l++) {
        size_t n = sizes[l];
        h->db_range.add(codes[l], n * (size_t)h->d);
        h->h_codes[l].assign(n * h->dpad, 0.f);
        for (size_t j = 0; j < n; j++) memcpy(&h->h_codes[l][j * h->dpad], codes[l] + j * h->d, h->d * sizeof(float));
        h->h_ids[l].assign(ids[l], ids[l] + n);
        nt += n;
    }
    h->ntotal = nt;
    h->lists_dirty = true;
    upload_lists(h);
    API_END
}

int amd_ivf_add(amd_ivf_t* h, size_t n, const float* x, const int64_t* xids, const int64_t* precomputed_idx) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    std::vector<int64_t> assign;
    const int64_t* idx = precomputed_idx;
    if (!idx) {
        // quantizer->assign(n, x, idx): nearest centroid with the exact kernel, in blocks
        assign.resize(n);
        const size_t bs = 1 << 18;
        h->w_x.ensure(std::min(bs, n) * h->dpad * sizeof(float));
        h->w_cdis.ensure(std::min(bs, n) * 4);
        h->w_ckeys.ensure(std::min(bs, n) * 8);
        for (size_t i0 = 0; i0 < n; i0 += bs) {
            size_t m = std::min(bs, n - i0);
            upload_rows(h, h->w_x.as<float>(), x + i0 * h->d, m);
            coarse_dev(h, h->w_x.as<float>(), m, 1, 0, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(), 0);
            HIP_CHECK(hipMemcpyAsync(assign.data() + i0, h->w_ckeys.p, m * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(stream_sync(h->stream));
        }
        idx = assign.data();
        double ms[NCAT], ln[NCAT];
        h->timer.collect(ms, NCAT, ln);
    }
    for (size_t i = 0; i < n; i++) {
        int64_t id = xids ? xids[i] : (int64_t)(h->ntotal + i);
        int64_t l = idx[i];
        if (l < 0) continue;
        if ((size_t)l >= h->nlist) throw EngineError("Invalid list number in add");
        std::vector<float>& c = h->h_codes[l];
        size_t o = c.size();
        c.resize(o + h->dpad, 0.f);
        memcpy(&c[o], x + i * h->d, h->d * sizeof(float));
        h->h_ids[l].push_back(id);
    }
    h->db_range.add(x, n * (size_t)h->d);
    h->ntotal += n;
    h->lists_dirty = true;
    API_END
}

int amd_ivf_ntotal(const amd_ivf_t* h, size_t* ntotal) {
    *ntotal = ix(h)->ntotal;
    return 0;
}

int amd_ivf_list_size(const amd_ivf_t* h, size_t list_no, size_t* size) {
    API_BEGIN
    if (list_no >= h->nlist) throw EngineError("Invalid list number");
    *size = ix(h)->h_ids[list_no].size();
    API_END
}

int amd_ivf_get_list(const amd_ivf_t* hc, size_t list_no, float* codes, int64_t* ids) {
    API_BEGIN
    OWNER_ONLY(hc);
    amd_ivf* h = const_cast<amd_ivf*>(hc);
    if (list_no >= h->nlist) throw EngineError("Invalid list number");
    upload_lists(h);
    size_t n = h->h_list_off[list_no + 1] - h->h_list_off[list_no];
    if (n) {
        // read back from HBM (the device copy is the one the kernels scan)
        HIP_CHECK(hipMemcpy2DAsync(codes, h->d * sizeof(float), h->d_codes.as<float>() + h->h_list_off[list_no] * h->dpad,
                                   h->dpad * sizeof(float), h->d * sizeof(float), n, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(ids, h->d_ids.as<int64_t>() + h->h_list_off[list_no], n * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(stream_sync(h->stream));
    }
    API_END
}

int amd_ivf_coarse(amd_ivf_t* h, size_t n, const float* x, size_t nprobe, float* coarse_dis, int64_t* keys, int mode) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (n == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    h->w_cdis.ensure(n * nprobe * 4);
    h->w_ckeys.ensure(n * nprobe * 8);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    coarse_dev(h, h->w_x.as<float>(), n, nprobe, mode, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(),
               h->allow_fused && h->centroid_range.fusable_with(qr, h->metric));
    HIP_CHECK(hipMemcpyAsync(coarse_dis, h->w_cdis.p, n * nprobe * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(keys, h->w_ckeys.p, n * nprobe * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_coarse_resident(amd_ivf_t* h, size_t start, size_t n, size_t nprobe, float* coarse_dis, int64_t* keys, int mode) {
    API_BEGIN
    use_device(h);
    // (a search context without resident queries of its own ranks its owner's: amd_ivf_submit_coarse_resident's contexts)
    const amd_ivf* src = h->is_clone && h->n_resident == 0 && h->parent ? h->parent : h;
    if (start + n > src->n_resident) throw EngineError("resident query range out of bounds");
    if (n == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_cdis.ensure(n * nprobe * 4);
    h->w_ckeys.ensure(n * nprobe * 8);
    coarse_dev(h, src->d_resident.as<float>() + start * h->dpad, n, nprobe, mode, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(),
               h->allow_fused && ix(h)->centroid_range.fusable_with(src->resident_range, h->metric));
    if (coarse_dis) HIP_CHECK(hipMemcpyAsync(coarse_dis, h->w_cdis.p, n * nprobe * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(keys, h->w_ckeys.p, n * nprobe * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_search_preassigned(amd_ivf_t* h, size_t n, const float* x, size_t k, size_t nprobe, const int64_t* keys,
                               const float* coarse_dis, float* D, int64_t* I, int store_pairs, size_t max_codes) {
    API_BEGIN
    (void)coarse_dis;  // IVF-Flat codes are not residuals: the scanner ignores it (IndexIVFFlat.cpp:106)
    use_device(h);
    if (n == 0 || k == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    static const bool host_plan = getenv("AUNCEL_AMD_HOST_PLAN") != nullptr;
    if (host_plan) {
        search_fixed_core(h, h->w_x.as<float>(), n, k, nprobe, keys, D, I, store_pairs, max_codes, qr);
    } else {
        h->w_ckeys.ensure(n * nprobe * 8);
        HIP_CHECK(hipMemcpyAsync(h->w_ckeys.p, keys, n * nprobe * 8, hipMemcpyHostToDevice, h->stream));
        search_fixed_device(h, h->w_x.as<float>(), n, k, nprobe, h->w_ckeys.as<int64_t>(), D, I, store_pairs, max_codes, qr);
    }
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_search(amd_ivf_t* h, size_t n, const float* x, size_t k, size_t nprobe, int coarse_mode, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    if (n == 0 || k == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    search_full(h, h->w_x.as<float>(), n, k, nprobe, coarse_mode, D, I, qr);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_set_queries(amd_ivf_t* h, size_t n, const float* x) {
    API_BEGIN
    use_device(h);
    h->d_resident.ensure(std::max<size_t>(n, 1) * h->dpad * sizeof(float));
    upload_rows(h, h->d_resident.as<float>(), x, n);
    HIP_CHECK(stream_sync(h->stream));
    h->n_resident = n;
    h->resident_gen++;
    h->resident_range = IntRange();
    h->resident_range.add(x, n * (size_t)h->d);
    API_END
}

// IndexIVF::range_search_preassigned (IndexIVF.cpp:759-857): device-planned rounds in threshold mode, radius as threshold
static void range_core(amd_ivf* h, const float* d_x, size_t n, float radius, size_t nprobe, const int64_t* d_keys, const IntRange& qr,
                       size_t* lims) {
    upload_lists(h);
    init_state(h, n, 1, false);
    launch_fill_f32(h->w_thr.as<float>(), n, radius, h->stream);
    h->r_lims.assign(n + 1, 0);
    h->r_labels.clear();
    h->r_dist.clear();
    RoundSpec base;
    base.range = true;
    base.k = 1;
    base.d_x = d_x;
    base.d_ckeys = d_keys;
    base.coarse_stride = (uint32_t)nprobe;
    base.fused = h->allow_fused && ix(h)->db_range.fusable_with(qr, h->metric);
    base.bytes = byte_queries(h, ix(h), d_x, n, qr);
    ix(h)->last_arith = base.bytes ? 2 : base.fused ? 1 : 0;
    run_rounds_device(h, base, n, nprobe, nprobe, nullptr);
    for (size_t i = 0; i < n; i++) h->r_lims[i + 1] += h->r_lims[i];
    memcpy(lims, h->r_lims.data(), (n + 1) * sizeof(size_t));
    fold_stats(h, n);
}

int amd_ivf_range_search_preassigned(amd_ivf_t* h, size_t n, const float* x, float radius, size_t nprobe, const int64_t* keys,
                                     size_t* lims) {
    API_BEGIN
    use_device(h);
    if (nprobe == 0) throw EngineError("nprobe must be positive");
    WallClock wc(h->stream);
    h->w_x.ensure(std::max<size_t>(n, 1) * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    h->w_ckeys.ensure(std::max<size_t>(n, 1) * nprobe * 8);
    if (n) HIP_CHECK(hipMemcpyAsync(h->w_ckeys.p, keys, n * nprobe * 8, hipMemcpyHostToDevice, h->stream));
    if (n == 0) {
        h->r_lims.assign(1, 0);
        h->r_labels.clear();
        h->r_dist.clear();
        lims[0] = 0;
        return 0;
    }
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    range_core(h, h->w_x.as<float>(), n, radius, nprobe, h->w_ckeys.as<int64_t>(), qr, lims);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_range_search(amd_ivf_t* h, size_t n, const float* x, float radius, size_t nprobe, int coarse_mode, size_t* lims) {
    API_BEGIN
    use_device(h);
    if (nprobe == 0) throw EngineError("nprobe must be positive");
    WallClock wc(h->stream);
    if (n == 0) {
        h->r_lims.assign(1, 0);
        h->r_labels.clear();
        h->r_dist.clear();
        lims[0] = 0;
        return 0;
    }
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    upload_lists(h);
    h->w_cdis.ensure(n * nprobe * 4);
    h->w_ckeys.ensure(n * nprobe * 8);
    coarse_dev(h, h->w_x.as<float>(), n, nprobe, coarse_mode, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(),
               h->allow_fused && ix(h)->centroid_range.fusable_with(qr, h->metric));
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    range_core(h, h->w_x.as<float>(), n, radius, nprobe, h->w_ckeys.as<int64_t>(), qr, lims);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_range_results(amd_ivf_t* h, int64_t* labels, float* distances) {
    API_BEGIN
    if (!h->r_labels.empty()) {
        memcpy(labels, h->r_labels.data(), h->r_labels.size() * sizeof(int64_t));
        memcpy(distances, h->r_dist.data(), h->r_dist.size() * sizeof(float));
    }
    API_END
}

int amd_ivf_search_resident_preassigned(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const int64_t* keys, float* D,
                                        int64_t* I) {
    API_BEGIN
    use_device(h);
    const amd_ivf* src = h->is_clone && h->n_resident == 0 && h->parent ? h->parent : h;  // (as amd_ivf_search_resident)
    if (start + n > src->n_resident) throw EngineError("resident query range out of bounds");
    if (n == 0 || k == 0) return 0;
    if (!keys) throw EngineError("keys are required");
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_ckeys.ensure(n * nprobe * 8);
    HIP_CHECK(hipMemcpyAsync(h->w_ckeys.p, keys, n * nprobe * 8, hipMemcpyHostToDevice, h->stream));
    search_fixed_device(h, src->d_resident.as<float>() + start * h->dpad, n, k, nprobe, h->w_ckeys.as<int64_t>(), D, I, 0, 0, src->resident_range);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_search_resident(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, int coarse_mode, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    // (a search context without resident queries of its own searches its owner's: amd_ivf_submit_search_resident's contexts)
    const amd_ivf* src = h->is_clone && h->n_resident == 0 && h->parent ? h->parent : h;
    if (start + n > src->n_resident) throw EngineError("resident query range out of bounds");
    if (n == 0 || k == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    search_full(h, src->d_resident.as<float>() + start * h->dpad, n, k, nprobe, coarse_mode, D, I, src->resident_range);
    finish_timing(h, wc.stop());
    API_END
}

// Error_sys::time_search (profile.cpp:229-244): IndexIVF::search with tune off and t->time_tune on -- the plain probe loop
// over nprobe = nlist probes, left when the time budget of the query is used up (IndexIVF.cpp:504-506,545-549).
static void timed_core(amd_ivf* h, const float* d_x, size_t start, size_t n, size_t k, size_t nprobe, const float* budget_ms, int coarse_mode,
                       uint64_t* nprobe_used, float* D, int64_t* I, const IntRange& qr) {
    const double t_start = now_us();
    nprobe = std::min<size_t>(nprobe, h->nlist);
    upload_lists(h);
    DevBuf& d_b = h->w_misc2;
    d_b.ensure((start + n) * 4);
    HIP_CHECK(hipMemcpyAsync(d_b.p, budget_ms, (start + n) * 4, hipMemcpyHostToDevice, h->stream));
    h->w_cdis.ensure(n * nprobe * 4);
    h->w_ckeys.ensure(n * nprobe * 8);
    // Runs of equal coarse distances stay in centroid-number order here unless AUNCEL_AMD_COARSE_TIES=heap: where the
    // clock decides how deep a query goes, which of two equidistant lists comes first is immaterial, and re-running the
    // reference's heap over all nlist entries (2.8 ms at 4096) would cost more than most budgets.
    h->ties_override = (int)opt(h, OPT_COARSE_TIES, -1) == 1 ? 1 : 0;
    coarse_dev(h, d_x, n, nprobe, coarse_mode, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(),
               h->allow_fused && ix(h)->centroid_range.fusable_with(qr, h->metric));
    h->ties_override = -1;
    init_state(h, n, k, false);
    RoundSpec base;
    base.k = (int)k;
    base.id_offset = start;
    base.d_x = d_x;
    base.d_ckeys = h->w_ckeys.as<int64_t>();
    base.coarse_stride = (uint32_t)nprobe;
    base.fused = h->allow_fused && ix(h)->db_range.fusable_with(qr, h->metric);
    base.bytes = byte_queries(h, ix(h), d_x, n, qr);
    ix(h)->last_arith = base.bytes ? 2 : base.fused ? 1 : 0;
    base.d_budget_ms = d_b.as<float>();
    base.t_start_us = t_start;
    static const size_t first_env = getenv("AUNCEL_AMD_TIMED_FIRST") ? (size_t)atoi(getenv("AUNCEL_AMD_TIMED_FIRST")) : 4;
    base.caller_checks_error = true;
    std::vector<uint32_t> stage(nprobe_used ? n : 0);
    with_select_fallback(h, [&] {
        if (h->force_heap_select) init_state(h, n, k, false);
        run_rounds_device(h, base, n, std::max<size_t>(1, first_env), nprobe, nullptr);
        finish_results(h, n, k, D, I, nprobe_used ? stage.data() : nullptr);
    });
    for (size_t i = 0; i < stage.size(); i++) nprobe_used[i] = stage[i];
}

int amd_ivf_search_timed(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const float* budget_ms, int coarse_mode,
                         uint64_t* nprobe_used, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    if (n == 0 || k == 0 || nprobe == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    timed_core(h, h->d_resident.as<float>() + start * h->dpad, start, n, k, nprobe, budget_ms, coarse_mode, nprobe_used, D, I,
               h->resident_range);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_search_timed_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t k, size_t nprobe, const float* budget_ms,
                           int coarse_mode, uint64_t* nprobe_used, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    if (n == 0 || k == 0) return 0;
    WallClock wc(h->stream);
    h->scan_bytes = h->scan_min_bytes = h->scan_min_bytes_thr = 0;
    h->scan_slots = h->scan_useful = 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    timed_core(h, h->w_x.as<float>(), id_offset, n, k, nprobe, budget_ms, coarse_mode, nprobe_used, D, I, qr);
    finish_timing(h, wc.stop());
    API_END
}

int amd_ivf_scan_codes(amd_ivf_t* h, const float* query, size_t list_no, int store_pairs, size_t k, float* simi,
                       int64_t* idxi, size_t* nup) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (list_no >= h->nlist) throw EngineError("Invalid key");
    upload_lists(h);
    h->w_x.ensure(h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), query, 1);
    init_state(h, 1, k, false);
    // import the caller's heap as the starting state
    HIP_CHECK(hipMemcpyAsync(h->w_heap_val.p, simi, k * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->w_heap_ref.p, idxi, k * 8, hipMemcpyHostToDevice, h->stream));
    int64_t key = (int64_t)list_no;
    RoundSpec r;
    r.slot = {0};
    r.p0 = {0};
    r.cnt = {1};
    r.keys = &key;
    r.key_stride = 1;
    r.k = (int)k;
    r.store_pairs = store_pairs;
    r.finalize_all = 1;
    r.raw_heap_out = 1;
    r.d_x = h->w_x.as<float>();
    {
        IntRange qr;
        qr.add(query, (size_t)h->d);
        r.fused = h->allow_fused && h->db_range.fusable_with(qr, h->metric);
    }
    exec_round(h, r);
    check_device_error(h);
    HIP_CHECK(hipMemcpyAsync(simi, h->w_D.p, k * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(idxi, h->w_I.p, k * 8, hipMemcpyDeviceToHost, h->stream));
    unsigned long long rows[4 * STATS_ROWS], st[4];
    HIP_CHECK(hipMemcpyAsync(rows, h->w_stats.p, sizeof rows, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    sum_stat_rows(rows, st);
    if (nup) *nup = st[2];
    double ms[NCAT], ln[NCAT];
    h->timer.collect(ms, NCAT, ln);
    API_END
}

int amd_ivf_distance_to_code(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, float* dis) {
    API_BEGIN
    use_device(h);
    if (list_no >= h->nlist) throw EngineError("Invalid key");
    upload_lists(h);
    const std::vector<uint64_t>& loff = ix(h)->h_list_off;
    if (offset >= loff[list_no + 1] - loff[list_no]) throw EngineError("offset beyond list size");
    h->w_x.ensure(h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), query, 1);
    h->w_dist.ensure(4);
    h->w_items.ensure(sizeof(ScanItem));
    h->w_pair_query.ensure(4);
    h->w_pair_out.ensure(8);
    ScanItem it{loff[list_no] + offset, 1, 0, 0, 1, 1, 0};  // qgroup 0
    uint32_t pq = 0;
    uint64_t po = 0;
    HIP_CHECK(hipMemcpyAsync(h->w_items.p, &it, sizeof(it), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->w_pair_query.p, &pq, 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->w_pair_out.p, &po, 8, hipMemcpyHostToDevice, h->stream));
    pack_query_tiles(h, h->w_x.as<float>(), {{0u, 1u}});
    ScanArgs sa{ix(h)->d_codes.as<float>(), h->w_x.as<float>(), h->w_items.as<ScanItem>(), h->w_pair_query.as<uint32_t>(),
                h->w_pair_out.as<uint64_t>(), h->w_dist.as<float>(), h->w_qtile.as<float>(), h->dpad, h->metric, 0};
    const size_t one_qg1[4] = {1, 0, 0, 0};
    launch_scan(sa, one_qg1, h->stream);
    HIP_CHECK(hipMemcpyAsync(dis, h->w_dist.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    API_END
}

// one query against vectors [offset, offset + n) of a list: the round of exec_round over that part
static RoundSpec list_part_round(amd_ivf* h, const float* query, size_t list_no, size_t offset, size_t n, const int64_t* key0) {
    if (list_no >= h->nlist) throw EngineError("Invalid key");
    upload_lists(h);
    const std::vector<uint64_t>& off = ix(h)->h_list_off;
    if (offset + n > off[list_no + 1] - off[list_no]) throw EngineError("codes beyond the end of the list");
    h->w_x.ensure(h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), query, 1);
    RoundSpec r;
    r.slot = {0};
    r.p0 = {0};
    r.cnt = {1};
    r.keys = key0;
    r.key_stride = 1;
    r.finalize_all = 1;
    r.d_x = h->w_x.as<float>();
    r.sub_base = off[list_no] + offset;
    r.sub_n = n;
    IntRange qr;
    qr.add(query, (size_t)h->d);
    r.fused = h->allow_fused && ix(h)->db_range.fusable_with(qr, h->metric);
    return r;
}

int amd_ivf_scan_codes_at(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, size_t n, int store_pairs, size_t k, float* simi,
                          int64_t* idxi, size_t* nup) {
    API_BEGIN
    use_device(h);
    if (nup) *nup = 0;
    const int64_t key0 = 0;
    RoundSpec r = list_part_round(h, query, list_no, offset, n, &key0);
    if (n == 0 || k == 0) return 0;
    init_state(h, 1, k, false);
    HIP_CHECK(hipMemcpyAsync(h->w_heap_val.p, simi, k * 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipMemcpyAsync(h->w_heap_ref.p, idxi, k * 8, hipMemcpyHostToDevice, h->stream));
    r.k = (int)k;
    r.store_pairs = store_pairs;
    r.pair_list = (long long)list_no;
    r.raw_heap_out = 1;
    exec_round(h, r);
    check_device_error(h);
    HIP_CHECK(hipMemcpyAsync(simi, h->w_D.p, k * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(idxi, h->w_I.p, k * 8, hipMemcpyDeviceToHost, h->stream));
    unsigned long long rows[4 * STATS_ROWS], st[4];
    HIP_CHECK(hipMemcpyAsync(rows, h->w_stats.p, sizeof rows, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    sum_stat_rows(rows, st);
    if (nup) *nup = st[2];
    double ms[NCAT], ln[NCAT];
    h->timer.collect(ms, NCAT, ln);
    API_END
}

int amd_ivf_scan_codes_range(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, size_t n, float radius, size_t* count) {
    API_BEGIN
    use_device(h);
    *count = 0;
    h->r_part_pos.clear();
    h->r_part_dis.clear();
    const int64_t key0 = 0;
    RoundSpec r = list_part_round(h, query, list_no, offset, n, &key0);
    if (n == 0) return 0;
    init_state(h, 1, 1, false);
    r.k = 1;
    r.collect = true;
    r.collect_radius = radius;
    exec_round(h, r);
    uint32_t cnt = 0;
    HIP_CHECK(hipMemcpyAsync(&cnt, h->w_sub_off.as<uint32_t>() + 4, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    if (cnt) {
        h->r_part_pos.resize(cnt);
        h->r_part_dis.resize(cnt);
        HIP_CHECK(hipMemcpyAsync(h->r_part_pos.data(), h->w_I.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(h->r_part_dis.data(), h->w_D.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(stream_sync(h->stream));
    }
    *count = cnt;
    double ms[NCAT], ln[NCAT];
    h->timer.collect(ms, NCAT, ln);
    API_END
}

int amd_ivf_scan_codes_range_results(amd_ivf_t* h, uint32_t* positions, float* distances) {
    API_BEGIN
    if (!h->r_part_pos.empty()) {
        memcpy(positions, h->r_part_pos.data(), h->r_part_pos.size() * sizeof(uint32_t));
        memcpy(distances, h->r_part_dis.data(), h->r_part_dis.size() * sizeof(float));
    }
    API_END
}

int amd_ivf_stats(amd_ivf_t* h, size_t stats[4], int reset) {
    for (int i = 0; i < 4; i++) stats[i] = h->stats_host[i];
    if (reset)
        for (int i = 0; i < 4; i++) h->stats_host[i] = 0;
    for (amd_ivf* c : h->async_ctx) {  // searches submitted with amd_ivf_submit_* count on their owner (call with none in flight)
        for (int i = 0; i < 4; i++) stats[i] += c->stats_host[i];
        if (reset)
            for (int i = 0; i < 4; i++) c->stats_host[i] = 0;
    }
    return 0;
}

int amd_ivf_set_interdis(amd_ivf_t* h, const float* table) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    const size_t nl = h->nlist, sz = nl * (nl - 1) / 2;
    h->d_interdis.ensure(std::max<size_t>(sz, 1) * 4);
    if (table) {
        HIP_CHECK(hipMemcpyAsync(h->d_interdis.p, table, sz * 4, hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(stream_sync(h->stream));
    } else {
        if (!h->have_centroids) throw EngineError("quantizer has no centroids");
        // all-pairs centroid distances with the scan kernel, then pack the upper triangle
        std::vector<float> cq(h->h_centroids);
        if (h->metric == METRIC_IP) {
            // reference quirk (IndexIVF.cpp:102-107): centroid 0 is renormalised nlist times, the others never
            for (size_t i = 0; i < nl; i++) {
                float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                for (int c = 0; c < h->dpad; c += 4) {
                    s0 += cq[c] * cq[c];
                    s1 += cq[c + 1] * cq[c + 1];
                    s2 += cq[c + 2] * cq[c + 2];
                    s3 += cq[c + 3] * cq[c + 3];
                }
                float norm = sqrtf((s0 + s1) + (s2 + s3));
                for (int c = 0; c < h->d; c++) cq[c] /= norm;
            }
        }
        DevBuf d_cq, d_full;
        d_cq.ensure(nl * h->dpad * 4);
        d_full.ensure(nl * nl * 4);
        HIP_CHECK(hipMemcpyAsync(d_cq.p, cq.data(), nl * h->dpad * 4, hipMemcpyHostToDevice, h->stream));
        const uint32_t qg = 4, tv = SCAN_WAVE_VECS;
        std::vector<ScanItem> items;
        std::vector<uint32_t> pq(nl);
        std::vector<uint64_t> po(nl);
        for (size_t i = 0; i < nl; i++) {
            pq[i] = (uint32_t)i;
            po[i] = (uint64_t)i * nl;
        }
        for (uint32_t vb = 0; vb < nl; vb += tv)
            for (uint32_t qb = 0; qb < nl; qb += qg * SCAN_RQ)
                items.push_back(ScanItem{vb, std::min<uint32_t>(tv, (uint32_t)nl - vb), vb, qb,
                                         std::min<uint32_t>(qg * SCAN_RQ, (uint32_t)nl - qb), qg, qb / SCAN_RQ});
        h->w_items.ensure(items.size() * sizeof(ScanItem));
        h->w_pair_query.ensure(nl * 4);
        h->w_pair_out.ensure(nl * 8);
        HIP_CHECK(hipMemcpyAsync(h->w_items.p, items.data(), items.size() * sizeof(ScanItem), hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipMemcpyAsync(h->w_pair_query.p, pq.data(), nl * 4, hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipMemcpyAsync(h->w_pair_out.p, po.data(), nl * 8, hipMemcpyHostToDevice, h->stream));
        // rows = (possibly renormalised) centroids as queries, columns = the same table as the vector tile
        pack_query_tiles(h, d_cq.as<float>(), {{0u, (uint32_t)nl}});
        ScanArgs sa{d_cq.as<float>(), d_cq.as<float>(), h->w_items.as<ScanItem>(), h->w_pair_query.as<uint32_t>(),
                    h->w_pair_out.as<uint64_t>(), d_full.as<float>(), h->w_qtile.as<float>(), h->dpad, h->metric, 0};
        const size_t all_qg4[4] = {0, 0, items.size(), 0};
        launch_scan(sa, all_qg4, h->stream);
        launch_pack_upper(d_full.as<float>(), (uint32_t)nl, h->d_interdis.as<float>(), h->stream);
        HIP_CHECK(stream_sync(h->stream));
        if (h->metric == METRIC_IP) {
            // acos on the host libm, as the reference does (IndexIVF.cpp:109-110)
            std::vector<float> t(sz);
            HIP_CHECK(hipMemcpy(t.data(), h->d_interdis.p, sz * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < sz; i++) t[i] = std::acos(t[i]);
            HIP_CHECK(hipMemcpy(h->d_interdis.p, t.data(), sz * 4, hipMemcpyHostToDevice));
        }
    }
    h->have_interdis = true;
    API_END
}

int amd_ivf_get_interdis(amd_ivf_t* h, float* table) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (!h->have_interdis) throw EngineError("centroid table not set");
    HIP_CHECK(hipMemcpy(table, h->d_interdis.p, h->nlist * (h->nlist - 1) / 2 * 4, hipMemcpyDeviceToHost));
    API_END
}

int amd_ivf_set_tuner(amd_ivf_t* h, size_t max_topk, size_t ntraces, const size_t* trace_len, const float* const* trace_x,
                      const float* const* trace_y, const float* const* trace_std, const float* arcos_list) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    std::vector<uint32_t> off(ntraces + 1, 0);
    for (size_t i = 0; i < ntraces; i++) off[i + 1] = off[i] + (uint32_t)trace_len[i];
    std::vector<float> x(off[ntraces]), y(off[ntraces]), sd(off[ntraces]);
    for (size_t i = 0; i < ntraces; i++) {
        if (trace_len[i] == 0) throw EngineError("empty trace");
        memcpy(&x[off[i]], trace_x[i], trace_len[i] * 4);
        memcpy(&y[off[i]], trace_y[i], trace_len[i] * 4);
        memcpy(&sd[off[i]], trace_std[i], trace_len[i] * 4);
    }
    h->d_trace_off.ensure(off.size() * 4);
    h->d_trace_x.ensure(x.size() * 4);
    h->d_trace_y.ensure(x.size() * 4);
    h->d_trace_std.ensure(x.size() * 4);
    h->d_arcos.ensure(500 * 4);
    HIP_CHECK(hipMemcpy(h->d_trace_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->d_trace_x.p, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->d_trace_y.p, y.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->d_trace_std.p, sd.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->d_arcos.p, arcos_list, 500 * 4, hipMemcpyHostToDevice));
    h->tuner_max_topk = max_topk;
    h->tuner_ntraces = ntraces;
    h->tuner_trace_cap = 0;
    for (size_t i = 0; i < ntraces; i++) h->tuner_trace_cap = std::max(h->tuner_trace_cap, trace_len[i]);
    h->have_tuner = true;
    API_END
}

// one slice of an adaptive batch: queries [q0, q0+n) of the call, on lane `L` (L == h or one of h's kids)
// The coarse ranking of a tune / train search into w_cdis / w_ckeys: the engine's own over all nlist centroids (what
// Error_sys::search asks its quantizer for, profile.cpp:220), or the rows the caller handed to search_preassigned
// (Auncel/IndexIVF.cpp:382-386).  Returns the row length = the length of the probe loop.
static size_t coarse_or_given(amd_ivf_t* L, const float* d_x, size_t n, int coarse_mode, bool fused_ok, size_t coarse_prefix) {
    const size_t nlist = L->nlist;
    if (L->spec_use) {  // second pass of the "redo" regime: these queries' rankings were re-ranked while the first pass ran
        L->w_cdis.ensure(n * nlist * 4);
        L->w_ckeys.ensure(n * nlist * 8);
        HIP_CHECK(hipStreamWaitEvent(L->stream, L->ev_spec_done, 0));
        launch_spec_gather(L->w_spec_pick.as<int32_t>(), (uint32_t)n, (uint32_t)nlist, (uint32_t)L->spec_ncopy, L->w_spec_dis.as<float>(),
                           L->w_spec_keys.as<int64_t>(), L->w_cdis.as<float>(), L->w_ckeys.as<int64_t>(), L->stream);
        L->tie_rows_host += n;
        return nlist;
    }
    if (!L->given_keys) {
        L->w_cdis.ensure(n * nlist * 4);
        L->w_ckeys.ensure(n * nlist * 8);
        coarse_dev(L, d_x, n, nlist, coarse_mode, L->w_cdis.as<float>(), L->w_ckeys.as<int64_t>(), fused_ok, coarse_prefix);
        return nlist;
    }
    const size_t np = L->given_nprobe;
    if (np <= nlist / 8 + 20) throw EngineError("tune / train mode reads coarse entries 0 .. nlist/8 + 20: nprobe too small");
    L->w_cdis.ensure(n * np * 4);
    L->w_ckeys.ensure(n * np * 8);
    HIP_CHECK(hipMemcpyAsync(L->w_cdis.p, L->given_dis, n * np * 4, hipMemcpyHostToDevice, L->stream));
    HIP_CHECK(hipMemcpyAsync(L->w_ckeys.p, L->given_keys, n * np * 8, hipMemcpyHostToDevice, L->stream));
    return np;
}

static void adaptive_slice(amd_ivf_t* L, const float* d_x, size_t id0, size_t n, size_t query_topk, float multipler, float std_m,
                           const float* dreq, const float* dgt, unsigned long long* dnp, float* dtr, int profile, int coarse_mode,
                           float* D, int64_t* I, const IntRange& qr, size_t coarse_prefix, bool defer_finish,
                           const std::function<void()>* tail_gathers = nullptr) {
    use_device(L);
    const size_t K = ix(L)->tuner_max_topk, nlist = L->nlist;
    // full coarse ranking (Error_sys::search sets nprobe = nlist, profile.cpp:220), or the caller's
    const size_t np_row = coarse_or_given(L, d_x, n, coarse_mode, ix(L)->allow_fused && ix(L)->centroid_range.fusable_with(qr, L->metric), coarse_prefix);
    host_stamp("coarse");
    // (at most four queries: the four small launches between the coarse ranking and the first round are made as one)
    const bool fuse_small = n <= 4 && !L->spec_wanted;
    if (L->want_first_tie) {
        L->w_first_tie.ensure(n * 4);
        // Which queries will have to be searched again is known only when this pass ends (first run < 2 my_nprobe + 14), but the
        // expensive part of searching them again -- the reference's heap over all nlist centroids, a serial 4096-element heap
        // sort per query -- needs nothing from this pass.  Every query that could qualify with the probes of the first two
        // rounds gets its rows set aside now (the distance table lives in w_dist, which round 0 overwrites) and the heap runs on
        // a side stream under this pass.
        static const bool no_spec = getenv("AUNCEL_AMD_NO_TIE_SPECULATION") != nullptr;
        // (slots for the rankings whose first run is in reach: 512 for calls of up to 5000 queries -- about 400 of the bench workload's
        // 5000 rankings take one -- and an eighth of the call beyond, so that larger calls, e.g. queued tickets served together, do not
        // send what overflows through the second pass)
        const uint32_t SPEC_CAP = n <= 5120 ? 512u : (uint32_t)(((n / 8 + 63) / 64) * 64);
        constexpr uint32_t SPEC_NEAR = 64, SPEC_WINDOW = 2 * (12 + 144) + 14;
        L->spec_cap = SPEC_CAP;
        L->spec_valid = false;
        const bool can_spec = !no_spec && L->spec_wanted && !L->given_keys && np_row == nlist && n <= L->dist_budget_floats / std::max<size_t>(nlist, 1) &&
                              heap_tie_order_lds((uint32_t)nlist, (uint32_t)nlist) <= 160 * 1024;
        if (!can_spec && !fuse_small)
            launch_first_tie(L->w_cdis.as<float>(), (uint32_t)n, (uint32_t)nlist, (uint32_t)L->first_tie_nreal, L->w_first_tie.as<uint32_t>(), L->stream);
        if (can_spec) {
            const size_t ncopy = std::min(L->first_tie_nreal, nlist);
            L->w_spec_full.ensure((size_t)SPEC_CAP * nlist * 4);
            L->w_spec_dis.ensure((size_t)SPEC_CAP * nlist * 4);
            L->w_spec_keys.ensure((size_t)SPEC_CAP * nlist * 8);
            L->w_spec_count.ensure(32);  // count | rankings the patch changed | rows the heap re-ranked (u64; not reported: most are never used) | scratch cursor (u64)
            L->w_spec_slot.ensure(n * 4);
            L->w_spec_query.ensure((size_t)SPEC_CAP * 4);
            ensure_context_streams(L);
            HIP_CHECK(hipStreamWaitEvent(L->stream, L->ev_spec_done, 0));  // (the previous search's slots are no longer being written)
            HIP_CHECK(hipMemsetAsync(L->w_spec_count.p, 0, 32, L->stream));
            // (the first run of every ranking and the slots of the rankings to re-rank in one launch: the heap starts right behind the
            // coarse ranking; slots go to the waves in the order they arrive -- a ranking that finds none is searched again at the end)
            (void)SPEC_NEAR;
            launch_tie_collect(L->w_cdis.as<float>(), (uint32_t)n, (uint32_t)L->first_tie_nreal, SPEC_WINDOW, SPEC_CAP, (uint32_t)nlist, (uint32_t)ncopy,
                               L->w_dist.as<float>(), L->w_ckeys.as<int64_t>(), L->w_first_tie.as<uint32_t>(), L->w_spec_count.as<uint32_t>(),
                               L->w_spec_slot.as<int32_t>(), L->w_spec_full.as<float>(), L->w_spec_dis.as<float>(), L->w_spec_keys.as<int64_t>(),
                               L->w_spec_query.as<uint32_t>(), L->stream);
            HIP_CHECK(hipEventRecord(L->ev_spec_go, L->stream));
            HIP_CHECK(hipStreamWaitEvent(L->spec_stream, L->ev_spec_go, 0));
            launch_heap_tie_order(L->w_spec_full.as<float>(), SPEC_CAP, (uint32_t)nlist, (uint32_t)nlist, (uint32_t)ncopy, L->metric,
                                  L->w_spec_dis.as<float>(), L->w_spec_keys.as<int64_t>(),
                                  reinterpret_cast<unsigned long long*>(L->w_spec_count.as<uint32_t>() + 2), L->spec_stream,
                                  L->w_spec_count.as<uint32_t>());
            HIP_CHECK(hipEventRecord(L->ev_spec_done, L->spec_stream));
            L->spec_valid = true;
            L->spec_ncopy = ncopy;
            // The heap's order can be applied to THIS pass when it arrives before anything has read the order inside a run of equal
            // distances: that is the selection of round 0 (the stop rule's boundary distances and the probe order; the planner
            // takes whole runs into a round, so the scan does not depend on it).  With the level-parallel filling and the pipelined
            // heap sort (nlist a power of two) the heap takes ~1.5 ms a row, about what coarse ranking -> planning -> scan of
            // round 0 take among other searches; the literal heap (12 ms a row) stays under the pass and feeds a second one.
            // (only run_rounds_device calls RoundSpec::before_first_select: with the host-planned rounds of AUNCEL_AMD_HOST_PLAN the
            // heap's order feeds the second pass instead, as for an nlist that is not a power of two)
            static const bool no_patch = getenv("AUNCEL_AMD_NO_TIE_PATCH") != nullptr || getenv("AUNCEL_AMD_HOST_PLAN") != nullptr;
            L->spec_inline = !no_patch && (nlist & (nlist - 1)) == 0 && nlist >= 64;
        }
    }
    const bool tie_inline = L->want_first_tie && L->spec_valid && L->spec_inline;
    struct FuseScope {  // (init_state and byte_queries record their launches while this is alive)
        amd_ivf* h;
        bool on;
        FuseScope(amd_ivf* hh, bool o) : h(hh), on(o) {
            if (on) {
                h->fuse = amd_ivf::SmallFuse{};
                h->fuse.active = true;
            }
        }
        ~FuseScope() { h->fuse.active = false; }
    } fuse_scope(L, fuse_small);
    init_state(L, n, K, true);
    auto set_online = [L, nlist, n, np_row](const uint32_t* only = nullptr, const uint32_t* only_count = nullptr, uint32_t cap = 0) {
        launch_set_online(L->metric, (uint32_t)nlist, only ? cap : (uint32_t)n, L->w_cdis.as<float>(), L->w_ckeys.as<int64_t>(), (uint32_t)np_row,
                          ix(L)->d_interdis.as<float>(), ix(L)->d_arcos.as<float>(), L->w_dtb.as<float>(), L->w_error.as<uint32_t>(), L->stream,
                          only, only_count);
    };
    if (!fuse_small && !tie_inline) set_online();
    RoundSpec base;
    if (tie_inline) {
        // room for the rows the patch moves: a round's rows of every slot if that is not beyond reason (a query that finds no room is
        // searched again)
        size_t maxlist = 0;
        for (size_t l = 0; l < nlist; l++) maxlist = std::max<size_t>(maxlist, ix(L)->h_list_off[l + 1] - ix(L)->h_list_off[l]);
        const size_t SCRATCH_FLOATS = std::min<size_t>((size_t)64 << 20, std::max<size_t>((size_t)8 << 20, (size_t)L->spec_cap * 16 * ((maxlist + 1023) & ~(size_t)1023)));
        L->w_spec_scratch.ensure(SCRATCH_FLOATS * 4);
        base.run_ties = true;
        base.before_first_select = [L, nlist, set_online, SCRATCH_FLOATS]() {
            HIP_CHECK(hipStreamWaitEvent(L->stream, L->ev_spec_done, 0));
            TiePatchArgs ta{};
            ta.count = L->w_spec_count.as<uint32_t>();
            ta.cap = L->spec_cap;
            ta.nlist = (uint32_t)nlist;
            ta.ncopy = (uint32_t)L->spec_ncopy;
            ta.key_stride = (uint32_t)nlist;
            ta.slot_query = L->w_spec_query.as<uint32_t>();
            ta.slot_of = L->w_spec_slot.as<int32_t>();
            ta.s_keys = L->w_spec_keys.as<int64_t>();
            ta.ckeys = L->w_ckeys.as<int64_t>();
            ta.seg_count = L->w_pl_cnt.as<uint32_t>();
            ta.seg_begin = L->w_seg_begin.as<uint32_t>();
            ta.seg_list = L->w_seg_list.as<int32_t>();
            ta.seg_off = L->w_seg_off.as<uint64_t>();
            ta.list_off = ix(L)->d_list_off.as<uint64_t>();
            ta.dist = L->w_dist.as<float>();
            ta.row_align = L->row_align_now;
            ta.scratch = L->w_spec_scratch.as<float>();
            ta.scratch_floats = SCRATCH_FLOATS;
            ta.cursor = reinterpret_cast<unsigned long long*>(L->w_spec_count.as<uint32_t>() + 4);
            ta.patched = L->w_spec_count.as<uint32_t>() + 1;
            launch_tie_patch(ta, L->stream);
            set_online();
        };
        // the first selection in two launches (chained rounds; AUNCEL_AMD_NO_SPLIT_SELECT: in one, behind the heap)
        static const bool no_split = getenv("AUNCEL_AMD_NO_SPLIT_SELECT") != nullptr;
        if (!no_split && n >= 256) {
            L->w_split.ensure((2 * n + 8) * 4);
            uint32_t* counts = L->w_split.as<uint32_t>();
            uint32_t* q_free = counts + 8;
            uint32_t* q_wait = q_free + n;
            const auto patch_hook = base.before_first_select;
            const uint32_t wait_cap = (uint32_t)std::min<size_t>(n, L->spec_cap);  // (tie_collect_kernel hands out at most that many slots)
            base.split_free = q_free;
            base.split_wait = q_wait;
            base.split_counts = counts;
            base.split_wait_cap = wait_cap;
            base.split_prepare = [L, n, counts, q_free, q_wait, set_online]() {
                HIP_CHECK(hipMemsetAsync(counts, 0, 8, L->stream));
                // (round 0 of an adaptive search: every query of the call is active, in order)
                launch_partition_qsel(nullptr, nullptr, (uint32_t)n, L->w_spec_slot.as<int32_t>(), q_free, q_wait, counts, L->stream);
                set_online();
            };
            base.split_between = [L, patch_hook, counts, q_wait, wait_cap, set_online]() {
                (void)set_online;
                patch_hook();  // (waits for the heap, patches, and derives the boundary distances again -- of every query: the patched
                               // rankings are a tenth of them and the launch is 0.045 ms)
            };
        }
    }
    base.fused = ix(L)->allow_fused && ix(L)->db_range.fusable_with(qr, L->metric);
    base.bytes = byte_queries(L, ix(L), d_x, n, qr);
    if (fuse_small) {
        L->fuse.active = false;
        SmallStateArgs sa{};
        sa.init = L->fuse.init;
        sa.metric = L->metric;
        sa.nlist = (uint32_t)nlist;
        sa.nq = (uint32_t)n;
        sa.coarse_dis = L->w_cdis.as<float>();
        sa.coarse_keys = L->w_ckeys.as<int64_t>();
        sa.coarse_stride = (uint32_t)np_row;
        sa.interdis = ix(L)->d_interdis.as<float>();
        sa.arcos = ix(L)->d_arcos.as<float>();
        sa.dtb = L->w_dtb.as<float>();
        if (L->want_first_tie) {
            sa.ft_sorted_dis = L->w_cdis.as<float>();
            sa.ft_stride = (uint32_t)nlist;
            sa.ft_nreal = (uint32_t)L->first_tie_nreal;
            sa.ft_out = L->w_first_tie.as<uint32_t>();
        }
        if (L->fuse.have_bytes) {
            sa.bx = L->fuse.bx;
            sa.bout = L->fuse.bout;
            sa.bcx = L->fuse.bcx;
        }
        sa.d = L->d;
        sa.dpad = L->dpad;
        launch_small_state(sa, L->stream);
    }
    ix(L)->last_arith = base.bytes ? 2 : base.fused ? 1 : 0;
    base.k = (int)K;
    base.id_offset = id0;
    base.d_x = d_x;
    base.d_cdis = L->w_cdis.as<float>();
    base.d_ckeys = L->w_ckeys.as<int64_t>();
    base.coarse_stride = (uint32_t)np_row;
    base.tuner = make_tuner(L, query_topk, multipler, std_m, dreq, dgt, dnp, dtr, profile);
    // (one query per call: a first round of 64 probes instead of 12 ends 95 % of the bench workload's queries in it instead of 73 %,
    // and moved neither the median nor the p90 of the call -- 0.25 / 0.56 ms: the slow tenth are not the queries that need a second
    // round but the ones in which equal distances met, whose result is the reference's heap replayed over ~500 admissions, 0.3 ms on
    // one wave; profiles/r05_latency_batch1.txt)
    const size_t first_env = std::max<size_t>(1, (size_t)opt(L, OPT_ROUND_FIRST, 12));
    static const bool host_plan = getenv("AUNCEL_AMD_HOST_PLAN") != nullptr;
    if (host_plan) {
        run_rounds(L, base, n, first_env, np_row, dnp, id0);
        HIP_CHECK(hipMemcpyAsync(D, L->w_D.p, n * K * 4, hipMemcpyDeviceToHost, L->stream));
        HIP_CHECK(hipMemcpyAsync(I, L->w_I.p, n * K * 8, hipMemcpyDeviceToHost, L->stream));
        HIP_CHECK(stream_sync(L->stream));
        fold_stats(L, n);
        return;
    }
    base.caller_checks_error = true;
    DirectOut direct(L, D, I);
    host_stamp("state");
    L->spec_done = false;
    if (defer_finish && tail_gathers && n < 20)  // (one lane, a few queries: the read-backs may ride with the first look)
        base.spec_finish = [&]() {
            finish_results(L, n, K, D, I, nullptr, true);
            (*tail_gathers)();
        };
    run_rounds_device(L, base, n, first_env, np_row, dnp);
    host_stamp("rounds");
    if (!L->spec_done) finish_results(L, n, K, D, I, nullptr, defer_finish);
}

static size_t lane_count(size_t n) {
    static const int env = getenv("AUNCEL_AMD_SLICES") ? atoi(getenv("AUNCEL_AMD_SLICES")) : 0;
    if (env > 0) return (size_t)env;
    (void)n;
    return 1;  // measured on MI355X: concurrent slices halve the queries per list and lose more in the
               // VALU-bound scan than they hide of selection + planning (bench: 1 lane 25 ms/step, 2 lanes 29)
}

static void adaptive_core_once(amd_ivf_t* h, const float* d_x, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                            const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                            uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I, const IntRange& qr) {
    use_device(h);
    CallScope call_scope(h);
    const amd_ivf* owner = ix(h);
    if (!owner->have_tuner || !owner->have_interdis)
        throw EngineError("Search tune start can't start without IVF_pro init and training");
    if (n == 0) return;
    const size_t K = owner->tuner_max_topk, nlist = h->nlist;
    if (nlist <= nlist / 8 + 20) throw EngineError("tune mode needs nprobe(=nlist) > nlist/8 + 20");
    size_t ntr = 0;
    while (((size_t)1 << ntr) <= nlist / 8) ntr++;
    if (owner->tuner_ntraces < ntr) throw EngineError("not enough traces for this nlist");
    if (query_topk == 0 || query_topk > K) throw EngineError("query_topk out of range");
    {
        const int pt = (int)opt(h, OPT_PHASE_TIMING, -1);
        h->timer.off = pt >= 0 ? pt == 0 : n < 20;
    }
    g_stamps.clear();
    host_stamp("begin");
    WallClock wc(h->stream, h->timer.off);
    upload_lists(h);
    host_stamp("clock+lists");
    const size_t nabs = start + n;
    // per-absolute-query arrays on the device, shared by all lanes
    DevBuf &d_req = h->w_misc2, &d_np = h->w_misc3;
    // (sized for every resident query at once: a caller that walks through slices of its resident queries would otherwise grow
    // these arrays slice by slice -- and a reallocation frees device memory, which waits for every stream of the device: with
    // four searches in flight the first visit of a context to a later slice cost a fifth of a step)
    const amd_ivf* rsrc = h->is_clone && h->n_resident == 0 && h->parent ? h->parent : h;
    const size_t ncap = std::max(nabs, rsrc->n_resident);
    d_req.ensure(ncap * 4 * 2 + (gt_D ? ncap * K * 4 : 0) + 64);
    float* dreq = d_req.as<float>();
    float* dtr = dreq + ncap;
    float* dgt = gt_D ? dtr + ncap : nullptr;
    d_np.ensure(ncap * 8);
    // only the entries of this call's queries are read or written on the device (everything is indexed by absolute id)
    h2d_small(h, dreq + start, require_acc + start, n * 4, h->stream);
    h2d_small(h, dtr + start, t_recalls + start, n * 4, h->stream);
    if (gt_D) HIP_CHECK(hipMemcpyAsync(dgt + start * K, gt_D + start * K, n * K * 4, hipMemcpyHostToDevice, h->stream));
    h2d_small(h, d_np.as<unsigned long long>() + start, my_nprobe + start, n * 8, h->stream);
    flush_h2d(h, h->stream);
    host_stamp("inputs");

    // How much of the coarse ranking can be consumed: set_online reads entries 0 .. nlist/8+20, and the probe loop ends at
    // my_nprobe <= floor((nlist/8) * multipler) (IndexIVF.cpp:615-632) or at a value the caller passed in.  When that is
    // well short of nlist only this prefix is ranked (launch_sort_rows).
    size_t coarse_prefix = std::max<size_t>(nlist / 8 + 21, (size_t)((double)(nlist / 8) * (double)multipler) + 2);
    for (size_t i = start; i < nabs; i++) coarse_prefix = std::max<size_t>(coarse_prefix, (size_t)my_nprobe[i] + 1);
    coarse_prefix += 16;
    if (coarse_prefix >= nlist || getenv("AUNCEL_AMD_FULL_COARSE_SORT")) coarse_prefix = 0;

    const size_t nl = std::min(lane_count(n), std::max<size_t>(1, n / 64));
    while (h->kids.size() + 1 < nl) {
        std::unique_ptr<amd_ivf> kid(new amd_ivf);
        kid->parent = ix(h);
        kid->d = h->d;
        kid->dpad = h->dpad;
        kid->nlist = h->nlist;
        kid->metric = h->metric;
        kid->device = h->device;
        kid->dist_budget_floats = h->dist_budget_floats;
        kid->stream = make_main_stream();
        h->kids.push_back(std::move(kid));
    }
    std::vector<amd_ivf*> lanes(nl);
    lanes[0] = h;
    for (size_t i = 1; i < nl; i++) lanes[i] = h->kids[i - 1].get();
    for (size_t i = 1; i < nl; i++) {  // the slices' rows of a caller-supplied coarse ranking
        const size_t q0 = n * i / nl;
        lanes[i]->given_keys = h->given_keys ? h->given_keys + q0 * h->given_nprobe : nullptr;
        lanes[i]->given_dis = h->given_keys ? h->given_dis + q0 * h->given_nprobe : nullptr;
        lanes[i]->given_nprobe = h->given_nprobe;
    }
    for (amd_ivf* L : lanes) {
        L->force_heap_select = h->force_heap_select;
        L->scan_bytes = L->scan_min_bytes = L->scan_min_bytes_thr = 0;
        L->scan_slots = L->scan_useful = 0;
    }
    // what this function reads back once the slices are done (one lane: queued together with the slice's own read-backs, or --
    // a few queries -- with the look after the first round)
    const std::function<void()> tail_gathers = [&]() {
        d2h_small(h, my_nprobe + start, d_np.as<unsigned long long>() + start, n * 8, h->stream);
        d2h_small(h, t_recalls + start, dtr + start, n * 4, h->stream);
        if (h->want_first_tie) {
            h->first_tie_host.assign(n, 0);
            d2h_small(h, h->first_tie_host.data(), h->w_first_tie.p, n * 4, h->stream);
            if (h->spec_valid && nl == 1) {
                h->spec_slot_host.assign(n, -1);
                d2h_small(h, h->spec_slot_host.data(), h->w_spec_slot.p, n * 4, h->stream);
                h->tie_patched_host = 0;
                if (h->spec_inline) d2h_small(h, &h->tie_patched_host, h->w_spec_count.as<uint32_t>() + 1, 4, h->stream);
            } else {
                h->spec_valid = false;
            }
        }
    };
    h->spec_done = false;
    std::vector<std::exception_ptr> errs(nl);
    auto run = [&](size_t i) {
        const size_t q0 = n * i / nl, q1 = n * (i + 1) / nl;
        try {
            // (one lane: the slice's read-back joins this function's own, one synchronisation ends the call)
            adaptive_slice(lanes[i], d_x + q0 * h->dpad, start + q0, q1 - q0, query_topk, multipler, std_m, dreq, dgt,
                           d_np.as<unsigned long long>(), dtr, profile, coarse_mode, D + q0 * K, I + q0 * K, qr, coarse_prefix, nl == 1,
                           nl == 1 ? &tail_gathers : nullptr);
        } catch (...) {
            errs[i] = std::current_exception();
        }
    };
    SmallCopies pending(h);  // (whatever is still queued when this scope is left by an exception names memory of this call)
    std::vector<std::thread> th;
    for (size_t i = 1; i < nl; i++) th.emplace_back(run, i);
    run(0);
    for (auto& t : th) t.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
    if (!(nl == 1 && h->spec_done)) {  // (else the look after the first round brought everything along)
        tail_gathers();
        sync_and_flush(h, h->stream);
    }
    h->spec_done = false;
    host_stamp("final-sync");
    // fold the kids' counters and kernel timings into the handle
    const double wall = wc.stop();
    host_stamp("clock");
    print_stamps();
    double ms[NCAT] = {0}, ln[NCAT] = {0};
    double bytes = 0, slots = 0, useful = 0, min_bytes = 0;
    for (amd_ivf* L : lanes) {
        double m[NCAT], c[NCAT];
        L->timer.collect(m, NCAT, c);
        for (int k = 0; k < NCAT; k++) ms[k] += m[k], ln[k] += c[k];
        bytes += L->scan_bytes;
        min_bytes += L->scan_min_bytes;
        slots += L->scan_slots;
        useful += L->scan_useful;
        if (L != h) {
            for (int k = 0; k < 4; k++) h->stats_host[k] += L->stats_host[k], L->stats_host[k] = 0;
        }
    }
    fill_timing(h, ms, ln);
    h->timer.off = false;
    h->timing[3] = wall;
    h->timing[5] = bytes;
    h->last_min_bytes = min_bytes;
    h->timing[6] = slots > 0 ? useful / slots : 0;
}

// A call of n >= 20 queries with the reference's exact-distance tie order: the whole call is searched with runs of equal
// coarse distances in centroid-number order (sorting; no heap), then the queries whose first run starts below what they read
// (entries < 2 my_nprobe + 14, see adaptive_core) are searched again as one small call with the heap's order, and their rows,
// my_nprobe, t_recalls and share of the statistics are replaced.
static void adaptive_redo_ties(amd_ivf_t* h, const float* d_x, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                               const float* require_acc, const float* gt_D, int profile, int coarse_mode, uint64_t* my_nprobe,
                               float* t_recalls, float* D, int64_t* I, const IntRange& qr) {
    const size_t K = ix(h)->tuner_max_topk, nlist = h->nlist;
    const std::vector<uint64_t> np0(my_nprobe + start, my_nprobe + start + n);
    const std::vector<float> tr0(t_recalls + start, t_recalls + start + n);
    size_t nreal = std::max<size_t>(nlist / 8 + 21, (size_t)((double)(nlist / 8) * (double)multipler) + 2);
    for (size_t i = 0; i < n; i++) nreal = std::max<size_t>(nreal, (size_t)np0[i] + 1);
    nreal = std::min(nreal + 16, nlist);
    struct Restore {
        amd_ivf_t* h;
        ~Restore() {
            h->ties_override = -1;
            h->want_first_tie = false;
            h->spec_use = false;
            h->spec_wanted = false;
        }
    } restore{h};
    h->ties_override = 0;
    h->want_first_tie = true;
    h->first_tie_nreal = nreal;
    h->spec_valid = false;
    h->spec_wanted = true;
    with_select_fallback(h, [&] { adaptive_core_once(h, d_x, start, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode, my_nprobe, t_recalls, D, I, qr); });
    h->want_first_tie = false;
    h->spec_wanted = false;
    std::vector<uint32_t> again;
    // (with the heap's order applied to the pass itself -- launch_tie_patch -- a query whose ranking had a slot is done: the order it
    // read was the reference's; left are the queries without a slot, the ones the patch gave up on (-2) and reads past what was ranked)
    const bool patched = h->spec_inline && h->spec_valid && h->spec_slot_host.size() == n;
    for (size_t i = 0; i < n; i++) {
        const uint64_t bound = 2 * my_nprobe[start + i] + 14;
        const bool in_order = patched && h->spec_slot_host[i] >= 0 && bound <= h->spec_ncopy;
        if ((h->first_tie_host[i] < bound && !in_order) || bound + 1 >= nreal) again.push_back((uint32_t)i);
    }
    h->last_tie_redone = again.size();
    h->last_tie_patched = patched ? h->tie_patched_host : 0;
    if (again.empty()) return;
    const size_t m = again.size();
    // what the first pass counted for these queries leaves the statistics (the device arrays are about to be reused)
    std::vector<unsigned long long> nscan(n);
    std::vector<uint2> qstat(n);
    HIP_CHECK(hipMemcpyAsync(nscan.data(), h->w_nscan.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(qstat.data(), h->w_qstat.p, n * sizeof(uint2), hipMemcpyDeviceToHost, h->stream));
    // their query rows, gathered on the device; their rankings from the slots re-ranked during the first pass if every one has one
    bool spec = h->spec_valid && h->spec_slot_host.size() == n;
    std::vector<int32_t> pick(m, -1);
    for (size_t j = 0; j < m && spec; j++) {
        pick[j] = h->spec_slot_host[again[j]];
        if (pick[j] <= -2) pick[j] = -2 - pick[j];  // (a slot whose order the first pass could not take over: launch_tie_patch)
        spec = pick[j] >= 0;
    }
    if (getenv("AUNCEL_AMD_DEBUG_REDO")) {
        size_t have = 0, nslots = 0;
        for (size_t j = 0; j < m; j++) have += h->spec_valid && h->spec_slot_host.size() == n && h->spec_slot_host[again[j]] >= 0;
        for (int32_t v : h->spec_slot_host) nslots += v >= 0;
        fprintf(stderr, "[redo] n %zu again %zu spec_valid %d slots given %zu, of the queries searched again %zu have one -> %s\n", n, m,
                (int)h->spec_valid, nslots, have, spec ? "slots" : "heap in the second pass");
    }
    h->w_redo_x.ensure(m * h->dpad * sizeof(float));
    h->w_redo_idx.ensure(m * 4);
    HIP_CHECK(hipMemcpyAsync(h->w_redo_idx.p, again.data(), m * 4, hipMemcpyHostToDevice, h->stream));
    launch_gather_rows(d_x, h->w_redo_idx.as<uint32_t>(), (uint32_t)m, (uint32_t)h->dpad, h->w_redo_x.as<float>(), h->stream);
    if (spec) {
        h->w_spec_pick.ensure(m * 4);
        HIP_CHECK(hipMemcpyAsync(h->w_spec_pick.p, pick.data(), m * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIP_CHECK(stream_sync(h->stream));
    for (size_t j = 0; j < m; j++) {
        h->stats_host[0] -= 1;
        h->stats_host[1] -= qstat[again[j]].x;
        h->stats_host[2] -= nscan[again[j]];
        h->stats_host[3] -= qstat[again[j]].y;
    }
    std::vector<float> req(m), tr(m), Dc(m * K), gt;
    std::vector<uint64_t> np(m);
    std::vector<int64_t> Ic(m * K);
    if (gt_D) gt.resize(m * K);
    for (size_t j = 0; j < m; j++) {
        const size_t id = start + again[j];
        req[j] = require_acc[id];
        np[j] = np0[again[j]];
        tr[j] = tr0[again[j]];
        if (gt_D) std::copy(gt_D + id * K, gt_D + (id + 1) * K, gt.begin() + j * K);
    }
    h->ties_override = 1;
    h->spec_use = spec;
    with_select_fallback(h, [&] {
        adaptive_core_once(h, h->w_redo_x.as<float>(), 0, m, query_topk, multipler, std_m, req.data(), gt_D ? gt.data() : nullptr, profile,
                           coarse_mode, np.data(), tr.data(), Dc.data(), Ic.data(), qr);
    });
    for (size_t j = 0; j < m; j++) {
        const size_t i = again[j];
        my_nprobe[start + i] = np[j];
        t_recalls[start + i] = tr[j];
        std::copy(Dc.begin() + j * K, Dc.begin() + (j + 1) * K, D + i * K);
        std::copy(Ic.begin() + j * K, Ic.begin() + (j + 1) * K, I + i * K);
    }
}

// Fewer than 20 queries per call is the regime in which the reference ranks exact coarse distances (utils.cpp:624-655), so
// there the order inside runs of bit-equal distances has to be its heap's.  Re-running that heap costs up to 2.8 ms a
// ranking (nlist 4096) and one ranking in eight holds such a run somewhere in the prefix that may be read -- but a query
// only ever reads entries below max(my_nprobe, 2 my_nprobe + 14) (the probes it scans; the set_online windows of the stages
// it evaluates: stage s reads entries up to 2^ceil(log2 s) + 14).  So: search with the runs in centroid-number order,
// which differs from the heap's only inside them; if no query's first run starts below its bound the result is the
// reference's, otherwise (about one call in a hundred) the call is repeated with the heap's order.
static void adaptive_core(amd_ivf_t* h, const float* d_x, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                          const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                          uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I, const IntRange& qr) {
    const int ties_opt = (int)opt(h, OPT_COARSE_TIES, -1);
    const bool can_speculate = !h->given_keys && n > 0 && h->nlist > 128 && multipler >= 1.f && !(profile & 2) && h->kids.empty();
    const bool speculate = can_speculate && n < 20 && ties_opt < 0;
    // larger calls: "redo" searches again, with the heap's order, exactly the queries whose first run of equal coarse distances
    // lies within what they read (a handful in thousands) -- the reference's exact-distance result for every query of the call
    const bool redo_some = can_speculate && n >= 20 && ties_opt == 2;
    if (redo_some) {
        adaptive_redo_ties(h, d_x, start, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode, my_nprobe, t_recalls, D, I, qr);
        return;
    }
    if (!speculate) {
        with_select_fallback(h, [&] { adaptive_core_once(h, d_x, start, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode, my_nprobe, t_recalls, D, I, qr); });
        return;
    }
    size_t stats0[4];
    for (int i = 0; i < 4; i++) stats0[i] = h->stats_host[i];
    const std::vector<uint64_t> np0(my_nprobe + start, my_nprobe + start + n);
    const std::vector<float> tr0(t_recalls + start, t_recalls + start + n);
    // the start of the first run of equal distances in every ranking of the call comes back with the results
    // (first_tie_kernel right behind the coarse ranking, over what adaptive_core_once ranks: its coarse_prefix)
    const size_t nlist = h->nlist;
    size_t nreal = std::max<size_t>(nlist / 8 + 21, (size_t)((double)(nlist / 8) * (double)multipler) + 2);
    for (size_t i = 0; i < n; i++) nreal = std::max<size_t>(nreal, (size_t)np0[i] + 1);
    nreal = std::min(nreal + 16, nlist);
    h->ties_override = 0;
    h->want_first_tie = true;
    h->first_tie_nreal = nreal;
    try {
        with_select_fallback(h, [&] { adaptive_core_once(h, d_x, start, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode, my_nprobe, t_recalls, D, I, qr); });
    } catch (...) {
        h->ties_override = -1;
        h->want_first_tie = false;
        throw;
    }
    h->ties_override = -1;
    h->want_first_tie = false;
    const std::vector<uint32_t>& first = h->first_tie_host;
    bool redo = false;
    for (size_t i = 0; i < n; i++) {
        const uint64_t bound = 2 * my_nprobe[start + i] + 14;
        redo = redo || first[i] < bound || bound + 1 >= nreal;  // (a run across the end of what was ranked: only deep queries get there)
    }
    if (!redo) return;
    for (int i = 0; i < 4; i++) h->stats_host[i] = stats0[i];
    std::copy(np0.begin(), np0.end(), my_nprobe + start);
    std::copy(tr0.begin(), tr0.end(), t_recalls + start);
    h->ties_override = 1;
    try {
        with_select_fallback(h, [&] { adaptive_core_once(h, d_x, start, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode, my_nprobe, t_recalls, D, I, qr); });
    } catch (...) {
        h->ties_override = -1;
        throw;
    }
    h->ties_override = -1;
}

int amd_ivf_search_adaptive(amd_ivf_t* h, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                            const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                            uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    // a search context that was never given resident queries of its own searches its owner's (amd_ivf_submit_adaptive's contexts)
    const amd_ivf* src = h->is_clone && h->n_resident == 0 && h->parent ? h->parent : h;
    if (start + n > src->n_resident) throw EngineError("resident query range out of bounds");
    adaptive_core(h, src->d_resident.as<float>() + start * h->dpad, start, n, query_topk, multipler, std_m, require_acc, gt_D,
                  profile, coarse_mode, my_nprobe, t_recalls, D, I, src->resident_range);
    API_END
}

int amd_ivf_search_adaptive_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t query_topk, float multipler,
                              float std_m, const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                              uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    if (n == 0) return 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    adaptive_core(h, h->w_x.as<float>(), id_offset, n, query_topk, multipler, std_m, require_acc, gt_D, profile, coarse_mode,
                  my_nprobe, t_recalls, D, I, qr);
    API_END
}

// scope of a caller-supplied coarse ranking on a handle
struct GivenCoarse {
    amd_ivf* h;
    GivenCoarse(amd_ivf* hh, size_t nprobe, const int64_t* keys, const float* dis) : h(hh) {
        if (!keys || !dis) throw EngineError("keys and coarse_dis are required");
        h->given_keys = keys;
        h->given_dis = dis;
        h->given_nprobe = nprobe;
    }
    ~GivenCoarse() {
        h->given_keys = nullptr;
        h->given_dis = nullptr;
        h->given_nprobe = 0;
    }
};

int amd_ivf_search_adaptive_pre(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t nprobe, const int64_t* keys,
                                const float* coarse_dis, size_t query_topk, float multipler, float std_m, const float* require_acc,
                                const float* gt_D, int profile, uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I) {
    API_BEGIN
    use_device(h);
    if (n == 0) return 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    GivenCoarse given(h, nprobe, keys, coarse_dis);
    adaptive_core(h, h->w_x.as<float>(), id_offset, n, query_topk, multipler, std_m, require_acc, gt_D, profile, 0, my_nprobe, t_recalls,
                  D, I, qr);
    API_END
}

static void train_core(amd_ivf_t* h, const float* d_x, size_t start, size_t n, size_t max_topk, const float* gt_D, size_t train_num,
                          int coarse_mode, float* const* raw, float* D, int64_t* I, const IntRange& qr) {
    use_device(h);
    if (!h->have_interdis) throw EngineError("Search tune start can't start without IVF_pro init and training");
    if (n == 0) return;
    const size_t K = max_topk, nlist = h->nlist;
    if (nlist <= nlist / 8 + 20) throw EngineError("train mode needs nprobe(=nlist) > nlist/8 + 20");
    size_t ntr = 0;
    while (((size_t)1 << ntr) <= nlist / 8) ntr++;
    upload_lists(h);
    const size_t per = train_num * (K / 4) * 2;  // floats per raw trace
    DevBuf d_raw;
    d_raw.ensure(ntr * per * 4 + 64);
    std::vector<float*> ptrs(ntr);
    for (size_t i = 0; i < ntr; i++) {
        ptrs[i] = d_raw.as<float>() + i * per;
        HIP_CHECK(hipMemcpyAsync(ptrs[i], raw[i], per * 4, hipMemcpyHostToDevice, h->stream));
    }
    h->w_rawptrs.ensure(ntr * sizeof(float*));
    HIP_CHECK(hipMemcpyAsync(h->w_rawptrs.p, ptrs.data(), ntr * sizeof(float*), hipMemcpyHostToDevice, h->stream));
    const size_t nabs = start + n;
    h->w_misc2.ensure(nabs * K * 4 + 64);
    HIP_CHECK(hipMemcpyAsync(h->w_misc2.p, gt_D, nabs * K * 4, hipMemcpyHostToDevice, h->stream));
    if (!h->d_arcos.p) {
        // construct_arcos (IVF_pro.cpp:151-160) on the host libm
        std::vector<float> lut(500);
        amd_ivf_arcos_table(lut.data());
        h->d_arcos.ensure(500 * 4);
        HIP_CHECK(hipMemcpy(h->d_arcos.p, lut.data(), 500 * 4, hipMemcpyHostToDevice));
    }
    // training stops after stage nlist/8 + 1 (IndexIVF.cpp:640-673); set_online reads entries 0 .. nlist/8+20
    size_t coarse_prefix = nlist / 8 + 21 + 16;
    if (coarse_prefix >= nlist || getenv("AUNCEL_AMD_FULL_COARSE_SORT")) coarse_prefix = 0;
    const size_t np_row = coarse_or_given(h, d_x, n, coarse_mode, h->allow_fused && h->centroid_range.fusable_with(qr, h->metric), coarse_prefix);
    init_state(h, n, K, true);
    launch_set_online(h->metric, (uint32_t)nlist, (uint32_t)n, h->w_cdis.as<float>(), h->w_ckeys.as<int64_t>(), (uint32_t)np_row,
                      h->d_interdis.as<float>(), h->d_arcos.as<float>(), h->w_dtb.as<float>(), h->w_error.as<uint32_t>(), h->stream);
    RoundSpec base;
    base.fused = h->allow_fused && h->db_range.fusable_with(qr, h->metric);
    base.bytes = byte_queries(h, ix(h), d_x, n, qr);
    ix(h)->last_arith = base.bytes ? 2 : base.fused ? 1 : 0;
    base.k = (int)K;
    base.id_offset = start;
    base.d_x = d_x;
    base.d_cdis = h->w_cdis.as<float>();
    base.d_ckeys = h->w_ckeys.as<int64_t>();
    base.coarse_stride = (uint32_t)np_row;
    base.train.enabled = 1;
    base.train.ntraces = (uint32_t)ntr;
    base.train.interdis = h->d_interdis.as<float>();
    base.train.arcos = h->d_arcos.as<float>();
    base.train.gt_D = h->w_misc2.as<float>();
    base.train.raw = reinterpret_cast<float* const*>(h->w_rawptrs.p);
    if (getenv("AUNCEL_AMD_HOST_PLAN")) run_rounds(h, base, n, 32, np_row, nullptr, start);
    else run_rounds_device(h, base, n, 32, np_row, nullptr);
    HIP_CHECK(hipMemcpyAsync(D, h->w_D.p, n * K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(I, h->w_I.p, n * K * 8, hipMemcpyDeviceToHost, h->stream));
    for (size_t i = 0; i < ntr; i++)
        HIP_CHECK(hipMemcpyAsync(raw[i], ptrs[i], per * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(stream_sync(h->stream));
    double ms[NCAT], ln[NCAT];
    h->timer.collect(ms, NCAT, ln);
}

int amd_ivf_train_samples(amd_ivf_t* h, size_t start, size_t n, size_t max_topk, const float* gt_D, size_t train_num,
                          int coarse_mode, float* const* raw, float* D, int64_t* I) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    train_core(h, h->d_resident.as<float>() + start * h->dpad, start, n, max_topk, gt_D, train_num, coarse_mode, raw, D, I,
               h->resident_range);
    API_END
}

int amd_ivf_train_samples_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t max_topk, const float* gt_D,
                            size_t train_num, int coarse_mode, float* const* raw, float* D, int64_t* I) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (n == 0) return 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    train_core(h, h->w_x.as<float>(), id_offset, n, max_topk, gt_D, train_num, coarse_mode, raw, D, I, qr);
    API_END
}

int amd_ivf_train_samples_pre(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t nprobe, const int64_t* keys,
                              const float* coarse_dis, size_t max_topk, const float* gt_D, size_t train_num, float* const* raw, float* D,
                              int64_t* I) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (n == 0) return 0;
    h->w_x.ensure(n * h->dpad * sizeof(float));
    upload_rows(h, h->w_x.as<float>(), x, n);
    IntRange qr;
    qr.add(x, n * (size_t)h->d);
    GivenCoarse given(h, nprobe, keys, coarse_dis);
    train_core(h, h->w_x.as<float>(), id_offset, n, max_topk, gt_D, train_num, 0, raw, D, I, qr);
    API_END
}

int amd_ivf_arcos_table(float out[500]) {
    const int len = 500;
    const float sc = len / 2;
    for (int i = 0; i < len; i++) {
        const float x = float(i - sc) / sc;
        out[i] = std::acos(x);
    }
    return 0;
}

int amd_ivf_trace_sb(const float* raw_xy, size_t n, size_t bs, float* out_x, float* out_y, float* out_std,
                     size_t* nbuckets) {
    API_BEGIN
    if (bs == 0) throw EngineError("bucket size must be positive");
    // std::sort with the reference's comparator and container type: equal keys must land in the
    // same order for the bucket means to agree bit for bit
    std::vector<std::pair<float, float>> tr(n);
    for (size_t i = 0; i < n; i++) tr[i] = std::make_pair(raw_xy[2 * i], raw_xy[2 * i + 1]);
    std::sort(tr.begin(), tr.end(),
              [](std::pair<float, float>& l, std::pair<float, float>& r) { return l.first > r.first; });
    size_t size = 0;
    for (auto& p : tr) size += (p.first < 0 && p.second < 0) ? 0 : 1;
    const size_t sz = (size + bs - 1) / bs;
    for (size_t b = 0; b < sz; b++) {
        const size_t left = b * bs, right = std::min((b + 1) * bs, size);
        float mx = 0, my = 0;
        for (size_t idx = left; idx < right; idx++) {
            const size_t j = idx - left;
            mx = (float)j / (float)(j + 1) * mx + tr[idx].first / (j + 1);
            my = (float)j / (float)(j + 1) * my + tr[idx].second / (j + 1);
        }
        double accum = 0.;
        for (size_t idx = left; idx < right; idx++) accum += (tr[idx].second - my) * (tr[idx].second - my);
        const float sd = std::sqrt(accum / bs);  // always / bs, also for the last partial bucket
        out_x[sz - 1 - b] = mx;
        out_y[sz - 1 - b] = my;
        out_std[sz - 1 - b] = sd;
    }
    *nbuckets = sz;
    API_END
}

int amd_ivf_merge_tables(int metric, size_t n, size_t k, size_t nshard, const float* all_D, const int64_t* all_I, float* D,
                         int64_t* I) {
    API_BEGIN
    if (k == 0) return 0;
    // IndexShards.cpp:44-105: per query, a heap over (distance, shard) of the shards' current heads.
    // L2 pops the smallest first, IP the largest; -1 ids end a shard's row; padding carries the
    // *heap's* neutral value (so -FLT_MAX for L2), as the reference does.
    const bool smallest_first = metric == METRIC_L2;
    auto before = [&](float a, float b) { return smallest_first ? a < b : a > b; };  // C::cmp
    const size_t stride = n * k;
    // (queries are independent: the reference runs this loop under "#pragma omp parallel for", IndexShards.cpp:56; here a few host
    // threads take ranges of queries when the tables are large -- eight shards x 10 000 queries are 1.5 ms on one thread, which
    // would be the slowest stage of the pipelined shards mode)
    auto merge_range = [&](size_t i_begin, size_t i_end) {
    std::vector<int> pointer(nshard), sid(nshard);
    std::vector<float> hv(nshard);
    for (size_t i = i_begin; i < i_end; i++) {
        const float* Din = all_D + i * k;
        const int64_t* Iin = all_I + i * k;
        size_t hs = 0;
        auto push = [&](float v, int s) {
            size_t c = ++hs;
            while (c > 1) {
                size_t f = c >> 1;
                if (!before(v, hv[f - 1])) break;
                hv[c - 1] = hv[f - 1];
                sid[c - 1] = sid[f - 1];
                c = f;
            }
            hv[c - 1] = v;
            sid[c - 1] = s;
        };
        auto pop = [&]() {
            size_t kk = hs--;
            float v = hv[kk - 1];
            size_t c = 1;
            for (;;) {
                size_t c1 = c << 1, c2 = c1 + 1;
                if (c1 > kk) break;
                if (c2 == kk + 1 || before(hv[c1 - 1], hv[c2 - 1])) {
                    if (before(v, hv[c1 - 1])) break;
                    hv[c - 1] = hv[c1 - 1];
                    sid[c - 1] = sid[c1 - 1];
                    c = c1;
                } else {
                    if (before(v, hv[c2 - 1])) break;
                    hv[c - 1] = hv[c2 - 1];
                    sid[c - 1] = sid[c2 - 1];
                    c = c2;
                }
            }
            hv[c - 1] = hv[kk - 1];
            sid[c - 1] = sid[kk - 1];
        };
        for (size_t s = 0; s < nshard; s++) {
            pointer[s] = 0;
            if (Iin[stride * s] >= 0) push(Din[stride * s], (int)s);
        }
        for (size_t j = 0; j < k; j++) {
            if (hs == 0) {
                I[i * k + j] = -1;
                D[i * k + j] = smallest_first ? -FLT_MAX : FLT_MAX;
            } else {
                int s = sid[0];
                int& p = pointer[s];
                D[i * k + j] = hv[0];
                I[i * k + j] = Iin[stride * s + p];
                pop();
                p++;
                if ((size_t)p < k && Iin[stride * s + p] >= 0) push(Din[stride * s + p], s);
            }
        }
    }
    };
    const size_t work = n * nshard * k;
    size_t nthreads = work >= ((size_t)1 << 18) ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 2)) : 1;
    if (const char* e = getenv("AUNCEL_AMD_MERGE_THREADS")) nthreads = std::max(1, atoi(e));
    if (nthreads <= 1) {
        merge_range(0, n);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthreads; t++) th.emplace_back(merge_range, n * t / nthreads, n * (t + 1) / nthreads);
        for (auto& t : th) t.join();
    }
    API_END
}

// Clustering::train over an IndexFlat (Clustering.cpp:75-226) with the training set resident on the device: every
// iteration's assignment runs through the coarse kernels, the centroid update through ivf_kmeans.hip (fp32 sums in point
// order); what is sequential in the reference -- permutations, the objective's running sum, void-cluster splitting -- stays
// on the host (kmeans_host.h).
int amd_ivf_kmeans(int d, size_t n, const float* x_in, size_t k, int metric, int niter, long seed, size_t max_points_per_centroid,
                   int spherical, int int_centroids, int coarse_mode, int device, float* centroids, float* obj) {
    API_BEGIN
    namespace km = amdivf_kmeans;
    if (d <= 0 || k == 0) throw EngineError("bad dimension / k");
    if (n < k) throw EngineError("Number of training points should be at least as large as number of clusters");
    const float* x = x_in;
    std::vector<float> sub;
    size_t nx = n;
    if (nx > k * max_points_per_centroid) {
        std::vector<int> perm(nx);
        km::rand_perm(perm.data(), nx, seed);
        nx = k * max_points_per_centroid;
        sub.resize(nx * (size_t)d);
        for (size_t i = 0; i < nx; i++) memcpy(&sub[i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
        x = sub.data();
    }
    if (nx == k) {  // the reference's corner case: the training set becomes the centroids
        memcpy(centroids, x_in, sizeof(float) * (size_t)d * k);
        return 0;
    }
    {
        std::vector<int> perm(nx);
        km::rand_perm(perm.data(), nx, seed + 1);
        for (size_t i = 0; i < k; i++) memcpy(centroids + i * d, x + (size_t)perm[i] * d, d * sizeof(float));
    }
    km::post_process(centroids, d, k, spherical != 0, int_centroids != 0);

    amd_ivf_t* raw = nullptr;
    if (amd_ivf_create(d, k, metric, device, &raw)) throw EngineError(g_last_error);
    std::unique_ptr<amd_ivf> h(raw);
    use_device(h.get());
    hipStream_t s = h->stream;
    const size_t dpad = h->dpad;
    h->d_resident.ensure(nx * dpad * sizeof(float));
    upload_rows(h.get(), h->d_resident.as<float>(), x, nx);
    IntRange qr;
    qr.add(x, nx * (size_t)d);
    DevBuf d_dis, d_keys, keys_in, keys_out, idx_in, idx_out, counts, seg, temp, d_cen;
    d_dis.ensure(nx * 4);
    d_keys.ensure(nx * 8);
    keys_in.ensure(nx * 4);
    keys_out.ensure(nx * 4);
    idx_in.ensure(nx * 4);
    idx_out.ensure(nx * 4);
    counts.ensure(k * 4);
    seg.ensure((k + 1) * 4);
    const size_t temp_bytes = kmeans_sort_temp_bytes(nx);
    temp.ensure(std::max<size_t>(temp_bytes, 16));
    d_cen.ensure(k * (size_t)d * 4);
    std::vector<float> dis(nx);
    std::vector<uint32_t> cnt(k), off(k + 1);
    std::vector<size_t> hassign(k);
    for (int it = 0; it < niter; it++) {
        if (amd_ivf_set_centroids(h.get(), centroids)) throw EngineError(g_last_error);
        coarse_dev(h.get(), h->d_resident.as<float>(), nx, 1, coarse_mode, d_dis.as<float>(), d_keys.as<int64_t>(),
                   h->allow_fused && h->centroid_range.fusable_with(qr, h->metric));
        HIP_CHECK(hipMemcpyAsync(dis.data(), d_dis.p, nx * 4, hipMemcpyDeviceToHost, s));
        launch_kmeans_group(d_keys.as<int64_t>(), nx, (uint32_t)k, keys_in.as<uint32_t>(), keys_out.as<uint32_t>(), idx_in.as<uint32_t>(),
                            idx_out.as<uint32_t>(), counts.as<uint32_t>(), temp.p, temp_bytes, s);
        HIP_CHECK(hipMemcpyAsync(cnt.data(), counts.p, k * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(stream_sync(s));
        float err = 0;  // the reference's running fp32 sum, in point order
        for (size_t j = 0; j < nx; j++) err += dis[j];
        if (obj) obj[it] = err;
        off[0] = 0;
        for (size_t c = 0; c < k; c++) {
            off[c + 1] = off[c] + cnt[c];
            hassign[c] = cnt[c];
        }
        HIP_CHECK(hipMemcpyAsync(seg.p, off.data(), (k + 1) * 4, hipMemcpyHostToDevice, s));
        launch_kmeans_sums(h->d_resident.as<float>(), dpad, d, idx_out.as<uint32_t>(), seg.as<uint32_t>(), (uint32_t)k, d_cen.as<float>(), s);
        HIP_CHECK(hipMemcpyAsync(centroids, d_cen.p, k * (size_t)d * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(stream_sync(s));
        km::split_void_clusters(centroids, hassign, d, k, nx);
        km::post_process(centroids, d, k, spherical != 0, int_centroids != 0);
    }
    API_END
}

int amd_ivf_scan_arith(amd_ivf_t* h) { return ix(h)->last_arith; }

int amd_ivf_set_byte_codes(amd_ivf_t* h, int enable) {
    h->allow_bytes = enable ? 1 : 0;
    for (amd_ivf* c : h->async_ctx) c->allow_bytes = h->allow_bytes;
    return 0;
}

static int opt_id(const char* key) {
    if (key)
        for (int i = 0; i < N_OPT; i++)
            if (!strcmp(key, OPT_TABLE[i].key)) return i;
    throw EngineError(std::string("unknown option: ") + (key ? key : "(null)"));
}
static double opt_default(OptId id) {
    switch (id) {
        case OPT_COARSE_TIES: case OPT_TIE_FIX: case OPT_PHASE_TIMING: return -1;
        case OPT_FIXED_ROUNDS: case OPT_ROUND_INC: case OPT_ROUND_GROW: return 0;  // (0: chosen per search)
        case OPT_ROUND_FIRST: return 12;
        case OPT_SCAN_PIPELINED: return 7;
        case OPT_FILTER: case OPT_COARSE_PICK: return 2;
        case OPT_ROW_LISTS: return ROW_LISTS_DEFAULT;
        default: return 1;
    }
}
int amd_ivf_set_option(amd_ivf_t* h, const char* key, double value) {
    API_BEGIN
    if (!h) throw EngineError("null handle");
    ix(h)->opt.v[opt_id(key)].store(std::isnan(value) ? OPT_UNSET : value, std::memory_order_relaxed);
    API_END
}
int amd_ivf_get_option(amd_ivf_t* h, const char* key, double* value) {
    API_BEGIN
    if (!h || !value) throw EngineError("null argument");
    const OptId id = (OptId)opt_id(key);
    *value = opt(h, id, opt_default(id));
    API_END
}

int amd_ivf_last_scan_min_bytes(amd_ivf_t* h, double* bytes) {
    *bytes = h->last_min_bytes;
    return 0;
}

int amd_ivf_last_round_hints(amd_ivf_t* h, uint64_t out[2]) {
    out[0] = h->hinted_rounds;
    out[1] = h->short_rounds;
    for (auto& kid : h->kids) {
        out[0] += kid->hinted_rounds;
        out[1] += kid->short_rounds;
    }
    return 0;
}

int amd_ivf_last_tie_redone(amd_ivf_t* h, uint64_t* queries) {
    *queries = h->last_tie_redone;
    return 0;
}

int amd_ivf_last_tie_patched(amd_ivf_t* h, uint64_t* rankings) {
    *rankings = h->last_tie_patched;
    return 0;
}

int amd_ivf_last_tie_fixed(amd_ivf_t* h, uint64_t* queries) {
    API_BEGIN
    use_device(h);
    *queries = 0;
    auto add = [&](amd_ivf* c) {
        if (!c->w_tie_flag.p || !c->last_state_n) return;
        std::vector<uint32_t> f(c->last_state_n);
        HIP_CHECK(stream_sync(c->stream));
        HIP_CHECK(hipMemcpy(f.data(), c->w_tie_flag.p, f.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t v : f) *queries += v == 2;
    };
    add(h);
    for (auto& kid : h->kids) add(kid.get());
    API_END
}
int amd_ivf_last_direct_out(amd_ivf_t* h) { return h ? h->last_direct_out : 0; }
int amd_ivf_last_coarse_pick(amd_ivf_t* h, uint64_t* rankings) {
    *rankings = h ? h->coarse_picked : 0;
    return 0;
}
int amd_ivf_last_filter(amd_ivf_t* h, uint64_t out[2]) {
    API_BEGIN
    use_device(h);
    out[0] = h->filter_launches;
    out[1] = 0;
    if (h->filter_launches && h->w_surv_cnt.p) {
        uint32_t v = 0;
        HIP_CHECK(stream_sync(h->stream));
        HIP_CHECK(hipMemcpy(&v, h->w_surv_cnt.p, 4, hipMemcpyDeviceToHost));
        out[1] = v;
    }
    API_END
}
int amd_ivf_coarse_tie_rows(amd_ivf_t* h, uint64_t* rows) {
    API_BEGIN
    use_device(h);
    *rows = 0;
    auto add = [&](amd_ivf* c) {
        *rows += c->tie_rows_host;
        if (!c->w_tie_rows.p) return;
        uint64_t v = 0;
        HIP_CHECK(stream_sync(c->stream));
        HIP_CHECK(hipMemcpy(&v, c->w_tie_rows.p, 8, hipMemcpyDeviceToHost));
        *rows += v;
    };
    add(h);
    for (auto& kid : h->kids) add(kid.get());  // the slices of an adaptive batch run on these
    API_END
}

int amd_ivf_last_timing(amd_ivf_t* h, double out[8]);
int amd_ivf_last_scan_min_bytes(amd_ivf_t* h, double* bytes);
int amd_ivf_last_round_hints(amd_ivf_t* h, uint64_t out[2]);
int amd_ivf_last_tie_redone(amd_ivf_t* h, uint64_t* queries);
int amd_ivf_last_direct_out(amd_ivf_t* h);

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Asynchronous searches.  One synchronous call leaves the GPU to one search at a time, whose selection and planning phases are
// latency-bound; several searches in flight fill those phases with each other's scans (1.6 x the throughput on the bench
// workload).  A caller with more than one batch at hand gets that without threads of its own: submit returns a ticket at
// once, the search runs on one of the handle's internal contexts (amd_ivf_clone: own stream and workspaces, the owner's index
// data and resident queries), wait returns its status.  At most `depth` searches run at a time; further tickets queue.
// what amd_ivf_submit_adaptive was asked for, kept next to the job so that a worker can see whether the ticket behind it in the queue
// continues it (async_take_group)
struct AdaptiveSpec {
    size_t start = 0, n = 0, query_topk = 0;
    float multipler = 0.f, std_m = 0.f;
    const float* require_acc = nullptr;
    const float* gt_D = nullptr;
    int profile = 0, coarse_mode = 0;
    uint64_t* my_nprobe = nullptr;
    float* t_recalls = nullptr;
    float* D = nullptr;
    int64_t* I = nullptr;
};
struct AsyncJob {
    std::function<int(amd_ivf_t*)> run;
    bool adaptive = false;
    AdaptiveSpec spec;
    uint32_t coalesced = 1;  // tickets the pass that served this one served
    int rc = 0;
    bool done = false;
    std::string error;
    double timing[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // amd_ivf_last_timing's eight + amd_ivf_last_scan_min_bytes
    uint64_t diag[4] = {0, 0, 0, 0};  // launches sized by a hint | hints too small | queries searched again (redo) | direct out
};
struct AsyncPool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    int running = 0, running_limit = 0;  // searches at a time: never more than the hardware queues of a priority class
    uint64_t served_tickets = 0, served_passes = 0;
    const struct amd_ivf* owner = nullptr;
    std::deque<AsyncJob*> queue;
    std::map<uint64_t, std::unique_ptr<AsyncJob>> jobs;
    std::vector<std::thread> workers;
    std::vector<amd_ivf_t*> ctx;
    uint64_t next = 1;
    bool stop = false;
};

// searches the pool runs at a time: its depth, cut to the hardware queues of a class (async_pool) and -- where the owner's lists have
// no byte codes, i.e. its large searches are fp32 searches -- to the option fp32_in_flight, so that no worker parks inside a search
static int async_running_limit(const AsyncPool* p) {
    int lim = p->running_limit;
    const amd_ivf* o = p->owner;
    static const bool no_cap = getenv("AUNCEL_AMD_NO_POOL_FP32_CAP") != nullptr;  // (experiments: the gate inside the searches only)
    if (!no_cap && o && !o->lists_dirty && !(o->have_codes8 && o->allow_bytes)) {
        const int f = (int)o->opt.get(OPT_FP32_IN_FLIGHT, 4);
        if (f > 0) lim = std::min(lim, f);
    }
    return std::max(lim, 1);
}

// Tickets of amd_ivf_submit_adaptive that ONE pass can serve: the queue's next ticket continues the group when it asks for the same
// search (parameters, require_acc / ground-truth arrays) over the resident queries right behind the group's, and its result buffers
// lie right behind the group's in memory -- the group then is one search of the joined range into one buffer.  A pass over the
// lists costs its bytes whatever the number of queries probing them (the scans are bound by the list stream), so two queued batches
// of 5000 cost little more than one; every query's result is the reference's whatever batch it travels in (the parity suites).
// Option "coalesce": tickets a group may hold (1: off).  The queue is only looked at, never waited on: a ticket without a
// successor in the queue runs alone.
static bool continues_group(const AdaptiveSpec& g, size_t g_n, const AdaptiveSpec& s, size_t K) {
    return K > 0 && g.n > 0 && s.query_topk == g.query_topk && s.multipler == g.multipler && s.std_m == g.std_m && s.require_acc == g.require_acc &&
           s.gt_D == g.gt_D && s.profile == g.profile && s.coarse_mode == g.coarse_mode && s.n == g.n && s.start == g.start + g_n &&
           s.D == g.D + g_n * K && s.I == g.I + g_n * K && s.my_nprobe && s.t_recalls && g.my_nprobe && g.t_recalls;
}
static int run_adaptive_group(amd_ivf_t* c, const std::vector<AsyncJob*>& group) {
    const AdaptiveSpec& g = group[0]->spec;
    if (group.size() == 1)
        return amd_ivf_search_adaptive(c, g.start, g.n, g.query_topk, g.multipler, g.std_m, g.require_acc, g.gt_D, g.profile, g.coarse_mode, g.my_nprobe,
                                       g.t_recalls, g.D, g.I);
    // my_nprobe / t_recalls are per-ticket arrays indexed by absolute query id: the joined search reads and writes a scratch pair that
    // holds every ticket's own entries of its own range, and each ticket gets its range back
    const size_t total = g.n * group.size(), end = g.start + total;
    std::vector<uint64_t> np(end, 0);
    std::vector<float> tr(end, 0.f);
    for (const AsyncJob* j : group) {
        std::copy(j->spec.my_nprobe + j->spec.start, j->spec.my_nprobe + j->spec.start + j->spec.n, np.begin() + j->spec.start);
        std::copy(j->spec.t_recalls + j->spec.start, j->spec.t_recalls + j->spec.start + j->spec.n, tr.begin() + j->spec.start);
    }
    const int rc = amd_ivf_search_adaptive(c, g.start, total, g.query_topk, g.multipler, g.std_m, g.require_acc, g.gt_D, g.profile, g.coarse_mode, np.data(),
                                           tr.data(), g.D, g.I);
    for (const AsyncJob* j : group) {
        std::copy(np.begin() + j->spec.start, np.begin() + j->spec.start + j->spec.n, j->spec.my_nprobe + j->spec.start);
        std::copy(tr.begin() + j->spec.start, tr.begin() + j->spec.start + j->spec.n, j->spec.t_recalls + j->spec.start);
    }
    return rc;
}

static void async_worker(AsyncPool* p, size_t i) {
    for (;;) {
        std::vector<AsyncJob*> group;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->stop || (!p->queue.empty() && p->running < async_running_limit(p)); });
            if (p->stop || p->queue.empty()) return;  // (the handle is going away: searches not yet started are dropped)
            group.push_back(p->queue.front());
            p->queue.pop_front();
            if (group[0]->adaptive && p->owner) {
                const size_t limit = (size_t)std::max(1.0, p->owner->opt.get(OPT_COALESCE, 1));
                const size_t K = p->owner->tuner_max_topk;
                while (group.size() < limit && !p->queue.empty() && p->queue.front()->adaptive &&
                       continues_group(group[0]->spec, group[0]->spec.n * group.size(), p->queue.front()->spec, K)) {
                    group.push_back(p->queue.front());
                    p->queue.pop_front();
                }
            }
            p->running++;
        }
        amd_ivf_t* c = p->ctx[i];
        int rc;
        std::string err;
        try {
            rc = group[0]->adaptive ? run_adaptive_group(c, group) : group[0]->run(c);
            if (rc) err = amd_ivf_last_error();
        } catch (const std::exception& e) {  // (the entry points catch their own: this is the group's scratch memory)
            rc = -4;
            err = e.what();
        }
        double timing[9];
        uint64_t diag[4];
        amd_ivf_last_timing(c, timing);
        amd_ivf_last_scan_min_bytes(c, &timing[8]);
        amd_ivf_last_round_hints(c, diag);
        amd_ivf_last_tie_redone(c, &diag[2]);
        diag[3] = (uint64_t)amd_ivf_last_direct_out(c);
        // (the pass's kernel times, bytes and launch counts are shared out evenly among the tickets it served -- sums over tickets stay
        // sums over passes; its wall time, slot efficiency and round count are every ticket's)
        const double share = 1.0 / (double)group.size();
        {
            std::lock_guard<std::mutex> lk(p->mu);
            for (size_t t = 0; t < group.size(); t++) {
                AsyncJob* j = group[t];
                for (int f = 0; f < 9; f++) j->timing[f] = (f == 3 || f == 6 || f == 7) ? timing[f] : timing[f] * share;
                for (int f = 0; f < 4; f++) j->diag[f] = f == 3 ? diag[f] : (t == 0 ? diag[f] : 0);
                j->coalesced = (uint32_t)group.size();
                j->error = err;
                j->rc = rc;
                j->done = true;
            }
            p->running--;
            p->served_tickets += group.size();
            p->served_passes++;
        }
        p->cv_done.notify_all();
        p->cv_job.notify_one();
    }
}

static AsyncPool* async_pool(amd_ivf* h) {
    // (two threads submitting on one handle for the first time must not both build a pool: ADVICE round 3; a mutex of its own --
    // amd_ivf_clone below takes the upload mutex)
    std::lock_guard<std::mutex> lock(h->async_mu);
    if (h->async) return h->async;
    std::unique_ptr<AsyncPool> p(new AsyncPool);
    const int depth = std::min(std::max(h->async_depth, 1), 16);
    // Every search context has one stream in each priority class and wants a hardware queue to itself in each (ensure_context_streams):
    // where the runtime was started with fewer queues a class than searches asked for, two searches' kernels would wait for each other
    // in a shared queue (2.0 instead of 3.0 M q/s, by luck of the order).  Then only as many run at a time as there are queues; the
    // other tickets wait their turn, the caller's code is the same.
    const int hwq = hw_queues_per_class();
    p->owner = h;
    p->running_limit = std::min(depth, std::max(hwq, 1));
    if (p->running_limit < depth && !getenv("AUNCEL_AMD_QUIET"))
        fprintf(stderr, "[auncel_amd] %d searches in flight asked for, the HIP runtime has %d hardware queues a priority class: %d run at a time "
                        "(export GPU_MAX_HW_QUEUES=8 before the first GPU call: include/auncel_amd.h)\n", depth, hwq, p->running_limit);
    for (int i = 0; i < depth; i++) {
        amd_ivf_t* c = nullptr;
        if (amd_ivf_clone(h, &c) != 0) {
            for (amd_ivf_t* made : p->ctx) amd_ivf_destroy(made);
            throw std::runtime_error(std::string("asynchronous search contexts: ") + amd_ivf_last_error());
        }
        p->ctx.push_back(c);
    }
    for (int i = 0; i < depth; i++) p->workers.emplace_back(async_worker, p.get(), (size_t)i);
    h->async_ctx.assign(p->ctx.begin(), p->ctx.end());
    h->async = p.release();
    return h->async;
}

static void async_shutdown(amd_ivf* h) {
    AsyncPool* p = h->async;
    if (!p) return;
    h->async_served[0] += p->served_tickets, h->async_served[1] += p->served_passes;
    h->async = nullptr;
    h->async_ctx.clear();
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_job.notify_all();
    for (auto& t : p->workers) t.join();  // (searches that are running finish; nobody can wait for the queued ones any more)
    for (amd_ivf_t* c : p->ctx) amd_ivf_destroy(c);
    delete p;
}

static uint64_t async_enqueue(amd_ivf* h, std::function<int(amd_ivf_t*)> run, const AdaptiveSpec* spec = nullptr) {
    AsyncPool* p = async_pool(h);
    std::unique_ptr<AsyncJob> j(new AsyncJob);
    j->run = std::move(run);
    if (spec) {
        j->adaptive = true;
        j->spec = *spec;
    }
    uint64_t id;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        id = p->next++;
        p->queue.push_back(j.get());
        p->jobs[id] = std::move(j);
    }
    p->cv_job.notify_one();
    return id;
}

extern "C" {

int amd_ivf_set_async_depth(amd_ivf_t* h, int depth) {
    API_BEGIN
    OWNER_ONLY(h);
    if (depth == 0) {  // give the internal contexts (streams, workspaces) and their threads back
        if (h->async) {
            std::lock_guard<std::mutex> lk(h->async->mu);
            if (!h->async->jobs.empty()) throw EngineError("tickets are still out");
        }
        async_shutdown(h);
        return 0;
    }
    if (depth < 1 || depth > 16) throw EngineError("asynchronous depth must be 1..16");
    if (h->async && depth != (int)h->async->ctx.size()) throw EngineError("asynchronous depth is fixed by the first submit");
    h->async_depth = depth;
    API_END
}

int amd_ivf_submit_adaptive(amd_ivf_t* h, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                            const float* require_acc, const float* gt_D, int profile, int coarse_mode, uint64_t* my_nprobe,
                            float* t_recalls, float* D, int64_t* I, uint64_t* ticket) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    AdaptiveSpec spec;
    spec.start = start, spec.n = n, spec.query_topk = query_topk, spec.multipler = multipler, spec.std_m = std_m;
    spec.require_acc = require_acc, spec.gt_D = gt_D, spec.profile = profile, spec.coarse_mode = coarse_mode;
    spec.my_nprobe = my_nprobe, spec.t_recalls = t_recalls, spec.D = D, spec.I = I;
    *ticket = async_enqueue(h, nullptr, &spec);
    API_END
}

int amd_ivf_submit_search_resident(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, int coarse_mode, float* D, int64_t* I,
                                   uint64_t* ticket) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    *ticket = async_enqueue(h, [=](amd_ivf_t* c) { return amd_ivf_search_resident(c, start, n, k, nprobe, coarse_mode, D, I); });
    API_END
}

int amd_ivf_submit_coarse_resident(amd_ivf_t* h, size_t start, size_t n, size_t nprobe, float* coarse_dis, int64_t* keys, int mode,
                                   uint64_t* ticket) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    *ticket = async_enqueue(h, [=](amd_ivf_t* c) { return amd_ivf_coarse_resident(c, start, n, nprobe, coarse_dis, keys, mode); });
    API_END
}

int amd_ivf_submit_search_resident_preassigned(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const int64_t* keys,
                                               float* D, int64_t* I, uint64_t* ticket) {
    API_BEGIN
    OWNER_ONLY(h);
    use_device(h);
    if (start + n > h->n_resident) throw EngineError("resident query range out of bounds");
    *ticket = async_enqueue(h, [=](amd_ivf_t* c) { return amd_ivf_search_resident_preassigned(c, start, n, k, nprobe, keys, D, I); });
    API_END
}

int amd_ivf_async_counts(amd_ivf_t* h, uint64_t out[2]) {
    out[0] = out[1] = 0;
    if (h && h->async) {
        std::lock_guard<std::mutex> lk(h->async->mu);
        out[0] = h->async->served_tickets + h->async_served[0];
        out[1] = h->async->served_passes + h->async_served[1];
    } else if (h) {
        out[0] = h->async_served[0], out[1] = h->async_served[1];
    }
    return 0;
}

int amd_ivf_wait(amd_ivf_t* h, uint64_t ticket, double timing[9], uint64_t diag[4]) {
    int rc = 0;
    try {
        if (!h || !h->async) throw EngineError("no asynchronous search was submitted on this handle");
        AsyncPool* p = h->async;
        std::unique_ptr<AsyncJob> j;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            auto it = p->jobs.find(ticket);
            if (it == p->jobs.end()) throw EngineError("unknown ticket");
            AsyncJob* raw = it->second.get();
            p->cv_done.wait(lk, [&] { return raw->done; });
            j = std::move(it->second);
            p->jobs.erase(it);
        }
        if (timing) std::copy(j->timing, j->timing + 9, timing);
        if (diag) std::copy(j->diag, j->diag + 4, diag);
        rc = j->rc;
        if (rc) g_last_error = j->error;
    } catch (const EngineError& e) {
        g_last_error = e.what();
        return -2;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -4;
    }
    return rc;
}

int amd_ivf_last_timing(amd_ivf_t* h, double out[8]) {
    for (int i = 0; i < 8; i++) out[i] = h->timing[i];
    return 0;
}
int amd_ivf_last_timing_detail(amd_ivf_t* h, double out[16]) {
    for (int i = 0; i < 2 * NCAT + 2; i++) out[i] = h->timing_detail[i];
    return 0;
}

}  // extern "C"
