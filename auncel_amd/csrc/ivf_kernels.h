// Device-side data structures and kernel launch wrappers of the gfx950 IVF-Flat engine.
// (internal: the public boundary is include/auncel_amd.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <stdexcept>
#include <string>

namespace amdivf {

// A launch the runtime rejects (grid, LDS or register limits) leaves nothing on the stream, and the later synchronisation
// succeeds: without this check a search would return untouched output buffers as if they were results.
inline void check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("kernel launch failed: ") + what + ": " + hipGetErrorString(e));
    // AUNCEL_AMD_SYNC_LAUNCH=1 (debugging): name every launch and wait for it, so that a device fault points at its kernel
    static const bool sync_each = getenv("AUNCEL_AMD_SYNC_LAUNCH") != nullptr;
    if (sync_each) {
        fprintf(stderr, "[launch] %.100s\n", what);
        const hipError_t s = hipDeviceSynchronize();
        if (s != hipSuccess) throw std::runtime_error(std::string("kernel failed: ") + what + ": " + hipGetErrorString(s));
    }
}
// (the error state is per thread and sticky: an unrelated earlier call, e.g. the elapsed time of an event pair that was
// never recorded, must not be taken for this launch's)
#define LAUNCH(...)                     \
    do {                                \
        (void)hipGetLastError();        \
        hipLaunchKernelGGL(__VA_ARGS__); \
        ::amdivf::check_launch(#__VA_ARGS__); \
    } while (0)

constexpr int METRIC_IP = 0;
constexpr int METRIC_L2 = 1;

// ---------------------------------------------------------------------------- distance tiles
// One workgroup (256 threads = 4 waves) computes the distances between up to QG*RQ queries and
// up to (4/QG)*64*RV consecutive vectors of one inverted list (or of the centroid table).
constexpr int SCAN_RQ = 8;    // queries per wave (query operands live in SGPRs)
constexpr int SCAN_RV = 2;    // vectors per lane
#ifndef AUNCEL_SCAN_DC
#define AUNCEL_SCAN_DC 16
#endif
constexpr int SCAN_DC = AUNCEL_SCAN_DC;   // dimensions staged through LDS per step
constexpr int SCAN_WAVE_VECS = 64 * SCAN_RV;

// fp32 tiles: vec_base carries two numbers -- bits 0..39 the global index of the tile's first vector (into `codes`), bits 40..63
// the tile's first 64-vector block in the lane-ordered copy of the lists (ScanArgs::lanes; 0 where there is none)
constexpr int SCAN_VB_BITS = 40;
constexpr uint64_t SCAN_VB_MASK = (1ull << SCAN_VB_BITS) - 1;
inline __host__ __device__ uint64_t scan_vec_base(uint64_t first_vector, uint64_t first_block32_of_list, uint32_t pos_in_list) {
    return first_vector | (((first_block32_of_list >> 1) + pos_in_list / 64u) << SCAN_VB_BITS);
}

struct ScanItem {
    uint64_t vec_base;   // global index (into codes) of the tile's first vector (fp32 tiles: see scan_vec_base)
    uint32_t nvec;       // vectors in the tile
    uint32_t vec_off;    // position of the tile's first vector inside its list
    uint32_t pair_begin; // first entry of this tile in the pair arrays
    uint32_t npair;      // queries in the tile (<= qg * SCAN_RQ)
    uint32_t qg;         // query groups per workgroup: 1, 2, 4 or 8
    uint32_t qgroup;     // index of the tile's first 8-query group in the packed query tiles
};

struct ScanArgs {
    const float* codes;       // vectors, row-major, row stride d floats (d % 4 == 0)
    const float* queries;     // query matrix, row stride d floats
    const ScanItem* items;
    const uint32_t* pair_query;  // query row of each pair
    const uint64_t* pair_out;    // offset of each pair's distance row in `dist`
    float* dist;
    const float* qtile;       // packed query operands: [group][d/4][8 queries][4 floats] (pack_queries)
    int d;
    int metric;
    int fused;  // 1: fma(t, t, acc) -- only when the operands make it bit-identical to mul + add (see engine)
    // threshold mode (non-null thr): store only distances that beat thr[query row] and write one mask bit per
    // candidate (mask[i / 64] covers dist[i .. i + 63]); rows of `dist` start on multiples of 64
    const float* thr;
    unsigned long long* mask;
    int xcd_chunks;     // 1: XCD x (workgroup id % 8) takes the x-th eighth of the item list (consecutive items share an L2)
    uint32_t nitems;    // items of the shape being launched (set by launch_scan)
    // Device-chained rounds: the item counts of the four shapes are only known on the device (PlanArgs::counters); the launch
    // is then a fixed grid of resident workgroups that walk the items of their shape.  dev_counts = that counter array.
    const uint32_t* dev_counts;
    // ... sized by what the same round needed last time (0: no idea, one resident grid): the grid is a hint, never an input to
    // correctness -- workgroups stride over the true count -- and a right-sized grid lets the dispatcher interleave the
    // workgroups of several search contexts where resident grids would run one context's launch to its end first.
    uint32_t hint_qg[4];
    int hint_valid;     // hint_qg is what the same round of the previous search needed (a 0 there: probably no item of the shape again)
    // the lists once more in LANE ORDER (launch_lanes_from_f32), or null: 64-vector blocks (lists padded like the fragment copies:
    // block_off / 2), block b = d/4 pieces of 1 KiB, piece s = lane l's elements 4s .. 4s+3 of vector 64 b + l.  A wave streams a
    // block with one fully coalesced 16-byte-a-lane load per piece: scan_lanes_kernel needs no LDS staging and no barrier.
    const float* lanes;
};

// items grouped by qg (1, then 2, 4, 8); n_qg = item count of each group
// the shapes are independent: s2 / s1 / s4 (optional) let the qg 2 / 1 / 4 launches run beside the qg 8 one
void launch_scan(const ScanArgs& a, const size_t n_qg[4], hipStream_t s, hipStream_t s2 = nullptr, hipStream_t s1 = nullptr,
                 hipStream_t s4 = nullptr);
// fp32 lists (CSR rows, row stride dpad) -> lane order; nblocks64 = block_off[nlist] / 2
void launch_lanes_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks64, int dpad,
                           float* out, hipStream_t s);
// counters of PlanArgs that the chained launches read
constexpr int CNT_ACTIVE = 0, CNT_SEGMENTS = 1, CNT_PAIRS = 2, CNT_GROUPS = 3, CNT_QG1 = 4, CNT_QG2 = 5, CNT_QG4 = 8, CNT_QG8 = 9;
unsigned resident_grid(unsigned workgroups_per_cu);  // CUs of the current device x workgroups_per_cu

// gather + interleave the query rows of every group of (up to) 8 pairs: group g holds pairs
// [group_p0[g], group_p0[g] + group_cnt[g]); missing slots are zero
void launch_pack_queries(const float* queries, const uint32_t* pair_query, const uint32_t* group_p0, const uint32_t* group_cnt,
                         size_t ngroups, int d, float* qtile, hipStream_t s, const uint32_t* dev_ngroups = nullptr, uint32_t hint = 0);

// ---------------------------------------------------------------------------- byte-code scan on the i8 matrix cores
// When lists and queries hold integers 0..255 (and d * max^2 <= 2^24, so that every fp32 partial sum of the reference is an
// exact integer whatever the order) the distance tiles are integer contractions: v_mfma_i32_32x32x32_i8 on bytes re-centred
// to signed (u - 128), L2 = |xs|^2 + |ys|^2 - 2 xs.ys (translation invariant), IP = xs.ys + 128 sum(x) + 128 sum(y) - 16384 d.
//
// Storage of the lists for that kernel ("fragment order"): every list is cut into blocks of 32 vectors (lists padded to an
// even number of blocks); a block is ks x 64 x 16 bytes, ks = ceil(d / 32): 16-byte piece (s, lane) holds bytes
// [h * 16 ks + 16 s, + 16) of vector (lane & 31), h = lane >> 5 -- exactly the B operand of K-step s, so a wave fetches an
// operand with one fully coalesced 1-KiB load and no LDS staging.  Padding vectors / dimensions are signed zeros.
// code_cy: one int32 per stored slot (block * 32 + vector): |ys|^2 (L2) or 128 sum(y) (IP).
constexpr uint32_t MFMA_BLOCK = 32;        // vectors per block = N of the MFMA
constexpr uint32_t MFMA_QBLOCK = 32;       // queries per item = M of the MFMA
uint32_t mfma_chunk();                     // vectors per work item (a multiple of 64; AUNCEL_AMD_MFMA_CHUNK, default 256)
uint32_t mfma_chunk_thr();                 // ... of a threshold round (AUNCEL_AMD_MFMA_CHUNK_THR, default 512)
inline __host__ __device__ uint32_t mfma_ksteps(int d) { return (uint32_t)(d + 31) / 32; }
inline __host__ __device__ uint64_t mfma_list_blocks(uint64_t size) { return ((size + 63) / 64) * 2; }

struct MfmaScanArgs {
    const uint8_t* codes_frag;    // lists in fragment order
    const int32_t* code_cy;       // per stored slot
    const int8_t* queries8;       // signed query bytes, row stride ks * 32
    const int32_t* query_cx;      // per query row: |xs|^2 (L2) or 128 sum(x) - 16384 d (IP)
    const ScanItem* items;        // vec_base = first 32-block of the chunk (global block number), nvec = vectors of the chunk,
                                  // vec_off = position of the chunk in its list, pair_begin / npair (<= 32) = its queries
    const uint32_t* pair_query;
    const uint64_t* pair_out;
    float* dist;
    const float* thr;             // threshold mode (see ScanArgs)
    unsigned long long* mask;
    int d;
    int metric;
    int xcd_chunks;
    uint32_t nitems;
    const uint32_t* dev_nitems;   // chained rounds: the item count lives on the device; the grid is a hint (see ScanArgs)
    uint32_t hint_nitems;         // items the same round had last time (0: unknown -> a resident grid)
    int debug;                    // timing experiments only (results are wrong): 1 no mask stores, 2 no distance stores, 4 no epilogue,
                                  // 8 no contraction (threshold rounds)
    int exact_mask;               // threshold mode: the mask bits are counted as results (range search), not re-tested by a selection
    int pipelined;                // bit 0: dense rounds, bit 1: threshold rounds through scan_mfma_thr_kernel (two blocks in flight
                                  // per wave; threshold rounds: threshold folded into the accumulator, the mask a superset of
                                  // the exact one -- where exact_mask allows it); bit 2: threshold rounds with up to 64
                                  // queries per item through scan_mfma_pair_kernel (mfma_thr_qblock)
};
void launch_scan_mfma(const MfmaScanArgs& a, hipStream_t s);
uint32_t mfma_thr_qblock(int d, int pipelined, bool exact_mask);  // queries per item the planner gives a threshold round (32 | 64)
// fp32 lists (CSR rows, row stride dpad floats, integers 0..255) -> fragment order + code_cy; block_off[l] = first block of list l
void launch_frag_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                          int dpad, int metric, uint8_t* out, int32_t* cy, hipStream_t s);
// fp32 query rows (integers 0..255) -> signed byte rows (row stride ks * 32, zero padded) + query_cx
void launch_sbytes_from_f32(const float* x, size_t n, int d, int dpad, int metric, int8_t* out, int32_t* cx, hipStream_t s);
// ---------------------------------------------------------------------------- fp32 filter + exact rescoring (ivf_filter.hip)
// pieces of 8 dimensions per vector in the fragment-ordered fp32 copy: d / 8 rounded up to the step counts the kernel is built
// for (4, 8, 12, 16: the query operand stays in registers) or as it is beyond 128 dimensions
inline __host__ __device__ uint32_t filter_steps(int d) {
    const uint32_t j = (uint32_t)(d + 7) / 8;
    return j <= 4 ? 4u : j <= 8 ? 8u : j <= 12 ? 12u : j <= 16 ? 16u : (j + 1u) & ~1u;  // (beyond 128 dimensions: an even count)
}
struct FilterScanArgs {
    const float* codes_frag;      // lists in fragment order (launch_frag32_from_f32)
    const float* yn;              // per stored slot: |y|^2 (L2) / |y| (IP)
    const float* xf;              // packed queries, row stride 8 filter_steps(d) (launch_filter_queries)
    const float* xn;              // per query row: |x|^2 (L2) / |x| (IP)
    const float* codes;           // the lists as stored (CSR rows, stride dpad): the exact distance is computed from these
    const float* queries;         // the query rows as passed (stride dpad)
    const ScanItem* items;        // as for scan_mfma_kernel; qgroup = global index of the chunk's first vector
    const uint32_t* pair_query;
    const uint64_t* pair_out;
    float* dist;
    const float* thr;
    unsigned long long* mask;
    uint4* surv;                  // survivors of the filter: (distance row position, query row, vector, -)
    uint32_t* surv_count;
    uint32_t surv_cap;
    int d, dpad;
    int metric;
    int xcd_chunks;
    uint32_t nitems;
    const uint32_t* dev_nitems;
    uint32_t hint_nitems;
    int half;                     // 1: codes_frag / xf hold scaled fp16 values in the fp16 fragment order (launch_frag16_from_f32,
                                  // launch_filter_queries16) and `params` the scales and the error constant that go with them
    const float* params;          // half: FilterParams on the device (written by launch_filter_queries16)
};
// what the fp16 form of the filter needs besides its operands (device memory, one per search): the dot product of the scaled
// halves times `ps` is x.y up to C16 (|x|^2 + |y|^2) / 2; C = the constant of the keep test (ivf_filter.hip)
struct FilterParams {
    float sx, ps, C, pad;
};
void launch_scan_filter(const FilterScanArgs& a, hipStream_t s);  // filter + rescoring, two launches
// pieces of 16 dimensions per vector in the fp16 fragment order: the step counts the one-wave kernel is built for (4, 6, 8) or an
// even count beyond 128 dimensions
inline __host__ __device__ uint32_t filter_steps16(int d) {
    const uint32_t j = (uint32_t)(d + 15) / 16;
    return j <= 4 ? 4u : j <= 6 ? 6u : j <= 8 ? 8u : (j + 1u) & ~1u;
}
// range of a row-major matrix for the fp16 form: info[0] = max |v| (bits of a non-negative float; +inf if a NaN or an infinity was
// seen), info[1] != 0: some element is not an integer of magnitude <= 2048 (fp16 holds those exactly), info[2] = 0x7f800000 - bits
// of the smallest row maximum among rows that are not all zero; info (4 words) zeroed by the caller
void launch_amax(const float* x, size_t rows, int stride, uint32_t* info, hipStream_t s);
// the scale that goes with such a range (0: none usable) -- ivf_filter.hip
float filter_half_scale(const uint32_t info[4], int d);
// FilterParams from the ranges of two matrices (sx, ps, C as above; pad = the second matrix's scale)
void launch_half_params(const uint32_t* qinfo, const uint32_t* yinfo, int d, FilterParams* params, hipStream_t s);
// fp32 lists -> fp16 fragment order, scaled by the power of two that brings info[0] into [2^14, 2^15); yn as launch_frag32_from_f32
void launch_frag16_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                            int dpad, int metric, const uint32_t* info, float* out, float* yn, hipStream_t s);
// query rows -> scaled fp16 rows (stride 16 filter_steps16(d) halves) + xn + the search's FilterParams
void launch_filter_queries16(const float* x, size_t n, int d, int dpad, int metric, const uint32_t* qinfo, const uint32_t* yinfo, float* xf,
                             float* xn, FilterParams* params, hipStream_t s);
// beyond 128 dimensions (the query operand no longer fits a wave's registers) an item of the filter is up to this many queries, computed by a whole workgroup (scan_filter_wide_kernel); AUNCEL_AMD_FILTER_NARROW=1 keeps the one-wave form
constexpr uint32_t FILTER_WIDE_QUERIES = 128;
constexpr uint32_t FILTER_WIDE_VECTORS = 128;  // ... x a chunk of this many vectors (one 32-vector block per wave)
uint32_t filter_item_queries(int d);
uint32_t filter_item_vectors(int d);
void launch_frag32_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                            int dpad, int metric, float* out, float* yn, hipStream_t s);
void launch_filter_queries(const float* x, size_t n, int d, int dpad, int metric, float* xf, float* xn, hipStream_t s);

inline __host__ __device__ int scan_qg_class(uint32_t qg) { return qg == 1 ? 0 : qg == 2 ? 1 : qg == 4 ? 2 : 3; }
// The queries of a list go into blocks of `qblock` (64: 8-wave tiles, the byte-code scan, which is short of HBM and
// issue slots rather than of VALU; 32: 4-wave tiles, the fp32 scans); the last block takes the narrowest shape that holds it.
constexpr uint32_t SCAN_QBLOCK = 8 * SCAN_RQ;  // the largest block
inline __host__ __device__ uint32_t scan_shape_of(uint32_t r) { return r <= SCAN_RQ ? 1u : r <= 2 * SCAN_RQ ? 2u : r <= 4 * SCAN_RQ ? 4u : 8u; }
inline __host__ __device__ uint32_t scan_tile_vecs(uint32_t qg) { return (qg >= 4 || qg == 1 ? 1u : 4u / qg) * SCAN_WAVE_VECS; }  // (ScanShape)

// ---------------------------------------------------------------------------- ordered selection
// One wave per query replays the reference's sequential heap (Heap.h) over the distance rows in
// probe order, evaluates the Auncel stop rule after every probe, and collects training samples.
struct TunerDev {
    int enabled;
    int profile;
    int overhead;   // error_pro::overhead_profile: the rule runs on every probe, its verdict is ignored, the loop ends at nlist / 8
    uint32_t max_topk, query_topk, ntraces;
    float multipler, std_m;
    const float* interdis;     // packed upper-triangular centroid table
    const float* arcos;        // 500-entry LUT
    const uint32_t* trace_off; // ntraces + 1
    const float *trace_x, *trace_y, *trace_std;
    const float* require_acc;  // by absolute query id
    const float* gt_D;         // by absolute query id, k wide (may be null)
    unsigned long long* my_nprobe;  // by absolute query id
    float* t_recalls;               // by absolute query id
};

struct TrainDev {
    int enabled;
    uint32_t ntraces;
    const float* interdis;
    const float* arcos;
    const float* gt_D;    // by absolute query id, k wide
    float* const* raw;    // ntraces device pointers
};

struct ReplayArgs {
    int metric;
    int k;
    uint32_t nlist;
    uint32_t nq;               // queries in this launch (chained rounds: an upper bound, the count is *nq_dev)
    const uint32_t* nq_dev;
    uint32_t nq_hint;          // expected number of queries (picks the kernel variant); 0: nq
    const uint32_t* qsel;      // [nq] query slot of each launch position (null: identity)
    uint32_t total_nprobe;     // length of the reference's probe loop (0: not bounded here)
    uint32_t round_probes;     // row stride of seg_* arrays
    uint64_t id_offset;        // absolute query id of query 0 of this launch
    const float* dist;
    const unsigned long long* mask;  // threshold mode: bit i of mask[j] set <=> dist[64 j + i] was stored (rows 64-aligned)
    float* thr;                // [nq] by query slot: heap top when the launch ends (may be null)
    const uint64_t* seg_off;   // [nq][round_probes] offset of the distance row       (by launch position)
    const int32_t* seg_list;   // [nq][round_probes] list number, <0 = missing centroid (by launch position)
    const uint32_t* seg_count; // [nq] probes supplied this round                      (by launch position)
    const uint32_t* seg_begin; // [nq] first entry of the query in seg_off / seg_list; null: li * round_probes
    int seg_by_slot;           // seg_count / seg_begin are indexed by query slot instead of launch position
    const uint64_t* list_off;  // nlist + 1 (vectors)
    const int64_t* ids;        // per stored vector
    int store_pairs;
    int identity_ids;          // ids are the positions themselves (coarse quantiser rows)
    unsigned long long max_codes;
    int finalize_all;          // fixed-nprobe mode: every query ends after this round
    const uint32_t* limit;     // [slot] time-bounded search: this query's probe loop ends at limit[slot] (null: total_nprobe)
    // per-query state carried across rounds
    float* heap_val;           // [nq][k]
    int64_t* heap_ref;         // [nq][k]  (list << 32 | position), -1 = empty
    uint32_t* stage;           // probes consumed so far
    unsigned long long* nscan; // codes visited so far
    uint32_t* done;            // 1 once the query has its final result
    float* pre_val;            // stop-rule state
    uint32_t* stoped;
    float* dtb;                // [nq][nlist/8+20] distance-to-boundary vector
    const float* coarse_dis;   // [nq][coarse_stride] full coarse ranking (tune / train)
    const int64_t* coarse_keys;
    uint32_t coarse_stride;
    uint32_t trace_cap;        // longest trace (LDS cache size), tune mode
    // outputs
    float* D;                  // [nq][k]
    int64_t* I;
    unsigned long long* stats; // {nlist, ndis, nheap, ties} x 8: one row per XCD (added with L2-local atomics: STATS_ROWS below)
    uint32_t* error;           // != 0: the reference would have thrown (code)
    int raw_heap_out;          // scanner API: leave the heap un-reordered in D/I
    long long pair_list = -1;  // scanner API over a list part (store_pairs): the list number new labels carry (< 0: the segment's own)
    unsigned long long* dbg;   // optional [nq][8]: wave cycles, heap updates, candidates, stages evaluated, cycles in the
                               // candidate stream, cycles in the stop rule, masked chunks fetched, probes consumed
    TunerDev tuner;
    TrainDev train;
    // Sorted-array selection (replay_kernel MODE 2; null log: the heap kernels).  heap_val / heap_ref then hold the k best in
    // best-first order (values, global positions) between rounds.
    uint2* log;                // [slot][log_cap] admissions in order: (value bits, global position = list_off[list] + position)
    uint32_t log_cap;
    uint32_t* log_cnt;         // [slot]
    uint32_t* amb;             // [slot] order key of a value whose id became ambiguous (0xffffffff: none)
    uint32_t* tie_flag;        // [slot] set when the query's result has to come from tie_fix_kernel
    uint32_t round;            // number of this round within the search
    uint32_t* log_snap;        // [2][nq_total] log_cnt as this round left it, by round parity (tie_fix_kernel of this round reads it
                               //               while the next round's selection already appends)
    uint32_t nq_total;
    uint32_t* fin_round;       // [slot] the round in which the query got its final state (0xffffffff: not yet)
    uint2* qstat;              // [slot] (lists scanned, heap updates) of this query so far (null: not kept)
    uint32_t* unfinished;      // null or 8 counters (one per XCD): += 1 for every query of this launch that goes on to another round (with the
                               // queries the planning deferred, PlanArgs counters[11], what is left after the round: the host
                               // needs no further planning pass to learn that a search has ended)
    // threshold rounds, optional: what compact_rows_kernel made of the masks (by launch position, like seg_off)
    const uint32_t* cl_cnt;    // marked candidates of the row (CL_WALK: find them in the masks)
    const uint2* cl_ent;       // [CL_CAP] per row: (position in the list, distance bits) in position order while cl_cnt <= CL_CAP; a longer
                               // list is in the arena, and entry 0 says where: (first entry, -)
    const uint2* cl_arena;
};

bool replay_sorted_applies(const ReplayArgs& a);
void launch_replay(const ReplayArgs& a, hipStream_t s);

// A threshold round leaves a mask bit in one candidate of two thousand: a query's selection walking its rows -- mask words, then
// the marked chunks -- pays a trip to memory per row, one row after the other (132 rows: 0.35 ms with the chip nearly empty).
// compact_rows_kernel does that walk for every row of the round at once (a wave per eight rows) and leaves each row's marked
// candidates as a short list at an address the selection knows in advance: it requests 64 rows' lists with its probe table.
// Rows with more candidates than a list holds (the first rows of a hard query: dozens) get a run of the arena: one eighth of it per
// XCD, places handed out by an add that the XCD's L2 serves (ivf_dev.h: xcd_local_add); when that is full: CL_WALK.
constexpr uint32_t CL_CAP = 8;
constexpr uint32_t CL_WALK = 0xffffffffu;
struct CompactArgs {
    const uint32_t* nseg_dev;  // rows of the round (device count), or null: nseg
    uint32_t nseg;
    uint32_t nseg_hint;        // sizes the grid
    uint32_t nlist;
    const int32_t* seg_list;
    const uint64_t* seg_off;
    const uint64_t* list_off;
    const float* dist;
    const unsigned long long* mask;
    uint32_t* cl_cnt;
    uint2* cl_ent;
    uint2* arena;
    uint32_t arena_per_xcd;    // entries
    uint32_t* cursor;          // [8 x 32] next free entry of every XCD's part (zeroed by the round's planning)
};
void launch_compact_rows(const CompactArgs& a, hipStream_t s);

// see tie_fix_kernel (ivf_select.hip)
struct TieFixArgs {
    int metric;
    int k;
    uint32_t nq, nlist;
    const uint2* log;
    uint32_t log_cap;
    uint32_t round;              // the selection round this launch follows
    int final_pass;              // 1: one launch at the end of the search instead of one per round: every flagged query, whole log
    const uint32_t* log_cnt;     // [nq] (final pass)
    const uint32_t* log_snap;    // [2][nq]
    const uint32_t* fin_round;   // [nq]
    float* fix_val;              // [nq][k] the reference's heap (node order) as far as the log has been replayed
    int64_t* fix_ref;            // [nq][k] global positions, -1 = empty
    uint32_t* fix_pos;           // [nq] log entries replayed so far
    uint32_t* tie_flag;
    const uint64_t* list_off;
    const int64_t* ids;
    int store_pairs, identity_ids;
    float* D;
    int64_t* I;
};
void launch_tie_fix(const TieFixArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------- range search
// IndexIVF::range_search_preassigned: the scan runs in threshold mode with the radius as every query's threshold,
// so the masks mark exactly the entries scan_codes_range would add.  One wave per query counts its entries, then
// (offsets known) writes (distance, label) in the reference's order: probes in order, list entries in order.
struct RangeArgs {
    uint32_t nq;                 // active queries of this round
    uint32_t nlist;
    const uint32_t* qsel;        // [nq] query slot of each launch position
    const uint32_t* seg_count;   // by slot: probes this round
    const uint32_t* seg_begin;   // by slot: first entry in seg_list / seg_off
    const int32_t* seg_list;
    const uint64_t* seg_off;
    const uint64_t* list_off;
    const int64_t* ids;
    const float* dist;
    const unsigned long long* mask;
    uint32_t* counts;                     // by slot: entries found (count pass)
    const unsigned long long* out_off;    // by slot: first output position (fill pass)
    int64_t* out_labels;
    float* out_dist;
    uint32_t* stage;
    uint32_t* done;
    unsigned long long* stats;   // {nlist, ndis, -}
    uint32_t* error;
};
void launch_range_count(const RangeArgs& a, hipStream_t s);
void launch_range_fill(const RangeArgs& a, hipStream_t s);

// error_pro::set_online for nq queries: dtb[q][nlist/8+20] from the full coarse ranking (before round 0)
void launch_set_online(int metric, uint32_t nlist, uint32_t nq, const float* coarse_dis, const int64_t* coarse_keys,
                       uint32_t coarse_stride, const float* interdis, const float* arcos, float* dtb, uint32_t* error, hipStream_t s,
                       const uint32_t* qsel = nullptr, const uint32_t* nq_dev = nullptr);  // (optional: a subset of the queries, count on the device)
void launch_partition_qsel(const uint32_t* qsel, const uint32_t* nq_dev, uint32_t nq, const int32_t* slot_of, uint32_t* q_free, uint32_t* q_wait,
                           uint32_t* counts, hipStream_t s);

// ---------------------------------------------------------------------------- device-side round planning
struct PlanArgs {
    uint32_t nq;               // query slots of this search (state arrays are indexed by slot)
    uint32_t nlist;
    uint32_t total_nprobe;     // length of the probe loop (= nlist in tune / train mode)
    uint32_t key_stride;
    uint32_t slot_base;        // row of slot 0 in the query matrix the scan reads (pair_query = slot_base + slot)
    uint32_t first_round, round_len;
    uint32_t min_inc;          // tune mode: probes a later round adds at least (round 0 runs first_round)
    int tune;
    int d;
    float multipler;
    double grow;
    unsigned long long id_offset;
    unsigned long long dist_budget;  // floats
    uint32_t seg_cap;
    const int64_t* keys;             // [nq][key_stride] coarse ranking
    const uint64_t* list_off;
    const uint32_t* stage;
    const uint32_t* done;
    const unsigned long long* my_nprobe;  // by absolute id, may be null
    // time-bounded search (IndexIVF.cpp:504-506,545-549): budgets in ms by absolute id, ms elapsed when this round is
    // planned, and the per-slot end of the probe loop the plan derives from them (read by the replay)
    const float* budget_ms;
    float elapsed_ms;
    uint32_t* limit;
    // outputs
    uint32_t* cnt;                   // [nq] probes this round (0: finished or deferred)
    unsigned long long* need;        // [nq] floats of distance rows (rows padded to multiples of `row_align`)
    uint32_t* pad;                   // [nq] padding inside need
    uint32_t row_align;              // 1 or 64
    uint32_t qblock;                 // queries per full block of a list: 64 (8-wave tiles) or 32 (4-wave tiles)
    uint32_t mfma_chunk;             // != 0: items for scan_mfma_kernel (vectors per item); counted as tiles of shape 8
    uint32_t mfma_qblock;            // queries per item in that form (MFMA_QBLOCK: one tile)
    int item_order;                  // ... items of a list: 0 chunk major (a chunk's query blocks are neighbours), 1 query block major
    const uint64_t* block_off;       // mfma: first 32-vector block of every list in the fragment-order storage
    const uint64_t* lane_block_off;  // fp32 tiles over the lane-ordered copy (ScanArgs::lanes): the same array, else null
    uint32_t* seg_begin;             // [nq]
    unsigned long long* dist_base;   // [nq]
    uint32_t* qsel;                  // active slots, compacted
    int32_t* seg_list;
    uint64_t* seg_off;
    uint32_t* lcount;                // [nlist] pairs per list (written by plan_lists_kernel)
    uint32_t* lstart;
    uint32_t* gbase;
    uint32_t* ibase;                 // [4][nlist]
    uint32_t* xcount;                // [8 x nlist] pairs per (XCD that counted them, list); after plan_lists_kernel: the XCDs' offsets inside a list
    uint32_t* seg_slot;              // [seg_cap] per pair: XCD << 28 | its place among that XCD's pairs of the list
    uint32_t* pair_query;
    uint64_t* pair_out;
    uint32_t* group_p0;
    uint32_t* group_cnt;
    ScanItem* items;
    uint32_t item_cap;
    uint32_t* error;                 // device error word (ERR_ITEM_OVERFLOW: the round needs more tiles than item_cap)
    unsigned long long* acc64;       // [0] += (query, vector) slots computed, [1] += pairs wanted (tile bookkeeping)
    uint32_t* history;               // null or 16 uint32: receives the counters as the previous round's planning left them
    int first_plan;                  // first planning pass of a search: the accumulators below and round_unfinished start from zero
    uint32_t* round_unfinished;      // null or [PLAN_MAX_ROUNDS][8]: per round and XCD, queries its selection left unfinished (ReplayArgs::unfinished)
    uint32_t* cl_cursor;             // null or [8 x 32]: CompactArgs::cursor, zeroed with the counters
    uint32_t* counters;              // [0] active queries [1] segments [2] pairs [3] groups [4] tiles qg1 [5] tiles qg2
                                     // [6] scratch (compaction cursor, zeroed by host) [7] MiB of distances [8] tiles qg4 [9] tiles qg8
                                     // [10] queries that may still be unfinished after this round, [11] of those: deferred by the
                                     //      budget cut
    double* bytes;                   // [0] += algorithmic bytes of the round's distances
    // [0] += bytes the round cannot avoid moving through HBM: every probed list once (row_bytes per stored vector) and the
    // rows it writes (4 bytes per distance of a dense round, one mask bit per distance in threshold mode)
    double* min_bytes;
    double* min_bytes_thr;           // the same for the threshold rounds alone (null: not kept)
    uint32_t row_bytes;
    int dense_round;
    // null or the sorted coarse distances [nq][key_stride]: a round never ends inside a run of exactly equal distances (the order
    // inside a run may still change -- launch_tie_patch -- and a round's rows can be reordered, not exchanged with the next round's)
    const float* run_dis;
};

// self-check of the per-XCD counters (ivf_plan.hip: xcd_check_kernel): counters [8][nkeys] zeroed by the caller
void launch_xcd_check(uint32_t* counters, uint32_t nkeys, uint32_t* slot, uint32_t* xcc_of, uint32_t n, hipStream_t s);

constexpr uint32_t PLAN_MAX_ROUNDS = 64;
// Counters that every wave of a selection adds to (statistics, queries left unfinished) exist once per XCD and are added to with
// workgroup-scope atomics, which that XCD's L2 serves: thousands of adds on ONE word from all eight XCDs are served one after the
// other at the memory side, ~12 ns each -- 30000 of them were 0.21 of a 0.48 ms selection launch (cfg 1, 10000 queries).  Rows
// 0..7: the XCDs; row 8: kernels that add with ordinary (agent-scope) atomics.  The host sums the rows.
constexpr uint32_t STATS_ROWS = 9;
void launch_plan(const PlanArgs& a, hipStream_t s);

constexpr uint32_t ERR_ARCOS_DOMAIN = 1;
constexpr uint32_t ERR_COSINE_PRECOND = 2;
constexpr uint32_t ERR_INVALID_KEY = 3;
constexpr uint32_t ERR_ITEM_OVERFLOW = 4;
constexpr uint32_t ERR_LOG_OVERFLOW = 5;  // a query admitted more candidates than its admission log holds

// ---------------------------------------------------------------------------- coarse helpers
// full ascending/descending sort of each row of `dis` (nlist entries) keeping the first nprobe
// prefix != 0: only the first `prefix` entries are needed in order; the rest may come out as (neutral, -1)
bool sort_rows_ranks_a_prefix(uint32_t nlist, uint32_t nprobe, uint32_t prefix);
void launch_sort_rows(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, int metric, float* out_dis,
                      int64_t* out_keys, hipStream_t s, uint32_t prefix = 0, bool tail_is_neutral = false);

// Rows whose first `nout` entries hold (or end inside) a run of exactly equal distances are re-ranked by the reference's
// own heap (utils.cpp:417-490, Heap.h:88-142,295-322), whose order inside such a run depends on its history; *nrows
// counts them.  false = heap + row exceed one workgroup's LDS, nothing launched.
bool launch_heap_tie_order(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, uint32_t nout, int metric, float* out_dis,
                           int64_t* out_keys, unsigned long long* nrows, hipStream_t s, const uint32_t* dev_nq = nullptr);

size_t heap_tie_order_lds(uint32_t nlist, uint32_t nprobe);  // LDS bytes of one row's workgroup

// Rankings that may need the heap's order, set aside while the search runs (AUNCEL_AMD_COARSE_TIES=redo): every query whose
// first run of equal distances starts in [lo, window) gets a slot (at most cap; *count keeps counting beyond) holding copies of
// its distance row (nlist floats) and of the first ncopy entries of its ranking (slot rows of nlist entries);
// slot_of[q] = its slot or -1.  launch_heap_tie_order(dev_nq = count) then re-ranks the slots on a side stream.
void launch_spec_collect(const uint32_t* first, uint32_t nq, uint32_t lo, uint32_t window, uint32_t cap, uint32_t nlist, uint32_t ncopy,
                         const float* full, const float* cdis, const int64_t* ckeys, uint32_t* count, int32_t* slot_of, float* s_full,
                         float* s_dis, int64_t* s_keys, hipStream_t s, uint32_t* slot_query = nullptr);
// The heap's order of the slots' rankings applied to the search that set them aside, between the scan and the selection of its
// first round (nothing before the selection reads the order inside a run of equal distances: the planner takes whole runs into
// the round, PlanArgs::run_dis).  Per slot: the ranking's keys are overwritten with the heap's (same distances in the same places);
// where that changes the order of rows the round has scanned, the rows and their table entries are put in the new order (through
// `scratch`).  slot_of[q] = -2 - slot where that is not possible (more than 64 rows, a run across the end of the round, scratch full): the
// caller searches those queries again.  *patched counts the rankings that changed.
struct TiePatchArgs {
    const uint32_t* count;       // slots handed out (may exceed cap)
    uint32_t cap, nlist, ncopy, key_stride;
    const uint32_t* slot_query;  // [cap]
    int32_t* slot_of;            // [nq]
    const int64_t* s_keys;       // [cap][nlist] the heap's order
    int64_t* ckeys;              // [nq][key_stride] the search's ranking
    const uint32_t* seg_count;   // [nq] rows of the query in the round
    const uint32_t* seg_begin;   // [nq]
    int32_t* seg_list;
    uint64_t* seg_off;
    const uint64_t* list_off;
    float* dist;
    uint32_t row_align;
    float* scratch;
    unsigned long long scratch_floats;
    unsigned long long* cursor;  // zeroed by the caller
    uint32_t* patched;
};
void launch_tie_patch(const TiePatchArgs& a, hipStream_t s);
// launch_first_tie + launch_spec_collect in one launch (sorted rankings of row stride nlist): first_out[q] as launch_first_tie
// leaves it, slots handed out in the order the waves arrive
void launch_tie_collect(const float* sorted_dis, uint32_t nq, uint32_t nreal, uint32_t window, uint32_t cap, uint32_t nlist, uint32_t ncopy,
                        const float* full, const int64_t* ckeys, uint32_t* first_out, uint32_t* count, int32_t* slot_of, float* s_full, float* s_dis,
                        int64_t* s_keys, uint32_t* slot_query, hipStream_t s);
// rows of a second search from those slots: ranking row j <- slot slots[j] (ncopy leading entries, rows of nlist entries)
void launch_spec_gather(const int32_t* slots, uint32_t m, uint32_t nlist, uint32_t ncopy, const float* s_dis, const int64_t* s_keys,
                        float* cdis, int64_t* ckeys, hipStream_t s);
// out row j <- x row idx[j] (rows of dpad floats)
void launch_gather_rows(const float* x, const uint32_t* idx, uint32_t m, uint32_t dpad, float* out, hipStream_t s);

// position of the first pair of equal neighbours among the first nreal entries of every sorted ranking (row stride `stride`
// floats), 0xffffffff if there is none
void launch_first_tie(const float* sorted_dis, uint32_t nq, uint32_t stride, uint32_t nreal, uint32_t* out, hipStream_t s);

// GEMM-formulated coarse distances on the fp32 matrix cores (row stride d, d % 4 == 0)
void launch_row_norms(const float* x, size_t n, int d, float* out, hipStream_t s);
// approximate distances from fp16 operands (the rows are scaled and rounded while they are staged; the error constant that goes
// with them is params->C: |approx - exact| <= C (|x|^2 + |y|^2), C < 0: no usable scale, the result means nothing)
void launch_coarse_gemm16(int metric, const float* X, const float* Y, const float* xn, const float* yn, int nq, int ny, int d, float* out,
                          const FilterParams* params, hipStream_t s);
void launch_coarse_gemm(int metric, const float* X, const float* Y, const float* xn, const float* yn, int nq, int ny, int d, float* out,
                        hipStream_t s);

// exact top-nprobe (nprobe << nlist <= 4096) from approximate distances + exact recomputation of the candidates (see
// coarse_pick_kernel); queries whose result could depend on the reference's heap history are counted in *nflag / listed in
// flagged[] and left to the caller
void launch_coarse_pick(int metric, const float* approx, const float* x, const float* centroids, const float* xn, float cmax, uint32_t n,
                        uint32_t nlist, uint32_t nprobe, int dpad, float* out_dis, int64_t* out_keys, uint32_t* nflag, uint32_t* flagged,
                        hipStream_t s, const FilterParams* params = nullptr, uint32_t* why = nullptr);  // params: the approximate distances came from launch_coarse_gemm16
void launch_scatter_rows(const void* in, const uint32_t* idx, uint32_t m, uint32_t words, void* out, hipStream_t s);

// packed upper triangle (IVF_pro.cpp:21-39 layout) of a full nlist x nlist distance matrix
void launch_pack_upper(const float* full, uint32_t nlist, float* out, hipStream_t s);

// k-means centroid update (ivf_kmeans.hip): points grouped by centroid with a stable sort, then fp32 sums in point order
size_t kmeans_sort_temp_bytes(size_t n);
void launch_kmeans_group(const int64_t* assign, size_t n, uint32_t k, uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_in,
                         uint32_t* idx_out, uint32_t* counts, void* temp, size_t temp_bytes, hipStream_t s);
void launch_kmeans_sums(const float* x, size_t stride, int d, const uint32_t* idx_sorted, const uint32_t* seg_off, uint32_t k, float* centroids,
                        hipStream_t s);

// Up to MAX small copies in ONE launch (one workgroup row per segment; sizes in 4-byte words).  A search starts and ends with a
// dozen copies of a few bytes to a few tens of KB (per-call inputs, error word, counters, my_nprobe, t_recalls ...): as
// hipMemcpyAsync each of them is a blit kernel of its own on the stream (18 % of the four-in-flight span in round 3's trace);
// through page-locked staging they are one kernel before the first round and one behind the last.
struct CopySegs {
    static constexpr int MAX = 12;
    const void* src[MAX];
    void* dst[MAX];
    uint32_t words[MAX];
    uint32_t n;
};
void launch_copy_segs(const CopySegs& c, hipStream_t s);

// fill helpers
struct InitStateArgs {
    size_t n, k;
    float neutral;  // FLT_MAX (L2) / -FLT_MAX (IP)
    float* heap_val;
    int64_t* heap_ref;
    float* thr;
    uint32_t* stage;
    unsigned long long* nscan;
    uint32_t* done;
    float* pre_val;
    uint32_t* stoped;
    unsigned long long* stats;  // 4 counters
    uint32_t* error;
    uint32_t *log_cnt, *amb, *tie_flag;  // sorted-array selection (may be null)
    uint32_t *log_snap, *fin_round, *fix_pos;
    uint2* qstat;
    float* fix_val;
    int64_t* fix_ref;
};
void launch_init_state(const InitStateArgs& a, hipStream_t s);
// A call of at most four queries: the per-query state (init_state_kernel), the boundary distances of the stop rule
// (set_online_kernel), the start of each ranking's first run of equal coarse distances (first_tie_kernel) and the signed-byte view of
// the queries (sbytes_from_f32_kernel) in ONE launch -- four launches of a few microseconds of work each are four dispatch
// latencies on the critical path of a 0.25 ms call.
struct SmallStateArgs {
    InitStateArgs init;
    int metric;
    uint32_t nlist, nq;
    const float* coarse_dis;
    const int64_t* coarse_keys;
    uint32_t coarse_stride;
    const float* interdis;
    const float* arcos;
    float* dtb;
    const float* ft_sorted_dis;  // first tie: null = not wanted
    uint32_t ft_stride, ft_nreal;
    uint32_t* ft_out;
    const float* bx;             // byte view: null = not wanted
    int d, dpad;
    int8_t* bout;
    int32_t* bcx;
};
void launch_small_state(const SmallStateArgs& a, hipStream_t s);
void launch_range_collect(const float* dist, uint32_t n, float radius, int metric, uint32_t* count, uint32_t* out_pos, float* out_dis, hipStream_t s);
void launch_fill_f32(float* p, size_t n, float v, hipStream_t s);
void launch_fill_i64(int64_t* p, size_t n, int64_t v, hipStream_t s);

}  // namespace amdivf
