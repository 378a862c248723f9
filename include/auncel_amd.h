/* auncel_amd.h -- C ABI of the MI355X (gfx950) IVF-Flat search engine.
 *
 * This is the drop-in boundary for Auncel's IVF-Flat search hot path: plain pointers and
 * sizes, no C++/torch types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repo's Auncel/ directory).  The host-side C++ mirror of
 * the reference classes (auncel_amd/csrc/host/, namespace faiss) calls only these functions;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (same as the reference's c_api, c_api/error_c.h:20-33, macros_impl.h:22-58):
 *   - every function returns 0 on success, -1 unknown error, -2 engine exception (what the
 *     reference raises as FaissException: invalid key, arcos domain, cosine-theorem
 *     precondition, tune without tuner...), -4 std::exception / HIP runtime failure;
 *     amd_ivf_last_error() returns the message of the last failure on the calling thread.
 *   - matrices are compact row-major; ids are int64 ("long" idx_t, Index.h:67); outputs are
 *     fully overwritten, sorted best first, padded with id -1 and +/-FLT_MAX (Heap.h:317-320).
 *   - x / D / I / keys / coarse_dis arguments are HOST pointers unless the function name ends
 *     in _dev, in which case they are device pointers on the handle's GPU.
 *   - there is NO CPU fallback: without a usable HIP device every call fails with -4.
 */
#ifndef AUNCEL_AMD_H
#define AUNCEL_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMD_IVF_METRIC_INNER_PRODUCT 0 /* MetricType, Index.h:49-52 */
#define AMD_IVF_METRIC_L2 1

typedef struct amd_ivf amd_ivf_t;

const char* amd_ivf_last_error(void);
int amd_ivf_device_count(int* count);

/* ---- index lifetime and contents ------------------------------------------------------ */

/* Self-check of an assumption the engine's device-side bookkeeping rests on: counters that exist once per XCD (statistics,
 * queries left unfinished by a round, pairs per list) are added to with workgroup-scope atomics, which the issuing XCD's L2
 * executes -- exact as long as no other XCD touches the row, and an XCD has a number below 8.  Runs the adds under contention
 * (2^20 returning adds on 8 x 1024 addresses).  out[0] adds made, out[1] sum of the counters (must equal out[0]), out[2] (XCD,
 * address) pairs whose returned values are not exactly 0 .. count - 1 (must be 0), out[3] bit mask of the XCD numbers seen.
 * amd_ivf_create runs it once per device and refuses to create an index where it fails.  (No reference counterpart: diagnostics.) */
int amd_ivf_self_check(int device, uint64_t out[4]);

/* IndexIVFFlat(quantizer, d, nlist, metric)  [IndexIVFFlat.cpp:28-33, IndexIVF.cpp:146-170] */
int amd_ivf_create(int d, size_t nlist, int metric, int device, amd_ivf_t** out);
int amd_ivf_destroy(amd_ivf_t* h);

/* A second search context over the same device-resident index: its own HIP stream, workspaces and resident
 * queries; lists, centroids, centroid table and traces stay with (and are owned by) `h`.  The reference's
 * search() / search_preassigned() are const and re-entrant over the index data (IndexIVF.h:189-207, and
 * IndexShards(threaded=true) calls them from one thread per shard, IndexShards.cpp:261-311); a context is what a
 * thread of the caller uses to keep its own batch in flight on the GPU while another thread's batch is in a
 * latency-bound phase.  Valid on a clone: set_queries, search, search_preassigned, search_resident,
 * search_adaptive(_x), stats, scan_arith, last_timing, ntotal, list_size, destroy; everything else must go through
 * `h` and returns -2 here.  Build the index first: clones do not see later add() / set_*() calls until `h` has
 * searched once, and `h` must outlive its clones. */
int amd_ivf_clone(amd_ivf_t* h, amd_ivf_t** out);

/* quantizer->add(nlist, centroids): IndexFlat::xb, row-major nlist x d  [IndexFlat.cpp:30-33] */
int amd_ivf_set_centroids(amd_ivf_t* h, const float* centroids);

/* ArrayInvertedLists contents: per list l, sizes[l] vectors codes[l] (sizes[l] x d fp32) and
 * ids[l]; copied to HBM CSR-packed  [InvertedLists.h:182-202, InvertedLists.cpp:138-196] */
int amd_ivf_set_lists(amd_ivf_t* h, const size_t* sizes, const float* const* codes, const int64_t* const* ids);

/* IndexIVFFlat::add_core: assign each vector to its nearest centroid (exact kernel) and append
 * it to that list in input order; xids == NULL -> ids ntotal..ntotal+n-1; precomputed_idx may
 * be NULL, entries < 0 are skipped  [IndexIVFFlat.cpp:41-80] */
int amd_ivf_add(amd_ivf_t* h, size_t n, const float* x, const int64_t* xids, const int64_t* precomputed_idx);

int amd_ivf_ntotal(const amd_ivf_t* h, size_t* ntotal);
int amd_ivf_list_size(const amd_ivf_t* h, size_t list_no, size_t* size);
/* InvertedLists::get_codes / get_ids: copy one list back to the host */
int amd_ivf_get_list(const amd_ivf_t* h, size_t list_no, float* codes, int64_t* ids);

/* ---- search ---------------------------------------------------------------------------- */

/* quantizer->search(n, x, nprobe, coarse_dis, keys)  [IndexFlat.cpp:42-56].
 * mode 0: exact per-pair kernel, same fp32 summation order as the reference's default (SSE)
 *         build of fvec_L2sqr / fvec_inner_product, i.e. knn_L2sqr_sse [utils.cpp:454-490];
 * mode 1: |x|^2+|y|^2-2x.y on the fp32 MFMA path, i.e. knn_L2sqr_blas [utils.cpp:538-608]
 *         (the reference takes it for n >= 20; its rounding belongs to the vendor BLAS);
 * mode -1: pick as the reference does (n < 20 && d % 4 == 0 -> 0, else 1)  [utils.cpp:644-655] */
int amd_ivf_coarse(amd_ivf_t* h, size_t n, const float* x, size_t nprobe, float* coarse_dis, int64_t* keys, int mode);

/* IndexIVF::search_preassigned (plain branch)  [IndexIVF.cpp:382-736; vanilla faiss/IndexIVF.cpp:250-428] */
int amd_ivf_search_preassigned(amd_ivf_t* h, size_t n, const float* x, size_t k, size_t nprobe, const int64_t* keys,
                               const float* coarse_dis, float* D, int64_t* I, int store_pairs, size_t max_codes);

/* IndexIVF::search(n, x, k, D, I): coarse + scan without leaving the device  [IndexIVF.cpp:335-353] */
int amd_ivf_search(amd_ivf_t* h, size_t n, const float* x, size_t k, size_t nprobe, int coarse_mode, float* D,
                   int64_t* I);

/* InvertedListScanner: set_query + set_list + scan_codes on a caller-owned raw binary heap
 * (simi/idxi, k entries, heapified by the caller as in tests/test_lowlevel_ivf.cpp:150-175);
 * returns the number of heap updates in *nup  [IndexIVFFlat.cpp:101-137] */
int amd_ivf_scan_codes(amd_ivf_t* h, const float* query, size_t list_no, int store_pairs, size_t k, float* simi,
                       int64_t* idxi, size_t* nup);
/* ... scan_codes over ANY run of codes of the list: the reference's scanner scans the n codes at whatever pointer it is handed
 * (IndexIVFFlat.cpp:117-137; tests/test_lowlevel_ivf.cpp:426-564 hands different parts of a list to different threads).  Here the
 * codes live in HBM, so the run is named: vectors [offset, offset + n) of list `list_no`.  Labels of the entries this call admits:
 * store_pairs = 0: the stored ids of those vectors; store_pairs = 1: list_no << 32 | j with j counted FROM `offset`, exactly what the
 * reference computes for a `codes` pointer that starts there (IndexIVFFlat.cpp:131).  Labels already in the heap pass through. */
int amd_ivf_scan_codes_at(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, size_t n, int store_pairs, size_t k,
                          float* simi, int64_t* idxi, size_t* nup);
/* InvertedListScanner::scan_codes_range over the same kind of run  [IndexIVF.h:349-354, IndexIVFFlat.cpp:139-155]: the entries with
 * C::cmp(radius, dis) -- dis < radius (L2) / dis > radius (IP) -- in position order, as RangeQueryResult::add receives them.  *count
 * entries; amd_ivf_scan_codes_range_results copies their positions j (counted from `offset`) and distances out of the handle. */
int amd_ivf_scan_codes_range(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, size_t n, float radius, size_t* count);
int amd_ivf_scan_codes_range_results(amd_ivf_t* h, uint32_t* positions, float* distances);
/* InvertedListScanner::distance_to_code for vector `offset` of list `list_no`  [IndexIVFFlat.cpp:110-115] */
int amd_ivf_distance_to_code(amd_ivf_t* h, const float* query, size_t list_no, size_t offset, float* dis);

/* IndexIVFStats {nq, nlist, ndis, nheap_updates}  [IndexIVF.h:361-374, IndexIVF.cpp:731-734] */
int amd_ivf_stats(amd_ivf_t* h, size_t stats[4], int reset);

/* amd_ivf_coarse over resident queries [start, start + n) (coarse_dis may be NULL: keys only) */
int amd_ivf_coarse_resident(amd_ivf_t* h, size_t start, size_t n, size_t nprobe, float* coarse_dis, int64_t* keys, int mode);
/* search_preassigned over resident queries [start, start + n): the caller's keys (n x nprobe, host), the engine's lists.
 * What a shard of an IndexShards runs when the coarse ranking was computed once, elsewhere. */
int amd_ivf_search_resident_preassigned(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const int64_t* keys, float* D,
                                        int64_t* I);
/* ---- resident query sets (Error_sys::set_queries keeps the query matrix and later searches
 *      slices of it, profile.cpp:173-227): upload once, search slices with data already in HBM */
int amd_ivf_set_queries(amd_ivf_t* h, size_t n, const float* x);
int amd_ivf_search_resident(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, int coarse_mode, float* D,
                            int64_t* I);

/* ---- Auncel error-bound machinery (IVF_pro.{h,cpp}, profile.{h,cpp}) -------------------- */

/* Level1Quantizer::train_q1 table: interdis_cem[(2 nlist-1-i) i/2 + j-1-i], i<j  [IndexIVF.cpp:97-117].
 * With table == NULL it is computed on the device from the current centroids (L2: squared
 * distances; IP: reference normalisation quirk + acos); otherwise it is uploaded as given. */
int amd_ivf_set_interdis(amd_ivf_t* h, const float* table);
int amd_ivf_get_interdis(amd_ivf_t* h, float* table); /* nlist(nlist-1)/2 floats */

/* IndexIVF::init_tune + trained traces: ntraces = log2(nlist/8)+1 maps sum-of-angles -> k-scaling,
 * trace i has trace_len[i] ascending bins (x, y, std)  [IndexIVF.cpp:203-244, IVF_pro.h:47-66].
 * arcos_list: 500-entry acos LUT (error_pro::construct_arcos, IVF_pro.cpp:151-160). */
int amd_ivf_set_tuner(amd_ivf_t* h, size_t max_topk, size_t ntraces, const size_t* trace_len, const float* const* trace_x,
                      const float* const* trace_y, const float* const* trace_std, const float* arcos_list);

/* search_preassigned, tune branch, driven as Error_sys::search does (nprobe = nlist, k = max_topk)
 * on resident queries [start, start+n): per-query adaptive stop  [IndexIVF.cpp:507-638, profile.cpp:211-227].
 * require_acc / gt_D (may be NULL unless profile) / my_nprobe / t_recalls are indexed by absolute query id
 * like the reference's arrays (id_q = i + offset, IndexIVF.cpp:487); my_nprobe entries must be 0 on entry
 * for queries that have not been searched (Error_sys::set_queries zeroes them).
 * profile: bit 0 = error_pro::profile (t_recalls gets the true recall at the stop, IndexIVF.cpp:628-631);
 *          bit 1 = error_pro::overhead_profile (IndexIVF.cpp:529-539,614,634-637; eval/overhead.cpp:284-290): the rule is
 *          evaluated on every probe, its verdict ignored, every query runs to stage nlist / 8 and my_nprobe stays as
 *          passed.  The reference prints the time of the list scans alone next to it; here that is the same search without
 *          the rule, amd_ivf_search_resident(start, n, max_topk, nlist / 8) (the class mirror times both). */
int amd_ivf_search_adaptive(amd_ivf_t* h, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                            const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                            uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I);

/* same, for queries passed by (host) pointer: IndexIVF::search(n, x, k, D, I, offset) with tune on
 * [IndexIVF.cpp:355-378]; arrays are indexed by id_offset + i */
int amd_ivf_search_adaptive_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t query_topk, float multipler,
                              float std_m, const float* require_acc, const float* gt_D, int profile, int coarse_mode,
                              uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I);

/* IndexIVF::search_preassigned with tune on, over the coarse ranking the CALLER passes (Auncel/IndexIVF.cpp:382-386: keys and
 * coarse_dis are arguments there; Error_sys::search gets them from its quantizer with nprobe = nlist, profile.cpp:220): rows of
 * nprobe entries per query, best first.  The probe loop has nprobe steps and set_online reads entries 0 .. nlist/8 + 20, so
 * nprobe must exceed nlist/8 + 20 (else -2).  Everything else as amd_ivf_search_adaptive_x.  This is what a subclass overriding
 * search_preassigned uses when the order of the reference's own quantizer (its BLAS, its tie order) has to be kept. */
int amd_ivf_search_adaptive_pre(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t nprobe, const int64_t* keys,
                                const float* coarse_dis, size_t query_topk, float multipler, float std_m, const float* require_acc,
                                const float* gt_D, int profile, uint64_t* my_nprobe, float* t_recalls, float* D, int64_t* I);

/* Time-bounded search: Error_sys::time_search (profile.cpp:229-244) = IndexIVF::search with tune off, nprobe = nlist and
 * error_pro::time_tune on: the plain probe loop (nprobe probes; time_search sets nprobe = nlist) over resident queries [start, start+n), heap of k, left when the
 * query's budget is used up.  budget_ms is indexed by absolute query id (the reference keeps the budgets in
 * t->require_acc[id_q], effect_time.cpp:274-281).  The reference tests after every probe ik whether one more is expected to
 * fit, `el >= 0.95 budget - el / (ik + 1)` (IndexIVF.cpp:545-549); the engine reads the clock between rounds and lets a
 * query take as many further probes as the same estimate says still fit (at most doubling per round).  The clock starts
 * when the call is entered (coarse quantisation included, which the reference's t0 leaves out).  Whatever the timing, the
 * result of query i is exactly search_preassigned's over its first nprobe_used[i] probes (nprobe_used may be NULL). */
int amd_ivf_search_timed(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const float* budget_ms, int coarse_mode,
                         uint64_t* nprobe_used, float* D, int64_t* I);
/* same for queries passed by (host) pointer; budget_ms[id_offset + i] belongs to query i */
int amd_ivf_search_timed_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t k, size_t nprobe,
                           const float* budget_ms, int coarse_mode, uint64_t* nprobe_used, float* D, int64_t* I);

/* search_preassigned, training branch, driven as Error_sys::sys_train does: raw (sum_angle, kscaling)
 * samples for stages 1,2,4..nlist/8 of resident queries [start, start+n); raw[i] has
 * train_num*(max_topk/4) (x,y) pairs and must be pre-filled with (-1,-1)  [IndexIVF.cpp:640-673, profile.cpp:88-156] */
int amd_ivf_train_samples(amd_ivf_t* h, size_t start, size_t n, size_t max_topk, const float* gt_D, size_t train_num,
                          int coarse_mode, float* const* raw, float* D, int64_t* I);

int amd_ivf_train_samples_x(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t max_topk, const float* gt_D,
                            size_t train_num, int coarse_mode, float* const* raw, float* D, int64_t* I);

/* the training branch over the caller's coarse ranking (search_preassigned with training on and its keys / coarse_dis
 * arguments, IndexIVF.cpp:382-386,640-673); nprobe > nlist/8 + 20 */
int amd_ivf_train_samples_pre(amd_ivf_t* h, size_t n, const float* x, size_t id_offset, size_t nprobe, const int64_t* keys,
                              const float* coarse_dis, size_t max_topk, const float* gt_D, size_t train_num, float* const* raw,
                              float* D, int64_t* I);

/* error_pro::construct_arcos: the 500-entry acos LUT, computed with the host libm exactly as the
 * reference does (LUT[i] = acosf((i-250)/250.f))  [IVF_pro.cpp:151-160] */
int amd_ivf_arcos_table(float out[500]);

/* Trace::SB: sort the raw samples by x descending, drop unset (-1,-1) ones, bucket by `bs` (250 in the
 * reference), mean x / mean y / std y per bucket, ascending on return; out_* need n/bs + 1 entries;
 * *nbuckets receives the count.  Host-side, like the reference  [IVF_pro.cpp:109-149] */
int amd_ivf_trace_sb(const float* raw_xy, size_t n, size_t bs, float* out_x, float* out_y, float* out_std,
                     size_t* nbuckets);

/* ---- sharding (IndexShards over list-id shards, IndexShards.cpp:44-105,261-311) ----------- */

/* k-way merge of nshard sorted result tables (all_D/all_I: nshard x n x k) -- host-side, as in the reference */
int amd_ivf_merge_tables(int metric, size_t n, size_t k, size_t nshard, const float* all_D, const int64_t* all_I,
                         float* D, int64_t* I);

/* ---- k-means -------------------------------------------------------------------------------------
 * Clustering::train (Clustering.cpp:75-226) over an IndexFlat of `metric`, as Level1Quantizer::train_q1 runs it
 * (IndexIVF.cpp:84-92): sub-sampling beyond k * max_points_per_centroid and seeding with the reference's rand_perm /
 * RandomGenerator (utils.cpp:111-137,229-239), niter x { assignment (coarse_mode as in amd_ivf_coarse: 0 exact kernel,
 * 1 |x|^2+|y|^2-2xy on the matrix cores, -1 the reference's switch), objective, km_update_centroids (utils.cpp:1078-1159)
 * in its fp32 summation order, void-cluster splitting }, spherical / int_centroids post-processing.  nredo 1, no input
 * centroids.  centroids: k x d out; obj: niter objective values out (may be NULL).  With coarse_mode 0 the centroids
 * equal the reference's bit for bit wherever its BLAS assignment picks the same centroids (tests/golden/kmeans_*). */
int amd_ivf_kmeans(int d, size_t n, const float* x, size_t k, int metric, int niter, long seed, size_t max_points_per_centroid,
                   int spherical, int int_centroids, int coarse_mode, int device, float* centroids, float* obj);

/* ---- range search ------------------------------------------------------------------------------
 * IndexIVF::range_search / range_search_preassigned (IndexIVF.cpp:740-857) with IVFFlatScanner::scan_codes_range
 * (IndexIVFFlat.cpp:139-155): all stored vectors of the probed lists with dis < radius (L2) / dis > radius (IP), per
 * query in the reference's order (probes in order, list entries in order), unsorted.  RangeSearchResult is filled in
 * two steps as in the reference (lims first, then do_allocation): the search call writes lims (n + 1 entries), the
 * caller allocates lims[n] labels / distances and fetches them with amd_ivf_range_results from the same handle. */
int amd_ivf_range_search_preassigned(amd_ivf_t* h, size_t n, const float* x, float radius, size_t nprobe, const int64_t* keys,
                                     size_t* lims);
int amd_ivf_range_search(amd_ivf_t* h, size_t n, const float* x, float radius, size_t nprobe, int coarse_mode, size_t* lims);
int amd_ivf_range_results(amd_ivf_t* h, int64_t* labels, float* distances);

/* ---- environment ----------------------------------------------------------------------------------
 * AUNCEL_AMD_BLOCKING_SYNC=1   host threads wait for the device on blocking events (sleep until the interrupt) instead of
 *                              hipStreamSynchronize's spinning: for hosts where the calling threads outnumber their cores
 * GPU_MAX_HW_QUEUES (the HIP runtime's own variable): hardware queues per stream-priority class.  The engine's streams -- a
 *                              high-priority main stream, a background stream and a low-priority side stream per search context --
 *                              are laid out for 8 (ROCm's default is 4, with which the background streams of several searches in
 *                              flight share queues and wait for each other's kernels).  The library never changes the environment:
 *                              a process that wants more than 4 searches in flight (amd_ivf_set_async_depth, or threads of its own on
 *                              amd_ivf_clone contexts) exports GPU_MAX_HW_QUEUES=8 before anything -- torch included -- starts the HIP
 *                              runtime (bench.py and the tests do).  With fewer queues the asynchronous entry points run only as
 *                              many searches at a time as there are queues in a class and say so once on stderr (AUNCEL_AMD_QUIET)
 * The other AUNCEL_AMD_* variables the sources read are measurement switches (DESIGN.md names the ones it quotes). */

/* ---- measurement hooks (bench.py): time of the kernels of the last search call, from HIP events
 *      recorded on the engine's own stream: {coarse_ms, scan_ms, select_ms, total_ms, scan_launches,
 *      bytes of the distances the scan tiles computed (x d x 4), fraction of the computed (query, vector)
 *      slots that were wanted pairs, select launches (= rounds x sub-batches)} */
int amd_ivf_last_timing(amd_ivf_t* h, double out[8]);
/* the same by phase, (ms, launches) pairs: coarse ranking, dense scan (round 0), selection of dense rounds, threshold scan,
 * selection of threshold rounds, tie_fix_kernel (its own stream), round planning; then the bytes the dense and the threshold
 * rounds of the search could not avoid moving (every probed list once per round + rows / mask bits written) -- bench.py's
 * roofline.per_launch */
int amd_ivf_last_timing_detail(amd_ivf_t* h, double out[16]);
/* bytes the scans of the last search could not avoid moving through HBM: every probed list once per round (as stored:
 * d bytes per vector in byte-code mode, 4 d in fp32) plus the rows written (4 bytes per distance of a dense round, one
 * mask bit per distance in threshold mode).  A lower bound of the traffic, counted on the device by the planning kernels. */
int amd_ivf_last_scan_min_bytes(amd_ivf_t* h, double* bytes);
/* Coarse rankings longer than 128 are sorted on the device; inside a run of exactly equal distances the reference's order
 * (knn_L2sqr_sse / knn_inner_product_sse, Auncel/utils.cpp:417-490: a binary heap over centroids 0..nlist-1, heap-sorted
 * at the end, Heap.h:295-322) depends on the heap's history.  Rankings that hold such a run in the part that is read can
 * be re-run through that heap on the device (up to 2.8 ms each at nlist 4096).  Policy:
 *   calls of fewer than 20 queries (the reference's exact regime; what the eval/ harnesses issue) -- fixed-nprobe search, coarse,
 *     training: re-run.  Adaptive search: searched with such runs in centroid-number order first; a query reads only
 *     entries below 2 my_nprobe + 14, so if no run starts below that the result is the reference's, else the call is
 *     repeated with the heap's order -- same results as always re-running, at a hundredth of the cost.  Time-bounded
 *     search: centroid-number order (the clock decides the depth);
 *   larger calls: centroid-number order (the reference ranks sgemm output there);
 *   AUNCEL_AMD_COARSE_TIES=heap: always re-run; =id: never; =redo (adaptive calls of >= 20 queries): the exact-distance result
 *     for every query of the call.  The rankings whose first run starts inside what the first two rounds can read are re-run
 *     through the heap on a side stream WHILE the call's first round is planned and scanned (the planner takes whole runs into a
 *     round, so nothing before the first selection depends on the order inside a run); the heap's order is then written over
 *     those rankings and, where it changes the order of rows already scanned, over the rows (amd_ivf_last_tie_patched: rankings
 *     changed) -- one pass, the reference's order.  What that cannot cover (no slot left, a ranking read past what was re-run,
 *     nlist not a power of two: the level-parallel heap does not apply) is searched again as one small call with the heap's
 *     order (amd_ivf_last_tie_redone: how many queries that was).
 * *rows = rankings re-run so far on this handle. */
int amd_ivf_coarse_tie_rows(amd_ivf_t* h, uint64_t* rows);
int amd_ivf_last_tie_redone(amd_ivf_t* h, uint64_t* queries);
int amd_ivf_last_tie_patched(amd_ivf_t* h, uint64_t* rankings);
/* Launch sizing of the last search.  The device-planned rounds are enqueued without reading anything back, so a scan's grid is
 * sized from what the same round of the previous search of this shape needed (+ 12 %); its workgroups stride over the item
 * count the planner left on the device, so a round that needs more is still complete -- it just runs on fewer workgroups than it
 * would have been given.  out[0] = scan launches of the last search that were sized by such a hint, out[1] = those whose hint
 * was too small. */
int amd_ivf_last_round_hints(amd_ivf_t* h, uint64_t out[2]);
/* Selection diagnostics of the last search on this handle.  The k best of a query are kept as a sorted array (same
 * admissions as the reference's heap, Auncel/Heap.h:88-142); a query in which equal distances met -- the only case in which
 * the heap's history decides an id or an output order -- gets its result from the reference's heap replayed over the query's
 * admission log.  *queries = how many queries of the last search took that second path. */
int amd_ivf_last_tie_fixed(amd_ivf_t* h, uint64_t* queries);
/* fp32 lists: threshold rounds compute x.y on the matrix cores and keep a candidate iff its distance, widened by a rigorous
 * bound on the difference to the reference's value (utils_simd.cpp:391-443 rounding sequence), can beat the query's threshold;
 * the kept ones are recomputed in the reference's rounding sequence, so (D, I) stay bit-identical.  out[0] = rounds of the last
 * search on this handle that ran that way, out[1] = survivor slots the last of them handed out (>= the candidates it kept for the
 * exact recomputation: a wave takes slots 64 at a time and marks the ones it leaves unused). */
int amd_ivf_last_filter(amd_ivf_t* h, uint64_t out[2]);
/* 1 if the last search on this handle wrote (D, I) straight into the caller's buffers: when both are page-locked, device-visible
 * host memory (hipHostMalloc / hipHostRegister; torch's pin_memory) the selection kernels store each query's row there as the
 * query finishes, under the later rounds, and the call ends without a copy; pageable buffers are filled by a copy at the end.
 * AUNCEL_AMD_DIRECT_OUT=0 always copies. */
int amd_ivf_last_direct_out(amd_ivf_t* h);
/* Exact coarse rankings of a large fixed-nprobe call (>= 256 queries, nprobe <= 128 << nlist <= 4096; IndexFlat::search in the
 * arithmetic of knn_L2sqr_sse / knn_inner_product_sse, utils.cpp:417-490): the matrix cores rank every centroid approximately, the
 * centroids that can be among the nprobe best -- by a rigorous bound on the difference to the reference's value -- are recomputed
 * in the reference's rounding sequence and sorted; a query in which exactly equal distances meet (the reference's order there is
 * its heap's history) is recomputed from exact distances to every centroid through that heap.  Returned distances and ids are
 * the reference's bit for bit either way.  *rankings = how many of the last coarse call's came the first way.
 * amd_ivf_set_option(h, "coarse_pick", 0) keeps the exact distances to every centroid for all queries. */
int amd_ivf_last_coarse_pick(amd_ivf_t* h, uint64_t* rankings);

/* ------------------------------------------------------------------------------------------------
 * Asynchronous form of amd_ivf_search_adaptive (same arguments, same results).  submit returns at once with a ticket; the
 * search runs on one of `depth` internal search contexts of the handle (amd_ivf_set_async_depth, 1..16, default 4, fixed by
 * the first submit; depth 0 with no ticket out releases them: their streams and workspaces are the price of the mode; each is an amd_ivf_clone: own stream and workspaces, the owner's lists, traces and resident queries);
 * further tickets queue.  wait blocks until that search has ended and returns its status (0 / -2 / -4 as the synchronous
 * call; amd_ivf_last_error() then holds its message); timing (amd_ivf_last_timing's 8 doubles + amd_ivf_last_scan_min_bytes) and diag (launches sized by a
 * hint, hints too small, queries searched again for the tie order, 1 if (D, I) were written directly) may be null.  Every
 * buffer passed to submit must stay valid, and the index and its resident queries unchanged, until the ticket has been waited
 * for; every ticket must be waited for exactly once.  One caller thread that keeps six 5000-query searches running this way (twelve
 * tickets out: bench.py's headline) reaches what six threads with a context each reach (the reference's callers would use
 * threads: IndexShards.cpp:48-120). */
int amd_ivf_set_async_depth(amd_ivf_t* h, int depth);
int amd_ivf_submit_adaptive(amd_ivf_t* h, size_t start, size_t n, size_t query_topk, float multipler, float std_m,
                            const float* require_acc, const float* gt_D, int profile, int coarse_mode, uint64_t* my_nprobe,
                            float* t_recalls, float* D, int64_t* I, uint64_t* ticket);
/* the same for amd_ivf_search_resident (IndexIVF::search with a fixed nprobe over resident queries [start, start + n)) */
int amd_ivf_submit_search_resident(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, int coarse_mode, float* D, int64_t* I,
                                   uint64_t* ticket);
/* ... and the two halves of a sharded search (IndexShards over sub-indexes that share one quantizer, IndexShards.cpp:261-311): the
 * coarse ranking of a share of the resident queries, and search_preassigned of a resident range with keys that came from elsewhere
 * (the other shards' ranks).  `keys` / `coarse_dis` / `D` / `I` must stay valid until amd_ivf_wait returns the ticket. */
int amd_ivf_submit_coarse_resident(amd_ivf_t* h, size_t start, size_t n, size_t nprobe, float* coarse_dis, int64_t* keys, int mode,
                                   uint64_t* ticket);
int amd_ivf_submit_search_resident_preassigned(amd_ivf_t* h, size_t start, size_t n, size_t k, size_t nprobe, const int64_t* keys,
                                               float* D, int64_t* I, uint64_t* ticket);
int amd_ivf_wait(amd_ivf_t* h, uint64_t ticket, double timing[9], uint64_t diag[4]);
/* tickets the asynchronous entry points have served since the handle was made, and the passes that served them (fewer where
 * option "coalesce" joined queued tickets) */
int amd_ivf_async_counts(amd_ivf_t* h, uint64_t out[2]);

/* Arithmetic the list scan of the last search ran in.  All three produce the reference's fp32 distance bit for
 * bit (utils_simd.cpp:391-443 order); the engine picks the cheapest one the data allows:
 *   0  fp32, the reference's four running sums, separate multiply and add
 *   1  fp32 with fma: lists and queries hold integers of magnitude <= 4095
 *   2  byte codes + integer dot products: lists and queries hold integers 0..255 and d * max^2 <= 2^24
 *      (SIFT / BIGANN descriptors), lists are kept as bytes on the device (1/4 of the HBM traffic) */
int amd_ivf_scan_arith(amd_ivf_t* h);
/* enable = 0: this handle (the index owner or a search context of amd_ivf_clone) scans the fp32 lists even where the byte
 * codes qualify (same results; bench.py's fp32_path leg).  enable = 1 restores the default. */
int amd_ivf_set_byte_codes(amd_ivf_t* h, int enable);

/* Options of an index: policy that can change a result's tie order, and tuning that cannot change a result at all.  The
 * reference exposes such choices as public fields of the index (nprobe, max_codes, parallel_mode ..., IndexIVF.h:97-143) and
 * through ParameterSpace::set_index_parameter(index, "name", value) (AutoTune.h) / c_api/IndexIVF_c.h:82-85 getter-setter
 * pairs; this is the same thing in C.  `h` may be the index or one of its search contexts (the option is the index's either
 * way); set it while no search is running.  get returns the effective value: what was set, else the value of the debugging
 * environment variable of the same meaning (AUNCEL_AMD_<KEY>), else the built-in default.  Unknown key: -2.
 *
 *   key               values                                                                 default
 *   "coarse_ties"     order inside runs of bit-equal coarse distances (knn_L2sqr_sse's heap   unset (-1): heap for calls of
 *                     history, utils.cpp:454-490): 0 centroid number, 1 the reference's       fewer than 20 queries (the regime
 *                     heap for every ranking, 2 search again exactly the queries whose        in which the reference ranks exact
 *                     result could depend on it                                               distances, utils.cpp:644-655), else 0
 *                     -- the ONLY option that can change returned ids (of queries whose probe order crosses such a run)
 *   "select"          0 the reference's binary heap replayed for every query, 1 sorted        1
 *                     arrays + heap replay of the queries in which equal distances met
 *   "tie_fix"         0 that replay once at the end of a search, 1 behind every round         unset: by call size / concurrency
 *   "filter"          fp32 threshold rounds: matrix-core filter + exact recomputation of what  2
 *                     it keeps, over an fp16 (2) or fp32 (1) copy of the lists; 0 vector ALU only
 *   "fixed_rounds"    fixed-nprobe search: 1 one dense round, 2 dense + threshold round       unset (0): by nprobe
 *   "round_first", "round_grow", "round_inc"   round schedule of the adaptive search           12, 12 (bytes) / 6 / 3.5, = first
 *   "direct_out"      1 results stored straight into page-locked (D, I), 0 copied at the end  1
 *   "scan_pipelined"  byte-code scan through scan_mfma_thr_kernel (two list blocks in flight per   7
 *                     wave): bit 0 dense rounds, bit 1 threshold rounds, bit 2 threshold rounds with up to 64
 *                     queries per item (scan_mfma_pair_kernel: a chunk is fetched once per 64 queries); 0: scan_mfma_kernel
 *   "plan_fused"      round planning in 3 launches (1) or 7 (0)                               1
 *   "coarse_pick"     large fixed-nprobe calls: coarse rankings from matrix-core distances (2: fp16    2
 *                     operands, 1: fp32) + exact recomputation of the candidates (amd_ivf_last_coarse_pick),
 *                     0 exact distances to every centroid
 *   "phase_timing"    HIP events around every phase (amd_ivf_last_timing): 1 always, 0 never          unset: calls of >= 20 queries
 *   "pinned_io"       per-call inputs / outputs through one page-locked block (1) or copies (0)                 1
 *   "row_lists"       threshold rounds of calls of >= 256 queries: 1 the rows' marked candidates are compacted    unset (-1): 1 when no other
 *                     into short lists before the selection (compact_rows_kernel), 0 the selection walks the masks  search of the index is running
 *   "lanes"           dense rounds of fp32 searches: 1 from a lane-ordered copy of the lists (one coalesced KiB per    1
 *                     64 vectors and 4 dimensions, no staging; the copy costs the lists' bytes once more, built by the
 *                     first fp32 search), 0 from the rows
 *   "fp32_in_flight"  searches of >= 256 queries in fp32 arithmetic that run on the index at a time; further ones      4
 *                     wait inside the call (four is the measured optimum; 0: no limit)
 *   "coalesce"        amd_ivf_submit_adaptive: how many QUEUED tickets one pass over the lists may serve together --   1
 *                     tickets that ask for the same search (parameters, require_acc / ground-truth arrays) over resident
 *                     ranges that follow each other, with result buffers that follow each other in memory.  A pass is
 *                     bound by the list stream, not by the queries probing it, so two queued 5000-query batches cost
 *                     little more than one; every query's result is what its own call would have returned.  Only
 *                     tickets already waiting are joined: a ticket alone in the queue runs alone
 * amd_ivf_set_option(h, key, NAN) returns the key to "unset".  May be called while search contexts of the index are searching: a
 * search reads what shapes its launches once, when it starts, so the change takes effect with the searches that start after it. */
int amd_ivf_set_option(amd_ivf_t* h, const char* key, double value);
int amd_ivf_get_option(amd_ivf_t* h, const char* key, double* value);

/* ------------------------------------------------------------------------------------------------
 * Dataset files of the reference's harness (Auncel/eval/bound.cpp:29-113; host only).  Buffers are malloc'ed
 * here and released with amd_ivf_free.  Where the harness aborts (missing file, size that is not a whole number of rows,
 * short read) these return -2 and amd_ivf_last_error() says why.
 *   read_fvecs / read_ivecs   fvecs_read / ivecs_read (:29-63): every row = int32 d followed by d 4-byte values; *x is n x d.
 *   read_fbin                 fbin_read (:65-109): int32 n, int32 d, then `num` rows (0: n rows) of d values of `bytes`
 *                             bytes each: 4 = fp32; 1 = one byte per value read as a SIGNED char and widened to float,
 *                             exactly as the harness does for its u8 SIFT files.  *n is the header's count, as there.
 *   read_ibin                 ibin_read (:111-113): the same container holding int32 (ground-truth ids). */
int amd_ivf_read_fvecs(const char* path, size_t* d, size_t* n, float** x);
int amd_ivf_read_ivecs(const char* path, size_t* d, size_t* n, int32_t** x);
int amd_ivf_read_fbin(const char* path, size_t num, int bytes, size_t* d, size_t* n, float** x);
int amd_ivf_read_ibin(const char* path, size_t num, size_t* d, size_t* n, int32_t** x);
void amd_ivf_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* AUNCEL_AMD_H */
