// Implementation of the host-side mirror of the reference's classes (namespace faiss) on top of the
// C ABI of include/auncel_amd.h.  No distance, selection or stop-rule arithmetic happens here: this
// file is bookkeeping, argument checking and the host-only pieces the reference also keeps on the
// host (list storage, Trace::SB ordering, the acos table, shard merging, k-means means).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <exception>
#include <typeinfo>
#include <vector>

#include "../../../include/auncel_amd.h"
#include "../kmeans_host.h"
#include "AutoTune.h"
#include "AuxIndexStructures.h"
#include "FaissAssert.h"
#include "Heap.h"
#include "IVF_pro.h"
#include "IndexFlat.h"
#include "IndexIVF.h"
#include "IndexIVFFlat.h"
#include "IndexShards.h"
#include "profile.h"

namespace faiss {

namespace {

void chk(int rc, const char* what) {
    if (rc == 0) return;
    std::string msg = std::string(what) + ": " + amd_ivf_last_error();
    if (rc == -2) throw FaissException(msg);
    throw std::runtime_error(msg);  // HIP failure / no GPU: there is no CPU path to fall back to
}
#define AMD(call) chk(call, #call)

int device_id(int assigned = -1) {
    if (assigned >= 0) return assigned;
    const char* e = getenv("AUNCEL_AMD_DEVICE");
    return e ? atoi(e) : 0;
}

static_assert(sizeof(long) == sizeof(int64_t), "idx_t must be 64-bit");
inline int64_t* i64(long* p) { return reinterpret_cast<int64_t*>(p); }
inline const int64_t* i64(const long* p) { return reinterpret_cast<const int64_t*>(p); }

}  // namespace

// ------------------------------------------------------------------------------------- Index
void Index::add_with_ids(idx_t, const float*, const long*) { FAISS_THROW_MSG("add_with_ids not implemented for this type of index"); }
long Index::remove_ids(const IDSelector&) { FAISS_THROW_MSG("remove_ids not implemented for this type of index"); }

void Index::assign(idx_t n, const float* x, idx_t* labels, idx_t k) {
    std::vector<float> dis(n * k);
    search(n, x, k, dis.data(), labels);
}

// ------------------------------------------------------------------------------------- IndexFlat
IndexFlat::IndexFlat(idx_t d, MetricType metric) : Index(d, metric) {}

IndexFlat::~IndexFlat() {
    if (gpu_) amd_ivf_destroy(gpu_);
}

void IndexFlat::add(idx_t n, const float* x) {
    xb.insert(xb.end(), x, x + n * d);
    ntotal += n;
    gpu_ntotal_ = -1;  // the device copy is stale
    version++;
}

void IndexFlat::reset() {
    xb.clear();
    ntotal = 0;
    gpu_ntotal_ = -1;
    version++;
}

void IndexFlat::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
    FAISS_THROW_IF_NOT_MSG(ntotal > 0, "empty flat index");
    std::lock_guard<std::mutex> lock(gpu_mu_);
    if (!gpu_ || gpu_ntotal_ != ntotal) {
        if (gpu_) amd_ivf_destroy(gpu_);
        gpu_ = nullptr;
        AMD(amd_ivf_create(d, (size_t)ntotal, (int)metric_type, device_id(amd_device), &gpu_));
        AMD(amd_ivf_set_centroids(gpu_, xb.data()));
        gpu_ntotal_ = ntotal;
    }
    AMD(amd_ivf_coarse(gpu_, (size_t)n, x, (size_t)k, distances, i64(labels), coarse_mode));
}

// ------------------------------------------------------------------------------------- lists
size_t ArrayInvertedLists::add_entries(size_t list_no, size_t n_entry, const idx_t* ids_in, const uint8_t* code) {
    if (n_entry == 0) return 0;
    FAISS_THROW_IF_NOT(list_no < nlist);
    size_t o = ids[list_no].size();
    ids[list_no].insert(ids[list_no].end(), ids_in, ids_in + n_entry);
    codes[list_no].insert(codes[list_no].end(), code, code + n_entry * code_size);
    version++;
    return o;
}

void ArrayInvertedLists::update_entries(size_t list_no, size_t offset, size_t n_entry, const idx_t* ids_in, const uint8_t* code) {
    FAISS_THROW_IF_NOT(list_no < nlist && offset + n_entry <= ids[list_no].size());
    memcpy(ids[list_no].data() + offset, ids_in, n_entry * sizeof(idx_t));
    memmove(codes[list_no].data() + offset * code_size, code, n_entry * code_size);  // (an entry may be moved within its own list)
    version++;
}

void ArrayInvertedLists::resize(size_t list_no, size_t new_size) {
    ids[list_no].resize(new_size);
    codes[list_no].resize(new_size * code_size);
    version++;
}

// ------------------------------------------------------------------------------------- error_pro / Trace
float Trace::search(float k, float std_m) {
    const size_t n = trace.size();
    if (k <= trace[0].first) return trace[0].second + std_m * stds[0];
    if (k >= trace[n - 1].first) return (trace[n - 1].second + std_m * stds[n - 1]) * (k / trace[n - 1].first);
    size_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        size_t mid = (lo + hi) / 2;
        if (trace[mid].first < k) lo = mid + 1;
        else hi = mid - 1;
    }
    if (trace[lo].first > k) lo--;
    return trace[lo].second + std_m * stds[lo];
}

void Trace::SB() {
    static_assert(sizeof(std::pair<float, float>) == 2 * sizeof(float), "pair layout");
    const size_t n = trace.size();
    std::vector<float> x(n / bs + 2), y(n / bs + 2), s(n / bs + 2);
    size_t nb = 0;
    AMD(amd_ivf_trace_sb(n ? &trace[0].first : nullptr, n, bs, x.data(), y.data(), s.data(), &nb));
    trace.resize(nb);
    stds.resize(nb);
    for (size_t i = 0; i < nb; i++) {
        trace[i] = std::make_pair(x[i], y[i]);
        stds[i] = s[i];
    }
}

void error_pro::construct_arcos() {
    arcos_list.resize(arcos_size);
    FAISS_THROW_IF_NOT(arcos_size == 500);
    AMD(amd_ivf_arcos_table(arcos_list.data()));
}

float error_pro::arcos(float x) {
    FAISS_THROW_IF_NOT_MSG((x <= 1. && x >= -1.), "arcos's domain definition is [-1, 1]");
    int index = x * arcos_size / 2 + arcos_size / 2;
    return arcos_list[index];
}

void error_pro::train(MetricType) {
    size_t i = 0;
    while ((size_t(1) << i) <= nlist / 8) {
        std::cout << "SB() " << (1 << i) << std::endl;
        traces[i].SB();
        i++;
    }
    traces_version++;
}

void error_pro::setparam(int id_) {
    // same file, same relative path as the reference (IVF_pro.cpp:240-256); silently keeps the
    // defaults when the file is missing, as the reference does
    std::ifstream in("../hyperparameter.txt");
    for (int i = 0; i < 12; i++) {
        float a, b;
        if (!(in >> a >> b)) break;
        if (i == id_ - 1) {
            multipler = a;
            std_m = b;
        }
    }
    profile = false;
}

error_pro::~error_pro() {
    delete[] train_ci;
    delete[] train_cd;
    delete[] my_nprobe;
    delete[] KD;
}

// ------------------------------------------------------------------------------------- Level1Quantizer
Level1Quantizer::Level1Quantizer(Index* quantizer, size_t nlist, bool t)
    : quantizer(quantizer), nlist(nlist), quantizer_trains_alone(0), own_fields(false), clustering_index(nullptr) {
    cp.niter = 25;
    if (t) quantizer->tune = t;
}

Level1Quantizer::Level1Quantizer() : quantizer(nullptr), nlist(0), quantizer_trains_alone(0), own_fields(false), clustering_index(nullptr) {}

Level1Quantizer::~Level1Quantizer() {
    if (own_fields) delete quantizer;
}

// ---- Clustering (Clustering.cpp:36-256, utils.cpp:111-137,229-239,1078-1159)
namespace {
namespace km = amdivf_kmeans;
// centroid = fp32 sum of its points in point order / count; void clusters are split off bigger ones (host version, for
// an assignment index that is not an IndexFlat; the IndexFlat case runs in the engine, amd_ivf_kmeans)
int km_update_centroids(const float* x, float* centroids, const long* assign, size_t d, size_t k, size_t n, size_t k_frozen) {
    k -= k_frozen;
    centroids += k_frozen * d;
    std::vector<size_t> hassign(k);
    memset(centroids, 0, sizeof(*centroids) * d * k);
    for (size_t i = 0; i < n; i++) {
        long ci = assign[i] - (long)k_frozen;
        if (ci < 0) continue;
        float* c = centroids + ci * d;
        hassign[ci]++;
        const float* xi = x + i * d;
        for (size_t j = 0; j < d; j++) c[j] += xi[j];
    }
    for (size_t ci = 0; ci < k; ci++) {
        float* c = centroids + ci * d;
        float ni = (float)hassign[ci];
        if (ni != 0)
            for (size_t j = 0; j < d; j++) c[j] /= ni;
    }
    return km::split_void_clusters(centroids, hassign, d, k, n);
}
}  // namespace

Clustering::Clustering(int d, int k) : d(d), k(k) {}
Clustering::Clustering(int d, int k, const ClusteringParameters& cp) : ClusteringParameters(cp), d(d), k(k) {}

void Clustering::post_process_centroids() { km::post_process(centroids.data(), d, k, spherical, int_centroids); }

void Clustering::train(idx_t nx, const float* x_in, Index& index) {
    FAISS_THROW_IF_NOT_MSG(nx >= (idx_t)k, "Number of training points should be at least as large as number of clusters");
    for (size_t i = 0; i < (size_t)nx * d; i++) FAISS_THROW_IF_NOT_MSG(std::isfinite(x_in[i]), "input contains NaN's or Inf's");
    if (IndexFlat* fl = dynamic_cast<IndexFlat*>(&index)) {
        // the whole procedure on the device (training set resident, assignment + centroid update on the GPU)
        if (nredo == 1 && centroids.empty() && !update_index && fl->d == (int)d) {
            centroids.resize(d * k);
            const size_t o = obj.size();
            obj.resize(o + niter);
            AMD(amd_ivf_kmeans((int)d, (size_t)nx, x_in, k, (int)fl->metric_type, niter, seed, (size_t)max_points_per_centroid,
                               spherical ? 1 : 0, int_centroids ? 1 : 0, fl->coarse_mode, device_id(fl->amd_device), centroids.data(), obj.data() + o));
            if ((size_t)nx == k) obj.resize(o);  // the corner case copies the training set and records no objective
            index.reset();
            index.add(k, centroids.data());
            return;
        }
    }
    const float* x = x_in;
    std::vector<float> sub;
    if ((size_t)nx > k * max_points_per_centroid) {
        if (verbose) printf("Sampling a subset of %ld / %ld for training\n", (long)(k * max_points_per_centroid), (long)nx);
        std::vector<int> perm(nx);
        km::rand_perm(perm.data(), nx, seed);
        nx = k * max_points_per_centroid;
        sub.resize((size_t)nx * d);
        for (idx_t i = 0; i < nx; i++) memcpy(&sub[i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
        x = sub.data();
    } else if ((size_t)nx < k * min_points_per_centroid) {
        fprintf(stderr, "WARNING clustering %ld points to %ld centroids: please provide at least %ld training points\n", (long)nx,
                (long)k, (long)(k * min_points_per_centroid));
    }
    if ((size_t)nx == k) {  // corner case: the training set becomes the centroids
        centroids.assign(x_in, x_in + d * k);
        index.reset();
        index.add(k, x_in);
        return;
    }
    std::vector<idx_t> assign(nx);
    std::vector<float> dis(nx);
    float best_err = HUGE_VALF;
    std::vector<float> best_obj, best_centroids;
    FAISS_THROW_IF_NOT_MSG(centroids.size() % d == 0, "size of provided input centroids not a multiple of dimension");
    const size_t n_input_centroids = centroids.size() / d;
    for (int redo = 0; redo < nredo; redo++) {
        centroids.resize(d * k);
        std::vector<int> perm(nx);
        km::rand_perm(perm.data(), nx, seed + 1 + redo * 15486557L);
        for (size_t i = n_input_centroids; i < k; i++) memcpy(&centroids[i * d], x + (size_t)perm[i] * d, d * sizeof(float));
        post_process_centroids();
        if (index.ntotal != 0) index.reset();
        if (!index.is_trained) index.train(k, centroids.data());
        index.add(k, centroids.data());
        float err = 0;
        for (int it = 0; it < niter; it++) {
            index.search(nx, x, 1, dis.data(), assign.data());  // the GPU part
            err = 0;
            for (idx_t j = 0; j < nx; j++) err += dis[j];
            obj.push_back(err);
            int nsplit = km_update_centroids(x, centroids.data(), assign.data(), d, k, nx, frozen_centroids ? n_input_centroids : 0);
            if (verbose) printf("  Iteration %d: objective=%g nsplit=%d\n", it, err, nsplit);
            post_process_centroids();
            index.reset();
            if (update_index) index.train(k, centroids.data());
            index.add(k, centroids.data());
        }
        if (nredo > 1) {
            if (err < best_err) {
                best_centroids = centroids;
                best_obj = obj;
                best_err = err;
            }
            index.reset();
        }
    }
    if (nredo > 1) {
        centroids = best_centroids;
        obj = best_obj;
        index.reset();
        index.add(k, best_centroids.data());
    }
}

float kmeans_clustering(size_t d, size_t n, size_t k, const float* x, float* centroids) {
    Clustering clus((int)d, (int)k);
    IndexFlatL2 index(d);
    clus.train(n, x, index);
    memcpy(centroids, clus.centroids.data(), sizeof(*centroids) * d * k);
    return clus.obj.back();
}

void Level1Quantizer::train_q1(size_t n, const float* x, bool verbose, MetricType metric_type) {
    const size_t d = quantizer->d;
    if (quantizer->is_trained && ((size_t)quantizer->ntotal == nlist)) {
        if (verbose) printf("IVF quantizer does not need training.\n");
        return;
    }
    FAISS_THROW_IF_NOT_MSG(quantizer_trains_alone == 0, "only k-means training of a flat quantizer is supported");
    if (verbose) printf("Training level-1 quantizer on %ld vectors in %ldD\n", (long)n, (long)d);
    Clustering clus((int)d, (int)nlist, cp);
    quantizer->reset();
    if (clustering_index) {
        clus.train(n, x, *clustering_index);
        quantizer->add(nlist, clus.centroids.data());
    } else {
        clus.train(n, x, *quantizer);
    }
    const std::vector<float>& cen = clus.centroids;
    if (quantizer->tune) {
        // centroid-to-centroid table of the Auncel geometry (IndexIVF.cpp:97-111), computed on the GPU
        amd_ivf* g = nullptr;
        AMD(amd_ivf_create((int)d, nlist, (int)metric_type, device_id(quantizer->amd_device), &g));
        interdis_cem.resize(nlist * (nlist - 1) / 2);
        int rc = amd_ivf_set_centroids(g, cen.data());
        if (rc == 0) rc = amd_ivf_set_interdis(g, nullptr);
        if (rc == 0) rc = amd_ivf_get_interdis(g, interdis_cem.data());
        amd_ivf_destroy(g);
        chk(rc, "interdis_cem");
    }
    quantizer->is_trained = true;
}

// ------------------------------------------------------------------------------------- IndexIVF
IndexIVFStats indexIVF_stats;
void IndexIVFStats::reset() { memset((void*)this, 0, sizeof(*this)); }

IndexIVF::IndexIVF(Index* quantizer, size_t d, size_t nlist, size_t code_size, MetricType metric)
    : Index(d, metric),
      Level1Quantizer(quantizer, nlist),
      invlists(new ArrayInvertedLists(nlist, code_size)),
      own_invlists(true),
      t(nullptr),
      code_size(code_size),
      nprobe(1),
      max_codes(0),
      parallel_mode(0),
      maintain_direct_map(false) {
    FAISS_THROW_IF_NOT(d == (size_t)quantizer->d);
    is_trained = quantizer->is_trained && ((size_t)quantizer->ntotal == nlist);
    if (metric_type == METRIC_INNER_PRODUCT) cp.spherical = true;
    type = IVF;
    tune = false;
}

IndexIVF::IndexIVF() : invlists(nullptr), own_invlists(false), t(nullptr), code_size(0), nprobe(1), max_codes(0), parallel_mode(0), maintain_direct_map(false) {
    type = IVF;
    tune = false;
}

IndexIVF::~IndexIVF() {
    if (own_invlists) delete invlists;
    if (gpu_) amd_ivf_destroy(gpu_);
}

void IndexIVF::set_tune_mode() {
    tune = true;
    quantizer->tune = true;
}
void IndexIVF::set_tune_off() {
    tune = false;
    quantizer->tune = false;
}
void IndexIVF::set_train_mode() {
    training = true;
    quantizer->tune = true;
}
void IndexIVF::set_train_off() {
    training = false;
    quantizer->tune = false;
}

void IndexIVF::init_tune(size_t train_num, size_t topk, const float* train_q, const float* train_D, const long* train_I,
                         float* train_cd, long* train_ci) {
    t = new error_pro;
    t->construct_arcos();
    for (size_t np = 1; np <= nlist / 8; np <<= 1) {
        Trace tr;
        tr.nprobe = np;
        tr.trace.assign((topk / 4) * train_num, std::make_pair(-1.f, -1.f));
        t->traces.push_back(tr);
    }
    t->m_type = metric_type == METRIC_INNER_PRODUCT ? error_pro::IP : error_pro::L2;
    t->count = 0;
    t->nlist = nlist;
    t->max_topk = topk;
    t->d = d;
    t->interdis_cem = interdis_cem.data();
    t->train_num = train_num;
    t->train_q = train_q;
    t->train_D = train_D;
    t->train_I = train_I;
    t->train_cd = train_cd;
    t->train_ci = train_ci;
}

void IndexIVF::reset() {
    direct_map.clear();
    invlists->reset();
    ntotal = 0;
}

void IndexIVF::train(idx_t n, const float* x) {
    train_q1(n, x, verbose, metric_type);
    is_trained = true;
}

void IndexIVF::add(idx_t n, const float* x) { add_with_ids(n, x, nullptr); }

void IndexIVF::set_device(int device) {
    FAISS_THROW_IF_NOT_MSG(gpu_ == nullptr || device == amd_device, "the index already lives on another device");
    amd_device = device;
    // a quantizer that already has a device keeps it (shards share one: its rankings come back to the host either way)
    if (quantizer && quantizer->amd_device < 0) quantizer->set_device(device);
}

bool IndexIVF::device_bound() const { return gpu_ != nullptr; }

void IndexIVF::replace_invlists(InvertedLists* il, bool own) {
    if (own_invlists) delete invlists;
    invlists = il;
    own_invlists = own;
    lists_version_ = (size_t)-1;
}

amd_ivf* IndexIVF::engine() const {
    sync_engine(false);
    return gpu_;
}

void IndexIVF::set_engine_option(const char* key, double value) {
    if (!gpu_) AMD(amd_ivf_create(d, nlist, (int)metric_type, device_id(amd_device), &gpu_));
    AMD(amd_ivf_set_option(gpu_, key, value));
}

void IndexIVF::sync_engine(bool need_tuner) const {
    if (!gpu_) AMD(amd_ivf_create(d, nlist, (int)metric_type, device_id(amd_device), &gpu_));
    if (applied_ties_ != coarse_tie_order || applied_select_ != selection) {
        AMD(amd_ivf_set_option(gpu_, "coarse_ties", coarse_tie_order < 0 ? std::nan("") : (double)coarse_tie_order));
        AMD(amd_ivf_set_option(gpu_, "select", selection < 0 ? std::nan("") : (double)selection));
        applied_ties_ = coarse_tie_order;
        applied_select_ = selection;
    }
    const IndexFlat* qf = dynamic_cast<const IndexFlat*>(quantizer);
    FAISS_THROW_IF_NOT_MSG(qf != nullptr, "the MI355X engine needs a flat coarse quantizer");
    FAISS_THROW_IF_NOT_MSG((size_t)qf->ntotal == nlist, "quantizer is not trained (ntotal != nlist)");
    // (the quantizer's version, not just its size: retraining with the same nlist must reach the device too)
    if (centroid_count_ != nlist || centroid_version_ != qf->version) {
        AMD(amd_ivf_set_centroids(gpu_, qf->xb.data()));
        centroid_count_ = nlist;
        centroid_version_ = qf->version;
        interdis_uploaded_ = nullptr;  // the centroid table belongs to the old centroids
    }
    if (lists_version_ != invlists->version) {
        FAISS_THROW_IF_NOT_MSG(code_size == sizeof(float) * d, "IVF-Flat codes expected");
        std::vector<size_t> sizes(nlist);
        std::vector<const float*> codes(nlist);
        std::vector<const int64_t*> ids(nlist);
        for (size_t l = 0; l < nlist; l++) {
            sizes[l] = invlists->list_size(l);
            codes[l] = reinterpret_cast<const float*>(invlists->get_codes(l));
            ids[l] = i64(invlists->get_ids(l));
        }
        AMD(amd_ivf_set_lists(gpu_, sizes.data(), codes.data(), ids.data()));
        lists_version_ = invlists->version;
        resident_ptr_ = nullptr;  // nothing else to refresh, but keep the invariant explicit
    }
    // interdis_cem is a public vector the caller may refill in place (train_q1 does): pointer and size alone would miss
    // that, so three of its values are remembered as well
    const size_t isz = interdis_cem.size();
    const float print[3] = {isz ? interdis_cem[0] : 0.f, isz ? interdis_cem[isz / 2] : 0.f, isz ? interdis_cem[isz - 1] : 0.f};
    if (isz && (interdis_uploaded_ != interdis_cem.data() || interdis_size_ != isz || memcmp(print, interdis_print_, sizeof(print)) != 0)) {
        AMD(amd_ivf_set_interdis(gpu_, interdis_cem.data()));
        interdis_uploaded_ = interdis_cem.data();
        interdis_size_ = isz;
        memcpy(interdis_print_, print, sizeof(print));
    }
    if (need_tuner) {
        FAISS_THROW_IF_NOT_MSG((t != nullptr), "Search tune start can't start without IVF_pro init and training");
        FAISS_THROW_IF_NOT_MSG(!interdis_cem.empty(), "interdis_cem missing: train() the index in tune mode first");
        if (traces_version_ != t->traces_version) {
            const size_t nt = t->traces.size();
            std::vector<size_t> len(nt);
            std::vector<std::vector<float>> xs(nt), ys(nt);
            std::vector<const float*> px(nt), py(nt), ps(nt);
            for (size_t i = 0; i < nt; i++) {
                const Trace& tr = t->traces[i];
                FAISS_THROW_IF_NOT_MSG(tr.stds.size() == tr.trace.size(), "traces are not trained (call error_pro::train)");
                len[i] = tr.trace.size();
                xs[i].resize(len[i]);
                ys[i].resize(len[i]);
                for (size_t j = 0; j < len[i]; j++) {
                    xs[i][j] = tr.trace[j].first;
                    ys[i][j] = tr.trace[j].second;
                }
                px[i] = xs[i].data();
                py[i] = ys[i].data();
                ps[i] = tr.stds.data();
            }
            AMD(amd_ivf_set_tuner(gpu_, t->max_topk, nt, len.data(), px.data(), py.data(), ps.data(), t->arcos_list.data()));
            traces_version_ = t->traces_version;
        }
    }
}

void IndexIVF::set_resident_queries(const float* x, size_t n) const {
    sync_engine(false);
    AMD(amd_ivf_set_queries(gpu_, n, x));
    resident_ptr_ = x;
    resident_n_ = n;
}

void IndexIVF::fold_stats() const {
    size_t st[4];
    amd_ivf_stats(gpu_, st, 1);
    indexIVF_stats.nq += st[0];
    indexIVF_stats.nlist += st[1];
    indexIVF_stats.ndis += st[2];
    indexIVF_stats.nheap_updates += st[3];
}

void IndexIVF::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
    if (tune || training || (t && t->time_tune)) {
        search(n, x, k, distances, labels, (size_t)0);
        return;
    }
    sync_engine(false);
    AMD(amd_ivf_search(gpu_, (size_t)n, x, (size_t)k, nprobe, coarse_mode, distances, i64(labels)));
    fold_stats();
}

void IndexIVF::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, size_t offset) const {
    if (!tune && !training) {
        sync_engine(false);
        if (t && t->time_tune) {
            // IndexIVF.cpp:504-506,545-549: the plain probe loop, left when the budget t->require_acc[id_q] (ms) is used up
            const bool res = resident_ptr_ && x == resident_ptr_ + offset * (size_t)d && offset + n <= resident_n_;
            if (res)
                AMD(amd_ivf_search_timed(gpu_, offset, (size_t)n, (size_t)k, nprobe, t->require_acc, coarse_mode, nullptr, distances, i64(labels)));
            else
                AMD(amd_ivf_search_timed_x(gpu_, (size_t)n, x, offset, (size_t)k, nprobe, t->require_acc, coarse_mode, nullptr, distances, i64(labels)));
        } else {
            AMD(amd_ivf_search(gpu_, (size_t)n, x, (size_t)k, nprobe, coarse_mode, distances, i64(labels)));
        }
        fold_stats();
        return;
    }
    FAISS_THROW_IF_NOT_MSG((t != nullptr), "Search tune start can't start without IVF_pro init and training");
    FAISS_THROW_IF_NOT_MSG(nprobe == nlist, "tune / train mode probes in full coarse order: set nprobe = nlist (profile.cpp:220)");
    FAISS_THROW_IF_NOT_MSG((size_t)k == t->max_topk, "tune / train mode needs k == max_topk (IndexIVF.cpp:560-561)");
    // queries that are rows [offset, offset+n) of the registered matrix are already in HBM
    bool resident = resident_ptr_ && x == resident_ptr_ + offset * (size_t)d && offset + n <= resident_n_;
    if (training) {
        sync_engine(false);
        std::vector<float*> raw(t->traces.size());
        for (size_t i = 0; i < raw.size(); i++) raw[i] = &t->traces[i].trace[0].first;
        if (resident)
            AMD(amd_ivf_train_samples(gpu_, offset, (size_t)n, (size_t)k, t->train_D, t->train_num, coarse_mode, raw.data(), distances, i64(labels)));
        else
            AMD(amd_ivf_train_samples_x(gpu_, (size_t)n, x, offset, (size_t)k, t->train_D, t->train_num, coarse_mode, raw.data(), distances, i64(labels)));
        return;
    }
    sync_engine(true);
    static_assert(sizeof(size_t) == sizeof(uint64_t), "my_nprobe layout");
    uint64_t* np = reinterpret_cast<uint64_t*>(t->my_nprobe);
    // error_pro::overhead_profile (IVF_pro.h:95; IndexIVF.cpp:529-539,614,634-637): rule on every probe, no stop before nlist / 8
    const int flags = (t->profile ? 1 : 0) | (t->overhead_profile ? 2 : 0);
    if (resident)
        AMD(amd_ivf_search_adaptive(gpu_, offset, (size_t)n, t->query_topk, t->multipler, t->std_m, t->require_acc, t->train_D,
                                    flags, coarse_mode, np, t->t_recalls, distances, i64(labels)));
    else
        AMD(amd_ivf_search_adaptive_x(gpu_, (size_t)n, x, offset, t->query_topk, t->multipler, t->std_m, t->require_acc, t->train_D,
                                      flags, coarse_mode, np, t->t_recalls, distances, i64(labels)));
    fold_stats();
    if (t->overhead_profile) {
        // The reference sums the time of its scan_one_list calls inside the loop and prints it (IndexIVF.cpp:679-680) for
        // eval/overhead.cpp to set against the whole search ("With ELP").  Kernels of a batch cannot be split that way: the
        // figure here is the same search without the rule -- the plain probe loop over nlist / 8 probes, k = max_topk --
        // timed on the host.  Its results equal the ones just returned (same probes, same heap); they are discarded.
        std::vector<float> D2((size_t)n * k);
        std::vector<idx_t> I2((size_t)n * k);
        const auto t0 = std::chrono::steady_clock::now();
        if (resident) AMD(amd_ivf_search_resident(gpu_, offset, (size_t)n, (size_t)k, nlist / 8, coarse_mode, D2.data(), i64(I2.data())));
        else AMD(amd_ivf_search(gpu_, (size_t)n, x, (size_t)k, nlist / 8, coarse_mode, D2.data(), i64(I2.data())));
        const double overh = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        size_t st[4];
        amd_ivf_stats(gpu_, st, 1);  // (not part of the search the caller asked for)
        printf("Without ELP search Time: %.3f s\n", overh);
    }
}

void IndexIVF::search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* keys, const float* coarse_dis,
                                  float* distances, idx_t* labels, bool store_pairs, const IVFSearchParameters* params) const {
    const idx_t offset = (k >> 32) & 0xffffffff;  // the reference packs the query offset into k (IndexIVF.cpp:389-392)
    k = k & 0xffffffff;
    (void)offset;
    FAISS_THROW_IF_NOT_MSG(!tune && !training,
                           "tune / train mode runs through IndexIVF::search(n, x, k, D, I, offset): the engine ranks the "
                           "centroids itself");
    const size_t np = params ? params->nprobe : nprobe;
    const size_t mc = params ? params->max_codes : max_codes;
    sync_engine(false);
    AMD(amd_ivf_search_preassigned(gpu_, (size_t)n, x, (size_t)k, np, i64(keys), coarse_dis, distances, i64(labels),
                                   store_pairs ? 1 : 0, mc));
    fold_stats();
}

// ---- stored vectors back (IndexIVF.cpp:305-328,869-945, IndexIVFFlat.cpp:226-230)
void Index::reconstruct(idx_t, float*) const { FAISS_THROW_MSG("reconstruct not implemented for this type of index"); }
void Index::reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    for (idx_t i = 0; i < ni; i++) reconstruct(i0 + i, recons + i * d);
}
void Index::search_and_reconstruct(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, float* recons) const {
    search(n, x, k, distances, labels);
    for (idx_t i = 0; i < n; ++i)
        for (idx_t j = 0; j < k; ++j) {
            const idx_t ij = i * k + j, key = labels[ij];
            float* reconstructed = recons + ij * d;
            if (key < 0) memset(reconstructed, -1, sizeof(*reconstructed) * d);  // NaNs, as the reference fills them
            else reconstruct(key, reconstructed);
        }
}

void IndexIVF::make_direct_map(bool new_maintain_direct_map) {
    if (new_maintain_direct_map == maintain_direct_map) return;
    if (new_maintain_direct_map) {
        direct_map.resize(ntotal, -1);
        for (size_t key = 0; key < nlist; key++) {
            const size_t list_size = invlists->list_size(key);
            const idx_t* idlist = invlists->get_ids(key);
            for (long ofs = 0; ofs < (long)list_size; ofs++) {
                FAISS_THROW_IF_NOT_MSG(0 <= idlist[ofs] && idlist[ofs] < ntotal, "direct map supported only for seuquential ids");
                direct_map[idlist[ofs]] = (long)key << 32 | ofs;
            }
        }
    } else {
        direct_map.clear();
    }
    maintain_direct_map = new_maintain_direct_map;
}

void IndexIVF::reconstruct(idx_t key, float* recons) const {
    FAISS_THROW_IF_NOT_MSG((idx_t)direct_map.size() == ntotal, "direct map is not initialized");
    FAISS_THROW_IF_NOT_MSG(key >= 0 && key < (idx_t)direct_map.size(), "invalid key");
    reconstruct_from_offset(direct_map[key] >> 32, direct_map[key] & 0xffffffff, recons);
}

void IndexIVF::reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    FAISS_THROW_IF_NOT(ni == 0 || (i0 >= 0 && i0 + ni <= ntotal));
    for (size_t list_no = 0; list_no < nlist; list_no++) {
        const size_t list_size = invlists->list_size(list_no);
        const idx_t* idlist = invlists->get_ids(list_no);
        for (size_t offset = 0; offset < list_size; offset++) {
            const idx_t id = idlist[offset];
            if (!(id >= i0 && id < i0 + ni)) continue;
            reconstruct_from_offset((idx_t)list_no, (idx_t)offset, recons + (id - i0) * d);
        }
    }
}

void IndexIVF::search_and_reconstruct(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, float* recons) const {
    std::vector<idx_t> idx(n * nprobe);
    std::vector<float> coarse_dis(n * nprobe);
    quantizer->search(n, x, nprobe, coarse_dis.data(), idx.data());
    // store_pairs: (list_no << 32 | offset) instead of ids, to find the codes
    search_preassigned(n, x, k, idx.data(), coarse_dis.data(), distances, labels, true);
    for (idx_t i = 0; i < n; ++i)
        for (idx_t j = 0; j < k; ++j) {
            const idx_t ij = i * k + j, key = labels[ij];
            float* reconstructed = recons + ij * d;
            if (key < 0) {
                memset(reconstructed, -1, sizeof(*reconstructed) * d);
            } else {
                const long list_no = key >> 32, offset = key & 0xffffffff;
                labels[ij] = invlists->get_ids(list_no)[offset];
                reconstruct_from_offset(list_no, offset, reconstructed);
            }
        }
}

void IndexIVF::reconstruct_from_offset(idx_t, idx_t, float*) const { FAISS_THROW_MSG("reconstruct_from_offset not implemented"); }

void IndexIVFFlat::reconstruct_from_offset(idx_t list_no, idx_t offset, float* recons) const {
    memcpy(recons, invlists->get_codes(list_no) + (size_t)offset * code_size, code_size);
}

// ---- range search (IndexIVF.cpp:740-857, AuxIndexStructures.cpp:26-60)
RangeSearchResult::RangeSearchResult(idx_t nq, bool alloc_lims) : nq((size_t)nq), lims(nullptr), labels(nullptr), distances(nullptr), buffer_size(1024 * 256) {
    if (alloc_lims) {
        lims = new size_t[nq + 1];
        memset(lims, 0, sizeof(*lims) * (nq + 1));
    }
}
void RangeSearchResult::do_allocation() {
    // lims holds one COUNT per query on entry and the offsets on return (AuxIndexStructures.cpp:40-50)
    size_t ofs = 0;
    for (size_t i = 0; i < nq; i++) {
        const size_t c = lims[i];
        lims[i] = ofs;
        ofs += c;
    }
    lims[nq] = ofs;
    labels = new idx_t[ofs];
    distances = new float[ofs];
}
RangeSearchResult::~RangeSearchResult() {
    delete[] labels;
    delete[] distances;
    delete[] lims;
}

void Index::range_search(idx_t, const float*, float, RangeSearchResult*) const {
    FAISS_THROW_MSG("range search not implemented");
}

void IndexIVF::range_search_preassigned(idx_t nx, const float* x, float radius, const idx_t* keys, const float*,
                                        RangeSearchResult* result) const {
    sync_engine(false);
    AMD(amd_ivf_range_search_preassigned(gpu_, (size_t)nx, x, radius, nprobe, i64(keys), result->lims));
    for (idx_t i = 0; i < nx; i++) result->lims[i] = result->lims[i + 1] - result->lims[i];  // (the engine's offsets -> counts)
    result->do_allocation();
    AMD(amd_ivf_range_results(gpu_, i64(result->labels), result->distances));
    fold_stats();
}

void IndexIVF::range_search(idx_t nx, const float* x, float radius, RangeSearchResult* result) const {
    sync_engine(false);
    AMD(amd_ivf_range_search(gpu_, (size_t)nx, x, radius, nprobe, coarse_mode, result->lims));
    for (idx_t i = 0; i < nx; i++) result->lims[i] = result->lims[i + 1] - result->lims[i];
    result->do_allocation();
    AMD(amd_ivf_range_results(gpu_, i64(result->labels), result->distances));
    fold_stats();
}

// ---- the scanners' range results (Auncel/AuxIndexStructures.cpp:65-215): host-side bookkeeping, blocks that never move
BufferList::BufferList(size_t buffer_size) : buffer_size(buffer_size), wp(buffer_size) {}
BufferList::~BufferList() {
    for (Buffer& b : buffers) {
        delete[] b.ids;
        delete[] b.dis;
    }
}
void BufferList::append_buffer() {
    buffers.push_back(Buffer{new idx_t[buffer_size], new float[buffer_size]});
    wp = 0;
}
void BufferList::add(idx_t id, float dis) {
    if (wp == buffer_size) append_buffer();
    buffers.back().ids[wp] = id;
    buffers.back().dis[wp] = dis;
    wp++;
}
void BufferList::copy_range(size_t ofs, size_t n, idx_t* dest_ids, float* dest_dis) {
    for (size_t b = ofs / buffer_size, at = ofs % buffer_size; n > 0; b++, at = 0) {
        const size_t take = std::min(n, buffer_size - at);
        memcpy(dest_ids, buffers[b].ids + at, take * sizeof(idx_t));
        memcpy(dest_dis, buffers[b].dis + at, take * sizeof(float));
        dest_ids += take;
        dest_dis += take;
        n -= take;
    }
}
void RangeQueryResult::add(float dis, idx_t id) {
    nres++;
    pres->add(id, dis);
}
RangeSearchPartialResult::RangeSearchPartialResult(RangeSearchResult* res_in) : BufferList(res_in->buffer_size), res(res_in) {}
RangeQueryResult& RangeSearchPartialResult::new_result(idx_t qno) {
    queries.push_back(RangeQueryResult{qno, 0, this});
    return queries.back();
}
void RangeSearchPartialResult::set_lims() {
    for (const RangeQueryResult& q : queries) res->lims[q.qno] = q.nres;
}
void RangeSearchPartialResult::copy_result(bool incremental) {
    size_t ofs = 0;
    for (const RangeQueryResult& q : queries) {
        copy_range(ofs, q.nres, res->labels + res->lims[q.qno], res->distances + res->lims[q.qno]);
        if (incremental) res->lims[q.qno] += q.nres;
        ofs += q.nres;
    }
}
void RangeSearchPartialResult::finalize() {
    // (the reference runs this inside an OpenMP team, with barriers around the one allocation: AuxIndexStructures.cpp:145-155)
    set_lims();
    res->do_allocation();
    copy_result();
}
void RangeSearchPartialResult::merge(std::vector<RangeSearchPartialResult*>& partial_results, bool do_delete) {
    if (partial_results.empty()) return;
    RangeSearchResult* result = partial_results[0]->res;
    const size_t nx = result->nq;
    for (const RangeSearchPartialResult* p : partial_results)
        if (p)
            for (const RangeQueryResult& q : p->queries) result->lims[q.qno] += q.nres;
    // (AuxIndexStructures.cpp:194-215: counts -> offsets by do_allocation, entries copied with incremental = true -- which leaves
    // every offset one query ahead -- and the table shifted back)
    result->do_allocation();
    for (RangeSearchPartialResult*& p : partial_results) {
        if (!p) continue;
        p->copy_result(true);
        if (do_delete) {
            delete p;
            p = nullptr;
        }
    }
    for (size_t i = nx; i > 0; i--) result->lims[i] = result->lims[i - 1];
    result->lims[0] = 0;
}

void InvertedListScanner::scan_codes_range(size_t, const uint8_t*, const idx_t*, float, RangeQueryResult&) const {
    FAISS_THROW_MSG("scan_codes_range not implemented");
}

/// The scanner of IndexIVFFlat over the engine (IVFFlatScanner, Auncel/IndexIVFFlat.cpp:95-158).  The codes it is handed live in HBM
/// already, so a `codes` pointer only NAMES a run of the current list -- [offset, offset + n), anywhere inside it.  Every scanner
/// searches on a context of its own (amd_ivf_clone), so scanners of one index may run in different threads, as the reference's
/// do ("distance_to_code and scan_codes can be called in multiple threads", IndexIVF.h:312-315; tests/test_lowlevel_ivf.cpp:426-564).
struct EngineScanner : InvertedListScanner {
    const IndexIVF* ix;
    bool store_pairs;
    std::vector<float> q;
    idx_t list_no = -1;
    mutable amd_ivf* ctx = nullptr;
    mutable size_t ctx_version = (size_t)-1;
    EngineScanner(const IndexIVF* ix, bool sp) : ix(ix), store_pairs(sp) {}
    ~EngineScanner() override {
        if (ctx) amd_ivf_destroy(ctx);
    }
    void set_query(const float* query) override { q.assign(query, query + ix->d); }
    void set_list(idx_t l, float) override { list_no = l; }
    /// this scanner's search context over the index as it is now
    amd_ivf* context() const {
        static std::mutex mu;  // (the index's own handle is brought up to date by one scanner at a time)
        std::lock_guard<std::mutex> lock(mu);
        ix->sync_engine(false);
        if (ctx && ctx_version != ix->invlists->version) {
            amd_ivf_destroy(ctx);
            ctx = nullptr;
        }
        if (!ctx) {
            AMD(amd_ivf_clone(ix->gpu_, &ctx));
            ctx_version = ix->invlists->version;
        }
        return ctx;
    }
    /// where in the current list the n codes at `codes` are
    size_t offset_of(size_t n, const uint8_t* codes) const {
        FAISS_THROW_IF_NOT_MSG(list_no >= 0, "set_list first");
        const uint8_t* base = ix->invlists->get_codes(list_no);
        const size_t sz = ix->invlists->list_size(list_no);
        FAISS_THROW_IF_NOT_MSG(n == 0 || (codes >= base && codes < base + sz * ix->code_size && (size_t)(codes - base) % ix->code_size == 0),
                               "codes must point at a code of the current list (the lists live in HBM)");
        const size_t offset = n ? (size_t)(codes - base) / ix->code_size : 0;
        FAISS_THROW_IF_NOT_MSG(offset + n <= sz, "codes run past the end of the current list");
        return offset;
    }
    float distance_to_code(const uint8_t* code) const override {
        const size_t offset = offset_of(1, code);
        float dis = 0;
        AMD(amd_ivf_distance_to_code(context(), q.data(), (size_t)list_no, offset, &dis));
        return dis;
    }
    size_t scan_codes(size_t n, const uint8_t* codes, const idx_t* ids, float* simi, idx_t* idxi, size_t k) const override {
        const size_t offset = offset_of(n, codes);
        if (n == 0) return 0;
        size_t nup = 0;
        const idx_t* own = ix->invlists->get_ids(list_no);
        if (store_pairs || ids == own + offset) {
            // labels straight from the engine: list_no << 32 | j (j from `codes`), or the ids stored with those vectors
            AMD(amd_ivf_scan_codes_at(context(), q.data(), (size_t)list_no, offset, n, store_pairs ? 1 : 0, k, simi, i64(idxi), &nup));
            return nup;
        }
        // the caller's own id array (the reference reads ids[j], whatever it is handed): the entries already in the heap travel as
        // tags, the admitted ones come back as positions
        std::vector<idx_t> lab(k);
        for (size_t i = 0; i < k; i++) lab[i] = -(idx_t)i - 2;
        AMD(amd_ivf_scan_codes_at(context(), q.data(), (size_t)list_no, offset, n, 1, k, simi, i64(lab.data()), &nup));
        std::vector<idx_t> old(idxi, idxi + k);
        for (size_t i = 0; i < k; i++) idxi[i] = lab[i] < 0 ? old[(size_t)(-lab[i] - 2)] : ids[lab[i] & 0xffffffffll];
        return nup;
    }
    void scan_codes_range(size_t n, const uint8_t* codes, const idx_t* ids, float radius, RangeQueryResult& res) const override {
        const size_t offset = offset_of(n, codes);
        if (n == 0) return;
        size_t count = 0;
        amd_ivf* c = context();
        AMD(amd_ivf_scan_codes_range(c, q.data(), (size_t)list_no, offset, n, radius, &count));
        if (!count) return;
        std::vector<uint32_t> pos(count);
        std::vector<float> dis(count);
        AMD(amd_ivf_scan_codes_range_results(c, pos.data(), dis.data()));
        for (size_t i = 0; i < count; i++) res.add(dis[i], store_pairs ? (idx_t)((idx_t)list_no << 32 | (idx_t)pos[i]) : ids[pos[i]]);
    }
};

InvertedListScanner* IndexIVF::get_InvertedListScanner(bool store_pairs) const { return new EngineScanner(this, store_pairs); }

// ------------------------------------------------------------------------------------- IndexIVFFlat
IndexIVFFlat::IndexIVFFlat(Index* quantizer, size_t d, size_t nlist, MetricType metric)
    : IndexIVF(quantizer, d, nlist, sizeof(float) * d, metric) {
    code_size = sizeof(float) * d;
}

void IndexIVFFlat::add_with_ids(idx_t n, const float* x, const long* xids) { add_core(n, x, xids, nullptr); }

void IndexIVFFlat::add_core(idx_t n, const float* x, const long* xids, const long* precomputed_idx) {
    FAISS_THROW_IF_NOT(is_trained);
    FAISS_THROW_IF_NOT_MSG(!(maintain_direct_map && xids), "cannot have direct map and add with ids");
    std::vector<long> own;
    const long* idx = precomputed_idx;
    if (!idx) {
        own.resize(n);
        quantizer->assign(n, x, own.data());
        idx = own.data();
    }
    long n_add = 0;
    for (idx_t i = 0; i < n; i++) {
        long id = xids ? xids[i] : ntotal + i;
        long list_no = idx[i];
        if (list_no < 0) continue;
        size_t o = invlists->add_entry(list_no, id, reinterpret_cast<const uint8_t*>(x + i * d));
        if (maintain_direct_map) direct_map.push_back(list_no << 32 | (long)o);
        n_add++;
    }
    if (verbose) printf("IndexIVFFlat::add_core: added %ld / %ld vectors\n", n_add, (long)n);
    ntotal += n;
}

// ------------------------------------------------------------------------------------- IndexIVFFlatDedup
// Auncel/IndexIVFFlat.cpp:233-380
IndexIVFFlatDedup::IndexIVFFlatDedup(Index* quantizer, size_t d, size_t nlist_, MetricType metric) : IndexIVFFlat(quantizer, d, nlist_, metric) {}

namespace {
// utils.cpp:1584-1593 (the classic string hash): keys the training-set dedup
uint64_t bytes_hash(const uint8_t* p, long n) {
    uint64_t h = (uint64_t)p[0] << 7;
    for (long i = 0; i < n; i++) h = (1000003 * h) ^ p[i];
    return h ^ (uint64_t)n;
}
}  // namespace

void IndexIVFFlatDedup::train(idx_t n, const float* x) {
    // keep the first copy of every vector; as in the reference a hash remembers only the last vector that produced it
    std::unordered_map<uint64_t, idx_t> last_with_hash;
    std::vector<float> uniq((size_t)n * d);
    long nu = 0;
    for (idx_t i = 0; i < n; i++) {
        const float* xi = x + i * d;
        const uint64_t hv = bytes_hash(reinterpret_cast<const uint8_t*>(xi), (long)code_size);
        auto it = last_with_hash.find(hv);
        if (it != last_with_hash.end() && !memcmp(uniq.data() + it->second * d, xi, code_size)) continue;
        last_with_hash[hv] = nu;
        memcpy(uniq.data() + nu * d, xi, code_size);
        nu++;
    }
    if (verbose) printf("IndexIVFFlatDedup::train: train on %ld points after dedup (was %ld points)\n", nu, (long)n);
    IndexIVFFlat::train(nu, uniq.data());
}

void IndexIVFFlatDedup::add_with_ids(idx_t na, const float* x, const long* xids) {
    FAISS_THROW_IF_NOT(is_trained);
    FAISS_THROW_IF_NOT_MSG(!maintain_direct_map, "IVFFlatDedup not implemented with direct_map");
    std::vector<long> list_of(na);
    quantizer->assign(na, x, list_of.data());
    long n_add = 0, n_dup = 0;
    for (idx_t i = 0; i < na; i++) {
        const idx_t id = xids ? xids[i] : ntotal + i;
        const long list_no = list_of[i];
        if (list_no < 0) continue;
        const uint8_t* xi = reinterpret_cast<const uint8_t*>(x + i * d);
        // first stored entry of the list holding the same bytes
        const uint8_t* codes = invlists->get_codes(list_no);
        const long ls = (long)invlists->list_size(list_no);
        long found = -1;
        for (long o = 0; o < ls && found < 0; o++)
            if (!memcmp(codes + (size_t)o * code_size, xi, code_size)) found = o;
        if (found < 0) {
            invlists->add_entry(list_no, id, xi);
        } else {
            instances.insert(std::make_pair(invlists->get_ids(list_no)[found], id));
            n_dup++;
        }
        n_add++;
    }
    if (verbose) printf("IndexIVFFlat::add_with_ids: added %ld / %ld vectors (out of which %ld are duplicates)\n", n_add, (long)na, n_dup);
    ntotal += n_add;
}

// From the first result that has copies on, every entry is followed by its copies at the same distance until the row
// is full (IndexIVFFlat.cpp:340-376).
void IndexIVFFlatDedup::expand_instances(idx_t n, idx_t k, float* distances, idx_t* labels) const {
    std::vector<idx_t> lab(k);
    std::vector<float> dis(k);
    for (idx_t i = 0; i < n; i++) {
        idx_t* li = labels + i * k;
        float* di = distances + i * k;
        idx_t first = 0;
        while (first < k && instances.find(li[first]) == instances.end()) first++;
        if (first == k) continue;
        idx_t w = first, r = first;  // write position in the expanded row, read position in the search result
        while (w < k) {
            auto range = instances.equal_range(li[r]);
            lab[w] = li[r];
            dis[w] = di[r];
            w++;
            for (auto it = range.first; w < k && it != range.second; ++it) {
                lab[w] = it->second;
                dis[w] = di[r];
                w++;
            }
            r++;
        }
        std::copy(lab.begin() + first, lab.end(), li + first);
        std::copy(dis.begin() + first, dis.end(), di + first);
    }
}

void IndexIVFFlatDedup::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
    IndexIVFFlat::search(n, x, k, distances, labels);  // the mirror's search ranks the centroids on the device itself
    expand_instances(n, k, distances, labels);
}

void IndexIVFFlatDedup::search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis,
                                           float* distances, idx_t* labels, bool store_pairs, const IVFSearchParameters* params) const {
    FAISS_THROW_IF_NOT_MSG(!store_pairs, "store_pairs not supported in IVFDedup");
    IndexIVFFlat::search_preassigned(n, x, k, assign, centroid_dis, distances, labels, false, params);
    expand_instances(n, k & 0xffffffff, distances, labels);
}

// Removal (IndexIVF.cpp:955-987): a removed entry's place is taken by the list's last entry, list by list; the lists in HBM are
// refreshed before the next search (the version counter of the inverted lists).
long IndexIVF::remove_ids(const IDSelector& sel) {
    FAISS_THROW_IF_NOT_MSG(!maintain_direct_map, "direct map remove not implemented");
    long nremove = 0;
    for (size_t l = 0; l < nlist; l++) {
        size_t len = invlists->list_size(l), j = 0;
        const size_t len0 = len;
        while (j < len) {
            if (sel.is_member(invlists->get_single_id(l, j))) {
                len--;
                invlists->update_entry(l, j, invlists->get_single_id(l, len), invlists->get_codes(l) + len * code_size);
            } else {
                j++;
            }
        }
        if (len != len0) invlists->resize(l, len);
        nremove += (long)(len0 - len);
    }
    ntotal -= nremove;
    return nremove;
}

// IndexIVFFlat.cpp:381-448: a removed id that stands for copies hands its entry to the first surviving copy (and that copy's
// remaining twins are re-keyed to it); only an entry without a surviving copy leaves its list
long IndexIVFFlatDedup::remove_ids(const IDSelector& sel) {
    std::unordered_map<idx_t, idx_t> heir;
    std::vector<std::pair<idx_t, idx_t>> rekeyed;
    for (auto it = instances.begin(); it != instances.end();) {
        const bool head_goes = sel.is_member(it->first), copy_goes = sel.is_member(it->second);
        if (head_goes && !copy_goes) {
            auto h = heir.find(it->first);
            if (h == heir.end()) heir[it->first] = it->second;
            else rekeyed.emplace_back(h->second, it->second);
        }
        if (head_goes || copy_goes) it = instances.erase(it);
        else ++it;
    }
    instances.insert(rekeyed.begin(), rekeyed.end());
    FAISS_THROW_IF_NOT_MSG(!maintain_direct_map, "direct map remove not implemented");
    long nremove = 0;
    for (size_t l = 0; l < nlist; l++) {
        size_t len = invlists->list_size(l), j = 0;
        const size_t len0 = len;
        while (j < len) {
            const idx_t id = invlists->get_single_id(l, j);
            if (!sel.is_member(id)) {
                j++;
                continue;
            }
            auto h = heir.find(id);
            if (h == heir.end()) {
                len--;
                invlists->update_entry(l, j, invlists->get_single_id(l, len), invlists->get_codes(l) + len * code_size);
            } else {
                invlists->update_entry(l, j, h->second, invlists->get_codes(l) + j * code_size);
                j++;
            }
        }
        if (len != len0) invlists->resize(l, len);
        nremove += (long)(len0 - len);
    }
    ntotal -= nremove;
    return nremove;
}

std::vector<int> ivf_list_owners(const IndexIVF& index, int nshard) {
    FAISS_THROW_IF_NOT(nshard > 0);
    std::vector<size_t> order(index.nlist);
    for (size_t l = 0; l < index.nlist; l++) order[l] = l;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return index.invlists->list_size(a) > index.invlists->list_size(b); });
    std::vector<size_t> load((size_t)nshard, 0);
    std::vector<int> owner(index.nlist, 0);
    for (size_t l : order) {
        const size_t r = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[l] = (int)r;
        load[r] += index.invlists->list_size(l);
    }
    return owner;
}

void IndexIVF::copy_subset_to(IndexIVF& other, int subset_type, idx_t a1, idx_t a2) const {
    FAISS_THROW_IF_NOT(nlist == other.nlist);
    FAISS_THROW_IF_NOT(code_size == other.code_size);
    FAISS_THROW_IF_NOT(!other.maintain_direct_map);
    FAISS_THROW_IF_NOT_FMT(subset_type >= 0 && subset_type <= 4, "subset type %d not implemented", subset_type);
    std::vector<int> owner;
    if (subset_type == 4) owner = ivf_list_owners(*this, (int)a1);
    size_t seen = 0, cut1 = 0, cut2 = 0;  // (type 2: entries met so far, and how many of them lie before a1 / before a2)
    for (size_t l = 0; l < nlist; l++) {
        const size_t n = invlists->list_size(l);
        const idx_t* ids = invlists->get_ids(l);
        const uint8_t* codes = invlists->get_codes(l);
        size_t i0 = 0, i1 = 0;  // a contiguous run of the list (types 2, 3, 4)
        if (subset_type == 0 || subset_type == 1) {
            for (size_t i = 0; i < n; i++) {
                const bool take = subset_type == 0 ? (a1 <= ids[i] && ids[i] < a2) : (ids[i] % a1 == a2);
                if (!take) continue;
                other.invlists->add_entry(l, ids[i], codes + i * code_size);
                other.ntotal++;
            }
        } else if (subset_type == 2) {
            const size_t next = seen + n, n1 = next * (size_t)a1 / (size_t)ntotal, n2 = next * (size_t)a2 / (size_t)ntotal;
            i0 = n1 - cut1;
            i1 = n2 - cut2;
            cut1 = n1;
            cut2 = n2;
        } else if (subset_type == 3 ? (idx_t)l % a1 == a2 : owner[l] == (int)a2) {
            i1 = n;
        }
        if (i1 > i0) {
            other.invlists->add_entries(l, i1 - i0, ids + i0, codes + i0 * code_size);
            other.ntotal += (idx_t)(i1 - i0);
        }
        seen += n;
    }
    FAISS_THROW_IF_NOT(seen == (size_t)ntotal);
}

IndexShardsByList::IndexShardsByList(const IndexIVFFlat& index, int nshard, bool threaded_, const int* devices)
    : IndexShards((idx_t)index.d, threaded_, false) {
    FAISS_THROW_IF_NOT(nshard > 0 && index.is_trained);
    for (int s = 0; s < nshard; s++) {
        IndexIVFFlat* sub = new IndexIVFFlat(index.quantizer, (size_t)index.d, index.nlist, index.metric_type);
        owned.push_back(sub);
        sub->is_trained = true;
        sub->nprobe = index.nprobe;
        sub->max_codes = index.max_codes;
        sub->coarse_mode = index.coarse_mode;
        index.copy_subset_to(*sub, 4, nshard, s);
        if (devices) sub->set_device(devices[s]);
        add_shard(sub);
    }
    FAISS_THROW_IF_NOT(ntotal == index.ntotal);
}

IndexShardsByList::~IndexShardsByList() {
    for (IndexIVFFlat* s : owned) delete s;
}

void IndexIVFFlatDedup::range_search(idx_t, const float*, float, RangeSearchResult*) const { FAISS_THROW_MSG("not implemented"); }
void IndexIVFFlatDedup::reconstruct_from_offset(idx_t, idx_t, float*) const { FAISS_THROW_MSG("not implemented"); }

// ------------------------------------------------------------------------------------- Error_sys
Error_sys::Error_sys(Index* in, size_t nq, size_t topk) : train_num(nq), max_topk(topk) {
    FAISS_THROW_IF_NOT_MSG(nq % 10 == 0, "Train num must be evenly divided by ten");
    key = "Base";
    if (IndexIVF* ix = dynamic_cast<IndexIVF*>(in)) {
        index = ix;
        key = "IVF";
        ix->t = nullptr;
    }
}

Error_sys::Error_sys() {}

void Error_sys::set_gt(const float* gt_D_in, const Index::idx_t* gt_I_in) {
    FAISS_THROW_IF_NOT_MSG((gt_D_in != nullptr && gt_I_in != nullptr), "the ground truth must not be null ptr when setting up");
    train_D.assign(gt_D_in, gt_D_in + train_num * max_topk);
    train_I.assign(gt_I_in, gt_I_in + train_num * max_topk);
}

float Error_sys::recall(Index::idx_t* I, Index::idx_t* gtI, size_t topk) {
    // sorts the caller's id row in place, like the reference (profile.cpp:246-280)
    std::sort(I, I + topk);
    size_t m = std::unique(I, I + topk) - I;
    size_t count = 0;
    std::vector<char> seen(m, 0);
    for (size_t i = 0; i < m; i++) {
        Index::idx_t* p = std::lower_bound(I, I + m, gtI[i]);
        if (p != I + m && *p == gtI[i] && !seen[p - I]) {
            seen[p - I] = 1;
            count++;
        }
    }
    return float(count) / m;
}

void Error_sys::set_train_point(float* D, Index::idx_t* I, size_t key_v, size_t nq) {
    FAISS_THROW_IF_NOT_MSG((index && index->t != nullptr), "your must init tune for index first");
    FAISS_THROW_IF_NOT_MSG((train_I.size() == train_num * max_topk), "ground truth not initialized");
    TrainPoint tp;
    tp.key = "nprobe";
    tp.key_value = key_v;
    tp.topk_dis.assign(D, D + nq * max_topk);
    tp.topk_id.assign(I, I + nq * max_topk);
    tp.acc.resize(nq);
    for (size_t i = 0; i < nq; i++) tp.acc[i] = recall(I + i * max_topk, train_I.data() + i * max_topk, max_topk);
    index->t->tps.push_back(tp);
}

void Error_sys::sys_train(size_t nq, const float* xq) {
    FAISS_THROW_IF_NOT_MSG(nq <= train_num, "Error sys training does not have the same nb of queries compared with creation");
    FAISS_THROW_IF_NOT_MSG((train_I.size() == train_num * max_topk), "ground truth not initialized");
    if (!index) return;
    IndexIVF* ix = index;
    ix->init_tune(nq, max_topk, xq, train_D.data(), train_I.data(), nullptr, nullptr);
    std::cout << "Init IVF done" << std::endl;
    ix->set_train_mode();
    ix->nprobe = ix->nlist;
    ix->set_resident_queries(xq, nq);
    std::vector<float> D(nq * max_topk);
    std::vector<Index::idx_t> I(nq * max_topk);
    const size_t bs = nq / 10;
    for (size_t q0 = 0; q0 < nq; q0 += bs) {
        size_t q1 = std::min(nq, q0 + bs);
        ix->search(q1 - q0, xq + q0 * ix->d, max_topk, D.data() + q0 * max_topk, I.data() + q0 * max_topk, q0);
    }
    set_train_point(D.data(), I.data(), ix->nlist, nq);
    ix->set_train_off();
    is_trained = true;
    std::cout << "Start t traing" << std::endl;
    ix->t->train(METRIC_L2);
    std::cout << "End t traing" << std::endl;
    for (size_t ij = 0; ij < std::min<size_t>(8, ix->t->traces.size()); ij++) {  // profile.cpp:158-169
        std::stringstream ss;
        ss << "Validation_" << ix->d << "_" << (1 << ij) << ".log";
        std::ofstream out(ss.str());
        for (auto& p : ix->t->traces[ij].trace) out << p.first << " " << p.second << std::endl;
    }
}

void Error_sys::set_queries(size_t n, const float* q, const float* acc, size_t allo_size) {
    num = n;
    queries = q;
    require_acc = acc;
    if (!index) return;
    error_pro* t = index->t;
    FAISS_THROW_IF_NOT_MSG((t != nullptr), "your must init tune for index first");
    t->alloc_s = allo_size;
    delete[] t->my_nprobe;
    t->my_nprobe = new size_t[allo_size]();
    size_t ind = 0;
    for (size_t np = 1; np <= index->nlist / 8; np <<= 1) ind++;
    delete[] t->KD;
    t->KD = new float[allo_size * ind]();
    delete[] t->t_recalls;
    t->t_recalls = new float[allo_size]();
    t->require_acc = acc;
    // every reference harness passes the whole query matrix (allo_size rows, the online ones last):
    // keep it resident in HBM so that search() only moves results
    index->set_resident_queries(q, allo_size);
}

void Error_sys::set_topk(size_t new_topk) {
    if (index) index->t->query_topk = new_topk;
}

void Error_sys::search(float* D, int64_t* I, size_t start, size_t search_size) {
    FAISS_THROW_IF_NOT_MSG(is_trained == true, "Error sys must be trained before searching");
    FAISS_THROW_IF_NOT_MSG(num <= train_num, "Error sys search num must be lower than all qeuries num");
    if (!index) return;
    index->set_tune_mode();
    index->nprobe = index->nlist;
    const size_t n = search_size == (size_t)-1 ? num : search_size;
    index->search(n, queries + start * index->d, max_topk, D, reinterpret_cast<Index::idx_t*>(I), start);
    index->set_tune_off();
}

void Error_sys::time_search(float* D, int64_t* I, size_t start, size_t search_size) {  // profile.cpp:229-244
    FAISS_THROW_IF_NOT_MSG(is_trained == true, "Error sys must be trained before searching");
    FAISS_THROW_IF_NOT_MSG(num <= train_num, "Error sys search num must be lower than all qeuries num");
    if (!index) return;
    FAISS_THROW_IF_NOT_MSG(index->t != nullptr, "time_search needs init_tune (it reads t->time_tune and the budgets in t->require_acc)");
    index->t->time_tune = true;
    index->nprobe = index->nlist;
    const size_t n = search_size == (size_t)-1 ? num : search_size;
    index->search(n, queries + start * index->d, max_topk, D, reinterpret_cast<Index::idx_t*>(I), start);
    index->t->time_tune = true;  // sic: the reference leaves it on
}

// ------------------------------------------------------------------------------------- IndexShards
IndexShards::IndexShards(idx_t d, bool threaded, bool successive_ids) : Index(d), threaded(threaded), successive_ids(successive_ids) {}

void IndexShards::add_shard(Index* index) {
    if (shards.empty()) {
        d = index->d;
        metric_type = index->metric_type;
    }
    FAISS_THROW_IF_NOT_MSG(index->d == d && index->metric_type == metric_type, "shards must agree on d and metric");
    // one MI355X per shard: a shard without a device of its own takes the next one of the node, round robin
    // (an index that already holds device state stays where it is)
    if (index->amd_device < 0 && !index->device_bound() && !getenv("AUNCEL_AMD_DEVICE")) {
        int ndev = 0;
        if (amd_ivf_device_count(&ndev) == 0 && ndev > 1) index->set_device((int)(shards.size() % (size_t)ndev));
    }
    shards.push_back(index);
    ntotal += index->ntotal;
    is_trained = index->is_trained;
}

// f(shard number, shard) on every shard: from one host thread per shard when `threaded` (the reference keeps a WorkerThread
// per shard, ThreadedIndex-inl.h:121-150; each thread then drives its own GPU through its shard's engine handle), one
// after the other otherwise.  The first exception is rethrown once every shard has returned.
template <class F> static void run_on_shards(const std::vector<Index*>& shards, bool threaded, F f) {
    const size_t ns = shards.size();
    if (!threaded || ns < 2) {
        for (size_t i = 0; i < ns; i++) f(i, shards[i]);
        return;
    }
    std::vector<std::exception_ptr> errs(ns);
    std::vector<std::thread> th;
    for (size_t i = 0; i < ns; i++)
        th.emplace_back([&, i] {
            try {
                f(i, shards[i]);
            } catch (...) {
                errs[i] = std::current_exception();
            }
        });
    for (auto& t : th) t.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
}

void IndexShards::train(idx_t n, const float* x) {
    run_on_shards(shards, threaded, [&](size_t, Index* s) { s->train(n, x); });
    is_trained = true;
}

void IndexShards::reset() {
    for (Index* s : shards) s->reset();
    ntotal = 0;
}

void IndexShards::add(idx_t n, const float* x) {
    // contiguous n/nshard slices, as the reference does without ids (tests/test_threaded_index.cpp:205-253)
    const idx_t ns = count();
    FAISS_THROW_IF_NOT(ns > 0);
    const idx_t base = ntotal;
    run_on_shards(shards, threaded, [&](size_t si, Index* sh) {
        const idx_t i = (idx_t)si, i0 = i * n / ns, i1 = (i + 1) * n / ns;
        if (successive_ids) {
            sh->add(i1 - i0, x + i0 * d);
        } else {
            std::vector<long> ids(i1 - i0);
            for (idx_t j = i0; j < i1; j++) ids[j - i0] = base + j;
            sh->add_with_ids(i1 - i0, x + i0 * d, ids.data());
        }
    });
    ntotal += n;
}

void IndexShards::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
    const size_t ns = shards.size();
    std::vector<float> all_D(ns * n * k);
    std::vector<int64_t> all_I(ns * n * k);
    // IndexShards.cpp:261-311: every shard searches the whole batch (its own lists, its own GPU), rows merged on the host
    run_on_shards(shards, threaded, [&](size_t s, Index* sh) {
        sh->search(n, x, k, all_D.data() + s * n * k, reinterpret_cast<idx_t*>(all_I.data()) + s * n * k);
    });
    if (successive_ids) {
        long base = 0;
        for (size_t s = 0; s < ns; s++) {
            if (s)
                for (size_t i = 0; i < (size_t)(n * k); i++)
                    if (all_I[s * n * k + i] >= 0) all_I[s * n * k + i] += base;
            base += shards[s]->ntotal;
        }
    }
    AMD(amd_ivf_merge_tables((int)metric_type, (size_t)n, (size_t)k, ns, all_D.data(), all_I.data(), distances, i64(labels)));
}

// ------------------------------------------------------------------------------------- factory
Index* index_factory(int d, const char* description, MetricType metric) {
    std::string s(description);
    if (s == "Flat") return new IndexFlat(d, metric);
    int nlist = 0;
    char tail[32] = {0};
    if (sscanf(description, "IVF%d,%31s", &nlist, tail) == 2 && std::string(tail) == "Flat" && nlist > 0) {
        IndexFlat* q = new IndexFlat(d, metric);
        IndexIVFFlat* ix = new IndexIVFFlat(q, d, nlist, metric);
        ix->own_fields = true;
        return ix;
    }
    FAISS_THROW_FMT("index_factory: only \"Flat\" and \"IVF<n>,Flat\" are built on this path, got \"%s\"", description);
}

}  // namespace faiss

// ------------------------------------------------------------------------------------- index_io
#include "index_io.h"

namespace faiss {

namespace {

uint32_t fourcc(const char* s) {
    const unsigned char* x = reinterpret_cast<const unsigned char*>(s);
    return x[0] | x[1] << 8 | x[2] << 16 | (uint32_t)x[3] << 24;
}

struct Writer {
    FILE* f;
    std::string name;
    template <class T> void put(const T& v) { raw(&v, sizeof(T)); }
    void raw(const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) FAISS_THROW_FMT("write error in %s", name.c_str());
    }
    template <class T> void vec(const std::vector<T>& v) {
        size_t n = v.size();
        put(n);
        raw(v.data(), n * sizeof(T));
    }
};

struct Reader {
    FILE* f;
    std::string name;
    template <class T> void get(T& v) { raw(&v, sizeof(T)); }
    void raw(void* p, size_t n) {
        if (n && fread(p, 1, n, f) != n) FAISS_THROW_FMT("read error in %s", name.c_str());
    }
    template <class T> void vec(std::vector<T>& v) {
        size_t n;
        get(n);
        FAISS_THROW_IF_NOT_MSG(n < (size_t(1) << 40), "implausible vector size in index file");
        v.resize(n);
        raw(v.data(), n * sizeof(T));
    }
};

// d (int), ntotal, two dummies, is_trained (bool), metric_type (int)   [index_io.cpp:205-213, 562-571]
void write_header(const Index* idx, Writer& w) {
    w.put(idx->d);
    w.put(idx->ntotal);
    Index::idx_t dummy = 1 << 20;
    w.put(dummy);
    w.put(dummy);
    w.put(idx->is_trained);
    w.put(idx->metric_type);
}

void read_header(Index* idx, Reader& r) {
    r.get(idx->d);
    r.get(idx->ntotal);
    Index::idx_t dummy;
    r.get(dummy);
    r.get(dummy);
    r.get(idx->is_trained);
    r.get(idx->metric_type);
    idx->verbose = false;
}

void write_any(const Index* idx, Writer& w);

void write_invlists(const InvertedLists* ils, Writer& w) {
    const ArrayInvertedLists* al = dynamic_cast<const ArrayInvertedLists*>(ils);
    if (!al) {
        w.put(fourcc("il00"));
        return;
    }
    w.put(fourcc("ilar"));
    w.put(al->nlist);
    w.put(al->code_size);
    size_t non0 = 0;
    for (size_t i = 0; i < al->nlist; i++) non0 += al->ids[i].empty() ? 0 : 1;
    std::vector<size_t> sizes;
    if (non0 > al->nlist / 2) {  // dense table of sizes, else (list, size) pairs  [index_io.cpp:296-320]
        w.put(fourcc("full"));
        for (size_t i = 0; i < al->nlist; i++) sizes.push_back(al->ids[i].size());
    } else {
        w.put(fourcc("sprs"));
        for (size_t i = 0; i < al->nlist; i++)
            if (!al->ids[i].empty()) {
                sizes.push_back(i);
                sizes.push_back(al->ids[i].size());
            }
    }
    w.vec(sizes);
    for (size_t i = 0; i < al->nlist; i++) {
        size_t n = al->ids[i].size();
        if (!n) continue;
        w.raw(al->codes[i].data(), n * al->code_size);
        w.raw(al->ids[i].data(), n * sizeof(Index::idx_t));
    }
}

void write_any(const Index* idx, Writer& w) {
    if (const IndexFlat* f = dynamic_cast<const IndexFlat*>(idx)) {
        w.put(fourcc(f->metric_type == METRIC_INNER_PRODUCT ? "IxFI" : "IxF2"));
        write_header(idx, w);
        w.vec(f->xb);
    } else if (const IndexIVFFlat* ivf = dynamic_cast<const IndexIVFFlat*>(idx)) {
        w.put(fourcc("IwFl"));
        write_header(ivf, w);
        w.put(ivf->nlist);
        w.put(ivf->nprobe);
        write_any(ivf->quantizer, w);
        w.put(ivf->maintain_direct_map);
        w.vec(ivf->direct_map);
        write_invlists(ivf->invlists, w);
    } else {
        FAISS_THROW_FMT("write_index: only IndexFlat and IndexIVFFlat are on this path (got %s)", idx ? typeid(*idx).name() : "null");
    }
}

Index* read_any(Reader& r) {
    uint32_t h;
    r.get(h);
    if (h == fourcc("IxFI") || h == fourcc("IxF2")) {
        IndexFlat* f = h == fourcc("IxFI") ? static_cast<IndexFlat*>(new IndexFlatIP()) : new IndexFlatL2();
        read_header(f, r);
        r.vec(f->xb);
        FAISS_THROW_IF_NOT(f->xb.size() == (size_t)f->ntotal * f->d);
        return f;
    }
    if (h == fourcc("IwFl")) {
        IndexIVFFlat* ivf = new IndexIVFFlat();
        read_header(ivf, r);
        r.get(ivf->nlist);
        r.get(ivf->nprobe);
        ivf->quantizer = read_any(r);
        ivf->own_fields = true;
        r.get(ivf->maintain_direct_map);
        r.vec(ivf->direct_map);
        ivf->code_size = ivf->d * sizeof(float);
        uint32_t hl;
        r.get(hl);
        if (hl == fourcc("il00")) {
            ivf->invlists = new ArrayInvertedLists(ivf->nlist, ivf->code_size);
        } else {
            FAISS_THROW_IF_NOT_MSG(hl == fourcc("ilar"), "unsupported inverted list type");
            size_t nlist, code_size;
            r.get(nlist);
            r.get(code_size);
            FAISS_THROW_IF_NOT(nlist == ivf->nlist && code_size == ivf->code_size);
            ArrayInvertedLists* al = new ArrayInvertedLists(nlist, code_size);
            uint32_t lt;
            r.get(lt);
            std::vector<size_t> tab, sizes(nlist, 0);
            r.vec(tab);
            if (lt == fourcc("full")) {
                FAISS_THROW_IF_NOT(tab.size() == nlist);
                sizes = tab;
            } else if (lt == fourcc("sprs")) {
                for (size_t j = 0; j + 1 < tab.size(); j += 2) sizes[tab[j]] = tab[j + 1];
            } else {
                FAISS_THROW_MSG("invalid list_type");
            }
            for (size_t i = 0; i < nlist; i++) {
                al->ids[i].resize(sizes[i]);
                al->codes[i].resize(sizes[i] * code_size);
                if (sizes[i]) {
                    r.raw(al->codes[i].data(), sizes[i] * code_size);
                    r.raw(al->ids[i].data(), sizes[i] * sizeof(Index::idx_t));
                }
            }
            al->version = 1;
            ivf->invlists = al;
        }
        ivf->own_invlists = true;
        return ivf;
    }
    FAISS_THROW_MSG("read_index: only IxF2 / IxFI / IwFl files are on this path");
}

}  // namespace

void write_index(const Index* idx, const char* fname) {
    Writer w{fopen(fname, "wb"), fname};
    FAISS_THROW_IF_NOT_FMT(w.f, "could not open %s for writing", fname);
    try {
        write_any(idx, w);
    } catch (...) {
        fclose(w.f);
        throw;
    }
    fclose(w.f);
}

Index* read_index(const char* fname, int) {
    Reader r{fopen(fname, "rb"), fname};
    FAISS_THROW_IF_NOT_FMT(r.f, "could not open %s for reading", fname);
    Index* idx = nullptr;
    try {
        idx = read_any(r);
    } catch (...) {
        fclose(r.f);
        throw;
    }
    fclose(r.f);
    return idx;
}

}  // namespace faiss
