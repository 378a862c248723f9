// Auncel's error-profile state (Auncel/IVF_pro.h:47-180): same public fields and method names as the
// reference, because the harnesses poke them directly (ix->t->setparam, ->profile, ->t_recalls, ...).
// The arithmetic itself (set_online, sum_angle, cur_num, the stop rule) runs inside the GPU replay
// kernel; what stays on the host is bookkeeping, Trace::SB and the acos table (host libm).
#pragma once
#include <string>
#include <sys/time.h>
#include <utility>
#include <vector>

#include "Index.h"

namespace faiss {

using idx = Index::idx_t;

struct Trace {
    size_t nprobe = 0;
    std::vector<std::pair<float, float>> trace;  ///< (sum of angles, k-scaling); raw samples before SB()
    std::vector<float> stds;
    size_t bs = 250;
    /// piecewise-constant lookup (IVF_pro.cpp:84-107); host copy for inspection, the search uses the device one
    float search(float k, float std_m);
    /// sort + bucket (IVF_pro.cpp:109-149)
    void SB();
};
using Traces = std::vector<Trace>;

struct TrainPoint {
    std::vector<float> acc;
    std::string key;
    size_t key_value = 0;
    std::vector<float> topk_dis;
    std::vector<idx> topk_id;
};

class error_pro {
   public:
    float std_m = 1.0;
    float multipler = 1.0;
    size_t arcos_size = 500;
    std::vector<float> arcos_list;

    const float* require_acc = nullptr;
    bool profile = false;
    bool overhead_profile = false;
    bool time_tune = false;
    size_t alloc_s = 0;
    float* KD = nullptr;
    float* t_recalls = nullptr;
    float cur_rc = 0;
    size_t* my_nprobe = nullptr;
    size_t id = 0;
    size_t count = 0;
    size_t query_topk = -1;

    std::vector<Trace> traces;
    enum metric { L2, IP };
    metric m_type = L2;
    size_t nlist = 0;
    size_t max_topk = 0;
    size_t d = 0;
    size_t train_num = 0;
    float* interdis_cem = nullptr;
    const float* train_q = nullptr;
    const float* train_D = nullptr;
    const idx* train_I = nullptr;
    float* train_cd = nullptr;
    idx* train_ci = nullptr;
    std::vector<TrainPoint> tps;

    /// bumped whenever traces change so that the owning index re-uploads them
    size_t traces_version = 0;

    float arcos(float x);
    void construct_arcos();
    void train(MetricType metric_type);
    void setparam(int id);
    ~error_pro();
};

}  // namespace faiss
