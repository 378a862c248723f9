// faiss::Error_sys (Auncel/profile.h:29-91): offline trace training + online error-bounded search driver.
#pragma once
#include <string>
// (as Auncel/profile.h:18-25)
#include <condition_variable>
#include <stdint.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/types.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "IndexIVF.h"

namespace faiss {

class Error_sys {
   public:
    const float* queries = nullptr;
    size_t num = 0;
    const float* require_acc = nullptr;
    bool is_trained = false;
    std::string key;
    size_t train_num = 0;
    size_t max_topk = 0;
    IndexIVF* index = nullptr;
    std::vector<float> train_D;
    std::vector<Index::idx_t> train_I;

    Error_sys(Index* in, size_t nq, size_t topk);
    Error_sys();

    void set_gt(const float* gt_D_in, const Index::idx_t* gt_I_in);
    void set_train_point(float* D, Index::idx_t* I, size_t key_v, size_t nq);
    void sys_train(size_t nq, const float* xq);
    void set_queries(size_t n, const float* q, const float* acc, size_t allo_size);
    void search(float* D, int64_t* I, size_t start, size_t search_size = -1);
    void time_search(float* D, int64_t* I, size_t start, size_t search_size = -1);
    void set_topk(size_t new_topk);
    float recall(Index::idx_t* I, Index::idx_t* gtI, size_t topk);
};

}  // namespace faiss
