// The evaluation harness's file readers under their own names and signatures (Auncel/eval/bound.cpp:29-113), on top of
// the C ABI (include/auncel_amd.h: amd_ivf_read_*).  Buffers come from new[] as in the harness (callers delete[] them);
// a malformed file throws FaissException where the harness aborts.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../../include/auncel_amd.h"
#include "FaissAssert.h"

namespace faiss {

namespace dataset_io_detail {
template <class T> T* to_new(T* m, size_t count) {
    T* x = new T[count ? count : 1];
    if (count) memcpy(x, m, count * sizeof(T));
    amd_ivf_free(m);
    return x;
}
}  // namespace dataset_io_detail

inline float* fvecs_read(const char* fname, size_t* d_out, size_t* n_out) {
    float* m = nullptr;
    FAISS_THROW_IF_NOT_MSG(amd_ivf_read_fvecs(fname, d_out, n_out, &m) == 0, amd_ivf_last_error());
    return dataset_io_detail::to_new(m, *d_out * *n_out);
}

inline int* ivecs_read(const char* fname, size_t* d_out, size_t* n_out) {
    int32_t* m = nullptr;
    FAISS_THROW_IF_NOT_MSG(amd_ivf_read_ivecs(fname, d_out, n_out, &m) == 0, amd_ivf_last_error());
    return reinterpret_cast<int*>(dataset_io_detail::to_new(m, *d_out * *n_out));
}

inline float* fbin_read(const char* fname, size_t* d_out, size_t* n_out, int num = 10000000, int bytes = 4) {
    float* m = nullptr;
    FAISS_THROW_IF_NOT_MSG(amd_ivf_read_fbin(fname, (size_t)num, bytes, d_out, n_out, &m) == 0, amd_ivf_last_error());
    return dataset_io_detail::to_new(m, *d_out * (size_t)num);
}

inline int* ibin_read(const char* fname, size_t* d_out, size_t* n_out, int num = 10000000, int bytes = 4) {
    (void)bytes;  // (the harness forwards it to fbin_read; ids are 4 bytes)
    int32_t* m = nullptr;
    FAISS_THROW_IF_NOT_MSG(amd_ivf_read_ibin(fname, (size_t)num, d_out, n_out, &m) == 0, amd_ivf_last_error());
    return reinterpret_cast<int*>(dataset_io_detail::to_new(m, *d_out * (size_t)num));
}

}  // namespace faiss
