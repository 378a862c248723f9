// Dataset files of the reference's evaluation harness (host only, no device code):
//   .fvecs / .ivecs  every row = int32 d, then d 4-byte values            (Auncel/eval/bound.cpp:29-63)
//   .fbin / .ibin    int32 n, int32 d, then rows of d values, 4 bytes each, or 1 byte each read as *signed* chars and
//                    widened -- the harness's `bytes = 1` mode, kept as it is  (Auncel/eval/bound.cpp:65-113)
// The harness aborts on a malformed file; here that is return code -2 with a message (amd_ivf_last_error).
#include "../../include/auncel_amd.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>

namespace amdivf {
void set_last_error(const std::string& msg);  // ivf_engine.hip
}

namespace {

struct File {
    FILE* f;
    explicit File(const char* path) : f(fopen(path, "rb")) {}
    ~File() {
        if (f) fclose(f);
    }
};

int fail(const std::string& msg) {
    amdivf::set_last_error(msg);
    return -2;
}

// rows of (int32 d | d x 4 bytes) -> compact n x d matrix of 4-byte values
int read_vecs(const char* path, size_t* d_out, size_t* n_out, void** out) {
    if (!path || !d_out || !n_out || !out) return fail("null argument");
    File fl(path);
    if (!fl.f) return fail(std::string("could not open ") + path);
    int32_t d = 0;
    if (fread(&d, 4, 1, fl.f) != 1) return fail(std::string("empty file ") + path);
    if (!(d > 0 && d < 1000000)) return fail("unreasonable dimension");
    struct stat st;
    if (fstat(fileno(fl.f), &st) != 0) return fail("fstat failed");
    const size_t sz = (size_t)st.st_size, row = ((size_t)d + 1) * 4;
    if (sz % row != 0) return fail("weird file size");
    const size_t n = sz / row;
    uint32_t* x = static_cast<uint32_t*>(malloc(n * (size_t)d * 4 + 4));
    if (!x) return fail("out of memory");
    fseek(fl.f, 0, SEEK_SET);
    // a slab of rows at a time, headers dropped on the way (the harness reads everything and shifts in place)
    const size_t slab = (((size_t)4 << 20) / row) + 1;
    uint32_t* buf = static_cast<uint32_t*>(malloc(slab * row));
    if (!buf) {
        free(x);
        return fail("out of memory");
    }
    for (size_t i0 = 0; i0 < n; i0 += slab) {
        const size_t m = n - i0 < slab ? n - i0 : slab;
        if (fread(buf, row, m, fl.f) != m) {
            free(buf);
            free(x);
            return fail("could not read whole file");
        }
        for (size_t i = 0; i < m; i++) {
            if ((int32_t)buf[i * ((size_t)d + 1)] != d) {
                free(buf);
                free(x);
                return fail("row header differs from the first row's dimension");
            }
            memcpy(x + (i0 + i) * (size_t)d, buf + i * ((size_t)d + 1) + 1, (size_t)d * 4);
        }
    }
    free(buf);
    *d_out = (size_t)d;
    *n_out = n;
    *out = x;
    return 0;
}

}  // namespace

extern "C" {

int amd_ivf_read_fvecs(const char* path, size_t* d, size_t* n, float** x) { return read_vecs(path, d, n, reinterpret_cast<void**>(x)); }

int amd_ivf_read_ivecs(const char* path, size_t* d, size_t* n, int32_t** x) { return read_vecs(path, d, n, reinterpret_cast<void**>(x)); }

int amd_ivf_read_fbin(const char* path, size_t num, int bytes, size_t* d_out, size_t* n_out, float** out) {
    if (!path || !d_out || !n_out || !out) return fail("null argument");
    if (bytes != 1 && bytes != 4) return fail("bytes per value must be 1 or 4");
    File fl(path);
    if (!fl.f) return fail(std::string("could not open ") + path);
    int32_t hdr[2];
    if (fread(hdr, 4, 2, fl.f) != 2) return fail(std::string("no header in ") + path);
    const int32_t n = hdr[0], d = hdr[1];
    if (!(d > 0 && d < 1000000)) return fail("unreasonable dimension");
    if (n < 0) return fail("negative row count");
    const size_t rows = num ? num : (size_t)n;  // the harness passes the number of rows it wants; 0: the header's
    const size_t total = rows * (size_t)d;
    float* x = static_cast<float*>(malloc(total * 4 + 4));
    if (!x) return fail("out of memory");
    if (bytes == 4) {
        if (fread(x, 4, total, fl.f) != total) {
            free(x);
            return fail("could not read whole file");
        }
    } else {
        int8_t* raw = static_cast<int8_t*>(malloc(total + 1));
        if (!raw) {
            free(x);
            return fail("out of memory");
        }
        if (fread(raw, 1, total, fl.f) != total) {
            free(raw);
            free(x);
            return fail("could not read whole file");
        }
        for (size_t i = 0; i < total; i++) x[i] = (float)raw[i];
        free(raw);
    }
    *d_out = (size_t)d;
    *n_out = (size_t)n;  // as the harness: the header's count, whatever `num` asked for
    *out = x;
    return 0;
}

int amd_ivf_read_ibin(const char* path, size_t num, size_t* d, size_t* n, int32_t** x) {
    return amd_ivf_read_fbin(path, num, 4, d, n, reinterpret_cast<float**>(x));
}

void amd_ivf_free(void* p) { free(p); }

}  // extern "C"
