// TEST INFRASTRUCTURE ONLY (oracle/): tiny named-tensor container used to move
// inputs/outputs between the Python test scripts and the C++ oracle programs.
//
// File layout (little endian):
//   char magic[8] = "TBND1\0\0\0"; u32 count;
//   per record: u32 name_len; char name[name_len]; u32 dtype; u32 ndim;
//               u64 dims[ndim]; raw data (row-major).
// dtype: 0=f32 1=i64 2=i32 3=u8 4=f64 5=u64
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace tb {

enum DType : uint32_t { F32 = 0, I64 = 1, I32 = 2, U8 = 3, F64 = 4, U64 = 5 };

inline size_t dtype_size(uint32_t t) {
    switch (t) {
        case F32: case I32: return 4;
        case I64: case F64: case U64: return 8;
        case U8: return 1;
    }
    throw std::runtime_error("tbundle: bad dtype");
}

struct Tensor {
    uint32_t dtype = F32;
    std::vector<uint64_t> dims;
    std::vector<uint8_t> data;
    size_t numel() const {
        size_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
    template <class T> const T* as() const { return reinterpret_cast<const T*>(data.data()); }
    template <class T> T* as() { return reinterpret_cast<T*>(data.data()); }
};

struct Bundle {
    std::map<std::string, Tensor> t;
    std::vector<std::string> order;

    bool has(const std::string& n) const { return t.count(n) != 0; }

    const Tensor& get(const std::string& n) const {
        auto it = t.find(n);
        if (it == t.end()) throw std::runtime_error("tbundle: missing tensor " + n);
        return it->second;
    }

    template <class T> T scalar(const std::string& n) const {
        const Tensor& x = get(n);
        if (x.numel() != 1) throw std::runtime_error("tbundle: not a scalar: " + n);
        switch (x.dtype) {
            case F32: return (T)x.as<float>()[0];
            case I64: return (T)x.as<int64_t>()[0];
            case I32: return (T)x.as<int32_t>()[0];
            case U8:  return (T)x.as<uint8_t>()[0];
            case F64: return (T)x.as<double>()[0];
            case U64: return (T)x.as<uint64_t>()[0];
        }
        throw std::runtime_error("tbundle: bad dtype");
    }

    template <class T> T scalar_or(const std::string& n, T dflt) const {
        return has(n) ? scalar<T>(n) : dflt;
    }

    void put(const std::string& n, uint32_t dtype, std::vector<uint64_t> dims, const void* p) {
        Tensor x;
        x.dtype = dtype;
        x.dims = std::move(dims);
        size_t bytes = x.numel() * dtype_size(dtype);
        x.data.resize(bytes);
        if (bytes) memcpy(x.data.data(), p, bytes);
        if (!t.count(n)) order.push_back(n);
        t[n] = std::move(x);
    }
    void put_f32(const std::string& n, std::vector<uint64_t> dims, const float* p) { put(n, F32, std::move(dims), p); }
    void put_i64(const std::string& n, std::vector<uint64_t> dims, const int64_t* p) { put(n, I64, std::move(dims), p); }
    void put_u64(const std::string& n, std::vector<uint64_t> dims, const uint64_t* p) { put(n, U64, std::move(dims), p); }
    void put_scalar_i64(const std::string& n, int64_t v) { put(n, I64, {1}, &v); }
    void put_scalar_f64(const std::string& n, double v) { put(n, F64, {1}, &v); }

    static Bundle load(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("tbundle: cannot open " + path);
        auto rd = [&](void* p, size_t n) {
            if (n && fread(p, 1, n, f) != n) { fclose(f); throw std::runtime_error("tbundle: short read " + path); }
        };
        char magic[8];
        rd(magic, 8);
        if (memcmp(magic, "TBND1\0\0\0", 8) != 0) { fclose(f); throw std::runtime_error("tbundle: bad magic"); }
        uint32_t count;
        rd(&count, 4);
        Bundle b;
        for (uint32_t i = 0; i < count; i++) {
            uint32_t nl;
            rd(&nl, 4);
            std::string name(nl, '\0');
            rd(&name[0], nl);
            Tensor x;
            uint32_t ndim;
            rd(&x.dtype, 4);
            rd(&ndim, 4);
            x.dims.resize(ndim);
            rd(x.dims.data(), 8 * ndim);
            x.data.resize(x.numel() * dtype_size(x.dtype));
            rd(x.data.data(), x.data.size());
            b.order.push_back(name);
            b.t[name] = std::move(x);
        }
        fclose(f);
        return b;
    }

    void save(const std::string& path) const {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("tbundle: cannot write " + path);
        fwrite("TBND1\0\0\0", 1, 8, f);
        uint32_t count = (uint32_t)order.size();
        fwrite(&count, 4, 1, f);
        for (auto& name : order) {
            const Tensor& x = t.at(name);
            uint32_t nl = (uint32_t)name.size(), ndim = (uint32_t)x.dims.size();
            fwrite(&nl, 4, 1, f);
            fwrite(name.data(), 1, nl, f);
            fwrite(&x.dtype, 4, 1, f);
            fwrite(&ndim, 4, 1, f);
            fwrite(x.dims.data(), 8, ndim, f);
            if (!x.data.empty()) fwrite(x.data.data(), 1, x.data.size(), f);
        }
        fclose(f);
    }
};

}  // namespace tb
