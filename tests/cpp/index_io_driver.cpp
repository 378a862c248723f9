// index_io of the host mirror against files written by the reference's write_index (CPU only).
// usage: index_io_driver <bundle.tb> <scratch dir>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../auncel_amd/csrc/host/FaissException.h"
#include "../../auncel_amd/csrc/host/IndexFlat.h"
#include "../../auncel_amd/csrc/host/IndexIVFFlat.h"
#include "../../auncel_amd/csrc/host/index_io.h"
#include "../../oracle/tbundle.h"

using namespace faiss;
static int fails = 0;
static void expect(bool ok, const char* what) {
    if (!ok) {
        printf("MISMATCH: %s\n", what);
        fails++;
    }
}
static std::vector<uint8_t> slurp(const std::string& fn) {
    std::vector<uint8_t> b;
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return b;
    int c;
    while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c);
    fclose(f);
    return b;
}
static void spit(const std::string& fn, const tb::Tensor& t) {
    FILE* f = fopen(fn.c_str(), "wb");
    fwrite(t.data.data(), 1, t.data.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    try {
        tb::Bundle in = tb::Bundle::load(argv[1]);
        std::string dir = argv[2];
        size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist");
        MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
        const tb::Tensor &cen = in.get("centroids"), &xb = in.get("xb");
        size_t nb = xb.dims[0];
        // (1) our writer reproduces the reference's bytes
        IndexFlat quantizer(d, mt);
        quantizer.add(nlist, cen.as<float>());
        IndexIVFFlat index(&quantizer, d, nlist, mt);
        index.nprobe = in.scalar<size_t>("nprobe");
        write_index(&index, (dir + "/e.index").c_str());
        std::vector<uint8_t> b = slurp(dir + "/e.index");
        expect(b.size() == in.get("index_empty").data.size() && memcmp(b.data(), in.get("index_empty").data.data(), b.size()) == 0,
               "empty index bytes");
        index.add_core(nb, xb.as<float>(), nullptr, reinterpret_cast<const long*>(in.get("assign").as<int64_t>()));
        write_index(&index, (dir + "/f.index").c_str());
        b = slurp(dir + "/f.index");
        expect(b.size() == in.get("index_full").data.size() && memcmp(b.data(), in.get("index_full").data.data(), b.size()) == 0,
               "populated index bytes");
        // (2) files written by the reference load, and round-trip byte for byte
        for (const char* name : {"index_empty", "index_full"}) {
            spit(dir + "/ref.index", in.get(name));
            std::unique_ptr<Index> r(read_index((dir + "/ref.index").c_str()));
            IndexIVFFlat* ivf = dynamic_cast<IndexIVFFlat*>(r.get());
            expect(ivf && ivf->nlist == nlist && (size_t)ivf->d == d && ivf->metric_type == mt && ivf->type == IVF, "header fields");
            if (!ivf) continue;
            expect(dynamic_cast<IndexFlat*>(ivf->quantizer) && (size_t)ivf->quantizer->ntotal == nlist, "quantizer");
            write_index(r.get(), (dir + "/rt.index").c_str());
            b = slurp(dir + "/rt.index");
            expect(b.size() == in.get(name).data.size() && memcmp(b.data(), in.get(name).data.data(), b.size()) == 0, "round trip bytes");
            if (std::string(name) == "index_full") {
                expect((size_t)ivf->ntotal == nb, "ntotal");
                size_t tot = 0;
                for (size_t l = 0; l < nlist; l++) tot += ivf->invlists->list_size(l);
                expect(tot == nb, "list sizes");
            }
        }
        // (3) a truncated file is an error, not a crash
        {
            FILE* f = fopen((dir + "/trunc.index").c_str(), "wb");
            fwrite(in.get("index_full").data.data(), 1, in.get("index_full").data.size() / 2, f);
            fclose(f);
            bool threw = false;
            try { std::unique_ptr<Index> r(read_index((dir + "/trunc.index").c_str())); } catch (const FaissException&) { threw = true; }
            expect(threw, "truncated file throws");
        }
        printf(fails ? "FAILED %d\n" : "ALL OK\n", fails);
        return fails ? 1 : 0;
    } catch (const std::exception& e) {
        printf("EXCEPTION: %s\n", e.what());
        return 3;
    }
}
