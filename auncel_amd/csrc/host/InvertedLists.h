// faiss::InvertedLists / ArrayInvertedLists (Auncel/InvertedLists.h:31-202): the host-visible store
// of the lists.  The engine keeps its own CSR-packed copy in HBM, refreshed lazily after writes.
#pragma once
#include <cstdint>
#include <vector>

#include "Index.h"

namespace faiss {

struct InvertedLists {
    typedef Index::idx_t idx_t;
    size_t nlist;
    size_t code_size;

    InvertedLists(size_t nlist, size_t code_size) : nlist(nlist), code_size(code_size) {}
    virtual ~InvertedLists() {}

    virtual size_t list_size(size_t list_no) const = 0;
    virtual const uint8_t* get_codes(size_t list_no) const = 0;
    virtual const idx_t* get_ids(size_t list_no) const = 0;
    virtual void release_codes(const uint8_t*) const {}
    virtual void release_ids(const idx_t*) const {}
    virtual size_t add_entries(size_t list_no, size_t n_entry, const idx_t* ids, const uint8_t* code) = 0;
    virtual size_t add_entry(size_t list_no, idx_t theid, const uint8_t* code) { return add_entries(list_no, 1, &theid, code); }
    virtual void resize(size_t list_no, size_t new_size) = 0;
    /// one stored id / entries overwritten in place (InvertedLists.h:66,88-93)
    virtual idx_t get_single_id(size_t list_no, size_t offset) const { return get_ids(list_no)[offset]; }
    virtual void update_entries(size_t list_no, size_t offset, size_t n_entry, const idx_t* ids, const uint8_t* code) = 0;
    virtual void update_entry(size_t list_no, size_t offset, idx_t id, const uint8_t* code) { update_entries(list_no, offset, 1, &id, code); }
    virtual void reset() {
        for (size_t i = 0; i < nlist; i++) resize(i, 0);
    }
    virtual void prefetch_lists(const long*, int) const {}

    /// bumped by every write: lets the owning index know its HBM copy is stale
    size_t version = 0;

    struct ScopedIds {
        const InvertedLists* il;
        const idx_t* ids;
        ScopedIds(const InvertedLists* il, size_t list_no) : il(il), ids(il->get_ids(list_no)) {}
        const idx_t* get() { return ids; }
        idx_t operator[](size_t i) const { return ids[i]; }
        ~ScopedIds() { il->release_ids(ids); }
    };
    struct ScopedCodes {
        const InvertedLists* il;
        const uint8_t* codes;
        ScopedCodes(const InvertedLists* il, size_t list_no) : il(il), codes(il->get_codes(list_no)) {}
        const uint8_t* get() { return codes; }
        ~ScopedCodes() { il->release_codes(codes); }
    };
};

struct ArrayInvertedLists : InvertedLists {
    std::vector<std::vector<uint8_t>> codes;  // nlist x (n_l * code_size)
    std::vector<std::vector<idx_t>> ids;

    ArrayInvertedLists(size_t nlist, size_t code_size) : InvertedLists(nlist, code_size), codes(nlist), ids(nlist) {}

    size_t list_size(size_t list_no) const override { return ids[list_no].size(); }
    const uint8_t* get_codes(size_t list_no) const override { return codes[list_no].data(); }
    const idx_t* get_ids(size_t list_no) const override { return ids[list_no].data(); }
    size_t add_entries(size_t list_no, size_t n_entry, const idx_t* ids_in, const uint8_t* code) override;
    void resize(size_t list_no, size_t new_size) override;
    void update_entries(size_t list_no, size_t offset, size_t n_entry, const idx_t* ids_in, const uint8_t* code) override;
};

}  // namespace faiss
