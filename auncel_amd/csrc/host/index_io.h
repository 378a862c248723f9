// write_index / read_index for the index types of the hot path, in the reference's binary format
// (Auncel/index_io.cpp): "IxF2"/"IxFI" flat quantiser, "IwFl" IVF-Flat with "ilar" array inverted lists.
// Files written by the reference load here and vice versa.  As in the reference, Auncel's tuner state
// (interdis_cem, traces) is not part of the file.
#pragma once
#include "Index.h"

namespace faiss {

void write_index(const Index* idx, const char* fname);
Index* read_index(const char* fname, int io_flags = 0);

}  // namespace faiss
