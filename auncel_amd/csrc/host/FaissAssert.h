// FAISS_THROW_* conventions of the reference (Auncel/FaissAssert.h:60-93), same macro names.
#pragma once
#include <cstdio>
#include <string>

#include "FaissException.h"

#define FAISS_THROW_MSG(MSG) throw faiss::FaissException(MSG, __PRETTY_FUNCTION__, __FILE__, __LINE__)

#define FAISS_THROW_FMT(FMT, ...)                                   \
    do {                                                            \
        char _buf[512];                                             \
        snprintf(_buf, sizeof(_buf), FMT, __VA_ARGS__);             \
        FAISS_THROW_MSG(std::string(_buf));                         \
    } while (0)

#define FAISS_THROW_IF_NOT(X)                                       \
    do {                                                            \
        if (!(X)) FAISS_THROW_MSG("Error: '" #X "' failed");        \
    } while (0)

#define FAISS_THROW_IF_NOT_MSG(X, MSG)                              \
    do {                                                            \
        if (!(X)) FAISS_THROW_MSG(std::string("Error: '" #X "' failed: ") + MSG); \
    } while (0)

#define FAISS_THROW_IF_NOT_FMT(X, FMT, ...)                         \
    do {                                                            \
        if (!(X)) FAISS_THROW_FMT("Error: '" #X "' failed: " FMT, __VA_ARGS__); \
    } while (0)

#define FAISS_ASSERT(X) FAISS_THROW_IF_NOT(X)
