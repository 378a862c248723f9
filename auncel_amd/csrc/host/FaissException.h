// Host-side mirror of the reference's exception type (Auncel/FaissException.h): thrown wherever
// the reference would throw, including engine errors reported through the C ABI (code -2).
#pragma once
#include <exception>
#include <string>

namespace faiss {

class FaissException : public std::exception {
   public:
    explicit FaissException(const std::string& m) : msg(m) {}
    FaissException(const std::string& m, const char* func, const char* file, int line)
        : msg(std::string("Error in ") + func + " at " + file + ":" + std::to_string(line) + ": " + m) {}
    const char* what() const noexcept override { return msg.c_str(); }
    std::string msg;
};

}  // namespace faiss
