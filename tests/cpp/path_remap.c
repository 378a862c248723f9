/* TEST INFRASTRUCTURE.  LD_PRELOAD shim for the reference's own harnesses (Auncel/eval/{bound,effect_error,overhead,effect_time}.cpp),
 * whose dataset paths are compiled in ("/workspace/data/sift/sift10M/query.fvecs" ...): a path under /workspace/data is opened
 * under $AUNCEL_DATA_ROOT instead, so the unmodified binaries run on small synthetic files in a test directory.
 *   gcc -shared -fPIC -O1 -o path_remap.so path_remap.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char* remap(const char* p, char* buf, size_t n) {
    static const char pre[] = "/workspace/data";
    const char* root = getenv("AUNCEL_DATA_ROOT");
    if (root && p && !strncmp(p, pre, sizeof pre - 1)) {
        snprintf(buf, n, "%s%s", root, p + sizeof pre - 1);
        return buf;
    }
    return p;
}

FILE* fopen(const char* path, const char* mode) {
    static FILE* (*real)(const char*, const char*);
    char buf[4096];
    if (!real) real = (FILE * (*)(const char*, const char*)) dlsym(RTLD_NEXT, "fopen");
    return real(remap(path, buf, sizeof buf), mode);
}

FILE* fopen64(const char* path, const char* mode) {
    static FILE* (*real)(const char*, const char*);
    char buf[4096];
    if (!real) real = (FILE * (*)(const char*, const char*)) dlsym(RTLD_NEXT, "fopen64");
    return real(remap(path, buf, sizeof buf), mode);
}
