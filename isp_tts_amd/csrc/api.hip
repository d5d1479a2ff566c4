// ABI bookkeeping: version, thread-local error string, device query.
#include <atomic>

#include "common.h"

char* ispk_err_buf() {
    static thread_local char buf[256] = {0};
    return buf;
}

extern "C" int32_t ispk_abi_version(void) { return ISPK_ABI_VERSION; }

extern "C" const char* ispk_last_error_string(void) { return ispk_err_buf(); }

extern "C" int32_t ispk_device_info(char* name, int32_t cap) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) ISPK_FAIL((int32_t)e, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) ISPK_FAIL((int32_t)e, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && cap > 0) snprintf(name, (size_t)cap, "%s", p.gcnArchName);
    return p.multiProcessorCount;
}

// Dropout seed source (dropout.h).  PROCESS-wide, not thread-local: autograd runs the backward nodes of a step on its own
// per-device worker thread, and the backward kernels must fold in the same word as the forward kernels the calling thread
// launched (one process drives one GPU, so one source per process is the right scope).
static std::atomic<const uint64_t*> g_seed_source{nullptr};
const uint64_t* ispk_seed_source() { return g_seed_source.load(std::memory_order_acquire); }
extern "C" int32_t ispk_set_dropout_seed_source(const uint64_t* device_word) {
    g_seed_source.store(device_word, std::memory_order_release);
    return 0;
}
