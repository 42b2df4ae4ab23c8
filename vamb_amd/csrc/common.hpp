// Shared host-side plumbing for libvambhip.so: error reporting across the C ABI, RAII device
// buffers, stream + event timing.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vambhip.h"
#include "../../include/vambhip_debug.h"

namespace vh {

extern thread_local std::string g_last_error;

inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

struct HipError {
    hipError_t err;
    const char* what;
    const char* file;
    int line;
};

#define VH_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) throw ::vh::HipError{_e, #expr, __FILE__, __LINE__}; \
    } while (0)

struct InvalidArg {
    std::string msg;
};
#define VH_REQUIRE(cond, ...)                                        \
    do {                                                             \
        if (!(cond)) {                                               \
            char _b[512];                                            \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
            throw ::vh::InvalidArg{std::string(_b)};                 \
        }                                                            \
    } while (0)

// Run `body` and translate C++ exceptions into status codes (nothing crosses the C boundary).
template <class F>
int guarded(F&& body) {
    try {
        body();
        return VH_OK;
    } catch (const HipError& e) {
        int code = (e.err == hipErrorOutOfMemory) ? VH_ERR_NOMEM : VH_ERR_HIP;
        return fail(code, "HIP error %d (%s) in `%s` at %s:%d", (int)e.err, hipGetErrorString(e.err), e.what,
                    e.file, e.line);
    } catch (const InvalidArg& e) {
        return fail(VH_ERR_INVALID, "%s", e.msg.c_str());
    } catch (const std::bad_alloc&) {
        return fail(VH_ERR_NOMEM, "host allocation failed");
    } catch (...) {
        return fail(VH_ERR_HIP, "unknown C++ exception");
    }
}

// Guard mode (option debug.guard_bytes, off by default; tools/gpu/gpu_guard_check.py): every device allocation is followed by
// that many canary bytes, vh_debug_check_guards reports the allocations whose canary was overwritten -- the search tool for
// a kernel that stores past the end of a buffer (no GPU address sanitizer on this pool).  Defined in cluster.hip.
size_t guard_bytes();
void guard_track(void* p, size_t bytes, size_t guard);
void guard_forget(void* p);

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool borrowed = false;   // p belongs to another DevBuf (borrow): never freed here
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), borrowed(o.borrowed) { o.p = nullptr; o.n = 0; o.borrowed = false; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; borrowed = o.borrowed; o.p = nullptr; o.n = 0; o.borrowed = false; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p && !borrowed) {
            guard_forget(p);
            (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
        borrowed = false;
    }
    // alias another buffer's storage (the owner must outlive this view)
    void borrow(const DevBuf& owner) {
        release();
        p = owner.p;
        n = owner.n;
        borrowed = true;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        const size_t guard = guard_bytes();
        VH_HIP(hipMalloc((void**)&p, count * sizeof(T) + guard));
        if (guard) guard_track(p, count * sizeof(T), guard);
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
    size_t bytes() const { return n * sizeof(T); }
};

// Pinned host staging buffer (D2H of small results without an extra pageable bounce).
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        VH_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool enabled = false;
    float last_ms = 0.f;
    ~EventTimer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    void enable(bool on) {
        if (on && !a) {
            VH_HIP(hipEventCreate(&a));
            VH_HIP(hipEventCreate(&b));
        }
        enabled = on;
    }
    void start(hipStream_t s) { if (enabled) VH_HIP(hipEventRecord(a, s)); }
    void stop(hipStream_t s) { if (enabled) VH_HIP(hipEventRecord(b, s)); }
    // call after the stream has been synchronised
    void collect() { if (enabled) VH_HIP(hipEventElapsedTime(&last_ms, a, b)); }
};

// Process-wide tuning / diagnostic options (vh_set_option / vh_get_option, defined in cluster.hip).  The library reads NO
// environment variables: the Python layer forwards VAMBHIP_* variables through vh_set_option (vamb_amd/_lib.py).
int64_t option(const char* name, int64_t dflt);
const char* option_string(const char* name);   // nullptr when unset

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

}  // namespace vh
