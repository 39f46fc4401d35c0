// Memory-safety gate for the GPU test suite (TEST INFRASTRUCTURE - never loaded by the product path).
//
// A torch.cuda "pluggable allocator" (torch.cuda.memory.CUDAPluggableAllocator) that serves EVERY device tensor of the test
// process from its own virtual-memory mapping, placed FLUSH AGAINST THE END of that mapping:
//
//      reserved VA:  [ mapped pages ................................ ][ one granule, never mapped ]
//                    [ head canary 0xA5.. | payload (size bytes) |t ]
//                                          ^ptr                   ^ tail slack (< 16 bytes, 0xA5) so that ptr stays 16-byte aligned
//
//  * a kernel that READS or WRITES past the end of a tensor touches the unmapped granule: the GPU raises a memory access
//    fault and the process aborts (loud) - exactly the class of bug that hid for two rounds in the attention kernels'
//    keep-bit table loads (csrc/attention_bf16.hip, fixed in round 4) because the caching allocator's 2 MB blocks and
//    neighbouring tensors made the stray reads land in mapped memory 13 runs out of 14;
//  * a kernel that WRITES in front of a tensor, or into the < 16-byte tail slack, trips a canary: checked when the tensor
//    is freed, counted in ttsmi_guard_violations() (tests/conftest.py fails the test that was running) and logged.
//
// Frees synchronise the device first (torch frees a tensor as soon as its last reference dies, in host order; the caching
// allocator would keep the block alive in stream order - here in-flight kernels must finish before the pages go away).
// Slow (a reserve + create + map per tensor, a device sync per free) - which is fine for a gate that runs a few times per
// round:  TTSMI_GUARD_ALLOC=1 python -m pytest tests -m gpu -n 1     (tools/sessions/r05_c.sh; xdist so that a faulting
// test costs one worker, not the run).
//
// Address ranges are NEVER handed back (TTSMI_GUARD_KEEP_VA=1, the default).  Measured on ROCm 7.2 / MI355X
// (tools/sessions/r05_b.sh, profiles/r05_guard_variants.txt): with hipMemAddressFree after every tensor, a later
// hipMemAddressReserve returns the same range and the GPU keeps using the OLD translation for it - 921 of 3 603 tensors of
// tools/guard_stress.py (plain torch fills, no library kernel) read back wrong and a third of all canaries were "violated";
// a device sync around the canary fill changes nothing; without address reuse: 0 / 0.  The ranges cost address space only
// (the pages are unmapped and released), and a test process ends long before 2^47 bytes are reserved.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

struct Rec {
    char* va;
    size_t va_size, mapped, size;
    hipMemGenericAllocationHandle_t handle;
    char* ptr;
    size_t head_checked;       // bytes of head canary in front of ptr that are filled / checked
};

std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
size_t g_gran = 0;
std::atomic<long> g_allocs{0}, g_frees{0}, g_violations{0}, g_live_bytes{0}, g_peak_bytes{0};
constexpr unsigned char kCanary = 0xA5;
constexpr size_t kHeadMax = 4096;

// Probe knobs (tools/sessions/r05_b.sh established which of them the ROCm 7.2 virtual-memory path needs: see the header)
int env_flag(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
const int g_keep_va = env_flag("TTSMI_GUARD_KEEP_VA", 1);          // 1: address ranges are never handed back (no VA reuse)
const int g_sync_alloc = env_flag("TTSMI_GUARD_SYNC_ALLOC", 0);    // 1: device sync before and after the canary fill
const int g_no_release = env_flag("TTSMI_GUARD_NO_RELEASE", 0);    // 1: frees check the canaries but keep the pages mapped

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    std::fflush(stderr);
    std::abort();
}
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(#call, e_); } while (0)

void log_line(const char* msg) {
    std::fprintf(stderr, "%s\n", msg);
    std::fflush(stderr);
    if (const char* path = std::getenv("TTSMI_GUARD_LOG")) {
        if (FILE* f = std::fopen(path, "a")) {
            std::fprintf(f, "%s\n", msg);
            std::fclose(f);
        }
    }
}

struct AtExit {
    ~AtExit() {
        char buf[256];
        std::snprintf(buf, sizeof buf, "guard_alloc: %ld allocations, %ld frees, peak %.1f MB live, %ld canary violations",
                      g_allocs.load(), g_frees.load(), g_peak_bytes.load() / 1e6, g_violations.load());
        log_line(buf);
    }
} g_at_exit;

}  // namespace

extern "C" {

// torch.cuda.memory.CUDAPluggableAllocator entry points
void* ttsmi_guard_alloc(size_t size, int device, hipStream_t /*stream*/) {
    std::lock_guard<std::mutex> lock(g_mu);
    int prev = 0;
    CK(hipGetDevice(&prev));
    if (prev != device) CK(hipSetDevice(device));
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) {
        CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
        if (g_gran < 4096) g_gran = 4096;
    }
    if (size == 0) size = 1;
    const size_t payload = round_up(size, 16);
    Rec r;
    r.size = size;
    r.mapped = round_up(payload + 256, g_gran);                 // >= 256 bytes of head canary
    r.va_size = r.mapped + g_gran;                              // + the granule that stays unmapped
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, r.va_size, 0, nullptr, 0));
    r.va = static_cast<char*>(va);
    CK(hipMemCreate(&r.handle, r.mapped, &prop, 0));
    CK(hipMemMap(r.va, r.mapped, 0, r.handle, 0));
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(r.va, r.mapped, &acc, 1));
    r.ptr = r.va + r.mapped - payload;
    if (g_sync_alloc) CK(hipDeviceSynchronize());
    r.head_checked = static_cast<size_t>(r.ptr - r.va) < kHeadMax ? static_cast<size_t>(r.ptr - r.va) : kHeadMax;
    CK(hipMemset(r.ptr - r.head_checked, kCanary, r.head_checked));
    if (payload > size) CK(hipMemset(r.ptr + size, kCanary, payload - size));
    if (const char* p = std::getenv("TTSMI_GUARD_POISON")) {   // optional: fresh memory reads as NaN (fp32 and bf16)
        if (p[0] == '1') CK(hipMemset(r.ptr, 0xFF, size));
    }
    if (g_sync_alloc) CK(hipDeviceSynchronize());
    g_live[r.ptr] = r;
    g_allocs++;
    long live = (g_live_bytes += static_cast<long>(r.mapped));
    long peak = g_peak_bytes.load();
    while (live > peak && !g_peak_bytes.compare_exchange_weak(peak, live)) {}
    if (prev != device) CK(hipSetDevice(prev));
    return r.ptr;
}

void ttsmi_guard_free(void* ptr, size_t /*size*/, int device, hipStream_t /*stream*/) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        log_line("guard_alloc: free of a pointer this allocator did not hand out");
        g_violations++;
        return;
    }
    Rec r = it->second;
    g_live.erase(it);
    int prev = 0;
    CK(hipGetDevice(&prev));
    if (prev != device) CK(hipSetDevice(device));
    CK(hipDeviceSynchronize());                                 // every kernel that may still touch the pages
    const size_t payload = round_up(r.size, 16);
    std::vector<unsigned char> host(r.head_checked + (payload - r.size));
    if (r.head_checked) CK(hipMemcpy(host.data(), r.ptr - r.head_checked, r.head_checked, hipMemcpyDeviceToHost));
    if (payload > r.size) CK(hipMemcpy(host.data() + r.head_checked, r.ptr + r.size, payload - r.size, hipMemcpyDeviceToHost));
    long first_bad = -1, n_bad = 0;
    for (size_t i = 0; i < host.size(); ++i)
        if (host[i] != kCanary) {
            if (first_bad < 0) first_bad = static_cast<long>(i);
            ++n_bad;
        }
    if (n_bad) {
        char buf[320];
        const bool in_head = static_cast<size_t>(first_bad) < r.head_checked;
        std::snprintf(buf, sizeof buf,
                      "guard_alloc: CANARY VIOLATION on a %zu-byte tensor at %p: %ld bytes overwritten, first %s (offset %ld)",
                      r.size, static_cast<void*>(r.ptr), n_bad, in_head ? "IN FRONT of the tensor" : "in the tail slack behind it",
                      in_head ? first_bad - static_cast<long>(r.head_checked) : static_cast<long>(r.size) + first_bad - static_cast<long>(r.head_checked));
        log_line(buf);
        g_violations++;
    }
    if (!g_no_release) {
        CK(hipMemUnmap(r.va, r.mapped));
        CK(hipMemRelease(r.handle));
        if (!g_keep_va) CK(hipMemAddressFree(r.va, r.va_size));
    }
    g_frees++;
    g_live_bytes -= static_cast<long>(r.mapped);
    if (prev != device) CK(hipSetDevice(prev));
}

// read by tests/conftest.py after every test
long ttsmi_guard_violations(void) { return g_violations.load(); }
long ttsmi_guard_allocations(void) { return g_allocs.load(); }
long ttsmi_guard_live_bytes(void) { return g_live_bytes.load(); }

}  // extern "C"
