// Host-only pieces of the C ABI: error slot, ABI version, hash-grid level table.
#include <math.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>
#include <stdio.h>
#include "common.hpp"

namespace dns {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- one-time kernel attributes (see common.hpp) ----
static AttrInitFn g_attr_fns[16];
static int g_n_attr_fns = 0;                       // filled by static constructors at load time (single-threaded)
static std::atomic<uint64_t> g_ready_devices{0};   // bit d: device d has its attributes
static std::mutex g_init_mutex;

AttrRegistrar::AttrRegistrar(AttrInitFn fn) {
  if (g_n_attr_fns < 16) g_attr_fns[g_n_attr_fns++] = fn;
}

// ---- sticky device-side error word (see common.hpp / dns_hip.h) ----
static uint32_t* g_err_host[64];             // pinned + mapped, one word per device, allocated by init_current_device()
static uint32_t* g_err_dev[64];

uint32_t* device_error_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return g_err_dev[dev];
}

int poll_device_error(const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_err_host[dev]) return DNS_OK;
  const uint32_t w = *reinterpret_cast<volatile uint32_t*>(g_err_host[dev]);
  if (!w) return DNS_OK;
  set_error("%s: a kernel of this library hit a device-side capacity check (error word 0x%x%s); results since then are "
            "incomplete -- dns_device_error(1) clears the word", what, w,
            (w & DNS_DEVERR_GROUP_CURSOR) ? ": group cursor past n_slots in dns_group_slots / dns_group_scatter"
            : (w & DNS_DEVERR_MLP_RANGE) ? ": dns_mlp_bwd was handed a row index beyond 2 GiB of its dy / d_x matrix" : "");
  return DNS_E_LAUNCH;
}

static int init_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("dns_init: no current HIP device");
    return DNS_E_STATE;
  }
  if (g_ready_devices.load(std::memory_order_acquire) & (1ull << dev)) return DNS_OK;
  std::lock_guard<std::mutex> lock(g_init_mutex);
  if (g_ready_devices.load(std::memory_order_acquire) & (1ull << dev)) return DNS_OK;
  for (int i = 0; i < g_n_attr_fns; ++i) {
    const int rc = g_attr_fns[i]();
    if (rc != DNS_OK) return rc;
  }
  if (!g_err_host[dev]) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      set_error("dns_init: cannot allocate the pinned error word: %s", hipGetErrorString(hipGetLastError()));
      return DNS_E_LAUNCH;
    }
    *reinterpret_cast<volatile uint32_t*>(h) = 0u;
    g_err_host[dev] = reinterpret_cast<uint32_t*>(h);
    g_err_dev[dev] = reinterpret_cast<uint32_t*>(d);
  }
  g_ready_devices.fetch_or(1ull << dev, std::memory_order_release);
  return DNS_OK;
}

int ensure_ready(hipStream_t st, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 &&
      (g_ready_devices.load(std::memory_order_acquire) & (1ull << dev)))
    return DNS_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    set_error("%s: first use of the library on this device inside a stream capture; call dns_init() before capturing", who);
    return DNS_E_STATE;
  }
  (void)hipGetLastError();
  return init_current_device();
}

// ---- fill by kernel (see common.hpp: no memset nodes in captured graphs) ----
__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* __restrict__ dst, uint32_t value, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = value;
}
int fill_words(void* dst, uint32_t value, size_t n_words, hipStream_t st, const char* who) {
  if (n_words == 0) return DNS_OK;
  const size_t want = (n_words + 255) / 256;
  const uint32_t blocks = (uint32_t)(want < 2048 ? want : 2048);
  DNS_LAUNCH(fill_words_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst), value, n_words);
  return check_launch(who);
}

// two ranges in one launch (dns_group_slots: the class counters and the slot -> row table)
__global__ __launch_bounds__(256) void fill_words2_kernel(uint32_t* __restrict__ a, uint32_t va, size_t na, uint32_t* __restrict__ b,
                                                          uint32_t vb, size_t nb) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < na) a[i] = va;
    else b[i - na] = vb;
  }
}
int fill_words2(void* a, uint32_t va, size_t na, void* b, uint32_t vb, size_t nb, hipStream_t st, const char* who) {
  if (na + nb == 0) return DNS_OK;
  const size_t want = (na + nb + 255) / 256;
  const uint32_t blocks = (uint32_t)(want < 2048 ? want : 2048);
  DNS_LAUNCH(fill_words2_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(a), va, na, reinterpret_cast<uint32_t*>(b), vb, nb);
  return check_launch(who);
}

// ---- per-kernel event timing (measurement aid; see KernelSpan in common.hpp) ----
struct SpanRec {
  const char* name;
  hipEvent_t e0, e1;
};
static std::atomic<int> g_timing{0};
static std::mutex g_span_mutex;                    // launches come from the caller's thread AND the autograd thread
static std::vector<SpanRec> g_spans;

KernelSpan::KernelSpan(const char* name_, hipStream_t st_) : name(name_), st(st_), e0(nullptr), on(false) {
  if (!g_timing.load(std::memory_order_relaxed)) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;   // no timing in a capture
  if (hipEventCreate(&e0) != hipSuccess) return;
  on = hipEventRecord(e0, st) == hipSuccess;
}
KernelSpan::~KernelSpan() {
  if (!on) return;
  hipEvent_t e1 = nullptr;
  if (hipEventCreate(&e1) != hipSuccess || hipEventRecord(e1, st) != hipSuccess) {
    (void)hipEventDestroy(e0);                     // no span without both ends: nothing is kept, nothing leaks
    if (e1) (void)hipEventDestroy(e1);
    return;
  }
  std::lock_guard<std::mutex> lock(g_span_mutex);
  g_spans.push_back({name, e0, e1});
}
}  // namespace dns

extern "C" int dns_kernel_timing(int enable) {
  std::lock_guard<std::mutex> lock(dns::g_span_mutex);
  if (enable) {
    for (auto& s : dns::g_spans) { (void)hipEventDestroy(s.e0); (void)hipEventDestroy(s.e1); }
    dns::g_spans.clear();
  }
  dns::g_timing.store(enable ? 1 : 0);
  return DNS_OK;
}
extern "C" int dns_kernel_timing_count(void) {
  std::lock_guard<std::mutex> lock(dns::g_span_mutex);
  return (int)dns::g_spans.size();
}
extern "C" int dns_kernel_timing_get(int i, char* name, int name_cap, float* ms) {
  std::lock_guard<std::mutex> lock(dns::g_span_mutex);
  DNS_REQUIRE(i >= 0 && i < (int)dns::g_spans.size(), "dns_kernel_timing_get: span %d of %d", i, (int)dns::g_spans.size());
  DNS_REQUIRE(name != nullptr && name_cap > 1 && ms != nullptr, "dns_kernel_timing_get: NULL output");
  const dns::SpanRec& s = dns::g_spans[i];
  if (hipEventSynchronize(s.e1) != hipSuccess || hipEventElapsedTime(ms, s.e0, s.e1) != hipSuccess) {
    dns::set_error("dns_kernel_timing_get: %s", hipGetErrorString(hipGetLastError()));
    return DNS_E_LAUNCH;
  }
  snprintf(name, (size_t)name_cap, "%s", s.name);
  return DNS_OK;
}

extern "C" int dns_init(void) { return dns::init_current_device(); }
extern "C" int dns_device_error(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !dns::g_err_host[dev]) return 0;
  volatile uint32_t* w = reinterpret_cast<volatile uint32_t*>(dns::g_err_host[dev]);
  const uint32_t v = *w;
  if (clear) *w = 0u;
  return (int)v;
}
extern "C" int dns_abi_version(void) { return DNS_ABI_VERSION; }
extern "C" const char* dns_last_error(void) { return dns::g_err; }

// tcnn GridEncoding constructor arithmetic:
//   scale = exp2(l * log2(pls)) * base - 1;  res = ceilf(scale) + 1;
// tcnn evaluates `scale` in float32 with CUDA's exp2f, whose last bits no other platform reproduces; at the
// finest level the exact value is an integer (desired_resolution - 1), so one ulp flips the resolution.  The
// table is therefore DEFINED here as the float64 value rounded once to float32 (finest level = exactly
// desired_resolution - 1), computed on the host only and handed to every kernel and to the oracle's check.
//   size  = min(next_multiple(res^3, 8), 2^log2_T);
//   dense index while the running stride stays <= size, hashed once it exceeds it.
extern "C" int dns_grid_meta_init(DnsGridMeta* meta, uint32_t n_levels, uint32_t n_features,
                                  uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale) {
  DNS_REQUIRE(meta != nullptr, "dns_grid_meta_init: meta is NULL");
  DNS_REQUIRE(n_levels >= 1 && n_levels <= DNS_MAX_LEVELS, "dns_grid_meta_init: n_levels %u out of range", n_levels);
  DNS_REQUIRE(n_features == 2, "dns_grid_meta_init: only 2 features per level are supported (got %u)", n_features);
  DNS_REQUIRE(log2_hashmap_size >= 3 && log2_hashmap_size <= 28, "dns_grid_meta_init: log2_hashmap_size %u", log2_hashmap_size);
  memset(meta, 0, sizeof(*meta));
  meta->n_levels = n_levels;
  meta->n_features = n_features;
  meta->log2_hashmap_size = log2_hashmap_size;
  meta->base_resolution = base_resolution;
  meta->per_level_scale = (float)per_level_scale;
  const double log2_pls = log2(per_level_scale);
  const uint32_t T = 1u << log2_hashmap_size;
  uint64_t offset = 0;
  for (uint32_t l = 0; l < n_levels; ++l) {
    volatile double e = exp2((double)l * log2_pls);      // volatile: no fused contraction into the mul/sub below
    volatile double m = e * (double)base_resolution;
    const float scale = (float)(m - 1.0);
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    uint64_t dense = (uint64_t)res * res * res;
    dense = (dense + 7u) / 8u * 8u;
    const uint32_t size = dense < T ? (uint32_t)dense : T;
    uint32_t stride = 1;
    for (int d = 0; d < 3 && stride <= size; ++d) stride *= res;   // uint32 wrap, as tcnn
    meta->scale[l] = scale;
    meta->resolution[l] = res;
    meta->size[l] = size;
    meta->offset[l] = (uint32_t)offset;
    meta->hashed[l] = size < stride ? 1u : 0u;
    offset += size;
    DNS_REQUIRE(offset < (1ull << 31), "dns_grid_meta_init: table too large");
  }
  meta->total_rows = (uint32_t)offset;
  return DNS_OK;
}
