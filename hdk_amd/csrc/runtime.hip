// runtime.hip -- HipMgr: the device manager that sits under HDK's BufferProvider.
//
// New code for the `GpuMgr` contract (reference omniscidb/DataMgr/GpuMgr.h:29-79); the reference's
// two implementations are CudaMgr (omniscidb/CudaMgr/CudaMgr.h:83-260, CUDA driver API) and
// L0Manager.  One HIP-runtime function per virtual, exported through the C ABI so that a
// `class HipMgr : public GpuMgr` in HDK is a list of one-line forwards.
#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "host_common.h"
#include "switches.h"

// ---- the HDK_HIP_* switches (switches.h): read once, re-read on request ---------------------------------------------------
namespace hdk {
namespace {
constexpr int kSwitchLen = 64;
const char* const kSwitchNames[SW_COUNT] = {
#define HDK_SW_NAME(name) "HDK_HIP_" #name,
    HDK_SWITCH_LIST(HDK_SW_NAME)
#undef HDK_SW_NAME
};
struct SwitchTable {
  bool set[SW_COUNT];
  char val[SW_COUNT][kSwitchLen];
};
SwitchTable g_switches;
std::once_flag g_switches_once;
void read_switches() {
  for (int i = 0; i < SW_COUNT; ++i) {
    const char* e = getenv(kSwitchNames[i]);
    g_switches.set[i] = e != nullptr;
    if (e) {
      strncpy(g_switches.val[i], e, kSwitchLen - 1);
      g_switches.val[i][kSwitchLen - 1] = 0;
    } else {
      g_switches.val[i][0] = 0;
    }
  }
}
}  // namespace
const char* hdk_sw(SwitchId id) {
  std::call_once(g_switches_once, read_switches);
  return g_switches.set[id] ? g_switches.val[id] : nullptr;
}
}  // namespace hdk

extern "C" void hdk_hip_reload_switches(void) {
  std::call_once(hdk::g_switches_once, hdk::read_switches);
  hdk::read_switches();
}

namespace hdk {

static thread_local char tl_error[1024] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tl_error, sizeof(tl_error), fmt, ap);
  va_end(ap);
}
void clear_error() { tl_error[0] = 0; }

constexpr int kMaxDevices = 16;  // Executor::max_gpu_count (reference QE/Execute.h:959-960)

struct DeviceState {
  std::once_flag once;
  hipStream_t stream = nullptr;
  hdk_hip_device_properties props;
  int32_t status = HDK_HIP_OK;
  int32_t* interrupt_word = nullptr;  // device memory: != 0 stops launches that poll it (hdk_hip_set_interrupt)
  hipStream_t interrupt_stream = nullptr;  // the flag travels on a stream of its own: it overtakes the kernels it stops
  std::mutex interrupt_mu;
};
static DeviceState g_dev[kMaxDevices];

static void init_device(int32_t d) {
  DeviceState& s = g_dev[d];
  hipDeviceProp_t hp;
  if (hipSetDevice(d) != hipSuccess || hipGetDeviceProperties(&hp, d) != hipSuccess ||
      hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&s.interrupt_stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s.interrupt_word), 256) != hipSuccess ||
      hipMemset(s.interrupt_word, 0, 256) != hipSuccess) {
    s.status = HDK_HIP_ERR_RUNTIME;
    return;
  }
  // Stream-ordered scratch (hipMallocAsync in the multi-pass strategies and the reductions) is recycled
  // by the device's default pool; without a release threshold the pool hands freed memory back at every
  // synchronisation and the next launch pays the mapping of its scratch again (measured: a 10 GB scratch
  // turned a 9.5 ms launch into 58 ms).  Keep up to a quarter of the device's memory cached (HDK_HIP_POOL_KEEP_MB
  // overrides); hdk_hip_mgr_allocate_device_mem trims the pool and retries when a plain hipMalloc runs out.
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, d) == hipSuccess && pool) {
    uint64_t keep = static_cast<uint64_t>(hp.totalGlobalMem) / 4;
    if (const char* e = hdk_sw(SW_POOL_KEEP_MB)) {
      keep = static_cast<uint64_t>(strtoull(e, nullptr, 10)) << 20;
    }
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
  }
  (void)hipGetLastError();
  memset(&s.props, 0, sizeof(s.props));
  s.props.global_mem = hp.totalGlobalMem;
  s.props.num_cu = hp.multiProcessorCount;
  s.props.max_threads_per_block = hp.maxThreadsPerBlock;
  s.props.wavefront_size = hp.warpSize;
  // CudaMgr::getGridSize is 2 x #SM (CudaMgr.h:176); the persistent kernels here want 4 blocks of
  // 256 threads per CU (16 waves/CU) -- measured choice, see DESIGN.md.
  s.props.grid_size = 4 * hp.multiProcessorCount;
  s.props.shared_mem_per_block = hp.sharedMemPerBlock;
  s.props.has_shared_memory_atomics = 1;
  s.props.can_load_async = 1;
  s.props.has_fp64 = 1;
  s.props.clock_khz = hp.clockRate;
  s.props.memory_clock_khz = hp.memoryClockRate;
  s.props.memory_bus_width = hp.memoryBusWidth;
  strncpy(s.props.arch_name, hp.gcnArchName, sizeof(s.props.arch_name) - 1);
}

static int32_t ensure_device(int32_t d) {
  if (d < 0 || d >= kMaxDevices) {
    set_error("device %d out of range", d);
    return HDK_HIP_ERR_INVALID_ARG;
  }
  int n = 0;
  HDK_HIP_CHECK(hipGetDeviceCount(&n));
  if (d >= n) {
    set_error("device %d out of range: %d device(s) visible", d, n);
    return HDK_HIP_ERR_INVALID_ARG;
  }
  std::call_once(g_dev[d].once, init_device, d);
  if (g_dev[d].status != HDK_HIP_OK) {
    set_error("failed to initialise device %d", d);
  }
  return g_dev[d].status;
}

int32_t device_enter(int32_t device_id, void* stream, hipStream_t* out) {
  const int32_t st = ensure_device(device_id);
  if (st != HDK_HIP_OK) {
    return st;
  }
  HDK_HIP_CHECK(hipSetDevice(device_id));
  *out = stream ? static_cast<hipStream_t>(stream) : g_dev[device_id].stream;
  return HDK_HIP_OK;
}

const hdk_hip_device_properties* device_props(int32_t device_id) {
  return ensure_device(device_id) == HDK_HIP_OK ? &g_dev[device_id].props : nullptr;
}

const int32_t* device_interrupt_word(int32_t device_id) {
  return ensure_device(device_id) == HDK_HIP_OK ? g_dev[device_id].interrupt_word : nullptr;
}

}  // namespace hdk

using namespace hdk;

extern "C" int32_t hdk_hip_set_interrupt(int32_t device_id, int32_t value) {
  hipStream_t main_stream;
  const int32_t st = device_enter(device_id, nullptr, &main_stream);
  if (st) return st;
  DeviceState& d = g_dev[device_id];
  std::lock_guard<std::mutex> lk(d.interrupt_mu);
  HDK_HIP_CHECK(hipMemcpyAsync(d.interrupt_word, &value, sizeof(int32_t), hipMemcpyHostToDevice, d.interrupt_stream));
  HDK_HIP_CHECK(hipStreamSynchronize(d.interrupt_stream));
  return HDK_HIP_OK;
}

extern "C" {

const char* hdk_hip_last_error(void) { return tl_error; }
int32_t hdk_hip_version(void) { return 1000; }

int32_t hdk_hip_mgr_get_device_count(int32_t* count) {
  HDK_REQUIRE(count, "count is NULL");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;  // like CudaMgr on a CPU-only box: no devices rather than a hard failure
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return HDK_HIP_ERR_RUNTIME;
  }
  *count = n;
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_set_context(int32_t device_num) {
  hipStream_t s;
  return device_enter(device_num, nullptr, &s);
}

int32_t hdk_hip_mgr_allocate_device_mem(size_t num_bytes, int32_t device_num, int8_t** device_ptr) {
  HDK_REQUIRE(device_ptr, "device_ptr is NULL");
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, num_bytes ? num_bytes : 1);
  if (e == hipErrorOutOfMemory) {
    // the stream-ordered pool may be sitting on freed scratch: give it back to the driver and try once more
    (void)hipGetLastError();
    hipMemPool_t pool = nullptr;
    if (hipDeviceSynchronize() == hipSuccess && hipDeviceGetDefaultMemPool(&pool, device_num) == hipSuccess && pool) {
      (void)hipMemPoolTrimTo(pool, 0);
    }
    (void)hipGetLastError();
    e = hipMalloc(&p, num_bytes ? num_bytes : 1);
  }
  HDK_HIP_CHECK(e);
  *device_ptr = static_cast<int8_t*>(p);
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_free_device_mem(int8_t* device_ptr) {
  if (device_ptr) {
    HDK_HIP_CHECK(hipFree(device_ptr));
  }
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_allocate_pinned_host_mem(size_t num_bytes, int8_t** host_ptr) {
  HDK_REQUIRE(host_ptr, "host_ptr is NULL");
  void* p = nullptr;
  HDK_HIP_CHECK(hipHostMalloc(&p, num_bytes ? num_bytes : 1, hipHostMallocDefault));
  *host_ptr = static_cast<int8_t*>(p);
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_free_pinned_host_mem(int8_t* host_ptr) {
  if (host_ptr) {
    HDK_HIP_CHECK(hipHostFree(host_ptr));
  }
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_copy_host_to_device(int8_t* device_ptr, const int8_t* host_ptr, size_t num_bytes,
                                        int32_t device_num) {
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipMemcpy(device_ptr, host_ptr, num_bytes, hipMemcpyHostToDevice));
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_copy_host_to_device_async(int8_t* device_ptr, const int8_t* host_ptr,
                                              size_t num_bytes, int32_t device_num) {
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipMemcpyAsync(device_ptr, host_ptr, num_bytes, hipMemcpyHostToDevice, s));
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_synchronize_stream(int32_t device_num) {
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipStreamSynchronize(s));
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_copy_device_to_host(int8_t* host_ptr, const int8_t* device_ptr, size_t num_bytes,
                                        int32_t device_num) {
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  // results of kernels enqueued on the manager's stream must be complete (CudaMgr copies are
  // synchronous on the context that ran the kernel)
  HDK_HIP_CHECK(hipStreamSynchronize(s));
  HDK_HIP_CHECK(hipMemcpy(host_ptr, device_ptr, num_bytes, hipMemcpyDeviceToHost));
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_copy_device_to_device(int8_t* dest_ptr, int8_t* src_ptr, size_t num_bytes,
                                          int32_t dest_device_num, int32_t src_device_num) {
  hipStream_t s;
  const int32_t st = device_enter(src_device_num, nullptr, &s);
  if (st) return st;
  if (dest_device_num == src_device_num) {
    HDK_HIP_CHECK(hipMemcpy(dest_ptr, src_ptr, num_bytes, hipMemcpyDeviceToDevice));
  } else {
    HDK_HIP_CHECK(hipMemcpyPeer(dest_ptr, dest_device_num, src_ptr, src_device_num, num_bytes));
  }
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_zero_device_mem(int8_t* device_ptr, size_t num_bytes, int32_t device_num) {
  return hdk_hip_mgr_set_device_mem(device_ptr, 0, num_bytes, device_num);
}

int32_t hdk_hip_mgr_set_device_mem(int8_t* device_ptr, unsigned char uc, size_t num_bytes,
                                   int32_t device_num) {
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipMemset(device_ptr, uc, num_bytes));
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_synchronize_devices(void) {
  int n = 0;
  HDK_HIP_CHECK(hipGetDeviceCount(&n));
  for (int d = 0; d < n; ++d) {
    HDK_HIP_CHECK(hipSetDevice(d));
    HDK_HIP_CHECK(hipDeviceSynchronize());
  }
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_get_stream(int32_t device_num, void** stream) {
  HDK_REQUIRE(stream, "stream is NULL");
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  *stream = s;
  return HDK_HIP_OK;
}

// ---- hipGraph capture / replay of a launch sequence (include/hdk_hip.h) ----------------------------------
int32_t hdk_hip_graph_begin_capture(int32_t device_id, void* stream) {
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  return HDK_HIP_OK;
}

int32_t hdk_hip_graph_end_capture(int32_t device_id, void* stream, void** graph_exec) {
  HDK_REQUIRE(graph_exec, "graph_exec is NULL");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  hipGraph_t graph = nullptr;
  HDK_HIP_CHECK(hipStreamEndCapture(s, &graph));
  HDK_REQUIRE(graph, "stream capture produced no graph (a captured call was not capturable)");
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  HDK_HIP_CHECK(e);
  *graph_exec = exec;
  return HDK_HIP_OK;
}

int32_t hdk_hip_graph_launch(void* graph_exec, int32_t device_id, void* stream) {
  HDK_REQUIRE(graph_exec, "graph_exec is NULL");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  HDK_HIP_CHECK(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), s));
  return HDK_HIP_OK;
}

int32_t hdk_hip_graph_destroy(void* graph_exec) {
  if (graph_exec) {
    HDK_HIP_CHECK(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  }
  return HDK_HIP_OK;
}

int32_t hdk_hip_mgr_get_device_properties(int32_t device_num, hdk_hip_device_properties* out) {
  HDK_REQUIRE(out, "out is NULL");
  const hdk_hip_device_properties* p = device_props(device_num);
  if (!p) {
    return HDK_HIP_ERR_RUNTIME;
  }
  *out = *p;
  return HDK_HIP_OK;
}

// ---- HBM calibration (measurement helper): what a plain streaming kernel reaches on THIS device -------------------
// SURVEY.md 8(d) asks for the measured peak next to the nominal one.  16 bytes per lane per load, 8 loads in flight,
// persistent grid -- the access shape of the scan kernels.
}  // extern "C"
namespace {
typedef float __attribute__((ext_vector_type(4))) cal_f4;
__global__ __launch_bounds__(256) void k_cal_copy(const cal_f4* __restrict__ src, cal_f4* __restrict__ dst, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  }
}
// read: the access shape of the streaming scan kernels -- tiles of 256 lanes x 8 x 16 B, a lane's eight loads issued
// back to back, blocks striding over the tiles
__global__ __launch_bounds__(256) void k_cal_read(const cal_f4* __restrict__ src, float* sink, size_t n) {
  constexpr size_t kTile = 256 * 8;
  cal_f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t ntiles = n / kTile;
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const cal_f4* p = src + t * kTile + threadIdx.x;
    cal_f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = __builtin_nontemporal_load(p + u * 256);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc += v[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) {
    *sink = acc.x;
  }
}
}  // namespace
extern "C" {

int32_t hdk_hip_mgr_measure_hbm(int32_t device_num, size_t bytes, int32_t reps, double* copy_gbps, double* read_gbps) {
  HDK_REQUIRE(copy_gbps && read_gbps && reps > 0 && bytes >= (1u << 20), "bad arguments");
  hipStream_t s;
  const int32_t st = device_enter(device_num, nullptr, &s);
  if (st) return st;
  const hdk_hip_device_properties* props = device_props(device_num);
  const size_t n = bytes / 16;
  // (every early return below -- HDK_HIP_CHECK -- releases the calibration buffers and the events)
  struct Guard {
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Guard() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
    }
  } g;
  if (hipMalloc(&g.a, n * 16) != hipSuccess || hipMalloc(&g.b, n * 16) != hipSuccess) {
    (void)hipGetLastError();
    set_error("hipMalloc failed for the calibration buffers");
    return HDK_HIP_ERR_OUT_OF_GPU_MEM;
  }
  void *const a = g.a, *const b = g.b;
  HDK_HIP_CHECK(hipMemsetAsync(a, 1, n * 16, s));
  HDK_HIP_CHECK(hipMemsetAsync(b, 0, n * 16, s));
  HDK_HIP_CHECK(hipEventCreate(&g.e0));
  HDK_HIP_CHECK(hipEventCreate(&g.e1));
  const hipEvent_t e0 = g.e0, e1 = g.e1;
  const unsigned grid = static_cast<unsigned>(props->num_cu) * 8;
  float ms = 0.f;
  double best_copy = 0, best_read = 0;
  for (int r = 0; r < reps + 1; ++r) {  // (first round warms up)
    HDK_HIP_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_cal_copy, dim3(grid), dim3(256), 0, s, static_cast<const cal_f4*>(a), static_cast<cal_f4*>(b), n);
    HDK_HIP_CHECK(hipGetLastError());
    HDK_HIP_CHECK(hipEventRecord(e1, s));
    HDK_HIP_CHECK(hipEventSynchronize(e1));
    HDK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms > 0) best_copy = std::max(best_copy, 2.0 * n * 16 / (ms * 1e-3) / 1e9);
    for (unsigned per_cu = 2; per_cu <= 8; per_cu *= 2) {  // the streaming kernels run 2-4 blocks per CU
      HDK_HIP_CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_cal_read, dim3(static_cast<unsigned>(props->num_cu) * per_cu), dim3(256), 0, s,
                         static_cast<const cal_f4*>(a), static_cast<float*>(b), n);
      HDK_HIP_CHECK(hipGetLastError());
      HDK_HIP_CHECK(hipEventRecord(e1, s));
      HDK_HIP_CHECK(hipEventSynchronize(e1));
      HDK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms > 0) best_read = std::max(best_read, 1.0 * (n / 2048 * 2048) * 16 / (ms * 1e-3) / 1e9);
    }
  }
  *copy_gbps = best_copy;
  *read_gbps = best_read;
  return HDK_HIP_OK;
}

// sizes of the ABI PODs, for the ctypes mirror's self-check (tests/test_abi.py)
size_t hdk_hip_sizeof_plan(void) { return sizeof(hdk_hip_plan); }
size_t hdk_hip_sizeof_expr(void) { return sizeof(hdk_hip_expr); }
size_t hdk_hip_sizeof_target(void) { return sizeof(hdk_hip_target); }
size_t hdk_hip_sizeof_qual(void) { return sizeof(hdk_hip_qual); }
size_t hdk_hip_sizeof_join(void) { return sizeof(hdk_hip_join); }
size_t hdk_hip_sizeof_device_properties(void) { return sizeof(hdk_hip_device_properties); }
size_t hdk_hip_sizeof_kernel_options(void) { return sizeof(hdk_hip_kernel_options); }

}  // extern "C"
