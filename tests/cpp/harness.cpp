// C++ harness: the HDK-side bindings of hdk_amd/glue/ EXECUTED, with no Python between main() and the kernels.
//
//   GpuMgr (reference DataMgr/GpuMgr.h, the real header where /root/reference exists)  <- hip_mgr::HipMgr
//   DeviceKernel / KernelOptions / CompilationContext (shaped like QE/DeviceKernel.h:25-65, hdk_decls.h) <- hip_rt::HipKernel
//   QueryMemoryDescriptor accessors (qmd_standin.h)  -> hip_rt::make_plan -> hdk_hip_plan
//   init_group_by_buffer_on_device / fill_hash_join_buff_on_device forwards (HipRuntimeOnDevice.h)
//
// It walks the steps of QueryExecutionContext::launchGpuCode (QE/QueryExecutionContext.cpp:236-480): fetch chunks to
// the device, prepareKernelParams (the 12 pointers, :788-964), initialise the output buffer, create_device_kernel,
// kernel->launch(ko, params), copy error codes and the buffer back.  Two steps:
//   c2    SELECT key, SUM(val), COUNT(*) FROM t GROUP BY key           (GroupByPerfectHash, row-wise, 4 fragments)
//   join  SELECT SUM(val + dval), COUNT(*) FROM fact JOIN dim ON fk = key   (NonGroupedAggregate, one-to-one table)
// and the runtime interrupt; then five steps whose plans come out of the PLAN EXTRACTOR (hdk_amd/glue/HipPlanExtractor.h
// over the stand-in hdk::ir tree of ir_standin.h -- the pattern match NativeCodegen would do for the HIP platform):
//   q3         taxi Q3: GROUP BY passenger_count, extract(year from pickup_datetime) -> COUNT(*)   (hdk_scan_agg_keys)
//   c5         GROUP BY key SUM(val), open addressing, 6 M entries, 8.8 M rows: the radix-partitioned passes with
//              8-byte tuples (column statistics in HipInputCol)
//   projection SELECT key, val * 2 WHERE val < c   (filter/project, TOTAL_MATCHED)
//   floats     GROUP BY k: SUM(f), AVG(f), COUNT(f) over a FLOAT column (float accumulators, HDK_FP_SLOT_FLOAT)
//   reduce     two partial C2 buffers merged on the device with hdk_hip_reduce_buffers
// Results go to stdout, one line per group; tests/test_gpu_cpp_harness.py compares them
// with tests/golden/cpp_harness_output.txt (made by tests/golden/gen_cpp_harness_golden.py with numpy).
#include <algorithm>
#include <array>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "hdk_decls.h"
#include "qmd_standin.h"

#include "HipKernel.h"
#include "HipMgr.h"
#include "HipArena.h"
#include "HipPlanBuilder.h"
#include "HipPlanExtractor.h"
#include "HipRuntimeOnDevice.h"
#include "ir_standin.h"

namespace {

constexpr int64_t kNullBigint = std::numeric_limits<int64_t>::min();
constexpr int kDevice = 0;

uint64_t mix(uint64_t x) {  // splitmix64 finaliser: the data is a pure function of the row number
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
int64_t gen_val(uint64_t i) {
  if (mix(i + (1ull << 41)) % 32 == 0) return kNullBigint;
  return static_cast<int64_t>(mix(i + (1ull << 40)) % 2000001) - 1000000;
}

// what HDK reaches through Executor::getDataMgr(): the device manager and -- through a BufferProvider
// (BufferProvider/BufferProvider.h:23-64; here the standalone arena with the same method set, glue/HipArena.h) -- the
// per-kernel allocator and the copies
struct DeviceServices : hip_rt::HipWorkspaceAllocator {
  explicit DeviceServices(hip_mgr::HipMgr* m) : mgr(m), arena(m) {}
  ~DeviceServices() override {
    for (auto* b : owned) arena.free(b);
  }
  int8_t* alloc(size_t num_bytes) override {
    auto* b = arena.alloc(hip_mgr::ARENA_GPU_LEVEL, kDevice, num_bytes);
    owned.push_back(b);
    return b->getMemoryPtr();
  }
  template <class T>
  int8_t* upload(const std::vector<T>& host) {
    int8_t* d = alloc(host.size() * sizeof(T));
    const size_t bytes = host.size() * sizeof(T);
    if (bytes <= (64u << 10)) {  // the small parameter arrays: staged through the arena's pinned bounce buffer
      arena.copyToDeviceAsyncIfPossible(d, reinterpret_cast<const int8_t*>(host.data()), bytes, kDevice);
      arena.synchronizeStream(kDevice);
    } else {
      arena.copyToDevice(d, reinterpret_cast<const int8_t*>(host.data()), bytes, kDevice);
    }
    return d;
  }
  template <class T>
  std::vector<T> download(const int8_t* dev, size_t n) {
    std::vector<T> host(n);
    arena.copyFromDevice(reinterpret_cast<int8_t*>(host.data()), dev, n * sizeof(T), kDevice);
    return host;
  }
  hip_mgr::HipMgr* mgr;
  hip_mgr::HipArena arena;
  std::vector<hip_mgr::HipArena::Buffer*> owned;
  uint64_t rows_in_step{0};  // QueryExecutionContext knows the fragments of the step
};
DeviceServices* g_services = nullptr;

}  // namespace

// the factory of QE/DeviceKernel.cpp:211-225 with the case it gains
std::unique_ptr<DeviceKernel> create_device_kernel(const CompilationContext* ctx, GpuMgrPlatform platform, int device_id) {
  if (platform == HDK_GPU_PLATFORM_HIP) {
    const auto* hip_ctx = dynamic_cast<const hip_rt::HipPlanContext*>(ctx);
    if (!hip_ctx) throw std::runtime_error("create_device_kernel: not a HIP plan context");
    return std::make_unique<hip_rt::HipKernel>(hip_ctx, device_id, g_services, g_services->rows_in_step, /*timed=*/true,
                                               g_services->mgr->getGridSize());
  }
  throw std::runtime_error("create_device_kernel: platform not built");
}

namespace {

struct ParamBlock {
  std::vector<int8_t*> params;
  int8_t* error_code;
};

// prepareKernelParams: col_buffers[frag][col], NUM_FRAGMENTS, NUM_ROWS / FRAG_ROW_OFFSETS [frag * num_tables + table] ...
ParamBlock prepare_kernel_params(DeviceServices& dev, const std::vector<std::vector<int8_t*>>& col_buffers,
                                 const std::vector<int64_t>& num_rows, const std::vector<uint64_t>& frag_offsets,
                                 uint32_t num_tables, int32_t max_matched, const std::vector<int64_t>& init_agg_vals,
                                 const std::vector<int8_t*>& group_by_buffers, int8_t* join_hash_table) {
  const size_t nfrag = col_buffers.size();
  std::vector<int8_t*> flat;
  for (const auto& f : col_buffers) flat.insert(flat.end(), f.begin(), f.end());
  int8_t* d_flat = dev.upload(flat);
  std::vector<int8_t*> frag_ptrs(nfrag);
  const size_t ncols = nfrag ? col_buffers[0].size() : 0;
  for (size_t f = 0; f < nfrag; ++f) frag_ptrs[f] = d_flat + f * ncols * sizeof(int8_t*);
  ParamBlock pb;
  pb.params.assign(HDK_KP_COUNT, nullptr);
  pb.params[HDK_KP_COL_BUFFERS] = dev.upload(frag_ptrs);
  pb.params[HDK_KP_NUM_FRAGMENTS] = dev.upload(std::vector<uint64_t>{nfrag});
  pb.params[HDK_KP_LITERALS] = nullptr;
  pb.params[HDK_KP_NUM_ROWS] = dev.upload(num_rows);
  pb.params[HDK_KP_FRAG_ROW_OFFSETS] = dev.upload(frag_offsets);
  pb.params[HDK_KP_MAX_MATCHED] = dev.upload(std::vector<int32_t>{max_matched});
  pb.params[HDK_KP_TOTAL_MATCHED] = dev.upload(std::vector<int32_t>{0, 0});
  pb.params[HDK_KP_INIT_AGG_VALS] = dev.upload(init_agg_vals);
  pb.params[HDK_KP_GROUPBY_BUF] = dev.upload(group_by_buffers);
  pb.error_code = dev.upload(std::vector<int32_t>{0});
  pb.params[HDK_KP_ERROR_CODE] = pb.error_code;
  pb.params[HDK_KP_NUM_TABLES] = dev.upload(std::vector<uint32_t>{num_tables});
  pb.params[HDK_KP_JOIN_HASH_TABLES] = join_hash_table;
  return pb;
}

KernelOptions kernel_options(const hip_mgr::HipMgr& mgr) {  // Executor::gridSize()/blockSize() defaults
  KernelOptions ko;
  ko.gridDimX = mgr.getGridSize();
  ko.blockDimX = mgr.getMaxBlockSize();
  return ko;
}

int run_c2(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 4, kFragRows = 500000, kKeys = 64;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;

  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int64_t> key(kFragRows), val(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      key[r] = static_cast<int64_t>(mix(i) % kKeys);
      val[r] = gen_val(i);
    }
    col_buffers.push_back({dev.upload(key), dev.upload(val)});
    num_rows.push_back(kFragRows);
    frag_offsets.push_back(f * kFragRows);
  }

  // what MemoryLayoutBuilder decides for one BIGINT key with range [0, 63] and three 8-byte slots
  QmdStandIn qmd;
  qmd.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
  qmd.group_col_widths_ = {8};
  qmd.padded_slot_widths_ = {8, 8, 8};
  qmd.entry_count_ = kKeys;
  qmd.min_val_ = 0;
  qmd.max_val_ = kKeys - 1;

  hip_rt::HipWorkUnit wu;
  wu.cols = {{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}};
  wu.keys = {hip_rt::column_expr(0, kNullBigint, false)};
  hip_rt::HipTargetDesc key_t{HDK_AGG_ID, true, hip_rt::column_expr(0, kNullBigint, false), false, false, 0, 0};
  hip_rt::HipTargetDesc sum_t{HDK_AGG_SUM, true, hip_rt::column_expr(1, kNullBigint, true), true, false, -1, kNullBigint};
  hip_rt::HipTargetDesc cnt_t{HDK_AGG_COUNT, false, hdk_hip_expr{}, false, false, -1, 0};
  wu.targets = {key_t, sum_t, cnt_t};
  const hip_rt::HipPlanContext ctx(hip_rt::make_plan(wu, qmd));

  const std::vector<int64_t> init_agg_vals{0, kNullBigint, 0};  // init_agg_val_vec: key slot, nullable SUM, COUNT
  int8_t* out = dev.alloc(qmd.getBufferSizeBytes());
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, 0, init_agg_vals, {out}, nullptr);
  const KernelOptions ko = kernel_options(mgr);
  hip_rt::init_group_by_buffer_on_device_hip(reinterpret_cast<int64_t*>(out),
                                             reinterpret_cast<const int64_t*>(pb.params[HDK_KP_INIT_AGG_VALS]),
                                             static_cast<uint32_t>(qmd.getEntryCount()), 1, 8,
                                             static_cast<uint32_t>(qmd.getRowSize() / 8), false, 1, ko.blockDimX,
                                             ko.gridDimX, kDevice);

  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  auto clock = kernel->make_clock();
  clock->start();
  kernel->launch(ko, pb.params);
  const int ms = clock->stop();
  mgr.synchronizeStream(kDevice);
  const int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  std::printf("c2 error_code %d\n", err);
  const size_t quads = qmd.getRowSize() / 8;
  const auto rows = dev.download<int64_t>(out, quads * qmd.getEntryCount());
  for (size_t e = 0; e < qmd.getEntryCount(); ++e) {
    const int64_t* row = &rows[e * quads];
    if (row[0] == std::numeric_limits<int64_t>::max()) continue;  // EMPTY_KEY_64
    std::printf("c2 key %" PRId64 " sum %" PRId64 " count %" PRId64 "\n", row[qmd.getColOffInBytes(0) / 8],
                row[qmd.getColOffInBytes(1) / 8], row[qmd.getColOffInBytes(2) / 8]);
  }
  char names[256];
  hdk_hip_kernel_options o{};
  o.total_rows = dev.rows_in_step;
  hip_rt::check(hdk_hip_describe_launch(&ctx.plan, &o, kDevice, names, sizeof(names)));
  std::fprintf(stderr, "c2: %s, %d ms\n", names, ms);
  g_services = nullptr;
  return err;
}

int run_join(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 3, kFragRows = 400000, kDimRows = 1000, kFkDomain = 1200;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;

  // inner table: one fragment; key is a permutation of [0, 1000), so the table is one-to-one
  std::vector<int64_t> dkey(kDimRows), dval(kDimRows);
  for (size_t i = 0; i < kDimRows; ++i) {
    dkey[i] = static_cast<int64_t>((i * 37) % kDimRows);
    dval[i] = static_cast<int64_t>(mix(i + (1ull << 42)) % 1000);
  }
  int8_t* d_dkey = dev.upload(dkey);
  int8_t* d_dval = dev.upload(dval);

  // PerfectJoinHashTable::reify -> PerfectHashTableBuilder::initOneToOneHashTableOnGpu (PerfectHashTableBuilder.h:93-145)
  const JoinChunk chunk{d_dkey, kDimRows, 0};
  int8_t* d_chunk = dev.upload(std::vector<JoinChunk>{chunk});
  const JoinColumn join_column{d_chunk, sizeof(JoinChunk), 1, kDimRows, 8};
  const JoinColumnTypeInfo type_info{8, 0, static_cast<int64_t>(kDimRows) - 1, kNullBigint, false, 0, Signed};
  int32_t* hash_table = reinterpret_cast<int32_t*>(dev.alloc(kDimRows * sizeof(int32_t)));
  int8_t* d_build_err = dev.upload(std::vector<int32_t>{0});
  hip_rt::init_hash_join_buff_on_device(hash_table, kDimRows, -1, kDevice);
  hip_rt::fill_hash_join_buff_on_device(hash_table, -1, false, reinterpret_cast<int*>(d_build_err), join_column, type_info,
                                        kDevice);
  mgr.synchronizeStream(kDevice);
  const int32_t build_err = dev.download<int32_t>(d_build_err, 1)[0];
  std::printf("join build_error %d\n", build_err);
  {
    // the same table together with its fused form [row id | dval] in one sweep over the inner rows
    // (HipRuntimeOnDevice.h: fill_hash_join_buff_fused_on_device)
    int32_t* table2 = reinterpret_cast<int32_t*>(dev.alloc(kDimRows * sizeof(int32_t)));
    int64_t* fused = reinterpret_cast<int64_t*>(dev.alloc(kDimRows * 2 * sizeof(int64_t)));
    hip_rt::init_hash_join_buff_on_device(table2, kDimRows, -1, kDevice);
    const int8_t* inner_cols[1] = {d_dval};
    const int32_t widths[1] = {8}, kinds[1] = {HDK_COL_INT};
    hip_rt::fill_hash_join_buff_fused_on_device(table2, -1, false, reinterpret_cast<int*>(d_build_err), join_column, type_info, 1,
                                                inner_cols, widths, kinds, 1, fused, nullptr, 0, kDevice);
    mgr.synchronizeStream(kDevice);
    const auto t1 = dev.download<int32_t>(reinterpret_cast<int8_t*>(hash_table), kDimRows);
    const auto t2 = dev.download<int32_t>(reinterpret_cast<int8_t*>(table2), kDimRows);
    const auto fz = dev.download<int64_t>(reinterpret_cast<int8_t*>(fused), kDimRows * 2);
    bool ok = dev.download<int32_t>(d_build_err, 1)[0] == 0;
    for (size_t i = 0; i < kDimRows; ++i) {
      ok = ok && t1[i] == t2[i] && fz[2 * i] == t1[i] && fz[2 * i + 1] == (t1[i] >= 0 ? dval[static_cast<size_t>(t1[i])] : 0);
    }
    std::printf("join fused_build %d\n", ok ? 1 : 0);
  }

  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int64_t> fk(kFragRows), val(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      const uint64_t h = mix(i + (1ull << 43));
      fk[r] = (h >> 32) % 64 == 0 ? kNullBigint : static_cast<int64_t>(h % kFkDomain);
      val[r] = gen_val(i + (1ull << 44));
    }
    col_buffers.push_back({dev.upload(fk), dev.upload(val), d_dval});  // inner columns: the linearised column, per fragment
    num_rows.push_back(kFragRows);
    num_rows.push_back(kDimRows);
    frag_offsets.push_back(f * kFragRows);
    frag_offsets.push_back(0);
  }

  QmdStandIn qmd;
  qmd.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
  qmd.padded_slot_widths_ = {8, 8};

  hip_rt::HipWorkUnit wu;
  wu.cols = {{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}, {1, 8, HDK_COL_INT}};
  hdk_hip_join jn;
  std::memset(&jn, 0, sizeof(jn));
  jn.outer_key = hip_rt::column_expr(0, kNullBigint, true);
  jn.min_key = type_info.min_val;
  jn.max_key = type_info.max_val;
  jn.null_val = kNullBigint;
  jn.kind = HDK_JOIN_ONE_TO_ONE;
  jn.type = HDK_JOIN_INNER;
  jn.null_mode = HDK_JOIN_NULL_NULLABLE;
  jn.table_idx = 0;
  jn.key_component_count = 1;
  jn.key_component_width = 8;
  jn.entry_count = kDimRows;
  wu.joins = {jn};
  hdk_hip_expr val_plus_dval = hip_rt::column_expr(1, kNullBigint, true);  // val + dval, BIGINT, checked
  val_plus_dval.nsteps = 1;
  val_plus_dval.steps[0].op = HDK_OP_ADD;
  val_plus_dval.steps[0].out_class = HDK_VC_INT;
  val_plus_dval.steps[0].rhs = hip_rt::column_expr(2, kNullBigint, false).leaf0;
  val_plus_dval.steps[0].null_out = kNullBigint;
  val_plus_dval.steps[0].check_width = 8;
  hip_rt::HipTargetDesc sum_t{HDK_AGG_SUM, true, val_plus_dval, true, false, -1, kNullBigint};
  hip_rt::HipTargetDesc cnt_t{HDK_AGG_COUNT, false, hdk_hip_expr{}, false, false, -1, 0};
  wu.targets = {sum_t, cnt_t};
  const hip_rt::HipPlanContext ctx(hip_rt::make_plan(wu, qmd));

  // non-grouped: out_vec, one int64 per slot, starting at init_agg_vals (QueryExecutionContext.cpp:452-458)
  const std::vector<int64_t> init_agg_vals{kNullBigint, 0};
  int8_t* out = dev.upload(init_agg_vals);
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 2, 0, init_agg_vals, {out, out + 8},
                                        reinterpret_cast<int8_t*>(hash_table));
  const KernelOptions ko = kernel_options(mgr);
  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel->initializeRuntimeInterrupter();  // launchGpuCode arms it when the interrupt is enabled (:330-340)
  kernel->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  const auto res = dev.download<int64_t>(out, 2);
  std::printf("join error_code %d\n", err);
  std::printf("join sum %" PRId64 " count %" PRId64 "\n", res[0], res[1]);

  // Executor::interrupt() while the step is queued: the launch stops with ERR_INTERRUPTED (QE/Execute.h:1029)
  mgr.copyHostToDevice(out, reinterpret_cast<const int8_t*>(init_agg_vals.data()), 16, kDevice);
  auto kernel2 = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel2->initializeRuntimeInterrupter();
  hip_rt::check(hdk_hip_set_interrupt(kDevice, 1));
  kernel2->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  const int32_t ierr = dev.download<int32_t>(pb.error_code, 1)[0];
  std::printf("interrupt error_code %d\n", ierr);
  hip_rt::check(hdk_hip_set_interrupt(kDevice, 0));

  char names[256];
  hdk_hip_kernel_options o{};
  o.total_rows = dev.rows_in_step;
  hip_rt::check(hdk_hip_describe_launch(&ctx.plan, &o, kDevice, names, sizeof(names)));
  std::fprintf(stderr, "join: %s\n", names);
  g_services = nullptr;
  return err || build_err || ierr != HDK_HIP_ERR_INTERRUPTED;
}


// ---- steps whose plan comes out of the extractor -------------------------------------------------------------------
using namespace standin;
using hdk::ir::AggType;
using hdk::ir::OpType;
using hip_rt::TypeDesc;

hdk_hip_plan extract_plan(std::vector<hip_rt::HipInputCol> cols, const hip_rt::UnitView<StandInIr>& unit, const QmdStandIn& qmd,
                          std::vector<hip_rt::HipWorkUnit::KeyRange> ranges = {}) {
  hip_rt::PlanExtractor<StandInIr> ex(std::move(cols), [](const standin::Expr* e) { return e->column_id; });
  hip_rt::HipWorkUnit wu = ex.extract(unit, {}, qmd.getQueryDescriptionType() == QueryDescriptionType::Projection);
  wu.key_ranges = std::move(ranges);
  return hip_rt::make_plan(wu, qmd);
}

void report_kernels(const char* step, const hdk_hip_plan& plan, uint64_t rows) {
  char names[256];
  hdk_hip_kernel_options o{};
  o.total_rows = rows;
  hip_rt::check(hdk_hip_describe_launch(&plan, &o, kDevice, names, sizeof(names)));
  std::fprintf(stderr, "%s: %s\n", step, names);
}

int run_q3(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 3, kFragRows = 400000;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;
  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int16_t> pc(kFragRows);
    std::vector<int64_t> ts(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      pc[r] = static_cast<int16_t>(mix(i + (1ull << 45)) % 7);
      ts[r] = 1230768000 + static_cast<int64_t>(mix(i + (1ull << 46)) % 220838400ull);  // 2009-01-01 .. 2015-12-31
    }
    col_buffers.push_back({dev.upload(pc), dev.upload(ts)});
    num_rows.push_back(kFragRows);
    frag_offsets.push_back(f * kFragRows);
  }
  auto pc = column(type_of(TypeDesc::Integer, 2, true), 0, 0), ts = column(type_of(TypeDesc::Timestamp, 8, true), 0, 1);
  auto year = extract(hdk::ir::DateExtractField::kYear, ts);
  auto cnt = agg_expr(bigint(false), AggType::kCount, nullptr);
  hip_rt::UnitView<StandInIr> u;
  u.groupby = {pc.get(), year.get()};
  u.targets = {pc.get(), year.get(), cnt.get()};
  QmdStandIn qmd;  // MemoryLayoutBuilder: perfect hash over [0,6] x [2009,2015], three 8-byte slots behind two 8-byte keys
  qmd.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
  qmd.group_col_widths_ = {8, 8};
  qmd.padded_slot_widths_ = {8, 8, 8};
  qmd.entry_count_ = 49;
  const hip_rt::HipPlanContext ctx(extract_plan({{0, 2, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, qmd, {{0, 6, 0, false}, {2009, 2015, 0, false}}));
  const std::vector<int64_t> init_agg_vals{0, 0, 0};
  int8_t* out = dev.alloc(qmd.getBufferSizeBytes());
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, 0, init_agg_vals, {out}, nullptr);
  const KernelOptions ko = kernel_options(mgr);
  hip_rt::init_group_by_buffer_on_device_hip(reinterpret_cast<int64_t*>(out), reinterpret_cast<const int64_t*>(pb.params[HDK_KP_INIT_AGG_VALS]),
                                             49, 2, 8, static_cast<uint32_t>(qmd.getRowSize() / 8), false, 1, ko.blockDimX, ko.gridDimX, kDevice);
  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  const int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  std::printf("q3 error_code %d\n", err);
  const size_t quads = qmd.getRowSize() / 8;
  const auto rows = dev.download<int64_t>(out, quads * 49);
  std::vector<std::array<int64_t, 3>> groups;
  for (size_t e = 0; e < 49; ++e) {
    const int64_t* row = &rows[e * quads];
    if (row[0] == std::numeric_limits<int64_t>::max()) continue;
    groups.push_back({row[qmd.getColOffInBytes(0) / 8], row[qmd.getColOffInBytes(1) / 8], row[qmd.getColOffInBytes(2) / 8]});
  }
  std::sort(groups.begin(), groups.end());
  for (const auto& g : groups) std::printf("q3 passenger_count %" PRId64 " year %" PRId64 " count %" PRId64 "\n", g[0], g[1], g[2]);
  report_kernels("q3", ctx.plan, dev.rows_in_step);
  g_services = nullptr;
  return err;
}

int run_c5(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 8, kFragRows = 1100000, kKeys = 3000000, kEntries = 6000011;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;
  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int64_t> key(kFragRows), val(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      key[r] = static_cast<int64_t>(mix(i + (1ull << 47)) % kKeys);
      val[r] = static_cast<int64_t>(mix(i + (1ull << 48)) % 2000001) - 1000000;
    }
    col_buffers.push_back({dev.upload(key), dev.upload(val)});
    num_rows.push_back(kFragRows);
    frag_offsets.push_back(f * kFragRows);
  }
  auto key = column(bigint(false), 0, 0), val = column(bigint(true), 0, 1);
  auto sum = agg_expr(bigint(true), AggType::kSum, val);
  hip_rt::UnitView<StandInIr> u;
  u.groupby = {key.get()};
  u.targets = {key.get(), sum.get()};
  QmdStandIn qmd;  // GroupByBaselineHash: 4-byte table key (range fits), the projected key has no slot, one 8-byte slot
  qmd.query_desc_type_ = QueryDescriptionType::GroupByBaselineHash;
  qmd.group_col_widths_ = {8};
  qmd.group_col_compact_width_ = 4;
  qmd.padded_slot_widths_ = {0, 8};
  qmd.entry_count_ = kEntries;
  // ChunkStats of the two columns (ChunkMetadata::chunkStats): key and value both fit 32 bits -> 8-byte tuples
  hip_rt::HipInputCol kcol{0, 8, HDK_COL_INT, 1, 0, 0, static_cast<int64_t>(kKeys) - 1};
  hip_rt::HipInputCol vcol{0, 8, HDK_COL_INT, 1, 0, -1000000, 1000000};
  const hip_rt::HipPlanContext ctx(extract_plan({kcol, vcol}, u, qmd));
  const std::vector<int64_t> init_agg_vals{kNullBigint};  // compact_init_vals: one word per quad of the slot region
  int8_t* out = dev.alloc(qmd.getBufferSizeBytes());
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, 0, init_agg_vals, {out}, nullptr);
  const KernelOptions ko = kernel_options(mgr);
  hip_rt::init_group_by_buffer_on_device_hip(reinterpret_cast<int64_t*>(out), reinterpret_cast<const int64_t*>(pb.params[HDK_KP_INIT_AGG_VALS]),
                                             kEntries, 1, 4, 2, false, 1, ko.blockDimX, ko.gridDimX, kDevice);
  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  const int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  std::printf("c5 error_code %d\n", err);
  const auto rows = dev.download<int64_t>(out, 2 * kEntries);
  uint64_t groups = 0, sum_of_sums = 0, mixed = 0;
  for (size_t e = 0; e < kEntries; ++e) {
    const int32_t k = static_cast<int32_t>(rows[2 * e]);  // [key int32 | padding][slot]
    if (k == std::numeric_limits<int32_t>::max()) continue;
    ++groups;
    sum_of_sums += static_cast<uint64_t>(rows[2 * e + 1]);
    mixed ^= mix(static_cast<uint64_t>(k) * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(rows[2 * e + 1]));
  }
  std::printf("c5 groups %" PRIu64 " sum_of_sums %" PRId64 " checksum %" PRIu64 "\n", groups, static_cast<int64_t>(sum_of_sums), mixed);
  report_kernels("c5", ctx.plan, dev.rows_in_step);
  g_services = nullptr;
  return err;
}

int run_projection(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 2, kFragRows = 500000;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;
  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int64_t> val(kFragRows), key(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      key[r] = static_cast<int64_t>(mix(i) % 64);
      val[r] = gen_val(i);
    }
    col_buffers.push_back({dev.upload(val), dev.upload(key)});  // COL_BUFFERS order: first use (the filter reads val)
    num_rows.push_back(kFragRows);
    frag_offsets.push_back(f * kFragRows);
  }
  auto val = column(bigint(true), 0, 0), key = column(bigint(false), 0, 1);
  auto lt = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kLt, val, int_literal(-900000));
  auto twice = bin_oper(bigint(true), OpType::kMul, val, int_literal(2));
  hip_rt::UnitView<StandInIr> u;
  u.quals = {lt.get()};
  u.targets = {key.get(), twice.get()};
  QmdStandIn qmd;  // Projection, row-wise: [int64 row position | two 8-byte slots], entry_count = the scan limit
  qmd.query_desc_type_ = QueryDescriptionType::Projection;
  qmd.group_col_widths_ = {8};
  qmd.padded_slot_widths_ = {8, 8};
  qmd.entry_count_ = kFragments * kFragRows;
  const hip_rt::HipPlanContext ctx(extract_plan({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, qmd));
  int8_t* out = dev.alloc(qmd.getBufferSizeBytes());
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, static_cast<int32_t>(qmd.entry_count_), {0, 0}, {out}, nullptr);
  const KernelOptions ko = kernel_options(mgr);
  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  const int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  const int32_t matched = dev.download<int32_t>(pb.params[HDK_KP_TOTAL_MATCHED], 1)[0];
  std::printf("projection error_code %d matched %d\n", err, matched);
  const auto rows = dev.download<int64_t>(out, 3 * static_cast<size_t>(matched));
  uint64_t sum_pos = 0, sum_key = 0, sum_v2 = 0;  // rows land in claim order: order-free sums
  for (int32_t r = 0; r < matched; ++r) {
    sum_pos += static_cast<uint64_t>(rows[3 * r]);
    sum_key += static_cast<uint64_t>(rows[3 * r + 1]);
    sum_v2 += static_cast<uint64_t>(rows[3 * r + 2]);
  }
  std::printf("projection sum_pos %" PRIu64 " sum_key %" PRIu64 " sum_v2 %" PRId64 "\n", sum_pos, sum_key, static_cast<int64_t>(sum_v2));
  report_kernels("projection", ctx.plan, dev.rows_in_step);
  g_services = nullptr;
  return err;
}

int run_floats(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 2, kFragRows = 300000, kKeys = 10;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = kFragments * kFragRows;
  std::vector<std::vector<int8_t*>> col_buffers;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> frag_offsets;
  const float null_float = std::numeric_limits<float>::min();  // NULL_FLOAT (Shared/InlineNullValues.h:39)
  for (size_t f = 0; f < kFragments; ++f) {
    std::vector<int32_t> k(kFragRows);
    std::vector<float> x(kFragRows);
    for (size_t r = 0; r < kFragRows; ++r) {
      const uint64_t i = f * kFragRows + r;
      k[r] = static_cast<int32_t>(mix(i + (1ull << 49)) % kKeys);
      const uint64_t h = mix(i + (1ull << 50));
      x[r] = h % 16 == 0 ? null_float : static_cast<float>((h >> 8) % 16);  // small integers: float sums are exact in any order
    }
    col_buffers.push_back({dev.upload(k), dev.upload(x)});
    num_rows.push_back(kFragRows);
    frag_offsets.push_back(f * kFragRows);
  }
  auto k = column(type_of(TypeDesc::Integer, 4, false), 0, 0), x = column(type_of(TypeDesc::Fp, 4, true), 0, 1);
  auto sf = agg_expr(type_of(TypeDesc::Fp, 4, true), AggType::kSum, x), af = agg_expr(fp64(true), AggType::kAvg, x),
       cf = agg_expr(bigint(false), AggType::kCount, x);
  hip_rt::UnitView<StandInIr> u;
  u.groupby = {k.get()};
  u.targets = {k.get(), sf.get(), af.get(), cf.get()};
  QmdStandIn qmd;  // perfect hash on [0, 9]; slots: key, SUM(float) padded to 8, AVG = (float sum padded to 8, count), COUNT
  qmd.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
  qmd.group_col_widths_ = {8};
  qmd.padded_slot_widths_ = {8, 8, 8, 8, 8};
  qmd.entry_count_ = kKeys;
  qmd.max_val_ = kKeys - 1;
  const hip_rt::HipPlanContext ctx(extract_plan({{0, 4, HDK_COL_INT}, {0, 4, HDK_COL_FLOAT}}, u, qmd));
  int32_t nf_bits;
  std::memcpy(&nf_bits, &null_float, 4);
  const int64_t fnull = static_cast<int64_t>(nf_bits);  // init_agg_val_vec: the float sentinel sign-extended (OutputBufferInitialization.cpp:52-65)
  const std::vector<int64_t> init_agg_vals{0, fnull, fnull, 0, 0};
  int8_t* out = dev.alloc(qmd.getBufferSizeBytes());
  ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, 0, init_agg_vals, {out}, nullptr);
  const KernelOptions ko = kernel_options(mgr);
  hip_rt::init_group_by_buffer_on_device_hip(reinterpret_cast<int64_t*>(out), reinterpret_cast<const int64_t*>(pb.params[HDK_KP_INIT_AGG_VALS]),
                                             kKeys, 1, 8, static_cast<uint32_t>(qmd.getRowSize() / 8), false, 1, ko.blockDimX, ko.gridDimX, kDevice);
  auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
  kernel->launch(ko, pb.params);
  mgr.synchronizeStream(kDevice);
  const int32_t err = dev.download<int32_t>(pb.error_code, 1)[0];
  std::printf("floats error_code %d\n", err);
  const size_t quads = qmd.getRowSize() / 8;
  const auto rows = dev.download<int64_t>(out, quads * kKeys);
  for (size_t e = 0; e < kKeys; ++e) {
    const int64_t* row = &rows[e * quads];
    if (row[0] == std::numeric_limits<int64_t>::max()) continue;
    float s, a;  // a float accumulator lives in the LOW 4 bytes of its padded slot (takes_float_argument)
    std::memcpy(&s, &row[qmd.getColOffInBytes(1) / 8], 4);
    std::memcpy(&a, &row[qmd.getColOffInBytes(2) / 8], 4);
    std::printf("floats key %" PRId64 " sum %.1f avg_sum %.1f avg_count %" PRId64 " count %" PRId64 "\n", row[qmd.getColOffInBytes(0) / 8],
                static_cast<double>(s), static_cast<double>(a), row[qmd.getColOffInBytes(3) / 8], row[qmd.getColOffInBytes(4) / 8]);
  }
  report_kernels("floats", ctx.plan, dev.rows_in_step);
  g_services = nullptr;
  return err;
}

// ResultSetReduction on the device: C2 over fragments {0, 1} and {2, 3} into two buffers, merged with hdk_hip_reduce_buffers
// (replaces the host's reduceMultiDeviceResultSets, QE/Execute.cpp:1224-1336)
int run_reduce(hip_mgr::HipMgr& mgr) {
  constexpr size_t kFragments = 4, kFragRows = 500000, kKeys = 64;
  DeviceServices dev(&mgr);
  g_services = &dev;
  dev.rows_in_step = 2 * kFragRows;
  auto key = column(bigint(false), 0, 0), val = column(bigint(true), 0, 1);
  auto sum = agg_expr(bigint(true), AggType::kSum, val), cnt = agg_expr(bigint(false), AggType::kCount, nullptr);
  hip_rt::UnitView<StandInIr> u;
  u.groupby = {key.get()};
  u.targets = {key.get(), sum.get(), cnt.get()};
  QmdStandIn qmd;
  qmd.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
  qmd.group_col_widths_ = {8};
  qmd.padded_slot_widths_ = {8, 8, 8};
  qmd.entry_count_ = kKeys;
  qmd.max_val_ = kKeys - 1;
  const hip_rt::HipPlanContext ctx(extract_plan({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, qmd));
  const std::vector<int64_t> init_agg_vals{0, kNullBigint, 0};
  const KernelOptions ko = kernel_options(mgr);
  int8_t* bufs[2];
  int32_t err = 0;
  for (int half = 0; half < 2; ++half) {
    std::vector<std::vector<int8_t*>> col_buffers;
    std::vector<int64_t> num_rows;
    std::vector<uint64_t> frag_offsets;
    for (size_t f = half * 2; f < static_cast<size_t>(half) * 2 + 2 && f < kFragments; ++f) {
      std::vector<int64_t> k(kFragRows), v(kFragRows);
      for (size_t r = 0; r < kFragRows; ++r) {
        const uint64_t i = f * kFragRows + r;
        k[r] = static_cast<int64_t>(mix(i) % kKeys);
        v[r] = gen_val(i);
      }
      col_buffers.push_back({dev.upload(k), dev.upload(v)});
      num_rows.push_back(kFragRows);
      frag_offsets.push_back(f * kFragRows);
    }
    bufs[half] = dev.alloc(qmd.getBufferSizeBytes());
    ParamBlock pb = prepare_kernel_params(dev, col_buffers, num_rows, frag_offsets, 1, 0, init_agg_vals, {bufs[half]}, nullptr);
    hip_rt::init_group_by_buffer_on_device_hip(reinterpret_cast<int64_t*>(bufs[half]), reinterpret_cast<const int64_t*>(pb.params[HDK_KP_INIT_AGG_VALS]),
                                               kKeys, 1, 8, static_cast<uint32_t>(qmd.getRowSize() / 8), false, 1, ko.blockDimX, ko.gridDimX, kDevice);
    auto kernel = create_device_kernel(&ctx, mgr.getPlatform(), kDevice);
    kernel->launch(ko, pb.params);
    mgr.synchronizeStream(kDevice);
    err |= dev.download<int32_t>(pb.error_code, 1)[0];
  }
  int8_t* d_err = dev.upload(std::vector<int32_t>{0});
  const int64_t* that[1] = {reinterpret_cast<const int64_t*>(bufs[1])};
  const uint32_t that_counts[1] = {kKeys};
  hip_rt::check(hdk_hip_reduce_buffers(&ctx.plan, reinterpret_cast<int64_t*>(bufs[0]), kKeys, that, that_counts, 1, init_agg_vals.data(),
                                       reinterpret_cast<int32_t*>(d_err), kDevice, nullptr));
  mgr.synchronizeStream(kDevice);
  err |= dev.download<int32_t>(d_err, 1)[0];
  std::printf("reduce error_code %d\n", err);
  const size_t quads = qmd.getRowSize() / 8;
  const auto rows = dev.download<int64_t>(bufs[0], quads * kKeys);
  for (size_t e = 0; e < kKeys; ++e) {
    const int64_t* row = &rows[e * quads];
    if (row[0] == std::numeric_limits<int64_t>::max()) continue;
    std::printf("reduce key %" PRId64 " sum %" PRId64 " count %" PRId64 "\n", row[qmd.getColOffInBytes(0) / 8], row[qmd.getColOffInBytes(1) / 8],
                row[qmd.getColOffInBytes(2) / 8]);
  }
  g_services = nullptr;
  return err;
}

}  // namespace

int main() {
  try {
    hip_mgr::HipMgr mgr(1, 0);
    GpuMgr& as_reference_interface = mgr;  // everything below could go through the reference's interface
    std::fprintf(stderr, "platform %d, %d device(s), wave %d, grid %u, block %u\n",
                 static_cast<int>(as_reference_interface.getPlatform()), as_reference_interface.getDeviceCount(),
                 static_cast<int>(as_reference_interface.getSubGroupSize()), as_reference_interface.getGridSize(),
                 as_reference_interface.getMaxBlockSize());
    int rc = run_c2(mgr);
    rc |= run_join(mgr);
    rc |= run_q3(mgr);
    rc |= run_c5(mgr);
    rc |= run_projection(mgr);
    rc |= run_floats(mgr);
    rc |= run_reduce(mgr);
    return rc ? 1 : 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "harness failed: %s\n", e.what());
    return 2;
  }
}
