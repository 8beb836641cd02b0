// C++ harness for the multi-device merge (hdk_amd/glue/HipReduce.h): ONE process, every device HipMgr reports, RCCL.
//
// The reference reduces multi-device results on the host (QE/Execute.cpp:1224-1336,2606-2641); here the partial results
// stay in HBM and move over RCCL.  Fragment f of a table lives on device f mod G; two steps, the data of
// tests/cpp/harness.cpp's c2 and c5 steps (so their goldens carry over, whatever G is):
//   mg_c2   GROUP BY key: SUM(val), COUNT(*) -- perfect hash, every device scans its fragments, ncclAllGather of the
//           partial buffers + hdk_hip_reduce_buffers: every device ends with the merged buffer (all of them are compared)
//   mg_c5   GROUP BY key SUM(val), open addressing, 6 M entries: hdk_hip_scatter_to_owners -> grouped ncclSend / ncclRecv
//           with equal splits -> hdk_hip_aggregate_from_ranks; the owners' tables hold disjoint key sets
// One host thread per device (HipDeviceGroup::parallel); a failed HIP / RCCL / library call ends the program non-zero.
// On a one-GPU box G = 1: every kernel and both collectives still run (a rank exchanging with itself).
#include <algorithm>
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
#include "HipPlanBuilder.h"
#include "HipPlanExtractor.h"
#include "HipReduce.h"
#include "ir_standin.h"

namespace {

using namespace standin;
using hdk::ir::AggType;

constexpr int64_t kNullBigint = std::numeric_limits<int64_t>::min();

uint64_t mix(uint64_t x) {  // (the data definition of harness.cpp)
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
int64_t gen_val(uint64_t i) {
  if (mix(i + (1ull << 41)) % 32 == 0) return kNullBigint;
  return static_cast<int64_t>(mix(i + (1ull << 40)) % 2000001) - 1000000;
}

hdk_hip_plan extract_plan(std::vector<hip_rt::HipInputCol> cols, const hip_rt::UnitView<StandInIr>& unit, const QmdStandIn& qmd) {
  hip_rt::PlanExtractor<StandInIr> ex(std::move(cols), [](const standin::Expr* e) { return e->column_id; });
  hip_rt::HipWorkUnit wu = ex.extract(unit, {}, false);
  return hip_rt::make_plan(wu, qmd);
}

// a device's allocations (freed at the end of the step) and copies, through the reference's GpuMgr interface
struct DeviceMem {
  DeviceMem(hip_mgr::HipMgr* m, int dev) : mgr(m), device(dev) {}
  ~DeviceMem() {
    for (int8_t* p : owned) mgr->freeDeviceMem(p);
  }
  int8_t* alloc(size_t bytes) {
    int8_t* p = mgr->allocateDeviceMem(std::max<size_t>(bytes, 256), device);
    owned.push_back(p);
    return p;
  }
  template <class T>
  int8_t* upload(const std::vector<T>& host) {
    int8_t* d = alloc(host.size() * sizeof(T));
    mgr->copyHostToDevice(d, reinterpret_cast<const int8_t*>(host.data()), host.size() * sizeof(T), device);
    return d;
  }
  template <class T>
  std::vector<T> download(const int8_t* dev_ptr, size_t n) {
    std::vector<T> host(n);
    mgr->copyDeviceToHost(reinterpret_cast<int8_t*>(host.data()), dev_ptr, n * sizeof(T), device);
    return host;
  }
  hip_mgr::HipMgr* mgr;
  int device;
  std::vector<int8_t*> owned;
};

// prepareKernelParams for the fragments one device holds (QE/QueryExecutionContext.cpp:788-964)
std::vector<int8_t*> kernel_params(DeviceMem& dev, const std::vector<std::vector<int8_t*>>& col_buffers, const std::vector<int64_t>& num_rows,
                                   const std::vector<uint64_t>& frag_offsets, const std::vector<int64_t>& init_agg_vals,
                                   const std::vector<int8_t*>& group_by_buffers) {
  const size_t nfrag = col_buffers.size();
  std::vector<int8_t*> flat;
  for (const auto& f : col_buffers) flat.insert(flat.end(), f.begin(), f.end());
  int8_t* d_flat = dev.upload(flat);
  const size_t ncols = nfrag ? col_buffers[0].size() : 0;
  std::vector<int8_t*> frag_ptrs(nfrag);
  for (size_t f = 0; f < nfrag; ++f) frag_ptrs[f] = d_flat + f * ncols * sizeof(int8_t*);
  std::vector<int8_t*> p(HDK_KP_COUNT, nullptr);
  p[HDK_KP_COL_BUFFERS] = dev.upload(frag_ptrs);
  p[HDK_KP_NUM_FRAGMENTS] = dev.upload(std::vector<uint64_t>{nfrag});
  p[HDK_KP_NUM_ROWS] = dev.upload(num_rows);
  p[HDK_KP_FRAG_ROW_OFFSETS] = dev.upload(frag_offsets);
  p[HDK_KP_MAX_MATCHED] = dev.upload(std::vector<int32_t>{0});
  p[HDK_KP_TOTAL_MATCHED] = dev.upload(std::vector<int32_t>{0, 0});
  p[HDK_KP_INIT_AGG_VALS] = dev.upload(init_agg_vals);
  p[HDK_KP_GROUPBY_BUF] = dev.upload(group_by_buffers);
  p[HDK_KP_ERROR_CODE] = dev.upload(std::vector<int32_t>{0});
  p[HDK_KP_NUM_TABLES] = dev.upload(std::vector<uint32_t>{1});
  return p;
}

int run_mg_c2(hip_mgr::HipMgr& mgr, hip_rt::HipDeviceGroup& group) {
  constexpr size_t kFragments = 4, kFragRows = 500000, kKeys = 64;
  const int G = group.size();
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
  const hdk_hip_plan plan = extract_plan({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, qmd);
  const std::vector<int64_t> init_agg_vals{0, kNullBigint, 0};
  const size_t quads = qmd.getBufferSizeBytes() / 8;
  std::vector<std::unique_ptr<DeviceMem>> mem;
  std::vector<hip_rt::PartialBuffer> parts(static_cast<size_t>(G));
  std::vector<std::vector<int8_t*>> params(static_cast<size_t>(G));
  std::vector<uint64_t> rows_on(static_cast<size_t>(G), 0);
  for (int d = 0; d < G; ++d) {
    mem.push_back(std::make_unique<DeviceMem>(&mgr, group.device(d)));
    DeviceMem& dev = *mem.back();
    std::vector<std::vector<int8_t*>> col_buffers;
    std::vector<int64_t> num_rows;
    std::vector<uint64_t> frag_offsets;
    for (size_t f = static_cast<size_t>(d); f < kFragments; f += static_cast<size_t>(G)) {  // fragment f -> device f mod G
      std::vector<int64_t> k(kFragRows), v(kFragRows);
      for (size_t r = 0; r < kFragRows; ++r) {
        const uint64_t i = f * kFragRows + r;
        k[r] = static_cast<int64_t>(mix(i) % kKeys);
        v[r] = gen_val(i);
      }
      col_buffers.push_back({dev.upload(k), dev.upload(v)});
      num_rows.push_back(kFragRows);
      frag_offsets.push_back(f * kFragRows);
      rows_on[static_cast<size_t>(d)] += kFragRows;
    }
    int8_t* out = dev.alloc(quads * 8);
    parts[static_cast<size_t>(d)].buf = reinterpret_cast<int64_t*>(out);
    parts[static_cast<size_t>(d)].gathered = reinterpret_cast<int64_t*>(dev.alloc(static_cast<size_t>(G) * quads * 8));
    parts[static_cast<size_t>(d)].dev_error = reinterpret_cast<int32_t*>(dev.upload(std::vector<int32_t>{0}));
    params[static_cast<size_t>(d)] = kernel_params(dev, col_buffers, num_rows, frag_offsets, init_agg_vals, {out});
  }
  // every device scans its fragments into its own buffer: one host thread per device (QE/Execute.cpp:2776-2788)
  group.parallel([&](int d) {
    const size_t sd = static_cast<size_t>(d);
    hdk_hip_kernel_options ko{};
    ko.total_rows = rows_on[sd];
    hip_rt::lib_check(hdk_hip_init_group_by_buffer(parts[sd].buf, reinterpret_cast<const int64_t*>(params[sd][HDK_KP_INIT_AGG_VALS]), kKeys, 1, 8,
                                                   static_cast<uint32_t>(qmd.getRowSize() / 8), 0, 1, 0, 0, group.device(d), group.stream(d)),
                      "hdk_hip_init_group_by_buffer");
    size_t ws_bytes = 0;
    hip_rt::lib_check(hdk_hip_workspace_size(&plan, &ko, group.device(d), &ws_bytes), "hdk_hip_workspace_size");
    int8_t* ws = mem[sd]->alloc(ws_bytes);
    hip_rt::lib_check(hdk_hip_launch(&plan, params[sd].data(), &ko, group.device(d), group.stream(d), ws, ws_bytes), "hdk_hip_launch");
    hip_rt::hip_check(hipStreamSynchronize(group.stream(d)), "hipStreamSynchronize");
  });
  hip_rt::all_gather_and_fold(group, plan, kKeys, quads, parts, init_agg_vals.data());
  int32_t err = 0;
  std::vector<int64_t> first;
  for (int d = 0; d < G; ++d) {
    const size_t sd = static_cast<size_t>(d);
    err |= mem[sd]->download<int32_t>(params[sd][HDK_KP_ERROR_CODE], 1)[0];
    err |= mem[sd]->download<int32_t>(reinterpret_cast<const int8_t*>(parts[sd].dev_error), 1)[0];
    const auto rows = mem[sd]->download<int64_t>(reinterpret_cast<const int8_t*>(parts[sd].buf), quads);
    if (d == 0) {
      first = rows;
    } else if (rows != first) {  // as after an all-reduce: the same merged buffer everywhere
      std::fprintf(stderr, "mg_c2: device %d holds another result than device 0\n", d);
      err |= 1;
    }
  }
  std::printf("mg_c2 error_code %d\n", err);
  const size_t rq = qmd.getRowSize() / 8;
  for (size_t e = 0; e < kKeys; ++e) {
    const int64_t* row = &first[e * rq];
    if (row[0] == std::numeric_limits<int64_t>::max()) continue;
    std::printf("mg_c2 key %" PRId64 " sum %" PRId64 " count %" PRId64 "\n", row[qmd.getColOffInBytes(0) / 8], row[qmd.getColOffInBytes(1) / 8],
                row[qmd.getColOffInBytes(2) / 8]);
  }
  return err;
}

int run_mg_c5(hip_mgr::HipMgr& mgr, hip_rt::HipDeviceGroup& group) {
  constexpr size_t kFragments = 8, kFragRows = 1100000, kKeys = 3000000, kEntries = 6000011;
  const int G = group.size();
  auto key = column(bigint(false), 0, 0), val = column(bigint(true), 0, 1);
  auto sum = agg_expr(bigint(true), AggType::kSum, val);
  hip_rt::UnitView<StandInIr> u;
  u.groupby = {key.get()};
  u.targets = {key.get(), sum.get()};
  QmdStandIn qmd;  // GroupByBaselineHash: 4-byte table key, the projected key has no slot, one 8-byte slot
  qmd.query_desc_type_ = QueryDescriptionType::GroupByBaselineHash;
  qmd.group_col_widths_ = {8};
  qmd.group_col_compact_width_ = 4;
  qmd.padded_slot_widths_ = {0, 8};
  qmd.entry_count_ = kEntries;
  hip_rt::HipInputCol kcol{0, 8, HDK_COL_INT, 1, 0, 0, static_cast<int64_t>(kKeys) - 1};
  hip_rt::HipInputCol vcol{0, 8, HDK_COL_INT, 1, 0, -1000000, 1000000};
  const hdk_hip_plan plan = extract_plan({kcol, vcol}, u, qmd);
  const std::vector<int64_t> init_agg_vals{kNullBigint};
  // an owner holds 1 / G of the keys: its table is the plan's split evenly (hdk_amd/distributed.py: owner_entry_count_for)
  const uint32_t owner_entries = static_cast<uint32_t>(std::max<size_t>((kEntries + static_cast<size_t>(G) - 1) / static_cast<size_t>(G), 1024));
  int64_t owner_quads = 0;
  hip_rt::lib_check(hdk_hip_baseline_table_quads(&plan, owner_entries, &owner_quads), "hdk_hip_baseline_table_quads");
  const size_t frags_per_dev = (kFragments + static_cast<size_t>(G) - 1) / static_cast<size_t>(G);
  hdk_hip_kernel_options ko{};
  ko.total_rows = frags_per_dev * kFragRows;  // the largest rank's rows: the same bound on every rank
  hdk_hip_exchange_shape shape;
  hip_rt::lib_check(hdk_hip_exchange_shape_for(&plan, &ko, G, owner_entries, group.device(0), &shape), "hdk_hip_exchange_shape_for");
  std::vector<std::unique_ptr<DeviceMem>> mem;
  std::vector<hip_rt::TupleExchangeRank> ranks(static_cast<size_t>(G));
  std::vector<std::vector<int8_t*>> scan_params(static_cast<size_t>(G)), owner_params(static_cast<size_t>(G));
  std::vector<int8_t*> tables(static_cast<size_t>(G));
  for (int d = 0; d < G; ++d) {
    const size_t sd = static_cast<size_t>(d);
    mem.push_back(std::make_unique<DeviceMem>(&mgr, group.device(d)));
    DeviceMem& dev = *mem.back();
    std::vector<std::vector<int8_t*>> col_buffers;
    std::vector<int64_t> num_rows;
    std::vector<uint64_t> frag_offsets;
    for (size_t f = sd; f < kFragments; f += static_cast<size_t>(G)) {
      std::vector<int64_t> k(kFragRows), v(kFragRows);
      for (size_t r = 0; r < kFragRows; ++r) {
        const uint64_t i = f * kFragRows + r;
        k[r] = static_cast<int64_t>(mix(i + (1ull << 47)) % kKeys);
        v[r] = static_cast<int64_t>(mix(i + (1ull << 48)) % 2000001) - 1000000;
      }
      col_buffers.push_back({dev.upload(k), dev.upload(v)});
      num_rows.push_back(kFragRows);
      frag_offsets.push_back(f * kFragRows);
    }
    tables[sd] = dev.alloc(static_cast<size_t>(owner_quads) * 8);
    scan_params[sd] = kernel_params(dev, col_buffers, num_rows, frag_offsets, init_agg_vals, {tables[sd]});
    owner_params[sd] = scan_params[sd];  // (GROUPBY_BUF[0] = the owner's table, INIT_AGG_VALS, ERROR_CODE: the same block serves)
    ranks[sd].scan_params = scan_params[sd].data();
    ranks[sd].owner_params = owner_params[sd].data();
    ranks[sd].send = dev.alloc(static_cast<size_t>(G) * shape.segment_bytes);
    ranks[sd].recv = dev.alloc(static_cast<size_t>(G) * shape.segment_bytes);
    ranks[sd].ws_scatter = dev.alloc(shape.scatter_workspace_bytes);
    ranks[sd].ws_aggregate = dev.alloc(shape.aggregate_workspace_bytes);
  }
  hip_rt::exchange_tuples(group, plan, ko, shape, ranks);
  int32_t err = 0;
  uint64_t groups = 0, sum_of_sums = 0, mixed = 0;
  for (int d = 0; d < G; ++d) {
    const size_t sd = static_cast<size_t>(d);
    err |= mem[sd]->download<int32_t>(owner_params[sd][HDK_KP_ERROR_CODE], 1)[0];
    const auto rows = mem[sd]->download<int64_t>(tables[sd], 2 * static_cast<size_t>(owner_entries));
    for (size_t e = 0; e < owner_entries; ++e) {
      const int32_t k = static_cast<int32_t>(rows[2 * e]);  // [key int32 | padding][slot]
      if (k == std::numeric_limits<int32_t>::max()) continue;
      ++groups;
      sum_of_sums += static_cast<uint64_t>(rows[2 * e + 1]);
      mixed ^= mix(static_cast<uint64_t>(k) * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(rows[2 * e + 1]));  // (order-free)
    }
  }
  std::printf("mg_c5 error_code %d\n", err);
  std::printf("mg_c5 groups %" PRIu64 " sum_of_sums %" PRId64 " checksum %" PRIu64 "\n", groups, static_cast<int64_t>(sum_of_sums), mixed);
  std::fprintf(stderr, "mg_c5: %d owner(s), %u entries each, %u-byte tuples, %" PRIu64 " bytes per segment\n", G, owner_entries, shape.tuple_bytes,
               static_cast<uint64_t>(shape.segment_bytes));
  return err;
}

}  // namespace

int main() {
  try {
    hip_mgr::HipMgr mgr(-1, 0);  // every device of the node
    std::vector<int> devices;
    for (int d = 0; d < mgr.getDeviceCount(); ++d) devices.push_back(d);
    std::fprintf(stderr, "multi_device: %d device(s)\n", static_cast<int>(devices.size()));
    hip_rt::HipDeviceGroup group(devices);
    int rc = run_mg_c2(mgr, group);
    rc |= run_mg_c5(mgr, group);
    return rc ? 1 : 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "multi_device failed: %s\n", e.what());
    return 2;
  }
}
