// Micro-benchmark: what does a read-only kernel reach with C2's access SHAPE -- two 8-byte columns read in lockstep
// (16 B per lane and load, 4 + 4 loads in flight), each column in 32 separately allocated 256 MB fragments -- against one
// stream of the same total bytes?  Separates "the kernel's row work" from "the access pattern" as the reason why
// hdk_scan_agg_direct (6.3 TB/s) stays below the one-stream read peak (7.1 TB/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

struct Frags { const f4* a[32]; const f4* b[32]; };

// MODE 0: one stream (a only, 8 loads); 1: two streams in lockstep (4 + 4); 2: two streams, b read half a fragment ahead
template <int MODE>
__global__ __launch_bounds__(256) void k_read(Frags fr, int nfrag, size_t n_per_frag, float* sink) {
  constexpr size_t kTile = 256 * (MODE == 0 ? 8 : 4);
  const size_t tiles_per_frag = n_per_frag / kTile;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t t = blockIdx.x; t < tiles_per_frag * nfrag; t += gridDim.x) {
    const int f = (int)(t / tiles_per_frag);
    const size_t tt = t % tiles_per_frag;
    const f4* pa = fr.a[f] + tt * kTile + threadIdx.x;
    f4 v[8];
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(pa + u * 256);
    } else {
      const size_t tb = MODE == 2 ? (tt + tiles_per_frag / 2) % tiles_per_frag : tt;
      const f4* pb = fr.b[f] + tb * kTile + threadIdx.x;
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[2 * u] = __builtin_nontemporal_load(pa + u * 256); v[2 * u + 1] = __builtin_nontemporal_load(pb + u * 256); }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) *sink = acc.x;
}

int main() {
  const int nfrag = 32;
  const size_t frag_bytes = 256ull << 20, n_per_frag = frag_bytes / 16;
  Frags fr;
  for (int f = 0; f < nfrag; ++f) { void *a, *b; CK(hipMalloc(&a, frag_bytes)); CK(hipMalloc(&b, frag_bytes)); CK(hipMemset(a, 1, frag_bytes)); CK(hipMemset(b, 1, frag_bytes)); fr.a[f] = (const f4*)a; fr.b[f] = (const f4*)b; }
  float* sink; CK(hipMalloc(&sink, 4));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int per_cu : {2, 4, 8}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      const int nf = mode == 0 ? nfrag : nfrag;  // mode 0 reads a[] only: 8 GB; modes 1, 2 read a[] and b[]: 16 GB
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) k_read<0><<<prop.multiProcessorCount * per_cu, 256>>>(fr, nf, n_per_frag, sink);
        if (mode == 1) k_read<1><<<prop.multiProcessorCount * per_cu, 256>>>(fr, nf, n_per_frag, sink);
        if (mode == 2) k_read<2><<<prop.multiProcessorCount * per_cu, 256>>>(fr, nf, n_per_frag, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      const double bytes = (mode == 0 ? 1.0 : 2.0) * nfrag * frag_bytes;
      printf("%d blocks/CU  %-34s %.3f ms  %.0f GB/s\n", per_cu, mode == 0 ? "one stream, 8 loads in flight" : mode == 1 ? "two streams in lockstep, 4 + 4" : "two streams, second half a fragment ahead", best, bytes / best / 1e6);
    }
  }
  return 0;
}
