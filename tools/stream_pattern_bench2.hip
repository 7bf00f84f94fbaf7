// Which part of the register-blocked tile access pattern costs HBM efficiency?  2 reads + 1 write (f32, 4097 x 4224
// pitch), every launch on a fresh buffer set (rotation over > 768 MiB: nothing is found in the 256 MiB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/spb2 tools/stream_pattern_bench2.hip && /tmp/spb2
// Variants (one workgroup = W waves x RPT rows x 1 KB, all loads first, then all stores):
//   base      : tile = region, rows start on 128-byte lines
//   shift64   : every row access starts 64 bytes into a line (9 lines per 1 KB instead of 8)
//   vhalo     : the region has H extra rows above and below (loaded, not stored)
//   lmask     : only lanes 4..59 store
//   all       : shift64 + vhalo + lmask + tile pitch 896 B = the register-blocked leg's footprint
//   nt        : `all` with non-temporal stores / non-temporal rhs loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct alignas(16) P { float v[4]; };

template <int W, int RPT, int H, bool SHIFT, bool LMASK, bool PITCH896, int NT>
__global__ __launch_bounds__(W * 64) void tile_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, int nx,
                                                  int ny, int ld, int tiles_j) {
  constexpr int RI = W * RPT, TI = RI - 2 * H;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j0 = tj * (PITCH896 ? 224 : 256) - (SHIFT ? 16 : 0);
  const int j = j0 + lane * 4;
  const int i0 = ti * TI - H + w * RPT;
  P x[RPT], y[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = i0 + k;
    if (r >= 0 && r < nx && j >= 0 && j < ny) {
      x[k] = *reinterpret_cast<const P*>(a + (size_t)r * ld + j);
      if (NT & 2) { const P* p = reinterpret_cast<const P*>(b + (size_t)r * ld + j); y[k].v[0] = __builtin_nontemporal_load(&p->v[0]); y[k].v[1] = __builtin_nontemporal_load(&p->v[1]); y[k].v[2] = __builtin_nontemporal_load(&p->v[2]); y[k].v[3] = __builtin_nontemporal_load(&p->v[3]); }
      else y[k] = *reinterpret_cast<const P*>(b + (size_t)r * ld + j);
    } else { x[k] = P{{0, 0, 0, 0}}; y[k] = P{{0, 0, 0, 0}}; }
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int rr = w * RPT + k, r = i0 + k;
    const bool st = rr >= H && rr < RI - H && r >= 0 && r < nx && j >= 0 && j < ny && (!LMASK || (lane >= 4 && lane < 60));
    if (st) {
      P o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = x[k].v[e] + y[k].v[e];
      P* q = reinterpret_cast<P*>(c + (size_t)r * ld + j);
      if (NT & 1) { __builtin_nontemporal_store(o.v[0], &q->v[0]); __builtin_nontemporal_store(o.v[1], &q->v[1]); __builtin_nontemporal_store(o.v[2], &q->v[2]); __builtin_nontemporal_store(o.v[3], &q->v[3]); }
      else *q = o;
    }
  }
}

int main() {
  const int nx = 4097, ny = 4097, ld = 4224;
  const size_t bytes = (size_t)nx * ld * 4;
  const int NSETS = 5;                                     // 5 x 3 x 69 MB = 1 GB
  std::vector<float*> bufs(3 * NSETS);
  for (auto& p : bufs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  hipDeviceSynchronize();
  const double mb = 3.0 * nx * ny * 4 / 1e6;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < NSETS; ++i) launch(bufs[3 * i], bufs[3 * i + 1], bufs[3 * i + 2]);
    hipDeviceSynchronize();
    const int reps = 30;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) { const int s = i % NSETS; launch(bufs[3 * s], bufs[3 * s + 1], bufs[3 * s + 2]); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3;
    printf("%-34s %7.1f us  %6.0f GB/s (of the 3 x 67 MB the arrays hold)\n", name, us, mb / us * 1e3 / 1e3);
  };
#define L(W, RPT, H, SHIFT, LMASK, P896, NT)                                                                          \
  [&](const float* a, const float* b, float* c) {                                                                    \
    constexpr int TI = W * RPT - 2 * H;                                                                              \
    const int tiles_j = (ny + (P896 ? 224 : 256) - 1) / (P896 ? 224 : 256) + (SHIFT ? 0 : 0), tiles_i = (nx + TI - 1) / TI; \
    hipLaunchKernelGGL((tile_k<W, RPT, H, SHIFT, LMASK, P896, NT>), dim3(tiles_i * tiles_j), dim3(W * 64), 0, 0, a, b, c, nx, ny, ld, tiles_j); \
  }
  run("base 4x8", L(4, 8, 0, false, false, false, 0));
  run("base 8x8", L(8, 8, 0, false, false, false, 0));
  run("base 4x16", L(4, 16, 0, false, false, false, 0));
  run("base 2x8", L(2, 8, 0, false, false, false, 0));
  run("base 4x4", L(4, 4, 0, false, false, false, 0));
  run("shift64 4x8", L(4, 8, 0, true, false, false, 0));
  run("vhalo1 4x8", L(4, 8, 1, false, false, false, 0));
  run("vhalo3 4x8", L(4, 8, 3, false, false, false, 0));
  run("lmask 4x8 (pitch 1 KB)", L(4, 8, 0, false, true, false, 0));
  run("lmask + pitch 896 + shift 4x8", L(4, 8, 0, true, true, true, 0));
  run("all (vhalo3) 4x8", L(4, 8, 3, true, true, true, 0));
  run("all (vhalo1) 4x8", L(4, 8, 1, true, true, true, 0));
  run("all (vhalo1) 8x8", L(8, 8, 1, true, true, true, 0));
  run("base 4x8 nt stores", L(4, 8, 0, false, false, false, 1));
  run("base 4x8 nt stores + nt b loads", L(4, 8, 0, false, false, false, 3));
  run("all (vhalo1) 4x8 nt stores", L(4, 8, 1, true, true, true, 1));
  run("all (vhalo1) 4x8 nt st + nt b", L(4, 8, 1, true, true, true, 3));
  return 0;
}
