// Micro-benchmark of HBM access patterns for a 2-read + 1-write row-major stream (f64, 4097 x 4097, pitch 4160):
// how wide / how bursty must a wave's row accesses be to reach the rate of a contiguous tile?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/spb tools/stream_pattern_bench.hip && /tmp/spb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct alignas(16) P { double v[2]; };

// pattern S: one wave = one 1-KB column strip, walks `rows` rows, loads `B` rows of both inputs at a time
template <int B>
__global__ __launch_bounds__(256) void stream_k(const double* a, const double* b, double* c, int nx, int ld, int nstrips, int rows) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int strip = wave % nstrips, chunk = wave / nstrips;
  const int j = strip * 128 + lane * 2;
  if (j >= ld) return;
  const int r0 = chunk * rows, r1 = min(r0 + rows, nx);
  if (r0 >= nx) return;
  for (int r = r0; r < r1; r += B) {
    P x[B], y[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const size_t off = (size_t)min(r + k, nx - 1) * ld + j;
      x[k] = *reinterpret_cast<const P*>(a + off);
      y[k] = *reinterpret_cast<const P*>(b + off);
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (r + k < r1) {
        P o; o.v[0] = x[k].v[0] + y[k].v[0]; o.v[1] = x[k].v[1] + y[k].v[1];
        *reinterpret_cast<P*>(c + (size_t)(r + k) * ld + j) = o;
      }
    }
  }
}

// pattern T: one workgroup = a TI x 64-column tile (512 B rows), all loads then all stores
template <int TI>
__global__ __launch_bounds__(256) void tile_k(const double* a, const double* b, double* c, int nx, int ld, int tiles_j) {
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
  const int cg = threadIdx.x % 32, rg = threadIdx.x / 32;     // 32 threads x 16 B = 512 B per row, 8 row groups
  const int j = tj * 64 + cg * 2;
  if (j >= ld) return;
  constexpr int RPT = TI / 8;
  P x[RPT], y[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = min(ti * TI + rg * RPT + k, nx - 1);
    x[k] = *reinterpret_cast<const P*>(a + (size_t)r * ld + j);
    y[k] = *reinterpret_cast<const P*>(b + (size_t)r * ld + j);
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = ti * TI + rg * RPT + k;
    if (r < nx) { P o; o.v[0] = x[k].v[0] + y[k].v[0]; o.v[1] = x[k].v[1] + y[k].v[1]; *reinterpret_cast<P*>(c + (size_t)r * ld + j) = o; }
  }
}

template <typename F> float time_it(F f, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps * 1e3f;
}

int main() {
  const int nx = 4097, ld = 4160;
  const size_t bytes = (size_t)nx * ld * 8;
  double *a, *b, *c; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
  const double mb = 3.0 * nx * ld * 8 / 1e6;
  const int nstrips = (ld + 127) / 128;
#define RUN_S(B, ROWS) { const int chunks = (nx + ROWS - 1) / ROWS; const int waves = nstrips * chunks; \
    float us = time_it([&] { hipLaunchKernelGGL(stream_k<B>, dim3((waves + 3) / 4), dim3(256), 0, 0, a, b, c, nx, ld, nstrips, ROWS); }); \
    printf("stream burst %2d rows/chunk %4d waves %5d : %7.1f us  %6.0f GB/s\n", B, ROWS, waves, us, mb / us * 1e3 / 1e3); }
  RUN_S(1, 128) RUN_S(2, 128) RUN_S(4, 128) RUN_S(8, 128) RUN_S(16, 128) RUN_S(32, 128)
  RUN_S(4, 64) RUN_S(8, 64) RUN_S(16, 64) RUN_S(8, 32) RUN_S(16, 32) RUN_S(32, 32) RUN_S(8, 256) RUN_S(16, 256) RUN_S(8, 512) RUN_S(8, 1024)
#define RUN_T(TI) { const int tiles_j = (ld + 63) / 64, tiles_i = (nx + TI - 1) / TI; \
    float us = time_it([&] { hipLaunchKernelGGL(tile_k<TI>, dim3(tiles_i * tiles_j), dim3(256), 0, 0, a, b, c, nx, ld, tiles_j); }); \
    printf("tile %2d x 512 B                          : %7.1f us  %6.0f GB/s\n", TI, us, mb / us * 1e3 / 1e3); }
  RUN_T(8) RUN_T(16) RUN_T(32) RUN_T(64)
  return 0;
}
