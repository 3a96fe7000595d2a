// copy_probe.hip -- what the MI355X memory system gives a streaming kernel, by access geometry.  Not part of the product.
//   copy_probe [GiB=4] [reps=5]
// Kernels move 16 bytes per lane and instruction.  MODE 0: grid-stride (consecutive workgroups interleave at 4 KB), MODE 1:
// one contiguous span per wave (the geometry of the product's streaming kernels: a wave walks its channel chunk).
// RW: 0 copy (1 read : 1 write), 1 read-only (sum kept live), 2 write-only, 3 four reads : 1 write (the CIC decimator's mix).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int RW, int U, bool NT>
__global__ void __launch_bounds__(256) probe(const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec, long span_vec, int flag) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  const long n_thr = (long)gridDim.x * 256;
  v4i acc = {0, 0, 0, 0};
  if (MODE == 0) {
    for (long i = tid; i < n_vec; i += n_thr * U) {
      v4i v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long j = i + u * n_thr;
        if (RW != 2) { v[u] = j < n_vec ? (NT ? __builtin_nontemporal_load(x + j) : x[j]) : (v4i){0, 0, 0, 0}; } else { v[u] = (v4i){(int)j, 1, 2, 3}; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long j = i + u * n_thr;
        if (RW == 1) { acc += v[u]; }
        else if (RW == 3) {   // one 16-byte store per four loads, lane-contiguous
          acc += v[u];
          const long o = ((j - tid) / n_thr / 4) * n_thr + tid;
          if ((u & 3) == 3 && j < n_vec) { if (NT) { __builtin_nontemporal_store(acc, y + o); } else { y[o] = acc; } }
        }
        else if (j < n_vec) { if (NT) { __builtin_nontemporal_store(v[u], y + j); } else { y[j] = v[u]; } }
      }
    }
  } else {
    const long wave = tid >> 6;
    const int lane = threadIdx.x & 63;
    const long b = wave * span_vec, e = b + span_vec < n_vec ? b + span_vec : n_vec;
    for (long i = b + lane; i < e; i += 64 * U) {
      v4i v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long j = i + 64 * u;
        if (RW != 2) { v[u] = j < e ? (NT ? __builtin_nontemporal_load(x + j) : x[j]) : (v4i){0, 0, 0, 0}; } else { v[u] = (v4i){(int)j, 1, 2, 3}; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long j = i + 64 * u;
        if (RW == 1) { acc += v[u]; }
        else if (RW == 3) {
          acc += v[u];
          const long o = b / 4 + 64 * ((j - b - lane) / 64 / 4) + lane;
          if ((u & 3) == 3 && j < e) { if (NT) { __builtin_nontemporal_store(acc, y + o); } else { y[o] = acc; } }
        }
        else if (j < e) { if (NT) { __builtin_nontemporal_store(v[u], y + j); } else { y[j] = v[u]; } }
      }
    }
  }
  if (RW == 1 && flag == acc.x + acc.y + acc.z + acc.w + 12345) { y[tid] = acc; }
}

// LDS-DMA transport (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPR on the way in): wave-span geometry, a
// wave-private double buffer of U KB per half; the batch two ahead is issued before the current one is read back with
// ds_read_b128.  The asm loads are invisible to hipcc's waitcnt bookkeeping, so the waits are counted by hand: behind the
// newest U loads (read-only) or the newest U loads + the U stores issued in front of them (copy) everything has landed.
template <bool NT>
__device__ __forceinline__ void glds16(const v4i *gsrc, unsigned lds_dst) {
  unsigned keep;
  if (NT) { asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory"); }
  else { asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory"); }
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int RW, int U, bool NT>
__global__ void __launch_bounds__(256) probe_glds(const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec, long span_vec, int flag) {
  __shared__ __attribute__((aligned(16))) v4i sm[4][2][U][64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + wave;
  const long b = gw * span_vec, e = b + span_vec < n_vec ? b + span_vec : n_vec;   // spans are whole batches (host)
  v4i acc = {0, 0, 0, 0};
  if (b < e) {
    const unsigned base = (unsigned)(uintptr_t)&sm[wave][0][0][0];   // LDS byte address, wave-uniform
    const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int u = 0; u < U; u++) { glds16<NT>(x + b + 64 * u + lane, sbase + 1024 * u); }
    int buf = 0;
    for (long i = b; i < e; i += 64 * U, buf ^= 1) {
      const long nx = i + 64 * U < e ? i + 64 * U : i;   // past the end: the last batch again (branch-free body)
      const unsigned nb = sbase + 1024 * U * (buf ^ 1);
#pragma unroll
      for (int u = 0; u < U; u++) { glds16<NT>(x + nx + 64 * u + lane, nb + 1024 * u); }
      if (RW == 0) { wait_vm<2 * U>(); } else { wait_vm<U>(); }
      v4i v[U];
#pragma unroll
      for (int u = 0; u < U; u++) { v[u] = sm[wave][buf][u][lane]; }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (RW == 1) { acc += v[u]; }
        else { y[i + 64 * u + lane] = v[u]; }
      }
      asm volatile("" ::: "memory");
    }
    wait_vm<0>();
  }
  if (RW == 1 && flag == acc.x + acc.y + acc.z + acc.w + 12345) { y[(long)blockIdx.x * 256 + threadIdx.x] = acc; }
}

struct Ctx { v4i *x, *y; long n_vec; int reps; };

template <int MODE, int RW, int U, bool NT>
static void run(const Ctx &c, long blocks, long span_kb) {
  const long span_vec = span_kb * 64;
  long nb = blocks;
  if (MODE == 1) { const long waves = (c.n_vec + span_vec - 1) / span_vec; nb = (waves + 3) / 4; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) { hipLaunchKernelGGL((probe<MODE, RW, U, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.x, c.y, c.n_vec, span_vec, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((probe<MODE, RW, U, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.x, c.y, c.n_vec, span_vec, 0); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  const double bytes = (double)c.n_vec * 16 * (RW == 0 ? 2.0 : (RW == 3 ? 1.25 : 1.0));
  static const char *rw[] = {"copy 1:1", "read-only", "write-only", "read 4:1 write"};
  printf("%-15s %-11s U=%2d nt=%d blocks=%7ld span=%5ld KB  %8.3f ms  %6.2f TB/s\n", rw[RW], MODE == 0 ? "grid-stride" : "wave-span", U, (int)NT, nb, MODE == 1 ? span_kb : 0, ms,
         bytes / ms / 1e9);
}

template <int RW, int U, bool NT>
static void run_glds(const Ctx &c, long span_kb) {
  const long span_vec = span_kb * 64;
  const long waves = (c.n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) { hipLaunchKernelGGL((probe_glds<RW, U, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.x, c.y, c.n_vec, span_vec, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((probe_glds<RW, U, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.x, c.y, c.n_vec, span_vec, 0); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  const double bytes = (double)c.n_vec * 16 * (RW == 0 ? 2.0 : 1.0);
  printf("%-15s %-11s U=%2d nt=%d blocks=%7ld span=%5ld KB  %8.3f ms  %6.2f TB/s\n", RW == 0 ? "copy 1:1" : "read-only", "lds-dma", U, (int)NT, nb, span_kb, ms, bytes / ms / 1e9);
}

int main(int argc, char **argv) {
  Ctx c;
  const long gib = argc > 1 ? atol(argv[1]) : 4;
  c.reps = argc > 2 ? atoi(argv[2]) : 5;
  c.n_vec = gib * (1L << 30) / 16;
  CK(hipMalloc((void **)&c.x, c.n_vec * 16));
  CK(hipMalloc((void **)&c.y, c.n_vec * 16));
  CK(hipMemset(c.x, 1, c.n_vec * 16));
  CK(hipMemset(c.y, 2, c.n_vec * 16));
  printf("# copy_probe: %ld GiB per buffer\n", gib);
  // grid-stride, persistent-size grids
  run<0, 0, 4, false>(c, 256 * 8, 0); run<0, 0, 8, false>(c, 256 * 8, 0); run<0, 0, 4, false>(c, 256 * 32, 0); run<0, 0, 8, true>(c, 256 * 8, 0);
  run<0, 1, 8, false>(c, 256 * 8, 0); run<0, 1, 8, true>(c, 256 * 8, 0); run<0, 2, 4, false>(c, 256 * 8, 0); run<0, 2, 4, true>(c, 256 * 8, 0);
  run<0, 3, 8, false>(c, 256 * 8, 0); run<0, 3, 8, true>(c, 256 * 8, 0);
  // one contiguous span per wave
  run<1, 0, 4, false>(c, 0, 64); run<1, 0, 8, false>(c, 0, 64); run<1, 0, 8, false>(c, 0, 16); run<1, 0, 8, false>(c, 0, 256); run<1, 0, 8, true>(c, 0, 64);
  run<1, 1, 8, false>(c, 0, 64); run<1, 1, 8, true>(c, 0, 64); run<1, 2, 4, false>(c, 0, 64); run<1, 2, 4, true>(c, 0, 64);
  run<1, 3, 8, false>(c, 0, 64); run<1, 3, 8, true>(c, 0, 64); run<1, 3, 8, false>(c, 0, 256);
  // LDS-DMA transport, wave-span geometry
  run_glds<1, 4, false>(c, 64); run_glds<1, 4, true>(c, 64); run_glds<1, 8, false>(c, 64); run_glds<1, 8, true>(c, 64); run_glds<1, 4, true>(c, 16);
  run_glds<0, 4, false>(c, 64); run_glds<0, 4, true>(c, 64); run_glds<0, 8, false>(c, 64); run_glds<0, 4, false>(c, 16); run_glds<0, 4, true>(c, 16);
  // register loads again, behind the LDS-DMA rows (clock / thermal drift check)
  run<1, 1, 8, true>(c, 0, 64); run<1, 0, 8, false>(c, 0, 16);
  return 0;
}
