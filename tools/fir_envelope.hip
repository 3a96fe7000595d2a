// fir_envelope.hip -- round 3: the envelope of "stream 2 B in + 2 B out per sample AND issue NM int8 32x32x32 MFMAs per 1024 samples"
// in the BEST streaming geometry this part offers (tools/copy_probe2.hip: 4-wave workgroups, 16 - 32 KB spans walked in memory order,
// batches of U loads followed by U stores, non-temporal both ways: 5.8 - 6.1 TB/s), with operands that have the statistics of the
// product kernel's: A = the 255-tap windowed-sinc Toeplitz fragments of bench.py's config-2 set (low-byte plane: dense random-looking
// bytes; high-byte plane: small values inside the band, zero outside), B = the loaded samples themselves (uniform bytes, as the
// byte planes of the uniform int16 stimulus are).  No byte-plane split, no LDS, no epilogue arithmetic: everything the product
// kernel does beyond this costs extra.  Not part of the product.
//
//   fir_envelope [reps=10] [stream | mfma | both: one row only, for power sampling]
//
// Each row: wall ms per launch of 1024 ch x 2^20 samples (4.29 GB algorithmic), the rate, MFMA rate, and what fraction of
// 8 TB/s that is.  The 70 % target is 0.767 ms.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MK = MFMAs per 1 KB wave-load (a 1024-sample step is two loads: MK 13 = the 26 MFMAs of config 2, 18 = the dense 36);
// HI = how many of them take a high-byte-plane A fragment (config 2: 8 of 26 -> 4 of 13).
template <int MK, int HI, int U, int OCC, bool NT>
__global__ void __launch_bounds__(256, OCC) env(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec,
                                               long span_vec) {
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i Al[4], Ah[2];
#pragma unroll
  for (int i = 0; i < 4; i++) { Al[i] = frag[i * 64 + lane]; }
#pragma unroll
  for (int i = 0; i < 2; i++) { Ah[i] = frag[(4 + i) * 64 + lane]; }
  v16i acc[4] = {{0}, {0}, {0}, {0}};
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  for (long i = s + lane; i < e; i += 64 * U) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = NT ? __builtin_nontemporal_load(x + i + 64 * u) : x[i + 64 * u]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      // the stored words come from the accumulators as the PREVIOUS load left them (one load of pipelining, as the product's
      // epilogue runs one step behind its MFMAs), so the store does not wait for this load's MFMAs
      v4i o = v[u];
      if (MK > 0) { o = (v4i){acc[0][u & 15], acc[1][(u + 1) & 15], acc[2][(u + 2) & 15], acc[3][(u + 3) & 15]}; }
      asm volatile("" : "+v"(o));                 // the four words are read HERE (else the scheduler sinks the reads below the MFMAs
      __builtin_amdgcn_sched_barrier(0);          // and keeps whole copies of the accumulators: 200+ spilled VGPRs)
#pragma unroll
      for (int m = 0; m < MK; m++) {
        const v4i a = m < HI ? Ah[m & 1] : Al[m & 3];
        acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, v[u], acc[m & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NT) { __builtin_nontemporal_store(o, y + i + 64 * u); } else { y[i + 64 * u] = o; }
    }
  }
}

// the same MFMA work with no memory traffic in the loop (B operands loaded once): what the matrix pipe alone costs
template <int MK, int HI, int OCC>
__global__ void __launch_bounds__(256, OCC) mfma_only(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec,
                                                     long span_vec, int zero) {
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i Al[4], Ah[2], v[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { Al[i] = frag[i * 64 + lane]; }
#pragma unroll
  for (int i = 0; i < 2; i++) { Ah[i] = frag[(4 + i) * 64 + lane]; }
#pragma unroll
  for (int u = 0; u < 8; u++) { v[u] = zero ? (v4i){0, 0, 0, 0} : x[wave * 512 + 64 * u + lane]; }
  v16i acc[4] = {{0}, {0}, {0}, {0}};
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  for (long i = s; i < e; i += 64 * 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int m = 0; m < MK; m++) {
        const v4i a = m < HI ? Ah[m & 1] : Al[m & 3];
        acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, v[u], acc[m & 3], 0, 0, 0);
      }
    }
  }
  y[wave * 64 + lane] = (v4i){acc[0][0], acc[1][1], acc[2][2], acc[3][3]};
}

// the same int8 MAC count per load on the other int8 shape (v_mfma_i32_16x16x64_i8: half the MACs of a 32x32x32, so 2 MK per load):
// is one shape cheaper per MAC than the other?
template <int MK, int U, int OCC, bool STREAM>
__global__ void __launch_bounds__(256, OCC) env16(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec,
                                                 long span_vec) {
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i A[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { A[i] = frag[i * 64 + lane]; }
  v4i acc[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  v4i v[U];
  if (!STREAM) {
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = x[wave * 64 * U + 64 * u + lane]; }
  }
  for (long i = s + lane; i < e; i += 64 * U) {
    if (STREAM) {
#pragma unroll
      for (int u = 0; u < U; u++) { v[u] = __builtin_nontemporal_load(x + i + 64 * u); }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4i o = (v4i){acc[0][u & 3], acc[1][(u + 1) & 3], acc[2][(u + 2) & 3], acc[3][(u + 3) & 3]};
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 2 * MK; m++) { acc[m & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[m & 3], v[u], acc[m & 7], 0, 0, 0); }
      __builtin_amdgcn_sched_barrier(0);
      if (STREAM) { __builtin_nontemporal_store(o, y + i + 64 * u); }
    }
  }
  if (!STREAM) { y[wave * 64 + lane] = (v4i){acc[0][0], acc[1][1], acc[2][2], acc[3][3]}; }
}

// Both shapes WITH the data operand going through LDS, as any Toeplitz kernel must fetch it (the K-blocks of a step are shifted
// copies of each other; LDS addressing is the shifter): every load is staged once (ds_write_b128) and RPL conflict-free ds_read_b128
// per load feed the MFMAs.  The 32x32x32 formulation of config 2 reads 18 fragments per 26 MFMAs per 1024 samples (RPL 9 of MK 13);
// the 16x16x64 formulation (16 outputs x 16 blocks, K = 64: 5 K-blocks, 2 of them in the high-byte band) issues 56 instructions
// = 28 equivalents and reads 40 fragments per 1024 samples (RPL 20 of 28 instructions per load).
template <bool S16, int NI, int RPL, int U, int OCC>
__global__ void __launch_bounds__(256, OCC) env_lds(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec,
                                                   long span_vec) {
  __shared__ __attribute__((aligned(16))) v4i sm[4][2][64 + 32];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + wv;
  v4i A[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { A[i] = frag[i * 64 + lane]; }
  v16i acc[S16 ? 1 : 4];
  v4i acc16[S16 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (S16 ? 1 : 4); i++) { acc[i] = (v16i){0}; }
#pragma unroll
  for (int i = 0; i < (S16 ? 8 : 1); i++) { acc16[i] = (v4i){0, 0, 0, 0}; }
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  for (long i = s + lane; i < e; i += 64 * U) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = __builtin_nontemporal_load(x + i + 64 * u); }
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4i o = S16 ? (v4i){acc16[0][u & 3], acc16[1][(u + 1) & 3], acc16[2][(u + 2) & 3], acc16[3][(u + 3) & 3]}
                  : (v4i){acc[0][u & 15], acc[1][(u + 1) & 15], acc[2][(u + 2) & 15], acc[3][(u + 3) & 15]};
      asm volatile("" : "+v"(o));
      sm[wv][u & 1][lane] = v[u];                     // stage the load (1 KB)
      v4i B[RPL];
#pragma unroll
      for (int r = 0; r < RPL; r++) { B[r] = sm[wv][u & 1][lane + (r & 31)]; }   // shifted windows of the staged data (garbage tail: timing only)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NI; m++) {
        if (S16) { acc16[m & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[m % 6], B[m % RPL], acc16[m & 7], 0, 0, 0); }
        else { acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[m % 6], B[m % RPL], acc[m & 3], 0, 0, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_nontemporal_store(o, y + i + 64 * u);
    }
  }
}

struct Ctx { v4i *frag, *x, *y; long n_vec; int reps; };

template <bool S16, int NI, int RPL, int U, int OCC>
static void run_lds(const Ctx &c, long span_kb) {
  const long span_vec = span_kb * 64, waves = (c.n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; w++) { hipLaunchKernelGGL((env_lds<S16, NI, RPL, U, OCC>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((env_lds<S16, NI, RPL, U, OCC>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  printf("stream + MFMA + LDS reads  %s: %2d instructions (%4.1f equivalents of 32x32x32) and %2d fragment reads per 1024 samples  %7.3f ms  frac of 8 TB/s %.3f\n",
         S16 ? "16x16x64" : "32x32x32", 2 * NI, S16 ? (double)NI : 2.0 * NI, 2 * RPL, ms, 2.0 * c.n_vec * 16 / ms / 1e9 / 8000.0);
  fflush(stdout);
}

template <int MK, int U, int OCC, bool STREAM>
static void run16(const Ctx &c, long span_kb) {
  const long span_vec = span_kb * 64, waves = (c.n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; w++) { hipLaunchKernelGGL((env16<MK, U, OCC, STREAM>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((env16<MK, U, OCC, STREAM>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  printf("%-26s 16x16x64 shape, %2d MFMA-equivalents of 32x32x32 per 1024 samples (%d instructions), %s  %7.3f ms\n", STREAM ? "stream + MFMA" : "MFMA only", 2 * MK, 4 * MK,
         STREAM ? "stream" : "no memory traffic", ms);
  fflush(stdout);
}

template <int MK, int HI, int OCC>
static void run_mfma(const Ctx &c, int zero) {
  const long span_vec = 32 * 64, waves = (c.n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; w++) { hipLaunchKernelGGL((mfma_only<MK, HI, OCC>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec, zero); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((mfma_only<MK, HI, OCC>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec, zero); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  const double mfma = (double)MK * (c.n_vec / 64);
  printf("%-26s MFMA/1024 samples=%2d (high-plane %2d) launch_bounds(256,%d) B operand %-7s  %7.3f ms  %6.0f TOP/s (%.0f %% of 5 POP/s)\n", "MFMA only", 2 * MK, 2 * HI, OCC,
         zero ? "zero" : "random", ms, mfma * 65536.0 / ms / 1e9, mfma * 65536.0 / ms / 1e9 / 50.0);
  fflush(stdout);
}

template <int MK, int HI, int U, int OCC, bool NT>
static void run(const Ctx &c, long span_kb, const char *tag) {
  const long span_vec = span_kb * 64, waves = (c.n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; w++) { hipLaunchKernelGGL((env<MK, HI, U, OCC, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((env<MK, HI, U, OCC, NT>), dim3((unsigned)nb), dim3(256), 0, 0, c.frag, c.x, c.y, c.n_vec, span_vec); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  const double bytes = 2.0 * c.n_vec * 16, mfma = (double)MK * (c.n_vec / 64);
  printf("%-26s MFMA/1024 samples=%2d (high-plane %2d) U=%2d launch_bounds(256,%d) nt=%d span=%3ld KB  %7.3f ms  %5.2f TB/s  %6.0f TOP/s  frac of 8 TB/s %.3f\n", tag,
         2 * MK, 2 * HI, U, OCC, (int)NT, span_kb, ms, bytes / ms / 1e9, mfma * 65536.0 / ms / 1e9, bytes / ms / 1e9 / 8000.0);
  fflush(stdout);
}

int main(int argc, char **argv) {
  Ctx c;
  c.reps = argc > 1 ? atoi(argv[1]) : 10;
  const long n_ch = 1024, n = 1L << 20;
  c.n_vec = n_ch * n * 2 / 16;
  // A fragments with the byte statistics of config 2's set: windowed sinc (cutoff 0.1, Hamming), 255 taps, <16,2>
  std::vector<int> cf(255);
  {
    double sum = 0;
    std::vector<double> h(255);
    for (int k = 0; k < 255; k++) {
      const double t = k - 127.0, sinc = t == 0 ? 1.0 : sin(2 * M_PI * 0.1 * t) / (2 * M_PI * 0.1 * t);
      h[k] = sinc * 0.2 * (0.54 - 0.46 * cos(2 * M_PI * k / 254.0));
      sum += h[k];
    }
    for (int k = 0; k < 255; k++) { cf[k] = (int)lrint(h[k] / sum * 16384.0); }
  }
  std::vector<uint32_t> frag(6 * 64 * 4, 0);
  auto put = [&](int slot, int blk, bool hi) {   // Toeplitz block blk of 9: A[i][k] = c[i - k + 32 (8 - blk)], lane = i + 32 (k / 16)
    for (int lane = 0; lane < 64; lane++) {
      for (int dw = 0; dw < 4; dw++) {
        uint32_t w = 0;
        for (int bj = 0; bj < 4; bj++) {
          const int i = lane & 31, k = 16 * (lane >> 5) + 4 * dw + bj, tap = i - k + 32 * (8 - blk);
          int v = (tap >= 0 && tap < 255) ? cf[tap] : 0;
          const int lo = ((v + 128) & 0xff) - 128, hv = (v - lo) / 256;
          w |= (uint32_t)(uint8_t)(hi ? hv : lo) << (8 * bj);
        }
        frag[(slot * 64 + lane) * 4 + dw] = w;
      }
    }
  };
  put(0, 1, false); put(1, 3, false); put(2, 4, false); put(3, 6, false);   // four low-plane blocks
  put(4, 4, true); put(5, 5, true);                                         // two high-plane blocks of the band
  CK(hipMalloc((void **)&c.frag, frag.size() * 4));
  CK(hipMemcpy(c.frag, frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&c.x, c.n_vec * 16));
  CK(hipMalloc((void **)&c.y, c.n_vec * 16));
  {   // uniform random bytes
    std::vector<uint32_t> r(1 << 22);
    uint64_t st = 0xACD5;
    for (auto &w : r) { st = st * 6364136223846793005ull + 1442695040888963407ull; w = (uint32_t)(st >> 32); }
    for (long off = 0; off < c.n_vec * 16; off += (long)r.size() * 4) { CK(hipMemcpy((char *)c.x + off, r.data(), r.size() * 4, hipMemcpyHostToDevice)); }
  }
  printf("# fir_envelope: 1024 ch x 2^20 samples, 2 B in + 2 B out per sample = %.3f GB per launch; target 0.767 ms (70 %% of 8 TB/s)\n", 2.0 * c.n_vec * 16 / 1e9);
  if (argc > 2) {   // one row only, `reps` launches back to back: a steady load for tools/power_sample_cmd.sh
    const std::string row = argv[2];
    if (row == "stream") { run<0, 0, 8, 2, true>(c, 32, "stream only"); }
    else if (row == "mfma") { run_mfma<13, 4, 2>(c, 0); }
    else { run<13, 4, 8, 2, true>(c, 32, "stream + MFMA (config 2)"); }
    return 0;
  }
  printf("# --- the stream alone, best geometry ---\n");
  run<0, 0, 8, 2, true>(c, 32, "stream only");
  run<0, 0, 8, 4, true>(c, 32, "stream only");
  run<0, 0, 8, 2, true>(c, 16, "stream only");
  printf("# --- the matrix pipe alone, same operands and counts, no memory traffic in the loop ---\n");
  run_mfma<13, 4, 2>(c, 0); run_mfma<13, 4, 2>(c, 1); run_mfma<18, 9, 2>(c, 0); run_mfma<8, 2, 2>(c, 0);
  printf("# --- MFMA count sweep (config 2 issues 26 per 1024 samples, 8 of them high-plane; dense sets 36 / 18) ---\n");
  run<4, 1, 8, 2, true>(c, 32, "stream + MFMA");
  run<8, 2, 8, 2, true>(c, 32, "stream + MFMA");
  run<10, 3, 8, 2, true>(c, 32, "stream + MFMA");
  run<12, 4, 8, 2, true>(c, 32, "stream + MFMA");
  run<13, 4, 8, 2, true>(c, 32, "stream + MFMA (config 2)");
  run<14, 4, 8, 2, true>(c, 32, "stream + MFMA");
  run<18, 9, 8, 2, true>(c, 32, "stream + MFMA (dense)");
  printf("# --- config 2's count in other geometries ---\n");
  run<13, 4, 8, 4, true>(c, 32, "stream + MFMA (config 2)");
  run<13, 4, 8, 3, true>(c, 32, "stream + MFMA (config 2)");
  run<13, 4, 4, 2, true>(c, 32, "stream + MFMA (config 2)");
  run<13, 4, 4, 4, true>(c, 32, "stream + MFMA (config 2)");
  run<13, 4, 16, 2, true>(c, 32, "stream + MFMA (config 2)");
  run<13, 4, 8, 2, true>(c, 16, "stream + MFMA (config 2)");
  run<13, 4, 8, 2, true>(c, 128, "stream + MFMA (config 2)");
  run<13, 4, 8, 2, false>(c, 32, "stream + MFMA (config 2)");
  printf("# --- the other int8 shape (v_mfma_i32_16x16x64_i8), same MAC count: is it cheaper per MAC? ---\n");
  run16<13, 8, 2, false>(c, 32); run16<13, 8, 2, true>(c, 32);
  printf("# --- both shapes with the data operand fetched through LDS (what a Toeplitz kernel has to do) ---\n");
  run_lds<false, 13, 9, 8, 2>(c, 32);    // config 2 on 32x32x32: 26 MFMAs, 18 fragment reads per 1024 samples
  run_lds<true, 28, 20, 8, 2>(c, 32);    // config 2 on 16x16x64: 56 instructions (28 equivalents), 40 fragment reads
  run_lds<true, 26, 18, 8, 2>(c, 32);    // 16x16x64 at the 32x32x32 formulation's counts (what the shape alone buys)
  run_lds<false, 13, 9, 8, 3>(c, 32);
  run_lds<true, 28, 20, 8, 3>(c, 32);
  printf("# --- drift check ---\n");
  run<13, 4, 8, 2, true>(c, 32, "stream + MFMA (config 2)");
  run<0, 0, 8, 2, true>(c, 32, "stream only");
  return 0;
}
