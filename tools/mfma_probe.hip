// mfma_probe.hip -- micro-benchmarks used while tuning fir_mfma.hip (not part of the product).
// Measures the int8 32x32x32 MFMA issue rate of one SIMD under the FIR kernel's instruction mix:
//   mode 0: NB*4 MFMAs per step on 3 accumulators, nothing else
//   mode 1: + the 32-bit epilogue VALU work (16 outputs/lane/step)
//   mode 2: + v_perm byte-plane split of a fresh operand each step
//   mode 3: + global loads (prefetch) and stores, FIR-like addressing (HBM-sized footprint: 32 channel groups)
//   mode 4: mode 3 without the stores     mode 5: mode 3 without the loads
//   mode 6: mode 0 with FOUR accumulators (the two middle products kept apart)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NB, int MODE, int WPS, int KPF>
__global__ void __launch_bounds__(64, WPS) probe(const v4i *__restrict__ frag, const short *__restrict__ x, short *__restrict__ y,
                                                 int steps, long stride) {
  const int lane = threadIdx.x;
  v4i Ah[NB], Al[NB], Xh[NB], Xl[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    Ah[b] = frag[b * 64 + lane]; Al[b] = frag[(NB + b) * 64 + lane];
    Xh[b] = frag[(2 * NB + b) * 64 + lane]; Xl[b] = frag[(3 * NB + b) * 64 + lane];
  }
  const long grp = blockIdx.x % 32, chunk = blockIdx.x / 32;
  const short *xr = x + (grp * 32 + (lane & 31)) * stride + (lane >> 5) * 16 + chunk * 32L * steps;
  short *yr = y + (grp * 32 + (lane & 31)) * stride + (lane >> 5) * 4 + chunk * 32L * steps;
  v4i na = {0, 0, 0, 0}, nb = {0, 0, 0, 0};
  v4i pa[KPF], pb[KPF];   // raw prefetch group: KPF consecutive blocks of this lane's row half
#pragma unroll
  for (int k = 0; k < KPF; k++) { pa[k] = na; pb[k] = nb; }
  if (MODE == 3 || MODE == 4) {
#pragma unroll
    for (int k = 0; k < KPF; k++) { pa[k] = *(const v4i *)(xr + k * 32); pb[k] = *(const v4i *)(xr + k * 32 + 8); }
  }
  int sink = 0;
  for (int s0 = 0; s0 < steps; s0 += NB) {
#pragma unroll
    for (int u = 0; u < NB; u++) {
      if (MODE >= 2) {
        Xh[u].x = __builtin_amdgcn_perm(na.y, na.x, 0x07050301u); Xh[u].y = __builtin_amdgcn_perm(na.w, na.z, 0x07050301u);
        Xh[u].z = __builtin_amdgcn_perm(nb.y, nb.x, 0x07050301u); Xh[u].w = __builtin_amdgcn_perm(nb.w, nb.z, 0x07050301u);
        Xl[u].x = __builtin_amdgcn_perm(na.y, na.x, 0x06040200u) ^ 0x80808080u; Xl[u].y = __builtin_amdgcn_perm(na.w, na.z, 0x06040200u) ^ 0x80808080u;
        Xl[u].z = __builtin_amdgcn_perm(nb.y, nb.x, 0x06040200u) ^ 0x80808080u; Xl[u].w = __builtin_amdgcn_perm(nb.w, nb.z, 0x06040200u) ^ 0x80808080u;
        if (MODE == 2 || MODE == 5) { na.x += 0x01010101; nb.y ^= na.x; }
      }
      if (MODE == 3 || MODE == 4) {
        na = pa[u % KPF]; nb = pb[u % KPF];
        if (u % KPF == KPF - 1) {   // group consumed: fetch the next KPF blocks back to back
          const short *src = xr + (long)(s0 + u + 1) * 32;
#pragma unroll
          for (int k = 0; k < KPF; k++) { pa[k] = *(const v4i *)(src + k * 32); pb[k] = *(const v4i *)(src + k * 32 + 8); }
        }
      }
      v16i hh = {0}, mid = {0}, ll = {0}, mid2 = {0};
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int slot = (u + 1 + b) % NB;
        hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Xh[slot], hh, 0, 0, 0);
        mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Xl[slot], mid, 0, 0, 0);
        ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Xl[slot], ll, 0, 0, 0);
        if (MODE == 6) { mid2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Xh[slot], mid2, 0, 0, 0); }
        else { mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Xh[slot], mid, 0, 0, 0); }
      }
      if (MODE == 0 || MODE == 6) {
        sink += hh[0] + mid[5] + ll[15] + mid2[3];
      } else {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          int o[4];
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int r = 4 * g + rr;
            const int A = (int)(((unsigned)hh[r] << 8) + (unsigned)mid[r]);
            const int B = (ll[r] + 77) >> 8;
            int q = (A + 1234 + B) >> 6;
            q = max(-32768, min(q, 32767));
            o[rr] = q;
          }
          if (MODE == 3 || MODE == 5) {
            v4s pk = {(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
            *(v4s *)(yr + (long)(s0 + u) * 32 + 8 * g) = pk;
          } else {
            sink += o[0] ^ o[1] ^ o[2] ^ o[3];
          }
        }
      }
    }
  }
  if (sink == 0x7fffffff) { y[lane] = (short)sink; }
}

template <int NB, int MODE, int WPS, int KPF = 1>
static void run(const char *name, int waves, int steps, const v4i *frag, const short *x, short *y, long stride) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NB, MODE, WPS, KPF>), dim3(waves), dim3(64), 0, 0, frag, x, y, steps, stride);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double mfma = (double)waves * steps * NB * 4;
  double cyc_per_mfma_per_simd = ms * 1e-3 * 2.4e9 / (mfma / 1024.0);
  printf("%-34s waves %5d steps %4d  %.3f ms  %.1f TOPS  %.1f cyc/MFMA/SIMD@2.4GHz\n", name, waves, steps, ms,
         mfma * 2 * 32768 / (ms * 1e-3) / 1e12, cyc_per_mfma_per_simd);
}

// mode P: software-pipelined: the epilogue VALU of step s-1 is interleaved with the MFMAs of step s
// (sched_group_barrier: 1 MFMA, then VPM VALU), four accumulators, 1 wave per SIMD.
static long long *g_clk;

template <int NB, int VPM, int WPS, int AG = 0>
__global__ void __launch_bounds__(64, WPS) probe_pipe(const v4i *__restrict__ frag, short *__restrict__ y, int steps, long long *clk) {
  const int lane = threadIdx.x;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  v4i Ah[NB], Al[NB], Xh[NB], Xl[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    Ah[b] = frag[b * 64 + lane]; Al[b] = frag[(NB + b) * 64 + lane];
    Xh[b] = frag[(2 * NB + b) * 64 + lane]; Xl[b] = frag[(3 * NB + b) * 64 + lane];
  }
  v16i ph = {0}, pm1 = {0}, pm2 = {0}, pl = {0};   // accumulators of the previous step
  int sink = 0;
  for (int s = 0; s < steps; s++) {
    v16i hh = {0}, m1 = {0}, m2 = {0}, ll = {0};
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (AG) {   // accumulators pinned to the AGPR half of the register file
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(hh) : "v"(Ah[b]), "v"(Xh[b]));
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(m1) : "v"(Ah[b]), "v"(Xl[b]));
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(ll) : "v"(Al[b]), "v"(Xl[b]));
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(m2) : "v"(Al[b]), "v"(Xh[b]));
      } else {
      hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Xh[b], hh, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Xl[b], m1, 0, 0, 0);
      ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Xl[b], ll, 0, 0, 0);
      m2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Xh[b], m2, 0, 0, 0);
      }
    }
    // epilogue of the previous step (independent of this step's MFMAs); VPM < 0: MFMAs only
#pragma unroll
    for (int r = 0; r < (VPM < 0 ? 2 : 16); r += 2) {
      const int A0 = (int)(((unsigned)ph[r] << 8) + (unsigned)pm1[r]);
      const int A1 = (int)(((unsigned)ph[r + 1] << 8) + (unsigned)pm1[r + 1]);
      const int q0 = (A0 + pm2[r] + ((pl[r] + 77) >> 8)) >> 6;
      const int q1 = (A1 + pm2[r + 1] + ((pl[r + 1] + 77) >> 8)) >> 6;
      typedef short v2s __attribute__((ext_vector_type(2)));
      const v2s pk = __builtin_amdgcn_cvt_pk_i16(q0, q1);
      sink += (int)pk.x ^ (int)pk.y;
    }
    if (VPM > 0) {
#pragma unroll
      for (int i = 0; i < 4 * NB; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VPM VALU
      }
    }
    ph = hh; pm1 = m1; pm2 = m2; pl = ll;
    Xh[0].x += sink & 1;   // keep the loop honest
  }
  if (blockIdx.x == 0 && lane == 0) { clk[0] = (long long)(__builtin_readcyclecounter() - c0); clk[1] = (long long)(__builtin_amdgcn_s_memrealtime() - r0); }
  if (sink == 0x7fffffff) { y[lane] = (short)(sink + ph[0] + pm1[1] + pm2[2] + pl[3]); }
  else if (lane == 77) { y[0] = (short)(ph[0] + pm1[1] + pm2[2] + pl[3]); }
}

template <int NB, int VPM, int WPS, int AG = 0>
static void run_pipe(const char *name, int waves, int steps, const v4i *frag, short *y) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_pipe<NB, VPM, WPS, AG>), dim3(waves), dim3(64), 0, 0, frag, y, steps, g_clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double mfma = (double)waves * steps * NB * 4;
  long long hc[2]; CK(hipMemcpy(hc, g_clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)hc[0] / ((double)hc[1] * 10.0);   // s_memrealtime ticks at 100 MHz
  printf("%-34s waves %5d steps %4d  %.3f ms  %.1f TOPS  %.1f cyc/MFMA/SIMD@2.4GHz  shader clock %.2f GHz -> %.1f real cyc/MFMA (wave 0: %.1f cyc/MFMA)\n",
         name, waves, steps, ms, mfma * 2 * 32768 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfma / 1024.0), ghz,
         ms * 1e-3 * ghz * 1e9 / (mfma / 1024.0), (double)hc[0] / (steps * NB * 4.0));
}

int main() {
  const int NB = 9;
  std::vector<int> hf(4 * NB * 64 * 4);
  for (size_t i = 0; i < hf.size(); i++) { hf[i] = (int)(i * 2654435761u); }
  v4i *frag; CK(hipMalloc(&frag, hf.size() * 4)); CK(hipMemcpy(frag, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  const long stride = (1 << 20) + 64;
  short *x, *y;
  CK(hipMalloc(&x, 1025 * stride * 2)); CK(hipMalloc(&y, 1025 * stride * 2));
  {
    std::vector<short> hx(1 << 24);
    for (size_t i = 0; i < hx.size(); i++) { hx[i] = (short)((i * 2654435761u) >> 11); }
    for (long off = 0; off + (long)hx.size() <= 1025 * stride; off += hx.size()) { CK(hipMemcpy(x + off, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); }
  }
  CK(hipMalloc(&g_clk, 16));
  run<9, 6, 1>("mfma only, 4 acc, 1 wave/SIMD", 1024, 512, frag, x, y, stride);
  run<9, 6, 2>("mfma only, 4 acc, 2 waves/SIMD", 2048, 512, frag, x, y, stride);
  run_pipe<9, 0, 1>("pipelined epi, no hints, 1 w/SIMD", 1024, 512, frag, y);
  run_pipe<9, 3, 1>("pipelined epi, 1 MFMA : 3 VALU, 1w", 1024, 512, frag, y);
  run_pipe<9, 4, 1>("pipelined epi, 1 MFMA : 4 VALU, 1w", 1024, 512, frag, y);
  run_pipe<9, 4, 2>("pipelined epi, 1 MFMA : 4 VALU, 2w", 2048, 512, frag, y);
  run_pipe<9, -1, 1>("pipe kernel, MFMA only, 1w", 1024, 512, frag, y);
  run_pipe<9, -1, 2>("pipe kernel, MFMA only, 2w", 2048, 512, frag, y);
  run_pipe<9, 0, 1, 1>("AGPR acc: pipelined epi, 1w", 1024, 512, frag, y);
  run_pipe<9, 0, 2, 1>("AGPR acc: pipelined epi, 2w", 2048, 512, frag, y);
  run_pipe<9, 0, 2, 0>("VGPR acc: pipelined epi, 2w", 2048, 512, frag, y);
  return 0;
}
