// intg_dump.hip -- integrate-and-dump: reference include/ac_dsp/ac_intg_dump.h:93-147.
//
// The reference walks an interleaved stream (round-major, channel-minor): per block it reads n_sample, then adds one
// sample per channel and round into ACC_TYPE temp[i] (`temp[i] = temp[i] + data_in`, every add quantised, :97) and, in
// round j == n_sample, writes OUT(temp[i]) and clears it (:98-102).  A block whose n_sample is 0 or > NS runs NS rounds
// and dumps nothing: its sums carry into the next block.  Here one thread owns one (object, block, channel): it starts
// from the handle's temp[] (first chain of the call) or 0, replays the rounds of its carry chain in order and writes
// the block's output; one more thread per (object, channel) leaves the trailing, undumped sum in the handle.
#include "fir_kernels.hpp"

namespace acdsp {

__device__ inline i128 id_shl128(i128 v, int s) { return (i128)((u128)v << s); }

__device__ int64_t intg_chain(const IntgDumpParams &p, int obj, int i, int first_blk, int last_blk) {
  int64_t acc = (first_blk == 0) ? p.temp[((int64_t)obj * p.chn) + i] : 0;
  if (p.lossless) {   // wrapping ACC_TYPE with at least IN_TYPE's fraction bits: every add is exact mod 2^W -> one wrap at the end
    uint64_t sum = (uint64_t)acc;
    const int sh = p.acc.F - p.in.F;
    for (int b = first_blk; b <= last_blk; b++) {
      const int64_t r0 = p.blk_off[b];
      for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
        sum += (uint64_t)load_raw(p.x, (int64_t)obj * p.in_stride + (r0 + r) * p.chn + i, p.in_eb, p.in.S) << sh;
      }
    }
    return wrap64((int64_t)sum, p.acc.W, p.acc.S);
  }
  const int f = p.in.F > p.acc.F ? p.in.F : p.acc.F;
  for (int b = first_blk; b <= last_blk; b++) {
    const int64_t r0 = p.blk_off[b];
    for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
      const int64_t x = load_raw(p.x, (int64_t)obj * p.in_stride + (r0 + r) * p.chn + i, p.in_eb, p.in.S);
      acc = requant128(id_shl128((i128)acc, f - p.acc.F) + id_shl128((i128)x, f - p.in.F), f, p.acc);
    }
  }
  return acc;
}

__global__ void intg_dump_kernel(IntgDumpParams p, int64_t *temp_next) {
  const int obj = blockIdx.y;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_work = (int64_t)(p.n_blocks + 1) * p.chn;     // blocks, then the state slot
  if (tid >= n_work) { return; }
  const int b = (int)(tid / p.chn), i = (int)(tid % p.chn);
  if (b < p.n_blocks) {
    if (p.blk_out[b] < 0) { return; }                            // no dump in this block
    const int64_t acc = intg_chain(p, obj, i, p.blk_chain[b], b);
    store_raw(p.y, (int64_t)obj * p.out_stride + p.blk_out[b] * p.chn + i, p.out_eb, requant64(acc, p.acc.F, p.out));
  } else {
    int64_t acc = 0;
    if (p.n_blocks == 0) { acc = p.temp[(int64_t)obj * p.chn + i]; }
    else if (p.blk_out[p.n_blocks - 1] < 0) { acc = intg_chain(p, obj, i, p.blk_chain[p.n_blocks - 1], p.n_blocks - 1); }
    temp_next[(int64_t)obj * p.chn + i] = acc;
  }
}

// Lossless class, every block dumps, nothing carried in: a workgroup takes 256 / CHN consecutive blocks of one object, pulls
// their (contiguous) samples into LDS with coalesced loads and lets one thread per (block, channel) add them up from there.
// (The general kernel's per-thread walk fetches 8-byte pieces 2 * CHN * n_sample bytes apart: 1.6 TB/s on the bench row.)
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <typename TIN>
__global__ void __launch_bounds__(256) intg_dump_tile_kernel(IntgDumpParams p, int blocks_per_wg, int lds_elems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char id_lds[];
  TIN *tile = (TIN *)id_lds;
  const int obj = blockIdx.y;
  const int b0 = blockIdx.x * blocks_per_wg;
  const int b1 = (b0 + blocks_per_wg < p.n_blocks) ? b0 + blocks_per_wg : p.n_blocks;
  const int64_t e0 = p.blk_off[b0] * p.chn;                                        // first element of the span
  const int64_t e1 = (p.blk_off[b1 - 1] + p.blk_rounds[b1 - 1]) * p.chn;
  const TIN *row = (const TIN *)p.x + (int64_t)obj * p.in_stride;
  const bool staged = e1 - e0 <= lds_elems;
  // LDS image padded by one dword per 512 bytes: the per-thread walks below start a block length apart (a power of two
  // in the common case) and would otherwise all sit on one bank
  constexpr int kPerDw = 4 / (int)sizeof(TIN) > 0 ? 4 / (int)sizeof(TIN) : 1;      // elements per pad
  constexpr int kShift = sizeof(TIN) == 2 ? 8 : (sizeof(TIN) == 4 ? 7 : 6);        // log2(elements per 512 bytes)
  auto pe = [&](int64_t e) -> int64_t { return e + (e >> kShift) * kPerDw; };
  if (staged) {
    const int64_t n_e = e1 - e0;
    if (sizeof(TIN) < 8 && ((uintptr_t)(row + e0) % 16) == 0) {                    // 16-byte loads, dword stores into the padded image
      constexpr int EPV = 16 / (int)sizeof(TIN);
      const int64_t n_v = n_e / EPV;
      for (int64_t v = threadIdx.x; v < n_v; v += 256) {
        const v4i_t q = ((const v4i_t *)(row + e0))[v];
        int *dst = (int *)(tile + pe(v * EPV));                                    // a 16-byte chunk never straddles a pad
        dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
      }
      for (int64_t e = n_v * EPV + threadIdx.x; e < n_e; e += 256) { tile[pe(e)] = row[e0 + e]; }
    } else {
      for (int64_t e = threadIdx.x; e < n_e; e += 256) { tile[pe(e)] = row[e0 + e]; }
    }
    __syncthreads();
  }
  const int b = b0 + (int)threadIdx.x / p.chn, i = (int)threadIdx.x % p.chn;
  if (b >= b1 || (int)threadIdx.x >= blocks_per_wg * p.chn) { return; }
  const int64_t base = p.blk_off[b] * p.chn - e0 + i;
  uint64_t sum = 0;
  const int sh = p.acc.F - p.in.F;
  for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
    int64_t x = staged ? (int64_t)tile[pe(base + r * p.chn)] : (int64_t)row[e0 + base + r * p.chn];
    if (!p.in.S) { x &= (int64_t)((uint64_t)(-1) >> (64 - 8 * (int)sizeof(TIN))); }
    sum += (uint64_t)x << sh;
  }
  const int64_t acc = wrap64((int64_t)sum, p.acc.W, p.acc.S);
  store_raw(p.y, (int64_t)obj * p.out_stride + p.blk_out[b] * p.chn + i, p.out_eb, requant64(acc, p.acc.F, p.out));
}

__global__ void intg_dump_zero_kernel(int64_t *temp_next, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) { temp_next[t] = 0; }
}

hipError_t launch_intg_dump(const IntgDumpParams &p, int64_t *temp_next, hipStream_t s) {
  if (p.tile_ok && p.chn <= 256) {
    const int bpw = 256 / p.chn;
    const int lds_elems = (47 * 1024) / p.in_eb;   // + one pad dword per 512 bytes stays inside the 48 KB
    dim3 grid((unsigned)((p.n_blocks + bpw - 1) / bpw), (unsigned)p.n_obj);
    switch (p.in_eb) {
      case 2: hipLaunchKernelGGL(intg_dump_tile_kernel<int16_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
      case 4: hipLaunchKernelGGL(intg_dump_tile_kernel<int32_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
      default: hipLaunchKernelGGL(intg_dump_tile_kernel<int64_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return e; }
    const int64_t nt = (int64_t)p.n_obj * p.chn;
    hipLaunchKernelGGL(intg_dump_zero_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, temp_next, nt);   // every block dumped: temp[] = 0
    return hipGetLastError();
  }
  const int64_t n_work = (int64_t)(p.n_blocks + 1) * p.chn;
  dim3 grid((unsigned)((n_work + 255) / 256), (unsigned)p.n_obj);
  hipLaunchKernelGGL(intg_dump_kernel, grid, dim3(256), 0, s, p, temp_next);
  return hipGetLastError();
}

}  // namespace acdsp
