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

hipError_t launch_intg_dump(const IntgDumpParams &p, int64_t *temp_next, hipStream_t s) {
  const int64_t n_work = (int64_t)(p.n_blocks + 1) * p.chn;
  dim3 grid((unsigned)((n_work + 255) / 256), (unsigned)p.n_obj);
  hipLaunchKernelGGL(intg_dump_kernel, grid, dim3(256), 0, s, p, temp_next);
  return hipGetLastError();
}

}  // namespace acdsp
