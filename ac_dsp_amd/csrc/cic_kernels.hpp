// cic_kernels.hpp -- launch interface between engine.hip and the CIC kernels.
#pragma once
#include <vector>

#include "acdsp_dev.hpp"

namespace acdsp {

constexpr int kCicTile = 64;    // input samples per LDS tile (per channel)
constexpr int kCicMaxN = 8;

struct CicParams {
  int32_t interp, R, me, N, n_ch;  // me = effective comb delay = min(M, 2) (see cic.hip)
  int32_t w_int;                   // width of the reference's INT_TYPE (wrap arithmetic)
  DFmt in, out;                    // in: only F and S are used
  int32_t in_eb, out_eb;
  int32_t hl;                      // history inputs kept per channel (multiple of kCicTile)
  int32_t warm_tiles;              // tiles simulated before a chunk to rebuild the filter memory
  int32_t vec_ok;                  // 16-byte aligned rows: vector loads allowed
  int32_t out_simple;              // OUT_TYPE has INT_TYPE's fraction and AC_WRAP: conversion is a bit-field wrap
  int64_t in_stride, out_stride, n_in;
  int64_t chunk;                   // inputs per wave chunk (multiple of kCicTile)
  const void *x;                   // inputs  [n_ch][in_stride]
  void *y;                         // outputs [n_ch][out_stride]
  const void *hist;                // [n_ch][hl]: the hl inputs before local t = 0
  // decimator: global phase of local t = 0 and local time of the first emitted sample
  int32_t phase0;
  int64_t first;
  // interpolator: iteration window [q_begin, q_end) of this call in global iteration numbers,
  // inputs consumed by earlier calls, and the number of start-up iterations that are never emitted
  int64_t q_begin, q_end, t_prev, q_skip;
  // interpolator, FIR-identity kernels: iterations [q_from, q_to) are produced by this launch (0, 0: the whole call);
  // the output index stays q - max(q_begin, q_skip)
  int64_t q_from, q_to;
  // decimator, recurrence kernel behind a two-stage launch (cic2.hip): chunks start at input t_from (a multiple of kCicTile) and only
  // outputs whose emitting sample is >= emit_from are stored (0, 0: the whole call)
  int64_t t_from, emit_from;
};

hipError_t launch_cic(const CicParams &p, hipStream_t s);
// Interpolator through its FIR identity (polyphase form), one thread per output; taps = z^-(N-1) * boxcar(R*me)^N as raw
// int64 words (n_taps = N*R*me), every tap < 2^31.
hipError_t launch_cic_intr_fir(const CicParams &p, const int64_t *d_taps, int n_taps, hipStream_t s);
hipError_t launch_cic_hist_update(const CicParams &p, void *hist_next, hipStream_t s);

// Two-stage decimator for R = R1 R2 (cic2.hip): stage 1 = the FIR identity of rate R1 on the matrix cores, stage 2 = N prefix-sum
// integrators at the R1-decimated rate + the combs at the output rate, one launch, nothing but the outputs leaves the CU.
struct FirGenPlan;
bool cic2_factor(int in_eb, int R, int me, int N, int *R1, int *R2, int *wu);      // false: no compiled stage-1 rate divides R
void cic2_stage1_taps(int R1, int N, std::vector<int64_t> *h);                      // z^-(N-1) boxcar(R1)^N
int cic2_hist_len(int in_eb, int R1, int N, int wu);                                // input history chunk 0 reads back to
// p as for launch_cic (p.first, p.hl, p.out_simple set); pl / d_frag: fir_gen_plan of the stage-1 taps for p.first % 16.
// *covered = outputs [0, covered) written by complete chunks; the caller runs launch_cic with t_from / emit_from on the rest.
hipError_t launch_cic2(const CicParams &p, const FirGenPlan &pl, const uint32_t *d_frag, int R1, int R2, int wu, int64_t n_out,
                       hipStream_t s, int64_t *covered);

}  // namespace acdsp
