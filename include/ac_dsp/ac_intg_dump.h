// ac_intg_dump.h -- drop-in for hlslibs/ac_dsp's integrate-and-dump block, MI355X back end.
//
// Same class template and run() signature as the reference (include/ac_dsp/ac_intg_dump.h:113-151).  run() drains the
// input FIFO block by block exactly like the reference loop (:133-147): one N_TYPE word per block from n_sample, then
// min(n_sample, NS) rounds (NS when n_sample is 0 or larger than NS -- no dump then, the sums carry on) of CHN
// interleaved samples; the accumulation (`temp[i] = temp[i] + data_in`, ACC_TYPE, :97) runs as a HIP kernel behind
// include/acdsp.h (acdsp_intgdump_*).  Like the reference, a block that the FIFO cannot complete is an error (the
// reference reads an empty ac_channel); here it aborts with a message before any sample is consumed.
#ifndef _INCLUDED_AC_INTG_DUMP_H_
#define _INCLUDED_AC_INTG_DUMP_H_

#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_channel.h>
#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template < class IN_TYPE, class ACC_TYPE, class OUT_TYPE, class N_TYPE, int NS, int CHN >
class ac_intg_dump
{
public:
  ac_intg_dump() : h_(0) {}
  ~ac_intg_dump() { if (h_) { acdsp_intgdump_destroy(h_); } }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)(ac_channel < IN_TYPE > &data_in, ac_channel < OUT_TYPE > &data_out, ac_channel < N_TYPE > &n_sample) {
    std::vector<int64_t> blocks, raw;
    while (data_in.available(1)) {
      if (!n_sample.available(1)) { die("data left in data_in but no n_sample word (the reference reads an empty channel)"); }
      const int64_t n = (int64_t)(long long)n_sample.read();
      const int64_t rounds = (n >= 1 && n <= NS) ? n : NS;
      if (!data_in.available((unsigned)(rounds * CHN))) { die("data_in ends inside a block (the reference reads an empty channel)"); }
      for (int64_t k = 0; k < rounds * CHN; k++) { raw.push_back(acdsp::raw_of(data_in.read())); }
      blocks.push_back(n);
    }
    if (blocks.empty()) { return; }
    ensure();
    const int ib = acdsp_elem_bytes(IN_TYPE::width), ob = acdsp_elem_bytes(OUT_TYPE::width);
    std::vector<unsigned char> bi, bo(blocks.size() * (size_t)CHN * (size_t)ob + 16);
    acdsp::pack(raw, ib, bi);
    int64_t n_out = 0;
    acdsp::check(acdsp_intgdump_run_host(h_, bi.data(), blocks.data(), (int64_t)blocks.size(), bo.data(), (int64_t)blocks.size() * CHN, &n_out),
                 "acdsp_intgdump_run_host");
    for (int64_t i = 0; i < n_out; i++) {
      data_out.write(acdsp::from_raw<OUT_TYPE>(acdsp::unpack_one(&bo[(size_t)i * ob], ob, OUT_TYPE::sign)));
    }
  }

private:
  ac_intg_dump(const ac_intg_dump &);
  ac_intg_dump &operator=(const ac_intg_dump &);
  static void die(const char *why) { fprintf(stderr, "ac_intg_dump (MI355X engine): %s\n", why); abort(); }
  void ensure() {
    if (h_) { return; }
    acdsp_intgdump_desc_t d;
    d.ns = NS; d.chn = CHN; d.n_objects = 1;
    d.in = acdsp::fmt_of<IN_TYPE>(); d.acc = acdsp::fmt_of<ACC_TYPE>(); d.out = acdsp::fmt_of<OUT_TYPE>();
    d.device = acdsp::default_device(); d.flags = 0;
    acdsp::check(acdsp_intgdump_create(&d, &h_), "acdsp_intgdump_create");
  }
  acdsp_intgdump_t h_;
};

#endif
