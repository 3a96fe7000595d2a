# Builds libacdsp.so (HIP engine, gfx950 only) in-tree and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = ac_dsp_amd/csrc
OUT = ac_dsp_amd/lib/libacdsp.so
SRCS = $(CSRC)/engine.hip $(CSRC)/engine_fir.hip $(CSRC)/engine_cic.hip $(CSRC)/engine_ddc.hip $(CSRC)/engine_poly.hip $(CSRC)/engine_misc.hip $(CSRC)/fir_generic.hip $(CSRC)/fir_mfma.hip $(CSRC)/fir_mfma_mid.hip $(CSRC)/fir_mfma_mid2.hip $(CSRC)/fir_mfma_mid3.hip $(CSRC)/fir_mfma_alt.hip $(CSRC)/fir_mfma_alt2.hip $(CSRC)/fir_gen.hip $(CSRC)/fir_up.hip $(CSRC)/fir_up_b.hip $(CSRC)/fir_up_c.hip $(CSRC)/polydec.hip $(CSRC)/polyintr.hip $(CSRC)/intg_dump.hip $(CSRC)/mv_avg.hip $(CSRC)/cic.hip $(CSRC)/cic2.hip $(CSRC)/cic2_b.hip $(CSRC)/cic2_c.hip $(CSRC)/cic2_d.hip $(CSRC)/cic2_e.hip $(CSRC)/cic2_f.hip $(CSRC)/wide.hip $(CSRC)/diag.hip $(CSRC)/node.hip
HDRS = $(CSRC)/acdsp_dev.hpp $(CSRC)/fir_kernels.hpp $(CSRC)/cic_kernels.hpp include/acdsp.h
OBJS = $(SRCS:.hip=.o)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function

all: $(OUT) oracle

# formats wider than 64 bits: only the C-ABI layer and wide.hip see these headers
ENGINE_OBJS = $(CSRC)/engine.o $(CSRC)/engine_fir.o $(CSRC)/engine_cic.o $(CSRC)/engine_ddc.o $(CSRC)/engine_poly.o $(CSRC)/engine_misc.o
$(ENGINE_OBJS) $(CSRC)/wide.o: $(CSRC)/wide_kernels.hpp $(CSRC)/wide_int.hpp
$(ENGINE_OBJS): $(CSRC)/engine_common.hpp

$(CSRC)/fir_mfma_mid.o $(CSRC)/fir_mfma_mid2.o $(CSRC)/fir_mfma_mid3.o $(CSRC)/fir_mfma_alt.o $(CSRC)/fir_mfma_alt2.o: $(CSRC)/fir_mfma.hip
$(CSRC)/fir_up_b.o $(CSRC)/fir_up_c.o: $(CSRC)/fir_up.hip
$(CSRC)/cic2_b.o $(CSRC)/cic2_c.o $(CSRC)/cic2_d.o $(CSRC)/cic2_e.o $(CSRC)/cic2_f.o: $(CSRC)/cic2.hip

%.o: %.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p ac_dsp_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(OUT)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
