# Builds libmerv_hip.so (gfx950 only) and the oracle's C restatement.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := merv_amd/csrc
LIB   := merv_amd/lib/libmerv_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
HIP_SRCS := $(CSRC)/gemm.hip $(CSRC)/attention.hip $(CSRC)/rowops.hip $(CSRC)/preproc.hip $(CSRC)/backward.hip $(CSRC)/mxfp8.hip $(CSRC)/decode.hip $(CSRC)/capi.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)

all: lib oracle

lib: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/prof.h include/merv_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/sampler.o: $(CSRC)/sampler.cpp include/merv_hip.h
	g++ -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -c $< -o $@

$(CSRC)/prof.o: $(CSRC)/prof.cpp $(CSRC)/prof.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(HIP_OBJS) $(CSRC)/sampler.o $(CSRC)/prof.o
	@mkdir -p merv_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
