# Builds libmerv_hip.so (gfx950 only) and the oracle's C restatement.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := merv_amd/csrc
LIB   := merv_amd/lib/libmerv_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
HIP_SRCS := $(CSRC)/gemm.hip $(CSRC)/attention.hip $(CSRC)/rowops.hip $(CSRC)/preproc.hip $(CSRC)/backward.hip $(CSRC)/mxfp8.hip $(CSRC)/decode.hip $(CSRC)/capi.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)

HOOKS_LIB := merv_amd/lib/libmerv_hip_hooks.so
HOOKS_DIR := build/hooks
HOOKS_OBJS := $(patsubst $(CSRC)/%.hip,$(HOOKS_DIR)/%.o,$(HIP_SRCS))

all: lib hooks oracle

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

# The same sources with the tuning hooks compiled in (-DMERV_TUNING_HOOKS: the MERV_* tuning variables are read, merv_debug_set_* work):
# test / probe infrastructure, loaded only under MERV_TUNING_HOOKS=1 (merv_amd/_lib.py). The product library above has none of it.
hooks: $(HOOKS_LIB)

$(HOOKS_DIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/prof.h include/merv_hip.h
	@mkdir -p $(HOOKS_DIR)
	$(HIPCC) $(HIPFLAGS) -DMERV_TUNING_HOOKS -c $< -o $@

$(HOOKS_LIB): $(HOOKS_OBJS) $(CSRC)/sampler.o $(CSRC)/prof.o
	@mkdir -p merv_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o $(LIB) $(HOOKS_LIB)
	rm -rf $(HOOKS_DIR)
	$(MAKE) -C oracle clean

.PHONY: all lib hooks oracle clean
