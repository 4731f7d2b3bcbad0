# Native build of libtmvb_hip.so (the C-ABI library of include/tmvb.h) without Python.
# `python -c "import __graft_entry__ as g; g.build()"` does the same (topicmodelsvb.jl_amd/_lib.py) and is what the
# tests use; this Makefile is for a Julia-side maintainer.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
ROCM  ?= /opt/rocm
CSRC  := topicmodelsvb.jl_amd/csrc
SRCS  := $(CSRC)/tmvb_core.hip $(CSRC)/tmvb_comm.hip $(CSRC)/tmvb_lda.hip $(CSRC)/tmvb_flda.hip $(CSRC)/tmvb_ctm.hip $(CSRC)/tmvb_ctpf.hip $(CSRC)/tmvb_ctpf_recs.hip
HDRS  := $(wildcard $(CSRC)/*.h) include/tmvb.h
OBJS  := $(patsubst $(CSRC)/%.hip,topicmodelsvb.jl_amd/build/%.hip.o,$(SRCS))
FLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wno-pass-failed -I include -I $(CSRC)
LIB   := topicmodelsvb.jl_amd/libtmvb_hip.so

all: $(LIB)

topicmodelsvb.jl_amd/build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p topicmodelsvb.jl_amd/build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(OBJS) -ldl -Wl,-rpath,$(ROCM)/lib

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf topicmodelsvb.jl_amd/build $(LIB)

.PHONY: all oracle clean
