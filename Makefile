# rt355 build: the gfx950 C-ABI library, the N-API shim, the CPU oracle (test infrastructure).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := compute_raytracer_amd/csrc
LIB      := compute_raytracer_amd/librt355.so
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -fno-fast-math -Wall -Wno-unused-function $(EXTRA)
SRCS     := $(sort $(wildcard $(CSRC)/*.hip $(CSRC)/*.h)) include/rt355.h
# identity of the build: profiles (profiles/traffic.json) are evidence for the sources they were taken with
BUILD_ID := $(shell cat $(SRCS) | sha256sum | cut -c1-16)
OBJS     := $(CSRC)/rt_api.o $(CSRC)/rt_kernels.o $(CSRC)/rt_bvh.o $(CSRC)/rt_triangles.o $(CSRC)/rt_assemble.o $(CSRC)/rt_comm.o

all: lib oracle node

lib: $(LIB)

# the ray-trace kernels: no FMA contraction, no SLP (packed-math) vectorisation -- see the
# header of rt_kernels.hip
$(CSRC)/rt_kernels.o: $(CSRC)/rt_kernels.hip $(CSRC)/rt_filter.h $(CSRC)/rt_device.h $(CSRC)/rt_types.h include/rt355.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -fno-slp-vectorize -c $< -o $@

$(CSRC)/rt_bvh.o: $(CSRC)/rt_bvh.hip $(CSRC)/rt_filter.h $(CSRC)/rt_device.h $(CSRC)/rt_types.h include/rt355.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -fno-slp-vectorize -c $< -o $@

$(CSRC)/rt_triangles.o: $(CSRC)/rt_triangles.hip $(CSRC)/rt_tri_device.h $(CSRC)/rt_device.h $(CSRC)/rt_types.h $(CSRC)/rt_tri_types.h include/rt355.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -fno-slp-vectorize -c $< -o $@

$(CSRC)/rt_assemble.o: $(CSRC)/rt_assemble.hip $(CSRC)/rt_types.h include/rt355.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/rt_api.o: $(SRCS)
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -DRT355_BUILD_ID='"$(BUILD_ID)"' -c $(CSRC)/rt_api.hip -o $@

# the RCCL entry points (rt_comm_init / rt_render_gather / rt_group_*)
$(CSRC)/rt_comm.o: $(CSRC)/rt_comm.hip $(CSRC)/rt_ctx.h $(CSRC)/rt_flow_build.h $(CSRC)/rt_types.h $(CSRC)/rt_tri_types.h $(CSRC)/rt_wait_poll.h $(CSRC)/rt_exchange_plan.h include/rt355.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -L/opt/rocm/lib -lrccl

oracle:
	$(MAKE) -s -C oracle

node:
	@if [ -f node/Makefile ]; then $(MAKE) -s -C node; fi

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -s -C oracle clean
	@if [ -f node/Makefile ]; then $(MAKE) -s -C node clean; fi

.PHONY: all lib oracle node clean
