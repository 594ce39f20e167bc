# Builds libsccd_hip.so (gfx950 only) and the CPU oracle.  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := scalable-ccd_amd/csrc
OUT     := scalable-ccd_amd/sccd/libsccd_hip.so
# -ffp-contract=off: Tight-Inclusion parity needs every product/sum rounded as written; the
# only FMAs are explicit __builtin_fma calls (SCCD_OPT_ARITH = 1).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
            -Wall -Wextra -Wno-unused-parameter -Wno-unused-function -Wno-missing-field-initializers
SRCS    := $(CSRC)/api.hip $(CSRC)/build.hip $(CSRC)/drivers.hip $(CSRC)/boxes.hip $(CSRC)/scan.hip $(CSRC)/sort.hip $(CSRC)/sweep.hip $(CSRC)/narrow.hip
OBJS    := $(SRCS:.hip=.o) $(CSRC)/ti_census.o
CPPTEST := tests/cpp/test_ccd_api
HDRS    := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc) include/sccd.h

WALKPROBE := tests/gpu_probe/walk_probe

all: $(OUT) oracle $(CPPTEST) $(WALKPROBE)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# host-only: one query's bisection in level order (the certificate of a check limit); same contract flags as the kernels
$(CSRC)/ti_census.o: $(CSRC)/ti_census.cpp $(CSRC)/ti_math.hpp
	g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unknown-pragmas -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -pthread -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(OBJS) $(OUT)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean

# C++ parity test of include/scalable_ccd/hip/ccd.hpp (compiles without a GPU, runs on one)
$(CPPTEST): tests/cpp/test_ccd_api.cpp include/scalable_ccd/hip/ccd.hpp include/sccd.h $(OUT) oracle
	g++ -std=c++17 -O1 -Wall -Wextra -Iinclude tests/cpp/test_ccd_api.cpp -o $@ \
	    -Lscalable-ccd_amd/sccd -lsccd_hip -Loracle -lsccd_oracle \
	    -Wl,-rpath,'$$ORIGIN/../../scalable-ccd_amd/sccd' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath,/opt/rocm/lib
cpptest: $(CPPTEST)
.PHONY: cpptest

# device-vs-host check of the stackless-walk helpers of ti_math.hpp (compiles without a GPU, runs on one: tests/test_gpu_parity.py)
$(WALKPROBE): tests/gpu_probe/walk_probe.hip tests/gpu_probe/walk_probe_host.cpp tests/gpu_probe/walk_probe_ops.h $(CSRC)/ti_math.hpp
	g++ -O1 -std=c++17 -I$(CSRC) -Itests/gpu_probe -c tests/gpu_probe/walk_probe_host.cpp -o tests/gpu_probe/walk_probe_host.o
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -I$(CSRC) -Itests/gpu_probe -c tests/gpu_probe/walk_probe.hip -o tests/gpu_probe/walk_probe_dev.o
	$(HIPCC) --offload-arch=$(ARCH) tests/gpu_probe/walk_probe_dev.o tests/gpu_probe/walk_probe_host.o -o $@
