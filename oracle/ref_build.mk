# oracle/ref_build.mk -- ONE COMMAND that pins the oracle to the reference's own CPU broad phase, the day real headers exist.
#
#   make -C oracle -f ref_build.mk SCCD_EIGEN_DIR=/path/to/eigen SCCD_TBB_DIR=/path/to/oneTBB SCCD_SPDLOG_DIR=/path/to/spdlog
#   python -m pytest tests/test_reference_build.py
#
# Compiles the reference's CPU sources UNMODIFIED, IN PLACE, from $(REF) (default /root/reference):
#   src/scalable_ccd/broad_phase/aabb.cpp            (build_vertex/edge/face_boxes, aabb.cpp:38-133)
#   src/scalable_ccd/broad_phase/sort_and_sweep.cpp  (sort_and_sweep, :198-240)
#   src/scalable_ccd/utils/merge_local_overlaps.cpp, src/scalable_ccd/utils/logger.cpp
# against REAL Eigen / oneTBB / spdlog trees handed in by the caller, plus oracle/ref_driver.cpp (ours: reads a mesh or a
# box list, calls the reference's functions, dumps boxes and sorted pairs).  Outputs go to oracle/_ref/ only (git-ignored,
# not gpurun-ignored).  The only generated file is scalable_ccd/config.hpp, made by CMAKE ITSELF (configure_file in script
# mode, ref_config.cmake) from the reference's own template src/scalable_ccd/config.hpp.in with the reference's default
# options (CMakeLists.txt:68-72: USE_DOUBLE=ON, everything else OFF) -- into oracle/_ref/include, because the reference tree
# is read-only here.
#
# It REFUSES to build without the three directories: no stand-in headers, ever (this image has none of the three, so here
# the recipe stops with the message below and tests/test_reference_build.py skips).
REF            ?= /root/reference
OUT            := _ref
CXX            ?= g++
SCCD_TBB_LIBDIR ?= $(SCCD_TBB_DIR)/lib

REF_SRCS := $(REF)/src/scalable_ccd/broad_phase/aabb.cpp \
            $(REF)/src/scalable_ccd/broad_phase/sort_and_sweep.cpp \
            $(REF)/src/scalable_ccd/utils/merge_local_overlaps.cpp \
            $(REF)/src/scalable_ccd/utils/logger.cpp

all: check $(OUT)/ref_driver

check:
	@test -d "$(REF)/src/scalable_ccd" || { echo "ref_build: no reference tree at REF=$(REF)"; exit 2; }
	@test -n "$(SCCD_EIGEN_DIR)" -a -f "$(SCCD_EIGEN_DIR)/Eigen/Core" || { echo "ref_build: SCCD_EIGEN_DIR must hold Eigen/Core (a real Eigen checkout; none is installed in this image) -- refusing to build"; exit 2; }
	@test -n "$(SCCD_TBB_DIR)" -a -f "$(SCCD_TBB_DIR)/include/tbb/parallel_for.h" || { echo "ref_build: SCCD_TBB_DIR must hold include/tbb/parallel_for.h (oneTBB) -- refusing to build"; exit 2; }
	@test -n "$(SCCD_SPDLOG_DIR)" -a -f "$(SCCD_SPDLOG_DIR)/include/spdlog/spdlog.h" || { echo "ref_build: SCCD_SPDLOG_DIR must hold include/spdlog/spdlog.h -- refusing to build"; exit 2; }

$(OUT)/include/scalable_ccd/config.hpp: $(REF)/src/scalable_ccd/config.hpp.in ref_config.cmake
	mkdir -p $(OUT)/include/scalable_ccd
	cmake -DIN=$(REF)/src/scalable_ccd/config.hpp.in -DOUT=$(abspath $(OUT))/include/scalable_ccd/config.hpp -P ref_config.cmake

# (header-only spdlog: SPDLOG_COMPILED_LIB stays undefined; oneTBB is linked from SCCD_TBB_LIBDIR)
$(OUT)/ref_driver: check $(OUT)/include/scalable_ccd/config.hpp ref_driver.cpp $(REF_SRCS)
	$(CXX) -std=c++17 -O2 -I$(OUT)/include -I$(REF)/src -I$(SCCD_EIGEN_DIR) -I$(SCCD_TBB_DIR)/include -I$(SCCD_SPDLOG_DIR)/include \
	    $(REF_SRCS) ref_driver.cpp -L$(SCCD_TBB_LIBDIR) -Wl,-rpath,$(SCCD_TBB_LIBDIR) -ltbb -pthread -o $@

clean:
	rm -rf $(OUT)/ref_driver $(OUT)/include

.PHONY: all check clean
