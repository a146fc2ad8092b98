# Builds the three in-tree shared libraries:
#   raytracing_amd/lib/librt_host.so         host scene front-end (g++)
#   raytracing_amd/lib/libraytracing_amd.so  HIP kernels + C ABI for gfx950 (hipcc)
#   oracle/liboracle.so                      CPU restatement used by tests / bench cpu_baseline (g++)
# and tests/bin/abi_harness, a plain C caller of the device library's ABI (gcc) that the GPU tests run as a process.
# `make emu` builds the CPU checkers of tests/emu with AddressSanitizer: tests/bin/emu_trace_m0 / _m1 (the tracing device
# headers) and tests/bin/emu_library (the whole device library, host code and kernels); tests/test_emu_*.py build them
# themselves when they are missing or stale.
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CXXFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -pthread
# -fno-slp-vectorize: the compiler pairs neighbouring float multiplies and adds into v_pk_mul_f32 / v_pk_add_f32, which a
# SIMD issues in 4.1 cycles - what the two plain instructions take (2.2 each, tools/valu_rate.hip) - and pays register moves
# to line the pairs up. Without it k_shade has 97 moves and 10 registers fewer; 12.7 -> 13.1 G rays/s on the cornellBox.
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function

HOST_SRC = $(wildcard raytracing_amd/csrc/host/*.cpp)
HOST_HDR = $(wildcard raytracing_amd/csrc/host/*.h) $(wildcard include/*.h)
DEV_SRC = $(wildcard raytracing_amd/csrc/device/*.hip)
DEV_HDR = $(wildcard raytracing_amd/csrc/device/*.h) $(wildcard include/*.h)
ORACLE_SRC = $(wildcard oracle/*.cpp)
ORACLE_HDR = $(wildcard oracle/*.h) $(wildcard include/*.h)

all: host device oracle harness
host: raytracing_amd/lib/librt_host.so
device: raytracing_amd/lib/libraytracing_amd.so
oracle: oracle/liboracle.so
harness: tests/bin/abi_harness

raytracing_amd/lib/librt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p raytracing_amd/lib
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC)

raytracing_amd/lib/libraytracing_amd.so: $(DEV_SRC) $(DEV_HDR)
	@mkdir -p raytracing_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEV_SRC)

oracle/liboracle.so: $(ORACLE_SRC) $(ORACLE_HDR)
	$(CXX) $(CXXFLAGS) -pthread -shared -o $@ $(ORACLE_SRC)

tests/bin/abi_harness: tests/abi_harness.c include/raytracing_amd.h raytracing_amd/lib/libraytracing_amd.so
	@mkdir -p tests/bin
	gcc -std=c99 -O2 -Wall -Iinclude -o $@ tests/abi_harness.c -Lraytracing_amd/lib -lraytracing_amd \
	    -Wl,-rpath,'$$ORIGIN/../../raytracing_amd/lib' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib

EMU_SRC = tests/emu/emu_trace.cpp tests/emu/simt.h tests/emu/hip/hip_runtime.h
EMU_FLAGS = -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -static-libasan -static-libubsan -fno-omit-frame-pointer -ffp-contract=off -Wall -Wno-unknown-pragmas \
            -Wno-unused-function -Wno-unused-variable -DRT_STACK_LDS_LARGE=4 -Itests/emu
emu: tests/bin/emu_trace_m0 tests/bin/emu_trace_m1 tests/bin/emu_library
tests/bin/emu_library: tests/emu/emu_library.cpp tests/emu/simt.h tests/emu/hip/hip_runtime.h tests/emu/rccl/rccl.h $(DEV_SRC) $(DEV_HDR)
	@mkdir -p tests/bin
	$(CXX) $(filter-out -DRT_STACK_LDS_LARGE=4,$(EMU_FLAGS)) -o $@ tests/emu/emu_library.cpp -ldl -lpthread
tests/bin/emu_trace_m%: $(EMU_SRC) $(DEV_HDR)
	@mkdir -p tests/bin
	$(CXX) $(EMU_FLAGS) -DRT_STREAM_MERGED=$* -o $@ tests/emu/emu_trace.cpp

clean:
	rm -f raytracing_amd/lib/*.so oracle/liboracle.so tests/bin/abi_harness tests/bin/emu_trace_m0 tests/bin/emu_trace_m1 tests/bin/emu_library

.PHONY: all host device oracle harness emu clean
