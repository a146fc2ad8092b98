# Builds the three in-tree shared libraries:
#   raytracing_amd/lib/librt_host.so         host scene front-end (g++)
#   raytracing_amd/lib/libraytracing_amd.so  HIP kernels + C ABI for gfx950 (hipcc)
#   oracle/liboracle.so                      CPU restatement used by tests / bench cpu_baseline (g++)
# and tests/bin/abi_harness, a plain C caller of the device library's ABI (gcc) that the GPU tests run as a process.
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CXXFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function

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

clean:
	rm -f raytracing_amd/lib/*.so oracle/liboracle.so tests/bin/abi_harness

.PHONY: all host device oracle harness clean
