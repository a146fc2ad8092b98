# Builds the three in-tree shared libraries:
#   raytracing_amd/lib/librt_host.so         host scene front-end (g++)
#   raytracing_amd/lib/libraytracing_amd.so  HIP kernels + C ABI for gfx950 (hipcc)
#   oracle/liboracle.so                      CPU restatement used by tests / bench cpu_baseline (g++)
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

all: host device oracle
host: raytracing_amd/lib/librt_host.so
device: raytracing_amd/lib/libraytracing_amd.so
oracle: oracle/liboracle.so

raytracing_amd/lib/librt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p raytracing_amd/lib
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC)

raytracing_amd/lib/libraytracing_amd.so: $(DEV_SRC) $(DEV_HDR)
	@mkdir -p raytracing_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEV_SRC)

oracle/liboracle.so: $(ORACLE_SRC) $(ORACLE_HDR)
	$(CXX) $(CXXFLAGS) -pthread -shared -o $@ $(ORACLE_SRC)

clean:
	rm -f raytracing_amd/lib/*.so oracle/liboracle.so

.PHONY: all host device oracle clean
