# Top-level build: everything is built IN-TREE (the .so files travel to the GPU box with
# gpurun; they are git-ignored).  `python -c "import __graft_entry__ as g; g.build()"` runs this.
#
#   spz_amd/lib/libspz_amd.so    HIP kernels (spz_kernels.hip, spz_ply_kernels.hip, spz_median.hip) + C ABI (spz_abi.hip, spz_hostpath.hip, spz_exchange.hip)   (hipcc, gfx950)
#   spz_amd/lib/libspz_host.so   C++ drop-in layer spz::saveSpz/loadSpz + gzip (g++, zlib)
#   spz_amd/spz*.so              Python module `spz` (pybind11) over the C++ layer
#   spz_amd/bin/{ply_to_spz,spz_to_ply,spz_info}   the reference's three CLI tools over the C++ layer
#   oracle/liboracle.so, oracle/_ref/libspz_ref.so   CPU checkers (tests only)

ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
PYTHON  ?= python3
ARCH    ?= gfx950
LIBDIR  := $(ROOT)spz_amd/lib
CSRC    := $(ROOT)spz_amd/csrc
INC     := $(ROOT)include

# -ffp-contract=off: bit-exact quantisation needs every a*b+c rounded twice.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
            -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I$(INC)
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -I$(INC)

PYEXT   := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC   := $(shell $(PYTHON) -c "import sysconfig, pybind11; print('-I' + sysconfig.get_paths()['include'] + ' -I' + pybind11.get_include())")

all: device host python cli oracle

device: $(LIBDIR)/libspz_amd.so
host:   $(LIBDIR)/libspz_host.so
python: $(ROOT)spz_amd/spz$(PYEXT)
cli:    $(ROOT)spz_amd/bin/spz_tool $(ROOT)spz_amd/bin/dropin_user_test $(ROOT)spz_amd/bin/host_bench

DEVICE_SRCS := $(CSRC)/spz_kernels.hip $(CSRC)/spz_abi.hip $(CSRC)/spz_hostpath.hip $(CSRC)/spz_ply_kernels.hip $(CSRC)/spz_median.hip $(CSRC)/spz_exchange.hip $(CSRC)/spz_lz77.hip $(CSRC)/spz_inflate_dev.hip $(CSRC)/spz_place.hip
$(LIBDIR)/libspz_amd.so: $(DEVICE_SRCS) $(CSRC)/spz_common.hpp $(CSRC)/spz_kernel_params.hpp $(CSRC)/spz_lz77_core.hpp $(CSRC)/spz_huff_core.hpp $(CSRC)/spz_inflate_core.hpp $(INC)/spz_amd.h
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEVICE_SRCS) -ldl

$(LIBDIR)/libspz_host.so: $(CSRC)/spz_host.cpp $(CSRC)/spz_ply.cpp $(CSRC)/spz_deflate.cpp $(CSRC)/spz_lz77_model.cpp $(CSRC)/spz_lz77_core.hpp $(CSRC)/spz_huff_core.hpp $(CSRC)/spz_deflate.hpp $(CSRC)/spz_inflate.cpp $(CSRC)/spz_inflate.hpp $(CSRC)/spz_inflate_core.hpp $(CSRC)/spz_host_util.hpp $(INC)/spz_amd_host.hpp $(INC)/spz_amd.h $(LIBDIR)/libspz_amd.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(CSRC)/spz_host.cpp $(CSRC)/spz_ply.cpp $(CSRC)/spz_deflate.cpp $(CSRC)/spz_lz77_model.cpp $(CSRC)/spz_inflate.cpp -L$(LIBDIR) -lspz_amd -lz -ldl -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

$(ROOT)spz_amd/spz$(PYEXT): $(CSRC)/spz_py.cpp $(INC)/spz_amd_host.hpp $(LIBDIR)/libspz_host.so
	$(CXX) $(CXXFLAGS) $(PYINC) -fvisibility=hidden -shared -o $@ $(CSRC)/spz_py.cpp \
	    -L$(LIBDIR) -lspz_host -lspz_amd -Wl,-rpath,'$$ORIGIN/lib' -Wl,-rpath,/opt/rocm/lib

$(ROOT)spz_amd/bin/spz_tool: $(CSRC)/spz_cli.cpp $(INC)/spz_amd_host.hpp $(LIBDIR)/libspz_host.so
	mkdir -p $(ROOT)spz_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $(CSRC)/spz_cli.cpp -L$(LIBDIR) -lspz_host -lspz_amd \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib
	for t in ply_to_spz spz_to_ply spz_info; do ln -sf spz_tool $(ROOT)spz_amd/bin/$$t; done

# A user program written against the reference's C++ API, built against the compat headers
# (tests/cpp/dropin_user.cpp; its expected output comes from the same source built against the reference).
$(ROOT)spz_amd/bin/dropin_user_test: $(ROOT)tests/cpp/dropin_user.cpp $(INC)/compat/load-spz.h $(LIBDIR)/libspz_host.so
	mkdir -p $(ROOT)spz_amd/bin
	$(CXX) $(CXXFLAGS) -I$(INC)/compat -o $@ $(ROOT)tests/cpp/dropin_user.cpp -L$(LIBDIR) -lspz_host -lspz_amd \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

# The C++ boundary (host vectors in and out) timed without a language binding in the way.
$(ROOT)spz_amd/bin/host_bench: $(ROOT)tools/host_bench.cpp $(INC)/spz_amd_host.hpp $(LIBDIR)/libspz_host.so
	mkdir -p $(ROOT)spz_amd/bin
	$(CXX) $(CXXFLAGS) -o $@ $(ROOT)tools/host_bench.cpp -L$(LIBDIR) -lspz_host -lspz_amd -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C $(ROOT)oracle

# Kernel ISA + resource usage for inspection (not part of `all`).
asm: $(CSRC)/spz_kernels.hip
	mkdir -p $(ROOT)build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
	    -o $(ROOT)build/spz_kernels.s $(CSRC)/spz_kernels.hip

clean:
	rm -rf $(LIBDIR) $(ROOT)build $(ROOT)spz_amd/spz*.so $(ROOT)spz_amd/bin
	$(MAKE) -C $(ROOT)oracle clean

.PHONY: fuzz gzip-campaign all device host python cli oracle asm clean

# The byte-identity of the multi-threaded gzip writer against zlib 1.2.11 on ~9 300 randomized inputs (host
# only, about twenty minutes on 8 cores); the result line goes to profiles/.
gzip-campaign: host python
	$(PYTHON) $(ROOT)tools/gzip_campaign.py --inputs 9300 | tee $(ROOT)profiles/gzip_campaign.json

# Host-side robustness: the gzip readers and the .ply header parser under AddressSanitizer + UBSan
# (CPU only; the GPU pool has no sanitizer support).  Mutated inputs; must finish without a report.
fuzz: $(LIBDIR)/libspz_amd.so
	mkdir -p $(ROOT)build
	for t in gunzip_fuzz ply_fuzz deflate_fuzz inflate_fuzz; do \
	  $(CXX) -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu++17 -I$(INC) -o $(ROOT)build/$$t \
	    $(ROOT)tools/fuzz/$$t.cpp $(CSRC)/spz_host.cpp $(CSRC)/spz_ply.cpp $(CSRC)/spz_deflate.cpp $(CSRC)/spz_lz77_model.cpp $(CSRC)/spz_inflate.cpp -L$(LIBDIR) -lspz_amd -lz -ldl -lpthread \
	    -Wl,-rpath,$(abspath $(LIBDIR)) && ASAN_OPTIONS=detect_leaks=0 $(ROOT)build/$$t || exit 1; \
	done
