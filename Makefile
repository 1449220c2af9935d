# Build everything that lives in-tree (the .so files travel to the GPU box with gpurun):
#   star_amd/lib/libstaramd_host.so   host side (index loader, FASTQ batcher, post-map, SAM/SJ writers)  g++
#   star_amd/lib/libstaramd.so        HIP engine behind include/star_amd.h                                hipcc gfx950
#   star_amd/bin/star_amd             CLI: drop-in for `STAR --runMode alignReads`
#   oracle/_build/liboracle.so        CPU restatement (test infrastructure)
#   oracle/_ref/STAR                  the reference itself, when /root/reference is present
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -Wno-sign-compare -pthread
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result
# k_stitch.hip is compiled without loop unrolling (round 2: fewer spills, 3 % faster; round 5: 106 VGPRs, no scratch -- tools/isa_stats.sh k_stitch says which of the
# register allocator's two regimes a build is in, tests/test_isa_static.py notices a flip); the other kernels keep the default
STITCH_WAVES ?= 4
STITCH_FLAGS := -fno-unroll-loops -DSTITCH_WAVES=$(STITCH_WAVES)

# $(call build_engine,<variant>,<extra defines>): every .hip file to its own object (in parallel), then one shared library
define build_engine
	@mkdir -p star_amd/lib/obj/$(1)
	@set -e; for f in $(HIP_SRC); do b=$$(basename $$f .hip); extra=""; if [ $$b = k_stitch ]; then extra="$(STITCH_FLAGS)"; fi; \
	  $(HIPCC) $(HIPFLAGS) $(2) $$extra -c $$f -o star_amd/lib/obj/$(1)/$$b.o & done; wait; \
	  for f in $(HIP_SRC); do test -s star_amd/lib/obj/$(1)/$$(basename $$f .hip).o; done
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(foreach f,$(HIP_SRC),star_amd/lib/obj/$(1)/$(basename $(notdir $(f))).o) -o $@
endef

HOST_SRC := $(wildcard star_amd/csrc/host/*.cpp)
HOST_LIB_SRC := $(filter-out star_amd/csrc/host/main.cpp star_amd/csrc/host/cli_run.cpp,$(HOST_SRC))
CLI_SRC := star_amd/csrc/host/main.cpp star_amd/csrc/host/cli_run.cpp
CLI_HDR := include/star_amd_host.h include/star_amd_index.h include/star_amd_cli.h include/star_amd.h
HIP_SRC  := $(wildcard star_amd/csrc/engine/*.hip) $(wildcard star_amd/csrc/index/*.hip)
HIP_HDR  := $(wildcard star_amd/csrc/engine/*.h) $(wildcard star_amd/csrc/index/*.h) include/star_amd.h include/star_amd_index.h

all: host engine shadow cli oracle

host: star_amd/lib/libstaramd_host.so
engine: star_amd/lib/libstaramd.so
cli: star_amd/bin/star_amd star_amd/lib/libstaramd_cli.so
oracle: oracle/_build/liboracle.so oracle/_build/libindex_emul.so oracle/_build/star_amd_oracle_cli oracle/_build/libstaramd_cli_oracle.so oracle/_build/libstaramd_cli_replay.so oracle/_build/libstaramd_emul.so oracle/_build/star_amd_emul_cli

star_amd/lib/libstaramd_host.so: $(HOST_LIB_SRC) star_amd/csrc/host/host.h include/star_amd.h
	@mkdir -p star_amd/lib
	$(CXX) $(CXXFLAGS) -shared $(HOST_LIB_SRC) -o $@ -lz

star_amd/lib/libstaramd.so: $(HIP_SRC) $(HIP_HDR)
	$(call build_engine,prod,)

# shadow-validation build of the engine (tests only): every cooperative stitch / extend call is re-run through the
# scalar restatement on the GPU and disagreements are counted (tests/test_gpu_parity.py::test_shadow_validation)
shadow: star_amd/lib/libstaramd_shadow.so
star_amd/lib/libstaramd_shadow.so: $(HIP_SRC) $(HIP_HDR)
	$(call build_engine,shadow,-DSTARAMD_SHADOW)

star_amd/bin/star_amd: $(CLI_SRC) star_amd/lib/libstaramd_host.so star_amd/lib/libstaramd.so $(CLI_HDR)
	@mkdir -p star_amd/bin
	$(CXX) $(CXXFLAGS) -fPIE $(CLI_SRC) -o $@ -Lstar_amd/lib -lstaramd_host -lstaramd -Wl,-rpath,'$$ORIGIN/../lib'

# the same front end as a library (bench.py and tests run the pipeline in-process through ctypes)
star_amd/lib/libstaramd_cli.so: star_amd/csrc/host/cli_run.cpp star_amd/lib/libstaramd_host.so star_amd/lib/libstaramd.so $(CLI_HDR)
	$(CXX) $(CXXFLAGS) -shared star_amd/csrc/host/cli_run.cpp -o $@ -Lstar_amd/lib -lstaramd_host -lstaramd -Wl,-rpath,'$$ORIGIN'

oracle/_build/liboracle.so: oracle/star_oracle.cpp include/star_amd.h
	@mkdir -p oracle/_build
	$(CXX) $(CXXFLAGS) -shared oracle/star_oracle.cpp -o $@

# test infrastructure: the index-building algorithm (star_amd/csrc/index/index_core.h) on a plain-loop backend, to check its logic without a GPU
oracle/_build/libindex_emul.so: oracle/index_emul.cpp $(wildcard star_amd/csrc/index/*.h)
	@mkdir -p oracle/_build
	$(CXX) $(CXXFLAGS) -fopenmp -shared oracle/index_emul.cpp -o $@

# test infrastructure: the command-line front end with the oracle behind the engine's C ABI (oracle/cli_shim.cpp), for CPU tests of main.cpp
oracle/_build/star_amd_oracle_cli: $(CLI_SRC) oracle/cli_shim.cpp oracle/_build/liboracle.so oracle/_build/libindex_emul.so star_amd/lib/libstaramd_host.so $(CLI_HDR)
	$(CXX) $(CXXFLAGS) -DSTARAMD_NO_RESIDENT_SJDB -fPIE $(CLI_SRC) oracle/cli_shim.cpp -o $@ -Lstar_amd/lib -lstaramd_host -Loracle/_build -loracle -lindex_emul -Wl,-rpath,'$$ORIGIN/../../star_amd/lib' -Wl,-rpath,'$$ORIGIN'

# the same front end as a shared library (the in-process pipeline of bench.py / capi.run_cli) with the oracle behind the engine ABI: CPU tests of
# the multi-rank hooks (tests/test_multi_rank_cpu.py)
oracle/_build/libstaramd_cli_oracle.so: star_amd/csrc/host/cli_run.cpp oracle/cli_shim.cpp oracle/_build/liboracle.so oracle/_build/libindex_emul.so star_amd/lib/libstaramd_host.so $(CLI_HDR)
	$(CXX) $(CXXFLAGS) -DSTARAMD_NO_RESIDENT_SJDB -shared star_amd/csrc/host/cli_run.cpp oracle/cli_shim.cpp -o $@ -Lstar_amd/lib -lstaramd_host -Loracle/_build -loracle -lindex_emul -Wl,-rpath,'$$ORIGIN/../../star_amd/lib' -Wl,-rpath,'$$ORIGIN'

# measurement infrastructure for the HOST stages on a box without a GPU: the front end over an engine stand-in that records a batch's results through the
# oracle once and plays them back afterwards (oracle/replay_shim.cpp, tools/host_bench.py)
oracle/_build/libstaramd_cli_replay.so: star_amd/csrc/host/cli_run.cpp oracle/replay_shim.cpp oracle/_build/liboracle.so star_amd/lib/libstaramd_host.so $(CLI_HDR)
	$(CXX) $(CXXFLAGS) -DSTARAMD_NO_RESIDENT_SJDB -shared star_amd/csrc/host/cli_run.cpp oracle/replay_shim.cpp -o $@ -Lstar_amd/lib -lstaramd_host -Loracle/_build -loracle -Wl,-rpath,'$$ORIGIN/../../star_amd/lib' -Wl,-rpath,'$$ORIGIN'

ref:
	$(MAKE) -f oracle/Makefile.ref -j8 all

clean:
	rm -rf star_amd/lib star_amd/bin oracle/_build

# profiling build: shader-clock time per section of the stitch walk (bench.py --profile-sections)
profile-lib: star_amd/lib/libstaramd_profile.so
star_amd/lib/libstaramd_profile.so: $(HIP_SRC) $(HIP_HDR)
	$(call build_engine,profile,-DSTARAMD_PROFILE)

# the engine's kernel sources compiled for the host by the wavefront emulator (oracle/wave_emul/emu.h): CPU tests of the kernel logic
oracle/_build/libstaramd_emul.so: $(HIP_SRC) $(HIP_HDR) $(wildcard oracle/wave_emul/*.cpp oracle/wave_emul/*.h oracle/wave_emul/hip/*.h oracle/wave_emul/rocprim/*.hpp) oracle/wave_emul/build.sh
	bash oracle/wave_emul/build.sh

# the shipped front end (main.cpp + cli_run.cpp, resident junction insertion and all) linked against the EMULATED engine: CPU tests of the whole
# binary, device index build and device junction insertion included (tests/test_wave_emul.py)
oracle/_build/star_amd_emul_cli: $(CLI_SRC) oracle/_build/libstaramd_emul.so star_amd/lib/libstaramd_host.so $(CLI_HDR)
	$(CXX) $(CXXFLAGS) -fPIE $(CLI_SRC) -o $@ -Lstar_amd/lib -lstaramd_host -Loracle/_build -lstaramd_emul -Wl,-rpath,'$$ORIGIN/../../star_amd/lib' -Wl,-rpath,'$$ORIGIN'

.PHONY: all host engine shadow cli oracle ref clean
