# Top-level build: the product library (HIP, gfx950 only), the CPU checkers and the test emulator.
#   make            -> everything
#   make lib        -> hh-suite_amd/lib/libhhviterbi_hip.so   (the C-ABI drop-in, include/hhviterbi_hip.h)
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref/libhhref.so when /root/reference is present)
#   make emul       -> tests/emul/libwave_emul.so             (host lock-step emulation, test only)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the reference build has no FMA; contraction would change low bits.
# -fno-slp-vectorize: keeps hipcc from pairing scalar fp32 ops into v_pk_* + register shuffles.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off $(HHV_EXTRA_HIPFLAGS) -fPIC -fvisibility=hidden -Wall -Wno-unused-result
CSRC     := hh-suite_amd/csrc
LIBDIR   := hh-suite_amd/lib
OBJDIR   := build/obj
LIB      := $(LIBDIR)/libhhviterbi_hip.so
OBJS     := $(OBJDIR)/hhv_kernels.o $(OBJDIR)/hhv_kernels_w32.o $(OBJDIR)/hhv_kernels_w16.o $(OBJDIR)/hhv_kernels_pair.o $(OBJDIR)/hhv_prep.o $(OBJDIR)/hhv_prefilter.o $(OBJDIR)/hhv_mac.o $(OBJDIR)/hhv_topk.o $(OBJDIR)/hhv_api.o $(OBJDIR)/hhv_api_db.o $(OBJDIR)/hhv_api_prep.o $(OBJDIR)/hhv_api_prefilter.o $(OBJDIR)/hhv_api_mac.o $(OBJDIR)/hhv_pack.o
HDRS     := $(wildcard $(CSRC)/*.h) include/hhviterbi_hip.h

RUNNER   := $(LIBDIR)/libhhv_runner.so

all: lib lib_fma lib_pto oracle emul

lib: $(LIB) $(RUNNER)

# OPT-IN build, never the default: the emission score with fused multiply-adds (viterbi_lane.h HHV_EMISSION_FMA) - a second
# library next to the bit-exact one; nothing links it, a caller chooses it by name (bench.py `fast_mode`, tests/test_gpu_fast_mode.py)
lib_fma:
	$(MAKE) $(LIBDIR)/libhhviterbi_hip_fma.so LIB=$(LIBDIR)/libhhviterbi_hip_fma.so OBJDIR=build/obj_fma HHV_EXTRA_HIPFLAGS="$(HHV_EXTRA_HIPFLAGS) -DHHV_EMISSION_FMA"

# measurement builds next to the product library (tools/README.md), e.g.
#   make lib_variant NAME=nq FLAGS=-DHHV_NO_QUEUE        a fixed stream range per wave in every variant (A/B of the work queue)
#   make lib_variant NAME=wt FLAGS=-DHHV_EXP_WAVETIME    per-workgroup entry / exit times (tools/wave_times.py)
lib_variant:
	$(MAKE) $(LIBDIR)/libhhviterbi_$(NAME).so LIB=$(LIBDIR)/libhhviterbi_$(NAME).so OBJDIR=build/obj_$(NAME) HHV_EXTRA_HIPFLAGS="$(HHV_EXTRA_HIPFLAGS) $(FLAGS)"

# TEST build (tests/test_gpu_errors.py): the product objects with ONE unit replaced - the pair kernels compiled with
# -DHHV_EXP_PAIR_TIMEOUT (the first wave of a two-wave workgroup never reports progress, short spin bound), so that the second
# wave's bounded wait runs out: the launch must end with HHV_E_DEVICE, not with a result
lib_pto: $(LIBDIR)/libhhviterbi_hip_pto.so
build/obj_pto/hhv_kernels_pair.o: $(CSRC)/hhv_kernels_pair.hip $(HDRS)
	@mkdir -p build/obj_pto
	$(HIPCC) $(HIPFLAGS) -DHHV_EXP_PAIR_TIMEOUT -fno-slp-vectorize -c $< -o $@
# ... and the MAC dataflow kernels with -DHHV_EXP_MAC_TIMEOUT (the first parallel-part wave never posts its units)
build/obj_pto/hhv_mac.o: $(CSRC)/hhv_mac.hip $(HDRS)
	@mkdir -p build/obj_pto
	$(HIPCC) $(HIPFLAGS) -DHHV_EXP_MAC_TIMEOUT -c $< -o $@
$(LIBDIR)/libhhviterbi_hip_pto.so: $(OBJS) build/obj_pto/hhv_kernels_pair.o build/obj_pto/hhv_mac.o
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter-out $(OBJDIR)/hhv_kernels_pair.o $(OBJDIR)/hhv_mac.o,$(OBJS)) build/obj_pto/hhv_kernels_pair.o build/obj_pto/hhv_mac.o

# C++ host layer above the C ABI (mirror of the reference's ViterbiRunner); plain g++, links only the C ABI
$(RUNNER): hh-suite_amd/host/viterbi_runner.cpp hh-suite_amd/host/viterbi_runner.h hh-suite_amd/host/prefilter.cpp hh-suite_amd/host/prefilter.h hh-suite_amd/host/posterior_decoder.cpp hh-suite_amd/host/posterior_decoder.h include/hhviterbi_hip.h $(LIB)
	g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared -Wall -Iinclude -o $@ hh-suite_amd/host/viterbi_runner.cpp hh-suite_amd/host/prefilter.cpp hh-suite_amd/host/posterior_decoder.cpp -L$(LIBDIR) -lhhviterbi_hip -lpthread -Wl,-rpath,'$$ORIGIN'

$(OBJDIR)/hhv_kernels.o: $(CSRC)/hhv_kernels.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -c $< -o $@
$(OBJDIR)/hhv_kernels_w%.o: $(CSRC)/hhv_kernels_w%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -c $< -o $@
$(OBJDIR)/hhv_kernels_pair.o: $(CSRC)/hhv_kernels_pair.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -c $< -o $@
$(OBJDIR)/hhv_prep.o: $(CSRC)/hhv_prep.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -c $< -o $@
$(OBJDIR)/hhv_prefilter.o: $(CSRC)/hhv_prefilter.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/hhv_mac.o: $(CSRC)/hhv_mac.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/hhv_topk.o: $(CSRC)/hhv_topk.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/hhv_api.o: $(CSRC)/hhv_api.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(OBJDIR)/hhv_api_%.o: $(CSRC)/hhv_api_%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(OBJDIR)/hhv_pack.o: $(CSRC)/hhv_pack.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle all

emul: tests/emul/libwave_emul.so
tests/emul/libwave_emul.so: tests/emul/wave_emul.cpp $(CSRC)/viterbi_lane.h
	g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -o $@ $<

clean:
	rm -rf build $(LIBDIR) tests/emul/libwave_emul.so
	$(MAKE) -C oracle clean

PHONY_EXTRA := rccl_runner
.PHONY: topk_fuzz example example_rccl rccl_runner all lib lib_fma lib_pto lib_variant oracle emul clean

# plain-C++ use of the host classes (no Python): examples/search_example.cpp
example: $(RUNNER)
	g++ -O2 -std=c++14 -Wall -Iinclude -o build/search_example examples/search_example.cpp -L$(LIBDIR) -lhhv_runner -lhhviterbi_hip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

# hhv::RcclShardedRunner (hh-suite_amd/host/rccl_runner.h): the sharded search for multi-process hosts, one process per GPU - the C
# ABI + librccl (a library of its own: the C-ABI library itself does not depend on RCCL)
RCCL_RUNNER := $(LIBDIR)/libhhv_rccl_runner.so
rccl_runner: $(RCCL_RUNNER)
$(RCCL_RUNNER): hh-suite_amd/host/rccl_runner.cpp hh-suite_amd/host/rccl_runner.h hh-suite_amd/host/viterbi_runner.h include/hhviterbi_hip.h $(LIB)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wall -fPIC -shared -Iinclude -o $@ hh-suite_amd/host/rccl_runner.cpp -L$(LIBDIR) -lhhviterbi_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

# the sharded search as a native multi-process program on top of it (examples/sharded_search_rccl.cpp)
example_rccl: build/sharded_search_rccl
build/sharded_search_rccl: examples/sharded_search_rccl.cpp examples/synth8d.h include/hhviterbi_hip.h $(RCCL_RUNNER)
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wall -Iinclude -o build/sharded_search_rccl examples/sharded_search_rccl.cpp -L$(LIBDIR) -lhhv_rccl_runner -lhhviterbi_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

# TEST program: the one-launch top-K / merge kernels of hhv_topk.hip on their own - timing of the phases and a fuzzer against
# std::sort (tools/topk_ubench.hip, tests/test_gpu_topk_fuzz.py); includes the kernel source, links nothing of the product
topk_fuzz: build/topk_ubench
build/topk_ubench: tools/topk_ubench.hip $(CSRC)/hhv_topk.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -DHHV_TOPK_TIMING -Wno-unused-value -I$(CSRC) -Iinclude -o $@ tools/topk_ubench.hip
