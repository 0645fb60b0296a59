# Build the gfx950 kernel library (product) and, for CPU-side kernel debugging only, the wave-simulator build.
HIPCC      ?= /opt/rocm/bin/hipcc
HOSTCXX    ?= /opt/rocm/lib/llvm/bin/clang++
CSRC       := clip-lite_amd/csrc
LIBDIR     := clip-lite_amd/lib
SRCS       := $(wildcard $(CSRC)/*.hip)
HDRS       := $(wildcard $(CSRC)/*.h) include/clite.h
OBJS       := $(patsubst $(CSRC)/%.hip,build/hip/%.o,$(SRCS))
SIMOBJS    := $(patsubst $(CSRC)/%.hip,build/sim/%.o,$(SRCS)) build/sim/wavesim.o
HIPFLAGS   := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wno-unused-value
SIMFLAGS   := -x c++ -std=c++20 -O1 -g -fPIC -pthread -Itests/wavesim -Iinclude -I$(CSRC) -include tests/wavesim/wavesim.h -Wno-unknown-attributes -Wno-unused-value -Wno-psabi -DCLITE_BN_SLOTS=4 -DCLITE_PATCH_WGS=2 -DCLITE_GROUP_KCHUNK=24 -DCLITE_FOLD_ROWS_WGS=3 $(SIM_EXTRA)

.PHONY: hip sim diag variant clean
hip: $(LIBDIR)/libclite_hip.so
# diagnostic build (tools/README.md): kernel-selection knobs from the environment + the round-1 register-staged engine; never loaded by the product
diag: build/diag/libclite_hip_diag.so
sim: tests/wavesim/_build/libclite_sim.so

$(LIBDIR)/libclite_hip.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

build/hip/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# A/B variant of the PRODUCT build (no diagnostic code): make variant VAR_EXTRA="-DCLITE_BN_PREFETCH=0"; load with CLITE_HIP_LIB=build/var/libclite_hip_var.so
VAROBJS    := $(patsubst $(CSRC)/%.hip,build/var/%.o,$(SRCS))
variant: build/var/libclite_hip_var.so
build/var/libclite_hip_var.so: $(VAROBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(VAROBJS)
build/var/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/var
	$(HIPCC) $(HIPFLAGS) $(VAR_EXTRA) -c $< -o $@

DIAGOBJS   := $(patsubst $(CSRC)/%.hip,build/diag/%.o,$(SRCS))
build/diag/libclite_hip_diag.so: $(DIAGOBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(DIAGOBJS)
build/diag/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/diag
	$(HIPCC) $(HIPFLAGS) -DCLITE_DIAG=1 $(DIAG_EXTRA) -c $< -o $@

tests/wavesim/_build/libclite_sim.so: $(SIMOBJS)
	@mkdir -p tests/wavesim/_build
	$(HOSTCXX) -shared -fPIC -pthread -o $@ $(SIMOBJS) $(SIM_EXTRA)

build/sim/%.o: $(CSRC)/%.hip $(HDRS) tests/wavesim/wavesim.h
	@mkdir -p build/sim
	$(HOSTCXX) $(SIMFLAGS) -c $< -o $@

build/sim/wavesim.o: tests/wavesim/wavesim.cpp tests/wavesim/wavesim.h
	@mkdir -p build/sim
	$(HOSTCXX) -std=c++20 -O1 -g -fPIC -pthread -c $< -o $@

clean:
	rm -rf build $(LIBDIR)/*.so tests/wavesim/_build
