# Builds libfsgm_hip.so (hand-written HIP for gfx950 + the C ABI of include/fsgm.h) in-tree,
# and the oracle (test infrastructure) under oracle/.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the fp64 geometry / parabola must be evaluated exactly as the reference
# writes it (no FMA contraction) for bit-exact parity.
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-function -Iinclude
CSRC     := fsgm_amd/csrc
SRCS     := $(wildcard $(CSRC)/*.hip)
OBJS     := $(SRCS:.hip=.o)
LIB      := fsgm_amd/libfsgm_hip.so

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/fsgm.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

# micro-benchmarks behind DESIGN.md's rates (run on the GPU box; binaries are not tracked)
ubench:
	for f in tools/ubench/*.hip; do $(HIPCC) -O3 --offload-arch=$(ARCH) -Wno-unused-value -o $${f%.hip} $$f; done

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean ubench
